"""Golden vectors from the reference's own code (run in the build container; /root/reference does not exist on the GPU box):
  * `get_config` of symbol/multitask_symbol_factory.py -- the FunctionDef compiled from the file where it lies (the module
    itself imports mxnet through the symbol builder), evaluated for every preset the reference defines;
  * the segmentation look-up table of dataset/iterator.py:358-363, built from dataset/cs_labels.py (a plain-python module,
    executed from its file) by the loop the iterator runs.
Writes tests/golden/factory_config.json (inputs and returned values only, no source text)."""
import ast
import importlib.util
import json
import logging
import os

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_function(path, name):
    tree = ast.parse(open(path).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"logging": logging}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    return ns[name]


def main():
    get_config = reference_function(os.path.join(REF, "symbol", "multitask_symbol_factory.py"), "get_config")
    cases = []
    for network in ("vgg16_reduced", "inceptionv3", "resnet-50", "resnet101", "resnet-101", "resnet50", "mobilenet"):
        for shape in (300, 320, 447, 448, 512, 1024):
            try:
                cfg = dict(get_config(network, shape))
                cfg.pop("kwargs", None)
                cases.append({"network": network, "data_shape": shape, "config": cfg})
            except NotImplementedError as e:
                cases.append({"network": network, "data_shape": shape, "error": "NotImplementedError"})
    spec = importlib.util.spec_from_file_location("ref_cs_labels", os.path.join(REF, "dataset", "cs_labels.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lut = np.ones(256) * 255                      # dataset/iterator.py:359-362, verbatim semantics
    for l in mod.labels:
        if l.trainId >= 0:
            lut[l.id] = l.id
    out = {"get_config": cases, "seg_lut": [int(v) for v in lut],
           "labels": [[l.name, int(l.id), int(l.trainId)] for l in mod.labels]}
    with open(os.path.join(HERE, "factory_config.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", len(cases), "get_config cases,", sum(1 for c in cases if "config" in c), "of them presets")


if __name__ == "__main__":
    main()
