"""Writes the gradient-arena layout of the REAL headline graph (resnet-50 multitask, 512x512) for the CPU gloo test of
the bucketed all-reduce (tests/test_dp_gloo.py): [(parameter name, offset, padded size, index of the graph node that owns
it)] in arena order, the arena length and the number of graph nodes.  Building the graph needs the GPU (MultiBoxPrior
generates the anchors at construction), so this runs on the MI355X box:

    gpurun -- python tests/golden/make_bucket_layout_golden.py gpurun_out/bucket_layout_resnet50_512.json

and the file is then copied to tests/golden/.  The layout does not depend on the batch size."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main(out):
    import torch
    from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
    from dspnet_amd.train.solver import MultiTaskSolver
    net = get_multi_symbol_train("resnet-50", 512, num_classes=8, batch_size=1, device=torch.device("cuda", 0))
    solver = MultiTaskSolver(net)
    g = net.g
    owner = {}
    for idx, n in enumerate(g.nodes):
        for v in vars(n).values():
            if hasattr(v, "offset") and hasattr(v, "wd_mult"):
                owner[v.name] = min(owner.get(v.name, idx), idx)
    rows = [[p.name, int(p.offset), int((p.size + 3) // 4 * 4), int(owner[p.name])] for p in g.param_order]
    doc = {"graph": "resnet-50 multitask 512x512, 8 det classes", "arena": int(g.arena.numel()), "nodes": len(g.nodes),
           "params": rows, "buckets_16mb": [[int(lo), int(hi), int(first)] for lo, hi, first in solver.buckets]}
    json.dump(doc, open(out, "w"))
    print("wrote", out, len(rows), "parameters,", doc["arena"], "floats,", len(doc["buckets_16mb"]), "buckets")


if __name__ == "__main__":
    main(sys.argv[1])
