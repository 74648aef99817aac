"""Parameter set of a multi-task network started from a pretrained backbone: the rules of multi_init.py:50-169
(`init_from_resnet`, called from multi_train.py:348-354 after `mx.model.load_checkpoint(pretrained, epoch)`).

The reference keeps every pretrained array, adds `affine_matrix = [[1,0,0,0,1,0]]` (:72) and then, for the layers the
multi-task graph adds on top of the backbone -- the segmentation decoder (`score*`, `res{3,4,5}_reduced*`,
`res{3,4}_bn`), the SSD heads (`_plusN_{cls,loc}_pred_conv`) and the SSD extras (`multi_feat_*`), listed by name
(:74-158) -- assigns
    *_weight                    U(-1/sqrt(max(shape)), +1/sqrt(max(shape)))   on the MXNet shape        (:74-76)
    *_bias, *_bn_beta, *_bn_bias  zeros                                                                  (:107-152)
    *_bn_gamma                   ones                                                                    (:152-158)
    Deconvolution weights (`bigscore_weight`, `score2_weight`, `score4_conv_weight`)
                                 the diagonal bilinear kernel of `upsample_filt`                         (:160-168)
A graph argument that is neither pretrained nor covered by these rules has no value and the executor bind of the
reference fails; `init_from_resnet` here raises KeyError for it.  The name lists are expressed as patterns over the
same prefixes.  Random draws come from numpy's PCG64 (MXNet's generator is not reproducible here)."""
import math
import re

import numpy as np

# layers the multi-task graph adds to the backbone (prefixes of the names listed in multi_init.py:74-158)
_ADDED = re.compile(r"^(score\d*_|score_|bigscore_|res[345]_reduced\d*_|res[34]_bn_|_plus\d+_(cls|loc)_pred_conv_|multi_feat_\d+_)")
_DECONV = ("bigscore_weight", "score2_weight", "score4_conv_weight")


def upsample_filt(size):
    """bilinear interpolation kernel (size, size) (multi_init.py:13-21)"""
    factor = (size + 1) // 2
    center = factor - 1.0 if size % 2 == 1 else factor - 0.5
    og = np.ogrid[:size, :size]
    return (1 - abs(og[0] - center) / factor) * (1 - abs(og[1] - center) / factor)


def is_added_layer(name):
    return _ADDED.match(name) is not None


def init_from_resnet(net, resnet_args, resnet_auxs, seed=0):
    """-> (arg_params, aux_params) in the reference's shapes, ready for `net.g.set_params(arg_params)`.
    net: MultiTaskNet (or Graph); resnet_args / resnet_auxs: dicts from model.load_checkpoint."""
    g = getattr(net, "g", net)
    rng = np.random.Generator(np.random.PCG64(seed))
    args = dict(resnet_args)
    auxs = dict(resnet_auxs)
    args["affine_matrix"] = np.array([[1, 0, 0, 0, 1, 0]], np.float32)
    for p in g.param_order:
        shape = tuple(p.logical or p.shape)
        if is_added_layer(p.name):
            if p.name in _DECONV and p.kind == "deconv":
                w = np.zeros(shape, np.float32)
                filt = upsample_filt(shape[3])
                n = min(shape[0], shape[1])
                w[range(n), range(n), :, :] = filt
                args[p.name] = w
            elif p.name.endswith("_weight"):
                lim = 1.0 / math.sqrt(max(shape))
                args[p.name] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
            elif p.name.endswith(("_bias", "_beta")):
                args[p.name] = np.zeros(shape, np.float32)
            elif p.name.endswith("_gamma"):
                args[p.name] = np.ones(shape, np.float32)
        elif p.name not in args:
            raise KeyError("init_from_resnet: %s is neither in the pretrained model nor a layer multi_init.py initialises"
                           % p.name)
    return args, auxs
