import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from dspnet_amd import synthetic
from dspnet_amd.symbol.multitask_symbol_factory import get_config
from oracle import dspnet_torch as ot
threads = int(sys.argv[1]); size = int(sys.argv[2]); images = int(sys.argv[3])
torch.set_num_threads(threads)
cfg = get_config("resnet-50", size)
# random params with the right shapes: build from a tiny fake by running export on CPU is impossible (needs GPU) -> synthesize
import math
def mk():
    vals = {}
    rng = np.random.default_rng(0)
    def conv(name, co, ci, k): vals[name + "_weight"] = (rng.standard_normal((co, ci, k, k)) / math.sqrt(ci * k * k)).astype(np.float32)
    def bnp(name, c, gamma=True):
        if gamma: vals[name + "_gamma"] = np.ones(c, np.float32)
        vals[name + "_beta"] = np.zeros(c, np.float32)
    bnp("bn_data", 3, False); conv("conv0", 64, 3, 7); bnp("bn0", 64)
    fl = [64, 256, 512, 1024, 2048]; cin = 64
    for i, n in enumerate([3, 4, 6, 3]):
        for j in range(n):
            nm = "stage%d_unit%d" % (i + 1, j + 1); nf = fl[i + 1]; q = nf // 4
            bnp(nm + "_bn1", cin); conv(nm + "_conv1", q, cin, 1); bnp(nm + "_bn2", q); conv(nm + "_conv2", q, q, 3)
            bnp(nm + "_bn3", q); conv(nm + "_conv3", nf, q, 1)
            if j == 0: conv(nm + "_sc", nf, cin, 1)
            cin = nf
    prev = 2048
    for k, nf in zip((2, 3, 4, 5), (512, 256, 256, 128)):
        n1 = max(128, nf // 2)
        conv("multi_feat_%d_conv_1x1_conv" % k, n1, prev, 1); vals["multi_feat_%d_conv_1x1_conv_bias" % k] = np.zeros(n1, np.float32)
        conv("multi_feat_%d_conv_3x3_conv" % k, nf, n1, 3); vals["multi_feat_%d_conv_3x3_conv_bias" % k] = np.zeros(nf, np.float32)
        prev = nf
    names = ["_plus12", "_plus15"] + ["multi_feat_%d_conv_3x3_relu" % k for k in (2, 3, 4, 5)]
    chans = [1024, 2048, 512, 256, 256, 128]; A = [4, 6, 6, 6, 4, 4]
    for nm, c, a in zip(names, chans, A):
        conv(nm + "_loc_pred_conv", a * 5, c, 3); vals[nm + "_loc_pred_conv_bias"] = np.zeros(a * 5, np.float32)
        conv(nm + "_cls_pred_conv", a * 9, c, 3); vals[nm + "_cls_pred_conv_bias"] = np.zeros(a * 9, np.float32)
    for nm, co, ci, k in [("res3_reduced", 128, 512, 1), ("res3_reduced2", 128, 128, 3), ("res4_reduced", 256, 1024, 1), ("res4_reduced2", 256, 256, 3),
                          ("score2_pool4", 128, 2048, 1), ("score2_pool2", 256, 2048, 1), ("score2_pool1", 512, 2048, 1), ("score3_conv", 19, 3328, 3)]:
        conv(nm, co, ci, k); vals[nm + "_bn_beta"] = np.zeros(co, np.float32)
    vals["res5_reduced_bn_beta"] = np.zeros(2048, np.float32)
    vals["score4_conv_weight"] = (rng.standard_normal((19, 19, 4, 4)) * 0.1).astype(np.float32)
    return vals
vals = mk()
gen = synthetic.rng(233)
data = synthetic.images(images, size, size, gen); lab = synthetic.det_labels(images, gen=gen, height=size, width=size, first_empty=False); seg = synthetic.seg_labels(images, size, size, gen=gen)
for it in range(2):
    t0 = time.perf_counter()
    ref = ot.forward_loss(vals, data, lab, seg, cfg["sizes"][1:], cfg["ratios"][1:], dtype=torch.float32)
    t1 = time.perf_counter()
    ref["objective"].backward()
    t2 = time.perf_counter()
    print("threads", threads, "iter", it, "fwd %.2fs bwd %.2fs" % (t1 - t0, t2 - t1), flush=True)
