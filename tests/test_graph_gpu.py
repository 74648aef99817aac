"""Graph-level parity of the multi-task training step on the GPU against the CPU restatement
(oracle/dspnet_torch.py, float64): the five outputs, the loss readouts (BASELINE.json: fp32 losses
within 1e-4 relative) and every parameter gradient; plus the recorded shapes of utils.py:38."""
import numpy as np
import pytest
import torch

import mbx_cases as mc
from dspnet_amd import synthetic
from dspnet_amd.symbol.multitask_symbol_factory import get_config, get_multi_symbol_train
from dspnet_amd.train.metric import MultiBoxMetric
from dspnet_amd.train.solver import MultiTaskSolver
from oracle import dspnet_torch as ot
from oracle import multibox as om

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["bf16x3", "fp32", "f16x2"])
def conv_math(request):
    """both fp32-result math modes of the float-tensor convolutions (functional.set_conv_math), same tolerances"""
    from dspnet_amd import functional as fn
    fn.set_conv_math(request.param)
    yield request.param
    fn.set_conv_math(fn.DEFAULT_CONV_MATH)


def make(batch, h, w, seed=233):
    dev = torch.device("cuda", 0)
    net = get_multi_symbol_train("resnet-50", (3, h, w), num_classes=8, batch_size=batch, device=dev, seed=1)
    gen = synthetic.rng(seed)
    data = synthetic.images(batch, h, w, gen)
    lab = synthetic.det_labels(batch, gen=gen, height=h, width=w)
    seg = synthetic.seg_labels(batch, h, w, gen=gen)
    solver = MultiTaskSolver(net)
    solver.set_batch(torch.from_numpy(data).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(seg).to(dev))
    return net, solver, data, lab, seg


def bf16_parity_report(net, data, lab, seg, cfg, num_classes=8, legacy=None):
    """Device (bf16 MFMA convolutions) against the oracle evaluating THE SAME ARITHMETIC (conv_quant="bf16": both operands
    of every convolution GEMM rounded to bfloat16, exact accumulation) in float64 -- and, as the yardstick, the same
    oracle evaluated in float32 against its float64 self.  Both comparisons see the same two effects: float32
    accumulation order, and activations computed in float32 landing on the other side of a bf16 rounding boundary
    than their float64 counterparts (one 2^-9 step on a small fraction of elements per layer, then amplified by the
    batch-statistics BatchNorm stack like any other perturbation).  -> (dev, ref32): dicts of relative errors."""
    dev_targets = [net.target.loc_target.cpu().numpy(), net.target.loc_mask.cpu().numpy(),
                   net.target.cls_target.cpu().numpy()]
    values = ot.export_params(net.g)
    kw = dict(num_classes=num_classes, targets=dev_targets, conv_quant="bf16")
    if legacy is not None:
        args = (values, data, lab, seg) + tuple(legacy)
    else:
        args, kw = (values, data, lab, seg), dict(kw, config=cfg)
    r64 = ot.forward_loss(*args, dtype=torch.float64, **kw)
    r32 = ot.forward_loss(*args, dtype=torch.float32, **kw)
    with ot.quantized("bf16"):
        r64["objective"].backward()
        r32["objective"].backward()

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

    outs = [o.cpu().numpy() for o in net.outputs()]
    m = MultiBoxMetric(); m.update(net)
    losses = dict(zip(*m.get()))
    dev, c32 = {}, {}
    for key, dv in (("loc_preds", net.loc_preds.data.cpu().numpy()), ("cls_prob", outs[0]), ("seg_out", outs[4])):
        dev[key], c32[key] = rel(dv, r64[key].numpy()), rel(r32[key].numpy(), r64[key].numpy())
    for n, v in losses.items():
        if n in r64:
            dev["loss:" + n], c32["loss:" + n] = abs(v - r64[n]) / abs(r64[n]), abs(r32[n] - r64[n]) / abs(r64[n])
    nd = dd = n3 = 0.0
    hd = h3 = hh = 0.0
    for p in net.g.param_order:
        if p.name == "affine_matrix":          # identity grid: on the interpolation kinks (see the fp32 resnet-50 test)
            continue
        g64 = ot.import_grad(p.name, r64["params"][p.name].grad).astype(np.float64)
        g32 = ot.import_grad(p.name, r32["params"][p.name].grad).astype(np.float64)
        gdev = p.grad.cpu().numpy().astype(np.float64)
        gdev = gdev[:g64.shape[0], :, :, :g64.shape[3]] if gdev.ndim == 4 else gdev[:g64.shape[0]]
        e_d, e_3, sq = float(((gdev - g64) ** 2).sum()), float(((g32 - g64) ** 2).sum()), float((g64 ** 2).sum())
        nd += e_d; n3 += e_3; dd += sq
        if p.name.endswith("pred_conv_weight"):
            hd += e_d; h3 += e_3; hh += sq
    dev["grad_L2:all"], c32["grad_L2:all"] = (nd / dd) ** 0.5, (n3 / dd) ** 0.5
    dev["grad_L2:heads"], c32["grad_L2:heads"] = (hd / hh) ** 0.5, (h3 / hh) ** 0.5
    return dev, c32


def assert_within_yardstick(dev, c32, factor=4.0, floor=2e-4, loss_factor=8.0):
    """the device may be `factor` x as far from the float64 evaluation as the CPU's own float32 evaluation of the same
    bf16-operand arithmetic is (floor: the plain fp32-accumulation level where that yardstick happens to be tiny).
    Tensors and gradient norms are maxima / sums over 1e5 .. 1e7 elements and make stable yardsticks.  The three loss
    read-outs are single scalars: the deviation of a mean of a few hundred perturbed terms has a random sign and can
    come out 10x below its typical size by luck (CrossEntropy 1.5e-4 next to SegCrossEntropy 4e-3 on the same run), so
    each loss is judged against the LARGEST of the three loss yardsticks, with twice the factor."""
    pooled = max([v for k, v in c32.items() if k.startswith("loss:")] or [0.0])
    for k in dev:
        if k.startswith("loss:"):
            assert dev[k] <= max(loss_factor * pooled, floor), (k, dev[k], c32[k], pooled)
        else:
            assert dev[k] <= max(factor * c32[k], floor), (k, dev[k], c32[k])


def test_recorded_shapes_512x1024(gpu_device):
    """utils.py:38 internal_out_shapes_512 (1x3x512x1024, 10 det classes)"""
    net = get_multi_symbol_train("resnet-50", (3, 512, 1024), num_classes=10, batch_size=1)
    t = net.g.tensors
    nchw = lambda s: (s[0], s[3], s[1], s[2])  # noqa: E731
    assert nchw(t["_plus6"].shape) == (1, 512, 64, 128)
    assert nchw(t["_plus12"].shape) == (1, 1024, 32, 64)
    assert nchw(t["_plus15"].shape) == (1, 2048, 16, 32)
    for k, shp in zip((2, 3, 4, 5), ((512, 8, 16), (256, 4, 8), (256, 2, 4), (128, 1, 2))):
        assert nchw(t["multi_feat_%d_conv_3x3_conv_out" % k].shape)[1:] == shp
    assert tuple(net.anchors.shape) == (1, 12264, 4)
    assert net.loc_preds.shape == (1, 61320)
    net.g.forward()
    outs = net.outputs()
    assert tuple(outs[0].shape) == (1, 11, 12264)          # cls_prob
    assert tuple(outs[3].shape) == (1, 12264, 7)           # det_out_output
    assert tuple(outs[4].shape) == (1, 19, 128, 256)       # seg_out_output


def test_forward_backward_matches_cpu_restatement(gpu_device, conv_math):
    net, solver, data, lab, seg = make(2, 256, 256)
    solver.forward()
    solver.backward()
    torch.cuda.synchronize()
    cfg = get_config("resnet-50", 256)
    anchors = net.anchors.cpu().numpy()
    cls_preds_dev = net.target.cls_preds.data.cpu().numpy()
    dev_targets = [net.target.loc_target.cpu().numpy(), net.target.loc_mask.cpu().numpy(),
                   net.target.cls_target.cpu().numpy()]
    # operators inside the graph: bit-exact against the C oracle on the device's own inputs
    np.testing.assert_array_equal(anchors, mc.r50_anchors(256, 256))
    mc.assert_target_equal(dev_targets, om.multibox_target(anchors, lab, cls_preds_dev, negative_mining_ratio=3))
    outs = [o.cpu().numpy() for o in net.outputs()]
    np.testing.assert_array_equal(outs[3], om.multibox_detection(outs[0], net.loc_preds.data.cpu().numpy(), anchors,
                                                                 nms_threshold=.5, nms_topk=400))
    # whole graph in float64 on the CPU, same parameters, matching pinned to the device's
    ref = ot.forward_loss(ot.export_params(net.g), data, lab, seg, cfg["sizes"][1:], cfg["ratios"][1:],
                          dtype=torch.float64, targets=dev_targets)
    np.testing.assert_array_equal(ref["anchors"], anchors)

    def rel(a, b):
        return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

    assert rel(cls_preds_dev, ref["cls_preds"].numpy()) < 1e-4
    assert rel(net.loc_preds.data.cpu().numpy(), ref["loc_preds"].numpy()) < 1e-4
    assert rel(outs[0], ref["cls_prob"].numpy()) < 1e-4
    assert rel(outs[1], ref["loc_loss"].numpy()) < 1e-4
    assert rel(outs[4], ref["seg_out"].numpy()) < 1e-4
    m = MultiBoxMetric(); m.update(net)
    names, vals = m.get()
    for n, v in zip(names, vals):
        assert abs(v - ref[n]) <= 1e-4 * abs(ref[n]), (n, v, ref[n])
    # gradients of every parameter (before the 1/batch rescale of the optimizer)
    ref["objective"].backward()

    def l2(a, b):
        return float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-30))

    emax, el2, num, den = {}, {}, 0.0, 0.0
    for p in net.g.param_order:
        gref = ref["params"][p.name].grad
        assert gref is not None, p.name
        gref = ot.import_grad(p.name, gref)
        gdev = p.grad.cpu().numpy()
        if gdev.ndim == 4:
            gdev = gdev[:gref.shape[0], :, :, :gref.shape[3]]
        else:
            gdev = gdev[:gref.shape[0]]
        emax[p.name], el2[p.name] = rel(gdev, gref), l2(gdev, gref)
        if p.name != "affine_matrix":
            num += float(((gdev - gref) ** 2).sum()); den += float((gref ** 2).sum())
    print("largest gradient L2 errs", sorted(el2.items(), key=lambda kv: -kv[1])[:5], "global", (num / den) ** 0.5)
    # Parameters with no ReLU between them and the losses (the SSD head convs and the whole ReLU-free seg
    # decoder) must agree element-wise.  Everywhere else an fp32 and an fp64 forward disagree on the sign of
    # ~1e-4 of the pre-activations that sit within rounding of zero; each such ReLU flip changes individual
    # gradient entries by O(1), so those tensors are held to an L2 bound.
    # (affine_matrix is not in this list: at the identity grid every target pixel of a pyramid level that already has the
    # target size samples EXACTLY on a source pixel, where bilinear interpolation has a kink -- left and right derivative
    # differ, and which one a float evaluation of (x_t + 1) * (W - 1) / 2 lands on is a matter of its last bit, in MXNet
    # as much as here or in torch.  Its gradient is compared where it is defined: after the grid has moved,
    # test_second_step_with_moved_affine_matrix_matches_cpu_restatement, and per operator in tests/test_nn_gpu.py.)
    top = [n for n in emax if n.startswith(("score", "res3_", "res4_", "res5_")) or "_pred_conv_" in n]
    assert len(top) > 35 and "affine_matrix" in emax and np.isfinite(emax["affine_matrix"])
    for name in top:
        assert emax[name] < 1e-3, (name, emax[name])
    for name, e in el2.items():
        assert e < 8e-2 or name == "affine_matrix", (name, e)
    assert (num / den) ** 0.5 < 2e-2


def device_decisions(net):
    """the discrete choices of the device's forward pass, in the oracle's format (oracle/dspnet_torch.py _DECISIONS): ReLU
    masks of every BatchNorm+ReLU (evaluated by the same fmaf the convolution loaders use: dspn_bn_apply_f32) and of
    every Convolution+ReLU, and the window positions every max-pool recorded"""
    from dspnet_amd import engine as E
    from dspnet_amd import functional as fn
    dec, pools = {}, 0
    for n in net.g.nodes:
        if isinstance(n, E.BatchNorm) and n.relu:
            c = n.x.channels or n.x.shape[-1]
            y = fn.bn_apply(n.x.data, n.scale, n.shift, relu=True)
            dec["relu:" + n.beta.name[:-len("_beta")]] = (y > 0).permute(0, 3, 1, 2)[:, :c].cpu()
        elif isinstance(n, E.Conv) and n.relu:
            dec["relu:" + n.w.name[:-len("_weight")]] = (n.out.data > 0).permute(0, 3, 1, 2)[:, :n.cout].cpu()
        elif isinstance(n, E.MaxPool):
            assert n.argmax is not None
            c = n.x.channels or n.x.shape[-1]
            dec["pool:%d" % pools] = n.argmax.permute(0, 3, 1, 2)[:, :c].cpu().long()
            pools += 1
    return dec


@pytest.mark.parametrize("network,batch,size", [("resnet-50", 2, 512), ("vgg16_reduced", 2, 512), ("inceptionv3", 1, 512)])
def test_full_size_gradients_elementwise_with_pinned_decisions(gpu_device, conv_math, network, batch, size):
    """BASELINE.json's shape (512x512) instead of a reduced one, and an ELEMENT-WISE bound on every parameter gradient,
    backbone included: the float64 restatement is handed the device's own discrete decisions (ReLU signs, max-pool
    picks, MultiBoxTarget matching), so both differentiate the same piecewise-linear function and what is left is fp32
    rounding.  Bounds: outputs and losses 1e-4 (BASELINE.json), every gradient tensor 1e-3 of its largest entry
    (affine_matrix excepted: identity grid, on the interpolation kinks).  Round 3: inceptionv3 too (its 94 Conv -> BN -> ReLU
    blocks and five max-pools pinned the same way)."""
    dev = torch.device("cuda", 0)
    net = get_multi_symbol_train(network, (3, size, size), num_classes=8, batch_size=batch, device=dev, seed=1)
    gen = synthetic.rng(321)
    data = synthetic.images(batch, size, size, gen)
    lab = synthetic.det_labels(batch, gen=gen, height=size, width=size, first_empty=False)
    seg = synthetic.seg_labels(batch, size, size, gen=gen)
    solver = MultiTaskSolver(net)
    solver.set_batch(torch.from_numpy(data).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(seg).to(dev))
    solver.forward(); solver.backward(); torch.cuda.synchronize()
    dec = device_decisions(net)
    assert sum(k.startswith("relu:") for k in dec) >= (49 if network == "resnet-50" else 15)
    cfg = get_config(network, size)
    dev_targets = [net.target.loc_target.cpu().numpy(), net.target.loc_mask.cpu().numpy(),
                   net.target.cls_target.cpu().numpy()]
    ref = ot.forward_loss(ot.export_params(net.g), data, lab, seg, num_classes=8, dtype=torch.float64,
                          targets=dev_targets, config=cfg, decisions=dec)

    def rel(a, b):
        return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

    outs = [o.cpu().numpy() for o in net.outputs()]
    # (inceptionv3: 94 batch-statistics BatchNorms over small maps; the float32 CPU restatement itself is 2e-4 .. 9e-4 from
    # its float64 run on forward tensors, see test_other_backbone_graphs_match_cpu_restatement -- tensors get 2e-3 there,
    # the loss read-outs and every gradient the same bounds as the other backbones)
    ttol = 2e-3 if network == "inceptionv3" else 1e-4
    assert rel(net.loc_preds.data.cpu().numpy(), ref["loc_preds"].numpy()) < ttol
    assert rel(outs[0], ref["cls_prob"].numpy()) < ttol
    assert rel(outs[4], ref["seg_out"].numpy()) < ttol
    m = MultiBoxMetric(); m.update(net)
    for n, v in zip(*m.get()):
        assert abs(v - ref[n]) <= 1e-4 * abs(ref[n]), (n, v, ref[n])
    ref["objective"].backward()
    grads = {}
    for p in net.g.param_order:
        if p.name == "affine_matrix":
            continue
        gref = ot.import_grad(p.name, ref["params"][p.name].grad)
        gdev = p.grad.cpu().numpy()
        gdev = gdev[:gref.shape[0], :, :, :gref.shape[3]] if gdev.ndim == 4 else gdev[:gref.shape[0]]
        grads[p.name] = (gdev, gref)
    # A gradient that is ZERO in exact arithmetic is rounding noise on both sides and has no relative error: bn0_gamma in
    # the resnet graph (bn0_beta = 0 at initialisation, so ReLU and max-pool commute with the per-channel scale and the
    # BatchNorm of stage1_unit1 removes it again).  Absolute floor: 1e-6 of the largest gradient entry of the whole model.
    gmax = max(float(np.abs(r).max()) for _, r in grads.values())
    worst = {k: float(np.abs(d - r).max()) / (float(np.abs(r).max()) + 1e-30) for k, (d, r) in grads.items()}
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    print("%s %dx%d bs %d, decisions pinned: worst gradient tensors" % (network, size, size, batch), top,
          "| largest gradient entry %.3e" % gmax,
          {k: float(np.abs(grads[k][1]).max()) for k, _ in top})
    # resnet-50 / vgg16_reduced: 1e-3 of every tensor's largest entry.  inceptionv3 (94 batch-statistics BatchNorms, at
    # batch 1 the deepest ones average over 14 x 14 and 7 x 7 values) is not that well conditioned in float32 AT ALL: the
    # yardstick is the same restatement, same pinned decisions and targets, evaluated in float32 on the CPU against its
    # float64 self -- the device may be 1e-3, or three times that yardstick's error on the tensor, away from float64
    # (measured: device 1.2e-4 .. 3.8e-3, worst on the SSD extras below mixed_10).
    yard = {}
    if network == "inceptionv3":
        ref32 = ot.forward_loss(ot.export_params(net.g), data, lab, seg, num_classes=8, dtype=torch.float32,
                                targets=dev_targets, config=cfg, decisions=dec)
        ref32["objective"].backward()
        for name, (_, r) in grads.items():
            g32 = ot.import_grad(name, ref32["params"][name].grad).astype(np.float64)
            yard[name] = float(np.abs(g32 - r).max()) / (float(np.abs(r).max()) + 1e-30)
        print("float32 CPU restatement vs float64, same tensors:", [(k, yard[k]) for k, _ in top])
    for name, (d, r) in grads.items():
        gtol = max(1e-3, 3.0 * yard.get(name, 0.0))
        assert float(np.abs(d - r).max()) <= gtol * float(np.abs(r).max()) + 1e-6 * gmax, (name, worst[name], yard.get(name))


def _channel_spread_run(math, batch=2, size=256, guard_pass=False, widen_between_passes=False):
    """resnet-50 multitask with every learnable stage BatchNorm gamma multiplied by 2^+12 / 2^-12 alternating per channel, in
    convolution math `math`: relative errors of outputs / losses / gradients against the float64 restatement (decisions
    pinned), and the range monitor's report"""
    from dspnet_amd import functional as fn
    fn.set_conv_math(math)
    try:
        dev = torch.device("cuda", 0)
        net = get_multi_symbol_train("resnet-50", (3, size, size), num_classes=8, batch_size=batch, device=dev, seed=1)
        def widen():
            with torch.no_grad():
                for p in net.g.param_order:
                    if p.name.endswith("_gamma") and p.name.startswith("stage"):
                        c = torch.arange(p.data.numel(), device=dev)
                        p.data.mul_(torch.where(c % 2 == 0, torch.tensor(2.0 ** 12, device=dev), torch.tensor(2.0 ** -12, device=dev)))
        if not widen_between_passes:
            widen()
        gen = synthetic.rng(77)
        data = synthetic.images(batch, size, size, gen)
        lab = synthetic.det_labels(batch, gen=gen, height=size, width=size, first_empty=False)
        seg = synthetic.seg_labels(batch, size, size, gen=gen)
        solver = MultiTaskSolver(net)
        solver.set_batch(torch.from_numpy(data).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(seg).to(dev))
        if widen_between_passes:      # a calibrated, unremarkable net (what MultiTaskSolver's first step leaves) ... then the jump
            solver.forward(); solver.backward()
            net.g.guard["decide_now"] = True
            solver.forward(); solver.backward(); torch.cuda.synchronize()
            assert net.g.guard_report()[0] == 0
            widen()
        solver.forward(); solver.backward(); torch.cuda.synchronize()
        if guard_pass:     # the pass above measured the spans; this one acts on them (what MultiTaskSolver's first step does)
            net.g.guard["decide_now"] = True
            solver.forward(); solver.backward(); torch.cuda.synchronize()
        report = net.g.range_report() + net.g.guard_report()
        dec = device_decisions(net)
        cfg = get_config("resnet-50", size)
        dev_targets = [net.target.loc_target.cpu().numpy(), net.target.loc_mask.cpu().numpy(), net.target.cls_target.cpu().numpy()]
        ref = ot.forward_loss(ot.export_params(net.g), data, lab, seg, num_classes=8, dtype=torch.float64,
                              targets=dev_targets, config=cfg, decisions=dec)

        def rel(a, b):
            return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

        outs = [o.cpu().numpy() for o in net.outputs()]
        err = dict(loc_preds=rel(net.loc_preds.data.cpu().numpy(), ref["loc_preds"].numpy()),
                   cls_prob=rel(outs[0], ref["cls_prob"].numpy()), seg_out=rel(outs[4], ref["seg_out"].numpy()))
        m = MultiBoxMetric(); m.update(net)
        for n, v in zip(*m.get()):
            err["loss:" + n] = abs(v - ref[n]) / abs(ref[n])
        ref["objective"].backward()
        gerr, gmax = {}, 0.0
        for p in net.g.param_order:
            if p.name == "affine_matrix":
                continue
            gref = ot.import_grad(p.name, ref["params"][p.name].grad)
            gdev = p.grad.cpu().numpy()
            gdev = gdev[:gref.shape[0], :, :, :gref.shape[3]] if gdev.ndim == 4 else gdev[:gref.shape[0]]
            gerr[p.name] = (float(np.abs(gdev - gref).max()), float(np.abs(gref).max()))
            gmax = max(gmax, float(np.abs(gref).max()))
        err["grad_worst"] = max((d - 1e-6 * gmax) / (r + 1e-30) for d, r in gerr.values())
        if widen_between_passes:      # how many further passes until the guard has caught up
            more = 0
            while net.g.guard_report()[0] == 0 and more < 32:
                solver.forward(); solver.backward(); more += 1
            torch.cuda.synchronize()
            return err, report, more
        return err, report
    finally:
        fn.set_conv_math(fn.DEFAULT_CONV_MATH)


def test_two_piece_math_under_a_2_to_24_channel_spread(gpu_device):
    """Round 4: the condition trained nets produce and unit-scale tests do not -- channels of ONE
    tensor whose scales differ by far more than the 2^17 window in which the two-piece math is relative-accurate (the inputs
    of the convolutions behind the rescaled BatchNorms span 2^24; the next BatchNorm renormalises, so the net stays finite).
    The three fp32-result modes side by side against float64, decisions pinned.  The perturbed net is also a harder problem
    for fp32 arithmetic itself, so the yardstick is the fp32 MFMA's own error on the same net: the two-piece math may be
    1e-4 / 1e-3 (outputs, losses / gradients) or three times that yardstick away from float64; and the range monitor
    (Graph.range_report) must SEE the spread."""
    res = {m: _channel_spread_run(m) for m in ("fp32", "bf16x3", "f16x2")}
    res["f16x2 guarded"] = _channel_spread_run("f16x2", guard_pass=True)
    for m, (err, report) in res.items():
        print(m, {k: "%.2e" % v for k, v in err.items()},
              "range monitor (tensors, > 2^16, widest bits) + guard (convolutions, calls, wide slots):", report)
    seen, wide, span = res["f16x2"][1][:3]
    assert seen >= 30 and wide >= 20 and span > 20
    assert res["f16x2"][1][3:5] == (0, 0), "nothing to go by in the first pass: no fallback yet"
    yard, got = res["fp32"][0], res["f16x2"][0]
    for k, v in got.items():
        floor = 1e-3 if k == "grad_worst" else 1e-4
        assert v <= max(floor, 3.0 * yard[k]), (k, v, yard[k])
    # round 5: the GUARD.  The second pass runs the convolutions whose operands spanned more than 2^16 in
    # the three-piece bf16 math: fallbacks happen, and the result is within 1.5x the fp32 MFMA's own error (3x unguarded)
    convs, calls, slots = res["f16x2 guarded"][1][3:6]
    assert convs >= 20 and calls >= 2 * convs and slots >= 20, (convs, calls, slots)
    for k, v in res["f16x2 guarded"][0].items():
        floor = 1e-3 if k == "grad_worst" else 1e-4
        assert v <= max(floor, 1.5 * yard[k]), (k, v, yard[k])


@pytest.mark.parametrize("network,batch", [("resnet-50", 32), ("vgg16_reduced", 16)])
def test_losses_at_the_bench_batch_match_the_float64_restatement(gpu_device, network, batch):
    """the tile dispatcher keys on M = B H W, so the B = 1 .. 4 graphs of the other tests and the graphs
    bench.py times do not run the same kernel set.  Here the WHOLE graph at BASELINE.json's batch (configs[2]: resnet-50
    512 x 512 bs 32; configs[1]: vgg16_reduced 512 x 512 bs 16), automatic tiles, forward only: the loss read-outs (the
    quantities of the 1e-4 parity target) and the three output tensors against oracle/dspnet_torch.py in float64, the
    device's MultiBoxTarget matching pinned (that matching is checked bit-exactly against the C oracle at this batch in
    test_multibox_gpu.py)."""
    size = 512
    dev = torch.device("cuda", 0)
    net = get_multi_symbol_train(network, (3, size, size), num_classes=8, batch_size=batch, device=dev, seed=1)
    gen = synthetic.rng(233)
    data = synthetic.images(batch, size, size, gen)
    lab = synthetic.det_labels(batch, gen=gen, height=size, width=size)
    seg = synthetic.seg_labels(batch, size, size, gen=gen)
    solver = MultiTaskSolver(net)
    solver.set_batch(torch.from_numpy(data).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(seg).to(dev))
    solver.forward(); torch.cuda.synchronize()
    cfg = get_config(network, size)
    dev_targets = [net.target.loc_target.cpu().numpy(), net.target.loc_mask.cpu().numpy(),
                   net.target.cls_target.cpu().numpy()]
    m = MultiBoxMetric(); m.update(net)
    names, got = m.get()
    outs = [net.outputs()[i].cpu().numpy() for i in (0, 4)]
    loc_preds = net.loc_preds.data.cpu().numpy()
    values = ot.export_params(net.g)
    del net, solver
    torch.cuda.empty_cache()
    with torch.no_grad():
        ref = ot.forward_loss(values, data, lab, seg, num_classes=8, dtype=torch.float64, targets=dev_targets, config=cfg)

    def rel(a, b):
        return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

    assert len(names) == 3
    for n, v in zip(names, got):
        assert abs(v - ref[n]) <= 1e-4 * abs(ref[n]), (n, v, ref[n])
    assert rel(loc_preds, ref["loc_preds"].numpy()) < 1e-4
    assert rel(outs[0], ref["cls_prob"].numpy()) < 1e-4
    assert rel(outs[1], ref["seg_out"].numpy()) < 1e-4


def test_range_guard_lag_is_bounded_and_the_exposed_pass_is_accurate(gpu_device):
    """the guard decides from spans measured EARLIER (Graph._update_guard: looked at every GUARD_PERIOD-th
    pass, from the copy of the look before), so the passes right after a tensor's span jumps past 2^16 still run two-piece.
    Here the jump happens BETWEEN two passes of a calibrated net (every stage BatchNorm gamma times 2^+-12, alternating per
    channel): (1) the very next pass runs with no fallback -- the exposure exists -- and its outputs / losses / gradients are
    within the bound the unguarded math has under this spread (3 x the fp32 MFMA's own error, floors 1e-4 / 1e-3:
    test_two_piece_math_under_a_2_to_24_channel_spread); (2) at most 2 GUARD_PERIOD passes later the fallback is on."""
    from dspnet_amd import functional as fn
    if fn.get_conv_math() != "f16x2":
        pytest.skip("the range guard belongs to the two-piece math")
    yard, _ = _channel_spread_run("fp32")
    err, report, passes_until_guarded = _channel_spread_run("f16x2", widen_between_passes=True)
    print({k: "%.2e" % v for k, v in err.items()}, report, "guarded after", passes_until_guarded, "further passes")
    assert report[3:5] == (0, 0), "the pass right after the jump is not guarded yet"
    for k, v in err.items():
        floor = 1e-3 if k == "grad_worst" else 1e-4
        assert v <= max(floor, 3.0 * yard[k]), (k, v, yard[k])
    from dspnet_amd.engine import Graph
    assert 1 <= passes_until_guarded <= 2 * Graph.GUARD_PERIOD, passes_until_guarded


def test_range_guard_acts_on_a_recorded_step(gpu_device):
    """Advisor r5: a step replayed from a HIP graph never re-runs forward() from Python, so the guard's decisions were frozen
    at the recording.  MultiTaskSolver now polls the spans every GUARD_PERIOD-th replay and re-records when the decision
    changes: a graph recorded BEFORE any step (calibrated inside capture()), then a jump of the channel spans between two
    replays -> the recording is dropped once, the fallback is on, the run stays finite."""
    from dspnet_amd import functional as fn
    from dspnet_amd.engine import Graph
    if fn.get_conv_math() != "f16x2":
        pytest.skip("the range guard belongs to the two-piece math")
    net, solver, *_ = make(2, 128, 128)
    solver.lr = 0.0
    assert solver.capture(warmup=0) and net.g.guard["have_stats"]
    for _ in range(Graph.GUARD_PERIOD):
        solver.step()
    assert solver.graph_rerecorded == 0 and net.g.guard_report()[0] == 0
    with torch.no_grad():
        for p in net.g.param_order:
            if p.name.endswith("_gamma") and p.name.startswith("stage"):
                c = torch.arange(p.data.numel(), device=p.data.device)
                p.data.mul_(torch.where(c % 2 == 0, 2.0 ** 12, 2.0 ** -12).to(p.data.dtype))
    for _ in range(3 * Graph.GUARD_PERIOD):
        solver.step()
    torch.cuda.synchronize()
    assert solver.graph_rerecorded >= 1 and solver._graph is not None
    assert net.g.guard_report()[0] >= 20
    assert bool(torch.isfinite(net.g.grad_arena).all())


@pytest.mark.parametrize("setting", [1, 2])
def test_tile_spanning_training_is_reproducible_at_the_bench_shape(gpu_device, setting):
    """Round 6: the tile-spanning loops rely on counted waits (`s_waitcnt vmcnt(63)` past an epilogue's 64 unconditional stores)
    and on ring slots reused across tile boundaries; a hazard there would show as run-to-run differences or non-finite values
    at the BENCH shape, where every qualifying layer walks 2 - 16 tiles per workgroup (the small graphs of the other tests give
    a workgroup one tile).  Eight training steps at 32 x 512 x 512, twice from the same state, per setting of
    dspn_conv_set_tile_spanning: the parameter arenas bit-identical, finite (40 steps: profiles/r06_tile_spanning_determinism_soak.txt)."""
    from dspnet_amd import _lib, functional as fn
    if fn.get_conv_math() != "f16x2":
        pytest.skip("the tile-spanning loop belongs to the two-piece math")
    L = _lib.lib()
    dev = torch.device("cuda", 0)

    def run():
        net = get_multi_symbol_train("resnet-50", (3, 512, 512), num_classes=8, batch_size=32, device=dev, seed=0)
        gen = synthetic.rng(233)
        solver = MultiTaskSolver(net)
        solver.set_batch(torch.from_numpy(synthetic.images(32, 512, 512, gen)).to(dev),
                         torch.from_numpy(synthetic.det_labels(32, gen=gen, height=512, width=512)).to(dev),
                         torch.from_numpy(synthetic.seg_labels(32, 512, 512, gen=gen)).to(dev))
        for _ in range(8):
            solver.step()
        torch.cuda.synchronize()
        a = net.g.arena.detach().clone()
        del solver, net
        import gc
        gc.collect(); torch.cuda.empty_cache()
        return a

    try:
        _lib.check(L.dspn_conv_set_tile_spanning(setting), "set_tile_spanning")
        a, b = run(), run()
    finally:
        L.dspn_conv_set_tile_spanning(1)
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b)


@pytest.mark.parametrize("case", [("resnet-50", 512, 32, 6), ("resnet-50", 512, 3, 4), ("vgg16_reduced", 512, 2, 4), ("inceptionv3", 512, 2, 3)])
def test_weight_gradients_on_their_own_stream_give_the_same_bits(gpu_device, case):
    """Round 6: the weight gradients of the step's stream run on a second stream beside the data-gradient chain, their slab
    sums behind them (engine.WGRAD_SIDE); the step's stream waits for them at the end of the pass and before an accumulation
    into a buffer one of them still reads (the residual stream's gradient).  Same kernels, same operands: after several
    training steps the parameter arena is bit for bit what the one-stream schedule (with the BatchNorm finalize riding in the
    weight-gradient launches) gives -- at the bench shape, where the weight gradients overlap whole units of the chain, and on
    the small graphs of the other backbones, where a missed ordering would race at once."""
    from dspnet_amd import engine as E
    network, size, B, steps = case
    dev = torch.device("cuda", 0)

    def run(side):
        # (side: EVERY weight gradient of the step's stream beside the chain -- the engine's own rule keeps the short ones and the
        # graphs without a BatchNorm chain on one stream)
        prev, E.WGRAD_SIDE = (E.WGRAD_SIDE, E.WGRAD_SIDE_MIN_US, E.Graph.batchnorm_chain), side
        E.WGRAD_SIDE_MIN_US, E.Graph.batchnorm_chain = 0.0, (lambda self: True)
        try:
            net = get_multi_symbol_train(network, (3, size, size), num_classes=8, batch_size=B, device=dev, seed=0)
            gen = synthetic.rng(233)
            solver = MultiTaskSolver(net)
            solver.set_batch(torch.from_numpy(synthetic.images(B, size, size, gen)).to(dev),
                             torch.from_numpy(synthetic.det_labels(B, gen=gen, height=size, width=size)).to(dev),
                             torch.from_numpy(synthetic.seg_labels(B, size, size, gen=gen)).to(dev))
            for _ in range(steps):
                solver.step()
            torch.cuda.synchronize()
            a = net.g.arena.detach().clone()
            del solver, net
            import gc
            gc.collect(); torch.cuda.empty_cache()
            return a
        finally:
            E.WGRAD_SIDE, E.WGRAD_SIDE_MIN_US, E.Graph.batchnorm_chain = prev

    a, b, c = run(1), run(0), run(1)
    assert bool(torch.isfinite(a).all())
    assert torch.equal(a, b) and torch.equal(a, c)


def test_second_step_with_moved_affine_matrix_matches_cpu_restatement(gpu_device):
    """`affine_matrix` is an ordinary argument of the reference's graph (multitask_symbol_builder.py:574, initialised by
    multi_init.py:72, updated by multi_solver.py:291-293).  After one SGD step (large learning rate, so that the grid
    visibly leaves the identity) the sampler runs its general-affine path: the seg output, the losses, d/d affine_matrix
    and the decoder's weight gradients of the SECOND forward/backward against the float64 restatement on the device's
    updated parameters."""
    dev = torch.device("cuda", 0)
    net = get_multi_symbol_train("resnet-50", (3, 256, 256), num_classes=8, batch_size=2, device=dev, seed=1)
    gen = synthetic.rng(5)
    data = synthetic.images(2, 256, 256, gen)
    lab = synthetic.det_labels(2, gen=gen, height=256, width=256)
    seg = synthetic.seg_labels(2, 256, 256, gen=gen)
    solver = MultiTaskSolver(net, learning_rate=0.02)
    solver.set_batch(torch.from_numpy(data).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(seg).to(dev))
    theta = net.g.params["affine_matrix"]
    assert theta.data.cpu().tolist() == [1, 0, 0, 0, 1, 0]
    solver.step(); solver.step()
    moved = theta.data.cpu().numpy() - np.array([1, 0, 0, 0, 1, 0], np.float32)
    assert np.abs(moved).max() > 1e-4, moved
    solver.forward(); solver.backward(); torch.cuda.synchronize()
    cfg = get_config("resnet-50", 256)
    dev_targets = [net.target.loc_target.cpu().numpy(), net.target.loc_mask.cpu().numpy(),
                   net.target.cls_target.cpu().numpy()]
    ref = ot.forward_loss(ot.export_params(net.g), data, lab, seg, cfg["sizes"][1:], cfg["ratios"][1:],
                          dtype=torch.float64, targets=dev_targets)
    ref["objective"].backward()
    outs = net.outputs()
    a, b = outs[4].cpu().numpy(), ref["seg_out"].numpy()
    assert float(np.abs(a - b).max()) <= 1e-4 * float(np.abs(b).max())
    m = MultiBoxMetric(); m.update(net)
    for n, v in zip(*m.get()):
        assert abs(v - ref[n]) <= 1e-4 * abs(ref[n]), (n, v, ref[n])
    for name in ("affine_matrix", "score3_conv_weight", "score4_conv_weight", "score2_pool4_weight", "res3_reduced2_weight"):
        p = net.g.params[name]
        gref = ot.import_grad(name, ref["params"][name].grad)
        gdev = p.grad.cpu().numpy()
        gdev = gdev[:gref.shape[0], :, :, :gref.shape[3]] if gdev.ndim == 4 else gdev[:gref.shape[0]]
        err = float(np.abs(gdev - gref).max() / (np.abs(gref).max() + 1e-30))
        assert err < 1e-3, (name, err)
    # the checkpoint carries it in the reference's shape
    assert net.g.get_params()["affine_matrix"].shape == (1, 6)


def test_gradient_reducer_path_is_bit_identical_at_world_1(gpu_device):
    """the RCCL path (GradBucketReducer: bucketed async all_reduce released during backward) on the REAL
    graph at world size 1: after two steps the parameter arena equals the plain path's bit for bit, every bucket was
    released exactly once, in the order backward completes them, and together they cover the gradient arena."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    made = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        made = True
    try:
        nets = []
        for force in (False, True):
            dev = torch.device("cuda", 0)
            net = get_multi_symbol_train("resnet-50", (3, 128, 128), num_classes=8, batch_size=2, device=dev, seed=1)
            gen = synthetic.rng(9)
            solver = MultiTaskSolver(net, force_reducer=force, bucket_mb=4.0)
            solver.set_batch(torch.from_numpy(synthetic.images(2, 128, 128, gen)).to(dev),
                             torch.from_numpy(synthetic.det_labels(2, gen=gen, height=128, width=128)).to(dev),
                             torch.from_numpy(synthetic.seg_labels(2, 128, 128, gen=gen)).to(dev))
            solver.step(); solver.step()
            torch.cuda.synchronize()
            nets.append((net, solver))
        (na, sa), (nb, sb) = nets
        assert sa.reducer is None and sb.reducer is not None
        # round 5: the reducer path runs the N = 1 schedule -- the side-stream part of backward stays active, and the buckets
        # that hold its gradients (the SSD extra layers and their heads) wait for the side stream before their release
        assert nb.g.side_bwd is not None and nb.g.side_bwd["active"] and len(nb.g.side_bwd["side"]) == 17
        assert any(sb.bucket_side) and not all(sb.bucket_side)
        assert torch.equal(na.g.arena, nb.g.arena) and torch.equal(na.g.grad_arena, nb.g.grad_arena)
        launched = sb.reducer.launched
        assert len(launched) == len(sb.buckets) > 4
        assert launched == [(lo, hi) for lo, hi, _ in sb.buckets]               # release order = completion order
        cover = sorted(launched)
        assert cover[0][0] == 0 and cover[-1][1] == nb.g.grad_arena.numel()
        assert all(a[1] == b[0] for a, b in zip(cover, cover[1:]))              # contiguous, no overlap, no gap
    finally:
        if made:
            dist.destroy_process_group()


def test_training_reduces_losses_and_is_deterministic(gpu_device):
    net, solver, *_ = make(2, 128, 128)
    m = MultiBoxMetric()
    hist = []
    for _ in range(4):
        solver.step()
        m.reset(); m.update(net); hist.append(m.get()[1])
    assert np.isfinite(hist).all()
    assert hist[-1][0] < hist[0][0] and hist[-1][2] < hist[0][2]
    net2, solver2, *_ = make(2, 128, 128)
    hist2 = []
    for _ in range(4):
        solver2.step()
        m.reset(); m.update(net2); hist2.append(m.get()[1])
    assert hist == hist2                      # bitwise run-to-run reproducible
    assert torch.equal(net.g.arena, net2.g.arena)


def test_two_backward_passes_on_one_forward_pass_give_the_same_gradients(gpu_device):
    """default (two-piece) math: the operand magnitudes of a step live in per-graph slots that forward() clears; a SECOND
    backward pass on the same forward pass takes the gradient magnitudes again (Graph.begin_backward) instead of trusting
    slots the first one filled -- the gradients come out bit-identical, and a backward pass whose head gradients are 1000x
    larger (the slots would understate them 1000-fold) stays finite and scales accordingly"""
    net, solver, *_ = make(2, 128, 128)
    # (affine_matrix is left out: BilinearConcatConv's backward reuses the buffers that hold its forward products for the
    # gradients -- by design one backward pass per forward pass -- and d/d affine_matrix of a second pass reads those)
    th = net.g.params["affine_matrix"]
    keep = torch.ones_like(net.g.grad_arena, dtype=torch.bool); keep[th.offset:th.offset + th.size] = False

    def grads():
        return net.g.grad_arena[keep].clone()
    solver.forward()
    solver.backward()
    g1 = grads()
    assert bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0
    solver.backward()
    assert torch.equal(grads(), g1)
    # a second pass whose head gradient is 1000x larger: magnitude slots left over from the first pass would understate the
    # segmentation branch's gradients a thousandfold (fp16 overflow in the two-piece math); re-taken, the result is exactly what
    # a single backward pass with that head gradient gives
    seg = net.seg_out
    solver.forward()
    seg.gbuf.mul_(1000.0)
    solver.backward()
    ref = grads()
    solver.forward()
    solver.backward()
    seg.gbuf.mul_(1000.0)
    solver.backward()
    assert bool(torch.isfinite(ref).all()) and torch.equal(grads(), ref)
    assert float((ref - g1).abs().max()) > 10 * float(g1.abs().max())       # (the larger head gradient did arrive)


def test_step_replayed_from_a_hip_graph_is_bit_identical(gpu_device):
    """MultiTaskSolver.capture(): the whole step (forward, backward incl. the side-stream MultiBoxDetection, SGD) recorded
    into one HIP graph after two eager steps and replayed -- every library call is capturable (explicit stream, no
    allocation, no synchronisation) -- gives the same parameters, bit for bit, as running the same steps eagerly."""
    net_a, solver_a, *_ = make(2, 128, 128)
    net_b, solver_b, *_ = make(2, 128, 128)
    for _ in range(5):
        solver_a.step()
    assert solver_b.capture(warmup=2)          # 2 eager steps, then the recording pass (recording runs nothing)
    for _ in range(3):
        solver_b.step()                        # 3 replays
    torch.cuda.synchronize()
    assert torch.equal(net_a.g.arena, net_b.g.arena)
    assert torch.equal(net_a.outputs()[3], net_b.outputs()[3])


def test_bf16_tensor_training_tracks_the_float_run(gpu_device):
    """End to end with bfloat16 tensors in HBM (every activation, every gradient of an activation, every convolution
    operand; float master weights, statistics, losses): five SGD steps of the resnet-50 multi-task graph stay finite,
    reduce the three losses, are bitwise reproducible run to run, and track the float-tensor run of the same graph (which
    uses the same bf16 MFMA math, so the difference is the storage rounding alone, 2^-9 per tensor): the two softmax
    cross-entropies within 8 % / 3 % at every step (measured 1.3 - 4.4 % / 0.7 - 1.0 %); SmoothL1 -- a mean over the few dozen positive
    anchors of two images, whose membership moves with every re-matching -- within 15 % (measured 9 % at step 5)."""
    from dspnet_amd import functional as fn

    def run(store):
        fn.set_conv_math("bf16")
        fn.set_activation_dtype(store)
        try:
            net, solver, *_ = make(2, 256, 256)
        finally:
            fn.set_activation_dtype("fp32")
        m, hist = MultiBoxMetric(), []
        for _ in range(5):
            solver.step()
            m.reset(); m.update(net); hist.append(m.get()[1])
        torch.cuda.synchronize()
        return net, np.asarray(hist)

    try:
        net_h, hist_h = run("bf16")
        net_h2, hist_h2 = run("bf16")
        net_f, hist_f = run("fp32")
    finally:
        fn.set_conv_math(fn.DEFAULT_CONV_MATH)
    assert net_h.g.tensors["_plus15"].data.dtype == torch.bfloat16 and net_f.g.tensors["_plus15"].data.dtype == torch.float32
    assert net_h.g.arena.dtype == torch.float32                       # master weights stay float
    assert np.isfinite(hist_h).all() and torch.isfinite(net_h.g.arena).all()
    assert np.array_equal(hist_h, hist_h2) and torch.equal(net_h.g.arena, net_h2.g.arena)
    assert (hist_h[-1] < hist_h[0]).all()
    dev = np.abs(hist_h / hist_f - 1).max(axis=0)          # columns: CrossEntropy, SmoothL1, SegCrossEntropy
    # (round 6: 1.3 % / 9 % / 0.7 % became 4.4 % / 4.4 % / 1.0 % when an ulp-level change in the sampler's data gradient moved
    # both trajectories -- this run amplifies rounding differences by about three orders of magnitude per step, see
    # test_split_math_training_tracks_the_fp32_mfma_run; the bounds are the level of that amplification, not of bf16)
    assert dev[0] < 8e-2 and dev[2] < 3e-2 and dev[1] < 0.15, (dev, hist_h, hist_f)


def test_split_math_training_tracks_the_fp32_mfma_run(gpu_device):
    """The default convolution math (DSPN_MATH_F32_BF16X3: fp32 products from six exact bf16 partial products) against the
    fp32 MFMA over five SGD steps of the resnet-50 multi-task graph from the same initial state.  The first step's losses
    agree to 2e-6 (the precision of either evaluation).  After that this training problem (random initialisation, batch
    statistics over two 256x256 images, re-matched anchors every step) amplifies ANY fp32 rounding difference by about three
    orders of magnitude per step, so the yardstick for the later steps is a control pair of two fp32-MFMA evaluations of the
    same run that differ only in summation order (BatchNorm folded into the convolutions vs the stand-alone kernels,
    scratch/chaos_control.py: losses 0.9 % / 3 % / 0.4 %, parameters 2.1e-3): the split math stays within 4x of that pair
    (measured 2 % / 8 % / 0.45 %, parameters 2.3e-3 = 1.1x).  Bitwise reproducible run to run."""
    from dspnet_amd import engine as E
    from dspnet_amd import functional as fn

    def run(math, fuse=True):
        fn.set_conv_math(math)
        E.FUSE_BATCHNORM = fuse
        try:
            net, solver, *_ = make(2, 256, 256)
        finally:
            E.FUSE_BATCHNORM = True
        m, hist = MultiBoxMetric(), []
        for _ in range(5):
            solver.step()
            m.reset(); m.update(net); hist.append(m.get()[1])
        torch.cuda.synchronize()
        return net.g.arena.double().clone(), np.asarray(hist)

    try:
        a_s, hist_s = run("bf16x3")
        a_s2, hist_s2 = run("bf16x3")
        a_f, hist_f = run("fp32")
        a_u, hist_u = run("fp32", fuse=False)
    finally:
        fn.set_conv_math(fn.DEFAULT_CONV_MATH)
    assert np.array_equal(hist_s, hist_s2) and torch.equal(a_s, a_s2)
    assert np.isfinite(hist_s).all() and (hist_s[-1] < hist_s[0]).all()
    dev, ctl = np.abs(hist_s / hist_f - 1), np.abs(hist_u / hist_f - 1)       # columns: CrossEntropy, SmoothL1, SegCrossEntropy
    pdev, pctl = float((a_s - a_f).norm() / a_f.norm()), float((a_u - a_f).norm() / a_f.norm())
    print("bf16x3 vs fp32 MFMA: first step", dev[0], "worst", dev.max(axis=0), "parameters %.2e | control (two fp32 orders): worst" % pdev,
          ctl.max(axis=0), "parameters %.2e" % pctl)
    assert dev[0].max() < 2e-6, dev[0]
    assert pdev < 4 * pctl, (pdev, pctl)
    # Round 6: the control pair's deviation is itself a sample of this chaotic run -- an ulp-level change in ONE kernel (the
    # sampler's data gradient, same sums in the same order) moved it from (3.1 %, 1.9 %, 0.36 %) to (0.88 %, 0.40 %, 0.34 %)
    # while the split math's deviation stayed where it was (2.4 / 3.2 / 0.68 % -> 2.0 / 3.3 / 0.55 %).  The yardstick is
    # therefore the control pair OR its level over the runs on record (scratch/chaos_control.py: 0.9 % / 3 % / 0.4 %),
    # whichever is larger; the parameter distance above (2.3e-3 vs 2.1e-3 in every run on record) is the stable statement.
    floor = np.array([0.009, 0.03, 0.004])
    assert (dev.max(axis=0) < 4 * np.maximum(ctl.max(axis=0), floor) + 1e-3).all(), (dev, ctl)


def test_test_graph_matches_training_graph_outputs(gpu_device):
    """get_multi_symbol (symbol/multitask_symbol_builder.py:595-726) yields the det / seg values of the
    training graph's outputs[3], outputs[4] (what detect/multitask_detector.py:234 reads)"""
    from dspnet_amd.detect.multitask_detector import Detector
    net, solver, data, lab, seg = make(2, 256, 256)
    solver.forward()
    outs = net.outputs()
    det = Detector("resnet-50", 256, num_classes=8, batch_size=2, seed=1)
    assert torch.equal(det.net.g.arena, net.g.arena)
    d, s = det.forward(torch.from_numpy(data).cuda())
    assert torch.equal(d, outs[3])
    assert torch.allclose(det.net.outputs()[1], outs[4], atol=0, rtol=0)
    rows, segp = det.detect(torch.from_numpy(data).cuda())
    assert len(rows) == 2 and rows[0].shape[1] == 7 and (rows[0][:, 0] >= 0).all()
    assert segp.shape == (2, 19, 64, 64)


def test_segmentation_only_graphs_match_cpu_restatement(gpu_device):
    """get_seg_symbol_train / get_seg_symbol (symbol/multitask_symbol_builder.py:211-440): backbone -> pyramid decoder
    only, conv_feat read from the backbone (`_plus15`), no SSD layers; output [seg_out].  Values, SegCrossEntropy and
    every gradient against the float64 restatement; the test graph reproduces the training graph's seg_out."""
    from dspnet_amd.symbol.multitask_symbol_factory import get_seg_symbol, get_seg_symbol_train
    dev = torch.device("cuda", 0)
    B, S = 2, 256
    net = get_seg_symbol_train("resnet-50", S, num_classes=8, batch_size=B, device=dev, seed=5)
    names = [p.name for p in net.g.param_order]
    assert not any(n.startswith("multi_feat") or "pred_conv" in n for n in names) and "affine_matrix" in names
    assert net.label_det is None and net.det is None
    gen = synthetic.rng(31)
    data = synthetic.images(B, S, S, gen)
    seg = synthetic.seg_labels(B, S, S, gen=gen)
    solver = MultiTaskSolver(net)
    solver.set_batch(torch.from_numpy(data).to(dev), None, torch.from_numpy(seg).to(dev))
    solver.forward(); solver.backward(); torch.cuda.synchronize()
    outs = net.outputs()
    assert len(outs) == 1 and tuple(outs[0].shape) == (B, 19, S // 4, S // 4)
    # the device's own ReLU signs / max-pool picks handed to the restatement (see the full-size test above): both then
    # differentiate the same piecewise-linear function, and every gradient tensor is held element-wise
    ref = ot.forward_loss(ot.export_params(net.g), data, None, seg, num_classes=8, dtype=torch.float64,
                          config=get_config("resnet-50", S), with_det=False, decisions=device_decisions(net))
    a, b = outs[0].cpu().numpy(), ref["seg_out"].numpy()
    assert np.abs(a - b).max() < 1e-4 * np.abs(b).max()
    m = MultiBoxMetric(); m.update(net)
    v = dict(zip(*m.get()))
    assert abs(v["SegCrossEntropy"] - ref["SegCrossEntropy"]) <= 1e-4 * ref["SegCrossEntropy"]
    assert np.isnan(v["CrossEntropy"])                     # nothing counted for the tasks the graph does not have
    ref["objective"].backward()
    grads = {}
    for p in net.g.param_order:
        if p.name == "affine_matrix":      # identity grid: interpolation kinks (see test_second_step_with_moved_affine...)
            continue
        gref = ot.import_grad(p.name, ref["params"][p.name].grad)
        gdev = p.grad.cpu().numpy()
        grads[p.name] = (gdev[:gref.shape[0], :, :, :gref.shape[3]] if gdev.ndim == 4 else gdev[:gref.shape[0]], gref)
    gmax = max(float(np.abs(r).max()) for _, r in grads.values())
    for name, (d, r) in grads.items():      # absolute floor for gradients that are zero in exact arithmetic (bn0_gamma)
        assert np.abs(d - r).max() <= 1e-3 * np.abs(r).max() + 1e-6 * gmax, (name, np.abs(d - r).max(), np.abs(r).max())

    test_net = get_seg_symbol("resnet-50", S, num_classes=8, batch_size=B, device=dev, seed=5)
    assert torch.equal(test_net.g.arena, net.g.arena)
    test_net.data.data.copy_(torch.from_numpy(data).to(dev))
    test_net.g.forward()
    t_outs = test_net.outputs()
    assert len(t_outs) == 1 and torch.equal(t_outs[0], outs[0])


def test_single_task_graphs_train(gpu_device):
    """MultiTaskSolver on the two single-task training graphs (get_seg_symbol_train / get_det_symbol_train, the graphs
    seg_solver.py / det_solver.py of the reference drive): five SGD steps reduce the task's losses, bitwise reproducibly"""
    from dspnet_amd.symbol.multitask_symbol_factory import get_det_symbol_train, get_seg_symbol_train
    dev = torch.device("cuda", 0)

    def run(kind):
        f = get_seg_symbol_train if kind == "seg" else get_det_symbol_train
        net = f("resnet-50", 256, num_classes=8, batch_size=2, device=dev, seed=11)
        gen = synthetic.rng(5)
        data = synthetic.images(2, 256, 256, gen)
        lab = synthetic.det_labels(2, gen=gen, height=256, width=256, first_empty=False)
        seg = synthetic.seg_labels(2, 256, 256, gen=gen)
        solver = MultiTaskSolver(net)
        solver.set_batch(torch.from_numpy(data).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(seg).to(dev))
        m, hist = MultiBoxMetric(), []
        for _ in range(5):
            solver.step()
            m.reset(); m.update(net); hist.append(m.get()[1])
        torch.cuda.synchronize()
        return np.asarray(hist), net.g.arena.clone()

    for kind, cols in (("seg", [2]), ("det", [0, 1])):
        h1, a1 = run(kind)
        h2, a2 = run(kind)
        assert torch.equal(a1, a2) and np.array_equal(h1[:, cols], h2[:, cols])
        assert np.isfinite(h1[:, cols]).all() and (h1[-1, cols] < h1[0, cols]).all(), (kind, h1)
        other = [c for c in range(3) if c not in cols]
        assert np.isnan(h1[:, other]).all() or (h1[:, other] == 0).all()      # the absent tasks are not counted


def test_detection_only_test_graph(gpu_device):
    """get_det_symbol (symbol/multitask_symbol_builder.py:123-209): output [det], equal to outputs[3] of
    get_det_symbol_train on the same parameters"""
    from dspnet_amd.symbol.multitask_symbol_factory import get_det_symbol, get_det_symbol_train
    dev = torch.device("cuda", 0)
    net, solver, data, lab, seg = _vgg_case("det", 300, 1, 20)
    solver.forward()
    t = get_det_symbol("vgg16_reduced", 300, num_classes=20, batch_size=1, device=dev, seed=3)
    assert torch.equal(t.g.arena, net.g.arena)
    t.data.data.copy_(torch.from_numpy(data).to(dev))
    t.g.forward()
    outs = t.outputs()
    assert len(outs) == 1 and torch.equal(outs[0], net.outputs()[3])


def _vgg_case(kind, size, batch, classes, network="vgg16_reduced"):
    from dspnet_amd.symbol.multitask_symbol_factory import get_det_symbol_train
    dev = torch.device("cuda", 0)
    f = get_det_symbol_train if kind == "det" else get_multi_symbol_train
    net = f(network, size, num_classes=classes, batch_size=batch, device=dev, seed=3)
    gen = synthetic.rng(77)
    data = synthetic.images(batch, size, size, gen)
    lab = synthetic.det_labels(batch, gen=gen, num_classes=classes, height=size, width=size, first_empty=False)
    seg = synthetic.seg_labels(batch, size, size, gen=gen) if kind != "det" else None
    solver = MultiTaskSolver(net)
    net.data.data.copy_(torch.from_numpy(data).to(dev))
    net.label_det.data.copy_(torch.from_numpy(lab).to(dev))
    if seg is not None:
        net.label_seg.data.copy_(torch.from_numpy(seg).to(dev))
    return net, solver, data, lab, seg


def test_inceptionv3_shapes_1024x512(gpu_device):
    """BASELINE.json configs[3] shape: the reference's stem/tower geometry (symbol/inceptionv3.py:114-160) at 512x1024"""
    net = get_multi_symbol_train("inceptionv3", (3, 512, 1024), num_classes=8, batch_size=1)
    t = net.g.tensors
    assert t["ch_concat_mixed_2_chconcat"].shape == (1, 61, 125, 288)
    assert t["ch_concat_mixed_7_chconcat"].shape == (1, 30, 62, 768)
    assert t["ch_concat_mixed_10_chconcat"].shape == (1, 14, 30, 2048)
    net.g.forward()
    outs = net.outputs()
    assert tuple(outs[4].shape) == (1, 19, 128, 256)
    assert torch.isfinite(outs[0]).all() and torch.isfinite(outs[4]).all()


@pytest.mark.parametrize("network,kind,size,batch,classes", [("vgg16_reduced", "det", 300, 1, 20),
                                                             ("vgg16_reduced", "multi", 320, 2, 8),
                                                             ("inceptionv3", "multi", 512, 1, 8)])
def test_other_backbone_graphs_match_cpu_restatement(gpu_device, conv_math, network, kind, size, batch, classes):
    """BASELINE.json configs[0] (vgg16_reduced SSD-300 single-task det, bs=1, 20 VOC classes: N = 2956 anchors
    after the reference's [1:] slice of the 8732-anchor preset), the build's vgg16_reduced multi-task wiring, and
    the inceptionv3 multi-task wiring (configs[3]'s backbone; every conv class of symbol/inceptionv3.py)"""
    net, solver, data, lab, seg = _vgg_case(kind, size, batch, classes, network)
    solver.forward(); solver.backward(); torch.cuda.synchronize()
    cfg = get_config(network, size)
    if kind == "det":
        assert tuple(net.anchors.shape) == (1, 2956, 4)
    anchors = net.anchors.cpu().numpy()
    dev_targets = [net.target.loc_target.cpu().numpy(), net.target.loc_mask.cpu().numpy(),
                   net.target.cls_target.cpu().numpy()]
    mc.assert_target_equal(dev_targets, om.multibox_target(anchors, lab, net.target.cls_preds.data.cpu().numpy(),
                                                           negative_mining_ratio=3))
    ref = ot.forward_loss(ot.export_params(net.g), data, lab, seg, num_classes=classes, dtype=torch.float64,
                          targets=dev_targets, config=cfg, with_seg=(kind != "det"))
    np.testing.assert_array_equal(ref["anchors"], anchors)

    def rel(a, b):
        return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

    outs = [o.cpu().numpy() for o in net.outputs()]
    # element-wise tensor bound: 1e-4 of the tensor's scale.  The 94-layer inceptionv3 stack of batch-statistics
    # BatchNorms over small maps is worse conditioned in fp32: the CPU restatement itself run in float32 differs from
    # its float64 run by 2.2e-4 (loc_preds) and 2.9e-4 .. 9e-4 (seg_out) on this input (scratch/incep_dbg.py on the
    # MI355X box; the device shows 2.3e-4 / 4.3e-4 .. 6.1e-4), so tensors get 2e-3 there.  The loss readouts -- the
    # quantity BASELINE.json's 1e-4 is stated for -- are held to 1e-4 for every backbone below.
    ttol = 2e-3 if network == "inceptionv3" else 1e-4
    assert rel(net.loc_preds.data.cpu().numpy(), ref["loc_preds"].numpy()) < ttol
    assert rel(outs[0], ref["cls_prob"].numpy()) < ttol
    assert rel(outs[1], ref["loc_loss"].numpy()) < ttol
    np.testing.assert_array_equal(outs[3], om.multibox_detection(outs[0], net.loc_preds.data.cpu().numpy(), anchors,
                                                                 nms_threshold=.5, nms_topk=400))
    if kind != "det":
        assert rel(outs[4], ref["seg_out"].numpy()) < ttol
    m = MultiBoxMetric(); m.update(net)
    names, vals = m.get()
    for n, v in zip(names, vals):
        if n in ref:
            assert abs(v - ref[n]) <= 1e-4 * abs(ref[n]), (n, v, ref[n])
    ref["objective"].backward()
    num = den = 0.0
    for p in net.g.param_order:
        gref = ot.import_grad(p.name, ref["params"][p.name].grad)
        gdev = p.grad.cpu().numpy()
        gdev = gdev[:gref.shape[0], :, :, :gref.shape[3]] if gdev.ndim == 4 else gdev[:gref.shape[0]]
        if p.name.endswith(("pred_conv_weight", "pred_conv_bias")):      # nothing but the loss below them
            assert rel(gdev, gref) < 10 * ttol, p.name
        if p.name == "affine_matrix":      # identity grid: d/d affine_matrix sits on the interpolation kinks (see above)
            continue
        num += float(((gdev - gref) ** 2).sum()); den += float((gref ** 2).sum())
    # global relative L2 over all parameter gradients (ReLU sign flips of pre-activations within rounding of zero
    # make an element-wise bound meaningless below the heads).  inceptionv3: the float32 CPU
    # restatement is itself 3.6e-2 away from the float64 one on this input; the device measures 2.8e-2.
    assert (num / den) ** 0.5 < (6e-2 if network == "inceptionv3" else 2e-2)


def test_bf16_mfma_graph_losses_close_to_fp32_restatement(gpu_device):
    """BASELINE.json configs[3] runs its convolutions on bf16 MFMA (fp32 tensors, fp32 accumulate): the three loss
    readouts of the resnet-50 multi-task graph stay within 2e-2 of the float64 CPU restatement (operand rounding
    2^-9 per factor, averaged over K >= 64 products), and the step is still deterministic."""
    from dspnet_amd import functional as fn
    fn.set_conv_math("bf16")
    try:
        net, solver, data, lab, seg = make(2, 256, 256)
        solver.forward(); solver.backward(); torch.cuda.synchronize()
        dev_targets = [net.target.loc_target.cpu().numpy(), net.target.loc_mask.cpu().numpy(),
                       net.target.cls_target.cpu().numpy()]
        cfg = get_config("resnet-50", 256)
        ref = ot.forward_loss(ot.export_params(net.g), data, lab, seg, cfg["sizes"][1:], cfg["ratios"][1:],
                              dtype=torch.float64, targets=dev_targets)
        m = MultiBoxMetric(); m.update(net)
        names, vals = m.get()
        for n, v in zip(names, vals):
            if n in ref:
                assert abs(v - ref[n]) <= 2e-2 * abs(ref[n]), (n, v, ref[n])
        # ... and against the oracle that rounds the conv operands to bf16 as well (same arithmetic), with the CPU's own
        # float32 evaluation of that arithmetic as the yardstick
        devq, c32 = bf16_parity_report(net, data, lab, seg, None, legacy=(cfg["sizes"][1:], cfg["ratios"][1:]))
        print("bf16 resnet-50 256x256 vs bf16-operand oracle: device", devq, "| oracle fp32 vs fp64", c32)
        assert_within_yardstick(devq, c32)
        g1 = net.g.grad_arena.clone()
        solver.forward(); solver.backward(); torch.cuda.synchronize()
        assert torch.equal(g1, net.g.grad_arena)
    finally:
        fn.set_conv_math(fn.DEFAULT_CONV_MATH)


def test_inceptionv3_bf16_1024x512_matches_cpu_restatement(gpu_device):
    """BASELINE.json configs[3] in its own precision and at its own shape: inceptionv3 multi-task graph, 1024x512
    (H 512, W 1024), convolutions on bf16 MFMA (operands rounded to bf16, fp32 accumulate).

    The oracle runs THE SAME ARITHMETIC: oracle/dspnet_torch.py with conv_quant="bf16" rounds both operands of every
    convolution GEMM (forward, data gradient, weight gradient, transposed convolution) to bfloat16 and accumulates in
    float64, so what is compared is the device's bf16 path against its specification, not bf16 against fp64.  (Against
    the UNQUANTISED float64 restatement this 94-layer stack of batch-statistics BatchNorms with random weights is not a
    usable yardstick: it amplifies rounding noise ~1e4-fold -- fp32 vs fp64 already differ by 2e-4 .. 9e-4, DESIGN.md
    section 3 -- and the bf16 run measures 0.7 .. 0.8 of the tensor scale away, 15 % on the cross-entropy.)
    What remains between device and oracle: fp32 accumulation order, and activations that the device computes in fp32
    and the oracle in fp64 before the SAME rounding to bf16 -- a value within fp32 noise of a bf16 rounding boundary
    lands on the neighbouring bf16 number (one 2^-9 step on a small fraction of elements per layer), which the same
    amplification carries to the outputs.  The tolerance is therefore not a constant but a yardstick measured in the
    test: the SAME oracle evaluated in float32 on the CPU shows both effects against its float64 self; the device has
    to stay within 4x of that on every tensor, loss and gradient norm (bf16_parity_report / assert_within_yardstick).
    A wrong tap, a wrong tile or a missing rounding would be orders of magnitude outside."""
    from dspnet_amd import functional as fn
    fn.set_conv_math("bf16")
    try:
        dev = torch.device("cuda", 0)
        H, W, B = 512, 1024, 1
        net = get_multi_symbol_train("inceptionv3", (3, H, W), num_classes=8, batch_size=B, device=dev, seed=3)
        gen = synthetic.rng(78)
        data = synthetic.images(B, H, W, gen)
        lab = synthetic.det_labels(B, gen=gen, height=H, width=W, first_empty=False)
        seg = synthetic.seg_labels(B, H, W, gen=gen)
        solver = MultiTaskSolver(net)
        solver.set_batch(torch.from_numpy(data).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(seg).to(dev))
        solver.forward(); solver.backward(); torch.cuda.synchronize()
        assert fn.get_conv_math() == "bf16"
        cfg = get_config("inceptionv3", H)
        anchors = net.anchors.cpu().numpy()
        dev_targets = [net.target.loc_target.cpu().numpy(), net.target.loc_mask.cpu().numpy(),
                       net.target.cls_target.cpu().numpy()]
        # the operators inside the graph stay bit-exact against the C oracle on the device's own (bf16-produced) inputs
        mc.assert_target_equal(dev_targets, om.multibox_target(anchors, lab, net.target.cls_preds.data.cpu().numpy(),
                                                               negative_mining_ratio=3))
        devq, c32 = bf16_parity_report(net, data, lab, seg, cfg)
        print("bf16 inceptionv3 512x1024 vs bf16-operand oracle: device", devq, "| oracle fp32 vs fp64", c32)
        assert_within_yardstick(devq, c32)
        # absolute backstops on the quantities BASELINE.json names (losses), whatever the yardstick says
        for k in devq:
            if k.startswith("loss:"):
                assert devq[k] <= 5e-2, (k, devq[k])
        g1 = net.g.grad_arena.clone()
        solver.forward(); solver.backward(); torch.cuda.synchronize()
        assert torch.equal(g1, net.g.grad_arena)              # deterministic in bf16 mode as well
    finally:
        fn.set_conv_math(fn.DEFAULT_CONV_MATH)


@pytest.mark.parametrize("network,H,W,B,store", [("inceptionv3", 512, 1024, 1, "fp32"), ("inceptionv3", 512, 1024, 1, "bf16"),
                                                 ("resnet-50", 256, 256, 2, "bf16")])
def test_bf16_every_convolution_layer_local(gpu_device, network, H, W, B, store):
    """configs[3] layer by layer, without the chaos: EVERY Conv node of the multi-task graph (inceptionv3 at 1024x512)
    in bf16 mode -- forward (with its folded BatchNorm-apply+ReLU loader where the graph uses one), data gradient and
    weight gradient -- against float64 convolutions of the SAME bf16-rounded operands, each fed with the device's OWN
    input tensors (so no error propagates from layer to layer).
    store = "fp32": float tensors, operands rounded on the way into LDS.  What is left is fp32 accumulation: 1e-5 of the
    output scale (1e-4 where a data gradient is accumulated onto earlier contributions and has to be recovered as a
    difference).
    store = "bf16": bfloat16 tensors in HBM (the `*_bf16` kernels).  Inputs are then bf16 exactly; an output tensor carries
    its own rounding on top: |err| <= 2^-8 |ref| + 1e-5 scale per element; weight gradients (float) stay at 1e-5."""
    import torch.nn.functional as F
    from dspnet_amd import engine as E
    from dspnet_amd import functional as fn
    fn.set_conv_math("bf16")
    fn.set_activation_dtype(store)
    half = store == "bf16"
    try:
        dev = torch.device("cuda", 0)
        net = get_multi_symbol_train(network, (3, H, W), num_classes=8, batch_size=B, device=dev, seed=3)
        fn.set_activation_dtype("fp32")
        assert net.g.tensors["multibox_loc_pred"].data.dtype == torch.float32
        gen = synthetic.rng(78)
        solver = MultiTaskSolver(net)
        solver.set_batch(torch.from_numpy(synthetic.images(B, H, W, gen)).to(dev),
                         torch.from_numpy(synthetic.det_labels(B, gen=gen, height=H, width=W, first_empty=False)).to(dev),
                         torch.from_numpy(synthetic.seg_labels(B, H, W, gen=gen)).to(dev))
        g = net.g
        g.forward()
        torch.cuda.synchronize()

        def q(t):                      # round to bf16 (RNE), hold as float64
            return t.float().bfloat16().double()

        def nchw64(t, c):              # device NHWC fp32 -> CPU NCHW float64, logical channels
            return t.detach().cpu().double().permute(0, 3, 1, 2)[:, :c].contiguous()

        def conv_input(n):
            """the tensor the convolution multiplies: (relu)(x_raw * scale + shift) evaluated like the loader's fmaf
            (exact product + one rounding), or the plain input"""
            cin = n.w.logical[1]
            if n.in_affine is None:
                return nchw64(n.x.data, cin)
            sc, sh, relu = n.in_affine
            x = nchw64(n.x_raw.data, cin)
            u = (x * sc.cpu().double()[:cin].view(1, -1, 1, 1) + sh.cpu().double()[:cin].view(1, -1, 1, 1)).float().double()
            return u.clamp_min(0) if relu else u

        def weight(n):
            cout, cin = n.w.logical[0], n.w.logical[1]
            return n.w.data.detach().cpu().double()[:cout, :, :, :cin].permute(0, 3, 1, 2).contiguous()

        def rel(a, b, stored=False):
            """error in units of the bound: fp32 accumulation (1e-5 of the scale), plus the tensor's own bf16 rounding
            when the result went through a bf16 store"""
            scale = float(b.abs().max()) + 1e-30
            if not stored:
                return float((a - b).abs().max() / scale) / 1e-5
            return float(((a - b).abs() / (b.abs() * 2.0 ** -8 + 1e-5 * scale)).max())

        convs = [n for n in g.nodes if isinstance(n, E.Conv)]
        assert len(convs) > (100 if network == "inceptionv3" else 60)
        assert all(n.out.data.dtype == (torch.bfloat16 if half else torch.float32) for n in convs)
        kinds, worst_f = set(), 0.0
        for n in convs:
            xq, wq = q(conv_input(n)), q(weight(n))
            bias = None if n.b is None else n.b.data.cpu().double()[:n.cout]
            ref = F.conv2d(xq, wq, bias, stride=n.stride, padding=n.pad, dilation=n.dil)
            if n.residual is not None:
                ref = ref + nchw64(n.residual.data, n.cout)
            if n.relu:
                ref = ref.clamp_min(0)
            e = rel(nchw64(n.out.data, n.cout), ref, stored=half)
            worst_f = max(worst_f, e)
            assert e <= 1.0, ("forward", n.w.name, e)
            kinds.add((tuple(n.w.shape[1:3]), n.stride, n.pad, n.in_affine is not None, n.tap_expand))
        if network == "inceptionv3":     # every conv class of symbol/inceptionv3.py is in there
            assert {(1, 7), (7, 1), (1, 3), (3, 1), (5, 5), (3, 3), (1, 1)} <= {k[0] for k in kinds}

        # backward, node by node, with the output gradient each convolution actually received
        g.begin_backward()
        worst_w = worst_d = 0.0
        checked_d = 0
        for n in reversed(g.nodes):
            is_conv = isinstance(n, E.Conv) and n.out._gw
            if is_conv:
                dy = n.out.grad.clone()
                had = n.x.requires_grad and n.x._gw
                before = n.x.grad.clone() if had else None
            n.backward()
            if not is_conv:
                continue
            if n.slabs is not None:
                g.flush_slabs(("one", n.w.name), [n])
            torch.cuda.synchronize()
            dyc = nchw64(dy, n.cout)
            if n.relu:
                dyc = dyc * (nchw64(n.out.data, n.cout) > 0)
            xq, wq, dq = q(conv_input(n)), q(weight(n)), q(dyc)
            gw = torch.nn.grad.conv2d_weight(xq, wq.shape, dq, stride=n.stride, padding=n.pad, dilation=n.dil)
            cout, cin = n.w.logical[0], n.w.logical[1]
            got_w = n.w.grad.detach().cpu().double()[:cout, :, :, :cin].permute(0, 3, 1, 2)
            e = rel(got_w, gw)
            worst_w = max(worst_w, e)
            assert e <= 1.0, ("wgrad", n.w.name, e)
            if n.x.requires_grad:
                gx = torch.nn.grad.conv2d_input(xq.shape, wq, dq, stride=n.stride, padding=n.pad, dilation=n.dil)
                after = nchw64(n.x.grad, cin)
                if had and half:
                    # accumulated onto an earlier (stored, rounded) contribution: the sum is rounded once more
                    e = rel(after, nchw64(before, cin) + gx, stored=True)
                    assert e <= 1.0, ("dgrad(acc)", n.w.name, e)
                elif had:
                    e = float(((after - nchw64(before, cin)) - gx).abs().max() / (after.abs().max() + 1e-30)) / 1e-4
                    assert e <= 1.0, ("dgrad(acc)", n.w.name, e)
                else:
                    e = rel(after, gx, stored=half)
                    assert e <= 1.0, ("dgrad", n.w.name, e)
                worst_d = max(worst_d, e)
                checked_d += 1
        assert checked_d > (90 if network == "inceptionv3" else 55)
        print("bf16 math, %s tensors, %s %dx%d layer-local: %d convolutions, worst error / bound: forward %.3f, wgrad %.3f, "
              "dgrad %.3f" % (store, network, H, W, len(convs), worst_f, worst_w, worst_d))
    finally:
        fn.set_conv_math(fn.DEFAULT_CONV_MATH)
        fn.set_activation_dtype("fp32")


def test_fused_and_unfused_batchnorm_graphs_agree(gpu_device):
    """The BatchNorm work folded into the convolutions (statistics in the producer's epilogue, apply+ReLU in the
    consumers' loaders, backward reductions in the data-gradient epilogue) against the same graph built from the
    stand-alone BatchNorm kernels: same losses to 1e-6; gradients to 2e-2 in global relative L2 -- the two builds get
    their batch statistics by different summation orders, scale/shift differ in the last bits, and every ReLU whose
    pre-activation lies within that rounding of zero flips (the same sensitivity as fp32 vs fp64, DESIGN.md section 3;
    measured 1.2e-2 at batch 2, 256x256)."""
    from dspnet_amd import engine as E

    def run(fuse):
        E.FUSE_BATCHNORM = fuse
        try:
            net, solver, data, lab, seg = make(2, 256, 256)
        finally:
            E.FUSE_BATCHNORM = True
        solver.forward(); solver.backward(); torch.cuda.synchronize()
        m = MultiBoxMetric(); m.update(net)
        return net, dict(zip(*m.get()))

    net_f, loss_f = run(True)
    net_u, loss_u = run(False)
    assert any(getattr(n, "defer_apply", False) for n in net_f.g.nodes) and not any(
        getattr(n, "defer_apply", False) for n in net_u.g.nodes)
    for k in loss_f:
        assert abs(loss_f[k] - loss_u[k]) <= 1e-6 * abs(loss_u[k]) + 1e-9, (k, loss_f[k], loss_u[k])
    gf = {p.name: p.grad for p in net_f.g.param_order}
    num = den = 0.0
    for p in net_u.g.param_order:
        a, b = gf[p.name].double(), p.grad.double()
        num += float(((a - b) ** 2).sum()); den += float((b ** 2).sum())
    assert (num / den) ** 0.5 < 2e-2


@pytest.mark.gpu
def test_score3_conv_per_level_matches_direct_form(gpu_device):
    """engine.BilinearConcatConv (each pyramid level multiplied at its own resolution, results resized and summed)
    against the direct form it replaces (3328-channel concatenation + tap-expanded convolution) on the same weights
    and inputs: the segmentation output, the three losses and every parameter gradient.  The two are equal in real
    arithmetic; in fp32 they differ by summation order only.  Tolerances: outputs / losses 1e-5 relative; gradients
    2e-2 in global relative L2 for the same reason as the fused / unfused BatchNorm builds (ReLU pre-activations that
    lie within rounding of zero flip once the decoder's gradient into the backbone moves in its last bits) and
    1e-4 for the parameters of the block itself and of everything downstream of it, where no ReLU intervenes."""
    from dspnet_amd import engine as E

    def run(commute):
        E.COMMUTE_RESIZE_CONV = commute
        try:
            net, solver, data, lab, seg = make(2, 256, 256)
        finally:
            E.COMMUTE_RESIZE_CONV = True
        solver.forward(); solver.backward(); torch.cuda.synchronize()
        m = MultiBoxMetric(); m.update(net)
        return net, dict(zip(*m.get()))

    net_c, loss_c = run(True)
    net_d, loss_d = run(False)
    assert any(isinstance(n, E.BilinearConcatConv) for n in net_c.g.nodes)
    assert not any(isinstance(n, E.BilinearConcatConv) for n in net_d.g.nodes)
    assert [p.name for p in net_c.g.param_order] == [p.name for p in net_d.g.param_order]
    torch.testing.assert_close(net_c.g.arena, net_d.g.arena, rtol=0, atol=0)       # same initial parameters
    a, b = net_c.seg_out.prob.data, net_d.seg_out.prob.data
    assert float((a - b).abs().max()) <= 1e-5
    for k in loss_c:
        assert abs(loss_c[k] - loss_d[k]) <= 1e-5 * abs(loss_d[k]) + 1e-9, (k, loss_c[k], loss_d[k])
    gc = {p.name: p.grad for p in net_c.g.param_order}
    num = den = 0.0
    for p in net_d.g.param_order:
        x, y = gc[p.name].double(), p.grad.double()
        num += float(((x - y) ** 2).sum()); den += float((y ** 2).sum())
        if p.name.startswith(("score3_conv", "score4_conv")):
            rel = float(((x - y) ** 2).sum() ** 0.5 / max(float((y ** 2).sum() ** 0.5), 1e-30))
            assert rel < 1e-4, (p.name, rel)
    assert (num / den) ** 0.5 < 2e-2


def test_detector_batch_64_det_out_is_the_oracles(gpu_device):
    """BASELINE.json configs[4] at its own size: the inference-only Detector at batch 64, 512x512.  det_out must be, bit for
    bit, what the C restatement of the reference's MultiBoxDetection CPU kernel (operator/multibox_detection.cc:44-169)
    returns on the device's own cls_prob / loc_preds / anchors -- all 64 samples, ~6000 valid rows each (random weights:
    nearly every anchor passes the score threshold, the worst case for the sort and the suppression scan)."""
    from dspnet_amd.detect.multitask_detector import Detector
    dev = torch.device("cuda", 0)
    B = 64
    det = Detector("resnet-50", 512, num_classes=8, batch_size=B, device=dev)
    data = torch.from_numpy(synthetic.images(B, 512, 512, synthetic.rng(233))).to(dev)
    out, seg = det.forward(data)
    node = det.net.det
    prob = node.cls_prob.data.cpu().numpy()
    loc = node.loc_preds.data.cpu().numpy()
    anchors = node.anchors.cpu().numpy()
    assert prob.shape == (B, 9, 6132) and loc.shape == (B, 6132 * 5) and tuple(out.shape) == (B, 6132, 7)
    exp = om.multibox_detection(prob, loc, anchors, nms_threshold=.5, force_suppress=False, nms_topk=400)
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got, exp)
    valid = (got[:, :, 0] >= 0).sum(1)
    assert valid.min() > 0 and (exp[:, :, 0] >= 0).sum() == valid.sum()
    # a second forward on the same batch: the side stream's hand-off is race free and the result deterministic
    out2, _ = det.forward()
    assert torch.equal(out2, out)
    assert tuple(seg.shape[:3]) == (B, 128, 128) and torch.isfinite(seg).all()


@pytest.mark.gpu
@pytest.mark.parametrize("network", ["resnet-50", "vgg16_reduced", "inceptionv3"])
def test_side_stream_detection_branch_gives_the_bits_of_the_main_stream_schedule(gpu_device, network):
    """Round 4: the detection branch (SSD extra layers, heads, packing, MultiBoxTarget) runs its forward, and the part of its
    backward whose gradients stay inside it, on a second stream beside the segmentation decoder (Graph.set_side_segment /
    set_side_backward).  Only the STREAMS change -- every accumulation keeps its order -- so parameters and gradients after
    each of several SGD steps must equal, bit for bit, those of the same graph built with the whole schedule on the main stream
    (DSPN_TARGET_SIDE = DSPN_DET_SIDE = 0), and a second run of the side-stream schedule must reproduce the first.
    Round 5: all three presets -- for vgg16_reduced / inceptionv3 the decoder reads a tensor written INSIDE the branch (conv_feat
    is the first SSD extra layer), which the stream ordering derived in Graph._plan_side_sync has to cover (structure:
    tests/test_side_plan.py) -- and the steps are issued back to back, so that the host runs ahead of the device as it
    does in training."""
    import os
    from dspnet_amd import synthetic
    from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
    from dspnet_amd.train.solver import MultiTaskSolver
    B, S, steps = (4, 256, 4) if network == "resnet-50" else (2, 512, 3)     # (the vgg16_reduced preset's seventh map needs 512)

    def run(side):
        saved = {k: os.environ.get(k) for k in ("DSPN_TARGET_SIDE", "DSPN_DET_SIDE", "DSPN_DET_SIDE_BWD")}
        for k in saved:
            os.environ[k] = "1" if side else "0"
        try:
            net = get_multi_symbol_train(network, S, num_classes=8, batch_size=B, device=torch.device("cuda", 0), seed=0)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        assert (net.g.side_segment is not None) == side and (net.g.side_bwd is not None) == side
        g = synthetic.rng(3)
        solver = MultiTaskSolver(net)
        solver.set_batch(torch.from_numpy(synthetic.images(B, S, S, g)).cuda(),
                         torch.from_numpy(synthetic.det_labels(B, gen=g, height=S, width=S, first_empty=False)).cuda(),
                         torch.from_numpy(synthetic.seg_labels(B, S, S, gen=g)).cuda())
        out = []
        for _ in range(steps):
            solver.step()       # (no synchronize: the clones are ordered behind the step on the current stream)
            out.append((net.g.arena.clone(), net.g.grad_arena.clone()))
        torch.cuda.synchronize()
        return out

    on1, off, on2 = run(True), run(False), run(True)
    for k in range(steps):
        for a, b, c in zip(on1[k], off[k], on2[k]):
            assert torch.equal(a, b), "step %d: side-stream schedule differs from the main-stream schedule" % k
            assert torch.equal(a, c), "step %d: the side-stream schedule is not reproducible" % k
