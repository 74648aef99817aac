#!/usr/bin/env python
"""Headline benchmark: multi-task DSPNet training images/sec at 512x512 on N MI355X.

One "step" = forward + backward (+ RCCL gradient all-reduce for N > 1) + SGD-momentum update of the
resnet-50 multi-task graph (SSD det + depth + seg) on one synthetic Cityscapes-shaped batch that is
already resident in HBM.  Workload: resnet-50, 512x512, 8 det classes, 19 seg classes, 32 images per
GPU, fp32 (the only preset the reference's graph builder accepts, SURVEY.md 2.1, and the shape
BASELINE.json's scaling target is quoted on).  Weak scaling: every rank holds a replica and its own
32-image shard; the only collective is the gradient all-reduce.

fp32 here means fp32 tensors, fp32 accumulation and fp32-accurate products.  The default convolution
math (--math f16x2, DSPN_MATH_F32_F16X2, round 3) forms every product from three exact fp16 x fp16
partial products of a two-piece split of both fp32 operands (after a per-tensor power-of-two scale that
puts the operand inside fp16's range) on the fp16 MFMA -- its error against float64 is the fp32 MFMA's
(tests/test_nn_gpu.py::test_split_math_is_as_accurate_as_the_fp32_mfma, and every graph parity test runs
in all three fp32-result modes at one tolerance); --math bf16x3 (six bf16 products, round 2's default) and
--math fp32 (v_mfma_f32_32x32x2_f32) are reported beside the headline under other_configs.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     - the implicit-GEMM convolution family: algorithmic conv FLOPs executed per step / conv
                 kernel time per step, the latter measured live with HIP events on the launch stream
                 over the timed region; peak = the instruction that bounds the mode (f16x2: 16-bit MFMA
                 2500 TFLOP/s / 3 products per multiply-add = 833.3; bf16x3: 2500 / 6 = 416.7; fp32: 157.3; bf16: 2500);
  math_accuracy_check - one backbone layer in the three math modes against float64 (N = 1): the default math's error next
                 to the fp32 MFMA's and the bf16 mode's, measured in this run;
  roofline_ops - the HBM-bound operators SURVEY.md 8(d) names (MultiBoxTarget, MultiBoxDetection, the BatchNorm backward
                 apply, SGD): algorithmic bytes / event-timed kernel time / 8 TB/s, measured after the timed region;
  cpu_baseline - the CPU restatement of the same step (oracle/dspnet_torch.py, fp32, all host cores) on a bounded sample
                 at B = 4 (`value`) and B = 1 (`conv_path`), and the multibox operators alone through the C restatement,
                 one thread and over the batch (`multibox_ops`); rank 0, N = 1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA, dense
# algorithmic (fp32 multiply-add) peak of each math mode: bf16x3 issues six bf16 MFMAs per fp32 multiply-add
MATH_PEAK_TFLOPS = {"fp32": FP32_MATRIX_PEAK_TFLOPS, "bf16": BF16_MATRIX_PEAK_TFLOPS, "bf16x3": BF16_MATRIX_PEAK_TFLOPS / 6.0,
                    "f16x2": BF16_MATRIX_PEAK_TFLOPS / 3.0}    # (the fp16 MFMA has the bf16 MFMA's rate)
MATH_LABEL = {"fp32": "fp32 MFMA", "bf16": "bf16 MFMA (operands rounded to bf16)",
              "bf16x3": "fp32 on the bf16 MFMA (3-piece split, 6 products)",
              "f16x2": "fp32 on the fp16 MFMA (2-piece split after a per-tensor power-of-two scale, 3 products)"}
# committed rocprofv3 --pmc passes of the headline workload per math mode, newest first (roofline.traffic is read from these)
TRAFFIC_PROFILES = {"f16x2": ["r06_f16x2_pmc_conv_family.json", "r05_f16x2_pmc_conv_family.json", "r04_f16x2_pmc_conv_family.json", "r03_f16x2_pmc_conv_family.json"],
                    "bf16x3": ["r06_x3_pmc_conv_family.json", "r05_x3_pmc_conv_family.json", "r04_x3_pmc_conv_family.json", "r03_x3_pmc_conv_family.json", "r02_x3_pmc_conv_family.json"],
                    "fp32": ["r06_fp32_pmc_conv_family.json", "r05_fp32_pmc_conv_family.json", "r04_fp32_pmc_conv_family.json", "r02_fp32_pmc_conv_family.json", "r02_pmc_conv_family.json",
                             "r01_j_pmc_conv_family.json"]}
PROF_STEPS = 2   # steps of the timed region whose convolution launches are bracketed by HIP events (roofline.achieved)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=512, help="image height (and width unless --width is given)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--network", default="resnet-50", choices=["resnet-50", "vgg16_reduced", "inceptionv3"],
                    help="backbone preset; the headline workload is resnet-50 (the other BASELINE.json configs: "
                         "vgg16_reduced --batch 16; inceptionv3 --size 512 --width 1024 --batch 8 --math bf16)")
    ap.add_argument("--math", default="f16x2", choices=["f16x2", "bf16x3", "fp32", "bf16"],
                    help="convolution math on float tensors: f16x2 = fp32 results from three exact fp16 products per "
                         "multiply (two fp16 pieces per operand after a per-tensor power-of-two scale; default), bf16x3 = "
                         "six exact bf16 products per multiply, fp32 = fp32 MFMA, bf16 = operands rounded to bf16")
    ap.add_argument("--store", default="fp32", choices=["fp32", "bf16"],
                    help="storage type of activation tensors and convolution operands in HBM: float32 (default) or "
                         "bfloat16 (the *_bf16 kernels: bf16 MFMA, fp32 accumulate, fp32 master weights)")
    ap.add_argument("--graph", type=int, default=0,
                    help="1: record the step into a HIP graph after the warm-up and replay it (single GPU; no roofline events)")
    ap.add_argument("--reserve-cus", type=int, default=-1,
                    help="CUs the persistent convolution grids leave free for RCCL (dspn_conv_set_reserved_cus).  -1 (default): "
                         "0 at N = 1; at N > 1 switched to 16 after warm-up iff the exposed all-reduce time exceeds 1 ms")
    ap.add_argument("--wide-tiles", type=int, default=0,
                    help="dspn_conv_set_wide_tiles: 0 automatic (default), 1 never (the round-4 128x128 kernels), 2 / 3 / 4 force a shape")
    ap.add_argument("--sustained-steps", type=int, default=-1,
                    help="further steps run AFTER the timed region for the `sustained` block of the line (0: skip; default: 200, "
                         "or 0 for the short experiment runs that pass --no-other-configs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short side measurements of BASELINE.json configs[1] (vgg16_reduced, bs 16), configs[3] "
                         "(inceptionv3 1024x512, bs 8, bf16) and configs[4] (inference p50, bs 64) at N=1")
    ap.add_argument("--cpu-images", type=int, default=2)
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher plumbing only (gloo, no GPU): every rank joins, rank 0 prints {dry_run, n_gpus}")
    ap.add_argument("--mode", choices=["train", "infer"], default="train",
                    help="infer = BASELINE.json configs[4]: forward-only test graph + MultiBoxDetection/NMS, p50 latency")
    return ap.parse_args()


def run_infer(args):
    """configs[4]: Inference-only path, bs=64 512x512, HIP multibox_detection + NMS, p50 latency over
    >= 100 iterations after 10 warm-ups."""
    import numpy as np
    import torch
    from dspnet_amd import synthetic
    from dspnet_amd.detect.multitask_detector import Detector
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B = args.batch if args.batch != 32 else 64
    det = Detector("resnet-50", args.size, num_classes=8, batch_size=B, device=dev)
    data = torch.from_numpy(synthetic.images(B, args.size, args.size, synthetic.rng(233))).to(dev)
    det.net.data.data.copy_(data)
    iters = max(args.steps, 100)
    for _ in range(10):
        det.forward()
    torch.cuda.synchronize()
    lat = []
    for _ in range(iters):
        t0 = time.perf_counter()
        det.forward()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.sort(np.asarray(lat))
    p50 = float(np.percentile(lat, 50))
    print(json.dumps({"metric": "inference p50 latency, multitask detector forward + MultiBoxDetection/NMS", "value": round(p50, 3),
                      "unit": "ms/batch", "n_gpus": 1, "steps": iters, "warmup": 10, "ms_per_step": round(p50, 3),
                      "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                      "data": "synthetic", "p90_ms": round(float(np.percentile(lat, 90)), 3),
                      "images_per_s": round(B / p50 * 1e3, 1),
                      "config": {"workload": "resnet-50 multitask test graph %dx%d, batch %d, random-init weights "
                                             "(nearly all 6132 rows valid: worst case for sort + NMS)" % (args.size, args.size, B)}}))


def host_cores():
    """cores this process may actually use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


HBM_PEAK_TBS = 8.0   # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def cpu_multibox_ops(anchors, batch=32, seconds=1.5):
    """the multibox operators ALONE on the host -- the C restatement of the reference's CPU kernels
    (oracle/multibox_oracle.c) at the bench shape (B = 32, N = 6132 anchors, L = 200 label rows, 8 + 1 classes), timed
    single-threaded (the reference loops over the batch serially, operator/multibox_target.cc:92) and with the batch
    spread over the host cores (one slice of the batch per core from a thread pool; ctypes releases the GIL: what an
    OpenMP `parallel for` over the batch would give)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from dspnet_amd import synthetic
    from oracle import multibox as om
    gen = synthetic.rng(233)
    A = anchors.shape[1]
    lab = synthetic.det_labels(batch, gen=gen, first_empty=False)
    pred = gen.standard_normal((batch, 9, A)).astype(np.float32)
    e = np.exp(pred - pred.max(1, keepdims=True))
    prob = (e / e.sum(1, keepdims=True)).astype(np.float32)
    loc = (0.1 * gen.standard_normal((batch, A * 5))).astype(np.float32)
    cores = host_cores()

    def target(b0, b1):
        om.multibox_target(anchors, lab[b0:b1], pred[b0:b1], negative_mining_ratio=3)

    def detection(b0, b1):
        om.multibox_detection(prob[b0:b1], loc[b0:b1], anchors, nms_topk=400)

    def rate(fn_, parallel):
        pool = ThreadPoolExecutor(cores) if parallel else None
        t0 = time.perf_counter()
        reps = 0
        while True:
            if parallel:     # one contiguous slice of the batch per core, each one C call
                cuts = [batch * i // cores for i in range(cores + 1)]
                list(pool.map(lambda i: fn_(cuts[i], cuts[i + 1]), [i for i in range(cores) if cuts[i] < cuts[i + 1]]))
            else:
                fn_(0, batch)
            reps += 1
            if time.perf_counter() - t0 > seconds:
                break
        dt = time.perf_counter() - t0
        if pool:
            pool.shutdown()
        return round(batch * reps / dt, 1)

    return {"workload": "B = %d, N = %d anchors, L = 200 label rows, 8 + 1 classes; C restatement of the reference's CPU "
                        "kernels (oracle/multibox_oracle.c)" % (batch, A), "unit": "samples/s", "cores": cores,
            "MultiBoxTarget": {"1_thread": rate(target, False), "batch_parallel": rate(target, True)},
            "MultiBoxDetection": {"1_thread": rate(detection, False), "batch_parallel": rate(detection, True)}}


def roofline_ops(net, solver, dev):
    """the HBM-bound operators of the step, each priced against the HBM roof: algorithmic bytes (read
    once + written once) / kernel time / 8 TB/s.  Timed with events on the stream the kernels are launched on (torch's
    current stream: the operator front ends and dspnet_amd.functional launch there), 20 back-to-back calls each, after
    and outside the timed region."""
    import torch
    from dspnet_amd import functional as fn, operator as op, synthetic

    def timed(f, reps=20):
        f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            f()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e-3

    def row(name, nbytes, sec, note):
        gbs = nbytes / sec / 1e9
        return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s",
                "frac": round(gbs / (HBM_PEAK_TBS * 1e3), 4), "algorithmic_bytes": int(nbytes),
                "ms": round(sec * 1e3, 4), "note": note}

    out = {}
    B = net.data.shape[0]
    anchors = net.anchors
    A = anchors.shape[1]
    g = synthetic.rng(7)
    lab = torch.from_numpy(synthetic.det_labels(B, gen=g)).to(dev)
    pred = torch.randn(B, 9, A, device=dev)
    prob = torch.softmax(pred, dim=1).contiguous()
    loc = 0.1 * torch.randn(B, A * 5, device=dev)
    # MultiBoxTarget: reads cls_pred + labels (+ the anchor table once), writes loc_target, loc_mask, cls_target
    tb = B * (9 * A * 4 + 200 * 6 * 4 + A * 11 * 4) + A * 16
    out["MultiBoxTarget"] = row("target", tb, timed(lambda: op.MultiBoxTarget(anchors, lab, pred, negative_mining_ratio=3)),
                                "B = %d; IoUs, column maxima and background keys on the whole chip, the greedy matching / mining on one workgroup per sample (registers + LDS)" % B)
    db = B * (9 * A * 4 + A * 5 * 4 + A * 7 * 4) + A * 16
    det_out = torch.empty(B, A, 7, device=dev)
    out["MultiBoxDetection"] = row("detection", db, timed(lambda: op.MultiBoxDetection(prob, loc, anchors, nms_topk=400, out=det_out)),
                                   "B = %d, nearly every row valid (random scores): worst case for sort + NMS" % B)
    # BatchNorm backward apply on the largest tensor of the graph it runs on (stage 1: 128 x 128 x 256 per image)
    x = torch.randn(B, 128, 128, 256, device=dev)
    dy = torch.randn_like(x)
    dx = torch.empty_like(x)
    C = 256
    gamma = torch.rand(C, device=dev) + 0.5
    beta = torch.randn(C, device=dev)
    mean, rstd, scale, shift = fn.bn_stats(x, 2e-5, gamma, beta)
    tiles = 64
    sums = torch.randn(tiles, 2, C, device=dev)
    out["bn_bwd_apply"] = row("bn_bwd_apply", 3 * 4 * x.numel(),
                              timed(lambda: fn.bn_backward_from_sums(x, scale, shift, dy, mean, rstd, gamma, sums, tiles, relu=True, dx=dx)),
                              "dx = a dy' + c1 x + c0 on a %d x 128 x 128 x 256 tensor (reads x, dy; writes dx), incl. its per-channel finalize launch" % B)
    del x, dy, dx
    gr = net.g
    n = gr.arena.numel()
    out["sgd"] = row("sgd", 5 * 4 * n, timed(lambda: fn.sgd_momentum(gr.arena, gr.grad_arena, gr.mom_arena, 0.0, 0.9, 0.0005, 1.0 / B)),
                     "one launch over the %.1f M-parameter arena: reads w, g, m; writes m, w (lr = 0 here: the weights do not move)" % (n / 1e6))
    return out


def cpu_baseline(size, images, cfg, width=None, seconds=12.0):
    """forward + backward of the same graph with torch CPU ops, fp32, all cores"""
    width = width or size
    import numpy as np
    import torch
    from dspnet_amd import synthetic
    from oracle import dspnet_torch as ot
    cores = host_cores()
    torch.set_num_threads(cores)
    gen = synthetic.rng(233)
    data = synthetic.images(images, size, width, gen)
    lab = synthetic.det_labels(images, gen=gen, height=size, width=width, first_empty=False)
    seg = synthetic.seg_labels(images, size, width, gen=gen)
    values = cpu_baseline.values

    def once():
        ref = ot.forward_loss(values, data, lab, seg, dtype=torch.float32, config=cfg)
        ref["objective"].backward()

    once()                                  # warm-up (allocator, thread pool)
    t0 = time.perf_counter()
    reps = 0
    while True:
        once()
        reps += 1
        if time.perf_counter() - t0 > seconds or reps >= 200:
            break
    dt = time.perf_counter() - t0
    return {"value": round(images * reps / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d x (forward+backward of the same %s multitask graph, %d images %dx%d, fp32 torch-CPU "
                      "ops + C multibox oracle; no optimizer step)" % (reps, cfg["network"], images, size, width)}


def math_accuracy_check(dev):
    """One backbone layer (3x3, 256 -> 256 channels, 8 x 32 x 32) in the three math modes against a float64 CPU convolution
    of the same float32 operands: forward and data gradient, largest error over the tensor's largest entry.  Shows in the
    bench line itself that the default math (bf16x3) is an fp32 evaluation and what the bf16 mode of configs[3] gives up.
    Outside the timed region, rank 0 at N = 1 only."""
    import torch
    import torch.nn.functional as F
    from dspnet_amd import functional as fn
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 256, 32, 32, generator=g)
    w = torch.randn(256, 256, 3, 3, generator=g) / 48.0
    dy = torch.randn(8, 256, 32, 32, generator=g)
    xr = x.double().requires_grad_()
    y_ref = F.conv2d(xr, w.double(), None, 1, 1)
    y_ref.backward(dy.double())
    xd, wd, dyd = (t.permute(0, 2, 3, 1).contiguous().to(dev) for t in (x, w, dy))
    wt = fn.weight_transpose(wd)
    out = {"layer": "3x3 conv 256->256, 8x32x32, K = 2304; max |err| / max |ref| against float64"}
    prev = fn.get_conv_math()
    try:
        for mode in ("f16x2", "bf16x3", "fp32", "bf16"):
            fn.set_conv_math(mode)
            y = fn.conv2d_forward(xd, wd, None, 1, 1, 1)
            dx = fn.conv2d_dgrad(dyd, wt, tuple(xd.shape), 1, 1, 1)
            ey = float((y.permute(0, 3, 1, 2).double().cpu() - y_ref.detach()).abs().max() / y_ref.detach().abs().max())
            ed = float((dx.permute(0, 3, 1, 2).double().cpu() - xr.grad).abs().max() / xr.grad.abs().max())
            out[mode] = {"forward": float("%.3g" % ey), "data_gradient": float("%.3g" % ed)}
    finally:
        fn.set_conv_math(prev)
    return out


class GpuSensors:
    """engine clock (MHz) and board power (W) of one GPU from the amdgpu driver's sysfs files -- read directly (no child
    process: a process that has initialised the GPU must not exec on this pool, and rocm-smi is a program).  Every field is
    optional: a box that hides the files gives an empty summary."""

    def __init__(self, index):
        import glob
        self.freq, self.power, self.dpm = None, None, None
        # the sysfs node of THIS device: matched by PCI address (a box shows every GPU of the node under /sys/class/drm, the
        # process sees one of them as cuda:0)
        self.pci = None
        try:
            buf = ctypes.create_string_buffer(64)
            if ctypes.CDLL("libamdhip64.so").hipDeviceGetPCIBusId(buf, 64, int(index)) == 0:
                self.pci = buf.value.decode().lower()
        except OSError:
            pass
        cards = [c for c in glob.glob("/sys/class/drm/card[0-9]*/device")
                 if self.pci and os.path.basename(os.path.realpath(c)).lower() == self.pci]
        if not cards:
            return
        dev = cards[0]
        for h in sorted(glob.glob(os.path.join(dev, "hwmon/hwmon*"))):
            for name in ("freq1_input",):
                if self.freq is None and os.path.exists(os.path.join(h, name)):
                    self.freq = os.path.join(h, name)
            for name in ("power1_average", "power1_input"):
                if self.power is None and os.path.exists(os.path.join(h, name)):
                    self.power = os.path.join(h, name)
        if os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
            self.dpm = os.path.join(dev, "pp_dpm_sclk")

    def read(self):
        out = {}
        try:
            if self.freq:
                out["sclk_mhz"] = int(open(self.freq).read()) / 1e6
            elif self.dpm:
                for l in open(self.dpm).read().splitlines():
                    if l.rstrip().endswith("*"):
                        out["sclk_mhz"] = float(l.split(":")[1].strip().lower().split("mhz")[0])
        except (OSError, ValueError, IndexError):
            pass
        try:
            if self.power:
                out["power_w"] = int(open(self.power).read()) / 1e6
        except (OSError, ValueError):
            pass
        return out

    @staticmethod
    def summarise(samples):
        out = {}
        for key in ("sclk_mhz", "power_w"):
            v = sorted(x[key] for x in samples if key in x)
            if v:
                out[key] = {"min": round(v[0], 1), "median": round(v[len(v) // 2], 1), "max": round(v[-1], 1), "samples": len(v)}
        if not out:
            out["sensors"] = "not readable from this process (no sysfs clock / power files for this device's PCI address)"
        return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start one process per GPU through torch.distributed.run and
    pass their exit status on.  Runs BEFORE anything in this process touches the GPU (no torch.cuda call, no HIP
    library loaded): the children are ordinary child processes, never an exec of a process that initialised HIP."""
    import socket
    import subprocess
    with socket.socket() as so:                       # a free rendezvous port on the loopback interface
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("NCCL_DEBUG", "WARN")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the launcher and its ranks form a process group of their own: when a rank fails (the elastic agent then stops the
    # others and exits non-zero) or this parent is interrupted, whatever is left of exactly THAT group is ended -- fresh
    # children only, never a re-exec of a process that touched the GPU, never a kill by pattern
    import signal
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        rc = child.wait()
    except BaseException:
        rc = 1
        raise
    finally:
        if rc != 0:
            try:
                os.killpg(child.pid, signal.SIGTERM)
                time.sleep(2.0)
                os.killpg(child.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
    if rc != 0:
        sys.stderr.write("bench.py: a rank failed (torch.distributed.run exit status %d); no result line\n" % rc)
    return rc if rc != 0 else 0


def check_distinct_devices(dist, identity, rank, world):
    """first-contact hygiene of an N > 1 run: every rank reports the device it sits on; N ranks must hold N DISTINCT devices
    (two ranks on one GPU would still produce a plausible-looking line at half the throughput).  -> the list of identities"""
    ids = [None] * world
    dist.all_gather_object(ids, identity)
    if len(set(ids)) != world:
        sys.exit("bench.py: %d ranks but only %d distinct devices: %s" % (world, len(set(ids)), ids))
    return ids


_WGRAD_SIDE_DEFAULT = None


def serial_schedule(on):
    """everything on the step's stream (engine.WGRAD_SIDE = 0) while `on`, the engine's own schedule otherwise"""
    global _WGRAD_SIDE_DEFAULT
    from dspnet_amd import engine as E
    if _WGRAD_SIDE_DEFAULT is None:
        _WGRAD_SIDE_DEFAULT = E.WGRAD_SIDE
    E.WGRAD_SIDE = 0 if on else _WGRAD_SIDE_DEFAULT


def instrument_step(lib, on):
    """HIP events around every convolution launch (dspn_profile_enable) for the steps that feed `roofline`.  Round 6: the
    engine runs the weight gradients on a stream of their own, beside the data-gradient chain (dspnet_amd/engine.py:
    WGRAD_SIDE) -- a kernel timed while another one shares the chip measures the pair, so the instrumented steps run
    everything on the step's stream, as `rocprofv3 --kernel-trace` of a DSPN_WGRAD_SIDE=0 run sees the kernels."""
    serial_schedule(on)
    lib.dspn_profile_enable(1 if on else 0)


def conv_family_roofline(lib, steps, flops_step, flops_3x_step, math, step_s, traffic=None, traffic_source=None):
    """roofline block of the implicit-GEMM convolution family from the HIP events the library recorded on the launch
    stream around every conv launch since dspn_profile_enable(1).  `frac` = executed multiply-adds / conv kernel time /
    peak; `frac_end_to_end_3x` = SURVEY.md 8(d)'s own definition, images/s x (3 x direct-conv forward FLOPs) / peak,
    i.e. the whole step (every non-conv kernel and launch gap included) priced against the matrix peak."""
    tot, cnt = ctypes.c_double(), ctypes.c_longlong()
    lib.dspn_profile_collect(0, ctypes.byref(tot), ctypes.byref(cnt))
    nt_ms, nt_n = tot.value, cnt.value
    lib.dspn_profile_collect(1, ctypes.byref(tot), ctypes.byref(cnt))
    wg_ms, wg_n = tot.value, cnt.value
    conv_s = (nt_ms + wg_ms) / 1e3 / steps
    if conv_s <= 0:
        return None
    ach = flops_step / conv_s / 1e12
    peak = MATH_PEAK_TFLOPS[math]
    out = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
           "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
           "frac_end_to_end_3x": round(flops_3x_step / step_s / 1e12 / peak, 4),
           "kernel": "conv_nt / conv_ntw / conv_ntv / conv_stem kernels (fwd+dgrad) + conv_wgrad_kernel: implicit-GEMM family, %s" % MATH_LABEL[math],
           "launches_per_step": (nt_n + wg_n) // steps,
           "avg_launch_us": round((nt_ms + wg_ms) * 1e3 / max(1, nt_n + wg_n), 2),
           "conv_ms_per_step": round(conv_s * 1e3, 3),
           "nt_ms_per_step": round(nt_ms / steps, 3), "wgrad_ms_per_step": round(wg_ms / steps, 3),
           "algorithmic_gflop_per_step": round(flops_step / 1e9, 1),
           "share_of_step_time": round(conv_s / step_s, 3)}
    if math in ("bf16x3", "f16x2"):
        # the same work priced two other ways: against the fp32 MFMA it replaces, and as issued 16-bit MFMA flops
        k = 6.0 if math == "bf16x3" else 3.0
        out["peak_note"] = "2500 TFLOP/s 16-bit MFMA / %d products per fp32 multiply-add" % k
        out["achieved_vs_fp32_mfma_peak"] = round(ach / FP32_MATRIX_PEAK_TFLOPS, 4)
        out["issued_16bit_mfma_tflops"] = round(k * ach, 1)
        if math == "f16x2":   # the bound round 2's frac (0.29) and VERDICT r02's target (>= 0.40) were quoted against
            out["frac_of_bf16x3_bound"] = round(ach / MATH_PEAK_TFLOPS["bf16x3"], 4)
    return out


def conv_nodes(net):
    from dspnet_amd import engine as E
    return [n for n in net.g.nodes if isinstance(n, (E.Conv, E.Deconv4x4s2, E.BilinearConcatConv))]


def side_train(network, H, W, B, math, steps, warmup, dev, store="fp32"):
    """a short side measurement of another BASELINE.json training config on this GPU (never part of `value`)"""
    import torch
    from dspnet_amd import _lib, functional as fn, synthetic
    from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
    from dspnet_amd.train.solver import MultiTaskSolver
    lib = _lib.lib()
    prev_math, prev_dtype = fn.get_conv_math(), fn.ACT_DTYPE
    fn.set_conv_math(math)
    fn.set_activation_dtype(store)
    try:
        net = get_multi_symbol_train(network, (3, H, W), num_classes=8, batch_size=B, device=dev, seed=0)
        fn.set_activation_dtype("fp32")
        solver = MultiTaskSolver(net)
        g = synthetic.rng(233)
        solver.set_batch(torch.from_numpy(synthetic.images(B, H, W, g)).to(dev),
                         torch.from_numpy(synthetic.det_labels(B, gen=g, height=H, width=W)).to(dev),
                         torch.from_numpy(synthetic.seg_labels(B, H, W, gen=g)).to(dev))
        convs = conv_nodes(net)
        flops_step = sum(n.flops_fwd + n.flops_bwd for n in convs)
        flops_3x = 3.0 * sum(getattr(n, "flops_direct", n.flops_fwd) for n in convs)
        for w in range(warmup):
            serial_schedule(w == warmup - 1)
            solver.step()
        serial_schedule(False)
        torch.cuda.synchronize()
        ps = min(PROF_STEPS, steps)
        t0 = time.perf_counter()
        for i in range(steps):
            instrument_step(lib, i < ps)
            solver.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        instrument_step(lib, False)
        return {"workload": "%s multitask (det+depth+seg) %dx%d (HxW), bs %d, %s, forward+backward+SGD, N=%d anchors"
                            % (network, H, W, B, MATH_LABEL[math] if math != "bf16" else ("bf16 MFMA convs" + (
                                ", bf16 tensors in HBM" if store == "bf16" else ", fp32 tensors in HBM")), net.anchors.shape[1]),
                "images_per_s": round(B * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 3),
                "steps": steps, "warmup": warmup, "dtype": "bf16" if math == "bf16" else "f32",
                "roofline": conv_family_roofline(lib, ps, flops_step, flops_3x, math, dt / steps)}
    finally:
        fn.set_conv_math(prev_math)
        fn.set_activation_dtype(prev_dtype)


def side_infer(B, size, iters, warmup, dev):
    """BASELINE.json configs[4]: inference-only detector path, p50 latency of forward + MultiBoxDetection/NMS"""
    import numpy as np
    import torch
    from dspnet_amd import _lib, functional as fn, synthetic
    from dspnet_amd.detect.multitask_detector import Detector
    lib = _lib.lib()
    det = Detector("resnet-50", size, num_classes=8, batch_size=B, device=dev)
    det.net.data.data.copy_(torch.from_numpy(synthetic.images(B, size, size, synthetic.rng(233))).to(dev))
    convs = conv_nodes(det.net)
    flops_fwd = sum(n.flops_fwd for n in convs)
    flops_direct = sum(getattr(n, "flops_direct", n.flops_fwd) for n in convs)
    for _ in range(warmup):
        det.forward()
    torch.cuda.synchronize()
    # convolution-family time from HIP events on a few instrumented iterations FIRST (the events cost ~1 ms per batch);
    # the latency percentiles come from uninstrumented iterations
    ps = min(10, iters)
    lib.dspn_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(ps):
        det.forward()
    torch.cuda.synchronize()
    inst_s = (time.perf_counter() - t0) / ps
    lib.dspn_profile_enable(0)
    lat = []
    for _ in range(iters):
        t0 = time.perf_counter()
        det.forward()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.sort(np.asarray(lat))
    p50 = float(np.percentile(lat, 50))
    return {"workload": "resnet-50 multitask test graph %dx%d, batch %d, fp32, forward + MultiBoxDetection/NMS, random-init "
                        "weights (nearly all 6132 rows valid: worst case for sort + NMS)" % (size, size, B),
            "p50_ms_per_batch": round(p50, 3), "p90_ms_per_batch": round(float(np.percentile(lat, 90)), 3),
            "images_per_s": round(B / p50 * 1e3, 1), "iterations": iters, "warmup": warmup, "dtype": "f32",
            "roofline": conv_family_roofline(lib, ps, flops_fwd, flops_direct, fn.get_conv_math(), inst_s)}


def dry_run(args):
    """--dry-run: the launcher / rendezvous / one-JSON-line plumbing of an N-rank run without a GPU (gloo): every rank
    joins the group, a sum all-reduce of (rank + 1) must give N (N + 1) / 2, rank 0 prints the line."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    line = {"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup}
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        # the distinct-device check of the real run, on pretended devices (LOCAL_RANK; DSPN_DRY_DEVICE pins every rank to one
        # index: the failure the check exists for)
        local = os.environ.get("DSPN_DRY_DEVICE", os.environ.get("LOCAL_RANK", str(rank)))
        sys.stderr.write("[bench rank %d/%d] device %s (dry run), buckets n/a\n" % (rank, world, local))
        check_distinct_devices(dist, "dry-device-" + local, rank, world)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        dist.barrier()
        assert float(t.item()) == world * (world + 1) / 2
        # the per-rank fields of the real N > 1 line, reduced the same way (rank r pretends to 10 + r ms per step and r / 10 ms
        # of exposed all-reduce): min / max over ranks of the step time, max of the exposure, the group's own size
        dt = torch.tensor([10.0 + rank], dtype=torch.float64)
        tmax, tmin, texp = dt.clone(), dt.clone(), torch.tensor([rank / 10.0], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(texp, op=dist.ReduceOp.MAX)
        line.update({"ms_per_step_min_over_ranks": float(tmin.item()), "ms_per_step_max_over_ranks": float(tmax.item()),
                     "allreduce_exposed_ms_max_over_ranks": float(texp.item()), "dist_world_size": dist.get_world_size(),
                     "rccl_version": "gloo (dry run)", "reserved_cus": max(args.reserve_cus, 0)})
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.dry_run:
        return dry_run(args)
    if args.mode == "infer":
        return run_infer(args)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and os.environ.get("DSPN_FORCE_DIST") != "1":
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # DSPN_FORCE_DIST=1 runs the RCCL code path (init, bucketed async all-reduce, barrier) even at
    # world_size 1, so that it can be exercised on a single-GPU box
    use_dist = world > 1 or os.environ.get("DSPN_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_DEBUG", "WARN")       # no RCCL version banner on stdout next to the JSON line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL prints a five-line version banner on STDOUT when the communicator is created: keep stdout for the one JSON
        # line by pointing fd 1 at stderr while the group comes up (init + a first collective), then restoring it
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
            torch.cuda.set_device(local)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            try:
                ctypes.CDLL(None).fflush(None)      # the banner sits in libc's stdout buffer: push it out while fd 1 is stderr
            except OSError:
                pass
            os.dup2(saved, 1)
            os.close(saved)
        assert dist.get_world_size() == world
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from dspnet_amd import _lib, synthetic
    from dspnet_amd.symbol.multitask_symbol_factory import get_config, get_multi_symbol_train
    from dspnet_amd.train.solver import MultiTaskSolver

    B, S = args.batch, args.size
    Wd = args.width or S
    from dspnet_amd import functional as fn
    if args.store == "bf16":
        args.math = "bf16"
    fn.set_conv_math(args.math)
    fn.set_activation_dtype(args.store)
    net = get_multi_symbol_train(args.network, (3, S, Wd), num_classes=8, batch_size=B, device=dev, seed=0)
    fn.set_activation_dtype("fp32")
    solver = MultiTaskSolver(net, process_group=None, world_size=world, force_reducer=use_dist)
    gen = synthetic.rng(233 + rank)
    solver.set_batch(torch.from_numpy(synthetic.images(B, S, Wd, gen)).to(dev),
                     torch.from_numpy(synthetic.det_labels(B, gen=gen, height=S, width=Wd)).to(dev),
                     torch.from_numpy(synthetic.seg_labels(B, S, Wd, gen=gen)).to(dev))
    convs = conv_nodes(net)
    # executed multiply-adds: score3_conv is evaluated per pyramid level before the resize (engine.BilinearConcatConv,
    # an exact linear identity) and is counted with what it executes, NOT with the 9.3x larger direct-form count, so
    # the MFMA fraction below is not inflated by the saving; the 3x convention uses the direct count of every layer
    flops_step = sum(n.flops_fwd + n.flops_bwd for n in convs)          # executed
    flops_3x = 3.0 * sum(getattr(n, "flops_direct", n.flops_fwd) for n in convs)   # SURVEY.md 8(d) convention

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    lib = _lib.lib()
    reserved = max(args.reserve_cus, 0)
    if lib.dspn_conv_set_reserved_cus(reserved) != 0:
        sys.exit("bench.py: --reserve-cus %d rejected by the library (0 .. 128)" % reserved)
    if lib.dspn_conv_set_wide_tiles(args.wide_tiles) != 0:
        sys.exit("bench.py: --wide-tiles %d rejected by the library (0 .. 4)" % args.wide_tiles)
    if solver.reducer is not None:
        solver.reducer.measure_exposed = True          # event records around the collective waits (and per bucket)
    if use_dist:
        # first contact with a multi-GPU node: who sits where, before any step (stderr; stdout carries the one JSON line)
        props = torch.cuda.get_device_properties(local)
        ident = "%s/%s" % (os.uname().nodename, getattr(props, "uuid", None) or getattr(props, "pci_bus_id", None) or local)
        sys.stderr.write("[bench rank %d/%d] cuda:%d %s (%s), %d gradient buckets of <= %.0f MB, reducer %s\n"
                         % (rank, world, local, torch.cuda.get_device_name(local), ident, len(solver.buckets),
                            max((hi - lo) * 4 / 2 ** 20 for lo, hi, _ in solver.buckets),
                            "on" if solver.reducer is not None else "off"))
        sys.stderr.flush()
        assert dist.get_world_size() == world
        check_distinct_devices(dist, ident, rank, world)
    for w in range(args.warmup):
        # (the last warm-up step runs the schedule of the instrumented steps -- everything on the step's stream -- so that
        # nothing of that path runs for the first time inside the timed region)
        serial_schedule(w == args.warmup - 1 and not args.no_roofline)
        solver.step()
    serial_schedule(False)
    sync()
    warm_exposed = solver.reducer.exposed_ms() if solver.reducer is not None else None
    if solver.reducer is not None:
        solver.reducer.bucket_latency_ms()             # (forget the warm-up's)
    if args.reserve_cus < 0 and world > 1 and warm_exposed is not None:
        # the persistent convolution kernels own every CU; if the warm-up shows the collectives are NOT hidden behind
        # backward, give RCCL 16 CUs for the timed region.  Every rank must take the same decision: the largest exposure
        t = torch.tensor([warm_exposed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if float(t.item()) > 1.0:
            reserved = 16
            if lib.dspn_conv_set_reserved_cus(reserved) != 0:
                sys.exit("bench.py: dspn_conv_set_reserved_cus(16) failed")
            for _ in range(2):
                solver.step()
            sync()
            solver.reducer.exposed_ms(); solver.reducer.bucket_latency_ms()
    graphed = bool(args.graph) and solver.capture(warmup=0)
    prof = not args.no_roofline and not graphed
    # HIP events around every convolution launch cost ~2.5 ms of a 60 ms step (measured: 517 vs 539 images/s with every
    # step instrumented), so only PROF_STEPS steps of the timed region carry them; the other steps run as in production
    prof_steps = min(PROF_STEPS, args.steps) if prof else 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == 0 and prof_steps:
            instrument_step(lib, True)
        if i == prof_steps:
            instrument_step(lib, False)
        solver.step()
    sync()
    dt = time.perf_counter() - t0
    instrument_step(lib, False)
    exposed_ms = solver.reducer.exposed_ms() if solver.reducer is not None else None
    bucket_ms = solver.reducer.bucket_latency_ms() if solver.reducer is not None else None
    n_buckets = len(solver.buckets)
    # OUTSIDE the timed region (never part of `value`): the same step for SUSTAINED_STEPS further steps -- the headline's K
    # steps last under a second and say nothing about the clock the chip settles to -- with the engine clock and board power
    # sampled from sysfs while they run, where the box lets this process read them
    sustained = None
    if args.sustained_steps < 0:
        args.sustained_steps = 0 if args.no_other_configs else 200
    if args.sustained_steps > 0:
        if solver.reducer is not None:
            solver.reducer.measure_exposed = False
        sensors = GpuSensors(local)
        samples = []
        sync()
        ts = time.perf_counter()
        for i in range(args.sustained_steps):
            solver.step()
            if i % 25 == 24:
                samples.append(sensors.read())
        sync()
        dts = time.perf_counter() - ts
        tsus = torch.tensor([dts], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(tsus, op=dist.ReduceOp.MAX)
        dts = float(tsus.item())
        sustained = {"value": round(world * B * args.sustained_steps / dts, 2), "unit": "images/s", "steps": args.sustained_steps,
                     "ms_per_step": round(dts / args.sustained_steps * 1e3, 3), "seconds": round(dts, 2),
                     "note": "the same step, run for this many further steps after the timed region (not part of `value`)"}
        sustained.update(GpuSensors.summarise(samples))
        if sensors.pci and ("sclk_mhz" in sustained or "power_w" in sustained):
            sustained["sensor_device"] = sensors.pci
    rerecorded = solver.graph_rerecorded
    range_rep = net.g.range_report()     # (one device -> host copy, after the timed region)
    dt_rank = dt
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    tmin = t.clone()
    texp = torch.tensor([exposed_ms if exposed_ms is not None else 0.0], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(texp, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    dt_min = float(tmin.item())
    world_dist = dist.get_world_size() if use_dist else 1
    rccl_version = None
    if use_dist:
        try:
            rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            rccl_version = "unknown"
    exposed_max = float(texp.item()) if exposed_ms is not None else None

    roofline = None
    if prof:
        # HBM bytes per conv launch: rocprofv3 --pmc cannot run inside bench.py, so this figure is OFFLINE -- read from
        # the committed PMC passes of the headline workload named in traffic_source -- not a measurement of this run
        traffic = tsrc = None
        headline = (args.network, S, Wd, B, args.store) == ("resnet-50", 512, 512, 32, "fp32")
        if headline:
            for name in TRAFFIC_PROFILES.get(args.math, []):
                try:
                    traffic = round(json.load(open(os.path.join(ROOT, "profiles", name)))["hbm_bytes_per_launch"])
                    tsrc = "offline: profiles/" + name
                    break
                except (OSError, KeyError, ValueError):
                    pass
        roofline = conv_family_roofline(lib, prof_steps, flops_step, flops_3x, args.math, dt / args.steps, traffic, tsrc)
        if roofline is not None:
            roofline["instrumented_steps"] = "%d of the %d timed steps" % (prof_steps, args.steps)
            if _WGRAD_SIDE_DEFAULT:
                roofline["instrumented_schedule"] = ("serial: in the instrumented steps every kernel runs alone on the step's stream "
                                                     "(DSPN_WGRAD_SIDE=0); the other timed steps run the weight gradients on a second "
                                                     "stream beside the data-gradient chain, where a per-launch duration would measure "
                                                     "the pair of kernels sharing the chip")

    ops_roofline = None
    if rank == 0 and world == 1 and not args.no_roofline and args.store == "fp32":
        try:
            ops_roofline = roofline_ops(net, solver, dev)
        except Exception as e:          # never costs the headline line
            ops_roofline = {"error": str(e)[:200]}

    # the other BASELINE.json configs beside the headline workload: short side measurements OUTSIDE the timed region
    # above (never part of `value`), N=1 only, each with its own roofline block
    other = None
    headline = (args.network, S, Wd, B, args.math, args.store) == ("resnet-50", 512, 512, 32, "f16x2", "fp32")
    if rank == 0 and world == 1 and headline and not args.no_other_configs and not args.no_cpu_baseline:
        del solver
        other = {}
        for key, f in (("headline shape, fp32 MFMA", lambda: side_train("resnet-50", 512, 512, 32, "fp32", 6, 2, dev)),
                       ("headline shape, bf16x3 (round-2 default)", lambda: side_train("resnet-50", 512, 512, 32, "bf16x3", 6, 2, dev)),
                       ("configs[1]", lambda: side_train("vgg16_reduced", 512, 512, 16, "f16x2", 5, 2, dev)),
                       ("configs[3]", lambda: side_train("inceptionv3", 512, 1024, 8, "bf16", 8, 3, dev, store="bf16")),
                       ("headline shape, bf16 tensors", lambda: side_train("resnet-50", 512, 512, 32, "bf16", 8, 3, dev, store="bf16")),
                       ("configs[4]", lambda: side_infer(64, 512, 100, 10, dev))):
            try:
                other[key] = f()
            except Exception as e:          # a side measurement must never cost the headline line
                other[key] = {"error": str(e)[:200]}
            import gc
            gc.collect()                    # (graphs hold reference cycles: without this the previous one's ~25 GB stay cached)
            torch.cuda.empty_cache()

    if rank == 0:
        cfg = get_config(args.network, S)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import dspnet_torch as ot
            cpu_baseline.values = ot.export_params(net.g)
            # the convolution path (whole graph, torch-CPU ops) at B = 1 and B = 4; `value` is the
            # B = 4 figure.  8(d)(i): the multibox operators alone (C restatement), serial and over the batch.
            cpu = cpu_baseline(S, 4 if args.cpu_images == 2 else args.cpu_images, cfg, Wd, seconds=8.0)
            try:
                b1 = cpu_baseline(S, 1, cfg, Wd, seconds=5.0)
                cpu["conv_path"] = {"B=1": {"value": b1["value"], "unit": "images/s", "sample": b1["sample"]},
                                    "B=%d" % (4 if args.cpu_images == 2 else args.cpu_images):
                                        {"value": cpu["value"], "unit": "images/s", "sample": cpu["sample"]}}
                cpu["multibox_ops"] = cpu_multibox_ops(net.anchors.detach().cpu().numpy())
            except Exception as e:      # never costs the headline line
                cpu["split_error"] = str(e)[:200]
        line = {
            "metric": "training images/sec at %dx%d multitask" % (S, Wd),
            "value": round(world * B * args.steps / dt, 2), "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.math == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "%s multitask (det+depth+seg) %dx%d, 8 det classes, 19 seg classes, "
                                   "N=%d anchors, forward+backward+SGD%s" % (args.network, S, Wd, net.anchors.shape[1],
                                                                            ", bf16 tensors in HBM" if args.store == "bf16" else ""),
                       "conv_math": MATH_LABEL[args.math],
                       "conv_tiles": {0: "automatic (plane-fed two-piece layers on the wide family: 64x64 outputs per wave, "
                                         "operands global -> LDS directly)", 1: "128x128 conv_nt_kernel only (the round-4 schedule)",
                                      2: "forced 256x128", 3: "forced 128x256", 4: "forced 128x128 on four waves"}[args.wide_tiles],
                       "batch_per_gpu": B, "global_batch": world * B, "parallelism": "dp%d" % world,
                       "train_gflop_per_image_3x_convention": round(flops_3x / B / 1e9, 2),
                       "train_gflop_per_image_executed": round(flops_step / B / 1e9, 2)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if exposed_ms is not None:
            # what rank 0's compute stream waited for the gradient all-reduce after the last backward kernel (mean per step):
            # the part of the bucketed collectives that backward did NOT hide
            line["allreduce_exposed_ms"] = round(exposed_ms, 4)
            line["allreduce_buckets"] = n_buckets
            # max over ranks, and what the step saw per bucket (issue -> completion, release order, this rank)
            line["allreduce_exposed_ms_max_over_ranks"] = round(exposed_max, 4)
            if bucket_ms is not None:
                line["allreduce_bucket_latency_ms"] = [round(v, 3) for v in bucket_ms]
        if use_dist:
            # per-rank skew of the timed region, the group as torch.distributed sees it, and the collective library
            line["ms_per_step_min_over_ranks"] = round(dt_min / args.steps * 1e3, 3)
            line["ms_per_step_max_over_ranks"] = round(dt / args.steps * 1e3, 3)
            line["dist_world_size"] = world_dist
            line["rccl_version"] = rccl_version
        line["reserved_cus"] = reserved
        if sustained is not None:
            line["sustained"] = sustained
        if graphed:
            line["config"]["hip_graph"] = {"replayed": True, "range_guard": "spans polled every %d replays; a changed decision "
                                           "drops the recording (re-recorded %d times)" % (net.g.GUARD_PERIOD, rerecorded)}
        if args.math == "f16x2" and args.store == "fp32":
            # range monitor of the two-piece math (Graph.range_report): convolution inputs whose per-channel magnitudes were
            # seen by a BatchNorm finalize, how many of them span more than 2^16, and the widest span in bits
            line["config"]["f16x2_range_monitor"] = {"tensors": range_rep[0], "wider_than_2^16": range_rep[1],
                                                     "widest_span_bits": round(range_rep[2], 1)}
            # round 5: the guard that acts on it (engine.Graph._update_guard): convolutions whose operand spans exceeded 2^16 in
            # the previous pass run in the three-piece bf16 math; 0 calls expected on synthetic data
            gconv, gcalls, gslots = net.g.guard_report()
            line["config"]["f16x2_fallback_calls"] = gcalls
            line["config"]["f16x2_range_guard"] = {"enabled": bool(net.g.guard["enabled"]), "convolutions_on_fallback": gconv,
                                                   "slots_wider_than_2^16_last_pass": gslots,
                                                   "fallback_calls_since_start": net.g.guard["calls_total"]}
        # round 4: which part of the step runs beside the main stream (engine.Graph.set_side_segment / set_side_backward).  While
        # it does, the convolution family's kernels share the chip with it, so `roofline.achieved` -- work / the HIP-event
        # brackets of those kernels -- is lower than with everything on one stream (DSPN_DET_SIDE=0), though the step is shorter
        gsched = net.g
        line["config"]["side_stream_schedule"] = {
            "detection_branch_forward": gsched.side_segment is not None,
            "detection_branch_backward_part": bool(gsched.side_bwd is not None and gsched.side_bwd.get("active", False)),
            "nodes_on_side_stream": (0 if gsched.side_segment is None else gsched.side_segment[1] - gsched.side_segment[0] + 1,
                                     0 if gsched.side_bwd is None else len(gsched.side_bwd["side"])),
            # round 6: the weight gradients (and their slab sums) on a stream of their own beside the data-gradient chain
            "weight_gradients_beside": bool(__import__("dspnet_amd.engine", fromlist=["WGRAD_SIDE"]).WGRAD_SIDE
                                            and gsched.wgrad_side_allowed and gsched.batchnorm_chain())}
        if other is not None:
            line["other_configs"] = other
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["math_accuracy_check"] = math_accuracy_check(dev)
            except Exception as e:      # never costs the headline line
                line["math_accuracy_check"] = {"error": str(e)[:200]}
        if world == 1 and not args.no_roofline and ops_roofline is not None:
            line["roofline_ops"] = ops_roofline
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line goes out last, after anything a native library still holds in its stdio buffer
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
