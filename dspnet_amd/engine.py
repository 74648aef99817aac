"""Static-graph executor for the DSPNet hot path.

MXNet binds a Symbol to an executor and calls forward()/backward()
(multi_solver.py:250-290).  This is the counterpart: a list of nodes with
statically shaped NHWC buffers, run in order for forward and in reverse for
backward.  Every node launches the HIP kernels of include/dspn_nn.h on torch's
current stream; nothing here computes with torch ops.

Gradient buffers: the first producer of a tensor's gradient in a backward pass
writes it, later ones accumulate (kernels take an `accumulate` flag).  A
residual add hands its output-gradient buffer to its inputs by aliasing, so
the gradient of a whole ResNet stage lives in one buffer that each unit
accumulates into (stream order makes this safe: every reader of the old value
was enqueued before the accumulating kernel).

All parameters live in one flat fp32 arena (with matching gradient and momentum
arenas) so the optimizer is one kernel and data-parallel reduction works on a
few large contiguous buckets.
"""
import math

import numpy as np
import torch

from . import functional as fn


# Test switch: with FUSE_BATCHNORM = False every BatchNorm runs its own statistics / apply / backward kernels and
# convolutions never read raw pre-BN tensors (the graph is then built from the plain kernels only).
FUSE_BATCHNORM = True
# Test switch: False builds score3_conv in the direct form (BilinearConcat + tap-expanded Conv) instead of
# BilinearConcatConv.
COMMUTE_RESIZE_CONV = True
# A/B switch (bf16 tensors): BatchNorm outputs read by multi-tap convolutions are materialised instead of being applied in
# the loaders ("auto" BatchNorms: inceptionv3 towers; the residual units set theirs in symbol/resnet.py)
import os as _os
MATERIALISE_MULTITAP_INPUT_BF16 = _os.environ.get("DSPN_MAT3X3", "1") != "0"


_SHARED_STREAMS = {}


def shared_stream(device, kind, priority=0):
    """ONE stream per (device, purpose) for the whole process -- the step stream of the solvers, the side streams of the
    detection branch / MultiBoxTarget / MultiBoxDetection.  Round 4, measured: with a fresh set of streams per graph, the
    SECOND graph of a process ran its convolutions 25 % slower (fp32 MFMA mode: 57.5 -> 79.0 ms per step, conv kernel time 52.9
    -> 65.9 ms): HIP multiplexes streams onto a few hardware queues, and a main stream that lands on the queue of a side
    stream waits behind its long one-workgroup-per-sample kernels.  Graphs of one process run one at a time; sharing the
    streams keeps the mapping of the first graph."""
    key = (str(device), kind)
    ent = _SHARED_STREAMS.get(key)
    if ent is None:
        ent = _SHARED_STREAMS[key] = (torch.cuda.Stream(device=device, priority=priority), priority)
    assert ent[1] == priority, "shared stream %r of %s was created with priority %d, asked for %d" % (kind, device, ent[1], priority)
    return ent[0]


# Round 6: weight gradients on a stream of their own (WGRAD_SIDE below).  A weight gradient READS its layer's output gradient;
# the only writers of that buffer later in the same backward pass are ACCUMULATIONS through an alias (the residual stream:
# conv3's output gradient is handed to the unit's input, whose first BatchNorm adds its own gradient in place).  Every
# accumulating writer gets its target from Tensor.grad_target() / give_grad(): they make the current stream wait for the
# weight gradient that still reads the buffer.
_WG_READS = {}      # data_ptr of an output-gradient buffer -> (event recorded behind the weight gradient that reads it, device)


def _wg_before_write(buf):
    if _WG_READS and buf is not None:
        ent = _WG_READS.pop(buf.data_ptr(), None)
        if ent is not None:
            torch.cuda.current_stream(ent[1]).wait_event(ent[0])


class Tensor:
    """An activation: NHWC (or any) device buffer + its gradient slot.  dtype: the graph's activation storage type
    (functional.ACT_DTYPE: float32, or bfloat16 for the `*_bf16` kernels) unless given -- graph inputs, loss inputs
    and graph outputs are float32 in both."""

    def __init__(self, shape, name, requires_grad=True, data=None, device=None, virtual=False, dtype=None):
        self.name = name
        self.shape = tuple(int(s) for s in shape)
        self.requires_grad = requires_grad
        self.device = device
        self.dtype = data.dtype if data is not None else (dtype or fn.ACT_DTYPE)
        # virtual: never materialised (a BatchNorm output that only convolutions consume: they apply the affine in
        # their tile loaders, see BatchNorm(defer_apply=True)); .data stays None so any other consumer fails loudly
        self.affine_src = None
        self.producer = None       # the Conv node that writes this tensor (it can emit BatchNorm tile statistics)
        # "f16x2" math: index of a magnitude block that BOUNDS |values| once the producer has run in this step (a materialised
        # BatchNorm takes it while it writes the tensor; pooled / aliased tensors inherit their source's -- a bound that is too
        # large by less than 2^17 costs the two-piece math nothing), and the tensor whose values this one shares (BlockGrad)
        self.am_slot = None
        self.alias_of = None
        self.grad_planes = False   # this backward pass: .grad holds fp16 piece planes, not floats (BatchNorm.dx_planes)
        self.channels = None       # logical channel count when the last axis is padded (19 -> 20, 3 -> 4)
        self.data = None if virtual else (data if data is not None else fn.zeros(*self.shape, device=device, dtype=self.dtype))
        self.grad = None
        self._own_grad = None
        self._gw = False  # gradient written in the current backward pass

    def own_grad(self):
        if self._own_grad is None:
            self._own_grad = (torch.zeros_like(self.data) if self.data is not None
                              else fn.zeros(*self.shape, device=self.device, dtype=self.dtype))
        return self._own_grad

    def grad_target(self):
        """(buffer, accumulate) for a kernel that produces this tensor's gradient"""
        if not self._gw:
            self.grad = self.own_grad()
            self._gw = True
            return self.grad, False
        _wg_before_write(self.grad)
        return self.grad, True

    def give_grad(self, buf):
        """hand over a finished gradient buffer (alias if first, else accumulate)"""
        if not self._gw:
            self.grad = buf
            self._gw = True
        else:
            _wg_before_write(self.grad)
            fn.add(self.grad, buf, out=self.grad)


class Param:
    def __init__(self, name, shape, init, wd_mult=1.0):
        self.name, self.shape, self.init, self.wd_mult = name, tuple(shape), init, wd_mult
        self.data = self.grad = None
        self.offset = 0
        # shape of the same parameter in the reference's checkpoint: (Cout, Cin, kh, kw) for a Convolution weight,
        # (in, out, kh, kw) for a Deconvolution weight, (C,) otherwise -- without the channel padding of `shape`
        self.logical = None
        self.kind = "vec"          # "conv" | "deconv" | "vec"

    @property
    def size(self):
        return int(np.prod(self.shape))


class Node:
    def forward(self):
        raise NotImplementedError

    def backward(self):
        pass


class Graph:
    def __init__(self, device):
        self.device = device
        self.nodes = []
        self.params = {}
        self.param_order = []
        self.bn_names = []         # (name, channels, fix_gamma) of every BatchNorm, for checkpoint files
        self.tensors = {}
        self.all_tensors = []
        self.pre_forward = []      # callables run at the top of forward() (joins of side-stream work)
        # round 6: weight gradients may run on a stream of their own (WGRAD_SIDE).  A solver with a gradient all-reduce clears it:
        # every bucket release would make the step's stream wait for that stream, and no N > 1 run has measured the mix
        self.wgrad_side_allowed = True
        # round 4: nodes [first, last] whose FORWARD runs on a second stream beside the nodes behind them (the detection
        # branch -- small SSD layers, heads, packing, target matching: tens of launches of 1 - 32 workgroups -- beside the
        # segmentation decoder); the first reader of their results calls join_side().  (first, last, stream, ready, done)
        self.side_segment = None
        self.side_pending = False
        # ... and the BACKWARD of some of those nodes beside the decoder's (set_side_backward): only nodes whose gradients
        # stay inside the branch; every main-stream node of the branch waits for what the side stream has been given so far
        self.side_bwd = None
        self.side_fwd_events, self.side_fwd_waits = {}, {}   # _plan_side_sync
        self.side_plan = None      # what the builder asked for (any device) and what _plan_side_sync made of it
        self.slab_tables = {}      # key -> (conv nodes, device table) of deferred split-K slab reductions
        self.wt_table = None       # descriptor table of every Conv's (weight, transposed weight) pair
        self.wt_batched = False    # True while backward() runs after one batched transpose launch
        self.arena = self.grad_arena = self.mom_arena = None
        self.half_operands = False  # bf16 convolution operands: refreshed from the float masters at the top of forward()
        # math of the float-tensor convolutions of THIS graph (include/dspn_nn.h DSPN_MATH_*), fixed when the graph is
        # created: a later functional.set_conv_math() does not change what an existing graph computes
        self.math = fn.get_conv_math()
        self.wp_table = None       # split math: descriptor table of every piece-plane weight operand (one launch per step)
        # "f16x2" math: one float per operand magnitude (functional.absmax), all in one arena zeroed at the top of forward();
        # slots are handed out while the graph is built (new_scalar) and become views of the arena in finalize()
        self._nscal = 0
        self.scalars = None
        self.am_table = None       # descriptor table of the weight magnitudes (one launch per step)
        self._am_x = {}            # (id of the raw input tensor, id of its affine scale) -> slot shared by its readers
        self._am_wanted = set()    # magnitude slots some two-piece convolution will read off its input tensor (want_magnitude)
        self._am_done = set()      # slots already computed in the current step
        self._am_bwd_slots, self._am_bwd_ran = set(), False
        self._am_table_slots = set()   # weight magnitudes taken by the one batched launch at the top of forward()
        # range monitor of the "f16x2" math: per convolution-input slot the smallest non-zero per-channel magnitude, where a
        # BatchNorm finalize sees the channels (range_report)
        self.scalars_min = None
        # round 5: the range GUARD of the "f16x2" math.  Per magnitude slot the smallest non-zero per-channel magnitude travels
        # beside the largest (scalars_min: inputs from the BatchNorm finalize, weights from one batched launch per step, output
        # gradients from the BatchNorm backward's per-channel bounds); a convolution one of whose operands spanned more than
        # 2^GUARD_BITS in the PREVIOUS pass runs all three of its calls in the three-piece bf16 math this step (guard_fb on the
        # node; the span of a trained net's channels moves over many steps, not within one).  DSPN_RANGE_GUARD=0 switches it off.
        self.guard = dict(enabled=_os.environ.get("DSPN_RANGE_GUARD", "1") != "0", have_stats=False, risk=frozenset(),
                          calls=0, calls_total=0, at_risk_slots=0)
        self.wmin_table = None     # descriptor table of the per-output-channel weight minima (one launch per step)

    # -- construction ---------------------------------------------------------
    def tensor(self, shape, name, requires_grad=True, data=None, virtual=False, dtype=None):
        t = Tensor(shape, name, requires_grad, data=data, device=self.device, virtual=virtual, dtype=dtype)
        assert name not in self.tensors, "duplicate tensor name " + name
        self.tensors[name] = t
        self.all_tensors.append(t)
        return t

    def param(self, name, shape, init):
        assert name not in self.params, name
        p = Param(name, shape, init)
        self.params[name] = p
        self.param_order.append(p)
        return p

    def add(self, node):
        self.nodes.append(node)
        return node

    def new_scalar(self, backward=False):
        """index of a fresh slot of the operand-magnitude arena ("f16x2" math).  backward=True: the magnitude of a gradient;
        a second backward pass after one forward pass takes those again (begin_backward)"""
        self._nscal += 1
        if backward:
            self._am_bwd_slots.add(self._nscal - 1)
        return self._nscal - 1

    def scalar(self, i):
        return None if (i is None or self.scalars is None) else self.scalars[i * fn.ABSMAX_SLOTS:(i + 1) * fn.ABSMAX_SLOTS]

    def want_magnitude(self, t):
        """a two-piece convolution will multiply tensor t as it is: whoever produces it (or what it was pooled from) is asked
        to leave its magnitude block (Conv._out_magnitude) -- nobody takes one for tensors no convolution reads"""
        while t is not None:
            if t.am_slot is not None:
                self._am_wanted.add(t.am_slot)
            t = t.alias_of

    def magnitude(self, slot, tensor, src=None):
        """"f16x2" math: the magnitude block of `tensor` in slot `slot` (a node-owned index from new_scalar), taken by a pass
        over the tensor the first time a step asks for it and reused afterwards; None in the other math modes.
        src: the engine Tensor behind `tensor` -- if its producer has already left a bound of it this step, that block"""
        if slot is None or self.scalars is None or tensor.dtype != torch.float32:
            return None
        if src is not None and src.am_slot is not None and src.am_slot in self._am_done:
            return self.scalar(src.am_slot)
        out = self.scalar(slot)
        if slot not in self._am_done:
            fn.absmax(tensor, out=out)
            self._am_done.add(slot)
        return out

    def range_report(self):
        """"f16x2" math, after a step (one device -> host copy): (tensors seen, tensors whose channel magnitudes span more than
        2^16, largest span in bits) over the convolution inputs whose BatchNorm finalize saw per-channel extremes.  The
        two-piece math keeps fp32 relative accuracy for elements within 2^17 of their tensor's largest magnitude; channels
        further below keep an absolute error of 2^-39 of that magnitude (include/dspn_nn.h)."""
        if self.scalars is None:
            return (0, 0, 0.0)
        mx = self.scalars.view(-1, fn.ABSMAX_SLOTS).max(dim=1).values.cpu().numpy()
        mn = self.scalars_min.cpu().numpy()
        ok = np.isfinite(mn) & (mn > 0) & np.isfinite(mx) & (mx > 0)
        if not ok.any():
            return (0, 0, 0.0)
        span = np.log2(mx[ok] / mn[ok])
        return (int(ok.sum()), int((span > 16).sum()), float(span.max()))

    GUARD_BITS = 16
    GUARD_PERIOD = 4

    def _spans_device(self):
        """(2, slots) device tensor: the largest (FINITE partial maxima, as the kernels read them) and the smallest non-zero
        per-channel magnitude per slot of the pass that has just been issued"""
        mx = self.scalars.view(-1, fn.ABSMAX_SLOTS)
        mx = torch.where(mx < 3.0e38, mx, torch.zeros_like(mx)).max(dim=1).values
        return torch.stack([mx, self.scalars_min])

    def slot_spans(self):
        """the same as numpy arrays (synchronises); nan / inf where a slot has no per-channel information"""
        both = self._spans_device().cpu().numpy()
        return both[0], both[1]

    def _update_guard(self, blocking=False):
        """Called at the top of forward(), before the magnitude arena is zeroed: which convolutions run their calls of THIS
        pass in the three-piece math.  The spans of the pass issued last are copied to pinned host memory behind it (no
        synchronisation); the decisions are taken from the copy made one pass EARLIER, which has long arrived -- the host
        runs ahead of the device and must not wait for it, and taking a fixed lag (not "whatever has arrived") keeps a run
        reproducible.  blocking=True (MultiTaskSolver's calibration pass, tests): decide from the pass issued last.
        A forward() that is being RECORDED into a HIP graph cannot look at the device: it keeps the decisions it is recorded
        with and takes the weights' minima every step; MultiTaskSolver polls guard_poll() between replays instead."""
        gd = self.guard
        gd["calls"] = 0
        capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if capturing:
            gd["wmin_now"] = True      # (the recorded launch sequence is the same every step: the minima are part of it)
            return
        gd["count"] = gd.get("count", 0) + 1
        # the weights' per-channel minima cost a launch over every weight: taken in the pass BEFORE one whose spans are read
        gd["wmin_now"] = blocking or gd["count"] % self.GUARD_PERIOD == self.GUARD_PERIOD - 1 or not gd["have_stats"]
        if not gd["enabled"] or self.scalars is None or not gd["have_stats"]:
            return
        if not blocking and gd["count"] % self.GUARD_PERIOD != 0:
            return                     # spans move over many steps: looked at every GUARD_PERIOD-th pass (0.7 % of the step -> 0.2 %)
        both = self._guard_fetch(blocking)
        if both is not None:
            self._guard_decide(both)

    def _guard_fetch(self, blocking):
        """the (largest, smallest) magnitudes per slot of the pass issued last: read now (blocking), or -- without waiting for
        the device -- the copy requested one call EARLIER (None while there is none)"""
        gd = self.guard
        if blocking:
            gd["queue"] = []
            return self._spans_device().cpu().numpy()
        pool = gd.setdefault("pinned", [torch.empty((2, self.scalars_min.numel()), dtype=torch.float32, pin_memory=True)
                                        for _ in range(3)])       # (at most two copies are in flight)
        host = pool[gd.get("pin_i", 0) % 3]
        gd["pin_i"] = gd.get("pin_i", 0) + 1
        host.copy_(self._spans_device(), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        gd.setdefault("queue", []).append((host, ev))
        if len(gd["queue"]) < 2:
            return None
        host, ev = gd["queue"].pop(0)
        ev.synchronize()
        return host.numpy()

    def _guard_decide(self, both):
        """-> True when the set of convolutions on the fallback changed"""
        gd = self.guard
        mx, mn = both[0], both[1]
        ok = np.isfinite(mn) & (mn > 0) & (mx > 0)
        wide = np.zeros(mx.shape, bool)
        wide[ok] = np.log2(mx[ok] / mn[ok]) > self.GUARD_BITS
        gd["at_risk_slots"] = int(wide.sum())
        risk = set()
        if wide.any():
            for n in self.nodes:
                if isinstance(n, Conv) and n.am_x is not None and not n.tap_expand:
                    if wide[n.am_x] or wide[n.am_w] or wide[n.am_dy]:
                        risk.add(id(n))
        if risk == gd["risk"]:
            return False
        gd["risk"] = frozenset(risk)
        for n in self.nodes:
            if isinstance(n, Conv):
                n.guard_fb = id(n) in risk
        return True

    def guard_poll(self, blocking=False):
        """The guard of a RECORDED step (MultiTaskSolver.capture; advisor r5): called between two replays, on the stream they
        run on.  Copies the spans of the replayed pass out behind it and decides from the copy of the previous poll
        (blocking=True: from this pass).  -> True when the set of convolutions on the fallback changed: the recorded launch
        sequence no longer is the step, the caller drops the graph and records a new one."""
        gd = self.guard
        if not gd["enabled"] or self.scalars is None or not gd["have_stats"]:
            return False
        both = self._guard_fetch(blocking)
        return both is not None and self._guard_decide(both)

    def guard_report(self):
        """(convolutions on the fallback this pass, fallback kernel calls this pass, slots whose span exceeded 2^GUARD_BITS last pass)"""
        return len(self.guard["risk"]), self.guard["calls"], self.guard["at_risk_slots"]

    def _resolve_auto_deferred(self):
        """BatchNorm(defer_apply="auto"): keep the output virtual only if every reader is a plain convolution input;
        otherwise (a concat, a pooling layer, a residual operand ...) materialise it and unhook the convolutions."""
        readers = {}
        for n in self.nodes:
            for k, v in vars(n).items():
                if k in ("out", "x_raw"):
                    continue
                for t in (v if isinstance(v, (list, tuple)) else [v]):
                    if isinstance(t, Tensor) and t.affine_src is not None:
                        ok = isinstance(n, Conv) and k == "x" and not n.tap_expand
                        # bf16 tensors: a multi-tap convolution would re-apply the affine once per tap on a main loop that
                        # is 3x shorter than in fp32 (scratch/fuse_cost.py bf16: +45 %); its input is materialised instead
                        if ok and MATERIALISE_MULTITAP_INPUT_BF16 and t.dtype == torch.bfloat16 and n.w.shape[1] * n.w.shape[2] > 1:
                            ok = False
                        readers.setdefault(id(t), []).append((n, ok))
        for n in self.nodes:
            if not isinstance(n, BatchNorm) or n.defer_apply != "auto":
                continue
            rs = readers.get(id(n.out), [])
            if rs and all(ok for _, ok in rs):
                n.defer_apply = True
                continue
            n.defer_apply = False
            n.out.affine_src = None
            n.out.data = fn.zeros(*n.out.shape, device=self.device, dtype=n.out.dtype)
            for c in n.conv_consumers:
                c.x_raw, c.in_affine = c.x, None
            n.mat_consumers = list(n.conv_consumers)     # still candidates for gathering the backward reductions
            n.conv_consumers = []

    def _plan_bn_backward_fusion(self):
        """For every BatchNorm whose output only convolutions read: the convolution that runs LAST in backward (the
        first in forward order) gathers the BatchNorm-backward reductions in its data-gradient epilogue."""
        # every reader of a tensor, to make sure the convolutions are the ONLY readers of a materialised BatchNorm output
        readers = {}
        for m in self.nodes:
            for k, v in vars(m).items():
                if k == "out":
                    continue
                for t in (v if isinstance(v, (list, tuple)) else [v]):
                    if isinstance(t, Tensor):
                        readers.setdefault(id(t), set()).add(id(m))
        for n in self.nodes:
            if not isinstance(n, BatchNorm) or not n.out.requires_grad:
                continue
            cons = n.conv_consumers if n.defer_apply else n.mat_consumers
            if not cons or (not n.defer_apply and readers.get(id(n.out), set()) != {id(c) for c in cons}):
                continue
            last = min(cons, key=self.nodes.index)
            if last.stride not in (1, 2) or (last.stride == 2 and last.dil != 1) or last.x.shape[3] % 4 != 0:
                continue
            tiles = fn.conv_dgrad_bn_tiles(last.x.shape, last.stride)
            if tiles <= 0:
                continue
            n.bwd_sums = (fn.zeros(tiles, 2, last.x.shape[3], device=self.device), tiles)
            last.bn_bwd_node = n

    def _plan_gradient_magnitudes(self):
        """"f16x2" math: a BatchNorm whose backward is the LAST writer of its input's gradient (backward runs the nodes in
        reverse: the reader with the smallest index) stores the complete gradient, so its apply kernel can also take the
        magnitude the producing convolution needs (BatchNorm.completes_x_grad)."""
        first_reader, readers = {}, {}
        for idx, m in enumerate(self.nodes):
            for k, v in vars(m).items():
                if k == "out":
                    continue
                for t in (v if isinstance(v, (list, tuple)) else [v]):
                    if isinstance(t, Tensor):
                        first_reader.setdefault(id(t), idx)
                        readers.setdefault(id(t), set()).add(idx)
        for idx, n in enumerate(self.nodes):
            if isinstance(n, BatchNorm):
                n.completes_x_grad = first_reader.get(id(n.x)) == idx
        # a convolution read by nothing but another convolution's residual add (the projection shortcut of a unit) receives
        # that convolution's output gradient itself (Tensor.give_grad aliases it): one magnitude slot serves both
        for idx, n in enumerate(self.nodes):
            r = getattr(n, "residual", None)
            if isinstance(n, Conv) and n.am_dy is not None and r is not None and r.requires_grad:
                prod = getattr(r, "producer", None)
                if (isinstance(prod, Conv) and prod.am_dy is not None and prod.out is r and readers.get(id(r)) == {idx}
                        and not prod.relu and prod.b is None):
                    prod.am_dy = n.am_dy

    def _plan_gradient_planes(self):
        """"f16x2" math, round 4: a BatchNorm whose backward (from the sums its consumer's data gradient gathered) is the ONLY
        writer of its input's gradient, and whose input is the dense output of a plain convolution, writes that gradient as
        fp16 piece planes -- same buffer, same bytes -- cut by a bound it forms beforehand (dspn_bn_backward_from_sums_f32,
        dx_planes); the convolution's data gradient and weight gradient then copy their dy operand instead of cutting it
        once per tap and column tile.  In the residual units: bn2 -> conv1 and bn3 -> conv2."""
        if self.math != "f16x2" or self.device.type != "cuda":
            return
        readers = {}
        for idx, m in enumerate(self.nodes):
            for k, v in vars(m).items():
                if k in ("out", "x_raw"):      # (x_raw: a convolution reading THROUGH a deferred BatchNorm; the gradient goes to the BatchNorm)
                    continue
                for t in (v if isinstance(v, (list, tuple)) else [v]):
                    if isinstance(t, Tensor):
                        readers.setdefault(id(t), set()).add(idx)
        gatherer = {id(n.bn_bwd_node): n for n in self.nodes if isinstance(n, Conv) and getattr(n, "bn_bwd_node", None) is not None}
        for idx, n in enumerate(self.nodes):
            if not isinstance(n, BatchNorm) or n.bwd_sums is None or id(n) not in gatherer:
                continue
            prod, x = getattr(n.x, "producer", None), n.x
            if not (isinstance(prod, Conv) and prod.out is x and readers.get(id(x)) == {idx} and x.requires_grad
                    and x.dtype == torch.float32 and n.completes_x_grad and n.tile_stats is not None):
                continue
            if (prod.tap_expand or prod.relu or prod.b is not None or prod.residual is not None or prod.input_sum_grad is not None
                    or prod.am_dy is None or x.shape[3] != prod.cout or prod.cout % 32 != 0 or prod.out_minmax is None
                    or gatherer[id(n)].math != "f16x2"):
                continue
            n.dx_planes = True
            n.x_ext = fn.zeros(2, x.shape[3], device=self.device)
            n.am_dyin = self.new_scalar(backward=True)

    def _plan_input_planes(self):
        """"f16x2" math, round 4: a deferred BatchNorm whose readers re-read it often (X_PLANES_MIN_READS) also writes
        (relu)(x * scale + shift) as fp16 piece planes (dspn_bn_apply_planes_f32), cut by the magnitude its statistics finalize
        has just formed; that convolution's forward and weight gradient then copy their x operand into LDS instead of applying
        the affine and cutting every element once per (tap, column tile) -- 9 x Cout / 128 times for a 3 x 3.  The 1 x 1 readers
        (and the BatchNorm backward) keep reading the raw tensor.  In the residual units: bn2 -> conv2."""
        if self.math != "f16x2" or self.device.type != "cuda":
            return
        for n in self.nodes:
            if not isinstance(n, BatchNorm) or n.defer_apply is not True or n.tile_stats is None:
                continue
            x = n.x
            if (x.dtype != torch.float32 or x.shape[3] % 32 != 0 or x.data is None
                    or getattr(getattr(x, "producer", None), "out_minmax", None) is None):
                continue
            cons = [c for c in n.conv_consumers if c.math == "f16x2" and c.wp is not None and c.x_raw is x]
            # how often the tile loaders would apply the affine to (and cut) one element: once per tap and 128-column tile
            reads = sum(c.w.shape[1] * c.w.shape[2] * ((c.cout + 127) // 128) for c in cons)
            if reads < X_PLANES_MIN_READS:
                continue
            n.planes = fn.zeros(*x.shape, device=self.device)
            for c in cons:
                c.x_planes_bn = n

    def finalize(self, seed=0):
        """allocate the flat parameter / gradient / momentum arenas and initialise"""
        self._resolve_auto_deferred()
        self._plan_side_sync()
        if _os.environ.get("DSPN_X_PLANES", "1") != "0":       # (A/B switch)
            self._plan_input_planes()
        self._plan_bn_backward_fusion()
        self._plan_gradient_magnitudes()
        if _os.environ.get("DSPN_DY_PLANES", "1") != "0":      # (A/B switch)
            self._plan_gradient_planes()
        self._wt_pairs_nodes = [n for n in self.nodes if isinstance(n, Conv) and (n.wt is not None or n.wh is not None)]
        off = 0
        for p in self.param_order:
            p.offset = off
            off += (p.size + 3) // 4 * 4
        self.arena = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grad_arena = torch.zeros_like(self.arena)
        self.mom_arena = torch.zeros_like(self.arena)
        rng = np.random.Generator(np.random.PCG64(seed))
        host = np.zeros(off, np.float32)
        for p in self.param_order:
            host[p.offset:p.offset + p.size] = np.asarray(p.init(rng, p.shape), np.float32).reshape(-1)
        self.arena.copy_(torch.from_numpy(host))
        for p in self.param_order:
            p.data = self.arena[p.offset:p.offset + p.size].view(p.shape)
            p.grad = self.grad_arena[p.offset:p.offset + p.size].view(p.shape)
        if self._wt_pairs_nodes and self.device.type == "cuda":
            # bf16 operands: every Conv has a bf16 copy wh of its float master (forward operand); a Conv whose input
            # needs no gradient still gets a (scratch) transposed operand so that one table row serves it
            for n in self._wt_pairs_nodes:
                if n.wt is None:
                    n.wt = fn.zeros(n.w.shape[3], n.w.shape[1], n.w.shape[2], n.out.shape[3], device=self.device,
                                    dtype=n.wh.dtype)
            # split math: a data gradient that contracts over whole 32-channel blocks reads piece planes, not the float
            # transpose (Conv.__init__ allocated them); only the other layers stay in the transpose table
            pairs = [n for n in self._wt_pairs_nodes if n.wtp is None or n.wh is not None]
            if pairs:
                self.wt_table = fn.weight_transpose_table([(n.w.data, n.wt, n.wh) for n in pairs], self.device)
                self.half_operands = self.wt_table[3]
        if self.device.type == "cuda":
            if self.math == "f16x2" and self._nscal:
                self.scalars = torch.zeros(self._nscal * fn.ABSMAX_SLOTS, dtype=torch.float32, device=self.device)
                self.scalars_min = torch.full((self._nscal,), float("inf"), dtype=torch.float32, device=self.device)
                wnodes = [n for n in self.nodes if getattr(n, "am_w", None) is not None and getattr(n, "w", None) is not None]
                pairs = [(n.w.data, self.scalar(n.am_w)) for n in wnodes]
                self._am_table_slots = {n.am_w for n in wnodes}
                if pairs:
                    self.am_table = fn.absmax_table(pairs, self.device)
                    if self.guard["enabled"]:
                        self.wmin_table = fn.absmin_rows_table(
                            [(n.w.data.view(n.w.shape[0], -1), self.scalars_min[n.am_w:n.am_w + 1]) for n in wnodes], self.device)
            planes = [(n.w.data, n.wp, n.wtp) + ((self.scalar(n.am_w),) if self.math == "f16x2" else ())
                      for n in self.nodes if isinstance(n, Conv) and (n.wp is not None or n.wtp is not None)]
            if planes:
                self.wp_table = fn.weight_planes_table(planes, self.device)
        if self.device.type == "cuda":
            # every Conv keeps the split-K partial sums of its weight gradient in a buffer of its own, so that the
            # slab sums of many layers run as one launch (flush_slabs) instead of one small kernel per layer
            for n in self.nodes:
                if isinstance(n, Conv):
                    n.alloc_slabs()
        return self

    def flush_slabs(self, key="all", convs=None, beside=False):
        """sum the pending split-K slabs of `convs` (default: every Conv) into their weight gradients, one launch.
        beside (the weight gradients run on a stream of their own, WGRAD_SIDE): the sum goes to THAT stream, behind the weight
        gradients it reads and behind what the current stream has issued; the current stream does not wait (join_side_backward
        does, at the end of the pass)"""
        if key not in self.slab_tables:
            nodes = [n for n in (convs if convs is not None else self.nodes) if isinstance(n, Conv) and n.slabs is not None]
            table = fn.slab_reduce_table([(n.slabs, n.w.grad, False) for n in nodes], self.device) if nodes else None
            self.slab_tables[key] = (nodes, table)
        nodes, table = self.slab_tables[key]
        st = self.__dict__.get("_wg")
        beside = beside and st is not None and st["last"] is not None
        if not beside:
            self.wgrad_beside_join()      # (before a bucket's gradients leave: its weight gradients on the second stream have finished)
        if not nodes:
            return
        ran = nodes if all(n.slabs_fresh for n in nodes) else [n for n in nodes if n.slabs_fresh]
        if ran:
            # (some convolution had no output gradient in this pass: reduce only the ones that ran)
            args = table if ran is nodes else fn.slab_reduce_table([(n.slabs, n.w.grad, False) for n in ran], self.device)
            if beside:
                main, side = torch.cuda.current_stream(self.device), st["stream"]
                ev = self._wg_event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    fn.slab_reduce_batch(*args)
                    done = self._wg_event()
                    done.record(side)
                st["last"] = done
            else:
                fn.slab_reduce_batch(*args)
        for n in nodes:
            n.slabs_fresh = False

    def load_params(self, values):
        """values: name -> numpy array in the param's (device-layout) shape"""
        for k, v in values.items():
            self.params[k].data.copy_(torch.from_numpy(np.ascontiguousarray(v, np.float32)).view(self.params[k].shape))


    # -- stream ordering of the side segment (round 5) ---------------------------------------------
    def _node_tensors(self, n):
        """every engine Tensor a node holds (attributes, lists / tuples of them), each with the tensors whose buffers it reads
        through: the raw input of a deferred BatchNorm (affine_src) and the source of a BlockGrad alias"""
        seen, out = set(), []

        def take(t):
            while t is not None and id(t) not in seen:
                seen.add(id(t))
                out.append(t)
                if t.affine_src is not None:
                    take(t.affine_src[0])
                t = t.alias_of

        for v in vars(n).values():
            for t in (v if isinstance(v, (list, tuple)) else [v]):
                if isinstance(t, Tensor):
                    take(t)
        return out

    @staticmethod
    def _node_outputs(n):
        outs = []
        for k, v in vars(n).items():
            if k == "out" or k.startswith("out_") or k in ("outs", "cls_prob", "cls_preds", "prob"):
                outs += [t for t in (v if isinstance(v, (list, tuple)) else [v]) if isinstance(t, Tensor)]
        return outs

    def _plan_side_sync(self):
        """Called by finalize().  The builder only says WHICH nodes run beside the main stream; what orders the two streams
        is derived here from the tensors the nodes hold, so that a preset whose wiring differs (vgg16_reduced, inceptionv3,
        resnet-101: the decoder reads an SSD extra layer's output, symbol/multitask_symbol_builder.py `conv_feat`) cannot
        silently become a race:
        * forward: a main-stream node behind the segment that holds a tensor written inside it waits for an event recorded
          behind the LAST in-segment node that writes (or, failing that, last holds) that tensor -- not for the whole branch;
        * backward: a node leaves the side set if it shares a tensor with a main-stream node that runs between the fork and
          itself (the decoder writes that tensor's gradient on the main stream while the side stream, which only waits for
          the fork event, would read or accumulate into it), repeated until nothing changes."""
        plan = self.side_plan
        self.side_fwd_events, self.side_fwd_waits = {}, {}
        if plan is None:
            return
        first, last = plan["first"], plan["last"]
        cuda = self.side_segment is not None
        refs = [self._node_tensors(n) for n in self.nodes]
        holders = {}
        for i, ts in enumerate(refs):
            for t in ts:
                holders.setdefault(id(t), []).append(i)
        before = set()
        for i in range(first):
            before.update(id(t) for t in refs[i])
        for tid, idxs in holders.items():
            inside = [i for i in idxs if first <= i <= last]
            outside = [i for i in idxs if i > last]
            if not inside or not outside or tid in before:
                continue
            writers = [i for i in inside if any(id(t) == tid for t in self._node_outputs(self.nodes[i]))]
            # no declared writer (a node type whose output attribute _node_outputs does not know): behind the LAST in-segment
            # holder, which is safe whichever of them writes
            src = max(writers) if writers else max(inside)
            if src not in self.side_fwd_events:
                self.side_fwd_events[src] = torch.cuda.Event() if cuda else None
            for r in outside:
                w = self.side_fwd_waits.setdefault(r, [])
                if src not in w:
                    w.append(src)
        plan["fwd_waits"] = {r: list(w) for r, w in self.side_fwd_waits.items()}
        sb = plan["bwd"]
        if sb is None:
            return
        side, fork_after = set(sb["side"]), sb["fork_after"]
        has_bwd = [type(n).backward is not Node.backward for n in self.nodes]
        changed = True
        while changed:
            changed = False
            for i in sorted(side):
                mine = {id(t) for t in refs[i]}
                for x in range(i + 1, fork_after):
                    if x in side or not has_bwd[x]:
                        continue
                    if mine & {id(t) for t in refs[x]}:
                        side.discard(i)
                        changed = True
                        break
        sb["removed"] = frozenset(sb["side"]) - side
        sb["side"] = frozenset(side)
        if self.side_bwd is not None:
            self.side_bwd["side"] = frozenset(side)
            if not side:
                self.side_bwd = None

    # -- execution ------------------------------------------------------------
    def forward(self):
        self.wt_batched = False        # the weights may have changed since the last batched transpose
        if self.half_operands:         # bf16 copies (forward) and transposes (data gradient) of every weight, one launch
            fn.weight_transpose_batch(*self.wt_table)
            self.wt_batched = True
        if self.scalars is not None:   # "f16x2" math: every operand magnitude of the step starts from zero; the weights' now
            self._update_guard(blocking=self.guard.pop("decide_now", False))
            self.scalars.zero_()
            self.scalars_min.fill_(float("inf"))
            if self.wmin_table is not None and self.guard.get("wmin_now", True):
                fn.absmin_rows_batch(*self.wmin_table)
            self._am_done = set()
            self._am_bwd_ran = False
            if self.am_table is not None:
                fn.absmax_batch(*self.am_table)
                self._am_done |= self._am_table_slots
        if self.wp_table is not None:  # split math: the piece planes of every weight (forward and data-gradient operands;
            fn.weight_planes_batch(*self.wp_table)     # "f16x2": cut relative to the magnitudes just taken)
        for f in self.pre_forward:
            f()
        seg = self.side_segment
        if seg is None:
            for n in self.nodes:
                n.forward()
            self.guard["have_stats"] = True
            return
        first, last, side, ready, done = seg
        self.join_side()
        for i, n in enumerate(self.nodes):
            if i < first or i > last:
                for src in self.side_fwd_waits.get(i, ()):     # (a tensor written inside the segment: _plan_side_sync)
                    torch.cuda.current_stream(self.device).wait_event(self.side_fwd_events[src])
                n.forward()
                continue
            if i == first:       # the branch reads what the main stream has produced so far
                ready.record(torch.cuda.current_stream(self.device))
                side.wait_event(ready)
            with torch.cuda.stream(side), fn.workspace_lane(1):
                n.forward()
                if i in self.side_fwd_events:
                    self.side_fwd_events[i].record(side)
                if i == last:
                    done.record(side)
                    self.side_pending = True
        self.guard["have_stats"] = True       # (forward-only use -- the Detector -- is guarded by the spans of inputs and weights)

    def set_side_segment(self, first, last):
        """nodes [first, last] run their forward on the branch stream.  On a CPU graph only the plan is kept (side_plan):
        tests/test_side_plan.py checks what _plan_side_sync derives from it for every preset without a GPU."""
        assert 0 <= first <= last < len(self.nodes)
        self.side_plan = dict(first=first, last=last, bwd=None)
        if self.device.type == "cuda":
            self.side_segment = (first, last, shared_stream(self.device, "branch", -1), torch.cuda.Event(), torch.cuda.Event())

    def set_side_backward(self, side_nodes, lo, hi, fork_after):
        """side_nodes: indices (within [lo, hi]) whose backward runs on the side stream; it starts behind the backward of
        node `fork_after` (the last producer of the branch's incoming gradients), not behind whatever the main stream was
        given after it"""
        assert self.side_plan is not None and all(lo <= i <= hi for i in side_nodes) and fork_after > hi
        self.side_plan["bwd"] = dict(side=frozenset(side_nodes), lo=lo, hi=hi, fork_after=fork_after)
        if self.side_segment is not None:
            self.side_bwd = dict(side=frozenset(side_nodes), lo=lo, hi=hi, fork_after=fork_after, fork_ev=torch.cuda.Event(),
                                 prog_ev=torch.cuda.Event(), forked=False, dirty=False, active=True)

    def backward_node(self, idx):
        """backward of node idx on the stream it belongs to (the main stream unless set_side_backward says otherwise)"""
        sb, n = self.side_bwd, self.nodes[idx]
        if sb is None or not sb["active"]:
            n.backward()
            return
        if idx in sb["side"]:
            side = self.side_segment[2]
            if not sb["forked"]:
                side.wait_event(sb["fork_ev"])
                sb["forked"] = True
            with torch.cuda.stream(side), fn.workspace_lane(1):
                n.backward()
                sb["prog_ev"].record(side)
            sb["dirty"] = True
            return
        if sb["dirty"] and idx <= sb["hi"]:      # a main-stream node of (or below) the branch: behind everything given to the side stream
            torch.cuda.current_stream(self.device).wait_event(sb["prog_ev"])
            sb["dirty"] = False
        n.backward()
        if idx == sb["fork_after"]:
            sb["fork_ev"].record(torch.cuda.current_stream(self.device))

    # -- round 6 (WGRAD_SIDE): every weight gradient of the step's stream on a second stream, off the data-gradient chain
    def batchnorm_chain(self):
        """True for a graph whose backward pass is mostly the chain data gradient -> BatchNorm finalize -> apply (more than half
        of its convolutions gather a BatchNorm's backward sums): there a weight gradient beside the chain fills gaps (resnet-50);
        without BatchNorm the chain is as MFMA-bound as the weight gradients and sharing the chip costs (vgg16_reduced)"""
        v = self.__dict__.get("_bn_chain")
        if v is None:
            convs = [n for n in self.nodes if isinstance(n, Conv)]
            v = self._bn_chain = 2 * sum(1 for n in convs if getattr(n, "bn_bwd_node", None) is not None) > len(convs)
        return v

    def _wg_event(self):
        st = self._wg
        if st["used"] == len(st["events"]):
            st["events"].append(torch.cuda.Event())
        st["used"] += 1
        return st["events"][st["used"] - 1]

    def wgrad_beside(self, conv, dy, planes, xa, dya):
        main = torch.cuda.current_stream(self.device)
        if main == shared_stream(self.device, "branch", -1) or conv.tap_expand:
            return False
        # the stream MultiBoxTarget uses in the forward pass, idle during backward.  (On the detection branch's stream, behind that
        # branch's own backward: +0.9 % instead of +1.9 %; a fifth stream of its own: the same +1.9 %; stream priorities make no
        # measurable difference: profiles/r06_stream_priority_ab.txt.)
        side = shared_stream(self.device, "target")
        st = self.__dict__.setdefault("_wg", dict(events=[], used=0, last=None, stream=side))
        ready, done = self._wg_event(), self._wg_event()
        ready.record(main)
        side.wait_event(ready)
        with torch.cuda.stream(side), fn.workspace_lane(2):
            conv._weight_gradient(dy, planes, xa, dya)
            done.record(side)
        _WG_READS[dy.data_ptr()] = (done, self.device)
        st["last"] = done
        return True

    def wgrad_beside_join(self, final=False):
        st = self.__dict__.get("_wg")
        if st and st["last"] is not None:
            torch.cuda.current_stream(self.device).wait_event(st["last"])
            st["last"] = None
        if st and final:
            st["used"] = 0
            for k in [k for k, v in _WG_READS.items() if v[1] == self.device]:
                del _WG_READS[k]

    def join_side_backward(self):
        self.wgrad_beside_join(final=True)
        sb = self.side_bwd
        if sb is not None and sb["dirty"]:
            torch.cuda.current_stream(self.device).wait_event(sb["prog_ev"])
            sb["dirty"] = False

    def side_backward_busy(self, idx):
        """True while gradients of the branch may still be in flight on the side stream (the solver holds its slab reductions)"""
        sb = self.side_bwd
        return sb is not None and sb["active"] and sb["dirty"] and idx >= sb["lo"]

    def join_side(self):
        """order the current stream behind the side segment's forward (idempotent; called by the first reader of its results)"""
        if self.side_pending:
            torch.cuda.current_stream(self.device).wait_event(self.side_segment[4])
            self.side_pending = False

    def begin_backward(self):
        """reset the gradient bookkeeping; all data-gradient operands (transposed weights) in one launch"""
        for t in self.all_tensors:
            t._gw = False
            t.grad = None
            t.grad_planes = False
        if self.side_bwd is not None:
            assert not self.side_bwd["dirty"], "the previous backward pass left side-stream work unjoined"
            self.side_bwd["forked"] = False
        self.guard["have_stats"] = True      # (the spans of a whole pass -- inputs, weights, gradients -- exist once this one has run)
        if self.scalars is not None:
            if self._am_bwd_ran:       # a second backward pass on the same forward pass: new gradients, new magnitudes
                for slot in self._am_bwd_slots & self._am_done:
                    self.scalar(slot).zero_()
                self._am_done -= self._am_bwd_slots
            self._am_bwd_ran = True
        if self.half_operands:
            return                     # prepared by forward(); the weights have not changed since
        self.wt_batched = self.wt_table is not None
        if self.wt_batched:
            fn.weight_transpose_batch(*self.wt_table)

    def backward(self):
        self.begin_backward()
        for idx in range(len(self.nodes) - 1, -1, -1):
            self.backward_node(idx)
        self.join_side_backward()
        self.flush_slabs()

    def num_params(self):
        return sum(p.size for p in self.param_order)

    # -- checkpoint exchange in the reference's shapes (mx.model.load_checkpoint -> arg_params) ----------------
    def set_params(self, arg_params, allow_missing=False, allow_extra=True):
        """arg_params: name -> numpy array in the reference's shape (Convolution weight (Cout, Cin, kh, kw),
        Deconvolution weight (in, out, kh, kw), vectors (C,)).  Converted to the device layout ([Cout][kh][kw][Cin
        padded to the input tensor's physical channels], pad channels zero) and copied into the arena.
        Shape mismatches raise; names the graph has no parameter for (the `*_gamma` of fix_gamma BatchNorms, the moving
        statistics, ...) are ignored unless allow_extra=False; graph parameters without a value keep their
        current contents if allow_missing else raise (Module.set_params semantics)."""
        missing = [p.name for p in self.param_order if p.name not in arg_params]
        if missing and not allow_missing:
            raise KeyError("set_params: no value for " + ", ".join(missing[:8]) + (" ..." if len(missing) > 8 else ""))
        extra = [k for k in arg_params if k not in self.params]
        if extra and not allow_extra:
            raise KeyError("set_params: graph has no parameter " + ", ".join(extra[:8]))
        for p in self.param_order:
            if p.name not in arg_params:
                continue
            v = np.asarray(arg_params[p.name], np.float32)
            logical = tuple(p.logical or p.shape)
            if tuple(v.shape) != logical:
                raise ValueError("set_params: %s has shape %s, the graph expects %s" % (p.name, tuple(v.shape), logical))
            dev = np.zeros(p.shape, np.float32)
            if p.kind == "raw":                                   # same values, only the shape differs ((1,6) <-> (6,))
                dev[:] = v.reshape(p.shape)
            elif p.kind in ("conv", "deconv"):
                t = v.transpose(0, 2, 3, 1)                       # -> [rows][kh][kw][cols]
                dev[:t.shape[0], :, :, :t.shape[3]] = t
            else:
                if v.shape[0] < p.shape[0]:                       # padded vector: keep the pad lanes as they are
                    dev[:] = p.data.detach().cpu().numpy()
                dev[:v.shape[0]] = v
            p.data.copy_(torch.from_numpy(dev))

    def get_params(self):
        """-> name -> numpy array in the reference's shapes (inverse of set_params; pad channels dropped)"""
        out = {}
        for p in self.param_order:
            v = p.data.detach().cpu().numpy()
            logical = tuple(p.logical or p.shape)
            if p.kind == "raw":
                v = v.reshape(logical)
            elif p.kind in ("conv", "deconv"):
                v = v[:logical[0], :, :, :logical[1]].transpose(0, 3, 1, 2)
            else:
                v = v[:logical[0]]
            out[p.name] = np.ascontiguousarray(v)
        return out


# ------------------------------------------------------------------ initialisers
def init_zeros(rng, shape):
    return np.zeros(shape, np.float32)


def init_ones(rng, shape):
    return np.ones(shape, np.float32)


def conv_weight_init(kind, cin_logical):
    """"maxdim": multi_init.py:74-76, U(+-1/sqrt(max(shape))) on the MXNet shape (Cout, Cin, kh, kw); "xavier":
    mx.init.Xavier(uniform, avg, magnitude 3).  Device layout (Cout, kh, kw, Cin_phys); pad channels stay 0."""
    def f(rng, shape):
        cout, r, s, cin_p = shape
        w = np.zeros(shape, np.float32)
        if kind == "maxdim":
            lim = 1.0 / math.sqrt(max(cout, cin_logical, r, s))
        else:  # "xavier": mx.init.Xavier(uniform, avg, magnitude 3) (train/train_multitask.py:313)
            fan_in, fan_out = cin_logical * r * s, cout * r * s
            lim = math.sqrt(3.0 / ((fan_in + fan_out) / 2.0))
        w[..., :cin_logical] = rng.uniform(-lim, lim, size=(cout, r, s, cin_logical))
        return w
    return f


def deconv_bilinear_init(channels):
    """multi_init.py:160-168 + upsample_filt: a diagonal bilinear kernel.  Device layout of the
    Deconvolution weight is [K = in channel][R][S][C = out channel] (phys channels)."""
    def f(rng, shape):
        k_p, r, s, c_p = shape
        factor = (r + 1) // 2
        center = factor - 1 if r % 2 == 1 else factor - 0.5
        og = np.ogrid[:r, :s]
        filt = (1 - abs(og[0] - center) / factor) * (1 - abs(og[1] - center) / factor)
        w = np.zeros(shape, np.float32)
        for i in range(channels):
            w[i, :, :, i] = filt
        return w
    return f


def init_affine_identity(rng, shape):
    """multi_init.py:72: affine_matrix = [[1, 0, 0, 0, 1, 0]]"""
    return np.array([1, 0, 0, 0, 1, 0], np.float32)


def affine_matrix_param(g, name="affine_matrix"):
    """mx.sym.var("affine_matrix", shape=(1,6)) (multitask_symbol_builder.py:574): an ordinary argument -- it gets
    a gradient through GridGenerator / BilinearSampler and the optimizer updates it (multi_solver.py:205-208,
    291-293), weight decay included.  Device shape (6,), checkpoint shape (1, 6)."""
    p = g.param(name, (6,), init_affine_identity)
    p.logical, p.kind = (1, 6), "raw"
    return p


# ------------------------------------------------------------------ nodes
class InputNCHW(Node):
    """data (B,3,H,W) -> NHWC with channels padded to 4 (symbol/resnet.py:89 `data`)"""

    def __init__(self, g, src, name="data_nhwc"):
        self.src = src
        N, C, H, W = src.shape
        self.out = g.tensor((N, H, W, fn.padc(C)), name, requires_grad=False)
        self.out.channels = C

    def forward(self):
        fn.nchw_to_nhwc(self.src.data, out=self.out.data)


# Graph._plan_input_planes: a deferred BatchNorm output is ALSO written as piece planes when its readers would otherwise apply
# the affine to each element at least this many times (taps x 128-column tiles, summed over the readers)
FUSE_MAXPOOL_BACKWARD = _os.environ.get("DSPN_FUSE_POOL_BWD", "1") != "0"      # (A/B switch)
X_PLANES_MIN_READS = int(_os.environ.get("DSPN_X_PLANES_MIN_READS", "6"))


class BatchNorm(Node):
    """mx.sym.BatchNorm with batch statistics (the solver always runs is_train=True,
    multi_solver.py:284) optionally fused with the following ReLU."""

    def __init__(self, g, x, name, fix_gamma=False, eps=2e-5, relu=False, beta_grad_from_consumer=False,
                 defer_apply=False):
        """defer_apply: the output is only consumed by convolutions, which apply scale/shift(+ReLU) to x in their
        tile loaders (dspn_conv2d_forward_bn_f32 / dspn_conv2d_wgrad_bn_f32): forward is the statistics pass alone and
        the normalised tensor is never written"""
        C = x.shape[-1]
        if not FUSE_BATCHNORM:
            defer_apply = False
        self.x, self.eps, self.relu = x, eps, relu
        self.gamma = None if fix_gamma else g.param(name + "_gamma", (C,), init_ones)
        self.beta = g.param(name + "_beta", (C,), init_zeros)
        for p in (self.gamma, self.beta):
            if p is not None:
                p.logical = (x.channels or C,)
        g.bn_names.append((name, x.channels or C, bool(fix_gamma)))   # gamma / moving_* entries of a checkpoint
        self.mean = fn.zeros(C, device=g.device)
        self.rstd = fn.zeros(C, device=g.device)
        self.scale = fn.zeros(C, device=g.device)
        self.shift = fn.zeros(C, device=g.device)
        # an input BN whose data has no gradient only needs sum(dy) for beta: its consumer supplies it
        self.defer_apply = defer_apply
        # statistics gathered by the producing convolution's epilogue (one (mean, M2) pair per row tile) instead of a
        # separate pass over x
        self.tile_stats = (x.producer.enable_out_stats()
                           if FUSE_BATCHNORM and getattr(x, "producer", None) is not None else None)
        self.out = g.tensor(x.shape, name + ("_relu" if relu else "_out"), requires_grad=not beta_grad_from_consumer,
                            virtual=bool(defer_apply))
        if defer_apply:
            self.out.affine_src = (x, self.scale, self.shift, relu)
        self.out.bn_node = self
        # "f16x2" math: a MATERIALISED output leaves its magnitude as a by-product of the apply kernel
        self.am_out = g.new_scalar() if (g.math == "f16x2" and self.out.dtype == torch.float32) else None
        self.out.am_slot = self.am_out
        self.conv_consumers = []     # convolutions reading self.out (deferred apply), in forward order
        self.mat_consumers = []      # plain convolutions reading a MATERIALISED self.out (their last data gradient can
                                     # still gather this BatchNorm's backward reductions in its epilogue)
        self.bwd_sums = None         # (buffer, tiles) written by the LAST data gradient into self.out.grad
        self.bwd_sums_ready = False
        # round 4 (Graph._plan_gradient_planes): dx leaves as fp16 piece planes; x_ext = per-channel extremes of x from the
        # forward finalize, am_dyin = slot of the magnitude of this node's own output gradient (from the data gradient's epilogue)
        self.dx_planes, self.x_ext, self.am_dyin = False, None, None
        # round 4 (Graph._plan_input_planes): the output as fp16 piece planes for the multi-tap convolutions behind a DEFERRED
        # apply; planes_ready: written by this step's forward (the magnitude they are cut by came out of the finalize)
        self.planes, self.planes_ready = None, False
        self.pool_grad = None        # (argmax, pooled gradient, k, stride, pad) left by a MaxPool that folds this node (one backward)
        self._g = g
        # True when this node's backward is the LAST writer of x's gradient (set by Graph.finalize): only then is the dx it
        # stores the complete gradient whose magnitude the producing convolution may use
        self.completes_x_grad = False
        # round 6: the finalize half of this node's backward is parked for / rode in the weight gradient of the convolution
        # whose data gradient gathered the sums: (args, kwargs) of the apply half, or None
        self._pending = None

    def forward(self):
        if self.tile_stats is not None:
            buf, tiles, tile_rows = self.tile_stats
            rows = int(np.prod(self.x.shape[:-1]))
            # "f16x2" math: the convolutions that fold this BatchNorm into their loaders multiply (relu)(x * scale + shift);
            # its magnitude comes out of this finalize kernel, from the extremes the producer wrote beside the statistics
            g, mm, am = self._g, getattr(self.x.producer, "out_minmax", None), None
            slot = g._am_x.get((id(self.x), id(self.scale))) if (mm is not None and g.scalars is not None) else None
            if slot is not None:
                am = g.scalar(slot)
                g._am_done.add(slot)
            fn.bn_stats_from_tiles(buf, tiles, tile_rows, rows, self.x.shape[-1], self.eps,
                                   None if self.gamma is None else self.gamma.data, self.beta.data,
                                   self.mean, self.rstd, self.scale, self.shift,
                                   tile_minmax=mm if am is not None else None, relu=self.relu, out_absmax=am,
                                   out_absmin=None if am is None else g.scalars_min[slot:slot + 1],
                                   out_chan_minmax=self.x_ext if am is not None else None)
            self.planes_ready = self.planes is not None and am is not None
            if self.planes_ready:
                fn.bn_apply_planes(self.x.data, self.scale, self.shift, am, relu=self.relu, out=self.planes)
        else:
            fn.bn_stats(self.x.data, self.eps, None if self.gamma is None else self.gamma.data, self.beta.data,
                        self.mean, self.rstd, self.scale, self.shift)
        if not self.defer_apply:
            am = self._g.scalar(self.am_out)
            fn.bn_apply(self.x.data, self.scale, self.shift, relu=self.relu, out=self.out.data, out_absmax=am)
            if am is not None:
                self._g._am_done.add(self.am_out)

    def _x_ext_valid(self):
        """the per-channel extremes of x were written by this step's forward finalize (the magnitude slot they come with is done)"""
        g = self._g
        slot = g._am_x.get((id(self.x), id(self.scale)))
        return slot is not None and slot in g._am_done

    def finalize_beside(self):
        """Round 6: called by the convolution whose data gradient has just gathered this node's backward
        sums in its epilogue, BEFORE it launches its weight gradient.  The finalize half of dspn_bn_backward_from_sums (one or
        two launches on 1 - 64 workgroups, 6 - 15 us of an otherwise idle chip) is PARKED in the library and rides in front of
        that weight gradient's grid (csrc/bn_final_job.h); backward() later launches the apply half.  (On a second stream
        instead -- two events per node -- the step LOST 1.5 %: profiles/r06_finalize_beside_ab.txt.)  -> True if it did."""
        g = self._g
        if (not FINALIZE_BESIDE or g.device.type != "cuda" or not self.bwd_sums_ready or self.pool_grad is not None
                or not self.out._gw or self._pending is not None):
            return False
        self.backward(beside=True)
        return self._pending is not None

    def _from_sums(self, beside, args, kw):
        if not beside:
            fn.bn_backward_from_sums(*args, **kw)
            return
        fn.bn_backward_from_sums(*args, phase=1, park=True, **kw)
        self._pending = (args, kw)

    def backward(self, beside=False):
        if self._pending is not None:
            args, kw = self._pending
            self._pending = None
            fn.bn_backward_from_sums(*args, phase=2, **kw)
            return
        if not self.out._gw:
            return
        am = None
        if self.x.requires_grad:
            dx, acc = self.x.grad_target()
            # "f16x2" math: dx is the output gradient of the convolution that produced x; when this call completes it (a
            # tensor read by this BatchNorm alone, or the residual stream, whose last writer in backward order is the next
            # unit's first BatchNorm), its magnitude comes out of the apply kernel instead of a pass of its own
            prod, g = getattr(self.x, "producer", None), self._g
            if (isinstance(prod, Conv) and prod.am_dy is not None and g.scalars is not None and dx.dtype == torch.float32
                    and self.completes_x_grad):
                am = g.scalar(prod.am_dy)
                g._am_done.add(prod.am_dy)
        else:  # parameters still need their gradients; dx goes to scratch
            dx, acc = (self.out.grad if self.pool_grad is None else self.out.own_grad()), False
        if self.pool_grad is not None:
            argmax, dyp, k, s, p = self.pool_grad
            self.pool_grad = None
            if not acc:
                fn.bn_backward_maxpool(self.x.data, self.scale, self.shift, dyp, argmax, k, s, p, self.mean, self.rstd,
                                       None if self.gamma is None else self.gamma.data, relu=self.relu, dx=dx,
                                       dgamma=None if self.gamma is None else self.gamma.grad, dbeta=self.beta.grad, dx_absmax=am)
                return
            # (a second writer of x's gradient: materialise the pooling backward after all)
            fn.maxpool_backward_argmax(argmax, dyp, self.out.shape, k, s, p, dx=self.out.own_grad())
            self.out.grad = self.out.own_grad()
        if (self.bwd_sums_ready and self.dx_planes and am is not None and not acc and self._x_ext_valid()
                and not getattr(prod, "guard_fb", False)):      # (a convolution on the guard's fallback reads a FLOAT gradient)
            # the gradient leaves as fp16 piece planes (same buffer): `am` receives the BOUND it is cut by
            self.bwd_sums_ready = False
            self._from_sums(beside, (self.x.data, self.scale, self.shift, self.out.grad, self.mean, self.rstd,
                                     None if self.gamma is None else self.gamma.data, self.bwd_sums[0], self.bwd_sums[1]),
                            dict(relu=self.relu, dx=dx, dgamma=None if self.gamma is None else self.gamma.grad,
                                 dbeta=self.beta.grad, accumulate=False, dx_absmax=am,
                                 dy_absmax=self._g.scalar(self.am_dyin), x_chan_minmax=self.x_ext, dx_planes=True,
                                 dx_absmin=g.scalars_min[prod.am_dy:prod.am_dy + 1] if g.guard["enabled"] else None))
            self.x.grad_planes = True
            return
        if self.bwd_sums_ready:      # the two reductions came out of the data-gradient kernel's epilogue
            self.bwd_sums_ready = False
            self._from_sums(beside, (self.x.data, self.scale, self.shift, self.out.grad, self.mean, self.rstd,
                                     None if self.gamma is None else self.gamma.data, self.bwd_sums[0], self.bwd_sums[1]),
                            dict(relu=self.relu, dx=dx, dgamma=None if self.gamma is None else self.gamma.grad,
                                 dbeta=self.beta.grad, accumulate=acc, dx_absmax=am))
            return
        assert not beside, "finalize_beside() without the data gradient's sums"
        fn.bn_backward(self.x.data, self.scale, self.shift, self.out.grad, self.mean, self.rstd,
                       None if self.gamma is None else self.gamma.data, relu=self.relu, dx=dx,
                       dgamma=None if self.gamma is None else self.gamma.grad, dbeta=self.beta.grad,
                       accumulate=acc, dx_absmax=am)


# round 6: the weight gradients of the step's stream on a stream of their own (Graph.wgrad_beside): nothing on the critical
# chain data gradient -> BatchNorm finalize -> apply -> next data gradient reads them, so they run BESIDE it -- in the gaps of
# the latency-bound finalize launches and under the HBM-bound apply passes -- and the bucket's slab sums follow them there;
# the step's stream waits for that stream once, at the end of the pass (and before an accumulation into a buffer one of
# them still reads: _wg_before_write).  Same kernels, same bits; +1.9 % on the step (profiles/r06_wgrad_beside_ab.txt).
# DSPN_WGRAD_SIDE=0: on the step's stream, with the BatchNorm finalize riding in their launches (below).  bench.py switches
# it off for its instrumented steps: a kernel timed beside another one measures the pair, not the kernel.
WGRAD_SIDE = int(_os.environ.get("DSPN_WGRAD_SIDE", "1"))
WGRAD_SIDE_MIN_US = float(_os.environ.get("DSPN_WGRAD_SIDE_MIN_US", "30"))      # (Conv._wgrad_worth_a_stream)

# round 6: the finalize half of a BatchNorm backward rides in the weight-gradient launch of the layer behind it
# (DSPN_FINALIZE_BESIDE=0: launches of its own, as round 5 -- same-box A/B; the results do not depend on it)
FINALIZE_BESIDE = _os.environ.get("DSPN_FINALIZE_BESIDE", "1") != "0"

# round 5: convolutions no BatchNorm reads leave the magnitude of their output (forward epilogue) and of their masked gradient
# (the ReLU-backward / bias-gradient pass) as by-products; DSPN_CONV_MAGNITUDES=0 keeps the stand-alone passes (same-box A/B)
FUSE_CONV_MAGNITUDES = _os.environ.get("DSPN_CONV_MAGNITUDES", "1") != "0"


def xa_unavailable(conv):
    """the output-magnitude epilogue needs a dense output of whole float4 rows (every Conv output of the engine is one)"""
    return conv.out.data is None or conv.out.data.shape[-1] % 4 != 0


class Conv(Node):
    """mx.sym.Convolution (+ bias) (+ ReLU epilogue); weight [Cout, R, S, Cin_phys]"""

    def __init__(self, g, x, name, num_filter, kernel, stride=1, pad=0, dilate=1, no_bias=True, relu=False,
                 init="xavier", cin_logical=None, cout_phys=None, input_sum_grad=None, residual=None, out_name=None,
                 tap_expand=False):
        N, H, W, Cin = x.shape
        kh, kw = fn._hw(kernel)
        ph, pw = fn._hw(pad)
        self.x, self.stride, self.pad, self.dil, self.relu = x, stride, (ph, pw), dilate, relu
        # input produced by a deferred BatchNorm: read the raw tensor and apply the affine in the loader
        self.x_raw, self.in_affine = x, None
        if x.affine_src is not None:
            raw, sc, sh, arelu = x.affine_src
            self.x_raw, self.in_affine = raw, (sc, sh, arelu)
            assert not tap_expand
            x.bn_node.conv_consumers.append(self)
        elif getattr(x, "bn_node", None) is not None and not tap_expand:
            x.bn_node.mat_consumers.append(self)
        self.cout = num_filter
        # Param that receives sum_pixels(dx) instead of a full data gradient (see conv2d_input_sum_grad)
        self.input_sum_grad = input_sum_grad
        cin_meta = cin_logical if cin_logical is not None else (x.channels or Cin)   # channels a checkpoint holds
        cin_logical = Cin if cin_logical is None else cin_logical
        self.w = g.param(name + "_weight", (num_filter, kh, kw, Cin), conv_weight_init(init, cin_logical))
        self.b = None if no_bias else g.param(name + "_bias", (num_filter,), init_zeros)
        self.w.logical, self.w.kind = (num_filter, cin_meta, kh, kw), "conv"
        if self.b is not None:
            self.b.logical = (num_filter,)
        Ho, Wo = fn.conv_out_size(H, kh, stride, ph, dilate), fn.conv_out_size(W, kw, stride, pw, dilate)
        ldc = fn.padc(num_filter) if cout_phys is None else cout_phys
        self.out = g.tensor((N, Ho, Wo, ldc), out_name or (name + "_out"))
        self.out.channels = num_filter
        self.out.producer = self
        self.slabs, self.slabs_fresh = None, False     # deferred split-K slab reduction (Graph.flush_slabs)
        self.out_stats = None      # (buffer, tiles, rows per tile) once a BatchNorm asked for them
        self.out_minmax = None     # "f16x2" math: (tiles, 2, Cout) smallest / largest output value per tile (with out_stats)
        self._g = g
        # residual: a tensor of the output's shape added in the conv epilogue (`conv3 + shortcut`,
        # symbol/resnet.py:51); its gradient is the output gradient itself
        self.residual = residual
        assert residual is None or residual.shape == self.out.shape
        half = self.out.dtype == torch.bfloat16
        # data-gradient operand: the transposed weight -- as piece planes (wtp, below) where the split math contracts over whole
        # 32-channel blocks, else as a tensor of the storage type (no float transpose is kept beside planes: ~100 MB on resnet-50)
        self.wt_shape = (Cin, kh, kw, ldc)
        planes_t = g.device.type == "cuda" and x.requires_grad and fn.needs_planes(self.out.dtype, ldc, g.math)
        self.wt = None if (not x.requires_grad or planes_t) else fn.zeros(Cin, kh, kw, ldc, device=g.device, dtype=self.out.dtype)
        # bf16 operands: the copy of the float master the forward pass multiplies (refreshed once per step, Graph.forward)
        self.wh = fn.zeros(num_filter, kh, kw, Cin, device=g.device, dtype=torch.bfloat16) if half else None
        # split math: piece planes of the weight (forward operand) and of its transpose (data-gradient operand), cut once
        # per step by Graph.forward for the whole graph instead of once per tile inside the kernels
        self.math = g.math
        # "f16x2" math: slots of the operand magnitudes.  Readers of one (raw tensor, affine) pair share the input's slot
        # (act1 feeds conv1 and the projection shortcut); the gradient's and the weight's belong to this node.
        self.am_x = self.am_dy = self.am_w = None
        if g.math == "f16x2" and self.out.dtype == torch.float32:
            key = (id(self.x_raw), None if self.in_affine is None else id(self.in_affine[0]))
            if key not in g._am_x:
                g._am_x[key] = g.new_scalar()
            self.am_x, self.am_dy, self.am_w = g._am_x[key], g.new_scalar(backward=True), g.new_scalar()
        # ... and of the OUTPUT as stored, where no BatchNorm reads it (no statistics epilogue: vgg16_reduced, the SSD extra layers):
        # the convolution's own epilogue leaves it (fn.conv2d_forward out_absmax), the next convolution needs no pass over the tensor
        self.am_out = g.new_scalar() if (self.am_x is not None and not tap_expand) else None
        if self.am_out is not None:
            self.out.am_slot = self.am_out       # (pooled / aliased tensors built on it inherit the slot)
        if self.am_x is not None and self.in_affine is None:
            g.want_magnitude(self.x_raw)         # the producer of this input leaves its magnitude if it can
        self.x_planes_bn = None      # the deferred BatchNorm that also leaves this node's input as piece planes (Graph._plan_input_planes)
        self.guard_fb = False        # range guard (Graph._update_guard): this pass's calls run in the three-piece bf16 math
        self.wp = self.wtp = None
        if g.device.type == "cuda":
            npc = fn.plane_pieces(g.math)
            if fn.needs_planes(self.out.dtype, Cin, g.math):
                self.wp = fn.zeros(num_filter, kh * kw, Cin // 32, npc, 32, device=g.device, dtype=torch.bfloat16)
            if x.requires_grad and fn.needs_planes(self.out.dtype, ldc, g.math):
                self.wtp = fn.zeros(Cin, kh * kw, ldc // 32, npc, 32, device=g.device, dtype=torch.bfloat16)
        # tap-expanded evaluation (few output channels, stride 1): 1x1 convolution to Cout*kh*kw channels
        # on the same weight buffer + shifted sum over taps (include/dspn_nn.h, dspn_tap_sum_f32)
        self.tap_expand = bool(tap_expand) and kh * kw > 1
        if self.tap_expand:
            assert stride == 1 and dilate == 1 and not relu and residual is None
            assert (Ho, Wo) == (H, W), "tap expansion needs a 'same' convolution"
            self.z = fn.act_zeros(N, H, W, fn.padc(num_filter * kh * kw), device=g.device)   # also holds dz in backward
        # algorithmic FLOPs per batch (direct-conv count, logical channels)
        self.flops_fwd = 2.0 * cin_logical * num_filter * kh * kw * Ho * Wo * N
        self.flops_bwd = self.flops_fwd * (2 if x.requires_grad else 1)

    def alloc_slabs(self):
        cout, kh, kw, cin = self.w.shape
        wshape = (cout * kh * kw, 1, 1, cin) if self.tap_expand else self.w.shape
        xs = self.x.shape
        splits = fn.conv2d_wgrad_splits(xs, self.z.shape if self.tap_expand else self.out.shape, wshape,
                                        1 if self.tap_expand else self.stride)
        if splits > 0 and self.w.data.numel() % 4 == 0:
            self.slabs = fn.zeros(splits, self.w.data.numel(), device=self._g.device)

    def wop(self):
        """the weight operand of the forward kernels: the float master, or its bf16 copy"""
        return self.w.data if self.wh is None else self.wh

    def enable_out_stats(self):
        """called by a BatchNorm on self.out: have the epilogue write per-tile statistics (None if unavailable)"""
        if self.tap_expand or self.out.shape[3] != self.cout or self.cout % 4 != 0:
            return None
        if self.out_stats is None:
            rows = self.out.shape[0] * self.out.shape[1] * self.out.shape[2]
            tiles, tile_rows = fn.conv_stats_layout(rows, self.cout)
            if tiles == 0:
                return None
            self.out_stats = (fn.zeros(tiles, 2, self.cout, device=self._g.device), tiles, tile_rows)
            if self.math == "f16x2" and self.out.dtype == torch.float32:
                # per-tile extremes of the output next to its statistics: the magnitude of BatchNorm(+ReLU)(out), which
                # the next convolution multiplies, is then read off this table instead of off the whole tensor
                self.out_minmax = fn.zeros(tiles, 2, self.cout, device=self._g.device)
        return self.out_stats

    def _magnitudes(self, which):
        """device scalars of this node's operand magnitudes ("f16x2" math; None otherwise).  `x`: computed by the first
        reader of the (tensor, affine) pair in a step; `dy`: by backward(); `w`: by Graph.forward for all weights."""
        g = self._g
        if self.am_x is None or g.scalars is None:
            return None
        if which == "x" and self.am_x not in g._am_done:
            src = self.x_raw
            while src.alias_of is not None:       # BlockGrad: the same values under another name
                src = src.alias_of
            prod = getattr(src, "producer", None)
            # the producer's per-tile extremes of x_raw (written when some BatchNorm reads it): with an affine, its extremes
            # sit at the extremes of x (monotone per channel); without one, |x| is largest at one of them
            table = getattr(prod, "out_minmax", None) if getattr(prod, "out", None) is src else None
            if table is not None:
                fn.absmax(table.view(-1, table.shape[-1]), self.in_affine, out=g.scalar(self.am_x))
            elif src.am_slot is not None and src.am_slot in g._am_done:
                # round 4: a bound the producer left while it wrote the tensor (or the bound of what it was pooled from)
                if self.in_affine is None:
                    return g.scalar(src.am_slot)
                fn.absmax_affine_bound(self.in_affine[0], self.in_affine[1], g.scalar(src.am_slot), g.scalar(self.am_x))
            else:
                fn.absmax(self.x_raw.data, self.in_affine, out=g.scalar(self.am_x))
            g._am_done.add(self.am_x)
        return g.scalar({"x": self.am_x, "dy": self.am_dy, "w": self.am_w}[which])

    def forward(self):
        xa, wa = self._magnitudes("x"), self._magnitudes("w")
        if self.tap_expand:
            cout, kh, kw, cin = self.w.shape
            fn.conv2d_forward(self.x.data, self.wop().view(cout * kh * kw, 1, 1, cin), None, 1, 0, 1, out=self.z,
                              w_planes=self.wp, math=self.math,     # (the planes of [Cout][taps][..] ARE those of the view)
                              x_absmax=xa, w_absmax=wa)
            fn.tap_sum(self.z, None if self.b is None else self.b.data, cout, kh, kw, self.pad, out=self.out.data)
            return
        if self.guard_fb:
            return self._forward_fallback()
        xp = self._x_planes()
        oa = self._out_magnitude()
        fn.conv2d_forward(self.x_raw.data if xp is None else xp, self.wop(), None if self.b is None else self.b.data, self.stride,
                          self.pad, self.dil, relu=self.relu, out=self.out.data,
                          residual=None if self.residual is None else self.residual.data,
                          in_affine=self.in_affine if xp is None else None,
                          out_stats=None if self.out_stats is None else self.out_stats[0], w_planes=self.wp,
                          math=self.math, x_absmax=xa, w_absmax=wa, out_minmax=self.out_minmax, x_planes=xp is not None,
                          out_absmax=oa)
        if oa is not None:
            self._g._am_done.add(self.am_out)

    def _out_magnitude(self):
        """the magnitude block this call's epilogue fills (None where a BatchNorm takes statistics -- its finalize hands the
        magnitude on -- or the math is not "f16x2")"""
        g = self._g
        if (not FUSE_CONV_MAGNITUDES or self.am_out is None or g.scalars is None or self.out_stats is not None
                or self.am_out not in g._am_wanted or xa_unavailable(self)):
            return None
        return g.scalar(self.am_out)

    def _x_planes(self):
        """the input as piece planes, when the BatchNorm in front wrote them this step (cut by this node's x magnitude)"""
        bn = self.x_planes_bn
        if self.guard_fb:
            return None
        return bn.planes if (bn is not None and bn.planes_ready and self.am_x in self._g._am_done) else None

    def _w3(self, transposed):
        """the weight operand of a fallback call: three bf16 piece planes cut on the spot (None where the three-piece kernels
        read the float weights: contractions that are not whole 32-channel blocks)"""
        cols = self.wt_shape[3] if transposed else self.w.shape[3]
        if cols % 32 != 0:
            return None
        return fn.weight_planes(self.w.data, transposed=transposed, cols=cols, math="bf16x3")

    def _forward_fallback(self):
        """range guard: the same convolution in the three-piece bf16 math, on the float input (affine in the loader); the
        per-tile extremes the two-piece consumers take their magnitudes from are filled by a pass of their own"""
        fn.conv2d_forward(self.x_raw.data, self.wop(), None if self.b is None else self.b.data, self.stride, self.pad, self.dil,
                          relu=self.relu, out=self.out.data, residual=None if self.residual is None else self.residual.data,
                          in_affine=self.in_affine, out_stats=None if self.out_stats is None else self.out_stats[0],
                          w_planes=self._w3(False), math="bf16x3")
        if self.out_minmax is not None:
            fn.tile_minmax(self.out.data, self.out_stats[2], self.out_minmax)
        oa = self._out_magnitude()
        if oa is not None:
            fn.absmax(self.out.data, out=oa)
            self._g._am_done.add(self.am_out)
        self._g.guard["calls"] += 1
        self._g.guard["calls_total"] += 1

    def backward(self):
        if not self.out._gw:
            return
        dy = self.out.grad
        planes = self.out.grad_planes            # fp16 piece planes from the BatchNorm behind this convolution (never with relu / bias / residual)
        if self.relu and self.b is not None:     # ReLU mask and bias gradient in one pass over dy
            # "f16x2" math: that pass also leaves the magnitude of the masked gradient, which both of its readers cut it by
            dya0 = None
            if (FUSE_CONV_MAGNITUDES and self.am_dy is not None and self._g.scalars is not None
                    and self.am_dy not in self._g._am_done and dy.dtype == torch.float32):
                dya0 = self._g.scalar(self.am_dy)
            fn.relu_backward_colsum(self.out.data, dy, self.cout, dx=dy, out=self.b.grad, dx_absmax=dya0)
            if dya0 is not None:
                self._g._am_done.add(self.am_dy)
        elif self.relu:
            fn.relu_backward(self.out.data, dy, dx=dy)
        if self.residual is not None and self.residual.requires_grad:
            self.residual.give_grad(dy)
        if self.b is not None and not self.relu:
            fn.colsum(dy, self.cout, out=self.b.grad)
        # "f16x2" math: the gradient's magnitude once (after the in-place ReLU mask above), for both of its readers
        xa, dya, wa = self._magnitudes("x"), self._magnitudes("dy"), self._magnitudes("w")
        if dya is not None and self.am_dy not in self._g._am_done:       # (else: the BatchNorm backward that completed dy took it)
            fn.absmax(dy, out=dya)
            self._g._am_done.add(self.am_dy)      # (a projection shortcut that shares this slot reads the same gradient)
        # round 6: when this data gradient gathers the backward sums of the BatchNorm in front (bn_bwd_node), it goes FIRST and
        # that node's finalize rides in front of the weight gradient's grid (BatchNorm.finalize_beside)
        bn = getattr(self, "bn_bwd_node", None) if self.x.requires_grad else None
        early = (bn is not None and FINALIZE_BESIDE and not self.guard_fb and self._g.device.type == "cuda"
                 and bn.pool_grad is None)
        if (WGRAD_SIDE and self._g.wgrad_side_allowed and self._g.device.type == "cuda" and (bn is not None or self._g.batchnorm_chain()) and self._wgrad_worth_a_stream()
                and self._g.wgrad_beside(self, dy, planes, xa, dya)):
            early = False          # (the weight gradient went to its own stream: nothing to ride in, and the finalize's gap is filled)
            if self.input_sum_grad is not None:
                fn.conv2d_input_sum_grad(dy, self.w.data, self.x.shape, self.stride, self.pad, self.dil,
                                         out=self.input_sum_grad.grad)
            if self.x.requires_grad:
                self._data_gradient(dy, planes, dya, wa)
            return
        if early:
            self._data_gradient(dy, planes, dya, wa)
            bn.finalize_beside()
        self._weight_gradient(dy, planes, xa, dya)
        if self.input_sum_grad is not None:
            fn.conv2d_input_sum_grad(dy, self.w.data, self.x.shape, self.stride, self.pad, self.dil,
                                     out=self.input_sum_grad.grad)
        if self.x.requires_grad and not early:
            self._data_gradient(dy, planes, dya, wa)

    def _wgrad_worth_a_stream(self):
        """The weight gradient goes beside the chain only where that pays: behind a data gradient that feeds a BatchNorm
        backward (its finalize launches leave gaps, its apply pass is HBM-bound; the caller checks that) and where the kernel is
        long enough to carry the two events it costs.  Measured with every weight gradient beside: resnet-50 +1.9 %,
        vgg16_reduced (no BatchNorm: the chain is MFMA-bound like the weight gradients) -3.4 %, inceptionv3 1024 x 512 bs 8
        (launches of ~15 us) -7 % (profiles/r06_wgrad_beside_other_configs.txt).  Estimate: the larger of multiply-adds at
        250 TFLOP/s and operand bytes at 4 TB/s, at least WGRAD_SIDE_MIN_US."""
        w = getattr(self, "_wg_worth", None)
        if w is None:
            P = float(np.prod(self.out.shape[:-1]))
            cout, kh, kw, cin = self.w.shape
            esz = 4.0 if self.out.dtype == torch.float32 else 2.0
            est = max(2.0 * P * cout * kh * kw * cin / 2.5e14, esz * (P * cout + float(np.prod(self.x.shape))) / 4e12)
            w = self._wg_worth = est * 1e6 >= WGRAD_SIDE_MIN_US
        return w

    def _weight_gradient(self, dy, planes, xa, dya):
        if self.tap_expand:
            cout, kh, kw, cin = self.w.shape
            fn.tap_spread(dy, cout, kh, kw, self.pad, out=self.z)
            # (the weight gradient multiplies the SPREAD gradient z: its magnitude is taken on the fly, these are small layers)
            if self.slabs is not None:
                fn.conv2d_wgrad_slabs(self.x.data, self.z, (cout * kh * kw, 1, 1, cin), self.slabs, 1, 0, 1, math=self.math,
                                      x_absmax=xa)
            else:
                fn.conv2d_wgrad(self.x.data, self.z, (cout * kh * kw, 1, 1, cin), 1, 0, 1,
                                out=self.w.grad.view(cout * kh * kw, 1, 1, cin), math=self.math, x_absmax=xa)
        elif self.guard_fb:
            assert not planes, "a convolution on the guard's fallback got its gradient as piece planes"
            if self.slabs is not None:
                fn.conv2d_wgrad_slabs(self.x_raw.data, dy, self.w.shape, self.slabs, self.stride, self.pad, self.dil,
                                      in_affine=self.in_affine, math="bf16x3")
            else:
                fn.conv2d_wgrad(self.x_raw.data, dy, self.w.shape, self.stride, self.pad, self.dil, out=self.w.grad,
                                in_affine=self.in_affine, math="bf16x3")
            self._g.guard["calls"] += 1
            self._g.guard["calls_total"] += 1
        elif self.slabs is not None:
            xp = self._x_planes()
            fn.conv2d_wgrad_slabs(self.x_raw.data if xp is None else xp, dy, self.w.shape, self.slabs, self.stride, self.pad,
                                  self.dil, in_affine=self.in_affine if xp is None else None, math=self.math, x_absmax=xa,
                                  dy_absmax=dya, dy_planes=planes, x_planes=xp is not None)
        else:
            xp = self._x_planes()
            fn.conv2d_wgrad(self.x_raw.data if xp is None else xp, dy, self.w.shape, self.stride, self.pad, self.dil,
                            out=self.w.grad, in_affine=self.in_affine if xp is None else None, math=self.math, x_absmax=xa,
                            dy_absmax=dya, dy_planes=planes, x_planes=xp is not None)
        self.slabs_fresh = self.slabs is not None

    def _data_gradient(self, dy, planes, dya, wa):
        if self.wtp is not None and self._g.wp_table is None:
            fn.weight_planes(self.w.data, transposed=True, cols=self.wtp.shape[2] * 32, out=self.wtp, math=self.math,
                             w_absmax=wa)
        elif self.wtp is None and not self._g.wt_batched:
            fn.weight_transpose(self.w.data, out=self.wt, copy=self.wh)
        dx, acc = self.x.grad_target()
        bn = getattr(self, "bn_bwd_node", None)   # set by Graph.finalize on the LAST writer of a deferred BN's gradient
        bn_bwd, bn_dya = None, None
        if bn is not None:
            bn_bwd = (bn.x.data, bn.scale, bn.shift, bn.mean, bn.rstd, bn.relu, bn.bwd_sums[0])
            bn.bwd_sums_ready = True
            if bn.dx_planes:         # that BatchNorm's backward bounds its dx from the largest gradient stored here
                bn_dya = self._g.scalar(bn.am_dyin)
        if self.guard_fb:
            w3 = self._w3(True)
            fn.conv2d_dgrad(dy, self.wt if w3 is None else None, self.x.shape, self.stride, self.pad, self.dil, out=dx,
                            accumulate=acc, bn_bwd=bn_bwd, wt_planes=w3, math="bf16x3", wt_shape=self.wt_shape)
            if bn_dya is not None:       # (the two-piece data gradient leaves this by-product in its epilogue)
                fn.absmax(dx, out=bn_dya)
            self._g.guard["calls"] += 1
            self._g.guard["calls_total"] += 1
            return
        fn.conv2d_dgrad(dy, self.wt, self.x.shape, self.stride, self.pad, self.dil, out=dx, accumulate=acc,
                        bn_bwd=bn_bwd, wt_planes=self.wtp, math=self.math, dy_absmax=dya, w_absmax=wa,
                        bn_dy_absmax=bn_dya, dy_planes=planes, wt_shape=self.wt_shape)


class BilinearConcatConv(Node):
    """Convolution(kernel k x k, stride 1, 'same', no bias) of the channel concatenation of bilinearly resized maps
    -- `score3_conv` over `score3_concat` (multitask_symbol_builder.py:574-585) -- evaluated WITHOUT the
    concatenation.

    With the k x k taps expanded (dspn_tap_sum_f32) the convolution is a per-pixel linear map W (Cout*k*k x Cin)
    followed by a shifted sum, and a per-pixel linear map commutes with the resize U_c of each component (both are
    linear; U_c acts on pixels, W on channels):
        z = W . concat_c U_c(x_c) = sum_c U_c(W_c . x_c),          W_c = the columns of W that face component c.
    So each component is multiplied at ITS OWN resolution (4x4 ... 64x64 instead of 64x64 for all 3328 channels:
    9.3x fewer multiply-adds at 512x512), the small Cout*k*k-channel results are resized and added, and the
    3328-channel concatenation (1.7 GB per batch of 32, written once and read three times) never exists.  Backward
    is the transpose of the same chain: dz_c = U_c^T dz, dx_c = W_c^T dz_c, dW_c = dz_c^T x_c.
    Exact in real arithmetic; in fp32 the result differs from the direct form by rounding only (the graph-level
    parity tests against the direct-form oracle cover it).  The parameter keeps the reference's name and shape."""

    def __init__(self, g, inputs, name, num_filter, kernel, pad, target_hw, theta, init="maxdim"):
        N = inputs[0].shape[0]
        kh, kw = fn._hw(kernel)
        self.pad = fn._hw(pad)
        assert (kh - 1) // 2 == self.pad[0] and (kw - 1) // 2 == self.pad[1], "needs a 'same' convolution"
        Ht, Wt = target_hw
        self.inputs = inputs
        self.theta = theta            # Param "affine_matrix": the sampling grid of every component
        self.offsets = np.cumsum([0] + [t.shape[3] for t in inputs]).tolist()
        Cin = self.offsets[-1]
        self.cout, self.kh, self.kw = num_filter, kh, kw
        self.w = g.param(name + "_weight", (num_filter, kh, kw, Cin), conv_weight_init(init, Cin))
        self.w.logical, self.w.kind = (num_filter, Cin, kh, kw), "conv"
        T = num_filter * kh * kw
        Tp = fn.padc(T)
        self.T = T
        half = fn.ACT_DTYPE == torch.bfloat16
        self.z = fn.act_zeros(N, Ht, Wt, Tp, device=g.device)        # tap-expanded map at the target size (dz in backward)
        self.zc, self.wc, self.wct, self.dwc, self.wch, self.wcp = [], [], [], [], [], []
        self.math = g.math
        self._g = g
        # "f16x2" math: per component, slots for the magnitudes of its input, its weight slice and its output gradient --
        # each taken once per step and shared by the forward, weight-gradient and data-gradient calls that multiply it
        f16 = g.math == "f16x2"
        self.am = [(g.new_scalar(), g.new_scalar(), g.new_scalar(backward=True)) if f16 else (None, None, None) for _ in inputs]
        # round 4: every weight slice is bounded by the magnitude of the whole weight, which the one batched launch at the top
        # of forward() takes with all the others (Graph.am_table); the inputs are materialised BatchNorm outputs, which leave
        # theirs while they are written; the sampler's data-gradient passes leave the gradients'
        self.am_w = g.new_scalar() if f16 else None
        for t in inputs:
            # W_c . x_c at the component's own resolution (its gradient in backward).  Every component goes through
            # the sampler, also the ones that already have the target size: once the optimizer has moved
            # affine_matrix off the identity they are resampled too, and their grid gradient is not zero
            self.zc.append(fn.act_zeros(N, t.shape[1], t.shape[2], Tp, device=g.device))
            self.wc.append(fn.zeros(T, 1, 1, t.shape[3], device=g.device))      # W_c, contiguous (float master slice)
            self.wct.append(fn.act_zeros(t.shape[3], 1, 1, Tp, device=g.device))    # its transpose (data-gradient operand)
            self.wch.append(fn.act_zeros(T, 1, 1, t.shape[3], device=g.device) if half else None)   # bf16 forward operand
            self.dwc.append(fn.zeros(T, 1, 1, t.shape[3], device=g.device))
            # split math: piece planes of W_c (the forward operand), refreshed after every gather of the slices
            self.wcp.append(fn.zeros(T, 1, t.shape[3] // 32, fn.plane_pieces(g.math), 32, device=g.device, dtype=torch.bfloat16)
                            if g.device.type == "cuda" and fn.needs_planes(self.zc[-1].dtype, t.shape[3], g.math) else None)
        self.sources = None           # fn.SamplerSources over zc, made at the first forward
        self.tpart = None             # (source pixels of all components, 6) float64: rows of the theta gradient
        self.out = g.tensor((N, Ht, Wt, fn.padc(num_filter)), name + "_out")
        self.out.channels = num_filter
        # multiply-adds actually executed (the direct form would be 2 * Cin * Cout * k * k * Ht * Wt * N)
        self.flops_fwd = sum(2.0 * t.shape[3] * T * t.shape[1] * t.shape[2] * N for t in inputs)
        self.flops_bwd = self.flops_fwd + sum(2.0 * t.shape[3] * T * t.shape[1] * t.shape[2] * N
                                              for t in inputs if t.requires_grad)
        self.flops_direct = 2.0 * Cin * T * Ht * Wt * N

    def _gather_weights(self):
        w2d = self.w.data
        Cin = self.offsets[-1]
        for c, t in enumerate(self.inputs):
            fn.copy_block(w2d, self.wc[c], 1, self.T, t.shape[3], 0, Cin, self.offsets[c], 0, t.shape[3], 0)

    def forward(self):
        self._gather_weights()
        for c, t in enumerate(self.inputs):
            if self.wch[c] is not None:      # bf16 operands of this slice: copy + transpose in one launch
                fn.weight_transpose(self.wc[c], out=self.wct[c], copy=self.wch[c])
            xa, wa = self._g.magnitude(self.am[c][0], t.data, src=t), self._g.scalar(self.am_w)
            if self.wcp[c] is not None:
                fn.weight_planes(self.wc[c], out=self.wcp[c], math=self.math, w_absmax=wa)
            fn.conv2d_forward(t.data, self.wc[c] if self.wch[c] is None else self.wch[c], None, 1, 0, 1, out=self.zc[c],
                              w_planes=self.wcp[c], math=self.math, x_absmax=xa, w_absmax=wa)
        if self.sources is None:
            self.sources = fn.SamplerSources([(z, 0) for z in self.zc])
        fn.affine_sampler_forward(self.sources, self.theta.data, self.z)       # z = sum_c U_c(theta)(W_c x_c), one pass
        fn.tap_sum(self.z, None, self.cout, self.kh, self.kw, self.pad, out=self.out.data)

    def backward(self):
        if not self.out._gw:
            return
        fn.tap_spread(self.out.grad, self.cout, self.kh, self.kw, self.pad, out=self.z)
        # d L / d affine_matrix needs the forward values W_c x_c: every source's data-gradient pass reads them (its own
        # pixel's, before it overwrites zc with the gradient) and leaves its share of the theta gradient as rows of tpart
        if self.tpart is None:
            rows = [fn.affine_sampler_theta_rows(z.shape, self.z.shape[1], z.dtype) for z in self.zc]
            self.tpart = torch.zeros(sum(rows), 6, dtype=torch.float64, device=self.z.device)
            self.tpart_off = np.cumsum([0] + rows).tolist()
        Cin = self.offsets[-1]
        for c, t in enumerate(self.inputs):
            dza = self._g.scalar(self.am[c][2])
            if dza is not None and self.zc[c].dtype == torch.float32:
                self._g._am_done.add(self.am[c][2])
            dz = fn.affine_sampler_backward_data_theta(self.z, self.theta.data, self.zc[c], 0,
                                                       self.tpart[self.tpart_off[c]:self.tpart_off[c + 1]], dx=self.zc[c],
                                                       dx_absmax=dza)
            xa, wa = self._g.magnitude(self.am[c][0], t.data, src=t), self._g.scalar(self.am_w)
            dza = self._g.magnitude(self.am[c][2], dz)
            fn.conv2d_wgrad(t.data, dz, (self.T, 1, 1, t.shape[3]), 1, 0, 1, out=self.dwc[c], math=self.math, x_absmax=xa,
                            dy_absmax=dza)
            fn.copy_block(self.dwc[c], self.w.grad, 1, self.T, t.shape[3], 0, t.shape[3], 0, 0, Cin, self.offsets[c])
            if t.requires_grad:
                if self.wch[c] is None:
                    fn.weight_transpose(self.wc[c], out=self.wct[c])
                dx, acc = t.grad_target()
                fn.conv2d_dgrad(dz, self.wct[c], t.shape, 1, 0, 1, out=dx, accumulate=acc, math=self.math, dy_absmax=dza,
                                w_absmax=wa)
        fn.affine_sampler_theta_reduce(self.tpart, self.theta.grad)


class Deconv4x4s2(Node):
    """mx.sym.Deconvolution(kernel 4, stride 2, pad 1, no bias) (multitask_symbol_builder.py:586):
    forward = data-gradient kernel of the 4x4/2 convolution with weight [K=in][R][S][C=out]."""

    def __init__(self, g, x, name, channels):
        N, H, W, Cp = x.shape
        self.x = x
        self.w = g.param(name + "_weight", (Cp, 4, 4, Cp), deconv_bilinear_init(channels))
        self.w.logical, self.w.kind = (channels, channels, 4, 4), "deconv"
        self.wt = fn.act_zeros(Cp, 4, 4, Cp, device=g.device)
        self.wh = fn.act_zeros(Cp, 4, 4, Cp, device=g.device) if fn.ACT_DTYPE == torch.bfloat16 else None
        self.out = g.tensor((N, 2 * H, 2 * W, Cp), name + "_out")
        self.out.channels = channels
        self.math = g.math
        self._g = g
        self.am = (g.new_scalar(), g.new_scalar(), g.new_scalar(backward=True)) if g.math == "f16x2" else (None, None, None)   # x, w, dy
        self.am_w = self.am[1]      # (taken with every other weight's by the batched launch at the top of forward())
        self.flops_fwd = 2.0 * channels * channels * 16 * H * W * N
        self.flops_bwd = self.flops_fwd * (2 if x.requires_grad else 1)

    def forward(self):
        fn.weight_transpose(self.w.data, out=self.wt, copy=self.wh)
        g = self._g
        xa, wa = g.magnitude(self.am[0], self.x.data, src=self.x), g.magnitude(self.am[1], self.w.data)
        fn.conv2d_dgrad(self.x.data, self.wt, self.out.shape, 2, 1, 1, out=self.out.data, math=self.math, dy_absmax=xa,
                        w_absmax=wa)

    def backward(self):
        if not self.out._gw:
            return
        dy = self.out.grad
        g = self._g
        xa, wa, dya = g.magnitude(self.am[0], self.x.data, src=self.x), g.magnitude(self.am[1], self.w.data), g.magnitude(self.am[2], dy)
        fn.conv2d_wgrad(dy, self.x.data, self.w.shape, 2, 1, 1, out=self.w.grad, math=self.math, x_absmax=dya, dy_absmax=xa)
        if self.x.requires_grad:
            dx, acc = self.x.grad_target()
            fn.conv2d_forward(dy, self.w.data if self.wh is None else self.wh, None, 2, 1, 1, out=dx, accumulate=acc,
                              math=self.math, x_absmax=dya, w_absmax=wa)


class Add(Node):
    """elementwise sum of the residual unit (symbol/resnet.py:51)"""

    def __init__(self, g, a, b, name):
        self.a, self.b = a, b
        self.out = g.tensor(a.shape, name)

    def forward(self):
        fn.add(self.a.data, self.b.data, out=self.out.data)

    def backward(self):
        if not self.out._gw:
            return
        for t in (self.a, self.b):
            if t.requires_grad:
                t.give_grad(self.out.grad)


class BlockGrad(Node):
    """mx.sym.BlockGrad: same values, no gradient (multitask_symbol_builder.py:542,549)"""

    def __init__(self, g, x, name):
        self.out = g.tensor(x.shape, name, requires_grad=False, data=x.data)
        self.out.alias_of, self.out.am_slot, self.out.channels = x, x.am_slot, x.channels

    def forward(self):
        pass


class MaxPool(Node):
    """mx.sym.Pooling(pool_type='max'); pooling_convention='full' (symbol/vgg16_reduced.py:40-42) rounds the
    output size up, windows that stick out of the image just see fewer pixels"""

    def __init__(self, g, x, name, kernel, stride, pad, full=False):
        N, H, W, C = x.shape
        self.x, self.k, self.s, self.p = x, kernel, stride, pad

        def osz(h):
            if not full:
                return fn.conv_out_size(h, kernel, stride, pad)
            return -(-(h + 2 * pad - kernel) // stride) + 1
        self.out = g.tensor((N, osz(H), osz(W), C), name)
        self.out.am_slot = x.am_slot        # |max over a window| <= the input's magnitude
        # input produced by a deferred BatchNorm(+ReLU) (round 4: bn0 -> relu0 -> pooling0 of the resnet stem): the pooling pass
        # reads the raw tensor and applies the affine itself, the normalised tensor is never written
        self.x_raw, self.in_affine, self._g = x, None, g
        self.am_out = None
        if x.affine_src is not None:
            raw, sc, sh, arelu = x.affine_src
            self.x_raw, self.in_affine = raw, (sc, sh, arelu)
            if g.math == "f16x2" and self.out.dtype == torch.float32:
                self.am_out = g.new_scalar()
                self.out.am_slot = self.am_out
        # one byte per output element: which window position held the maximum (read by backward
        # instead of x and y)
        self.argmax = (torch.zeros(self.out.shape, dtype=torch.uint8, device=g.device)
                       if x.requires_grad and kernel * kernel < 255 else None)

    def forward(self):
        if self.in_affine is not None:
            am = self._g.scalar(self.am_out)
            fn.maxpool_forward(self.x_raw.data, self.k, self.s, self.p, out=self.out.data, argmax=self.argmax,
                               in_affine=self.in_affine, out_absmax=am)
            if am is not None:
                self._g._am_done.add(self.am_out)
            return
        fn.maxpool_forward(self.x.data, self.k, self.s, self.p, out=self.out.data, argmax=self.argmax)

    def backward(self):
        if not self.out._gw or not self.x.requires_grad:
            return
        assert not self.x._gw, "maxpool input has a single consumer"
        bn = getattr(self.x, "bn_node", None)
        if (self.in_affine is not None and self.argmax is not None and bn is not None and FUSE_MAXPOOL_BACKWARD
                and self.out.grad.dtype == torch.float32 and self.out.grad.device.type == "cuda"):
            # round 4: the BatchNorm(+ReLU) folded into this pooling forms its output gradient from (pooled gradient, argmax)
            # itself (dspn_bn_backward_maxpool_f32): the dense gradient of the pooling input is never written
            bn.pool_grad = (self.argmax, self.out.grad, self.k, self.s, self.p)
            self.x._gw = True
            return
        dx, _ = self.x.grad_target()
        if self.argmax is not None:
            fn.maxpool_backward_argmax(self.argmax, self.out.grad, self.x.shape, self.k, self.s, self.p, dx=dx)
        else:
            assert self.in_affine is None, "a pooling layer behind a deferred BatchNorm keeps its argmax record"
            fn.maxpool_backward(self.x.data, self.out.data, self.out.grad, self.k, self.s, self.p, dx=dx)


class AvgPool(Node):
    """mx.sym.Pooling(pool_type='avg', kernel=(k,k), stride=(k,k)) (multitask_symbol_builder.py:560-562)"""

    def __init__(self, g, x, name, k):
        N, H, W, C = x.shape
        self.x, self.k = x, k
        if k == 1:
            self.out = x
        else:
            self.out = g.tensor((N, H // k, W // k, C), name)
            self.out.am_slot = x.am_slot    # |mean over a window| <= the input's magnitude

    def forward(self):
        if self.k > 1:
            fn.avgpool_forward(self.x.data, self.k, out=self.out.data)

    def backward(self):
        if self.k == 1 or not self.out._gw or not self.x.requires_grad:
            return
        dx, acc = self.x.grad_target()
        fn.avgpool_backward(self.out.grad, self.x.shape, self.k, dx=dx, accumulate=acc)


class AvgPool2d(Node):
    """mx.sym.Pooling(pool_type='avg', kernel k, stride, pad) with overlapping windows
    (symbol/inceptionv3.py:31,74,115); the divisor is k*k including padding"""

    def __init__(self, g, x, name, kernel, stride, pad):
        N, H, W, C = x.shape
        self.x, self.k, self.s, self.p = x, kernel, stride, pad
        self.out = g.tensor((N, fn.conv_out_size(H, kernel, stride, pad), fn.conv_out_size(W, kernel, stride, pad), C),
                            name)

    def forward(self):
        fn.avgpool2d_forward(self.x.data, self.k, self.s, self.p, out=self.out.data)

    def backward(self):
        if not self.out._gw or not self.x.requires_grad:
            return
        dx, acc = self.x.grad_target()
        fn.avgpool2d_backward(self.out.grad, self.x.shape, self.k, self.s, self.p, dx=dx, accumulate=acc)


class Concat(Node):
    """mx.sym.Concat along channels (symbol/inceptionv3.py:33): each input is copied into its channel
    slice of the NHWC output; backward copies (or accumulates) the slice of the output gradient back"""

    def __init__(self, g, inputs, name):
        N, H, W, _ = inputs[0].shape
        assert all(t.shape[:3] == (N, H, W) for t in inputs), [t.shape for t in inputs]
        self.inputs = inputs
        self.offsets = np.cumsum([0] + [t.shape[3] for t in inputs]).tolist()
        self.out = g.tensor((N, H, W, self.offsets[-1]), name)

    def forward(self):
        N, H, W, C = self.out.shape
        for t, off in zip(self.inputs, self.offsets):
            c = t.shape[3]
            fn.copy_block(t.data, self.out.data, N, H * W, c, H * W * c, c, 0, H * W * C, C, off)

    def backward(self):
        if not self.out._gw:
            return
        N, H, W, C = self.out.shape
        for t, off in zip(self.inputs, self.offsets):
            if not t.requires_grad:
                continue
            c = t.shape[3]
            dx, acc = t.grad_target()
            fn.copy_block(self.out.grad, dx, N, H * W, c, H * W * C, C, off, H * W * c, c, 0, accumulate=acc)


class BilinearConcat(Node):
    """GridGenerator(affine_matrix, target) + BilinearSampler on each input, concatenated along
    channels (multitask_symbol_builder.py:574-582)."""

    def __init__(self, g, inputs, name, target_hw, theta):
        N = inputs[0].shape[0]
        self.inputs = inputs
        self.theta = theta
        self.offsets = np.cumsum([0] + [t.shape[3] for t in inputs]).tolist()
        self.out = g.tensor((N, target_hw[0], target_hw[1], self.offsets[-1]), name)
        self.sources = None           # made at the first forward (Graph.finalize may still re-allocate an input)

    def forward(self):
        if self.sources is None:
            self.sources = fn.SamplerSources([(t.data, off) for t, off in zip(self.inputs, self.offsets)])
        fn.affine_sampler_forward(self.sources, self.theta.data, self.out.data)

    def backward(self):
        if not self.out._gw:
            return
        fn.affine_sampler_backward_theta(self.sources, self.theta.data, self.out.grad, self.theta.grad)
        for t, off in zip(self.inputs, self.offsets):
            if not t.requires_grad:
                continue
            dx, acc = t.grad_target()
            fn.affine_sampler_backward_data(self.out.grad, self.theta.data, t.shape, off, dx=dx, accumulate=acc)
