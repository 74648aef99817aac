"""One training step = forward, backward, gradient all-reduce, SGD-momentum update.

Numerics contract of MultiTaskSolver.fit (multi_solver.py:221, :284-293): `sgd` with
rescale_grad = 1/batch_size, momentum, wd applied to every parameter (the reference creates the
optimizer without the symbol, so no lr_mult / wd_mult takes effect), batch-statistics BatchNorm
(is_train=True always).  What the reference does around it per batch -- re-binding the executor,
copying all five outputs to the host -- is not reproduced.

Data parallelism (new in this build; the reference is single-device): one process per GPU, each with
a full replica and its own batch shard; gradients are summed with RCCL all-reduce in contiguous
buckets of the flat gradient arena, launched as soon as backward has produced every gradient of a
bucket so the reduction overlaps the remaining dgrad/wgrad kernels, then scaled by 1/world_size
inside the SGD kernel (`rescale_grad = 1/len(ctx)` convention of train/train_multitask.py:248).
"""
from .. import functional as fn


def plan_buckets(params, owner_index, total, bucket_elems):
    """Cut the flat gradient arena into contiguous buckets.

    params: [(name, offset, padded_size)] in arena order; owner_index[name] = index of the graph node
    that produces the gradient.  Returns [(lo, hi, first_node)] sorted by the order in which backward
    (which runs nodes from last to first) completes them: a bucket is complete once the node with the
    smallest index among its owners has run."""
    buckets, lo, first = [], 0, None
    for name, offset, size in params:
        idx = owner_index[name]
        first = idx if first is None else min(first, idx)
        end = offset + size
        if end - lo >= bucket_elems:
            buckets.append((lo, end, first))
            lo, first = end, None
    if lo < total:
        buckets.append((lo, total, first if first is not None else 0))
    buckets.sort(key=lambda b: -b[2])
    return buckets


def side_buckets(buckets, params, owner_index, side_nodes):
    """per bucket (release order): does it hold a parameter whose gradient a node of `side_nodes` writes?"""
    side_nodes = set(side_nodes)
    return [any(o < hi and o + s > lo and owner_index[n] in side_nodes for n, o, s in params) for lo, hi, _ in buckets]


class GradBucketReducer:
    """Sum-all-reduce of the gradient arena in buckets, each launched (async) as soon as backward has
    passed the first node that writes into it, so RCCL traffic overlaps the remaining kernels.
    Device-agnostic (used with RCCL on GPUs, exercised with gloo on CPU tensors in the tests)."""

    def __init__(self, grad_arena, buckets, process_group=None):
        self.arena, self.buckets, self.pg = grad_arena, buckets, process_group
        self.pending, self.works, self.launched = [], [], []
        # measure_exposed = True: bracket the waits of finish() with events on the compute stream; the time between them
        # is what the step waits for the collectives AFTER backward has run out of kernels to overlap them with
        # (bench.py reports the mean as allreduce_exposed_ms)
        self.measure_exposed, self.exposed_events = False, []
        # ... and per bucket, an event when it is handed to the collective and one when the compute stream has seen it
        # complete: bucket_latency_ms() = issue -> completion as the step sees it (at world 1 that is the rest of backward;
        # on a real node the first SCALE run can be read bucket by bucket)
        self.bucket_events = []

    def begin(self):
        self.pending = list(self.buckets)
        self.works, self.launched = [], []
        self._issue_ev = []

    def node_done(self, idx, before_release=None):
        """release every bucket whose last gradient the backward of node idx has produced.  before_release(bucket index in
        release order), if given, runs first: the solver sums the bucket's split-K slabs there and, when part of the bucket
        was written on the side stream, orders the current stream behind it -- the collective starts behind the CURRENT
        stream's work, whichever stream wrote the bucket's last gradient"""
        import torch.distributed as dist
        while self.pending and self.pending[0][2] >= idx:
            if before_release is not None:
                before_release(len(self.buckets) - len(self.pending))
            lo, hi, _ = self.pending.pop(0)
            self.launched.append((lo, hi))
            if self.measure_exposed and self.arena.is_cuda:
                import torch
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                self._issue_ev.append(ev)
            self.works.append(dist.all_reduce(self.arena[lo:hi], group=self.pg, async_op=True))

    def finish(self):
        assert not self.pending, "backward ended before every bucket was released"
        timed = self.measure_exposed and self.arena.is_cuda
        if timed:
            import torch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        done_ev = []
        for w in self.works:
            w.wait()
            if timed and len(done_ev) < len(self._issue_ev):
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                done_ev.append(ev)
        if timed:
            e1.record()
            self.exposed_events.append((e0, e1))
            if len(done_ev) == len(self._issue_ev):
                self.bucket_events.append(list(zip(self._issue_ev, done_ev)))

    def exposed_ms(self):
        """mean time per step the compute stream spent blocked in finish() (call after a device synchronize)"""
        if not self.exposed_events:
            return None
        t = sum(a.elapsed_time(b) for a, b in self.exposed_events) / len(self.exposed_events)
        self.exposed_events = []
        return t

    def bucket_latency_ms(self):
        """per bucket (release order), mean over the measured steps: milliseconds from the all-reduce's issue to the point
        where the compute stream has waited for it (call after a device synchronize)"""
        if not self.bucket_events:
            return None
        n = len(self.bucket_events[0])
        out = [sum(step[b][0].elapsed_time(step[b][1]) for step in self.bucket_events) / len(self.bucket_events) for b in range(n)]
        self.bucket_events = []
        return out


class MultiTaskSolver:
    def __init__(self, net, learning_rate=0.0005, momentum=0.9, wd=0.0005, process_group=None,
                 world_size=1, bucket_mb=16.0, force_reducer=False, high_priority=True):
        self.net, self.g = net, net.g
        # the step runs on a HIGH-priority stream of its own: MultiBoxDetection's side stream (normal priority) then never
        # gets its workgroups handed out ahead of a main-path kernel that is ready (HIP has two levels, and torch's
        # current stream already sits on the lower one).  step() orders that stream behind the caller's current stream
        # on entry and the caller's stream behind it on exit, so callers see ordinary stream semantics.
        self.stream = None
        if high_priority and net.g.device.type == "cuda":
            import torch
            from ..engine import shared_stream
            self.stream = shared_stream(net.g.device, "step", -1)
        self.lr, self.momentum, self.wd = learning_rate, momentum, wd
        self.world_size, self.pg = world_size, process_group
        self.batch_size = net.data.shape[0]
        self._graph, self._graph_hyper = None, None
        self._ran_eager = False                         # has forward / backward of this solver run outside a recording?
        self._rerecord = False
        self._replays, self.graph_rerecorded = 0, 0     # replays of the recorded step / recordings dropped by the range guard
        g = self.g
        owner = {}
        for idx, n in enumerate(g.nodes):
            for v in vars(n).values():
                if hasattr(v, "offset") and hasattr(v, "wd_mult"):
                    owner.setdefault(v.name, idx)
                    owner[v.name] = min(owner[v.name], idx)
        params = [(p.name, p.offset, (p.size + 3) // 4 * 4) for p in g.param_order]
        self.buckets = plan_buckets(params, owner, g.arena.numel(), int(bucket_mb * (1 << 20) / 4))
        self.reducer = (GradBucketReducer(g.grad_arena, self.buckets, process_group)
                        if (world_size > 1 or force_reducer) else None)
        # round 6: with an all-reduce to feed, the weight gradients stay on the step's stream (engine.WGRAD_SIDE): a bucket's
        # release would make that stream wait for theirs eight times per pass, and no N > 1 run has measured the mix
        g.wgrad_side_allowed = self.reducer is None
        # convolutions whose weight gradient lies in each bucket: their split-K slabs are summed in one launch
        # right before the bucket is released to the all-reduce
        self.bucket_convs = []
        for lo, hi, first in self.buckets:
            self.bucket_convs.append((first, [n for n in g.nodes if getattr(n, "slabs", None) is not None
                                              and lo <= n.w.offset < hi]))
        # buckets that hold a gradient written by the side-stream part of backward (Graph.set_side_backward): their release
        # waits for the side stream's progress event; the others are released as before, beside it
        self.bucket_side = side_buckets(self.buckets, params, owner,
                                        g.side_bwd["side"] if g.side_bwd is not None else ())

    def set_batch(self, data, label_det, label_seg):
        """device tensors in the reference's layouts: (B,3,H,W), (B,200,6), (B,H/4,W/4); a label the graph has
        no input for (detection-only / segmentation-only graphs) is ignored"""
        self.net.data.data.copy_(data)
        if self.net.label_det is not None:
            self.net.label_det.data.copy_(label_det)
        if self.net.label_seg is not None:
            self.net.label_seg.data.copy_(label_seg)

    def forward(self):
        self.g.forward()

    def backward(self):
        g = self.g
        g.begin_backward()
        if self.reducer is not None:
            self.reducer.begin()
        if self.reducer is None:
            pending = list(range(len(self.bucket_convs)))      # buckets in release order
            for idx in range(len(g.nodes) - 1, -1, -1):
                g.backward_node(idx)
                while pending and self.bucket_convs[pending[0]][0] >= idx and not g.side_backward_busy(idx):
                    b = pending.pop(0)
                    g.flush_slabs(("bucket", b), self.bucket_convs[b][1], beside=True)
            g.join_side_backward()
            for b in pending:
                g.flush_slabs(("bucket", b), self.bucket_convs[b][1])
            return
        # N > 1 (round 5): the SAME schedule as N = 1 -- the side-stream part of backward stays on its stream.  A bucket is
        # released when backward has passed its first owner, as before; one that holds side-stream gradients first makes the
        # main stream wait for the side stream's progress event (every side node of the bucket has been issued by then)

        def before_release(b):
            if self.bucket_side[b]:
                g.join_side_backward()
            g.flush_slabs(("bucket", b), self.bucket_convs[b][1])

        for idx in range(len(g.nodes) - 1, -1, -1):
            g.backward_node(idx)
            self.reducer.node_done(idx, before_release)
        g.join_side_backward()
        self.reducer.finish()

    def update(self):
        g = self.g
        fn.sgd_momentum(g.arena, g.grad_arena, g.mom_arena, self.lr, self.momentum, self.wd,
                        1.0 / (self.batch_size * self.world_size))

    def _on_step_stream(self, body):
        """run body() on the solver's stream with ordinary stream semantics for the caller (behind the caller's current
        stream on entry, the caller's stream behind it on exit)"""
        if self.stream is None:
            return body()
        import torch
        cur = torch.cuda.current_stream(self.g.device)
        if cur == self.stream:
            return body()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            body()
        cur.wait_stream(self.stream)

    def step(self):
        self._on_step_stream(self._step)

    def _calibrate_guard(self):
        """range guard of the "f16x2" math (engine.Graph._update_guard): the decisions of a pass come from the spans the previous
        pass measured, so the very first step is preceded by ONE forward / backward whose only product is those spans (no
        update, no reduction: parameters, momenta and the batch are untouched; a net loaded from a checkpoint whose channels
        span more than 2^16 is then guarded from its first update on)"""
        g = self.g
        if g.scalars is None or not g.guard["enabled"] or g.guard["have_stats"]:
            return
        g.forward()
        g.backward()
        g.guard["decide_now"] = True      # the next forward() decides from this pass (one synchronisation, once)

    def _step(self):
        self._calibrate_guard()
        if self._graph is not None:
            # the recorded SGD launch carries lr / momentum / wd BY VALUE: a schedule that moved them since the recording
            # (the reference's optimizer takes an lr_scheduler, multi_solver.py:221) drops the graph and records a new one
            if (self.lr, self.momentum, self.wd) != self._graph_hyper:
                self._graph = None
                if not self.capture(warmup=0):
                    self.forward(); self.backward(); self.update()
                    return
            self._graph.replay()
            # the range guard cannot act from inside a recorded step: every GUARD_PERIOD-th replay its spans are looked at from
            # here, and a changed decision drops the recording (advisor r5)
            self._replays += 1
            if self._replays % self.g.GUARD_PERIOD == 0 and self.g.guard_poll():
                # the convolutions that changed sides run kernels this process may never have launched: the next step runs
                # eagerly (first-use work outside a recording), the recording is made behind it
                self._graph = None
                self.graph_rerecorded += 1
                self._rerecord = True
            return
        self.forward()
        self.backward()
        self.update()
        self._ran_eager = True
        if self._rerecord:
            self._rerecord = False
            self.capture(warmup=0)

    def capture(self, warmup=2):
        """Record one whole step (forward, backward, update: ~700 .. 1100 kernel launches issued from Python through
        ctypes) into a HIP graph; step() then replays it with one host call.  Every launch of the library is
        graph-capturable (no allocation, no synchronisation, explicit stream), the batch lives in fixed buffers
        (set_batch copies into them) and the host-side bookkeeping of a step is the same every step, so the recorded
        launch sequence IS the step.  The gradient all-reduce is not captured: with a reducer the step stays eager.
        Returns True if the graph is in use."""
        import torch
        if self.reducer is not None or self.g.device.type != "cuda" or self._graph is not None:
            return self._graph is not None
        # the recording keeps the range guard's decisions it is made with: they come from a calibration pass (a graph
        # recorded before any step would otherwise never be calibrated) and are settled BEFORE the recording starts
        def settle():
            self._calibrate_guard()
            for _ in range(warmup):               # first-use work (function attributes, workspace growth) happens eagerly
                self._step()
            if not self._ran_eager:
                # ... and so does the first use of THIS class's backward (the per-bucket tables of the slab sums are built and
                # uploaded on first use: inside a recording that is a host copy of a temporary): one pass without an update
                self.forward(); self.backward()
                self._ran_eager = True
            if self.g.guard.pop("decide_now", False):
                self.g.guard_poll(blocking=True)
        self._on_step_stream(settle)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        det = getattr(self.net, "det", None)
        try:
            # (captured on the solver's own high-priority stream when it has one: a replayed step then runs where an eager
            # step runs, not on torch's normal-priority capture stream)
            with torch.cuda.graph(graph, **({"stream": self.stream} if self.stream is not None else {})):
                self.forward()
                self.backward()
                self.update()
                if det is not None:
                    det.join()                    # the side stream of MultiBoxDetection joins before the capture ends
        except Exception as e:                    # noqa: BLE001 -- whatever broke the recording, the step stays eager
            import logging
            logging.getLogger(__name__).warning("MultiTaskSolver.capture: recording failed, the step stays eager: %r", e)
            torch.cuda.synchronize()
            if det is not None:
                # an event recorded inside the aborted capture must not be waited on by the next eager join()
                det.pending = False
            return False
        self._graph = graph
        self._graph_hyper = (self.lr, self.momentum, self.wd)
        return True


class BatchEndParam(object):
    """what the reference hands to batch_end_callback (mx.model.BatchEndParam: epoch, nbatch, eval_metric)"""

    def __init__(self, epoch, nbatch, eval_metric):
        self.epoch, self.nbatch, self.eval_metric = epoch, nbatch, eval_metric


def do_checkpoint(prefix):
    """mx.callback.do_checkpoint(prefix) (multi_train.py:370): epoch_end_callback that writes prefix-%04d.params
    for epoch + 1"""
    from ..model import save_checkpoint

    def _callback(epoch, net, aux_params=None):
        save_checkpoint(prefix, epoch + 1, net, aux_params=aux_params)
    return _callback


def fit(solver, train_data, begin_epoch=0, num_epoch=1, batch_end_callback=None, epoch_end_callback=None,
        eval_data=None, class_names=None, seg_class_names=None, logger=None, eval_score_thresh=0.25,
        check_label_errors=True):
    """The epoch loop of MultiTaskSolver.fit (multi_solver.py:229-345 + the evaluation pass :353-436): per epoch
    train_data.reset(), metric reset, one solver.step() per batch with the MultiBoxMetric / CustomAccuracyMetric
    read-outs, batch_end_callback(BatchEndParam), epoch_end_callback(epoch, net), the 'Train-<name>' log lines, and
    -- when eval_data is given -- evaluate.multi_eval.evaluate_net over it.
    train_data / eval_data: dspnet_amd.dataset.iterator.MultiTaskRecordIter (or any iterator of (batch, fnames) with
    batch.data[0], batch.label[0], batch.label[1] device tensors).  -> list of per-epoch dicts of metric values.
    eval_score_thresh: the in-training evaluation keeps detections with score > .25 (multi_solver.py:428; multi_eval.py
    uses .1).  check_label_errors: after every step (at the metric read-out, which synchronises anyway) raise DspnError
    where the reference's MultiBoxTarget would have CHECK-failed on the batch (multibox_target.cc:98-101, :236)."""
    import logging
    from .metric import CustomAccuracyMetric, MultiBoxMetric
    logger = logger or logging
    net = solver.net
    multibox_metric = MultiBoxMetric()
    # a detection-only graph (get_det_symbol_train) has no segmentation output: no pixel-accuracy metric, as in the
    # reference's det_solver.py; a segmentation-only graph keeps MultiBoxMetric's SegCrossEntropy slot only
    has_seg = getattr(net, "seg_out", None) is not None
    acc_metric = CustomAccuracyMetric(num_classes=net.seg_out.C) if has_seg else None
    history = []
    for epoch in range(begin_epoch, num_epoch):
        nbatch = 0
        train_data.reset()
        multibox_metric.reset()
        if acc_metric is not None:
            acc_metric.reset()
        while train_data.iter_next():
            batch, _ = train_data.next()
            nbatch += 1
            solver.set_batch(batch.data[0], batch.label[0], batch.label[1])
            solver.step()
            multibox_metric.update(net)
            if acc_metric is not None:
                acc_metric.update([net.label_seg.data], [net.seg_out.prob.data])
            if check_label_errors and getattr(net, "target", None) is not None:
                net.target.raise_on_errors()
            if batch_end_callback is not None:
                batch_end_callback(BatchEndParam(epoch, nbatch, (multibox_metric, acc_metric) if acc_metric is not None
                                                 else (multibox_metric,)))
        if epoch_end_callback is not None:
            epoch_end_callback(epoch, net)
        names, values = multibox_metric.get()
        out = dict(zip(names, values))
        if acc_metric is not None:
            name, value = acc_metric.get()
            out[name] = value
        for k, v in out.items():
            logger.info("                     --->Epoch[%d] Train-%s=%f", epoch, k, v)
        if eval_data is not None:
            from ..evaluate.multi_eval import evaluate_net

            def batches():
                eval_data.reset()
                while eval_data.iter_next():
                    b, _ = eval_data.next()
                    yield {"data": b.data[0], "label_det": b.label[0], "label_seg": b.label[1]}
            ev = evaluate_net(net, batches(), class_names, seg_class_names, score_thresh=eval_score_thresh)
            for k, v in ev.items():
                if not isinstance(v, list):
                    logger.info("                     --->Epoch[%d] Validation-%s=%f", epoch, k, v)
            out["validation"] = ev
        out["nbatch"] = nbatch
        history.append(out)
    return history
