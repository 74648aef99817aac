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
import torch

from .. import functional as fn


class MultiTaskSolver:
    def __init__(self, net, learning_rate=0.0005, momentum=0.9, wd=0.0005, process_group=None,
                 world_size=1, bucket_mb=16.0):
        self.net, self.g = net, net.g
        self.lr, self.momentum, self.wd = learning_rate, momentum, wd
        self.world_size, self.pg = world_size, process_group
        self.batch_size = net.data.shape[0]
        self._plan_buckets(bucket_mb)

    def _plan_buckets(self, bucket_mb):
        """contiguous arena slices, each ready once backward has passed its first node"""
        g = self.g
        owner = {}
        for idx, n in enumerate(g.nodes):
            for v in vars(n).values():
                if hasattr(v, "offset") and hasattr(v, "wd_mult"):
                    owner[v.name] = idx
        self.buckets = []   # (lo, hi, first_node_index)
        limit = int(bucket_mb * (1 << 20) / 4)
        lo, first = 0, None
        total = g.arena.numel()
        for p in g.param_order:
            if first is None:
                first = owner[p.name]
            end = p.offset + (p.size + 3) // 4 * 4
            if end - lo >= limit:
                self.buckets.append((lo, end, first))
                lo, first = end, None
        if lo < total:
            self.buckets.append((lo, total, first if first is not None else 0))
        self.buckets.sort(key=lambda b: -b[2])     # order in which backward completes them

    def set_batch(self, data, label_det, label_seg):
        """device tensors in the reference's layouts: (B,3,H,W), (B,200,6), (B,H/4,W/4)"""
        self.net.data.data.copy_(data)
        self.net.label_det.data.copy_(label_det)
        self.net.label_seg.data.copy_(label_seg)

    def forward(self):
        self.g.forward()

    def backward(self):
        g = self.g
        for t in g.all_tensors:
            t._gw = False
            t.grad = None
        pending = list(self.buckets) if self.world_size > 1 else []
        works = []
        for idx in range(len(g.nodes) - 1, -1, -1):
            g.nodes[idx].backward()
            while pending and pending[0][2] >= idx:
                lo, hi, _ = pending.pop(0)
                works.append(torch.distributed.all_reduce(g.grad_arena[lo:hi], group=self.pg, async_op=True))
        for w in works:
            w.wait()

    def update(self):
        g = self.g
        fn.sgd_momentum(g.arena, g.grad_arena, g.mom_arena, self.lr, self.momentum, self.wd,
                        1.0 / (self.batch_size * self.world_size))

    def step(self):
        self.forward()
        self.backward()
        self.update()
