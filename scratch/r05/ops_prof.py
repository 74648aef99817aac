"""MultiBoxTarget / MultiBoxDetection alone at the bench shapes (B 32, 6132 anchors, 9 classes): event-timed totals; run under
rocprofv3 --kernel-trace --stats for the per-kernel split.  usage: ops_prof.py [B]"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import operator as op, synthetic
from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
net = get_multi_symbol_train("resnet-50", 512, num_classes=8, batch_size=2, device=dev)
anchors = net.anchors
A = anchors.shape[1]
g = synthetic.rng(7)
lab = torch.from_numpy(synthetic.det_labels(B, gen=g)).to(dev)
pred = torch.randn(B, 9, A, device=dev)
prob = torch.softmax(pred, dim=1).contiguous()
loc = 0.1 * torch.randn(B, A * 5, device=dev)
det_out = torch.empty(B, A, 7, device=dev)
def timed(f, reps=20):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print("B %d A %d" % (B, A))
print("MultiBoxTarget     %.4f ms" % timed(lambda: op.MultiBoxTarget(anchors, lab, pred, negative_mining_ratio=3)))
print("MultiBoxDetection  %.4f ms (topk 400)" % timed(lambda: op.MultiBoxDetection(prob, loc, anchors, nms_topk=400, out=det_out)))
print("MultiBoxDetection  %.4f ms (threshold 0.2)" % timed(lambda: op.MultiBoxDetection(prob, loc, anchors, threshold=0.2, nms_topk=400, out=det_out)))
print("MultiBoxDetection  %.4f ms (force)" % timed(lambda: op.MultiBoxDetection(prob, loc, anchors, force_suppress=True, out=det_out)))
