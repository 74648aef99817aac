"""which operand magnitudes of a training step are still taken by a standalone pass over the tensor ("f16x2" math)"""
import sys, inspect
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import engine as E, functional as fn, synthetic
from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
net = get_multi_symbol_train("resnet-50", 512, num_classes=8, batch_size=B, device=dev)
gen = synthetic.rng(1)
net.data.data.copy_(torch.from_numpy(synthetic.images(B, 512, 512, gen)))
net.label_det.data.copy_(torch.from_numpy(synthetic.det_labels(B, gen=gen)))
net.label_seg.data.copy_(torch.from_numpy(synthetic.seg_labels(B, gen=gen)))
log = []
real = fn.absmax
def traced(x, in_affine=None, out=None):
    who, fnname = None, "?"
    for fr in inspect.stack()[1:8]:
        me = fr.frame.f_locals.get("self")
        if isinstance(me, E.Node):
            who, fnname = me, fr.function
            break
    w = getattr(who, "w", None)
    log.append((x.numel() * 4 / 1e6, "%s:%s" % (type(who).__name__, getattr(w, "name", "?")), fnname, in_affine is not None))
    return real(x, in_affine, out=out)
fn.absmax = traced
net.g.forward(); net.g.backward()
torch.cuda.synchronize()
log.sort(reverse=True)
print("standalone magnitude passes: %d, %.1f MB" % (len(log), sum(l[0] for l in log)))
for mb, name, f, aff in log[:60]:
    print("%9.1f MB  %-40s %-12s affine=%s" % (mb, name, f, aff))
