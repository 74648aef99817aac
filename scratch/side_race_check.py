"""race check of the side-stream detection branch: N SGD steps on one synthetic batch, a 64-bit hash of the parameter arena and
of the gradient arena after every step.  Run with DSPN_DET_SIDE=0 and with the default: the streams differ, the order of
every accumulation does not, so the two outputs must be identical line by line (and so must two runs of the same setting)."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dspnet_amd import synthetic
from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
from dspnet_amd.train.solver import MultiTaskSolver
dev = torch.device("cuda", 0)
B, S, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 512, int(sys.argv[2]) if len(sys.argv) > 2 else 25
net = get_multi_symbol_train("resnet-50", S, num_classes=8, batch_size=B, device=dev, seed=0)
g = synthetic.rng(3)
solver = MultiTaskSolver(net)
solver.set_batch(torch.from_numpy(synthetic.images(B, S, S, g)).to(dev),
                 torch.from_numpy(synthetic.det_labels(B, gen=g, height=S, width=S, first_empty=False)).to(dev),
                 torch.from_numpy(synthetic.seg_labels(B, S, S, gen=g)).to(dev))
for step in range(steps):
    solver.step()
    torch.cuda.synchronize()
    h = lambda t: hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]
    print(step, h(net.g.arena), h(net.g.grad_arena), flush=True)
