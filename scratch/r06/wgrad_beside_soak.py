"""Round 6: the weight gradients on a stream of their own (engine.WGRAD_SIDE) are ordered against the step's stream by events; a
missed ordering would show as run-to-run differences or as a difference from the one-stream schedule.  STEPS training steps at the bench
shape from the same state: twice with the weight gradients beside the chain, once on one stream; the parameter arenas must be bit-identical."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hashlib
import torch
from dspnet_amd import engine as E, synthetic
from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
from dspnet_amd.train.solver import MultiTaskSolver
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)


def run(side):
    E.WGRAD_SIDE = side
    net = get_multi_symbol_train("resnet-50", (3, 512, 512), num_classes=8, batch_size=32, device=dev, seed=0)
    gen = synthetic.rng(233)
    solver = MultiTaskSolver(net)
    solver.set_batch(torch.from_numpy(synthetic.images(32, 512, 512, gen)).to(dev),
                     torch.from_numpy(synthetic.det_labels(32, gen=gen, height=512, width=512)).to(dev),
                     torch.from_numpy(synthetic.seg_labels(32, 512, 512, gen=gen)).to(dev))
    for _ in range(STEPS):
        solver.step()
    torch.cuda.synchronize()
    a = net.g.arena.detach().cpu()
    h = hashlib.sha256(a.numpy().tobytes()).hexdigest()[:16]
    fin = bool(torch.isfinite(a).all())
    del solver, net
    import gc; gc.collect(); torch.cuda.empty_cache()
    return h, fin


r = [run(1), run(0), run(1)]
print("%d steps at 32 x 512 x 512: arena sha256 beside %s / one stream %s / beside %s  finite %s  -> %s" % (
    STEPS, r[0][0], r[1][0], r[2][0], all(x[1] for x in r), "bit-identical" if r[0][0] == r[1][0] == r[2][0] else "DIFFERENT"))
