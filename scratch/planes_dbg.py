import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspnet_amd import engine as E, functional as fn
from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
net = get_multi_symbol_train("resnet-50", 512, num_classes=8, batch_size=2, device=torch.device("cuda", 0))
bns = [n for n in net.g.nodes if isinstance(n, E.BatchNorm)]
print("BatchNorms", len(bns), "with bwd_sums", sum(n.bwd_sums is not None for n in bns), "dx_planes", sum(n.dx_planes for n in bns))
for n in bns[:12]:
    prod = getattr(n.x, "producer", None)
    print(n.beta.name, "bwd_sums", n.bwd_sums is not None, "completes", n.completes_x_grad, "tile_stats", n.tile_stats is not None,
          "prod", type(prod).__name__, "minmax", getattr(prod, "out_minmax", None) is not None, "planes", n.dx_planes)
