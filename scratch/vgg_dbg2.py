import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch, torch.nn.functional as F
from test_graph_gpu import _vgg_case
from oracle import dspnet_torch as ot
net, solver, data, lab, seg = _vgg_case("det", 300, 1, 20)
solver.forward(); torch.cuda.synchronize()
P = ot.Params(ot.export_params(net.g), torch.float64)
x = torch.tensor(data, dtype=torch.float64)
def c(x, name, pad=1, dil=1):
    return F.relu(F.conv2d(x, P[name + "_weight"], P[name + "_bias"], padding=pad, dilation=dil))
def rel(a, b): return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
def dev(name): return net.g.tensors[name].data.cpu().numpy().transpose(0, 3, 1, 2)
seq = [("conv1_1", 1, 1), ("conv1_2", 1, 1), "pool1", ("conv2_1", 1, 1), ("conv2_2", 1, 1), "pool2", ("conv3_1", 1, 1), ("conv3_2", 1, 1), ("conv3_3", 1, 1), "pool3",
       ("conv4_1", 1, 1), ("conv4_2", 1, 1), ("conv4_3", 1, 1), "pool4", ("conv5_1", 1, 1), ("conv5_2", 1, 1), ("conv5_3", 1, 1), "pool5", ("fc6", 6, 6), ("fc7", 0, 1)]
with torch.no_grad():
    for it in seq:
        if isinstance(it, str):
            if it == "pool3": x = F.max_pool2d(x, 2, 2, ceil_mode=True)
            elif it == "pool5": x = F.max_pool2d(x, 3, 1, 1)
            else: x = F.max_pool2d(x, 2, 2)
            d = dev(it)
        else:
            x = c(x, it[0], it[1], it[2]); d = dev(it[0] + "_out")
        print(it if isinstance(it, str) else it[0], tuple(x.shape), "rel err %.2e" % rel(d[:, :x.shape[1]], x.numpy()))
