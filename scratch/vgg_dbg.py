import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import engine as E
from dspnet_amd.symbol import vgg16_reduced as v
from dspnet_amd.symbol.common import multi_layer_feature
from dspnet_amd.symbol.multitask_symbol_factory import get_config
dev = torch.device("cuda", 0)
g = E.Graph(dev)
data = g.tensor((1, 3, 300, 300), "data", requires_grad=False)
internals = v.get_symbol(g, data)
for k, t in internals.items(): print(k, t.shape)
c = get_config("vgg16_reduced", 300)
layers = multi_layer_feature(g, internals, c["from_layers"][1:], c["num_filters"][1:], c["strides"][1:], c["pads"][1:])
for t in layers: print(t.name, t.shape)
