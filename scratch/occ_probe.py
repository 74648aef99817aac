"""DSPN_DEBUG_PRINT=1 python scratch/occ_probe.py: one 3x3 and one 1x1 convolution (default math) -> the occupancy lines of their kernels"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn
x = torch.randn(32, 32, 32, 256, device="cuda"); w = torch.randn(256, 3, 3, 256, device="cuda") * 0.05
y = fn.conv2d_forward(x, w, None, 1, 1, 1)
w1 = torch.randn(256, 1, 1, 256, device="cuda") * 0.05
y1 = fn.conv2d_forward(x, w1, None, 1, 0, 1)
torch.cuda.synchronize()
