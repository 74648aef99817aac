import sys; sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn
x = torch.randn(4, 16, 16, 64, device="cuda"); dy = torch.randn_like(x)
gamma = torch.rand(64, device="cuda") + 0.5; beta = torch.randn(64, device="cuda")
mean, rstd, scale, shift = fn.bn_stats(x, 2e-5, gamma, beta)
torch.cuda.synchronize(); print("stats ok")
try:
    fn.bn_backward(x, scale, shift, dy, mean, rstd, gamma, relu=True)
    torch.cuda.synchronize(); print("bwd ok")
except Exception as e:
    print("ERR", e)
am = torch.zeros(64, device="cuda")
fn.bn_backward(x, scale, shift, dy, mean, rstd, gamma, relu=True, dx_absmax=am)
torch.cuda.synchronize(); print("bwd+am ok", float(am.max()))
