"""Host time of one fused-Adam step (tiny tensors: the launch is negligible, ms/step ~ host time).  usage: python tools/micro/host_adam.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import gspl_amd  # noqa: F401
from gspl_amd import optimizers as gopt

dev = torch.device("cuda", 0)
N = 2000
shapes = [(N, 3), (N, 3), (N, 4), (N, 1), (N, 1, 3), (N, 15, 3)]
params = [torch.randn(s, device=dev).requires_grad_(True) for s in shapes]
opt = gopt.FusedAdam([{"params": [p], "lr": 1e-3, "name": str(i)} for i, p in enumerate(params)], eps=1e-15)
grads = [[torch.randn_like(p) for p in params] for _ in range(2)]
for k in range(20):
    for p, g in zip(params, grads[k % 2]):
        p.grad = g
    opt.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(2000):
    for p, g in zip(params, grads[k % 2]):
        p.grad = g
    opt.step()
torch.cuda.synchronize()
print("fused Adam host time per step: %.1f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
