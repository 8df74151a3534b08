"""Fused Adam (gspl_selective_adam, the model's six tensors, no mask) at several model sizes, beside a plain streaming kernel of the
same read / write mix — what this part streams at 28 B per element.  usage: python tools/micro/adam_bandwidth.py [variant .so via GSPL_HIP_LIB]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import gspl_amd  # noqa: F401
from gspl_amd import optimizers as gopt

dev = torch.device("cuda", 0)


def timed(fn, n=8):
    ms = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ms.append(a.elapsed_time(b))
    return statistics.median(ms)


for N in (250_000, 1_000_000, 2_000_000, 4_000_000, 6_000_000):
    shapes = [(N, 3), (N, 3), (N, 4), (N, 1), (N, 1, 3), (N, 15, 3)]
    params = [torch.randn(s, device=dev).requires_grad_(True) for s in shapes]
    opt = gopt.FusedAdam([{"params": [p], "lr": 1e-3, "name": str(i)} for i, p in enumerate(params)], eps=1e-15)
    for p in params:
        p.grad = torch.randn_like(p)
    opt.step(); torch.cuda.synchronize()
    t = timed(opt.step)
    gb = N * 59 * 28 / 1e9
    # the same mix with torch: 4 reads + 3 writes of the biggest tensor (three fused-multiply-add style kernels would read more; this is
    # a lower bound of the traffic: out = a + b (2 reads, 1 write) three times over ... kept simple: one copy = 1 read + 1 write
    big = params[-1].detach()
    other = torch.empty_like(big)
    tc = timed(lambda: other.copy_(big))
    print(f"N={N:>8}: fused Adam {t:7.3f} ms = {gb / t:6.2f} TB/s... x1e-3" if False else
          f"N={N:>8}: fused Adam {t:7.3f} ms  {gb / (t * 1e-3) / 1e3:5.2f} TB/s   |  copy of shs_rest ({big.numel() * 8 / 1e9:.2f} GB moved) {tc:6.3f} ms  {big.numel() * 8 / 1e9 / (tc * 1e-3) / 1e3:5.2f} TB/s")
    del params, opt, big, other
    torch.cuda.empty_cache()
