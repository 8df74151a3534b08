"""Launch times of the fused photometric loss (forward / backward) at 1080p, HIP events on the current stream.
usage: python tools/micro/loss_time.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import gspl_amd  # noqa: F401,E402
from gspl_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
img = torch.rand(3, 1080, 1920, generator=g).to(dev).requires_grad_(True)
gt = torch.rand(3, 1080, 1920, generator=g).to(dev)
for _ in range(10):
    img.grad = None
    ops.photometric_loss(img, gt, 0.2).backward()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * iters)]
for i in range(iters):
    img.grad = None
    ev[3 * i].record()
    loss = ops.photometric_loss(img, gt, 0.2)
    ev[3 * i + 1].record()
    loss.backward()
    ev[3 * i + 2].record()
torch.cuda.synchronize()
f = sorted(ev[3 * i].elapsed_time(ev[3 * i + 1]) for i in range(iters))
b = sorted(ev[3 * i + 1].elapsed_time(ev[3 * i + 2]) for i in range(iters))
print(f"photometric loss 3x1080x1920: forward (2 launches) median {f[iters // 2] * 1e3:.1f} us, backward (incl. autograd's fill) median {b[iters // 2] * 1e3:.1f} us; loss {float(loss):.6f}")
