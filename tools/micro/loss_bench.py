"""usage (on the GPU box): GSPL_HIP_LIB=<variant .so> python tools/micro/loss_bench.py
Launch durations of the fused photometric loss at 3 x 1080 x 1920 (forward kernels, backward kernel), HIP events, median of 30."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import gspl_amd  # noqa: F401
from gspl_amd import ops

g = torch.Generator().manual_seed(1)
a = torch.rand(3, 1080, 1920, generator=g).cuda().requires_grad_(True)
b = torch.rand(3, 1080, 1920, generator=g).cuda()
fw, bw = [], []
for it in range(40):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    a.grad = None
    e[0].record(); loss = ops.photometric_loss(a, b); e[1].record(); loss.backward(); e[2].record(); torch.cuda.synchronize()
    if it >= 10:
        fw.append(e[0].elapsed_time(e[1]) * 1e3); bw.append(e[1].elapsed_time(e[2]) * 1e3)
fw.sort(); bw.sort()
print(f"forward {fw[len(fw) // 2]:.1f} us  backward (incl. torch's grad fill) {bw[len(bw) // 2]:.1f} us  sum {fw[len(fw) // 2] + bw[len(bw) // 2]:.1f} us  loss {float(loss):.6f}")
