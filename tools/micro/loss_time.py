import torch, time, sys
sys.path.insert(0, "/root/repo")
import gspl_amd
from gspl_amd import ops, _lib as L
g = torch.Generator().manual_seed(1)
a = torch.rand(3, 1080, 1920, generator=g).cuda().requires_grad_(True)
b = torch.rand(3, 1080, 1920, generator=g).cuda()
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("fwd eval (no maps): %.3f ms" % t(lambda: ops.l1_ssim(a.detach(), b, train=False)))
print("fwd train (maps):   %.3f ms" % t(lambda: ops.l1_ssim(a, b, train=True)))
l1, s = ops.l1_ssim(a, b)
def bw():
    a.grad = None; (0.8 * l1 + 0.2 * (1 - s)).backward(retain_graph=True)
print("bwd only:           %.3f ms" % t(bw))
