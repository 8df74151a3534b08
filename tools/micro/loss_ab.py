import sys, torch, os
sys.path.insert(0, "/root/repo")
import gspl_amd
from gspl_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
a = torch.rand(3, 1080, 1920, generator=g).to(dev).requires_grad_(True)
b = torch.rand(3, 1080, 1920, generator=g).to(dev)
def step():
    a.grad = None
    ops.photometric_loss(a, b, 0.2).backward()
for _ in range(10): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): step()
e1.record(); torch.cuda.synchronize()
print(os.path.basename(os.environ.get("GSPL_HIP_LIB", "base")), "%.1f us per fwd+bwd (incl. launch gaps)" % (e0.elapsed_time(e1) * 1000 / 200))
