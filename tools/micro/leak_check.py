import sys, torch
sys.path.insert(0, "/root/repo")
import gspl_amd, bench
from gspl_amd import synthetic, ops
wl = synthetic.WORKLOADS["S-800-100k"]
dev = torch.device("cuda:0")
means, scales, quats, opac, shs = synthetic.scene(wl["n"], seed=42)
cam = synthetic.camera(wl["width"], wl["height"], wl["fx"])
tensors = [t.to(dev).requires_grad_(True) for t in (means, scales, quats, opac, shs)]
for api in ("vanilla", "gsplat"):
    step = bench.make_step(api, dev, wl, cam, tensors, "photometric")
    for i in range(5): step()
    torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated(); r0 = torch.cuda.memory_reserved()
    for i in range(300): step()
    torch.cuda.synchronize(); m1 = torch.cuda.memory_allocated(); r1 = torch.cuda.memory_reserved()
    for i in range(600): step()
    torch.cuda.synchronize(); m2 = torch.cuda.memory_allocated(); r2 = torch.cuda.memory_reserved()
    print(api, "allocated", m0, "->", m1, "->", m2, "reserved", r0, "->", r1, "->", r2, "pinned words", len(ops._PINNED_WORDS))
