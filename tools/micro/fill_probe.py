import sys, os
sys.path.insert(0, "/root/repo")
import torch, bench
import gspl_amd
from gspl_amd import synthetic
dev = torch.device("cuda", 0)
wl = synthetic.WORKLOADS["S-1080p-1M"]
cam = synthetic.camera(wl["width"], wl["height"], wl["fx"], distance=4.0)
tensors = [t.to(dev).requires_grad_(True) for t in synthetic.scene(wl["n"], seed=42)]
step = bench.make_step("vanilla", dev, wl, cam, tensors, "photometric")
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step()
    torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::ones_like", "aten::zeros_like") and e.device_time > 0:
        print(e.name, e.input_shapes, "dev_us=%.1f" % e.device_time)
