"""Host-side profile of the training step: how much CPU time one step costs and where (cProfile, cumulative).
A tiny scene makes the GPU work negligible, so ms/step ~ host time per step.
usage: python tools/micro/host_profile.py [n_gaussians] [width] [height]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
import gspl_amd  # noqa: F401
from gspl_amd import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 320
H = int(sys.argv[3]) if len(sys.argv) > 3 else 208
api = sys.argv[4] if len(sys.argv) > 4 else "vanilla"
dev = torch.device("cuda", 0)
wl = {"n": n, "width": W, "height": H, "fx": 0.8 * W}
cam = synthetic.camera(W, H, wl["fx"], distance=4.0)
tensors = [t.to(dev).requires_grad_(True) for t in synthetic.scene(n, seed=42)]
step = bench.make_step(api, dev, wl, cam, tensors, "photometric")
for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print("host-bound step: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
