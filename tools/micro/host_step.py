"""Host time of the segments of one training step (vanilla API, S-1080p-1M): perf_counter around each Python-level call, no device
synchronisation added.  The forward call contains the frame's one wait (list length), so its figure includes blocking; the others
are pure enqueue time.  usage: python tools/micro/host_step.py [steps] [deferred 0|1]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import gspl_amd  # noqa: F401,E402
from gspl_amd import ops, synthetic  # noqa: E402
from gspl_amd.density import update_densification_stats  # noqa: E402
from gspl_amd.optimizers import FusedAdam  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
deferred = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
dev = torch.device("cuda:0")
wl = synthetic.WORKLOADS["S-1080p-1M"]
W, H = wl["width"], wl["height"]
means, scales, quats, opac, shs = synthetic.scene(wl["n"], seed=42)
tensors = [t.contiguous().to(dev).requires_grad_(True) for t in (means, scales, quats, opac, shs[:, :1], shs[:, 1:])]
m, s, q, o, dc, rest = tensors
cams = synthetic.camera_set(W, H, wl["fx"], count=16, distance=wl.get("distance", 4.0))
bg = torch.zeros(3, device=dev)
target = torch.full((3, H, W), 0.5, device=dev)
rasts = [ops.GaussianRasterizer(ops.GaussianRasterizationSettings(
    image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
    viewmatrix=c["world_to_camera"].to(dev), projmatrix=c["full_projection"].to(dev), sh_degree=3, campos=c["camera_center"].to(dev))) for c in cams]
names = ("means", "scales", "rotations", "opacities", "shs_dc", "shs_rest")
opt = FusedAdam([{"params": [t], "lr": 1e-6, "name": n} for t, n in zip(tensors, names)], eps=1e-15, deferred=("shs_rest",) if deferred else None)
accum, denom, max_radii = (torch.zeros(wl["n"], device=dev) for _ in range(3))
T = {k: 0.0 for k in ("zero", "forward", "loss", "backward", "stats", "optimizer")}
pc = time.perf_counter


def one(k, acc):
    t0 = pc()
    for t in tensors:
        t.grad = None
    screen = torch.empty_like(m).requires_grad_(True)
    t1 = pc()
    render, radii = rasts[k % 16](means3D=m, means2D=screen, opacities=o, shs=dc, shs_rest=rest, scales=s, rotations=q)
    t2 = pc()
    loss = ops.photometric_loss(render, target, 0.2)
    t3 = pc()
    loss.backward()
    t4 = pc()
    with torch.no_grad():
        update_densification_stats(screen.grad, None, radii, accum, denom, max_radii, scale=None)
        t5 = pc()
        opt.step()
    t6 = pc()
    if acc:
        for key, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
            T[key] += v


for k in range(20):
    one(k, False)
import gc
gc.collect(); gc.freeze()
torch.cuda.synchronize()
t0 = pc()
for k in range(steps):
    one(k, True)
host = pc() - t0
torch.cuda.synchronize()
total = pc() - t0
print(f"deferred={deferred}: {steps} steps, {total / steps * 1e3:.4f} ms per step (host loop {host / steps * 1e3:.4f})")
print("host ms per step: " + ", ".join(f"{k} {v / steps * 1e3:.4f}" for k, v in T.items()) + f"; sum without forward {sum(v for k, v in T.items() if k != 'forward') / steps * 1e3:.4f}")
