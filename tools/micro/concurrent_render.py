"""Re-entrancy under load: T host threads, each on its own stream, render the same 1 M-splat scene concurrently (the viewer
situation: one thread per client calling `forward` under no_grad, SURVEY.md §8b "threads / streams").  Checks that every
frame is identical to the single-thread frame and that nothing dead-locks (the look-back kernels trap instead of hanging).
usage: python tools/micro/concurrent_render.py [threads] [frames]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import gspl_amd  # noqa: F401
from gspl_amd import ops, synthetic

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
FRAMES = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
wl = synthetic.WORKLOADS["S-1080p-1M"]
cam = synthetic.camera(wl["width"], wl["height"], wl["fx"], distance=4.0)
m, s, q, o, c = [t.to(dev) for t in synthetic.scene(wl["n"], seed=42)]
settings = ops.GaussianRasterizationSettings(
    image_height=wl["height"], image_width=wl["width"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
    scale_modifier=1.0, viewmatrix=cam["world_to_camera"].to(dev), projmatrix=cam["full_projection"].to(dev), sh_degree=3,
    campos=cam["camera_center"].to(dev))
rast = ops.GaussianRasterizer(settings)


def frame():
    with torch.no_grad():
        img, radii = rast(means3D=m, means2D=torch.empty_like(m), opacities=o, shs=c, scales=s, rotations=q)
    return img


ref = frame().clone()
torch.cuda.synchronize()
errors = []


def worker(k):
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        for _ in range(FRAMES):
            img = frame()
            if not torch.equal(img, ref):
                errors.append((k, float((img - ref).abs().max())))
        st.synchronize()


t0 = time.perf_counter()
threads = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
for th in threads:
    th.start()
for th in threads:
    th.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{T} threads x {FRAMES} frames: {T * FRAMES / dt:.1f} frames/s, mismatching frames: {len(errors)} {errors[:3]}")
assert not errors
