"""cProfile of the sharded renderer's training step at W = 1 (bench.py --parallelism sharded): where the host time goes.
usage: python tools/micro/host_sharded_profile.py [steps]"""
import cProfile
import os
import pstats
import sys

import torch

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, root)
import gspl_amd  # noqa: F401,E402
from gspl_amd import ops, synthetic  # noqa: E402
from gspl_amd.renderers import HipGSplatDistributedRenderer  # noqa: E402
from gspl_amd.optimizers import FusedAdam  # noqa: E402
from gspl_amd.density import update_densification_stats  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
wl = synthetic.WORKLOADS["S-1080p-1M"]
W, H = wl["width"], wl["height"]
means, scales, quats, opac, shs = synthetic.scene(wl["n"], seed=42)
model = synthetic.ModelObject(*[t.contiguous().to(dev) for t in (means, scales, quats, opac, shs)])
cams = [synthetic.CameraObject(c, dev, idx=i) for i, c in enumerate(synthetic.camera_set(W, H, wl["fx"], count=16, distance=wl.get("distance", 4.0)))]
renderer = HipGSplatDistributedRenderer(tile_based_culling=True).instantiate()
renderer.world_size, renderer.global_rank = 1, 0
renderer.camera_lookup = lambda idx, training: cams[idx]
renderer.train()
bg = torch.zeros(3, device=dev)
target = torch.full((3, H, W), 0.5, device=dev)
tensors = model.leaves()
opt = FusedAdam([{"params": [t], "lr": 1e-6} for t in tensors], eps=1e-15)
N = wl["n"]
accum, denom, max_radii = (torch.zeros(N, device=dev) for _ in range(3))
grad_scale = torch.tensor([0.5 * W, 0.5 * H], device=dev)


def step(k):
    for t in tensors:
        t.grad = None
    out = renderer(cams[k % 16], model, bg)
    for r in out["projection_results_list"]:
        r[1].retain_grad()
    loss = ops.photometric_loss(out["render"], target, 0.2)
    loss.backward()
    with torch.no_grad():
        for r, vis in zip(out["projection_results_list"], out["visible_mask_list"]):
            update_densification_stats(r[1].grad, vis, r[0], accum, denom, max_radii, scale=grad_scale)
        opt.step()


for k in range(20):
    step(k)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for k in range(steps):
    step(k)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative")
import io
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(28)
print(f"{steps} steps under cProfile")
print("\n".join(l[:170] for l in buf.getvalue().splitlines()[4:44]))
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(30)
print("\n".join(l[:170] for l in buf.getvalue().splitlines()[4:44]))
