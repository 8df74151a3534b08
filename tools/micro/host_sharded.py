"""Host time per call of the pieces of the sharded renderer's step (no profiler: perf_counter around the calls, the device runs
asynchronously).  usage: python tools/micro/host_sharded.py [steps]"""
import sys, time, collections
sys.path.insert(0, "/root/repo")
import torch
import gspl_amd  # noqa: F401
from gspl_amd import ops, synthetic, distributed as D
from gspl_amd.renderers import HipGSplatDistributedRenderer
from gspl_amd.renderers import hip_gsplat_distributed_renderer as R
from gspl_amd.optimizers import FusedAdam

acc = collections.defaultdict(lambda: [0.0, 0])


def timed(mod, name, label=None):
    fn = getattr(mod, name)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc[label or name]; e[0] += time.perf_counter() - t; e[1] += 1
    setattr(mod, name, w)


for n in ("fully_fused_projection", "sh_view_colors_batched", "pack_visible_records", "unpack_visible_records", "bin_gaussians_begin",
          "bin_gaussians_end", "rasterize_to_pixels", "photometric_loss", "unbind_cameras"):
    timed(ops, n)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
wl = synthetic.WORKLOADS["S-1080p-1M"]
W, H = wl["width"], wl["height"]
model = synthetic.ModelObject(*[t.to(dev) for t in synthetic.scene(wl["n"], seed=42)])
cam = synthetic.CameraObject(synthetic.camera(W, H, wl["fx"]), dev, idx=0)
r = HipGSplatDistributedRenderer(tile_based_culling=True).instantiate()
r.world_size, r.global_rank = 1, 0
r.camera_lookup = lambda idx, training: cam
r.train()
timed(r, "gather_cameras"); timed(r, "batch_project"); timed(r, "isect_encode")
bg = torch.zeros(3, device=dev)
target = torch.full((3, H, W), 0.5, device=dev)
tensors = model.leaves()
opt = FusedAdam([{"params": [t], "lr": 1e-6} for t in tensors], eps=1e-15)
tot = collections.defaultdict(float)
for k in range(steps + 20):
    if k == 20:
        torch.cuda.synchronize(); acc.clear(); tot.clear(); t_all = time.perf_counter()
    for t in tensors:
        t.grad = None
    t0 = time.perf_counter()
    out = r(cam, model, bg)
    t1 = time.perf_counter()
    loss = ops.photometric_loss(out["render"], target, 0.2)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    tot["forward"] += t1 - t0; tot["loss"] += t2 - t1; tot["backward"] += t3 - t2; tot["adam"] += t4 - t3
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / steps
print(f"wall {wall * 1e3:.3f} ms/step; host: " + ", ".join(f"{k} {v / steps * 1e6:.0f} us" for k, v in tot.items()))
for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:28s} {s / steps * 1e6:7.1f} us/step  ({n / steps:.1f} calls)")
