"""How far do the gradients of the sharded renderer (one rank, exchange through RCCL or short-cut) differ between runs?"""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.getcwd())
import gspl_amd  # noqa
from gspl_amd import distributed as D, synthetic, ops
from gspl_amd.renderers import HipGSplatDistributedRenderer
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
W, H = 320, 240
scene = synthetic.scene(20000, seed=3)
cam = synthetic.CameraObject(synthetic.camera(W, H, 300.0), dev, idx=0)
bg = torch.zeros(3, device=dev)
def render(shortcut):
    D.SINGLE_RANK_SHORTCUT = shortcut
    model = synthetic.ModelObject(*[t.clone().to(dev) for t in scene])
    r = HipGSplatDistributedRenderer(tile_based_culling=True).instantiate()
    r.world_size, r.global_rank = 1, 0
    r.camera_lookup = lambda idx, training: cam
    r.train()
    out = r(cam, model, bg)
    out["render"].square().sum().backward()
    render.vis = out["visible_mask_list"][0]
    render.xy_grad = out["projection_results_list"][0][1].grad
    return out["render"].detach(), [t.grad for t in model.leaves()]
names = ("means", "scales", "quats", "opac", "shs_dc", "shs_rest")
img0, g0 = render(True)
for k in range(8):
    sc = bool(k % 2)
    img, g = render(sc)
    line = []
    for n, a, b in zip(names, g0, g):
        d = (a - b).abs()
        i = int(d.argmax())
        line.append(f"{n} {float(d.max()) / (float(a.abs().max()) + 1e-30):.1e}")
    print("shortcut" if sc else "rccl    ", "image equal" if torch.equal(img, img0) else "IMAGE DIFFERS", " ".join(line), ops.SPECULATION)
# where does the first frame differ from the last one?
a, b = g0[2], g[2]
d = (a - b).abs().sum(dim=1)
top = torch.topk(d, 8).indices
vis = render.vis
print("rows with a quat-gradient difference > 1e-9:", int((d > 1e-9).sum()), "of", d.numel(), "; visible rows:", int(vis.sum()))
for i in top.tolist():
    print(i, "visible" if bool(vis[i]) else "INVISIBLE", "first", [f"{x:.3e}" for x in a[i].tolist()], "last", [f"{x:.3e}" for x in b[i].tolist()],
          "means first/last", [f"{x:.3e}" for x in g0[0][i].tolist()], [f"{x:.3e}" for x in g[0][i].tolist()])
dist.destroy_process_group()
