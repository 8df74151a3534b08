"""Loop the two-rank shared-GPU sharded renderer (tests/test_distributed_renderer.py) ITERS times per exchange format and report,
per iteration, the worst gradient ratio against (a) the one-process HIP result and (b) the fp64 oracle; rows that leave the
tolerance are printed with their per-camera visibility, radii and screen-space gradients.  Development tool.

    python tools/diag/padded_loop.py [--iters 40] [--exchange padded counted]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def ratio(got, ref):
    got, ref = got.detach().cpu().double().numpy(), ref.detach().cpu().double().numpy()
    rms = float(np.sqrt(np.mean(ref * ref))) + 1e-30
    return np.abs(got - ref) / (np.abs(ref) + rms)


def worker(rank, world, port, iters, exchanges, tol):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gspl_amd  # noqa: F401
    from gspl_amd import distributed as D
    from gspl_amd.renderers import HipGSplatDistributedRenderer
    from fakes import FakeCamera, FakePropertyModel
    import test_distributed_renderer as T
    dev = torch.device("cuda:0")
    params = T._scene(torch.float32)
    cams, weights = T._cameras(world), T._weights(torch.float32, world)
    bg = torch.tensor([0.2, 0.1, 0.4])
    N = params[0].shape[0]
    lo, hi = D.shard_bounds(N, world, rank)
    names = ("means", "scales", "rotations", "opacities", "shs_dc", "shs_rest")
    # fp64 oracle of the one-process result, once
    p64 = [p.double() for p in params]
    _, oracle_grads, oracle_xy = T._reference_full_model(False, p64, cams, [w.double() for w in weights], bg.double(), torch.device("cpu"))
    camset = [FakeCamera(c, dev) for c in cams]
    n_bad = 0
    for exchange in exchanges:
        for it in range(iters):
            model = FakePropertyModel(*[p.clone().to(dev) for p in params], extra={"ids": torch.arange(N, dtype=torch.float32).to(dev)})
            opt = model.named_optimizer()
            module = T._Module(model, [opt], world, rank, dev)
            renderer = HipGSplatDistributedRenderer(exchange=exchange).instantiate()
            renderer.training_setup(module)
            renderer.camera_lookup = lambda idx, training: camset[idx]
            renderer.train()
            out = renderer(camset[rank], model, bg.to(dev))
            for r in out["projection_results_list"]:
                r[1].retain_grad()
            (out["render"] * weights[rank].to(dev)).sum().backward()
            ref_renders, ref_grads, ref_xy = T._reference_full_model(True, params, cams, weights, bg, dev)
            torch.cuda.synchronize()
            line = []
            for name, ref, orc in zip(names, ref_grads, oracle_grads):
                got = model.get_property(name).grad
                r_hip, r_orc, r_ref = ratio(got, ref[lo:hi]), ratio(got, orc[lo:hi]), ratio(ref[lo:hi], orc[lo:hi])
                line.append(f"{name} {r_hip.max():.1e}/{r_orc.max():.1e}/{r_ref.max():.1e}")
                if r_hip.max() > tol:
                    n_bad += 1
                    rows = np.unique(np.argwhere(r_hip > tol)[:, 0])
                    print(f"[rank {rank}] {exchange} it {it} {name}: {len(rows)} rows outside {tol:g} (sharded vs one-process HIP), worst {r_hip.max():.3e}", flush=True)
                    for row in rows[:6]:
                        g = got[row].detach().cpu().numpy().ravel()[:6]
                        h = ref[lo + row].cpu().numpy().ravel()[:6]
                        o = orc[lo + row].numpy().ravel()[:6]
                        vis = [bool(out["visible_mask_list"][c][row]) for c in range(world)]
                        rad = [int(out["projection_results_list"][c][0][row]) for c in range(world)]
                        xyg = [out["projection_results_list"][c][1].grad[row].cpu().numpy() for c in range(world)]
                        xyr = [ref_xy[c].reshape(N, 2)[lo + row].numpy() for c in range(world)]
                        print(f"    row {lo + row}: sharded {g}\n      one-process HIP {h}\n      oracle {o}\n      visible {vis} radii {rad}"
                              f"\n      xy grad sharded {xyg}\n      xy grad one-process {xyr}", flush=True)
            dr = float((out["render"].detach().cpu() - ref_renders[rank]).abs().max())
            if rank == 0 or dr > 2e-5:
                print(f"[rank {rank}] {exchange} it {it}: render {dr:.1e} | (vs HIP / vs oracle / HIP vs oracle) " + " | ".join(line), flush=True)
    print(f"[rank {rank}] done, {n_bad} gradient tensors outside {tol:g}", flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--exchange", nargs="+", default=["padded", "counted"])
    ap.add_argument("--tol", type=float, default=2e-4)
    args = ap.parse_args()
    from conftest import free_port
    mp.spawn(worker, args=(2, free_port(), args.iters, args.exchange, args.tol), nprocs=2, join=True)


if __name__ == "__main__":
    main()
