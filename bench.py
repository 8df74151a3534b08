#!/usr/bin/env python
"""
bench.py — headline benchmark of the rasterizer hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload S-1080p-1M] [--api vanilla|gsplat]

A "step" is one pass of the hot path over one camera of synthetic input: renderer forward
(preprocess + SH + binning + compositing), an L1 image loss, and the full backward down to the
activated Gaussian properties (means, scales, rotations, opacities, SH).  Inputs are resident in HBM
before the timed region.  `value` = images/s over all ranks; `ms_per_step` = wall per step.

Multi-GPU (driver launches `torch.distributed.run ... bench.py --gpus N`): one process per GPU over
RCCL; Gaussians replicated, one camera per rank per step (weak scaling), and — as BASELINE.json's
north_star prescribes — an all-reduce of the densification statistics only (per-Gaussian screen-space
gradient norm: SUM, visibility count: SUM, max radius: MAX; 12 B/Gaussian), issued when a densification would
consume them (every 100 steps, the reference's cadence) and once at the end of the timed region.

Extra objects on the JSON line:
  roofline      dominant kernel = composite backward; achieved = algorithmic bytes (76*I + 20*P,
                SURVEY.md §8d) / its mean launch duration measured with HIP events inside the timed steps;
                peak = 8000 GB/s (MI355X_MICROARCH.md).  `traffic` = PMC-measured HBM bytes per launch when
                profiles/ holds them for this round, else null.
  cpu_baseline  the oracle (kind "port": torch fp32 projection+SH restatement of the reference's Python +
                the OpenMP C compositing loops) timed on the host cores, rank 0, N=1 only, one bounded pass.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md "HBM3E peak BW"
FP32_PEAK_TFLOPS = 157.3       # vector fp32


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=30)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--workload", default="S-1080p-1M")
    p.add_argument("--api", default="vanilla", choices=["vanilla", "gsplat"])
    p.add_argument("--loss", default="photometric", choices=["l1", "photometric"],
                   help="l1: mean |render - target| with torch ops; photometric: the reference's training loss "
                        "0.8 L1 + 0.2 (1 - SSIM) (vanilla_metrics.py:66-68) through the fused HIP loss kernels")
    p.add_argument("--optimizer", default="none", choices=["none", "fused-adam", "selective-adam", "torch-adam"],
                   help="optionally put an optimizer step inside the timed step (single-GPU study; the multi-GPU protocol of "
                        "north_star exchanges densification statistics only, so the default step has none)")
    p.add_argument("--stage-times", action="store_true",
                   help="time EVERY C-ABI call with events (stages_ms); default: only the compositing kernels the roofline needs")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample", default="auto", help="workload name for the CPU baseline leg, or 'auto'")
    p.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL; default) or gloo (code-path test on one GPU)")
    p.add_argument("--share-device", action="store_true", help="testing only: every rank uses cuda:0")
    return p.parse_args()


def _mark():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def make_step(api, dev, wl, cam, tensors, loss_kind="l1"):
    """One training step (forward, loss, backward).  With state["marks"] = [] the step leaves three events per call
    (start, before backward, after backward) for the fwd_ms / bwd_ms of the bench line."""
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    m, s, q, o, c = tensors
    W, H = wl["width"], wl["height"]
    bg = torch.zeros(3, device=dev)
    target = torch.full((3, H, W), 0.5, device=dev)
    state = {}
    if loss_kind == "photometric":
        loss_fn = lambda img: ops.photometric_loss(img, target, 0.2)
    else:
        loss_fn = lambda img: (img - target).abs().mean()
    if api == "vanilla":
        settings = ops.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg, scale_modifier=1.0,
            viewmatrix=cam["world_to_camera"].to(dev), projmatrix=cam["full_projection"].to(dev), sh_degree=3,
            campos=cam["camera_center"].to(dev))
        rast = ops.GaussianRasterizer(settings)

        def step():
            for t in tensors:
                t.grad = None
            marks = state.get("marks")
            if marks is not None:
                marks.append(_mark())
            screen = torch.empty_like(m).requires_grad_(True)      # gradient carrier, as HipVanillaRenderer creates it (values unused)
            render, radii = rast(means3D=m, means2D=screen, opacities=o, shs=c, scales=s, rotations=q)
            loss = loss_fn(render)
            if marks is not None:
                marks.append(_mark())
            loss.backward()
            if marks is not None:
                marks.append(_mark())
            state["vs_grad"], state["radii"], state["loss"] = screen.grad, radii, loss
            state["grad_scale"] = None
            return state
    else:
        vm = cam["world_to_camera"].T.contiguous().to(dev)
        center = cam["camera_center"].to(dev)
        grad_scale = torch.tensor([0.5 * W, 0.5 * H], device=dev)      # what the renderers return as viewspace_points_grad_scale

        def step():
            for t in tensors:
                t.grad = None
            marks = state.get("marks")
            if marks is not None:
                marks.append(_mark())
            xys, depths, radii, conics, comp, tiles, _ = ops.project_gaussians(
                m, s, 1.0, q, vm[:3], cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W, 16)
            xys.retain_grad()
            opac = o * comp[:, None]
            # same order as HipGSplatRenderer.forward: count half of the binning, SH, emit half, compositing
            pending = ops.bin_gaussians_begin(xys, depths, radii, H, W, 16, conics=conics, opacities=opac)
            rgbs = ops.sh_view_colors(3, m, center, c, None, radii > 0)
            img = ops.rasterize_gaussians(xys, depths, radii, conics, tiles, rgbs, opac, H, W, 16, bg,
                                          isects=ops.bin_gaussians_end(pending), channels_first=True)
            loss = loss_fn(img)
            if marks is not None:
                marks.append(_mark())
            loss.backward()
            if marks is not None:
                marks.append(_mark())
            state["vs_grad"], state["radii"], state["loss"] = xys.grad, radii, loss
            state["grad_scale"] = grad_scale
            return state
    step.state = state
    return step


def densification_stats(state, accum, denom, max_radii):
    """What VanillaDensityControllerImpl.update_states accumulates
    (internal/density_controllers/vanilla_density_controller.py:101-123), on the fused kernel the package ships for it
    (gspl_amd.density.HipDensityStatsMixin): masked max of the radii, masked sum of the scaled gradient norms, masked count."""
    from gspl_amd.density import update_densification_stats
    update_densification_stats(state["vs_grad"], None, state["radii"], accum, denom, max_radii, scale=state["grad_scale"])


def cpu_baseline(workload_name, api):
    """Oracle port timed on the host cores (one bounded pass; fp32; all threads)."""
    import numpy as np
    from gspl_amd import synthetic
    from oracle import gsplat_oracle as O
    wl = synthetic.WORKLOADS[workload_name]
    # elementwise torch ops degrade badly when oversubscribed (28 s vs ~1 s for the projection on a
    # 256-thread host), so the baseline uses at most 32 threads and reports that number
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    means, scales, quats, opac, shs = synthetic.scene(wl["n"], seed=42)
    cam = synthetic.camera(wl["width"], wl["height"], wl["fx"])
    W, H = wl["width"], wl["height"]
    leaves = [t.clone().requires_grad_(True) for t in (means, scales, quats, opac, shs)]
    m, s, q, o, c = leaves
    lib = O._lib()
    import ctypes
    lib.oracle_set_threads(ctypes.c_int(cores))

    t0 = time.perf_counter()
    xys, depths, radii, conics, comp, n_tiles, _, mask, _, _ = O.project_gaussians(
        m, s, 1.0, q, cam["world_to_camera"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W)
    rgbs = O.sh_colors(3, c, m, cam["camera_center"], detach_dirs=True)
    op = o.reshape(-1) * comp
    t1 = time.perf_counter()
    tiles, ids, flat, offs = O.isect_tiles(O.MODE_GSPLAT, xys, radii, depths, W, H)
    t2 = time.perf_counter()
    f = lambda t: np.ascontiguousarray(t.detach().numpy(), np.float32)
    a_xy, a_con, a_col, a_op = f(xys), f(conics), f(rgbs), f(op)
    bg = np.zeros(3, np.float32)
    tw, th = (W + 15) // 16, (H + 15) // 16
    out = np.empty((H, W, 3), np.float32)
    alpha = np.empty((H, W), np.float32)
    last = np.empty((H, W), np.int32)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.oracle_composite_fwd_f32(ctypes.c_int(0), ctypes.c_int64(flat.shape[0]), ctypes.c_int(3), P(a_xy), P(a_con), P(a_col),
                                 P(a_op), P(bg), ctypes.c_int(W), ctypes.c_int(H), ctypes.c_int(tw), ctypes.c_int(th),
                                 P(offs), P(flat), P(out), P(alpha), P(last))
    t3 = time.perf_counter()
    v_out = (np.sign(out - 0.5) / out.size).astype(np.float32)
    g_xy, g_con, g_col, g_op = (np.zeros_like(a) for a in (a_xy, a_con, a_col, a_op))
    lib.oracle_composite_bwd_f32(ctypes.c_int(0), ctypes.c_int(wl["n"]), ctypes.c_int64(flat.shape[0]), ctypes.c_int(3),
                                 P(a_xy), P(a_con), P(a_col), P(a_op), P(bg), ctypes.c_int(W), ctypes.c_int(H),
                                 ctypes.c_int(tw), ctypes.c_int(th), P(offs), P(flat), P(alpha), P(last), P(v_out),
                                 P(g_xy), P(g_con), P(g_col), P(g_op))
    t4 = time.perf_counter()
    torch.autograd.backward([xys, conics, rgbs, op],
                            [torch.from_numpy(g_xy), torch.from_numpy(g_con), torch.from_numpy(g_col), torch.from_numpy(g_op)])
    t5 = time.perf_counter()
    total = t5 - t0
    return {
        "value": 1.0 / total, "unit": "images/s", "cores": cores, "kind": "port",
        "sample": f"one fwd+bwd pass of {workload_name} (N={wl['n']}, {W}x{H}, I={int(flat.shape[0])}), fp32, "
                  f"torch CPU projection+SH (restating the reference's gaussian_projection.py/sh_utils.py) + OpenMP C compositing",
        "ms": {"project_sh_fwd": (t1 - t0) * 1e3, "binning": (t2 - t1) * 1e3, "composite_fwd": (t3 - t2) * 1e3,
               "composite_bwd": (t4 - t3) * 1e3, "project_sh_bwd": (t5 - t4) * 1e3},
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP extension is the only compute path)"
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    import gspl_amd  # noqa: F401
    from gspl_amd import _lib, synthetic
    _lib.lib()
    wl = synthetic.WORKLOADS[args.workload]
    means, scales, quats, opac, shs = synthetic.scene(wl["n"], seed=42)
    # every rank renders its own camera (cameras sharded): a small per-rank dolly keeps the work equal
    cam = synthetic.camera(wl["width"], wl["height"], wl["fx"], distance=wl.get("distance", 4.0) + 0.01 * rank)
    tensors = [t.to(dev).requires_grad_(True) for t in (means, scales, quats, opac, shs)]
    step = make_step(args.api, dev, wl, cam, tensors, args.loss)
    optimizer = None
    if args.optimizer != "none":
        # eps 1e-15 as the reference (internal/models/vanilla_gaussian.py:266-300).  The bench tensors are ACTIVATED values
        # (post-exp scales, post-sigmoid opacities), so the reference's learning rates — meant for the raw parameters —
        # are scaled down by 1e3: the optimizer's cost is measured without letting the synthetic scene drift.
        groups = [{"params": [t], "lr": lr * 1e-3} for t, lr in zip(tensors, (1.6e-4, 5e-3, 1e-3, 5e-2, 2.5e-3))]
        if args.optimizer == "torch-adam":
            optimizer = torch.optim.Adam(groups, eps=1e-15)
        else:
            from gspl_amd import optimizers as gopt
            optimizer = (gopt.FusedAdam if args.optimizer == "fused-adam" else gopt.SelectiveAdam)(groups, eps=1e-15)
    N = wl["n"]
    accum = torch.zeros(N, device=dev)
    denom = torch.zeros(N, device=dev)
    max_radii = torch.zeros(N, device=dev)                       # float, as the reference's buffer (vanilla_density_controller.py:61)

    from gspl_amd import distributed as gdist
    DENSIFY_INTERVAL = 100      # the reference consumes the statistics every 100 steps (vanilla_density_controller.py:16,86)
    counter = {"n": 0}

    def full_step(force_reduce=False):
        st = step()
        with torch.no_grad():
            densification_stats(st, accum, denom, max_radii)
            if optimizer is not None:
                if args.optimizer == "selective-adam":
                    optimizer.step(st["radii"] > 0)
                else:
                    optimizer.step()
            counter["n"] += 1
            # statistics are accumulated locally and made identical on all ranks when a densification would consume
            # them (every DENSIFY_INTERVAL steps) — and once at the end of the timed region so that every run pays
            # for at least one reduction
            if dist is not None and (force_reduce or counter["n"] % DENSIFY_INTERVAL == 0):
                gdist.reduce_densification_stats(accum, denom, max_radii)
        return st

    for _ in range(args.warmup):
        full_step(force_reduce=True)
    # Host hygiene: a full (generation-2) pass of Python's cyclic GC walks every object torch has imported — 30-40 ms during
    # which no kernel is launched, once every ~130 steps (seen as one 38 ms step in GSPL_BENCH_DUMP=1 runs).  Freezing what
    # exists after warm-up keeps later collections to the objects of the steps themselves.
    import gc
    gc.collect()
    gc.freeze()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    step.state["marks"] = []
    _lib.profile_start(None if args.stage_times else ("gspl_composite_bwd_packed", "gspl_composite_bwd", "gspl_composite_fwd"))
    t0 = time.perf_counter()
    for k in range(args.steps):
        st = full_step(force_reduce=(k == args.steps - 1))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = _lib.profile_stop()
    marks = step.state.pop("marks")
    phase_fwd = sum(marks[i].elapsed_time(marks[i + 1]) for i in range(0, len(marks), 3)) / args.steps
    phase_bwd = sum(marks[i + 1].elapsed_time(marks[i + 2]) for i in range(0, len(marks), 3)) / args.steps
    if os.environ.get("GSPL_BENCH_DUMP") and rank == 0:
        # per-step device times, for hunting outliers: step span = start of step i to start of step i+1
        starts = marks[0::3]
        spans = [starts[i].elapsed_time(starts[i + 1]) for i in range(len(starts) - 1)]
        fw = [marks[i].elapsed_time(marks[i + 1]) for i in range(0, len(marks), 3)]
        srt = sorted(spans)
        print("step spans ms: median %.3f p90 %.3f p99 %.3f max %.3f; outliers (>2x median): %s" % (
            srt[len(srt) // 2], srt[int(len(srt) * 0.9)], srt[int(len(srt) * 0.99)], srt[-1],
            [(i, round(x, 2), round(fw[i], 2)) for i, x in enumerate(spans) if x > 2 * srt[len(srt) // 2]][:40]), file=sys.stderr)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        mean = lambda name: (sum(prof[name]) / len(prof[name])) if prof.get(name) else None
        # per-STEP totals (an entry point called twice per step, e.g. the two phases of the Inria preprocess, counts twice)
        stages = {k: round(sum(v) / args.steps, 4) for k, v in prof.items()}
        # intersections of this workload (for the algorithmic byte model)
        from gspl_amd import ops
        with torch.no_grad():
            if args.api == "vanilla":
                I = None
            m, s, q, o, c = tensors
            vm = cam["world_to_camera"].T.contiguous().to(dev)
            _, _, radii, _, _, tiles, _ = ops.project_gaussians(m, s, 1.0, q, vm[:3], cam["fx"], cam["fy"], cam["cx"], cam["cy"],
                                                                wl["height"], wl["width"], 16)
            I_gsplat = int(tiles.sum().item())
        P = wl["width"] * wl["height"]
        bwd_ms = mean("gspl_composite_bwd_packed") or mean("gspl_composite_bwd")
        fwd_ms = mean("gspl_composite_fwd")
        # vanilla rect convention gives a slightly different I; measure it from the sort call count instead
        I = I_gsplat
        alg_bytes = 76.0 * I + 20.0 * P
        roofline = None
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                traffic = json.load(f).get(f"{args.workload}/{args.api}", {}).get("composite_bwd_kernel", {}).get("traffic_bytes")
        except OSError:
            pass
        if bwd_ms:
            achieved = alg_bytes / (bwd_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": "composite_bwd_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "algorithmic_bytes": alg_bytes, "avg_ms": round(bwd_ms, 4), "intersections": I}
        line = {
            "metric": "training images/sec + fwd/bwd ms @1080p, 1M Gaussians, 1/2/4/8 MI355X",
            "value": round(world * args.steps / elapsed, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "api": args.api, "n_gaussians": N, "width": wl["width"], "height": wl["height"],
                       "sh_degree": 3, "loss": args.loss, "optimizer": args.optimizer,
                       "step": "renderer fwd + " + ("L1 loss" if args.loss == "l1" else "0.8 L1 + 0.2 (1-SSIM) loss (fused)") + " + full bwd + densification stats"
                               + ("" if args.optimizer == "none" else " + " + args.optimizer + " step"),
                       "parallelism": f"replicated Gaussians, {world} camera(s)/step, all-reduce of densification stats only"},
            "stages_ms": stages,
            # device time between the events at the start of the step, before loss.backward() and after it
            "fwd_ms": round(phase_fwd, 4),
            "bwd_ms": round(phase_bwd, 4),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = args.workload if args.cpu_sample == "auto" else args.cpu_sample
            try:
                line["cpu_baseline"] = cpu_baseline(sample, args.api)
            except Exception as e:  # the baseline leg must never take the bench line down
                line["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": os.cpu_count(), "kind": "port",
                                        "sample": f"failed: {e!r}"}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
