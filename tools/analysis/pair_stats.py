"""Offline statistics of the compositing backward's work (CPU only; development tool for DESIGN §4.2).

    python tools/analysis/pair_stats.py [--workload S-1080p-1M] [--n N]

Projects the synthetic scene with the oracle's Inria preprocess (fp32), builds the tile lists and prints, per shape of
the culling unit, candidates visited, pixel slots and lane utilisation (valid pairs / slots)."""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gsplat_oracle as O  # noqa: E402

SHAPES = [(16, 16), (16, 8), (8, 8), (16, 4), (8, 4), (4, 4), (16, 2), (8, 2), (4, 2), (16, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--fx", type=float, default=1600.0)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--distance", type=float, default=4.0)
    args = ap.parse_args()
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "_pair_stats.so")
    src = os.path.join(here, "pair_stats.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", src, "-o", so, "-lm"])
    lib = ctypes.CDLL(so)
    W, H = args.width, args.height
    means, scales, quats, opac, shs = O.synthetic_scene(args.n)
    cam = O.synthetic_camera(W, H, args.fx, distance=args.distance)
    with torch.no_grad():
        xy, depths, radii, conics, mask = O.inria_preprocess(means, scales, 1.0, quats, cam["world_to_camera"], cam["full_projection"],
                                                             cam["tanfovx"], cam["tanfovy"], H, W)
    tiles, ids, flat, offs = O.isect_tiles(O.MODE_INRIA, xy, radii, depths, W, H)
    print(f"N={args.n} visible={int((radii > 0).sum())} I={flat.shape[0]}")
    xy, conics, op = O._f32(xy), O._f32(conics), O._f32(opac).reshape(-1)
    offs, flat = O._i32(offs).reshape(-1), O._i32(flat)
    tw, th = (W + 15) // 16, (H + 15) // 16
    out = np.zeros((len(SHAPES), 6))
    hist = np.zeros(257)
    lay = np.zeros(8)
    rm = np.zeros((4, 3))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.pair_stats(ctypes.c_int(1), ctypes.c_int64(flat.shape[0]), p(xy), p(conics), p(op), ctypes.c_int(W), ctypes.c_int(H),
                   ctypes.c_int(tw), ctypes.c_int(th), p(offs), p(flat), p(out), p(hist), p(lay), p(rm))
    valid = out[0, 2]
    print(f"valid pairs {valid / 1e6:.1f} M; processed tile entries {hist.sum() / 1e6:.2f} M "
          f"(with no valid pixel: {hist[0] / 1e6:.2f} M; median valid px {np.searchsorted(np.cumsum(hist) / hist.sum(), 0.5)})")
    print(f"{'unit':>7} {'cand(geo&depth)':>16} {'slots M':>9} {'util':>6} | {'ideal cand':>11} {'util':>6} | {'geo only cand':>13}")
    for (uw, uh), r in zip(SHAPES, out):
        px = uw * uh
        print(f"{uw:>4}x{uh:<2} {r[0] / 1e6:>14.2f} M {r[0] * px / 1e6:>9.1f} {valid / (r[0] * px):>6.3f} | {r[1] / 1e6:>9.2f} M {valid / (r[1] * px):>6.3f} | {r[3] / 1e6:>11.2f} M")
    names = ["wave 16x8, one queue (today)", "wave 16x8, two 8x8 queues", "wave 16x8, four 8x4 queues", "wave 16x8, four 16x2 strips",
             "wave 8x8 1px/lane, four 4x4 queues", "wave 16x8, eight 4x4 queues", "wave 16x16, four 8x8 queues", "wave 16x16, eight 8x4 queues"]
    slots = [128, 128, 128, 128, 64, 128, 256, 256]
    print("wave iterations (sum over waves of the longest unit queue) and slot utilisation:")
    for n, v, sl in zip(names, lay, slots):
        print(f"  {n:<38} {v / 1e6:6.2f} M iterations  util {valid / (v * sl):.3f}")
    print("composite_bwd4_kernel model (eight 8x4 units, rounds counted back from the tile's deepest last contributor):")
    for name, r in zip(("32", "64", "128", "whole list"), rm):
        print(f"  round length {name:>10}: {r[0] / 1e3:8.1f} k rounds, {r[1] / 1e6:6.3f} M iterations, {r[2] / 1e6:6.2f} M staged entries, util {valid / (r[1] * 256):.3f}")


if __name__ == "__main__":
    main()

