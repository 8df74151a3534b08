"""Lane-utilisation statistics of the compositing backward (instrumentation build: -DGSPL_COUNT_PAIRS)."""
import ctypes, sys
sys.path.insert(0, "/root/repo")
import torch
import gspl_amd
from gspl_amd import _lib as L, ops, synthetic
import bench
wl = synthetic.WORKLOADS["S-1080p-1M"]
dev = torch.device("cuda:0")
means, scales, quats, opac, shs = synthetic.scene(wl["n"], seed=42)
cam = synthetic.camera(wl["width"], wl["height"], wl["fx"])
tensors = [t.to(dev).requires_grad_(True) for t in (means, scales, quats, opac, shs)]
step = bench.make_step("vanilla", dev, wl, cam, tensors, "l1")
step(); torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so")
lib = L.lib()
sym = ctypes.c_void_p(); size = ctypes.c_size_t()
# the counters are a __device__ array: read through hipMemcpyFromSymbol is not available from ctypes without the fat binary
# handle, so the instrumentation build exports an accessor instead
lib.gspl_debug_pair_stats.restype = ctypes.c_int
out = (ctypes.c_ulonglong * 8)()
lib.gspl_debug_pair_stats(out, 1)
step(); torch.cuda.synchronize()
lib.gspl_debug_pair_stats(out, 0)
cand, valid, anyv, both = out[0], out[1], out[2], out[3]
print(f"half-tile candidates {cand}, with a valid pixel {anyv} ({anyv / cand:.3f}), touching both quadrants {both} ({both / max(anyv, 1):.3f} of those), "
      f"valid (pixel, splat) pairs {valid}, pixel utilisation among processed candidates {valid / (128.0 * anyv):.3f}")
rounds, fcand, fiter = out[4], out[5], out[6]
print(f"forward: {rounds} staging rounds of 64 entries (per quadrant wave), {fcand} candidates ({fcand / max(rounds, 1):.1f} per round), "
      f"{fiter} two-candidate iterations")
