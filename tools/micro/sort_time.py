"""Times the binning-stage radix sorts (include/gspl_hip.h §10) at the metric workload's sizes.
usage: python tools/micro/sort_time.py"""
import ctypes
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import gspl_amd  # noqa: F401
from gspl_amd import _lib as L

DEV = "cuda:0"


def time_call(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    lib = L.lib()
    rng = np.random.default_rng(0)
    n = 1_000_000
    d = rng.uniform(2.7, 5.3, size=n).astype(np.float32).view(np.int32)
    k_src = torch.from_numpy(d).to(DEV)
    k0, k1 = k_src.clone(), torch.empty_like(k_src)
    v0 = torch.arange(n, dtype=torch.int32, device=DEV)
    v1 = torch.empty_like(v0)
    ws_bytes = lib.gspl_radix_sort_workspace_bytes(n, 4, 0, 32)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=DEV)
    res = ctypes.c_int(0)

    def depth():
        k0.copy_(k_src)
        L.call("gspl_radix_sort_pairs_u32", n, L.ptr(k0), L.ptr(k1), L.ptr(v0), L.ptr(v1), 0, 32, ctypes.byref(res), L.ptr(ws), ws_bytes, L.stream())
    tc = time_call(lambda: k0.copy_(k_src))
    print(f"depth sort  {n} u32 pairs, 32 bits: {time_call(depth) - tc:8.1f} us   (torch.sort: {time_call(lambda: torch.sort(k_src)):8.1f} us)")

    ni = int(os.environ.get("SORT_NI", "13818945"))
    tile = torch.randint(0, 8160, (ni,), device=DEV, dtype=torch.int64)
    rec = (tile << 32) | torch.arange(ni, device=DEV, dtype=torch.int64)
    r0, r1 = rec.clone(), torch.empty_like(rec)
    ws2_bytes = lib.gspl_radix_sort_workspace_bytes(ni, 8, 32, 45)
    ws2 = torch.empty((ws2_bytes,), dtype=torch.uint8, device=DEV)

    def tiles():
        r0.copy_(rec)
        L.call("gspl_radix_sort_keys_u64", ni, L.ptr(r0), L.ptr(r1), 32, 45, ctypes.byref(res), L.ptr(ws2), ws2_bytes, L.stream())
    tc = time_call(lambda: r0.copy_(rec))
    print(f"tile sort   {ni} u64 keys, 13 bits: {time_call(tiles) - tc:8.1f} us")


if __name__ == "__main__":
    main()
