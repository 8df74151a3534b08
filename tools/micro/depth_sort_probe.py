"""The depth sort alone (gspl_radix_sort_pairs_u32 over 32 bits) on depth-like keys, a few repetitions: run it under
`rocprofv3 --kernel-trace --stats` or `--pmc ...` to look at the count / scatter kernels without the rest of a frame.
usage: SORT_N=6000000 python tools/micro/depth_sort_probe.py"""
import ctypes
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import gspl_amd  # noqa: F401
from gspl_amd import _lib as L


def main():
    lib = L.lib()
    n = int(os.environ.get("SORT_N", "6000000"))
    reps = int(os.environ.get("SORT_REPS", "8"))
    rng = np.random.default_rng(0)
    d = rng.uniform(2.7, 5.3, size=n).astype(np.float32).view(np.int32)
    k_src = torch.from_numpy(d).to("cuda:0")
    k0, k1 = k_src.clone(), torch.empty_like(k_src)
    v0 = torch.arange(n, dtype=torch.int32, device="cuda:0")
    v1 = torch.empty_like(v0)
    ws_bytes = lib.gspl_radix_sort_workspace_bytes(n, 4, 0, 32)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device="cuda:0")
    res = ctypes.c_int(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(reps):
        k0.copy_(k_src)
        e0.record()
        L.call("gspl_radix_sort_pairs_u32", n, L.ptr(k0), L.ptr(k1), L.ptr(v0), L.ptr(v1), 0, 32, ctypes.byref(res), L.ptr(ws), ws_bytes, L.stream())
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    out = (k0, k1)[res.value]
    assert bool((out[1:] >= out[:-1]).all())
    print(f"depth sort of {n} pairs: {tot / reps * 1e3:.1f} us per sort (events around the call)")


if __name__ == "__main__":
    main()
