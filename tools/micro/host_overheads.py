"""Host-side cost of the calls between the count read-back and the emit launch (the GPU idles while they run)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import gspl_amd
from gspl_amd import _lib as L
lib = L.lib()
N, n = 1_000_000, 6_500_000
def t(fn, k=200):
    fn(); t0 = time.perf_counter()
    for _ in range(k): fn()
    return (time.perf_counter() - t0) / k * 1e6
print("gspl_bin_workspace_bytes: %.1f us" % t(lambda: lib.gspl_bin_workspace_bytes(N, n)))
print("torch.empty(6.5M i32):    %.1f us" % t(lambda: torch.empty((n,), dtype=torch.int32, device="cuda")))
ev = torch.cuda.Event()
def sync():
    ev.record(); ev.synchronize()
print("event record+synchronize (idle stream): %.1f us" % t(sync))
pinned = torch.empty((1,), dtype=torch.int64).pin_memory()
src = torch.ones((1,), dtype=torch.int64, device="cuda")
def rb():
    pinned.copy_(src, non_blocking=True); ev.record(); ev.synchronize(); return int(pinned[0])
print("async 8-byte read-back + event sync: %.1f us" % t(rb))
print(".item(): %.1f us" % t(lambda: src.item()))
