"""usage (on the GPU box): python tools/micro/lone_walk_probe.py
What does a LONE wave's walk of a long list cost per entry in the compositing forward (the tail of a heavy-tailed frame is four such waves)?
One 16 x 16 image = one tile = four quadrant waves, a list of L entries, launch duration from HIP events (median of 20):
  miss     every splat lies far outside the tile (the quadrant test rejects it): the cost of a round of 64 tests
  hit      every splat covers the tile with alpha 0.004 (above 1/255; T stays above 1e-4 for 2 300 entries): a round of 64 candidates
  third    one entry in three is such a candidate
  scattered / sorted: splat ids in random order (the gathers of a round touch 64 different lines) or ascending (neighbouring records)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
import gspl_amd  # noqa: F401
from gspl_amd import _lib as L
from hip_helpers import hip_composite_fwd

dev = torch.device("cuda:0")
W = H = 16
N = 200_000


def case(kind, Ln, scattered):
    g = torch.Generator().manual_seed(1)
    means = torch.full((N, 2), 8.0)
    conics = torch.tensor([0.002, 0.0, 0.002]).repeat(N, 1)       # sigma ~ 22 px: alpha nearly flat over the tile
    opac = torch.full((N,), 0.004)
    far = torch.zeros(N, dtype=torch.bool)
    if kind == "miss":
        far[:] = True
    elif kind == "third":
        far[torch.arange(N) % 3 != 0] = True
    means[far] = 400.0
    conics[far] = torch.tensor([2.0, 0.0, 2.0])
    colors = torch.rand(N, 3, generator=g)
    ids = (torch.randperm(N, generator=g)[:Ln] if scattered else torch.arange(Ln)).to(torch.int32)
    offsets = torch.tensor([0], dtype=torch.int32)
    a = [t.contiguous().to(dev) for t in (means, conics, colors, opac)]
    return a, offsets.to(dev), ids.to(dev)


bg = torch.zeros(3, device=dev)
for kind in ("miss", "hit", "third"):
    for scattered in (False, True):
        row = []
        for Ln in (1024, 2048):
            (means, conics, colors, opac), offsets, ids = case(kind, Ln, scattered)
            ts = []
            for it in range(25):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                out = hip_composite_fwd(L.GSPL_MODE_INRIA, means, conics, colors, opac, bg, W, H, offsets, ids, layout=L.GSPL_LAYOUT_CHW)
                b.record(); torch.cuda.synchronize()
                if it >= 5:
                    ts.append(a.elapsed_time(b) * 1e3)
            ts.sort()
            row.append(ts[len(ts) // 2])
            last = int(out[3].max())
        per = (row[1] - row[0]) / 1024 * 1e3
        print(f"{kind:6s} {'scattered' if scattered else 'sorted   '}: L=1024 {row[0]:7.1f} us  L=2048 {row[1]:7.1f} us  -> {per:6.1f} ns per entry, {per * 64 / 1e3:5.2f} us per round of 64 (deepest blended entry {last})")
