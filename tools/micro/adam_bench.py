"""usage (on the GPU box): GSPL_HIP_LIB=<variant .so> python tools/micro/adam_bench.py [N ...]
Launch duration of the fused Adam update over the reference's six parameter groups (3, 3, 4, 1, 3, 45 floats per Gaussian), HIP events
around 20 launches after 5 warm-up launches, all rows visible.  Prints µs per launch and the rate over 28 B per element."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import gspl_amd
from gspl_amd import optimizers as gopt

for N in [int(a) for a in sys.argv[1:]] or [1_000_000, 6_000_000]:
    rows = [3, 3, 4, 1, 3, 45]
    params = [torch.nn.Parameter(torch.randn(N, r, device="cuda")) for r in rows]
    opt = gopt.FusedAdam([{"params": [p], "lr": 1e-3, "name": f"g{k}"} for k, p in enumerate(params)], eps=1e-15)
    grads = [torch.randn_like(p) for p in params]
    def step():
        for p, g in zip(params, grads):
            p.grad = g
        opt.step()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); step(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"N {N}: median {med:.1f} us  min {ts[0]:.1f}  -> {N * sum(rows) * 28 / med / 1e6:.2f} TB/s")
    del params, opt, grads
    torch.cuda.empty_cache()
