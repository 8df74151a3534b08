"""The RCCL code paths of gspl_amd.distributed on a machine with ONE GPU: a process group of a single rank on the "nccl"
backend (RCCL on ROCm), with the single-rank shortcuts switched off, so that every collective the two multi-GPU modes use
— all-gather of ints, all-to-all of counts, the differentiable all-to-all of rows with split sizes, the SUM / MAX / AVG
all-reduces, the chunked all-reduce overlapped with the fused Adam — is really issued to RCCL on device tensors.  With one
rank every one of them must be the identity, which is what is asserted; what the test buys is that the calls, dtypes, split
sizes and stream semantics are accepted by the backend the 8-GPU runs use (the W = 2 tests run on gloo)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import copy, os, sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.environ["GSPL_ROOT"])
    import gspl_amd  # noqa: F401
    from gspl_amd import distributed as D
    from gspl_amd.optimizers import FusedAdam

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert D.is_rccl()
    D.SINGLE_RANK_SHORTCUT = False
    g = torch.Generator().manual_seed(7)

    assert D.gather_ints(41, dev) == [41]
    assert D.exchange_counts([1234], dev) == [1234]

    # differentiable all-to-all of rows: forward and backward are both the identity route
    rows = torch.randn(1234, D.RECORD_FLOATS, generator=g).to(dev).requires_grad_(True)
    out = D.all_to_all_rows(rows, [1234], [1234])
    assert out.is_cuda and torch.equal(out.detach(), rows.detach())
    w = torch.randn(1234, D.RECORD_FLOATS, generator=g).to(dev)
    (out * w).sum().backward()
    assert torch.equal(rows.grad, w)
    empty = D.all_to_all_rows(rows[:0], [0], [0])                     # a rank that sees nothing
    assert empty.shape == (0, D.RECORD_FLOATS)

    # rebalancing route (rows + Adam rows travel through the same call)
    local = torch.randn(1000, 7, generator=g).to(dev)
    moved = D.redistribute_rows(local, torch.zeros(1000, dtype=torch.int64, device=dev))
    assert torch.equal(moved, local)

    # densification statistics: SUM, SUM, MAX
    a, b, c = (torch.rand(5000, generator=g).to(dev) for _ in range(3))
    a0, b0, c0 = a.clone(), b.clone(), c.clone()
    D.reduce_densification_stats(a, b, c)
    assert torch.equal(a, a0) and torch.equal(b, b0) and torch.equal(c, c0)

    # gradient all-reduce + step, plain and chunked/overlapped: same parameters and moments as a local step
    shapes = [(20000, 3), (20000, 4), (20000, 1), (20000, 15, 3)]
    def make():
        gg = torch.Generator().manual_seed(11)
        ps = [torch.randn(*s, generator=gg).to(dev).requires_grad_(True) for s in shapes]
        for p in ps:
            p.grad = torch.randn(*p.shape, generator=gg).to(dev)
        return ps, FusedAdam([{"params": [p], "lr": 1e-2} for p in ps], eps=1e-15)
    ref_p, ref_o = make()
    for _ in range(3):
        ref_o.step()
    p1, o1 = make()
    for _ in range(3):
        D.all_reduce_gradients(p1)
        o1.step()
    p2, o2 = make()
    for _ in range(3):
        D.all_reduce_and_step(o2, p2, chunk_bytes=64 << 10)            # several chunks per tensor
    torch.cuda.synchronize()
    for r, x, y in zip(ref_p, p1, p2):
        assert torch.equal(r, x), "all_reduce_gradients + step"
        assert torch.equal(r, y), "all_reduce_and_step"
    for r, y in zip(ref_p, p2):
        assert torch.equal(ref_o.state[r]["exp_avg"], o2.state[y]["exp_avg"])
        assert torch.equal(ref_o.state[r]["exp_avg_sq"], o2.state[y]["exp_avg_sq"])

    # the sharded renderer with its exchange forced through RCCL == the same renderer without a process group's exchange
    from gspl_amd import synthetic
    from gspl_amd.renderers import HipGSplatDistributedRenderer
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
        return out["render"].detach(), [t.grad for t in model.leaves()]
    img_a, grads_a = render(True)
    img_b, grads_b = render(False)
    assert torch.equal(img_a, img_b)
    for k, (ga, gb) in enumerate(zip(grads_a, grads_b)):
        # The compositing backward accumulates with fp32 atomics: the order of the additions differs between two launches, most of all
        # between a process's FIRST launch of the kernel (code being loaded, workgroups start one after the other) and the later ones.
        # Measured over fresh processes (tools/diag/r03rccl2.py): opacities / SH coefficients agree to 8e-8 of the tensor's maximum,
        # means / scales / rotations — behind the conic -> cov2D -> cov3D chain, which amplifies a last-bit difference of one
        # near-degenerate splat — to 1e-6 / 5e-6 / 1.1e-5 for the first frame and 3e-7 between later frames.
        tol = 1e-4 if k < 3 else 1e-5                     # leaves: means, scales, rotations | opacities, shs_dc, shs_rest
        assert float((ga - gb).abs().max()) <= tol * float(ga.abs().max()) + 1e-12, (k, float((ga - gb).abs().max()), float(ga.abs().max()))
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL-SINGLE-RANK-OK")
""")


@pytest.mark.gpu
def test_collectives_of_both_modes_run_on_rccl_with_one_rank(tmp_path):
    from conftest import free_port
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, GSPL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-SINGLE-RANK-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
