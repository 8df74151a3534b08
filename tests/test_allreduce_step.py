"""Replicated-Gaussian mode (bench.py --parallelism replicated, the reference's configs/ddp.yaml shape): the chunked gradient
all-reduce overlapped with the chunk-wise fused Adam (`distributed.all_reduce_and_step`) gives the same parameters and moments
as the plain sequence `all_reduce_gradients` + `optimizer.step()`.  Two processes on one shared GPU, gloo (host staged)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for p in (HERE, os.path.dirname(HERE)):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gspl_amd  # noqa: F401
        from gspl_amd import distributed as D
        from gspl_amd import optimizers as gopt
        dev = torch.device("cuda:0")
        N = 10007                                            # not a multiple of the chunk rows
        shapes = [(N, 3), (N, 4), (N, 1), (N, 1, 3), (N, 15, 3)]
        g0 = torch.Generator().manual_seed(5)
        init = [torch.randn(s, generator=g0) for s in shapes]

        def run(chunked):
            params = [t.clone().to(dev).requires_grad_(True) for t in init]
            opt = gopt.FusedAdam([{"params": [p], "lr": 1e-2 * (i + 1), "name": str(i)} for i, p in enumerate(params)], eps=1e-15)
            for step in range(3):
                g = torch.Generator().manual_seed(100 * step + rank)          # every rank its own gradients
                for p in params:
                    p.grad = torch.randn(p.shape, generator=g).to(dev)
                if chunked:
                    D.all_reduce_and_step(opt, params, chunk_bytes=64 << 10)   # many chunks, also inside one tensor
                else:
                    D.all_reduce_gradients(params)
                    opt.step()
            torch.cuda.synchronize()
            return params, opt

        ref_p, ref_o = run(False)
        got_p, got_o = run(True)
        for a, b in zip(ref_p, got_p):
            assert torch.allclose(a, b, rtol=0, atol=2e-7), float((a - b).abs().max())
            sa, sb = ref_o.state[a], got_o.state[b]
            assert sa["step"] == sb["step"] == 3
            assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=0, atol=1e-7) and torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=0, atol=1e-7)
        # replicas are identical after the step
        flat = torch.cat([p.detach().reshape(-1) for p in got_p]).cpu()
        other = flat.clone()
        dist.broadcast(other, src=0)
        assert torch.equal(flat, other)
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    except BaseException:
        dist.destroy_process_group()
        raise
    from conftest import leave_process_group
    leave_process_group(dist)


@pytest.mark.gpu
def test_chunked_all_reduce_with_chunkwise_adam_matches_the_plain_sequence(tmp_path):
    from conftest import free_port
    port = free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
