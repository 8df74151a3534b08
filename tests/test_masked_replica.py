"""Replicated-Gaussian mode with the visibility-masked exchange (`distributed.MaskedReplicaAdam`; SURVEY.md §5.8, §8e "secondary";
VERDICT r2 item 9): W CPU processes on gloo.

  * after every step the parameters equal the dense formulation — average of the ranks' gradients, visibility-masked Adam
    (oracle/adam_oracle.py, the update internal/optimizers.py:26-58 wraps) on the rows ANY rank saw — and the sharded moments,
    put together, equal its moments;
  * the replicas are BIT-identical after every step;
  * a densification performed identically on every rank — statistics all-reduced (`reduce_densification_stats`), rows selected from
    them, split samples drawn with a shared RNG seed (vanilla_density_controller.py:180-182), rows pruned — keeps the replicas
    bit-identical, and the moments follow their rows through `reshard`.
"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
SHAPES = ((3,), (4,), (1,), (1, 3), (15, 3))
LRS = (1e-2, 2e-2, 5e-2, 3e-3, 1e-3)


def _worker(rank, world, port, tmpdir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    for p in (HERE, os.path.dirname(HERE)):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gspl_amd  # noqa: F401
        from gspl_amd import distributed as D
        from oracle.adam_oracle import selective_adam_step
        N = 1003
        g0 = torch.Generator().manual_seed(11)
        init = [torch.randn((N,) + s, generator=g0) for s in SHAPES]
        params = [t.clone().requires_grad_(True) for t in init]
        opt = D.MaskedReplicaAdam([(str(i), p, lr) for i, (p, lr) in enumerate(zip(params, LRS))], eps=1e-15)
        # dense fp64 reference carried along on every rank
        ref_p = [t.double().clone() for t in init]
        ref_m = [torch.zeros_like(t) for t in ref_p]
        ref_v = [torch.zeros_like(t) for t in ref_p]

        def identical_everywhere(tensors, what):
            flat = torch.cat([t.detach().reshape(-1) for t in tensors])
            other = flat.clone()
            dist.broadcast(other, src=0)
            assert torch.equal(flat, other), f"replicas differ ({what})"

        def gathered_moments():
            sizes = [b[1] - b[0] for b in opt.bounds]
            return [D._all_to_all_rows_raw(t.repeat(world, 1), [t.shape[0]] * world, sizes, None) for t in (opt.exp_avg, opt.exp_avg_sq)]

        def one_step(step, n_rows):
            vis_all, grads_all = [], []
            for r in range(world):                                   # every rank can form every rank's inputs (seeded): the reference needs them
                g = torch.Generator().manual_seed(1000 * step + r)
                vis = torch.rand(n_rows, generator=g) < (0.15 + 0.2 * r)
                vis_all.append(vis)
                grads_all.append([torch.randn((n_rows,) + s, generator=g) * vis.reshape(-1, *([1] * len(s))) for s in SHAPES])
            for p, gr in zip(params, grads_all[rank]):
                p.grad = gr.clone()
            seen = opt.step(vis_all[rank])
            union = torch.stack(vis_all).any(0)
            assert seen == int(union.sum())
            for k in range(len(SHAPES)):
                avg = sum(grads_all[r][k].double() for r in range(world)) / world
                selective_adam_step(ref_p[k], avg, ref_m[k], ref_v[k], union, LRS[k], 0.9, 0.999, 1e-15)
                assert torch.allclose(params[k].detach().double(), ref_p[k], rtol=0, atol=2e-6), (step, k, float((params[k].detach().double() - ref_p[k]).abs().max()))
                assert torch.equal(params[k].detach()[~union], (init_now[k])[~union])          # rows nobody saw: untouched, bit for bit
            m, v = gathered_moments()
            assert torch.allclose(m.double(), torch.cat([t.reshape(n_rows, -1) for t in ref_m], dim=1), rtol=0, atol=1e-6)
            assert torch.allclose(v.double(), torch.cat([t.reshape(n_rows, -1) for t in ref_v], dim=1), rtol=0, atol=1e-6)
            identical_everywhere(params, f"step {step}")

        init_now = [p.detach().clone() for p in params]
        for step in range(3):
            one_step(step, N)
            init_now = [p.detach().clone() for p in params]

        # ---- a densification, identically on every rank
        g = torch.Generator().manual_seed(77 + rank)
        accum, denom, max_radii = torch.rand(N, generator=g), torch.randint(1, 4, (N,), generator=g).float(), torch.rand(N, generator=g) * 30
        D.reduce_densification_stats(accum, denom, max_radii)                        # SUM / SUM / MAX over the ranks
        identical_everywhere([accum, denom, max_radii], "densification statistics")
        score = accum / denom
        split = score > score.quantile(0.9)                                          # rows to split (replaced by two samples)
        prune = (max_radii > max_radii.quantile(0.97)) & ~split
        torch.manual_seed(4242)                                                      # the SHARED seed of the split sampling
        n_split = int(split.sum())
        noise = torch.randn(2 * n_split, 3)
        keep = ~(split | prune)
        new_params = []
        for k, p in enumerate(params):
            d = p.detach()
            extra = d[split].repeat(2, *([1] * (d.dim() - 1)))
            if k == 0:
                extra = extra + 0.01 * noise
            new_params.append(torch.cat([d[keep], extra]).clone().requires_grad_(True))
        opt.reshard(new_params, keep=keep, appended=2 * n_split)
        params[:] = new_params
        N2 = int(keep.sum()) + 2 * n_split
        assert params[0].shape[0] == N2 and opt.N == N2 and opt.exp_avg.shape[0] == opt.bounds[rank][1] - opt.bounds[rank][0]
        identical_everywhere(params, "after the densification")
        # the reference's moments follow their rows; appended rows start from zero
        for k in range(len(SHAPES)):
            pad = lambda t: torch.cat([t[keep], torch.zeros((2 * n_split,) + t.shape[1:], dtype=t.dtype)])
            ref_m[k], ref_v[k] = pad(ref_m[k]), pad(ref_v[k])
            ref_p[k] = params[k].detach().double().clone()
        m, v = gathered_moments()
        assert torch.allclose(m.double(), torch.cat([t.reshape(N2, -1) for t in ref_m], dim=1), rtol=0, atol=1e-6)
        init_now = [p.detach().clone() for p in params]
        for step in range(3, 5):
            one_step(step, N2)
            init_now = [p.detach().clone() for p in params]
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    except BaseException:
        dist.destroy_process_group()
        raise
    from conftest import leave_process_group
    leave_process_group(dist)


@pytest.mark.parametrize("world", [2, 3])
def test_masked_exchange_keeps_replicas_identical_and_matches_the_dense_formulation(tmp_path, world):
    from conftest import free_port
    mp.spawn(_worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
