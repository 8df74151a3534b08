"""distributed.PeerExchange / csrc/peer.hip: the direct peer-write transport of the sharded renderer's records.
One process (its own and only peer) exercises the three launches — put, signal, wait —, the two receive areas used alternately,
buffer growth and the give-up path of the wait; two processes on one GPU (IPC mapping of each other's buffers) run in
tests/test_distributed_renderer.py::test_world2_sharded_renderer_shared_gpu[padded-peer]."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_single_rank_round_trips_and_growth():
    import gspl_amd  # noqa: F401
    from gspl_amd import distributed as D
    dev = torch.device("cuda:0")
    px = D.PeerExchange(0, None, dev)
    g = torch.Generator().manual_seed(0)
    last = None
    for n in (1000, 1000, 1000, 777, 5000, 0, 12):          # 5000 outgrows the first allocation (1.5 x 1000 + 1024 rows)
        capacity_before = px.cap_total
        fwd, bwd = px.route([n])
        rows = torch.randn(n, 12, generator=g).to(dev)
        got = fwd(rows)
        assert got.shape == (n, 12) and torch.equal(got, rows)
        if last is not None and last[0].shape[0] and n and px.cap_total == capacity_before:
            assert torch.equal(last[0], last[1])             # the previous step's rows are still intact (the other receive area)
        if n == 5000:
            assert px.cap_total > capacity_before      # (the buffers were re-allocated — possibly at the same address; views of the old ones are dead)
        v = torch.randn(n, 12, generator=g).to(dev)
        back = bwd(v)
        assert back.shape == (n, 12) and torch.equal(back, v)
        last = (got, rows.clone())
    px.check()
    assert px.step == 7 and px.cap_total >= 5000
    px.close()


def test_wait_gives_up_and_reports_the_missing_source():
    """A flag that never arrives must not hang the GPU: the wait kernel stops after max_polls polls and raises the error word."""
    import gspl_amd  # noqa: F401
    from gspl_amd import _lib as L
    dev = torch.device("cuda:0")
    flags = torch.zeros(2, dtype=torch.int64, device=dev)
    flags[0] = 5
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        L.call("gspl_peer_wait", L.ptr(flags), 2, 5, 2000, L.ptr(err), L.stream())
    torch.cuda.synchronize()
    assert int(err.item()) == 2          # 1 + the source whose flag stayed below 5
    err.zero_()
    flags[1] = 9
    with torch.cuda.device(dev):
        L.call("gspl_peer_wait", L.ptr(flags), 2, 5, 2000, L.ptr(err), L.stream())
    torch.cuda.synchronize()
    assert int(err.item()) == 0


def test_single_rank_counted_matrix():
    """The counted format: fewer rows than the shard holds, as the count matrix says."""
    import gspl_amd  # noqa: F401
    from gspl_amd import distributed as D
    dev = torch.device("cuda:0")
    px = D.PeerExchange(0, None, dev)
    g = torch.Generator().manual_seed(1)
    for n, c in ((1000, 640), (1000, 0), (1000, 1000)):
        fwd, bwd = px.route([n], [[c]])
        rows = torch.randn(c, 12, generator=g).to(dev)
        got = fwd(rows)
        assert got.shape == (c, 12) and torch.equal(got, rows)
        v = torch.randn(c, 12, generator=g).to(dev)
        assert torch.equal(bwd(v), v)
    with pytest.raises(ValueError):
        px.route([10], [[11]])
    px.check()
    px.close()


def _failing_setup_worker(rank, world, port, tmpdir, fail_at):
    import os
    import sys
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gspl_amd  # noqa: F401
        from gspl_amd import distributed as D, _lib as L
        dev = torch.device("cuda:0")
        if rank == 1:          # this rank's allocation / IPC mapping is refused, the other rank's works
            lib = L.lib()
            real = getattr(lib, fail_at)
            setattr(lib, fail_at, lambda *a: 1)
        px = D.PeerExchange(rank, None, dev)
        try:
            px.route([100, 100])
            outcome = "no error"
        except RuntimeError as e:
            outcome = str(e)
        # the set-up failed on BOTH ranks, with the failing rank named, and nobody is left inside a collective: this one pairs up
        t = torch.tensor([rank + 1])
        dist.all_reduce(t)
        assert int(t) == 3
        assert "rank 1" in outcome and "failed" in outcome, outcome
        assert px._mine is None and px._opened == []
        if rank == 1:
            setattr(lib, fail_at, real)
        # ... and the same object sets up fine afterwards
        fwd, _ = px.route([100, 100])
        rows = torch.full((200, D.RECORD_FLOATS), float(rank + 1), device=dev)
        got = fwd(rows)
        torch.cuda.synchronize()
        px.check()
        assert got.shape == (200, D.RECORD_FLOATS) and torch.equal(got[:100], torch.full_like(got[:100], 1.0)) and torch.equal(got[100:], torch.full_like(got[100:], 2.0))
        dist.barrier()
        px.close()
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    except BaseException:
        dist.destroy_process_group()
        raise
    from conftest import leave_process_group
    leave_process_group(dist)


@pytest.mark.gpu
@pytest.mark.parametrize("fail_at", ["gspl_peer_alloc", "gspl_peer_open"])
def test_a_set_up_failure_on_one_rank_is_raised_on_every_rank(tmp_path, fail_at):
    """PeerExchange set-up is collective (handles all-gathered, a barrier before the first put): a rank whose allocation or IPC
    mapping is refused must not leave the others waiting in it — every rank raises, naming the rank that failed, and the caller
    (bench.py's `auto` transport) falls back to the collective route on all ranks together.  Two processes on one GPU."""
    import torch.multiprocessing as mp
    from conftest import free_port
    mp.spawn(_failing_setup_worker, args=(2, free_port(), str(tmp_path), fail_at), nprocs=2, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(2))


def _lost_peer_worker(rank, world, port, tmpdir, dying=1):
    import os
    import sys
    import time
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    # a fraction of a second instead of the collective-scale default (eight processes time-share one GPU's queues: a few seconds)
    polls = 300000 if world <= 2 else 3000000
    os.environ["GSPL_PEER_MAX_POLLS"] = str(polls)
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gspl_amd  # noqa: F401
    from gspl_amd import distributed as D
    dev = torch.device("cuda:0")
    px = D.PeerExchange(rank, None, dev)
    assert px.MAX_POLLS == polls
    rows = torch.full((100 * world, D.RECORD_FLOATS), float(rank + 1), device=dev)
    fwd, _ = px.route([100] * world)           # step 1: every rank takes part
    got = fwd(rows)
    torch.cuda.synchronize()
    for s_ in range(world):
        assert torch.equal(got[100 * s_:100 * (s_ + 1)], torch.full_like(got[:100], float(s_ + 1)))
    dist.barrier()
    if rank == dying:                          # ... then one rank is gone (no signal, no clean-up, no collective)
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("gone")
        os._exit(0)
    fwd, _ = px.route([100] * world)           # step 2 on the survivors: their wait for the lost rank's rows can only give up
    t0 = time.monotonic()
    got = fwd(rows)
    torch.cuda.synchronize()                   # the stream DRAINS: the wait kernel is bounded, the GPU is not left spinning
    waited = time.monotonic() - t0
    assert waited < 60.0, waited
    for s_ in range(world):                    # the rows of the ranks that ARE there arrived (the other receive area of the double buffer)
        if s_ != dying:
            assert torch.equal(got[100 * s_:100 * (s_ + 1)], torch.full_like(got[:100], float(s_ + 1)))
    try:                                       # ... and the very next use raises, naming the rank that never signalled
        px.route([100] * world)
        outcome = "no error"
    except RuntimeError as e:
        outcome = str(e)
    assert f"gave up waiting for the records of rank {dying}" in outcome, outcome
    assert int(px._err_host[0]) == 0           # cleared by the raise: the object can be torn down (or used with a new peer set)
    open(os.path.join(tmpdir, f"ok{rank}"), "w").write(f"{waited:.3f}")
    os._exit(0)                                # (no barrier with a dead peer: leave without the collective tear-down)


@pytest.mark.gpu
@pytest.mark.parametrize("world,dying", [(2, 1), (8, 5)])
def test_a_rank_that_dies_mid_run_makes_the_survivor_raise_within_the_poll_budget(tmp_path, world, dying):
    """ADVICE r4 / VERDICT r4 #7: the peer transport's device-side wait is bounded (GSPL_PEER_MAX_POLLS) and its error word lives in
    pinned host memory that `route()` reads before every step — a peer that is lost mid-run turns into a RuntimeError on the next step
    of the survivor (a collective would block for the process group's timeout), never into a hang or a silent run over stale rows.
    Two processes on one GPU; and eight with rank 5 dying (VERDICT r5 #1d): seven survivors each give up on exactly that rank."""
    import torch.multiprocessing as mp
    from conftest import free_port
    ctx = mp.spawn(_lost_peer_worker, args=(world, free_port(), str(tmp_path), dying), nprocs=world, join=False)
    import time
    deadline = time.monotonic() + 240
    while time.monotonic() < deadline and not all((tmp_path / f"ok{r}").exists() for r in range(world)):
        time.sleep(0.2)
    for p in ctx.processes:
        p.join(timeout=10)
        if p.is_alive():
            p.kill()
    missing = [r for r in range(world) if not (tmp_path / f"ok{r}").exists()]
    assert not missing, f"ranks {missing} never finished (hung in the wait?)"
