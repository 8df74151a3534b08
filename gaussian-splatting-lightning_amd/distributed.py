"""
distributed.py — multi-GPU plumbing of the rasterizer path (one process per GPU, `torch.distributed`;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests and in the shared-GPU debugging mode).

Two modes:

1. Gaussian-sharded (the reference's `configs/distributed.yaml`,
   internal/renderers/gsplat_distributed_renderer.py:132-211): every rank projects its shard for all W cameras
   and the visible splats travel to the rank that renders that camera.  Where the reference sends two
   messages per peer (a float [n,11] and an int [n] tensor, :195-202), this sends ONE packed 48-byte record
   per splat — xy(2) depth(1) conic(3) compensation(1) opacity(1) rgb(3) radius(1, int32 bits) — through a single
   autograd-aware all-to-all with split sizes; the backward pass is the reverse all-to-all of the same
   records' gradients.  On the MI355X full mesh each peer message rides its own xGMI link.
2. Replicated Gaussians, cameras sharded (BASELINE.json north_star wording; `configs/ddp.yaml` in the reference):
   parameter gradients are all-reduced every step (`all_reduce_gradients`), the densification statistics when a
   densification consumes them (`reduce_densification_stats`).

Backends: with RCCL the collectives take device tensors.  Any other backend (gloo: no GPU all-to-all) gets the
same calls with the payload staged through host memory — slow, but it lets two processes share one GPU for tests.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import os

import torch
import torch.distributed as dist

RECORD_FLOATS = 12     # 48 B per visible splat
# A group of one rank needs no exchange and the reductions below return at once.  False makes them issue their collectives
# anyway: how tests/test_rccl_single_rank.py and `bench.py --init-dist` drive the RCCL code paths on a one-GPU machine.
SINGLE_RANK_SHORTCUT = True


def _nothing_to_exchange(group) -> bool:
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return SINGLE_RANK_SHORTCUT and dist.get_world_size(group) == 1


def shard_bounds(n_gaussians: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block owned by `rank` (gsplat_distributed_renderer.py:76-83: round(N/W) rows, last rank takes the rest)."""
    per = round(n_gaussians / world_size)
    lo = per * rank
    hi = n_gaussians if rank + 1 == world_size else lo + per
    return lo, hi


def is_rccl(group=None) -> bool:
    return dist.get_backend(group) == "nccl"


def _staged(t: torch.Tensor, group) -> bool:
    """Does `t` have to travel through host memory for this group's backend?"""
    return t.is_cuda and not is_rccl(group)


def _wire(t: torch.Tensor, group) -> torch.Tensor:
    return t.cpu() if _staged(t, group) else t


def gather_ints(value: int, device, group=None) -> List[int]:
    """One integer per rank, in rank order (camera ids, Gaussian counts: gsplat_distributed_renderer.py:319-323,426)."""
    world = dist.get_world_size(group)
    wire_dev = device if (is_rccl(group) or torch.device(device).type == "cpu") else "cpu"
    mine = torch.tensor([int(value)], dtype=torch.int64, device=wire_dev)
    out = torch.empty((world,), dtype=torch.int64, device=wire_dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    return [int(v) for v in out.tolist()]


def gather_int_rows(values: Sequence[int], device, group=None) -> List[List[int]]:
    """A short row of integers per rank, in rank order: ONE all-gather for everything a step has to agree on before it starts
    (camera id, local Gaussian count, the rank's vote on the exchange format)."""
    world = dist.get_world_size(group)
    wire_dev = device if (is_rccl(group) or torch.device(device).type == "cpu") else "cpu"
    mine = torch.tensor([int(v) for v in values], dtype=torch.int64, device=wire_dev)
    out = torch.empty((world * mine.numel(),), dtype=torch.int64, device=wire_dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    return [[int(v) for v in row] for row in out.reshape(world, -1).tolist()]


def exchange_counts(send_counts: Sequence[int], device, group=None) -> List[int]:
    """send_counts[j] = rows this rank sends to rank j -> rows it receives from each rank."""
    wire_dev = device if (is_rccl(group) or torch.device(device).type == "cpu") else "cpu"
    send = torch.tensor([int(c) for c in send_counts], dtype=torch.int64, device=wire_dev)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    return [int(v) for v in recv.tolist()]


def _all_to_all_rows_raw(send: torch.Tensor, send_counts: List[int], recv_counts: List[int], group) -> torch.Tensor:
    send = send.contiguous()
    wire = _wire(send, group)
    out = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=wire.device)
    dist.all_to_all_single(out, wire, output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group)
    return out.to(send.device) if out.device != send.device else out


class _AllToAllRows(torch.autograd.Function):
    """Rows grouped by destination rank -> rows grouped by source rank; the backward is the reverse route."""

    @staticmethod
    def forward(ctx, send, send_counts, recv_counts, group):
        ctx.route = (list(send_counts), list(recv_counts), group)
        return _all_to_all_rows_raw(send, list(send_counts), list(recv_counts), group)

    @staticmethod
    def backward(ctx, v_out):
        send_counts, recv_counts, group = ctx.route
        return _all_to_all_rows_raw(v_out, recv_counts, send_counts, group), None, None, None


def all_to_all_rows(send: torch.Tensor, send_counts: Sequence[int], recv_counts: Sequence[int], group=None) -> torch.Tensor:
    """Differentiable variable-size all-to-all of the rows of `send` ([sum(send_counts), ...], grouped by destination)."""
    return _AllToAllRows.apply(send, list(send_counts), list(recv_counts), group)


def all_to_all_route(send_counts: Sequence[int], recv_counts: Sequence[int], group=None):
    """(forward, backward) callables of the same exchange for `ops.sharded_exchange`: rows grouped by destination -> rows grouped
    by source, and the gradients of the received rows back to the rows that were sent."""
    send_counts, recv_counts = list(send_counts), list(recv_counts)
    return (lambda rows: _all_to_all_rows_raw(rows, send_counts, recv_counts, group),
            lambda v_rows: _all_to_all_rows_raw(v_rows, recv_counts, send_counts, group))


_TRACKER_LOCK = __import__("threading").Lock()


class HostMailbox:
    """What the ranks of ONE node have to agree on before a step — camera id, local Gaussian count, vote on the exchange format:
    a short row of integers per rank — through POSIX shared memory instead of a collective: every rank stores its row and a sequence
    number into its slot and polls the other slots.  Host to host, a few microseconds, no stream, no kernel, no device
    synchronisation (the all-gather this replaces, `gather_int_rows`, is an RCCL launch, a copy back and a stream wait: 0.1-0.2 ms of
    host time per step on the critical path — profiles/r05d_*).  Created once (the name travels through one collective)."""

    def __init__(self, rank: int, group, width: int = 4):
        import numpy as np
        from multiprocessing import shared_memory
        self.rank, self.group = int(rank), group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.width, self.seq = int(width), 0
        # the collectives' patience, not a few seconds: a rank that writes a checkpoint or logs images keeps the others waiting here
        self.timeout_s = float(os.environ.get("GSPL_MAILBOX_TIMEOUT_S", "1800"))
        # two copies of the table, used alternately: a rank can be one call ahead of the slowest reader of the previous call (never
        # two: the call after the next needs everybody's rows of the next), so its new row must not land where that reader looks
        nbytes = 8 * 2 * self.world * (self.width + 1)
        names = [None]
        if self.rank == 0:
            self._shm = shared_memory.SharedMemory(create=True, size=nbytes)
            self._shm.buf[:nbytes] = bytes(nbytes)
            names = [self._shm.name]
        if self.world > 1:
            dist.broadcast_object_list(names, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        attach_error = None
        if self.rank != 0:
            # Python < 3.13 registers an ATTACHED segment with the resource tracker too (no `track=False` yet).  The segment belongs to
            # rank 0: an attaching rank must leave no trace there — registering and unregistering again is not the same thing when the
            # ranks share one tracker (multiprocessing children do): the name is one set entry, and rank 0's unlink would then
            # unregister a name that is already gone (a KeyError traceback from the tracker at exit).  The tracker's `register` is
            # swapped out under a process-wide lock, for the attach alone (another thread creating a segment of its own at
            # that moment waits for the lock instead of going untracked).
            import sys
            try:
                if sys.version_info >= (3, 13):
                    self._shm = shared_memory.SharedMemory(name=names[0], track=False)
                else:
                    from multiprocessing import resource_tracker
                    with _TRACKER_LOCK:
                        register = resource_tracker.register
                        resource_tracker.register = lambda *a, **k: None
                        try:
                            self._shm = shared_memory.SharedMemory(name=names[0])
                        finally:
                            resource_tracker.register = register
            except Exception as e:      # noqa: BLE001 — a rank on another node (or in another IPC namespace) cannot see the segment
                attach_error = f"rank {self.rank}: {e!r}"
        if self.world > 1:
            # ONE node (one IPC namespace), verified collectively BY THE ATTACH ITSELF (host names lie in both directions —
            # containers of one node differ, nodes of a cluster may share one): every rank says whether it sees rank 0's segment, and a
            # failure anywhere is raised everywhere — nobody is left in the barrier below
            verdicts = [None] * self.world
            dist.all_gather_object(verdicts, attach_error, group=group)
            failed = [v for v in verdicts if v is not None]
            if failed:
                if getattr(self, "_shm", None) is not None:
                    self._shm.close()
                    if self.rank == 0:
                        self._shm.unlink()
                    self._shm = None
                raise RuntimeError("HostMailbox: shared host memory (and the peer transport that uses it) is a one-node transport — "
                                   f"{len(failed)} rank(s) cannot attach rank 0's segment ({'; '.join(failed)}); use exchange_transport='collective'")
        self._rows = np.ndarray((2, self.world, self.width + 1), dtype=np.int64, buffer=self._shm.buf)      # [..., -1] = sequence number
        if self.world > 1:
            dist.barrier(group=group)
        # every rank has the segment mapped: its NAME can go now (the memory lives as long as a mapping does), so that no way of ending
        # a process — an exception, a kill, a multiprocessing child that skips its exit handlers — leaves a segment behind in /dev/shm
        if self.rank == 0:
            try:
                self._shm.unlink()
            except FileNotFoundError:
                pass
        # ... and nobody leaves the constructor before it is gone: "the name no longer exists" holds on every rank from here on
        if self.world > 1:
            dist.barrier(group=group)

    def exchange(self, values: Sequence[int], timeout_s: Optional[float] = None) -> List[List[int]]:
        """My row in, everybody's rows out (rank order); blocks until every rank has posted its row of this call (at most `timeout_s`
        seconds, default `self.timeout_s` = GSPL_MAILBOX_TIMEOUT_S or 1800: the scale of a process group's timeout)."""
        import time
        timeout_s = self.timeout_s if timeout_s is None else timeout_s
        self.seq += 1
        table = self._rows[self.seq & 1]
        table[self.rank, :self.width] = [int(v) for v in values]
        table[self.rank, self.width] = self.seq          # (x86: stores become visible in program order)
        deadline = None
        while not bool((table[:, self.width] == self.seq).all()):
            if deadline is None:
                deadline = time.monotonic() + timeout_s
            elif time.monotonic() > deadline:
                raise RuntimeError(f"HostMailbox: rank {self.rank} waited {timeout_s} s for the rows of call {self.seq} "
                                   f"(have {table[:, self.width].tolist()})")
        return [[int(v) for v in table[r, :self.width]] for r in range(self.world)]

    def close(self):
        shm, self._shm = getattr(self, "_shm", None), None
        if shm is not None:
            self._rows = None
            shm.close()          # (the name was unlinked as soon as every rank had attached)


class _DevicePtr:
    """A region of raw device memory as something `torch.as_tensor` understands (CUDA array interface)."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class PeerExchange:
    """The record exchange of the Gaussian-sharded renderer (fixed-size "padded" format, or "counted" with the count matrix the ranks
    post through a `HostMailbox`) WITHOUT a collective: every rank writes its rows
    straight into the receive buffer of the rank that renders that camera — device memory of the peer PROCESS mapped once through HIP
    IPC (csrc/peer.hip: fine-grained memory, reached over xGMI or on the same GPU) — and raises a flag word per destination; the
    receiver's stream waits for the flag words of its sources.  Three small launches per direction, nothing for the host to wait
    for, no count exchange.  Replaces the per-step `all_to_all` of gsplat_distributed_renderer.py:141-202 (`all_to_all_route` above
    stays as the route of the gloo / CPU tests and of the variable-size "counted" format).

    Every rank derives the SAME layout from the same `rows_per_rank` (the local Gaussian counts, which the per-step all-gather of the
    camera ids carries).  One allocation per rank:
        [flags fwd: W x u64][flags bwd: W x u64][error: i32] | 2 x forward receive (rows of every source for MY camera, by source)
        | 2 x backward receive (gradients of the rows I sent, by destination)
    Two copies of each receive area, used alternately (step parity): a source may run one exchange ahead of the slowest reader of
    the previous one, never two (its next forward needs every peer's backward rows of this step).  Training steps only — a
    forward without backward has no such bound and takes the collective route.
    Buffers grow (x 1.5, identically on every rank) when the Gaussian counts outgrow them; the handles are exchanged again then."""

    FLOATS = RECORD_FLOATS
    # Polls (~1 us each: a system-scope load + s_sleep 32) before a device-side wait gives up and raises the error word instead of hanging
    # the GPU.  A collective would block for the process group's timeout (minutes): the default is of that scale — 120 M polls = about
    # TWO MINUTES during which the waiting stream holds one wave busy-polling — so that a peer that is merely slow (a first-use path, an
    # allocator stall, a checkpoint on another rank) does not turn into an error; GSPL_PEER_MAX_POLLS overrides it (tests: 3e5 = 0.3 s).
    MAX_POLLS = int(os.environ.get("GSPL_PEER_MAX_POLLS", "120000000"))

    def __init__(self, rank: int, group, device):
        self.rank, self.group, self.device = int(rank), group, torch.device(device)
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if self.world > 16:
            raise NotImplementedError("PeerExchange: up to 16 ranks (one node)")
        self.step = 0
        self.cap_total = self.cap_rank = 0
        self.base: List[int] = []          # base pointer of every rank's buffer in THIS process
        self._mine = None
        self._opened: List[int] = []
        self._layout = None
        # the waits' error word lives in PINNED HOST memory: the wait kernel stores into it, the host reads it without a
        # synchronisation — `check()` is free and runs on every step (a wait that gave up is an exception on the next step at the
        # latest, never a silent stretch of steps over stale rows)
        self._err_host = torch.zeros(1, dtype=torch.int32).pin_memory() if self.device.type == "cuda" else torch.zeros(1, dtype=torch.int32)

    # ---- layout (identical on every rank) -----------------------------------------------------------------------------
    def _plan(self, cap_total: int, cap_rank: int):
        W, row = self.world, self.FLOATS * 4
        up = lambda x: (x + 255) // 256 * 256
        flags_f, flags_b = 0, up(8 * W)
        err = flags_b + up(8 * W)
        fwd = err + 256
        fwd_stride = up(cap_total * row)
        bwd = fwd + 2 * fwd_stride
        bwd_stride = up(W * cap_rank * row)
        return dict(flags_f=flags_f, flags_b=flags_b, err=err, fwd=fwd, fwd_stride=fwd_stride, bwd=bwd, bwd_stride=bwd_stride, total=bwd + 2 * bwd_stride)

    def _ensure(self, rows_per_rank: Sequence[int]):
        total, biggest = int(sum(rows_per_rank)), int(max(rows_per_rank))
        if self._mine is not None and total <= self.cap_total and biggest <= self.cap_rank:
            return
        import ctypes
        from . import _lib as L
        self.close()
        self.cap_total, self.cap_rank = int(total * 1.5) + 1024, int(biggest * 1.5) + 1024
        self._layout = lay = self._plan(self.cap_total, self.cap_rank)
        # Set-up is COLLECTIVE and must fail on every rank or on none: a rank that raised out of it alone would leave the others in the
        # barrier below (or in the next collective of the caller).  So every step that can fail locally is followed by an exchange of
        # the ranks' verdicts, and an error anywhere is raised everywhere — the caller (bench.py's `auto` transport, a renderer
        # configured with a fall-back) can then take the collective route on all ranks together.
        def agree(local_error, what):
            errors = [None] * self.world
            if self.world > 1:
                dist.all_gather_object(errors, None if local_error is None else f"rank {self.rank}: {local_error}", group=self.group)
            else:
                errors = [None if local_error is None else str(local_error)]
            failed = [e for e in errors if e is not None]
            if failed:
                self._release_local()
                raise RuntimeError(f"PeerExchange: {what} failed ({'; '.join(failed)})")

        with torch.cuda.device(self.device):
            ptr, handle = ctypes.c_void_p(), ctypes.create_string_buffer(64)
            err = None
            try:
                L.check(L.lib().gspl_peer_alloc(lay["total"], ctypes.byref(ptr), handle), "gspl_peer_alloc")
                self._mine = int(ptr.value)
            except Exception as e:      # noqa: BLE001 — whatever it was, the other ranks must hear of it
                err = e
            agree(err, "allocation of the receive buffers")
            handles = [None] * self.world
            if self.world > 1:
                dist.all_gather_object(handles, (self.rank, handle.raw), group=self.group)
            else:
                handles = [(self.rank, handle.raw)]
            self.base, self._opened = [0] * self.world, []
            err = None
            try:
                for r, raw in handles:
                    if r == self.rank:
                        self.base[r] = self._mine
                        continue
                    p = ctypes.c_void_p()
                    L.check(L.lib().gspl_peer_open(ctypes.create_string_buffer(raw, 64), ctypes.byref(p)), "gspl_peer_open")
                    self.base[r] = int(p.value)
                    self._opened.append(int(p.value))
            except Exception as e:      # noqa: BLE001
                err = e
            agree(err, "mapping of the peers' receive buffers (HIP IPC)")
        if self.world > 1:
            dist.barrier(group=self.group)          # nobody writes into a buffer its owner has not mapped and cleared yet

    def _release_local(self):
        """Frees this rank's buffer and mappings without any collective (the failure path of `_ensure`)."""
        if self._mine is None and not self._opened:
            return
        from . import _lib as L
        with torch.cuda.device(self.device):
            for p in self._opened:
                L.lib().gspl_peer_close(p)
            if self._mine is not None:
                L.lib().gspl_peer_free(self._mine)
        self._mine, self._opened, self.base = None, [], []

    def close(self):
        if self._mine is None:
            return
        from . import _lib as L
        torch.cuda.synchronize(self.device)
        if self.world > 1:
            dist.barrier(group=self.group)          # the peers are done with the old buffers
        with torch.cuda.device(self.device):
            for p in self._opened:
                L.lib().gspl_peer_close(p)
            L.lib().gspl_peer_free(self._mine)
        self._mine, self._opened, self.base = None, [], []

    def check(self):
        """Raises if a wait of this rank gave up (a peer that never signalled) — as soon as the kernel that gave up has run; no
        synchronisation (the word is in pinned host memory).  Clears the word: the exchange can be used again (or torn down)."""
        err = int(self._err_host[0])
        if err:
            self._err_host[0] = 0
            raise RuntimeError(f"PeerExchange: rank {self.rank} gave up waiting for the records of rank {err - 1} after {self.MAX_POLLS} polls "
                               f"(step {self.step}); rows received since then are stale")

    # ---- one exchange ---------------------------------------------------------------------------------------------------
    def _send(self, rows: torch.Tensor, begin: List[int], dst: List[int], flag_dst: List[int], my_flags: int, value: int):
        import ctypes
        from . import _lib as L
        W = self.world
        rows = rows.contiguous()
        with torch.cuda.device(self.device):
            L.call("gspl_peer_put_rows", W, L.ptr(rows), (ctypes.c_int64 * (W + 1))(*begin), (ctypes.c_void_p * W)(*dst), self.FLOATS, L.stream())
            L.call("gspl_peer_signal", W, (ctypes.c_void_p * W)(*flag_dst), value, L.stream())
            L.call("gspl_peer_wait", ctypes.c_void_p(my_flags), W, value, self.MAX_POLLS, ctypes.c_void_p(self._err_host.data_ptr()), L.stream())

    def route(self, rows_per_rank: Sequence[int], matrix: Optional[Sequence[Sequence[int]]] = None):
        """(forward, backward) callables for `ops.sharded_exchange`, as `all_to_all_route` returns them, for ONE training step.
        matrix[s][d] = rows rank s sends to rank d, identical on every rank (None: the fixed-size format, n_s rows to everybody; the
        counted format passes the per-destination visible counts every rank posted through a `HostMailbox`).
        forward: my rows grouped by destination -> the rows of every source for my camera, grouped by source (a view of this rank's
        receive buffer, valid until the exchange after the next one); backward: gradients of those -> gradients of the rows I sent."""
        rows_per_rank = [int(v) for v in rows_per_rank]
        W, me, row = self.world, self.rank, self.FLOATS * 4
        if matrix is None:
            matrix = [[n] * W for n in rows_per_rank]
        matrix = [[int(v) for v in r] for r in matrix]
        if len(matrix) != W or any(len(r) != W for r in matrix) or any(v < 0 or v > rows_per_rank[s] for s, r in enumerate(matrix) for v in r):
            raise ValueError("PeerExchange.route: matrix must be W x W with 0 <= matrix[s][d] <= rows_per_rank[s]")
        self.check()                    # a wait of an earlier step that gave up: raise before any more stale rows are used
        self._ensure(rows_per_rank)
        self.step += 1
        step, par, lay = self.step, self.step & 1, self._layout
        sent = [0]                      # my rows, grouped by destination
        for d in range(W):
            sent.append(sent[-1] + matrix[me][d])
        got = [0]                       # the rows I receive, grouped by source
        for s_ in range(W):
            got.append(got[-1] + matrix[s_][me])
        before_me_at = [sum(matrix[s_][d] for s_ in range(me)) for d in range(W)]       # my slot in destination d's forward area
        my_block_at = [sum(matrix[s_][d] for d in range(me)) for s_ in range(W)]        # my slot in source s's backward area
        fwd_of = lambda r: self.base[r] + lay["fwd"] + par * lay["fwd_stride"]
        bwd_of = lambda r: self.base[r] + lay["bwd"] + par * lay["bwd_stride"]
        view = lambda ptr, n: torch.as_tensor(_DevicePtr(ptr, (n, self.FLOATS), "<f4"), device=self.device)

        def forward(rows: torch.Tensor) -> torch.Tensor:
            assert rows.shape == (sent[-1], self.FLOATS) and rows.dtype == torch.float32, (tuple(rows.shape), sent[-1])
            self._send(rows, sent, [fwd_of(d) + before_me_at[d] * row for d in range(W)],
                       [self.base[d] + lay["flags_f"] + 8 * me for d in range(W)], self._mine + lay["flags_f"], step)
            return view(fwd_of(me), got[-1]) if got[-1] else rows.new_zeros((0, self.FLOATS))

        def backward(v_rows: torch.Tensor) -> torch.Tensor:
            assert v_rows.shape == (got[-1], self.FLOATS), (tuple(v_rows.shape), got[-1])
            self._send(v_rows.to(torch.float32), got, [bwd_of(s_) + my_block_at[s_] * row for s_ in range(W)],
                       [self.base[s_] + lay["flags_b"] + 8 * me for s_ in range(W)], self._mine + lay["flags_b"], step)
            return view(bwd_of(me), sent[-1]) if sent[-1] else v_rows.new_zeros((0, self.FLOATS))
        return forward, backward


def pack_visible(radii, means2d, depths, conics, compensations, opacities, rgbs, visibility) -> torch.Tensor:
    """[n_vis, 12] fp32 records of the splats `visibility` selects (one camera)."""
    rbits = radii.to(torch.int32).view(torch.float32)
    rec = torch.cat([means2d, depths.unsqueeze(-1), conics, compensations.unsqueeze(-1), opacities.reshape(-1, 1), rgbs,
                     rbits.unsqueeze(-1)], dim=-1)
    return rec[visibility]


def pack_all(radii, means2d, depths, conics, compensations, opacities, rgbs) -> torch.Tensor:
    """[N, 12] records of EVERY local splat for one camera, rows of invisible splats (radius <= 0) zeroed — radius 0 keeps them
    out of the receiver's lists.  The fixed-size counterpart of `pack_visible` (no count to exchange)."""
    rbits = radii.to(torch.int32).view(torch.float32)
    rec = torch.cat([means2d, depths.unsqueeze(-1), conics, compensations.unsqueeze(-1), opacities.reshape(-1, 1), rgbs,
                     rbits.unsqueeze(-1)], dim=-1)
    return torch.where((radii > 0).unsqueeze(-1), rec, torch.zeros((), dtype=rec.dtype, device=rec.device))


def unpack_records(rec: torch.Tensor):
    """-> radii i32 [n], means2d [n,2], depths [n], conics [n,3], compensations [n], opacities [n,1], rgbs [n,3]"""
    means2d, depths, conics, comp, opac, rgbs, rbits = torch.split(rec, [2, 1, 3, 1, 1, 3, 1], dim=-1)
    radii = rbits.detach().to(torch.float32).contiguous().view(torch.int32).squeeze(-1)     # (fp64 records in the CPU tests: exact)
    return radii, means2d, depths.squeeze(-1), conics, comp.squeeze(-1), opac, rgbs


def exchange_visible_splats(records_per_camera: Sequence[torch.Tensor], group=None) -> Tuple[torch.Tensor, List[int]]:
    """records_per_camera[j] = this rank's records visible from rank j's camera.  Returns the records every rank
    sent for THIS rank's camera, concatenated in rank order, and the per-source counts.  Differentiable."""
    world = dist.get_world_size(group)
    assert len(records_per_camera) == world
    dev = records_per_camera[0].device
    send_list = [int(r.shape[0]) for r in records_per_camera]
    recv_list = exchange_counts(send_list, dev, group)
    send = torch.cat(list(records_per_camera), dim=0)
    return all_to_all_rows(send, send_list, recv_list, group), recv_list


def reduce_densification_stats(xyz_gradient_accum: torch.Tensor, denom: torch.Tensor, max_radii2D: torch.Tensor, group=None):
    """Replicated-Gaussian mode: make the densification statistics identical on every rank
    (buffers of VanillaDensityControllerImpl, internal/density_controllers/vanilla_density_controller.py:60-67)."""
    if _nothing_to_exchange(group):
        return
    dist.all_reduce(xyz_gradient_accum, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(denom, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group)


def all_reduce_gradients(params: Iterable[torch.Tensor], group=None, average: bool = True) -> None:
    """Replicated-Gaussian mode: sum (or average) `p.grad` of every parameter over the ranks, in place — what DDP does for
    the reference's `configs/ddp.yaml`.  One collective per parameter tensor (five or six large tensors: each is its own
    bucket), all in flight before the first wait, so RCCL pipelines them over the xGMI links."""
    if _nothing_to_exchange(group):
        return
    world = dist.get_world_size(group)
    grads = [p.grad for p in params if p.grad is not None]
    works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True) for g in grads]
    for w in works:
        w.wait()
    if average:
        torch._foreach_mul_(grads, 1.0 / world)


def all_reduce_and_step(optimizer, params: Iterable[torch.Tensor], group=None, chunk_bytes: int = 48 << 20) -> None:
    """Replicated-Gaussian mode, gradient exchange OVERLAPPED with the optimizer: the gradient rows of every parameter go out as
    all-reduces of at most `chunk_bytes` each (all enqueued at once, smallest tensors first, so RCCL keeps the xGMI links busy
    back to back), and as soon as one chunk is reduced the fused Adam updates exactly those rows — while the collectives of the
    following chunks are still on the wire.  Same result as `all_reduce_gradients` + `optimizer.step()` (averaged gradients,
    one step count per parameter); needs an optimizer with `begin_chunked_step` / `step_rows` (gspl_amd.optimizers.FusedAdam)."""
    params = [p for p in params if p.grad is not None]
    if not (hasattr(optimizer, "begin_chunked_step") and hasattr(optimizer, "step_rows")):
        raise TypeError("all_reduce_and_step needs an optimizer with begin_chunked_step / step_rows (gspl_amd.optimizers.FusedAdam); "
                        f"got {type(optimizer).__name__} — use all_reduce_gradients + optimizer.step(...) instead")
    # `begin_chunked_step` advances the step count of EVERY parameter of the optimizer that has a gradient: each of them must be
    # reduced and updated here, or its bias correction would drift without an update
    owned = {id(p) for g in optimizer.param_groups for p in g["params"] if p.grad is not None}
    given = {id(p) for p in params}
    if owned != given:
        raise ValueError(f"all_reduce_and_step: `params` must be exactly the optimizer's parameters that have a gradient "
                         f"({len(owned - given)} of them missing, {len(given - owned)} not in the optimizer)")
    if any(p.dim() == 0 for p in params):
        raise ValueError("all_reduce_and_step: 0-dim parameters have no rows to chunk")
    if _nothing_to_exchange(group):
        optimizer.step()
        return
    world = dist.get_world_size(group)
    ready = optimizer.begin_chunked_step()
    rccl = is_rccl(group)
    chunks = []
    for p in sorted(params, key=lambda t: t.numel()):
        n = p.shape[0]
        row_bytes = max(p.numel() // max(n, 1), 1) * p.element_size()
        rows = max(4, (chunk_bytes // row_bytes) // 4 * 4)      # multiples of four rows: every chunk starts 16-byte aligned
        rows = min(n, rows)
        for lo in range(0, n, rows):
            hi = min(n, lo + rows)
            g = p.grad[lo:hi]                                   # a contiguous view: reduced in place
            if rccl:
                work = dist.all_reduce(g, op=dist.ReduceOp.AVG, group=group, async_op=True)
                chunks.append((work, None, p, lo, hi))
            else:                                               # gloo (tests on one shared GPU): host staged, summed then scaled
                wire = g.cpu()
                work = dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=group, async_op=True)
                chunks.append((work, wire, p, lo, hi))
    for work, wire, p, lo, hi in chunks:
        work.wait()                                             # RCCL: the current stream waits, the host does not
        if wire is not None:
            p.grad[lo:hi].copy_(wire.to(p.device, non_blocking=False)).mul_(1.0 / world)
        optimizer.step_rows(ready, p, lo, hi)


def redistribute_rows(local: torch.Tensor, destination: torch.Tensor, group=None, recv_counts: Optional[List[int]] = None) -> torch.Tensor:
    """Move row i of `local` to rank destination[i] (random rebalancing, gsplat_distributed_renderer.py:440-510):
    one all-to-all per tensor with rows grouped by destination (rows keep their relative order per source)."""
    world = dist.get_world_size(group)
    order = torch.argsort(destination, stable=True)
    send_list = [int(v) for v in torch.bincount(destination, minlength=world).tolist()]
    if recv_counts is None:
        recv_counts = exchange_counts(send_list, local.device, group)
    send = local.detach()[order].contiguous()
    return _all_to_all_rows_raw(send, send_list, list(recv_counts), group)


class MaskedReplicaAdam:
    """Replicated-Gaussian mode with a VISIBILITY-MASKED exchange and the optimizer state SHARDED by row ownership
    (SURVEY.md §5.8 / §8e "secondary": reduce-scatter + all-gather on the rows somebody saw, instead of the dense 236 B per Gaussian
    all-reduce of `all_reduce_and_step`).

    Every rank holds all N Gaussians (the replicas) and renders its own camera; row i of every parameter is OWNED by the rank whose
    `shard_bounds` contain i, and only the owner keeps Adam moments for it.  One step:
      1. all-reduce (MAX) of the per-rank visibility bytes [N] -> the rows ANY rank saw (N bytes);
      2. reduce-scatter by ownership: every rank sends each owner its gradient rows (all parameters side by side, F floats per row) of
         the seen rows in that owner's range — one variable-size all-to-all — and the owner adds the W contributions in rank order
         (a fixed order: the sum is bit-identical wherever it is formed) and averages;
      3. the owner applies the visibility-masked Adam (no bias correction, rows nobody saw untouched: the update of gsplat's
         `SelectiveAdam`, internal/optimizers.py:26-58) to ITS seen rows;
      4. all-gather of the UPDATED parameter rows (a second variable-size all-to-all), scattered into every replica.
    Wire bytes per rank and step: 2 (W-1)/W x F x 4 B x (rows seen), against 2 (W-1)/W x F x 4 B x N for the dense all-reduce; Adam
    work and moment memory are 1/W.  Replicas stay bit-identical: every rank writes the same received bytes into the same rows.
    `reshard` carries the moments across a densification (rows appended / pruned identically on every rank).
    The row arithmetic is a handful of torch ops on compact [rows seen / W, F] buffers (this is the secondary mode; the fused HIP
    Adam serves the primary paths)."""

    def __init__(self, named_params: Sequence[Tuple[str, torch.Tensor, float]], group=None, betas=(0.9, 0.999), eps: float = 1e-15):
        self.group = group
        self.names = [n for n, _, _ in named_params]
        self.params = [p for _, p, _ in named_params]
        self.lrs = [float(lr) for _, _, lr in named_params]
        self.betas, self.eps = betas, float(eps)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._layout()
        lo, hi = self.bounds[self.rank]
        dev = self.params[0].device
        self.exp_avg = torch.zeros((hi - lo, self.F), dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)

    def _layout(self):
        self.N = int(self.params[0].shape[0])
        assert all(int(p.shape[0]) == self.N for p in self.params), "every parameter has one row per Gaussian"
        self.widths = [p.numel() // max(self.N, 1) for p in self.params]
        self.F = sum(self.widths)
        self.bounds = [shard_bounds(self.N, self.world, r) for r in range(self.world)]
        dev = self.params[0].device
        self.lr_row = torch.cat([torch.full((w,), lr, dtype=torch.float32, device=dev) for w, lr in zip(self.widths, self.lrs)])

    def _exchange(self, rows: torch.Tensor, send_counts, recv_counts):
        if self.world == 1:
            return rows
        return _all_to_all_rows_raw(rows, list(send_counts), list(recv_counts), self.group)

    @torch.no_grad()
    def step(self, visible: torch.Tensor):
        """visible [N] bool: the rows THIS rank's camera saw (its gradient rows elsewhere are zero / absent)."""
        dev = self.params[0].device
        seen = visible.reshape(-1).to(torch.uint8)
        if self.world > 1:
            wire = _wire(seen, self.group)
            dist.all_reduce(wire, op=dist.ReduceOp.MAX, group=self.group)
            seen = wire.to(dev)
        idx = seen.nonzero().reshape(-1)                                   # ascending: grouped by owner
        edges = torch.tensor([b[0] for b in self.bounds] + [self.N], device=dev)
        cuts = torch.searchsorted(idx, edges).tolist()                     # the one host read-back of the step
        counts = [cuts[r + 1] - cuts[r] for r in range(self.world)]
        mine = counts[self.rank]
        lo, _ = self.bounds[self.rank]
        # 2. reduce-scatter by ownership
        grads = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(self.N, -1) for p in self.params], dim=1)
        recv = self._exchange(grads[idx].contiguous(), counts, [mine] * self.world)
        g = recv.reshape(self.world, mine, self.F)
        total = g[0].clone()
        for r in range(1, self.world):
            total += g[r]                                                  # rank order: the same sum on every owner
        total /= self.world
        # 3. visibility-masked Adam on the owner's seen rows
        my_idx = idx[cuts[self.rank]:cuts[self.rank + 1]]
        local = my_idx - lo
        b1, b2 = self.betas
        m = self.exp_avg[local].mul_(b1).add_(total, alpha=1 - b1)
        v = self.exp_avg_sq[local].mul_(b2).addcmul_(total, total, value=1 - b2)
        self.exp_avg[local] = m
        self.exp_avg_sq[local] = v
        rows = torch.cat([p.detach().reshape(self.N, -1)[my_idx] for p in self.params], dim=1)
        rows -= self.lr_row * m / (v.sqrt() + self.eps)
        # 4. all-gather of the updated rows
        out = self._exchange(rows.repeat(self.world, 1) if self.world > 1 else rows, [mine] * self.world, counts)
        col = 0
        for p, w in zip(self.params, self.widths):
            p.detach().reshape(self.N, -1)[idx] = out[:, col:col + w]
            col += w
        return int(idx.numel())

    @torch.no_grad()
    def reshard(self, new_params: Sequence[torch.Tensor], keep: Optional[torch.Tensor] = None, appended: int = 0):
        """After a densification that every rank performed identically: `keep` [old N] bool = rows that survive (None: all),
        `appended` new rows at the end (zero moments, as the reference's cat_tensors_to_optimizer gives them).  The moments are
        gathered, row-edited and cut to the new ownership ranges (472 B per Gaussian once per densification)."""
        full = [self.exp_avg, self.exp_avg_sq]
        if self.world > 1:
            sizes = [b[1] - b[0] for b in self.bounds]
            full = [_all_to_all_rows_raw(t.repeat(self.world, 1), [t.shape[0]] * self.world, sizes, self.group) for t in full]
        if keep is not None:
            full = [t[keep.to(t.device)] for t in full]
        if appended:
            full = [torch.cat([t, torch.zeros((appended, t.shape[1]), dtype=t.dtype, device=t.device)]) for t in full]
        self.params = list(new_params)
        self._layout()
        assert full[0].shape[0] == self.N, (full[0].shape, self.N)
        lo, hi = self.bounds[self.rank]
        self.exp_avg, self.exp_avg_sq = full[0][lo:hi].contiguous(), full[1][lo:hi].contiguous()
