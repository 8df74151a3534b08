"""The mutable run-time state of the operator layer, in ONE object instead of module-level globals scattered over the wrappers:
the process-wide switches (tests and bench flip them), and everything that is kept between frames — keyed by DEVICE (and tile grid
where it matters), so that several devices in one process and the viewer's threads (one renderer call per client thread,
internal/viewer/client.py:114) never share a slot that belongs to another device.

    STATE.fused_inria, .device_side_list_length, .speculative_emit, .track_hit_pixels, .keep_last_raster, .side_low_priority,
          .segmented_backward
    STATE.last_isects[(device index, tiles x, tiles y)]    list length of the last frame (introspection)
    STATE.capacity                                          ListCapacity: the room the next frame's speculative emission gets (per device, tile grid)
    STATE.speculation                                       how the guesses fared (frames / cold / misses; bench.py reports them)
    STATE.events[device index], .pinned_words, .pinned_ends[C]    free lists (an event / a pinned word costs ~15 us to construct)
    STATE.pending_updates[data_ptr]                         parameter updates in flight on the colour stream (optimizers.FusedAdam)
    STATE.last_raster                                       introspection: the last compositing forward's inputs (keep_last_raster)
    STATE.consts, .identity_slots, .zero_scalars            small per-device constant tensors
    STATE.backward_optimizers                               optimizers whose update the fused backward may apply itself (opt-in)
    STATE.stats_in_backward, .backward_stats                the density controller's statistics applied by the fused backward (density.py)

`gspl_amd.ops` keeps the historical module-level spellings (ops.FUSED_INRIA, ops._LAST_ISECTS, ...) as properties of the package
that read and write this object.  Container operations used on the hot path (dict get / set, list pop / append) are atomic under
the GIL; the statistics counters are advisory."""
from __future__ import annotations

import os
from typing import Optional


class ListCapacity:
    """Room (list entries) the next frame's SPECULATIVE emission is given, per (device index, tiles x, tiles y): the emission, the
    tile sort and the compositing launch are enqueued before the host knows the frame's list length, on buffers and grids sized by
    this number; a frame that needs more repeats them (a "miss": emission + sort + compositing again and a host wait, 0.35-0.45 ms).

    Policy (round 6): a DECAYED RUNNING MAXIMUM of the list entries PER SPLAT, times the frame's splat count, times a
    margin.  Rounds 3-5 used the previous frame's length x 1.25: right for a camera path that changes slowly, wrong for what the
    reference's data loader serves — a fresh random permutation of the training views every epoch (internal/dataset.py:216-217,
    258-259), neighbours in the stream whose lists differ by 2-3 x.  After one pass over the views the maximum covers all of them;
    per splat, so that a densification step (N grows by tens of per cent at once, the lists with it) is covered as well; decayed
    slowly (half-life ~3500 frames) so that a scene that shrinks for good — opacity reset, pruning — gives the room back.
    What over-capacity costs: address space (24 B per entry of capacity from the caching allocator: tens of MB against 288 GB)
    and workgroups of the capacity-sized sort grids that find nothing to do (measured: profiles/r20_capacity_*.txt)."""
    MARGIN, PAD, DECAY = float(os.environ.get("GSPL_CAPACITY_MARGIN", "1.125")), 65536, 0.9998      # (the env knob: A/B runs of what room costs)

    def __init__(self):
        self.peak: dict = {}

    def hint(self, key, n_splats: int) -> int:
        r = self.peak.get(key)
        return 0 if not r else int(r * max(int(n_splats), 1) * self.MARGIN) + self.PAD

    def observe(self, key, n_splats: int, n_isects: int) -> None:
        r = float(n_isects) / max(int(n_splats), 1)
        old = self.peak.get(key)
        self.peak[key] = r if old is None else max(r, old * self.DECAY)

    def set(self, key, n_splats: int, n_isects: int) -> None:
        """Tests: pretend the history so far consisted of one frame of `n_isects` entries for `n_splats` splats."""
        self.peak[key] = float(n_isects) / max(int(n_splats), 1)

    def clear(self) -> None:
        self.peak.clear()


class RuntimeState:
    __slots__ = ("fused_inria", "device_side_list_length", "speculative_emit", "track_hit_pixels", "keep_last_raster", "side_low_priority",
                 "segmented_backward",
                 "last_isects", "capacity", "speculation", "events", "pinned_words", "pinned_ends", "pending_updates", "last_raster", "consts",
                 "identity_slots", "zero_scalars", "new_event", "backward_optimizers", "stats_in_backward", "backward_stats")

    def __init__(self):
        env = os.environ.get
        # one C-ABI call per direction for the Inria rasterizer (csrc/fused.hip); GSPL_FUSED_INRIA=0: the stage-by-stage calls
        self.fused_inria: bool = env("GSPL_FUSED_INRIA", "1") != "0"
        # staged binning: with a speculative emission in flight the tile sort is enqueued before the host has read the list length
        # (False: wait for the count first, then sort — the round-1 order; kept for A/B runs and the tests of both orders)
        self.device_side_list_length: bool = True
        self.speculative_emit: bool = env("GSPL_SPECULATIVE_EMIT", "1") != "0"
        # the compositing backward also reports which splats some pixel actually composited and attaches the mask as
        # `has_hit_any_pixels` to the caller's screen-space tensor (the fork-only side channel gsplat's SelectiveAdam adapter reads,
        # internal/optimizers.py:39); off by default: one more byte store per (tile, splat) in the hot kernel
        self.track_hit_pixels: bool = False
        # introspection for bench.py / tools: the last compositing forward leaves its per-splat inputs and tile lists in `last_raster`
        self.keep_last_raster: bool = False
        # the colour kernel on the library's lowest-priority stream instead of a default-priority torch stream (measured: no gain)
        self.side_low_priority: bool = env("GSPL_SIDE_LOW_PRIORITY", "0") != "0"
        # segmented backward of the fused Inria call (csrc/gspl_composite.h): while tiles whose walk is longer than 768 list entries
        # are being met — the heavy-tailed lists of a trained scene — the forward leaves per-pixel checkpoints and the backward cuts
        # such a walk into segments for independent workgroups.  True (GSPL_SEGMENTED_BWD=1, default): adaptive, a scene without long
        # walks never leaves the plain kernels; "always": every frame (tests); False (=0): never.
        _seg = env("GSPL_SEGMENTED_BWD", "1")
        self.segmented_backward = False if _seg == "0" else ("always" if _seg == "always" else True)
        self.last_isects: dict = {}
        self.capacity = ListCapacity()
        self.speculation: dict = {"frames": 0, "cold": 0, "misses": 0}
        self.events: dict = {}
        self.pinned_words: list = []
        self.pinned_ends: dict = {}
        self.pending_updates: dict = {}
        self.last_raster: Optional[dict] = None
        self.consts: dict = {}
        self.identity_slots: dict = {}
        self.zero_scalars: dict = {}
        # optimizers constructed with fuse_into_backward=True (weak references): the fused Inria backward asks them for the moments of the
        # parameters it is differentiating and, if every one is claimed, applies the update itself (optimizers._FusedAdamBase)
        self.backward_optimizers: list = []
        # the density controller's statistics of a frame (density.update_densification_stats) applied by that frame's fused Inria
        # backward instead of by a launch of their own after it: `backward_stats` is the ONE pending request (density.StatsRequest:
        # the frame's radii tensor + the three state buffers), taken by the backward that owns those radii.  GSPL_STATS_IN_BACKWARD=0:
        # requests are refused and the controller's own launch runs, as before.
        self.stats_in_backward: bool = env("GSPL_STATS_IN_BACKWARD", "1") != "0"
        self.backward_stats = None
        self.new_event = _new_event      # (constructor of the events the free lists hand out; the host-only tests put a stand-in here)


def _new_event():
    import torch
    return torch.cuda.Event()


STATE = RuntimeState()
