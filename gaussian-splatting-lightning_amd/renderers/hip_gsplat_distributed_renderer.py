"""HipGSplatDistributedRenderer — the reference's Gaussian-sharded multi-GPU renderer
(internal/renderers/gsplat_distributed_renderer.py:16-516, `configs/distributed.yaml`) on the HIP ops.

Per step, on each of W ranks (one process per GPU, MPStrategy = RCCL on ROCm):
  1. all_gather the W camera ids                                   (reference :319-323)
  2. project THIS rank's shard for all W cameras in ONE batched launch (`fully_fused_projection`, C = W) and
     evaluate the SH colours of all W cameras in ONE launch that reads the coefficients once (reference :252-311)
  3. ONE packed 48-B-record all-to-all (`distributed.exchange_visible_splats`; the reference sends a float and
     an int message, :195-202), autograd-aware: backward is the reverse all-to-all
  4. bin (list-only two-level binning, optional tile-based culling) + composite the received splats for the local
     camera                                                         (reference :356-389)
Outputs follow the reference (:407-414): `render`, `cameras`, `projection_results_list`, `visible_mask_list`,
`xys_grad_scale_required` — what `DistributedVanillaDensityControllerImpl` consumes.
Random redistribution (:432-510) uses one all_to_all_single per tensor (`distributed.redistribute_rows`).
"""
from __future__ import annotations

import contextlib
from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist

from .. import distributed as D
from .. import ops
from ..optim_utils import replace_tensors_to_properties
from .hip_gsplat_v1_renderer import GSplatV1
from .renderer import Renderer, RendererConfig, RendererOutputInfo, RendererOutputTypes, camera_hw, camera_scalars, implementation_tile_size


@dataclass
class HipGSplatDistributedRenderer(RendererConfig):
    block_size: int = 16
    anti_aliased: bool = True
    filter_2d_kernel_size: float = 0.3
    tile_based_culling: bool = False
    redistribute_interval: int = 1000
    redistribute_until: int = 15_000
    redistribute_threshold: float = 1.1
    # Format of the per-step exchange of rasterizer inputs.  "counted": the records of the VISIBLE splats, compacted, after an
    # exchange of their counts (the reference's scheme; here the counts leave the device before the colour kernel runs and their
    # exchange rides a control stream, but the host still waits for them and for the peers').  "padded": one record per (camera,
    # local splat), invisible rows zeroed — every size is known beforehand (the peers' Gaussian counts travel with the camera ids),
    # so the exchange has no count collective and no read-back, at 1 / (visible fraction) times the bytes.  "auto" (default): with ONE
    # rank (nothing travels) padded while at least `padded_min_visible` of the (camera, splat) pairs were visible in the last step,
    # else counted; with peers always counted — the padded bytes on a real interconnect are unmeasured.  Measured on one MI355X with the three-node
    # step (S-1080p-1M, 95 % visible; profiles/r04b_*): 1.42 ms padded against 1.46 ms counted at W = 1, 1.75-1.78 against 1.83-1.85 ms
    # with every collective issued to RCCL in a one-rank group.
    exchange: str = "auto"
    padded_min_visible: float = 0.5
    auto_padded_with_peers: bool = False      # let "auto" vote for the padded format with more than one rank too
    # How the records of a training step travel.  "collective": torch.distributed all-to-all (RCCL over xGMI; gloo in the tests) —
    # the reference's transport.  "peer": every rank writes its rows straight into the destination rank's receive buffer (HIP IPC
    # mapping, fine-grained device memory) and raises a flag the receiver's stream waits for (distributed.PeerExchange, csrc/peer.hip):
    # no collective and no host round trip in the step.  With exchange="auto" it uses the fixed-size "padded" format (every size is
    # known before the step); with "counted" the W x W matrix of visible counts travels through shared host memory first.
    # GPU, three-node step, training steps only — anything else takes the collective route.
    exchange_transport: str = "collective"
    # The step as three autograd nodes (`ops.sharded_front` / `sharded_exchange` / `sharded_back`: project + colours + pack, the
    # all-to-all, unpack + bin + composite) instead of eleven — same kernels, same numbers, less host work per step.  Taken on
    # the GPU (either exchange format) when nothing is overridden (`get_rgbs`) and only "rgb" is asked for; that path hands out
    # depths / conics / compensations of `projection_results_list` detached.  False: always the stage-by-stage formulation.
    fused_step: bool = True

    def instantiate(self, *args, **kwargs) -> Renderer:
        return HipGSplatDistributedRendererImpl(self)


class _Range:
    """Profiler span: the Lightning profiler's `profile(name)` when the trainer has one (as the reference,
    gsplat_distributed_renderer.py:316-379) and a roctx range (torch.cuda.nvtx = roctx on ROCm) for rocprofv3 --marker-trace."""

    def __init__(self, profiler, name):
        self.name = name
        self.ctx = profiler.profile(name) if profiler is not None else contextlib.nullcontext()

    def __enter__(self):
        self.ctx.__enter__()
        torch.cuda.nvtx.range_push(self.name)

    def __exit__(self, *exc):
        torch.cuda.nvtx.range_pop()
        return self.ctx.__exit__(*exc)


class HipGSplatDistributedRendererImpl(Renderer):
    profile_prefix = "[Renderer]GSplatDistributedRenderer."

    def __init__(self, config: HipGSplatDistributedRenderer) -> None:
        super().__init__()
        self.config = config
        # the rasterizer consumes (flatten_ids, isect_offsets) only: the list-only two-level binning builds exactly the lists of
        # isect_tiles + isect_offset_encode (bit-identical, tests/test_hip_parity.py) without 64-bit keys; with
        # tile_based_culling (configs/distributed-accel.yaml; reference :54-55) tile hits that cannot reach alpha >= 1/255 are dropped
        self.isect_encode = GSplatV1.isect_encode_lists_only
        if config.tile_based_culling:
            self.isect_encode = GSplatV1.isect_encode_tile_based_culling
        self.world_size = 1
        self.global_rank = 0
        self.group = None
        self.batched = True                                                   # one projection + one SH launch for all W cameras
        self.camera_lookup: Optional[Callable[[int, bool], object]] = None   # (camera idx, training) -> Camera
        self.on_density_changed = lambda: None
        self.profiler = None
        if config.exchange not in ("auto", "counted", "padded"):
            raise ValueError(f"exchange must be auto | counted | padded, got {config.exchange!r}")
        if config.exchange_transport not in ("collective", "peer"):
            raise ValueError(f"exchange_transport must be collective | peer, got {config.exchange_transport!r}")
        self._peer = None                         # distributed.PeerExchange, created by the first step that uses it
        self.last_exchange = None                 # format the last forward used ("counted" | "padded"; None: nothing exchanged)
        self._visible_permille = -1               # share of (camera, splat) pairs visible in this rank's last step; -1: unknown
        self._visible_pending = None              # (event, pinned word, total) of a count still on its way to the host
        self._peer_rows = None                    # per rank: [camera id, local Gaussian count, visible permille]

    def _span(self, name):
        return _Range(self.profiler, self.profile_prefix + name)

    def _backward_follows(self, records) -> bool:
        """The peer transport double-buffers on the promise of ONE backward per forward (a source may run one exchange ahead of the
        slowest reader, never two: its next forward needs every peer's backward rows of this step).  A forward that no backward can
        follow — no_grad, eval mode, records that carry no graph — takes the collective route."""
        return torch.is_grad_enabled() and self.training and bool(records.requires_grad)

    def close(self):
        """Releases what the peer transport holds for the life of the process otherwise: the fine-grained IPC receive buffers and the
        peers' mappings (`PeerExchange.close`: COLLECTIVE — every rank calls it, it waits for the device and meets the others in a
        barrier before anything is unmapped) and the shared-memory mailboxes.  Idempotent; the renderer can be used again afterwards
        (everything is created on first use)."""
        peer, self._peer = self._peer, None
        if peer is not None:
            peer.close()
        for name in ("_mailbox", "_count_mailbox"):
            box = self.__dict__.pop(name, None)
            if box is not None:
                box.close()

    def teardown(self, *args, **kwargs):      # (the name Lightning's hooks use for the end of fit / validate)
        self.close()

    # ---- setup: shard the Gaussians (reference :63-118) -------------------------------------------------
    def training_setup(self, module):
        self.world_size = module.trainer.world_size
        self.global_rank = module.trainer.global_rank
        lo, hi = D.shard_bounds(module.gaussian_model.n_gaussians, self.world_size, self.global_rank)
        new_tensors = {k: v[lo:hi] for k, v in module.gaussian_model.properties.items()}
        module.gaussian_model.properties = replace_tensors_to_properties(new_tensors, module.gaussian_optimizers)
        self.on_density_changed = module.density_updated_by_renderer
        self.on_density_changed()
        self.profiler = getattr(module.trainer, "profiler", None)

        def lookup(idx: int, training: bool):
            loader = module.trainer.train_dataloader if training else module.trainer.val_dataloaders
            return loader.dataset.image_cameras[idx]

        self.camera_lookup = lookup
        # batched projection needs one image size for every camera (reference :104-116)
        try:
            outputs = module.trainer.datamodule.dataparser_outputs
            for cams in (outputs.train_set.cameras, outputs.val_set.cameras):
                if cams.width.unique().shape[0] + cams.height.unique().shape[0] != 2:
                    self.batched = False
                    break
        except AttributeError:
            pass
        return None, None

    # ---- forward -----------------------------------------------------------------------------------------
    def _world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _poll_visible(self):
        """The visible count of the last padded step arrives through pinned memory; taken when it is there, never waited for."""
        pending = self._visible_pending
        if pending is not None and pending[0].query():
            self._visible_permille = int(1000 * int(pending[1][0]) // max(pending[2], 1))
            self._visible_pending = None
            ops._EVENTS.setdefault(pending[3], []).append(pending[0])

    def gather_cameras(self, viewpoint_camera, n_local: int = 0):
        """Camera of every rank, in rank order.  The same all-gather carries what else the ranks have to agree on before the
        step: every rank's local Gaussian count (the fixed split sizes of a padded exchange) and its visible share."""
        if self._world() == 1 and (D.SINGLE_RANK_SHORTCUT or not dist.is_initialized()):
            self._poll_visible()
            self._peer_rows = [[0, int(n_local), self._visible_permille]]
            return [viewpoint_camera]
        cams = []
        idx = int(camera_scalars(viewpoint_camera, ("idx",))[0])      # (a device read-back: orders the host past the previous forward)
        self._poll_visible()
        mine = [idx, int(n_local), self._visible_permille]
        dev = torch.device(viewpoint_camera.device)
        # The ids travel on a stream of their own: the collective and the read-back of its result then wait for nothing but
        # each other — on the current stream they would sit behind the previous step's backward and optimizer, and the host
        # could not start enqueueing this step before the device had drained.
        if self.config.exchange_transport == "peer" and dev.type == "cuda":
            # one node, one process per GPU: the rows go through shared host memory (a few microseconds, no stream involved)
            if self.__dict__.get("_mailbox") is None:
                self._mailbox = D.HostMailbox(dist.get_rank(self.group), self.group, width=len(mine))
            rows = self._mailbox.exchange(mine)
        else:
            rows = self._on_control_stream(D.gather_int_rows, mine, dev, self.group)
        self._peer_rows = rows
        for i in (r[0] for r in rows):
            cam = self.camera_lookup(i, self.training)
            if cam.device != viewpoint_camera.device:
                cam.to_device(viewpoint_camera.device)
            cams.append(cam)
        return cams

    def _on_control_stream(self, collective, payload, dev, group):
        """A small host-valued collective (ids, counts) on the renderer's control stream when the backend is RCCL: it is ordered
        behind nothing the step has enqueued on the current stream."""
        dev = torch.device(dev)
        if dev.type == "cuda" and D.is_rccl(group):
            ctl = self.__dict__.get("_ctl_stream")
            if ctl is None or ctl.device != dev:
                ctl = self.__dict__["_ctl_stream"] = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(ctl):
                return collective(payload, dev, group)
        return collective(payload, dev, group)

    def _post_visible_count(self, vis: torch.Tensor, pairs: int):
        """A padded step's visible count, for the next steps' votes: summed on the device, copied to pinned memory, an event
        behind it — picked up by `_poll_visible` when it has arrived, never waited for."""
        with torch.no_grad():
            n_vis = vis.sum(dtype=torch.int64).reshape(1)
            word = self.__dict__.get("_visible_word")
            if word is None:
                word = self.__dict__["_visible_word"] = torch.empty((1,), dtype=torch.int64).pin_memory()
            stale = self._visible_pending
            if stale is not None:           # a count nobody picked up: its event goes back to the pool once it has fired
                stale[0].synchronize()
                ops._EVENTS.setdefault(stale[3], []).append(stale[0])
            word.copy_(n_vis, non_blocking=True)
            ev = ops._take_event(vis.device)
            ev.record()
            self._visible_pending = (ev, word, pairs, vis.device.index)

    def _exchange_format(self) -> str:
        """Identical on every rank: a function of the configuration and of the gathered rows only."""
        c = self.config
        if c.exchange_transport == "peer" and c.exchange == "auto":
            return "padded"             # (every size known before the step: not even the counts have to travel)
        if c.exchange != "auto":
            return c.exchange
        if self._world() > 1 and not c.auto_padded_with_peers:
            # Every measurement behind "padded" was taken with ONE rank (nothing on the wire, profiles/r04b_*); with peers, padding
            # sends up to 1 / (visible fraction) times the bytes over xGMI, which nobody has measured: the reference's counted
            # scheme stays the default there until a multi-rank run says otherwise (exchange="padded" / auto_padded_with_peers=True opt in).
            return "counted"
        votes = [r[2] for r in self._peer_rows]
        return "padded" if min(votes) >= int(1000 * c.padded_min_visible) else "counted"

    def batch_project(self, cameras, pc, scales, scaling_modifier):
        """ONE projection launch and ONE SH launch for all W cameras (reference :252-311 loops over the cameras for the SH
        colours; here the coefficient rows are read once)."""
        viewmats, Ks, centers = self._camera_batch(cameras, scales.device)
        W, H = camera_hw(cameras[0])
        if scaling_modifier != 1.:
            scales = scales * scaling_modifier
        radii, means2d, depths, conics, comps = ops.fully_fused_projection(
            pc.get_means(), None, pc.get_rotations(), scales, viewmats=viewmats, Ks=Ks, width=W, height=H,
            eps2d=self.config.filter_2d_kernel_size, calc_compensations=True, packed=False)
        vis = radii > 0
        # per-camera views whose backward passes a batched gradient through (no zero fill + copy per slice)
        m2, dep, con, cmp_ = (ops.unbind_cameras(t) for t in (means2d, depths, conics, comps))
        results = [(radii[i], m2[i], dep[i], con[i], cmp_[i], vis[i]) for i in range(len(cameras))]
        if type(self).get_rgbs is HipGSplatDistributedRendererImpl.get_rgbs:
            rgbs = ops.sh_view_colors_batched(pc.active_sh_degree, pc.get_xyz, centers, pc.get_shs_dc(), pc.get_shs_rest(), radii)
            return results, list(ops.unbind_cameras(rgbs))
        # a subclass supplies its own colours (the reference's appearance-embedding renderer overrides `get_rgbs`,
        # gsplat_distributed_appearance_embedding_renderer.py:67-84): per camera, as the reference calls it (:308)
        return results, [self.get_rgbs(pc, cam, r) for cam, r in zip(cameras, results)]

    def _camera_rows(self, c, device):
        """(view matrix [4,4], intrinsics [3,3], centre [3]) of ONE camera, built once per camera: the intrinsics matrix alone is an
        `eye` and four element copies — six tiny launches.  An entry KEEPS its source tensors and is valid only while every one of
        them is the same object at the same version (as `GSplatV1.preprocess_camera`): an address handed out again by the allocator
        to a new camera's tensor can never match, an in-place pose edit misses."""
        src = (c.world_to_camera, c.camera_center, c.fx, c.fy, c.cx, c.cy)
        key = tuple((id(v), v._version) if isinstance(v, torch.Tensor) else float(v) for v in src) + (str(device),)
        cache = self.__dict__.setdefault("_camera_rows_cache", {})
        hit = cache.get(key)
        if hit is not None and all((a is b) if isinstance(b, torch.Tensor) else (a == b) for a, b in zip(hit[0], src)):
            return hit[1]
        if len(cache) > 4096:
            cache.clear()
        rows = (c.world_to_camera.T.contiguous(), GSplatV1.get_intrinsics_matrix(c.fx, c.fy, c.cx, c.cy, device), c.camera_center)
        cache[key] = (src, rows)
        return rows

    def _camera_batch(self, cameras, device):
        """Stacked view matrices [W,4,4], intrinsics [W,3,3] and centres [W,3] of this step's cameras.  Two levels of caching: the
        per-CAMERA rows (`_camera_rows`: a data set's cameras are built once) and the stacks of a camera TUPLE.  A loader that serves
        a fresh permutation every epoch (the reference's, internal/dataset.py:216-217) hands W > 1 ranks a new tuple nearly every step:
        with the tuple cache alone that was ~50 tiny launches per step at W = 8 (profiles/r27_w8_shared_gpu_sequence_process_0.txt:
        eight times `eye` + four element copies, plus the stacks); now it is the three `stack` launches."""
        src = tuple(v for c in cameras for v in (c.world_to_camera, c.camera_center, c.fx, c.fy, c.cx, c.cy))
        key = tuple((id(v), v._version) if isinstance(v, torch.Tensor) else float(v) for v in src) + (str(device),)
        cache = self.__dict__.setdefault("_camera_batches", {})
        hit = cache.get(key)
        if hit is not None and len(hit[0]) == len(src) and all((a is b) if isinstance(b, torch.Tensor) else (a == b) for a, b in zip(hit[0], src)):
            return hit[1]
        if len(cache) > 256:
            cache.clear()
        rows = [self._camera_rows(c, device) for c in cameras]
        stacks = tuple(torch.stack([r[k] for r in rows]).contiguous() for k in range(3))
        cache[key] = (src, stacks)
        return stacks

    def non_batch_project(self, cameras, pc, scales, scaling_modifier):
        """Cameras of different sizes: one projection and one SH launch per camera (reference :238-250)."""
        if scaling_modifier != 1.:
            scales = scales * scaling_modifier
        results, rgbs = [], []
        for cam in cameras:
            radii, means2d, depths, conics, comps = GSplatV1.project(
                GSplatV1.preprocess_camera(cam), means3d=pc.get_means(), scales=scales, quats=pc.get_rotations(),
                eps2d=self.config.filter_2d_kernel_size, anti_aliased=True)
            r = (radii[0], means2d[0], depths[0], conics[0], comps[0], radii[0] > 0)
            results.append(r)
            rgbs.append(self.get_rgbs(pc, cam, r))
        return results, rgbs

    def get_rgbs(self, pc, camera, projection_results):
        """Colours of the local splats for one camera (reference :416-421); `projection_results[-1]` is the visibility mask."""
        return ops.sh_view_colors(pc.active_sh_degree, pc.get_xyz, camera.camera_center, pc.get_shs_dc(), pc.get_shs_rest(),
                                  projection_results[-1])

    def forward(self, viewpoint_camera, pc, bg_color: torch.Tensor, scaling_modifier=1.0, render_types: list = None, **kwargs):
        if render_types is None:
            render_types = ["rgb"]
        with self._span("forward"):
            scales, opacities = pc.get_scales(), pc.get_opacities()
            n_local = int(opacities.shape[0])
            with self._span("gather_cameras"):
                cameras = self.gather_cameras(viewpoint_camera, n_local)
            rank = dist.get_rank(self.group) if dist.is_initialized() else 0
            fmt = self._exchange_format()
            exchanging = len(cameras) > 1 or (dist.is_initialized() and not D.SINGLE_RANK_SHORTCUT)
            peer_counts = [r[1] for r in self._peer_rows]
            pairs = max(len(cameras) * n_local, 1)
            if self._takes_fused_step(opacities, fmt, render_types):
                return self._forward_fused(cameras, rank, pc, scales, opacities, bg_color, scaling_modifier, exchanging, pairs, fmt, peer_counts)
            with self._span("project"):
                project = self.batch_project if self.batched else self.non_batch_project
                projection_results_list, rgb_list = project(cameras, pc, scales, scaling_modifier)
            for r in projection_results_list:
                if r[1].requires_grad:
                    r[1].retain_grad()              # per-camera xys: what the distributed density controller reads

            with self._span("rasterizer_required_data_all2all"):
                self.last_exchange = fmt
                if opacities.is_cuda and fmt == "padded":
                    # one record per (camera, local splat): sizes known beforehand, no read-back; the visible count follows through
                    # pinned memory for the next steps' votes
                    records = ops.pack_all_records(projection_results_list, rgb_list, opacities)
                    self._post_visible_count(torch.stack([r[5] for r in projection_results_list]), pairs)
                    if exchanging:
                        records = D.all_to_all_rows(records, [n_local] * len(cameras), peer_counts, self.group)
                    radii, means2d, depths, conics, opac, rgbs = ops.unpack_visible_records(records, self.config.anti_aliased)
                    opac = opac.unsqueeze(0)
                elif opacities.is_cuda:
                    # one pack kernel for all cameras, one all-to-all, one unpack kernel (csrc/records.hip)
                    records, send_counts = ops.pack_visible_records(projection_results_list, rgb_list, opacities)
                    self._visible_permille, self._visible_pending = int(1000 * sum(send_counts) // pairs), None
                    if exchanging:
                        recv_counts = D.exchange_counts(send_counts, records.device, self.group)
                        records = D.all_to_all_rows(records, send_counts, recv_counts, self.group)
                    radii, means2d, depths, conics, opac, rgbs = ops.unpack_visible_records(records, self.config.anti_aliased)
                    opac = opac.unsqueeze(0)
                else:
                    # host tensors (the CPU tests of the exchange logic): the same steps as torch ops
                    if fmt == "padded":
                        send = torch.cat([D.pack_all(r[0], r[1], r[2], r[3], r[4], opacities, rgb)
                                          for r, rgb in zip(projection_results_list, rgb_list)], dim=0)
                        self._visible_permille = int(1000 * sum(int(r[5].sum()) for r in projection_results_list) // pairs)
                        received = D.all_to_all_rows(send, [n_local] * len(cameras), peer_counts, self.group) if len(cameras) > 1 else send
                    else:
                        records = [D.pack_visible(r[0], r[1], r[2], r[3], r[4], opacities, rgb, r[5])
                                   for r, rgb in zip(projection_results_list, rgb_list)]
                        self._visible_permille = int(1000 * sum(int(r.shape[0]) for r in records) // pairs)
                        if len(cameras) > 1:
                            received, _ = D.exchange_visible_splats(records, self.group)
                        else:
                            received = records[0]
                    radii, means2d, depths, conics, comps, opac, rgbs = D.unpack_records(received)
                    if self.config.anti_aliased:
                        opac = opac * comps.unsqueeze(-1)
                    opac = opac.squeeze(-1).unsqueeze(0)

            local = cameras[rank]
            W, H = camera_hw(local)
            pre = (None, None, (W, H))
            projections = (radii.unsqueeze(0), means2d, depths.unsqueeze(0), conics.unsqueeze(0), None)
            isects = self.isect_encode(pre, (projections[0], means2d.unsqueeze(0), projections[2], projections[3], None),
                                       opac, tile_size=implementation_tile_size(self.config.block_size), lazy=True)
            with self._span("rasterize"):
                # [3,H,W] straight from the kernel (the reference permutes an [H,W,3] image)
                rgb, _ = GSplatV1.rasterize(pre, projections, isects, opac, colors=rgbs, background=bg_color,
                                            tile_size=implementation_tile_size(self.config.block_size), absgrad=False, channels_first=True)
                hard_inverse_depth_im = None
                if "hard_inverse_depth" in render_types:
                    inverse_depth = 1. / (depths.clamp_min(0.) + 1e-8).unsqueeze(-1)
                    hard_inverse_depth_im, _ = GSplatV1.rasterize(pre, projections, isects, opac + (1 - opac.detach()), colors=inverse_depth,
                                                                  background=torch.zeros((1,), dtype=torch.float, device=bg_color.device),
                                                                  tile_size=implementation_tile_size(self.config.block_size), absgrad=False, channels_first=True)
        return {
            "render": rgb,
            "hard_inverse_depth": hard_inverse_depth_im,
            "cameras": cameras,
            "projection_results_list": projection_results_list,
            "visible_mask_list": [r[5] for r in projection_results_list],
            "xys_grad_scale_required": True,
        }

    def _takes_fused_step(self, opacities, fmt, render_types) -> bool:
        return (self.config.fused_step and opacities.is_cuda and self.batched and "hard_inverse_depth" not in render_types
                and type(self).get_rgbs is HipGSplatDistributedRendererImpl.get_rgbs)

    def _forward_fused(self, cameras, rank, pc, scales, opacities, bg_color, scaling_modifier, exchanging, pairs, fmt, peer_counts):
        """The same step as three autograd nodes (see `fused_step`); called inside the "forward" span."""
        c = self.config
        W, H = camera_hw(cameras[0])
        with self._span("project"):
            viewmats, Ks, centers = self._camera_batch(cameras, scales.device)
            if scaling_modifier != 1.:
                scales = scales * scaling_modifier
            stash = {}
            records, send_counts, radii, means2d, depths, conics, comps = ops.sharded_front(
                pc.get_means(), scales, pc.get_rotations(), opacities, pc.get_shs_dc(), pc.get_shs_rest(), viewmats, Ks, centers,
                W, H, c.filter_2d_kernel_size, pc.active_sh_degree, stash, padded=(fmt == "padded"))
            vis = radii > 0
            xys = ops.unbind_cameras(means2d)
            projection_results_list = [(radii[i], xys[i], depths[i], conics[i], comps[i], vis[i]) for i in range(len(cameras))]
            for x in xys:
                if x.requires_grad:
                    x.retain_grad()                 # per-camera xys: what the distributed density controller reads
        with self._span("rasterizer_required_data_all2all"):
            self.last_exchange = fmt
            route = None
            if fmt == "padded":
                # every size was known before the step (the peers' Gaussian counts came with the camera ids): no count exchange,
                # no read-back; the visible share follows through pinned memory for the next steps' votes — every eighth padded
                # step (a reduction, a cast and two copies: ~28 us of small launches at 1 M Gaussians; the vote only has to
                # notice a scene that is drifting out of view, and every rank votes with what it has)
                n = self.__dict__["_padded_steps"] = self.__dict__.get("_padded_steps", 0) + 1
                if n % 8 == 1:
                    self._post_visible_count(vis, pairs)
                if exchanging and c.exchange_transport == "peer" and self._backward_follows(records):
                    if self._peer is None:
                        self._peer = D.PeerExchange(rank, self.group, records.device)
                    route = self._peer.route(peer_counts)      # (raises first if a wait of an earlier step gave up: the error word is
                    #                                             in pinned host memory, the check costs nothing and runs every step)
                elif exchanging:
                    route = D.all_to_all_route(send_counts, peer_counts, self.group)
            else:
                self._visible_permille, self._visible_pending = int(1000 * sum(send_counts) // pairs), None
                if exchanging and c.exchange_transport == "peer" and self._backward_follows(records):
                    # the whole W x W count matrix through shared host memory (every rank posts its row), then direct peer writes
                    if self.__dict__.get("_count_mailbox") is None:
                        self._count_mailbox = D.HostMailbox(rank, self.group, width=len(cameras))
                    matrix = self._count_mailbox.exchange(send_counts)
                    if self._peer is None:
                        self._peer = D.PeerExchange(rank, self.group, records.device)
                    route = self._peer.route(peer_counts, matrix)
                elif exchanging:
                    # the counts were on the host before the colour kernel and the scatter had run (two-phase pack): on RCCL their
                    # exchange goes out on the control stream, next to those kernels instead of behind them
                    recv_counts = self._on_control_stream(D.exchange_counts, send_counts, records.device, self.group)
                    route = D.all_to_all_route(send_counts, recv_counts, self.group)
            records = ops.sharded_exchange(records, stash, xys, route)
        with self._span("rasterize"):
            local = cameras[rank]
            W, H = camera_hw(local)
            rgb, _ = ops.sharded_back(records, bg_color, W, H, implementation_tile_size(c.block_size), c.anti_aliased, c.tile_based_culling)
        return {
            "render": rgb,
            "hard_inverse_depth": None,
            "cameras": cameras,
            "projection_results_list": projection_results_list,
            "visible_mask_list": [r[5] for r in projection_results_list],
            "xys_grad_scale_required": True,
        }

    # ---- periodic rebalancing (reference :423-510) -------------------------------------------------------
    def after_training_step(self, step: int, module):
        c = self.config
        if c.redistribute_interval < 0 or step >= c.redistribute_until or step % c.redistribute_interval != 0:
            return
        self.redistribute(module)

    def redistribute(self, module):
        with torch.no_grad():
            counts = D.gather_ints(module.gaussian_model.get_xyz.shape[0], module.gaussian_model.get_xyz.device, self.group)
            if min(counts) * self.config.redistribute_threshold >= max(counts):
                return
            self.random_redistribute(module)

    def random_redistribute(self, module, destination: Optional[torch.Tensor] = None):
        """Every Gaussian moves to a uniformly random rank; the Adam moments travel with their parameters (reference :440-510)."""
        xyz = module.gaussian_model.get_xyz
        n = xyz.shape[0]
        if destination is None:
            destination = torch.randint(0, self.world_size, (n,), device=xyz.device)
        send = [int(v) for v in torch.bincount(destination, minlength=self.world_size).tolist()]
        recv = D.exchange_counts(send, xyz.device, self.group)
        move = lambda t: D.redistribute_rows(t, destination, self.group, recv_counts=recv)
        new_tensors = {}
        with torch.no_grad():
            for opt in module.gaussian_optimizers:
                for group in opt.param_groups:
                    assert len(group["params"]) == 1
                    old = group["params"][0]
                    state = opt.state.get(old, None)
                    new = torch.nn.Parameter(move(old).requires_grad_(True))
                    if state is not None:
                        state["exp_avg"], state["exp_avg_sq"] = move(state["exp_avg"]), move(state["exp_avg_sq"])
                        del opt.state[old]
                        opt.state[new] = state
                    group["params"][0] = new
                    new_tensors[group["name"]] = new
            for name in module.gaussian_model.get_property_names():
                if name not in new_tensors:
                    new_tensors[name] = move(module.gaussian_model.get_property(name))
        module.gaussian_model.properties = new_tensors
        self.on_density_changed()

    def get_available_outputs(self) -> Dict:
        return {"rgb": RendererOutputInfo("render"),
                "hard_inverse_depth": RendererOutputInfo("hard_inverse_depth", type=RendererOutputTypes.GRAY)}
