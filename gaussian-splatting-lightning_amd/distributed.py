"""
distributed.py — multi-GPU plumbing of the rasterizer path (one process per GPU, `torch.distributed`;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Two modes:

1. Gaussian-sharded (the reference's `configs/distributed.yaml`,
   internal/renderers/gsplat_distributed_renderer.py:132-211): every rank projects its shard for all W cameras
   and the visible splats travel to the rank that renders that camera.  Where the reference sends two
   messages per peer (a float [n,11] and an int [n] tensor, :195-202), this sends ONE packed 48-byte record
   per splat — xy(2) depth(1) conic(3) compensation(1) opacity(1) rgb(3) radius(1, int32 bits) — through a single
   autograd-aware `all_to_all_single` with split sizes; the backward pass is the reverse all-to-all of the same
   records' gradients.  On the MI355X full mesh each peer message rides its own xGMI link.
2. Replicated Gaussians, cameras sharded (BASELINE.json north_star wording; what bench.py --gpus N runs):
   only the densification statistics are all-reduced (`reduce_densification_stats`).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.distributed.nn.functional as dist_fn

RECORD_FLOATS = 12     # 48 B per visible splat


def shard_bounds(n_gaussians: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block owned by `rank` (gsplat_distributed_renderer.py:76-83: round(N/W) rows, last rank takes the rest)."""
    per = round(n_gaussians / world_size)
    lo = per * rank
    hi = n_gaussians if rank + 1 == world_size else lo + per
    return lo, hi


def pack_visible(radii, means2d, depths, conics, compensations, opacities, rgbs, visibility) -> torch.Tensor:
    """[n_vis, 12] fp32 records of the splats `visibility` selects (one camera)."""
    rbits = radii.to(torch.int32).view(torch.float32)
    rec = torch.cat([means2d, depths.unsqueeze(-1), conics, compensations.unsqueeze(-1), opacities.reshape(-1, 1), rgbs,
                     rbits.unsqueeze(-1)], dim=-1)
    return rec[visibility]


def unpack_records(rec: torch.Tensor):
    """-> radii i32 [n], means2d [n,2], depths [n], conics [n,3], compensations [n], opacities [n,1], rgbs [n,3]"""
    means2d, depths, conics, comp, opac, rgbs, rbits = torch.split(rec, [2, 1, 3, 1, 1, 3, 1], dim=-1)
    radii = rbits.detach().contiguous().view(torch.int32).squeeze(-1)
    return radii, means2d, depths.squeeze(-1), conics, comp.squeeze(-1), opac, rgbs


def exchange_visible_splats(records_per_camera: Sequence[torch.Tensor], group=None) -> Tuple[torch.Tensor, List[int]]:
    """records_per_camera[j] = this rank's records visible from rank j's camera.  Returns the records every rank
    sent for THIS rank's camera, concatenated in rank order, and the per-source counts.  Differentiable."""
    world = dist.get_world_size(group)
    assert len(records_per_camera) == world
    dev = records_per_camera[0].device
    send_counts = torch.tensor([r.shape[0] for r in records_per_camera], dtype=torch.int64, device=dev)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    recv_list = [int(v) for v in recv_counts.tolist()]
    send_list = [int(r.shape[0]) for r in records_per_camera]
    send = torch.cat(list(records_per_camera), dim=0).contiguous()
    out = torch.empty((sum(recv_list), RECORD_FLOATS), dtype=send.dtype, device=dev)
    out = dist_fn.all_to_all_single(out, send, output_split_sizes=recv_list, input_split_sizes=send_list, group=group)
    return out, recv_list


def reduce_densification_stats(xyz_gradient_accum: torch.Tensor, denom: torch.Tensor, max_radii2D: torch.Tensor, group=None):
    """Replicated-Gaussian mode: make the densification statistics identical on every rank
    (buffers of VanillaDensityControllerImpl, internal/density_controllers/vanilla_density_controller.py:60-67)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(xyz_gradient_accum, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(denom, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group)


def redistribute_rows(local: torch.Tensor, destination: torch.Tensor, group=None) -> torch.Tensor:
    """Move row i of `local` to rank destination[i] (random rebalancing, gsplat_distributed_renderer.py:440-510):
    one all_to_all_single per tensor with rows grouped by destination."""
    world = dist.get_world_size(group)
    order = torch.argsort(destination, stable=True)
    send_counts = torch.bincount(destination, minlength=world).to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    send = local[order].contiguous()
    flat = send.reshape(send.shape[0], -1)
    recv_list, send_list = [int(v) for v in recv_counts.tolist()], [int(v) for v in send_counts.tolist()]
    out = torch.empty((sum(recv_list), flat.shape[1]), dtype=flat.dtype, device=flat.device)
    dist.all_to_all_single(out, flat, output_split_sizes=recv_list, input_split_sizes=send_list, group=group)
    return out.reshape((out.shape[0],) + tuple(local.shape[1:]))
