"""Optimizer surgery the sharded renderer needs when it swaps a model's property tensors (sharding at setup).

Contract (what `HipGSplatDistributedRendererImpl.training_setup` relies on; the reference gets the same effect from its density
controller utilities, internal/density_controllers/density_controller.py:148-203, which are used when that package is importable):
    * `tensors` maps property names to the tensors that replace the model's current ones;
    * a name that an optimizer trains (a param group called `name` with exactly one parameter) becomes a fresh trainable Parameter in
      that group, and the optimizer state of the old parameter moves to the new one with its per-row moments cleared — all rows, or
      only the rows `selector` picks;
    * every other name becomes a frozen Parameter;
    * the result maps every name to its Parameter.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch


def _groups_by_name(optimizers: List[torch.optim.Optimizer]) -> Dict[str, Tuple[torch.optim.Optimizer, dict]]:
    index: Dict[str, Tuple[torch.optim.Optimizer, dict]] = {}
    for optimizer in optimizers:
        for group in optimizer.param_groups:
            name = group.get("name")
            if name is None:
                continue
            if name in index:
                raise ValueError(f"property {name!r} is trained by more than one optimizer group")
            index[name] = (optimizer, group)
    return index


def _swap_parameters(tensors: Dict[str, torch.Tensor], optimizers: List[torch.optim.Optimizer], selector=None) -> Dict[str, torch.nn.Parameter]:
    index = _groups_by_name(optimizers)
    out: Dict[str, torch.nn.Parameter] = {}
    for name, tensor in tensors.items():
        owner = index.get(name)
        if owner is None:
            out[name] = torch.nn.Parameter(tensor, requires_grad=False)
            continue
        optimizer, group = owner
        if len(group["params"]) != 1:
            raise ValueError(f"optimizer group {name!r} holds {len(group['params'])} parameters, expected one")
        previous = group["params"][0]
        fresh = torch.nn.Parameter(tensor.requires_grad_(True))
        state = optimizer.state.pop(previous, None)
        if state is not None:
            for key in ("exp_avg", "exp_avg_sq"):
                if key not in state:
                    continue
                if selector is None:
                    state[key] = torch.zeros_like(tensor)
                else:
                    state[key][selector] = 0
            optimizer.state[fresh] = state
        group["params"][0] = fresh
        out[name] = fresh
    return out


def replace_tensors_to_properties(tensors: Dict[str, torch.Tensor], optimizers, selector=None) -> Dict[str, torch.Tensor]:
    try:  # pragma: no cover - only inside the reference repo (needs lightning)
        from internal.density_controllers.density_controller import Utils  # type: ignore
    except Exception:
        return _swap_parameters(tensors, optimizers, selector)
    return Utils.replace_tensors_to_properties(tensors, optimizers, selector)
