"""Optimizer surgery the renderers need when they replace a model's tensors (sharding at setup).

Inside the reference repository `internal.density_controllers.density_controller.Utils.replace_tensors_to_properties`
(density_controller.py:148-203) is used as is; stand-alone (tests, bench) the function below has the same behaviour:
every optimizer group whose name appears in `tensors` gets the tensor as its new (single) parameter and its Adam moments
reset (or only the `selector` rows of them); names no optimizer knows become non-trainable parameters."""
from __future__ import annotations

from typing import Dict, List

import torch


def _own_replace(tensors: Dict[str, torch.Tensor], optimizers: List[torch.optim.Optimizer], selector=None) -> Dict[str, torch.Tensor]:
    new_parameters = {}
    for opt in optimizers:
        for group in opt.param_groups:
            tensor = tensors.get(group["name"], None)
            if tensor is None:
                continue
            assert len(group["params"]) == 1
            assert group["name"] not in new_parameters, "parameter `{}` appears in multiple optimizers".format(group["name"])
            old = group["params"][0]
            state = opt.state.get(old, None)
            new = torch.nn.Parameter(tensor.requires_grad_(True))
            if state is not None:
                if selector is not None:
                    state["exp_avg"][selector] = 0
                    state["exp_avg_sq"][selector] = 0
                else:
                    state["exp_avg"] = torch.zeros_like(tensor)
                    state["exp_avg_sq"] = torch.zeros_like(tensor)
                del opt.state[old]
                opt.state[new] = state
            group["params"][0] = new
            new_parameters[group["name"]] = new
    for k, v in tensors.items():
        if k not in new_parameters:
            new_parameters[k] = torch.nn.Parameter(v, requires_grad=False)
    return new_parameters


def replace_tensors_to_properties(tensors: Dict[str, torch.Tensor], optimizers, selector=None) -> Dict[str, torch.Tensor]:
    try:  # pragma: no cover - only inside the reference repo (needs lightning)
        from internal.density_controllers.density_controller import Utils  # type: ignore
        return Utils.replace_tensors_to_properties(tensors, optimizers, selector)
    except Exception:
        return _own_replace(tensors, optimizers, selector)
