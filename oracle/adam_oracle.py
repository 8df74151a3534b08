"""CPU oracle for the fused (visibility-masked) Adam update.

TEST INFRASTRUCTURE ONLY (imported by tests/); the product path is gaussian-splatting-lightning_amd/csrc/adam.hip.

Two references:
  * `selective_adam_step`: restates gsplat's published `selective_adam` kernel as the reference wraps it
    (internal/optimizers.py:26-58): no bias correction, rows with visibility False untouched.  **Parity unpinned**
    against that un-vendored CUDA package.
  * `torch.optim.Adam` itself (in-tree dependency of the reference's default configuration,
    internal/models/vanilla_gaussian.py:266-300) pins the bias-corrected, unmasked mode: the tests run it side by side.
"""
from __future__ import annotations

import torch


def selective_adam_step(p, g, m, v, visible, lr, b1, b2, eps):
    """In place on fp64 (or fp32) tensors p, m, v of shape [N, ...]; visible [N] bool."""
    sel = visible.reshape(-1, *([1] * (p.dim() - 1))).expand_as(p)
    m_new = b1 * m + (1 - b1) * g
    v_new = b2 * v + (1 - b2) * g * g
    p_new = p - lr * m_new / (v_new.sqrt() + eps)
    m.copy_(torch.where(sel, m_new, m))
    v.copy_(torch.where(sel, v_new, v))
    p.copy_(torch.where(sel, p_new, p))
