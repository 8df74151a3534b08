"""TEST INFRASTRUCTURE ONLY (never imported by the product).  CPU restatement of the densification statistics of
`VanillaDensityControllerImpl.update_states` / `_add_densification_stats`
(reference internal/density_controllers/vanilla_density_controller.py:101-123), written exactly as the reference does it
(boolean-mask gathers and scatters).  Pinned: the reference module imports `lightning` (absent here), so the lines are
restated, not imported — they are plain tensor indexing."""
import torch


def update_states(max_radii2D, xyz_gradient_accum, denom, grad, visibility_filter, radii, scale=None):
    """Returns the updated (max_radii2D [N], xyz_gradient_accum [N,1], denom [N,1]); inputs are not modified."""
    max_radii2D, xyz_gradient_accum, denom = max_radii2D.clone(), xyz_gradient_accum.clone(), denom.clone()
    # :107-110
    max_radii2D[visibility_filter] = torch.max(max_radii2D[visibility_filter], radii[visibility_filter].to(max_radii2D.dtype))
    # :116-123
    scaled_grad = grad[visibility_filter, :2]
    if scale is not None:
        scaled_grad = scaled_grad * scale
    grad_norm = torch.norm(scaled_grad, dim=-1, keepdim=True)
    xyz_gradient_accum[visibility_filter] += grad_norm
    denom[visibility_filter] += 1
    return max_radii2D, xyz_gradient_accum, denom
