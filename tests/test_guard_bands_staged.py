"""Guard bands for the STAGED paths (tests/test_guard_bands.py covers the fused call's own allocator): while the test runs, every CUDA tensor the
package asks torch for — outputs, gradient buffers, workspaces sized by the `gspl_*_workspace_bytes` functions — is carved out of a larger block
with a poisoned band on either side, and all bands must be intact afterwards.  Catches a kernel that writes past the end (or before the start)
of a buffer it was handed; the pool has no GPU sanitizer."""
import math
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

GUARD = 1024
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


class Guards:
    def __init__(self, monkeypatch, poison=None):
        """poison: a byte every handed-out buffer is filled with (0xFF: NaN as fp32, -1 as an index) — results that depend on what
        `torch.empty` happened to leave in a buffer change with it."""
        self.blocks = []
        self.poison = poison
        real_empty, real_empty_like = torch.empty, torch.empty_like
        guards = self

        def is_cuda(device):
            if device is None:
                return False
            d = torch.device(device) if not isinstance(device, torch.device) else device
            return d.type == "cuda"

        def carve(shape, dtype, device):
            dtype = dtype or torch.get_default_dtype()
            n = math.prod(shape) * torch.empty((), dtype=dtype).element_size()
            n_up = (n + 255) // 256 * 256
            outer = real_empty((n_up + 2 * GUARD,), dtype=torch.uint8, device=device)
            outer[:GUARD] = 0xA5
            outer[GUARD + n:] = 0xA5
            guards.blocks.append((n, outer))
            if guards.poison is not None:
                outer[GUARD:GUARD + n] = guards.poison
            return outer[GUARD:GUARD + n].view(dtype).view(tuple(shape))

        def empty(*size, dtype=None, device=None, **kw):
            if not is_cuda(device) or kw.get("pin_memory") or kw.get("layout") not in (None, torch.strided) or kw.get("memory_format") not in (None, torch.contiguous_format):
                return real_empty(*size, dtype=dtype, device=device, **kw)
            shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(int(x) for x in size)
            t = carve(shape, dtype, device)
            return t.requires_grad_(True) if kw.get("requires_grad") else t

        def zeros(*size, dtype=None, device=None, **kw):
            return empty(*size, dtype=dtype, device=device, **kw).zero_() if is_cuda(device) else real_zeros(*size, dtype=dtype, device=device, **kw)

        def empty_like(x, dtype=None, device=None, **kw):
            dev = device if device is not None else x.device
            if not is_cuda(dev) or kw.get("memory_format") not in (None, torch.contiguous_format, torch.preserve_format) or not x.is_contiguous():
                return real_empty_like(x, dtype=dtype, device=device, **kw)
            return carve(tuple(x.shape), dtype or x.dtype, dev)

        def zeros_like(x, dtype=None, device=None, **kw):
            dev = device if device is not None else x.device
            return empty_like(x, dtype=dtype, device=device, **kw).zero_() if is_cuda(dev) and x.is_contiguous() else real_zeros_like(x, dtype=dtype, device=device, **kw)

        real_zeros, real_zeros_like = torch.zeros, torch.zeros_like
        monkeypatch.setattr(torch, "empty", empty)
        monkeypatch.setattr(torch, "zeros", zeros)
        monkeypatch.setattr(torch, "empty_like", empty_like)
        monkeypatch.setattr(torch, "zeros_like", zeros_like)

    def check(self, what):
        torch.cuda.synchronize()
        for n, outer in self.blocks:
            assert bool((outer[:GUARD] == 0xA5).all()), f"{what}: a write BELOW a {n}-byte buffer"
            assert bool((outer[GUARD + n:] == 0xA5).all()), f"{what}: a write ABOVE a {n}-byte buffer"
        k = len(self.blocks)
        del self.blocks[:]
        return k


def test_no_write_outside_the_buffers_of_the_staged_ops(monkeypatch):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import fuzz_parity as FP
    import fuzz_differential as FD
    from gspl_amd import ops
    from gspl_amd.ops._state import STATE as S
    g = Guards(monkeypatch)
    total = 0
    for seed in range(8000, 8030):
        desc, case = FP.random_case(seed)
        for tile in (8, 16, 32):
            FD.gsplat(case, tile)
            total += g.check(f"seed {seed} {desc}: gsplat ops, tile {tile}")
        FD.inria(case, fused_inria=False)
        total += g.check(f"seed {seed} {desc}: staged Inria ops")
        FD.inria(case, fused_inria=False, device_side_list_length=False, speculative_emit=False)
        total += g.check(f"seed {seed} {desc}: staged Inria ops, host-side list length")
        means, scales, quats, opac, shs, cam, wimg, bg = case
        W, H = cam["width"], cam["height"]
        a = torch.rand(3, H, W, generator=torch.Generator().manual_seed(seed)).cuda().requires_grad_(True)
        b = torch.rand(3, H, W, generator=torch.Generator().manual_seed(seed + 1)).cuda()
        ops.photometric_loss(a, b).backward()
        total += g.check(f"seed {seed} {desc}: loss {H}x{W}")
        if means.shape[0] >= 4:
            ops.distCUDA2(means.cuda())
            total += g.check(f"seed {seed} {desc}: knn")
    assert total > 2000 and S is not None


def test_no_write_outside_the_buffers_sort_adam_density(monkeypatch):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import gspl_amd  # noqa: F401
    from gspl_amd import density, ops, optimizers
    g = Guards(monkeypatch)
    gen = torch.Generator().manual_seed(5)
    for n in (1, 2, 63, 2047, 2049, 70_001, 300_003):
        keys = torch.randint(0, 2 ** 31 - 1, (n,), generator=gen, dtype=torch.int64).to(torch.int32).cuda()
        vals = torch.arange(n, dtype=torch.int32).cuda()
        ops.radix_sort_pairs(keys, vals)
        g.check(f"radix sort of {n} pairs")
        rows = [3, 3, 4, 1, 3, 45]
        params = [torch.nn.Parameter(torch.randn(n, r, generator=gen).cuda()) for r in rows]
        opt = optimizers.FusedAdam([{"params": [p], "lr": 1e-3, "name": f"g{k}"} for k, p in enumerate(params)], eps=1e-15)
        for p in params:
            p.grad = torch.empty_like(p).normal_()
        opt.step()
        g.check(f"Adam over {n} rows")
        accum, denom, maxr = torch.zeros(n, 1, device="cuda"), torch.zeros(n, 1, device="cuda"), torch.zeros(n, device="cuda")
        radii = torch.randint(0, 9, (n,), generator=gen, dtype=torch.int32).cuda()
        density.update_densification_stats(torch.randn(n, 3, generator=gen).cuda(), None, radii, accum, denom, maxr, scale=320.0)
        g.check(f"densification statistics of {n} rows")


def test_the_guard_bands_do_detect_a_stray_write(monkeypatch):
    """The mechanism itself: a kernel (here: a torch copy through a view that reaches one element past the buffer) trips the check."""
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    g = Guards(monkeypatch)
    t = torch.empty(100, dtype=torch.float32, device="cuda")
    assert g.check("clean") == 1
    t = torch.zeros((7, 3), dtype=torch.float32, device="cuda:0")
    n, outer = g.blocks[-1]
    assert n == 84 and t.data_ptr() == outer.data_ptr() + GUARD and float(t.abs().sum()) == 0.0
    outer[GUARD + n:GUARD + n + 4].view(torch.float32).fill_(1.0)      # the element right behind t
    with pytest.raises(AssertionError, match="ABOVE"):
        g.check("stray")
    del g.blocks[:]
    t = torch.empty_like(t)
    n, outer = g.blocks[-1]
    outer[GUARD - 1] = 0
    with pytest.raises(AssertionError, match="BELOW"):
        g.check("stray")


def test_no_write_outside_the_buffers_of_the_renderer_plugins(monkeypatch):
    """The renderer plugins (v0, v1 with and without tile culling, the Gaussian-sharded renderer at world size 1 in its fused and staged
    forms and both exchange formats) on random cases, forward + backward, every buffer between guard bands."""
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fuzz_parity as FP
    from fakes import FakeCamera, FakeGaussianModel
    from gspl_amd.renderers import HipGSplatDistributedRenderer, HipGSplatRenderer, HipGSplatV1Renderer
    g = Guards(monkeypatch)
    dev = FP.dev
    renderers = {"v0": HipGSplatRenderer(absgrad=True), "v1": HipGSplatV1Renderer().instantiate(),
                 "v1-culling": HipGSplatV1Renderer(tile_based_culling=True).instantiate(),
                 "sharded-fused-counted": HipGSplatDistributedRenderer(fused_step=True, exchange="counted").instantiate(),
                 "sharded-fused-padded": HipGSplatDistributedRenderer(tile_based_culling=True, fused_step=True, exchange="padded").instantiate(),
                 "sharded-staged": HipGSplatDistributedRenderer(fused_step=False).instantiate()}
    total = 0
    for seed in range(9000, 9024):
        desc, (means, scales, quats, opac, shs, cam, wimg, bg) = FP.random_case(seed)
        deg = int(math.isqrt(shs.shape[1])) - 1
        for name, renderer in renderers.items():
            model = FakeGaussianModel(*[p.to(dev) for p in (means, scales, quats, opac, shs)], active_sh_degree=deg)
            out = renderer(FakeCamera(cam, dev), model, bg.to(dev))
            (out["render"] * wimg.to(dev)).sum().backward()
            total += g.check(f"seed {seed} {desc}: {name}")
    assert total > 1500



def test_results_do_not_depend_on_what_the_buffers_held(monkeypatch):
    """Every buffer the package takes from `torch.empty` (and every block of the fused call's allocator) pre-filled with 0xFF — NaN as
    fp32, -1 as an index or a count — and again with 0x00: the images are bit-equal to each other and the gradients finite and equal
    within the re-ordering of atomic sums, i.e. nothing is read before it is written."""
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import fuzz_parity as FP
    import fuzz_differential as FD
    from gspl_amd import _lib as L
    from gspl_amd.ops import inria
    g = Guards(monkeypatch, poison=0xFF)

    def filled(_ctx, tag, nbytes):      # the fused call's allocation call-back
        holder = inria._ALLOC_TLS.holder
        try:
            t = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=holder["device"])      # (guarded + poisoned by `g`)
            holder.setdefault(tag, []).append(t)
            return t.data_ptr()
        except Exception as e:      # noqa: BLE001
            holder["error"] = e
            return 0

    monkeypatch.setattr(inria, "_ALLOC_CB", L.ALLOC_FN(filled))
    loose = dict(rel=2e-4, frac=0.99, cap=5e-2)
    for seed in range(9500, 9530):
        desc, case = FP.random_case(seed)
        runs = {}
        for poison in (0xFF, 0x00):
            g.poison = poison
            runs[poison] = (FD.inria(case, segmented_backward=False), FD.inria(case, segmented_backward="always"),
                            FD.inria(case, fused_inria=False, segmented_backward=False), FD.gsplat(case, 16))
            g.check(f"seed {seed} {desc}: poison {poison:#x}")
        for a, b, what in zip(runs[0xFF], runs[0x00], ("fused", "fused, segmented", "staged", "gsplat ops")):
            FD.same(a, b, f"seed {seed} {desc}: {what}, buffers pre-filled with 0xFF against 0x00", **loose)
