"""Host logic of the sharded renderer's three-node step (`ops.sharded_front` / `sharded_exchange` / `sharded_back`) without a GPU.

The C-ABI is replaced by a recorder that checks every call against the binding's signature table (argument count, ctypes
convertibility) and writes the two host-visible results the control flow depends on (the per-camera record ends of the pack kernel,
the list length of the binning's scan); buffers hold whatever `torch.empty` left in them.  What is checked is what the host side
owns: the order of the launches forward and backward, that every parameter and every camera's screen-space positions receive a
gradient of the right shape through the stash hand-over between the exchange node and the front node, the speculative second frame
(lists whose length stays on the device), and the output contract of the plugin.  The numbers are the GPU tests' business
(tests/test_distributed_renderer.py, tests/test_renderers_gpu.py: fused against staged, bit for bit)."""
import contextlib
import ctypes
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


class _Event:
    def record(self, *a): pass
    def synchronize(self): pass
    def query(self): return True


@pytest.fixture()
def recorded_abi(monkeypatch):
    import gspl_amd  # noqa: F401
    from gspl_amd import _lib as L, ops
    calls = []

    def call(name, *args):
        restype, argtypes = L._SIGNATURES[name]
        assert len(args) == len(argtypes), (name, len(args), len(argtypes))
        for k, (a, t) in enumerate(zip(args, argtypes)):
            try:
                t.from_param(a)
            except (TypeError, ctypes.ArgumentError) as e:       # pragma: no cover
                raise AssertionError(f"{name}: argument {k} ({a!r}) does not convert to {t}") from e
        if name == "gspl_records_count_fwd":
            C, N, host = args[0], args[1], args[5]
            ends = (ctypes.c_int64 * C).from_address(host if isinstance(host, int) else host.value)
            for c in range(C):
                ends[c] = (c + 1) * (N // 2)
        if name == "gspl_bin_count":
            host = args[14]
            words = (ctypes.c_int64 * 2).from_address(host if isinstance(host, int) else host.value)
            words[0], words[1] = 1234, 0
        calls.append(name)

    def ptr(t, dtype=None, offset_bytes=0):
        if t is None:
            return None
        assert t.is_contiguous(), "gspl ops need contiguous tensors"
        return ctypes.c_void_p(t.data_ptr() + offset_bytes)


    monkeypatch.setattr(L, "call", call)
    monkeypatch.setattr(L, "ptr", ptr)
    monkeypatch.setattr(L, "stream", lambda: None)
    monkeypatch.setattr(L, "device_guard", lambda t: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    monkeypatch.setattr(torch.Tensor, "pin_memory", lambda self, *a, **k: self)
    monkeypatch.setattr(ops.STATE, "new_event", _Event)
    monkeypatch.setattr(ops, "_EVENTS", {})
    monkeypatch.setattr(ops, "_LAST_ISECTS", {})
    monkeypatch.setattr(ops.STATE, "capacity", type(ops.STATE.capacity)())
    monkeypatch.setattr(ops, "_PINNED_ENDS", {})
    monkeypatch.setattr(ops, "_PINNED_WORDS", [])
    return calls


# (the record counts leave for the host BEFORE the colour kernel: the two-phase pack)
FWD = ["gspl_project_fwd", "gspl_records_count_fwd", "gspl_sh_fwd_batched", "gspl_records_scatter_fwd", "gspl_records_unpack_fwd", "gspl_bin_count"]
BWD = ["gspl_composite_bwd_packed", "gspl_records_unpack_bwd", "gspl_records_pack_bwd", "gspl_sh_bwd_batched", "gspl_project_bwd"]


def _scene(n=400):
    from oracle import gsplat_oracle as O
    from fakes import FakeCamera, FakeGaussianModel
    means, scales, quats, opac, shs = O.synthetic_scene(n, seed=5)
    model = FakeGaussianModel(*[t.float() for t in (means, scales, quats, opac, shs)])
    cam = O.synthetic_camera(96, 64, 90.0, 91.0)
    cam["idx"] = 0
    return model, FakeCamera(cam, "cpu")


@pytest.mark.parametrize("exchange", ["counted", "padded"])
def test_three_node_step_launch_order_and_gradient_routes(recorded_abi, monkeypatch, exchange):
    from gspl_amd import ops
    from gspl_amd.renderers import HipGSplatDistributedRenderer
    from gspl_amd.renderers.hip_gsplat_distributed_renderer import HipGSplatDistributedRendererImpl
    calls = recorded_abi
    monkeypatch.setattr(HipGSplatDistributedRendererImpl, "_takes_fused_step",
                        lambda self, opacities, fmt, render_types: self.config.fused_step and "hard_inverse_depth" not in render_types)
    monkeypatch.setattr(HipGSplatDistributedRendererImpl, "_post_visible_count", lambda self, vis, pairs: None)     # (a CUDA event)
    model, cam = _scene()
    N = model.means.shape[0]
    renderer = HipGSplatDistributedRenderer(tile_based_culling=True, exchange=exchange).instantiate()
    # the fixed-size format has no count phase: colours, then one kernel that writes every (camera, splat) row
    front = FWD[:1] + ["gspl_sh_fwd_batched", "gspl_records_pad_fwd"] + FWD[4:] if exchange == "padded" else FWD
    renderer.camera_lookup = lambda idx, training: cam
    renderer.train()
    bg = torch.zeros(3)
    for frame in range(2):
        del calls[:]
        for t in model.leaves():
            t.grad = None
        out = renderer(cam, model, bg)
        assert set(out) == {"render", "hard_inverse_depth", "cameras", "projection_results_list", "visible_mask_list", "xys_grad_scale_required"}
        assert out["render"].shape == (3, 64, 96) and out["xys_grad_scale_required"] is True and renderer.last_exchange == exchange
        (radii, xys, depths, conics, comps, vis), = out["projection_results_list"]
        assert radii.shape == (N,) and radii.dtype == torch.int32 and xys.shape == (N, 2) and depths.shape == (N,)
        assert conics.shape == (N, 3) and comps.shape == (N,) and vis.dtype == torch.bool and out["visible_mask_list"][0] is vis
        assert xys.requires_grad and not depths.requires_grad and not conics.requires_grad
        if frame == 0:
            # no guess of the list length yet: count, wait, emit, sort, composite
            assert calls == front + ["gspl_bin_emit", "gspl_bin_sort", "gspl_composite_fwd"], calls
        else:
            # speculative emission, the sort reads the length on the device, compositing launched before the host looks at the count
            assert calls == front + ["gspl_bin_emit", "gspl_bin_sort_device_count", "gspl_composite_fwd"], calls
        del calls[:]
        out["render"].sum().backward()
        assert calls == BWD, calls
        for t in model.leaves():
            assert t.grad is not None and t.grad.shape == t.shape
        assert xys.grad is not None and xys.grad.shape == (N, 2)          # what DistributedVanillaDensityControllerImpl reads
    assert ops.SPECULATION["frames"] >= 2


def test_exchange_node_routes_both_directions_and_feeds_the_front_node(recorded_abi):
    from gspl_amd import ops
    model, cam = _scene(300)
    N = model.means.shape[0]
    C = 2
    viewmats = torch.eye(4).repeat(C, 1, 1)
    Ks = torch.eye(3).repeat(C, 1, 1)
    centers = torch.zeros(C, 3)
    stash = {}
    records, counts, radii, means2d, depths, conics, comps = ops.sharded_front(
        model.means, model.scales_, model.rotations_, model.opacities_, model.shs_dc, model.shs_rest, viewmats, Ks, centers, 96, 64, 0.3, 3, stash)
    assert counts == [N // 2, N // 2] and records.shape == (2 * (N // 2), 12) and means2d.shape == (C, N, 2) and radii.shape == (C, N)
    xys = ops.unbind_cameras(means2d)
    for x in xys:
        x.retain_grad()
    seen = []
    route = (lambda rows: (seen.append(("fwd", tuple(rows.shape))), rows.clone())[1],
             lambda v: (seen.append(("bwd", tuple(v.shape))), v.clone())[1])
    received = ops.sharded_exchange(records, stash, xys, route)
    image, alphas = ops.sharded_back(received, torch.zeros(3), 96, 64, 16, True, False)
    assert image.shape == (3, 64, 96) and alphas.shape == (64, 96)
    del recorded_abi[:]
    (image.sum() + alphas.sum()).backward()
    assert recorded_abi == BWD
    assert seen == [("fwd", tuple(records.shape)), ("bwd", tuple(records.shape))]
    assert all(x.grad is not None and x.grad.shape == (N, 2) for x in xys)
    assert "pack" not in stash and "grads" not in stash               # both consumed
    for t in model.leaves():
        assert t.grad is not None and t.grad.shape == t.shape


def test_records_bypassing_the_exchange_node_are_refused(recorded_abi):
    from gspl_amd import ops
    model, cam = _scene(100)
    stash = {}
    records, *_ = ops.sharded_front(model.means, model.scales_, model.rotations_, model.opacities_, model.shs_dc, model.shs_rest,
                                    torch.eye(4)[None], torch.eye(3)[None], torch.zeros(1, 3), 96, 64, 0.3, 3, stash)
    with pytest.raises(RuntimeError, match="sharded_exchange"):
        records.sum().backward()


def test_second_backward_of_the_three_node_step_says_what_happened(recorded_abi):
    """the exchange node releases the pack stage's state in its backward; a second backward over the same graph
    (retain_graph=True) must say so instead of raising a bare KeyError."""
    from gspl_amd import ops
    model, cam = _scene(100)
    stash = {}
    records, counts, radii, means2d, *_ = ops.sharded_front(model.means, model.scales_, model.rotations_, model.opacities_, model.shs_dc, model.shs_rest,
                                                            torch.eye(4)[None], torch.eye(3)[None], torch.zeros(1, 3), 96, 64, 0.3, 3, stash)
    received = ops.sharded_exchange(records, stash, ops.unbind_cameras(means2d), None)
    image, _ = ops.sharded_back(received, torch.zeros(3), 96, 64, 16, True, False)
    image.sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="single-use"):
        image.sum().backward()
