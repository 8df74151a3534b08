"""Checkpoint / PLY formats (gspl_amd.formats) against the reference's `GaussianPlyUtils`
(internal/utils/gaussian_utils.py:52-255; its own test: tests/ckpt2ply_test.py) and `utils/merge_distributed_ckpts.py`."""
import os
import sys
import types

import numpy as np
import pytest
import torch

import gspl_amd  # noqa: F401
from gspl_amd import formats as F

REF_ROOT = os.environ.get("GSPL_REFERENCE_ROOT", "/root/reference")


def _state_dict(n=257, k_rest=15, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    p = "gaussian_model.gaussians."
    return {p + "means": r(n, 3), p + "shs_dc": r(n, 1, 3), p + "shs_rest": r(n, k_rest, 3), p + "opacities": r(n, 1), p + "scales": r(n, 3),
            p + "rotations": r(n, 4), "gaussian_model._active_sh_degree": torch.tensor(3, dtype=torch.uint8)}


@pytest.mark.parametrize("k_rest", [15, 8, 3, 0])
def test_ckpt_to_ply_and_back(tmp_path, k_rest):
    """The reference's ckpt2ply_test: state dict -> ply -> parameters, every property equal."""
    sd = _state_dict(k_rest=k_rest)
    props = F.GaussianProperties.from_state_dict(sd)
    assert props.sh_degree == {15: 3, 8: 2, 3: 1, 0: 0}[k_rest]
    path = str(tmp_path / "a" / "scene.ply")
    props.save_ply(path)
    back = F.GaussianProperties.load_ply(path)
    for k in F.GaussianProperties.NAMES:
        assert torch.equal(sd["gaussian_model.gaussians." + k], getattr(back, k)), k
    # the file itself: header as plyfile writes it, channel-major f_rest, little-endian float32 rows
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().split("\n")
    n = sd["gaussian_model.gaussians.means"].shape[0]
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", f"element vertex {n}"]
    names = [l.split()[-1] for l in lines[3:] if l.startswith("property float ")]
    expect = ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(3 * k_rest)] + ["opacity", "scale_0", "scale_1", "scale_2",
                                                                                                          "rot_0", "rot_1", "rot_2", "rot_3"]
    assert names == expect and len(body) == n * 4 * len(expect)
    rows = np.frombuffer(body, "<f4").reshape(n, len(expect))
    rest = sd["gaussian_model.gaussians.shs_rest"].numpy()
    if k_rest:
        assert np.array_equal(rows[:, 6 + 1], rest[:, 1, 0])                    # f_rest_1 = channel 0, coefficient 1
        assert np.array_equal(rows[:, 6 + k_rest], rest[:, 0, 1])               # f_rest_K = channel 1, coefficient 0
    # old-style checkpoints (gaussian_utils.py:141-159)
    old = {"gaussian_model._xyz": props.means, "gaussian_model._features_dc": props.shs_dc, "gaussian_model._features_rest": props.shs_rest,
           "gaussian_model._opacity": props.opacities, "gaussian_model._scaling": props.scales, "gaussian_model._rotation": props.rotations}
    assert torch.equal(F.GaussianProperties.from_state_dict(old).shs_rest, props.shs_rest)


def test_ply_with_colors_and_ascii(tmp_path):
    props = F.GaussianProperties.from_state_dict(_state_dict(n=33))
    p = str(tmp_path / "c.ply")
    props.save_ply(p, with_colors=True)
    v = F.read_ply_vertices(p)
    assert v.dtype.names[-3:] == ("red", "green", "blue") and v["red"].dtype == np.uint8
    assert torch.equal(F.GaussianProperties.load_ply(p).means, props.means)
    # an ascii file of the same element loads too
    q = str(tmp_path / "ascii.ply")
    names = [n for n in v.dtype.names if n not in ("red", "green", "blue")]
    with open(q, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment test\nelement vertex %d\n" % v.shape[0])
        f.write("".join(f"property float {n}\n" for n in names) + "end_header\n")
        for row in v:
            f.write(" ".join(repr(float(row[n])) for n in names) + "\n")
    a = F.GaussianProperties.load_ply(q)
    assert torch.allclose(a.shs_rest, props.shs_rest) and torch.allclose(a.rotations, props.rotations)
    with pytest.raises(ValueError):
        F.GaussianProperties.load_ply(p, sh_degree=2)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_ROOT, "internal", "utils", "gaussian_utils.py")), reason="reference tree not present")
def test_interoperates_with_the_reference_ply_utils(tmp_path):
    """The reference's own GaussianPlyUtils (imported with a `plyfile` stand-in built on this codec, plyfile itself is not
    installed here) writes files this module reads and reads files this module writes: attribute naming, ordering and the
    channel-major transposes agree."""
    class _Prop:
        def __init__(self, name): self.name = name

    class PlyElement:
        def __init__(self, data): self.data = data
        @staticmethod
        def describe(data, name): assert name == "vertex"; return PlyElement(data)
        @property
        def properties(self): return [_Prop(n) for n in self.data.dtype.names]
        def __getitem__(self, k): return self.data[k]

    class PlyData:
        def __init__(self, elements): self.elements = elements
        def write(self, path): F.write_ply_vertices(path, self.elements[0].data)
        @staticmethod
        def read(path): return PlyData([PlyElement(F.read_ply_vertices(path))])

    stub = types.ModuleType("plyfile")
    stub.PlyData, stub.PlyElement = PlyData, PlyElement
    saved = sys.modules.get("plyfile")
    sys.modules["plyfile"] = stub
    sys.path.insert(0, REF_ROOT)
    try:
        from internal.utils.gaussian_utils import GaussianPlyUtils
        sd = _state_dict(n=101)
        ref_path, my_path = str(tmp_path / "ref.ply"), str(tmp_path / "mine.ply")
        GaussianPlyUtils.load_from_state_dict(sd).to_ply_format().save_to_ply(ref_path)
        mine = F.GaussianProperties.load_ply(ref_path)
        for k in F.GaussianProperties.NAMES:
            assert torch.equal(sd["gaussian_model.gaussians." + k], getattr(mine, k)), k
        F.GaussianProperties.from_state_dict(sd).save_ply(my_path)
        assert open(ref_path, "rb").read() == open(my_path, "rb").read()          # byte-identical files
        theirs = GaussianPlyUtils.load_from_ply(my_path).to_parameter_structure()
        assert theirs.sh_degrees == 3
        for a, b in (("means", "xyz"), ("shs_dc", "features_dc"), ("shs_rest", "features_rest"), ("scales", "scales"), ("rotations", "rotations"),
                     ("opacities", "opacities")):
            assert torch.equal(sd["gaussian_model.gaussians." + a], getattr(theirs, b)), a
    finally:
        sys.path.remove(REF_ROOT)
        if saved is None:
            del sys.modules["plyfile"]
        else:
            sys.modules["plyfile"] = saved


def test_merge_rank_checkpoints(tmp_path):
    """utils/merge_distributed_ckpts.py: Gaussian properties, density-controller state and Adam moments concatenated in rank order."""
    names = ["means", "shs_dc", "shs_rest", "opacities", "scales", "rotations"]

    def rank_ckpt(rank, n):
        sd = _state_dict(n=n, seed=rank)
        sd["density_controller.max_radii2D"] = torch.full((n,), float(rank))
        sd["density_controller.denom"] = torch.full((n, 1), float(rank))
        sd["renderer.some_weight"] = torch.tensor([float(rank)])
        groups = [{"name": k, "lr": 1e-3} for k in names]
        state = {i: {"step": torch.tensor(7.0), "exp_avg": torch.full_like(sd["gaussian_model.gaussians." + k], rank + 0.5),
                     "exp_avg_sq": torch.full_like(sd["gaussian_model.gaussians." + k], rank + 0.25)} for i, k in enumerate(names)}
        return {"state_dict": sd, "optimizer_states": [{"param_groups": groups, "state": state}], "global_step": 100}

    d = tmp_path / "checkpoints"
    d.mkdir()
    sizes = [5, 9, 3]
    for r, n in enumerate(sizes):
        torch.save(rank_ckpt(r, n), str(d / f"epoch=3-step=100-rank={r}.ckpt"))
    torch.save(rank_ckpt(0, 2), str(d / "epoch=1-step=50-rank=0.ckpt"))           # an older step is ignored
    files = F.find_rank_checkpoints(str(d))
    assert [os.path.basename(f) for f in files] == [f"epoch=3-step=100-rank={r}.ckpt" for r in range(3)]
    F.main(["merge", str(tmp_path)])
    merged = torch.load(str(d / "epoch=3-step=100.ckpt"), map_location="cpu", weights_only=False)
    sd = merged["state_dict"]
    assert sd["gaussian_model.gaussians.means"].shape[0] == sum(sizes)
    expect = torch.cat([_state_dict(n=n, seed=r)["gaussian_model.gaussians.shs_rest"] for r, n in enumerate(sizes)])
    assert torch.equal(sd["gaussian_model.gaussians.shs_rest"], expect)
    assert torch.equal(sd["density_controller.max_radii2D"], torch.cat([torch.full((n,), float(r)) for r, n in enumerate(sizes)]))
    st = merged["optimizer_states"][0]["state"]
    assert st[0]["exp_avg"].shape == (sum(sizes), 3) and float(st[0]["exp_avg"][0, 0]) == 0.5 and float(st[0]["exp_avg"][-1, 0]) == 2.5
    assert st[2]["exp_avg_sq"].shape == (sum(sizes), 15, 3) and float(st[0]["step"]) == 7.0
    assert float(sd["renderer.some_weight"]) == 2.0                                 # not sharded: the last rank's
    # and the merged checkpoint converts to a ply
    F.main(["ckpt2ply", str(d / "epoch=3-step=100.ckpt"), str(tmp_path / "m.ply")])
    assert F.GaussianProperties.load_ply(str(tmp_path / "m.ply")).means.shape[0] == sum(sizes)


def test_merge_rank_checkpoints_renderer_swap_first_match_and_no_mutation():
    """the merged checkpoint names the NON-distributed renderer with the distributed one's options, the DDP-wrapped
    appearance keys are renamed, only the first optimizer group named after a property is merged, the inputs stay untouched, and the
    pre-property-dict layout is rejected."""
    import copy
    import gspl_amd  # noqa: F401
    from gspl_amd.renderers import HipGSplatDistributedRenderer, HipGSplatV1Renderer

    def rank_ckpt(rank, n):
        sd = _state_dict(n=n, seed=rank)
        sd["renderer.appearance_model.module.w"] = torch.tensor([float(rank)])
        g1 = {"param_groups": [{"name": "means"}, {"name": "opacities"}],
              "state": {0: {"exp_avg": torch.full((n, 3), float(rank)), "exp_avg_sq": torch.full((n, 3), 1.0 + rank)},
                        1: {"exp_avg": torch.full((n, 1), float(rank)), "exp_avg_sq": torch.full((n, 1), 1.0 + rank)}}}
        g2 = {"param_groups": [{"name": "means"}],            # another optimizer that reuses the name: must be left alone
              "state": {0: {"exp_avg": torch.full((4,), 9.0), "exp_avg_sq": torch.full((4,), 9.0)}}}
        return {"state_dict": sd, "optimizer_states": [g1, g2],
                "hyper_parameters": {"renderer": HipGSplatDistributedRenderer(block_size=16, anti_aliased=False, filter_2d_kernel_size=0.1,
                                                                              tile_based_culling=True), "other": 5}}
    ckpts = [rank_ckpt(0, 4), rank_ckpt(1, 6)]
    before = copy.deepcopy(ckpts)
    merged = F.merge_rank_checkpoints(ckpts)
    r = merged["hyper_parameters"]["renderer"]
    assert isinstance(r, HipGSplatV1Renderer) and (r.block_size, r.anti_aliased, r.filter_2d_kernel_size, r.tile_based_culling) == (16, False, 0.1, True)
    assert merged["hyper_parameters"]["other"] == 5
    sd = merged["state_dict"]
    assert "renderer.model.w" in sd and not any(k.startswith("renderer.appearance_model.module.") for k in sd)
    assert sd["gaussian_model.gaussians.means"].shape[0] == 10
    o1, o2 = merged["optimizer_states"]
    assert o1["state"][0]["exp_avg"].shape == (10, 3) and o1["state"][1]["exp_avg_sq"].shape == (10, 1)
    assert o2["state"][0]["exp_avg"].shape == (4,)                                    # second match of `means`: untouched
    for a, b in zip(ckpts, before):                                                   # inputs not modified
        assert a["state_dict"].keys() == b["state_dict"].keys()
        assert all(torch.equal(a["state_dict"][k], b["state_dict"][k]) for k in b["state_dict"])
        assert a["optimizer_states"][0]["state"][0]["exp_avg"].shape == b["optimizer_states"][0]["state"][0]["exp_avg"].shape
        assert isinstance(a["hyper_parameters"]["renderer"], HipGSplatDistributedRenderer)
    with pytest.raises(ValueError, match="pre-property-dict"):
        F.merge_rank_checkpoints([{"state_dict": {"gaussian_model._xyz": torch.zeros(3, 3)}, "gaussian_model_extra_state_dict": {}}])
