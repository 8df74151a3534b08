"""`renderer.model_raw_parameters`: which models hand the rasterizer their RAW parameters (activations inside the preprocess
kernels) and which keep their getters.  CPU; the kernels' side is tests/test_renderers_gpu.py."""
import os
import sys
import types

import pytest
import torch

REF_ROOT = "/root/reference"


def _imitation_module():
    """A module with the reference's module path and class name, holding what vanilla_gaussian.py:345-358,421-441 define."""
    name = "internal.models.vanilla_gaussian"
    mod = types.ModuleType(name)
    src = '''
import torch
class VanillaGaussianModel:
    def __init__(self, n=5):
        g = torch.Generator().manual_seed(0)
        self.gaussians = {"scales": torch.randn(n, 3, generator=g), "rotations": torch.randn(n, 4, generator=g), "opacities": torch.randn(n, 1, generator=g)}
        self.is_pre_activated = False
    def get_property(self, name): return self.gaussians[name]
    def opacity_activation(self, v): return torch.sigmoid(v)
    def scale_activation(self, v): return torch.exp(v)
    def rotation_activation(self, v): return torch.nn.functional.normalize(v)
    @staticmethod
    def _return_as_is(v): return v
    @property
    def get_scaling(self): return self.scale_activation(self.gaussians["scales"])
    @property
    def get_rotation(self): return self.rotation_activation(self.gaussians["rotations"])
    @property
    def get_opacity(self): return self.opacity_activation(self.gaussians["opacities"])
'''
    exec(compile(src, name, "exec"), mod.__dict__)
    mod.VanillaGaussianModel.__module__ = name
    for v in vars(mod.VanillaGaussianModel).values():
        f = v.fget if isinstance(v, property) else getattr(v, "__func__", v)
        if callable(f):
            f.__module__ = name
    return mod


def test_declared_and_imitated_reference_models():
    import gspl_amd  # noqa: F401
    from gspl_amd.renderers.renderer import model_raw_parameters
    mod = _imitation_module()
    Model = mod.VanillaGaussianModel
    m = Model()
    raw = model_raw_parameters(m)
    assert raw is not None and raw[0] is m.gaussians["scales"] and raw[1] is m.gaussians["rotations"] and raw[2] is m.gaussians["opacities"]

    # pre-activated: identities installed on the instance, flag set (vanilla_gaussian.py:370-400)
    p = Model()
    p.is_pre_activated = True
    assert model_raw_parameters(p) is None
    p = Model()
    p.scale_activation = p._return_as_is
    assert model_raw_parameters(p) is None

    # a subclass that changes an activation or a getter keeps its getters
    class Filtered(Model):
        def scale_activation(self, v): return torch.exp(v) + 0.1
    Filtered.scale_activation.__module__ = "internal.models.mip_splatting"
    assert model_raw_parameters(Filtered()) is None

    class OtherGetter(Model):
        @property
        def get_opacity(self): return torch.sigmoid(self.gaussians["opacities"]) * 0.5
    assert model_raw_parameters(OtherGetter()) is None

    # a subclass that changes neither qualifies
    class Plain(Model):
        pass
    assert model_raw_parameters(Plain()) is not None

    # declared
    class Declared:
        fused_activations = {"scales": "exp", "rotations": "normalize", "opacities": "sigmoid"}
        def __init__(self): self.g = {"scales": torch.zeros(2, 3), "rotations": torch.ones(2, 4), "opacities": torch.zeros(2, 1)}
        def get_property(self, n): return self.g[n]
    assert model_raw_parameters(Declared()) is not None
    class DeclaredOther(Declared):
        fused_activations = {"scales": "softplus", "rotations": "normalize", "opacities": "sigmoid"}
    assert model_raw_parameters(DeclaredOther()) is None

    # anything else: the test fakes store activated values behind plain getters
    from fakes import FakeGaussianModel
    from oracle import gsplat_oracle as O
    assert model_raw_parameters(FakeGaussianModel(*O.synthetic_scene(10, seed=1))) is None


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_ROOT, "internal", "models", "vanilla_gaussian.py")), reason="reference tree not present")
def test_the_reference_model_itself_qualifies_until_it_is_pre_activated():
    if "lightning" not in sys.modules:
        L = types.ModuleType("lightning")
        L.LightningModule = type("LightningModule", (), {})
        sys.modules["lightning"] = L
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    from internal.models.vanilla_gaussian import VanillaGaussian
    import gspl_amd  # noqa: F401
    from gspl_amd.renderers.renderer import model_raw_parameters
    model = VanillaGaussian(sh_degree=1).instantiate()
    model.setup_from_number(7)
    raw = model_raw_parameters(model)
    assert raw is not None
    assert raw[0] is model.gaussians["scales"] and raw[1] is model.gaussians["rotations"] and raw[2] is model.gaussians["opacities"]
    assert torch.equal(model.get_scaling, torch.exp(raw[0])) and torch.equal(model.get_opacity, torch.sigmoid(raw[2]))
    assert torch.equal(model.get_rotation, torch.nn.functional.normalize(raw[1]))
    model.pre_activate_all_properties()
    assert model_raw_parameters(model) is None
