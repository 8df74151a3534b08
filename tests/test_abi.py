"""No-GPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/gspl_hip.h declares, the ctypes table covers exactly that set, and the product path fails
loudly instead of falling back (missing library, CPU tensors)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gspl_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gspl_\w+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    import gspl_amd  # noqa: F401
    from gspl_amd import _lib
    declared = _declared()
    assert len(declared) >= 14
    assert os.path.exists(_lib.LIB_PATH), "build the extension first: python -c 'import __graft_entry__ as g; g.build()'"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in gspl_hip.h but not exported"
    assert declared == _lib.exported_symbols(), "ctypes signature table out of sync with the header"
    assert _lib.lib().gspl_abi_version() == _lib.ABI_VERSION
    assert _lib.lib().gspl_last_error() is not None


def test_no_cpu_fallback():
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.spherical_harmonics(0, torch.zeros(4, 3), torch.zeros(4, 1, 3))
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.project_gaussians(torch.zeros(4, 3), torch.ones(4, 3), 1.0, torch.ones(4, 4), torch.eye(4), 100.0, 100.0, 50.0, 50.0, 100, 100, 16)


def test_missing_library_fails_loudly(monkeypatch):
    import gspl_amd  # noqa: F401
    from gspl_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgspl_hip.so")
    with pytest.raises(_lib.HipLibraryError, match="only compute path"):
        _lib.lib()


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "gaussian-splatting-lightning_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text and "gsplat_oracle" not in text, f


def test_renderer_plugins_importable_and_picklable():
    import pickle
    import gspl_amd  # noqa: F401
    from gspl_amd.renderers import HipGSplatRenderer, HipGSplatV1Renderer, HipVanillaRenderer, Renderer
    cfg = HipGSplatV1Renderer(block_size=16, anti_aliased=False)
    assert pickle.loads(pickle.dumps(cfg)) == cfg           # stored in checkpoint hparams by the reference
    mod = cfg.instantiate()
    assert isinstance(mod, Renderer) and isinstance(HipVanillaRenderer(), Renderer) and isinstance(HipGSplatRenderer(), Renderer)
    assert "rgb" in mod.get_available_outputs() and mod.parse_render_types(["rgb", "alpha"]) == 1 | 2 | 4
    from gspl_amd.renderers import GSplatV1
    culling = HipGSplatV1Renderer(tile_based_culling=True).instantiate()
    assert culling.isect_encode == GSplatV1.isect_encode_tile_based_culling and mod.isect_encode == GSplatV1.isect_encode_lists_only


def test_v1_renderer_viewer_tab_writes_the_runtime_options():
    """`setup_web_viewer_tabs` (Renderer plugin API, reference gsplat_v1_renderer.py:350-352,615-661) against a stand-in for the
    viser server: the two controls exist with the reference's labels and ranges, and their callbacks update `runtime_options`
    and ask for a re-render."""
    import contextlib
    import gspl_amd  # noqa: F401
    from gspl_amd.renderers import HipGSplatV1Renderer

    class Control:
        def __init__(self, **kw):
            self.kw, self.value, self.cb = kw, kw["initial_value"], None

        def on_update(self, fn):
            self.cb = fn
            return fn

    class Gui:
        def __init__(self):
            self.controls = {}

        def add_number(self, **kw):
            self.controls[kw["label"]] = Control(**kw)
            return self.controls[kw["label"]]

        def add_dropdown(self, **kw):
            self.controls[kw["label"]] = Control(**kw)
            return self.controls[kw["label"]]

    class Server:
        gui = Gui()

    class Tabs:
        names = []

        def add_tab(self, name):
            self.names.append(name)
            return contextlib.nullcontext()

    class Viewer:
        rerenders = 0

        def rerender_for_all_client(self):
            Viewer.rerenders += 1

    renderer = HipGSplatV1Renderer().instantiate()
    renderer.setup_web_viewer_tabs(Viewer(), Server(), Tabs())
    assert Tabs.names == ["gsplat"]
    clip, model = Server.gui.controls["Radius Clip"], Server.gui.controls["Camera Model"]
    assert clip.kw["min"] == 0. and clip.kw["max"] == 65535. and model.kw["options"] == ["pinhole", "ortho", "fisheye"]
    clip.value = 2.5
    clip.cb(None)
    model.value = "fisheye"
    model.cb(None)
    assert renderer.runtime_options.radius_clip == 2.5 and renderer.runtime_options.camera_model == "fisheye" and Viewer.rerenders == 2


def test_camera_scalars_are_read_once_per_camera_object_and_follow_changes():
    """`camera_scalars` keeps the python values of a camera's 0-d tensor fields on the camera object (the reference reads them
    with `.item()` on every call); a field replaced or modified in place is read again."""
    import types
    import gspl_amd  # noqa: F401
    from gspl_amd.renderers.renderer import camera_hw, camera_scalars
    cam = types.SimpleNamespace(width=torch.tensor(640, dtype=torch.int32), height=torch.tensor(480, dtype=torch.int32),
                                fov_x=torch.tensor(0.75), idx=torch.tensor(7, dtype=torch.int32), plain=3)
    assert camera_hw(cam) == (640, 480) and all(isinstance(v, int) for v in camera_hw(cam))
    fov, idx, plain = camera_scalars(cam, ("fov_x", "idx", "plain"))
    assert fov == float(torch.tensor(0.75)) and idx == 7 and isinstance(idx, int) and plain == 3
    reads = {"n": 0}
    real_item = torch.Tensor.item

    def counting_item(self):
        reads["n"] += 1
        return real_item(self)
    torch.Tensor.item = counting_item
    try:
        assert camera_hw(cam) == (640, 480) and camera_scalars(cam, ("fov_x", "idx")) == (fov, 7)
        assert reads["n"] == 0                              # served from the object
        cam.width.fill_(800)                                # modified in place: version counter moves
        assert camera_hw(cam) == (800, 480) and reads["n"] == 1
        cam.height = torch.tensor(600, dtype=torch.int32)   # replaced
        assert camera_hw(cam) == (800, 600) and reads["n"] == 2
        assert camera_hw(cam) == (800, 600) and reads["n"] == 2
    finally:
        torch.Tensor.item = real_item

    class Slotted:                                          # an object that takes no new attributes: read every time, still right
        __slots__ = ("width", "height")
    s = Slotted()
    s.width, s.height = torch.tensor(32), torch.tensor(16)
    assert camera_hw(s) == (32, 16) and camera_hw(s) == (32, 16)


def test_v1_camera_constants_are_cached_per_camera_and_follow_edits():
    """`GSplatV1.preprocess_camera` keeps (view matrix, K) on the camera object; an in-place change of a source field, a replaced
    field, or a caller that edits the returned tensors in place all lead to a rebuild."""
    import types
    import gspl_amd  # noqa: F401
    from gspl_amd.renderers import GSplatV1
    cam = types.SimpleNamespace(world_to_camera=torch.eye(4), fx=torch.tensor(100.), fy=torch.tensor(101.), cx=torch.tensor(50.),
                                cy=torch.tensor(40.), width=torch.tensor(100), height=torch.tensor(80))
    a = GSplatV1.preprocess_camera(cam)
    b = GSplatV1.preprocess_camera(cam)
    assert a[0] is b[0] and a[1] is b[1] and a[2] == (100, 80)
    assert torch.equal(a[0][0], cam.world_to_camera.T) and float(a[1][0, 1, 1]) == 101.0 and float(a[1][0, 2, 2]) == 1.0
    b[1][0, 0, 2] -= 5.0                                     # a caller edits K in place
    c = GSplatV1.preprocess_camera(cam)
    assert c[1] is not b[1] and float(c[1][0, 0, 2]) == 50.0
    cam.fx.mul_(2)                                           # source modified in place
    assert float(GSplatV1.preprocess_camera(cam)[1][0, 0, 0]) == 200.0
    cam.world_to_camera = torch.eye(4) * 2                   # source replaced
    assert float(GSplatV1.preprocess_camera(cam)[0][0, 0, 0]) == 2.0
    pose = torch.eye(4, requires_grad=True)                  # a pose being optimised is never cached
    cam.world_to_camera = pose
    v = GSplatV1.preprocess_camera(cam)[0]
    assert v.requires_grad and GSplatV1.preprocess_camera(cam)[0] is not v
