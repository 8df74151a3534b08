"""Import shim: exposes the hyphen-named directory `gaussian-splatting-lightning_amd/` as package `gspl_amd`.

    import gspl_amd                      # -> gaussian-splatting-lightning_amd/__init__.py
    from gspl_amd import ops             # -> gaussian-splatting-lightning_amd/ops/
    --model.renderer gspl_amd.renderers.HipGSplatRenderer   (reference CLI)
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gaussian-splatting-lightning_amd")
_spec = importlib.util.spec_from_file_location("gspl_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gspl_amd"] = _mod
_spec.loader.exec_module(_mod)
