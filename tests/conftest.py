"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI symbol export, gloo world_size-2 path.
`-m gpu`       : HIP path vs oracle through the C-ABI (needs an MI355X).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def free_port() -> int:
    """A TCP port nobody listens on right now (rendezvous of the multi-process tests): asked from the kernel per test, so that
    two tests of one session never meet on a port that is still in TIME_WAIT."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / by the driver)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _built_extension():
    """A fresh checkout has no libgspl_hip.so (built artefacts are git-ignored): build it once per session.
    hipcc cross-compiles for gfx950 without a GPU (~30 s cold)."""
    lib = os.path.join(ROOT, "gaussian-splatting-lightning_amd", "libgspl_hip.so")
    if not os.path.exists(lib) and not os.environ.get("GSPL_HIP_LIB"):
        import gspl_amd  # noqa: F401
        from gspl_amd import _lib
        _lib.build()
    yield
