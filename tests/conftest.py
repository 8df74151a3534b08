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


_PORTS_HANDED_OUT = set()


def free_port() -> int:
    """A TCP port nobody listens on right now (rendezvous of the multi-process tests): a fresh one per test, so that two tests of one
    session never meet on a port that is still in TIME_WAIT — and taken BELOW the kernel's ephemeral range (/proc/sys/net/ipv4/
    ip_local_port_range, 32768+ by default): a port the kernel hands out for `bind(0)` can be given to any other process's outgoing
    connection between this probe and the rendezvous that binds it a moment later."""
    import os
    import random
    import socket
    lo = 32768
    try:
        lo = int(open("/proc/sys/net/ipv4/ip_local_port_range").read().split()[0])
    except (OSError, ValueError, IndexError):
        pass
    top = max(min(lo, 32768), 12000)
    rng = random.Random(os.getpid() * 1000003 + len(_PORTS_HANDED_OUT))
    for _ in range(200):
        port = rng.randrange(10000, top)
        if port in _PORTS_HANDED_OUT:
            continue
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            try:
                s.bind(("127.0.0.1", port))
            except OSError:
                continue
        _PORTS_HANDED_OUT.add(port)
        return port
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:      # (fallback: the kernel's choice)
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def leave_process_group(dist) -> None:
    """The SUCCESS end of a multi-process test worker: a barrier (every rank is past its last collective and its last assertion), then the
    process leaves at once — no collective tear-down, no interpreter finalisation.  Why: with `destroy_process_group()` + a normal exit one
    rank of eight was seen to die of SIGABRT ("terminate called without an active exception": a gloo thread torn down while its peers were
    closing their ends) AFTER every assertion of every rank had passed — once in twenty runs of the world-8 tests.  A worker that FAILS
    still unwinds normally (its traceback is what `mp.spawn` reports)."""
    import os
    import sys
    dist.barrier()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / by the driver)")


# Collection order (the driver runs `pytest -x`): the hot path's HIP-vs-oracle parity first, the functions either side of it next,
# everything that spawns processes or drives whole training loops last — so that a failure in a side feature can never keep the
# hot-path parity tests from running.  Files not listed keep their alphabetical place between the two groups.
_FIRST = ["test_abi", "test_oracle_golden", "test_oracle_composite", "test_hip_parity", "test_metric_point_parity", "test_locked_parity", "test_renderers_gpu",
          "test_tile_sizes", "test_sort", "test_sh_batched", "test_camera_models", "test_backward_spread", "test_loss", "test_knn", "test_scores",
          "test_records", "test_adam", "test_density", "test_formats", "test_upstream_golden"]
_LAST = ["test_training_loop", "test_reference_lightning_loop", "test_package_shims", "test_bench_contract", "test_distributed_gloo", "test_masked_replica", "test_allreduce_step",
         "test_rccl_single_rank", "test_distributed_renderer"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _FIRST:
            return (0, _FIRST.index(name))
        if name in _LAST:
            return (2, _LAST.index(name))
        return (1, 0)
    items.sort(key=rank)        # stable: the order inside a file is kept


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
