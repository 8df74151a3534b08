"""bench.py's launch contract as the driver uses it: `python bench.py` on one GPU, and `python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N` for several ranks — ONE JSON line from rank 0, last on stdout, with the fields the
driver reads.  The two-rank launch shares the one GPU of the test box and stages its collectives through gloo
(`--share-device --dist-backend gloo`): slow, but the same code path as the RCCL launch apart from the backend."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def _run(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    line = json.loads(lines[-1])                     # the JSON line is the LAST line of stdout
    for k in REQUIRED:
        assert k in line, k
    return line


@pytest.mark.gpu
def test_single_gpu_line():
    line = _run([sys.executable, "bench.py", "--workload", "S-800-100k", "--steps", "5", "--warmup", "2"])
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["warmup"] == 2 and line["unit"] == "images/s"
    assert line["config"]["workload"] == "S-800-100k" and line["config"]["parallelism_mode"] == "single"
    assert abs(line["value"] - 1e3 / line["ms_per_step"]) <= 1e-3 * line["value"]
    assert line["images_per_s_with_optimizer"] == line["value"] and line["images_per_s_renderer_only"] > 0.0      # (5-step regions: no ordering asserted)
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and 0.0 < roof["frac"] and roof["unit"] == "GB/s" and roof["valu_frac"] > 0.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) <= 2e-3
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["value"] > 0.0
    # round 3: the steps cycle through a camera set; the line says how the list-length guesses fared, how the step times spread,
    # and carries a roofline for every stage of the step
    assert line["cameras"]["count"] == 16 and len(line["cameras"]["per_camera"]) == 16
    assert len({c["I"] for c in line["cameras"]["per_camera"]}) > 8                                  # different views, different lists
    spec = line["speculation"]
    assert spec["frames"] == 5 and 0 <= spec["misses"] <= spec["frames"] and spec["miss_rate"] == round(spec["misses"] / spec["frames"], 4)
    assert line["step_ms"]["p50"] > 0 and line["step_ms"]["p99"] >= line["step_ms"]["p50"]
    sr = line["stage_rooflines"]
    for k in ("inria_preprocess_fwd", "sh_fwd_alone", "sh_fwd_overlapped_with_binning", "binning", "composite_fwd", "composite_bwd",
              "inria_preprocess_bwd_with_sh_bwd", "adam"):
        assert sr[k]["ms"] > 0 and sr[k]["bytes"] > 0 and abs(sr[k]["frac"] - sr[k]["GBps"] / 8000.0) <= 2e-4, k
    assert roof["traffic"] is None or "same ABI and kernel sources" in roof["traffic_source"]      # stale PMC files are refused
    # round 4: the timed region carries no per-step events; fwd_ms / bwd_ms / step_ms come from the instrumented pass after it
    assert line["instrumented_pass"]["events_per_step"] == 3 and line["instrumented_pass"]["ms_per_step"] > 0
    assert line["fwd_ms"] > 0 and line["bwd_ms"] > 0 and line["step_ms"]["p50"] > 0 and line["roofline"]["avg_ms"] > 0
    # round 4: the reference-shaped loop (raw parameters, density controller, N changes) rides in the same line
    loop = line["reference_shaped_loop"]
    assert "failed" not in loop, loop
    assert loop["images_per_s_densifying"] > 0 and loop["steps"] == 450 and loop["sh_degree_end"] == 3
    assert sum(1 for e in loop["events"] if "n_after" in e) >= 3 and any(e.get("opacity_reset") for e in loop["events"])
    assert loop["speculation"]["frames"] == 450 and loop["loss_first_last"][1] < loop["loss_first_last"][0]
    # the plugin's default for a raw-parameter model (activations inside the kernels) beside the torch-getter run of the same loop
    other = loop["with_torch_activations"]
    assert "inside the preprocess kernels" in loop["activations"] and "torch getters" in other["activations"]
    # (no ORDER is asserted between the two: at this size both loops wait for the host, and a busy box turns the 15 % around —
    # seen once in nine runs of the suite; the 1 M numbers are the bench line's, profiles/r33_bench.json)
    assert loop["ms_per_step_between_events_p50"] > 0 and other["ms_per_step_between_events_p50"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["default", "replicated", "replicated-masked"])
def test_two_rank_launch_line(mode):
    from conftest import free_port
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "S-800-100k",
           "--share-device", "--dist-backend", "gloo"]
    if mode != "default":
        cmd += ["--parallelism", "replicated"] + (["--optimizer", "masked-adam"] if mode == "replicated-masked" else [])
    line = _run(cmd)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["config"]["parallelism_mode"] == ("sharded" if mode == "default" else "replicated")      # sharded unless told otherwise
    assert ("masked" in line["config"]["parallelism"]) == (mode == "replicated-masked")
    if mode == "default":
        # --exchange-transport auto: the peer transport (records written straight into the other process's IPC-mapped buffer) is taken
        # only after a validation frame reproduced the collective route's image bit for bit on every rank — which two processes on one
        # GPU can do — and the line says which transport the numbers come from
        assert "over the peer transport (validated" in line["config"]["parallelism"], line["config"]["parallelism"]
    assert abs(line["value"] - 2e3 / line["ms_per_step"]) <= 1e-3 * line["value"]               # whole-job images/s
    assert "cpu_baseline" not in line or line["cpu_baseline"] is None                           # rank 0, N = 1 only
    # `frac` is priced on the list entries the launch WALKS — from each tile's deepest blended entry to the head of its list (round 5;
    # round 4 priced the whole culled lists) —, the figures on the culled lists and on the rect intersections beside it
    roof = line["roofline"]
    assert roof["intersections"] >= roof["list_entries"] >= roof["entries_walked"] > 0
    P = line["config"]["width"] * line["config"]["height"]
    assert abs(roof["algorithmic_bytes"] - (76.0 * roof["entries_walked"] + 20.0 * P)) < 1.0
    assert abs(roof["frac"] - roof["algorithmic_bytes"] / (roof["avg_ms"] * 1e-3) / 1e9 / roof["peak"]) <= 2e-3 * roof["frac"] + 1e-5
    on_lists = (76.0 * roof["list_entries"] + 20.0 * P) / (roof["avg_ms"] * 1e-3) / 1e9 / roof["peak"]
    assert abs(roof["frac_on_list_entries"] - on_lists) <= 2e-3 * on_lists + 1e-5 and roof["frac_on_list_entries"] >= roof["frac"]
    on_rects = (76.0 * roof["intersections"] + 20.0 * P) / (roof["avg_ms"] * 1e-3) / 1e9 / roof["peak"]
    assert abs(roof["frac_on_rect_intersections"] - on_rects) <= 2e-3 * on_rects + 1e-5 and roof["frac_on_rect_intersections"] >= roof["frac_on_list_entries"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sharded", "replicated"])
def test_eight_rank_launch_line(mode):
    """The driver's launch line at the world size BASELINE configs[3] / [4] name (VERDICT r5 #1b): eight processes on the one GPU of
    the test box (`--share-device`, collectives staged through gloo), both parallelism modes.  One JSON line, last on stdout, `n_gpus`
    8, whole-job images/s; the sharded mode says which transport passed the validation frame (eight processes mapping each other's
    IPC buffers on one device can run the peer route — between eight devices the same code runs over xGMI)."""
    from conftest import free_port
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), "bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1", "--workload", "S-800-100k",
           "--share-device", "--dist-backend", "gloo", "--parallelism", mode]
    line = _run(cmd)
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["steps"] == 3
    assert line["config"]["parallelism_mode"] == mode
    assert abs(line["value"] - 8e3 / line["ms_per_step"]) <= 1e-3 * line["value"]               # whole-job images/s
    par = line["config"]["parallelism"]
    if mode == "sharded":
        assert "sharded over 8 rank(s), 8 camera(s)/step" in par, par
        # the transport named is the one the timed steps used: peer only after the validation frame on all eight ranks
        assert ("over the peer transport (validated" in par) or ("over the collective transport" in par), par
        assert "over the peer transport (validated" in par, par      # on one device the eight-way IPC mapping works: the peer route must be taken
    else:
        assert "replicated Gaussians, 8 camera(s)/step" in par, par
    assert "cpu_baseline" not in line or line["cpu_baseline"] is None                           # rank 0, N = 1 only
    roof = line["roofline"]
    assert roof["intersections"] >= roof["list_entries"] >= roof["entries_walked"] > 0


def test_cpu_baseline_leg_runs_without_a_gpu():
    """`bench.py --cpu-baseline-only`: the oracle port of one forward + backward pass (and, where the reference tree is present,
    its own projection + SH) timed on the host cores — the `cpu_baseline` object of the bench line, runnable on its own."""
    r = subprocess.run([sys.executable, "bench.py", "--cpu-baseline-only", "--workload", "S-smoke"], cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    cpu = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert cpu["kind"] == "port" and cpu["unit"] == "images/s" and cpu["value"] > 0 and cpu["cores"] >= 1
    assert "S-smoke" in cpu["sample"] and all(v > 0 for v in cpu["ms"].values())
    ref = cpu.get("reference_projection_sh")
    if ref is not None:                   # only where GSPL_REFERENCE_ROOT (default /root/reference) exists
        assert ref["kind"] == "reference" and ref["fwd_ms"] > 0 and ref["bwd_ms"] > 0
