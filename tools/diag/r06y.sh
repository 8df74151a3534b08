#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
nproc; uptime; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
for i in 1 2; do
python bench.py 2>/dev/null | tail -1 > $O/r06y_bench_$i.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r06y_bench_driver_form_$i.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06y_bench*.json")):
    d = json.loads(open(f).read())
    sr = d["stage_rooflines"]
    print(f.split("/")[-1], "value", d["value"], "ms/step", d["ms_per_step"], "bwd", d["roofline"]["avg_ms"], "binning", sr["binning"]["ms"], "loss", sr["loss_fwd_bwd"]["ms"], "adam", sr["adam"]["ms"],
          "loop p50", d["reference_shaped_loop"]["ms_per_step_between_events_p50"])
PY
uptime
