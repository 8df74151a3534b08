#!/bin/bash
# end-of-session verification on a fresh box: whole GPU suite, smoke(), the default bench line and the driver's short form
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > $O/r06z_gpu_tests.txt 2>&1
tail -5 $O/r06z_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/r06z_smoke.txt
( time python bench.py 2>/dev/null | tail -1 > $O/r06z_bench.json ) 2>&1 | grep real
( time python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r06z_bench_driver_form.json ) 2>&1 | grep real
python - <<'PY'
import json
for f in ("r06z_bench.json", "r06z_bench_driver_form.json"):
    d = json.loads(open("gpurun_out/" + f).read())
    loop = d.get("reference_shaped_loop") or {}
    print(f, "value", d["value"], "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("frac", "avg_ms", "traffic")},
          "loop p50", loop.get("ms_per_step_p50_between_events", loop.get("p50_ms")), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
