#!/bin/bash
# The ONE scratch script of a gpurun call (rewritten per call; outputs under gpurun_out/<tag>/, the keepers are copied to profiles/).
tag=${1:-r07k}
cd /root/repo
O=gpurun_out/$tag; mkdir -p $O
timeout 600 python -m pytest tests/test_fused_backward_adam.py -q -m gpu > $O/${tag}_fused_adam_tests.txt 2>&1; grep -E "passed|failed|part at step|AssertionError" $O/${tag}_fused_adam_tests.txt | cut -c1-400
b() { timeout 600 python bench.py "$@" 2>>$O/${tag}_bench.err | tail -1; }
b --no-cpu-baseline --loop none --no-stage-rooflines --optimizer fused-bwd-adam > $O/${tag}_bench_fused_bwd_adam.json
b --no-cpu-baseline --loop none --no-stage-rooflines --workload S-1080p-6M --steps 60 --optimizer fused-bwd-adam > $O/${tag}_bench_6M_fused_bwd_adam.json
b --no-cpu-baseline --loop none --no-stage-rooflines --steps 20 --warmup 5 --optimizer fused-bwd-adam > $O/${tag}_bench_fused_driver_form.json
python - <<PY
import json
for n in ("bench_fused_bwd_adam", "bench_6M_fused_bwd_adam", "bench_fused_driver_form"):
    try:
        d = json.load(open("$O/${tag}_%s.json" % n))
        print(n, d["value"], d["ms_per_step"], d["stages_ms"])
    except Exception as e:
        print(n, "failed", e)
PY
