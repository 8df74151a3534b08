#!/bin/bash
# one gpurun lease: parity of the new backward kernel, A/B against the old one, and the padded-exchange loop
out=gpurun_out/r03b; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_metric_point_parity.py tests/test_renderers_gpu.py tests/test_backward_spread.py tests/test_scores.py tests/test_training_loop.py -q -m gpu -p no:cacheprovider > $out/pytest_bwd4.log 2>&1
tail -25 $out/pytest_bwd4.log
for i in 1 2; do
  for k in 2 4; do
    GSPL_BWD_KERNEL=$k timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-renderer-only 2>/dev/null | tail -1 > $out/bench_k${k}_$i.json
    python - $out/bench_k${k}_$i.json $k <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d.get("roofline") or {}
print("kernel", sys.argv[2], "ms/step", d["ms_per_step"], "bwd avg_ms", r.get("avg_ms"), "frac", r.get("frac"), r.get("kernel"))
PY
  done
done
timeout 600 python tools/diag/padded_loop.py --iters 30 > $out/padded_loop.log 2>&1
grep -c "outside" $out/padded_loop.log; tail -4 $out/padded_loop.log
