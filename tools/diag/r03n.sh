#!/bin/bash
out=gpurun_out/r03n; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_locked_parity.py tests/test_renderers_gpu.py tests/test_training_loop.py -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1; tail -4 $out/pytest.log
python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python - $out/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","fwd_ms","bwd_ms","step_ms","speculation")})
for k,v in (d["stage_rooflines"] or {}).items():
    if isinstance(v,dict) and "ms" in v: print("  %-36s %7.4f ms %8.1f GB/s frac %.3f" % (k, v["ms"], v["GBps"], v["frac"]))
PY
