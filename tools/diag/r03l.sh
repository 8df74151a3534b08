#!/bin/bash
out=gpurun_out/r03l; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.err; python - $out/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","steps","fwd_ms","bwd_ms","step_ms","speculation")})
print("roofline", {k:d["roofline"][k] for k in ("frac","avg_ms","intersections","list_entries","valid_pairs","traffic","traffic_source")})
for k,v in (d["stage_rooflines"] or {}).items():
    if isinstance(v,dict) and "ms" in v: print("  %-36s %7.4f ms %8.1f GB/s frac %.3f" % (k, v["ms"], v["GBps"], v["frac"]))
print("per-camera I:", [c["I"] for c in d["cameras"]["per_camera"]])
print("renderer_only", d["renderer_only"])
PY
timeout 900 python -m pytest tests/test_bench_contract.py tests/test_distributed_renderer.py tests/test_rccl_single_rank.py tests/test_allreduce_step.py -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1; tail -5 $out/pytest.log
