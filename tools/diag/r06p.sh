#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r06_bench_other_workloads.jsonl; : > $O
for w in S-800-100k S-1080p-6M S-garden-6M S-1080p-20M-sh0-absgrad; do
  timeout 900 python bench.py --workload $w --stage-times --no-cpu-baseline --loop none 2>/dev/null | tail -1 >> $O
done
python - <<'PY'
import json
for l in open("gpurun_out/r06_bench_other_workloads.jsonl"):
    try:
        d = json.loads(l); print(d["config"]["workload"], d["ms_per_step"], d["value"], d.get("stages_ms"))
    except Exception as e:
        print("ERR", e, l[:200])
PY
