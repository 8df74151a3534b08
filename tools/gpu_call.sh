#!/bin/bash
# The ONE scratch script of a gpurun call (rewritten per call; outputs under gpurun_out/<tag>/, the keepers are copied to profiles/).
tag=${1:-r09}
cd /root/repo
bash tools/collect_profiles_r06.sh $tag 2>&1 | tail -3
python - <<PY
import json
O="gpurun_out/$tag"
for n in ("bench", "bench_driver_form", "bench_fused_bwd_adam", "bench_surfaces", "bench_gsplat", "bench_sharded_1gpu", "bench_sharded_1gpu_collective_issued", "bench_sharded_1gpu_peer_issued"):
    try:
        d = json.load(open(f"{O}/${tag}_{n}.json"))
        r = d.get("roofline") or {}
        print(n, d["value"], d["ms_per_step"], "bwd", r.get("avg_ms"), "frac", r.get("frac"), r.get("frac_on_list_entries"), (d.get("with_fused_bwd_adam") or {}).get("ms_per_step"), r.get("traffic"))
    except Exception as e:
        print(n, "failed", e)
for line in open(f"{O}/${tag}_bench_other_workloads.jsonl"):
    try:
        d = json.loads(line); print(d["config"]["workload"], d["config"]["optimizer"], d["value"], d["ms_per_step"])
    except Exception as e:
        print("other failed", e)
PY
timeout 600 python -m pytest tests/test_segmented_backward.py tests/test_locked_parity.py -q -m gpu > gpurun_out/$tag/${tag}_seg_locked_tests.txt 2>&1; grep -E "passed|failed|^FAILED" gpurun_out/$tag/${tag}_seg_locked_tests.txt | cut -c1-200
