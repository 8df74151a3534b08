#!/bin/bash
# round 4, session 2, call a: tagged lists — parity tests, A/B bench lines, kernel stats of both
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tagged_lists.py tests/test_locked_parity.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -25 > $O/r06a_tests.txt
tail -5 $O/r06a_tests.txt
B="--steps 60 --warmup 10 --no-cpu-baseline --loop none"
for i in 1 2; do
  GSPL_TAGGED_LISTS=0 python bench.py $B 2>/dev/null | tail -1 > $O/r06a_bench_untagged_$i.json
  python bench.py $B 2>/dev/null | tail -1 > $O/r06a_bench_tagged_$i.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06a_bench_*.json")):
    try:
        d = json.loads(open(f).read())
        sr = d.get("stage_rooflines", {})
        print(f.split("/")[-1], "ms/step", d["ms_per_step"], "img/s", d["value"], "bwd avg_ms", d["roofline"].get("avg_ms"),
              {k: v.get("ms") for k, v in sr.items() if isinstance(v, dict)})
    except Exception as e:
        print(f, "ERR", e)
PY
for v in untagged tagged; do
  rm -rf /tmp/prof
  if [ $v = untagged ]; then export GSPL_TAGGED_LISTS=0; else unset GSPL_TAGGED_LISTS; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-renderer-only --loop none --no-stage-rooflines --no-workload-stats > /tmp/log_$v.txt 2>&1)
  f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); python tools/prof_summary.py stats $f 25 $O/r06a_kstats_$v.csv | head -24
  f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); python tools/prof_summary.py seq $f composite_fwd $O/r06a_seq_$v.txt > /dev/null
done
