#!/bin/bash
# usage (on the GPU box): tools/ab_env.sh <rounds> <ENV_NAME> [-- bench args]
# Interleaved short bench runs with ENV_NAME=0 and ENV_NAME=1 (a process-start switch of the library); prints medians of ms/step.
rounds=$1; name=$2; shift 2
[ "$1" == "--" ] && shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$(mktemp)
for i in $(seq $rounds); do
  for v in 0 1; do
    line=$(env $name=$v python $root/bench.py --steps 30 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | tail -1)
    echo "$v $line" >> $out
  done
done
python - $out $name <<'PY'
import sys, json, statistics as st
rows = {}
for l in open(sys.argv[1]):
    n, j = l.split(" ", 1)
    try: d = json.loads(j)
    except Exception: continue
    rows.setdefault(n, []).append(d["ms_per_step"])
for n, v in sorted(rows.items()):
    print(f"{sys.argv[2]}={n}: step median {st.median(v):.4f} ms  min {min(v):.4f}  max {max(v):.4f}  runs {len(v)}")
PY
