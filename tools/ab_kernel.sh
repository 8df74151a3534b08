#!/bin/bash
# usage (on the GPU box): tools/ab_kernel.sh <rounds> <variant name>... [-- bench args]
# Interleaved short bench runs of the in-tree library ("base") and of variants/libgspl_hip_<name>.so; prints, per library, the medians of
# the compositing backward / forward launch durations (HIP events inside the timed steps) and of ms/step.
rounds=$1; shift
names=(base)
while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done
[ "$1" == "--" ] && shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$(mktemp)
for i in $(seq $rounds); do
  for n in "${names[@]}"; do
    lib=$root/gaussian-splatting-lightning_amd/libgspl_hip.so
    [ "$n" != base ] && lib=$root/gaussian-splatting-lightning_amd/variants/libgspl_hip_$n.so
    line=$(GSPL_HIP_LIB=$lib python $root/bench.py --steps 30 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | tail -1)
    echo "$n $line" >> $out
  done
done
python - $out <<'PY'
import sys, json, statistics as st
rows = {}
for l in open(sys.argv[1]):
    n, j = l.split(" ", 1)
    try:
        d = json.loads(j)
    except Exception:
        rows.setdefault(n, []).append(None); continue
    s = d.get("stages_ms", {})
    rows.setdefault(n, []).append((d["roofline"]["avg_ms"] if d.get("roofline") else float("nan"), s.get("gspl_composite_fwd", float("nan")), d["ms_per_step"]))
base = None
for n, v in rows.items():
    ok = [x for x in v if x]
    if not ok:
        print(f"{n:12s} FAILED"); continue
    med = [st.median(x[i] for x in ok) for i in range(3)]
    if base is None: base = med
    print(f"{n:12s} bwd {med[0]:.4f} ms ({med[0]/base[0]:.3f})  fwd {med[1]:.4f} ms ({med[1]/base[1]:.3f})  step {med[2]:.4f} ms ({med[2]/base[2]:.3f})  runs {len(ok)}/{len(v)}")
PY
