#!/bin/bash
# usage (on the GPU box): tools/micro/loss_prof.sh <variant name | base> ...   -> average kernel durations of the fused loss (rocprofv3 kernel stats)
root=$(cd "$(dirname "$0")/../.." && pwd)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$root/gaussian-splatting-lightning_amd/variants/libgspl_hip_$v.so
  [ "$v" == base ] && lib=$root/gaussian-splatting-lightning_amd/libgspl_hip.so
  rm -rf /tmp/lp_$v
  GSPL_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp_$v -- python $root/tools/micro/loss_bench.py > /tmp/lp_$v.log 2>&1
  f=$(find /tmp/lp_$v -name "*kernel_stats.csv" | head -1)
  python - "$f" "$v" <<'PY'
import csv, sys
rows = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
def avg(prefix):
    for n, r in rows.items():
        if prefix in n:
            return float(r["AverageNs"]) / 1e3
    return float("nan")
f, b, r = avg("loss_fwd_rows"), avg("loss_bwd_rows"), avg("loss_reduce")
print(f"{sys.argv[2]:14s} fwd_rows {f:6.1f} us  bwd_rows {b:6.1f} us  reduce {r:4.1f} us  sum {f + b + r:6.1f} us")
PY
done
