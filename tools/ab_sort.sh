#!/bin/bash
# usage (on the GPU box): tools/ab_sort.sh <variant|base|base:ENV=VALUE>...   per-step kernel time of the binning stage (rocprofv3 kernel
# trace of a short bench run) at the metric workload and at S-1080p-6M, for the in-tree library and variants/libgspl_hip_<name>.so
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  n=${spec%%:*}; envs=""; [ "$spec" != "$n" ] && envs=${spec#*:}
  lib=$root/gaussian-splatting-lightning_amd/libgspl_hip.so
  [ "$n" != base ] && lib=$root/gaussian-splatting-lightning_amd/variants/libgspl_hip_$n.so
  for wl in S-1080p-1M S-1080p-6M; do
    rm -rf /tmp/prof_ab
    env $envs GSPL_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-renderer-only --workload $wl > /tmp/log_ab.txt 2>&1
    f=$(find /tmp/prof_ab -name "*kernel_stats.csv" | head -1)
    python - "$f" "$spec" $wl <<'PY'
import csv, sys, re
g = {"radix_count": 0.0, "radix_scatter": 0.0, "scan_": 0.0, "bin_keys": 0.0, "bin_emit": 0.0, "tile_offsets": 0.0}
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    us = float(r["TotalDurationNs"]) / 25 / 1e3
    tot += us
    for k in g:
        if k in r["Name"]: g[k] += us
print("%-28s %-12s binning %7.1f us/step  (%s)  all kernels %7.1f" % (sys.argv[2], sys.argv[3], sum(g.values()), "  ".join("%s %.1f" % (k, v) for k, v in g.items()), tot))
PY
  done
done
