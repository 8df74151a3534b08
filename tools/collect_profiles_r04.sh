#!/bin/bash
# usage (on the GPU box): tools/collect_profiles_r04.sh <tag>   -> gpurun_out/<tag>/...
# rocprofv3 kernel stats + last-step sequence of the default bench command and of the sharded step (three autograd nodes, exchange
# auto).  No PMC passes: the compositing kernels' sources and the ABI version are those of profiles/r03b_pmc_* (bench.py checks).
tag=${1:-r04}
cd /root/repo
O=gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
one() {   # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python /root/repo/bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-renderer-only --no-stage-rooflines --no-workload-stats "$@" > /tmp/log_$name.txt 2>&1
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1); python /root/repo/tools/prof_summary.py stats $f 64 /root/repo/$O/${tag}_${name}kernel_stats.csv > /dev/null
  f=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1); python /root/repo/tools/prof_summary.py seq $f composite_fwd /root/repo/$O/${tag}_${name}sequence.txt > /dev/null
  tail -1 /tmp/log_$name.txt > /root/repo/$O/${tag}_${name}bench_under_rocprof.json
}
one ""
one sharded_ --parallelism sharded
head -12 /root/repo/$O/${tag}_kernel_stats.csv
tail -3 /root/repo/$O/${tag}_sharded_sequence.txt
