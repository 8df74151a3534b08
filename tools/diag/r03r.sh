#!/bin/bash
out=gpurun_out/r03r; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python /root/repo/bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-renderer-only --no-stage-rooflines --no-workload-stats > /tmp/log.txt 2>&1
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); python /root/repo/tools/prof_summary.py seq $f composite_fwd /root/repo/$out/sequence.txt > /dev/null; cat /root/repo/$out/sequence.txt
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); python /root/repo/tools/prof_summary.py stats $f 64 /root/repo/$out/kernel_stats.csv > /dev/null
