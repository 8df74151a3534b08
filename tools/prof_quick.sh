cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-renderer-only "$@" > /tmp/log.txt 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); python /root/repo/tools/prof_summary.py stats $f 30 /root/repo/gpurun_out/kstats.csv | head -40
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); python /root/repo/tools/prof_summary.py seq $f composite_fwd /root/repo/gpurun_out/kseq.txt > /dev/null; head -50 /root/repo/gpurun_out/kseq.txt
