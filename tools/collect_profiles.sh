set -x
cd /root/repo
O=gpurun_out/r01f; mkdir -p $O
python bench.py 2>/dev/null | tail -1 > $O/r01f_bench.json
python bench.py --no-cpu-baseline --stage-times 2>/dev/null | tail -1 > $O/r01f_bench_stage_times.json
python bench.py --no-cpu-baseline --api gsplat 2>/dev/null | tail -1 > $O/r01f_bench_gsplat.json
python bench.py --no-cpu-baseline --loss l1 2>/dev/null | tail -1 > $O/r01f_bench_l1.json
python bench.py --no-cpu-baseline --optimizer fused-adam 2>/dev/null | tail -1 > $O/r01f_bench_fused_adam.json
for w in S-800-100k S-1080p-6M S-garden-6M S-4k-2M S-1080p-1M-inside; do python bench.py --no-cpu-baseline --stage-times --workload $w 2>/dev/null | tail -1 >> $O/r01f_bench_other_workloads.jsonl; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/log.txt 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); python /root/repo/tools/prof_summary.py stats $f 25 /root/repo/$O/r01f_kernel_stats.csv > /dev/null
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); python /root/repo/tools/prof_summary.py seq $f composite_fwd /root/repo/$O/r01f_vanilla_sequence.txt > /dev/null
ls -la /root/repo/$O
