#!/bin/bash
# The ONE scratch script of a gpurun call (rewritten per call; outputs under gpurun_out/<tag>/, the keepers are copied to profiles/).
tag=${1:-r10a}
cd /root/repo
O=gpurun_out/$tag; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_sort.py tests/test_metric_point_parity.py tests/test_renderers_gpu.py tests/test_sharded_fused_host.py -q -m gpu -x > $O/${tag}_tests.txt 2>&1; grep -E "passed|failed|^FAILED|Error|assert " $O/${tag}_tests.txt | cut -c1-300 | head -20
timeout 300 python bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err; cut -c1-400 $O/${tag}_bench.json
timeout 300 python bench.py --workload S-1080p-6M --steps 10 --warmup 3 > $O/${tag}_bench_6M.json 2>> $O/${tag}_bench.err; cut -c1-300 $O/${tag}_bench_6M.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_x
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python /root/repo/bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-renderer-only --no-stage-rooflines --no-workload-stats --loop none > /tmp/log_x.txt 2>&1
f=$(find /tmp/prof_x -name "*kernel_trace.csv" | head -1); python /root/repo/tools/prof_summary.py seq $f composite_fwd /root/repo/$O/${tag}_sequence.txt > /dev/null
tail -36 /root/repo/$O/${tag}_sequence.txt
