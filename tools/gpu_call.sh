#!/bin/bash
# The ONE scratch script of a gpurun call (rewritten per call; outputs under gpurun_out/<tag>/, the keepers are copied to profiles/).
tag=${1:-r18}
cd /root/repo
O=gpurun_out/$tag; mkdir -p $O
timeout 200 python -m pytest tests/test_stats_in_backward.py tests/test_density.py tests/test_bench_loop.py -q -m gpu > $O/${tag}_stats_tests.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR|Error" $O/${tag}_stats_tests.txt | cut -c1-300 | head
