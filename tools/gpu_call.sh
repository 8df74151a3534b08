#!/bin/bash
# The ONE scratch script of a gpurun call (rewritten per call; outputs under gpurun_out/<tag>/, the keepers are copied to profiles/).
tag=${1:-r09b}
cd /root/repo
O=gpurun_out/$tag; mkdir -p $O
timeout 900 python -m pytest tests/test_bench_loop.py tests/test_segmented_backward.py tests/test_bench_contract.py -q -m gpu > $O/${tag}_tests.txt 2>&1; grep -E "passed|failed|^FAILED|Error|assert " $O/${tag}_tests.txt | cut -c1-300 | head -20
