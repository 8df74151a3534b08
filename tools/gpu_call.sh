#!/bin/bash
# The ONE scratch script of a gpurun call (rewritten per call; outputs under gpurun_out/<tag>/, the keepers are copied to profiles/).
tag=${1:-r17}
cd /root/repo
O=gpurun_out/$tag; mkdir -p $O
timeout 280 python -m pytest tests/test_bench_contract.py -q -m gpu -k "not two_rank_launch_line" > $O/${tag}_bench_contract.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR|Error" $O/${tag}_bench_contract.txt | cut -c1-300 | head
