#!/bin/bash
# The ONE scratch script of a gpurun call (rewritten per call; outputs under gpurun_out/<tag>/, the keepers are copied to profiles/).
tag=${1:-r07t}
cd /root/repo
O=gpurun_out/$tag; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/${tag}_gpu_suite.txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $O/${tag}_gpu_suite.txt | cut -c1-300
timeout 300 python __graft_entry__.py > $O/${tag}_smoke.txt 2>&1; tail -1 $O/${tag}_smoke.txt | cut -c1-400
