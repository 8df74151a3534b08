#!/bin/bash
# The ONE scratch script of a gpurun call (rewritten per call; outputs under gpurun_out/<tag>/, the keepers are copied to profiles/).
tag=${1:-r10final}
cd /root/repo
O=gpurun_out/$tag; mkdir -p $O
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/${tag}_smoke.txt 2>&1; tail -3 $O/${tag}_smoke.txt
timeout 640 python -m pytest tests -q -m gpu > $O/${tag}_gpu_suite.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/${tag}_gpu_suite.txt | cut -c1-300 | head -20
