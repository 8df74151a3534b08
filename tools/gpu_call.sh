#!/bin/bash
# The ONE scratch script of a gpurun call (rewritten per call; outputs under gpurun_out/<tag>/, the keepers are copied to profiles/).
tag=${1:-r12}
cd /root/repo
O=gpurun_out/$tag; mkdir -p $O
timeout 400 bash tools/collect_profiles_r06.sh $tag > $O/${tag}_collect.log 2>&1; tail -3 $O/${tag}_collect.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/${tag}final_smoke.txt 2>&1; tail -2 $O/${tag}final_smoke.txt
timeout 560 python -m pytest tests -q -m gpu > $O/${tag}final_gpu_suite.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/${tag}final_gpu_suite.txt | cut -c1-300 | head -20
