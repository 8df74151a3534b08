#!/bin/bash
# The ONE scratch script of a gpurun call (rewritten per call; outputs under gpurun_out/<tag>/, the keepers are copied to profiles/).
tag=${1:-r15}
cd /root/repo
O=gpurun_out/$tag; mkdir -p $O
timeout 500 bash tools/collect_profiles_r06.sh $tag > $O/${tag}_collect.log 2>&1; tail -3 $O/${tag}_collect.log
tail -30 $O/${tag}_sequence.txt
