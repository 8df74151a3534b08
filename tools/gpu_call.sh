#!/bin/bash
# The ONE scratch script of a gpurun call (rewritten per call; outputs under gpurun_out/<tag>/, the keepers are copied to profiles/).
tag=${1:-r10b}
cd /root/repo
O=gpurun_out/$tag; mkdir -p $O
tools/ab_sort.sh base ballot base ballot > $O/${tag}_depth_sort_ballot_ranking.txt 2>&1
cat $O/${tag}_depth_sort_ballot_ranking.txt
