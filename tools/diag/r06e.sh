#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/micro/splat_parallel_ceiling.py 2>&1 | tail -8 | tee gpurun_out/r06e_splat_parallel_ceiling.txt
