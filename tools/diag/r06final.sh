#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) > gpurun_out/r06final_gpu_tests.txt 2>&1
tail -6 gpurun_out/r06final_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r06final_smoke.txt
ls /dev/shm | head -3
