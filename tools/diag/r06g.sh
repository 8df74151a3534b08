#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --share-device --dist-backend gloo --parallelism sharded --no-cpu-baseline --loop none > $O/r06g_stdout.txt 2> $O/r06g_stderr.txt
echo "rc=$?"; echo "--- stdout lines:"; wc -l $O/r06g_stdout.txt; cut -c1-300 $O/r06g_stdout.txt | tail -3; echo "--- stderr tail:"; tail -12 $O/r06g_stderr.txt | cut -c1-300
