#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_distributed_renderer.py tests/test_peer_exchange.py tests/test_rccl_single_rank.py tests/test_distributed_gloo.py -q -m gpu 2>&1 | tail -2
echo "--- /dev/shm after:"; ls /dev/shm | head
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --share-device --dist-backend gloo --parallelism sharded --no-cpu-baseline --loop none > gpurun_out/r06l_stdout.txt 2> gpurun_out/r06l_stderr.txt
echo "rc=$?"; tail -1 gpurun_out/r06l_stdout.txt | cut -c1-200; grep -i "leak\|resource_tracker" gpurun_out/r06l_stderr.txt | head -3; echo "--- /dev/shm after the 2-rank bench:"; ls /dev/shm | head
