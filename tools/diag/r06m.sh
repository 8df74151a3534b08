#!/bin/bash
# what slows the bench down right behind the test suite?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
b() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --loop none --no-stage-rooflines --no-workload-stats --no-renderer-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: ms/step', d['ms_per_step'], 'p50', d['step_ms']['p50'])"; }
b "fresh box"
timeout 900 python -m pytest tests/test_distributed_renderer.py tests/test_peer_exchange.py tests/test_rccl_single_rank.py tests/test_masked_replica.py tests/test_allreduce_step.py -q -m gpu 2>&1 | tail -1
echo "--- processes after the multi-process tests:"; ps -eo pid,ppid,stat,pcpu,etime,cmd | grep -i "python\|pytest" | grep -v grep | head; ls /dev/shm | head
b "after the multi-process tests"
timeout 900 python -m pytest tests/test_metric_point_parity.py tests/test_locked_parity.py -q -m gpu 2>&1 | tail -1
echo "--- processes after the oracle-heavy tests:"; ps -eo pid,ppid,stat,pcpu,etime,cmd | grep -i "python\|pytest" | grep -v grep | head
b "after the oracle-heavy tests"
sleep 20
b "20 s later"
