#!/bin/bash
# call w: the list length from the key pass (early ticket) — fused-path tests, A/B bench lines and host time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_locked_parity.py tests/test_renderers_gpu.py tests/test_training_loop.py tests/test_bench_loop.py -x -q -m gpu 2>&1 | tail -3
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --loop none --no-stage-rooflines --no-workload-stats --no-renderer-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 ms/step', d['ms_per_step'], 'p50', d['step_ms']['p50'])"; }
for rep in 1 2 3; do
  GSPL_EARLY_LENGTH=0 run "early_length=0"
  GSPL_EARLY_LENGTH=1 run "early_length=1"
done
GSPL_EARLY_LENGTH=0 python tools/micro/host_step.py 300 0 2>&1 | tail -2
GSPL_EARLY_LENGTH=1 python tools/micro/host_step.py 300 0 2>&1 | tail -2
# with a busy host: 64 spinning processes beside the bench
for i in $(seq 64); do (timeout 60 python -c "while True: pass" &) ; done
sleep 1
GSPL_EARLY_LENGTH=0 run "busy host, early_length=0"
GSPL_EARLY_LENGTH=1 run "busy host, early_length=1"
GSPL_EARLY_LENGTH=0 run "busy host, early_length=0"
GSPL_EARLY_LENGTH=1 run "busy host, early_length=1"
