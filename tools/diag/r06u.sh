#!/bin/bash
# early list length under a SLOW HOST: the bench pinned to two cores, one / two spinners on the same cores
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { taskset -c 2,3 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --loop none --no-stage-rooflines --no-workload-stats --no-renderer-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 ms/step', d['ms_per_step'], 'p50', d['step_ms']['p50'], 'p99', d['step_ms']['p99'])"; }
echo "--- pinned to 2 cores, no competition"
GSPL_EARLY_LENGTH=0 run "early_length=0"; GSPL_EARLY_LENGTH=1 run "early_length=1"
for n in 1 2 3; do
  pids=""
  for i in $(seq $n); do taskset -c 2,3 python -c "while True: pass" & pids="$pids $!"; done
  sleep 0.5
  echo "--- pinned to 2 cores, $n spinner(s) on the same cores"
  GSPL_EARLY_LENGTH=0 run "early_length=0"; GSPL_EARLY_LENGTH=1 run "early_length=1"
  GSPL_EARLY_LENGTH=0 run "early_length=0"; GSPL_EARLY_LENGTH=1 run "early_length=1"
  kill $pids
done
