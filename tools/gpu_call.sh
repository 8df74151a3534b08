#!/bin/bash
# The ONE scratch script of a gpurun call (rewritten per call; outputs under gpurun_out/<tag>/, the keepers are copied to profiles/).
tag=${1:-r11c}
cd /root/repo
O=gpurun_out/$tag; mkdir -p $O
{
for e in 1 0 1 0 1 0; do
  echo "== GSPL_BENCH_GC_EARLY=$e --steps 20 --warmup 5"
  GSPL_BENCH_GC_EARLY=$e timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --loop none --no-renderer-only --no-stage-rooflines 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['step_ms'], d['instrumented_pass'])"
done
for e in 1 0; do
echo "== GSPL_BENCH_GC_EARLY=$e --steps 200 --warmup 20"
GSPL_BENCH_GC_EARLY=$e timeout 120 python bench.py --no-cpu-baseline --loop none --no-renderer-only --no-stage-rooflines 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['step_ms'], d['instrumented_pass'])"
done
} > $O/${tag}_gc_early_ab.txt 2>&1
cat $O/${tag}_gc_early_ab.txt
