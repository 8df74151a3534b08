#!/bin/bash
# three-node sharded step: parity tests, then fused vs staged step time at W = 1 (and with every collective issued to RCCL in a
# one-rank group), then the host profile of the fused step
out=gpurun_out/r04a
mkdir -p $out
timeout 420 python -m pytest tests/test_records.py tests/test_renderers_gpu.py tests/test_distributed_renderer.py tests/test_rccl_single_rank.py \
    -x -q -m gpu -k "two_phase or distributed or sharded or rccl" 2>&1 | tail -15 > $out/tests.txt
cat $out/tests.txt
F="--parallelism sharded --no-cpu-baseline --no-stage-rooflines --no-workload-stats"
show='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d["step_ms"]["p50"], d["step_ms"]["p99"])'
for i in 1 2; do
  python bench.py $F 2>/dev/null | tee -a $out/bench_fused.jsonl | python -c "$show" fused
  python bench.py $F --staged-sharded-step 2>/dev/null | tee -a $out/bench_staged.jsonl | python -c "$show" staged
done
python bench.py $F --init-dist 2>/dev/null | tee -a $out/bench_fused_rccl1.jsonl | python -c "$show" fused-rccl1
python bench.py $F --init-dist --staged-sharded-step 2>/dev/null | tee -a $out/bench_staged_rccl1.jsonl | python -c "$show" staged-rccl1
python tools/micro/host_sharded_profile.py 100 2>&1 | grep -v amdgpu.ids | head -75 > $out/host_profile_fused.txt
head -40 $out/host_profile_fused.txt
