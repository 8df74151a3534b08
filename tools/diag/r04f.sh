#!/bin/bash
# the per-camera pass before the warm-up: the driver's short form and the default form, single and sharded
out=gpurun_out/r04f; mkdir -p $out
show='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get("roofline") or {}; print(sys.argv[1], d["ms_per_step"], d["value"], d["step_ms"]["p50"], d["step_ms"]["p99"], d["allocator"], r.get("frac"), r.get("intersections"), r.get("traffic"))'
for i in 1 2 3; do
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee $out/bench_driver_form_$i.json | python -c "$show" short
done
python bench.py 2>/dev/null | tail -1 | tee $out/bench.json | python -c "$show" default
python bench.py --parallelism sharded --no-cpu-baseline --no-stage-rooflines --steps 20 --warmup 5 2>/dev/null | tail -1 | tee $out/bench_sharded_short.json | python -c "$show" sharded-short
python bench.py --parallelism sharded --no-cpu-baseline --no-stage-rooflines 2>/dev/null | tail -1 | tee $out/bench_sharded.json | python -c "$show" sharded
python bench.py --api gsplat --no-cpu-baseline --no-stage-rooflines 2>/dev/null | tail -1 | tee $out/bench_gsplat.json | python -c "$show" gsplat
