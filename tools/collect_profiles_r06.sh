#!/bin/bash
# usage (on the GPU box): tools/collect_profiles_r06.sh <tag>   -> gpurun_out/<tag>/...  (copy the files worth keeping to profiles/)
# The bench line as the driver runs it, rocprofv3 kernel stats + last-step sequence of the same command, the gsplat-API line, the
# sharded step (W = 1 with every exchange issued) over the collective and over the peer transport with their sequences, and the PMC
# passes (each in its own run) + the traffic file bench.py reads (valid for THIS code state only).
tag=${1:-r07}
cd /root/repo
O=gpurun_out/$tag; mkdir -p $O
b() { python bench.py "$@" 2>/dev/null | tail -1; }
b > $O/${tag}_bench.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/${tag}_bench_driver_form.json
b --no-cpu-baseline --api gsplat --no-stage-rooflines --loop none > $O/${tag}_bench_gsplat.json
b --no-cpu-baseline --loop none --optimizer fused-bwd-adam > $O/${tag}_bench_fused_bwd_adam.json
b --no-cpu-baseline --loop none --workload S-1080p-1M-surfaces > $O/${tag}_bench_surfaces.json
( b --no-cpu-baseline --loop none --stage-times --workload S-1080p-6M --steps 60; b --no-cpu-baseline --loop none --stage-times --workload S-1080p-6M --steps 60 --optimizer fused-bwd-adam; b --no-cpu-baseline --loop none --stage-times --workload S-garden-6M --steps 60; b --no-cpu-baseline --loop none --stage-times --workload S-800-100k ) > $O/${tag}_bench_other_workloads.jsonl
b --no-cpu-baseline --parallelism sharded --no-stage-rooflines > $O/${tag}_bench_sharded_1gpu.json
b --no-cpu-baseline --parallelism sharded --init-dist --exchange padded --exchange-transport collective --no-stage-rooflines > $O/${tag}_bench_sharded_1gpu_collective_issued.json
b --no-cpu-baseline --parallelism sharded --init-dist --exchange padded --exchange-transport peer --no-stage-rooflines > $O/${tag}_bench_sharded_1gpu_peer_issued.json
cd /tmp && export TMPDIR=/tmp
one() {   # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python /root/repo/bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-renderer-only --no-stage-rooflines --no-workload-stats --loop none "$@" > /tmp/log_$name.txt 2>&1
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1); python /root/repo/tools/prof_summary.py stats $f 112 /root/repo/$O/${tag}_${name}kernel_stats.csv > /dev/null
  # step 41 of the run: inside the 48 timed steps (16 warm-up steps before them; the steps past 64 are the instrumented pass, three event
  # records each), and not one of the every-fourth steps whose graded launch the roofline brackets with an event pair (40, 44, ...)
  f=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1); python /root/repo/tools/prof_summary.py seq $f composite_fwd /root/repo/$O/${tag}_${name}sequence.txt 41 > /dev/null
  python /root/repo/tools/prof_summary.py seq $f composite_fwd /root/repo/$O/${tag}_${name}sequence_instrumented_pass.txt > /dev/null
}
one ""
one fused_bwd_adam_ --optimizer fused-bwd-adam
one surfaces_ --workload S-1080p-1M-surfaces
one sharded_collective_ --parallelism sharded --init-dist --exchange padded --exchange-transport collective
one sharded_peer_ --parallelism sharded --init-dist --exchange padded --exchange-transport peer
pmc() {    # name, counters...
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python /root/repo/bench.py --steps 16 --warmup 16 --no-cpu-baseline --no-renderer-only --no-stage-rooflines --no-workload-stats --loop none > /tmp/logp_$name.txt 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1); python /root/repo/tools/prof_summary.py pmc $f /root/repo/$O/${tag}_pmc_$name.csv > /dev/null
}
pmc FETCH_SIZE FETCH_SIZE
pmc WRITE_SIZE WRITE_SIZE
pmc SQ SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
python /root/repo/tools/make_pmc_traffic.py /root/repo/$O $tag S-1080p-1M/vanilla /root/repo/$O/${tag}_pmc_traffic.json
ls /root/repo/$O
