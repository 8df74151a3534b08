#!/bin/bash
out=gpurun_out/r03c; mkdir -p $out
tools/micro/lds_atomic_bench.bin > $out/lds_atomic_bench.txt 2>&1; cat $out/lds_atomic_bench.txt
tools/ab_kernel.sh 2 acc1 abl -- --no-renderer-only > $out/ab.txt 2>&1; cat $out/ab.txt
