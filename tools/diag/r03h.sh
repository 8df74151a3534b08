#!/bin/bash
out=gpurun_out/r03h; mkdir -p $out
tools/ab_kernel.sh 2 pk4 pk4p64 pk5p64 pk4p40 pk3 -- --no-renderer-only > $out/ab.txt 2>&1; cat $out/ab.txt
