#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags...]  -> gaussian-splatting-lightning_amd/variants/libgspl_hip_<name>.so
# A/B builds of the C-ABI library for kernel experiments; select one at run time with GSPL_HIP_LIB=<path>.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/gaussian-splatting-lightning_amd/csrc
out=$root/gaussian-splatting-lightning_amd/variants
mkdir -p $out/obj_$name
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-function -Wno-unused-variable -Wno-pass-failed"
all="projection sh binning composite composite_bwd inria knn loss adam sort density fused records peer"
# ONLY="sort binning" recompiles just those files and takes the other objects from the regular build (csrc/build)
for f in $all; do
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $f "; then cp $src/build/$f.o $out/obj_$name/$f.o; continue; fi
  /opt/rocm/bin/hipcc $flags "$@" -c $src/$f.hip -o $out/obj_$name/$f.o &
done
if [ -n "$ONLY" ]; then cp $src/build/api.o $out/obj_$name/api.o; else /opt/rocm/bin/hipcc $flags "$@" -x hip -c $src/api.cpp -o $out/obj_$name/api.o & fi
wait
for f in $all api; do test -f $out/obj_$name/$f.o || { echo "build failed: $f"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libgspl_hip_$name.so $out/obj_$name/*.o
rm -rf $out/obj_$name
echo $out/libgspl_hip_$name.so
