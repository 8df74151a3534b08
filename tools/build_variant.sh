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
for f in projection sh binning composite inria knn loss adam; do
  /opt/rocm/bin/hipcc $flags "$@" -c $src/$f.hip -o $out/obj_$name/$f.o &
done
/opt/rocm/bin/hipcc $flags "$@" -x hip -c $src/api.cpp -o $out/obj_$name/api.o &
wait
for f in projection sh binning composite inria knn loss adam api; do test -f $out/obj_$name/$f.o || { echo "build failed: $f"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libgspl_hip_$name.so $out/obj_$name/*.o
rm -rf $out/obj_$name
echo $out/libgspl_hip_$name.so
