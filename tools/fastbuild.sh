#!/bin/bash
# Development build of the DDP kernels only (two-slot instantiations: P <= 12 planes per polytope), ~25 s instead of
# 2.5 min: usage tools/fastbuild.sh <name> [extra hipcc flags]  ->  direct_amd/lib/dev_<name>.so (DIRECT_DDP_LIB=...).
# The cluster / label-model objects are compiled once into build_variants/.  Never the product build (direct_amd/build.py).
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); name=$1; shift
mkdir -p $ROOT/build_variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDDP_WAVES_F32=3 -DDDP_WAVES_F64=3 -mllvm -amdgpu-load-store-vectorizer=0 -Xclang -target-feature -Xclang -load-store-opt -mllvm -amdgpu-sched-strategy=iterative-ilp"
for m in direct_cluster direct_quad; do
  [ -f $ROOT/build_variants/$m.o ] && [ $ROOT/build_variants/$m.o -nt $ROOT/direct_amd/csrc/$m.hip ] || hipcc $F -c $ROOT/direct_amd/csrc/$m.hip -o $ROOT/build_variants/$m.o 2>&1 | grep -v "not a recognized feature" || true
done
hipcc $F -DDDP_DEV_RPL2 "$@" -c $ROOT/direct_amd/csrc/direct_ddp.hip -o $ROOT/build_variants/ddp_$name.o 2>&1 | grep -v "not a recognized feature" || true
hipcc --offload-arch=gfx950 -shared -fPIC $ROOT/build_variants/ddp_$name.o $ROOT/build_variants/direct_cluster.o $ROOT/build_variants/direct_quad.o -o $ROOT/direct_amd/lib/dev_$name.so
echo $ROOT/direct_amd/lib/dev_$name.so
