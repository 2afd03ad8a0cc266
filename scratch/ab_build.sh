#!/bin/bash
# A/B build of the library: scratch/ab_build.sh <name> <file.hip> <extra hipcc flags...>  ->  scratch/libdeepclr_<name>.so
# (the other objects come from the last regular build; select with DCLR_LIB=scratch/libdeepclr_<name>.so)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../deepclr_amd/csrc"
base=${src%.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Rpass-analysis=kernel-resource-usage "$@" -c $src -o /tmp/ab_${name}_$base.o 2> /tmp/ab_${name}_$base.log || { tail -20 /tmp/ab_${name}_$base.log; exit 1; }
objs=""
for o in api fps grouping knn sa gemm flow gemm16 flow16 prep forward; do
  if [ $o = $base ]; then objs="$objs /tmp/ab_${name}_$base.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libdeepclr_$name.so $objs
echo built scratch/libdeepclr_$name.so
