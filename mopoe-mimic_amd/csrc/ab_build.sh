#!/bin/bash
# build an alternative library for same-box A/B timing: ./ab_build.sh <name> <extra hipcc flags...>
set -e
name=$1; shift
cd "$(dirname "$0")"
mkdir -p ab
objs=""
for f in adam api conv_gemm conv_gemm_bf16 edge elementwise latent likelihood; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c $f.hip -o ab/${name}_$f.o
  objs="$objs ab/${name}_$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib_${name}.so $objs
echo built ab/lib_${name}.so
