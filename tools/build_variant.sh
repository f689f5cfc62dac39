#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFOO=1 ...]   -> tools/ab/lib<name>.so (quantizer TU rebuilt with the defines;
# the GEMM TU's object is reused from the last regular build)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/tools/ab
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize -I$root/include -DFPQ_BUILD_TAG="\"$name\"" "$@" \
  -c -o /tmp/fpq_kernels_$name.o $root/fpqvar_amd/csrc/fpq_kernels.hip 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/ab/lib$name.so /tmp/fpq_kernels_$name.o $root/fpqvar_amd/csrc/fpq_gemm.o
echo built tools/ab/lib$name.so
