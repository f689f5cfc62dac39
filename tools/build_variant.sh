#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFOO=1 ...]          -> tools/ab/lib<name>.so: the quantizer TU rebuilt with the defines,
#                                                               the GEMM TU's object reused from the last regular build
#        tools/build_variant.sh --gemm <name> [-DFOO=1 ...]   -> the GEMM TU rebuilt with the defines, the quantizer TU's object reused
set -e
tu=fpq_kernels; other=fpq_gemm; extra=-fno-slp-vectorize
if [ "$1" = --gemm ]; then tu=fpq_gemm; other=fpq_kernels; extra=; shift; fi
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/tools/ab
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 $extra -I$root/include -DFPQ_BUILD_TAG="\"$name\"" "$@" \
  -c -o /tmp/${tu}_$name.o $root/fpqvar_amd/csrc/$tu.hip 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/ab/lib$name.so /tmp/${tu}_$name.o $root/fpqvar_amd/csrc/$other.o
echo built tools/ab/lib$name.so
