#!/usr/bin/env bash
# Regenerates tests/golden/*.f32|*.i32 from the reference's own device functions, compiled as host
# C++.  Runs ONLY where /root/reference is mounted (the build container); the GPU box and the tests
# only ever read the committed vectors.  Scratch copies of five math headers are made in a temp dir
# because qr.cuh:55 names an undeclared identifier inside a never-instantiated template (accepted by
# nvcc, rejected by g++); the one token is fixed in the scratch copy, nothing is written to the repo
# or to /root/reference.
set -euo pipefail
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$(cd "$HERE/.." && pwd)
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$TMP/shim" "$TMP/scratch/MnBase/Math/Matrix"
echo '#include "cuda_host_shim.h"' > "$TMP/shim/cuda.h"
cp "$REF"/Library/MnBase/Math/Matrix/{svd.cuh,qr.cuh,Givens.cuh,Utility.h,MatrixUtils.h} "$TMP/scratch/MnBase/Math/Matrix/"
sed -i 's/rotation.fill<2>(R);/rotation.fill<2>(r);/' "$TMP/scratch/MnBase/Math/Matrix/qr.cuh"
# G7: the arithmetic of calculate_contribution_and_store_particle_data<J_FLUID> (Projects/GMPM/mgmpm_kernels.cuh:473-504) lives in a header
# that pulls in every CUDA kernel of the project; its statements - from the J update to the nine contrib[] lines, without the write-back that
# follows - are cut out of the file AS TEXT into a scratch include and compiled inside a function whose locals carry the names the text uses
# (data.J, particle_buffer.volume / bulk / gamma / viscosity, dt.count(), A, contrib: gen_golden.cpp, jfluid_reference_block)
sed -n '/^__forceinline__ __device__ void calculate_contribution_and_store_particle_data<MaterialE::J_FLUID>/,/^}/p' "$REF/Projects/GMPM/mgmpm_kernels.cuh" \
  | sed -n '/Update determinante/,/Write back particle data/p' | sed '$d' > "$TMP/scratch/jfluid_block.inc"
grep -q "powf(data.J, -particle_buffer.gamma)" "$TMP/scratch/jfluid_block.inc"
g++ -std=c++17 -O1 -ffp-contract=off -fpermissive -w \
    -I"$HERE" -I"$TMP/shim" -I"$TMP/scratch" -I"$REF/Library" -I"$REF/Projects/GMPM" \
    -I"$REF/Externals/function_ref" -I"$REF/Externals/variant" -I"$REF/Externals/optional" \
    "$HERE/gen_golden.cpp" -o "$TMP/gen_golden"
"$TMP/gen_golden" "$OUT"
