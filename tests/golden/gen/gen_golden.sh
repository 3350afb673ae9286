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
g++ -std=c++17 -O1 -ffp-contract=off -fpermissive -w \
    -I"$HERE" -I"$TMP/shim" -I"$TMP/scratch" -I"$REF/Library" -I"$REF/Projects/GMPM" \
    -I"$REF/Externals/function_ref" -I"$REF/Externals/variant" -I"$REF/Externals/optional" \
    "$HERE/gen_golden.cpp" -o "$TMP/gen_golden"
"$TMP/gen_golden" "$OUT"
