#!/usr/bin/env bash
# Regenerates tests/golden/g12_* .. g15_* from the MGSP project's own host-compilable functions (compute_dt,
# SignedDistanceGrid::rot_angle_to_matrix / get_signed_distance_and_normal / query_sdf / detect_and_resolve_collision), compiled as
# host C++.  Runs ONLY where /root/reference is mounted (the build container); the GPU box and the tests only ever read the
# committed vectors.  A scratch copy of boundary_condition.cuh is made in a temp dir, because the file as it lies pulls in what this
# image lacks and what is not arithmetic: <fmt/...> (a CPM-fetched library), grid_buffer.cuh (which includes every CUDA kernel of the
# project) and cudaMemcpyAsync.  The scratch copy drops those include lines, the init() method (a cudaMemcpyAsync), the fmt::print
# statements and init_from_signed_distance_file (file IO); the two domain typedefs boundary_condition.cuh takes from grid_buffer.cuh
# are extracted from that file into a scratch header.  Every arithmetic line is the reference's own; nothing is written to the repo
# (beyond the vectors) or to /root/reference.
set -euo pipefail
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=${1:-$(cd "$HERE/.." && pwd)}
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$TMP/shim" "$TMP/scratch"
echo '#include "cuda_host_shim.h"' > "$TMP/shim/cuda.h"
cp "$REF/Projects/MGSP/boundary_condition.cuh" "$TMP/scratch/"
sed -i -e '/#include <fmt\//d' \
       -e 's/#include "grid_buffer.cuh"/#include "grid_domains_scratch.h"/' \
       -e '/void init(base_t& host_data, cudaStream_t stream) {/,/^\t}/d' \
       -e '/fmt::print(/,/);/d' \
       -e '/^inline void init_from_signed_distance_file/,/^}/d' "$TMP/scratch/boundary_condition.cuh"
{
  echo '#pragma once'; echo '#include "settings.h"'; echo 'namespace mn {'; echo 'using namespace placeholder;'
  grep -E "^using (BlockDomain|GridDomain)[[:space:]]" "$REF/Projects/MGSP/grid_buffer.cuh"
  echo '}'
} > "$TMP/scratch/grid_domains_scratch.h"
g++ -std=c++17 -O1 -ffp-contract=off -fpermissive -w \
    -I"$HERE" -I"$TMP/shim" -I"$TMP/scratch" -I"$REF/Library" -I"$REF/Projects/MGSP" \
    -I"$REF/Externals/function_ref" -I"$REF/Externals/variant" -I"$REF/Externals/optional" \
    "$HERE/gen_golden_mgsp.cpp" -o "$TMP/gen_golden_mgsp"
"$TMP/gen_golden_mgsp" "$OUT"
