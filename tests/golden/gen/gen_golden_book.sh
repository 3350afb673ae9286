#!/usr/bin/env bash
# Regenerates tests/golden/g20_* from the reference's own STATEMENTS of the integer bookkeeping of the particle path:
# Partition::insert / query / reinsert (Projects/GMPM/hash_table.cuh:118-135), ParticleBufferImpl::add_advection
# (particle_buffer.cuh:100-135), and the kernels activate_blocks, build_particle_cell_buckets, cell_bucket_to_block, compute_bin_capacity,
# register_neighbor_blocks, register_exterior_blocks, mark_active_particle_blocks, update_partition, update_buckets
# (mgmpm_kernels.cuh:21-151, :954-1000) and exclusive_scan_inverse (Library/MnBase/Algorithm/MappingKernels.cuh:44-55),
# cut out of the files as text into a temp dir and compiled over a serial thread loop (gen_golden_book.cpp).
# Runs ONLY where /root/reference is mounted; nothing of the reference's text is written to the repo.
set -euo pipefail
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=${1:-$(cd "$HERE/.." && pwd)}
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$TMP/shim"
echo '#include "cuda_host_shim.h"' > "$TMP/shim/cuda.h"
S="$TMP"
K="$REF/Projects/GMPM/mgmpm_kernels.cuh"
# a __global__ function by name, with the template line in front of it if there is one
cutfn() { awk -v name="$2" '{ if(!p && index($0, "__global__ void " name "(")) { if(prev ~ /^template</) print prev; p = 1 } if(p) { print; if($0 == "}") exit } prev = $0 }' "$1"; }
: > "$S/kernels.inc"
for f in activate_blocks build_particle_cell_buckets cell_bucket_to_block compute_bin_capacity register_neighbor_blocks register_exterior_blocks mark_active_particle_blocks update_partition update_buckets; do
  cutfn "$K" $f >> "$S/kernels.inc"
done
cutfn "$REF/Library/MnBase/Algorithm/MappingKernels.cuh" exclusive_scan_inverse >> "$S/kernels.inc"
test "$(grep -c '__global__ void' "$S/kernels.inc")" = 10
grep -q "partition.insert(blockid);" "$S/kernels.inc"
grep -q "const int particle_id_in_block = atomic_agg_inc<int>(particle_bucket_sizes + blockIdx.x);" "$S/kernels.inc"
grep -q "next_partition.reinsert(static_cast<int>(blockno));" "$S/kernels.inc"
grep -q "map_inv\[map_idx\] = idx;" "$S/kernels.inc"
# the three member functions of Partition<Opt> (hash_table.cuh:117-135) and add_advection (particle_buffer.cuh:99-135)
sed -n '/__forceinline__ __device__ value_t insert(key_t key) noexcept {/,/^};/p' "$REF/Projects/GMPM/hash_table.cuh" | sed '$d' > "$S/part_methods.inc"
grep -q "value_t tag = atomicCAS(&this->index(key), sentinel_v, 0);" "$S/part_methods.inc"
grep -q "this->index(this->active_keys\[index\]) = index;" "$S/part_methods.inc"
sed -n '/__forceinline__ __device__ void add_advection(Partition<1>& table/,/^	}/p' "$REF/Projects/GMPM/particle_buffer.cuh" > "$S/add_advection.inc"
grep -q "const int particle_id_in_cell = atomicAdd(cell_particle_counts" "$S/add_advection.inc"
grep -q "(dirtag \* config::G_PARTICLE_NUM_PER_BLOCK) | particle_id_in_block;" "$S/add_advection.inc"
g++ -std=c++17 -O1 -fpermissive -w -I"$HERE" -I"$TMP/shim" -I"$S" -I"$REF/Library" -I"$REF/Projects/GMPM" \
    -I"$REF/Externals/function_ref" -I"$REF/Externals/variant" -I"$REF/Externals/optional" \
    "$HERE/gen_golden_book.cpp" -o "$TMP/gen_golden_book"
"$TMP/gen_golden_book" "$OUT"
