#!/usr/bin/env bash
# Regenerates tests/golden/g16_*, g19_* from the reference's own STATEMENTS of the g2p2g kernel body and of the grid update
# (Projects/GMPM/mgmpm_kernels.cuh), cut out of the file as text into a temp dir and compiled between locals (gen_golden_kernel.cpp).
# Runs ONLY where /root/reference is mounted; nothing of the reference's text is written to the repo.
set -euo pipefail
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=${1:-$(cd "$HERE/.." && pwd)}
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$TMP/shim" "$TMP/scratch/MnBase/Math/Matrix"
echo '#include "cuda_host_shim.h"' > "$TMP/shim/cuda.h"
cp "$REF"/Library/MnBase/Math/Matrix/{svd.cuh,qr.cuh,Givens.cuh,Utility.h,MatrixUtils.h} "$TMP/scratch/MnBase/Math/Matrix/"
sed -i 's/rotation.fill<2>(R);/rotation.fill<2>(r);/' "$TMP/scratch/MnBase/Math/Matrix/qr.cuh"   # (see gen_golden.sh)
K="$REF/Projects/GMPM/mgmpm_kernels.cuh"
S="$TMP/scratch"
# the kernel g2p2g (:665-937)
sed -n '/^__global__ void g2p2g(Duration dt, Duration new_dt/,/^}/p' "$K" > "$S/g2p2g.txt"
# G16: from the stencil base to the advection (:772-838)
sed -n '/\/\/Get position of grid cell/,/pos += vel \* dt.count();/p' "$S/g2p2g.txt" > "$S/gather.inc"
grep -q "ivec3 global_base_index = get_block_id(pos.data_arr()) - 1;" "$S/gather.inc"
grep -q "A\[8\] += W \* vi\[2\] \* xixp\[2\];" "$S/gather.inc"
test "$(tail -n 1 "$S/gather.inc" | tr -d '\t ')" = "pos+=vel*dt.count();"
# G17: from the contrib line to the end of the scatter loops (:848-905); the four lines behind them (the particle loop's brace,
# __syncthreads, a blank line, the next comment) are dropped
sed -n '/\/\/Update momentum?/,/\/\/Store data from shared memory to grid/p' "$S/g2p2g.txt" | head -n -4 > "$S/scatter.inc"
grep -q "contrib = (A \* particle_buffer.mass - contrib \* new_dt.count()) \* config::G_D_INV;" "$S/scatter.inc"
grep -q "next_particle_buffer.add_advection(partition, new_global_base_index - 1, dirtag, particle_id_in_block);" "$S/scatter.inc"
test "$(grep -c atomicAdd "$S/scatter.inc")" = 4
! grep -q "__syncthreads" "$S/scatter.inc"
# G18: the bodies of calculate_contribution_and_store_particle_data<J_FLUID | FC | SAND | NACC> (:473-663), without their signature and closing brace
sed -n "/^__forceinline__ __device__ void calculate_contribution_and_store_particle_data<MaterialE::J_FLUID>/,/^}/p" "$K" | sed '1d;$d' > "$S/body_jfluid.inc"
grep -q "powf(data.J, -particle_buffer.gamma)" "$S/body_jfluid.inc"
for m in FIXED_COROTATED:fc SAND:sand NACC:nacc; do
  sed -n "/^__forceinline__ __device__ void calculate_contribution_and_store_particle_data<MaterialE::${m%%:*}>/,/^}/p" "$K" | sed '1d;$d' > "$S/body_${m##*:}.inc"
  grep -q "compute_stress<float, MaterialE::${m%%:*}>" "$S/body_${m##*:}.inc"
  grep -q "matrix_matrix_multiplication_3d(dws.data_arr(), contrib, F.data_arr());" "$S/body_${m##*:}.inc"
done
# G19: the cell arithmetic of update_grid_velocity_query_max (:353-388): from the mass fetch to the NaN rule
sed -n '/^__global__ void update_grid_velocity_query_max(/,/^}/p' "$K" \
  | sed -n '/const float mass = grid_block.val_1d(_0, cell_id_in_block);/,/vel_sqr = std::numeric_limits<float>::infinity();/p' > "$S/gridcell.inc"
echo "			}" >> "$S/gridcell.inc"   # (closes the if(isnan) whose last line the range ends on)
grep -q "vel\[1\] += config::G_GRAVITY \* dt.count();" "$S/gridcell.inc"
g++ -std=c++17 -O1 -ffp-contract=off -fpermissive -w \
    -I"$HERE" -I"$TMP/shim" -I"$S" -I"$REF/Library" -I"$REF/Projects/GMPM" \
    -I"$REF/Externals/function_ref" -I"$REF/Externals/variant" -I"$REF/Externals/optional" \
    "$HERE/gen_golden_kernel.cpp" -o "$TMP/gen_golden_kernel"
"$TMP/gen_golden_kernel" "$OUT"
