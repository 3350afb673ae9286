#!/usr/bin/env bash
# Resource / instruction summary of the four g2p2g instantiations.  usage: tools/kstat.sh [extra hipcc flags...]
D=$(mktemp -d); trap 'rm -rf $D' EXIT
cd "$(dirname "$0")/../claymore_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Wno-unused-value -fno-slp-vectorize "$@" -save-temps=obj -o $D/x.so claymore_hip.hip 2>&1 | grep -E "error|warning: v" | head
S=$D/claymore_hip-hip-amdgcn-amd-amdhsa-gfx950.s
for m in 0 1 2 3 P0 P1 P2 P3; do
  case $m in P*) sym="_ZN3mpm17g2p2g_pair_kernelILi${m#P}E";; *) sym="_ZN3mpm12g2p2g_kernelILi${m}E";; esac   # (P*: two particles per lane, mpm_g2p2g_pair.hpp)
  awk "/^${sym}/,/\.end_amdhsa_kernel/" $S > $D/k$m.s
  [ -n "$KEEP" ] && cp $D/k$m.s $KEEP.$m.s
  echo "MAT $m: valu $(grep -cE '^\s+v_' $D/k$m.s) pk $(grep -cE '^\s+v_pk' $D/k$m.s) salu $(grep -cE '^\s+s_' $D/k$m.s) lds $(grep -cE '^\s+ds_' $D/k$m.s) vmem $(grep -cE '^\s+(global|buffer|scratch)_' $D/k$m.s) scratch_ops $(grep -cE '^\s+scratch_' $D/k$m.s) | $(grep -E 'next_free_vgpr|private_segment_fixed_size|group_segment_fixed_size' $D/k$m.s | awk '{printf "%s=%s ", $1, $2}')"
done
