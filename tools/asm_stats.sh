#!/usr/bin/env bash
# Instruction histogram of one g2p2g instantiation.  usage: tools/asm_stats.sh <material 0-3> [extra hipcc flags...]
set -e
MAT=${1:-2}; shift || true
D=$(mktemp -d); trap 'rm -rf $D' EXIT
cd "$(dirname "$0")/../claymore_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-value "$@" -save-temps=obj -o $D/x.so claymore_hip.hip 2>/dev/null
S=$D/claymore_hip-hip-amdgcn-amd-amdhsa-gfx950.s
awk "/^_ZN3mpm12g2p2g_kernelILi${MAT}ELi0E/,/\.end_amdhsa_kernel/" $S > $D/k.s
[ -n "$KEEP" ] && cp $D/k.s $KEEP
grep -oE "^\s+[a-z_0-9]+" $D/k.s | sort | uniq -c | sort -rn | awk '{t+=$1; if(NR<=28) printf "%s:%s ",$2,$1} END{print "\nTOTAL",t}'
grep -E "next_free_vgpr|private_segment_fixed_size|group_segment_fixed_size" $D/k.s | head -4
exit 0
