#!/usr/bin/env bash
# writes build_stamp.txt (not tracked; travels with gpurun): the commit, whether the tree is dirty, and the hash of the engine library
# the GPU box will load - every profile file starts with this line
cd "$(dirname "$0")/.."
c=$(git rev-parse --short=12 HEAD)
d=$(git status --porcelain -- claymore_amd include bench.py | grep -v '^??' | wc -l)
h=$(sha256sum claymore_amd/csrc/libclaymore_hip.so | cut -c1-16)
echo "git $c$([ "$d" != 0 ] && echo "+dirty($d)") libclaymore_hip.so sha256 $h" > build_stamp.txt
cat build_stamp.txt
