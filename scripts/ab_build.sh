#!/bin/bash
# Builds an A/B variant of libbtba.so into /tmp/ab/<name>/ with the ISA kept, prints the cost model of the dense loop.
#   scripts/ab_build.sh <name> [-DMACRO ...]
name=$1; shift
mkdir -p /tmp/ab/$name && cd /tmp/ab/$name || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -shared -fvisibility=hidden -save-temps=obj "$@" -o /tmp/ab/$name/libbtba.so /root/repo/bundletrack_amd/csrc/btba_api.hip 2>&1 | grep -v warning | grep -B2 -A6 "error" | head -30
S=/tmp/ab/$name/btba_api-hip-amdgcn-amd-amdhsa-gfx950.s
echo "== $name $*"
python /root/repo/scripts/isa_cost.py $S k_fused_sweepsILi1E --loop ${LOOP:-1} | head -1
grep "k_fused_sweepsILi1E.*\.num_vgpr\|k_fused_sweepsILi1E.*private_seg" $S | sed 's/.*PKi//'
