#!/bin/bash
# usage: tools/kregs.sh <file.hip> [grep pattern]   -- VGPR / AGPR / SGPR / LDS / spill figures of every kernel of one source (device asm of a gfx950 compile)
F=$1; P=${2:-.}
D=$(dirname "$F")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I"$D" --cuda-device-only -S "$F" -o /tmp/kregs_$$.s || exit 1
awk '/^\s*\.amdhsa_kernel /{k=$2} /\.amdhsa_next_free_vgpr|\.amdhsa_accum_offset|\.amdhsa_group_segment_fixed_size|; ScratchSize|; Occupancy|\.amdhsa_next_free_sgpr/{print k, $0}' /tmp/kregs_$$.s | grep -E "$P" | sed 's/\s\+/ /g'
grep -E "^; (ScratchSize|Occupancy|NumVgprs|NumAgprs|codeLenInByte)" -B0 /tmp/kregs_$$.s > /dev/null
cp /tmp/kregs_$$.s /tmp/kregs_last.s; rm -f /tmp/kregs_$$.s
