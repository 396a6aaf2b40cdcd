#!/bin/bash
# Device ISA of one kernel of csrc/render.hip: line count, VGPRs, and the mnemonic histogram.
# usage: tools/isa_stats.sh [kernel-substring] [extra hipcc flags...]
cd "$(dirname "$0")/../gpu-pathtracer_amd" || exit 1
k=${1:-k_pt_extend_persistentILb0}; shift
mkdir -p build_isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize "$@" -S --cuda-device-only -o build_isa/render.s csrc/render.hip 2>/dev/null
n=$(grep -n "^_ZN4mipt.*$k.*:" build_isa/render.s | head -1 | cut -d: -f1)
awk -v n=$n 'NR>=n{print} NR>n && /s_endpgm/{exit}' build_isa/render.s > build_isa/kernel.s
echo "lines: $(grep -cE '^\s+(v_|s_|ds_|global_|buffer_|flat_|scratch_)' build_isa/kernel.s)  valu: $(grep -cE '^\s+v_' build_isa/kernel.s)  salu: $(grep -cE '^\s+s_' build_isa/kernel.s)"
grep -A40 "\.name:.*$k" build_isa/render.s | grep -E "\.vgpr_count|\.sgpr_count|spill_count|private_segment_fixed" | head -5
