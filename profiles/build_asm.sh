#!/bin/bash
# compile the engine and keep the gfx950 assembly of one kernel: profiles/build_asm.sh [kernel name regex]  -> /tmp/asm/<kernel>.s + resource lines
set -e
cd /root/repo
mkdir -p /tmp/asm
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-result -Wno-unused-value -Wno-parentheses -Wno-pass-failed -Iinclude -o /tmp/asm/libmlst_hip_asm.so metamlst_amd/csrc/mlst_engine.hip -save-temps=obj   # temps land beside the output: /tmp/asm, never the source tree
K=${1:-k_extend_160}
awk -v k="$K" '$0 ~ "^_Z[0-9]+"k"[A-Za-z0-9_]*:" {p=1} p{print} /s_endpgm/{if(p){exit}}' /tmp/asm/mlst_engine-hip-amdgcn-amd-amdhsa-gfx950.s > /tmp/asm/$K.s
grep -A40 "\.name:.*$K" /tmp/asm/mlst_engine-hip-amdgcn-amd-amdhsa-gfx950.s | grep -E "vgpr_count|sgpr_count|private_segment_fixed|group_segment_fixed|spill" | head -6
wc -l /tmp/asm/$K.s; echo "readlane/writelane: $(grep -c 'v_readlane\|v_writelane' /tmp/asm/$K.s)"
