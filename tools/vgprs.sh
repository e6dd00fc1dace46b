#!/bin/bash
# registers / LDS / scratch of the kernels whose (demangled) name matches $1, from a device-only compile of csrc/kernels.hip
cd "$(dirname "$0")/.."
[ -n "$VGPRS_REUSE" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTKAMD_BUILD --cuda-device-only -S tokenizers_amd/csrc/kernels.hip -o /tmp/tkamd_kernels.s || exit 1
awk '/^[ \t]*\.amdhsa_kernel /{k=$2} /\.amdhsa_next_free_vgpr|\.amdhsa_group_segment_fixed_size|\.amdhsa_private_segment_fixed_size|\.amdhsa_accum_offset/{print k, $1, $2}' /tmp/tkamd_kernels.s | c++filt | grep -E "${1:-.}" 
