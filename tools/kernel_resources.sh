#!/bin/bash
# VGPRs / LDS / scratch of the kernels matching $1 (default k_rp_step): the step kernel must stay at <= 128 VGPRs and <= 40 KB of
# LDS or its 1,024-workgroup grid needs a second round per launch
cd "$(dirname "$0")/../voxblox_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I../../include ${EXTRA_FLAGS} -Rpass-analysis=kernel-resource-usage -c vbx_hip.hip -o /dev/null 2>&1 \
  | grep -A12 "Function Name: .*${1:-k_rp_step}" | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|LDS Size|Occupancy" | sed 's/.*remark: [^ ]* //'
