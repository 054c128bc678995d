#!/bin/bash
# register / LDS / scratch use of the kernels of one csrc unit (hipcc -Rpass-analysis=kernel-resource-usage); usage: tools/kres.sh det_roialign.hip [grep pattern] [extra flags]
cd "$(dirname "$0")/../waymo_2d_tracking_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -Wno-unused-function $3 -Rpass-analysis=kernel-resource-usage -S --cuda-device-only -o /tmp/kres_$$.s "$1" 2>&1 \
  | grep -i "error\|remark.*\(Function Name\| VGPRs:\|AGPRs\|Occupancy\|Scratch\|LDS Size\|SGPRs:\)" | sed 's/.*remark: *//; s/\[-Rpass.*//; s/Function Name: /\n/' | tr '\n' ' ' | sed 's/ _Z/\n_Z/g' | grep "${2:-.}"
echo
echo "asm: /tmp/kres_$$.s"
