#!/bin/bash
# resource usage (and optionally the ISA) of ONE step-kernel instantiation.  usage: ru.sh "<template args>" [out.s] [extra flags...]
#   HT regular : ru.sh "5,8,true,false,0,0,0,4,false,1"
#   HT replay  : ru.sh "5,128,true,false,0,0,0,4,true,1"
#   A1 bench   : ru.sh "3,6,false,false,1,0,0,4,false,2"
set -e
cd "$(dirname "$0")/../../.."
ARGS="$1"; OUT="${2:-}"; shift; [ $# -gt 0 ] && shift
FLAGS="--offload-arch=gfx950 -Os -std=c++17 -fPIC -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -Wno-unused-result -Wno-unused-value"
if [ -n "$OUT" ]; then
  /opt/rocm/bin/hipcc $FLAGS "$@" -DK_ARGS="$ARGS" -I loco_mujoco_amd/csrc --cuda-device-only -S -o "$OUT" tools/probes/r6/one_kernel.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs:|AGPRs|Spill|ScratchSize" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ' '; echo
else
  /opt/rocm/bin/hipcc $FLAGS "$@" -DK_ARGS="$ARGS" -I loco_mujoco_amd/csrc --cuda-device-only -c -o /dev/null tools/probes/r6/one_kernel.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs:|AGPRs|Spill|ScratchSize" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ' '; echo
fi
