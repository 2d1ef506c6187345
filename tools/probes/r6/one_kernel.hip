// one_kernel.hip — ONE instantiation of the step kernel, for the register-diet loop of round 6:
//   hipcc --offload-arch=gfx950 -Os <the Makefile's FLAGS> -DK_ARGS="5,8,true,false,0,0,0,4,false,1" -Rpass-analysis=kernel-resource-usage \
//         -I loco_mujoco_amd/csrc -c -o /dev/null tools/probes/r6/one_kernel.hip            (tools/probes/r6/ru.sh)
// prints VGPR / AGPR / scratch / spills of that kernel alone in ~25 s instead of the family object's 80 s.
#include "lm_step.h"
namespace lmk {
template __global__ void step_kernel<K_ARGS>(KArgs);
}
