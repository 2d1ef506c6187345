#!/bin/bash
# Reproducer of the miscompile of kernels that CALL the convex-pair collider (csrc/lm_step.h LM_MPR_CALL; Makefile: SCHED;
# profiles/r3_notes.md §4). The shipped library inlines the collider; -DLM_MPR_CALL makes it a real function again.
# Builds kernel family 8 (<5 links, 8 slots, RK4, pyramids, PAIRS>) and family 10 (<5,8,Euler,pyramids,muscles,PAIRS>) of
# csrc/lm_family.hip three times — default strategy (max-occupancy), max-ILP, scheduler off — links each against the shipped objects of
# the other families and runs, on the GPU box:
#   * HumanoidTorque.run reset-table states through one control step against the fp64 oracle (family 8, plain kernel)
#   * test_fused_rollout_is_bitwise_the_single_step_rollout[HumanoidMuscle.run] (family 10, fused kernel)
# Observed (ROCm 7.2.0 hipcc, gfx950, -Os): max-occupancy -> family 8 wrong by O(1) in EVERY environment (1.2 rad after one step; the
# same source with printf statements in the pair pass is right); max-ILP -> family 8 right, the fused family-10 kernel wrong;
# -enable-misched=false -> both right at -Os (but <5,8,Euler,muscles,PAIRS> wrong at -O2). Every one of these builds is right WITHOUT the call
# (-DLM_NO_MPR, or the collider inlined: what ships). The kernels sit at the register ceiling: 256 VGPR + 256 AGPR + ~480 B scratch.
#   usage (from the repo root, CPU container):  bash tools/probes/r3/sched_repro.sh build      (three libraries, ~3 min on 8 cores)
#         (GPU box, through gpurun):            bash tools/probes/r3/sched_repro.sh run
set -e
cd "$(dirname "$0")/../../../loco_mujoco_amd/csrc"
F="-DLM_MPR_CALL --offload-arch=gfx950 -Os -std=c++17 -fPIC -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -Wno-unused-result -Wno-unused-value"
declare -A SCHED=( [occ]="" [ilp]="-mllvm -amdgpu-sched-strategy=max-ilp" [off]="-mllvm -enable-misched=false" )
if [ "$1" = build ]; then
  for v in occ ilp off; do
    mkdir -p build_repro_$v
    for f in 8 10; do for p in 0 1 2; do
      /opt/rocm/bin/hipcc $F ${SCHED[$v]} -DLM_FAMILY=$f -DLM_PART=$p -c -o build_repro_$v/lm_family_f${f}p$p.o lm_family.hip 2>/dev/null &
    done; done
    wait
    objs="build/lm_kernels.o"
    for f in 0 1 2 3 4 5 6 7 8 9 10; do for p in 0 1 2; do
      if [ -f build_repro_$v/lm_family_f${f}p$p.o ]; then objs="$objs build_repro_$v/lm_family_f${f}p$p.o"; else objs="$objs build/lm_family_f${f}p$p.o"; fi
    done; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblocohip_repro_$v.so $objs
    /opt/rocm/lib/llvm/bin/llvm-objdump --version > /dev/null 2>&1 && echo "built liblocohip_repro_$v.so"
  done
else
  cd ../..
  for v in occ ilp off; do
    echo "== scheduler: $v"
    LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/liblocohip_repro_$v.so python tools/probes/r3/ht_kat_debug.py HumanoidTorque.run 1,16 2>&1 | tail -2 | cut -c1-200
    LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/liblocohip_repro_$v.so python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fused_rollout_is_bitwise and Muscle" 2>&1 | tail -1
  done
fi
