#!/bin/bash
# Round 5: which code-generation switch makes the FUSED kernel of family 8 (<5 links, 8 slots, RK4, pyramids, self-collisions>, part 1)
# come out right at -Os with the DEFAULT machine-scheduler strategy? (csrc/Makefile SCHED_f8p1; profiles/r4_notes.md §9: every tenth
# environment non-finite after two control steps; max-ILP / -enable-misched=false / -O1 / -O2 are right.) Each variant rebuilds ONLY
# lm_family_f8p1.o with one extra switch and links it against the shipped objects; on the GPU box tools/probes/r4/fused_vs_single.py
# says whether the fused rollout is bitwise the single-step one. The switches separate the hypotheses:
#   waitcnt-forcezero  -> a missing s_waitcnt (memory ordering: source hand-over or the compiler's counter insertion)
#   snop-padding       -> a data hazard the hazard recognizer misses
#   sched stage / pre-RA switches -> a defect of one scheduling stage or pre-RA optimisation (liveness, rematerialisation)
#   usage (CPU container):  bash tools/probes/r5/miscompile_matrix.sh build
#         (GPU box):        bash tools/probes/r5/miscompile_matrix.sh run
set -e
cd "$(dirname "$0")/../../../loco_mujoco_amd/csrc"
F="--offload-arch=gfx950 -Os -std=c++17 -fPIC -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -Wno-unused-result -Wno-unused-value"
declare -A V=(
  [bad]=""
  [ilp]="-mllvm -amdgpu-sched-strategy=max-ilp"
  [wait0]="-mllvm -amdgpu-waitcnt-forcezero"
  [snop]="-mllvm -amdgpu-snop-padding=4"
  [nounclu]="-mllvm -amdgpu-disable-unclustered-high-rp-reschedule"
  [noclu]="-mllvm -amdgpu-disable-clustered-low-occupancy-reschedule"
  [noprera]="-mllvm -amdgpu-enable-pre-ra-optimizations=0"
  [nolive]="-mllvm -amdgpu-opt-vgpr-liverange=0 -mllvm -amdgpu-opt-exec-mask-pre-ra=0"
  [nodce]="-mllvm -amdgpu-dce-in-ra=0 -mllvm -amdgpu-enable-rewrite-partial-reg-uses=0"
  [trackers]="-mllvm -amdgpu-use-amdgpu-trackers"
  [nopost]="-mllvm -enable-post-misched=0"
  [prealloc]="-mllvm -amdgpu-prealloc-sgpr-spill-vgprs"
  [nosink]="-mllvm -disable-machine-sink"
  [bias100]="-mllvm -amdgpu-schedule-metric-bias=100"
)
if [ "$1" = build ]; then
  mkdir -p build_mm
  n=0
  for v in "${!V[@]}"; do
    ( /opt/rocm/bin/hipcc $F ${V[$v]} -DLM_FAMILY=8 -DLM_PART=1 -c -o build_mm/f8p1_$v.o lm_family.hip 2> build_mm/f8p1_$v.log \
      && { objs="build/lm_kernels.o"; for f in 0 1 2 3 4 5 6 7 8 9 10; do for p in 0 1 2; do
             if [ $f = 8 ] && [ $p = 1 ]; then objs="$objs build_mm/f8p1_$v.o"; else objs="$objs build/lm_family_f${f}p$p.o"; fi; done; done
           /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_mm/liblocohip_mm_$v.so $objs; echo "built $v"; } || echo "FAILED $v" ) &
    n=$((n+1)); if [ $((n % 7)) = 0 ]; then wait; fi
  done
  wait
else
  cd ../..
  for lib in loco_mujoco_amd/csrc/build_mm/liblocohip_mm_*.so; do
    v=$(basename $lib .so); v=${v#liblocohip_mm_}
    echo "== $v: ${V[$v]}"
    LOCOHIP_LIB=$PWD/$lib timeout 300 python tools/probes/r4/fused_vs_single.py HumanoidTorque.run 2>&1 | grep "replay 0" | cut -c1-220
  done
fi
