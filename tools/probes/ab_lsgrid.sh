#!/bin/bash
# A/B of the first-round step lengths of the four-point line search (fractions of the Newton step besides 1)
for g in "0.25,0.0625,0.015625" "0.5,0.25,0.125" "0.5,0.1,0.01" "0.3,0.09,0.027" "0.2,0.04,0.008" "0.35,0.1,0.02" "0.6,0.3,0.1"; do
  for t in UnitreeA1.simple HumanoidTorque.run; do
    LM_LS_GRID=$g python bench.py --task $t --steps 300 --warmup 30 --no-cpu-baseline --fuse 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid %-22s %-20s %.4f ms  fused %.4f ms  ls-iters/fwd %.3f' % ('$g', '$t', d['ms_per_step'], d['rollout_fused']['ms_per_step'], d['stats']['newton_iters_per_forward_pass']))"
  done
done
