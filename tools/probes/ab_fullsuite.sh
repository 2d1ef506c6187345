#!/bin/bash
# A/B of whole-library builds: each tools/probes/_bin/full_*.so replaces the library; GPU parity suite + bench per task.
cp loco_mujoco_amd/csrc/liblocohip.so /tmp/liblocohip_full.so
for f in tools/probes/_bin/full_*.so; do
  cp $f loco_mujoco_amd/csrc/liblocohip.so
  echo "=== $(basename $f)"
  python -m pytest tests -m gpu -q 2>&1 | tail -1
  for t in UnitreeA1.simple HumanoidTorque.walk Atlas.walk HumanoidMuscle.walk Talos.walk; do
    python bench.py --task $t --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-22s %.3f ms  %.0f env-steps/s' % ('$t', d['ms_per_step'], d['value']))"
  done
done
cp /tmp/liblocohip_full.so loco_mujoco_amd/csrc/liblocohip.so
