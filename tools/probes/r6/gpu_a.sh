#!/bin/bash
# round 6, first GPU call: same-box A/B of the round-5 library (liblocohip_r5.so), the PV-array variant and the current tree on the pair-pass
# families, then the bitwise guards + golden KATs of those families with the current library.
# usage (inside gpurun): bash tools/probes/r6/gpu_a.sh
O=gpurun_out/r6a; mkdir -p $O
C=loco_mujoco_amd/csrc
rate() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s %s: %.3f ms/step %.0f env-steps/s parity %s' % ('$1', '$2', d['ms_per_step'], d['value'], d.get('parity', {}).get('within_tolerance')))"; }
for T in HumanoidTorque.run UnitreeG1.walk; do
  for V in liblocohip_r5.so liblocohip_pvarr.so liblocohip.so; do
    [ -f $C/$V ] || continue
    LOCOHIP_LIB=$PWD/$C/$V timeout 600 python bench.py --task $T --steps 100 --warmup 30 --no-cpu-baseline --configs off --surface-steps 0 2>$O/err_${T}_$V.log | rate $T $V >> $O/ab.log 2>&1
  done
done
for V in liblocohip_r5.so liblocohip.so; do
  LOCOHIP_LIB=$PWD/$C/$V timeout 600 python bench.py --task HumanoidMuscle.run --envs-per-gpu 2048 --steps 100 --warmup 30 --no-cpu-baseline --configs off --surface-steps 0 2>$O/err_hm_$V.log | rate HumanoidMuscle.run2048 $V >> $O/ab.log 2>&1
  LOCOHIP_LIB=$PWD/$C/$V timeout 600 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --configs off --surface-steps 0 2>$O/err_a1_$V.log | rate A1 $V >> $O/ab.log 2>&1
done
cat $O/ab.log
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_rollout_is_bitwise or replay_kernel_is_bitwise or humanoid_torque_one_control_step or humanoid_torque_random_states or no_contact_is_dropped or unitree_g1_one_control or unitree_h1_one_control or humanoid_muscle_one_control or six_link_self or 4_ages" 2>&1 | tail -15 > $O/guards.log
cat $O/guards.log
