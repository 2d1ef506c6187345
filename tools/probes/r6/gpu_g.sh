#!/bin/bash
# round 6: unrolled arrow loops + PV array + 14 hint positions against the library before them (liblocohip_base.so): guards, A/B, counters;
# the float64 surface with the conversion kernel writing the pinned slot itself (pinB) and the step kernel reading the pinned action (pinC)
O=gpurun_out/r6g; mkdir -p $O
C=loco_mujoco_amd/csrc
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_rollout_is_bitwise or replay_kernel_is_bitwise or humanoid_torque_one_control_step or humanoid_torque_random_states or no_contact_is_dropped or unitree_g1_one_control or unitree_h1_one_control or humanoid_muscle_one_control or six_link_self or 4_ages or auto_reset_and_sharding or a1_self_contacts or tangled" 2>&1 | tail -8 > $O/guards.log
cat $O/guards.log
for T in HumanoidTorque.run UnitreeG1.walk "HumanoidMuscle.run --envs-per-gpu 2048" UnitreeA1.simple; do
  bash tools/probes/r6/ab.sh $O/ab_$(echo $T | cut -d' ' -f1) "$T" liblocohip_base.so liblocohip.so > /dev/null 2>&1
  cat $O/ab_$(echo $T | cut -d' ' -f1)/ab.log
done
LOCOHIP_LIB=$PWD/$C/liblocohip_timers.so timeout 600 python tools/probes/r3/slow_waves.py HumanoidTorque.run 4096 1 40 > $O/slow_waves_ht.txt 2>&1
grep -E "launch ms|SLOWEST|mean cycles|convex collider" $O/slow_waves_ht.txt
for V in liblocohip.so liblocohip_pinB.so liblocohip_pinC.so liblocohip.so liblocohip_pinB.so liblocohip_pinC.so; do
  LOCOHIP_LIB=$PWD/$C/$V timeout 300 python tools/probes/r6/surface.py 300 2>&1 | tail -1 >> $O/surface.log
done
cat $O/surface.log
