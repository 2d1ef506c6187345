#!/bin/bash
# round 6: the convex collider as a real function (-DLM_MPR_CALL, family 8) against the inlined one: clock probe, bench A/B, guards
O=gpurun_out/r6e; mkdir -p $O
C=loco_mujoco_amd/csrc
rate() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s %s: %.3f ms/step %.0f env-steps/s' % ('$1', '$2', d['ms_per_step'], d['value']))"; }
LM_MPR_CLOCK=1 LOCOHIP_LIB=$PWD/$C/liblocohip_callclk.so timeout 600 python tools/probes/r3/slow_waves.py HumanoidTorque.run 4096 1 40 > $O/slow_waves_callclk.txt 2>&1
grep -E "launch ms|SLOWEST|mean cycles|clock probe|convex collider" $O/slow_waves_callclk.txt
for V in liblocohip.so liblocohip_call.so liblocohip.so liblocohip_call.so; do
  LOCOHIP_LIB=$PWD/$C/$V timeout 600 python bench.py --task HumanoidTorque.run --steps 100 --warmup 30 --no-cpu-baseline --configs off --surface-steps 0 2>$O/err_$V.log | rate HumanoidTorque.run $V >> $O/ab.log 2>&1
done
cat $O/ab.log
LOCOHIP_LIB=$PWD/$C/liblocohip_call.so timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_rollout_is_bitwise or replay_kernel_is_bitwise or humanoid_torque_one_control_step or humanoid_torque_random_states or no_contact_is_dropped" 2>&1 | tail -5
