#!/bin/bash
# round 6, third GPU call: the convex collider's warm-start cache — guards (bitwise invariants, golden rows), same-box A/B against the
# library without it, counters of the timers build; the python_surface leg of bench.py
O=gpurun_out/r6c; mkdir -p $O
C=loco_mujoco_amd/csrc
rate() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s %s: %.3f ms/step %.0f env-steps/s' % ('$1', '$2', d['ms_per_step'], d['value']))"; }
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_rollout_is_bitwise or replay_kernel_is_bitwise or humanoid_torque_one_control_step or humanoid_torque_random_states or no_contact_is_dropped or unitree_g1_one_control or unitree_h1_one_control or humanoid_muscle_one_control or six_link_self or 4_ages or auto_reset_and_sharding or gymnasium_wrapper or step_on_device_buffers" 2>&1 | tail -15 > $O/guards.log
cat $O/guards.log
for T in HumanoidTorque.run UnitreeG1.walk UnitreeH1.run; do
  for V in liblocohip_defer.so liblocohip.so; do
    LOCOHIP_LIB=$PWD/$C/$V timeout 600 python bench.py --task $T --steps 100 --warmup 30 --no-cpu-baseline --configs off --surface-steps 0 2>$O/err_${T}_$V.log | rate $T $V >> $O/ab.log 2>&1
  done
done
for V in liblocohip_defer.so liblocohip.so; do
  LOCOHIP_LIB=$PWD/$C/$V timeout 600 python bench.py --task HumanoidMuscle.run --envs-per-gpu 2048 --steps 100 --warmup 30 --no-cpu-baseline --configs off --surface-steps 0 2>$O/err_hm_$V.log | rate HumanoidMuscle.run2048 $V >> $O/ab.log 2>&1
done
cat $O/ab.log
LOCOHIP_LIB=$PWD/$C/liblocohip_timers.so timeout 600 python tools/probes/r3/slow_waves.py HumanoidTorque.run 4096 1 40 > $O/slow_waves_ht.txt 2>&1
cat $O/slow_waves_ht.txt
timeout 900 python bench.py --steps 20 --warmup 10 --configs off > $O/bench_a1.json 2> $O/bench_a1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6c/bench_a1.json").read().strip().splitlines()[-1])
print("A1 value", d["value"], "ms", d["ms_per_step"], "parity", d.get("parity"))
print("python_surface", d.get("python_surface"))
PY
tail -5 $O/bench_a1.err
