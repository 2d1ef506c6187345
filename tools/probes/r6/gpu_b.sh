#!/bin/bash
# round 6, second GPU call: where the slowest waves of HumanoidTorque.run spend their cycles (timers build of family 8 on the DEFER tree)
O=gpurun_out/r6b; mkdir -p $O
LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/liblocohip_timers.so timeout 600 python tools/probes/r3/slow_waves.py HumanoidTorque.run 4096 1 40 > $O/slow_waves_ht.txt 2>&1
cat $O/slow_waves_ht.txt
LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/liblocohip_timers.so timeout 600 python tools/probes/r5/replay_profile.py HumanoidTorque.run 32 > $O/replay_profile_ht.txt 2>&1
tail -30 $O/replay_profile_ht.txt
