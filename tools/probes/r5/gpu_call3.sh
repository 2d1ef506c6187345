#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
for lib in liblocohip_timers.so liblocohip_timers_rep4.so; do
  echo "== $lib" >> $O/replay_profile.txt
  LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/$lib timeout 600 python tools/probes/r5/replay_profile.py HumanoidTorque.run 32 >> $O/replay_profile.txt 2>&1
done
cat $O/replay_profile.txt
