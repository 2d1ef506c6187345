#!/bin/bash
# round 5, first GPU call: the new bench line, where HumanoidTorque.run's time goes (timers build), replay latency, miscompile matrix
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/log.txt
LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/liblocohip_timers.so timeout 400 python tools/probes/r3/slow_waves.py HumanoidTorque.run 4096 1 40 > $O/slow_waves_ht.txt 2>&1; echo "slow_waves rc $?" >> $O/log.txt
timeout 400 python tools/probes/r4/replay_latency.py HumanoidTorque.run > $O/replay_latency.txt 2>&1; echo "replay_latency rc $?" >> $O/log.txt
timeout 900 bash tools/probes/r5/miscompile_matrix.sh run > $O/mm.txt 2>&1; echo "mm rc $?" >> $O/log.txt
tail -c 1500 $O/bench.json; cat $O/log.txt; cat $O/mm.txt | tail -40
