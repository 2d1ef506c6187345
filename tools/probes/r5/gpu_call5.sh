#!/bin/bash
# pollers per launch and hand-off thresholds with resume (same box; probes build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
B="python bench.py --steps 150 --warmup 30 --sustained 0 --fuse 0 --no-cpu-baseline --configs off"
P='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("ms %.3f value %.4g replayed %d overflow %d nan %d" % (d["ms_per_step"], d["value"], d["stats"]["replayed_env_steps"], d["stats"]["overflow_contacts"], d["stats"]["nan_resets"]))
'
export LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/liblocohip_probes.so
for t in HumanoidTorque.run Atlas.walk; do
for pol in 2,2,32 3,4,64 4,4,64 6,8,128; do
  for h in 0,0,0 7,0,0; do
    echo "== $t pollers $pol handoff $h" >> $O/ab.txt
    LM_POLLERS=$pol timeout 300 $B --task $t --handoff $h 2>&1 | python -c "$P" >> $O/ab.txt 2>&1
  done
done
done
for h in 7,16,0 6,0,0 7,12,0; do
    echo "== HumanoidTorque.run pollers 4,4,64 handoff $h" >> $O/ab.txt
    LM_POLLERS=4,4,64 timeout 300 $B --task HumanoidTorque.run --handoff $h 2>&1 | python -c "$P" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
