#!/bin/bash
# the gate in front of the regular launch (pollers resident first): A/B on one box, time line, targeted tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; mkdir -p $O
B="python bench.py --steps 150 --warmup 30 --sustained 0 --fuse 25 --no-cpu-baseline --configs off"
P='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("ms %.3f value %.4g fused %.4g replayed %d overflow %d nan %d" % (d["ms_per_step"], d["value"], d.get("rollout_fused", {}).get("value", 0), d["stats"]["replayed_env_steps"], d["stats"]["overflow_contacts"], d["stats"]["nan_resets"]))
'
export LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/liblocohip_probes.so
for t in HumanoidTorque.run Atlas.walk UnitreeG1.walk Talos.walk; do
  for mode in nogate gate; do
    unset LM_NO_GATE
    if [ $mode = nogate ]; then export LM_NO_GATE=1; fi
    echo "== $t $mode" >> $O/ab.txt
    timeout 300 $B --task $t 2>&1 | python -c "$P" >> $O/ab.txt 2>&1
  done
done
unset LM_NO_GATE
for pol in 3,4,48 4,4,64; do
  echo "== HumanoidTorque.run gate pollers $pol" >> $O/ab.txt
  LM_POLLERS=$pol timeout 300 $B --task HumanoidTorque.run 2>&1 | python -c "$P" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/liblocohip_timers.so timeout 600 python tools/probes/r5/timeline.py HumanoidTorque.run 5 > $O/timeline_ht.txt 2>&1; cat $O/timeline_ht.txt
unset LOCOHIP_LIB
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "replay or fused or hand_off or root_dof or sharding or folded or masked" > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -8 $O/pytest.txt
