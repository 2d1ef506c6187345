#!/bin/bash
# same-box A/B over several configurations: csrc/liblocohip_old.so against the tree's library
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5z; mkdir -p $O; rm -f $O/ab.txt
P='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("ms %.4f value %.5g iters/pass %.3f parity %s %s overflow %d" % (d["ms_per_step"], d["value"], d["stats"]["newton_iters_per_forward_pass"], d.get("parity", {}).get("qpos_linf"), d.get("parity", {}).get("qvel_linf"), d["stats"]["overflow_contacts"]))
'
run() {   # label, bench arguments
  for lib in old new; do
    if [ $lib = old ]; then export LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/liblocohip_old.so; else unset LOCOHIP_LIB; fi
    echo "== $1 $lib" >> $O/ab.txt
    timeout 300 python bench.py $2 --fuse 0 --sustained 0 --configs off --no-cpu-baseline 2>&1 | python -c "$P" >> $O/ab.txt
  done
}
run "A1 (1)" "--steps 500 --warmup 80"
run "A1 (2)" "--steps 500 --warmup 80"
run "Atlas.walk --dr 2048" "--task Atlas.walk --dr --envs-per-gpu 2048 --steps 300 --warmup 50"
run "Talos.walk" "--task Talos.walk --steps 300 --warmup 50"
run "HumanoidTorque.run" "--task HumanoidTorque.run --steps 200 --warmup 40"
run "UnitreeG1.walk" "--task UnitreeG1.walk --steps 150 --warmup 30"
cat $O/ab.txt
