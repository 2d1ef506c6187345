#!/bin/bash
# Newton bookkeeping (row residuals moved by alpha x Jv instead of rebuilt; no line evaluation at alpha = 0): old library vs new, one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q; mkdir -p $O
P='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("ms %.4f value %.5g iters/pass %.3f parity %s %s" % (d["ms_per_step"], d["value"], d["stats"]["newton_iters_per_forward_pass"], d.get("parity", {}).get("qpos_linf"), d.get("parity", {}).get("qvel_linf")))
'
for rep in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/liblocohip_old.so; else unset LOCOHIP_LIB; fi
  echo "== A1 $lib (run $rep)" >> $O/ab.txt
  timeout 300 python bench.py --steps 500 --warmup 80 --fuse 0 --sustained 0 --configs off $( [ $rep = 1 ] || echo --no-cpu-baseline ) 2>&1 | python -c "$P" >> $O/ab.txt
done
done
for t in "--task HumanoidTorque.run" "--task Atlas.walk --dr --envs-per-gpu 2048" "--task HumanoidMuscle.run --envs-per-gpu 2048" "--task Talos.walk" "--task UnitreeG1.walk"; do
for lib in old new; do
  if [ $lib = old ]; then export LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/liblocohip_old.so; else unset LOCOHIP_LIB; fi
  echo "== $t $lib" >> $O/ab.txt
  timeout 300 python bench.py $t --steps 150 --warmup 30 --fuse 0 --sustained 0 --configs off 2>&1 | python -c "$P" >> $O/ab.txt
done
done
cat $O/ab.txt
