#!/bin/bash
# generic same-box A/B: csrc/liblocohip_old.so against the tree's library, the quadruped's bench rollout (3 runs) and Talos / Atlas
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5y; mkdir -p $O; rm -f $O/ab.txt
P='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("ms %.4f value %.5g iters/pass %.3f parity %s %s" % (d["ms_per_step"], d["value"], d["stats"]["newton_iters_per_forward_pass"], d.get("parity", {}).get("qpos_linf"), d.get("parity", {}).get("qvel_linf")))
'
for rep in 1 2 3; do
for lib in old new; do
  if [ $lib = old ]; then export LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/liblocohip_old.so; else unset LOCOHIP_LIB; fi
  echo "== A1 $lib (run $rep)" >> $O/ab.txt
  timeout 300 python bench.py --steps 500 --warmup 80 --fuse 0 --sustained 0 --configs off $( [ $rep = 1 ] || echo --no-cpu-baseline ) 2>&1 | python -c "$P" >> $O/ab.txt
done
done
cat $O/ab.txt
