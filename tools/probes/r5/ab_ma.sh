#!/bin/bash
# M a carried across Newton iterations (new) against recomputed (jar): per family, one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s; mkdir -p $O
P='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("ms %.4f value %.5g" % (d["ms_per_step"], d["value"]))
'
for t in "--steps 500 --warmup 80" "--task HumanoidTorque.run --steps 150 --warmup 30" "--task Atlas.walk --dr --envs-per-gpu 2048 --steps 150 --warmup 30" "--task HumanoidMuscle.run --envs-per-gpu 2048 --steps 150 --warmup 30" "--task Talos.walk --steps 300 --warmup 30"; do
for rep in 1 2; do
for v in jar new; do
  if [ $v = jar ]; then export LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/build_v/lib_jar.so; else unset LOCOHIP_LIB; fi
  echo "== $t $v" >> $O/ab.txt
  timeout 300 python bench.py $t --fuse 0 --sustained 0 --configs off --no-cpu-baseline 2>&1 | python -c "$P" >> $O/ab.txt
done
done
done
cat $O/ab.txt
