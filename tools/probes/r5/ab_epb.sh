#!/bin/bash
# environments per workgroup at 2048 environments per GPU (BASELINE configs 4 and 5): 4 (512 waves: half the SIMDs idle) against 2 and 1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5n; mkdir -p $O
export LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/liblocohip_probes.so
P='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("ms %.3f value %.4g replayed %d" % (d["ms_per_step"], d["value"], d["stats"]["replayed_env_steps"]))
'
for cfg in "--task Atlas.walk --dr --envs-per-gpu 2048" "--task HumanoidMuscle.run --envs-per-gpu 2048" "--task HumanoidTorque.run --envs-per-gpu 2048" "--envs-per-gpu 2048" "--envs-per-gpu 1024"; do
  for epb in 4 2 1; do
    echo "== $cfg epb $epb" >> $O/ab.txt
    LM_ENVS_PER_BLOCK=$epb timeout 300 python bench.py $cfg --steps 200 --warmup 40 --sustained 0 --fuse 0 --no-cpu-baseline --configs off 2>&1 | python -c "$P" >> $O/ab.txt 2>&1
  done
done
cat $O/ab.txt
