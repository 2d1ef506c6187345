#!/bin/bash
# A/B of experimental builds of family 8 part 0 (timers builds under csrc/build_x/lib_<name>.so): cost of a hard robot + the bench rollout
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5x; mkdir -p $O
for lib in loco_mujoco_amd/csrc/build_x/lib_*.so; do
  n=$(basename $lib .so)
  echo "== $n" >> $O/ab.txt
  LOCOHIP_LIB=$PWD/$lib timeout 300 python tools/probes/r5/replay_profile.py HumanoidTorque.run 24 2>&1 | tail -2 >> $O/ab.txt
  LOCOHIP_LIB=$PWD/$lib timeout 300 python bench.py --task HumanoidTorque.run --steps 100 --warmup 30 --sustained 0 --fuse 0 --configs off 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   bench ms %.3f replayed %d parity %.2e %.2e %s' % (d['ms_per_step'], d['stats']['replayed_env_steps'], d['parity']['qpos_linf'], d['parity']['qvel_linf'], d['parity']['within_tolerance']))
" >> $O/ab.txt
done
cat $O/ab.txt
