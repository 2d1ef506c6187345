#!/bin/bash
# do the two "miscompile" reproducers of rounds 3 / 4 still reproduce on the round-5 source?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i; mkdir -p $O
for v in occ ilp off; do
  echo "== out-of-line collider (-DLM_MPR_CALL), family 8 part 0, scheduler $v" >> $O/repro.txt
  LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/build_call/liblocohip_call_$v.so timeout 300 python tools/probes/r3/ht_kat_debug.py HumanoidTorque.run 1,16 2>&1 | tail -2 | cut -c1-260 >> $O/repro.txt
  LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/build_call/liblocohip_call_$v.so timeout 300 python bench.py --task HumanoidTorque.run --steps 60 --warmup 20 --sustained 0 --fuse 0 --no-cpu-baseline --configs off 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   bench ms %.3f replayed %d nan %d' % (d['ms_per_step'], d['stats']['replayed_env_steps'], d['stats']['nan_resets']))
" >> $O/repro.txt
done
echo "== fused kernel of family 8 with the DEFAULT scheduler (no SCHED_f8p1)" >> $O/repro.txt
LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/build_call/liblocohip_f8p1bad.so timeout 300 python tools/probes/r4/fused_vs_single.py HumanoidTorque.run 2>&1 | cut -c1-230 >> $O/repro.txt
echo "== shipped" >> $O/repro.txt
timeout 300 python tools/probes/r4/fused_vs_single.py HumanoidTorque.run 2>&1 | cut -c1-230 >> $O/repro.txt
cat $O/repro.txt
