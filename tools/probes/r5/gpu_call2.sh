#!/bin/bash
# round 5, second GPU call: the whole GPU suite on the sixteen-replica replay kernels, then the hand-off thresholds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/log.txt
tail -5 $O/pytest.txt >> $O/log.txt
B="python bench.py --steps 100 --warmup 30 --sustained 0 --fuse 0 --no-cpu-baseline --configs off"
for h in 0,0,0 6,10,0 5,6,0 4,4,0 5,6,8 6,10,10; do
  echo "== HT.run handoff $h" >> $O/sweep.txt
  timeout 300 $B --task HumanoidTorque.run --handoff $h 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms %.3f value %.4g replayed %d overflow %d' % (d['ms_per_step'], d['value'], d['stats']['replayed_env_steps'], d['stats']['overflow_contacts']))
" >> $O/sweep.txt 2>&1
done
for h in 0,0,0 0,0,5 0,0,7 0,0,9 0,0,12; do
  echo "== A1 handoff $h" >> $O/sweep.txt
  timeout 300 python bench.py --steps 300 --warmup 50 --sustained 0 --fuse 0 --no-cpu-baseline --configs off --handoff $h 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms %.4f value %.4g replayed %d' % (d['ms_per_step'], d['value'], d['stats']['replayed_env_steps']))
" >> $O/sweep.txt 2>&1
done
for t in Atlas.walk UnitreeG1.walk Talos.walk UnitreeH1.run; do
  for h in 0,0,0 5,6,8; do
  echo "== $t handoff $h" >> $O/sweep.txt
  timeout 300 $B --task $t --handoff $h 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms %.3f value %.4g replayed %d overflow %d' % (d['ms_per_step'], d['value'], d['stats']['replayed_env_steps'], d['stats']['overflow_contacts']))
" >> $O/sweep.txt 2>&1
  done
done
cat $O/log.txt; cat $O/sweep.txt
for v in bad ilp; do
  echo "== mm $v" >> $O/mm.txt
  LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/build_mm/liblocohip_mm_$v.so timeout 300 python tools/probes/r4/fused_vs_single.py HumanoidTorque.run >> $O/mm.txt 2>&1
done
cat $O/mm.txt | cut -c1-230
