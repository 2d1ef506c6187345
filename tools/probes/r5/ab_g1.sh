#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5r; mkdir -p $O
P='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("ms %.4f value %.5g iters/pass %.3f" % (d["ms_per_step"], d["value"], d["stats"]["newton_iters_per_forward_pass"]))
'
for rep in 1 2; do
for v in old jar new; do
  echo "== G1 $v (run $rep)" >> $O/ab.txt
  LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/build_v/lib_$v.so timeout 300 python bench.py --task UnitreeG1.walk --steps 150 --warmup 30 --fuse 0 --sustained 0 --configs off --no-cpu-baseline 2>&1 | python -c "$P" >> $O/ab.txt
done
done
cat $O/ab.txt
