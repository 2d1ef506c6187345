#!/bin/bash
# the quadruped's bench line: round 4's tree (its own bench.py, package and library, staged under gpurun_r4/) against this round's, same box, alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l; mkdir -p $O
P='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("ms %.4f value %.5g" % (d["ms_per_step"], d["value"]))
'
for rep in 1 2 3; do
  echo "== r4 (run $rep)" >> $O/ab.txt
  (cd gpurun_r4 && timeout 300 python bench.py --steps 600 --warmup 100 --no-cpu-baseline --fuse 0 --sustained 0 2>&1 | python -c "$P") >> $O/ab.txt
  echo "== r5 (run $rep)" >> $O/ab.txt
  timeout 300 python bench.py --steps 600 --warmup 100 --no-cpu-baseline --fuse 0 --sustained 0 --configs off 2>&1 | python -c "$P" >> $O/ab.txt
done
cat $O/ab.txt
