#!/bin/bash
# code-generation switches for the quadruped's bench kernel (family 0 part 0 rebuilt per variant, linked on the GPU box): ms per control step
cd $GRAFT_REPO_ROOT/loco_mujoco_amd/csrc/build_fl
O=$GRAFT_REPO_ROOT/gpurun_out/r5m; mkdir -p $O
base=$(ls *.o | grep -v "^f0p0_")
P='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("ms %.4f value %.5g parity %s" % (d["ms_per_step"], d["value"], d.get("parity", {}).get("within_tolerance")))
'
for rep in 1 2; do
for o in f0p0_*.o; do
  v=$(basename $o .o); v=${v#f0p0_}
  [ -f /tmp/lib_$v.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_$v.so $base $o
  echo "== $v (run $rep)" >> $O/lottery.txt
  (cd $GRAFT_REPO_ROOT && LOCOHIP_LIB=/tmp/lib_$v.so timeout 200 python bench.py --steps 500 --warmup 80 --no-cpu-baseline --fuse 0 --sustained 0 --configs off 2>&1 | python -c "$P") >> $O/lottery.txt
done
done
cat $O/lottery.txt
