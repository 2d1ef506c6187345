#!/bin/bash
# The quadruped's bench line on builds of its family with other code-generation switches (same box): scheduler strategies, -O2,
# five slots, the regular kernel without the inlined convex collider. `parity` = device vs oracle on 64 dataset states.
OUT=gpurun_out/a1_variants; mkdir -p $OUT
for round in 1 2; do
for lib in liblocohip.so liblocohip_v_ilp.so liblocohip_v_iter.so liblocohip_v_o2.so liblocohip_v_ns5.so liblocohip_v_pm2.so; do
  [ -f loco_mujoco_amd/csrc/$lib ] || continue
  LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/$lib python bench.py --steps 600 --warmup 100 --fuse 0 > $OUT/$lib.$round.json 2>> $OUT/err.log
  python - $OUT/$lib.$round.json $lib <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-26s %9.0f env-steps/s %7.4f ms  kernel %7.4f ms  parity %.2e / %.2e %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_launch"], d["parity"]["qpos_linf"], d["parity"]["qvel_linf"], d["parity"]["within_tolerance"]))
PY
done
done
