#!/bin/bash
# A/B of two builds of the library on the same box: ab_libs.sh <out dir> <lib A> <lib B> ["task [bench args]" ...]
OUT=gpurun_out/$1; A=$2; B=$3; shift 3; mkdir -p $OUT
if [ $# -eq 0 ]; then set -- "UnitreeA1.simple" "HumanoidTorque.run" "UnitreeH1.run" "HumanoidMuscle.run --envs-per-gpu 2048" "Atlas.walk"; fi
for t in "$@"; do
  for lib in $A $B; do
    tag=$(echo $t | tr ' ' '_' | tr -d '-').$(basename $lib .so)
    LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/$lib python bench.py --task $t --steps ${STEPS:-300} --warmup 60 --no-cpu-baseline > $OUT/$tag.json 2>> $OUT/err.log
    python - $OUT/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-52s %9.0f env-steps/s %7.3f ms  fused %7.3f ms  overflow %d selfcon %d replayed %d" % (sys.argv[2], d["value"], d["ms_per_step"], d.get("rollout_fused", {}).get("ms_per_step", 0),
      d["stats"]["overflow_contacts"], d["stats"]["self_contacts"], d["stats"].get("replayed_env_steps", -1)))
PY
  done
done
