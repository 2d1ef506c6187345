#!/bin/bash
# A/B of prebuilt library variants (variants/*.so, built here with different flags) on the bench workload + humanoid rollouts.
# usage: bash tools/probes/ab_variants.sh <out.log> <variant.so> [<variant.so> ...]
OUT=$1; shift
: > $OUT
for V in "$@"; do
  echo "== $V" >> $OUT
  LOCOHIP_LIB=$PWD/$V python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A1 4096: %.3f ms/step %.0f env-steps/s' % (d['ms_per_step'], d['value']))" >> $OUT 2>&1
  LOCOHIP_LIB=$PWD/$V python -m pytest tests/test_gpu_parity.py -q -s -k "batch_rollout_properties or one_control_step_kats" 2>&1 | grep -E "envs:|KAT errors|passed|failed" >> $OUT
done
cat $OUT
