#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2_exp3
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
timeout 420 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s --timeout 200 > $OUT/suite_Os.log 2>&1
echo "=== current tree -Os: $(tail -1 $OUT/suite_Os.log)"
grep -E "^(FAILED|ERROR)" $OUT/suite_Os.log | cut -c1-220 | head -40
grep -E "4096 reachable|self-contact states|cylinder states" $OUT/suite_Os.log | cut -c1-900
for t in UnitreeA1.simple HumanoidTorque.run Atlas.walk HumanoidMuscle.run Talos.walk; do
  python bench.py --task $t --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %-22s %.3f ms  %.0f env-steps/s  fused %s  stats %s' % ('$t', d['ms_per_step'], d['value'], d.get('rollout_fused', {}).get('ms_per_step'), {k: v for k, v in d.get('stats', {}).items() if 'self' in k or 'overflow' in k or 'unhandled' in k}))"
done
for n in 16384 65536; do
  python bench.py --envs-per-gpu $n --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   A1 $n envs  %.3f ms  %.0f env-steps/s' % (d['ms_per_step'], d['value']))"
done
