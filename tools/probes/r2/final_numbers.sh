#!/bin/bash
# Round 2, final library: the bench line of every configuration (with the CPU baseline), the large batches, smoke(), then the
# rocprofv3 passes of three configurations. Everything lands in gpurun_out/r2_final/ and gpurun_out/profiles/.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2_final
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
# the rocprofv3 passes first (SKIP_PROF=1: bench lines only), installed into profiles/ of THIS copy so that the bench lines below
# quote the counters of the library they run (profiles/<tag>_pmc.json carries the sha256 of liblocohip.so)
if [ -z "$SKIP_PROF" ]; then
  bash tools/probes/prof_run.sh r2 200 > $OUT/prof_a1.log 2>&1
  bash tools/probes/prof_run.sh r2_HumanoidTorque.run 100 "--task HumanoidTorque.run" > $OUT/prof_ht.log 2>&1
  bash tools/probes/prof_run.sh r2_Atlas.walk.dr2048 100 "--task Atlas.walk --dr --envs-per-gpu 2048" > $OUT/prof_atlas.log 2>&1
  cp gpurun_out/profiles/r2_* profiles/
  ls gpurun_out/profiles
fi
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json
for t in HumanoidTorque.run Atlas.walk HumanoidMuscle.run Talos.walk UnitreeH1.walk UnitreeG1.walk; do
  timeout 240 python bench.py --task $t --steps 300 --warmup 30 > $OUT/bench_$t.json 2> $OUT/bench_$t.err
done
timeout 240 python bench.py --task Atlas.walk --dr --envs-per-gpu 2048 --steps 300 --warmup 30 > $OUT/bench_Atlas.walk.dr2048.json 2> /dev/null
timeout 240 python bench.py --task HumanoidMuscle.run --envs-per-gpu 2048 --steps 300 --warmup 30 > $OUT/bench_HumanoidMuscle.run.2048.json 2> /dev/null
for n in 16384 65536; do
  timeout 240 python bench.py --envs-per-gpu $n --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_a1_$n.json 2> /dev/null
done
for f in $OUT/bench*.json; do python -c "
import sys, json
d = json.loads(open('$f').read().strip().splitlines()[-1])
print('%-40s %.3f ms  %.0f env-steps/s  fused %s  cpu %s' % ('$f'.split('/')[-1], d['ms_per_step'], d['value'], (d.get('rollout_fused') or {}).get('ms_per_step'), (d.get('cpu_baseline') or {}).get('value')))
"; done
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
