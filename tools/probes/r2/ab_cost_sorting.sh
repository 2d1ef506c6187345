#!/bin/bash
cd $GRAFT_REPO_ROOT
b() { python bench.py --steps 400 --warmup 40 --no-cpu-baseline $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %-44s %.3f ms  %.0f env-steps/s  fused %s iters/pass %.2f' % ('$1', d['ms_per_step'], d['value'], d.get('rollout_fused', {}).get('ms_per_step'), d['stats']['newton_iters_per_forward_pass']))"; }
b "A1 sorted" ""
LM_NO_SORT=1 b "A1 unsorted" ""
b "A1 16384 sorted" "--envs-per-gpu 16384 --steps 100"
LM_NO_SORT=1 b "A1 16384 unsorted" "--envs-per-gpu 16384 --steps 100"
b "A1 65536 sorted" "--envs-per-gpu 65536 --steps 50"
LM_NO_SORT=1 b "A1 65536 unsorted" "--envs-per-gpu 65536 --steps 50"
for t in HumanoidTorque.run Atlas.walk HumanoidMuscle.run Talos.walk; do b "$t sorted" "--task $t"; LM_NO_SORT=1 b "$t unsorted" "--task $t"; done
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 200 -k "sharding or fused or ragged or bench_two" 2>&1 | tail -3
