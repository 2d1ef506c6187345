#!/bin/bash
# Solver stopping rules vs speed and parity: LM_TOLERANCE (scaled gradient / Newton decrement), LM_LS_TOL (relative slope of
# the exact line search). Parity = worst KAT error against the golden rows (A1, Atlas, HumanoidTorque), printed by the tests.
for cfg in "1e-6 1e-2" "3e-6 1e-2" "1e-5 1e-2" "3e-5 1e-2" "1e-6 3e-2" "1e-6 1e-1" "1e-5 3e-2" "1e-5 1e-1"; do
  set -- $cfg
  export LM_TOLERANCE=$1 LM_LS_TOL=$2
  par=$(python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "one_control_step_kats and not 4_ages and not carry" 2>&1 | grep -E "KAT errors" | sed -E 's/.*(qpos max [0-9.e+-]+).*(qvel max [0-9.e+-]+).*/\1 \2/' | tr '\n' ';')
  for t in UnitreeA1.simple HumanoidTorque.run; do
    python bench.py --task $t --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tol $1 ls_tol $2 %-20s %.4f ms fused %.4f ms its/fwd %.3f' % ('$t', d['ms_per_step'], d['rollout_fused']['ms_per_step'], d['stats']['newton_iters_per_forward_pass']))"
  done
  echo "   parity: $par"
done
