#!/bin/bash
cd $GRAFT_REPO_ROOT
b() { python bench.py --steps 300 --warmup 30 --no-cpu-baseline $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %-40s %.3f ms  %.0f env-steps/s  fused %s' % ('$1', d['ms_per_step'], d['value'], d.get('rollout_fused', {}).get('ms_per_step')))"; }
b "default" ""
LM_NO_PAIRS=1 b "pairs compiled, link-pair list empty" ""
for v in variants/a1_*.so; do LOCOHIP_LIB=$PWD/$v b "$v" ""; done
