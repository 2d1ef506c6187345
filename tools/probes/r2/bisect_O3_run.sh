#!/bin/bash
# GPU side of bisect_O3.sh: the parity test of the kernel under every library in variants/f5/
cd $GRAFT_REPO_ROOT
for v in $(ls variants/f5/*.so); do
  export LOCOHIP_LIB=$GRAFT_REPO_ROOT/$v
  r=$(timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=no -p no:cacheprovider -s --timeout 100 -k "per_environment_joint_parameters and HumanoidMuscle" 2>&1 | grep -E "vs oracle|passed|failed" | tr '\n' ' ')
  echo "$v: $r"
done
