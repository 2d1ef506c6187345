#!/bin/bash
# A/B of compiler flags on one kernel family: each tools/probes/_bin/t_*.so replaces the library, dr_kernel_vs_oracle.py reports the
# worst per-environment velocity error of the per-environment-parameter kernel against the oracle.
cp loco_mujoco_amd/csrc/liblocohip.so /tmp/liblocohip_full.so
for f in tools/probes/_bin/t_*.so; do
  cp $f loco_mujoco_amd/csrc/liblocohip.so
  r=$(python tools/probes/dr_kernel_vs_oracle.py 2>&1 | grep -E "^(nominal-params|dr|nominal) " | awk '{ if ($4+0 > m[$1]) m[$1] = $4+0 } END { for (k in m) printf "%s %.1e  ", k, m[k] }')
  echo "$(basename $f): $r"
done
cp /tmp/liblocohip_full.so loco_mujoco_amd/csrc/liblocohip.so
