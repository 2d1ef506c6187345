#!/bin/bash
# variant.sh <name> "<extra compile flags>" "<families>" ["<parts>"] — an A/B library liblocohip_<name>.so: the objects of the listed kernel families
# (default parts: 0) compiled with the extra flags into csrc/build_<name>/, everything else taken from the main build (csrc/build/).
set -e
cd "$(dirname "$0")/../../../loco_mujoco_amd/csrc"
NAME="$1"; EXTRA="$2"; FAMS="$3"; PARTS="${4:-0}"
B=build_$NAME; mkdir -p $B
FLAGS="--offload-arch=gfx950 -Os -std=c++17 -fPIC -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -Wno-unused-result -Wno-unused-value"
pids=()
for f in $FAMS; do for p in $PARTS; do
  /opt/rocm/bin/hipcc $FLAGS $EXTRA -DLM_FAMILY=$f -DLM_PART=$p -c -o $B/lm_family_f${f}p${p}.o lm_family.hip & pids+=($!)
done; done
for p in "${pids[@]}"; do wait $p; done
OBJS=""
for o in build/*.o; do b=$(basename $o); if [ -f $B/$b ]; then OBJS="$OBJS $B/$b"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblocohip_$NAME.so $OBJS
ls -la liblocohip_$NAME.so
