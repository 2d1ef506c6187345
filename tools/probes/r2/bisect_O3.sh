#!/bin/bash
# Round 2: which LLVM pass turns the -O3 build of ONE kernel (<5 links, 4 slots, Euler, pyramids, 48 muscles, per-environment
# parameters, 4 replicas> = family 5, part 1) into one that fails its parity test. Builds that object with
# `-mllvm -opt-bisect-limit=N` for each N given (8 at a time fit the build box), links each into a library next to the -Os
# objects of the other families (variants/f5/b<N>.so), and the companion run script tests every library on the GPU:
#   bash tools/probes/r2/bisect_O3.sh 2300 4600 ...        (build box)
#   gpurun -- 'bash tools/probes/r2/bisect_O3_run.sh'       (GPU box)
# Outcome (profiles/r2_ab_probes.md §5): limit 18196 passes, 18197 fails = "machine-scheduler" (GCN max-occupancy, pre-RA)
# on that one function.
cd "$(dirname "$0")/../../../loco_mujoco_amd/csrc" || exit 1
ROOT=$(cd ../.. && pwd)
FL="--offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -Wno-unused-result -Wno-unused-value -DLM_FAMILY=5 -DLM_PART=1"
rm -rf $ROOT/variants/f5; mkdir -p $ROOT/variants/f5 /tmp/f5
OBJS=$(ls build/lm_kernels.o build/lm_family_f*.o | grep -v f5p1)
for n in "$@"; do
  ( /opt/rocm/bin/hipcc $FL -O3 -mllvm -opt-bisect-limit=$n -c -o /tmp/f5/b$n.o lm_family.hip 2> /dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/variants/f5/b$n.so $OBJS /tmp/f5/b$n.o ) &
done
wait
ls $ROOT/variants/f5
