#!/bin/bash
# Runs the device core (loco_mujoco_amd/csrc/lm_core.h, compiled for the CPU lane emulator) under MemorySanitizer and
# under AddressSanitizer + UBSan on start states of all four robots: the check that was used to tell a compiler problem
# (-O3 miscompare of the RK4 pyramid kernels on gfx950, see csrc/Makefile) from undefined behaviour in the source.
#   bash tools/sanitize_core.sh          (needs /opt/rocm/lib/llvm/bin/clang++ for MSan, g++ for ASan/UBSan)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
python - "$W" <<PY
import sys, struct, numpy as np
sys.path.insert(0, "$ROOT")
from loco_mujoco_amd import LocoEnv, lowering
for task, nu in [("HumanoidTorque.run", 13), ("Atlas.walk", 10), ("UnitreeA1.simple", 12), ("HumanoidMuscle.run", 92), ("Talos.walk", 12)]:
    np.random.seed(0)
    env = LocoEnv.make(task, debug=True)
    m = env._model
    cmod, _ = lowering.lower(m, env._device_task())
    tab = env._reset_table()
    with open("%s/%s.bin" % (sys.argv[1], task), "wb") as f:
        f.write(struct.pack("5i", len(cmod), m.nv, nu, getattr(m, "na", 0), 5))
        f.write(np.asarray(cmod, dtype=np.float64).tobytes())
        for i in range(5):
            row = tab[i * 9]
            f.write(np.asarray(row[:m.nv], dtype=np.float64).tobytes())
            f.write(np.asarray(row[m.nv:2 * m.nv], dtype=np.float64).tobytes())
            f.write(np.random.uniform(-1, 1, nu).astype(np.float64).tobytes())
PY
SRC="$ROOT/tests/emu/san_main.cpp $ROOT/tests/emu/emu.cpp"
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++20 -pthread -ffp-contract=off -fsanitize=memory -fsanitize-memory-track-origins \
    -DEMU_LS_POINTS=4 -DEMU_PYRAMID_ONLY -o $W/msan $SRC
g++ -O1 -std=c++20 -pthread -ffp-contract=off -fsanitize=undefined,address -fno-sanitize-recover=undefined \
    -DEMU_LS_POINTS=4 -DEMU_PYRAMID_ONLY -o $W/asan $SRC
# the replicated small-batch layout: 16 threads per environment, private lane memory reconciled at Q::fence()
g++ -O1 -std=c++20 -pthread -ffp-contract=off -fsanitize=undefined,address -fno-sanitize-recover=undefined \
    -DEMU_LS_POINTS=4 -DEMU_REP=4 -DEMU_PYRAMID_ONLY -o $W/asan_rep4 $SRC
for t in HumanoidTorque.run Atlas.walk UnitreeA1.simple HumanoidMuscle.run Talos.walk; do
  echo "== $t (MemorySanitizer)"; $W/msan $W/$t.bin
  echo "== $t (AddressSanitizer + UBSan)"; ASAN_OPTIONS=detect_leaks=0 $W/asan $W/$t.bin
  echo "== $t (AddressSanitizer + UBSan, 4 replicas)"; ASAN_OPTIONS=detect_leaks=0 $W/asan_rep4 $W/$t.bin
done
rm -rf $W
echo "sanitizers: clean"
