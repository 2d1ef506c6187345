#!/usr/bin/env python3
"""kernel_resources.py <round tag> — register / scratch / spill figures of the profiled step kernels from the COMPILER's own analysis
(-Rpass-analysis=kernel-resource-usage on ONE instantiation each, the Makefile's flags): profiles/<tag>_kernel_resources.json.
rocprofv3's `vgpr_count` / `accum_vgpr_count` columns do not describe these kernels (round-5 review: 248 / 0 for a kernel the compiler
allocates 256 + 237 for); tools/summarize_rocprof.py takes the figures from this file. Runs on the build machine (no GPU)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ("--offload-arch=gfx950 -Os -std=c++17 -fPIC -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt "
         "-fgpu-flush-denormals-to-zero -Wno-unused-result -Wno-unused-value -Wno-cuda-compat").split()
# step_kernel<MC, NS, RK4, FWD, CONE, NM, DR, REP, FUSED, PM> as rocprofv3 prints them -> template arguments
KERNELS = {
    "step_kernel<3, 6, false, false, 1, 0, 0, 4, false, 2>": "3,6,false,false,1,0,0,4,false,2",          # quadruped, the bench line
    "step_kernel<5, 8, true, false, 0, 0, 0, 4, false, 1>": "5,8,true,false,0,0,0,4,false,1",            # HumanoidTorque regular
    "step_kernel<5, 128, true, false, 0, 0, 0, 4, true, 1>": "5,128,true,false,0,0,0,4,true,1",          # ... its replay kernel
    "step_kernel<5, 8, true, false, 0, 0, 1, 4, false, 0>": "5,8,true,false,0,0,1,4,false,0",            # Atlas with per-environment joint parameters
    "step_kernel<5, 8, false, false, 0, 48, 0, 4, false, 1>": "5,8,false,false,0,48,0,4,false,1",        # HumanoidMuscle
    "step_kernel<6, 8, false, false, 0, 0, 0, 4, false, 3>": "6,8,false,false,0,0,0,4,false,3",          # UnitreeG1 regular (detection only)
}


def main():
    tag = sys.argv[1]
    out = {}
    for name, args in KERNELS.items():
        cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ["-DK_ARGS=" + args, "-I", os.path.join(ROOT, "loco_mujoco_amd", "csrc"), "--cuda-device-only", "-c", "-o", "/dev/null",
               os.path.join(ROOT, "tools", "probes", "r6", "one_kernel.hip"), "-Rpass-analysis=kernel-resource-usage"]
        txt = subprocess.run(cmd, capture_output=True, text=True).stderr
        def field(label):
            m = re.search(re.escape(label) + r":\s*(\d+)", txt)
            return int(m.group(1)) if m else None
        out[name] = dict(vgpr=field("VGPRs"), agpr=field("AGPRs"), sgpr=field("TotalSGPRs"), scratch_bytes_per_lane=field("ScratchSize [bytes/lane]"),
                         vgpr_spills=field("VGPRs Spill"), sgpr_spills=field("SGPRs Spill"), occupancy_waves_per_simd=field("Occupancy [waves/SIMD]"))
        print(name, out[name])
    tc = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.splitlines()[:2]
    json.dump({"toolchain": tc, "flags": " ".join(FLAGS), "kernels": out, "source": "hipcc -Rpass-analysis=kernel-resource-usage, tools/kernel_resources.py"},
              open(os.path.join(ROOT, "profiles", tag + "_kernel_resources.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
