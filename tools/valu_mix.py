"""
Static VALU instruction mix of one step kernel: flops per VALU lane-operation, from the gfx950 ISA of the built objects.

bench.py turns the profile's SQ_INSTS_VALU (wave-instructions per launch) into counted FP32 flops with it:
    flops per launch = SQ_INSTS_VALU x 64 lanes x flops_per_valu_lane_op
The mix is STATIC (every instruction of the kernel's text weighs the same, whatever its trip count): the dynamic mix of
a kernel whose hot loops are the solver's fma chains is a little richer in flops, so the figure is a slight under-count.
Weights per lane: v_fma / v_fmac / v_mac / v_mad / v_fmaak / v_fmamk (f32, f64) 2; v_pk_fma_f32 4; v_pk_mul_f32 / v_pk_add_f32 2;
v_add / v_sub / v_subrev / v_mul (f32, f64) 1; v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos 1; everything else
that issues on the vector ALU (moves, selects, compares, integer and address arithmetic, DPP / lane moves, conversions,
accumulator-register copies, min / max) 0.

  python tools/valu_mix.py "step_kernel<3, 6, false, false, 1, 0, 0, 4, false, 2>"        (prints JSON)
  python tools/valu_mix.py --annotate profiles/r5_pmc.json [...]     (adds "valu_mix" of the profile's kernel to each file; run
                                                                      where the objects are: csrc/build/ does not travel to the GPU box)
"""
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

_W2 = re.compile(r"^v_(fma|fmac|mac|mad|fmaak|fmamk|madak|madmk)_(f32|f64|legacy_f32)")
_W1 = re.compile(r"^v_(add|sub|subrev|mul|mul_legacy|rcp|rsq|sqrt|exp|log|sin|cos|rcp_iflag|exp_legacy|log_legacy)_(f32|f64)")


def weight(mn):
    if mn.startswith("v_pk_fma_f32"):
        return 4
    if mn.startswith("v_pk_mul_f32") or mn.startswith("v_pk_add_f32"):
        return 2
    if _W2.match(mn):
        return 2
    if _W1.match(mn):
        return 1
    return 0


def disassemble(obj, workdir):
    fat = os.path.join(workdir, "fat.bin")
    co = os.path.join(workdir, "dev.co")
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + fat, "--output=" + co])
    return subprocess.check_output([LLVM + "/llvm-objdump", "-d", "-C", co], text=True)


def kernel_mix(kernel, build_dir=None):
    """kernel: 'step_kernel<3, 6, false, false, 1, 0, 0, 4, false, 2>' (as rocprofv3 / the profile summary spell it)."""
    build_dir = build_dir or os.path.join(ROOT, "loco_mujoco_amd", "csrc", "build")
    want = "lmk::" + kernel + "("
    # the family is the first template argument's object: scan the family objects until the symbol shows up
    for obj in sorted(glob.glob(os.path.join(build_dir, "lm_family_f*p*.o"))):
        with tempfile.TemporaryDirectory() as wd:
            try:
                text = disassemble(obj, wd)
            except subprocess.CalledProcessError:
                continue
        if want not in text:
            continue
        counts = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "other": 0}
        flops = 0
        w_hist = {0: 0, 1: 0, 2: 0, 4: 0}
        inside = False
        for line in text.splitlines():
            if line.endswith(">:") and "<" in line and not line.startswith(" "):
                inside = want in line
                continue
            if not inside:
                continue
            parts = line.strip().split()
            if not parts:
                continue
            mn = parts[0]
            if mn.startswith("v_"):
                counts["valu"] += 1
                w = weight(mn)
                flops += w
                w_hist[w] += 1
            elif mn.startswith("s_"):
                counts["salu"] += 1
            elif mn.startswith("ds_"):
                counts["lds"] += 1
            elif mn.startswith(("global_", "buffer_", "scratch_", "flat_")):
                counts["vmem"] += 1
            else:
                counts["other"] += 1
        if counts["valu"] == 0:
            continue
        return dict(kernel=kernel, object=os.path.basename(obj), static_instructions=counts,
                    valu_by_flop_weight={str(k): v for k, v in w_hist.items()},
                    flops_per_valu_lane_op=flops / counts["valu"],
                    replicas=int(kernel.split("<")[1].split(">")[0].split(",")[7]) if kernel.count(",") >= 7 else 1)
    return None


if __name__ == "__main__":
    if sys.argv[1] == "--annotate":
        for path in sys.argv[2:]:
            prof = json.load(open(path))
            mix = kernel_mix(prof["kernel"])
            if mix is None:
                print("%s: kernel %s not found in the built objects" % (path, prof["kernel"]))
                continue
            prof["valu_mix"] = mix
            json.dump(prof, open(path, "w"), indent=1)
            print("%s: %s -> %.3f flops per VALU lane-op" % (path, prof["kernel"], mix["flops_per_valu_lane_op"]))
    else:
        print(json.dumps(kernel_mix(sys.argv[1]), indent=1))
