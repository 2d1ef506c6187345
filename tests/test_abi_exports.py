"""The C-ABI library builds for gfx950, loads, and exports every symbol include/locohip.h declares (no GPU needed)."""

import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_all_declared_symbols():
    subprocess.check_call(["make", "-j%d" % (os.cpu_count() or 4), "-C", os.path.join(ROOT, "loco_mujoco_amd", "csrc"), "liblocohip.so"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    header = open(os.path.join(ROOT, "include", "locohip.h")).read()
    declared = set(re.findall(r"\b(lm_[a-z_]+)\s*\(", header))
    assert len(declared) >= 17
    lib = ctypes.CDLL(os.path.join(ROOT, "loco_mujoco_amd", "csrc", "liblocohip.so"))
    for name in declared:
        assert hasattr(lib, name), name
    from loco_mujoco_amd import backend
    assert set(backend.EXPORTS) == declared
    # ... and nothing of its own beyond them (the lm_debug_* entry points of the profiling builds exist under -DLM_TIMERS only)
    syms = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "loco_mujoco_amd", "csrc", "liblocohip.so")], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (lm_[a-z_0-9]+)$", syms, flags=re.M))
    assert exported == declared, sorted(exported ^ declared)


# the toolchain the GPU suite was last run green on (round 6): ROCm 7.2.0's hipcc, the library at -Os
VALIDATED_TOOLCHAIN = ("HIP version: 7.2.26015-fc0010cf6a", "roc-7.2.0 26014 7b800a19466229b8479a78de19143dc33c3ab9b5", "| -Os")


def test_library_records_the_toolchain_it_was_validated_on():
    """The step kernels sit at the 512-register ceiling, where a code-generation defect of this toolchain was met and fenced by tests
    (csrc/Makefile, profiles/r5_notes.md §5): the library says what built it, and a build by another compiler or at another optimisation
    level fails HERE — re-run the GPU suite (bitwise fused-vs-single and replay-vs-regular tests, the -O2 guard) and update the strings."""
    lib = ctypes.CDLL(os.path.join(ROOT, "loco_mujoco_amd", "csrc", "liblocohip.so"))
    lib.lm_toolchain.restype = ctypes.c_char_p
    got = lib.lm_toolchain().decode()
    for part in VALIDATED_TOOLCHAIN:
        assert part in got, (part, got)


def test_shipped_library_reads_no_environment_variable():
    """The A/B switches of the probe builds (LM_NO_PAIRS, LM_TOLERANCE, LM_ABLATE, LM_LS_*, LM_NO_REPLICAS, LM_GENERIC_KERNELS,
    LM_NO_XCD_MAP, LM_ENVS_PER_BLOCK) exist only under -DLM_PROBES: the default library does not import getenv at all."""
    lib = os.path.join(ROOT, "loco_mujoco_amd", "csrc", "liblocohip.so")
    syms = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True, check=True).stdout
    assert "hipLaunchKernel" in syms or "hipModuleLaunchKernel" in syms or "__hipRegisterFunction" in syms     # (the listing is the real one)
    assert "getenv" not in syms
    src = open(os.path.join(ROOT, "loco_mujoco_amd", "csrc", "lm_kernels.hip")).read()
    assert src.count("getenv(") == 1 and "#define LM_PROBE_ENV(name) getenv(name)" in src


def test_model_create_rejects_bad_input():
    from loco_mujoco_amd import backend
    import numpy as np
    lib = backend.load_library()
    h = ctypes.c_void_p()
    bad = np.zeros(10)
    rc = lib.lm_model_create(bad.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(bad), 0, ctypes.byref(h))
    assert rc != 0 and b"too short" in lib.lm_last_error()


def test_layout_header_is_in_sync():
    import sys
    sys.path.insert(0, ROOT)
    before = open(os.path.join(ROOT, "include", "lm_layout.h")).read()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_layout_header.py")], stdout=subprocess.DEVNULL)
    assert open(os.path.join(ROOT, "include", "lm_layout.h")).read() == before
