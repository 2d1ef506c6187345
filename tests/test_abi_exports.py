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
