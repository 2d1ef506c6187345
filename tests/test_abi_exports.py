"""The C-ABI library builds for gfx950, loads, and exports every symbol include/locohip.h declares (no GPU needed)."""

import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_all_declared_symbols():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "loco_mujoco_amd", "csrc"), "liblocohip.so"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    header = open(os.path.join(ROOT, "include", "locohip.h")).read()
    declared = set(re.findall(r"\b(lm_[a-z_]+)\s*\(", header))
    assert len(declared) >= 17
    lib = ctypes.CDLL(os.path.join(ROOT, "loco_mujoco_amd", "csrc", "liblocohip.so"))
    for name in declared:
        assert hasattr(lib, name), name
    from loco_mujoco_amd import backend
    assert set(backend.EXPORTS) == declared


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
