"""ctypes wrapper of the CPU lane emulator (tests/emu/emu.cpp) — test tooling only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_VARIANTS = [(1, 0, 1), (4, 0, 1), (4, 1, 1), (4, 0, 4), (4, 1, 4), (4, 2, 4)]          # what the tests use: built together, in parallel (g++ needs ~1 min each)


def _lib_path(ls_points, dr, rep=1):
    name = ("libemu.so" if ls_points == 1 else "libemu%d.so" % ls_points).replace(".so", {0: ".so", 1: "_dr.so", 2: "_drv.so"}[int(dr)])
    return os.path.join(_HERE, name.replace(".so", "_rep%d.so" % rep if rep > 1 else ".so"))


def build(ls_points=1, dr=False, rep=1):
    """ls_points = 1: the one-point-at-a-time line search of full waves; 4: the four-points-per-round line search of the
    replicated small-batch layout (evaluated by one lane here); dr: 1 = the per-environment joint-parameter code path, 2 = with
    model variants (the kernels' DR template level)."""
    srcs = [os.path.join(_HERE, "emu.cpp"), os.path.join(_HERE, "../../loco_mujoco_amd/csrc/lm_core.h"),
            os.path.join(_HERE, "../../include/lm_layout.h")]

    def stale(lib):
        return not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs)

    dr = int(dr)
    want = [(ls_points, dr, rep)] + [v for v in _VARIANTS if v != (ls_points, dr, rep)]
    procs = []
    for lp, d, rp in want:
        lib = _lib_path(lp, d, rp)
        if stale(lib):
            tmp = lib + ".tmp%d" % os.getpid()
            procs.append((subprocess.Popen(["g++", "-O2", "-std=c++20", "-pthread", "-fPIC", "-shared", "-ffp-contract=off",
                                            "-DEMU_LS_POINTS=%d" % lp, "-DEMU_REP=%d" % rp, "-DEMU_PYRAMID_ONLY"] + (["-DEMU_DR=%d" % d] if d else [])
                                           + ["-o", tmp, srcs[0]]), tmp, lib))
    for p, tmp, lib in procs:
        if p.wait() != 0:
            raise RuntimeError("building %s failed" % lib)
        os.replace(tmp, lib)
    return _lib_path(ls_points, dr, rep)


def run(chain_model, qpos, qvel, action, nsub=1, warm=None, debug_env=-1, act=None, ls_points=1, dof_params=None, dr=False, rep=1,
        variant=None, replay=True):
    """variant: (record, geom table, geom-pair table) of ``lowering.variant_tables`` — one model variant for all environments
    (implies dr). replay: the library's speculate / replay protocol (a control step beyond the regular instantiation's capacity is run by
    the big one; the counters' "replayed" says for how many environments); False = the regular instantiation alone. dof_params: (3, n, nv) per-environment damping / stiffness / frictionloss (implies dr); dr=True alone runs the
    per-environment code path on the table's nominal values."""
    dr = 2 if variant is not None else int(bool(dr or dof_params is not None))
    if rep > 1:
        ls_points = 4            # the replicated layout always evaluates four step lengths per round
    lib = C.CDLL(build(ls_points, dr, rep))
    cmod = np.ascontiguousarray(chain_model, dtype=np.float64)
    nv = int(cmod[2])
    q = np.array(qpos, dtype=np.float64).reshape(-1, nv)
    v = np.array(qvel, dtype=np.float64).reshape(-1, nv)
    n = q.shape[0]
    w = np.zeros_like(q) if warm is None else np.array(warm, dtype=np.float64).reshape(n, nv)
    a = np.ascontiguousarray(action, dtype=np.float64).reshape(n, -1)
    M = np.zeros((nv, nv), dtype=np.float32)
    d5 = np.zeros((5, nv), dtype=np.float32)
    cnt = np.zeros(8, dtype=np.int32)
    dp = lambda x: x.ctypes.data_as(C.c_void_p)
    na = int(cmod[31])        # H_NMUSCLE
    actv = None
    if na:
        actv = np.zeros((n, na)) if act is None else np.array(act, dtype=np.float64).reshape(n, na)
    prm = None
    if dr:
        prm = None if dof_params is None else np.ascontiguousarray(dof_params, dtype=np.float64).reshape(3, n, nv)
        lib.emu_set_dof_params(dp(prm) if prm is not None else None)
        vt = None
        if variant is not None:
            vt = [np.ascontiguousarray(t, dtype=np.float32) for t in variant]
            lib.emu_set_model_variant(dp(vt[0]), dp(vt[1]), dp(vt[2]) if len(vt[2]) else None)
        else:
            lib.emu_set_model_variant(None, None, None)
    replayed = np.zeros(n, dtype=np.int32)
    rc = lib.emu_run2(dp(cmod), n, dp(q), dp(v), dp(w), dp(a), int(nsub), int(debug_env),
                      dp(M) if debug_env >= 0 else None, dp(d5) if debug_env >= 0 else None, dp(cnt),
                      dp(actv) if actv is not None else None, int(replay), dp(replayed))
    assert rc == 0, "emulator returned %d (-2: two replicas changed the same lane-memory word to different values)" % rc
    dbg = dict(M=M, bias=d5[0], smooth=d5[1], qacc_smooth=d5[2], qacc=d5[3], qfrc_constraint=d5[4])
    if actv is not None:
        dbg["act"] = actv
    return q, v, w, dict(solver_iters=int(cnt[0]), overflow=int(cnt[1]), unhandled=int(cnt[2]), ncon=int(cnt[3]), ls_evals=int(cnt[4]), ls_capped=int(cnt[5]), selfprox=int(cnt[6]), selfcon=int(cnt[7]), replayed=int(replayed.sum())), dbg
