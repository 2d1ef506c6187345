"""
ctypes binding of the fp64 CPU oracle (oracle/oracle.c). TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by loco_mujoco_amd/.
"""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liblmoracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liblmoracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


class Contact(C.Structure):
    _fields_ = [("dist", C.c_double), ("pos", C.c_double * 3), ("frame", C.c_double * 9),
                ("includemargin", C.c_double), ("margin", C.c_double), ("friction", C.c_double * 5),
                ("solref", C.c_double * 2), ("solimp", C.c_double * 5), ("mu", C.c_double),
                ("dim", C.c_int), ("geom1", C.c_int), ("geom2", C.c_int), ("efc_address", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("ncon", C.c_int), ("nefc", C.c_int), ("solver_iter_total", C.c_int),
                ("solver_iter_max", C.c_int), ("unhandled_pairs", C.c_int), ("convex_contacts", C.c_int),
                ("max_self_depth", C.c_double), ("native_contacts", C.c_int),
                ("own_contacts", C.c_int), ("own_face_contacts", C.c_int)]


_DP = C.POINTER(C.c_double)


class ForwardOut(C.Structure):
    _fields_ = [(n, _DP) for n in ("M", "bias", "passive", "actuator", "qacc_smooth", "qacc", "qfrc_constraint",
                                   "xpos", "xmat", "geom_xpos")] + \
               [("contacts", C.POINTER(Contact)), ("max_con", C.c_int)] + \
               [(n, _DP) for n in ("efc_J", "efc_aref", "efc_R", "efc_force")] + \
               [("efc_type", C.POINTER(C.c_int)), ("max_efc", C.c_int), ("actuator_force", _DP), ("actuator_length", _DP),
                ("ncon", C.c_int), ("nefc", C.c_int), ("solver_iter", C.c_int), ("unhandled_pairs", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.lmo_model_create.restype = C.c_void_p
        _lib.lmo_model_create.argtypes = [_DP, C.c_long]
        _lib.lmo_model_destroy.argtypes = [C.c_void_p]
        _lib.lmo_set_option.argtypes = [C.c_void_p, C.c_int, C.c_double]
        _lib.lmo_set_mesh.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.lmo_step.argtypes = [C.c_void_p, _DP, _DP, _DP, _DP, C.c_int, C.POINTER(Stats)]
        _lib.lmo_forward.argtypes = [C.c_void_p, _DP, _DP, _DP, _DP, C.POINTER(ForwardOut)]
        _lib.lmo_step_act.argtypes = [C.c_void_p, _DP, _DP, _DP, _DP, _DP, C.c_int, C.POINTER(Stats)]
        _lib.lmo_forward_act.argtypes = [C.c_void_p, _DP, _DP, _DP, _DP, _DP, C.POINTER(ForwardOut)]
        _lib.lmo_na.argtypes = [C.c_void_p]
        _lib.lmo_step_contact_forces.argtypes = [C.c_void_p, _DP, _DP, _DP, _DP, _DP, _DP, C.c_int, C.POINTER(C.c_int)]
    return _lib


def _p(a):
    return a.ctypes.data_as(_DP)


class Oracle:
    MAX_CON, MAX_EFC = 96, 400

    def __init__(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.float64)
        self._h = lib().lmo_model_create(_p(blob), len(blob))
        if not self._h:
            raise ValueError("oracle rejected the model blob")
        self.nv = int(blob[3])
        self.nu = int(blob[5])
        self.nbody = int(blob[2])
        self.ngeom = int(blob[4])
        self.na = int(blob[19])

    def __del__(self):
        if getattr(self, "_h", None):
            lib().lmo_model_destroy(self._h)
            self._h = None

    def set_mesh(self, geom, vert):
        """Convex hull of a mesh geom (hull vertices [n, 3] in the frame of the geom's body): enables its plane contact."""
        v = np.ascontiguousarray(vert, dtype=np.float64)
        rc = lib().lmo_set_mesh(self._h, int(geom), len(v), v.ctypes.data_as(C.c_void_p))
        assert rc == 0

    def clear_mesh_graph(self, geom):
        """Switch the further plane-hull contacts of mesh geom ``geom`` off (single contact at the support vertex): tests."""
        lib().lmo_mesh_nvert.argtypes = [C.c_void_p, C.c_int]
        n = lib().lmo_mesh_nvert(self._h, int(geom))
        adr = np.zeros(n + 1, dtype=np.int32)
        nbr = np.zeros(1, dtype=np.int32)
        lib().lmo_set_mesh_graph.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double]
        assert lib().lmo_set_mesh_graph(self._h, int(geom), adr.ctypes.data, nbr.ctypes.data, 0.0) == 0

    def set_option(self, what, value):
        lib().lmo_set_option(self._h, {"disable_self_collision": 0, "iterations": 1, "tolerance": 2, "skip_pair_counter": 3, "disable_ccd": 4, "disable_native": 5}[what], float(value))

    def step(self, qpos, qvel, ctrl, nsub=1, warmstart=None):
        """Returns new (qpos, qvel, warmstart, stats-dict). Inputs are not modified."""
        q = np.array(qpos, dtype=np.float64)
        v = np.array(qvel, dtype=np.float64)
        c = np.ascontiguousarray(ctrl, dtype=np.float64)
        w = np.zeros(self.nv) if warmstart is None else np.array(warmstart, dtype=np.float64)
        st = Stats()
        rc = lib().lmo_step(self._h, _p(q), _p(v), _p(c), _p(w), int(nsub), C.byref(st))
        if rc != 0:
            raise ValueError("model has activation states: use step_act")
        return q, v, w, {n: getattr(st, n) for n, _ in Stats._fields_}

    def step_act(self, qpos, qvel, act, ctrl, nsub=1, warmstart=None):
        """Models with muscle activations: returns new (qpos, qvel, act, warmstart, stats-dict)."""
        q = np.array(qpos, dtype=np.float64)
        v = np.array(qvel, dtype=np.float64)
        a = np.array(act, dtype=np.float64)
        assert a.shape == (self.na,)
        c = np.ascontiguousarray(ctrl, dtype=np.float64)
        w = np.zeros(self.nv) if warmstart is None else np.array(warmstart, dtype=np.float64)
        st = Stats()
        rc = lib().lmo_step_act(self._h, _p(q), _p(v), _p(a), _p(c), _p(w), int(nsub), C.byref(st))
        if rc != 0:
            raise ValueError("oracle rejected the step (RK4 with activation states is not restated)")
        return q, v, a, w, {n: getattr(st, n) for n, _ in Stats._fields_}

    def step_contact_forces(self, qpos, qvel, ctrl, warmstart=None, act=None):
        """One substep; returns (qpos, qvel, act, warmstart, contacts) with contacts = [(geom1, geom2, f[3])] of the
        substep's last forward pass, in the engine's contact order."""
        q = np.array(qpos, dtype=np.float64)
        v = np.array(qvel, dtype=np.float64)
        a = np.zeros(max(self.na, 1)) if act is None else np.array(act, dtype=np.float64)
        c = np.ascontiguousarray(ctrl, dtype=np.float64)
        w = np.zeros(self.nv) if warmstart is None else np.array(warmstart, dtype=np.float64)
        out = np.zeros((self.MAX_CON, 5))
        n = C.c_int(0)
        rc = lib().lmo_step_contact_forces(self._h, _p(q), _p(v), _p(a), _p(c), _p(w), _p(out), self.MAX_CON, C.byref(n))
        if rc != 0:
            raise ValueError("oracle rejected the step")
        return q, v, a[:self.na], w, [(int(r[0]), int(r[1]), r[2:5].copy()) for r in out[:n.value]]

    def forward(self, qpos, qvel, ctrl, warmstart=None, act=None):
        nv = self.nv
        q = np.ascontiguousarray(qpos, dtype=np.float64)
        v = np.ascontiguousarray(qvel, dtype=np.float64)
        c = np.ascontiguousarray(ctrl, dtype=np.float64)
        w = np.zeros(nv) if warmstart is None else np.ascontiguousarray(warmstart, dtype=np.float64)
        res = dict(M=np.zeros((nv, nv)), bias=np.zeros(nv), passive=np.zeros(nv), actuator=np.zeros(nv),
                   qacc_smooth=np.zeros(nv), qacc=np.zeros(nv), qfrc_constraint=np.zeros(nv),
                   xpos=np.zeros((self.nbody, 3)), xmat=np.zeros((self.nbody, 9)), geom_xpos=np.zeros((self.ngeom, 3)),
                   efc_J=np.zeros((self.MAX_EFC, nv)), efc_aref=np.zeros(self.MAX_EFC), efc_R=np.zeros(self.MAX_EFC),
                   efc_force=np.zeros(self.MAX_EFC), actuator_force=np.zeros(self.nu), actuator_length=np.zeros(self.nu))
        out = ForwardOut()
        for k, a in res.items():
            setattr(out, k, _p(a))
        cons = (Contact * self.MAX_CON)()
        etype = np.zeros(self.MAX_EFC, dtype=np.int32)
        out.contacts = cons
        out.max_con = self.MAX_CON
        out.efc_type = etype.ctypes.data_as(C.POINTER(C.c_int))
        out.max_efc = self.MAX_EFC
        av = np.zeros(max(self.na, 1)) if act is None else np.ascontiguousarray(act, dtype=np.float64)
        lib().lmo_forward_act(self._h, _p(q), _p(v), _p(av), _p(c), _p(w), C.byref(out))
        ne, nc = out.nefc, out.ncon
        for k in ("efc_J", "efc_aref", "efc_R", "efc_force"):
            res[k] = res[k][:ne]
        res["efc_type"] = etype[:ne]
        res["contacts"] = [dict(dist=cons[i].dist, pos=np.array(cons[i].pos), frame=np.array(cons[i].frame).reshape(3, 3),
                                dim=cons[i].dim, geom1=cons[i].geom1, geom2=cons[i].geom2, mu=cons[i].mu,
                                friction=np.array(cons[i].friction), efc_address=cons[i].efc_address)
                           for i in range(nc)]
        res.update(ncon=nc, nefc=ne, solver_iter=out.solver_iter, unhandled_pairs=out.unhandled_pairs)
        return res


def native_pair(t1, p1, R1, s1, t2, p2, R2, s2, margin):
    """One of the oracle's native box / cylinder colliders on its own (tests): returns [(dist, pos, normal)] or None when the pair
    type has no native collider. Geom types as in include/lm_layout.h, R 3x3 rotation (columns = geom axes)."""
    f = lambda x, n: np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1)[:n] if np.size(x) >= n else np.pad(np.asarray(x, dtype=np.float64).reshape(-1), (0, n - np.size(x))))
    out = np.zeros((8, 7))
    a = [f(p1, 3), f(R1, 9), f(s1, 3), f(p2, 3), f(R2, 9), f(s2, 3)]
    lib().lmo_test_native_pair.argtypes = [C.c_int, _DP, _DP, _DP, C.c_int, _DP, _DP, _DP, C.c_double, _DP]
    n = lib().lmo_test_native_pair(int(t1), _p(a[0]), _p(a[1]), _p(a[2]), int(t2), _p(a[3]), _p(a[4]), _p(a[5]), float(margin), _p(out))
    if n < 0:
        return None
    return [(out[i, 0], out[i, 1:4].copy(), out[i, 4:7].copy()) for i in range(n)]
