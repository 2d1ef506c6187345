"""
ctypes loader for ``liblocohip.so`` (C-ABI: ``include/locohip.h``) — the ONLY physics path of the
product. There is no CPU fallback: if the library or a GPU is missing this module raises.
"""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LOCOHIP_LIB", os.path.join(_HERE, "csrc", "liblocohip.so"))   # override: kernel A/B builds

_F = C.POINTER(C.c_float)
_U8 = C.POINTER(C.c_uint8)
_D = C.POINTER(C.c_double)


class Dims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("nq", "nv", "nu", "nobs", "ngoal", "n_substeps", "n_chains", "max_chain_dofs", "na")]


class Stats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("env_steps", "episodes", "reward_sum", "nan_resets", "solver_iters",
                                          "overflow_contacts", "unhandled_geoms", "linesearch_evals", "linesearch_capped", "steps_with_8plus_iters", "kernel_ms",
                                          "self_proximity", "self_contacts", "replayed_env_steps", "own_manifold_contacts")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class ForwardOut(C.Structure):
    _fields_ = [(n, _F) for n in ("M", "qfrc_bias", "qfrc_smooth", "qacc_smooth", "qacc", "qfrc_constraint")] + \
               [("ncon", C.POINTER(C.c_int)), ("solver_iter", C.POINTER(C.c_int))]


EXPORTS = ["lm_device_count", "lm_last_error", "lm_toolchain", "lm_model_create", "lm_model_destroy", "lm_model_dims",
           "lm_batch_create", "lm_batch_destroy", "lm_batch_set_layout", "lm_batch_set_replay", "lm_batch_set_active", "lm_get_replay_marks", "lm_set_state", "lm_get_state", "lm_set_activation", "lm_get_activation",
           "lm_set_dof_params", "lm_get_dof_params", "lm_set_dof_randomization", "lm_set_goal", "lm_step", "lm_step_device",
           "lm_pinned_slot", "lm_set_obs_order", "lm_step_pinned",
           "lm_set_reset_table", "lm_set_auto_reset", "lm_rollout", "lm_rollout_fused", "lm_forward_debug", "lm_get_stats", "lm_sync",
           "lm_get_flags", "lm_set_model_variants", "lm_set_variant_index", "lm_get_variant_index", "lm_set_variant_rows",
           "lm_set_model_compiler", "lm_compile_models", "lm_get_model_draws", "lm_get_model_tables"]

_lib = None


class BackendError(RuntimeError):
    pass


def load_library():
    """Load liblocohip.so and declare its prototypes. Raises BackendError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendError("HIP library not built: %s is missing (run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` or `make -C loco_mujoco_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.lm_device_count.restype = C.c_int
    lib.lm_last_error.restype = C.c_char_p
    lib.lm_toolchain.restype = C.c_char_p
    lib.lm_model_create.argtypes = [_D, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
    lib.lm_model_destroy.argtypes = [C.c_void_p]
    lib.lm_model_destroy.restype = None
    lib.lm_model_dims.argtypes = [C.c_void_p, C.POINTER(Dims)]
    lib.lm_batch_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.lm_batch_set_layout.argtypes = [C.c_void_p, C.c_int]
    lib.lm_batch_set_replay.argtypes = [C.c_void_p, C.c_int]
    lib.lm_batch_set_active.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
    lib.lm_get_replay_marks.argtypes = [C.c_void_p, _U8, C.c_int]
    lib.lm_batch_destroy.argtypes = [C.c_void_p]
    lib.lm_batch_destroy.restype = None
    lib.lm_set_state.argtypes = [C.c_void_p, _F, _F, _U8]
    lib.lm_get_state.argtypes = [C.c_void_p, _F, _F]
    lib.lm_set_goal.argtypes = [C.c_void_p, _F, _U8]
    lib.lm_set_dof_params.argtypes = [C.c_void_p, _F, _F, _F, _U8]
    lib.lm_get_dof_params.argtypes = [C.c_void_p, _F, _F, _F]
    lib.lm_set_dof_randomization.argtypes = [C.c_void_p, _F]
    lib.lm_step_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.lm_set_activation.argtypes = [C.c_void_p, _F, _U8]
    lib.lm_get_activation.argtypes = [C.c_void_p, _F]
    lib.lm_step.argtypes = [C.c_void_p, _F, _F, _F, _U8]
    lib.lm_pinned_slot.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_double)), C.POINTER(_U8)]
    lib.lm_set_obs_order.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
    lib.lm_step_pinned.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    lib.lm_set_reset_table.argtypes = [C.c_void_p, _F, C.c_int, C.c_uint64, C.c_int64]
    lib.lm_set_auto_reset.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.lm_rollout.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.POINTER(Stats)]
    lib.lm_rollout_fused.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(Stats)]
    lib.lm_forward_debug.argtypes = [C.c_void_p, _F, C.POINTER(ForwardOut)]
    lib.lm_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats), C.c_int]
    lib.lm_sync.argtypes = [C.c_void_p]
    lib.lm_get_flags.argtypes = [C.c_void_p, _U8]
    lib.lm_set_model_variants.argtypes = [C.c_void_p, _F, _F, _F, C.c_int, C.c_int]
    lib.lm_set_variant_index.argtypes = [C.c_void_p, C.POINTER(C.c_int32), _U8]
    lib.lm_get_variant_index.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.lm_set_variant_rows.argtypes = [C.c_void_p, C.c_int]
    lib.lm_set_model_compiler.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_longlong, C.POINTER(C.c_double), C.c_longlong, _F, _F, _F,
                                          C.c_int, C.c_uint64]
    lib.lm_compile_models.argtypes = [C.c_void_p, _U8]
    lib.lm_get_model_draws.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    lib.lm_get_model_tables.argtypes = [C.c_void_p, C.c_int, _F, _F, _F]
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise BackendError("liblocohip: %s (code %d)" % (load_library().lm_last_error().decode(), rc))


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _fp(a):
    return a.ctypes.data_as(_F)


def _mask(mask, n):
    if mask is None:
        return None, None
    m = np.ascontiguousarray(mask, dtype=np.uint8).reshape(n)
    return m, m.ctypes.data_as(_U8)


class HipModel:
    def __init__(self, chain_model, device=0):
        lib = load_library()
        if lib.lm_device_count() <= 0:
            raise BackendError("no HIP device visible: the batched simulator needs an MI355X (no CPU fallback)")
        cm = np.ascontiguousarray(chain_model, dtype=np.float64)
        h = C.c_void_p()
        _check(lib.lm_model_create(cm.ctypes.data_as(_D), len(cm), device, C.byref(h)))
        self._h = h
        self._lib = lib                      # kept on the object: module globals are gone when __del__ runs at interpreter exit
        d = Dims()
        _check(lib.lm_model_dims(h, C.byref(d)))
        self.dims = d

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lm_model_destroy(self._h)
            self._h = None

    __del__ = close


class HipBatch:
    """A batch of ``n_envs`` lock-step environments resident on one GPU."""

    def __init__(self, model, n_envs, envs_per_workgroup=None):
        """``envs_per_workgroup``: None / 4 = the replicated layout (default), 8 or 16 = the plain layout (lm_batch_set_layout)."""
        self.model = model
        self.n = int(n_envs)
        self._lib = load_library()
        h = C.c_void_p()
        _check(self._lib.lm_batch_create(model._h, self.n, C.byref(h)))
        self._h = h
        if envs_per_workgroup is not None:
            _check(self._lib.lm_batch_set_layout(self._h, int(envs_per_workgroup)))
        d = model.dims
        self.nq, self.nv, self.nu, self.nobs, self.ngoal = d.nq, d.nv, d.nu, d.nobs, d.ngoal
        self.na = d.na

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lm_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def set_replay(self, enabled):
        """Speculate / replay (``lm_batch_set_replay``): on by default; off = the regular kernels alone, contacts beyond their slots
        are dropped and counted (A/B measurements)."""
        # 2 (tests): everything through the replay kernel; 3 / 4: like 1 / 2 without the concurrent pollers (profilers)
        _check(self._lib.lm_batch_set_replay(self._h, int(enabled) if enabled in (2, 3, 4) else int(bool(enabled))))

    def replay_marks(self, reset=False):
        """Per environment: did the replay kernel run one of its control steps since the marks were last cleared?"""
        out = np.empty(self.n, dtype=np.uint8)
        _check(self._lib.lm_get_replay_marks(self._h, out.ctypes.data_as(_U8), int(bool(reset))))
        return out != 0

    def set_state(self, qpos, qvel, mask=None):
        q, v = _f32(qpos, (self.n, self.nq)), _f32(qvel, (self.n, self.nv))
        keep, mp = _mask(mask, self.n)
        _check(self._lib.lm_set_state(self._h, _fp(q), _fp(v), mp))

    def get_state(self):
        q = np.empty((self.n, self.nq), dtype=np.float32)
        v = np.empty((self.n, self.nv), dtype=np.float32)
        _check(self._lib.lm_get_state(self._h, _fp(q), _fp(v)))
        return q, v

    def step_device(self, action=None, obs=None, reward=None, done=None, stream=None, sync=True):
        """One control step on DEVICE buffers: arguments are device pointers (ints) or objects with ``data_ptr()`` such as
        torch tensors — float32 action [n, nu], obs [n, nobs], reward [n], uint8 done [n]; None = zero action / library
        buffers. ``stream``: raw hipStream_t (e.g. ``torch.cuda.current_stream().cuda_stream``).

        The done byte is a bit field, NOT a boolean: bit 0 (``done & 1``) = absorbing state, what the reference's ``step()``
        returns; bit 1 (``done & 2``) = the episode ended on the device in this step (restarted from the reset table — ``obs``
        is then the first observation of the new episode — or, without auto-reset, the step that reached the horizon).
        ``done.bool()`` mixes truncation and restarts into the terminal flag; mask the bits."""
        def ptr(x):
            if x is None:
                return None
            return C.c_void_p(int(x.data_ptr()) if hasattr(x, "data_ptr") else int(x))
        _check(self._lib.lm_step_device(self._h, ptr(action), ptr(obs), ptr(reward), ptr(done),
                                        None if stream is None else C.c_void_p(int(stream)), int(bool(sync))))

    def set_dof_params(self, damping=None, stiffness=None, frictionloss=None, mask=None):
        """Per-environment joint parameters [n, nv] (domain randomisation); None leaves a parameter as it is."""
        arrs = [None if x is None else _f32(x, (self.n, self.nv)) for x in (damping, stiffness, frictionloss)]
        keep, mp = _mask(mask, self.n)
        _check(self._lib.lm_set_dof_params(self._h, *[None if x is None else _fp(x) for x in arrs], mp))

    def get_dof_params(self):
        out = [np.empty((self.n, self.nv), dtype=np.float32) for _ in range(3)]
        _check(self._lib.lm_get_dof_params(self._h, *[_fp(x) for x in out]))
        return dict(damping=out[0], stiffness=out[1], frictionloss=out[2])

    def set_dof_randomization(self, spec):
        """Device-side redraw rule at episode restarts: spec[3, nv, 3] = (kind, a, b); None disables."""
        if spec is None:
            _check(self._lib.lm_set_dof_randomization(self._h, None))
            return
        s = _f32(spec, (3, self.nv, 3))
        _check(self._lib.lm_set_dof_randomization(self._h, _fp(s)))

    def set_model_variants(self, tables):
        """``tables``: list of (record, geom table, geom-pair table) from ``lowering.variant_tables`` — the pool of randomised
        models of this batch (None / empty removes it). Every environment starts on variant 0; a device-side restart redraws."""
        if not tables:
            _check(self._lib.lm_set_model_variants(self._h, None, None, None, 0, 0))
            self.n_variants = 0
            return
        rec = np.ascontiguousarray(np.stack([t[0] for t in tables]), dtype=np.float32)
        gt = np.ascontiguousarray(np.stack([t[1] for t in tables]), dtype=np.float32)
        npair = len(tables[0][2])
        gpt = np.ascontiguousarray(np.stack([t[2] for t in tables]), dtype=np.float32) if npair else None
        _check(self._lib.lm_set_model_variants(self._h, _fp(rec), _fp(gt), _fp(gpt) if gpt is not None else None, npair, len(tables)))
        self.n_variants = len(tables)

    def set_model_compiler(self, program, nominal_tables, seed=0):
        """The model compiler on the device: ``program`` = (int32, float64) of ``lowering.model_compiler_tables``, ``nominal_tables`` =
        ``lowering.variant_tables(nominal, nominal)``. Every environment gets a slot and a freshly drawn model of its own, now and at
        every device-side restart; :meth:`compile_models` is the host-side reset."""
        ib = np.ascontiguousarray(program[0], dtype=np.int32)
        db = np.ascontiguousarray(program[1], dtype=np.float64)
        # the library checks the program's sizes and where its gather / contact ops write; what its draws and drawn bodies index is
        # checked here (the kernel's tables in LDS are sized by the header's counts)
        if len(ib) < 16:
            raise BackendError("model-compiler program: header missing")
        nv, nrb, ngs, nd, nslot, nbody = int(ib[1]), int(ib[2]), int(ib[3]), int(ib[4]), int(ib[5]), int(ib[8])
        if len(ib) < 16 + 4 * nd + 4 * nrb:
            raise BackendError("model-compiler program: draw / body tables cut short")
        draws = ib[16:16 + 4 * nd].reshape(nd, 4)
        rb = ib[16 + 4 * nd:16 + 4 * nd + 4 * nrb].reshape(nrb, 4)
        limit = {0: (nv, 1), 1: (nrb, 1), 2: (nrb, 3), 3: (nrb, 3), 4: (ngs, 3)}
        for kind, target, idx, comp in draws:
            if kind not in (1, 2, 3) or target not in limit or not (0 <= idx < limit[int(target)][0] and 0 <= comp < limit[int(target)][1]):
                raise BackendError("model-compiler program: a draw outside its table (kind %d target %d index %d component %d)" % (kind, target, idx, comp))
        for body, kind, slot, has_sv in rb:
            if not (0 < body < nbody and kind in (1, 2) and 0 <= slot < nslot and has_sv in (0, 1)):
                raise BackendError("model-compiler program: a drawn body outside the model (body %d kind %d slot %d)" % (body, kind, slot))
        rec, gt = _f32(nominal_tables[0], (len(nominal_tables[0]),)), _f32(nominal_tables[1], (len(nominal_tables[1]),))
        npair = len(nominal_tables[2])
        gpt = _f32(nominal_tables[2], (npair,)) if npair else None
        _check(self._lib.lm_set_model_compiler(self._h, ib.ctypes.data_as(C.POINTER(C.c_int32)), len(ib), db.ctypes.data_as(C.POINTER(C.c_double)),
                                               len(db), _fp(rec), _fp(gt), _fp(gpt) if gpt is not None else None, npair, int(seed) & (2 ** 64 - 1)))
        self.n_variants = self.n
        self.n_model_draws = int(ib[4])
        self._table_sizes = (len(rec), len(gt), npair)

    def compile_models(self, mask=None):
        """A fresh model for the masked environments (None: all), drawn and compiled on the device."""
        keep, mp = _mask(mask, self.n)
        _check(self._lib.lm_compile_models(self._h, mp))

    def get_model_draws(self):
        """(draws [n, n_draw] float64 of every environment's CURRENT model, models each environment has had [n])."""
        d = np.empty((self.n, self.n_model_draws), dtype=np.float64)
        g = np.empty(self.n, dtype=np.uint32)
        _check(self._lib.lm_get_model_draws(self._h, d.ctypes.data_as(C.POINTER(C.c_double)), g.ctypes.data_as(C.POINTER(C.c_uint32))))
        return d, g

    def get_model_tables(self, env, sizes=None):
        """(record, geom table, geom-pair table) environment ``env`` runs on."""
        nr, ng, npair = sizes if sizes is not None else self._table_sizes
        rec, gt, gpt = np.empty(nr, dtype=np.float32), np.empty(ng, dtype=np.float32), np.empty(npair, dtype=np.float32)
        _check(self._lib.lm_get_model_tables(self._h, int(env), _fp(rec), _fp(gt), _fp(gpt) if npair else None))
        return rec, gt, gpt

    def set_variant_index(self, index, mask=None):
        idx = np.ascontiguousarray(np.broadcast_to(np.asarray(index, dtype=np.int32), (self.n,)))
        keep, mp = _mask(mask, self.n)
        _check(self._lib.lm_set_variant_index(self._h, idx.ctypes.data_as(C.POINTER(C.c_int32)), mp))

    def set_variant_rows(self, rows_per_variant):
        """The reset table is n_variants blocks of ``rows_per_variant`` rows: a device-side restart from row i puts the environment
        on variant i // rows_per_variant (0: variants are redrawn independently of the row)."""
        _check(self._lib.lm_set_variant_rows(self._h, int(rows_per_variant)))

    def get_variant_index(self):
        idx = np.zeros(self.n, dtype=np.int32)
        _check(self._lib.lm_get_variant_index(self._h, idx.ctypes.data_as(C.POINTER(C.c_int32))))
        return idx

    def set_activation(self, act, mask=None):
        """Muscle activations [n, na] (set_state zeroes them like mj_resetData; this is for checkpoints and tests)."""
        a = _f32(act, (self.n, self.na))
        keep, mp = _mask(mask, self.n)
        _check(self._lib.lm_set_activation(self._h, _fp(a), mp))

    def get_activation(self):
        a = np.empty((self.n, self.na), dtype=np.float32)
        _check(self._lib.lm_get_activation(self._h, _fp(a)))
        return a

    def set_goal(self, goal, mask=None):
        if self.ngoal == 0:
            return
        g = _f32(goal, (self.n, self.ngoal))
        keep, mp = _mask(mask, self.n)
        _check(self._lib.lm_set_goal(self._h, _fp(g), mp))

    def step(self, action):
        a = _f32(action, (self.n, self.nu))
        obs = np.empty((self.n, self.nobs), dtype=np.float32)
        rew = np.empty(self.n, dtype=np.float32)
        done = np.empty(self.n, dtype=np.uint8)
        _check(self._lib.lm_step(self._h, _fp(a), _fp(obs), _fp(rew), done.ctypes.data_as(_U8)))
        # done byte: bit 0 = absorbing state, bit 1 = the device ended the episode in this step (restarted it from the
        # reset table, or the horizon was reached)
        self.last_restarted = (done & 2) != 0
        return obs, rew, (done & 1) != 0

    def set_active(self, env_ids):
        """Run only the listed environments from now on (``lm_batch_set_active``); None: all of them again."""
        if env_ids is None:
            _check(self._lib.lm_batch_set_active(self._h, None, 0))
            return
        ids = np.ascontiguousarray(env_ids, dtype=np.int32).reshape(-1)
        _check(self._lib.lm_batch_set_active(self._h, ids.ctypes.data_as(C.POINTER(C.c_int32)), len(ids)))

    PINNED_SLOTS = 4

    def set_obs_order(self, perm):
        """Column order of the float64 observation of step_pinned (None: the kernel's own)."""
        if perm is None:
            _check(self._lib.lm_set_obs_order(self._h, None, 0))
        else:
            p = np.ascontiguousarray(perm, dtype=np.int32)
            _check(self._lib.lm_set_obs_order(self._h, p.ctypes.data_as(C.POINTER(C.c_int32)), len(p)))

    def _pinned_views(self):
        views = []
        for slot in range(self.PINNED_SLOTS):
            o, r, d = C.POINTER(C.c_double)(), C.POINTER(C.c_double)(), _U8()
            _check(self._lib.lm_pinned_slot(self._h, slot, C.byref(o), C.byref(r), C.byref(d)))
            arrs = []
            for ptr, ct, dt, shape in ((o, C.c_double, np.float64, (self.n, self.nobs)), (r, C.c_double, np.float64, (self.n,)), (d, C.c_uint8, np.uint8, (self.n,))):
                buf = (ct * int(np.prod(shape))).from_address(C.addressof(ptr.contents))
                buf._owner = self          # a view handed to the caller keeps the batch (and with it the pinned memory) alive
                arrs.append(np.frombuffer(buf, dtype=dt).reshape(shape))
            views.append(tuple(arrs))
        return views

    def step_pinned(self, action64):
        """One control step through the library's float64 host surface (lm_step_pinned): ``action64`` is a C-contiguous float64
        array [n, nu]; returns (obs float64 [n, nobs], reward float64 [n], absorbing bool [n]). obs and reward are VIEWS of a ring of
        PINNED_SLOTS pinned result sets owned by the batch: what a call returned stays intact for the next PINNED_SLOTS - 1 calls."""
        if getattr(self, "_pinned", None) is None:
            self._pinned = self._pinned_views()
            self._slot = -1
        self._slot = (self._slot + 1) % self.PINNED_SLOTS
        _check(self._lib.lm_step_pinned(self._h, action64.ctypes.data_as(C.POINTER(C.c_double)), self._slot))
        obs, rew, done = self._pinned[self._slot]
        self.last_restarted = (done & 2) != 0
        return obs, rew, (done & 1) != 0

    def set_reset_table(self, rows, seed=0, global_env_offset=0):
        r = _f32(rows)
        assert r.ndim == 2 and r.shape[1] == self.nq + self.nv + self.ngoal
        _check(self._lib.lm_set_reset_table(self._h, _fp(r), r.shape[0], int(seed), int(global_env_offset)))

    def set_auto_reset(self, enabled, horizon=0):
        _check(self._lib.lm_set_auto_reset(self._h, int(bool(enabled)), int(horizon)))

    def rollout(self, n_steps, action_mode=0, seed=0, steps_per_launch=1):
        """n_steps control steps on the device with zero (0) or uniform random (1) actions. steps_per_launch > 1 fuses that
        many control steps into one launch (same results, no device-wide join between control steps)."""
        st = Stats()
        _check(self._lib.lm_rollout_fused(self._h, int(n_steps), int(steps_per_launch), int(action_mode), int(seed), C.byref(st)))
        return st.as_dict()

    def forward_debug(self, action):
        a = _f32(action, (self.n, self.nu))
        nv = self.nv
        res = dict(M=np.zeros((self.n, nv, nv), np.float32), qfrc_bias=np.zeros((self.n, nv), np.float32),
                   qfrc_smooth=np.zeros((self.n, nv), np.float32), qacc_smooth=np.zeros((self.n, nv), np.float32),
                   qacc=np.zeros((self.n, nv), np.float32), qfrc_constraint=np.zeros((self.n, nv), np.float32))
        ncon = np.zeros(self.n, np.int32)
        it = np.zeros(self.n, np.int32)
        out = ForwardOut()
        for k, arr in res.items():
            setattr(out, k, _fp(arr))
        out.ncon = ncon.ctypes.data_as(C.POINTER(C.c_int))
        out.solver_iter = it.ctypes.data_as(C.POINTER(C.c_int))
        _check(self._lib.lm_forward_debug(self._h, _fp(a), C.byref(out)))
        res["ncon"], res["solver_iter"] = ncon, it
        return res

    def stats(self, reset=False):
        st = Stats()
        _check(self._lib.lm_get_stats(self._h, C.byref(st), int(reset)))
        return st.as_dict()

    def flags(self):
        """Validity flags of the last control step per environment (``lm_get_flags``): 1 dropped contact, 2 self pair without a
        collider in reach, 4 collider-less geom at the floor."""
        out = np.empty(self.n, dtype=np.uint8)
        _check(self._lib.lm_get_flags(self._h, out.ctypes.data_as(_U8)))
        return out

    def sync(self):
        _check(self._lib.lm_sync(self._h))
