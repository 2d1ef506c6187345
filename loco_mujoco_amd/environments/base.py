"""
``LocoEnv`` — host-side mirror of the reference's ``loco_mujoco/environments/base.py`` for the
``step()``/``reset()`` hot path, batched over ``n_envs`` lock-step environments that live on one
MI355X behind the C-ABI (``include/locohip.h``).

What stays identical to the reference (for ``n_envs=1``): ``LocoEnv.make(task_id, **kw)``,
``reset(obs=None) -> obs``, ``step(action) -> (obs, reward, absorbing, info)``, ``info.observation_space``
/ ``info.action_space`` (actions normalised to [-1, 1], ``base.py:122-126``), ``create_dataset()``,
``load_trajectory()``, the ``np.random`` draw order of ``reset`` (``base.py:187-191``,
``trajectory.py:253,259``) and float64 NumPy outputs of shape ``(nobs,)``.
With ``n_envs=N`` every array gains a leading batch dimension.

What is different by design: the physics, observation gather, reward and termination of ``step()``
run inside one HIP kernel launch per control step (``csrc/``); this class only converts dtypes and
keeps the episode bookkeeping. There is no CPU physics path in the product.
"""

import copy
import os
import warnings
from itertools import product
from pathlib import Path

import numpy as np

from ..utils.reward import (CustomReward, NoReward, PosReward, TargetVelocityReward)
from ..utils.trajectory import Trajectory
from .observation import Box, MDPInfo, ObservationHelper, ObservationType

_PKG = Path(__file__).resolve().parent.parent


class _HostState:
    """Host copy of one environment's simulation state, used to assemble the reset observation."""

    def __init__(self, nq, nv):
        self.qpos = np.zeros(nq)
        self.qvel = np.zeros(nv)
        self.site_xmat = {}


class LocoEnv:
    """Base class of all locomotion environments (reference ``base.py:25``)."""

    _registered_envs = dict()

    def __init__(self, model, action_spec, observation_spec, collision_groups=None, gamma=0.99, horizon=1000,
                 n_substeps=10, reward_type=None, reward_params=None, traj_params=None, random_start=True,
                 init_step_no=None, timestep=0.001, use_foot_forces=False, default_camera_mode="follow",
                 use_absorbing_states=True, domain_randomization_config=None, parallel_dom_rand=True,
                 N_worker_per_xml_dom_rand=4, n_envs=1, device=0, n_model_variants=None, model_variants_per_reset=4,
                 copy_outputs=False, **viewer_params):
        self._model = model
        # n_envs > 1: step() returns VIEWS of a ring of four pinned result sets (intact for the next three calls) unless copy_outputs
        self._copy_outputs = bool(copy_outputs)
        # models compiled per batch for the randomisation rules that change compile-time constants (inertial, armature, geom
        # friction): a pool the environments draw from per episode, see utils/domain_randomization.py
        # The reference compiles one freshly drawn model at every reset (base.py:183-185). Here: a pool of `n_model_variants`
        # models, of which every reset() REPLACES `model_variants_per_reset` (round-robin; 0 = fixed pool), so that a long training
        # is not limited to the first pool; a batch of at most that many environments runs one brand-new model per environment
        # and episode — the reference's behaviour exactly. Device-side restarts (auto-reset inside a rollout) draw from the pool.
        # Round 5: `n_model_variants=None` (the default) takes the MODEL COMPILER ON THE DEVICE instead (csrc/lm_compile.hip): every
        # environment draws and compiles a model of its own at reset() and at every device-side restart — the reference's behaviour,
        # at any batch size. An explicit `n_model_variants` keeps the host-compiled pool (and several models in one batch always do).
        self._compiler_wanted = n_model_variants is None
        self._pending_compile = False
        n_model_variants = 32 if n_model_variants is None else n_model_variants
        self._n_model_variants = int(n_model_variants)
        self._variants_per_reset = int(model_variants_per_reset)
        self._variant_models = {}          # model index -> [CompiledModel] (kept for inspection and the parity tests)
        # the pools, lowered — ONE PER MODEL of the environment (the reference's MultiMuJoCo randomises whichever model the episode
        # drew): model index -> {"tables", "cursor", "dirty"}; built at the first reset() of that model, rebuilt after seed().
        # `_variant_tables` / `_variant_cursor` / `_variant_dirty` are the current model's.
        self._variant_pools = {}
        # joint-parameter randomisation per episode (reference base.py:103-107,183-185). The reference draws in worker
        # processes, i.e. outside the main np.random stream — so does this (own RandomState, reseeded by seed()).
        self._domain_rand = None
        if domain_randomization_config is not None:
            from ..utils.domain_randomization import JointRandomization
            self._domain_rand = JointRandomization(model, domain_randomization_config)
            self._domain_rand_rs = np.random.RandomState(0)
            if self._domain_rand.has_model_rules and self._n_model_variants < 1:
                raise ValueError("n_model_variants must be >= 1 with randomisation rules that change compile-time constants "
                                 "(inertial, armature, geom friction), got %d" % self._n_model_variants)
            if not 0 <= self._variants_per_reset:
                raise ValueError("model_variants_per_reset must be >= 0")
        assert abs(model.timestep - timestep) < 1e-12, "compile the model with the environment's timestep"
        self._timestep = timestep
        # foot forces: the reference turns the control step into n_substeps intermediate steps of one substep each and
        # averages the per-foot contact force over them (base.py:94-98,623-631); the device does the same inside one launch
        self._n_substeps = n_substeps
        self._n_intermediate_steps = 1
        self._collision_groups = dict(collision_groups or [])
        self.n_envs = int(n_envs)
        self._device = device

        # ---- observation / action bookkeeping (what mushroom-rl's MuJoCo.__init__ does)
        self.obs_helper = ObservationHelper(observation_spec, model)
        self._action_spec = list(action_spec) if action_spec else list(model.act_names)
        self._action_indices = np.array([model.act_id(n) for n in self._action_spec], dtype=np.int64)
        low = model.act_ctrlrange[self._action_indices, 0].copy()
        high = model.act_ctrlrange[self._action_indices, 1].copy()
        action_space = Box(low, high)
        observation_space = Box(*self.obs_helper.get_obs_limits())
        self._mdp_info = MDPInfo(observation_space, action_space, gamma, horizon, dt=self.dt)

        self._reward_function = self._get_reward_function(reward_type, reward_params)
        self._use_foot_forces = bool(use_foot_forces)
        self.info.observation_space = Box(*self._get_observation_space())

        # actions are normalised to [-1, 1] (reference base.py:122-126)
        self.norm_act_mean = (high + low) / 2.0
        self.norm_act_delta = (high - low) / 2.0
        self.info.action_space.low[:] = -1.0
        self.info.action_space.high[:] = 1.0

        self._dataset = None
        self.trajectories = None
        if traj_params:
            self.load_trajectory(traj_params)
        self._random_start = random_start
        self._init_step_no = init_step_no
        self._use_absorbing_states = use_absorbing_states

        # ---- simulation state
        self._host = [_HostState(model.nv, model.nv) for _ in range(self.n_envs)]
        self._obs = None
        self._backend = None           # created on first use (needs a GPU)
        self._pending_state = False
        self._auto_reset = False
        self._n_models = 1
        self._current_model_idx = 0
        self._blocks = False             # several models in one BATCH: contiguous blocks of environments, one per model
        self._grouped = False            # ... blocks whose environments draw their model per episode (one batch per model over ALL ids)
        self._redrawn = None
        self._pooled = False             # ... or ONE batch whose environments draw their model per episode (model variants)
        self._random_env_reset = True

    # ------------------------------------------------------------------ registry / factory
    @classmethod
    def register(cls):
        LocoEnv._registered_envs.setdefault(cls.__name__, cls)

    @staticmethod
    def list_registered_loco_mujoco():
        return list(LocoEnv._registered_envs.keys())

    @staticmethod
    def make(env_name, *args, **kwargs):
        """``LocoEnv.make("UnitreeA1.simple.real", **kw)`` -> ``UnitreeA1.generate("simple", "real", **kw)``."""
        parts = env_name.split(".")
        name, gen_args = parts[0], parts[1:]
        if name not in LocoEnv._registered_envs:
            raise ValueError("Environment '%s' is not registered. Available: %s"
                             % (name, LocoEnv.list_registered_loco_mujoco()))
        return LocoEnv._registered_envs[name].generate(*gen_args, *args, **kwargs)

    @classmethod
    def get_all_task_names(cls):
        names = []
        for e, env in cls._registered_envs.items():
            for conf in env.valid_task_confs.get_all_combinations():
                names.append(".".join([e] + list(conf.values())))
        return names

    # ------------------------------------------------------------------ properties
    @property
    def info(self):
        return self._mdp_info

    @property
    def dt(self):
        return self._timestep * self._n_intermediate_steps * self._n_substeps

    @property
    def backend(self):
        """The device batch (``HipBatch``). Created lazily; raises if the HIP library/GPU is missing."""
        if self._backend is None:
            from ..backend import HipBatch, HipModel
            nominal = self._chain_model(self._models[0] if self._pooled else None)
            self._hip_model = HipModel(nominal, self._device)
            self._backend = HipBatch(self._hip_model, len(self._model_envs(self._current_model_idx)) if (self._blocks and not self._grouped) else self.n_envs)
            if self._pooled:
                from ..lowering import variant_tables
                self._backend.set_model_variants([variant_tables(nominal, self._chain_model(m)) for m in self._models])
            elif self._use_model_compiler:
                self._install_model_compiler(nominal)
            elif self._domain_rand is not None and self._domain_rand.has_model_rules:
                self._ensure_variant_pool()
                self._backend.set_model_variants(self._variant_tables)
                self._variant_dirty = False
        return self._backend

    def _install_model_compiler(self, nominal=None):
        """(Re-)install the model compiler on the device: its program and a seed from the randomisation's own generator — after
        ``seed()`` the models of the following episodes are a function of that seed again."""
        from ..lowering import model_compiler_tables, variant_tables
        nominal = self._chain_model() if nominal is None else nominal
        if getattr(self, "_compiler_program", None) is None:
            ops, svd = self._domain_rand.model_draw_ops()
            ib, db, self._compiler_info = model_compiler_tables(self._model, self._device_task(), ops, svd)
            self._compiler_program = (ib, db, variant_tables(nominal, nominal))
        ib, db, tabs = self._compiler_program
        self._backend.set_model_compiler((ib, db), tabs, seed=int(self._domain_rand_rs.randint(0, 2 ** 31 - 1)))
        self._compiler_reseed = False

    @property
    def _use_model_compiler(self):
        """True if the randomisation rules that change compile-time constants run through the model compiler on the device."""
        return bool(self._compiler_wanted and self._domain_rand is not None and self._domain_rand.has_model_rules
                    and not self._blocks and not self._pooled and self._n_models == 1)

    def model_of_env(self, e):
        """The compiled model environment ``e`` currently runs on when the device compiles the models: rebuilt on the host from the
        draws the device reports (``lm_get_model_draws``) — for inspection, the parity tests and the oracle."""
        if not self._use_model_compiler:
            raise ValueError("the environments of this batch do not compile their models on the device")
        draws, _ = self.backend.get_model_draws()
        return self._domain_rand.variant_from_draws(draws[int(e)])

    # the current model's pool (see __init__)
    def _pool(self):
        return self._variant_pools.setdefault(getattr(self, "_current_model_idx", 0), dict(tables=None, cursor=0, dirty=False))

    @property
    def _variant_tables(self):
        return self._pool()["tables"]

    @_variant_tables.setter
    def _variant_tables(self, v):
        if v is None:
            self._variant_pools.clear()           # seed(): every model's pool is a function of the seed
        else:
            self._pool()["tables"] = v

    @property
    def _variant_cursor(self):
        return self._pool()["cursor"]

    @_variant_cursor.setter
    def _variant_cursor(self, v):
        self._pool()["cursor"] = v

    @property
    def _variant_dirty(self):
        return self._pool()["dirty"]

    @_variant_dirty.setter
    def _variant_dirty(self, v):
        self._pool()["dirty"] = v

    def _build_model_variants(self, nominal, count=None):
        """`count` (default: the whole pool) randomised models of the current model's batch: drawn with the randomisation's own
        generator, compiled (``mjcf.model_variant``), lowered, reduced to what differs from the nominal tables
        (``lowering.variant_tables``). Returns (models, tables)."""
        from ..lowering import variant_tables
        state = np.random.get_state()
        np.random.set_state(self._domain_rand_rs.get_state())
        models = [self._domain_rand.sample_model_variant(self._model) for _ in range(self._n_model_variants if count is None else count)]
        self._domain_rand_rs.set_state(np.random.get_state())
        np.random.set_state(state)
        return models, [variant_tables(nominal, self._chain_model(v)) for v in models]

    def _ensure_variant_pool(self):
        """Host-only and deterministic: the pool is a function of the randomisation generator's state (``seed()``) alone — built
        at the first reset() before anything else is drawn from that generator, never as a side effect of the first step()."""
        if self._variant_tables is None:
            if self._pooled:
                # (several models with one device batch per model — contiguous blocks, or n_envs = 1 — keep one pool per model)
                raise NotImplementedError("several models in ONE device batch AND randomised compile-time constants: both use the model variants")
            models, self._variant_tables = self._build_model_variants(self._chain_model())
            self._variant_models[self._current_model_idx] = models
            self._variant_cursor = 0
            self._variant_dirty = True

    def refresh_model_variants(self, count=None):
        """Replace `count` models of the pool (default: all of them) by fresh draws, round-robin. Returns the pool indices that
        were replaced. Takes effect on the device at the next upload of a reset (environments in mid-episode on a replaced index
        would change model: call it where every environment restarts, as reset() does). With the model compiler on the device: a
        fresh model for EVERY environment, at once."""
        if self._use_model_compiler:
            self.backend.compile_models()
            return np.arange(self.n_envs)
        self._ensure_variant_pool()
        k = self._n_model_variants if count is None else min(int(count), self._n_model_variants)
        if k <= 0:
            return np.zeros(0, dtype=np.int64)
        models, tables = self._build_model_variants(self._chain_model(), k)
        idx = (self._variant_cursor + np.arange(k)) % self._n_model_variants
        for j, mdl, tab in zip(idx, models, tables):
            self._variant_models[self._current_model_idx][j] = mdl
            self._variant_tables[j] = tab
        self._variant_cursor = int((self._variant_cursor + k) % self._n_model_variants)
        self._variant_dirty = True
        return idx

    def _init_models(self, models):
        """Several models in one environment (the reference's ``MultiMuJoCo``: sizes of the humanoid, carried weights).
        One compiled model and, lazily, one device batch per model; one of them is current (``base.py:186-193``)."""
        self._models = list(models)
        self._n_models = len(self._models)
        self._current_model_idx = 0
        self._model_backends = [None] * self._n_models
        # A device batch has ONE constant table. Models that differ only in what a model VARIANT carries (inertial numbers,
        # invweights, geom tables: the carried weights) share one batch, and every environment draws its model per episode
        # like the reference (base.py:186-190) — `_pooled`. Models that differ in geometry (the humanoid's four sizes) get
        # contiguous blocks of environments, one block (= one device batch) per model: environment e keeps model
        # e * n_models // n_envs for its whole life — the same mixture over the batch — `_blocks`.
        self._pooled = self.n_envs > 1 and self._n_models > 1 and self._models_differ_like_variants()
        self._blocks = self.n_envs > 1 and self._n_models > 1 and not self._pooled
        # GROUPED (round 6): models that differ in geometry AND are drawn per episode like the reference's (the humanoid's four sizes,
        # base_humanoid_4_ages.py:106-135, base.py:186-190). Every model's device batch spans ALL environment ids; environment e is
        # ACTIVE in the batch of the model it drew for this episode (lm_batch_set_active: one launch per model over its active list)
        # and changes batch when an episode ends — at reset() on the host, and at a device-side restart through a host redraw keyed by
        # (seed, global environment id, episodes so far), so that sharding a batch does not change what any environment draws.
        # (rounds 2-5: `_blocks` alone — environment e kept model e * n_models // n_envs for its whole life.)
        self._grouped = (self._blocks and self._models_regroup_per_episode()
                         and not (self._domain_rand is not None and self._domain_rand.active))      # (with domain randomisation: blocks, as before)
        if self._blocks and not self._grouped and self.n_envs < self._n_models:
            raise ValueError("n_envs=%d cannot hold %d models" % (self.n_envs, self._n_models))
        self._env_model = np.zeros(self.n_envs, dtype=np.int64)
        if self._blocks:
            n, m = self.n_envs, self._n_models
            for i in range(m):
                self._env_model[(i * n + m - 1) // m:((i + 1) * n + m - 1) // m] = i
        self._active_version = [None] * self._n_models        # grouped: the active list each model's batch was last given
        self._episodes = np.zeros(self.n_envs, dtype=np.int64)  # grouped: device-side restarts redrawn by the host, per environment

    def _models_differ_like_variants(self):
        """True if the models of this environment can live in one batch as model variants (``lowering.variant_tables``)."""
        return False

    def _models_regroup_per_episode(self):
        """True if a batch draws one of its (geometrically different) models per environment and episode (`_grouped`)."""
        return False

    def _model_envs(self, idx):
        """Environment indices of model ``idx``: its block, or (grouped) the environments that drew it for their current episode."""
        if self._grouped:
            return np.nonzero(self._env_model == idx)[0]
        n, m = self.n_envs, self._n_models
        return np.arange((idx * n + m - 1) // m, ((idx + 1) * n + m - 1) // m)

    def _activate(self, idx, envs):
        """grouped: hand model ``idx``'s batch its active list (only when it changed)."""
        key = envs.tobytes()
        if self._active_version[idx] != key:
            self.backend.set_active(envs)
            self._active_version[idx] = key

    def _block_of(self, e):
        """Model of environment ``e`` in block mode."""
        return int(self._env_model[e])

    def _select_model(self, idx):
        """Make model ``idx`` (drawn per episode, ``base.py:186-190``) current."""
        if self._n_models <= 1:
            return
        if self._pooled:                   # one shared batch: only the host-side model changes
            self._current_model_idx = idx
            self._model = self._models[idx]
            return
        self._model_backends[self._current_model_idx] = self._backend
        self._current_model_idx = idx
        self._model = self._models[idx]
        self._backend = self._model_backends[idx]
        self._hip_model = None

    def _obs_perm(self):
        """Index array that turns the device observation [q, v, constants, foot forces] into the reference's order, or
        None when they coincide (they differ when an environment appends constants AFTER the foot forces)."""
        return None

    # ------------------------------------------------------------------ trajectories / datasets
    def load_trajectory(self, traj_params, warn=True):
        if self.trajectories is not None:
            warnings.warn("New trajectories loaded, which overrides the old ones.", RuntimeWarning)
        self.trajectories = Trajectory(keys=self.get_all_observation_keys(),
                                       low=self.info.observation_space.low,
                                       high=self.info.observation_space.high,
                                       joint_pos_idx=self.obs_helper.joint_pos_idx,
                                       interpolate_map=self._interpolate_map,
                                       interpolate_remap=self._interpolate_remap,
                                       interpolate_map_params=self._get_interpolate_map_params(),
                                       interpolate_remap_params=self._get_interpolate_remap_params(),
                                       warn=warn, **traj_params)

    def create_dataset(self, ignore_keys=None):
        if self._dataset is None:
            if self.trajectories is None:
                raise ValueError("No trajectory was passed to the environment. "
                                 "To create a dataset pass a trajectory first.")
            dataset = self.trajectories.create_dataset(ignore_keys=ignore_keys)
            for state in dataset["states"]:
                has_fallen, msg = self._has_fallen(state, return_err_msg=True)
                if has_fallen:
                    raise ValueError("Some of the states in the created dataset are terminal states. "
                                     "This should not happen.\n\nViolations:\n" + msg)
            self._dataset = copy.deepcopy(dataset)
            return dataset
        return copy.deepcopy(self._dataset)

    def load_dataset_and_get_traj_files(self, dataset_path, freq=None):
        """A recorded ("perfect") dataset -> per-key trajectories (reference ``base.py:499-548``). The dataset holds
        ``states`` [n, nobs] (the observation layout: no horizontal root position), ``actions``, ``last`` ...; the two
        missing coordinates are integrated from their velocities at ``freq`` Hz (restarting at 0 after every ``last``) or
        left at zero without ``freq``; ``split_points`` mark the episode starts. The dataset itself becomes what
        ``create_dataset`` returns."""
        root = Path(os.environ.get("LOCO_MUJOCO_AMD_DATA", _PKG))
        path = root / dataset_path
        if not path.exists():
            raise FileNotFoundError("dataset %s not found: the recorded datasets are downloads of the reference project; put "
                                    "them under $LOCO_MUJOCO_AMD_DATA/datasets/... (same relative paths)" % path)
        with np.load(str(path), allow_pickle=True) as f:
            dataset = {k: np.asarray(f[k]) for k in f.files}
        self._dataset = copy.deepcopy(dataset)
        states, last = np.atleast_2d(dataset["states"]), dataset["last"]
        keys = [spec[0] for spec in self.obs_helper.observation_spec]
        trajectories = dict()
        for i, key in enumerate(keys):
            if i >= 2:
                trajectories[key] = states[:, i - 2]
            elif freq is None:
                trajectories[key] = np.zeros(len(states))
            else:
                assert len(states) > 2
                vel = states[:-1, keys.index("d" + key) - 2] / float(freq)
                pos = np.zeros(len(states))
                for j in range(1, len(states)):
                    pos[j] = 0.0 if (last is not None and last[j - 1] == 1) else pos[j - 1] + vel[j - 1]
                trajectories[key] = pos
        if len(states) > 2:
            trajectories["split_points"] = np.concatenate([[0], np.squeeze(np.argwhere(last == 1) + 1, axis=-1)])
        return trajectories

    def _load_task_trajectory(self, path, dataset_type, debug, clip_trajectory_to_joint_ranges):
        """The trajectory part of the task factories: 500 Hz mocap files ("real", with the bundled mini files as fall-back)
        or a recorded 100 Hz dataset ("perfect")."""
        if dataset_type == "perfect":
            traj_files = self.load_dataset_and_get_traj_files(path, 100)
            self.load_trajectory(dict(traj_files=traj_files, traj_dt=1.0 / 100, control_dt=self.dt,
                                      clip_trajectory_to_joint_ranges=clip_trajectory_to_joint_ranges), warn=False)
            return
        root = Path(os.environ.get("LOCO_MUJOCO_AMD_DATA", _PKG))
        use_mini = not (root / path).exists()
        if debug or use_mini:
            if use_mini and not debug:
                warnings.warn("Datasets not found, falling back to test datasets. Please download and install "
                              "the datasets to use this environment for imitation learning!")
            parts = path.split("/")
            parts.insert(3, "mini_datasets")
            path = "/".join(parts)
        traj_path = root / path
        if not traj_path.exists():
            traj_path = _PKG / path
        self.load_trajectory(dict(traj_path=traj_path, traj_dt=1.0 / 500, control_dt=self.dt,
                                  clip_trajectory_to_joint_ranges=clip_trajectory_to_joint_ranges), warn=False)

    def get_all_observation_keys(self):
        return [k for k, _, _ in self.obs_helper.observation_spec]

    # ------------------------------------------------------------------ reset
    def reset(self, obs=None):
        """
        Start new episodes in ALL environments (reference ``base.py:178-203``). RNG draw order per
        environment: model index, trajectory number, step in trajectory [, yaw]. Returns the initial
        observation(s), computed on the host in float64 exactly like the reference (no physics call).
        """
        rows = []
        for e in range(self.n_envs):
            self._reset_one(e, obs)
            rows.append(self._create_observation(self.obs_helper._build_obs(self._host[e])))
        self._pending_state = True
        self._pending_dof_params = None
        self._pending_variants = None
        if self._domain_rand is not None and self._domain_rand.active:
            fresh = {}
            # the models whose environments restart: every block's model, or the one this episode drew
            idxs = list(range(self._n_models)) if self._blocks else [self._current_model_idx]
            if self._use_model_compiler:
                self._pending_compile = True            # drawn and compiled on the device when the state goes up
            elif self._domain_rand.has_model_rules:
                for idx in idxs:
                    if self._blocks:
                        self._select_model(idx)
                    n_here = len(self._model_envs(idx)) if self._blocks else self.n_envs
                    first = self._variant_tables is None
                    self._ensure_variant_pool()             # first reset (or first after seed()): the whole pool, freshly drawn
                    # a batch smaller than `model_variants_per_reset` compiles only what it can use (n_envs = 1: ONE model per episode)
                    k = min(self._variants_per_reset, n_here)
                    if not first and k > 0:
                        fresh[idx] = self.refresh_model_variants(k)
            state = np.random.get_state()
            np.random.set_state(self._domain_rand_rs.get_state())
            self._pending_dof_params = self._domain_rand.sample(self.n_envs)
            if self._domain_rand.has_model_rules and not self._use_model_compiler:
                self._pending_variants = np.zeros(self.n_envs, dtype=np.int64)
                for idx in idxs:
                    envs = self._model_envs(idx) if self._blocks else np.arange(self.n_envs)
                    fr = fresh.get(idx)
                    if fr is not None and len(envs) <= len(fr):
                        self._pending_variants[envs] = fr[:len(envs)]      # one brand-new model per environment and episode
                    else:
                        self._pending_variants[envs] = np.random.randint(0, self._n_model_variants, len(envs))
            self._domain_rand_rs.set_state(np.random.get_state())
            np.random.set_state(state)
        if self._pooled:
            self._pending_variants = self._env_model.copy()          # the model every environment drew for this episode
        self._obs = np.stack(rows)
        return self._out(self._obs)

    def _reset_one(self, e, obs):
        h = self._host[e]
        h.qpos[:] = self._model.qpos0          # mj_resetData
        h.qvel[:] = 0.0
        if self._blocks and not self._grouped:
            self._select_model(self._block_of(e))
        elif self._random_env_reset:
            self._select_model(np.random.randint(0, self._n_models))
        elif self._n_models > 1:
            self._select_model((self._current_model_idx + 1) % self._n_models)
        if self._pooled or self._grouped:
            self._env_model[e] = self._current_model_idx
        self._cur_env = e
        self.setup(obs)

    def setup(self, obs):
        """Initial state of the current environment (reference ``base.py:205-241``)."""
        self._reward_function.reset_state()
        if obs is not None:
            self._init_sim_from_obs(obs)
            return
        self._check_start_mode()
        if self.trajectories is not None:
            self.set_sim_state(self._sample_start())

    def _check_start_mode(self):
        if not self.trajectories and self._random_start:
            raise ValueError("Random start not possible without trajectory data.")
        elif not self.trajectories and self._init_step_no is not None:
            raise ValueError("Setting an initial step is not possible without trajectory data.")
        elif self._init_step_no is not None and self._random_start:
            raise ValueError("Either use a random start or set an initial step, not both.")

    def _sample_start(self):
        if self._random_start:
            return self.trajectories.reset_trajectory()
        if self._init_step_no is not None:
            traj_len = self.trajectories.trajectory_length
            n_traj = self.trajectories.number_of_trajectories
            assert self._init_step_no <= traj_len * n_traj
            return self.trajectories.reset_trajectory(int(self._init_step_no % traj_len),
                                                      int(self._init_step_no / traj_len))
        return self.trajectories.reset_trajectory(substep_no=0)

    def set_sim_state(self, sample):
        """Write a trajectory sample into the simulation state, entry by entry, by NAME (``base.py:478-497``)."""
        spec = self.obs_helper.observation_spec
        assert len(sample) == len(spec)
        h = self._host[self._cur_env]
        for (key, name, ot), value in zip(spec, sample):
            if ot == ObservationType.JOINT_POS:
                h.qpos[self._model.jnt_id(name)] = np.asarray(value).reshape(-1)[0]
            elif ot == ObservationType.JOINT_VEL:
                h.qvel[self._model.jnt_id(name)] = np.asarray(value).reshape(-1)[0]
            else:
                h.site_xmat[name] = np.asarray(value, dtype=np.float64).reshape(9).copy()

    def _init_sim_from_obs(self, obs):
        assert len(obs.shape) == 1
        obs = np.concatenate([[0.0, 0.0], obs])
        spec = self.obs_helper.observation_spec
        assert len(obs) >= len(spec)
        self.set_sim_state(obs[:len(spec)])

    # ------------------------------------------------------------------ step
    def step(self, action):
        """
        One control step (= ``n_substeps`` physics steps) for every environment; the whole reference
        sequence of SURVEY.md §3.3 — un-normalise action, physics, observation, absorbing, reward on the
        previous observation — runs in one kernel launch on the device.
        """
        if self._obs is None:
            raise RuntimeError("call reset() before step()")
        a = np.asarray(action, dtype=np.float64).reshape(self.n_envs, -1)
        prev_obs = self._obs
        if self._blocks:
            if self._pending_state:
                self._upload_state()
            if self._grouped:
                obs32 = rew32 = done = None
                restarted = np.zeros(self.n_envs, dtype=bool)
                for idx in range(self._n_models):
                    envs = self._model_envs(idx)
                    if len(envs) == 0:
                        continue
                    self._select_model(idx)
                    self._activate(idx, envs)
                    o, r, d = self.backend.step(a)              # the launch runs `envs` only; the other rows come back as they were
                    if obs32 is None:
                        obs32, rew32, done = np.zeros_like(o), np.zeros_like(r), np.zeros_like(d)
                    obs32[envs], rew32[envs], done[envs] = o[envs], r[envs], d[envs]
                    restarted[envs] = self.backend.last_restarted[envs]
                self._grouped_restarted = restarted
                self._redrawn = None
                if self._auto_reset and restarted.any():
                    which = np.nonzero(restarted)[0]
                    self._redrawn = (which, self._redraw_restarted(which))
            else:
                parts = []
                for idx in range(self._n_models):
                    self._select_model(idx)
                    parts.append(self.backend.step(a[self._model_envs(idx)]))
                obs32, rew32, done = (np.concatenate([p[i] for p in parts]) for i in range(3))
        else:
            b = self.backend
            if self._pending_state:
                self._upload_state()
            if self.n_envs > 1 and not self._copy_outputs and self._reward_device_spec() is not None and hasattr(b, "step_pinned"):
                # THE FAST SURFACE of a batch (round 6): one library call (lm_step_pinned) converts the float64 action into a pinned
                # staging buffer, runs the step, converts observation (in the reference's column order) and reward to float64 on the
                # device and brings them back in ONE copy into a ring of pinned result sets — no astype / fancy-index / copy of
                # [n_envs, nobs] arrays on the host. The arrays returned are VIEWS of that ring (as the tensors of batched GPU
                # simulators are): intact for the next three step() calls; LocoEnv.make(..., copy_outputs=True) returns fresh arrays
                if not getattr(b, "_obs_order_set", False):
                    b.set_obs_order(self._obs_perm())
                    b._obs_order_set = True
                obs, reward, done = b.step_pinned(np.ascontiguousarray(a))
                self._obs = obs
                restarted = self._restarted_flags()
                info = {}
                if restarted is not None and (self._auto_reset or restarted.any()):
                    info = {"episode_restarted": restarted}
                return obs, reward, done, info
            obs32, rew32, done = b.step(a)
        obs = obs32.astype(np.float64)
        perm = self._obs_perm()
        if perm is not None:
            obs = obs[:, perm]
        if self._grouped and self._redrawn is not None:
            obs[self._redrawn[0]] = self._redrawn[1]          # the first observation of the episodes the host redrew (model + start row)
        if self._reward_device_spec() is None:
            # host-side functors (custom callbacks, ...) see what the reference's see (mushroom-rl MuJoCo.step ->
            # reward(cur_obs, action, obs, absorbing)): ONE environment's 1-D state and the UN-normalised action of
            # _preprocess_action (base.py:606-621), environment by environment
            ctrl = self._preprocess_action(a)
            from ..utils.reward import CustomReward
            overridden = type(self).reward is not LocoEnv.reward
            if isinstance(self._reward_function, CustomReward) or overridden or self.n_envs == 1:
                reward = np.array([float(self.reward(prev_obs[e], ctrl[e], obs[e], bool(done[e]))) for e in range(self.n_envs)], dtype=np.float64)
            else:
                # the built-in functors take (N, nobs) batches: one call, not a Python loop over 4096 environments
                if not getattr(self, "_warned_host_reward", False):
                    import warnings
                    warnings.warn("%s: the reward is evaluated on the host every step (the step kernel's reward is bypassed, "
                                  "e.g. because the observation carries foot forces)" % type(self).__name__)
                    self._warned_host_reward = True
                reward = np.broadcast_to(np.asarray(self._reward_function(prev_obs, ctrl, obs, done), dtype=np.float64), (self.n_envs,)).copy()
        else:
            reward = rew32.astype(np.float64)
        self._obs = obs
        # episode boundaries the DEVICE crossed in this step (auto reset: the observation already belongs to the new
        # episode; horizon reached): reported so that a learner does not bootstrap across them. Empty like the
        # reference's info dict otherwise.
        restarted = self._restarted_flags()
        info = {}
        if restarted is not None and (self._auto_reset or restarted.any()):
            # always present with device-side restarts enabled (all False in most steps): a stable key for learners
            info = {"episode_restarted": restarted if self.n_envs > 1 else bool(restarted[0])}
        if self.n_envs == 1:
            return obs[0].copy(), float(reward[0]), bool(done[0]), info
        return obs.copy(), reward, done, info

    def _restarted_flags(self):
        """bit 1 of the device's done byte per environment (episode restarted / horizon reached in the last step)."""
        if self._grouped:
            return getattr(self, "_grouped_restarted", None)
        if self._blocks:
            flags = []
            for idx in range(self._n_models):
                self._select_model(idx)
                f = getattr(self.backend, "last_restarted", None)
                if f is None:
                    return None
                flags.append(f)
            return np.concatenate(flags)
        return getattr(self.backend, "last_restarted", None)

    def _upload_state(self):
        qpos = np.stack([h.qpos for h in self._host])
        qvel = np.stack([h.qvel for h in self._host])
        prm = getattr(self, "_pending_dof_params", None)
        if self._grouped:
            self._upload_grouped(qpos, qvel, np.arange(self.n_envs))
            self._pending_dof_params = None
            self._pending_variants = None
            self._pending_compile = False
            self._pending_state = False
            return
        for idx in (range(self._n_models) if self._blocks else [self._current_model_idx]):
            envs = self._model_envs(idx) if self._blocks else np.arange(self.n_envs)
            if self._blocks:
                self._select_model(idx)
            b = self.backend
            if self._pending_compile and getattr(self, "_compiler_reseed", False):
                self._install_model_compiler()          # seed() since the last episode: the draws follow the new seed (before the
                                                        # state and the joint parameters go up: it re-creates the batch's tables)
            b.set_state(qpos[envs], qvel[envs])
            if prm is not None:
                b.set_dof_params(damping=prm[0][envs], stiffness=prm[1][envs], frictionloss=prm[2][envs])
            if self._variant_dirty and self._variant_tables is not None:      # pool entries replaced since the last upload
                b.set_model_variants(self._variant_tables)
                self._variant_dirty = False
            if getattr(self, "_pending_variants", None) is not None:
                b.set_variant_index(self._pending_variants[envs])
            if self._pending_compile:
                b.compile_models()               # reset(): a freshly drawn model per environment (reference base.py:183-185)
            goal = self._goal_rows()
            if goal is not None:
                b.set_goal(goal[:len(envs)])
        self._pending_dof_params = None
        self._pending_variants = None
        self._pending_compile = False
        self._pending_state = False

    def _upload_grouped(self, qpos, qvel, which):
        """grouped: the states of the environments ``which`` (rows of the full-size arrays) go to the batch of the model each of them
        is on — masked uploads; the batches keep what they hold for everybody else."""
        if (self._domain_rand is not None and self._domain_rand.active) or getattr(self, "_pending_variants", None) is not None:
            raise NotImplementedError("domain randomisation with a model drawn per episode among geometrically different models")
        sel = np.zeros(self.n_envs, dtype=bool)
        sel[which] = True
        for idx in range(self._n_models):
            mask = sel & (self._env_model == idx)
            if not mask.any():
                continue
            self._select_model(idx)
            b = self.backend
            b.set_state(qpos, qvel, mask)
            goal = self._goal_rows()
            if goal is not None:
                b.set_goal(goal, mask)

    def _redraw_restarted(self, envs):
        """grouped, device-side restarts: the step kernel restarted these environments on the model they had; the reference draws the
        MODEL anew with every episode (base.py:186-190). The host redraws model and start row — counter-based, keyed by (seed, global
        environment id, episodes so far): independent of batch size and sharding —, writes the row into the drawn model's batch and
        returns the episodes' first observations (built on the host from the row as the device holds it, float32, like reset() does)."""
        seed, off = self._auto_reset_seed
        tabs = self._grouped_tables
        nv = self._model.nv
        qpos = np.zeros((self.n_envs, nv)); qvel = np.zeros((self.n_envs, nv))
        rows = []
        for e in envs:
            self._episodes[e] += 1
            rng = np.random.Generator(np.random.Philox(key=int(seed) & 0xFFFFFFFFFFFFFFFF, counter=[int(off) + int(e), int(self._episodes[e]), 0, 0]))
            idx = int(rng.integers(self._n_models))
            row = tabs[idx][int(rng.integers(len(tabs[idx])))]
            self._env_model[e] = idx
            qpos[e], qvel[e] = row[:nv].astype(np.float32), row[nv:2 * nv].astype(np.float32)
            self._select_model(idx)
            h = self._host[e]
            h.qpos[:], h.qvel[:] = qpos[e], qvel[e]
            rows.append(self._create_observation(self.obs_helper._build_obs(h)))
        self._upload_grouped(qpos, qvel, envs)
        return np.stack(rows)

    def _goal_rows(self):
        """(n_envs, n_goal) constants appended to the device observation, or None."""
        return None

    def enable_auto_reset(self, seed=0, horizon=None, global_env_offset=0):
        """
        Device-side episode handling for batched rollouts: finished environments restart from a random
        trajectory sample inside the step kernel (counter-based RNG keyed by global env id, so results
        do not depend on how environments are sharded over GPUs).
        """
        if self.trajectories is None:
            raise ValueError("auto reset needs trajectory data")
        for idx in (range(self._n_models) if self._blocks else [self._current_model_idx]):
            if self._blocks:
                self._select_model(idx)
            b = self.backend
            first = int(self._model_envs(idx)[0]) if (self._blocks and not self._grouped) else 0
            if self._grouped:
                self._grouped_tables = getattr(self, "_grouped_tables", None) or [None] * self._n_models
                self._grouped_tables[idx] = self._reset_table()
                self._auto_reset_seed = (seed, global_env_offset)
            if self._pooled:
                # one block of reset rows per model (the rows carry the model's constants, e.g. the weight): a device-side
                # restart from row i puts the environment on model i // rows_per_model
                tabs = []
                for i in range(self._n_models):
                    self._select_model(i)
                    tabs.append(self._reset_table())
                b.set_reset_table(np.concatenate(tabs), seed=seed, global_env_offset=global_env_offset)
                b.set_variant_rows(len(tabs[0]))
            else:
                b.set_reset_table(self._reset_table(), seed=seed, global_env_offset=global_env_offset + first)
            if self._domain_rand is not None and self._domain_rand.active:
                b.set_dof_randomization(self._domain_rand.spec)
            if self._use_model_compiler and global_env_offset + first != 0:
                # the device compiler keys its draws by (seed, GLOBAL environment id, models had); the models it drew when it was installed
                # were keyed with offset 0 — a rank of a sharded run redraws them under its own ids (round-5 advisor: ranks that share a
                # seed started their first episodes on identical models)
                b.compile_models()
            b.set_auto_reset(True, self.info.horizon if horizon is None else horizon)
        self._auto_reset = True

    def _reset_table(self):
        """Rows [qpos | qvel | goal] for every trajectory sample."""
        tab = self.trajectories.as_state_table()
        spec = self.obs_helper.observation_spec
        nv = self._model.nv
        rows = np.zeros((tab.shape[0], 2 * nv))
        col = 0
        for key, name, ot in spec:
            if ot == ObservationType.JOINT_POS:
                rows[:, self._model.jnt_id(name)] = tab[:, col]
                col += 1
            elif ot == ObservationType.JOINT_VEL:
                rows[:, nv + self._model.jnt_id(name)] = tab[:, col]
                col += 1
            else:
                col += 9
        return rows

    # ------------------------------------------------------------------ hooks (same names as the reference)
    def _preprocess_action(self, action):
        return np.asarray(action) * self.norm_act_delta + self.norm_act_mean

    def _create_observation(self, obs):
        """Host-side observation at reset (``base.py:584-604``): the running mean of the foot forces starts at zero."""
        obs = np.asarray(obs)[2:].copy()
        return np.concatenate([obs, np.zeros(self._get_grf_size())]) if self._use_foot_forces else obs

    def _get_grf_size(self):
        return 12

    def _grf_group_names(self):
        """Force groups in observation order (``base.py:667-679``)."""
        return ["foot_r", "front_foot_r", "foot_l", "front_foot_l"]

    def is_absorbing(self, obs):
        return self._has_fallen(obs) if self._use_absorbing_states else False

    def reward(self, state, action, next_state, absorbing):
        return self._reward_function(state, action, next_state, absorbing)

    def _has_fallen(self, obs, return_err_msg=False):
        raise NotImplementedError

    def _termination_spec(self):
        """[(obs index, low, high)]: absorbing iff any listed entry leaves [low, high]."""
        raise NotImplementedError

    def _get_observation_space(self):
        """``base.py:566-583``: the simulator's entries without the two horizontal root coordinates [+ foot forces]."""
        low, high = self.info.observation_space.low[2:], self.info.observation_space.high[2:]
        if self._use_foot_forces:
            inf = np.full(self._get_grf_size(), np.inf)
            return np.concatenate([low, -inf]), np.concatenate([high, inf])
        return low, high

    def _get_reward_function(self, reward_type, reward_params):
        if reward_type == "custom":
            return CustomReward(**reward_params)
        elif reward_type == "target_velocity":
            idx = self.get_obs_idx("dq_pelvis_tx")
            assert len(idx) == 1
            return TargetVelocityReward(x_vel_idx=idx[0], **reward_params)
        elif reward_type == "x_pos":
            idx = self.get_obs_idx("q_pelvis_tx")
            assert len(idx) == 1
            return PosReward(pos_idx=idx[0])
        elif reward_type is None:
            return NoReward()
        raise NotImplementedError("The specified reward has not been implemented: %s" % reward_type)

    def get_obs_idx(self, key):
        return [i - 2 for i in self.obs_helper.obs_idx_map[key]]

    def _get_idx(self, keys):
        if not isinstance(keys, list):
            keys = [keys]
        return np.concatenate([self.obs_helper.obs_idx_map[k] for k in keys]) - 2

    def _get_from_obs(self, obs, keys):
        obs = np.concatenate([[0.0, 0.0], obs])
        if not isinstance(keys, list):
            keys = [keys]
        return np.concatenate([self.obs_helper.get_from_obs(obs, k) for k in keys])

    def get_kinematic_obs_mask(self):
        return np.arange(len(self.obs_helper.observation_spec) - 2)

    def _len_qpos_qvel(self):
        keys = self.get_all_observation_keys()
        return len([k for k in keys if k.startswith("q_")]), len([k for k in keys if k.startswith("dq_")])

    def _get_interpolate_map_params(self):
        return None

    def _get_interpolate_remap_params(self):
        return None

    @staticmethod
    def _interpolate_map(traj, **interpolate_map_params):
        return np.array(traj)

    @staticmethod
    def _interpolate_remap(traj, **interpolate_remap_params):
        return [obs for obs in traj]

    # ------------------------------------------------------------------ task description for the device
    def _n_goal(self):
        return 0

    def _device_task(self):
        """Everything around the physics that the step kernel evaluates itself (see ``lowering.lower``)."""
        qpos_idx, qvel_idx = [], []
        for key, name, ot in self.obs_helper.observation_spec[2:]:
            if ot == ObservationType.JOINT_POS:
                qpos_idx.append(self._model.jnt_id(name))
            elif ot == ObservationType.JOINT_VEL:
                qvel_idx.append(self._model.jnt_id(name))
        n_goal = self._n_goal()
        grf_groups = [list(self._collision_groups[g]) for g in self._grf_group_names()] if self._use_foot_forces else []
        nobs = len(qpos_idx) + len(qvel_idx) + n_goal + 3 * len(grf_groups)
        assert nobs == self.info.observation_space.shape[0], "device observation layout does not match the space"
        spec = self._reward_device_spec(nobs, 3 * len(grf_groups))
        rtype, rparams = spec if spec is not None else (0, [])
        term = self._termination_spec() if self._use_absorbing_states else []
        return dict(nobs=nobs, qpos_obs_idx=qpos_idx, qvel_obs_idx=qvel_idx, n_goal=n_goal, grf_groups=grf_groups,
                    act_ctrl_idx=self._action_indices, act_mean=self.norm_act_mean, act_delta=self.norm_act_delta,
                    term=term, reward_type=rtype, reward_params=rparams, n_substeps=self._n_substeps)

    def _reward_device_spec(self, nobs=None, n_grf=None):
        """The reward functor's device form, or None when it has to run on the host. With ``use_foot_forces`` the foot
        forces END the observation, so the reference's negative indices (UnitreeA1's velocity-vector reward reads
        ``state[-3:]``, ``unitreeA1.py:491-497``) land on foot-force entries: that quirk is reproduced by evaluating the
        functor on the host — the kernel's reward only reads joint and goal entries."""
        spec = self._reward_function.device_spec()
        if spec is None:
            return None
        if nobs is None:
            nobs = self.info.observation_space.shape[0]
            n_grf = self._get_grf_size() if self._use_foot_forces else 0
        idx = {1: spec[1][:1], 2: spec[1][:5]}.get(spec[0], [])
        if n_grf and any(int(i) % nobs >= nobs - n_grf for i in idx):
            return None
        return spec

    def _chain_model(self, model=None):
        from ..lowering import lower
        return lower(self._model if model is None else model, self._device_task())[0]

    # ------------------------------------------------------------------ misc surface
    def _out(self, obs):
        return obs[0].copy() if self.n_envs == 1 else obs.copy()

    def render(self, record=False):
        raise NotImplementedError("rendering is out of scope of the headless batched simulator (SURVEY.md §2 row 22)")

    def stop(self):
        pass

    def seed(self, seed=None):
        np.random.seed(seed)
        if self._domain_rand is not None:
            self._domain_rand_rs = np.random.RandomState(seed)
            self._variant_tables = None          # the pool is a function of the seed: rebuilt at the next reset()
            self._compiler_reseed = self._backend is not None      # ... and so is the device compiler's draw sequence

    def play_trajectory(self, n_episodes=None, n_steps_per_episode=None, render=False, **kwargs):
        """Kinematic replay of the loaded trajectory (reference ``base.py:314-386``): yields the observation
        of every replayed sample; no dynamics, no rendering."""
        assert self.trajectories is not None
        out = []
        sample = self.trajectories.reset_trajectory(substep_no=1)
        self._cur_env = 0
        n_episodes = 1 if n_episodes is None else n_episodes
        for _ in range(n_episodes):
            steps = 0
            while sample is not None and (n_steps_per_episode is None or steps < n_steps_per_episode):
                self.set_sim_state(sample)
                out.append(self._create_observation(self.obs_helper._build_obs(self._host[0])))
                sample = self.trajectories.get_next_sample()
                steps += 1
            sample = self.trajectories.reset_trajectory(substep_no=1)
        return np.array(out)


    def play_trajectory_from_velocity(self, n_episodes=None, n_steps_per_episode=None, render=False, **kwargs):
        """Replay of the loaded trajectory from its joint VELOCITIES (reference ``base.py:388-476``): the positions of the
        first sample, then ``qpos += dt * qvel`` with the trajectory's velocities; the goal / site entries of every sample
        are taken as they are. Returns the observation of every replayed step; no dynamics, no rendering.

        At the end of the trajectory the replay restarts from a new reset and continues, like the reference. The reference's
        defaults are unbounded (it renders until stopped); this replay returns an array, so ``n_episodes`` defaults to 1 and an
        episode without ``n_steps_per_episode`` ends at the end of the trajectory."""
        assert self.trajectories is not None
        self._cur_env = 0
        self.reset()
        sample = self.trajectories.get_current_sample()
        self.set_sim_state(sample)
        len_qpos, len_qvel = self._len_qpos_qvel()
        curr_qpos = np.array([np.asarray(x, dtype=np.float64).reshape(-1)[0] for x in sample[0:len_qpos]])
        n_episodes = 1 if n_episodes is None else n_episodes
        out = []
        for _ in range(n_episodes):
            steps = 0
            while n_steps_per_episode is None or steps < n_steps_per_episode:
                qvel = sample[len_qpos:len_qpos + len_qvel]
                sample = list(sample)
                sample[:len_qpos] = [qp + self.dt * np.asarray(qv, dtype=np.float64).reshape(-1)[0] for qp, qv in zip(curr_qpos, qvel)]
                self.set_sim_state(sample)
                curr_qpos = self._get_joint_pos()
                out.append(self._create_observation(self.obs_helper._build_obs(self._host[0])))
                steps += 1
                sample = self.trajectories.get_next_sample()
                if sample is None:                      # end of the trajectory: restart and go on (reference base.py:452-455)
                    if n_steps_per_episode is None:     # ... unbounded there; a replay that RETURNS its observations ends here
                        break
                    self.reset()
                    sample = self.trajectories.get_current_sample()
                    curr_qpos = np.array([np.asarray(x, dtype=np.float64).reshape(-1)[0] for x in sample[0:len_qpos]])
            self.reset()
            sample = self.trajectories.get_current_sample()
            curr_qpos = np.array([np.asarray(x, dtype=np.float64).reshape(-1)[0] for x in sample[0:len_qpos]])
        return np.array(out)

    def _get_joint_pos(self):
        """Positions of the observed joints, in observation-specification order (reference ``base.py:709-718``)."""
        h = self._host[self._cur_env]
        return np.array([h.qpos[self._model.jnt_id(name)] for key, name, ot in self.obs_helper.observation_spec
                         if ot == ObservationType.JOINT_POS])

    def _get_joint_vel(self):
        """Velocities of the observed joints, in observation-specification order (reference ``base.py:720-729``)."""
        h = self._host[self._cur_env]
        return np.array([h.qvel[self._model.jnt_id(name)] for key, name, ot in self.obs_helper.observation_spec
                         if ot == ObservationType.JOINT_VEL])

    @staticmethod
    def _delete_from_xml_handle(xml_handle, joints_to_remove, motors_to_remove, equ_constraints):
        """Remove joints, motors and equality constraints from an MJCF handle (reference ``base.py:899-922``)."""
        from .atlas import Atlas
        return Atlas._delete_from_xml_handle(xml_handle, joints_to_remove, motors_to_remove, equ_constraints)


class ValidTaskConf:
    """Valid (task, mode, dataset type) combinations of an environment (reference ``base.py:972-1041``)."""

    def __init__(self, tasks=None, modes=None, data_types=None, non_combinable=None):
        self.tasks, self.modes, self.data_types, self.non_combinable = tasks, modes, data_types, non_combinable
        for nc in (non_combinable or []):
            assert len(nc) == 3

    def get_all(self):
        return (copy.deepcopy(self.tasks), copy.deepcopy(self.modes), copy.deepcopy(self.data_types),
                copy.deepcopy(self.non_combinable))

    def get_all_combinations(self):
        confs = []
        for t, m, dt in product(self.tasks or [None], self.modes or [None], self.data_types or [None]):
            conf = {}
            if t is not None:
                conf["task"] = t
            if m is not None:
                conf["mode"] = m
            if dt is not None:
                conf["data_type"] = dt
            if self.non_combinable is None:
                confs.append(conf)
                continue
            for bad_t, bad_m, bad_dt in self.non_combinable:
                if not ((bad_t is None or t == bad_t) and (bad_m is None or m == bad_m)
                        and (bad_dt is None or dt == bad_dt)):
                    confs.append(conf)
        return confs
