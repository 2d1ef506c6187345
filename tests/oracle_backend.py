"""
TEST-ONLY stand-in for ``loco_mujoco_amd.backend.HipBatch`` built on the fp64 CPU oracle, so that the
host logic of ``LocoEnv`` (reset, observation assembly, action un-normalisation, termination, reward)
can be checked end-to-end against the reference's golden rollouts on a machine without a GPU.
Never imported by the product.
"""

import numpy as np

from oracle.model_blob import pack_model
from oracle.pyoracle import Oracle


class OracleBatch:
    def __init__(self, env, n=None):
        self.env = env
        self.oracle = Oracle(pack_model(env._model))
        self.n = env.n_envs if n is None else n
        self.qpos = np.zeros((self.n, env._model.nv))
        self.qvel = np.zeros((self.n, env._model.nv))
        self.warm = np.zeros((self.n, env._model.nv))
        self.act = np.zeros((self.n, getattr(env._model, "na", 0)))      # muscle activations (mj_resetData zeroes them)
        self.goal = None
        self.prev_obs = None
        self.stats_log = []
        self.active = None              # set_active: the environments a step runs (None: all)
        self.last_restarted = np.zeros(self.n, dtype=bool)
        self._last = None               # what the last step returned: rows of inactive environments come back as they were
        self._stale = np.ones(self.n, dtype=bool)

    def set_active(self, env_ids):
        self.active = None if env_ids is None else np.array(env_ids, dtype=np.int64)

    def set_state(self, qpos, qvel, mask=None):
        m = slice(None) if mask is None else np.asarray(mask, dtype=bool)
        self.qpos[m] = np.asarray(qpos)[m]
        self.qvel[m] = np.asarray(qvel)[m]
        self.warm[m] = 0
        self.act[m] = 0
        self._stale = np.ones(self.n, dtype=bool) if mask is None else (self._stale | m)      # their "previous observation" is the new state's

    def set_goal(self, goal, mask=None):
        g = np.array(goal, dtype=np.float64)
        if mask is None or self.goal is None:
            self.goal = g
        else:
            m = np.asarray(mask, dtype=bool)
            self.goal[m] = g[m]

    def _obs(self, e):
        env = self.env
        qi = [env._model.jnt_id(n) for k, n, t in env.obs_helper.observation_spec[2:] if k.startswith("q_")]
        vi = [env._model.jnt_id(n) for k, n, t in env.obs_helper.observation_spec[2:] if k.startswith("dq_")]
        parts = [self.qpos[e, qi], self.qvel[e, vi]]
        if self.goal is not None:
            parts.append(self.goal[e])
        return np.concatenate(parts)

    def step(self, action):
        env = self.env
        run = range(self.n) if self.active is None else [int(e) for e in self.active]
        if self.prev_obs is None:
            self.prev_obs = np.stack([self._obs(e) for e in range(self.n)])
        for e in run:
            if self._stale[e]:
                self.prev_obs[e] = self._obs(e)
                self._stale[e] = False
        obs, rew, done = {}, {}, {}
        for e in run:
            ctrl = np.zeros(env._model.nu)
            ctrl[env._action_indices] = env._preprocess_action(action[e])
            if env._use_foot_forces:
                obs[e] = self._step_with_foot_forces(e, ctrl)
                done[e] = bool(env.is_absorbing(obs[e]))
                rew[e] = env.reward(self.prev_obs[e], action[e], obs[e], done[e])
                continue
            if self.act.shape[1]:
                q, v, a, w, st = self.oracle.step_act(self.qpos[e], self.qvel[e], self.act[e], ctrl, env._n_substeps, self.warm[e])
                self.act[e] = a
            else:
                q, v, w, st = self.oracle.step(self.qpos[e], self.qvel[e], ctrl, env._n_substeps, self.warm[e])
            self.qpos[e], self.qvel[e], self.warm[e] = q, v, w
            self.stats_log.append(st)
            o = self._obs(e)
            obs[e] = o
            done[e] = bool(env.is_absorbing(o))
            rew[e] = env.reward(self.prev_obs[e], action[e], o, done[e])
        if self._last is None:
            width = len(next(iter(obs.values()))) if obs else self.prev_obs.shape[1]
            self._last = [np.zeros((self.n, width)), np.zeros(self.n), np.zeros(self.n, dtype=bool)]
        for e in run:
            self._last[0][e], self._last[1][e], self._last[2][e] = obs[e], rew[e], done[e]
            self.prev_obs[e] = obs[e][:self.prev_obs.shape[1]]
        return self._last[0].copy(), self._last[1].copy(), self._last[2].copy()


def _grf_step(self, e, ctrl):
    """The reference's loop with use_foot_forces (base.py:94-98,623-631): n intermediate steps of one substep, after each
    the contact-frame force of the first floor contact of every foot group; observation += mean / 1000."""
    env = self.env
    m = env._model
    floor = m.geom_names.index("floor")
    groups = [[m.geom_names.index(g) for g in env._collision_groups[name]] for name in env._grf_group_names()]
    total = np.zeros((len(groups), 3))
    for _ in range(env._n_substeps):
        q, v, a, w, cons = self.oracle.step_contact_forces(self.qpos[e], self.qvel[e], ctrl, self.warm[e],
                                                           self.act[e] if self.act.shape[1] else None)
        self.qpos[e], self.qvel[e], self.warm[e] = q, v, w
        if self.act.shape[1]:
            self.act[e] = a
        for gi, ids in enumerate(groups):
            for g1, g2, f in cons:
                if (g1 == floor and g2 in ids) or (g2 == floor and g1 in ids):
                    total[gi] += f
                    break
    o = self._obs(e)
    n_goal = 0 if self.goal is None else self.goal.shape[1]
    grf = (total / env._n_substeps / 1000.0).ravel()
    return np.concatenate([o, grf])


OracleBatch._step_with_foot_forces = _grf_step


def attach(env):
    if getattr(env, "_n_models", 1) > 1:                 # one oracle per model of a multi-model environment
        current = env._current_model_idx
        for idx in range(env._n_models):
            env._select_model(idx)
            env._backend = OracleBatch(env, len(env._model_envs(idx)) if (env._blocks and not getattr(env, "_grouped", False)) else None)
        env._select_model(current)
        return env
    env._backend = OracleBatch(env)
    return env
