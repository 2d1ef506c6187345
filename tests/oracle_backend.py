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

    def set_state(self, qpos, qvel, mask=None):
        self.qpos[:] = qpos
        self.qvel[:] = qvel
        self.warm[:] = 0
        self.act[:] = 0

    def set_goal(self, goal, mask=None):
        self.goal = np.array(goal, dtype=np.float64)

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
        if self.prev_obs is None:
            self.prev_obs = np.stack([self._obs(e) for e in range(self.n)])
        obs, rew, done = [], [], []
        for e in range(self.n):
            ctrl = np.zeros(env._model.nu)
            ctrl[env._action_indices] = env._preprocess_action(action[e])
            if env._use_foot_forces:
                obs.append(self._step_with_foot_forces(e, ctrl))
                done.append(bool(env.is_absorbing(obs[-1])))
                rew.append(env.reward(self.prev_obs[e], action[e], obs[-1], done[-1]))
                continue
            if self.act.shape[1]:
                q, v, a, w, st = self.oracle.step_act(self.qpos[e], self.qvel[e], self.act[e], ctrl, env._n_substeps, self.warm[e])
                self.act[e] = a
            else:
                q, v, w, st = self.oracle.step(self.qpos[e], self.qvel[e], ctrl, env._n_substeps, self.warm[e])
            self.qpos[e], self.qvel[e], self.warm[e] = q, v, w
            self.stats_log.append(st)
            o = self._obs(e)
            obs.append(o)
            done.append(bool(env.is_absorbing(o)))
            rew.append(env.reward(self.prev_obs[e], action[e], o, done[-1]))
        self.prev_obs = np.stack(obs)
        return np.stack(obs), np.array(rew, dtype=np.float64), np.array(done)


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
            env._backend = OracleBatch(env, len(env._model_envs(idx)) if env._blocks else None)
        env._select_model(current)
        return env
    env._backend = OracleBatch(env)
    return env
