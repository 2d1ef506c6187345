"""
Observation/action-space plumbing that the reference takes from mushroom-rl (third party, absent
here): ``ObservationType``, ``ObservationHelper``, ``spaces.Box``, ``MDPInfo`` (SURVEY.md Appendix C;
used by the reference at ``loco_mujoco/environments/base.py:12-17,119-126,478-497``).
"""

from enum import Enum

import numpy as np


class ObservationType(Enum):
    """Kinds of observation entries used by the supported environments (the reference uses exactly
    JOINT_POS, JOINT_VEL and SITE_ROT)."""
    JOINT_POS = 0
    JOINT_VEL = 1
    SITE_ROT = 2


_OBS_DIM = {ObservationType.JOINT_POS: 1, ObservationType.JOINT_VEL: 1, ObservationType.SITE_ROT: 9}


class Box:
    """Continuous box space (same attributes as ``mushroom_rl.utils.spaces.Box``)."""

    def __init__(self, low, high):
        self._low = np.array(low, dtype=np.float64)
        self._high = np.array(high, dtype=np.float64)
        assert self._low.shape == self._high.shape

    @property
    def low(self):
        return self._low

    @property
    def high(self):
        return self._high

    @property
    def shape(self):
        return self._low.shape


class MDPInfo:
    def __init__(self, observation_space, action_space, gamma, horizon, dt=None):
        self.observation_space = observation_space
        self.action_space = action_space
        self.gamma = gamma
        self.horizon = horizon
        self.dt = dt


class ObservationHelper:
    """
    Maps an observation spec ``[(key, name, ObservationType), ...]`` to flat indices, limits and
    gather operations on host-side state arrays (``qpos``, ``qvel``, site rotation matrices).
    Entries are resolved BY NAME, so the observation follows the spec order, not the model order
    (SURVEY.md Appendix G).
    """

    def __init__(self, observation_spec, model):
        self.observation_spec = list(observation_spec)
        self.model = model
        self.obs_idx_map = {}
        self.joint_pos_idx, self.joint_vel_idx = [], []
        self.obs_low, self.obs_high = [], []
        self.build_obs_map = []                      # (kind, model index) per spec entry
        k = 0
        for key, name, ot in self.observation_spec:
            n = _OBS_DIM[ot]
            idx = list(range(k, k + n))
            self.obs_idx_map[key] = idx
            if ot == ObservationType.JOINT_POS:
                j = model.jnt_id(name)
                self.joint_pos_idx += idx
                if model.jnt_limited[j]:
                    self.obs_low.append(model.jnt_range[j, 0])
                    self.obs_high.append(model.jnt_range[j, 1])
                else:
                    self.obs_low.append(-np.inf)
                    self.obs_high.append(np.inf)
                self.build_obs_map.append((ot, j))
            elif ot == ObservationType.JOINT_VEL:
                j = model.jnt_id(name)
                self.joint_vel_idx += idx
                self.obs_low.append(-np.inf)
                self.obs_high.append(np.inf)
                self.build_obs_map.append((ot, j))
            else:
                self.obs_low += [-np.inf] * n
                self.obs_high += [np.inf] * n
                self.build_obs_map.append((ot, name))
            k += n
        self.obs_length = k

    def get_obs_limits(self):
        return np.array(self.obs_low), np.array(self.obs_high)

    def get_from_obs(self, obs, key):
        return obs[self.obs_idx_map[key]]

    def get_joint_pos_from_obs(self, obs):
        return obs[self.joint_pos_idx]

    def get_joint_vel_from_obs(self, obs):
        return obs[self.joint_vel_idx]

    def _build_obs(self, data):
        """``data``: object with ``qpos`` (nq,), ``qvel`` (nv,), ``site_xmat`` {name: (9,)}."""
        parts = []
        for ot, ref in self.build_obs_map:
            if ot == ObservationType.JOINT_POS:
                parts.append([data.qpos[ref]])
            elif ot == ObservationType.JOINT_VEL:
                parts.append([data.qvel[ref]])
            else:
                parts.append(np.asarray(data.site_xmat[ref]).reshape(9))
        return np.concatenate(parts)
