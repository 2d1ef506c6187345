"""
Common part of the robot humanoids (Atlas, Talos) — host-side mirror of the reference's
``loco_mujoco/environments/humanoids/base_robot_humanoid.py``: the dataset keys, the observation mask, the carried
weight (``hold_weight``: a box fixed to the torso, its mass appended to the observation; without ``weight_mass`` one model
per mass in 0.1 / 1 / 5 / 10 kg, one of them drawn per episode) and the task factory shared by the robots.

On the device one batch = one model table: with ``n_envs=1`` the four-weights variant switches between four batches at
reset like the reference; with ``n_envs>1`` the environments form four contiguous blocks, one weight each
(``LocoEnv._init_models``).
"""

import os
import warnings
from pathlib import Path

import numpy as np

from .base import LocoEnv

_PKG = Path(__file__).resolve().parent.parent


class BaseRobotHumanoid(LocoEnv):

    _valid_weights = [0.1, 1.0, 5.0, 10.0]
    _hidable_obs = ("positions", "velocities", "foot_forces", "weight")

    # ------------------------------------------------------------------ carried weight
    def _weight_list(self, hold_weight, weight_mass, n_envs):
        """Masses of the models this environment holds (``atlas.py:318-331`` / ``talos.py:310-324``)."""
        if not hold_weight:
            return [None]
        if weight_mass is not None:
            return [float(weight_mass)]
        return list(self._valid_weights)

    def _init_weight_models(self, models, weights):
        self._weights = weights
        if len(models) > 1:
            self._init_models(models)

    def _models_differ_like_variants(self):
        """The carried-weight models differ in the torso link's inertial numbers only: one batch, a weight per episode."""
        return bool(self._hold_weight)

    def _weight_obs(self):
        """Mass of the current model's ``weight`` body (``base_robot_humanoid.py:118-122``)."""
        return np.array([self._model.body_mass[self._model.body_names.index("weight")]])

    def create_dataset(self, ignore_keys=None):
        """``base_robot_humanoid.py:18-37``: the two horizontal pelvis coordinates are not part of the dataset."""
        return super().create_dataset(["q_pelvis_tx", "q_pelvis_tz"] if ignore_keys is None else ignore_keys)

    def get_mask(self, obs_to_hide):
        """Boolean mask over the observation that hides groups of entries (``base_robot_humanoid.py:39-91``)."""
        if type(obs_to_hide) == str:
            obs_to_hide = (obs_to_hide,)
        assert all(x in self._hidable_obs for x in obs_to_hide), "Some of the observations you want to hide are not" \
                                                                 "supported. Valid observations to hide are %s." \
                                                                 % (self._hidable_obs,)
        pos_dim, vel_dim = self._len_qpos_qvel()
        mask = [np.full(pos_dim - 2, "positions" not in obs_to_hide), np.full(vel_dim, "velocities" not in obs_to_hide)]
        if self._use_foot_forces:
            mask.append(np.full(self._get_grf_size(), "foot_forces" not in obs_to_hide))
        else:
            assert "foot_forces" not in obs_to_hide, "Creating a mask to hide foot forces without activating " \
                                                     "the latter is not allowed."
        if self._hold_weight:
            mask.append(np.full(1, "weight" not in obs_to_hide))
        else:
            assert "weight" not in obs_to_hide, "Creating a mask to hide the carried weight without activating " \
                                                "the latter is not allowed."
        return np.concatenate(mask).ravel()

    def _get_observation_space(self):
        low, high = super()._get_observation_space()
        if self._hold_weight:
            low, high = np.concatenate([low, [self._valid_weights[0]]]), np.concatenate([high, [self._valid_weights[-1]]])
        return low, high

    def _create_observation(self, obs):
        obs = super()._create_observation(obs)
        return np.concatenate([obs, self._weight_obs()]) if self._hold_weight else obs

    # the device appends constants before the foot forces, the reference the weight after them
    def _n_goal(self):
        return 1 if self._hold_weight else 0

    def _goal_rows(self):
        if not self._hold_weight:
            return None
        if self._pooled:
            w = np.array([m.body_mass[m.body_names.index("weight")] for m in self._models])
            return w[self._env_model][:, None]
        return np.tile(self._weight_obs(), (self.n_envs, 1))

    def _reset_table(self):
        rows = super()._reset_table()
        return np.concatenate([rows, np.tile(self._weight_obs(), (len(rows), 1))], axis=1) if self._hold_weight else rows

    def _obs_perm(self):
        if not (self._hold_weight and self._use_foot_forces):
            return None
        n, g = self.info.observation_space.shape[0], self._get_grf_size()
        return np.concatenate([np.arange(n - g - 1), np.arange(n - g, n), [n - g - 1]])

    # ------------------------------------------------------------------ task factory
    @staticmethod
    def generate(env, path, task="walk", dataset_type="real", debug=False, clip_trajectory_to_joint_ranges=False, **kwargs):
        """``base_robot_humanoid.py:145-260``: walk / carry at 1.25 m/s, run at 2.5 m/s; "real" = 500 Hz mocap
        trajectories, "perfect" = a recorded 100 Hz dataset (states, actions, ...: a download of the reference project)."""
        reward_type = kwargs.pop("reward_type", "target_velocity")
        reward_params = kwargs.pop("reward_params", dict(target_velocity=2.5 if task == "run" else 1.25))
        if task == "carry":
            kwargs["hold_weight"] = True
        mdp = env(reward_type=reward_type, reward_params=reward_params, **kwargs)
        mdp._load_task_trajectory(path, dataset_type, debug, clip_trajectory_to_joint_ranges)
        return mdp
