"""
Reward functors (mirror of the reference's ``loco_mujoco/utils/reward.py``: same class names and constructor arguments). All
of them look at the PREVIOUS observation only (``reward.py:73,110-115``) and take one observation ``(nobs,)`` or a batch
``(N, nobs)``. A functor that the step kernel can evaluate itself says so through ``device_spec()`` -> (reward type, up to 8
parameters); the others run on the host (``n_envs=1`` / custom rewards). The parity tests compare kernel and functor.
"""

import numpy as np


class RewardInterface:
    """``reward(state, action, next_state, absorbing)``; subclasses implement ``_of(state)`` on a numpy state."""

    def __call__(self, state, action, next_state, absorbing):
        return self._of(np.asarray(state))

    def _of(self, state):
        raise NotImplementedError

    def reset_state(self):
        """Nothing to forget: none of the functors below keeps history."""

    def device_spec(self):
        return None


class NoReward(RewardInterface):
    def _of(self, state):
        return 0

    def device_spec(self):
        return 0, []


class PosReward(RewardInterface):
    """The observation entry ``pos_idx`` itself."""

    def __init__(self, pos_idx):
        self._pos_idx = pos_idx

    def _of(self, state):
        return state[..., self._pos_idx]


class CustomReward(RewardInterface):
    """``reward_callback(state, action, next_state)`` supplied by the user; host only."""

    def __init__(self, reward_callback=None):
        self._reward_callback = reward_callback

    def __call__(self, state, action, next_state, absorbing):
        return 0 if self._reward_callback is None else self._reward_callback(state, action, next_state)


class TargetVelocityReward(RewardInterface):
    """exp(-(v_x - v*)^2)  (``reward.py:66-74``)."""

    def __init__(self, target_velocity, x_vel_idx):
        self._target_vel, self._x_vel_idx = target_velocity, x_vel_idx

    def _of(self, state):
        return np.exp(-(state[..., self._x_vel_idx] - self._target_vel) ** 2)

    def device_spec(self):
        return 1, [self._x_vel_idx, self._target_vel]


class MultiTargetVelocityReward(RewardInterface):
    """exp(-(v_x - s v*)^2): the size factor s is looked up with the indicator bits that end the state, most significant
    bit first (``reward.py:77-97``)."""

    def __init__(self, target_velocity, x_vel_idx, env_id_len, scalings):
        self._target_vel, self._x_vel_idx = target_velocity, x_vel_idx
        self._env_id_len, self._scalings = env_id_len, np.asarray(scalings)
        self._bit_weights = 1 << np.arange(env_id_len)[::-1]

    def _scaling(self, state):
        bits = np.asarray(state)[..., -self._env_id_len:].astype(int)
        return self._scalings[(bits * self._bit_weights).sum(axis=-1)]

    def _of(self, state):
        return np.exp(-(state[..., self._x_vel_idx] - self._target_vel * self._scaling(state)) ** 2)

    # no device_spec: a batch of one size has a constant target, BaseHumanoid4Ages supplies that spec per model


class VelocityVectorReward(RewardInterface):
    """exp(-5 |v_xy - v_goal (cos, sin)|)  (``reward.py:100-117``)."""

    def __init__(self, x_vel_idx, y_vel_idx, angle_idx, goal_vel_idx):
        self._x_vel_idx, self._y_vel_idx = x_vel_idx, y_vel_idx
        self._angle_idx, self._goal_vel_idx = angle_idx, goal_vel_idx

    def _of(self, state):
        planar = np.stack([state[..., self._x_vel_idx], state[..., self._y_vel_idx]], axis=-1)
        wanted = state[..., self._goal_vel_idx] * state[..., self._angle_idx]
        return np.exp(-5.0 * np.linalg.norm(planar - wanted, axis=-1))

    def device_spec(self):
        return 2, [self._x_vel_idx, self._y_vel_idx, self._angle_idx[0], self._angle_idx[1], self._goal_vel_idx[0]]
