"""
Reward functors (reference: ``loco_mujoco/utils/reward.py``). All are evaluated on the PREVIOUS
observation ``state`` (``reward.py:73,110-115``). They accept a single observation ``(nobs,)`` or a
batch ``(N, nobs)``. The batched device path evaluates the same formulas inside the step kernel
(``include/lm_model_blob.h`` reward_type); these host classes are the n_envs=1 / custom-reward path
and what the parity tests compare the kernel with.
"""

import numpy as np


class RewardInterface:
    def __call__(self, state, action, next_state, absorbing):
        raise NotImplementedError

    def reset_state(self):
        pass

    # (reward_type, params[8]) understood by the device kernel, or None if host-only
    def device_spec(self):
        return None


class NoReward(RewardInterface):
    def __call__(self, state, action, next_state, absorbing):
        return 0

    def device_spec(self):
        return 0, []


class PosReward(RewardInterface):
    def __init__(self, pos_idx):
        self._pos_idx = pos_idx

    def __call__(self, state, action, next_state, absorbing):
        return np.asarray(state)[..., self._pos_idx]


class CustomReward(RewardInterface):
    def __init__(self, reward_callback=None):
        self._reward_callback = reward_callback

    def __call__(self, state, action, next_state, absorbing):
        if self._reward_callback is None:
            return 0
        return self._reward_callback(state, action, next_state)


class TargetVelocityReward(RewardInterface):
    """exp(-(v_x - v*)^2)  (reference ``reward.py:66-74``)."""

    def __init__(self, target_velocity, x_vel_idx):
        self._target_vel = target_velocity
        self._x_vel_idx = x_vel_idx

    def __call__(self, state, action, next_state, absorbing):
        x_vel = np.asarray(state)[..., self._x_vel_idx]
        return np.exp(-np.square(x_vel - self._target_vel))

    def device_spec(self):
        return 1, [self._x_vel_idx, self._target_vel]


class MultiTargetVelocityReward(RewardInterface):
    """exp(-(v_x - s v*)^2) with the size factor s decoded from the indicator bits at the end of the state
    (reference ``reward.py:77-97``)."""

    def __init__(self, target_velocity, x_vel_idx, env_id_len, scalings):
        self._target_vel = target_velocity
        self._env_id_len = env_id_len
        self._scalings = scalings
        self._x_vel_idx = x_vel_idx

    def _scaling(self, state):
        env_id = np.asarray(state)[..., -self._env_id_len:].astype(int)
        ind = (env_id * (1 << np.arange(self._env_id_len)[::-1])).sum(axis=-1)
        return np.asarray(self._scalings)[ind]

    def __call__(self, state, action, next_state, absorbing):
        x_vel = np.asarray(state)[..., self._x_vel_idx]
        return np.exp(-np.square(x_vel - self._target_vel * self._scaling(state)))

    def device_spec(self):
        return None          # overridden per environment: a single-size batch has a constant target (see BaseHumanoid4Ages)


class VelocityVectorReward(RewardInterface):
    """exp(-5 * || v_xy - v_goal * (cos, sin) ||)  (reference ``reward.py:100-117``)."""

    def __init__(self, x_vel_idx, y_vel_idx, angle_idx, goal_vel_idx):
        self._x_vel_idx = x_vel_idx
        self._y_vel_idx = y_vel_idx
        self._angle_idx = angle_idx
        self._goal_vel_idx = goal_vel_idx

    def __call__(self, state, action, next_state, absorbing):
        state = np.asarray(state)
        vel = np.stack([state[..., self._x_vel_idx], state[..., self._y_vel_idx]], axis=-1)
        cos_sine = state[..., self._angle_idx]
        des = state[..., self._goal_vel_idx] * cos_sine
        return np.exp(-5.0 * np.linalg.norm(vel - des, axis=-1))

    def device_spec(self):
        return 2, [self._x_vel_idx, self._y_vel_idx, self._angle_idx[0], self._angle_idx[1], self._goal_vel_idx[0]]
