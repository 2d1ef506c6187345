"""Goal holder (reference: ``loco_mujoco/utils/goals.py:4``)."""

import copy


class GoalDirectionVelocity:
    """Desired walking direction (yaw, rad) and speed (m/s)."""

    def __init__(self):
        self._direction = None
        self._velocity = None

    def set_goal(self, direction, velocity):
        self._direction, self._velocity = direction, velocity

    def get_goal(self):
        return self.get_direction(), self.get_velocity()

    __call__ = get_goal

    def get_direction(self):
        assert self._direction is not None
        return copy.deepcopy(self._direction)

    def get_velocity(self):
        assert self._velocity is not None
        return copy.deepcopy(self._velocity)
