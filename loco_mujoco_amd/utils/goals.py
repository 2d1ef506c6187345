"""Goal holder of the quadruped tasks (mirror of the reference's ``loco_mujoco/utils/goals.py:4``): desired walking
direction (yaw, rad) and speed (m/s), handed out as copies."""

import copy


class GoalDirectionVelocity:

    _FIELDS = ("direction", "velocity")

    def __init__(self):
        self._goal = dict.fromkeys(self._FIELDS)

    def set_goal(self, direction, velocity):
        self._goal.update(direction=direction, velocity=velocity)

    def _get(self, field):
        value = self._goal[field]
        assert value is not None, "goal %s was never set" % field
        return copy.deepcopy(value)

    def get_direction(self):
        return self._get("direction")

    def get_velocity(self):
        return self._get("velocity")

    def get_goal(self):
        return tuple(self._get(f) for f in self._FIELDS)

    __call__ = get_goal
