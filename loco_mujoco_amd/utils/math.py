"""
Angle / rotation helpers with the semantics of the reference's ``loco_mujoco/utils/math.py:5-78``
(which imports ``euler_to_mat``/``mat_to_euler`` from mushroom-rl: SciPy ``Rotation``, sequence 'xyz',
SURVEY.md Appendix C).
"""

import numpy as np
from scipy.spatial.transform import Rotation


def euler_to_mat(euler_angles, kw="xyz"):
    return Rotation.from_euler(kw, euler_angles).as_matrix()


def mat_to_euler(mat, kw="xyz"):
    return Rotation.from_matrix(mat).as_euler(kw)


def transform_angle_2pi(angle):
    """Wrap an angle (or array of angles) to [-pi, pi)."""
    return (angle + np.pi) % (2 * np.pi) - np.pi


def mat2angle_xy(mat):
    """Yaw (rotation in the x-y plane) of a rotation matrix given as 9 or 3x3 numbers."""
    return mat_to_euler(np.asarray(mat).reshape((3, 3)))[-1]


def angle2mat_xy(angle):
    """3x3 rotation matrix of a rotation by ``angle`` about z."""
    return euler_to_mat(np.array([0.0, 0.0, angle]))


def rotate_obs(state, angle, idx_rot, idx_xvel, idx_yvel):
    """
    Rotate a state about the vertical axis: add ``angle`` to the yaw entry (wrapped) and rotate the
    planar velocity (reference: ``math.py:5-30``).
    """
    state = np.array(state)
    out = state.copy()
    c, s = np.cos(angle), np.sin(angle)
    out[idx_rot] = transform_angle_2pi(state[idx_rot] + angle)
    out[idx_xvel] = c * state[idx_xvel] - s * state[idx_yvel]
    out[idx_yvel] = s * state[idx_xvel] + c * state[idx_yvel]
    return out
