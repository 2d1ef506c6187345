"""
Talos humanoid environment — host-side mirror of the reference's ``loco_mujoco/environments/humanoids/talos.py``
(+ ``base_robot_humanoid.py``) for its default configuration: arms disabled and re-oriented (``talos.py:296-321,503-520``),
the two back joints kept (``:350-352``), no carried weight.
18 dofs (6 pelvis + 2 back + 2 x 5 leg), 12 torque actuators (ctrl range +-0.95), 34-dim observation, Euler integrator
with implicit damping, pyramidal cones, one box foot per leg; the other robot geoms collide with the floor only
(``data/talos/talos.xml:14``; cylinders have no device collider and are counted as proximity).
"""

import os
import warnings
from pathlib import Path

import numpy as np

from .. import mjcf
from ..utils.checks import check_validity_task_mode_dataset
from .atlas import Atlas
from .base import LocoEnv, ValidTaskConf
from .base_robot_humanoid import BaseRobotHumanoid
from .observation import ObservationType

_PKG = Path(__file__).resolve().parent.parent

_ARM = ["l_arm_shz", "l_arm_shx", "l_arm_ely", "l_arm_elx", "l_arm_wry", "l_arm_wrx",
        "r_arm_shz", "r_arm_shx", "r_arm_ely", "r_arm_elx", "r_arm_wry", "r_arm_wrx"]
_BACK = ["back_bkz", "back_bky"]
_PELVIS = ["pelvis_tx", "pelvis_tz", "pelvis_ty", "pelvis_tilt", "pelvis_list", "pelvis_rotation"]
_LEG = ["hip_flexion", "hip_adduction", "hip_rotation", "knee_angle", "ankle_angle"]


class Talos(BaseRobotHumanoid):

    valid_task_confs = ValidTaskConf(tasks=["walk", "carry"], data_types=["real", "perfect"],
                                     non_combinable=[("carry", None, "perfect")])

    def __init__(self, disable_arms=True, disable_back_joint=False, hold_weight=False, weight_mass=None,
                 xml_path=None, timestep=0.001, **kwargs):
        if not disable_arms:
            raise NotImplementedError("Talos with free arms is not built (SURVEY.md §8f rank 3): the arms would branch off "
                                      "the back chain")
        self._disable_arms, self._disable_back_joint, self._hold_weight = disable_arms, disable_back_joint, hold_weight
        self._weight_mass = weight_mass
        joints_to_remove, motors_to_remove, _ = self._get_xml_modifications()
        drop = ["q_" + j for j in joints_to_remove] + ["dq_" + j for j in joints_to_remove]
        observation_spec = [e for e in self._get_observation_specification() if e[0] not in drop]
        action_spec = [a for a in self._get_action_specification() if a not in motors_to_remove]
        weights = self._weight_list(hold_weight, weight_mass, kwargs.get("n_envs", 1))
        variant = "noback" if disable_back_joint else "default"
        models = []
        for w in weights:
            if xml_path is not None:
                models.append(self._compile(mjcf.MjcfHandle.from_path(xml_path), timestep, joints_to_remove, motors_to_remove, w))
                continue
            name = "Talos.%s.model.npz" % variant if w is None else "Talos.carry.%s.w%g.model.npz" % (variant, w)
            if not (_PKG / "assets" / name).exists():
                raise NotImplementedError("no compiled model %s in the package (shipped: no weight, or 0.1 / 1 / 5 / 10 kg with "
                                          "the back joints); pass xml_path=... to compile another one" % name)
            models.append(mjcf.CompiledModel.load(_PKG / "assets" / name))
            assert abs(models[-1].timestep - timestep) < 1e-12
        collision_groups = [("floor", ["floor"]), ("foot_r", ["right_foot"]), ("foot_l", ["left_foot"])]
        super().__init__(models[0], action_spec, observation_spec, collision_groups, timestep=timestep, **kwargs)
        self._init_weight_models(models, weights)

    @classmethod
    def _compile(cls, handle, timestep, joints_to_remove, motors_to_remove, weight=None):
        Atlas._delete_from_xml_handle(handle, joints_to_remove, motors_to_remove, [])
        if weight is not None:
            cls._add_weight(handle, weight)
        else:
            cls._reorient_arms(handle)
        # the collision meshes are kept with their convex hulls (plane vs hull on the device; hull pairs are counted by the oracle)
        return mjcf.compile_mjcf(handle, timestep=timestep, drop_mesh_geoms=True)

    @staticmethod
    def _add_weight(xml_handle, mass, color=None):
        """A box held in front of the robot, fixed to the upper torso, elbows and wrists turned towards it
        (``talos.py:468-500``; the colour only matters to the viewer)."""
        weight = xml_handle.add(xml_handle.find("body", "torso_2_link"), "body", name="weight")
        xml_handle.add(weight, "geom", type="box", size="0.1 0.25 0.1", pos="0.45 0 -0.20", group="0", mass=repr(float(mass)))
        for body in ("arm_right_4_link", "arm_left_4_link"):
            xml_handle.find("body", body).set("quat", "1.0 0.0 -0.65 0.0")
        for body in ("arm_right_6_link", "arm_left_6_link"):
            xml_handle.find("body", body).set("quat", "1.0 0.0 -0.0 1.0")
        return xml_handle

    @staticmethod
    def _reorient_arms(xml_handle):
        """Elbows turned so that the fixed arms clear the hips (``talos.py:503-520``)."""
        for body in ("arm_right_4_link", "arm_left_4_link"):
            xml_handle.find("body", body).set("quat", "1.0 0.0 -0.25 0.0")
        return xml_handle

    def _get_xml_modifications(self):
        joints, motors = [], []
        if self._disable_arms:
            joints += _ARM
            motors += [j + "_actuator" for j in _ARM]
        if self._disable_back_joint:
            joints += _BACK
            motors += [j + "_actuator" for j in _BACK]
        return joints, motors, []

    # ------------------------------------------------------------------ termination
    def _bounds(self):
        b = [(None, -0.3, 0.1, "pelvis_y_condition"), ("q_pelvis_tilt", -np.pi / 4.5, np.pi / 12, "pelvis_tilt_condition"),
             ("q_pelvis_list", -np.pi / 12, np.pi / 8, "pelvis_list_condition"),
             ("q_pelvis_rotation", -np.pi / 10, np.pi / 10, "pelvis_rotation_condition")]
        if not self._disable_back_joint:
            b += [("q_back_bky", -np.pi / 4, np.pi / 10, "back_extension_condition"),
                  ("q_back_bkz", -np.pi / 10, np.pi / 10, "back_rotation_condition")]
        return b

    def _has_fallen(self, obs, return_err_msg=False):
        """Pelvis height / orientation and back angles outside their bands (``talos.py:356-405``; the message names the
        first violated condition only, like the reference's elif chain)."""
        bad = [name for key, lo, hi, name in self._bounds()
               if not (lo <= (obs[0] if key is None else self._get_from_obs(obs, [key])[0]) <= hi)]
        if not return_err_msg:
            return bool(bad)
        return bool(bad), (bad[0] + " violated.\n") if bad else ""

    def _termination_spec(self):
        return [(0 if key is None else self.get_obs_idx(key)[0], lo, hi) for key, lo, hi, _ in self._bounds()]

    def _get_grf_size(self):
        return 6

    def _grf_group_names(self):
        """``talos.py:407-417``."""
        return ["foot_r", "foot_l"]

    # ------------------------------------------------------------------ task factory
    @staticmethod
    def generate(task="walk", dataset_type="real", debug=False, **kwargs):
        """``LocoEnv.make("Talos.walk.real")`` (``talos.py:429-466`` -> ``base_robot_humanoid.py:145-260``)."""
        if "disable_arms" in kwargs:
            assert kwargs["disable_arms"] is True, "Activating the arms in the Talos environment is currently not supported."
        check_validity_task_mode_dataset(Talos.__name__, task, None, dataset_type, *Talos.valid_task_confs.get_all())
        clip = kwargs.pop("clip_trajectory_to_joint_ranges", True)
        path = "datasets/humanoids/real/02-constspeed_TALOS.npz"
        if dataset_type == "perfect":
            assert kwargs.get("use_foot_forces", False) is False and kwargs.get("disable_back_joint", False) is False
            assert kwargs.get("hold_weight", False) is False
            path = "datasets/humanoids/perfect/talos_walk/perfect_expert_dataset_det.npz"
        return BaseRobotHumanoid.generate(Talos, path, task, dataset_type,
                                          debug=debug, clip_trajectory_to_joint_ranges=clip, **kwargs)

    # ------------------------------------------------------------------ specs
    @staticmethod
    def _get_observation_specification():
        """``talos.py:522-598``: pelvis, back, left arm, right arm, right leg, left leg."""
        joints = _PELVIS + _BACK + _ARM + [j + "_r" for j in _LEG] + [j + "_l" for j in _LEG]
        return ([("q_" + j, j, ObservationType.JOINT_POS) for j in joints]
                + [("dq_" + j, j, ObservationType.JOINT_VEL) for j in joints])

    @staticmethod
    def _get_action_specification():
        """``talos.py:600-619``."""
        return [j + "_actuator" for j in _BACK + _ARM + [j + "_r" for j in _LEG] + [j + "_l" for j in _LEG]]
