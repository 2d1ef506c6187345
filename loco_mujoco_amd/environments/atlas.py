"""
Atlas humanoid environment — host-side mirror of the reference's
``loco_mujoco/environments/humanoids/atlas.py`` (+ ``base_robot_humanoid.py``) for the default configuration of
BASELINE config 4: arms disabled, back joints disabled by default (``atlas.py:275,338-364``; ``disable_back_joint=False``
adds the three-joint back chain, the only joints the shipped domain-randomisation file touches), no carried weight.
10 torque actuators (ctrl range +-0.95), 30-dim observation (14 joint positions without the two horizontal
root coordinates, 16 joint velocities), RK4 integrator, pyramidal friction cones, box/cylinder geoms that
collide with the floor only (``data/atlas/atlas.xml:27,65``).
Observation/action vectors follow the SPEC order (right leg first) although the XML declares the left leg first
 — both are gathered by name.
"""

import os
import warnings
from pathlib import Path

import numpy as np

from .. import mjcf
from ..utils.checks import check_validity_task_mode_dataset
from .base import LocoEnv, ValidTaskConf
from .base_robot_humanoid import BaseRobotHumanoid
from .observation import ObservationType

_PKG = Path(__file__).resolve().parent.parent

_ARM = ["l_arm_shz", "l_arm_shx", "l_arm_ely", "l_arm_elx", "l_arm_wry", "l_arm_wrx",
        "r_arm_shz", "r_arm_shx", "r_arm_ely", "r_arm_elx", "r_arm_wry", "r_arm_wrx"]
_BACK = ["back_bkz", "back_bky", "back_bkx"]
_PELVIS = ["pelvis_tx", "pelvis_tz", "pelvis_ty", "pelvis_tilt", "pelvis_list", "pelvis_rotation"]
_LEG = ["hip_flexion", "hip_adduction", "hip_rotation", "knee_angle", "ankle_angle"]


class Atlas(BaseRobotHumanoid):

    valid_task_confs = ValidTaskConf(tasks=["walk", "carry"], data_types=["real", "perfect"])

    def __init__(self, disable_arms=True, disable_back_joint=True, hold_weight=False, weight_mass=None,
                 xml_path=None, timestep=0.001, **kwargs):
        if not disable_arms:
            raise NotImplementedError("Atlas with free arms is not built: an Atlas arm has 7 joints (shz, shx, ely, elx, wry, wrx, wry2), the "
                                      "step kernels are compiled for chains of at most 6 links; with the back joints the arms would also "
                                      "branch behind a 3-dof chain (UnitreeH1 / UnitreeG1, one torso joint and shorter arms, run with free arms)")
        self._disable_arms, self._disable_back_joint, self._hold_weight = disable_arms, disable_back_joint, hold_weight
        self._weight_mass = weight_mass
        joints_to_remove, motors_to_remove, _ = self._get_xml_modifications()
        drop = ["q_" + j for j in joints_to_remove] + ["dq_" + j for j in joints_to_remove]
        observation_spec = [e for e in self._get_observation_specification() if e[0] not in drop]
        action_spec = [a for a in self._get_action_specification() if a not in motors_to_remove]
        weights = self._weight_list(hold_weight, weight_mass, kwargs.get("n_envs", 1))
        models = [self._load_model(xml_path, timestep, joints_to_remove, motors_to_remove,
                                   "default" if disable_back_joint else "back", w) for w in weights]
        collision_groups = [("floor", ["floor"]), ("foot_r", ["right_foot_back"]), ("front_foot_r", ["right_foot_front"]),
                            ("foot_l", ["left_foot_back"]), ("front_foot_l", ["left_foot_front"])]
        super().__init__(models[0], action_spec, observation_spec, collision_groups, timestep=timestep, **kwargs)
        self._init_weight_models(models, weights)

    @classmethod
    def _load_model(cls, xml_path, timestep, joints_to_remove, motors_to_remove, variant="default", weight=None):
        if xml_path is not None:
            handle = mjcf.MjcfHandle.from_path(xml_path)
            cls._delete_from_xml_handle(handle, joints_to_remove, motors_to_remove, [])
            if weight is not None:
                cls._add_weight(handle, weight)
            return mjcf.compile_mjcf(handle, timestep=timestep)
        name = "Atlas.%s.model.npz" % variant if weight is None else "Atlas.carry.%s.w%g.model.npz" % (variant, weight)
        if not (_PKG / "assets" / name).exists():
            raise NotImplementedError("no compiled model %s in the package (shipped: no weight, or 0.1 / 1 / 5 / 10 kg with the "
                                      "back joints disabled); pass xml_path=... to compile another one" % name)
        m = mjcf.CompiledModel.load(_PKG / "assets" / name)
        assert abs(m.timestep - timestep) < 1e-12
        return m

    @staticmethod
    def _add_weight(xml_handle, mass, color=None):
        """A box held in front of the robot, fixed to the upper torso; the arms are turned towards it
        (``atlas.py:455-482``; the colour only matters to the viewer)."""
        weight = xml_handle.add(xml_handle.find("body", "utorso"), "body", name="weight")
        xml_handle.add(weight, "geom", type="box", size="0.1 0.27 0.1", pos="0.72 0 -0.25", group="0", mass=repr(float(mass)))
        xml_handle.find("body", "r_clav").set("quat", "1.0 0.0 -0.35 0.0")
        xml_handle.find("body", "l_clav").set("quat", "0.0 -0.35 0.0 1.0")
        return xml_handle

    @staticmethod
    def _delete_from_xml_handle(xml_handle, joints_to_remove, motors_to_remove, equ_constraints):
        """Remove joints / motors / equality constraints by name (reference ``base.py:865-890``)."""
        for j in joints_to_remove:
            assert xml_handle.remove(xml_handle.find("joint", j)), j
        for mname in motors_to_remove:
            assert xml_handle.remove(xml_handle.find("motor", mname)), mname
        for e in equ_constraints:
            for tag in ("joint", "weld", "connect"):
                el = xml_handle.find(tag, e)
                if el is not None:
                    xml_handle.remove(el)
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

    # ------------------------------------------------------------------ termination / reward
    _BOUNDS = dict(height=(-0.3, 0.1), tilt=(-np.pi / 4.5, np.pi / 12), list=(-np.pi / 12, np.pi / 8),
                   rotation=(-np.pi / 10, np.pi / 10))

    def _has_fallen(self, obs, return_err_msg=False):
        """Pelvis height / tilt / list / rotation outside their bands (``atlas.py:366-418``)."""
        e = self._get_from_obs(obs, ["q_pelvis_tilt", "q_pelvis_list", "q_pelvis_rotation"])
        vals = dict(height=obs[0], tilt=e[0], list=e[1], rotation=e[2])
        bad = [k for k in ("height", "tilt", "list", "rotation") if vals[k] < self._BOUNDS[k][0] or vals[k] > self._BOUNDS[k][1]]
        if not return_err_msg:
            return bool(bad)
        names = dict(height="pelvis_y_condition", tilt="pelvis_tilt_condition", list="pelvis_list_condition",
                     rotation="pelvis_rotation_condition")
        return bool(bad), (names[bad[0]] + " violated.\n") if bad else ""

    def _termination_spec(self):
        i = self.get_obs_idx
        b = self._BOUNDS
        return [(0, *b["height"]), (i("q_pelvis_tilt")[0], *b["tilt"]), (i("q_pelvis_list")[0], *b["list"]),
                (i("q_pelvis_rotation")[0], *b["rotation"])]

    # ------------------------------------------------------------------ task factory
    @staticmethod
    def generate(task="walk", dataset_type="real", debug=False, clip_trajectory_to_joint_ranges=False, **kwargs):
        """``LocoEnv.make("Atlas.walk.real")`` (``atlas.py:420-453`` -> ``base_robot_humanoid.py:145-260``)."""
        check_validity_task_mode_dataset(Atlas.__name__, task, None, dataset_type, *Atlas.valid_task_confs.get_all())
        path = "datasets/humanoids/real/02-constspeed_ATLAS.npz"
        if dataset_type == "perfect":
            # recorded with these settings (``atlas.py:438-451``)
            assert kwargs.get("use_foot_forces", False) is False and kwargs.get("disable_arms", True) is True
            assert kwargs.get("disable_back_joint", False) is False and kwargs.get("hold_weight", False) is False
            path = {"walk": "datasets/humanoids/perfect/atlas_walk/perfect_expert_dataset_det.npz",
                    "carry": "datasets/humanoids/perfect/atlas_carry/Atlas_carry_stochastic_dataset.npz"}[task]
        return BaseRobotHumanoid.generate(Atlas, path, task, dataset_type,
                                          debug=debug, clip_trajectory_to_joint_ranges=clip_trajectory_to_joint_ranges, **kwargs)

    # ------------------------------------------------------------------ specs
    @staticmethod
    def _get_observation_specification():
        joints = _PELVIS + ["back_bkz", "back_bkx", "back_bky"] + _ARM + [j + "_r" for j in _LEG] + [j + "_l" for j in _LEG]
        return ([("q_" + j, j, ObservationType.JOINT_POS) for j in joints]
                + [("dq_" + j, j, ObservationType.JOINT_VEL) for j in joints])

    @staticmethod
    def _get_action_specification():
        return ([j + "_actuator" for j in ["back_bkz", "back_bky", "back_bkx"] + _ARM]
                + [j + "_r_actuator" for j in _LEG] + [j + "_l_actuator" for j in _LEG])
