"""
Unitree H1 humanoid — host-side mirror of the reference's ``loco_mujoco/environments/humanoids/unitreeH1.py``
(+ ``base_robot_humanoid.py``) for its default configuration: arms disabled and re-oriented (``unitreeH1.py:265-275,447-468``),
the back joint kept, optionally a carried weight (``:427-445``).
17 dofs (6 pelvis + back + 2 x 5 leg), 11 torque actuators, 32-dim observation, Euler integrator (the XML default), pyramidal
cones. Every link is a collision MESH: the device collides their convex hulls with the floor (one contact at the hull's
support vertex, the rule the golden rollouts of this robot pinned); hull against hull (thigh vs hip-yaw link, a
third of the golden rows) is the engine's libccd path and has no collider here — such pairs are counted when their bounding
capsules come within reach (``lm_get_flags`` bit 2 on the oracle's side: ``unhandled_pairs``).
"""

from pathlib import Path

import numpy as np

from .. import mjcf
from ..utils.checks import check_validity_task_mode_dataset
from .atlas import Atlas
from .base import ValidTaskConf
from .base_robot_humanoid import BaseRobotHumanoid
from .observation import ObservationType

_PKG = Path(__file__).resolve().parent.parent

_ARM = ["l_arm_shy", "l_arm_shx", "l_arm_shz", "left_elbow", "r_arm_shy", "r_arm_shx", "r_arm_shz", "right_elbow"]
_BACK = ["back_bkz"]
_PELVIS = ["pelvis_tx", "pelvis_tz", "pelvis_ty", "pelvis_tilt", "pelvis_list", "pelvis_rotation"]
_LEG = ["hip_flexion", "hip_adduction", "hip_rotation", "knee_angle", "ankle_angle"]


class UnitreeH1(BaseRobotHumanoid):

    valid_task_confs = ValidTaskConf(tasks=["walk", "run", "carry"], data_types=["real", "perfect"],
                                     non_combinable=[("carry", None, "perfect")])

    def __init__(self, disable_arms=True, disable_back_joint=False, hold_weight=False, weight_mass=None,
                 xml_path=None, timestep=0.001, **kwargs):
        if hold_weight:
            assert disable_arms is True, "If you want Unitree H1 to carry a weight, please disable the arms. They will be kept fixed."
        if not disable_arms:
            # the reference takes the file as it is when the arms are free and no weight is held — `disable_back_joint` has no
            # effect then (unitreeH1.py:265-296: the modifications sit inside `if disable_arms or hold_weight`). Same here.
            disable_back_joint = False
        self._disable_arms, self._disable_back_joint, self._hold_weight = disable_arms, disable_back_joint, hold_weight
        self._weight_mass = weight_mass
        joints_to_remove, motors_to_remove, _ = self._get_xml_modifications()
        drop = ["q_" + j for j in joints_to_remove] + ["dq_" + j for j in joints_to_remove]
        observation_spec = [e for e in self._get_observation_specification() if e[0] not in drop]
        action_spec = [a for a in self._get_action_specification() if a not in motors_to_remove]
        weights = self._weight_list(hold_weight, weight_mass, kwargs.get("n_envs", 1))
        variant = "arms" if not disable_arms else ("noback" if disable_back_joint else "default")
        models = []
        for w in weights:
            if xml_path is not None:
                models.append(self._compile(mjcf.MjcfHandle.from_path(xml_path), timestep, joints_to_remove, motors_to_remove, w, reorient=disable_arms))
                continue
            name = "UnitreeH1.%s.model.npz" % variant if w is None else "UnitreeH1.carry.%s.w%g.model.npz" % (variant, w)
            if not (_PKG / "assets" / name).exists():
                raise NotImplementedError("no compiled model %s in the package; pass xml_path=... to compile another one" % name)
            models.append(mjcf.CompiledModel.load(_PKG / "assets" / name))
            assert abs(models[-1].timestep - timestep) < 1e-12
        collision_groups = [("floor", ["floor"]), ("foot_r", ["right_foot"]), ("foot_l", ["left_foot"])]
        super().__init__(models[0], action_spec, observation_spec, collision_groups, timestep=timestep, **kwargs)
        self._init_weight_models(models, weights)

    @classmethod
    def _compile(cls, handle, timestep, joints_to_remove, motors_to_remove, weight=None, reorient=True):
        Atlas._delete_from_xml_handle(handle, joints_to_remove, motors_to_remove, [])
        if weight is not None:
            cls._add_weight(handle, weight)
        elif reorient:
            cls._reorient_arms(handle)
        return mjcf.compile_mjcf(handle, timestep=timestep, drop_mesh_geoms=True)       # meshes kept with their convex hulls

    @staticmethod
    def _add_weight(xml_handle, mass, color=None):
        """A box held in front of the robot, fixed to the torso link (``unitreeH1.py:427-445``)."""
        weight = xml_handle.add(xml_handle.find("body", "torso_link"), "body", name="weight")
        xml_handle.add(weight, "geom", type="box", size="0.1 0.18 0.1", pos="0.35 0 0.1", group="0", mass=repr(float(mass)))
        return xml_handle

    @staticmethod
    def _reorient_arms(xml_handle):
        """Elbows turned so that the fixed arms clear the hips (``unitreeH1.py:447-468``)."""
        for body, quat in (("left_shoulder_pitch_link", "1.0 0.25 0.1 0.0"), ("right_elbow_link", "1.0 0.0 0.25 0.0"),
                           ("right_shoulder_pitch_link", "1.0 -0.25 0.1 0.0"), ("left_elbow_link", "1.0 0.0 0.25 0.0")):
            xml_handle.find("body", body).set("quat", quat)
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
        return [(None, -0.3, 0.1, "pelvis_y_condition"), ("q_pelvis_tilt", -np.pi / 4.5, np.pi / 12, "pelvis_tilt_condition"),
                ("q_pelvis_list", -np.pi / 12, np.pi / 8, "pelvis_list_condition"),
                ("q_pelvis_rotation", -np.pi / 8, np.pi / 8, "pelvis_rotation_condition")]

    def _has_fallen(self, obs, return_err_msg=False):
        """Pelvis height / orientation outside their bands (``unitreeH1.py:341-377``)."""
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
        """``unitreeH1.py:297-308``."""
        return ["foot_r", "foot_l"]

    # ------------------------------------------------------------------ task factory
    @staticmethod
    def generate(task="walk", dataset_type="real", debug=False, **kwargs):
        """``LocoEnv.make("UnitreeH1.walk.real")`` (``unitreeH1.py:379-425`` -> ``base_robot_humanoid.py:145-260``)."""
        check_validity_task_mode_dataset(UnitreeH1.__name__, task, None, dataset_type, *UnitreeH1.valid_task_confs.get_all())
        if dataset_type == "real":
            path = "datasets/humanoids/real/05-run_UnitreeH1.npz" if task == "run" else "datasets/humanoids/real/02-constspeed_UnitreeH1.npz"
        else:
            assert kwargs.get("use_foot_forces", False) is False and kwargs.get("disable_arms", True) is True
            assert kwargs.get("disable_back_joint", False) is False and kwargs.get("hold_weight", False) is False
            path = "datasets/humanoids/perfect/unitreeh1_%s/perfect_expert_dataset_det.npz" % ("run" if task == "run" else "walk")
        return BaseRobotHumanoid.generate(UnitreeH1, path, task, dataset_type, debug=debug, clip_trajectory_to_joint_ranges=True, **kwargs)

    # ------------------------------------------------------------------ specs
    @staticmethod
    def _get_observation_specification():
        """``unitreeH1.py:470-546``: pelvis, back, left arm, right arm, right leg, left leg."""
        joints = _PELVIS + _BACK + _ARM + [j + "_r" for j in _LEG] + [j + "_l" for j in _LEG]
        return ([("q_" + j, j, ObservationType.JOINT_POS) for j in joints]
                + [("dq_" + j, j, ObservationType.JOINT_VEL) for j in joints])

    @staticmethod
    def _get_action_specification():
        """``unitreeH1.py:548-555``."""
        return [j + "_actuator" for j in _BACK + _ARM + [j + "_r" for j in _LEG] + [j + "_l" for j in _LEG]]


_G1_PELVIS = ["pelvis_tx", "pelvis_tz", "pelvis_ty", "pelvis_tilt", "pelvis_list", "pelvis_rotation"]
_G1_LEG = ["hip_pitch_joint", "hip_roll_joint", "hip_yaw_joint", "knee_joint", "ankle_pitch_joint", "ankle_roll_joint"]
_G1_ARM = ["shoulder_pitch_joint", "shoulder_roll_joint", "shoulder_yaw_joint", "elbow_pitch_joint", "elbow_roll_joint"]
_G1_BACK = ["torso_joint"]


class UnitreeG1(BaseRobotHumanoid):
    """
    Unitree G1 — host-side mirror of ``loco_mujoco/environments/humanoids/unitreeG1.py``. The reference's default keeps the
    torso joint and the arms: 29 dofs (6 pelvis + 2 x 6 leg + torso + 2 x 5 arm), 23 torque motors, 56-dim observation; joints
    with damping 0.5, armature 0.01, frictionloss 0.1; Euler; pyramidal cones; four 1 mm spheres per foot, a cylinder per shin,
    collision meshes everywhere else (convex hulls against the floor, hull against hull counted, like UnitreeH1).

    Device: the kernels simulate a root body with up to four serial chains. The arms hang off the torso link — a branch: the two
    arm chains SHARE that link as their first (owner lane + massless copy, the two copies of the torso dof tied together in every
    solve; ``lowering.py`` ``shared_first``, ``csrc/lm_core.h`` ``tie_shared_dof``), kernel family ``<6 links, 8 slots, Euler,
    pyramids>``. The golden rollouts of this configuration are reproduced by the oracle (1e-14) and on the device (2.6e-6 / 2e-4).
    ``disable_back_joint`` / ``disable_arms`` give plain root + chains models.
    """

    valid_task_confs = ValidTaskConf(tasks=["walk", "run"], data_types=["real"])

    def __init__(self, disable_arms=False, disable_back_joint=False, xml_path=None, timestep=0.001, **kwargs):
        self._disable_arms, self._disable_back_joint, self._hold_weight = disable_arms, disable_back_joint, False
        joints_to_remove, motors_to_remove, _ = self._get_xml_modifications()
        drop = ["q_" + j for j in joints_to_remove] + ["dq_" + j for j in joints_to_remove]
        observation_spec = [e for e in self._get_observation_specification() if e[0] not in drop]
        action_spec = [a for a in self._get_action_specification() if a not in motors_to_remove]
        variant = self._variant_name(disable_arms, disable_back_joint)
        if xml_path is not None:
            model = self._compile(mjcf.MjcfHandle.from_path(xml_path), timestep, joints_to_remove, motors_to_remove, disable_arms)
        else:
            name = "UnitreeG1.%s.model.npz" % variant
            if not (_PKG / "assets" / name).exists():
                raise NotImplementedError("no compiled model %s in the package; pass xml_path=... to compile another one" % name)
            model = mjcf.CompiledModel.load(_PKG / "assets" / name)
            assert abs(model.timestep - timestep) < 1e-12
        collision_groups = [("floor", ["floor"])] + [("%s_foot_%d" % (s, i), ["%s_foot_%d_col" % (s, i)])
                                                     for s in ("right", "left") for i in (1, 2, 3, 4)]
        super().__init__(model, action_spec, observation_spec, collision_groups, timestep=timestep, **kwargs)
        self._init_weight_models([model], [None])

    @staticmethod
    def _variant_name(disable_arms, disable_back_joint):
        return {(False, False): "default", (False, True): "noback", (True, False): "noarms", (True, True): "legs"}[(bool(disable_arms), bool(disable_back_joint))]

    @classmethod
    def _compile(cls, handle, timestep, joints_to_remove, motors_to_remove, reorient):
        Atlas._delete_from_xml_handle(handle, joints_to_remove, motors_to_remove, [])
        if reorient:
            cls._reorient_arms(handle)
        return mjcf.compile_mjcf(handle, timestep=timestep, drop_mesh_geoms=True)       # meshes kept with their convex hulls

    @staticmethod
    def _reorient_arms(xml_handle):
        """Elbows turned so that the fixed arms clear the hips (``unitreeG1.py:425-446``)."""
        for body, quat in (("left_shoulder_pitch_link", "1.0 0.25 0.1 0.0"), ("right_elbow_pitch_link", "1.0 0.0 0.25 0.0"),
                           ("right_shoulder_pitch_link", "1.0 -0.25 0.1 0.0"), ("left_elbow_pitch_link", "1.0 0.0 0.25 0.0")):
            xml_handle.find("body", body).set("quat", quat)
        return xml_handle

    def _get_xml_modifications(self):
        """``unitreeG1.py:322-353``: the motors carry their joints' names."""
        joints = []
        if self._disable_arms:
            joints += [s + "_" + j for s in ("right", "left") for j in _G1_ARM]
        if self._disable_back_joint:
            joints += _G1_BACK
        return joints, list(joints), []

    # ------------------------------------------------------------------ termination (same bands as UnitreeH1)
    _bounds = UnitreeH1._bounds
    _has_fallen = UnitreeH1._has_fallen
    _termination_spec = UnitreeH1._termination_spec

    def _get_grf_size(self):
        return 24

    def _grf_group_names(self):
        """Four force points per foot, right foot first (``unitreeG1.py:295-317``): 8 groups x 3 = 24 entries."""
        return ["%s_foot_%d" % (s, i) for s in ("right", "left") for i in (1, 2, 3, 4)]

    # ------------------------------------------------------------------ task factory
    @staticmethod
    def generate(task="walk", dataset_type="real", debug=False, **kwargs):
        """``LocoEnv.make("UnitreeG1.walk.real")`` (``unitreeG1.py:395-423``)."""
        check_validity_task_mode_dataset(UnitreeG1.__name__, task, None, dataset_type, *UnitreeG1.valid_task_confs.get_all())
        path = "datasets/humanoids/real/05-run_UnitreeG1.npz" if task == "run" else "datasets/humanoids/real/02-constspeed_UnitreeG1.npz"
        return BaseRobotHumanoid.generate(UnitreeG1, path, task, dataset_type, debug=debug, clip_trajectory_to_joint_ranges=True, **kwargs)

    # ------------------------------------------------------------------ specs (XML order: ``unitreeG1.py:448-481``)
    @staticmethod
    def _joint_names():
        return (_G1_PELVIS + ["left_" + j for j in _G1_LEG] + ["right_" + j for j in _G1_LEG] + _G1_BACK
                + ["left_" + j for j in _G1_ARM] + ["right_" + j for j in _G1_ARM])

    @staticmethod
    def _get_observation_specification():
        joints = UnitreeG1._joint_names()
        return ([("q_" + j, j, ObservationType.JOINT_POS) for j in joints]
                + [("dq_" + j, j, ObservationType.JOINT_VEL) for j in joints])

    @staticmethod
    def _get_action_specification():
        return UnitreeG1._joint_names()[6:]
