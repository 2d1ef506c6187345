"""
Regenerates the bundled assets from a loco-mujoco checkout (default /root/reference). Run in the build
container only; nothing at run time reads the checkout.

  python tools/build_assets.py [--ref /path/to/loco-mujoco]

Writes
  loco_mujoco_amd/assets/{UnitreeA1.torque,Atlas.default,HumanoidTorque.default,HumanoidMuscle.default}.model.npz   compiled models (after the env's XML surgery)
  loco_mujoco_amd/datasets/quadrupeds/real/mini_datasets/walk_straight.npz   re-encoded mini dataset
  tests/golden/reference_rollouts.npz                     the reference's golden rollouts for our tasks
"""

import argparse
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from loco_mujoco_amd import mjcf                      # noqa: E402
from loco_mujoco_amd.environments.unitree_a1 import UnitreeA1   # noqa: E402
from loco_mujoco_amd.environments.atlas import Atlas, _ARM, _BACK   # noqa: E402
from loco_mujoco_amd.environments.talos import Talos   # noqa: E402
from loco_mujoco_amd.environments.unitree_h1 import UnitreeG1, UnitreeH1   # noqa: E402
from loco_mujoco_amd.environments.humanoids import (HumanoidMuscle, HumanoidMuscle4Ages, HumanoidTorque,   # noqa: E402
                                                    HumanoidTorque4Ages)

GOLDEN_TASKS = ["UnitreeA1.simple.real", "UnitreeA1.hard.real", "HumanoidTorque.run.real", "HumanoidTorque.walk.real",
                "Atlas.walk.real", "Atlas.carry.real", "Talos.walk.real", "Talos.carry.real", "HumanoidMuscle.run.real", "HumanoidMuscle.walk.real",
                "UnitreeH1.walk.real", "UnitreeH1.run.real", "UnitreeH1.carry.real", "UnitreeG1.walk.real", "UnitreeG1.run.real"] + [
                "Humanoid%s4Ages.%s.%s.real" % (a, t, k) for a in ("Torque", "Muscle") for t in ("run", "walk") for k in (1, 2, 3, 4, "all")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ref = Path(ap.parse_args().ref)
    pkg = ref / "loco_mujoco"

    # --- models
    (ROOT / "loco_mujoco_amd" / "assets").mkdir(exist_ok=True)
    h = mjcf.MjcfHandle.from_path(pkg / "environments" / "data" / "quadrupeds" / "unitree_a1_torque.xml")
    m = mjcf.compile_mjcf(UnitreeA1._add_dir_vector_to_xml_handle(h), timestep=0.001)
    m.save(ROOT / "loco_mujoco_amd" / "assets" / "UnitreeA1.torque.model.npz")
    print("UnitreeA1: nbody %d nv %d ngeom %d nu %d" % (m.nbody, m.nv, m.ngeom, m.nu))
    h = mjcf.MjcfHandle.from_path(pkg / "environments" / "data" / "quadrupeds" / "unitree_a1_position.xml")
    m = mjcf.compile_mjcf(UnitreeA1._add_dir_vector_to_xml_handle(h), timestep=0.001)
    m.save(ROOT / "loco_mujoco_amd" / "assets" / "UnitreeA1.position.model.npz")
    print("UnitreeA1 (position servos): kp %s force range %s" % (m.act_gainprm[0, 0], m.act_forcerange[0]))

    h = mjcf.MjcfHandle.from_path(pkg / "environments" / "data" / "atlas" / "atlas.xml")
    Atlas._delete_from_xml_handle(h, _ARM + _BACK, [j + "_actuator" for j in _ARM + _BACK], [])
    m = mjcf.compile_mjcf(h, timestep=0.001)
    m.save(ROOT / "loco_mujoco_amd" / "assets" / "Atlas.default.model.npz")
    print("Atlas: nbody %d nv %d ngeom %d nu %d" % (m.nbody, m.nv, m.ngeom, m.nu))
    h = mjcf.MjcfHandle.from_path(pkg / "environments" / "data" / "atlas" / "atlas.xml")
    Atlas._delete_from_xml_handle(h, _ARM, [j + "_actuator" for j in _ARM], [])
    m = mjcf.compile_mjcf(h, timestep=0.001)
    m.save(ROOT / "loco_mujoco_amd" / "assets" / "Atlas.back.model.npz")
    print("Atlas (back joints): nbody %d nv %d ngeom %d nu %d" % (m.nbody, m.nv, m.ngeom, m.nu))

    for w in Atlas._valid_weights:                       # Atlas.carry: a box of 0.1 / 1 / 5 / 10 kg fixed to the torso
        h = mjcf.MjcfHandle.from_path(pkg / "environments" / "data" / "atlas" / "atlas.xml")
        Atlas._delete_from_xml_handle(h, _ARM + _BACK, [j + "_actuator" for j in _ARM + _BACK], [])
        m = mjcf.compile_mjcf(Atlas._add_weight(h, w), timestep=0.001)
        m.save(ROOT / "loco_mujoco_amd" / "assets" / ("Atlas.carry.default.w%g.model.npz" % w))
    print("Atlas.carry: total mass %.2f kg with the 10 kg box" % m.body_mass.sum())

    for w in Talos._valid_weights:
        t = Talos.__new__(Talos)
        t._disable_arms, t._disable_back_joint = True, False
        j, mo, _ = t._get_xml_modifications()
        m = Talos._compile(mjcf.MjcfHandle.from_path(pkg / "environments" / "data" / "talos" / "talos.xml"), 0.001, j, mo, w)
        m.save(ROOT / "loco_mujoco_amd" / "assets" / ("Talos.carry.default.w%g.model.npz" % w))
    print("Talos.carry: total mass %.2f kg with the 10 kg box" % m.body_mass.sum())

    for variant, no_back in (("default", False), ("noback", True)):
        t = Talos.__new__(Talos)
        t._disable_arms, t._disable_back_joint = True, no_back
        j, mo, _ = t._get_xml_modifications()
        m = Talos._compile(mjcf.MjcfHandle.from_path(pkg / "environments" / "data" / "talos" / "talos.xml"), 0.001, j, mo)
        m.save(ROOT / "loco_mujoco_amd" / "assets" / ("Talos.%s.model.npz" % variant))
        print("Talos (%s): nbody %d nv %d ngeom %d nu %d integrator %d cone %d" % (variant, m.nbody, m.nv, m.ngeom, m.nu, m.integrator, m.cone))

    # UnitreeH1: collision meshes kept with their convex hulls (plane-mesh collider)
    for variant, no_back, no_arms in (("default", False, True), ("noback", True, True), ("arms", False, False)):
        t = UnitreeH1.__new__(UnitreeH1)
        t._disable_arms, t._disable_back_joint = no_arms, no_back
        j, mo, _ = t._get_xml_modifications()
        m = UnitreeH1._compile(mjcf.MjcfHandle.from_path(pkg / "environments" / "data" / "unitree_h1" / "h1.xml"), 0.001, j, mo, reorient=no_arms)
        m.save(ROOT / "loco_mujoco_amd" / "assets" / ("UnitreeH1.%s.model.npz" % variant))
        print("UnitreeH1 (%s): nbody %d nv %d ngeom %d nu %d integrator %d cone %d hull vertices %d" % (variant, m.nbody, m.nv, m.ngeom, m.nu, m.integrator, m.cone, len(m.hull_vert)))
    # UnitreeG1: the reference's default (torso joint + arms) for the host side and the oracle, the reduced ones for the device
    for no_arms, no_back in ((False, False), (False, True), (True, False), (True, True)):
        t = UnitreeG1.__new__(UnitreeG1)
        t._disable_arms, t._disable_back_joint = no_arms, no_back
        j, mo, _ = t._get_xml_modifications()
        m = UnitreeG1._compile(mjcf.MjcfHandle.from_path(pkg / "environments" / "data" / "unitree_g1" / "g1.xml"), 0.001, j, mo, no_arms)
        variant = UnitreeG1._variant_name(no_arms, no_back)
        m.save(ROOT / "loco_mujoco_amd" / "assets" / ("UnitreeG1.%s.model.npz" % variant))
        print("UnitreeG1 (%s): nbody %d nv %d ngeom %d nu %d integrator %d cone %d hull vertices %d" % (variant, m.nbody, m.nv, m.ngeom, m.nu, m.integrator, m.cone, len(m.hull_vert)))
    for w in UnitreeH1._valid_weights:
        t = UnitreeH1.__new__(UnitreeH1)
        t._disable_arms, t._disable_back_joint = True, False
        j, mo, _ = t._get_xml_modifications()
        m = UnitreeH1._compile(mjcf.MjcfHandle.from_path(pkg / "environments" / "data" / "unitree_h1" / "h1.xml"), 0.001, j, mo, w)
        m.save(ROOT / "loco_mujoco_amd" / "assets" / ("UnitreeH1.carry.default.w%g.model.npz" % w))

    h = mjcf.MjcfHandle.from_path(pkg / "environments" / "data" / "humanoid" / "humanoid_torque.xml")
    ht = HumanoidTorque.__new__(HumanoidTorque)
    ht._use_muscles, ht._use_box_feet, ht._disable_arms = False, True, True
    m = ht._compile(h, 0.001, *ht._get_xml_modifications()[:3])
    m.save(ROOT / "loco_mujoco_amd" / "assets" / "HumanoidTorque.default.model.npz")
    print("HumanoidTorque: nbody %d nv %d ngeom %d nu %d (mesh geoms kept as proximity spheres: %d)"
          % (m.nbody, m.nv, m.ngeom, m.nu, m.n_dropped_mesh_geoms))

    h = mjcf.MjcfHandle.from_path(pkg / "environments" / "data" / "humanoid" / "humanoid_muscle.xml")
    hm = HumanoidMuscle.__new__(HumanoidMuscle)
    hm._use_muscles, hm._use_box_feet, hm._disable_arms = True, True, True
    m = hm._compile(h, 0.001, *hm._get_xml_modifications()[:3])
    m.save(ROOT / "loco_mujoco_amd" / "assets" / "HumanoidMuscle.default.model.npz")
    print("HumanoidMuscle: nbody %d nv %d ngeom %d nu %d na %d tendons %d path sites %d"
          % (m.nbody, m.nv, m.ngeom, m.nu, m.na, m.ntendon, len(m.wrap_site)))

    # --- the humanoid in four sizes (base_humanoid_4_ages.py): one compiled model per size and actuation
    for cls, xml, mus in ((HumanoidTorque4Ages, "humanoid_torque.xml", False), (HumanoidMuscle4Ages, "humanoid_muscle.xml", True)):
        for scale in (0.4, 0.6, 0.8, 1.0):
            h = mjcf.MjcfHandle.from_path(pkg / "environments" / "data" / "humanoid" / xml)
            e = cls.__new__(cls)
            e._use_muscles, e._use_box_feet, e._disable_arms, e._model_scale = mus, True, True, scale
            m = e._compile(h, 0.001, *e._get_xml_modifications()[:3])
            m.save(ROOT / "loco_mujoco_amd" / "assets" / e._asset_name())
            print("%s: total mass %.2f kg" % (e._asset_name(), m.body_mass.sum()))

    # --- UnitreeH1 fixture for the oracle's plane-vs-convex-mesh contact (the robot itself is not built: its thigh and
    #     hip-yaw hulls collide in a third of the golden rows, which needs the engine's convex-convex collider)
    from scipy.spatial import ConvexHull
    import struct
    arm = ["l_arm_shy", "l_arm_shx", "l_arm_shz", "left_elbow", "r_arm_shy", "r_arm_shx", "r_arm_shz", "right_elbow"]
    h1dir = pkg / "environments" / "data" / "unitree_h1"
    h = mjcf.MjcfHandle.from_path(h1dir / "h1.xml")
    Atlas._delete_from_xml_handle(h, arm, [j + "_actuator" for j in arm], [])
    for body, quat in (("left_shoulder_pitch_link", "1.0 0.25 0.1 0.0"), ("right_elbow_link", "1.0 0.0 0.25 0.0"),
                       ("right_shoulder_pitch_link", "1.0 -0.25 0.1 0.0"), ("left_elbow_link", "1.0 0.0 0.25 0.0")):
        h.find("body", body).set("quat", quat)                       # unitreeH1.py:447-468
    m = mjcf.compile_mjcf(h, timestep=0.001, drop_mesh_geoms=True)
    m.save(ROOT / "tests" / "golden" / "UnitreeH1.model.npz")

    def stl_hull(path):
        d = open(path, "rb").read()
        n = struct.unpack("<I", d[80:84])[0]
        tri = np.frombuffer(d[84:84 + 50 * n], dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))["v"]
        pts = np.unique(tri.reshape(-1, 3).astype(np.float64), axis=0)
        return pts[ConvexHull(pts).vertices]
    np.savez_compressed(ROOT / "tests" / "golden" / "UnitreeH1.fixture.npz",
                        left_foot=stl_hull(h1dir / "assets" / "left_ankle_link.stl"),
                        right_foot=stl_hull(h1dir / "assets" / "right_ankle_link.stl"),
                        walk=np.load(ref / "tests" / "test_datasets" / "UnitreeH1.walk.real.npy"),
                        run=np.load(ref / "tests" / "test_datasets" / "UnitreeH1.run.real.npy"))
    print("UnitreeH1 fixture: nv %d, total mass %.2f kg" % (m.nv, m.body_mass.sum()))

    # --- domain-randomisation configurations (plain YAML, copied verbatim: they are data, not code)
    for rel in ["atlas/domain_randomization_atlas.yaml", "humanoid/domain_randomization_humanoid.yaml",
                "quadrupeds/domain_randomization_unitree_a1.yaml"]:
        dst = ROOT / "loco_mujoco_amd" / "environments" / "data" / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        dst.write_text((pkg / "environments" / "data" / rel).read_text())

    # --- mini datasets (same keys/values, re-encoded)
    for rel in ["datasets/quadrupeds/real/mini_datasets/walk_straight.npz",
                "datasets/humanoids/real/mini_datasets/02-constspeed_ATLAS.npz",
                "datasets/humanoids/real/mini_datasets/02-constspeed_TALOS.npz",
                "datasets/humanoids/real/mini_datasets/02-constspeed_UnitreeH1.npz",
                "datasets/humanoids/real/mini_datasets/05-run_UnitreeH1.npz",
                "datasets/humanoids/real/mini_datasets/02-constspeed_UnitreeG1.npz",
                "datasets/humanoids/real/mini_datasets/05-run_UnitreeG1.npz",
                "datasets/humanoids/real/mini_datasets/02-constspeed_reduced_humanoid.npz",
                "datasets/humanoids/real/mini_datasets/05-run_reduced_humanoid.npz"] + [
                "datasets/humanoids/real/mini_datasets/%s_reduced_humanoid_POMDP_%s.npz" % (t, k)
                for t in ("02-constspeed", "05-run") for k in (1, 2, 3, 4, "all")]:
        src = np.load(pkg / rel, allow_pickle=True)
        dst = ROOT / "loco_mujoco_amd" / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        np.savez_compressed(dst, **{k: np.asarray(src[k]) for k in src.files})
        print("dataset", rel, len(src.files), "keys")

    # --- golden rollouts of the reference's own test (tests/test_environments.py:67-94)
    gold = {}
    for t in GOLDEN_TASKS:
        p = ref / "tests" / "test_datasets" / (t + ".npy")
        if p.exists():
            gold[t] = np.load(p)
    (ROOT / "tests" / "golden").mkdir(parents=True, exist_ok=True)
    np.savez_compressed(ROOT / "tests" / "golden" / "reference_rollouts.npz", **gold)
    print("golden:", {k: v.shape for k, v in gold.items()})


if __name__ == "__main__":
    main()
