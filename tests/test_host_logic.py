"""Host-side surface of the drop-in (no GPU): task registry, spaces, reset row 0, datasets, rewards."""

import os
import sys
import numpy as np
import pytest

import loco_mujoco_amd
from loco_mujoco_amd import LocoEnv, mjcf
from loco_mujoco_amd.environments import gymnasium as lm_gym
from loco_mujoco_amd.utils.reward import TargetVelocityReward, VelocityVectorReward

GOLD = np.load(__file__.replace("test_host_logic.py", "golden/reference_rollouts.npz"))


@pytest.fixture(scope="module")
def env():
    np.random.seed(0)
    return LocoEnv.make("UnitreeA1.simple", debug=True)


def test_task_names_and_errors():
    names = loco_mujoco_amd.get_all_task_names()
    assert "UnitreeA1.simple.real" in names and "UnitreeA1.hard.perfect" in names
    with pytest.raises(ValueError):
        LocoEnv.make("UnitreeA1.fast")
    with pytest.raises(ValueError):
        LocoEnv.make("NoSuchRobot.walk")


def test_spaces_match_reference(env):
    assert env.info.observation_space.shape == (37,)
    assert env.info.action_space.shape == (12,)
    assert np.all(env.info.action_space.low == -1) and np.all(env.info.action_space.high == 1)
    assert np.allclose(env.norm_act_mean, 0) and np.allclose(env.norm_act_delta, 1)
    low, high = env.info.observation_space.low, env.info.observation_space.high
    assert low[0] == -np.inf and np.isclose(low[4], -0.802851) and np.isclose(high[6], -0.916298)
    assert list(low[-3:]) == [-1, -1, -np.inf] and list(high[-3:]) == [1, 1, np.inf]
    assert abs(env.dt - 0.01) < 1e-15 and env.info.horizon == 1000 and env.info.gamma == 0.99


def test_reset_row0_reproduces_golden():
    """seed(0) -> (model 0, traj 0, step 47) -> golden row 0 to 1e-12 (reference reset needs no physics)."""
    np.random.seed(0)
    e = LocoEnv.make("UnitreeA1.simple", debug=True)
    obs = e.reset()
    assert obs.dtype == np.float64 and obs.shape == (37,)
    assert np.abs(obs - GOLD["UnitreeA1.simple.real"][0]).max() < 1e-12
    assert e.trajectories.traj_no == 0 and e.trajectories.subtraj_step_no == 47


def test_gym_wrapper_reset():
    np.random.seed(0)
    w = lm_gym.make("LocoMujoco", env_name="UnitreeA1.simple", debug=True)
    obs, info = w.reset()
    assert info == {} and np.abs(obs - GOLD["UnitreeA1.simple.real"][0]).max() < 1e-12
    assert w.observation_space.shape == (37,) and w.action_space.shape == (12,)


def test_batched_reset_draw_order():
    np.random.seed(0)
    e = LocoEnv.make("UnitreeA1.simple", debug=True, n_envs=3)
    obs = e.reset()
    assert obs.shape == (3, 37)
    assert np.abs(obs[0] - GOLD["UnitreeA1.simple.real"][0]).max() < 1e-12
    assert not np.allclose(obs[0], obs[1])


def test_has_fallen_thresholds(env):
    g = GOLD["UnitreeA1.simple.real"]
    assert [bool(env._has_fallen(r)) for r in g] == [False] * 17 + [True]
    o = g[0].copy()
    o[1] = 0.28
    assert env._has_fallen(o)
    o = g[0].copy()
    o[2] = -0.2
    assert env._has_fallen(o)
    spec = env._termination_spec()
    for row in g:
        dev = any(row[i] < lo or row[i] > hi for i, lo, hi in spec)
        assert dev == bool(env._has_fallen(row))


def test_rewards_formulas(env):
    g = GOLD["UnitreeA1.simple.real"]
    r = env.reward(g[3], None, g[4], False)
    v = g[3][[16, 17]]
    want = np.exp(-5 * np.linalg.norm(v - g[3][36] * g[3][[34, 35]]))
    assert np.isclose(r, want)
    assert isinstance(env._reward_function, VelocityVectorReward)
    batch = env._reward_function(g[:5], None, None, None)
    assert batch.shape == (5,) and np.isclose(batch[3], want)
    tv = TargetVelocityReward(2.5, 17)
    assert np.isclose(tv(g[0], None, None, None), np.exp(-(g[0][17] - 2.5) ** 2))


def test_create_dataset(env):
    d = env.create_dataset()
    assert d["states"].shape == (3 * 99, 37) and d["next_states"].shape == (3 * 99, 37)
    assert d["absorbing"].sum() == 0 and d["last"].sum() == 3
    assert np.allclose(d["states"][1:99], d["next_states"][0:98])
    d2 = env.create_dataset()
    assert np.array_equal(d["states"], d2["states"])


def test_reset_table_rows_equal_reset_samples(env):
    tab = env._reset_table()
    assert tab.shape == (300, 18 + 18 + 3)
    np.random.seed(0)
    e = LocoEnv.make("UnitreeA1.simple", debug=True)
    obs = e.reset()
    row = tab[0 * 100 + 47]
    assert np.abs(np.concatenate([row[2:18], row[18:36], row[36:]]) - obs).max() < 1e-12


def test_compiled_model_facts():
    m = mjcf.CompiledModel.load(mjcf.__file__.replace("mjcf.py", "assets/UnitreeA1.torque.model.npz"))
    assert (m.nbody, m.nv, m.nu, m.ngeom) == (15, 18, 12, 38)
    assert m.cone == mjcf.CONE_ELLIPTIC and m.impratio == 100 and m.integrator == mjcf.INT_EULER
    assert np.isclose(m.body_mass.sum(), 4.713 + 4 * (0.696 + 1.013 + 0.226))
    assert np.all(m.dof_frictionloss == 0.2)            # root dofs inherit it too (SURVEY.md §0.6 ii)
    assert list(m.dof_damping[:9]) == [0, 0, 0, 0, 0, 0, 1, 2, 2]
    foot = m.geom_names.index("FR_foot")
    assert m.geom_condim[foot] == 6 and m.geom_priority[foot] == 1 and np.isclose(m.geom_margin[foot], 0.001)
    assert np.allclose(m.geom_solimp[foot], [0.015, 1, 0.031, 0.5, 2])
    assert np.allclose(m.act_gear, 34) and np.allclose(m.act_ctrlrange, [[-1, 1]] * 12)
    # M^-1 diagonal / body inverse weights are consistent with an independent solve
    mm = mjcf.mass_matrix(m, m.qpos0)[0]
    assert np.allclose(np.diag(np.linalg.inv(mm)), m.dof_invweight0)


def test_step_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    np.random.seed(0)
    e = LocoEnv.make("UnitreeA1.simple", debug=True)
    e.reset()
    from loco_mujoco_amd.backend import BackendError
    with pytest.raises(BackendError):
        e.step(np.zeros(12))


def test_recorded_dataset_task(tmp_path, monkeypatch):
    """dataset_type="perfect": a recorded dataset (states / actions / last ..., 100 Hz) under $LOCO_MUJOCO_AMD_DATA becomes
    the trajectory source (x / z integrated from the velocities, restarting at every episode end) and IS the dataset that
    create_dataset returns (reference base.py:499-548, 308-312). The file here is synthetic: 3 episodes cut from the
    bundled Atlas mocap samples."""
    np.random.seed(0)
    src = LocoEnv.make("Atlas.walk", debug=True).create_dataset()
    n = 90
    states = src["states"][:n]
    last = np.zeros(n, dtype=np.int64)
    last[[29, 59, 89]] = 1
    rec = dict(states=states, actions=np.random.uniform(-1, 1, (n, 10)), rewards=np.ones(n), next_states=src["next_states"][:n],
               absorbing=np.zeros(n, dtype=np.int64), last=last)
    d = tmp_path / "datasets" / "humanoids" / "perfect" / "atlas_walk"
    d.mkdir(parents=True)
    np.savez(d / "perfect_expert_dataset_det.npz", **rec)
    monkeypatch.setenv("LOCO_MUJOCO_AMD_DATA", str(tmp_path))
    env = LocoEnv.make("Atlas.walk.perfect")
    t = env.trajectories
    assert t.number_of_trajectories == 3 and list(t.split_points) == [0, 30, 60, 90]
    ds = env.create_dataset()
    assert set(ds) == set(rec) and np.array_equal(ds["actions"], rec["actions"]) and np.array_equal(ds["last"], last)
    ds["actions"][:] = 0
    assert np.array_equal(env.create_dataset()["actions"], rec["actions"])             # a copy every time
    # forward position = running integral of the forward velocity at 100 Hz, restarting with every episode
    files = env.load_dataset_and_get_traj_files("datasets/humanoids/perfect/atlas_walk/perfect_expert_dataset_det.npz", 100)
    vx = states[:, env.get_obs_idx("dq_pelvis_tx")[0]]
    assert files["q_pelvis_tx"][0] == 0 and files["q_pelvis_tx"][30] == 0 and files["q_pelvis_tx"][60] == 0
    assert np.allclose(files["q_pelvis_tx"][1:30], np.cumsum(vx[:29]) / 100)
    assert np.array_equal(files["q_pelvis_tilt"], states[:, env.get_obs_idx("q_pelvis_tilt")[0]])
    obs = env.reset()
    assert obs.shape == (30,) and np.isfinite(obs).all()
    with pytest.raises(AssertionError):
        LocoEnv.make("Atlas.walk.perfect", use_foot_forces=True)


def test_several_models_in_one_batch():
    """n_envs > 1 with several models (the reference's MultiMuJoCo draws one per episode, base.py:186-190). Carried weights differ
    like model variants: one batch, a weight per environment and episode (``_pooled``). The humanoid's four sizes differ in
    geometry: contiguous blocks of environments, one model (= one device batch) each (``_blocks``)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_backend import attach
    from loco_mujoco_amd import lowering
    np.random.seed(0)
    env = LocoEnv.make("Atlas.carry", debug=True, n_envs=10)
    assert env._pooled and not env._blocks and env._n_models == 4
    obs = env.reset()
    w = np.array([0.1, 1.0, 5.0, 10.0])
    assert obs.shape == (10, 31) and np.array_equal(obs[:, -1], w[env._env_model]) and len(set(env._env_model)) > 1
    assert np.array_equal(env._pending_variants, env._env_model) and np.array_equal(env._goal_rows()[:, 0], w[env._env_model])
    first = env._env_model.copy()
    env.reset()
    assert (env._env_model != first).any()                       # a new draw per episode
    nominal = env._chain_model(env._models[0])
    tabs = [lowering.variant_tables(nominal, env._chain_model(m)) for m in env._models]      # what the backend uploads
    assert len(tabs) == 4 and all((tabs[0][0] != t[0]).any() for t in tabs[1:])
    assert not LocoEnv.make("Atlas.carry", debug=True, n_envs=4, weight_mass=5.0)._pooled
    # the humanoid's four sizes (round 6: GROUPED — a size per environment and episode, one batch per size over all environment ids,
    # each stepping its active list): every size's environments are the single-model environment of that size on the same states / actions
    np.random.seed(0)
    env = attach(LocoEnv.make("HumanoidTorque4Ages.walk.all", debug=True, n_envs=6))
    assert env._blocks and env._grouped
    obs = env.reset()
    drawn = env._env_model.copy()
    assert len(set(drawn)) > 1 and np.array_equal(obs[:, -2] * 2 + obs[:, -1], drawn.astype(float))      # the size bits follow the draw
    assert sorted(np.concatenate([env._model_envs(i) for i in range(4)])) == list(range(6))
    a = np.random.uniform(-0.2, 0.2, (6, 13))
    o1, r, d, _ = env.step(a)
    assert o1.shape == (6, 38) and np.array_equal(o1[:, -2:], obs[:, -2:]) and r.shape == (6,) and d.shape == (6,)
    for idx in range(4):
        envs = env._model_envs(idx)
        if len(envs) == 0:
            continue
        np.random.seed(1)
        one = attach(LocoEnv.make("HumanoidTorque4Ages.walk.%d" % (idx + 1), debug=True, n_envs=len(envs)))
        one.reset()
        for k, e in enumerate(envs):
            one._host[k].qpos[:], one._host[k].qvel[:] = obs_state(env, e)
        one._pending_state = True
        o2 = one.step(a[envs])[0]
        assert np.abs(np.atleast_2d(o2) - o1[envs]).max() < 1e-12
    env.reset()
    assert (env._env_model != drawn).any()                       # a new draw per episode
    # every size restarts from its own size's trajectories and shows its size bits
    np.random.seed(0)
    h = LocoEnv.make("HumanoidTorque4Ages.walk.all", debug=True, n_envs=8)
    oh = h.reset()
    assert oh.shape == (8, 38) and np.array_equal(oh[:, -2] * 2 + oh[:, -1], h._env_model.astype(float))
    for idx in range(4):
        h._select_model(idx)
        tab = h._reset_table()
        assert len(tab) == 100 and np.all(tab[:, -2:] == h._env_id())
    # with domain randomisation the sizes keep their contiguous blocks (one pool of model variants per block)
    # (test_model_rule_randomisation_with_several_models)


def obs_state(env, e):
    """(qpos, qvel) the multi-model environment handed to its backend for environment ``e`` at reset."""
    return env._host[e].qpos.copy(), env._host[e].qvel.copy()


def test_a1_position_mode_surface():
    np.random.seed(0)
    e = LocoEnv.make("UnitreeA1.simple", debug=True, action_mode="position")
    m = e._model
    assert e.info.action_space.shape == (12,) and np.all(e.info.action_space.low == -1) and np.all(e.info.action_space.high == 1)
    lo, hi = m.act_ctrlrange[e._action_indices].T                       # joint ranges: actions are normalised onto them
    assert np.allclose(e._preprocess_action(np.ones(12)), hi) and np.allclose(e._preprocess_action(-np.ones(12)), lo)
    assert np.allclose(e.norm_act_mean[:3], [0.0, 1.570795, -1.806414]) and e.reset().shape == (37,)
    e2 = LocoEnv.make("UnitreeA1.simple", debug=True, action_mode="position_difference")
    assert e2._model.act_kind.tolist() == [2] * 12
    with pytest.raises(AssertionError):
        LocoEnv.make("UnitreeA1.simple", debug=True, action_mode="velocity")


def test_atlas_surface():
    np.random.seed(0)
    e = LocoEnv.make("Atlas.walk", debug=True)
    assert e.info.observation_space.shape == (30,) and e.info.action_space.shape == (10,)
    assert np.allclose(e.norm_act_delta, 0.95) and np.allclose(e.norm_act_mean, 0)
    m = e._model
    assert (m.nv, m.nu) == (16, 10) and m.integrator == mjcf.INT_RK4 and m.cone == mjcf.CONE_PYRAMIDAL
    # the XML declares the left leg first; obs/actions follow the spec (right leg first): gathered by name
    assert m.jnt_names[6] == "hip_flexion_l" and e._action_spec[0] == "hip_flexion_r_actuator"
    assert [m.act_names[i] for i in e._action_indices][:2] == ["hip_flexion_r_actuator", "hip_adduction_r_actuator"]
    obs = e.reset()
    assert np.abs(obs - GOLD["Atlas.walk.real"][0]).max() < 1e-14
    assert "Atlas.walk.real" in loco_mujoco_amd.get_all_task_names()
    with pytest.raises(FileNotFoundError):
        LocoEnv.make("Atlas.walk.perfect")           # recorded datasets are downloads, none is bundled


def test_talos_surface():
    np.random.seed(0)
    e = LocoEnv.make("Talos.walk", debug=True)
    assert e.info.observation_space.shape == (34,) and e.info.action_space.shape == (12,)
    assert np.allclose(e.norm_act_delta, 0.95) and np.allclose(e.norm_act_mean, 0)
    m = e._model
    assert (m.nv, m.nu) == (18, 12) and m.integrator == mjcf.INT_EULER and m.cone == mjcf.CONE_PYRAMIDAL
    assert abs(m.timestep - 0.001) < 1e-15 and e._n_substeps == 10
    assert e._action_spec[:2] == ["back_bkz_actuator", "back_bky_actuator"]
    obs = e.reset()
    assert np.abs(obs - GOLD["Talos.walk.real"][0]).max() < 1e-14
    assert "Talos.walk.real" in loco_mujoco_amd.get_all_task_names()
    ok, msg = e._has_fallen(np.r_[0.2, np.zeros(33)], return_err_msg=True)
    assert ok and msg.startswith("pelvis_y_condition")
    e2 = LocoEnv.make("Talos.walk", debug=True, disable_back_joint=True)
    assert e2.info.observation_space.shape == (30,) and e2.info.action_space.shape == (10,) and e2._model.nv == 16
    for bad in (dict(task="Talos.walk", disable_arms=False), dict(task="Talos.walk.perfect")):
        with pytest.raises((NotImplementedError, AssertionError, FileNotFoundError)):
            LocoEnv.make(bad.pop("task"), **bad)


def test_carry_surface():
    """hold_weight: 4 models, weight in the observation, mask, model drawn per episode (random or cyclic)."""
    np.random.seed(0)
    e = LocoEnv.make("Atlas.carry", debug=True)
    assert e.info.observation_space.shape == (31,) and e._n_models == 4
    assert e.info.observation_space.low[-1] == 0.1 and e.info.observation_space.high[-1] == 10.0
    assert e.get_mask(("weight",)).tolist() == [True] * 30 + [False]
    assert e.get_mask(("positions", "velocities")).tolist() == [False] * 30 + [True]
    with pytest.raises(AssertionError):
        e.get_mask(("foot_forces",))
    seen = set()
    for _ in range(12):
        seen.add(float(e.reset()[-1]))
    assert seen == {0.1, 1.0, 5.0, 10.0}
    e._random_env_reset = False                      # the reference's cyclic mode (base.py:189-191)
    order = [float(e.reset()[-1]) for _ in range(5)]
    assert all(order[i + 1] == [0.1, 1.0, 5.0, 10.0][([0.1, 1.0, 5.0, 10.0].index(order[i]) + 1) % 4] for i in range(4))
    t = LocoEnv.make("Talos.carry", debug=True, weight_mass=5.0)
    assert t.info.observation_space.shape == (35,) and t._n_models == 1 and t.reset()[-1] == 5.0
    f = LocoEnv.make("Talos.carry", debug=True, weight_mass=1.0, use_foot_forces=True)
    assert f.info.observation_space.shape == (41,) and f._obs_perm().tolist() == list(range(34)) + list(range(35, 41)) + [34]
    assert f.reset()[-1] == 1.0
    assert LocoEnv.make("Atlas.carry", debug=True, n_envs=8, weight_mass=10.0).reset().shape == (8, 31)
    with pytest.raises(NotImplementedError):
        LocoEnv.make("Atlas.carry", debug=True, weight_mass=2.5)     # not a shipped model
    assert "Atlas.carry.real" in loco_mujoco_amd.get_all_task_names() and "Talos.carry.perfect" not in loco_mujoco_amd.get_all_task_names()


def test_humanoid_torque_surface():
    np.random.seed(0)
    e = LocoEnv.make("HumanoidTorque.run", debug=True)
    assert e.info.observation_space.shape == (36,) and e.info.action_space.shape == (13,)
    assert np.allclose(e.norm_act_delta, 1.0) and np.allclose(e.norm_act_mean, 0)
    m = e._model
    assert (m.nv, m.nu, m.nbody) == (19, 13, 21) and m.integrator == mjcf.INT_RK4 and m.cone == mjcf.CONE_PYRAMIDAL
    assert e._action_spec[:4] == ["mot_lumbar_ext", "mot_lumbar_bend", "mot_lumbar_rot", "mot_hip_flexion_r"]
    # bones are proximity-only bounding capsules; the box feet are the floor colliders
    assert m.n_dropped_mesh_geoms == 75 and int((m.geom_type == mjcf.GEOM_BOX).sum()) == 2
    assert e._reward_function._target_vel == 2.5
    assert LocoEnv.make("HumanoidTorque.walk", debug=True)._reward_function._target_vel == 1.25
    obs = e.reset()
    assert np.abs(obs - GOLD["HumanoidTorque.run.real"][0]).max() < 1e-14
    assert e._has_fallen(GOLD["HumanoidTorque.run.real"][-1], return_err_msg=True)[0]
    d = e.create_dataset()
    assert d["states"].shape[1] == 36 and len(d["states"]) == len(d["next_states"])
    assert "HumanoidTorque.run.real" in loco_mujoco_amd.get_all_task_names()
    for kw in (dict(use_box_feet=False), dict(disable_arms=False)):
        with pytest.raises(NotImplementedError):
            loco_mujoco_amd.HumanoidTorque(**kw)
    with pytest.raises(FileNotFoundError):
        LocoEnv.make("HumanoidTorque.walk.perfect")


def test_humanoid_muscle_surface():
    np.random.seed(0)
    e = LocoEnv.make("HumanoidMuscle.walk", debug=True)
    assert e.info.observation_space.shape == (36,) and e.info.action_space.shape == (92,)
    m = e._model
    assert (m.nv, m.nu, m.na, m.ntendon) == (19, 92, 92, 92) and m.integrator == mjcf.INT_EULER
    # muscle controls live in [0, 1]: actions in [-1, 1] are mapped onto them (base.py:122-126, 606-621)
    assert np.allclose(e.norm_act_mean, 0.5) and np.allclose(e.norm_act_delta, 0.5)
    assert e._action_spec[0] == "glut_med1_r" and e._action_spec[43] == "glut_med1_l" and e._action_spec[-1] == "extobl_l"
    assert [m.act_names[i] for i in e._action_indices] == e._action_spec
    obs = e.reset()
    assert np.abs(obs - GOLD["HumanoidMuscle.walk.real"][0]).max() < 1e-14
    assert "HumanoidMuscle.run.real" in loco_mujoco_amd.get_all_task_names()
    from loco_mujoco_amd import lowering
    cm, info = lowering.lower(m, e._device_task())
    assert info["muscles_per_chain"] == [43, 43, 6, 0]      # every tendon runs over the pelvis and one chain


def test_domain_randomization_config_and_atlas_back_chain(tmp_path):
    from loco_mujoco_amd.utils.domain_randomization import JointRandomization, KIND_CLIPPED_NORMAL, KIND_NORMAL, KIND_UNIFORM
    data = os.path.join(os.path.dirname(loco_mujoco_amd.__file__), "environments", "data")
    cfg = os.path.join(data, "atlas", "domain_randomization_atlas.yaml")
    np.random.seed(0)
    # shipped file + default robot: the randomised back joints do not exist -> a no-op
    e0 = LocoEnv.make("Atlas.walk", debug=True, domain_randomization_config=cfg)
    assert not e0._domain_rand.active
    # with the back chain: 19 dofs, 13 motors, third chain of 3 links; damping of back_bkz / back_bkx ~ U[4, 6]
    e = LocoEnv.make("Atlas.walk", debug=True, disable_back_joint=False, domain_randomization_config=cfg)
    m = e._model
    assert (m.nv, m.nu) == (19, 13) and e.info.observation_space.shape == (36,)
    from loco_mujoco_amd import lowering
    cm, info = lowering.lower(m, e._device_task())
    assert info["n_chains"] == 3 and sorted(len(c) for c in info["chains"]) == [3, 5, 5]
    spec = e._domain_rand.spec
    hit = [m.jnt_names[d] for d in np.nonzero(spec[0, :, 0])[0]]
    assert hit == ["back_bkz", "back_bkx"] and (spec[1:, :, 0] == 0).all()
    assert tuple(spec[0, m.jnt_id("back_bkz")]) == (KIND_UNIFORM, 4.0, 6.0)
    state = np.random.get_state()[1].copy()
    e.reset()
    assert (np.random.get_state()[1] != state).any()                   # trajectory draws happened ...
    d = e._pending_dof_params[0][0]
    assert 4.0 <= d[m.jnt_id("back_bkz")] <= 6.0 and d[m.jnt_id("back_bky")] == m.dof_damping[m.jnt_id("back_bky")]
    e.seed(3); e.reset(); a = e._pending_dof_params.copy()
    e.seed(3); e.reset()
    assert np.array_equal(a, e._pending_dof_params)                     # ... and the parameter stream is reproducible
    # rule semantics incl. the reference's quirks (domain_randomization.py:299-383)
    y = tmp_path / "dr.yaml"
    y.write_text("Joints:\n  FR_hip_joint:\n    damping: {sigma: 0.5}\n    stiffness: {uniform_range: [1.0, 2.0]}\n"
                 "  FL_hip_joint:\n    damping: {uniform_range_delta: 0.25}\n    armature: {sigma: 0.0}\n")
    a1 = LocoEnv.make("UnitreeA1.simple", debug=True)
    jr = JointRandomization(a1._model, str(y))
    i, j = a1._model.jnt_id("FR_hip_joint"), a1._model.jnt_id("FL_hip_joint")
    assert tuple(jr.spec[0, i]) == (KIND_CLIPPED_NORMAL, a1._model.dof_damping[i], 0.5)
    assert tuple(jr.spec[1, i]) == (KIND_NORMAL, 1.0, 2.0)              # "uniform_range" on stiffness draws a normal
    assert tuple(jr.spec[0, j]) == (KIND_UNIFORM, a1._model.dof_damping[j] - 0.25, a1._model.dof_damping[j] + 0.25)
    s = jr.sample(2000)
    assert s.shape == (3, 2000, 18) and s[0, :, i].min() >= 0 and abs(np.median(s[1, :, i]) - 1.0) < 0.2
    assert s.min() >= 0                                                 # the N(a, b) quirk is clipped at 0: no negative joint parameter
    assert not jr.has_model_rules


def test_domain_randomization_of_compile_time_constants(tmp_path):
    """Armature, ``Inertial`` and ``Geoms`` rules (reference utils/domain_randomization.py:386-514) produce model VARIANTS:
    ``mjcf.model_variant`` recompiles the derived constants, ``lowering.variant_tables`` reduces a variant to what differs
    from the nominal tables."""
    from loco_mujoco_amd import lowering, mjcf
    from loco_mujoco_amd.utils.domain_randomization import JointRandomization
    np.random.seed(0)
    env = LocoEnv.make("Talos.walk", debug=True)
    m = env._model
    cfg = os.path.join(os.path.dirname(__file__), "golden", "dr_talos_inertial.yaml")
    jr = JointRandomization(m, cfg)
    assert jr.has_model_rules and jr.active and [d for d, *_ in jr.armature_rules] == [m.jnt_id("back_bkz")]
    assert sorted(m.body_names[b] for b, _, _ in jr.body_rules) == ["leg_left_4_link", "leg_right_6_link"]
    np.random.seed(5)
    v = jr.sample_model_variant()
    b6, b4 = m.body_id("leg_right_6_link"), m.body_id("leg_left_4_link")
    assert abs(v.body_mass[b6] - m.body_mass[b6]) <= 0.5 and v.body_mass[b6] != m.body_mass[b6] and v.body_mass[b4] != m.body_mass[b4]
    # diaginertia: the principal moments move by at most delta, the principal axes stay
    ev0, ev1 = np.linalg.eigvalsh(m.body_inertia[b6]), np.linalg.eigvalsh(v.body_inertia[b6])
    assert np.abs(np.sort(ev1) - np.sort(ev0)).max() <= 0.001 + 1e-12 and not np.allclose(ev0, ev1)
    untouched = [b for b in range(m.nbody) if b not in (b6, b4)]
    assert np.array_equal(v.body_mass[untouched], m.body_mass[untouched]) and np.array_equal(v.body_inertia[untouched], m.body_inertia[untouched])
    g6 = np.nonzero(np.asarray(m.geom_body) == b6)[0]
    assert len(g6) and (np.abs(v.geom_friction[g6] - m.geom_friction[g6]) <= [0.2, 0.001, 0.00005]).all() and not np.array_equal(v.geom_friction[g6], m.geom_friction[g6])
    assert v.dof_armature[m.jnt_id("back_bkz")] > 0 and np.array_equal(np.delete(v.dof_armature, m.jnt_id("back_bkz")), np.delete(m.dof_armature, m.jnt_id("back_bkz")))
    # derived constants are recomputed; an unchanged variant reproduces the nominal ones bit for bit
    assert not np.allclose(v.dof_invweight0, m.dof_invweight0) and v.meaninertia != m.meaninertia
    same = mjcf.model_variant(m)
    assert np.array_equal(same.dof_invweight0, m.dof_invweight0) and np.array_equal(same.body_invweight0, m.body_invweight0)
    # lowering: only the inertial record and the geom table differ
    nominal = env._chain_model()
    rec0, gt0, gp0 = lowering.variant_tables(nominal, nominal)
    rec, gt, gp = lowering.variant_tables(nominal, env._chain_model(v))
    assert rec.shape == (lowering.IR_SIZE * lowering.NCHAIN,) and len(gp) == 0 and (rec != rec0).any() and (gt != gt0).any()
    changed_fields = set(np.nonzero(rec != rec0)[0] // lowering.NCHAIN)
    assert lowering.IR_SCALE in changed_fields and any(f < lowering.IR_ROOT and f % lowering.IR_LINK == 0 for f in changed_fields)
    other = LocoEnv.make("Talos.walk", debug=True, disable_back_joint=True)
    with pytest.raises(lowering.UnsupportedModel):
        lowering.variant_tables(nominal, other._chain_model())
    # the environment builds its pool with the randomisation's own generator: reproducible, untouched main stream
    e = LocoEnv.make("Talos.walk", debug=True, n_envs=3, domain_randomization_config=cfg, n_model_variants=4, model_variants_per_reset=0)
    state = np.random.get_state()[1].copy()
    m1, t1 = e._build_model_variants(e._chain_model())
    assert np.array_equal(np.random.get_state()[1], state) and len(t1) == 4 and len(m1) == 4
    e.seed(0); np.random.seed(0)
    _, ta = e._build_model_variants(e._chain_model())
    e.seed(0)
    _, tb = e._build_model_variants(e._chain_model())
    assert all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(ta, tb))
    # the pool is built at the first reset() from the seed alone (no GPU, no dependence on what ran before the first step()) ...
    e.seed(3); e.reset()
    pool_a = [x[0].copy() for x in e._variant_tables]
    assert e._pending_variants.shape == (3,) and (e._pending_variants >= 0).all() and (e._pending_variants < 4).all()
    e.reset(); e.reset()
    assert all(np.array_equal(x, y[0]) for x, y in zip(pool_a, e._variant_tables))          # model_variants_per_reset=0: a fixed pool
    e.seed(3); assert e._variant_tables is None
    e.reset()
    assert all(np.array_equal(x, y[0]) for x, y in zip(pool_a, e._variant_tables))
    # ... and with the default rolling refresh every reset() replaces pool entries round-robin by fresh draws; a batch of at most
    # that many environments runs one brand-new model per environment and episode (the reference: a new model at every reset)
    r = LocoEnv.make("Talos.walk", debug=True, n_envs=2, domain_randomization_config=cfg, n_model_variants=6, model_variants_per_reset=2)
    r.seed(1); r.reset()
    p0 = [x[0].copy() for x in r._variant_tables]
    r.reset()
    changed = [j for j in range(6) if not np.array_equal(p0[j], r._variant_tables[j][0])]
    assert changed == [0, 1] and list(r._pending_variants) == [0, 1] and r._variant_dirty
    r.reset()
    assert list(r._pending_variants) == [2, 3] and not np.array_equal(p0[2], r._variant_tables[2][0]) and np.array_equal(p0[4], r._variant_tables[4][0])
    assert list(r.refresh_model_variants()) == [4, 5, 0, 1, 2, 3]
    with pytest.raises(ValueError, match="n_model_variants must be >= 1"):
        LocoEnv.make("Talos.walk", debug=True, n_envs=2, domain_randomization_config=cfg, n_model_variants=0)
    # the reference's own Talos file (data/talos/domain_randomization_talos.yaml:32-41) asks for `fullinertia` on a body whose
    # <inertial> has diaginertia + quat: its assertion fires (domain_randomization.py:497), and so does this one
    y = tmp_path / "dr.yaml"
    y.write_text("Inertial:\n  leg_right_5_link:\n    fullinertia:\n      uniform_range_delta: 0.001\n")
    with pytest.raises(AssertionError, match="fullinertia not allowed"):
        JointRandomization(m, str(y)).sample_model_variant()
    # fullinertia rule: singular values of the upper-triangular matrix (A1 trunk has a fullinertia attribute)
    a1 = LocoEnv.make("UnitreeA1.simple", debug=True)
    y.write_text("Inertial:\n  trunk:\n    fullinertia:\n      uniform_range_delta: 0.001\n    mass: {sigma: 0.2}\n")
    ja = JointRandomization(a1._model, str(y))
    va = ja.sample_model_variant()
    bt = a1._model.body_id("trunk")
    assert int(a1._model.body_inertial_kind[bt]) == 2 and not np.allclose(va.body_inertia[bt], a1._model.body_inertia[bt])
    assert np.allclose(va.body_inertia[bt], va.body_inertia[bt].T) and np.linalg.eigvalsh(va.body_inertia[bt]).min() > 0
    # geom mass / density on a body without <inertial>: not built
    y.write_text("Geoms:\n  trunk:\n    friction:\n      sigma: [0.1, 0.0, 0.0]\n")
    assert JointRandomization(a1._model, str(y)).has_model_rules


def test_humanoid_4_ages_surface():
    np.random.seed(0)
    e = LocoEnv.make("HumanoidMuscle4Ages.run.2", debug=True)
    assert e.info.observation_space.shape == (38,) and e.info.action_space.shape == (92,)
    assert list(e.info.observation_space.low[-2:]) == [0, 0] and list(e.info.observation_space.high[-2:]) == [1, 1]
    assert list(e._env_id()) == [0, 1] and e._n_goal() == 2
    m1 = LocoEnv.make("HumanoidMuscle.run", debug=True)._model
    i = e._model.act_names.index("glut_med1_r")
    assert np.isclose(e._model.act_gainprm[i, 2], m1.act_gainprm[i, 2] * 0.36)            # muscle force ~ s^2
    assert np.allclose(e._model.act_lengthrange[i], m1.act_lengthrange[i] * 0.6)           # tendon range ~ s
    t = LocoEnv.make("HumanoidTorque4Ages.walk.4", debug=True)
    assert np.allclose(t._model.act_gear, LocoEnv.make("HumanoidTorque.walk", debug=True)._model.act_gear)
    # reward: target speed scales with the size; on the device it is the plain target-velocity reward
    st = np.zeros(38); st[17] = 1.5; st[-2:] = [0, 1]
    assert np.isclose(e.reward(st, None, st, False), np.exp(-(1.5 - 2.5 * 0.6) ** 2))
    assert e._reward_function.device_spec() == (1, [17, 2.5 * 0.6])
    tab = e._reset_table()
    assert tab.shape[1] == 2 * 19 + 2 and (tab[:, -2:] == [0, 1]).all()
    assert "HumanoidTorque4Ages.walk.3.real" in loco_mujoco_amd.get_all_task_names()
    # mode "all": four models in one environment, one drawn per episode (n_envs = 1 only)
    np.random.seed(0)
    a = LocoEnv.make("HumanoidTorque4Ages.walk.all", debug=True)
    assert a._n_models == 4 and a.more_than_one_env and len(a._scaling_trajectory_map) >= 4
    seen = set()
    for _ in range(12):
        ob = a.reset()
        idx = a._current_model_idx
        seen.add(idx)
        assert list(ob[-2:]) == [idx >> 1, idx & 1] and a._model is a._models[idx]
        lo, hi = a._scaling_trajectory_map[idx]
        assert lo <= a.trajectories.traj_no < hi                      # start state from the trajectories of that size
    assert len(seen) >= 3
    assert LocoEnv.make("HumanoidTorque4Ages.walk.all", debug=True, n_envs=8)._grouped       # batches: a size per environment and episode
    with pytest.raises(TypeError):
        e.reset(obs=np.zeros(38))


def test_custom_reward_callback_sees_reference_shapes():
    """a ``reward_type="custom"`` callback written for the reference gets what mushroom-rl's MuJoCo.step gives it:
    ONE environment's 1-D state and the UN-normalised action of ``_preprocess_action`` (``base.py:606-621``), also in a batch
    (environment by environment). The physics behind the step is the oracle stand-in here (no GPU)."""
    from oracle_backend import attach
    seen = []

    def callback(state, action, next_state):
        seen.append((np.shape(state), np.array(action, dtype=float), np.shape(next_state)))
        return float(state[16]) + 0.0 * float(next_state[16])             # v_x: indexes like the reference's users do

    for n in (1, 3):
        np.random.seed(0)
        env = attach(LocoEnv.make("UnitreeA1.simple", debug=True, n_envs=n, reward_type="custom", reward_params=dict(reward_callback=callback)))
        prev = np.atleast_2d(env.reset())
        a = np.random.RandomState(1).uniform(-0.5, 0.5, (n, 12))
        seen.clear()
        obs, r, done, info = env.step(a if n > 1 else a[0])
        calls = seen[-n:]                       # (the oracle stand-in of the backend evaluates the functor too, before LocoEnv.step does)
        assert len(seen) >= n and all(s[0] == (37,) and s[2] == (37,) for s in calls)
        for e in range(n):
            assert np.allclose(calls[e][1], env._preprocess_action(a[e]))    # torques in N m (gear 33.5...), not [-1, 1]
            assert np.isclose(np.atleast_1d(r)[e], prev[e, 16])
        assert info == {}


def test_foot_force_layout_and_reward_quirk():
    """with ``use_foot_forces`` the goal entries sit BEFORE the twelve foot-force entries; the reference's
    velocity-vector reward then reads foot forces through its negative indices (``unitreeA1.py:491-497``) — reproduced by
    evaluating the functor on the host (no device form), while the device's observation map puts the goal where the
    observation has it."""
    from loco_mujoco_amd import lowering
    np.random.seed(0)
    env = LocoEnv.make("UnitreeA1.simple", debug=True, use_foot_forces=True)
    assert env.info.observation_space.shape == (49,)
    assert env._reward_device_spec() is None and env._reward_function.device_spec() is not None
    task = env._device_task()
    assert task["reward_type"] == 0 and task["nobs"] == 49 and len(task["grf_groups"]) == 4
    cmod, info = lowering.lower(env._model, task)                                # a positive goal index no longer raises
    plain = LocoEnv.make("UnitreeA1.simple", debug=True)
    assert plain._reward_device_spec() is not None and plain._device_task()["reward_type"] == 2


def test_recorded_dataset_tasks_unitree_a1_and_4_ages(tmp_path, monkeypatch):
    """dataset_type="perfect" for the quadruped (``unitreeA1.py:354-418,694-706``: the arrow's rotation matrix is rebuilt
    from the (cos, sin) columns, ``goal_speed`` is the file's mean planar trunk speed) and for the four-sizes torque humanoid
    (``base_humanoid_4_ages.py:410-412,449-454``, file ``<name>_<mode>.npz``). Synthetic files cut from the bundled mocap
    samples; the muscle variant of the four sizes only has mocap data in the reference (``humanoids.py:945-947``)."""
    from loco_mujoco_amd.utils.math import mat2angle_xy
    np.random.seed(0)
    src = LocoEnv.make("UnitreeA1.simple", debug=True).create_dataset()
    n = 60
    last = np.zeros(n, dtype=np.int64)
    last[[19, 39, 59]] = 1
    rec = dict(states=src["states"][:n], actions=np.random.uniform(-1, 1, (n, 12)), rewards=np.ones(n), next_states=src["next_states"][:n],
               absorbing=np.zeros(n, dtype=np.int64), last=last)
    d = tmp_path / "datasets" / "quadrupeds" / "perfect" / "unitreea1_simple"
    d.mkdir(parents=True)
    np.savez(d / "perfect_expert_dataset_det.npz", **rec)
    monkeypatch.setenv("LOCO_MUJOCO_AMD_DATA", str(tmp_path))
    env = LocoEnv.make("UnitreeA1.simple.perfect")
    t = env.trajectories
    assert t.number_of_trajectories == 3 and list(t.split_points) == [0, 20, 40, 60]
    ds = env.create_dataset()
    assert np.array_equal(ds["actions"], rec["actions"]) and np.array_equal(ds["last"], last)
    files = env.load_dataset_and_get_traj_files(d / "perfect_expert_dataset_det.npz", 100)
    s = rec["states"]
    assert np.allclose([mat2angle_xy(np.reshape(m, (3, 3))) for m in files["dir_arrow"][:5]], np.arctan2(s[:5, 35], s[:5, 34]))
    speed = np.linalg.norm(s[:, 16:18], axis=1)
    assert np.allclose(files["goal_speed"], speed.mean()) and files["q_trunk_tx"][20] == 0 and files["q_trunk_tx"][40] == 0
    assert np.allclose(files["q_trunk_tx"][1:20], np.cumsum(s[:19, 16]) / 100)
    obs = env.reset()
    assert obs.shape == (37,) and np.isfinite(obs).all() and abs(obs[36] - speed.mean()) < 1e-12
    with pytest.raises(AssertionError):
        LocoEnv.make("UnitreeA1.simple.perfect", use_foot_forces=True)
    # the four-sizes torque humanoid
    src4 = LocoEnv.make("HumanoidTorque4Ages.walk.2", debug=True).create_dataset()
    rec4 = dict(states=src4["states"][:n], actions=np.random.uniform(-1, 1, (n, 13)), rewards=np.ones(n), next_states=src4["next_states"][:n],
                absorbing=np.zeros(n, dtype=np.int64), last=last)
    d4 = tmp_path / "datasets" / "humanoids" / "perfect" / "humanoid4ages_torque_walk"
    d4.mkdir(parents=True)
    np.savez(d4 / "HumanoidTorque4Ages_walk_stochastic_dataset_2.npz", **rec4)
    e4 = LocoEnv.make("HumanoidTorque4Ages.walk.2.perfect")
    assert e4.trajectories.number_of_trajectories == 3 and np.array_equal(e4.create_dataset()["actions"], rec4["actions"])
    o4 = e4.reset()
    assert o4.shape == (38,) and list(o4[-2:]) == [0.0, 1.0]
    with pytest.raises(Exception):
        LocoEnv.make("HumanoidMuscle4Ages.walk.2.perfect")


def test_unitree_g1_surface():
    """UnitreeG1 (reference humanoids/unitreeG1.py): default = torso joint + arms, lowered to four six-link chains of which the two
    arm chains share the torso link; the reduced configurations are plain root + chains models."""
    from loco_mujoco_amd import lowering
    np.random.seed(0)
    env = LocoEnv.make("UnitreeG1.walk", debug=True)
    m = env._model
    assert (m.nv, m.nu) == (29, 23) and env.info.observation_space.shape == (56,) and env.info.action_space.shape == (23,)
    assert [k for k, _, _ in env.obs_helper.observation_spec][:8] == ["q_pelvis_tx", "q_pelvis_tz", "q_pelvis_ty", "q_pelvis_tilt", "q_pelvis_list",
                                                                     "q_pelvis_rotation", "q_left_hip_pitch_joint", "q_left_hip_roll_joint"]
    assert np.allclose(m.dof_frictionloss, 0.1) and np.allclose(m.dof_damping[6:], 0.5) and np.allclose(m.dof_armature[6:], 0.01)
    obs = env.reset()
    assert obs.shape == (56,) and not env._has_fallen(obs)
    low = obs.copy(); low[0] = -0.31
    assert env._has_fallen(low) and env._has_fallen(low, return_err_msg=True)[1].startswith("pelvis_y_condition")
    # the arms hang off the torso link: two chains share it as their first link (owner lane 2, massless copy in lane 3)
    cm, info = lowering.lower(m, env._device_task())
    assert [len(c) for c in info["chains"]] == [6, 6, 6, 6] and info["shared_first"] == {3: 2}
    assert info["chains"][2][0] == info["chains"][3][0] == m.body_id("torso_link")
    role = [int(cm[lowering.HEADER_SIZE + lowering.CM_CHAINS + lowering.C_DUPROLE * lowering.NCHAIN + c]) for c in range(4)]
    assert role == [0, 0, 1, -1]
    link0 = lambda c, f: cm[lowering.HEADER_SIZE + lowering.CM_CHAINS + (lowering.C_LINKS + f) * lowering.NCHAIN + c]
    t = m.jnt_id("torso_joint")
    assert link0(2, lowering.D_DOF) == link0(3, lowering.D_DOF) == t and link0(2, lowering.D_DAMP) == 0.5 and link0(3, lowering.D_DAMP) == 0
    assert link0(2, lowering.D_SIZE + lowering.L_MASS) > 5 and link0(3, lowering.D_SIZE + lowering.L_MASS) == 0
    assert link0(2, lowering.D_LIMITED) == 1 and link0(3, lowering.D_LIMITED) == 0 and link0(3, lowering.D_QOBS) == -1
    # foot forces: four force points per foot, 8 groups x 3 entries behind the state (unitreeG1.py:295-317); four groups per leg chain
    # are compiled in the six-link kernels
    ff = LocoEnv.make("UnitreeG1.walk", debug=True, use_foot_forces=True)
    assert ff.info.observation_space.shape == (56 + 24,) and ff._get_grf_size() == 24
    cmf, inf = lowering.lower(ff._model, ff._device_task())
    obs_slots = [int(cmf[lowering.HEADER_SIZE + lowering.CM_CHAINS + k * lowering.NCHAIN + c]) for c in range(4) for k in lowering.C_GRF_OBS]
    assert sorted(x for x in obs_slots if x >= 0) == [56 + 3 * i for i in range(8)]
    with pytest.raises(ValueError):
        LocoEnv.make("UnitreeG1.carry")
    for kw, nv, nu, chains in ((dict(disable_back_joint=True), 28, 22, [5, 5, 6, 6]), (dict(disable_arms=True), 19, 13, [1, 6, 6]),
                               (dict(disable_arms=True, disable_back_joint=True), 18, 12, [6, 6])):
        e = LocoEnv.make("UnitreeG1.run", debug=True, **kw)
        assert (e._model.nv, e._model.nu) == (nv, nu) and e.info.observation_space.shape == (2 * nv - 2,)
        cm, info = lowering.lower(e._model, e._device_task())
        assert sorted(len(c) for c in info["chains"]) == chains and int(cm[lowering.H_MAXLINKS]) == 6
        assert e.reset().shape == (2 * nv - 2,)
    ds = LocoEnv.make("UnitreeG1.run", debug=True, disable_back_joint=True).create_dataset()
    assert ds["states"].shape[1] == 54 and len(ds["states"]) > 50


def test_play_trajectory_from_velocity_integrates_the_dataset_velocities():
    """Reference ``base.py:388-476``: positions of the first sample, then qpos += dt * qvel with the trajectory's velocities."""
    np.random.seed(0)
    env = LocoEnv.make("UnitreeA1.simple", debug=True)
    obs = env.play_trajectory_from_velocity(n_episodes=1, n_steps_per_episode=12)
    assert obs.shape == (12, 37) and np.isfinite(obs).all()
    nq = 16                                              # observed joint positions (trunk x, y dropped)
    np.random.seed(0)
    env2 = LocoEnv.make("UnitreeA1.simple", debug=True)
    env2.reset()
    sample = env2.trajectories.get_current_sample()
    q = np.array([np.asarray(x).reshape(-1)[0] for x in sample[:18]])
    for k in range(12):
        v = np.array([np.asarray(x).reshape(-1)[0] for x in sample[18:36]])
        q = q + env2.dt * v
        assert np.abs(obs[k, :nq] - q[2:]).max() < 1e-12 and np.abs(obs[k, nq:nq + 18] - v).max() < 1e-12
        sample = env2.trajectories.get_next_sample()
    assert env._get_joint_pos().shape == (18,) and env._get_joint_vel().shape == (18,)
    assert hasattr(LocoEnv, "_delete_from_xml_handle")


def test_get_mask_of_the_four_sizes_and_wrapper_surface():
    """``base_humanoid_4_ages.py:187-241`` (the size bits are only part of the mask when the environment holds more than one
    size — the reference's behaviour, although the observation always carries them) and the rest of the Gymnasium wrapper's
    surface (``gymnasium.py:112-165``)."""
    from loco_mujoco_amd.environments.gymnasium import GymnasiumWrapper
    np.random.seed(0)
    e = LocoEnv.make("HumanoidTorque4Ages.walk.all", debug=True)
    m = e.get_mask(("velocities", "env_type"))
    assert m.shape == (38,) and m.sum() == 17 and m[:17].all() and not m[17:].any()
    assert e.get_mask("positions").sum() == 21
    one = LocoEnv.make("HumanoidTorque4Ages.walk.1", debug=True)
    assert one.get_mask("positions").shape == (36,) and one.info.observation_space.shape == (38,)
    with pytest.raises(AssertionError):
        one.get_mask("env_type")
    with pytest.raises(AssertionError):
        e.get_mask("foot_forces")
    g = GymnasiumWrapper("UnitreeA1.simple", debug=True)
    assert g._set_action_space().shape == (12,) and g._set_observation_space().shape == (37,)
    assert g.play_trajectory_from_velocity(n_steps_per_episode=3).shape == (3, 37)


def test_bench_leg_rate_identity_and_parity_flag():
    """bench.py computes every rate with ONE function from (environments, steps, seconds) of the leg itself, and the parity
    sample says whether it is inside the stated tolerance (no GPU needed for either)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    r = bench.leg_rate(4096, 1, 200, 0.3163)
    assert abs(r["value"] - 4096 * 200 / 0.3163) < 1e-6 and abs(r["ms_per_step"] - 1.5815) < 1e-9
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 - 4096) < 1e-6
    r2 = bench.leg_rate(2048, 4, 20, 0.05)
    assert abs(r2["value"] - 2048 * 4 * 20 / 0.05) < 1e-6
    src = open(os.path.join(root, "bench.py")).read()
    assert "within_tolerance" in src and "sys.exit(3)" in src
    # no rate is formed from a cumulative counter any more
    assert 'vals[13] / ' not in src and 'env_steps / elapsed' not in src
    # the neighbour rule (round 6): a state beyond the tolerance is set aside only when the oracle itself, for an input within one float32
    # ulp, produces the DEVICE'S result to within the tolerance — not when its own result merely moves
    q0, v0 = np.zeros(3), np.zeros(3)
    jump = lambda qa, va: (qa + (0.5 if qa[0] > 0 else 0.0), va + (0.5 if qa[0] > 0 else 0.0))      # an oracle with a switch at q[0] = 0
    assert bench.oracle_reaches(jump, q0, v0, q0 + 0.5, v0 + 0.5)[0]            # the device took the other branch: a probe gets there
    reached, near = bench.oracle_reaches(jump, q0, v0, q0 + 5.0, v0 + 5.0)      # the device is 10 x further than the oracle ever jumps
    assert not reached and near > 100
    smooth = lambda qa, va: (qa, va)
    assert not bench.oracle_reaches(smooth, q0, v0, q0 + 2e-4, v0)[0]           # a stable oracle excuses nothing
    # the parity sample comes from the states of the timed rollout, and UnitreeH1 is reported, not gating
    assert "lm_get_state after the timed block" in src and set(bench.PARITY_REPORTED_NOT_GATING) == {"UnitreeH1.walk", "UnitreeH1.run", "UnitreeH1.carry"}


def test_model_rule_randomisation_with_several_models(tmp_path):
    """Several models in one environment (the reference's MultiMuJoCo) TOGETHER with randomisation rules that change compile-time
    constants: one pool of model variants per model, for n_envs = 1 (one device batch per model, the model drawn per episode) and
    for contiguous blocks. Only several models inside ONE device batch (the carried weights at n_envs > 1) exclude the pool."""
    cfg = os.path.join(os.path.dirname(__file__), "golden", "dr_talos_inertial.yaml")
    np.random.seed(0)
    e = LocoEnv.make("Talos.carry", debug=True, domain_randomization_config=cfg, n_model_variants=2)
    assert e._n_models == 4 and not e._pooled and not e._blocks
    seen = set()
    for _ in range(12):
        e.reset()
        idx = e._current_model_idx
        seen.add(idx)
        pool = e._variant_pools[idx]
        assert len(pool["tables"]) == 2 and e._pending_variants.shape == (1,) and 0 <= int(e._pending_variants[0]) < 2
        assert len(e._variant_models[idx]) == 2
    assert len(seen) > 1 and set(e._variant_pools) == seen          # a pool per model the episodes drew, none for the others
    # n_envs = 1 compiles ONE fresh model per reset, not model_variants_per_reset of them
    e2 = LocoEnv.make("Talos.walk", debug=True, domain_randomization_config=cfg, n_model_variants=4, model_variants_per_reset=4)
    e2.reset()
    p0 = [x[0].copy() for x in e2._variant_tables]
    e2.reset()
    assert [j for j in range(4) if not np.array_equal(p0[j], e2._variant_tables[j][0])] == [0] and list(e2._pending_variants) == [0]
    # blocks: the humanoid's four sizes with n_envs = 8 -> four device batches of two environments, four pools
    y = tmp_path / "dr.yaml"
    y.write_text("Joints:\n  knee_angle_r:\n    armature:\n      sigma: 0.005\n")
    np.random.seed(0)
    b = LocoEnv.make("HumanoidTorque4Ages.walk.all", debug=True, n_envs=8, domain_randomization_config=str(y), n_model_variants=3, model_variants_per_reset=2)
    assert b._blocks and b._n_models == 4
    b.reset()
    assert sorted(b._variant_pools) == [0, 1, 2, 3] and all(len(p["tables"]) == 3 for p in b._variant_pools.values())
    assert b._pending_variants.shape == (8,) and (b._pending_variants >= 0).all() and (b._pending_variants < 3).all()
    first = {i: [t[0].copy() for t in b._variant_pools[i]["tables"]] for i in range(4)}
    b.reset()
    for i in range(4):
        changed = [j for j in range(3) if not np.array_equal(first[i][j], b._variant_pools[i]["tables"][j][0])]
        assert changed == [0, 1] and list(b._pending_variants[b._model_envs(i)]) == [0, 1]
    # the four sizes differ in their inertial numbers: so do their pools
    assert not np.array_equal(b._variant_pools[0]["tables"][2][0], b._variant_pools[3]["tables"][2][0])


def test_lowering_keeps_every_collider_and_builds_the_six_link_pair_tables():
    """Round 4: no floor-collidable geom is demoted to a proximity sphere any more (MAXG covers the humanoid's trunk chain with its 65
    geoms — the hand and finger bones used to be counted only), and the six-link robots get their self-collision tables: link-pair
    lists of up to 128 entries per lane that stay out of the workgroup's LDS copy of the constant table (H_CM_USED ends before them)."""
    from loco_mujoco_amd import lowering
    np.random.seed(0)
    for task, kw in (("HumanoidTorque.run", {}), ("HumanoidMuscle.walk", {}), ("HumanoidTorque4Ages.walk.1", {}), ("UnitreeA1.simple", {}),
                     ("Atlas.walk", {}), ("Talos.walk", {}), ("UnitreeH1.run", {}), ("UnitreeG1.walk", {}), ("UnitreeH1.walk", dict(disable_arms=False))):
        env = LocoEnv.make(task, debug=True, **kw)
        cmod, info = lowering.lower(env._model, env._device_task())
        assert not info.get("demoted_geoms"), (task, info.get("demoted_geoms"))
        assert int(cmod[lowering.H_CM_USED]) * 4 <= 13 * 1024          # the constant table of every robot fits its LDS share
        if info["max_links"] == 6:
            t = info["self_collision_tables"]
            assert t["convex"] > 100 and max(t["link_pairs"]) <= lowering.MAXLP and int(cmod[lowering.H_NGPAIR]) == t["closed_form"] + t["native"] + t["convex"]
            assert int(cmod[lowering.H_CM_USED]) == int(cmod[lowering.H_OFF_PRUNE]) < int(cmod[lowering.H_OFF_LPAIR])      # (round 6: prune records and link groups stay out too)
        elif info.get("self_collision_tables"):
            assert max(info["self_collision_tables"]["link_pairs"]) <= 64 and int(cmod[lowering.H_CM_USED]) > int(cmod[lowering.H_OFF_LPAIR])
    # five-link humanoids without muscles run in the eight-slot families (a fifth contact on a leg must not abandon the control step)
    talos = LocoEnv.make("Talos.walk", debug=True)
    assert lowering.lower(talos._model, talos._device_task())[1]["max_contacts"] == 8


# ---------------------------------------------------------------------------------------------------------------
# The model compiler on the device (csrc/lm_compile.hip): its program, executed here in numpy
# ---------------------------------------------------------------------------------------------------------------

def _run_model_compiler_program(ib, db, values, nominal_tables):
    """numpy restatement of csrc/lm_compile.hip's kernel for ONE environment: the tables it writes from the draws ``values``."""
    from loco_mujoco_amd import lowering as L
    nv, nrb, ngs, nd, nslot, nrec, ncon, nbody, pyramidal, balance = [int(x) for x in ib[1:11]]
    impratio, boundmass, boundinertia, minval = db[0:4]
    io, do = L.MC_IH_SIZE, L.MC_DH_SIZE
    i_draw = ib[io:io + nd * 4].reshape(nd, 4); io += nd * 4
    i_rb = ib[io:io + nrb * 4].reshape(nrb, 4); io += nrb * 4
    i_rec = ib[io:io + nrec * 2].reshape(nrec, 2); io += nrec * 2
    i_con = ib[io:io + ncon * 8].reshape(ncon, 8); io += ncon * 8
    assert io == len(ib)
    do += nd * 2
    d_rb = db[do:do + nrb * 55].reshape(nrb, 55); do += nrb * 55
    jac = db[do:do + nbody * 6 * nv].reshape(nbody, 6, nv); do += nbody * 6 * nv
    mbase = db[do:do + nv * nv].reshape(nv, nv); do += nv * nv
    arm = db[do:do + nv].copy(); do += nv
    fd = db[do:do + nv]; do += nv
    slot = db[do:do + nslot * 10].reshape(nslot, 10).copy(); do += nslot * 10
    fric = db[do:do + ngs * 3].reshape(ngs, 3).copy(); do += ngs * 3
    assert do == len(db)
    mass, vals, sv = d_rb[:, 0].copy(), d_rb[:, 1:7].copy(), np.zeros((nrb, 3))
    for (kind, target, idx, comp), v in zip(i_draw, values):
        if target == 0:
            arm[idx] = v
        elif target == 1:
            mass[idx] = v
        elif target == 2:
            vals[idx, comp] = v
        elif target == 3:
            sv[idx, comp] = v
        else:
            fric[idx, comp] = v
    sym = lambda a: np.array([[a[0], a[3], a[4]], [a[3], a[1], a[5]], [a[4], a[5], a[2]]])
    M = mbase + np.diag(arm)
    for j in range(nrb):
        rd = d_rb[j]
        if i_rb[j, 3]:
            t = rd[16:25].reshape(3, 3) @ np.diag(sv[j]) @ rd[25:34].reshape(3, 3)
            vals[j] = [t[0, 0], t[1, 1], t[2, 2], t[0, 1], t[0, 2], t[1, 2]]
        if i_rb[j, 1] == 2:
            inr = sym(vals[j])
        else:
            r = rd[7:16].reshape(3, 3)
            inr = r @ np.diag(vals[j, :3]) @ r.T
        if boundinertia > 0 or balance:
            ev, evec = np.linalg.eigh(inr)
            if boundinertia > 0:
                ev = np.maximum(ev, boundinertia)
            if balance and ev[0] + ev[1] < ev[2]:
                ev[:] = ev.mean()
            inr = evec @ np.diag(ev) @ evec.T
        if boundmass > 0:
            mass[j] = max(mass[j], boundmass)
        x = rd[46:55].reshape(3, 3)
        J = jac[i_rb[j, 0]]
        M += mass[j] * J[:3].T @ J[:3] + J[3:].T @ (x @ inr @ x.T) @ J[3:]
        c, r = rd[34:37], rd[37:46].reshape(3, 3)
        o = r @ inr @ r.T + mass[j] * (c @ c * np.eye(3) - np.outer(c, c))
        slot[i_rb[j, 2]] += np.concatenate([[mass[j]], mass[j] * c, [o[0, 0], o[1, 1], o[2, 2], o[0, 1], o[0, 2], o[1, 2]]])
    minv = np.linalg.inv(M)
    V = np.zeros(3 * nv + 1 + nslot * 10)
    V[0:nv], V[nv:2 * nv] = arm, np.diag(minv)
    V[2 * nv:3 * nv] = np.where(fd >= 0, np.maximum(minval, fd * np.diag(minv)), 0.0)
    V[3 * nv] = 1.0 / ((np.trace(M) / nv) * nv)
    for s in range(nslot):
        a = slot[s]
        if a[0] <= 0:
            continue
        m_, c = a[0], a[1:4] / a[0]
        o = sym(a[4:10]) - m_ * (c @ c * np.eye(3) - np.outer(c, c))
        V[3 * nv + 1 + s * 10:3 * nv + 11 + s * 10] = [m_, c[0], c[1], c[2], o[0, 0], o[1, 1], o[2, 2], o[0, 1], o[0, 2], o[1, 2]]
    biw = np.array([np.trace(jac[b, :3] @ minv @ jac[b, :3].T) / 3.0 for b in range(nbody)])
    rec, gt, gpt = [np.array(t, dtype=np.float64) for t in nominal_tables]
    for dst, src in i_rec:
        rec[dst] = V[src]
    for tab_i, at, st, code, ga, gb, ba, bb in i_con:
        tab = gt if tab_i == 0 else gpt
        dim, mix, pair = code & 15, (code >> 4) & 15, (code >> 8) & 1
        f3 = np.maximum(fric[ga], fric[gb]) if mix == 0 else (fric[ga] if mix == 1 else fric[gb])
        fr = np.array([f3[0], f3[0], f3[1], f3[2], f3[2]])
        tran = biw[ba] + biw[bb]
        if pyramidal:
            mu = 0.0 if (pair and dim != 3) else fr[0]
            vt = 2 * mu * mu * (1 + mu * mu) * tran if dim == 3 else (4.0 * tran if pair else tran)
            vmu, rr = mu, np.ones(5)
        else:
            vt, vmu, rr1 = tran, fr[0] / np.sqrt(max(minval, impratio)), 1.0 / max(minval, impratio)
            rr = np.array([rr1, rr1 * fr[0] ** 2 / fr[1] ** 2] + [rr1 * fr[0] ** 2 / fr[k] ** 2 for k in (2, 3, 4)])
        tab[at], tab[at + 2 * st] = vt, vmu
        tab[at + 3 * st:at + 8 * st:st], tab[at + 8 * st:at + 13 * st:st] = fr, rr
    return rec, gt, gpt


_DR_ALL_RULES = {
    "Talos.walk": "golden/dr_talos_inertial.yaml",
}


@pytest.mark.parametrize("task, rules", [
    ("Talos.walk", None),
    ("UnitreeA1.simple", "Inertial:\n  trunk:\n    mass: {sigma: 1.0}\n    fullinertia:\n      uniform_range_delta: 0.002\n"
                         "Geoms:\n  FR_calf:\n    friction:\n      sigma: [0.1, 0.001, 0.00001]\n"
                         "Joints:\n  FR_hip_joint:\n    armature:\n      uniform_range: [0.01, 0.02]\n"),
    ("HumanoidTorque.walk", "Default:\n  Inertial:\n    mass:\n      sigma: 0.3\n  Geoms:\n    friction:\n      sigma: [0.1, 0.001, 0.00001]\n"),
])
def test_model_compiler_program_reproduces_the_host_compiler(task, rules, tmp_path):
    """The program the device-side model compiler runs (lowering.model_compiler_tables), executed in numpy from a set of draws,
    gives the tables of the host path for the same draws: ``variant_tables(lower(mjcf.model_variant(...)))`` — inertial record
    (link inertias, armature, dof_invweight0, friction-loss regulariser, solver scale), geom table and geom-pair table. Talos: the
    golden configuration (mass, diaginertia, friction, armature); the quadruped: fullinertia through the singular values, elliptic
    cones, self-collision pairs; the humanoid: EVERY body and geom drawn, the compiler's inertia bounds, pyramids."""
    from loco_mujoco_amd import lowering
    from loco_mujoco_amd.utils.domain_randomization import JointRandomization
    if rules is None:
        cfg = os.path.join(os.path.dirname(__file__), "golden", "dr_talos_inertial.yaml")
    else:
        cfg = tmp_path / "dr.yaml"
        cfg.write_text(rules)
        cfg = str(cfg)
    env = LocoEnv.make(task, debug=True)
    m = env._model
    jr = JointRandomization(m, cfg)
    ops, svd = jr.model_draw_ops()
    assert len(ops) > 0
    # the ops ARE the sampler: drawn in sequence with the reference's generator they give sample_model_variant's model
    np.random.seed(11)
    values = [jr.draw_op(op) for op in ops]
    np.random.seed(11)
    ref = jr.sample_model_variant()
    v = jr.variant_from_draws(values)
    assert np.array_equal(v.body_mass, ref.body_mass) and np.array_equal(v.body_inertia, ref.body_inertia)
    assert np.array_equal(v.dof_armature, ref.dof_armature) and np.array_equal(v.geom_friction, ref.geom_friction)
    assert np.array_equal(v.dof_invweight0, ref.dof_invweight0)
    ib, db, info = lowering.model_compiler_tables(m, env._device_task(), ops, svd)
    assert ib.dtype == np.int32 and db.dtype == np.float64 and info["n_draw"] == len(ops)
    nominal = env._chain_model()
    want = lowering.variant_tables(nominal, env._chain_model(v))
    got = _run_model_compiler_program(ib, db, values, lowering.variant_tables(nominal, nominal))
    for name, w, g in zip(("record", "geom table", "pair table"), want, got):
        assert w.shape == g.shape
        err = np.abs(w - g) / np.maximum(np.abs(w), 1e-30)
        assert err[w != g].max(initial=0.0) < 1e-9, (name, np.nonzero(err > 1e-9)[0][:8], w[err > 1e-9][:4], g[err > 1e-9][:4])
    assert (want[0] != lowering.variant_tables(nominal, nominal)[0]).any()


def test_model_compiler_is_the_default_and_the_pool_an_option():
    """Rules that change compile-time constants: `n_model_variants` unset -> the model compiler on the device (one model per
    environment and episode, like the reference); an explicit pool size keeps the host-compiled pool; several models in one batch
    (the carried weights) use the variant tables for the MODELS and keep the pool; rules without compile-time constants need neither."""
    cfg = os.path.join(os.path.dirname(__file__), "golden", "dr_talos_inertial.yaml")
    assert LocoEnv.make("Talos.walk", debug=True, n_envs=4, domain_randomization_config=cfg)._use_model_compiler
    assert not LocoEnv.make("Talos.walk", debug=True, n_envs=4, domain_randomization_config=cfg, n_model_variants=8)._use_model_compiler
    assert not LocoEnv.make("Talos.carry", debug=True, n_envs=4, domain_randomization_config=cfg)._use_model_compiler
    assert not LocoEnv.make("Talos.walk", debug=True, n_envs=4)._use_model_compiler
    with pytest.raises(ValueError, match="do not compile their models on the device"):
        LocoEnv.make("Talos.walk", debug=True, n_envs=4).model_of_env(0)


def test_geom_mass_rule_on_a_body_with_inertial_is_no_model_rule(tmp_path):
    """A geom `mass` / `density` rule on a body WITH an <inertial> element draws nothing and changes nothing (the engine's compiler
    prefers the <inertial>): such a file must ask for neither the model compiler (an empty draw program is refused by the library)
    nor a variant pool — and with a friction rule beside it only the friction is drawn."""
    y = tmp_path / "dr.yaml"
    y.write_text("Geoms:\n  trunk:\n    mass:\n      sigma: 0.3\n")
    e = LocoEnv.make("UnitreeA1.simple", debug=True, n_envs=2, domain_randomization_config=str(y))
    assert not e._domain_rand.has_model_rules and not e._use_model_compiler
    assert e._domain_rand.model_draw_ops()[0] == []
    y.write_text("Geoms:\n  trunk:\n    mass:\n      sigma: 0.3\n    friction:\n      sigma: [0.1, 0.0, 0.0]\n")
    e = LocoEnv.make("UnitreeA1.simple", debug=True, n_envs=2, domain_randomization_config=str(y))
    ops = e._domain_rand.model_draw_ops()[0]
    assert e._use_model_compiler and len(ops) > 0 and all(op[3] == e._domain_rand.TARGET_FRICTION for op in ops)


def test_model_compiler_program_is_checked_before_it_reaches_the_device():
    """The binding refuses a program whose draws or drawn bodies index outside the tables the device kernel sizes from the header
    (the library itself checks the sizes and every gather / contact op)."""
    from loco_mujoco_amd import backend, lowering
    from loco_mujoco_amd.utils.domain_randomization import JointRandomization
    env = LocoEnv.make("Talos.walk", debug=True)
    jr = JointRandomization(env._model, os.path.join(os.path.dirname(__file__), "golden", "dr_talos_inertial.yaml"))
    ib, db, _ = lowering.model_compiler_tables(env._model, env._device_task(), *jr.model_draw_ops())
    nominal = env._chain_model()
    tabs = lowering.variant_tables(nominal, nominal)

    class Recorder:             # the wrapper up to the library call
        n, _h, calls = 4, None, []

        class _lib:
            @staticmethod
            def lm_set_model_compiler(*a):
                Recorder.calls.append(a)
                return 0

    backend.HipBatch.set_model_compiler(Recorder, (ib, db), tabs, seed=1)
    assert len(Recorder.calls) == 1 and Recorder.n_model_draws == int(ib[4]) and Recorder.n_variants == 4
    nd = int(ib[4])
    for at, value in ((lowering.MC_IH_SIZE + 2, 999), (lowering.MC_IH_SIZE + 1, 7), (lowering.MC_IH_SIZE + 4 * nd + 2, 99), (lowering.MC_IH_SIZE + 4 * nd, 0)):
        bad = ib.copy()
        bad[at] = value
        with pytest.raises(backend.BackendError, match="model-compiler program"):
            backend.HipBatch.set_model_compiler(Recorder, (bad, db), tabs, seed=1)
    assert len(Recorder.calls) == 1
