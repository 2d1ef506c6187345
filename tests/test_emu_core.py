"""
The device code (loco_mujoco_amd/csrc/lm_core.h) executed on the CPU by the 4-thread lane emulator
(tests/emu) against the fp64 oracle and the golden rollout: lets the float32 quad algorithm be checked in
a container without a GPU. Test tooling only — the product never runs this path.
"""

import os
import numpy as np
import pytest

from loco_mujoco_amd import LocoEnv, lowering
from oracle.model_blob import pack_model
from oracle.pyoracle import Oracle
from emu import pyemu

GOLD = np.load(__file__.replace("test_emu_core.py", "golden/reference_rollouts.npz"))


def _tables_are_contiguous(cmod, with_muscles):
    """The chain model = header, constant table, geom table, [muscle table], then the global tables one behind the other: geom pairs,
    hull vertices, their neighbour lists, body pairs, adjacency blocks."""
    L = lowering
    off = L.HEADER_SIZE + L.CM_SIZE + L.GT_SIZE + (L.MT_SIZE if with_muscles else 0)
    for h_off, n in ((L.H_OFF_GPT, L.GPAIR_SIZE * int(cmod[L.H_NGPAIR])), (L.H_OFF_MESHV, 4 * int(cmod[L.H_NMESHV])), (L.H_OFF_MESHN, int(cmod[L.H_NMESHN])),
                     (L.H_OFF_BPT, L.BP_SIZE * int(cmod[L.H_NBPAIR])), (L.H_OFF_MESHADJ, 4 * int(cmod[L.H_NMESHADJ]))):
        if int(cmod[h_off]) != off:
            return False
        off += n
    return off == len(cmod)


@pytest.fixture(scope="module")
def setup():
    np.random.seed(0)
    env = LocoEnv.make("UnitreeA1.simple", debug=True)
    cmod, info = lowering.lower(env._model, env._device_task())
    o = Oracle(pack_model(env._model))
    return env, cmod, info, o


def actions(n):
    np.random.seed(0)
    np.random.randint(0, 1), np.random.randint(0, 3), np.random.randint(0, 100)
    return [np.random.randn(12) * 0.1 for _ in range(n)]


def test_lowering_structure(setup):
    env, cmod, info, o = setup
    assert info["n_chains"] == 4 and info["max_links"] == 3
    assert _tables_are_contiguous(cmod, with_muscles=False)
    assert sorted(int(x) for x in info["dof_to_lane"][6:]) == [0] * 3 + [1] * 3 + [2] * 3 + [3] * 3


@pytest.mark.parametrize("ls_points", [1, 4])
def test_core_stages_and_golden_steps(setup, ls_points):
    env, cmod, info, o = setup
    g = GOLD["UnitreeA1.simple.real"]
    acts = actions(17)
    for k in (0, 3, 8, 12, 16):
        qpos, qvel = np.concatenate([[0, 0], g[k, :16]]), g[k, 16:34]
        f = o.forward(qpos, qvel, acts[k])
        q, v, w, cnt, d = pyemu.run(cmod, qpos, qvel, acts[k], nsub=1, debug_env=0, ls_points=ls_points)
        assert cnt["ncon"] == f["ncon"] and cnt["overflow"] == 0
        assert np.abs(d["M"] - f["M"]).max() < 1e-5
        assert np.abs(d["bias"] - f["bias"]).max() < 1e-4
        assert np.abs(d["qacc"] - f["qacc"]).max() < 1e-4 * max(1.0, np.abs(f["qacc"]).max())   # float32 Newton-decrement stop
        q10, v10, _, _, _ = pyemu.run(cmod, qpos, qvel, acts[k], nsub=10, ls_points=ls_points)
        assert np.abs(q10[0, 2:] - g[k + 1, :16]).max() < 1e-5
        assert np.abs(v10[0] - g[k + 1, 16:34]).max() < 1e-3


@pytest.mark.parametrize("task,nu,rows", [("Atlas.walk", 10, (0, 7, 20)), ("HumanoidTorque.run", 13, (0, 9, 30))])
def test_core_humanoids_rk4_pyramidal(task, nu, rows):
    """kernel variant <5,8,RK4> (chains of 5 links, pyramidal cones, plane-box feet) against oracle and golden rows."""
    np.random.seed(0)
    env = LocoEnv.make(task, debug=True)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    o = Oracle(pack_model(m))
    g = GOLD[task + ".real"]
    qidx = [m.jnt_id(n) for k, n, t in env.obs_helper.observation_spec if k.startswith("q_")]
    nq = len(qidx) - 2
    np.random.seed(0)
    np.random.randint(0, 1), np.random.randint(0, 1), np.random.randint(0, 100)
    acts = [np.random.randn(nu) * 0.1 for _ in range(max(rows) + 1)]
    for k in rows:
        qpos, qvel = np.zeros(m.nv), np.zeros(m.nv)
        qpos[qidx[2:]] = g[k, :nq]
        qvel[qidx] = g[k, nq:]
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(acts[k])
        f = o.forward(qpos, qvel, ctrl)
        q, v, w, cnt, d = pyemu.run(cmod, qpos, qvel, acts[k], nsub=1, debug_env=0)
        assert cnt["ncon"] == 4 * f["ncon"] and cnt["overflow"] == 0          # 4 pyramid edges per frictional contact
        assert np.abs(d["M"] - f["M"]).max() < 2e-5
        assert np.abs(d["qacc"] - f["qacc"]).max() < 1e-4 * max(1.0, np.abs(f["qacc"]).max())
        q10, v10, _, _, _ = pyemu.run(cmod, qpos, qvel, acts[k], nsub=10)
        assert np.abs(q10[0][qidx[2:]] - g[k + 1, :nq]).max() < 1e-5
        assert np.abs(v10[0][qidx] - g[k + 1, nq:]).max() < 1e-3


@pytest.mark.parametrize("ls_points", [1, 4])
def test_core_talos_euler_pyramidal(ls_points):
    """kernel variant <5,8,Euler,pyramidal> (Talos: 3 chains, implicit damping, frictionloss rows) vs oracle and golden rows."""
    np.random.seed(0)
    env = LocoEnv.make("Talos.walk", debug=True)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    assert info["n_chains"] == 3 and info["max_links"] == 5 and info["max_contacts"] == 8      # (stands on four per leg; the eight-slot family: lowering.py)
    o = Oracle(pack_model(m))
    g = GOLD["Talos.walk.real"]
    qidx = [m.jnt_id(n) for k, n, t in env.obs_helper.observation_spec if k.startswith("q_")]
    np.random.seed(0)
    np.random.randint(0, 1), np.random.randint(0, 1), np.random.randint(0, 100)
    acts = [np.random.randn(12) * 0.1 for _ in range(14)]
    for k in (0, 4, 13):
        qpos, qvel = np.zeros(m.nv), np.zeros(m.nv)
        qpos[qidx[2:]] = g[k, :16]
        qvel[qidx] = g[k, 16:34]
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(acts[k])
        f = o.forward(qpos, qvel, ctrl)
        q, v, w, cnt, d = pyemu.run(cmod, qpos, qvel, acts[k], nsub=1, debug_env=0, ls_points=ls_points)
        assert cnt["ncon"] == f["ncon"] and cnt["overflow"] == 0 and cnt["unhandled"] == 0
        assert np.abs(d["M"] - f["M"]).max() < 2e-5
        assert np.abs(d["qacc"] - f["qacc"]).max() < 1e-4 * max(1.0, np.abs(f["qacc"]).max())
        q10, v10, _, _, _ = pyemu.run(cmod, qpos, qvel, acts[k], nsub=10, ls_points=ls_points)
        assert np.abs(q10[0][qidx[2:]] - g[k + 1, :16]).max() < 1e-5
        assert np.abs(v10[0][qidx] - g[k + 1, 16:34]).max() < 1e-3


def test_core_per_environment_joint_parameters():
    """DR = true code path (damping / stiffness / frictionloss per environment) of the Talos family vs the oracle run on a
    model compiled with those values; with the table's own values it must reproduce the nominal kernel bit for bit."""
    import copy
    np.random.seed(0)
    env = LocoEnv.make("Talos.walk", debug=True)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    g = GOLD["Talos.walk.real"]
    qidx = [m.jnt_id(n) for k, n, t in env.obs_helper.observation_spec if k.startswith("q_")]
    rs = np.random.RandomState(3)
    n = 4
    qpos, qvel = np.zeros((n, m.nv)), np.zeros((n, m.nv))
    qpos[:, qidx[2:]] = g[[0, 3, 4, 13], :16]
    qvel[:, qidx] = g[[0, 3, 4, 13], 16:34]
    acts = rs.uniform(-0.3, 0.3, (n, 12))
    q0, v0, _, _, _ = pyemu.run(cmod, qpos, qvel, acts, nsub=10, ls_points=4)
    q1, v1, _, _, _ = pyemu.run(cmod, qpos, qvel, acts, nsub=10, ls_points=4, dr=True)
    assert np.array_equal(q0, q1) and np.array_equal(v0, v1)
    damp = (np.tile(m.dof_damping, (n, 1)) * rs.uniform(0.5, 2.0, (n, m.nv)) + (m.dof_damping > 0) * rs.uniform(0, 1, (n, m.nv))).astype(np.float32)
    floss = (np.tile(m.dof_frictionloss, (n, 1)) * rs.uniform(0.5, 1.5, (n, m.nv))).astype(np.float32)
    stiff = np.tile(m.jnt_stiffness, (n, 1)).astype(np.float32)
    q2, v2, _, _, _ = pyemu.run(cmod, qpos, qvel, acts, nsub=10, ls_points=4, dof_params=np.stack([damp, stiff, floss]))
    assert np.abs(v2 - v0).max() > 1e-3
    for i in range(n):
        m2 = copy.copy(m)
        m2.dof_damping, m2.jnt_stiffness, m2.dof_frictionloss = damp[i].astype(float), stiff[i].astype(float), floss[i].astype(float)
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(acts[i])
        qo, vo, _, _ = Oracle(pack_model(m2)).step(qpos[i], qvel[i], ctrl, nsub=10)
        assert np.abs(q2[i] - qo).max() < 1e-5 and np.abs(v2[i] - vo).max() < 1e-3


def test_core_position_servos():
    """UnitreeA1 with position servos (kp 100, force range +-33.5): device float32 code vs the oracle, one control step
    from dataset states with actions that drive some servos into their force limit."""
    np.random.seed(0)
    env = LocoEnv.make("UnitreeA1.simple", debug=True, action_mode="position")
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    assert cmod[lowering.H_ACTMODE] == 1
    o = Oracle(pack_model(m))
    tab = env._reset_table()
    rs = np.random.RandomState(5)
    rows = tab[rs.randint(0, len(tab), 6)]
    sat = 0
    for i, row in enumerate(rows):
        qpos, qvel = row[:m.nv], row[m.nv:2 * m.nv]
        a = np.clip((qpos[m.act_dof][np.argsort(env._action_indices)] - env.norm_act_mean) / env.norm_act_delta + rs.uniform(-0.3, 0.3, 12), -1, 1)
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(a)
        f = o.forward(qpos, qvel, ctrl)
        sat += int((np.abs(f["actuator_force"]) == 33.5).sum())
        q, v, w, cnt, d = pyemu.run(cmod, qpos, qvel, a, nsub=1, debug_env=0, ls_points=4)
        assert np.abs(d["smooth"] - (f["passive"] - f["bias"] + f["actuator"])).max() < 2e-3
        assert np.abs(d["qacc"] - f["qacc"]).max() < 1e-4 * max(1.0, np.abs(f["qacc"]).max())
        qo, vo, _, _ = o.step(qpos, qvel, ctrl, nsub=10)
        q10, v10, _, _, _ = pyemu.run(cmod, qpos, qvel, a, nsub=10, ls_points=4)
        assert np.abs(q10[0] - qo).max() < 1e-5 and np.abs(v10[0] - vo).max() < 2e-3
    assert sat > 0


@pytest.mark.parametrize("task,nu,row", [("UnitreeA1.simple", 12, 5), ("Talos.walk", 12, 4), ("Atlas.walk", 10, 7),
                                         ("HumanoidTorque.run", 13, 9), ("HumanoidMuscle.walk", 92, 0)])
def test_core_replicated_layout(task, nu, row):
    """The small-batch layout of the GPU (4 replicas per environment: 16 threads here) with every replica on a PRIVATE copy
    of its lane memory that is reconciled only inside Q::fence(): a hand-over between replicas that the device code does
    not bracket with a fence reads stale (NaN-poisoned) data and fails. One control step from a golden row against the
    golden successor and against the one-replica run (equal up to the summation order of the dealt work)."""
    np.random.seed(0)
    env = LocoEnv.make(task, debug=True)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    g = GOLD[task + ".real"]
    qidx = [m.jnt_id(n) for k, n, t in env.obs_helper.observation_spec if k.startswith("q_")]
    nq = len(qidx) - 2
    np.random.seed(0)
    np.random.randint(0, 1), np.random.randint(0, 3 if task.startswith("UnitreeA1") else 1), np.random.randint(0, 100)
    acts = [np.random.randn(nu) * 0.1 for _ in range(row + 1)]
    qpos, qvel = np.zeros(m.nv), np.zeros(m.nv)
    qpos[qidx[2:]] = g[row, :nq]
    qvel[qidx] = g[row, nq:nq + len(qidx)]
    act0 = np.zeros(m.na) if m.na else None
    q1, v1, _, c1, _ = pyemu.run(cmod, qpos, qvel, acts[row], nsub=10, ls_points=4, act=act0)
    q4, v4, _, c4, _ = pyemu.run(cmod, qpos, qvel, acts[row], nsub=10, rep=4, act=act0)
    assert np.isfinite(q4).all() and np.isfinite(v4).all()
    assert np.abs(q4[0][qidx[2:]] - g[row + 1, :nq]).max() < 1e-5 and np.abs(v4[0][qidx] - g[row + 1, nq:nq + len(qidx)]).max() < 1e-3
    assert np.abs(q4 - q1).max() < 1e-6 and np.abs(v4 - v1).max() < 5e-5
    assert c4["overflow"] == 0 and c4["ncon"] > 0


@pytest.mark.parametrize("task,nu", [("UnitreeA1.simple", 12), ("Atlas.walk", 10), ("HumanoidTorque.walk", 13),
                                     ("HumanoidMuscle.run", 92), ("Talos.carry", 12)])
def test_core_replicated_layout_random_states(task, nu):
    """Six dataset states per robot, full-range random actions, two control steps in the replicated layout (private lane
    memory per replica, NaN-poisoned, reconciled at the fences only) against the fp64 oracle: contacts making and breaking,
    several contacts per lane (the dealt gradient / Hessian paths), saturated actuators."""
    np.random.seed(0)
    kw = dict(weight_mass=5.0) if task.endswith("carry") else {}
    env = LocoEnv.make(task, debug=True, **kw)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    o = Oracle(pack_model(m))
    tab = env._reset_table()
    rs = np.random.RandomState(11)
    rows = tab[rs.randint(0, len(tab), 6)]
    acts = rs.uniform(-1, 1, (2, 6, nu))
    q, v = rows[:, :m.nv].copy(), rows[:, m.nv:2 * m.nv].copy()
    w = np.zeros_like(q)
    act = np.zeros((6, m.na)) if m.na else None
    qo, vo, wo, ao = q.copy(), v.copy(), np.zeros_like(q), (np.zeros((6, m.na)) if m.na else None)
    used = np.ones(6, dtype=bool)
    for k in range(2):
        q, v, w, cnt, dbg = pyemu.run(cmod, q, v, acts[k], nsub=10, rep=4, warm=w, act=act)
        if m.na:
            act = dbg["act"]
        for i in range(6):
            ctrl = np.zeros(m.nu)
            ctrl[env._action_indices] = env._preprocess_action(acts[k, i])
            if m.na:
                qo[i], vo[i], ao[i], wo[i], st = o.step_act(qo[i], vo[i], ao[i], ctrl, 10, wo[i])
            else:
                qo[i], vo[i], wo[i], st = o.step(qo[i], vo[i], ctrl, 10, wo[i])
            used[i] &= st["unhandled_pairs"] == 0          # a mesh / cylinder within reach of the floor: no collider on either side
    assert np.isfinite(q).all() and np.isfinite(v).all() and used.sum() >= 3
    assert np.abs(q - qo)[used].max() < 2e-4 and np.abs(v - vo)[used].max() < 2e-2


def test_core_replicated_layout_with_per_environment_parameters():
    """The kernel variant that tripped the -O2 build on the GPU (<5,4,Euler,pyramid,DR,4 replicas>), on the CPU."""
    np.random.seed(0)
    env = LocoEnv.make("Talos.walk", debug=True)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    tab = env._reset_table()
    rs = np.random.RandomState(1)
    n = 3
    rows = tab[rs.randint(0, len(tab), n)]
    acts = rs.uniform(-0.3, 0.3, (n, 12))
    damp = (np.tile(m.dof_damping, (n, 1)) * rs.uniform(0.5, 2.0, (n, m.nv))).astype(np.float32)
    floss = (np.tile(m.dof_frictionloss, (n, 1)) * rs.uniform(0.5, 1.5, (n, m.nv))).astype(np.float32)
    prm = np.stack([damp, np.zeros_like(damp), floss])
    q1, v1, _, _, _ = pyemu.run(cmod, rows[:, :m.nv], rows[:, m.nv:2 * m.nv], acts, nsub=10, ls_points=4, dof_params=prm)
    q4, v4, _, _, _ = pyemu.run(cmod, rows[:, :m.nv], rows[:, m.nv:2 * m.nv], acts, nsub=10, rep=4, dof_params=prm)
    assert np.isfinite(v4).all() and np.abs(q4 - q1).max() < 1e-6 and np.abs(v4 - v1).max() < 5e-5


def test_core_muscles():
    """kernel variant <5,8,Euler,muscles>: tendon paths, muscle forces and activation dynamics in float32 vs oracle/golden."""
    np.random.seed(0)
    env = LocoEnv.make("HumanoidMuscle.walk", debug=True)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    assert info["muscles_per_chain"] == [43, 43, 6, 0] and _tables_are_contiguous(cmod, with_muscles=True)
    o = Oracle(pack_model(m))
    g = GOLD["HumanoidMuscle.walk.real"]
    qidx = [m.jnt_id(n) for k, n, t in env.obs_helper.observation_spec if k.startswith("q_")]
    np.random.seed(0)
    np.random.randint(0, 1), np.random.randint(0, 1), np.random.randint(0, 100)
    act = np.zeros(m.na)
    for k in range(18):                       # rows 15-17: a muscle shorter than lmin (force-length fall-through)
        a = np.random.randn(92) * 0.1
        qpos, qvel = np.zeros(m.nv), np.zeros(m.nv)
        qpos[qidx[2:]] = g[k, :17]
        qvel[qidx] = g[k, 17:36]
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(a)
        if k in (0, 9, 16):
            f = o.forward(qpos, qvel, ctrl, act=act)
            q1, v1, w1, cnt, d = pyemu.run(cmod, qpos, qvel, a, nsub=1, debug_env=0, act=act)
            want = f["passive"] - f["bias"] + f["actuator"]
            assert np.abs(d["smooth"] - want).max() < 1e-3 * max(1.0, np.abs(want).max())
            q10, v10, _, _, d10 = pyemu.run(cmod, qpos, qvel, a, nsub=10, act=act)
            assert np.abs(q10[0][qidx[2:]] - g[k + 1, :17]).max() < 1e-5
            assert np.abs(v10[0][qidx] - g[k + 1, 17:36]).max() < 1e-3
        qo, vo, act_new, wo, st = o.step_act(qpos, qvel, act, ctrl, nsub=10)
        if k in (0, 9, 16):
            assert np.abs(d10["act"][0] - act_new).max() < 1e-5
        act = act_new


def test_core_plane_cylinder_contacts():
    """Plane vs cylinder on the device code (engine: mjc_PlaneCylinder — deepest rim point of the near cap, the rim point
    under it on the far cap, two more points of the near cap at +-120 degrees): Atlas states with a thigh / shin cylinder on
    the floor (tests/golden/atlas_cylinder_states.npz, found with the oracle by lowering tilted, folded robots onto the
    floor), one forward pass and then three control steps against the fp64 oracle, which restates the same construction
    (unpinned: no golden rollout of the reference has a cylinder on the floor)."""
    np.random.seed(0)
    env = LocoEnv.make("Atlas.walk", debug=True)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    o = Oracle(pack_model(m))
    d = np.load(__file__.replace("test_emu_core.py", "golden/atlas_cylinder_states.npz"))
    pick = [0, 3, 11, 40, 77, 180]          # (the GPU test runs all 213)
    n_multi = 0
    for i in pick:
        q, v = d["q"][i].copy(), d["v"][i].copy()
        ctrl = np.zeros(m.nu)
        f = o.forward(q, v, ctrl)
        assert any(m.geom_type[c["geom2"]] == 3 for c in f["contacts"])
        _, _, _, cnt, dbg = pyemu.run(cmod, q, v, np.zeros(10), nsub=1, debug_env=0, ls_points=4)
        assert 4 * f["ncon"] <= cnt["ncon"] <= 4 * f["ncon"] + 3 and cnt["overflow"] == 0, (i, cnt, f["ncon"])     # RK4: four passes, contacts may join in the later ones
        assert np.abs(dbg["qacc"] - f["qacc"]).max() < 2e-4 * max(1.0, np.abs(f["qacc"]).max()), i
        qe, ve, we, qo, vo, wo = q[None], v[None], None, q.copy(), v.copy(), np.zeros(m.nv)
        ok = True
        for k in range(3):
            qe, ve, we, cnt, _ = pyemu.run(cmod, qe, ve, np.zeros(10), nsub=10, rep=4, warm=we)
            qo, vo, wo, st = o.step(qo, vo, ctrl, 10, wo)
            n_multi += st["ncon"] >= 2
            if cnt["overflow"] or cnt["unhandled"] or st["unhandled_pairs"]:
                ok = False          # a ninth contact on a leg, or the pelvis / torso (root geoms: no collider) reached the floor
                break
            assert np.abs(qe[0] - qo).max() < 2e-4 and np.abs(ve[0] - vo).max() < 2e-2, (i, k, np.abs(qe[0] - qo).max(), np.abs(ve[0] - vo).max())
    assert n_multi >= 4


@pytest.mark.parametrize("rep", [1, 4])
def test_core_self_contacts(rep):
    """Sphere / capsule contacts between two legs of the quadruped (and a leg and the trunk) on the device code: broad phase
    over link bounding spheres, closest points of the capsule segments, contact frame in general position, "mirror" slots in
    the two chain lanes, cross block between the two chains in the Newton factorisation. States of oracle rollouts under
    full-range random torques at the moment two legs touch (tests/golden/a1_self_contact_states.npz: 1..9 self-contacts, up
    to three pairs of legs at once), one forward pass and one control step against the fp64 oracle WITH self-collisions.
    Unpinned: no golden rollout of the reference has a self-contact (the oracle restates mjc_CapsuleCapsule / mjc_SphereCapsule)."""
    np.random.seed(0)
    env = LocoEnv.make("UnitreeA1.simple", debug=True)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    assert info["self_collision_tables"]["geom_pairs"] > 300
    o = Oracle(pack_model(m))
    d = np.load(__file__.replace("test_emu_core.py", "golden/a1_self_contact_states.npz"))
    pick = [0, 26, 65, 104, 143, 195, 208, 247] if rep == 1 else [65, 143, 247]
    seen_multi = 0
    for i in pick:
        q, v, a, w = d["q"][i], d["v"][i], d["a"][i], d["w"][i]
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(a)
        f = o.forward(q, v, ctrl, w)
        nself = sum(1 for c in f["contacts"] if m.geom_type[c["geom1"]] != 0)
        assert nself == d["nself"][i] and f["unhandled_pairs"] == 0
        _, _, _, cnt, dbg = pyemu.run(cmod, q, v, a, nsub=1, debug_env=0, ls_points=4, warm=w, rep=rep)
        assert cnt["selfcon"] == nself and cnt["overflow"] == 0 and cnt["selfprox"] == 0
        assert cnt["ncon"] == f["ncon"] + nself          # a contact between two chains occupies a slot in both lanes
        assert np.abs(dbg["qacc"] - f["qacc"]).max() < 2e-5 * max(1.0, np.abs(f["qacc"]).max()), i
        assert abs(cnt["solver_iters"] - f["solver_iter"]) <= 2        # the cross block makes the Newton step exact
        qo, vo, wo, st = o.step(q, v, ctrl, 10, w)
        qe, ve, we, cnt10, _ = pyemu.run(cmod, q, v, a, nsub=10, ls_points=4, warm=w, rep=rep)
        assert cnt10["overflow"] == 0 and st["unhandled_pairs"] == 0
        assert np.abs(qe[0] - qo).max() < 1e-4 and np.abs(ve[0] - vo).max() < 1e-2, (i, np.abs(qe[0] - qo).max(), np.abs(ve[0] - vo).max())
        seen_multi += d["npairs"][i] > 1
    assert seen_multi >= 2


@pytest.mark.parametrize("rep", [1, 4])
def test_core_convex_pairs_humanoid_bones(rep):
    """Convex-convex self-contacts on the device code (float64 MPR as an out-of-line helper, hill climbing on the hull graph, work queue
    worked off one pair per replica, condim-1 contacts as mu = 0 pyramids, mirror slots + cross block): HumanoidTorque.walk's golden
    rows with the fingers of the left hand on the left femur (rows 19-28), one control step against the golden successor —
    kernel <5 links, 8 slots, RK4, pyramids, PAIRS>."""
    np.random.seed(0)
    env = LocoEnv.make("HumanoidTorque.walk", debug=True)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    t = info["self_collision_tables"]
    assert t["convex"] == 692 and t["native"] == 1 and t["body_pairs"] < 200 and max(t["link_pairs"]) <= 16
    o = Oracle(pack_model(m))
    g = GOLD["HumanoidTorque.walk.real"]
    qidx = [m.jnt_id(n) for k, n, tt in env.obs_helper.observation_spec if k.startswith("q_")]
    np.random.seed(0)
    np.random.randint(0, 1), np.random.randint(0, 1), np.random.randint(0, 100)
    acts = [np.random.randn(13) * 0.1 for _ in range(len(g))]
    for k in ((3, 19, 22, 28) if rep == 1 else (20, 25)):
        qpos, qvel = np.zeros(m.nv), np.zeros(m.nv)
        qpos[qidx[2:]] = g[k, :17]
        qvel[qidx] = g[k, 17:36]
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(acts[k])
        qo, vo, _, st = o.step(qpos, qvel, ctrl, nsub=10)
        q, v, _, cnt, _ = pyemu.run(cmod, qpos, qvel, acts[k], nsub=10, rep=rep)
        assert cnt["overflow"] == 0 and cnt["selfprox"] == 0 and (cnt["selfcon"] > 0) == (k >= 19) and st["convex_contacts"] == cnt["selfcon"]
        assert np.abs(q[0][qidx[2:]] - g[k + 1, :17]).max() < 1e-5 and np.abs(v[0][qidx] - g[k + 1, 17:36]).max() < 1e-3, k
        assert np.abs(q[0] - qo).max() < 1e-5 and np.abs(v[0] - vo).max() < 1e-3


def test_core_convex_pairs_same_chain_and_every_coupling_pattern():
    """(i) UnitreeH1's hip-yaw CYLINDER against the thigh mesh of the SAME leg (two links of one chain: one slot, no mirror, the joints
    up to the nearer link cancel in its Jacobian), condim-3 pyramid in a general frame, golden walk row 13; (ii) a humanoid folded
    up by random torques: bone hulls of the two legs and the trunk in contact with each other in every pattern — two pairs of
    chains, and all three (tests/golden/ht_folded_states.npz, from oracle rollouts) — the factorisation carries a cross block for
    every coupled pair of chains and the fill-in between them (arrow_factor_g), so the Newton iteration counts stay those of the
    oracle instead of hitting the cap."""
    from test_oracle_golden import _h1_kat_inputs
    np.random.seed(0)
    env = LocoEnv.make("UnitreeH1.walk", debug=True)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    assert info["self_collision_tables"]["convex"] == 135
    g, qidx, rows = _h1_kat_inputs(env, "walk")
    qpos, qvel, a = rows[13]
    q, v, _, cnt, _ = pyemu.run(cmod, qpos, qvel, a, nsub=10, rep=4)
    assert cnt["selfcon"] == 10 and cnt["overflow"] == 0                      # one contact in each of the ten substeps
    assert np.abs(q[0][qidx[2:]] - g[14, :15]).max() < 1e-5 and np.abs(v[0][qidx] - g[14, 15:32]).max() < 1e-3

    np.random.seed(0)
    env = LocoEnv.make("HumanoidTorque.run", debug=True)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    o = Oracle(pack_model(m))
    d = np.load(__file__.replace("test_emu_core.py", "golden/ht_folded_states.npz"))
    good = 0
    for i in (0, 3, 7, 8, 9):                      # [(0,2),(1,2)], [(0,1),(1,2)], all three pairs, [(0,1),(0,2)], [(0,1),(1,2)]
        q0, v0, a = d["q"][i], d["v"][i], d["a"][i]
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(a)
        f = o.forward(q0, v0, ctrl)
        _, _, _, cnt, dbg = pyemu.run(cmod, q0, v0, a, nsub=1, debug_env=0)
        assert np.abs(dbg["qacc"] - f["qacc"]).max() < 1e-4 * max(1.0, np.abs(f["qacc"]).max()), i
        qo, vo, _, st = o.step(q0, v0, ctrl, nsub=10)
        qe, ve, _, c10, _ = pyemu.run(cmod, q0, v0, a, nsub=10)
        assert c10["overflow"] == 0 and c10["solver_iters"] < 3 * st["solver_iter_total"] + 40
        good += np.abs(qe[0] - qo).max() < 1e-5 and np.abs(ve[0] - vo).max() < 1e-3
    assert good >= 4


@pytest.mark.parametrize("rep", [1, 4])
def test_core_plane_mesh_unitree_h1(rep):
    """Plane vs convex hull on the device code (one contact at the hull's support vertex; the vertices come from the mesh-vertex
    table, in the replicated layout every replica searches a quarter of the hull): UnitreeH1's golden rows whose only contacts
    are its mesh feet on the floor (walk rows 15-20: heel strike to double support), one control step against the golden
    successor and the oracle."""
    from test_oracle_golden import _h1_kat_inputs
    np.random.seed(0)
    env = LocoEnv.make("UnitreeH1.walk", debug=True)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    assert info["mesh_vertices"] > 5000 and info["max_links"] == 5
    o = Oracle(pack_model(m))
    g, qidx, rows = _h1_kat_inputs(env, "walk")
    for k in ((15, 17, 19, 20) if rep == 1 else (16, 19)):
        qpos, qvel, a = rows[k]
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(a)
        f = o.forward(qpos, qvel, ctrl)
        assert f["ncon"] >= 1 and all(m.geom_type[c["geom2"]] == 5 for c in f["contacts"])        # mesh feet only
        _, _, _, cnt, dbg = pyemu.run(cmod, qpos, qvel, a, nsub=1, debug_env=0, ls_points=4, rep=rep)
        assert cnt["ncon"] == f["ncon"] and cnt["overflow"] == 0, (k, cnt, f["ncon"])
        assert np.abs(dbg["qacc"] - f["qacc"]).max() < 1e-4 * max(1.0, np.abs(f["qacc"]).max())
        q10, v10, _, c10, _ = pyemu.run(cmod, qpos, qvel, a, nsub=10, ls_points=4, rep=rep)
        assert c10["overflow"] == 0
        assert np.abs(q10[0][qidx[2:]] - g[k + 1, :15]).max() < 1e-5 and np.abs(v10[0][qidx] - g[k + 1, 15:32]).max() < 2e-3, (k, np.abs(v10[0][qidx] - g[k + 1, 15:32]).max())


def test_core_model_variants_vs_oracle_compiled_with_the_same_numbers():
    """Inertial / armature / geom-friction randomisation (reference utils/domain_randomization.py:386-514): the kernel reads the
    environment's inertial record and geom table from its model VARIANT; checked against the oracle built from the variant's
    compiled model. Kernel <5,4,Euler,pyramid,DR,4 replicas>, on the CPU."""
    from loco_mujoco_amd.utils.domain_randomization import JointRandomization
    np.random.seed(0)
    env = LocoEnv.make("Talos.walk", debug=True)
    m = env._model
    nominal = env._chain_model()
    jr = JointRandomization(m, os.path.join(os.path.dirname(__file__), "golden", "dr_talos_inertial.yaml"))
    np.random.seed(1)
    variant = jr.sample_model_variant()
    tables = lowering.variant_tables(nominal, env._chain_model(variant))
    tab = env._reset_table()
    rs = np.random.RandomState(1)
    n = 4
    rows = tab[rs.randint(0, len(tab), n)]
    qpos, qvel = rows[:, :m.nv], rows[:, m.nv:2 * m.nv]
    acts = rs.uniform(-0.3, 0.3, (n, 12))

    def oracle_step(model):
        o = Oracle(pack_model(model))
        out = []
        for i in range(n):
            ctrl = np.zeros(model.nu)
            ctrl[env._action_indices] = env._preprocess_action(acts[i])
            q1, v1, _, _ = o.step(qpos[i], qvel[i], ctrl, 10)
            out.append(np.concatenate([q1, v1]))
        return np.array(out)

    ref_nom, ref_var = oracle_step(m), oracle_step(variant)
    qn, vn, _, _, _ = pyemu.run(nominal, qpos, qvel, acts, nsub=10, rep=4, dr=True)
    qv, vv, _, _, _ = pyemu.run(nominal, qpos, qvel, acts, nsub=10, rep=4, variant=tables)
    assert np.abs(qn - ref_nom[:, :m.nv]).max() < 1e-4 and np.abs(vn - ref_nom[:, m.nv:]).max() < 1e-2
    assert np.abs(qv - ref_var[:, :m.nv]).max() < 1e-4 and np.abs(vv - ref_var[:, m.nv:]).max() < 1e-2
    # the variant is a different robot: the two oracles disagree by far more than the tolerance
    assert np.abs(ref_var[:, m.nv:] - ref_nom[:, m.nv:]).max() > 0.1


def test_core_six_link_chains_unitree_g1():
    """Kernel family <6 links, 8 slots, Euler, pyramids> on the CPU: UnitreeG1 with its torso joint welded (legs of 6 joints,
    arms of 5; four 1 mm spheres per foot), replicated layout, one control step vs the oracle."""
    np.random.seed(0)
    env = LocoEnv.make("UnitreeG1.walk", debug=True, disable_back_joint=True)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    assert sorted(len(c) for c in info["chains"]) == [5, 5, 6, 6]
    tab = env._reset_table()
    rs = np.random.RandomState(1)
    n = 2
    rows = tab[rs.randint(0, len(tab), n)]
    acts = rs.uniform(-0.3, 0.3, (n, m.nu))
    o = Oracle(pack_model(m))
    q, v, _, cnt, _ = pyemu.run(cmod, rows[:, :m.nv], rows[:, m.nv:2 * m.nv], acts, nsub=10, rep=4)
    assert cnt["ncon"] > 0 and cnt["overflow"] == 0
    for i in range(n):
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(acts[i])
        qo, vo, _, _ = o.step(rows[i, :m.nv], rows[i, m.nv:2 * m.nv], ctrl, 10)
        assert np.abs(q[i] - qo).max() < 1e-5 and np.abs(v[i] - vo).max() < 1e-3


def test_core_shared_first_link_unitree_g1_default():
    """The default UnitreeG1: the two arm chains share the torso link (owner lane + massless copy, tied together in every
    solve — tie_shared_dof). Two golden rows as one-control-step known-answer tests of the device code on the CPU."""
    np.random.seed(0)
    env = LocoEnv.make("UnitreeG1.walk", debug=True)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    assert info["shared_first"] == {3: 2}
    g = GOLD["UnitreeG1.walk.real"]
    qidx = [m.jnt_id(n) for k_, n, t in env.obs_helper.observation_spec if k_.startswith("q_")]
    np.random.seed(0)
    np.random.randint(0, 1), np.random.randint(0, 1), np.random.randint(0, 100)
    acts = np.array([np.random.randn(23) * 0.1 for _ in range(3)])
    qpos, qvel = np.zeros((3, m.nv)), np.zeros((3, m.nv))
    qpos[:, qidx[2:]] = g[:3, :27]
    qvel[:, qidx] = g[:3, 27:56]
    q, v, _, cnt, _ = pyemu.run(cmod, qpos[1:], qvel[1:], acts[1:], nsub=10, rep=4)
    assert np.abs(q[:, qidx[2:]] - g[2:4, :27]).max() < 1e-5 and np.abs(v[:, qidx] - g[2:4, 27:56]).max() < 1e-3


def test_core_shared_first_link_with_per_environment_joint_parameters():
    """UnitreeG1 (default) in the kernels with per-environment joint parameters: the massless COPY of the shared torso link must
    not apply the torso joint's damping / frictionloss a second time (csrc/lm_core.h DUPK). vs the oracle compiled per environment."""
    import copy
    np.random.seed(0)
    env = LocoEnv.make("UnitreeG1.walk", debug=True)
    m = env._model
    cmod = env._chain_model()
    tab = env._reset_table()
    rs = np.random.RandomState(1)
    n = 2
    rows = tab[rs.randint(0, len(tab), n)]
    acts = rs.uniform(-0.3, 0.3, (n, 23))
    damp = np.tile(m.dof_damping, (n, 1)) * rs.uniform(0.5, 2.0, (n, m.nv))
    floss = np.tile(m.dof_frictionloss, (n, 1)) * rs.uniform(0.5, 1.5, (n, m.nv))
    prm = np.stack([damp, np.tile(m.jnt_stiffness, (n, 1)), floss]).astype(np.float32)
    q, v, _, _, _ = pyemu.run(cmod, rows[:, :m.nv], rows[:, m.nv:2 * m.nv], acts, nsub=10, rep=4, dof_params=prm)
    for i in range(n):
        m2 = copy.copy(m)
        m2.dof_damping, m2.jnt_stiffness, m2.dof_frictionloss = (prm[p][i].astype(np.float64) for p in range(3))
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(acts[i])
        qo, vo = Oracle(pack_model(m2)).step(rows[i, :m.nv], rows[i, m.nv:2 * m.nv], ctrl, 10)[:2]
        qn, vn = Oracle(pack_model(m)).step(rows[i, :m.nv], rows[i, m.nv:2 * m.nv], ctrl, 10)[:2]
        assert np.abs(q[i] - qo).max() < 1e-5 and np.abs(v[i] - vo).max() < 1e-3 and np.abs(vo - vn).max() > 0.1


def test_core_unitree_h1_free_arms_shared_torso_link():
    """UnitreeH1 with `disable_arms=False` (reference unitreeH1.py:235-296: the file as it is — torso joint + two
    4-dof arms): the arm chains share the torso link like UnitreeG1's, the five-link chains run in the six-link kernels with an idle
    link slot. One control step of three reset-table states with random actions against the oracle (self-collisions masked on both
    sides: this family has no pair tables, the device counts proximity)."""
    np.random.seed(0)
    env = LocoEnv.make("UnitreeH1.walk", debug=True, disable_arms=False)
    m = env._model
    assert m.nv == 25 and m.nu == 19 and env.info.action_space.shape == (19,)
    cmod, info = lowering.lower(m, env._device_task())
    assert info["shared_first"] == {3: 2} and info["max_links"] == 6 and info["n_chains"] == 4
    tab = env._reset_table()
    rs = np.random.RandomState(3)
    rows = tab[rs.randint(0, len(tab), 3)]
    acts = rs.uniform(-0.3, 0.3, (3, 19))
    q, v, _, cnt, _ = pyemu.run(cmod, rows[:, :m.nv], rows[:, m.nv:2 * m.nv], acts, nsub=10, rep=4)
    o = Oracle(pack_model(m))
    compared = 0
    for i in range(3):
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(acts[i])
        qo, vo, _, st = o.step(rows[i, :m.nv], rows[i, m.nv:2 * m.nv], ctrl, 10)
        if st["convex_contacts"] or st["unhandled_pairs"]:
            continue
        compared += 1
        assert np.abs(q[i] - qo).max() < 1e-5 and np.abs(v[i] - vo).max() < 1e-3, (i, np.abs(q[i] - qo).max(), np.abs(v[i] - vo).max())
    assert compared >= 2


def test_core_cross_chain_contacts_are_admitted_in_both_lanes_or_in_neither():
    """Two tangled quadruped states with more self-contacts than contact slots (tests/golden/a1_tangled_states.npz, found on the GPU
    with tools/probes/r3/find_nonfinite.py): a contact between two chains recorded in one lane only — the partner lane out of slots
    — is a force without its reaction; the step went to |v| = 289 and to non-finite numbers where the fp64 oracle stays at 12-14 m/s.
    With the admission decided once per environment the device code drops whole contacts and stays with the oracle's magnitudes."""
    np.random.seed(0)
    env = LocoEnv.make("UnitreeA1.simple", debug=True)
    m = env._model
    cmod = env._chain_model()
    o = Oracle(pack_model(m))
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "a1_tangled_states.npz"))
    for i in range(len(d["q"])):
        q0, v0, a = d["q"][i].astype(np.float64), d["v"][i].astype(np.float64), d["a"][i]
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(a)
        qo, vo = o.step(q0, v0, ctrl, 10)[:2]
        # the regular instantiation alone (replay off): the states DO drop contacts (six slots per leg), whole ones
        q, v, _, cnt, _ = pyemu.run(cmod, q0, v0, a, nsub=10, rep=4, replay=False)
        assert cnt["overflow"] > 0 and cnt["replayed"] == 0
        assert np.isfinite(q).all() and np.isfinite(v).all() and np.abs(v).max() < 1.5 * np.abs(vo).max()
        # speculate / replay (round 4, what the library does): the control step is abandoned and run by the big instantiation —
        # a slot for every contact: nothing dropped, and the result is the oracle's within the stated tolerance
        q, v, _, cnt, _ = pyemu.run(cmod, q0, v0, a, nsub=10, rep=4)
        assert cnt["overflow"] == 0 and cnt["replayed"] == 1
        assert np.abs(q[0] - qo).max() < 1e-4 and np.abs(v[0] - vo).max() < 1e-2, (np.abs(q[0] - qo).max(), np.abs(v[0] - vo).max())


@pytest.mark.parametrize("robot", ["a1", "ht"])
def test_core_native_box_and_cylinder_pairs_vs_oracle(robot):
    """The engine's native colliders for box / cylinder pairs on the device code (lm_core.h nat_*: sphere-box, sphere-cylinder,
    capsule-box of the quadruped's trunk boxes and hip cylinders against its legs, box-box of the humanoid's feet) against the
    oracle's float64 restatement of the same constructions: states of tests/golden/native_pair_states.npz (tools/make_native_fixtures.py:
    found in oracle rollouts, and sampled configurations in the air), one control step. Nothing is merely counted any more."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "native_pair_states.npz"))
    np.random.seed(0)
    env = LocoEnv.make("UnitreeA1.simple" if robot == "a1" else "HumanoidTorque.run", debug=True)
    m = env._model
    cmod = env._chain_model()
    o = Oracle(pack_model(m))
    q0, v0, a0 = d[robot + "_q"], d[robot + "_v"], d[robot + "_a"]
    pick = list(range(0, len(q0), 6 if robot == "a1" else 7))           # a sixth / a seventh of the fixture here (the GPU test runs all of it)
    worst_q = worst_v = 0.0
    replayed = selfcon = 0
    for i in pick:
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(a0[i])
        qo, vo, _, st = o.step(q0[i], v0[i], ctrl, 10)
        assert st["native_contacts"] > 0 and st["unhandled_pairs"] == 0
        q, v, _, cnt, _ = pyemu.run(cmod, q0[i], v0[i], a0[i], nsub=10, rep=4)
        assert cnt["selfprox"] == 0 and cnt["overflow"] == 0 and cnt["selfcon"] > 0, (i, cnt)
        worst_q, worst_v = max(worst_q, np.abs(q[0] - qo).max()), max(worst_v, np.abs(v[0] - vo).max())
        replayed += cnt["replayed"]; selfcon += cnt["selfcon"]
        assert np.abs(q[0] - qo).max() < 1e-4 and np.abs(v[0] - vo).max() < 1e-2, (robot, i, np.abs(q[0] - qo).max(), np.abs(v[0] - vo).max())
    print("native pairs (%s): %d states, qpos max %.2e qvel max %.2e, self-contacts %d, replayed %d" % (robot, len(pick), worst_q, worst_v, selfcon, replayed))


@pytest.mark.parametrize("robot", ["g1", "h1arms"])
def test_core_six_link_self_collisions_detect_and_replay(robot):
    """self-collisions of the six-link family (UnitreeG1 default, UnitreeH1 with its arms; reference
    humanoids/unitreeG1.py:246, unitreeH1.py:235): the whole pair pass for six-link chains - link-pair lists of up to 128 entries per
    lane read from global memory, cross blocks between all four chains, the shared torso link tied in the coupled solves, a pair of the
    torso with the arm that carries its copy as a pair of ONE lane. (A detection-only variant of the regular kernels, a contact handing
    the control step to the replay kernel, is kept as the switch LM_SIX_PAIRS = 3 of lm_family.hip / EMU_SIX_PAIRS here.)
    States of tests/golden/six_link_self_contact_states.npz (oracle rollouts of
    stumbling robots, arm-on-arm poses: tools/make_six_link_fixtures.py) vs the fp64 oracle; the well-conditioned ones at the stated
    tolerance, the others (the oracle itself moves under float32-sized input noise) at three times the oracle's own spread."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "six_link_self_contact_states.npz"))
    np.random.seed(0)
    env = LocoEnv.make("UnitreeG1.walk", debug=True) if robot == "g1" else LocoEnv.make("UnitreeH1.walk", debug=True, disable_arms=False)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    assert info["max_links"] == 6 and info["self_collision_tables"]["convex"] > 100 and int(cmod[lowering.H_NGPAIR]) > 0
    assert int(cmod[lowering.H_CM_USED]) == int(cmod[lowering.H_OFF_PRUNE]) < int(cmod[lowering.H_OFF_LPAIR])     # prune records, link groups and link-pair lists stay out of the LDS copy (round 6: four workgroups per CU)
    o = Oracle(pack_model(m))
    q0, v0, a0, spread = d[robot + "_qpos"], d[robot + "_qvel"], d[robot + "_action"], d[robot + "_oracle_spread"]
    pick = list(range(0, len(q0), 3 if robot == "g1" else 2)) + ([len(q0) - 1, len(q0) - 3] if robot == "g1" else [])       # (the GPU test runs all of them; the last four of g1: arm on arm)
    q, v, _, cnt, _ = pyemu.run(cmod, q0[pick], v0[pick], a0[pick], nsub=10, rep=4)
    assert cnt["overflow"] == 0 and cnt["selfcon"] > 0, cnt      # (LM_SIX_PAIRS 1: the regular instantiation has the pair pass; 3: every one of them replayed)
    held = 0
    for j, i in enumerate(pick):
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(a0[i])
        qo, vo, _, st = o.step(q0[i], v0[i], ctrl, 10)
        assert st["convex_contacts"] > 0
        eq, ev = np.abs(q[j] - qo).max(), np.abs(v[j] - vo).max()
        well = spread[i, 0] < 1e-5 and spread[i, 1] < 1e-3
        held += well
        assert (eq < 1e-4 and ev < 1e-2) if well else (eq < 3 * spread[i, 0] + 1e-4 and ev < 3 * spread[i, 1] + 1e-2), (robot, i, eq, ev, spread[i])
    assert held >= 2
    # an upright gait does not leave the regular instantiation: the hulls of neighbouring links sit inside each other's bounding
    # capsules for good, the colliders say "no contact"
    tab = env._reset_table()
    rows = tab[np.random.RandomState(0).randint(0, len(tab), 4)]
    _, _, _, cnt, _ = pyemu.run(cmod, rows[:, :m.nv], rows[:, m.nv:2 * m.nv], np.zeros((4, len(env._action_indices))), nsub=10, rep=4)
    assert cnt["replayed"] == 0 and cnt["selfcon"] == 0, cnt


def _root_limit_states(env):
    """HumanoidMuscle in the air (no contacts) with pelvis rotations beyond their joint limits (humanoid_muscle.xml: the pelvis joints
    are `limited`, +-pi/2): limit rows on the replicated root dofs."""
    m = env._model
    tab = env._reset_table()
    rs = np.random.RandomState(0)
    rows = tab[rs.randint(0, len(tab), 4)].copy()
    q, v = rows[:, :m.nv].copy(), rows[:, m.nv:2 * m.nv].copy()
    q[:, 1] += 1.0                         # pelvis_ty
    q[0, 3], v[0, 3] = 1.62, 0.5           # pelvis_tilt beyond its upper limit, still moving out
    q[1, 5], v[1, 5] = -1.60, -1.0         # pelvis_rotation beyond its lower limit
    q[2, 3], v[2, 3] = -1.60, -2.0         # pelvis_tilt, the other side
    q[3, 3], q[3, 5] = -1.65, 1.63         # two at once
    return q, v, rs.uniform(-1, 1, (4, len(env._action_indices)))


def test_core_root_dof_limit_rows():
    """limit rows on the (replicated) root dofs — compiled into the muscle families (lm_core.h ROOT_LIM);
    HumanoidMuscle's pelvis joints are limited (reference data/humanoid/humanoid_muscle.xml), the lowering used to drop them after
    proving that they cannot become active. States with the pelvis beyond its limits, one control step vs the oracle; the rows act
    (the tilt velocity is reversed) and every state is inside the stated tolerance, plain and replicated layout."""
    np.random.seed(0)
    env = LocoEnv.make("HumanoidMuscle.run", debug=True)
    m = env._model
    cmod, info = lowering.lower(m, env._device_task())
    assert info["dropped_root_limits"] == []
    o = Oracle(pack_model(m))
    q, v, acts = _root_limit_states(env)
    for rep in (1, 4):
        qe, ve, _, cnt, _ = pyemu.run(cmod, q, v, acts, nsub=10, rep=rep)
        for i in range(4):
            ctrl = np.zeros(m.nu)
            ctrl[env._action_indices] = env._preprocess_action(acts[i])
            qo, vo, _, _, _ = o.step_act(q[i], v[i], np.zeros(m.na), ctrl, 10)
            assert np.abs(qe[i] - qo).max() < 1e-5 and np.abs(ve[i] - vo).max() < 1e-3, (rep, i)
        assert ve[0, 3] < -0.3 and v[0, 3] > 0       # the limit row turned the tilt around


def _a1_with_limited_root_tz():
    """A synthetic quadruped whose root joint trunk_tz is `limited` to [-0.5, -0.18] (the reference's file has it unlimited,
    unitree_a1_torque.xml:80-85): the dataset states stand at -0.168, i.e. BEYOND the upper limit — the limit row pushes the trunk down.
    The termination band of trunk_tz (> -0.24) does not lie inside the range, so the lowering must keep the limit."""
    np.random.seed(0)
    env = LocoEnv.make("UnitreeA1.simple", debug=True)
    m = env._model
    m.jnt_limited[2] = 1
    m.jnt_range[2] = (-0.5, -0.18)
    cmod, info = lowering.lower(m, env._device_task())
    assert info["dropped_root_limits"] == [] and cmod[lowering.HEADER_SIZE + lowering.R_DOFS + 2 * lowering.D_SIZE + lowering.D_LIMITED] == 1.0
    tab = env._reset_table()
    rs = np.random.RandomState(1)
    rows = tab[rs.randint(0, len(tab), 4)].copy()
    q, v = rows[:, :m.nv].copy(), rows[:, m.nv:2 * m.nv].copy()
    q[3, 2] = -0.21                      # one state INSIDE the range: no row, no hand-over
    return env, m, cmod, q, v, rs.uniform(-0.3, 0.3, (4, 12))


def test_core_root_limit_rows_in_every_family_through_the_replay_instantiation():
    """a limited root joint is accepted for EVERY kernel family. The regular instantiations of the families without
    muscles only look (lm_core.h ROOT_LIM): a root dof beyond its range hands the control step to the replay instantiation, which
    carries the limit rows — device code vs the fp64 oracle compiled from the same model, regular + replay like the library."""
    env, m, cmod, q, v, acts = _a1_with_limited_root_tz()
    o = Oracle(pack_model(m))
    qe, ve, _, cnt, _ = pyemu.run(cmod, q, v, acts, nsub=10, rep=4)
    assert cnt["replayed"] == 3 and cnt["overflow"] == 0, cnt          # the three states beyond the limit; the fourth stays in the regular instantiation
    for i in range(4):
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(acts[i])
        qo, vo, _, _ = o.step(q[i], v[i], ctrl, 10)
        # (the fourth state was pushed 4 cm into the floor to get inside the range: stiff contacts, the stated tolerance)
        tq, tv = (1e-5, 1e-3) if i < 3 else (1e-4, 1e-2)
        assert np.abs(qe[i] - qo).max() < tq and np.abs(ve[i] - vo).max() < tv, (i, np.abs(qe[i] - qo).max(), np.abs(ve[i] - vo).max())
    # the row acts: without it (the unlimited model) the trunk ends higher
    np.random.seed(0)
    env0 = LocoEnv.make("UnitreeA1.simple", debug=True)
    q0, _, _, cnt0, _ = pyemu.run(env0._chain_model(), q, v, acts, nsub=10, rep=4)
    assert cnt0["replayed"] == 0 and (qe[:3, 2] < q0[:3, 2] - 2e-4).all() and abs(qe[3, 2] - q0[3, 2]) < 1e-6


def test_core_wide_replay_instantiation_sixteen_replicas():
    """Round 5: the replay kernels run ONE environment per wave on SIXTEEN replicas (lm_step.h kReplayRep; 64 emulator threads here):
    contact slots, floor geoms, hull vertices and the work lists of the pair pass are dealt to 64 lanes, ballots are 64 bits wide,
    replica sums are four-stage butterflies. Tangled quadrupeds (more self-contacts than the regular kernel's slots) and humanoid states
    with box / hull pairs in contact, every control step through the big instantiation: within the stated tolerance of the fp64 oracle,
    within rounding of the four-replica run of the same instantiation, and no lane-memory word written differently by two replicas."""
    np.random.seed(0)
    env = LocoEnv.make("UnitreeA1.simple", debug=True)
    m, cmod = env._model, env._chain_model()
    o = Oracle(pack_model(m))
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "a1_tangled_states.npz"))
    for i in range(1):                  # (the second state of the fixture takes 73 Newton iterations: left to the four-replica test above)
        q0, v0, a = d["q"][i].astype(np.float64), d["v"][i].astype(np.float64), d["a"][i]
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(a)
        qo, vo = o.step(q0, v0, ctrl, 10)[:2]
        q4, v4, _, c4, _ = pyemu.run(cmod, q0, v0, a, nsub=10, rep=4, replay=2)
        q16, v16, _, c16, _ = pyemu.run(cmod, q0, v0, a, nsub=10, rep=16, replay=2)
        assert c16["overflow"] == 0 and c16["ncon"] == c4["ncon"] and c16["selfcon"] == c4["selfcon"], (c4, c16)
        assert np.abs(q16[0] - qo).max() < 1e-4 and np.abs(v16[0] - vo).max() < 1e-2
        assert np.abs(q16 - q4).max() < 2e-5 and np.abs(v16 - v4).max() < 2e-3
    np.random.seed(0)
    env = LocoEnv.make("HumanoidTorque.run", debug=True)
    m, cmod = env._model, env._chain_model()
    o = Oracle(pack_model(m))
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "native_pair_states.npz"))
    for i in (9,):
        q0, v0, a = d["ht_q"][i], d["ht_v"][i], d["ht_a"][i]
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(a)
        qo, vo, _, st = o.step(q0, v0, ctrl, 10)
        q16, v16, _, c16, _ = pyemu.run(cmod, q0, v0, a, nsub=10, rep=16, replay=2)
        assert st["convex_contacts"] > 100 and c16["selfcon"] > 100 and c16["overflow"] == 0 and c16["selfprox"] == 0
        assert np.abs(q16[0] - qo).max() < 1e-4 and np.abs(v16[0] - vo).max() < 1e-2
