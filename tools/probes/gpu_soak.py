"""Soak: long random-policy rollouts of every robot at 4096 environments with device-side restarts (and parameter redraws
for the Atlas randomisation config); counts non-finite resets, dropped contacts, proximity flags, slow solves."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import loco_mujoco_amd
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
yaml = os.path.join(os.path.dirname(loco_mujoco_amd.__file__), "environments", "data", "atlas", "domain_randomization_atlas.yaml")
for task, kw in (("UnitreeA1.simple", {}), ("UnitreeA1.simple", dict(action_mode="position")), ("HumanoidTorque.run", {}),
                 ("HumanoidMuscle.walk", {}), ("Atlas.walk", dict(disable_back_joint=False, domain_randomization_config=yaml)),
                 ("Talos.walk", {}), ("Atlas.carry", dict(weight_mass=10.0)), ("Talos.carry", dict(weight_mass=0.1)),
                 ("HumanoidTorque4Ages.walk.1", {}), ("HumanoidMuscle4Ages.run.3", {}), ("UnitreeH1.run", {}), ("UnitreeG1.walk", {}),
                 ("UnitreeG1.run", dict(disable_back_joint=True)),
                 ("Talos.walk", dict(domain_randomization_config=os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "dr_talos_inertial.yaml")))):
    np.random.seed(0)
    env = LocoEnv.make(task, debug=True, **kw)
    m = env._model
    tab = env._reset_table()
    n = 4096
    nominal = env._chain_model()
    b = HipBatch(HipModel(nominal), n)
    if kw.get("domain_randomization_config") and env._domain_rand.has_model_rules:
        b.set_model_variants(env._build_model_variants(nominal)[1])
        b.set_variant_index(np.random.RandomState(1).randint(0, b.n_variants, n))
    rows = tab[np.random.RandomState(0).randint(0, len(tab), n)]
    b.set_reset_table(tab, seed=0)
    b.set_auto_reset(True, horizon=1000)
    b.set_state(rows[:, :m.nv], rows[:, m.nv:2 * m.nv])
    if rows.shape[1] > 2 * m.nv:
        b.set_goal(rows[:, 2 * m.nv:])
    if kw.get("domain_randomization_config"):
        d = env._domain_rand.sample(n)
        b.set_dof_params(damping=d[0], stiffness=d[1], frictionloss=d[2])
        b.set_dof_randomization(env._domain_rand.spec)
    t0 = time.perf_counter()
    st = b.rollout(STEPS, action_mode=1, seed=1, steps_per_launch=25)
    dt = time.perf_counter() - t0
    q, v = b.get_state()
    print("%-28s %s: %d steps x 4096 envs in %.1f s (%.2f M env-steps/s): episodes %d, nan_resets %d, overflow %d, proximity flags %d, "
          "self-contacts %d, pairs without collider in reach %d, control steps with >= 8 iterations %.3f %%, finite %s"
          % (task, "position" if kw.get("action_mode") else ("dr" if kw.get("domain_randomization_config") else ""), STEPS, dt,
             n * STEPS / dt / 1e6, st["episodes"], st["nan_resets"], st["overflow_contacts"], st["unhandled_geoms"], st["self_contacts"], st["self_proximity"],
             100.0 * st["steps_with_8plus_iters"] / (n * STEPS), bool(np.isfinite(q).all() and np.isfinite(v).all())), flush=True)
