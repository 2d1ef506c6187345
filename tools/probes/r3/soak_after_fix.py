import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel
for task in ("UnitreeA1.simple", "HumanoidTorque.run", "UnitreeH1.run"):
    np.random.seed(0)
    env = LocoEnv.make(task, debug=True); m = env._model; tab = env._reset_table(); n = 4096
    b = HipBatch(HipModel(env._chain_model()), n)
    rows = tab[np.random.RandomState(0).randint(0, len(tab), n)]
    b.set_reset_table(tab, seed=0); b.set_auto_reset(True, horizon=1000)
    b.set_state(rows[:, :m.nv], rows[:, m.nv:2 * m.nv])
    if rows.shape[1] > 2 * m.nv: b.set_goal(rows[:, 2 * m.nv:])
    steps = 6000 if task == "UnitreeA1.simple" else 1500
    st = b.rollout(steps, action_mode=1, seed=1, steps_per_launch=25)
    q, v = b.get_state()
    print(task, steps, "steps: nan_resets", st["nan_resets"], "overflow", st["overflow_contacts"], "self-contacts", st["self_contacts"], "finite", bool(np.isfinite(q).all() and np.isfinite(v).all()))
