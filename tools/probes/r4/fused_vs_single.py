"""fused launches vs single-step launches, regular kernels alone (replay off) and with replay: per task the number of environments whose
state differs after k control steps, and the first step at which any differs."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel
for task in sys.argv[1:] or ["HumanoidTorque.run", "UnitreeH1.run", "HumanoidMuscle.run", "Atlas.walk", "Talos.walk", "UnitreeG1.walk", "UnitreeA1.simple"]:
    np.random.seed(0)
    env = LocoEnv.make(task, debug=True)
    tab = env._reset_table(); nv = env._model.nv
    hm = HipModel(env._chain_model())
    n = 512
    rows = tab[np.random.RandomState(0).randint(0, len(tab), n)]
    for mode in (0, 1):
        for k in (2, 5, 12):
            out = []
            for fuse in (1, k):
                b = HipBatch(hm, n); b.set_replay(mode)
                b.set_reset_table(tab, seed=1); b.set_auto_reset(True, horizon=1000)
                b.set_state(rows[:, :nv], rows[:, nv:2 * nv])
                if rows.shape[1] > 2 * nv: b.set_goal(rows[:, 2 * nv:])
                st = b.rollout(k, action_mode=1, seed=5, steps_per_launch=fuse)
                q, v = b.get_state()
                out.append((q, v, st, b.replay_marks()))
            (q1, v1, s1, m1), (q2, v2, s2, m2) = out
            diff = (np.abs(q1 - q2).max(axis=1) > 0) | (np.abs(v1 - v2).max(axis=1) > 0)
            unmarked = ~(m1 | m2)
            print("%s replay %d, %d steps fused vs single: %d of %d envs differ (%d of them never replayed), max |dq| %.3g; nan %d / %d, overflow %d / %d, episodes %d / %d"
                  % (task, mode, k, diff.sum(), n, (diff & unmarked).sum(), np.abs(q1 - q2).max(), s1["nan_resets"], s2["nan_resets"], s1["overflow_contacts"], s2["overflow_contacts"],
                     s1["episodes"], s2["episodes"]), flush=True)
