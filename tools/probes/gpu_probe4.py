"""GPU probe: kernel time vs Newton iteration cap (ablation of the solver cost)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv, lowering
from loco_mujoco_amd.backend import HipBatch, HipModel
np.random.seed(0)
env = LocoEnv.make("UnitreeA1.simple", debug=True)
tab = env._reset_table()
N = 4096
rs = np.random.RandomState(0)
rows = tab[rs.randint(0, 3, N) * 100 + rs.randint(0, 100, N)]
res = {}
for cap in [0, 1, 2, 3, 5, 100]:
    cmod = env._chain_model().copy()
    cmod[lowering.H_ITERATIONS] = cap
    hm = HipModel(cmod)
    b = HipBatch(hm, N)
    b.set_reset_table(tab, seed=0)
    b.set_auto_reset(True, horizon=1000)
    b.set_state(rows[:, :18], rows[:, 18:36]); b.set_goal(rows[:, 36:39])
    b.rollout(20)
    st = b.rollout(50)
    res[cap] = dict(ms_per_step=round(st["kernel_ms"] / 50, 4), iters=round(st["solver_iters"] / st["env_steps"] / 10, 3))
print(json.dumps(res))
