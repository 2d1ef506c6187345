"""GPU probe: time the bench workload for one library build and several envs-per-block values."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel
np.random.seed(0)
env = LocoEnv.make("UnitreeA1.simple", debug=True)
hm = HipModel(env._chain_model())
tab = env._reset_table()
N = int(os.environ.get("N_ENVS", 4096))
rs = np.random.RandomState(0)
rows = tab[rs.randint(0, 3, N) * 100 + rs.randint(0, 100, N)]
res = {}
for epb in [int(x) for x in sys.argv[1:]] or [16, 4, 2, 1]:
    os.environ["LM_ENVS_PER_BLOCK"] = str(epb)
    b = HipBatch(hm, N)
    b.set_reset_table(tab, seed=0)
    b.set_auto_reset(True, horizon=1000)
    b.set_state(rows[:, :18], rows[:, 18:36]); b.set_goal(rows[:, 36:39])
    b.rollout(20)
    st = b.rollout(100)
    res[epb] = dict(ms_per_step=round(st["kernel_ms"] / 100, 4), env_steps_per_s=round(N * 100 / (st["kernel_ms"] * 1e-3)), iters=st["solver_iters"] / st["env_steps"] / 10)
print(os.environ.get("LOCOHIP_LIB", "default"), json.dumps(res))
