"""GPU probe (LM_TIMERS build): cycles per solver region on the bench workload."""
import os, sys, json, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv, backend
from loco_mujoco_amd.backend import HipBatch, HipModel
np.random.seed(0)
env = LocoEnv.make("UnitreeA1.simple", debug=True)
hm = HipModel(env._chain_model()); tab = env._reset_table()
N = 4096
rs = np.random.RandomState(0)
rows = tab[rs.randint(0, 3, N) * 100 + rs.randint(0, 100, N)]
b = HipBatch(hm, N)
b.set_reset_table(tab, seed=0); b.set_auto_reset(True, horizon=1000)
b.set_state(rows[:, :18], rows[:, 18:36]); b.set_goal(rows[:, 36:39])
b.rollout(20)
lib = backend.load_library()
buf = (ctypes.c_ulonglong * 16)()
lib.lm_debug_timers.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.lm_debug_timers(b._h, buf)
st = b.rollout(100)
lib.lm_debug_timers(b._h, buf)
t = np.array(list(buf)[:11], dtype=np.float64)
names = ["kinematics+contacts", "M+bias", "rows+a0", "warmstart", "gradient", "hessian", "factor+solve", "jv/Mv", "linesearch", "integrate", "lockstep wait"]
nblocks = (N + b_epb - 1) // b_epb if (b_epb := int(os.environ.get("LM_ENVS_PER_BLOCK", 4))) else 0
tot = t.sum()
print(json.dumps(dict(ms_per_step=st["kernel_ms"] / 100, iters=st["solver_iters"] / st["env_steps"] / 10, ls_per_iter=st["linesearch_evals"] / st["solver_iters"],
                      cycles_per_wave_step={n: round(v / 100 / 1024) for n, v in zip(names, t)}, share={n: round(v / tot, 3) for n, v in zip(names, t)},
                      total_cycles_per_wave_step=round(tot / 100 / 1024))))
