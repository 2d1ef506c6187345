"""GPU probe (LM_TIMERS build): cycles per solver region for any task.  usage: region_timers.py <task> <n_envs> <action_mode>"""
import os, sys, json, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv, backend
from loco_mujoco_amd.backend import HipBatch, HipModel
task, N, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
np.random.seed(0)
env = LocoEnv.make(task, debug=True)
hm = HipModel(env._chain_model()); tab = env._reset_table()
nv = env._model.nv
rs = np.random.RandomState(0)
rows = tab[rs.randint(0, len(tab), N)]
b = HipBatch(hm, N)
b.set_reset_table(tab, seed=0); b.set_auto_reset(True, horizon=1000)
b.set_state(rows[:, :nv], rows[:, nv:2 * nv])
if rows.shape[1] > 2 * nv: b.set_goal(rows[:, 2 * nv:])
b.rollout(20, action_mode=mode, seed=3)
lib = backend.load_library()
buf = (ctypes.c_ulonglong * 16)()
lib.lm_debug_timers.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.lm_debug_timers(b._h, buf)
st = b.rollout(50, action_mode=mode, seed=4)
lib.lm_debug_timers(b._h, buf)
t = np.array(list(buf)[:16], dtype=np.float64)
names = ["self-collision pairs + slot bookkeeping", "M+bias", "rows+a0", "warmstart", "gradient", "hessian", "factor+solve", "jv/Mv", "linesearch", "integrate", "lockstep wait",
         "kinematics", "floor: broad phase + primitives", "floor: hulls", "pairs: tests", "pairs: MPR"]
tot = t.sum()
print(json.dumps(dict(task=task, n=N, ms_per_step=st["kernel_ms"] / 50, iters_per_forward=st["solver_iters"] / st["env_steps"] / (40 if env._model.integrator else 10),
                      share={n: round(v / tot, 3) for n, v in zip(names, t)})))
