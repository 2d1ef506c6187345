"""GPU probe (LM_TIMERS build): wall-clock time line of single launches — when does the regular kernel end, when is a control step
handed to the replay kernel, when is it taken, when is it done, at which substep does it resume?
usage: LOCOHIP_LIB=<timers lib> timeline.py <task> [launches]"""
import os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv, backend
from loco_mujoco_amd.backend import HipBatch, HipModel
task = sys.argv[1]; L = int(sys.argv[2]) if len(sys.argv) > 2 else 12
np.random.seed(0)
env = LocoEnv.make(task, debug=True)
hm = HipModel(env._chain_model()); tab = env._reset_table(); nv = env._model.nv
N = 4096
rows = tab[np.random.RandomState(0).randint(0, len(tab), N)]
b = HipBatch(hm, N)
b.set_reset_table(tab, seed=0); b.set_auto_reset(True, horizon=1000)
b.set_state(rows[:, :nv], rows[:, nv:2 * nv])
if rows.shape[1] > 2 * nv: b.set_goal(rows[:, 2 * nv:])
b.rollout(40, action_mode=1, seed=3)
lib = backend.load_library()
nb = (N + 3) // 4
buf = (ctypes.c_ulonglong * (4 * N + 2 * nb))()
lib.lm_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.lm_debug_timeline(b._h, buf)
for k in range(L):
    st = b.rollout(1, action_mode=1, seed=100 + k)
    lib.lm_debug_timeline(b._h, buf)
    t = np.array(list(buf), dtype=np.float64)
    env_t, wg = t[:4 * N].reshape(N, 4), t[4 * N:].reshape(nb, 2)
    t0 = wg[:, 0][wg[:, 0] > 0].min()
    ms = lambda x: (x - t0) * 1e-5          # 100 MHz ticks -> ms
    rep = np.nonzero(env_t[:, 0] > 0)[0]
    print("launch %d: kernel_ms %.2f | regular workgroups: first start 0, last start %.2f, end p50 %.2f p99 %.2f last %.2f | replayed %d"
          % (k, st["kernel_ms"], ms(wg[:, 0].max()), ms(np.percentile(wg[:, 1], 50)), ms(np.percentile(wg[:, 1], 99)), ms(wg[:, 1].max()), len(rep)))
    order = rep[np.argsort(-env_t[rep, 2])][:6]
    for e in order:
        print("    env %4d resumes at substep %d: listed %.2f taken %.2f (+%.2f) done %.2f (%.2f in the replay kernel)" % (
            e, env_t[e, 3], ms(env_t[e, 0]), ms(env_t[e, 1]), ms(env_t[e, 1]) - ms(env_t[e, 0]), ms(env_t[e, 2]), ms(env_t[e, 2]) - ms(env_t[e, 1])))
