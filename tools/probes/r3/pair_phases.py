import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from loco_mujoco_amd import LocoEnv, backend
from loco_mujoco_amd.backend import HipBatch, HipModel
task, N, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
np.random.seed(0)
env = LocoEnv.make(task, debug=True)
hm = HipModel(env._chain_model()); tab = env._reset_table(); nv = env._model.nv
rows = tab[np.random.RandomState(0).randint(0, len(tab), N)]
b = HipBatch(hm, N)
b.set_reset_table(tab, seed=0); b.set_auto_reset(True, horizon=1000)
b.set_state(rows[:, :nv], rows[:, nv:2 * nv])
b.rollout(40, action_mode=mode, seed=3)
lib = backend.load_library()
m8 = (ctypes.c_ulonglong * 16)()
lib.lm_debug_mpr_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.lm_debug_mpr_counters(b._h, m8)
st = b.rollout(20, action_mode=mode, seed=5)
lib.lm_debug_mpr_counters(b._h, m8)
m = np.array(list(m8), dtype=np.float64)
ph = m[10:15]; print('  stage 1a (sphere tests, dealt to the replicas, + mask exchange): %.0f' % (m[15] / max(m[8] * 16, 1)))
# cycles are summed over ALL lanes that ran the block: per lane-detection
print(task, "ms/step %.3f" % (st["kernel_ms"] / 20), "detections (env) %d of %d passes" % (m[8], m[9]))
print("  cycles per detecting lane: stage1 (link pairs) %.0f  stage2 (body pairs) %.0f  stage3 (geom pairs) + next chunk's stage 1 %.0f  queue %.0f  results+gaps %.0f" % tuple(ph / max(m[8] * 16, 1)))
