"""GPU probe: substep-by-substep dump (n_substeps=1 model) for the random-state parity case."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv, lowering
from loco_mujoco_amd.backend import HipBatch, HipModel
np.random.seed(0)
env = LocoEnv.make("UnitreeA1.simple", debug=True)
task = env._device_task(); task["n_substeps"] = 1
hm = HipModel(lowering.lower(env._model, task)[0])
tab = env._reset_table()
rs = np.random.RandomState(1)
n = 256
rows = tab[rs.randint(0, len(tab), n)]
qpos = rows[:, :18] + rs.uniform(-0.03, 0.03, (n, 18))
qpos[:, 2] -= rs.uniform(0, 0.03, n)
qvel = rows[:, 18:36] * rs.uniform(0.5, 1.0, (n, 1))
acts = rs.uniform(-1, 1, (n, 12))
b = HipBatch(hm, n)
b.set_state(qpos, qvel)
out = dict(qpos=qpos, qvel=qvel, acts=acts)
for s in range(10):
    d = b.forward_debug(acts)
    for k in ("M", "qfrc_bias", "qfrc_smooth", "qacc_smooth", "qacc", "qfrc_constraint", "ncon", "solver_iter"):
        out["%s_%d" % (k, s)] = d[k]
    b.step(acts)
    q, v = b.get_state()
    out["q_%d" % s], out["v_%d" % s] = q, v
np.savez(os.path.join(ROOT, "gpurun_out", "probe2.npz"), **out)
print("done")
