"""GPU probe: HumanoidTorque.run golden rows through b.step for several batch sizes, against the oracle (CPU)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel
from oracle.model_blob import pack_model
from oracle.pyoracle import Oracle
task = sys.argv[1] if len(sys.argv) > 1 else "HumanoidTorque.run"
np.random.seed(0)
env = LocoEnv.make(task, debug=True)
m = env._model
hm = HipModel(env._chain_model())
tab = env._reset_table()
o = Oracle(pack_model(m))
nv = m.nv
for n in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,5,16").split(",")]:
    rows = tab[(np.arange(n) * 37) % len(tab)]
    q0 = rows[:, :nv].astype(np.float32).astype(np.float64); v0 = rows[:, nv:2 * nv].astype(np.float32).astype(np.float64)
    b = HipBatch(hm, n)
    b.set_state(q0, v0)
    b.step(np.zeros((n, m.nu if not hasattr(env, "_action_indices") else len(env._action_indices))))
    q, v = b.get_state()
    st = b.stats()
    errs = []
    for i in range(n):
        qo, vo = o.step(q0[i], v0[i], np.zeros(m.nu), nsub=10)[:2]
        errs.append(np.abs(q[i] - qo).max())
    print(task, "n=%d" % n, "qpos err per env:", " ".join("%.1e" % e for e in errs[:16]), "| self_contacts %d overflow %d prox %d" % (st["self_contacts"], st["overflow_contacts"], st["self_proximity"]))
