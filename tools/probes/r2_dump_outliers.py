"""GPU probe (round 2): the states of a parity scenario in which the device leaves the fp64 oracle by the most, with the
per-substep state history of the device, written to gpurun_out/ for a post-mortem on the CPU (emulator vs oracle).
  python tools/probes/r2_dump_outliers.py Atlas.walk 3steps | HumanoidMuscle.run 4096"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel
from oracle.model_blob import pack_model
from oracle.pyoracle import Oracle

task, mode = sys.argv[1], sys.argv[2]
np.random.seed(0)
env = LocoEnv.make(task, debug=True)
m = env._model
oracle = Oracle(pack_model(m))
tab = env._reset_table()
nu = len(env._action_indices)
if mode == "3steps":
    n = 128
    rs = np.random.RandomState(7)
    rows = tab[rs.randint(0, len(tab), n)]
    acts = rs.uniform(-1, 1, (3, n, nu))
    q0, v0, a0 = rows[:, :m.nv], rows[:, m.nv:2 * m.nv], np.zeros((n, m.na))
else:
    n = 4096
    rs = np.random.RandomState(2024)
    rows = tab[rs.randint(0, len(tab), n)]
    b = HipBatch(HipModel(env._chain_model()), n)
    b.set_state(rows[:, :m.nv], rows[:, m.nv:2 * m.nv])
    for _ in range(12):
        b.step(rs.uniform(-1, 1, (n, nu)))
    q0, v0 = b.get_state()
    a0 = b.get_activation() if m.na else np.zeros((n, 0))
    acts = rs.uniform(-1, 1, (1, n, nu)).astype(np.float32)
b = HipBatch(HipModel(env._chain_model()), n)
b.set_state(q0, v0)
if m.na:
    b.set_activation(a0)
hist = []
for k in range(len(acts)):
    b.step(acts[k])
    q, v = b.get_state()
    hist.append((q.copy(), v.copy(), b.flags().copy()))
err = []
for i in range(n):
    qo, vo, ao, w = q0[i].astype(np.float32).astype(np.float64), v0[i].astype(np.float32).astype(np.float64), a0[i].astype(np.float64), np.zeros(m.nv)
    flagged = 0
    for k in range(len(acts)):
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(acts[k, i].astype(np.float32))
        if m.na:
            qo, vo, ao, w, st = oracle.step_act(qo, vo, ao, ctrl, 10, w if mode == "3steps" else None)
        else:
            qo, vo, w, st = oracle.step(qo, vo, ctrl, 10, w if mode == "3steps" else None)
        flagged += st["unhandled_pairs"]
    err.append((np.abs(hist[-1][1][i] - vo).max(), np.abs(hist[-1][0][i] - qo).max(), flagged))
err = np.array(err)
order = np.argsort(-err[:, 0] * (err[:, 2] == 0))[:6]
print(task, mode, "worst states:", [(int(i), "dv %.2e dq %.2e" % (err[i, 0], err[i, 1])) for i in order])
os.makedirs("gpurun_out/r2_outliers", exist_ok=True)
np.savez("gpurun_out/r2_outliers/%s_%s.npz" % (task, mode), idx=order, q0=q0[order], v0=v0[order], a0=a0[order], acts=acts[:, order],
         dev_q=np.array([h[0][order] for h in hist]), dev_v=np.array([h[1][order] for h in hist]), flags=np.array([h[2][order] for h in hist]), err=err[order])
