"""GPU probe: hunt the states behind `stats.nan_resets` (quadruped, random actions): step a 4096-environment batch with host-drawn
actions and device-side restarts; every environment that ends its episode is stepped again from the same (state, action) in a second
batch WITHOUT restarts, where a non-finite result shows (parked at zero). Saves the offending (qpos, qvel, action) rows.
usage: find_nonfinite.py [steps] [task]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
task = sys.argv[2] if len(sys.argv) > 2 else "UnitreeA1.simple"
np.random.seed(0)
env = LocoEnv.make(task, debug=True)
m = env._model; nv = m.nv; nu = len(env._action_indices)
hm = HipModel(env._chain_model()); tab = env._reset_table()
n = 4096
b = HipBatch(hm, n)
rows = tab[np.random.RandomState(0).randint(0, len(tab), n)]
b.set_reset_table(tab, seed=0); b.set_auto_reset(True, horizon=1000)
b.set_state(rows[:, :nv], rows[:, nv:2 * nv])
if rows.shape[1] > 2 * nv: b.set_goal(rows[:, 2 * nv:])
rs = np.random.RandomState(1)
found = []
for s in range(steps):
    q0, v0 = b.get_state()
    a = rs.uniform(-1, 1, (n, nu))
    obs, rew, done = b.step(a)
    idx = np.nonzero(done)[0]
    if len(idx) == 0:
        continue
    b2 = HipBatch(hm, len(idx))
    b2.set_state(q0[idx], v0[idx])
    if rows.shape[1] > 2 * nv: b2.set_goal(np.tile(rows[:1, 2 * nv:], (len(idx), 1)))
    b2.step(a[idx])
    q2, v2 = b2.get_state()
    bad = np.nonzero((~np.isfinite(q2).all(1)) | (~np.isfinite(v2).all(1)) | ((q2 == 0).all(1) & (v2 == 0).all(1)))[0]
    for k in bad:
        found.append((q0[idx[k]].copy(), v0[idx[k]].copy(), a[idx[k]].copy()))
        print("step %d env %d: non-finite after one control step; z %.3f |v| %.2f" % (s, idx[k], q0[idx[k]][2], np.abs(v0[idx[k]]).max()), flush=True)
st = b.stats()
print("steps %d: nan_resets %d, found %d" % (steps, st["nan_resets"], len(found)))
if found:
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r3_nonfinite"), exist_ok=True)
    np.savez(os.path.join(ROOT, "gpurun_out", "r3_nonfinite", "%s.npz" % task), q=np.stack([f[0] for f in found]), v=np.stack([f[1] for f in found]), a=np.stack([f[2] for f in found]))
