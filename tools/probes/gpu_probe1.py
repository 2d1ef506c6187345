"""GPU probe: dump outputs for the random-state parity case and time the step kernel for several envs-per-block."""
import os, sys, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel
np.random.seed(0)
env = LocoEnv.make("UnitreeA1.simple", debug=True)
hm = HipModel(env._chain_model())
tab = env._reset_table()
rs = np.random.RandomState(1)
n = 256
rows = tab[rs.randint(0, len(tab), n)]
qpos = rows[:, :18] + rs.uniform(-0.03, 0.03, (n, 18))
qpos[:, 2] -= rs.uniform(0, 0.03, n)
qvel = rows[:, 18:36] * rs.uniform(0.5, 1.0, (n, 1))
acts = rs.uniform(-1, 1, (n, 12))
out = {}
for epb in (16, 4, 1):
    os.environ["LM_ENVS_PER_BLOCK"] = str(epb)
    b = HipBatch(hm, n)
    b.set_state(qpos, qvel)
    d = b.forward_debug(acts)
    obs, _, _ = b.step(acts)
    q1, v1 = b.get_state()
    out["q1_%d" % epb], out["v1_%d" % epb], out["qacc_%d" % epb], out["iter_%d" % epb] = q1, v1, d["qacc"], d["solver_iter"]
np.savez(os.path.join(ROOT, "gpurun_out", "probe1.npz"), qpos=qpos, qvel=qvel, acts=acts, **out)
res = {}
N = 4096
rs = np.random.RandomState(0)
rows = tab[rs.randint(0, 3, N) * 100 + rs.randint(0, 100, N)]
for epb in (16, 8, 4, 2, 1):
    os.environ["LM_ENVS_PER_BLOCK"] = str(epb)
    b = HipBatch(hm, N)
    b.set_reset_table(tab, seed=0)
    b.set_auto_reset(True, horizon=1000)
    b.set_state(rows[:, :18], rows[:, 18:36]); b.set_goal(rows[:, 36:39])
    b.rollout(20)
    st = b.rollout(100)
    res[epb] = dict(ms_per_step=st["kernel_ms"] / 100, env_steps_per_s=N * 100 / (st["kernel_ms"] * 1e-3))
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "probe1_timing.json"), "w"))
