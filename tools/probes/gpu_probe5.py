"""GPU probe: bench-workload timing + parity error vs the fp64 oracle (256 random states) for the current knobs."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel
from loco_mujoco_amd.model_blob import pack_model
from oracle.pyoracle import Oracle
np.random.seed(0)
env = LocoEnv.make("UnitreeA1.simple", debug=True)
hm = HipModel(env._chain_model())
tab = env._reset_table()
N = 4096
rs = np.random.RandomState(0)
rows = tab[rs.randint(0, 3, N) * 100 + rs.randint(0, 100, N)]
b = HipBatch(hm, N)
b.set_reset_table(tab, seed=0); b.set_auto_reset(True, horizon=1000)
b.set_state(rows[:, :18], rows[:, 18:36]); b.set_goal(rows[:, 36:39])
b.rollout(20)
st = b.rollout(100)
out = dict(ms_per_step=round(st["kernel_ms"] / 100, 4), iters=round(st["solver_iters"] / st["env_steps"] / 10, 3), ls_per_iter=round(st["linesearch_evals"] / max(st["solver_iters"], 1), 3), ls_capped_frac=st["linesearch_capped"] / max(st["solver_iters"], 1), steps8=st["steps_with_8plus_iters"] / st["env_steps"])
rs = np.random.RandomState(1)
n = 256
r2 = tab[rs.randint(0, len(tab), n)]
qpos = r2[:, :18] + rs.uniform(-0.03, 0.03, (n, 18)); qpos[:, 2] -= rs.uniform(0, 0.03, n)
qvel = r2[:, 18:36] * rs.uniform(0.5, 1.0, (n, 1)); acts = rs.uniform(-1, 1, (n, 12))
cache = "/tmp/oracle_ref256.npz"
if os.path.exists(cache):
    Q, V = np.load(cache)["Q"], np.load(cache)["V"]
else:
    o = Oracle(pack_model(env._model)); o.set_option("disable_self_collision", 1)
    Q, V = [], []
    for k in range(n):
        q, v, _, _ = o.step(qpos[k].astype(np.float32), qvel[k].astype(np.float32), acts[k].astype(np.float32), 10)
        Q.append(q); V.append(v)
    Q, V = np.array(Q), np.array(V)
    np.savez(cache, Q=Q, V=V)
b2 = HipBatch(hm, n); b2.set_state(qpos, qvel); b2.step(acts)
q1, v1 = b2.get_state()
eq, ev = np.abs(q1 - Q).max(axis=1), np.abs(v1 - V).max(axis=1)
out.update(q_max=float(eq.max()), q_p99=float(np.percentile(eq, 99)), v_max=float(ev.max()), v_p99=float(np.percentile(ev, 99)))
print(os.environ.get("TAG", ""), json.dumps(out))
