"""GPU probe (LM_TIMERS build): per-workgroup cycles of single launches against what the workgroup's environments did
(solver iterations, contact slots, line-search evaluations).  usage: wave_time_distribution.py <task> <n_envs> <action_mode>"""
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
b.rollout(40, action_mode=mode, seed=3)
lib = backend.load_library()
nb = (N + 3) // 4
buf = (ctypes.c_ulonglong * (16 * nb))()
lib.lm_debug_wg_records.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
recs = []
hard = []
for k in range(30):
    q0, v0 = b.get_state()
    st = b.rollout(1, action_mode=mode, seed=100 + k)
    assert lib.lm_debug_wg_records(b._h, buf, nb) == 0
    rec = np.array(list(buf), dtype=np.float64).reshape(nb, 16)
    recs.append((st["kernel_ms"], rec))
    it_env = rec[:, 1:5].ravel()[:N]
    for e in np.argsort(-it_env)[:3]:
        hard.append((it_env[e], rec[:, 9:13].ravel()[e], q0[e].copy(), v0[e].copy()))
ms = np.array([r[0] for r in recs])
R = np.stack([r[1] for r in recs])             # [launch][wg][16]
cyc = R[:, :, 0]
it = R[:, :, 1:5]; ncon = R[:, :, 5:9]; ls = R[:, :, 9:13]
print("launch ms: mean %.3f min %.3f max %.3f" % (ms.mean(), ms.min(), ms.max()))
print("per-wave cycles: mean %.0f  p50 %.0f p90 %.0f p99 %.0f max(mean over launches) %.0f  -> max/mean %.2f" % (
    cyc.mean(), np.percentile(cyc, 50), np.percentile(cyc, 90), np.percentile(cyc, 99), cyc.max(1).mean(), cyc.max(1).mean() / cyc.mean()))
x = dict(iters_max=it.max(2).ravel(), iters_sum=it.sum(2).ravel(), ncon_max=ncon.max(2).ravel(), ncon_sum=ncon.sum(2).ravel(), ls_max=ls.max(2).ravel(),
         restarted=R[:, :, 13:15].sum(2).ravel())
y = cyc.ravel()
for k, v in x.items():
    print("corr(cycles, %s) = %.3f   mean %.1f p99 %.1f" % (k, np.corrcoef(y, v)[0, 1], v.mean(), np.percentile(v, 99)))
A = np.stack([np.ones_like(y), x["iters_max"], x["ncon_max"], x["ls_max"]], 1)
coef = np.linalg.lstsq(A, y, rcond=None)[0]
print("cycles ~ %.0f + %.0f * iters_max + %.0f * ncon_max + %.0f * ls_max   (R2 %.3f)" % (*coef, 1 - ((A @ coef - y) ** 2).sum() / ((y - y.mean()) ** 2).sum()))
top = np.argsort(-y)[:10]
print("slowest waves: cycles / iters per env / ncon per env / ls per env")
for i in top:
    l, w = divmod(i, nb)
    print("  %.0f  %s  %s  %s" % (y[i], it[l, w].astype(int), ncon[l, w].astype(int), ls[l, w].astype(int)))
# what a perfectly balanced launch would take: mean vs max
print("if every wave took the mean: %.3f ms of the %.3f ms launch" % (ms.mean() * cyc.mean() / cyc.max(1).mean(), ms.mean()))

if mode == 0:
    hard.sort(key=lambda h: -h[0])
    os.makedirs(os.path.join(ROOT, "gpurun_out", "probe"), exist_ok=True)
    np.savez(os.path.join(ROOT, "gpurun_out", "probe", "hard_states_%s.npz" % task), iters=np.array([h[0] for h in hard]), ls=np.array([h[1] for h in hard]),
             q=np.stack([h[2] for h in hard]), v=np.stack([h[3] for h in hard]))
