"""GPU probe (LM_TIMERS build): region breakdown of the SLOWEST waves of single launches (the launch ends with its slowest wave).
usage: slow_waves.py <task> <n_envs> <action_mode> [warm steps]"""
import os, sys, json, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv, backend
from loco_mujoco_amd.backend import HipBatch, HipModel
task, N, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 40
np.random.seed(0)
env = LocoEnv.make(task, debug=True)
hm = HipModel(env._chain_model()); tab = env._reset_table()
nv = env._model.nv
rows = tab[np.random.RandomState(0).randint(0, len(tab), N)]
b = HipBatch(hm, N)
b.set_reset_table(tab, seed=0); b.set_auto_reset(True, horizon=1000)
b.set_state(rows[:, :nv], rows[:, nv:2 * nv])
if rows.shape[1] > 2 * nv: b.set_goal(rows[:, 2 * nv:])
b.rollout(warm, action_mode=mode, seed=3)
lib = backend.load_library()
nb = (N + 3) // 4
buf = (ctypes.c_ulonglong * (16 * nb))(); buf2 = (ctypes.c_ulonglong * (16 * nb))()
for f in (lib.lm_debug_wg_records, lib.lm_debug_wg_regions):
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
names = ["pairs+slots", "M+bias", "rows+a0", "warmstart", "gradient", "hessian", "factor+solve", "jv/Mv", "linesearch", "integrate", "lockstep",
         "kinematics", "floor prim", "floor hulls", "pair tests", "pair MPR"]
slow, allr, ms = [], [], []
for k in range(20):
    st = b.rollout(1, action_mode=mode, seed=100 + k)
    assert lib.lm_debug_wg_records(b._h, buf, nb) == 0 and lib.lm_debug_wg_regions(b._h, buf2, nb) == 0
    rec = np.array(list(buf), dtype=np.float64).reshape(nb, 16); reg = np.array(list(buf2), dtype=np.float64).reshape(nb, 16)
    ms.append(st["kernel_ms"]); allr.append(reg)
    for w in np.argsort(-reg.sum(1))[:4]:
        slow.append((reg[w].sum(), reg[w], rec[w, 1:5], rec[w, 5:9], rec[w, 9:13]))
allr = np.stack(allr)
tot = allr.sum(2)
print("launch ms mean %.3f; wave cycles mean %.0f p50 %.0f p90 %.0f p99 %.0f max(mean over launches) %.0f" % (
    np.mean(ms), tot.mean(), np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(1).mean()))
print("ALL waves, share: " + ", ".join("%s %.1f%%" % (n, 100 * v) for n, v in zip(names, allr.sum((0, 1)) / allr.sum())))
S = np.stack([s[1] for s in slow])
print("SLOWEST 4 waves of each launch, share: " + ", ".join("%s %.1f%%" % (n, 100 * v) for n, v in zip(names, S.sum(0) / S.sum())))
print("  mean cycles of those waves %.0f (%.2f ms at 2.4 GHz)" % (S.sum(1).mean(), S.sum(1).mean() / 2.4e6))
slow.sort(key=lambda s: -s[0])
for s in slow[:8]:
    print("  %.0f cycles  iters %s  contacts(sum over passes) %s  ls %s | %s" % (s[0], s[2].astype(int), s[3].astype(int), s[4].astype(int),
          " ".join("%s %.0f%%" % (n, 100 * v / s[0]) for n, v in zip(names, s[1]) if v / s[0] > 0.04)))
if hasattr(lib, "lm_debug_mpr_counters"):
    m8 = (ctypes.c_ulonglong * 16)()
    lib.lm_debug_mpr_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.lm_debug_mpr_counters(b._h, m8)
    st = b.rollout(10, action_mode=mode, seed=777)
    lib.lm_debug_mpr_counters(b._h, m8)
    m = np.array(list(m8), dtype=np.float64)
    passes = 10 * N * (40 if env._model.integrator else 10)
    print("convex collider over 10 steps (all lanes, mirrors included): calls %.0f (%.2f per env and pass), ended at the one-direction test %.1f%%, "
          "no contact %.1f%%, contact %.1f%%; support pairs per call %.2f, hill steps per support %.2f, refinement iterations per contact %.2f; "
          "rounds per wave and pass %.2f" % (m[0], m[0] / passes, 100 * m[1] / max(m[0], 1), 100 * m[2] / max(m[0], 1), 100 * m[3] / max(m[0], 1),
                                          m[4] / max(m[0], 1), m[5] / max(2 * m[4], 1), m[6] / max(m[3], 1), m[7] / 64 / (passes / 4)))
    if os.environ.get("LM_MPR_CLOCK"):      # -DLM_MPR_CLOCK builds: [10] cycles in the hull climbs, [13] cycles in the portal searches (lane sums)
        print("MPR clock probe: portal searches %.0f (contact or none), cycles per search %.0f, directions per search %.2f; per direction: %.0f cycles, of which the support call %.0f, of which the hull climbs %.0f" % (
            m[2] + m[3], m[13] / max(m[2] + m[3], 1), m[14] / max(m[2] + m[3], 1), m[13] / max(m[14], 1), m[11] / max(m[14], 1), m[10] / max(m[14], 1)))
    if os.environ.get("LM_MPR_CLOCK"):
        print("native colliders: %.0f runs, %.0f cycles each (lane sums); against the portal searches' %.0f x %.0f" % (m[15], m[12] / max(m[15], 1), m[2] + m[3], m[13] / max(m[2] + m[3], 1)))
    print("pair pass: detection ran in %.1f%% of the forward passes; geom pairs tested per detection (kind 0 / 1 / 2, all lanes and replicas): %.1f / %.1f / %.1f, "
          "of which within the margin: %.2f / %.2f / %.2f" % (100 * m[8] / max(m[9], 1), *(m[10:13] / max(m[8], 1)), *(m[13:16] / max(m[8], 1))))
