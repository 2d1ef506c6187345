"""GPU probe (LM_TIMERS builds): what does ONE hard environment cost in the replay kernel, and where?
usage: LOCOHIP_LIB=<timers lib> replay_profile.py <task> [n_hard]
1. 4096 robots under the random policy, 40 control steps; then single steps until `n_hard` (state, action-seed) pairs of environments
   that the regular kernel hands to the replay kernel are collected (their states BEFORE the step).
2. groups of four hard states: the regular kernel alone (one wave, replay off: drops) vs every control step through the replay kernel
   without pollers (ONE workgroup runs the four after the other): kernel ms per environment, and the region breakdown of the replay
   workgroup's LAST environment (cycles per region)."""
import os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv, backend
from loco_mujoco_amd.backend import HipBatch, HipModel
task = sys.argv[1]
n_hard = int(sys.argv[2]) if len(sys.argv) > 2 else 32
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
rs = np.random.RandomState(1)
hq, hv, ha = [], [], []
nu = len(env._action_indices)
for k in range(30):
    q0, v0 = b.get_state()
    acts = rs.uniform(-1, 1, (N, nu))
    b.replay_marks(reset=True)
    b.step(acts)
    m = b.replay_marks()
    for e in np.nonzero(m)[0]:
        hq.append(q0[e]); hv.append(v0[e]); ha.append(acts[e])
    if len(hq) >= n_hard: break
print("%s: %d hard states collected in %d steps (%.1f per launch)" % (task, len(hq), k + 1, len(hq) / (k + 1)))
hq, hv, ha = np.array(hq[:n_hard]), np.array(hv[:n_hard]), np.array(ha[:n_hard])
lib = backend.load_library()
names = ["pairs+slots", "M+bias", "rows+a0", "warmstart", "gradient", "hessian", "factor+solve", "jv/Mv", "linesearch", "integrate", "lockstep",
         "kinematics", "floor prim", "floor hulls", "pair tests", "pair MPR"]
has_t = hasattr(lib, "lm_debug_wg_regions")
if has_t:
    for f in (lib.lm_debug_wg_records, lib.lm_debug_wg_regions): f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
res = []
for g in range(0, len(hq) - 3, 4):
    out = {}
    for mode in (0, 4):
        bb = HipBatch(hm, 4); bb.set_replay(mode)
        bb.set_state(hq[g:g + 4], hv[g:g + 4])
        bb.step(ha[g:g + 4])                      # warm the code / tables (state advances: set it again)
        bb.set_state(hq[g:g + 4], hv[g:g + 4])
        bb.stats(reset=True)
        # one control step with the SAME actions, timed by the library's events: lm_step has no timer, so use lm_step then stats? use rollout(1) with the device policy instead
        st = bb.rollout(1, action_mode=1, seed=77 + g)
        out[mode] = (st["kernel_ms"], st["solver_iters"], st["overflow_contacts"], st["replayed_env_steps"])
        if has_t and mode == 4:
            buf = (ctypes.c_ulonglong * 16)()
            lib.lm_debug_wg_regions(bb._h, buf, 1)
            out["regions"] = np.array(list(buf), dtype=np.float64)
        bb.close()
    res.append(out)
    print("group %2d: regular wave %.2f ms (iters %d, dropped %d) | replay kernel %.2f ms for 4 = %.2f ms per environment (iters %d)" % (
        g // 4, out[0][0], out[0][1], out[0][2], out[4][0], out[4][0] / 4, out[4][1]))
reg = np.mean([r[0][0] for r in res]); rep = np.mean([r[4][0] / 4 for r in res])
print("MEAN: regular wave of four hard robots %.2f ms; replay kernel %.2f ms per hard robot" % (reg, rep))
if has_t:
    R = np.stack([r["regions"] for r in res])
    tot = R.sum(1).mean()
    print("replay workgroup, last environment: %.0f cycles (%.2f ms at 2.4 GHz); share: " % (tot, tot / 2.4e6) + ", ".join("%s %.1f%%" % (n, 100 * v) for n, v in zip(names, R.sum(0) / R.sum())))
