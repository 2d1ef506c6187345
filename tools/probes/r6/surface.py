"""GPU probe: what LocoEnv.step() costs at n_envs = 4096 (UnitreeA1.simple, zero action, device-side auto-reset) beside the policy-free
rollout of the same batch: ms per call, the part inside the library call (lm_step_pinned), and the rollout's ms per control step.
usage: surface.py [steps]      (LOCOHIP_LIB selects the library variant)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = 4096
np.random.seed(0)
env = LocoEnv.make("UnitreeA1.simple", debug=True, n_envs=n)
env.reset()
env.enable_auto_reset(seed=0)
act = np.zeros((n, 12))
for _ in range(30):
    env.step(act)
b = env.backend
inner = [0.0]
orig = b.step_pinned
def timed(a):
    t = time.perf_counter(); r = orig(a); inner[0] += time.perf_counter() - t; return r
b.step_pinned = timed
t0 = time.perf_counter()
for _ in range(steps):
    obs, rew, absorbing, info = env.step(act)
dt = time.perf_counter() - t0
b.step_pinned = orig
st = b.rollout(steps, action_mode=0, seed=1)
print("%s: LocoEnv.step %.4f ms per call (library call %.4f, Python around it %.4f); policy-free rollout %.4f ms per control step; ratio %.4f" % (
    os.path.basename(os.environ.get("LOCOHIP_LIB", "liblocohip.so")), 1e3 * dt / steps, 1e3 * inner[0] / steps, 1e3 * (dt - inner[0]) / steps,
    st["kernel_ms"] / steps, 1e3 * dt / steps / (st["kernel_ms"] / steps)))
