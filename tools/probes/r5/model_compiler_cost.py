"""Cost of the model compiler on the device: Talos.walk with the golden inertial rules, 4096 environments, random actions,
device-side restarts — (a) pool of 32 host-compiled variants (round 4), (b) a freshly compiled model per environment and restart.
Run on the GPU box: python tools/probes/r5/model_compiler_cost.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from loco_mujoco_amd.environments import LocoEnv

cfg = os.path.join(os.path.dirname(__file__), "..", "..", "..", "tests", "golden", "dr_talos_inertial.yaml")
n = 4096
for label, kw in (("pool of 32", dict(n_model_variants=32)), ("compiler", dict())):
    for horizon in (1000, 50):
        np.random.seed(0)
        env = LocoEnv.make("Talos.walk", debug=True, n_envs=n, domain_randomization_config=cfg, **kw)
        env.reset()
        env.enable_auto_reset(seed=3, horizon=horizon)
        env.step(np.zeros((n, 12)))
        b = env.backend
        b.rollout(50, action_mode=1, seed=1)
        t = time.time()
        st = b.rollout(300, action_mode=1, seed=2)
        dt = time.time() - t
        extra = ""
        if env._use_model_compiler:
            extra = " models compiled so far: %d" % int(b.get_model_draws()[1].sum())
        print("%-11s horizon %4d: %.3f ms per control step (kernel events %.3f ms), %d episodes ended, overflow %d%s"
              % (label, horizon, 1e3 * dt / 300, st["kernel_ms"] / 300, st["episodes"], st["overflow_contacts"], extra), flush=True)
