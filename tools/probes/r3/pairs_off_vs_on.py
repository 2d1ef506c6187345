"""GPU probe: the same library with the self-collision tables switched OFF in the lowering (what round 2 simulated: the robots' own
geom pairs ignored) against the shipped configuration — steady-state ms per control step at 4096 environments under each workload's
policy, device-side restarts. Usage: python tools/probes/r3/pairs_off_vs_on.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from loco_mujoco_amd import LocoEnv, lowering
from loco_mujoco_amd.backend import HipBatch, HipModel
for task, mode in (("UnitreeA1.simple", 0), ("HumanoidTorque.run", 1), ("HumanoidMuscle.run", 1), ("UnitreeH1.run", 1)):
    out = []
    for pairs in (0, 1):
        np.random.seed(0)
        env = LocoEnv.make(task, debug=True)
        t = env._device_task()
        t["self_collisions"] = bool(pairs)
        hm = HipModel(lowering.lower(env._model, t)[0])
        tab = env._reset_table(); nv = env._model.nv; n = 4096
        rows = tab[np.random.RandomState(0).randint(0, len(tab), n)]
        b = HipBatch(hm, n)
        b.set_reset_table(tab, seed=0); b.set_auto_reset(True, horizon=1000)
        b.set_state(rows[:, :nv], rows[:, nv:2 * nv])
        if rows.shape[1] > 2 * nv: b.set_goal(rows[:, 2 * nv:])
        b.rollout(100, action_mode=mode, seed=11)
        st = b.rollout(300, action_mode=mode, seed=12)
        out.append(st["kernel_ms"] / 300)
    print("%-20s pair tables off %.3f ms   on %.3f ms" % (task, out[0], out[1]), flush=True)
