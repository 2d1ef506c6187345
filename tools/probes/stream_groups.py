"""Stream groups: the batch split into G sub-batches, each stepping on its own stream without a device-wide join between
control steps (the launch of a step ends with its slowest environment; sub-batches let the next step of the others start).
Wall-clock throughput of K control steps at a fixed total number of environments."""
import sys, os, time, threading
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel

task = sys.argv[1] if len(sys.argv) > 1 else "UnitreeA1.simple"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = 300
default_task = task == "UnitreeA1.simple"
action_mode = 0 if default_task else 1
np.random.seed(0)
env = LocoEnv.make(task, debug=True)
table = env._reset_table()
hm = HipModel(env._chain_model())
nv = env._model.nv
rs = np.random.RandomState(0)
pick = (rs.randint(0, 3, N) * 100 + rs.randint(0, 100, N)) if default_task else rs.randint(0, len(table), N)
for G in (1, 2, 4, 8, 16):
    n = N // G
    batches = []
    for g in range(G):
        b = HipBatch(hm, n)
        rows = table[pick[g * n:(g + 1) * n]]
        b.set_reset_table(table, seed=0, global_env_offset=g * n)
        b.set_auto_reset(True, horizon=env.info.horizon)
        b.set_state(rows[:, :nv], rows[:, nv:2 * nv])
        if rows.shape[1] > 2 * nv:
            b.set_goal(rows[:, 2 * nv:])
        batches.append(b)
    def run(steps, seed):
        ths = [threading.Thread(target=lambda b=b: b.rollout(steps, action_mode=action_mode, seed=seed)) for b in batches]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        return time.perf_counter() - t0
    run(50, 11)
    el = run(K, 12)
    kms = [b.stats()["kernel_ms"] for b in batches]
    print("%s N=%d G=%2d: %.3f ms per control step of the whole batch, %.0f env-steps/s (per-group stream time %.3f ms/step)"
          % (task, N, G, el / K * 1e3, N * K / el, np.mean(kms) / (K + 50)))
    del batches
