"""Round 4 probe: what does ONE environment's control step cost in the replay kernel (one environment per workgroup, 128 slots per chain)
against the regular kernel (four per wave)? Same states, small batch (every workgroup resident at once): replay off (0), on (1), every
control step through the replay kernel (2). ms per control step of single-step launches."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel

for task in sys.argv[1:] or ["Talos.walk", "Atlas.walk", "HumanoidTorque.run", "UnitreeA1.simple"]:
    np.random.seed(0)
    env = LocoEnv.make(task, debug=True)
    table = env._reset_table(); hm = HipModel(env._chain_model()); nv = env._model.nv
    for n in (64, 1024):
        res = {}
        for mode in (0, 1, 2):
            b = HipBatch(hm, n); b.set_replay(mode)
            rows = table[np.random.RandomState(0).randint(0, len(table), n)]
            b.set_reset_table(table, seed=0); b.set_auto_reset(True, horizon=1000)
            b.set_state(rows[:, :nv], rows[:, nv:2 * nv])
            if rows.shape[1] > 2 * nv: b.set_goal(rows[:, 2 * nv:])
            am = 0 if task.startswith("UnitreeA1") else 1
            b.rollout(20, action_mode=am, seed=11); b.stats(reset=True)
            st = b.rollout(40, action_mode=am, seed=12)
            res[mode] = dict(ms=round(st["kernel_ms"] / 40, 4), replayed=st["replayed_env_steps"], overflow=st["overflow_contacts"])
            b.close()
        print(task, n, json.dumps(res), flush=True)
