import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel
np.random.seed(0)
task = sys.argv[1] if len(sys.argv) > 1 else "HumanoidTorque.run"
env = LocoEnv.make(task, debug=True)
tab = env._reset_table(); nv = env._model.nv
hm = HipModel(env._chain_model())
rows = tab[np.random.RandomState(0).randint(0, len(tab), 1024)]
res = {}
for mode in (0, 1, 2):
    for fuse in (1, 5, 20):
        b = HipBatch(hm, 1024); b.set_replay(mode)
        b.set_reset_table(tab, seed=1); b.set_auto_reset(True, horizon=1000)
        b.set_state(rows[:, :nv], rows[:, nv:2 * nv])
        st = b.rollout(40, action_mode=1, seed=5, steps_per_launch=fuse)
        q, v = b.get_state()
        res["mode%d_fuse%d" % (mode, fuse)] = dict(nan=st["nan_resets"], ep=st["episodes"], over=st["overflow_contacts"], rep=st["replayed_env_steps"], steps=st["env_steps"],
                                                    finite=bool(np.isfinite(q).all()), qsum=float(np.abs(q).sum()))
        print("mode", mode, "fuse", fuse, res["mode%d_fuse%d" % (mode, fuse)], flush=True)
