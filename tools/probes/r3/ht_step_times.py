"""Per-control-step kernel time of a 4096-environment batch while a random policy folds the humanoids up (round 3: cost of the
self-collision path). Usage: python tools/probes/r3/ht_step_times.py [task] [steps] [self_collisions 0/1]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel

task = sys.argv[1] if len(sys.argv) > 1 else "HumanoidTorque.run"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 1
np.random.seed(0)
env = LocoEnv.make(task, debug=True)
if not pairs:
    t = env._device_task()
    t["self_collisions"] = False
    from loco_mujoco_amd import lowering
    cmod = lowering.lower(env._model, t)[0]
else:
    cmod = env._chain_model()
hm = HipModel(cmod)
tab = env._reset_table()
n = 4096
rs = np.random.RandomState(0)
rows = tab[rs.randint(0, len(tab), n)]
nv = env._model.nv
b = HipBatch(hm, n)
b.set_reset_table(tab, seed=1)
b.set_auto_reset(True, horizon=1000)
b.set_state(rows[:, :nv], rows[:, nv:2 * nv])
prev = None
for s in range(steps):
    st = b.rollout(1, action_mode=mode, seed=5)
    cur = dict(st)
    d = {k: cur[k] - (prev[k] if prev else 0) for k in ("overflow_contacts", "self_contacts", "self_proximity", "episodes", "solver_iters")}
    print("step %2d kernel %.3f ms  %s" % (s, st["kernel_ms"], d), flush=True)
    prev = cur
