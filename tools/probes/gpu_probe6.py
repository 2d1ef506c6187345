"""GPU probe: one-Newton-iteration cost split by region (iteration cap 1, LM_ABLATE bitmask)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %r)
from loco_mujoco_amd import LocoEnv, lowering
from loco_mujoco_amd.backend import HipBatch, HipModel
np.random.seed(0)
env = LocoEnv.make("UnitreeA1.simple", debug=True)
tab = env._reset_table()
N = 4096
rs = np.random.RandomState(0)
rows = tab[rs.randint(0, 3, N) * 100 + rs.randint(0, 100, N)]
cmod = env._chain_model().copy(); cmod[lowering.H_ITERATIONS] = int(os.environ["CAP"])
hm = HipModel(cmod); b = HipBatch(hm, N)
b.set_state(rows[:, :18], rows[:, 18:36]); b.set_goal(rows[:, 36:39])
b.rollout(3)
st = b.rollout(10)
print(round(st["kernel_ms"] / 10, 4))
''' % ROOT
res = {}
for cap, abl in [(0, 0), (1, 0), (1, 8), (1, 14), (1, 14 + 16), (1, 14 + 32), (1, 14 + 64), (1, 14 + 112), (3, 14), (3, 14 + 112)]:
    env = dict(os.environ, CAP=str(cap), LM_ABLATE=str(abl))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    res["cap%d_abl%d" % (cap, abl)] = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]
print(json.dumps(res))
