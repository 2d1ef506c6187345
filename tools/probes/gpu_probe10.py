"""Forward-pass stage comparison device vs fp64 oracle on golden rows of a humanoid task."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel
from loco_mujoco_amd.model_blob import pack_model
from oracle.pyoracle import Oracle
task = sys.argv[1] if len(sys.argv) > 1 else "HumanoidTorque.run"
np.random.seed(0)
env = LocoEnv.make(task, debug=True)
m = env._model
g = np.load(os.path.join(ROOT, "tests/golden/reference_rollouts.npz"))[task + ".real"]
qidx = [m.jnt_id(n) for k, n, t in env.obs_helper.observation_spec if k.startswith("q_")]
nq = len(qidx) - 2
n = 6
qpos, qvel = np.zeros((n, m.nv)), np.zeros((n, m.nv))
qpos[:, qidx[2:]] = g[:n, :nq]; qvel[:, qidx] = g[:n, nq:]
hm = HipModel(env._chain_model()); b = HipBatch(hm, n)
b.set_state(qpos, qvel)
acts = np.zeros((n, len(env._action_indices)))
d = b.forward_debug(acts)
o = Oracle(pack_model(m))
for i in range(n):
    f = o.forward(qpos[i], qvel[i], np.zeros(m.nu))
    print(i, "ncon dev %d oracle %d | M %.2e bias %.2e qacc_smooth %.2e qacc %.2e (|qacc| %.0f)" % (
        d["ncon"][i], f["ncon"], np.abs(d["M"][i] - f["M"]).max(), np.abs(d["qfrc_bias"][i] - f["bias"]).max(),
        np.abs(d["qacc_smooth"][i] - f["qacc_smooth"]).max(), np.abs(d["qacc"][i] - f["qacc"]).max(), np.abs(f["qacc"]).max()))
