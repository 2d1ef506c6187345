import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from loco_mujoco_amd import LocoEnv
from oracle.model_blob import pack_model
from oracle.pyoracle import Oracle
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
y = "/tmp/dr.yaml"
open(y, "w").write("Inertial:\n  trunk:\n    mass: {sigma: 1.0}\n    fullinertia:\n      uniform_range_delta: 0.002\nGeoms:\n  FR_calf:\n    friction:\n      uniform_range_delta: [0.3, 0.004, 0.00005]\n")
np.random.seed(0)
env = LocoEnv.make("UnitreeA1.simple", debug=True, n_envs=12, domain_randomization_config=y, n_model_variants=3)
m = env._model
st = np.load(os.path.join(root, "tests", "golden", "a1_self_contact_states.npz"))
env.reset()
variants = env._pending_variants.copy()
pick = np.argsort(-st["nself"])[:12]
q0 = st["q"][pick].astype(np.float32).astype(np.float64); v0 = st["v"][pick].astype(np.float32).astype(np.float64)
b = env.backend
env._upload_state()
b.set_state(q0, v0)
b.step(np.zeros((12, 12)))
q, v = b.get_state()
s = b.stats(); fl = b.flags()
print("stats overflow", s["overflow_contacts"], "selfcon", s["self_contacts"], "flags", fl.tolist(), "nself", st["nself"][pick].tolist())
for i in range(12):
    o = Oracle(pack_model(env._variant_models[0][variants[i]]))
    qo, vo, _, so = o.step(q0[i], v0[i], np.zeros(m.nu), nsub=10)
    print(i, "err %.2e %.2e" % (np.abs(q[i] - qo).max(), np.abs(v[i] - vo).max()), {k: so[k] for k in ("ncon", "unhandled_pairs", "convex_contacts") if k in so})
