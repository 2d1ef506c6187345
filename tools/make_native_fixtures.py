"""
tests/golden/native_pair_states.npz — states in which one of the engine's NATIVE box / cylinder colliders has a contact (sphere-box,
sphere-cylinder, capsule-box of the quadruped: trunk boxes / hip cylinders against the legs; box-box of the humanoid: one foot box
on the other), found with the fp64 oracle: rollouts from dataset states under full-range random actions without restarts, a state
is kept at the first control step in which a native pair is within its margin. Test fixture generator (oracle = test infrastructure).
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv, mjcf
from oracle.model_blob import pack_model
from oracle.pyoracle import Oracle

NAMES = {mjcf.GEOM_SPHERE: "sphere", mjcf.GEOM_CAPSULE: "capsule", mjcf.GEOM_CYLINDER: "cylinder", mjcf.GEOM_BOX: "box", mjcf.GEOM_MESH: "mesh"}


def native_types(m, f):
    out = set()
    for c in f["contacts"]:
        g1, g2 = c["geom1"], c["geom2"]
        if g1 == 0 or g2 == 0:
            continue
        t = tuple(sorted((int(m.geom_type[g1]), int(m.geom_type[g2]))))
        if t in ((mjcf.GEOM_SPHERE, mjcf.GEOM_BOX), (mjcf.GEOM_SPHERE, mjcf.GEOM_CYLINDER), (mjcf.GEOM_CAPSULE, mjcf.GEOM_BOX), (mjcf.GEOM_BOX, mjcf.GEOM_BOX)):
            out.add("%s-%s" % (NAMES[t[0]], NAMES[t[1]]))
    return out


def collect(task, want, seed, max_eps=400, steps=40, scale=1.0, per_type=12):
    np.random.seed(0)
    env = LocoEnv.make(task, debug=True)
    m = env._model
    o = Oracle(pack_model(m))
    o.set_option("skip_pair_counter", 1)
    tab = env._reset_table()
    rs = np.random.RandomState(seed)
    nu = len(env._action_indices)
    found = {k: [] for k in want}
    for ep in range(max_eps):
        row = tab[rs.randint(0, len(tab))]
        q, v, w = row[:m.nv].copy(), row[m.nv:2 * m.nv].copy(), np.zeros(m.nv)
        for s in range(steps):
            a = rs.uniform(-scale, scale, nu)
            ctrl = np.zeros(m.nu); ctrl[env._action_indices] = env._preprocess_action(a)
            q1, v1, w1, st = o.step(q, v, ctrl, 10, w)
            if not (np.isfinite(q1).all() and np.abs(v1).max() < 60):
                break
            if st["native_contacts"] > 0:
                qf, vf = q.astype(np.float32).astype(np.float64), v.astype(np.float32).astype(np.float64)
                types = set()
                qq, vv, ww = qf.copy(), vf.copy(), np.zeros(m.nv)
                for sub in range(10):
                    types |= native_types(m, o.forward(qq, vv, ctrl, ww))
                    qq, vv, ww, _ = o.step(qq, vv, ctrl, 1, ww)
                for t in types:
                    if t in found and len(found[t]) < per_type:
                        found[t].append((qf, vf, a.astype(np.float32)))
            q, v, w = q1, v1, w1
        if all(len(x) >= per_type for x in found.values()):
            break
    print(task, {k: len(x) for k, x in found.items()})
    return found


def sample(task, want, seed, beyond=0.0, height=0.3, per_type=16, tries=200000):
    """Configurations in the air: joints uniform in their ranges (+- `beyond` rad outside them: the quadruped's trunk boxes are out of
    its legs' reach inside the joint limits), small random velocities, a random action; kept by the native pair types in contact."""
    np.random.seed(0)
    env = LocoEnv.make(task, debug=True)
    m = env._model
    o = Oracle(pack_model(m))
    o.set_option("skip_pair_counter", 1)
    rs = np.random.RandomState(seed)
    lo, hi = m.jnt_range[:, 0].copy(), m.jnt_range[:, 1].copy()
    nu = len(env._action_indices)
    zdof = [i for i in range(6) if m.jnt_type[i] == 0 and abs(m.jnt_axis[i][2]) > 0.9][0]
    found = {k: [] for k in want}
    for it in range(tries):
        q = np.zeros(m.nv)
        q[zdof] = height
        q[3:6] = rs.uniform(-0.3, 0.3, 3)
        q[6:] = rs.uniform(lo[6:] - beyond, hi[6:] + beyond)
        v = rs.normal(0, 0.3, m.nv)
        q, v = q.astype(np.float32).astype(np.float64), v.astype(np.float32).astype(np.float64)
        types = native_types(m, o.forward(q, v, np.zeros(m.nu)))
        for t in types:
            if t in found and len(found[t]) < per_type and len(types) == 1:
                a = rs.uniform(-0.3, 0.3, nu).astype(np.float32)
                ctrl = np.zeros(m.nu); ctrl[env._action_indices] = env._preprocess_action(a)
                q1, v1, _, st = o.step(q, v, ctrl, 10)
                if np.isfinite(q1).all() and np.abs(v1).max() < 40 and st["unhandled_pairs"] == 0:
                    found[t].append((q, v, a))
        if all(len(x) >= per_type for x in found.values()):
            break
    print(task, "sampled", {k: len(x) for k, x in found.items()})
    return found


if __name__ == "__main__":
    out = {}
    a1 = collect("UnitreeA1.simple", ["sphere-cylinder"], seed=3, per_type=5)                   # reachable: a foot against a hip cylinder
    a1b = sample("UnitreeA1.simple", ["sphere-cylinder"], seed=5, per_type=12)
    a1c = sample("UnitreeA1.simple", ["capsule-box", "sphere-box"], seed=6, beyond=1.2, per_type=16)
    rows, types = [], []
    for d in (a1, a1b, a1c):
        for k in d:
            rows += d[k]; types += [k] * len(d[k])
    out["a1_q"], out["a1_v"], out["a1_a"] = [np.array([r[i] for r in rows]) for i in range(3)]
    out["a1_type"] = np.array(types)
    ht = collect("HumanoidTorque.run", ["box-box"], seed=4, max_eps=600, per_type=2)           # reachable: one foot on the other
    htb = sample("HumanoidTorque.run", ["box-box"], seed=7, height=0.5, per_type=30)
    rows = ht["box-box"] + htb["box-box"]
    out["ht_q"], out["ht_v"], out["ht_a"] = [np.array([r[i] for r in rows]) for i in range(3)]
    np.savez(os.path.join(ROOT, "tests", "golden", "native_pair_states.npz"), **out)
