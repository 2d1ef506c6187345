"""Generates tests/golden/six_link_self_contact_states.npz: states of the six-link robots (UnitreeG1 default; UnitreeH1 with its arms) in
which links of the robot touch each other — the engine's convex collider between link hulls, between the two arm chains that share the
torso link, between the torso and the arm that carries its massless copy. From fp64 oracle rollouts under a random policy (robots that
stumble and fold) and, for UnitreeG1, a random search over arm poses for contacts between the two arms. Every state comes with the
oracle's own spread under float32-sized input noise (8 probes): the tests hold the well-conditioned ones to the stated tolerance.

    python tools/make_six_link_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from loco_mujoco_amd import LocoEnv, lowering            # noqa: E402
from oracle.model_blob import pack_model        # noqa: E402
from oracle.pyoracle import Oracle                       # noqa: E402


def spread(o, m, q, v, ctrl, rs, n=8):
    qo, vo, _, _ = o.step(q, v, ctrl, 10)
    sq = sv = 0.0
    for _ in range(n):
        q2, v2, _, _ = o.step(q * (1 + 6e-8 * rs.randn(m.nv)), v * (1 + 6e-8 * rs.randn(m.nv)), ctrl, 10)
        sq, sv = max(sq, np.abs(q2 - qo).max()), max(sv, np.abs(v2 - vo).max())
    return sq, sv


out = {}
for name, task, kw, want in (("g1", "UnitreeG1.walk", {}, 18), ("h1arms", "UnitreeH1.walk", dict(disable_arms=False), 10)):
    np.random.seed(0)
    env = LocoEnv.make(task, debug=True, **kw)
    m = env._model
    _, info = lowering.lower(m, env._device_task())
    chains = info["chains"]
    o = Oracle(pack_model(m))
    tab = env._reset_table()
    rs = np.random.RandomState(3)
    nu = len(env._action_indices)
    Q, V, A = [], [], []
    for ep in range(200):
        row = tab[rs.randint(0, len(tab))]
        q, v, w = row[:m.nv].copy(), row[m.nv:2 * m.nv].copy(), np.zeros(m.nv)
        for k in range(14):
            a = rs.uniform(-1, 1, nu)
            ctrl = np.zeros(m.nu)
            ctrl[env._action_indices] = env._preprocess_action(a)
            q0, v0 = q.copy(), v.copy()
            q, v, w, st = o.step(q, v, ctrl, 10, w)
            if st["convex_contacts"] > 0 and st["unhandled_pairs"] == 0 and len(Q) < want:
                Q.append(q0), V.append(v0), A.append(a)
        if len(Q) >= want:
            break
    if name == "g1":
        # contacts between the two arm chains (they share the torso link): random arm poses on dataset states
        arm_dofs = [d for c in (2, 3) for b in chains[c][1:] for d in range(m.body_jntadr[b], m.body_jntadr[b] + m.body_jntnum[b])]

        def lane_of(b):
            w_ = m.body_weldid[b]
            for c in (3, 2, 1, 0):
                if w_ in chains[c] and not (c == 3 and w_ == chains[3][0]):
                    return c, chains[c].index(w_)
            return -1, 7
        rs2, found = np.random.RandomState(5), 0
        for _ in range(6000):
            row = tab[rs2.randint(0, len(tab))]
            q, v = row[:m.nv].copy(), 0.2 * row[m.nv:2 * m.nv].copy()
            for d in arm_dofs:
                lo, hi = m.jnt_range[d] if m.jnt_limited[d] else (-1.5, 1.5)
                q[d] = rs2.uniform(lo, hi)
            f = o.forward(q, v, np.zeros(m.nu))
            cons = [c for c in f["contacts"] if c["geom1"] != 0]
            lanes = {(lane_of(m.geom_body[c["geom1"]]), lane_of(m.geom_body[c["geom2"]])) for c in cons}
            cross = [p for p in lanes if {p[0][0], p[1][0]} == {2, 3} and p[0][1] > 0 and p[1][1] > 0]
            if cross and min(c["dist"] for c in cons) > -0.01:
                Q.append(q), V.append(v), A.append(rs2.uniform(-0.3, 0.3, nu))
                found += 1
                if found >= 4:
                    break
    Q, V, A = np.array(Q), np.array(V), np.array(A)
    rs3 = np.random.RandomState(0)
    S = []
    for i in range(len(Q)):
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(A[i])
        S.append(spread(o, m, Q[i], V[i], ctrl, rs3))
    S = np.array(S)
    print(name, len(Q), "states; well-conditioned (oracle spread below 1e-5 / 1e-3):", int(((S[:, 0] < 1e-5) & (S[:, 1] < 1e-3)).sum()))
    out[name + "_qpos"], out[name + "_qvel"], out[name + "_action"], out[name + "_oracle_spread"] = Q, V, A, S
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "six_link_self_contact_states.npz"), **out)
