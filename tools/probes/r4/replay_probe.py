"""Round 4 probe: speculate / replay on the GPU. For each workload: a rollout with replay off (the regular kernels alone, contacts beyond
their slots dropped) and on; ms per control step, dropped contacts, replayed env-steps. Then the fixture states with more contacts
than slots against the fp64 oracle, replay off / on."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel
from oracle.model_blob import pack_model
from oracle.pyoracle import Oracle

def rollout(task, n=4096, steps=60, warm=30, kw={}, random_a1=False):
    np.random.seed(0)
    env = LocoEnv.make(task, debug=True, **kw)
    table = env._reset_table()
    hm = HipModel(env._chain_model())
    out = {}
    for replay in (0, 1):
        b = HipBatch(hm, n)
        b.set_replay(replay)
        rs = np.random.RandomState(0)
        rows = table[rs.randint(0, len(table), n)]
        nv = env._model.nv
        b.set_reset_table(table, seed=0); b.set_auto_reset(True, horizon=env.info.horizon)
        b.set_state(rows[:, :nv], rows[:, nv:2 * nv])
        if rows.shape[1] > 2 * nv: b.set_goal(rows[:, 2 * nv:])
        mode = 0 if task.startswith("UnitreeA1") and not random_a1 else 1
        b.rollout(warm, action_mode=mode, seed=11); b.stats(reset=True)
        st = b.rollout(steps, action_mode=mode, seed=12)
        q, v = b.get_state()
        out[replay] = dict(ms=st["kernel_ms"] / steps, overflow=st["overflow_contacts"], replayed=st["replayed_env_steps"], env_steps=st["env_steps"],
                           nan=st["nan_resets"], finite=bool(np.isfinite(q).all()), selfcon=st["self_contacts"], prox=st["self_proximity"])
        stf = b.rollout(50, action_mode=mode, seed=13, steps_per_launch=25)
        out[replay]["fused_ms"] = stf["kernel_ms"] / 50
        out[replay]["fused_overflow"] = stf["overflow_contacts"] - st["overflow_contacts"]
    print(task, kw, json.dumps(out), flush=True)

def fixture(task, path, kw={}):
    np.random.seed(0)
    env = LocoEnv.make(task, debug=True, **kw)
    m = env._model
    hm = HipModel(env._chain_model())
    oracle = Oracle(pack_model(m))
    d = np.load(os.path.join(ROOT, "tests", "golden", path))
    q0, v0 = d["q"], d["v"]
    n = len(q0)
    a = d["a"] if "a" in d.files else np.zeros((n, len(env._action_indices)))
    res = {}
    for replay in (0, 1):
        b = HipBatch(hm, n); b.set_replay(replay)
        b.set_state(q0, v0); b.step(a)
        q, v = b.get_state(); fl = b.flags(); st = b.stats()
        eq, ev = [], []
        for i in range(n):
            ctrl = np.zeros(m.nu); ctrl[env._action_indices] = env._preprocess_action(a[i])
            qo, vo, _, so = oracle.step(q0[i].astype(np.float32).astype(np.float64), v0[i].astype(np.float32).astype(np.float64), ctrl, 10)
            eq.append(np.abs(q[i] - qo).max()); ev.append(np.abs(v[i] - vo).max())
        res[replay] = dict(n=n, dropped_flags=int((fl & 1).sum()), overflow=st["overflow_contacts"], replayed=st["replayed_env_steps"],
                           qmax=float(max(eq)), vmax=float(max(ev)), qmed=float(np.median(eq)), vmed=float(np.median(ev)))
    print("fixture", task, path, json.dumps(res), flush=True)

if __name__ == "__main__":
    which = sys.argv[1:] or ["fix", "roll"]
    if "fix" in which:
        fixture("UnitreeA1.simple", "a1_tangled_states.npz")
        fixture("UnitreeA1.simple", "a1_self_contact_states.npz")
        fixture("HumanoidTorque.run", "ht_folded_states.npz")
        fixture("Atlas.walk", "atlas_cylinder_states.npz")
    if "roll" in which:
        rollout("UnitreeA1.simple")
        rollout("UnitreeA1.simple", random_a1=True)
        rollout("HumanoidTorque.run", steps=40, warm=40)
        rollout("Atlas.walk", steps=40, warm=40)
        rollout("Talos.walk", steps=40, warm=40)
        rollout("UnitreeH1.run", steps=40, warm=40)
        rollout("HumanoidMuscle.run", n=2048, steps=40, warm=40)
        rollout("UnitreeG1.walk", steps=40, warm=40)
