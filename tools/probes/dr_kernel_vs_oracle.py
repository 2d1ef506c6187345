"""Talos: per-environment joint parameters vs oracle, per-env errors (DR kernel) and the nominal kernel on the same states."""
import sys, os, copy
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel
from oracle.model_blob import pack_model
from oracle.pyoracle import Oracle
np.random.seed(0)
env = LocoEnv.make("Talos.walk", debug=True); m = env._model
hm = HipModel(env._chain_model()); tab = env._reset_table()
n = 8; rs = np.random.RandomState(1)
rows = tab[rs.randint(0, len(tab), n)]
acts = rs.uniform(-0.3, 0.3, (n, 12))
damp = np.tile(m.dof_damping, (n, 1)) * rs.uniform(0.5, 2.0, (n, m.nv)) + (m.dof_damping > 0) * rs.uniform(0, 1, (n, m.nv))
stiff = np.tile(m.jnt_stiffness, (n, 1)) * rs.uniform(0.5, 1.5, (n, m.nv))
floss = np.tile(m.dof_frictionloss, (n, 1)) * rs.uniform(0.5, 1.5, (n, m.nv))
def run(mode):
    b = HipBatch(hm, n); b.set_state(rows[:, :m.nv], rows[:, m.nv:2*m.nv])
    d, s, f = damp, stiff, floss
    if mode == "nominal-params": d, s, f = np.tile(m.dof_damping,(n,1)), np.tile(m.jnt_stiffness,(n,1)), np.tile(m.dof_frictionloss,(n,1))
    if mode != "nominal": b.set_dof_params(damping=d, stiffness=s, frictionloss=f, mask=None)
    b.step(acts); q, v = b.get_state()
    for i in range(n):
        m2 = copy.copy(m)
        if mode == "dr": m2.dof_damping, m2.jnt_stiffness, m2.dof_frictionloss = d[i].astype(np.float32).astype(float), s[i].astype(np.float32).astype(float), f[i].astype(np.float32).astype(float)
        o = Oracle(pack_model(m2)); o.set_option("disable_self_collision", 1)
        ctrl = np.zeros(m.nu); ctrl[env._action_indices] = env._preprocess_action(acts[i])
        q0, v0 = rows[i,:m.nv].astype(np.float32).astype(float), rows[i,m.nv:2*m.nv].astype(np.float32).astype(float)
        qo, vo, w, st = o.step(q0, v0, ctrl, nsub=10)
        print(mode, i, '%.2e %.2e' % (np.abs(q[i]-qo).max(), np.abs(v[i]-vo).max()), 'argmax dof', int(np.abs(v[i]-vo).argmax()), st['ncon'])
    print(b.stats())
for mode in ("nominal", "nominal-params", "dr"): run(mode)
