"""
TEST INFRASTRUCTURE (lives with the oracle, not in the product package). Pack a :class:`loco_mujoco_amd.mjcf.CompiledModel` into the flat float64 "general model" array
(``include/lm_model_blob.h``) that the fp64 test oracle consumes. The product path uses
``loco_mujoco_amd.lowering.lower`` instead.
"""

import numpy as np

LM_BLOB_MAGIC = 0x4C4D4231
HEADER_SIZE = 32


def pack_model(m):
    h = np.zeros(HEADER_SIZE)
    h[0] = LM_BLOB_MAGIC
    h[1] = 6
    h[2:9] = [m.nbody, m.nv, m.ngeom, m.nu, m.cone, m.integrator, m.iterations]
    h[9:12] = [m.timestep, m.impratio, m.tolerance]
    h[12:15] = m.gravity
    h[15] = m.meaninertia
    # tendon paths only reference a subset of the sites: pack those (renumbered)
    used = sorted(set(int(i) for i in getattr(m, "wrap_site", [])))
    renum = {s: i for i, s in enumerate(used)}
    wrap = np.array([renum[int(i)] for i in getattr(m, "wrap_site", [])], dtype=np.float64)
    nt = int(getattr(m, "ntendon", 0))
    h[16:20] = [len(used), nt, len(wrap), int(getattr(m, "na", 0))]
    hull_vert = np.asarray(getattr(m, "hull_vert", np.zeros((0, 3))), dtype=np.float64)
    h[20] = len(hull_vert)
    hull_nbr_adr = np.asarray(getattr(m, "hull_nbr_adr", np.zeros(len(hull_vert) + 1)), dtype=np.float64)
    hull_nbr = np.asarray(getattr(m, "hull_nbr", np.zeros(0)), dtype=np.float64)
    assert len(hull_nbr_adr) == len(hull_vert) + 1
    h[21] = len(hull_nbr)
    zeros = lambda *shape: np.zeros(shape)
    nu = m.nu
    parts = [h,
             m.body_parent, m.body_pos, m.body_quat, m.body_mass, m.body_ipos, m.body_inertia, m.body_jntadr,
             m.body_jntnum, m.body_weldid, m.body_invweight0,
             m.jnt_type, m.jnt_body, m.jnt_pos, m.jnt_axis, m.jnt_limited, m.jnt_range, m.jnt_stiffness,
             m.jnt_margin, m.jnt_solref, m.jnt_solimp,
             m.dof_damping, m.dof_armature, m.dof_frictionloss, m.dof_solref, m.dof_solimp, m.dof_parent,
             m.dof_invweight0,
             m.geom_type, m.geom_body, m.geom_pos, m.geom_quat, m.geom_size, m.geom_contype, m.geom_conaffinity,
             m.geom_condim, m.geom_priority, m.geom_friction, m.geom_solmix, m.geom_solref, m.geom_solimp,
             m.geom_margin, m.geom_gap,
             m.act_dof, m.act_gear, m.act_ctrlrange, m.act_ctrllimited,
             m.site_body[used] if used else zeros(0), m.site_pos[used] if used else zeros(0),
             getattr(m, "tendon_adr", zeros(0)), getattr(m, "tendon_num", zeros(0)), wrap,
             getattr(m, "act_kind", zeros(nu)), getattr(m, "act_tendon", -np.ones(nu)), getattr(m, "act_dynprm", zeros(nu, 3)),
             getattr(m, "act_gainprm", zeros(nu, 9)), getattr(m, "act_lengthrange", zeros(nu, 2)),
             getattr(m, "act_biasprm", zeros(nu, 3)), getattr(m, "act_forcerange", zeros(nu, 2)),
             getattr(m, "act_forcelimited", zeros(nu)),
             getattr(m, "geom_hull_adr", -np.ones(m.ngeom)), getattr(m, "geom_hull_num", zeros(m.ngeom)), hull_vert, hull_nbr_adr, hull_nbr,
             getattr(m, "geom_center", m.geom_pos)]
    return np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float64).ravel() for p in parts]))
