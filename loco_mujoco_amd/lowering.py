"""
Lowering of a :class:`~loco_mujoco_amd.mjcf.CompiledModel` to the "root + chains" device model the HIP
step kernel is written for (csrc/lm_core.h).

Topology the kernel supports (all four BASELINE robots have it): one root body
hanging off the world with up to 6 scalar joints, and up to 4 serial chains of 1-dof links below it
(A1: four 3-link legs; humanoids: two 5-dof legs + a 3-dof lumbar chain). One GPU lane simulates one
chain of one environment; the 4 lanes of a quad share the root.

Lowering steps:
* bodies without joints are welded into their parent (mass, centre of mass, inertia tensor and geoms are
  re-expressed in the parent's frame) — e.g. the A1 goal-arrow body;
* every scalar joint becomes one "link"; a body with several joints becomes consecutive links of which only
  the last carries the mass and the geoms;
* constraint constants that do not depend on the state are folded on the host: friction-loss regulariser
  R = (1-d0)/d0 * dof_invweight0 and its damping gain, joint-limit stiffness/damping from solref/solimp,
  per-geom floor-contact parameters after MuJoCo's priority/max mixing with the floor plane (condim,
  friction, solref -> (K, B), solimp, margin, the R_j/R_0 ratios of the elliptic friction rows and the cone mu).

The result is ONE float64 array, the "chain model" that crosses the C-ABI (``lm_model_create``): a 32-slot
header (``H_*``) followed by the constant table (``D_*``/``L_*``/``G_*``/``C_*``/``R_*`` offsets below).
``include/lm_layout.h`` is generated from these enums by ``tools/gen_layout_header.py``.
"""

import os
import numpy as np

from . import mjcf

MAXC = 6          # links per chain the table has room for (5 for every robot but UnitreeG1's legs)
NCHAIN = 4
NROOT = 6
MAXG = 68         # floor-collidable geoms per chain (with / without a device collider, each); the humanoid's trunk chain with its welded arms has 65 (every hand bone)
MAXRG = 80        # geoms welded to the root (no device collider: proximity is counted)

# ---- per-dof parameter block (used for root dofs and chain links)
(D_TYPE, D_AX, D_AY, D_AZ, D_PX, D_PY, D_PZ, D_DAMP, D_ARM, D_STIFF, D_FLOSS, D_FLOSS_R, D_FLOSS_B, D_LIMITED,
 D_LO, D_HI, D_LIM_K, D_LIM_B, D_LIM_S0, D_LIM_S1, D_LIM_S2, D_LIM_S3, D_LIM_S4, D_INVW, D_GEAR, D_CTRL_LO,
 D_CTRL_HI, D_ACT, D_ACT_MEAN, D_ACT_DELTA, D_QOBS, D_VOBS, D_TERM_QLO, D_TERM_QHI, D_TERM_VLO, D_TERM_VHI, D_DOF,
 D_FLO, D_FHI, D_SIZE) = range(40)
# D_FLO / D_FHI: force range of a position servo (H_ACTMODE = 1: torque = clamp(D_GEAR * (ctrl - q), D_FLO, D_FHI))
# ---- per-link extras (chain links only), after the dof block
(L_HAS_T, L_TX, L_TY, L_TZ, L_R0, L_R1, L_R2, L_R3, L_R4, L_R5, L_R6, L_R7, L_R8, L_MASS, L_CX, L_CY, L_CZ,
 L_IXX, L_IYY, L_IZZ, L_IXY, L_IXZ, L_IYZ, L_BSX, L_BSY, L_BSZ, L_BSR, L_SIZE) = range(28)
# L_BS*: centre and radius of the link's bounding sphere in the link frame (self-collision broad phase)
LINK_SIZE = D_SIZE + L_SIZE
# ---- per-geom block
(G_LINK, G_TYPE, G_PX, G_PY, G_PZ, G_AX, G_AY, G_AZ, G_RADIUS, G_HALF, G_RBOUND, G_MARGIN, G_K, G_B, G_S0, G_S1,
 G_S2, G_S3, G_S4, G_TRAN, G_DIM, G_MU, G_F0, G_F1, G_F2, G_F3, G_F4, G_RR1, G_RR2, G_RR3, G_RR4, G_RR5,
 G_SX, G_SY, G_SZ, G_R0, G_R1, G_R2, G_R3, G_R4, G_R5, G_R6, G_R7, G_R8, G_GRF, G_SIZE) = range(46)
# G_GRF: ground-reaction-force group of this geom within its chain (0/1), -1 = not reported
# G_TRAN: elliptic: tran (R_normal = (1-imp)/imp * tran); pyramidal: 2 mu^2 (1+mu^2) tran (shared R of all edges)
# ---- chain block (LDS, interleaved [field][chain]) = [nlinks, ngeoms, unsupported geoms (count), force-group slots, links...]
C_NLINKS, C_NGEOMS, C_NUNSUP, C_GRF_OBS0, C_GRF_OBS1, C_NLPAIR, C_NLGROUP, C_DUPROLE = 0, 1, 2, 3, 4, 5, 6, 7
# where this chain's tail lists start in the constant table (floats from its beginning): every chain's list is contiguous
# ([entry][field]) and only as long as the chain needs — a humanoid's trunk has 40 geoms, its legs 8
C_OFF_CUNSUP, C_OFF_PRUNE, C_OFF_LGROUP, C_OFF_LPAIR, C_GRF_OBS2, C_GRF_OBS3, C_LINKS = 8, 9, 10, 11, 12, 13, 14
C_GRF_OBS = (C_GRF_OBS0, C_GRF_OBS1, C_GRF_OBS2, C_GRF_OBS3)     # groups 2, 3: six-link kernels only (UnitreeG1: four force points per foot)
# C_DUPROLE: +1 = this chain's first link is SHARED with another chain and this lane owns its dof, -1 = this lane carries the
# massless copy of that link (a torso with two arms: two chains [torso, arm], one dof for the torso), 0 = neither
# C_NLPAIR: link-pair entries of the self-collision broad phase that involve this chain
# C_GRF_OBS0..3: observation index of the (normal, t1, t2) mean force of the chain's force group 0..3, -1 = none
CHAIN_SIZE = C_LINKS + MAXC * LINK_SIZE
U_SIZE = 6        # collider-less ("unsupported") geom: (link, px,py,pz, rbound, margin)
P_SIZE = 6        # prune record of a geom with a collider: (link, px,py,pz, rbound, type) — the full record is in the geom table
LG_SIZE = 7       # geoms of one link as a group: (link, first geom, count, bounding sphere cx,cy,cz, r); link -1 = root body
MAXLG = MAXC + 1
# ---- root block (replicated for all lanes)
(R_NDOF, R_TX, R_TY, R_TZ, R_R0, R_R1, R_R2, R_R3, R_R4, R_R5, R_R6, R_R7, R_R8, R_MASS, R_CX, R_CY, R_CZ, R_IXX,
 R_IYY, R_IZZ, R_IXY, R_IXZ, R_IYZ, R_NUNSUP, R_BSX, R_BSY, R_BSZ, R_DOFS) = range(28)
ROOT_SIZE = R_DOFS + NROOT * D_SIZE
# ---- the LDS constant table: root block, the chain blocks interleaved [field][chain], then a TAIL whose offsets are
# header fields (H_OFF_*): it starts right behind the last link slot the model uses, so a robot with 3-link chains and a
# few geoms ships a short table: [root unsupported geoms x U_SIZE][chain unsupported geoms, chain by chain][prune records, chain by chain] ...
CM_ROOT = 0
CM_CHAINS = ROOT_SIZE
# ---- self-collisions (kernels compiled with PAIRS): per lane a list of link pairs (LP_SIZE floats each, chain by chain
# [entry][field] in the tail): code = own link + 8 * partner link (7 = root body) + 64 * partner lane (the own link is the pair's FIRST
# link: a cross-chain pair is listed in that lane only; + 256 would mark the second link's view, which the device still decodes), range = first body pair + 65536 * number of body pairs (<= 24), squared reach of the two bounding spheres
MAXLP = 128        # link-pair entries per lane (64 for the three- and five-link families: one mask word on the device)
PAIR_PAD = 0.03   # metres: link pairs closer than touching + PAIR_PAD go through the narrow phase (csrc/lm_core.h LM_PAIR_PAD)
LP_SIZE = 3
# geom-pair records (global memory): kind (0 sphere/capsule pair with a collider, 1 counted only: bounding capsules of a pair the
# engine collides natively with a box or a sphere against a cylinder, 2 convex pair: MPR), geom 1 on the pair's second link?, geom 1 / geom 2 as capsules in their link frames (centre, axis, half length,
# radius), then the contact parameters after the engine's mixing rules
(GP_KIND, GP_G1Q, GP_P1, GP_P1Y, GP_P1Z, GP_A1, GP_A1Y, GP_A1Z, GP_H1, GP_R1, GP_P2, GP_P2Y, GP_P2Z, GP_A2, GP_A2Y, GP_A2Z, GP_H2,
 GP_R2, GP_MARGIN, GP_K, GP_B, GP_S0, GP_S1, GP_S2, GP_S3, GP_S4, GP_TRAN, GP_DIM, GP_MU, GP_F0, GP_F1, GP_F2, GP_F3, GP_F4,
 GP_RR1, GP_RR2, GP_RR3, GP_RR4, GP_RR5, GP_X1) = range(40)
# kind 2 (convex pair: the engine's MPR collider, csrc/lm_core.h mpr_contact): the P/A/H/R fields above hold the geom's BOUNDING
# capsule (exact for spheres and capsules, and for a cylinder's axis / radius / half length) for the mid phase; the extension
# GP_X1 / GP_X2 (GX_SIZE floats per geom) holds what the support function needs: engine geom type, the centre MPR starts from
# (link frame; a mesh's centre of mass), then for a box its half sizes and x / y axes (link frame), for a mesh the first vertex
# and the vertex count of its convex hull in the mesh-vertex table (link frame)
GX_TYPE, GX_CX, GX_CY, GX_CZ, GX_E0, GX_RBOUND, GX_SIZE = 0, 1, 2, 3, 4, 13, 14
GP_X2 = GP_X1 + GX_SIZE
GPAIR_SIZE = GP_X2 + GX_SIZE
# body-pair records (global memory, the level between a link pair and its geom pairs): the geoms of ONE body of either link as a
# bounding capsule (centre, axis, half length, radius in the link frames; "1" = the pair's first link), the geom pairs between the
# two bodies [BP_FIRST, BP_FIRST + BP_N) (<= 24) and their largest margin. A link pair's entry points at its body pairs
(BP_P1, BP_A1, BP_H1, BP_R1, BP_P2, BP_A2, BP_H2, BP_R2, BP_FIRST, BP_N, BP_MARGIN, BP_SIZE) = (0, 3, 6, 7, 8, 11, 14, 15, 16, 17, 18, 20)
CM_SIZE = ROOT_SIZE + CHAIN_SIZE * NCHAIN + MAXRG * U_SIZE + MAXG * (U_SIZE + P_SIZE) * NCHAIN + MAXLG * LG_SIZE * NCHAIN + MAXLP * LP_SIZE * NCHAIN
# ---- the geom table (global memory, read when a geom's bounding sphere reaches the floor): full records interleaved
# [geom][field][chain]
GT_SIZE = MAXG * G_SIZE * NCHAIN

GEOM_SUPPORTED = (mjcf.GEOM_SPHERE, mjcf.GEOM_CAPSULE, mjcf.GEOM_BOX, mjcf.GEOM_CYLINDER)     # + meshes that come with a convex hull
MINIMP, MAXIMP, MINVAL = 1e-4, 0.9999, 1e-15
MINMU = 1e-5          # mjMINMU: the engine's floor for a friction coefficient (applied when a contact is created)


class UnsupportedModel(ValueError):
    pass


def _kb(solref, solimp, timestep):
    dmax = min(MAXIMP, max(MINIMP, solimp[1]))
    if solref[0] > 0:
        tc = max(solref[0], 2 * timestep)
        return 1.0 / max(MINVAL, dmax * dmax * tc * tc * solref[1] * solref[1]), 2.0 / max(MINVAL, dmax * tc)
    return -solref[0] / max(MINVAL, dmax * dmax), -solref[1] / max(MINVAL, dmax)


def _clip_solimp(s):
    return [min(MAXIMP, max(MINIMP, s[0])), min(MAXIMP, max(MINIMP, s[1])), max(0.0, s[2]),
            min(MAXIMP, max(MINIMP, s[3])), max(1.0, s[4])]


def _mix_with_floor(m, g, gf):
    """MuJoCo's contact-parameter combination of geom ``g`` with the floor plane ``gf`` (priority wins,
    else max condim / max friction / solmix-weighted solref+solimp); margin = max, gap = max."""
    p, pf = m.geom_priority[g], m.geom_priority[gf]
    if p != pf:
        w = g if p > pf else gf
        dim, solref, solimp, fr = m.geom_condim[w], m.geom_solref[w], m.geom_solimp[w], m.geom_friction[w]
    else:
        dim = max(m.geom_condim[g], m.geom_condim[gf])
        s1, s2 = m.geom_solmix[g], m.geom_solmix[gf]
        mix = s1 / (s1 + s2) if (s1 >= MINVAL and s2 >= MINVAL) else (0.5 if (s1 < MINVAL and s2 < MINVAL) else
                                                                        (0.0 if s1 < MINVAL else 1.0))
        r1, r2 = m.geom_solref[g], m.geom_solref[gf]
        solref = mix * r1 + (1 - mix) * r2 if (r1[0] > 0 and r2[0] > 0) else np.minimum(r1, r2)
        solimp = mix * m.geom_solimp[g] + (1 - mix) * m.geom_solimp[gf]
        fr = np.maximum(m.geom_friction[g], m.geom_friction[gf])
    friction = [fr[0], fr[0], fr[1], fr[2], fr[2]]
    margin = max(m.geom_margin[g], m.geom_margin[gf])
    gap = max(m.geom_gap[g], m.geom_gap[gf])
    return int(dim), np.asarray(solref, float), np.asarray(solimp, float), friction, margin, gap


def _fill_contact_params(blk, m, b, dim, fr):
    """cone-dependent floor-contact constants of a geom on body ``b`` (G_DIM, G_F*, G_MU, G_TRAN, G_RR*)."""
    tran = m.body_invweight0[b, 0] + m.body_invweight0[0, 0]
    fr = np.maximum(np.asarray(fr, dtype=np.float64), MINMU)      # the engine clamps friction at contact creation (mjMINMU): a randomised friction
                                                                  # drawn at exactly 0 would make rr = 0 / 0 below (csrc/lm_compile.hip does the same)
    blk[G_DIM] = dim
    blk[G_F0:G_F0 + 5] = fr
    if m.cone == mjcf.CONE_ELLIPTIC:
        if dim not in (1, 3, 4, 6):
            raise UnsupportedModel("condim %d" % dim)
        blk[G_TRAN] = tran
        blk[G_MU] = fr[0] / np.sqrt(max(MINVAL, m.impratio))
        rr1 = 1.0 / max(MINVAL, m.impratio)
        blk[G_RR1], blk[G_RR2] = rr1, rr1 * fr[0] * fr[0] / (fr[1] * fr[1])
        blk[G_RR3:G_RR3 + 3] = [rr1 * fr[0] * fr[0] / (fr[k] * fr[k]) for k in (2, 3, 4)]
    else:
        if dim not in (1, 3):
            raise UnsupportedModel("pyramidal condim %d is not built on the device" % dim)
        if fr[0] != fr[1]:
            raise UnsupportedModel("anisotropic sliding friction")
        mu = fr[0]
        blk[G_MU] = mu
        # every edge of the pyramid: diagApprox = (1+mu^2) tran, shared regulariser Rpy = 2 mu^2 R
        blk[G_TRAN] = 2 * mu * mu * (1 + mu * mu) * tran if dim == 3 else tran
        blk[G_RR1:G_RR1 + 5] = 1.0


HEADER_SIZE = 56
LMC_MAGIC = 0x4C4D4332  # "LMC2"
(H_MAGIC, H_VERSION, H_NV, H_NU, H_NCHAINS, H_MAXLINKS, H_TIMESTEP, H_GX, H_GY, H_GZ, H_IMPRATIO, H_ITERATIONS,
 H_TOLERANCE, H_NSUBSTEPS, H_NOBS, H_NGOAL, H_REWARD_TYPE, H_REWARD_P0) = range(18)
H_NGRF, H_MEANINERTIA, H_CM_SIZE, H_INTEGRATOR, H_CONE, H_MAXCONTACTS, H_NMUSCLE, H_CM_USED, H_ACTMODE = 25, 26, 27, 28, 29, 30, 31, 32, 33
H_OFF_RUNSUP, H_OFF_CUNSUP, H_OFF_PRUNE, H_GT_SIZE, H_OFF_LPAIR, H_NGPAIR, H_OFF_GPT, H_OFF_LGROUP, H_NMESHV, H_OFF_MESHV = 34, 35, 36, 37, 38, 39, 40, 41, 42, 43
H_NMESHN, H_OFF_MESHN = 44, 45     # neighbour table of the hull vertices (floats: hull-local indices, -1 ends a vertex's list)
H_NBPAIR, H_OFF_BPT = 46, 47       # body-pair table of the self-collision mid phase (BP_SIZE floats per record)
H_NMESHADJ, H_OFF_MESHADJ = 48, 49  # adjacency blocks of the hull vertices (4 floats per entry): convex-pair collider
# H_NGPAIR geom-pair records start H_OFF_GPT floats into the chain-model array (behind the geom table and the muscle table)
# H_OFF_*: offsets (floats from the start of the constant table) of the tail lists, see CM_SIZE
# H_ACTMODE: 0 = joint motors (torque = gear * ctrl), 1 = position servos on every actuated joint
# H_CM_USED: floats of the constant table that are actually read (through the tail lists)
# H_NGRF: number of ground-reaction-force observation entries (3 per force group); they follow the goal entries

# ---- muscle table (optional; follows the constant table): MT_HEAD floats [first muscle of chain c] x NCHAIN,
# [muscle count of chain c] x NCHAIN, then MU_SIZE floats per muscle (sorted by chain), then 4 floats per tendon
# path entry (link index in the chain, -1 = root body; xyz in that link's frame)
MAXMUS = 48           # muscles per chain (lane memory holds their activation and control)
MT_MAXMUS = 96        # muscles per model
MT_MAXSITE = 352      # tendon path entries per model
(MU_ACT, MU_ACT_MEAN, MU_ACT_DELTA, MU_CTRL_LO, MU_CTRL_HI, MU_GEAR, MU_LR0, MU_INV_L0, MU_RANGE0, MU_FORCE, MU_LMIN,
 MU_LMAX, MU_INV_L0VMAX, MU_FPMAX, MU_FVMAX, MU_TAU_ACT, MU_TAU_DEACT, MU_SITE_ADR, MU_SITE_NUM, MU_STATE,
 MU_SIZE) = range(21)
MT_HEAD = 2 * NCHAIN
MT_SITES = MT_HEAD + MT_MAXMUS * MU_SIZE
MT_SIZE = MT_SITES + MT_MAXSITE * 4
SRC_ROOT_QVEL, SRC_GOAL, SRC_ROOT_QPOS = 0, 100, 200


def _merge_proximity_spheres(spheres, cap):
    """Reduce [link, x, y, z, r, margin] proximity spheres to at most `cap` by repeatedly replacing the two
    spheres of one link whose enclosing sphere is smallest by that enclosing sphere (conservative)."""
    spheres = [list(map(float, u)) for u in spheres]
    while len(spheres) > cap:
        best = None
        for i in range(len(spheres)):
            for j in range(i + 1, len(spheres)):
                a, b = spheres[i], spheres[j]
                if a[0] != b[0]:
                    continue
                d = np.linalg.norm(np.subtract(b[1:4], a[1:4]))
                r = max(a[4], b[4], 0.5 * (d + a[4] + b[4]))
                if best is None or r < best[0]:
                    best = (r, i, j, d)
        if best is None:
            raise UnsupportedModel("too many geoms without a device collider")
        r, i, j, d = best
        a, b = spheres[i], spheres[j]
        if r == a[4]:
            c = a[1:4]
        elif r == b[4]:
            c = b[1:4]
        else:
            c = np.add(a[1:4], np.subtract(b[1:4], a[1:4]) * ((r - a[4]) / d))
        merged = [a[0], c[0], c[1], c[2], r, max(a[5], b[5])]
        spheres = [u for k, u in enumerate(spheres) if k not in (i, j)] + [merged]
    return spheres


def lower(m, task):
    """
    ``task``: dict(nobs, qpos_obs_idx, qvel_obs_idx, n_goal, act_ctrl_idx, act_mean, act_delta,
    term=[(obs idx, lo, hi)], reward_type, reward_params, n_substeps) — see ``LocoEnv._device_task``.
    Returns ``(chain_model float64[HEADER_SIZE + CM_SIZE], info dict)``; raises :class:`UnsupportedModel`.
    """
    qobs = {int(d): i for i, d in enumerate(task["qpos_obs_idx"])}
    nq_obs = len(task["qpos_obs_idx"])
    vobs = {int(d): nq_obs + i for i, d in enumerate(task["qvel_obs_idx"])}
    action_of_act = {int(a): k for k, a in enumerate(task["act_ctrl_idx"])}
    obs_src = {}
    for d, i in qobs.items():
        obs_src[i] = ("q", d)
    for d, i in vobs.items():
        obs_src[i] = ("v", d)
    n_grf_obs = 3 * len(task.get("grf_groups") or [])        # foot forces end the observation, the goal sits before them
    for i in range(task["n_goal"]):
        obs_src[task["nobs"] - n_grf_obs - task["n_goal"] + i] = ("g", i)
    for i in range(n_grf_obs):
        obs_src[task["nobs"] - n_grf_obs + i] = ("f", i)
    term_q, term_v = {}, {}
    for idx, lo, hi in task["term"]:
        kind, d = obs_src[int(idx)]
        if kind in ("g", "f"):
            raise UnsupportedModel("termination on a goal / foot-force entry")
        (term_q if kind == "q" else term_v)[d] = (max(lo, -3e38), min(hi, 3e38))
    nb = m.nbody
    jointed = [b for b in range(1, nb) if m.body_jntnum[b] > 0]
    roots = [b for b in jointed if m.body_weldid[m.body_parent[b]] == 0]
    if len(roots) != 1:
        raise UnsupportedModel("exactly one jointed root body expected")
    root = roots[0]
    if m.body_parent[root] != 0:
        raise UnsupportedModel("root body must be a child of the world")
    if m.body_jntnum[root] > NROOT:
        raise UnsupportedModel("root body has more than %d dofs" % NROOT)

    # jointed children of each jointed body (through welded bodies)
    def jointed_parent(b):
        return m.body_weldid[m.body_parent[b]]
    children = {b: [c for c in jointed if c != b and jointed_parent(c) == b] for b in jointed}
    chains = []
    shared_first = {}                  # chain index of a copy -> chain index of the owner of its (shared) first link
    for start in children[root]:
        heads = [[start]]
        if len(children[start]) == 2 and m.body_jntnum[start] == 1:
            # ONE branch, right behind a one-dof first link (a torso joint with two arms): two chains that share that link.
            # The first owns the dof, the link's mass and geoms; the second carries a massless copy of it; the kernel ties the two
            # copies of the dof together in every solve (csrc/lm_core.h, arrow_solve_shared)
            heads = [[start, children[start][0]], [start, children[start][1]]]
            shared_first[len(chains) + 1] = len(chains)
        for chain in heads:
            b = chain[-1]
            while True:
                if len(children[b]) == 0:
                    break
                if len(children[b]) > 1:
                    raise UnsupportedModel("branching below the root is not supported (except one branch right behind a one-dof first link)")
                b = children[b][0]
                chain.append(b)
            chains.append(chain)
    if len(chains) > NCHAIN:
        raise UnsupportedModel("more than %d chains" % NCHAIN)

    kin = mjcf.forward_kinematics(m, m.qpos0)

    # ---- weld jointless bodies into their jointed ancestor: pose of every body relative to its weld body at qpos0
    def rel_pose(b):
        w = m.body_weldid[b]
        rw = kin["xmat"][w]
        return rw.T @ (kin["xpos"][b] - kin["xpos"][w]), rw.T @ kin["xmat"][b]

    def merged_inertial(w):
        """mass, com, inertia (about com) of weld group ``w`` in w's frame."""
        members = [b for b in range(1, nb) if m.body_weldid[b] == w]
        mass = sum(m.body_mass[b] for b in members)
        if mass <= 0:
            return 0.0, np.zeros(3), np.zeros((3, 3))
        coms, rots = {}, {}
        for b in members:
            p, r = rel_pose(b)
            coms[b], rots[b] = p + r @ m.body_ipos[b], r
        com = sum(m.body_mass[b] * coms[b] for b in members) / mass
        inertia = np.zeros((3, 3))
        for b in members:
            d = coms[b] - com
            inertia += rots[b] @ m.body_inertia[b] @ rots[b].T + m.body_mass[b] * (d @ d * np.eye(3) - np.outer(d, d))
        return mass, com, inertia

    floor = [g for g in range(m.ngeom) if m.geom_type[g] == mjcf.GEOM_PLANE]
    if len(floor) != 1 or m.geom_body[floor[0]] != 0:
        raise UnsupportedModel("exactly one floor plane on the world body expected")
    gf = floor[0]
    if np.abs(m.geom_pos[gf]).max() > 0 or np.abs(m.geom_quat[gf] - [1, 0, 0, 0]).max() > 1e-12:
        raise UnsupportedModel("floor plane must be z=0")

    cm = np.zeros(CM_SIZE, dtype=np.float64)
    act_of_dof = {int(d): a for a, d in enumerate(m.act_dof)}
    kinds = set(int(k) for k, d in zip(getattr(m, "act_kind", np.zeros(m.nu)), m.act_dof) if d >= 0)
    if kinds - {mjcf.ACT_MOTOR, mjcf.ACT_POSITION} or len(kinds) > 1:
        raise UnsupportedModel("joint actuators must be all motors or all position servos")
    act_mode = 1 if kinds == {mjcf.ACT_POSITION} else 0
    if act_mode == 1 and any(act_of_dof.get(int(d), -1) >= 0 for d in range(6)):
        raise UnsupportedModel("position servos on the root joints")
    dof_to_lane = -np.ones(m.nv, dtype=np.int64)

    dropped_root_limits = []

    def fill_dof(block, d, qobs, vobs, is_root=False):
        block[D_TYPE] = m.jnt_type[d]
        block[D_AX:D_AX + 3] = m.jnt_axis[d]
        block[D_PX:D_PX + 3] = m.jnt_pos[d]
        block[D_DAMP], block[D_ARM], block[D_STIFF] = m.dof_damping[d], m.dof_armature[d], m.jnt_stiffness[d]
        fl = m.dof_frictionloss[d]
        block[D_FLOSS] = fl
        if fl > 0 or m.dof_invweight0[d] > 0:            # also for fl == 0: a per-environment frictionloss may switch the row on
            si = _clip_solimp(m.dof_solimp[d])
            d0 = si[0] if not (si[0] == si[1] or si[2] <= MINVAL) else 0.5 * (si[0] + si[1])
            block[D_FLOSS_R] = max(MINVAL, (1 - d0) * m.dof_invweight0[d] / d0)
            block[D_FLOSS_B] = _kb(m.dof_solref[d], m.dof_solimp[d], m.timestep)[1]
        limited = bool(m.jnt_limited[d])
        if limited and is_root and getattr(m, "na", 0) > 0:
            pass        # the muscle families carry limit rows for the root dofs (csrc/lm_core.h ROOT_LIM): HumanoidMuscle's pelvis joints
        elif limited and is_root:
            # the regular kernels of the other families have no limit rows for the (replicated) root dofs: a root dof beyond its range
            # hands the control step to the family's replay kernel, which has them (csrc/lm_core.h ROOT_LIM). A root limit that cannot
            # become active is dropped, so that the robots of the path never take that detour: translation ranges of tens of metres, or
            # angles whose termination band (evaluated every control step) lies strictly inside the joint range.
            lo, hi = m.jnt_range[d]
            tlo, thi = term_q.get(int(d), (-np.inf, np.inf))
            if (m.jnt_type[d] == 0 and min(-lo, hi) >= 50.0) or (tlo > lo and thi < hi):
                limited = False
                dropped_root_limits.append(int(d))
        block[D_LIMITED] = limited
        block[D_LO], block[D_HI] = m.jnt_range[d]
        if limited:
            assert m.jnt_margin[d] == 0, "joint margin != 0 not supported on the device"
            block[D_LIM_K], block[D_LIM_B] = _kb(m.jnt_solref[d], m.jnt_solimp[d], m.timestep)
            block[D_LIM_S0:D_LIM_S0 + 5] = _clip_solimp(m.jnt_solimp[d])
        block[D_INVW] = m.dof_invweight0[d]
        a = act_of_dof.get(int(d), -1)
        block[D_ACT] = -1
        block[D_FLO], block[D_FHI] = -3e38, 3e38
        if a >= 0:
            block[D_GEAR] = m.act_gear[a]
            if act_mode == 1:
                # position servo: force = kp * ctrl - kp * q (gain [kp], bias [0, -kp, 0]), clamped, times gear = 1
                kp = m.act_gainprm[a][0]
                if m.act_gear[a] != 1.0 or abs(m.act_biasprm[a][1] + kp) > 0 or m.act_biasprm[a][0] != 0 or m.act_biasprm[a][2] != 0:
                    raise UnsupportedModel("position servo %s: only gear 1 and bias [0, -kp, 0] are lowered" % m.act_names[a])
                block[D_GEAR] = kp
                if m.act_forcelimited[a]:
                    block[D_FLO], block[D_FHI] = m.act_forcerange[a]
            lo, hi = (m.act_ctrlrange[a] if m.act_ctrllimited[a] else (-np.inf, np.inf))
            block[D_CTRL_LO], block[D_CTRL_HI] = max(lo, -3e38), min(hi, 3e38)
            k = action_of_act.get(a, -1)
            block[D_ACT] = k
            if k >= 0:
                block[D_ACT_MEAN], block[D_ACT_DELTA] = task["act_mean"][k], task["act_delta"][k]
        block[D_QOBS], block[D_VOBS] = qobs.get(int(d), -1), vobs.get(int(d), -1)
        block[D_TERM_QLO], block[D_TERM_QHI] = term_q.get(int(d), (-3e38, 3e38))
        block[D_TERM_VLO], block[D_TERM_VHI] = term_v.get(int(d), (-3e38, 3e38))
        block[D_DOF] = d

    # ground-reaction-force groups (foot-force observations): geom id -> group id in observation order
    grf_groups = task.get("grf_groups") or []
    grf_of_geom = {}
    for gi, names in enumerate(grf_groups):
        for nm in names:
            grf_of_geom[m.geom_names.index(nm)] = gi
    n_grf = 3 * len(grf_groups)
    grf_obs_base = task["nobs"] - n_grf

    blk_geom = {}                      # id(geom block) -> model geom (the blocks live until the table is written)

    def geom_blocks(w, link_index):
        """floor-collidable geoms of weld group w: (supported blocks, unsupported blocks)."""
        sup, unsup = [], []
        for g in range(m.ngeom):
            b = m.geom_body[g]
            if b == 0 or m.body_weldid[b] != w:
                continue
            if not ((m.geom_contype[g] & m.geom_conaffinity[gf]) or (m.geom_contype[gf] & m.geom_conaffinity[g])):
                continue
            p, r = rel_pose(b)
            gpos = p + r @ m.geom_pos[g]
            grot = r @ mjcf.quat_to_mat(m.geom_quat[g])
            dim, solref, solimp, fr, margin, gap = _mix_with_floor(m, g, gf)
            assert gap == 0, "contact gap != 0 not supported on the device"
            t, size = m.geom_type[g], m.geom_size[g]
            rbound = {mjcf.GEOM_SPHERE: size[0], mjcf.GEOM_CAPSULE: size[0] + size[1], mjcf.GEOM_MESH: size[0] + size[1],
                      mjcf.GEOM_CYLINDER: np.hypot(size[0], size[1]), mjcf.GEOM_BOX: np.linalg.norm(size)}[t]
            hull_n = int(getattr(m, "geom_hull_num", np.zeros(m.ngeom, int))[g])
            if t == mjcf.GEOM_MESH and hull_n > 0:
                # plane vs convex hull: a contact at the support vertex (+ its neighbours). The hull's vertices go into the
                # mesh-vertex table in the frame of the LINK; the prune sphere is the hull's bounding sphere
                first_vert, hv = register_hull(g)
                ctr = 0.5 * (hv.min(0) + hv.max(0))
                blk = np.zeros(G_SIZE)
                blk[G_GRF] = grf_of_geom.get(g, -1)
                blk[G_LINK], blk[G_TYPE] = link_index, t
                blk[G_PX:G_PX + 3] = ctr
                blk[G_R0:G_R0 + 9] = np.eye(3).reshape(9)
                blk[G_RBOUND], blk[G_MARGIN] = float(np.linalg.norm(hv - ctr, axis=1).max()) * (1 + 1e-6), margin
                blk[G_SX], blk[G_SY] = first_vert, hull_n                # first vertex, vertex count in the mesh-vertex table
                # further contacts at the hull-graph neighbours of the support vertex keep this far from the support contact
                # (0.3 x the bounding capsule's radius + half length: DESIGN.md §2 item 10)
                blk[G_SZ] = 0.3 * (size[0] + size[1])
                blk[G_K], blk[G_B] = _kb(solref, solimp, m.timestep)
                blk[G_S0:G_S0 + 5] = _clip_solimp(solimp)
                _fill_contact_params(blk, m, b, dim, fr)
                blk_geom[id(blk)] = g
                sup.append(blk)
                continue
            if t == mjcf.GEOM_MESH:
                # proximity-only bounding capsule (mjcf.compile_mjcf): the floor is within reach of a capsule exactly
                # when it is within reach of one of its two end spheres
                ends = [gpos] if size[1] == 0 else [gpos - size[1] * grot[:, 2], gpos + size[1] * grot[:, 2]]
                unsup += [[link_index, e[0], e[1], e[2], size[0], margin] for e in ends]
                continue
            if t not in GEOM_SUPPORTED:
                unsup.append([link_index, gpos[0], gpos[1], gpos[2], rbound, margin])
                continue
            blk = np.zeros(G_SIZE)
            blk[G_GRF] = -1
            if g in grf_of_geom:
                blk[G_GRF] = grf_of_geom[g]       # provisional: global group id, turned into the chain's slot below
            blk[G_LINK], blk[G_TYPE] = link_index, t
            blk[G_PX:G_PX + 3] = gpos
            blk[G_AX:G_AX + 3] = grot[:, 2]
            blk[G_SX:G_SX + 3] = size
            blk[G_R0:G_R0 + 9] = grot.reshape(9)
            blk[G_RADIUS], blk[G_HALF], blk[G_RBOUND], blk[G_MARGIN] = size[0], (size[1] if t in (mjcf.GEOM_CAPSULE, mjcf.GEOM_CYLINDER) else 0), rbound, margin
            blk[G_K], blk[G_B] = _kb(solref, solimp, m.timestep)
            blk[G_S0:G_S0 + 5] = _clip_solimp(solimp)
            _fill_contact_params(blk, m, b, dim, fr)
            blk_geom[id(blk)] = g
            sup.append(blk)
        return sup, unsup

    info = dict(root=root, chains=chains, shared_first=shared_first)
    # provenance of everything a model VARIANT changes (model_compiler_tables): record rows of the dofs [(lane or -1, link or root dof
    # index, dof, copy)], weld group behind every inertial slot [(lane or -1, link, body)], model geom behind every geom-table record
    # [(slot, lane, geom)], geoms behind every geom-pair record [(geom a, geom b)]
    prov = info["variant_provenance"] = dict(dofs=[], links=[(-1, -1, root)], geoms=[], pairs=[])
    mesh_verts = []                    # hull vertices of the mesh colliders, link frame (device table, global memory)
    mesh_nbr_first, mesh_nbr = [], []  # per vertex: start of its neighbour list in the neighbour table (hull-local indices, -1 ends)
    hull_slot = {}                     # geom -> (first vertex in the table, vertices in the link frame)
    mesh_adj, hull_block = [], {}      # adjacency blocks (4 floats per entry), geom -> block start of every hull vertex

    def register_hull(g):
        """The convex hull of mesh geom g in the mesh-vertex table (once per geom): its vertices in the frame of the geom's LINK
        (plane-hull contacts and the convex-pair collider read them there), each with the start of its neighbour list."""
        if g not in hull_slot:
            p_, r_ = rel_pose(m.geom_body[g])
            a0, n_ = int(m.geom_hull_adr[g]), int(m.geom_hull_num[g])
            hv_ = p_ + m.hull_vert[a0:a0 + n_].astype(np.float64) @ r_.T
            hull_slot[g] = (len(mesh_verts), hv_)
            nbrs = []
            for i in range(n_):
                e0, e1 = int(m.hull_nbr_adr[a0 + i]), int(m.hull_nbr_adr[a0 + i + 1])
                mesh_nbr_first.append(len(mesh_nbr))                     # 4th component of the vertex: its neighbour list ...
                nbrs.append([int(j) for j in m.hull_nbr[e0:e1]])
                mesh_nbr.extend(nbrs[-1])
                mesh_nbr.append(-1)                                      # ... which a -1 ends
            mesh_verts.extend(hv_.tolist())
            # adjacency blocks for the hill climbing of the convex-pair collider: per vertex [x y z degree] then one [x y z block] per
            # neighbour (block = where that neighbour's own block starts, in 4-float entries): a step reads ONE contiguous block
            starts, pos = [], len(mesh_adj)
            for i in range(n_):
                starts.append(pos)
                pos += 1 + len(nbrs[i])
            for i in range(n_):
                mesh_adj.append([hv_[i][0], hv_[i][1], hv_[i][2], len(nbrs[i])])
                mesh_adj.extend([hv_[j][0], hv_[j][1], hv_[j][2], starts[j]] for j in nbrs[i])
            hull_block[g] = starts
        return hull_slot[g]

    # ---- root block
    rb = cm[CM_ROOT:CM_ROOT + ROOT_SIZE]
    rb[R_NDOF] = m.body_jntnum[root]
    rb[R_TX:R_TX + 3] = m.body_pos[root]
    rb[R_R0:R_R0 + 9] = mjcf.quat_to_mat(m.body_quat[root]).reshape(9)
    mass, com, inertia = merged_inertial(root)
    rb[R_MASS], rb[R_CX:R_CX + 3] = mass, com
    rb[R_IXX:R_IXX + 6] = [inertia[0, 0], inertia[1, 1], inertia[2, 2], inertia[0, 1], inertia[0, 2], inertia[1, 2]]
    for k in range(m.body_jntnum[root]):
        d = m.body_jntadr[root] + k
        fill_dof(rb[R_DOFS + k * D_SIZE:R_DOFS + (k + 1) * D_SIZE], d, qobs, vobs, is_root=True)
        dof_to_lane[d] = -2
        prov["dofs"].append((-1, k, int(d), False))
    sup, unsup = geom_blocks(root, 0)
    if any(sb[G_GRF] >= 0 for sb in sup):
        raise UnsupportedModel("foot-force group on the root body")
    root_geoms = sup                   # colliders of the root body: dealt to the chains' lanes below (link -1)
    for gb in root_geoms:
        gb[G_LINK] = -1
    root_unsup = unsup                 # + colliders that find no geom slot (below); merged when the tail is written
    gt = np.zeros(GT_SIZE, dtype=np.float64)
    chain_unsup, chain_prune, chain_groups_tab = [[] for _ in range(NCHAIN)], [[] for _ in range(NCHAIN)], [[] for _ in range(NCHAIN)]

    # ---- chains, interleaved [field][chain]
    for c in range(NCHAIN):                      # unused lanes report no foot force
        for k in C_GRF_OBS:
            cm[CM_CHAINS + k * NCHAIN + c] = -1
    max_links = 0
    max_groups_seen = 0
    standing = []
    # every lane of the quad takes its share of the ROOT body's colliders, also the lanes without a chain (a two-legged robot
    # without arm chains leaves two lanes idle: their contact slots and their share of the floor pass are free capacity)
    for c in range(NCHAIN):
        chain = chains[c] if c < len(chains) else []
        blk = np.zeros(CHAIN_SIZE)
        links = []
        for b in chain:
            for k in range(m.body_jntnum[b]):
                links.append((b, m.body_jntadr[b] + k, k == 0, k == m.body_jntnum[b] - 1))
        if len(links) > MAXC:
            raise UnsupportedModel("chain with more than %d dofs" % MAXC)
        max_links = max(max_links, len(links))
        blk[C_NLINKS] = len(links)
        geoms, unsup = [], []
        blk[C_DUPROLE] = -1 if c in shared_first else (1 if c in shared_first.values() else 0)
        for li, (b, d, first, last) in enumerate(links):
            lb = blk[C_LINKS + li * LINK_SIZE:C_LINKS + (li + 1) * LINK_SIZE]
            copy = c in shared_first and li == 0
            fill_dof(lb[:D_SIZE], d, qobs, vobs)
            prov["dofs"].append((c, li, int(d), bool(copy)))
            if copy:
                # the massless copy of the shared link: same joint and dof index (both lanes load the same q, v), but everything
                # that acts ON the dof — damping, stiffness, armature, friction loss, limit, motor, observation, termination —
                # belongs to the owner's record
                for f in (D_DAMP, D_ARM, D_STIFF, D_FLOSS, D_LIMITED, D_GEAR):      # (the friction-loss regulariser stays: a row with
                    lb[f] = 0.0                                                     # frictionloss 0 is inactive, its R must not be 0)
                lb[D_ACT], lb[D_QOBS], lb[D_VOBS] = -1, -1, -1
                lb[D_TERM_QLO], lb[D_TERM_QHI], lb[D_TERM_VLO], lb[D_TERM_VHI] = -3e38, 3e38, -3e38, 3e38
            else:
                dof_to_lane[d] = c
            ex = lb[D_SIZE:]
            if first:
                # pose of body b relative to its jointed parent's frame (through welded ancestors), at qpos0
                pw = jointed_parent(b)
                rp = kin["xmat"][pw]
                ex[L_HAS_T] = 1
                ex[L_TX:L_TX + 3] = rp.T @ (kin["xpos"][b] - kin["xpos"][pw])
                ex[L_R0:L_R0 + 9] = (rp.T @ kin["xmat"][b]).reshape(9)
            else:
                ex[L_R0:L_R0 + 9] = np.eye(3).reshape(9)
            if last and not copy:
                mass, com, inertia = merged_inertial(b)
                prov["links"].append((c, li, int(b)))
                ex[L_MASS], ex[L_CX:L_CX + 3] = mass, com
                ex[L_IXX:L_IXX + 6] = [inertia[0, 0], inertia[1, 1], inertia[2, 2], inertia[0, 1], inertia[0, 2],
                                       inertia[1, 2]]
                s, u = geom_blocks(b, li)
                geoms += s
                unsup += u
        geoms += root_geoms[c::NCHAIN]               # this lane's share of the root body's colliders (after the chain's own)
        for k in C_GRF_OBS:
            blk[k] = -1
        chain_groups = sorted(set(int(gb[G_GRF]) for gb in geoms if gb[G_GRF] >= 0))
        max_groups_seen = max(max_groups_seen, len(chain_groups))
        if len(chain_groups) > 4:
            raise UnsupportedModel("more than four foot-force groups on one chain")
        for slot, gi in enumerate(chain_groups):
            blk[C_GRF_OBS[slot]] = grf_obs_base + 3 * gi
        for gb in geoms:
            gb[G_GRF] = chain_groups.index(int(gb[G_GRF])) if gb[G_GRF] >= 0 else -1
        if len(geoms) > MAXG:
            # more colliders than geom slots: the surplus, in model order, becomes proximity-only (bounding sphere,
            # counted in `unhandled_geoms` when it reaches the floor)
            info.setdefault("demoted_geoms", []).append((c, len(geoms) - MAXG))
            unsup += [[gb[G_LINK], gb[G_PX], gb[G_PY], gb[G_PZ], gb[G_RBOUND], gb[G_MARGIN]] for gb in geoms[MAXG:] if gb[G_LINK] >= 0]
            root_unsup += [[0, gb[G_PX], gb[G_PY], gb[G_PZ], gb[G_RBOUND], gb[G_MARGIN]] for gb in geoms[MAXG:] if gb[G_LINK] < 0]
            geoms = geoms[:MAXG]
        unsup = _merge_proximity_spheres(unsup, MAXG)
        blk[C_NGEOMS], blk[C_NUNSUP] = len(geoms), len(unsup)
        # contact slots the chain needs in regular operation: what the geoms that stand on the floor in the reference pose
        # (qpos0; bottom of the bounding sphere within 3 cm of the lowest one of the model) can produce. Geoms further up
        # only reach the floor once the robot has fallen; contacts beyond the kernel's slots are dropped and counted in
        # `overflow_contacts`
        for gb in geoms:
            cap = {mjcf.GEOM_SPHERE: 1, mjcf.GEOM_CAPSULE: 2, mjcf.GEOM_BOX: 4, mjcf.GEOM_CYLINDER: 4, mjcf.GEOM_MESH: 4}[int(gb[G_TYPE])]
            b = links[int(gb[G_LINK])][0] if gb[G_LINK] >= 0 else root
            rot = kin["xmat"][b] @ gb[G_R0:G_R0 + 9].reshape(3, 3)            # geom axes in the world at qpos0
            t, size = int(gb[G_TYPE]), gb[G_SX:G_SX + 3]
            if t == mjcf.GEOM_MESH:
                hv = np.array(mesh_verts[int(size[0]):int(size[0]) + int(size[1])])
                reach = -float(((hv - gb[G_PX:G_PX + 3]) @ kin["xmat"][b].T)[:, 2].min())
            else:
                reach = {mjcf.GEOM_SPHERE: size[0], mjcf.GEOM_CAPSULE: size[0] + size[1] * abs(rot[2, 2]),
                         mjcf.GEOM_CYLINDER: size[1] * abs(rot[2, 2]) + size[0] * np.sqrt(max(0.0, 1 - rot[2, 2] ** 2)),
                         mjcf.GEOM_BOX: float(np.abs(rot[2]) @ size)}[t]
            bottom = (kin["xpos"][b] + kin["xmat"][b] @ gb[G_PX:G_PX + 3])[2] - reach
            standing.append((c, bottom, cap))
        for i, gblk in enumerate(geoms):
            gt[(i * G_SIZE + np.arange(G_SIZE)) * NCHAIN + c] = gblk
            prov["geoms"].append((i, c, blk_geom[id(gblk)]))
            chain_prune[c].append([gblk[G_LINK], gblk[G_PX], gblk[G_PY], gblk[G_PZ], gblk[G_RBOUND], gblk[G_TYPE]])
        # the geoms of one link form a group with a bounding sphere: a link far above the floor costs one test per pass
        i = 0
        while i < len(geoms):
            j = i
            while j < len(geoms) and geoms[j][G_LINK] == geoms[i][G_LINK]:
                j += 1
            ctr = np.mean([gb[G_PX:G_PX + 3] for gb in geoms[i:j]], axis=0)
            rad = max(np.linalg.norm(gb[G_PX:G_PX + 3] - ctr) + gb[G_RBOUND] for gb in geoms[i:j])
            chain_groups_tab[c].append([geoms[i][G_LINK], i, j - i, ctr[0], ctr[1], ctr[2], rad])
            i = j
        assert len(chain_groups_tab[c]) <= MAXLG
        blk[C_NLGROUP] = len(chain_groups_tab[c])
        chain_unsup[c] = unsup
        cm[CM_CHAINS + np.arange(CHAIN_SIZE) * NCHAIN + c] = blk

    if (dof_to_lane == -1).any():
        raise UnsupportedModel("some dofs are outside the root+chains structure")
    if shared_first:
        # a first link shared by two chains (tie_shared_dof) is compiled in the six-link kernels only: a robot with shorter chains
        # (UnitreeH1 with its torso joint and free arms: 1 + 4 links) runs there with an idle link slot per chain
        max_links = MAXC

    # ---- muscles: every tendon must run over the root body and ONE chain, so that lane c owns it
    mt = None
    muscles = [a for a in range(m.nu) if getattr(m, "act_kind", np.zeros(m.nu))[a] == mjcf.ACT_MUSCLE]
    if muscles:
        if m.integrator != mjcf.INT_EULER:
            raise UnsupportedModel("muscle activations under RK4 are not built on the device")
        weld_link = {root: (-1, -1)}
        for c, chain in enumerate(chains):
            li = -1
            for b in chain:
                li += m.body_jntnum[b]
                weld_link[b] = (c, li)                       # frame after the body's last joint
        per_chain = [[] for _ in range(NCHAIN)]
        state_of = {a: i for i, a in enumerate(muscles)}      # activation state index = order among the muscles
        for a in muscles:
            t = m.act_tendon[a]
            path = []
            for s_id in m.wrap_site[m.tendon_adr[t]:m.tendon_adr[t] + m.tendon_num[t]]:
                b = m.site_body[s_id]
                w = m.body_weldid[b]
                if w not in weld_link:
                    raise UnsupportedModel("tendon site on a body outside the root+chains structure")
                p, r = rel_pose(b)
                path.append((weld_link[w], p + r @ m.site_pos[s_id]))
            lanes = sorted(set(cl[0] for cl, _ in path if cl[0] >= 0))
            if len(lanes) > 1:
                raise UnsupportedModel("tendon %s spans two chains" % m.tendon_names[t])
            if m.act_dynprm[a][2] != 0:
                raise UnsupportedModel("smoothed muscle time constants (tausmooth) are not built on the device")
            per_chain[lanes[0] if lanes else 0].append((a, path))
        if max(len(x) for x in per_chain) > MAXMUS or len(muscles) > MT_MAXMUS:
            raise UnsupportedModel("too many muscles")
        mt = np.zeros(MT_SIZE)
        n_mus = n_site = 0
        for c in range(NCHAIN):
            mt[c], mt[NCHAIN + c] = n_mus, len(per_chain[c])
            for a, path in per_chain[c]:
                rec = mt[MT_HEAD + n_mus * MU_SIZE:MT_HEAD + (n_mus + 1) * MU_SIZE]
                k = action_of_act.get(a, -1)
                rec[MU_ACT] = k
                if k >= 0:
                    rec[MU_ACT_MEAN], rec[MU_ACT_DELTA] = task["act_mean"][k], task["act_delta"][k]
                lo, hi = (m.act_ctrlrange[a] if m.act_ctrllimited[a] else (-np.inf, np.inf))
                rec[MU_CTRL_LO], rec[MU_CTRL_HI] = max(lo, -3e38), min(hi, 3e38)
                prm, lr = m.act_gainprm[a], m.act_lengthrange[a]
                l0 = (lr[1] - lr[0]) / max(MINVAL, prm[1] - prm[0])
                rec[MU_GEAR], rec[MU_LR0], rec[MU_INV_L0], rec[MU_RANGE0] = m.act_gear[a], lr[0], 1.0 / max(MINVAL, l0), prm[0]
                rec[MU_FORCE], rec[MU_LMIN], rec[MU_LMAX] = prm[2], prm[4], prm[5]
                rec[MU_INV_L0VMAX], rec[MU_FPMAX], rec[MU_FVMAX] = 1.0 / max(MINVAL, l0 * prm[6]), prm[7], prm[8]
                rec[MU_TAU_ACT], rec[MU_TAU_DEACT] = m.act_dynprm[a][0], m.act_dynprm[a][1]
                rec[MU_SITE_ADR], rec[MU_SITE_NUM], rec[MU_STATE] = n_site, len(path), state_of[a]
                if n_site + len(path) > MT_MAXSITE:
                    raise UnsupportedModel("too many tendon path entries")
                for (cl, li), p in path:
                    mt[MT_SITES + 4 * n_site:MT_SITES + 4 * n_site + 4] = [li, p[0], p[1], p[2]]
                    n_site += 1
                n_mus += 1
        info["muscles_per_chain"] = [len(x) for x in per_chain]
    elif getattr(m, "na", 0):
        raise UnsupportedModel("activation states without muscles")

    lowest = min([b for _, b, _ in standing], default=0.0)
    # "stands on the floor": within 2 % of the robot's height of the lowest geom bottom (3 cm for a 1.5 m humanoid; the
    # scaled-down humanoids keep their bones out of the count like the full-size one)
    height = max([kin["xpos"][b][2] for b in range(1, nb)], default=1.0) - lowest
    max_contacts = max([sum(cap for c2, b, cap in standing if c2 == c and b <= lowest + 0.02 * height) for c in range(NCHAIN)], default=0)
    if muscles:
        max_contacts = min(max_contacts, 4)       # the muscle family is compiled with four slots per chain (box feet)
    elif max_links > 3 and max_contacts <= 4:
        # five-link humanoids without muscles run in the eight-slot families even when they STAND on four contacts per leg (Talos'
        # box feet): a control step that needs a fifth slot is abandoned and replayed (csrc/lm_step.h), which costs the launch a
        # good part of a control step's latency — measured on Talos.walk at 4096 environments under the random policy: 1.46 ms per
        # step with four slots (1.2 abandoned steps per launch), 1.34 ms with eight (0.2); profiles/r4_notes.md
        max_contacts = 8

    def src_code(obs_idx):
        kind, i = obs_src[int(obs_idx) % task["nobs"]]
        if kind == "g":
            return SRC_GOAL + i
        if kind == "f":
            raise UnsupportedModel("device reward reads a foot-force entry (evaluate it on the host)")
        k = i - m.body_jntadr[root]
        if not (0 <= k < m.body_jntnum[root]):
            raise UnsupportedModel("device reward reads a non-root dof")
        return (SRC_ROOT_QVEL if kind == "v" else SRC_ROOT_QPOS) + k

    h = np.zeros(HEADER_SIZE)
    h[H_MAGIC], h[H_VERSION] = LMC_MAGIC, 1
    h[H_NV], h[H_NU], h[H_NCHAINS], h[H_MAXLINKS] = m.nv, len(task["act_ctrl_idx"]), len(chains), max_links
    h[H_TIMESTEP] = m.timestep
    h[H_GX:H_GX + 3] = m.gravity
    h[H_IMPRATIO], h[H_ITERATIONS], h[H_TOLERANCE] = m.impratio, m.iterations, m.tolerance
    h[H_NSUBSTEPS], h[H_NOBS], h[H_NGOAL] = task["n_substeps"], task["nobs"], task["n_goal"]
    rt, rp = task["reward_type"], list(task["reward_params"])
    h[H_REWARD_TYPE] = rt
    if rt == 1:
        h[H_REWARD_P0:H_REWARD_P0 + 2] = [src_code(rp[0]), rp[1]]
    elif rt == 2:
        h[H_REWARD_P0:H_REWARD_P0 + 5] = [src_code(p) for p in rp[:5]]
    h[H_MEANINERTIA], h[H_CM_SIZE] = m.meaninertia, CM_SIZE
    h[H_INTEGRATOR], h[H_CONE], h[H_MAXCONTACTS] = m.integrator, m.cone, max_contacts
    h[H_NMUSCLE] = len(muscles)
    h[H_ACTMODE] = act_mode
    # ---- the tail of the constant table, right behind the last link slot in use
    off = CM_CHAINS + (C_LINKS + max_links * LINK_SIZE) * NCHAIN
    h[H_OFF_RUNSUP] = off
    root_unsup = _merge_proximity_spheres(root_unsup, MAXRG)
    rb[R_NUNSUP] = len(root_unsup)
    for i, u in enumerate(root_unsup):
        cm[off + i * U_SIZE:off + (i + 1) * U_SIZE] = u
    off += len(root_unsup) * U_SIZE
    def write_lists(lists, size, c_field):
        """every chain's list contiguous ([entry][field]); its start goes into the chain block"""
        nonlocal off
        for c in range(NCHAIN):
            cm[CM_CHAINS + c_field * NCHAIN + c] = off
            for i, u in enumerate(lists[c]):
                cm[off + i * size:off + (i + 1) * size] = u
            off += len(lists[c]) * size
    h[H_OFF_CUNSUP] = off
    write_lists(chain_unsup, U_SIZE, C_OFF_CUNSUP)
    h[H_OFF_PRUNE] = off
    write_lists(chain_prune, P_SIZE, C_OFF_PRUNE)
    h[H_OFF_LGROUP] = off
    write_lists(chain_groups_tab, LG_SIZE, C_OFF_LGROUP)
    # self-collisions: the quadruped family (<= 3 links per chain, Euler, elliptic cones) and the five-link humanoid families
    # (pyramids) have kernels with the pair path; a model without a candidate pair (Atlas, Talos: contype 0) keeps the plain ones
    pairs_on = (task.get("self_collisions", True) and _count_self_pairs(m) > 0
                and ((m.cone == mjcf.CONE_ELLIPTIC and max_links <= 3 and m.integrator == mjcf.INT_EULER and not muscles)
                     or (m.cone == mjcf.CONE_PYRAMIDAL and 3 < max_links <= 5 and not shared_first)
                     # six-link chains (UnitreeG1, UnitreeH1 with its arms): the regular kernels only DETECT (a geom pair within reach
                     # hands the control step to the family's replay kernel, which has the pair pass: csrc/lm_family.hip)
                     or (m.cone == mjcf.CONE_PYRAMIDAL and max_links == 6 and m.integrator == mjcf.INT_EULER and not muscles)))
    pair_tab = _self_collision_tables(m, root, chains, kin, register_hull, hull_block, prov["pairs"]) if pairs_on else None
    h[H_OFF_LPAIR] = off
    off_before_lp = off
    gpt, bpt = np.zeros(0), np.zeros(0)
    if pair_tab is not None:
        lanes_lp, gpt, spheres, kinds, bpt = pair_tab
        for c in range(NCHAIN):
            cm[CM_CHAINS + C_NLPAIR * NCHAIN + c] = len(lanes_lp[c])
        write_lists(lanes_lp, LP_SIZE, C_OFF_LPAIR)
        for (cl, li), ctr in spheres.items():
            if cl < 0:
                rb[R_BSX:R_BSX + 3] = ctr[0]
            else:
                cm[CM_CHAINS + (C_LINKS + li * LINK_SIZE + D_SIZE + L_BSX + np.arange(4)) * NCHAIN + cl] = list(ctr[0]) + [ctr[1]]
        info["self_collision_tables"] = dict(link_pairs=[len(x) for x in lanes_lp], body_pairs=len(bpt) // BP_SIZE, geom_pairs=len(gpt) // GPAIR_SIZE,
                                             closed_form=kinds[0], native=kinds[1], convex=kinds[2])
        if m.cone == mjcf.CONE_PYRAMIDAL:
            max_contacts = 8                  # the pair families are compiled with eight slots per chain (floor + self-contacts)
            h[H_MAXCONTACTS] = max_contacts
    assert off <= CM_SIZE
    # what a workgroup copies into its LDS: everything — but the six-link kernels read their link-pair lists (3 KB for UnitreeG1) and, since
    # the end of round 6, their prune records and link groups (1.8 KB) from the table's copy in global memory: with them in LDS the family
    # sat at 42.7 KB per workgroup = THREE per CU, and a batch of 4096 ran its last quarter of workgroups behind the first finishers
    h[H_CM_USED] = h[H_OFF_PRUNE] if max_links > 5 else off
    h[H_GT_SIZE] = GT_SIZE
    h[H_NGRF] = n_grf
    if max_groups_seen > 2 and max_links < MAXC:
        raise UnsupportedModel("more than two foot-force groups on one chain (four are compiled in the six-link kernels only)")
    if n_grf and sum(1 for c in range(len(chains)) for k in C_GRF_OBS if cm[CM_CHAINS + k * NCHAIN + c] >= 0) != len(grf_groups):
        raise UnsupportedModel("a foot-force group has no geom with a device collider")
    info.update(dropped_root_limits=dropped_root_limits, n_chains=len(chains), max_links=max_links, max_contacts=max_contacts, dof_to_lane=dof_to_lane,
                self_collision_pairs=_count_self_pairs(m))
    h[H_NGPAIR] = len(gpt) // GPAIR_SIZE
    h[H_OFF_GPT] = HEADER_SIZE + CM_SIZE + GT_SIZE + (MT_SIZE if mt is not None else 0)
    meshv = np.zeros((len(mesh_verts), 4))
    if mesh_verts:
        meshv[:, :3] = mesh_verts
        meshv[:, 3] = mesh_nbr_first
    assert len(mesh_nbr) < 2 ** 24                         # the indices travel as float32
    h[H_NMESHV], h[H_OFF_MESHV] = len(mesh_verts), h[H_OFF_GPT] + len(gpt)
    h[H_NMESHN], h[H_OFF_MESHN] = len(mesh_nbr), h[H_OFF_MESHV] + 4 * len(mesh_verts)
    h[H_NBPAIR], h[H_OFF_BPT] = len(bpt) // BP_SIZE, h[H_OFF_MESHN] + len(mesh_nbr)
    if pair_tab is None:
        mesh_adj = []                                      # only the convex-pair collider reads the adjacency blocks
    assert len(mesh_adj) < 2 ** 24
    h[H_NMESHADJ], h[H_OFF_MESHADJ] = len(mesh_adj), h[H_OFF_BPT] + len(bpt)
    info["mesh_vertices"] = len(mesh_verts)
    return np.concatenate([h, cm, gt] + ([mt] if mt is not None else []) + [gpt, meshv.ravel(), np.array(mesh_nbr, dtype=np.float64), bpt,
                                                                           np.array(mesh_adj, dtype=np.float64).ravel()]), info


# ---- model variants (inertial / armature / geom-friction randomisation): what differs between two lowerings of the same robot
# with other inertial numbers. Constant-table entries: per link mass, centre of mass, inertia tensor, armature, dof_invweight0 and
# the friction-loss regulariser (IR_LINK floats), the same for the root body and its 6 dofs, and the solver's termination scale
# 1 / (meaninertia * nv). Everything else that changes lives in the global tables (geom table, geom-pair table), which a variant
# brings whole. Layout of a record: [IR_SIZE][NCHAIN] (field-major, the chain's lane reads its own column).
IR_LINK = 13
IR_ROOT = MAXC * IR_LINK
IR_ROOT_DOF = IR_ROOT + 10
IR_SCALE = IR_ROOT_DOF + NROOT * 3
IR_SIZE = IR_SCALE + 1
_IR_LINK_FIELDS = [D_SIZE + L_MASS + i for i in range(10)] + [D_ARM, D_INVW, D_FLOSS_R]
_IR_DOF_FIELDS = [D_ARM, D_INVW, D_FLOSS_R]


def variant_tables(nominal, variant):
    """
    ``(record [IR_SIZE * NCHAIN], geom table, geom-pair table)`` of ``variant`` (a chain model from :func:`lower` of
    ``mjcf.model_variant(...)``) relative to ``nominal`` — the per-variant data of ``lm_set_model_variants``. Raises if the
    two lowerings differ anywhere else (topology, geometry, solver options): those cannot vary inside one batch.
    """
    if len(nominal) != len(variant):
        raise UnsupportedModel("variant and nominal model have different tables")
    cmn, cmv = nominal[HEADER_SIZE:HEADER_SIZE + CM_SIZE], variant[HEADER_SIZE:HEADER_SIZE + CM_SIZE]
    rec = np.zeros((IR_SIZE, NCHAIN))
    covered = []
    for c in range(NCHAIN):
        for k in range(MAXC):
            for j, f in enumerate(_IR_LINK_FIELDS):
                i = CM_CHAINS + (C_LINKS + k * LINK_SIZE + f) * NCHAIN + c
                rec[k * IR_LINK + j, c] = cmv[i]
                covered.append(i)
        for j in range(10):
            rec[IR_ROOT + j, c] = cmv[CM_ROOT + R_MASS + j]
        for d in range(NROOT):
            for j, f in enumerate(_IR_DOF_FIELDS):
                rec[IR_ROOT_DOF + 3 * d + j, c] = cmv[CM_ROOT + R_DOFS + d * D_SIZE + f]
        rec[IR_SCALE, c] = 1.0 / (variant[H_MEANINERTIA] * variant[H_NV])
    covered += [CM_ROOT + R_MASS + j for j in range(10)]
    covered += [CM_ROOT + R_DOFS + d * D_SIZE + f for d in range(NROOT) for f in _IR_DOF_FIELDS]
    other = np.setdiff1d(np.nonzero(cmn != cmv)[0], covered)
    hdiff = [i for i in np.nonzero(nominal[:HEADER_SIZE] != variant[:HEADER_SIZE])[0] if i != H_MEANINERTIA]
    if len(other) or hdiff:
        raise UnsupportedModel("model variant differs from the nominal model outside the inertial record "
                               "(constant-table entries %s, header entries %s)" % (other[:8], hdiff))
    g0 = HEADER_SIZE + CM_SIZE
    o_gpt, n_gpt, o_mv = int(nominal[H_OFF_GPT]), int(nominal[H_NGPAIR]) * GPAIR_SIZE, int(nominal[H_OFF_MESHV])
    if not (np.array_equal(nominal[g0 + GT_SIZE:o_gpt], variant[g0 + GT_SIZE:o_gpt]) and np.array_equal(nominal[o_mv:], variant[o_mv:])):
        raise UnsupportedModel("model variant changes the muscle table or the mesh vertices")
    return rec.ravel(), variant[g0:g0 + GT_SIZE].copy(), variant[o_gpt:o_gpt + n_gpt].copy()


def _union_capsule(caps):
    """Bounding capsule [centre 3, axis 3, half length, radius] of capsules (centre, axis, half, radius, ...): axis = principal direction
    of their end points; every end point lies within (R - r) of the segment, and the distance to a segment is convex along a
    capsule's own segment, so the union is covered."""
    ends, rads = [], []
    for c_, a_, h_, r_, *_ in caps:
        ends += [np.asarray(c_) - h_ * np.asarray(a_), np.asarray(c_) + h_ * np.asarray(a_)]
        rads += [r_, r_]
    ends = np.array(ends)
    ctr = ends.mean(0)
    if len(caps) == 1:
        c_, a_, h_, r_ = caps[0][:4]
        return np.concatenate([c_, a_, [h_, r_]])
    w, v = np.linalg.eigh(np.cov((ends - ctr).T) + 1e-18 * np.eye(3))
    axis = v[:, 2]
    t = (ends - ctr) @ axis
    lo, hi = float(t.min()), float(t.max())
    c0, half = ctr + 0.5 * (lo + hi) * axis, 0.5 * (hi - lo)
    tt = np.clip((ends - c0) @ axis, -half, half)
    d = np.linalg.norm((ends - c0) - np.outer(tt, axis), axis=1)
    return np.concatenate([c0, axis, [half, float((d + np.array(rads)).max()) * (1 + 1e-9)]])


def _engine_uses_ccd(t1, t2):
    """Pair types (t1 <= t2 in the engine's order sphere < capsule < cylinder < box < mesh) the engine's collision table routes to its
    general convex collider (mjc_Convex: libccd MPR); the others have native colliders."""
    if t2 == mjcf.GEOM_MESH:
        return True
    if t1 == mjcf.GEOM_CAPSULE and t2 == mjcf.GEOM_CYLINDER:
        return True
    return t1 == mjcf.GEOM_CYLINDER and t2 in (mjcf.GEOM_CYLINDER, mjcf.GEOM_BOX)


def _self_collision_tables(m, root, chains, kin, register_hull, hull_block, pair_prov=None):
    """
    Tables of the self-collision path (kernels compiled with PAIRS): candidate geom pairs after the engine's filters (different
    weld groups, not parent and child, contype / conaffinity — the floor is handled elsewhere), grouped by link pair.
    Kind 0: sphere / capsule pairs (closed form). Kind 2: the pairs the engine collides through its general convex collider
    (anything against a mesh, capsule / cylinder / box against a cylinder): MPR on the device, hull vertices in the mesh-vertex
    table. Kind 1: pairs with one of the engine's NATIVE box / cylinder colliders (sphere / capsule / box against a box, sphere
    against a cylinder): restated on the device (csrc/lm_core.h nat_*) like in the oracle; a mesh without a hull in such a pair
    stays counted only (``self_proximity`` statistic).
    Two links of ONE chain may form a pair (UnitreeH1: hip-yaw cylinder against the thigh of the same leg): the entry then lives in
    that lane only. Returns (per-lane link-pair entries [code, range, reach^2], geom-pair records (flat),
    {(lane, link): bounding sphere}).
    """
    pyramidal = m.cone != mjcf.CONE_ELLIPTIC
    where = {root: (-1, 7)}                                   # weld group -> (lane, link index after its last joint)
    for c, chain in enumerate(chains):
        li = -1
        for b in chain:
            li += m.body_jntnum[b]
            if b not in where:              # (a first link shared by two chains belongs to the first of them: its owner)
                where[b] = (c, li)

    def rel_pose(b):
        w = m.body_weldid[b]
        rw = kin["xmat"][w]
        return rw.T @ (kin["xpos"][b] - kin["xpos"][w]), rw.T @ kin["xmat"][b]

    def capsule_of(g):
        """(centre, axis, half, radius, rbound) of geom g in the frame of its weld body; exact for spheres and capsules."""
        p, r = rel_pose(m.geom_body[g])
        pos, rot = p + r @ m.geom_pos[g], r @ mjcf.quat_to_mat(m.geom_quat[g])
        t, size = m.geom_type[g], m.geom_size[g]
        if t == mjcf.GEOM_SPHERE:
            return pos, rot[:, 2], 0.0, size[0], size[0]
        if t == mjcf.GEOM_BOX:
            k = int(np.argmax(size))
            rad = float(np.sqrt(sum(size[j] ** 2 for j in range(3) if j != k)))
            return pos, rot[:, k], size[k], rad, size[k] + rad
        return pos, rot[:, 2], size[1], size[0], size[0] + size[1]        # capsule, cylinder, mesh (bounding capsule)

    def convex_of(g):
        """GX_SIZE floats of geom g for the convex-pair collider (link frame)."""
        p, r = rel_pose(m.geom_body[g])
        rot = r @ mjcf.quat_to_mat(m.geom_quat[g])
        t, size = int(m.geom_type[g]), m.geom_size[g]
        x = np.zeros(GX_SIZE)
        x[GX_TYPE] = t
        x[GX_CX:GX_CX + 3] = p + r @ getattr(m, "geom_center", m.geom_pos)[g]
        if t == mjcf.GEOM_MESH:
            first, hv = register_hull(g)
            x[GX_E0], x[GX_E0 + 1] = first, len(hv)
            # where the hill climbing of the support search starts: the hull's extreme vertices along +x, -x, +y, -y, +z, -z (link frame)
            # (as positions of their adjacency blocks in the table the climbing reads)
            x[GX_E0 + 2:GX_E0 + 8] = [hull_block[g][int(np.argmax(sg * hv[:, k]))] for k in range(3) for sg in (1.0, -1.0)]
        elif t == mjcf.GEOM_BOX:
            x[GX_E0:GX_E0 + 3] = size
            x[GX_E0 + 3:GX_E0 + 6], x[GX_E0 + 6:GX_E0 + 9] = rot[:, 0], rot[:, 1]
        # radius of the engine's bounding sphere around the geom position (margin-less mid phase of a geom pair)
        x[GX_RBOUND] = {mjcf.GEOM_SPHERE: size[0], mjcf.GEOM_CAPSULE: size[0] + size[1], mjcf.GEOM_MESH: size[0] + size[1],
                          mjcf.GEOM_CYLINDER: float(np.hypot(size[0], size[1])), mjcf.GEOM_BOX: float(np.linalg.norm(size))}[t]
        return x

    by_links = {}
    for g1 in range(m.ngeom):
        for g2 in range(g1 + 1, m.ngeom):
            b1, b2 = m.geom_body[g1], m.geom_body[g2]
            w1, w2 = m.body_weldid[b1], m.body_weldid[b2]
            if w1 == 0 or w2 == 0 or w1 == w2:
                continue
            p1, p2 = m.body_weldid[m.body_parent[w1]], m.body_weldid[m.body_parent[w2]]
            if w1 == p2 or w2 == p1:
                continue
            if not ((m.geom_contype[g1] & m.geom_conaffinity[g2]) or (m.geom_contype[g2] & m.geom_conaffinity[g1])):
                continue
            a, b = (g1, g2) if m.geom_type[g1] <= m.geom_type[g2] else (g2, g1)      # the engine's order: lower type first
            by_links.setdefault((min(w1, w2), max(w1, w2)), []).append((a, b))
    handled = (mjcf.GEOM_SPHERE, mjcf.GEOM_CAPSULE)
    has_hull = getattr(m, "geom_hull_num", np.zeros(m.ngeom, int))
    lanes_lp = [[] for _ in range(NCHAIN)]
    records, spheres_r, bodypairs = [], {}, []
    # bounding sphere per weld group over the geoms that take part in any pair
    members = {}
    for (wp, wq), pairs in by_links.items():
        for a, b in pairs:
            members.setdefault(m.body_weldid[m.geom_body[a]], set()).add(a)
            members.setdefault(m.body_weldid[m.geom_body[b]], set()).add(b)
    sphere = {}
    for w, gs in members.items():
        caps = [capsule_of(g) for g in gs]
        ctr = np.mean([cp[0] for cp in caps], axis=0)
        sphere[w] = (ctr, max(np.linalg.norm(cp[0] - ctr) + cp[4] for cp in caps))
    kinds = [0, 0, 0]
    for (wp, wq), pairs in sorted(by_links.items()):
        if wp not in where or wq not in where:
            raise UnsupportedModel("self-collision pair outside the root+chains structure")
        (lp, kp), (lq, kq) = where[wp], where[wq]
        # a first link SHARED by two chains (a torso with an arm on either side) against a link of the chain that carries its
        # massless COPY: both are links of that chain — the relative motion does not depend on the shared dof, and as a pair of ONE
        # lane its rows hold that chain's own joints only (between the owner's lane and the copy's it would couple the two copies)
        if lp >= 0 and lq >= 0 and lp != lq:
            if kp == 0 and chains[lq][0] == wp:
                lp = lq
            elif kq == 0 and chains[lp][0] == wq:
                lq = lp
        if lp < 0:                                                   # the root body is always the PARTNER of an entry
            (wp, wq), (lp, kp), (lq, kq) = (wq, wp), (lq, kq), (lp, kp)
        margin_max = 0.0
        # geom pairs grouped by the two BODIES they join (the mid phase tests one bounding capsule per body), <= 24 per group
        def side(g):
            return m.body_weldid[m.geom_body[g]] == wq
        groups = {}
        for a, b in pairs:
            gp_, gq_ = (b, a) if side(a) else (a, b)
            groups.setdefault((int(m.geom_body[gp_]), int(m.geom_body[gq_])), []).append((a, b))
        chunks = []
        for key in sorted(groups):
            for i0 in range(0, len(groups[key]), 24):
                chunks.append(groups[key][i0:i0 + 24])
        first_bp = len(bodypairs)
        for chunk in chunks:
            bp = np.zeros(BP_SIZE)
            for base, want_q in ((BP_P1, False), (BP_P2, True)):
                gs = sorted(set(g for ab in chunk for g in ab if side(g) == want_q))
                bp[base:base + 8] = _union_capsule([capsule_of(g) for g in gs])
            bp[BP_FIRST], bp[BP_N] = len(records), len(chunk)
            bp_margin = 0.0
            for a, b in chunk:
                dim, solref, solimp, fr, margin, gap = _mix_with_floor(m, a, b)
                assert gap == 0
                ta, tb = int(m.geom_type[a]), int(m.geom_type[b])
                if ta in handled and tb in handled:
                    kind = 0
                elif _engine_uses_ccd(ta, tb) and all(m.geom_type[g] != mjcf.GEOM_MESH or has_hull[g] > 0 for g in (a, b)):
                    kind = 2
                elif (ta, tb) in ((mjcf.GEOM_SPHERE, mjcf.GEOM_BOX), (mjcf.GEOM_SPHERE, mjcf.GEOM_CYLINDER), (mjcf.GEOM_CAPSULE, mjcf.GEOM_BOX)) \
                        or ((ta, tb) == (mjcf.GEOM_BOX, mjcf.GEOM_BOX) and pyramidal):      # (the quadruped family is compiled without the box-box collider)
                    kind = 1                                           # the engine's native box / cylinder colliders
                else:
                    kind = 3                                           # a mesh without a convex hull: counted only
                kinds[min(kind, 2) if kind != 3 else 1] += 1
                rec = np.zeros(GPAIR_SIZE)
                rec[GP_KIND], rec[GP_G1Q] = kind, float(m.body_weldid[m.geom_body[a]] == wq)
                for base, g in ((GP_P1, a), (GP_P2, b)):
                    pos, axis, half, rad, _ = capsule_of(g)
                    rec[base:base + 3], rec[base + 3:base + 6], rec[base + 6], rec[base + 7] = pos, axis, half, rad
                if kind != 0:      # convex pairs (MPR) and native pairs (the engine's box / cylinder colliders): type, centre, box axes of both geoms
                    rec[GP_X1:GP_X1 + GX_SIZE], rec[GP_X2:GP_X2 + GX_SIZE] = convex_of(a), convex_of(b)
                rec[GP_MARGIN] = margin
                rec[GP_K], rec[GP_B] = _kb(solref, solimp, m.timestep)
                rec[GP_S0:GP_S0 + 5] = _clip_solimp(solimp)
                tran = m.body_invweight0[m.geom_body[a], 0] + m.body_invweight0[m.geom_body[b], 0]
                fr = np.maximum(np.asarray(fr, dtype=np.float64), MINMU)
                rec[GP_F0:GP_F0 + 5] = fr
                if pyramidal:
                    # condim 3: the four edges of the pyramid share R = 2 mu^2 (1 + mu^2) tran (like the floor contacts, G_TRAN).
                    # condim 1 (frictionless, the humanoid's bones): ONE row with R = tran, written as a pyramid with mu = 0 — its four
                    # edges coincide, each with a quarter of the row's D — so that the kernels compiled for pyramids need no other row type
                    if dim not in (1, 3):
                        raise UnsupportedModel("pyramidal condim %d is not built on the device" % dim)
                    if dim == 3 and fr[0] != fr[1]:
                        raise UnsupportedModel("anisotropic sliding friction")
                    mu = fr[0] if dim == 3 else 0.0
                    rec[GP_DIM], rec[GP_MU] = 3, mu
                    rec[GP_TRAN] = 2 * mu * mu * (1 + mu * mu) * tran if dim == 3 else 4.0 * tran
                    rec[GP_RR1:GP_RR1 + 5] = 1.0
                else:
                    if dim not in (1, 3, 4, 6):
                        raise UnsupportedModel("condim %d" % dim)
                    rec[GP_TRAN] = tran
                    rec[GP_DIM] = dim
                    rec[GP_MU] = fr[0] / np.sqrt(max(MINVAL, m.impratio))
                    rr1 = 1.0 / max(MINVAL, m.impratio)
                    rec[GP_RR1], rec[GP_RR2] = rr1, rr1 * fr[0] * fr[0] / (fr[1] * fr[1])
                    rec[GP_RR3:GP_RR3 + 3] = [rr1 * fr[0] * fr[0] / (fr[k] * fr[k]) for k in (2, 3, 4)]
                records.append(rec)
                if pair_prov is not None:
                    pair_prov.append((int(a), int(b)))
                margin_max = max(margin_max, margin)
                bp_margin = max(bp_margin, margin)
            bp[BP_MARGIN] = bp_margin
            bodypairs.append(bp)
        reach2 = (sphere[wp][1] + sphere[wq][1] + margin_max + PAIR_PAD) ** 2
        # the device exchanges the hits of an entry as a 24-bit mask: a link pair with more body pairs is several entries
        n_all = len(bodypairs) - first_bp
        for first in range(first_bp, first_bp + n_all, 24):
            n = min(24, first_bp + n_all - first)
            assert first < 65536
            # ONE entry per link pair, in the lane of its first link: that lane tests the pair, the partner lane takes the contacts over as
            # mirror slots (round 2 listed a cross-chain pair in both lanes, code + 256 for the second link's view: both ran the tests)
            # (either link's lane can hold it — the device decodes both views, + 256 = "my link is the pair's second": the shorter list gets it)
            if lq >= 0 and lq != lp and len(lanes_lp[lq]) < len(lanes_lp[lp]):
                lanes_lp[lq].append([kq + 8 * kp + 64 * lp + 256, first + 65536 * n, reach2])
            else:
                lanes_lp[lp].append([kp + 8 * kq + 64 * max(lq, 0), first + 65536 * n, reach2])
        spheres_r[where[wp]], spheres_r[where[wq]] = sphere[wp], sphere[wq]
        spheres_r[(lp, kp)], spheres_r[(lq, kq)] = sphere[wp], sphere[wq]          # (a shared first link seen as the copy lane's link 0)
    six = max(len(ch) for ch in chains) > 5            # (six-link chains: two mask words on the device, lists read from global memory)
    if max(len(x) for x in lanes_lp) > (MAXLP if six else 64):
        raise UnsupportedModel("too many self-collision link pairs (%s)" % [len(x) for x in lanes_lp])
    # what the device packs into one float32 (csrc/lm_core.h: the pair pass): a work item = entry * 65536 + geom-pair record; a body
    # pair in reach = entry * 32 + body pair of the entry + 4096 * its geom pairs; an entry in reach = entry + 64 * its body pairs
    if len(records) >= 65536 or len(bodypairs) >= 65536 or max(len(x) for x in lanes_lp) > (MAXLP if six else 64):
        raise UnsupportedModel("self-collision tables beyond the device's packing (%d geom pairs, %d body pairs, %s link pairs per lane)"
                               % (len(records), len(bodypairs), [len(x) for x in lanes_lp]))
    assert all(int(bp[BP_N]) <= 24 for bp in bodypairs) and all((int(e[1]) >> 16) <= 24 for x in lanes_lp for e in x)
    return lanes_lp, (np.concatenate(records) if records else np.zeros(0)), spheres_r, kinds, (np.concatenate(bodypairs) if bodypairs else np.zeros(0))


def _count_self_pairs(m):
    """Number of non-floor geom pairs MuJoCo's filters would let collide (ignored by the device path)."""
    n = 0
    for g1 in range(m.ngeom):
        for g2 in range(g1 + 1, m.ngeom):
            b1, b2 = m.geom_body[g1], m.geom_body[g2]
            w1, w2 = m.body_weldid[b1], m.body_weldid[b2]
            if w1 == 0 or w2 == 0 or w1 == w2:
                continue
            p1, p2 = m.body_weldid[m.body_parent[w1]], m.body_weldid[m.body_parent[w2]]
            if w1 == p2 or w2 == p1:
                continue
            if (m.geom_contype[g1] & m.geom_conaffinity[g2]) or (m.geom_contype[g2] & m.geom_conaffinity[g1]):
                n += 1
    return n


# ---------------------------------------------------------------------------------------------------------------
# The model compiler on the device (csrc/lm_compile.hip): a fresh model per device-side restart
# ---------------------------------------------------------------------------------------------------------------
# The reference re-compiles its model at every reset (base.py:183-185, utils/domain_randomization.py:219-227). What a re-compile
# changes on this path is the data of `variant_tables`; the device writes it per environment from that environment's own draws:
# the draws (`JointRandomization.model_draw_ops`), the inertial numbers of the drawn bodies (mjcf.inertia_from_spec), the mass matrix
# at qpos0 (mjcf.mass_matrix: a sum over bodies of J^T I J, linear in the drawn masses / inertia tensors, so the bodies without a
# rule are summed here once), its inverse -> dof_invweight0 / body_invweight0 / meaninertia (mjcf._set_const), and from those the
# record fields and the contact constants exactly as `lower` writes them. This function prepares the constants of that program.
MC_MAGIC = 0x4C4D4D43  # "LMMC"
MC_IH_SIZE, MC_DH_SIZE = 16, 8
MC_RB_INTS, MC_RB_DBLS = 4, 55
MC_DRAW_INTS, MC_DRAW_DBLS = 4, 2
MC_CON_INTS = 8
MC_NSLOT = 1 + NCHAIN * MAXC


def model_compiler_tables(m, task, draw_ops, svd):
    """
    ``(int32 blob, float64 blob, info)`` for ``lm_set_model_compiler``: the constants of the device-side model compiler for model
    ``m`` lowered with ``task`` and the draws ``draw_ops`` / ``svd`` of ``JointRandomization.model_draw_ops``.

    int32:  header [magic, nv, n_rbody, n_gslot, n_draw, n_slot, n_recop, n_conop, n_body, pyramidal, balanceinertia, 0...]
            draws   [n_draw][kind, target, index, component]      (index: dof | drawn-body index | friction slot)
            rbodies [n_rbody][body, inertial kind, inertial slot, has singular values]
            recops  [n_recop][destination in the record (row * NCHAIN + lane), source value]
            conops  [n_conop][table (0 geom table, 1 pair table), index of the TRAN entry, stride between fields, dim + 16 * mix
                              (0 max, 1 first, 2 second) + 256 * pair, friction slot a, friction slot b, body a, body b]
    float64: header [impratio, boundmass, boundinertia, MINVAL, nv as float, 0...]
            draws [n_draw][a, b]; rbodies [n_rbody][mass, vals 6, R(quat) 9, U 9, Vt 9, com in the slot frame 3, R in the slot
            frame 9, R in the world at qpos0 9]; jac [n_body][6][nv] at the bodies' centres of mass (qpos0); mbase [nv][nv] (bodies
            without a rule, no armature); armature [nv]; friction-loss factor (1-d0)/d0 [nv] (< 0: no row); slot base [n_slot][mass,
            mass*com 3, inertia about the slot origin 6]; friction [n_gslot][3]
    Values the device gathers the record from: ARM[nv] INVW[nv] RFL[nv] SCALE LINK[n_slot][10].
    """
    from .utils.domain_randomization import JointRandomization as JR
    chain, info = lower(m, task)
    prov = info["variant_provenance"]
    nv, nb = m.nv, m.nbody
    kin = mjcf.forward_kinematics(m, m.qpos0)
    jac = mjcf.body_jacobians(m, kin, kin["xipos"])            # [nbody][6][nv]: rows 0-2 translation at the COM, 3-5 rotation
    rbodies = sorted(set(op[4] for op in draw_ops if op[3] in (JR.TARGET_MASS, JR.TARGET_DIAGINERTIA, JR.TARGET_SINGULAR)))
    rb_index = {b: j for j, b in enumerate(rbodies)}
    gslots = {}                                                 # model geom -> friction slot

    def gslot(g):
        return gslots.setdefault(int(g), len(gslots))

    # inertial slots: 0 = root body, 1 + lane * MAXC + link
    slot_of_weld = {}
    for c, li, w in prov["links"]:
        slot_of_weld[int(w)] = 0 if c < 0 else 1 + c * MAXC + li

    def rel_pose(b):
        w = m.body_weldid[b]
        rw = kin["xmat"][w]
        return rw.T @ (kin["xpos"][b] - kin["xpos"][w]), rw.T @ kin["xmat"][b]

    slot_base = np.zeros((MC_NSLOT, 10))
    mbase = np.zeros((nv, nv))
    for b in range(1, nb):
        w = int(m.body_weldid[b])
        if b in rb_index:
            continue
        jp, jr = jac[b, 0:3], jac[b, 3:6]
        iw = kin["xmat"][b] @ m.body_inertia[b] @ kin["xmat"][b].T
        mbase += m.body_mass[b] * jp.T @ jp + jr.T @ iw @ jr
        if w == 0:
            continue
        if w not in slot_of_weld:
            raise UnsupportedModel("body %s outside the inertial slots" % m.body_names[b])
        p, r = rel_pose(b)
        cb = p + r @ m.body_ipos[b]
        io = r @ m.body_inertia[b] @ r.T + m.body_mass[b] * (cb @ cb * np.eye(3) - np.outer(cb, cb))
        slot_base[slot_of_weld[w]] += np.concatenate([[m.body_mass[b]], m.body_mass[b] * cb,
                                                       [io[0, 0], io[1, 1], io[2, 2], io[0, 1], io[0, 2], io[1, 2]]])
    ints = [np.zeros(MC_IH_SIZE, dtype=np.int64)]
    dbls = [np.zeros(MC_DH_SIZE)]
    # draws
    di, dd = np.zeros((len(draw_ops), MC_DRAW_INTS), dtype=np.int64), np.zeros((len(draw_ops), MC_DRAW_DBLS))
    for i, (kind, a, b, target, index, comp) in enumerate(draw_ops):
        idx = index if target == JR.TARGET_ARMATURE else (gslot(index) if target == JR.TARGET_FRICTION else rb_index[index])
        di[i], dd[i] = (kind, target, idx, comp), (a, b)
    # drawn bodies
    ri, rd = np.zeros((len(rbodies), MC_RB_INTS), dtype=np.int64), np.zeros((len(rbodies), MC_RB_DBLS))
    for j, b in enumerate(rbodies):
        kind = int(m.body_inertial_kind[b])
        if kind == 0:
            raise ValueError("body %s has no <inertial> element" % m.body_names[b])
        w = int(m.body_weldid[b])
        if w not in slot_of_weld:
            raise UnsupportedModel("body %s outside the inertial slots" % m.body_names[b])
        ri[j] = (b, kind, slot_of_weld[w], 1 if b in svd else 0)
        p, r = rel_pose(b)
        u, vt = svd.get(b, (np.eye(3), np.eye(3)))
        rd[j] = np.concatenate([[m.body_xml_mass[b]], m.body_inertial_vals[b], mjcf.quat_to_mat(m.body_inertial_quat[b]).ravel(),
                                np.asarray(u).ravel(), np.asarray(vt).ravel(), p + r @ m.body_ipos[b], r.ravel(), kin["xmat"][b].ravel()])
    # record gather: which value lands where
    V_ARM, V_INVW, V_RFL, V_SCALE, V_LINK = 0, nv, 2 * nv, 3 * nv, 3 * nv + 1
    recops = []
    fd = -np.ones(nv)
    for c, li, d, copy in prov["dofs"]:
        fl = m.dof_frictionloss[d]
        if fl > 0 or m.dof_invweight0[d] > 0:
            si = _clip_solimp(m.dof_solimp[d])
            d0 = si[0] if not (si[0] == si[1] or si[2] <= MINVAL) else 0.5 * (si[0] + si[1])
            fd[d] = (1 - d0) / d0
        row = (IR_ROOT_DOF + 3 * li) if c < 0 else (li * IR_LINK + 10)
        for lane in (range(NCHAIN) if c < 0 else [c]):
            if not copy:
                recops.append((row * NCHAIN + lane, V_ARM + d))
            recops.append(((row + 1) * NCHAIN + lane, V_INVW + d))
            if fd[d] >= 0:
                recops.append(((row + 2) * NCHAIN + lane, V_RFL + d))
    for c, li, w in prov["links"]:
        s = slot_of_weld[int(w)]
        row = IR_ROOT if c < 0 else li * IR_LINK
        for lane in (range(NCHAIN) if c < 0 else [c]):
            recops += [((row + f) * NCHAIN + lane, V_LINK + s * 10 + f) for f in range(10)]
    recops += [(IR_SCALE * NCHAIN + lane, V_SCALE) for lane in range(NCHAIN)]
    # contact constants
    floor = [g for g in range(m.ngeom) if m.geom_type[g] == mjcf.GEOM_PLANE][0]

    def mix_mode(a, b):
        pa, pb = m.geom_priority[a], m.geom_priority[b]
        return 0 if pa == pb else (1 if pa > pb else 2)

    conops = []
    for i, c, g in prov["geoms"]:
        dim = _mix_with_floor(m, g, floor)[0]
        conops.append((0, (i * G_SIZE + G_TRAN) * NCHAIN + c, NCHAIN, dim + 16 * mix_mode(g, floor), gslot(g), gslot(floor),
                       int(m.geom_body[g]), 0))
    for k, (a, b) in enumerate(prov["pairs"]):
        dim = _mix_with_floor(m, a, b)[0]
        conops.append((1, k * GPAIR_SIZE + GP_TRAN, 1, dim + 16 * mix_mode(a, b) + 256, gslot(a), gslot(b),
                       int(m.geom_body[a]), int(m.geom_body[b])))
    assert G_MU - G_TRAN == GP_MU - GP_TRAN == 2 and G_F0 - G_TRAN == GP_F0 - GP_TRAN == 3 and G_RR1 - G_TRAN == GP_RR1 - GP_TRAN == 8
    fric = np.zeros((len(gslots), 3))
    for g, s in gslots.items():
        fric[s] = m.geom_friction[g]
    pyramidal = m.cone != mjcf.CONE_ELLIPTIC
    head_i, head_d = ints[0], dbls[0]
    head_i[:11] = (MC_MAGIC, nv, len(rbodies), len(gslots), len(draw_ops), MC_NSLOT, len(recops), len(conops), nb, int(pyramidal),
                   int(bool(m.compiler_bounds[2])))
    head_d[:5] = (m.impratio, m.compiler_bounds[0], m.compiler_bounds[1], MINVAL, nv)
    ints += [di.ravel(), ri.ravel(), np.asarray(recops, dtype=np.int64).reshape(-1, 2).ravel(),
             np.asarray(conops, dtype=np.int64).reshape(-1, MC_CON_INTS).ravel()]
    dbls += [dd.ravel(), rd.ravel(), np.asarray(jac, dtype=np.float64)[:, :6].ravel(), mbase.ravel(), np.asarray(m.dof_armature, dtype=np.float64),
             fd, slot_base.ravel(), fric.ravel()]
    out_info = dict(n_draw=len(draw_ops), n_rbody=len(rbodies), n_gslot=len(gslots), n_recop=len(recops), n_conop=len(conops),
                    rbodies=rbodies, gslots=dict(gslots))
    return np.concatenate(ints).astype(np.int32), np.concatenate(dbls).astype(np.float64), out_info
