/*
 * lm_model_blob.h — flat GENERAL model description (body tree, joints, geoms, actuators) consumed by the
 * fp64 test oracle (oracle/oracle.c). The product library takes the lowered "chain model" instead
 * (lm_layout.h, lm_model_create in locohip.h).
 *
 * The reference hands MuJoCo an XML string (mushroom-rl MultiMuJoCo.__init__, invoked from
 * /root/reference/loco_mujoco/environments/base.py:109-111 -> mujoco.MjModel.from_xml_string) and
 * keeps task facts (observation spec, thresholds, reward) in Python
 * (/root/reference/loco_mujoco/environments/quadrupeds/unitreeA1.py:778-854, :503-536).
 * Here the host-side mini-compiler (loco_mujoco_amd/mjcf.py) produces ONE array of float64 ("blob");
 * integers are stored as doubles. This header only fixes the layout; it contains no algorithm.
 */
#ifndef LM_MODEL_BLOB_H
#define LM_MODEL_BLOB_H

#define LM_BLOB_MAGIC 0x4C4D4231 /* "LMB1" */
#define LM_BLOB_VERSION 6

/* header slots (doubles) */
enum {
  LMH_MAGIC = 0, LMH_VERSION, LMH_NBODY, LMH_NV, LMH_NGEOM, LMH_NU, LMH_CONE, LMH_INTEGRATOR,
  LMH_ITERATIONS, LMH_TIMESTEP, LMH_IMPRATIO, LMH_TOLERANCE, LMH_GRAV_X, LMH_GRAV_Y, LMH_GRAV_Z,
  LMH_MEANINERTIA, LMH_NSITE, LMH_NTENDON, LMH_NWRAP, LMH_NA, LMH_NHULLVERT, LMH_NHULLNBR, LMH_HEADER_SIZE = 32
};

/* geom types / joint types / cones / integrators (private numbering of this framework) */
enum { LM_GEOM_PLANE = 0, LM_GEOM_SPHERE, LM_GEOM_CAPSULE, LM_GEOM_CYLINDER, LM_GEOM_BOX, LM_GEOM_MESH };
enum { LM_JNT_SLIDE = 0, LM_JNT_HINGE = 1 };
enum { LM_CONE_PYRAMIDAL = 0, LM_CONE_ELLIPTIC = 1 };
enum { LM_INT_EULER = 0, LM_INT_RK4 = 1 };
enum { LM_ACT_MOTOR = 0, LM_ACT_MUSCLE = 1, LM_ACT_POSITION = 2 };

/*
 * Array order after the header. "nb"=nbody, "nv"=number of dofs (= joints, all 1-dof), "ng"=ngeom.
 *
 *  body_parent[nb] body_pos[3nb] body_quat[4nb] body_mass[nb] body_ipos[3nb] body_inertia[9nb]
 *  body_jntadr[nb] body_jntnum[nb] body_weldid[nb] body_invweight0[2nb]
 *  jnt_type[nv] jnt_body[nv] jnt_pos[3nv] jnt_axis[3nv] jnt_limited[nv] jnt_range[2nv] jnt_stiffness[nv]
 *  jnt_margin[nv] jnt_solref[2nv] jnt_solimp[5nv]
 *  dof_damping[nv] dof_armature[nv] dof_frictionloss[nv] dof_solref[2nv] dof_solimp[5nv] dof_parent[nv]
 *  dof_invweight0[nv]
 *  geom_type[ng] geom_body[ng] geom_pos[3ng] geom_quat[4ng] geom_size[3ng] geom_contype[ng]
 *  geom_conaffinity[ng] geom_condim[ng] geom_priority[ng] geom_friction[3ng] geom_solmix[ng]
 *  geom_solref[2ng] geom_solimp[5ng] geom_margin[ng] geom_gap[ng]
 *  act_dof[nu] act_gear[nu] act_ctrlrange[2nu] act_ctrllimited[nu]
 *  -- spatial tendons through sites and muscle actuators (ns = sites on tendon paths, nt = tendons, nw = path entries)
 *  site_body[ns] site_pos[3ns] tendon_adr[nt] tendon_num[nt] wrap_site[nw]
 *  act_kind[nu] act_tendon[nu] act_dynprm[3nu] act_gainprm[9nu] act_lengthrange[2nu]
 *  (act_kind: LM_ACT_MOTOR joint torque gear*ctrl | LM_ACT_MUSCLE: activation state + force-length-velocity
 *   curves on a tendon; gainprm = range0 range1 force scale lmin lmax vmax fpmax fvmax, dynprm = tau_act tau_deact tausmooth
 *   | LM_ACT_POSITION: affine joint servo, force = gainprm[0]*ctrl + biasprm[0] + biasprm[1]*length + biasprm[2]*velocity)
 *  -- version 3
 *  act_biasprm[3nu] act_forcerange[2nu] act_forcelimited[nu]   (force clamped to forcerange when forcelimited)
 *  -- version 4: convex hulls of the mesh geoms (nh = LMH_NHULLVERT vertices in all, each in the frame of its geom's BODY)
 *  geom_hull_adr[ng] (-1: none) geom_hull_num[ng] hull_vert[3nh]   (plane vs mesh: a contact at the hull's support vertex ...)
 *  -- version 5: the hulls' vertex graph (nn = LMH_NHULLNBR entries): neighbours of hull vertex i of geom g, as indices into the
 *  geom's hull, nearest first, are hull_nbr[hull_nbr_adr[geom_hull_adr[g] + i] : hull_nbr_adr[geom_hull_adr[g] + i + 1]]
 *  hull_nbr_adr[nh + 1] hull_nbr[nn]   (... and further contacts at penetrating neighbours of the support vertex)
 *  -- version 6: geom_center[3ng], body frame: the centre the convex-convex collider starts from (geom frame origin; mesh: its centre of mass)
 */

#endif
