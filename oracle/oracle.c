/*
 * oracle.c — fp64 CPU restatement of the physics step behind LocoEnv.step().
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product (loco_mujoco_amd/) may import, link or call this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the checker.
 *
 * What it restates: the reference's hot path is mushroom-rl's MuJoCo.step -> mujoco.mj_step(model,
 * data, 10) (third-party, mujoco==2.3.7 pinned in /root/reference/pyproject.toml:10; call chain in
 * SURVEY.md §3.3; the reference's own call sites: loco_mujoco/environments/base.py:109-111 (model
 * build, timestep 0.001), base.py:180 (mj_resetData), gymnasium.py:63 (step)). MuJoCo's source is
 * not under /root/reference, so this file restates MuJoCo 2.3.7's *published algorithm* for the
 * feature subset the BASELINE models use: hinge/slide trees, explicit
 * inertials, primitive-vs-plane and sphere/capsule pair collisions, friction-loss / joint-limit /
 * contact constraints with the solref/solimp impedance model, pyramidal and elliptic cones, a
 * Newton solver on the convex primal problem, semi-implicit Euler with implicit joint damping, RK4.
 * Parity is pinned by the reference's golden rollouts (tests/test_datasets/<task>.npy, generator
 * tests/test_environments.py:15-38,67-94) replayed one control step at a time (tests/test_oracle_golden.py).
 * Also restated: spatial tendons through sites with Hill-type muscles and their activation states, position servos
 * (affine actuators with a force range), plane vs convex mesh (one contact at the hull's support vertex), mass and
 * inertia of bodies defined by their geoms (in loco_mujoco_amd/mjcf.py).
 * Pinned / unpinned: every golden file of UnitreeA1 (.simple, .hard rows), Atlas (.walk, .carry), Talos (.walk, .carry),
 * HumanoidTorque / HumanoidMuscle (+ the 4Ages variants) is reproduced row by row, the rows with convex-convex (bone mesh)
 * contacts through the restated general convex collider (MPR, below), the box-on-box rows of the 4Ages "all" file through the
 * restated native box collider; plane-mesh is pinned on the UnitreeH1 rows without hull-hull contact; 26 UnitreeH1 rows with a
 * cylinder-cap-on-hull MPR contact are ill-conditioned in float64 and reproduced by nothing (tests/test_oracle_golden.py).
 * UNPINNED (no golden row exercises them): sphere/capsule self-collisions, sphere-box / sphere-cylinder / capsule-box, plane-cylinder, position servos, the muscle
 * curve beyond lmax, the reward values, per-environment joint parameters, foot-force observations.
 *
 * Deliberately simple and dense (O(nbody*nv^2) mass matrix from body Jacobians, dense Cholesky):
 * it shares no code and no algorithmic shortcuts with the HIP path it checks.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "../include/lm_model_blob.h"
#include "oracle.h"

#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999

/* ------------------------------------------------------------------------------------------ */
/* small vector helpers                                                                        */
/* ------------------------------------------------------------------------------------------ */
static inline double dot3(const double* a, const double* b) { return a[0]*b[0]+a[1]*b[1]+a[2]*b[2]; }
static inline void cross3(double* r, const double* a, const double* b) {
  double x = a[1]*b[2]-a[2]*b[1], y = a[2]*b[0]-a[0]*b[2], z = a[0]*b[1]-a[1]*b[0];
  r[0]=x; r[1]=y; r[2]=z;
}
static inline void copy3(double* r, const double* a) { r[0]=a[0]; r[1]=a[1]; r[2]=a[2]; }
static inline void add3(double* r, const double* a, const double* b) { r[0]=a[0]+b[0]; r[1]=a[1]+b[1]; r[2]=a[2]+b[2]; }
static inline void sub3(double* r, const double* a, const double* b) { r[0]=a[0]-b[0]; r[1]=a[1]-b[1]; r[2]=a[2]-b[2]; }
static inline void addscl3(double* r, const double* a, double s) { r[0]+=a[0]*s; r[1]+=a[1]*s; r[2]+=a[2]*s; }
static inline double norm3(const double* a) { return sqrt(dot3(a,a)); }
static inline double normalize3(double* a) {
  double n = norm3(a);
  if (n < MINVAL) { a[0]=1; a[1]=0; a[2]=0; return n; }
  a[0]/=n; a[1]/=n; a[2]/=n; return n;
}
/* r = M(3x3 row-major) * v */
static inline void mulmat3(double* r, const double* m, const double* v) {
  double x = m[0]*v[0]+m[1]*v[1]+m[2]*v[2], y = m[3]*v[0]+m[4]*v[1]+m[5]*v[2], z = m[6]*v[0]+m[7]*v[1]+m[8]*v[2];
  r[0]=x; r[1]=y; r[2]=z;
}
static void quat_mul(double* r, const double* a, const double* b) {
  double w = a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3];
  double x = a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2];
  double y = a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1];
  double z = a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0];
  r[0]=w; r[1]=x; r[2]=y; r[3]=z;
}
static void quat2mat(double* m, const double* q) {
  double w=q[0], x=q[1], y=q[2], z=q[3];
  m[0]=1-2*(y*y+z*z); m[1]=2*(x*y-w*z);   m[2]=2*(x*z+w*y);
  m[3]=2*(x*y+w*z);   m[4]=1-2*(x*x+z*z); m[5]=2*(y*z-w*x);
  m[6]=2*(x*z-w*y);   m[7]=2*(y*z+w*x);   m[8]=1-2*(x*x+y*y);
}

/* ------------------------------------------------------------------------------------------ */
/* model                                                                                       */
/* ------------------------------------------------------------------------------------------ */
struct lmo_model {
  int nbody, nv, ngeom, nu, cone, integrator, iterations;
  double timestep, impratio, tolerance, gravity[3], meaninertia;
  double* blob;
  const double *body_parent, *body_pos, *body_quat, *body_mass, *body_ipos, *body_inertia, *body_jntadr,
      *body_jntnum, *body_weldid, *body_invweight0;
  const double *jnt_type, *jnt_body, *jnt_pos, *jnt_axis, *jnt_limited, *jnt_range, *jnt_stiffness,
      *jnt_margin, *jnt_solref, *jnt_solimp;
  const double *dof_damping, *dof_armature, *dof_frictionloss, *dof_solref, *dof_solimp, *dof_parent,
      *dof_invweight0;
  const double *geom_type, *geom_body, *geom_pos, *geom_quat, *geom_size, *geom_contype, *geom_conaffinity,
      *geom_condim, *geom_priority, *geom_friction, *geom_solmix, *geom_solref, *geom_solimp, *geom_margin,
      *geom_gap, *geom_center;
  const double *act_dof, *act_gear, *act_ctrlrange, *act_ctrllimited;
  int nsite, ntendon, nwrap, na;
  const double *site_body, *site_pos, *tendon_adr, *tendon_num, *wrap_site;
  const double *act_kind, *act_tendon, *act_dynprm, *act_gainprm, *act_lengthrange;
  const double *act_biasprm, *act_forcerange, *act_forcelimited;
  unsigned char affects[LMO_MAXBODY][LMO_MAXV]; /* dof d moves body b */
  /* static candidate geom pairs after type/affinity/parent filtering */
  int npair;
  int pair_g1[LMO_MAXPAIR], pair_g2[LMO_MAXPAIR];
  /* run-time switches (test hooks) */
  int disable_self_collision;
  int disable_ccd;             /* 1: no convex-convex (MPR) contacts; such pairs are counted instead (A/B tests) */
  int skip_pair_counter;       /* 1: pairs without a restated collider are not examined (no `unhandled_pairs`): timing runs */
  int disable_native;          /* 1: no contacts from the native box / cylinder colliders; such pairs are counted instead (A/B tests) */
  /* convex hulls attached to mesh geoms (lmo_set_mesh): hull vertices in the frame of the geom's BODY */
  int mesh_nvert[LMO_MAXGEOM]; double* mesh_vert[LMO_MAXGEOM];
  /* hull-graph neighbours of every hull vertex (CSR, nearest first; from the model blob or lmo_set_mesh_graph) and the
     distance below which a further plane-hull contact is too close to one already found */
  int* mesh_nbr_adr[LMO_MAXGEOM]; int* mesh_nbr[LMO_MAXGEOM]; double mesh_tol[LMO_MAXGEOM];
};

#define IDX(a, i) ((int)((a)[i]))
#define IDX2(a, i) ((a)[i])

lmo_model* lmo_model_create(const double* blob, long n) {
  if (n < LMH_HEADER_SIZE || (unsigned)blob[LMH_MAGIC] != LM_BLOB_MAGIC || (int)blob[LMH_VERSION] != LM_BLOB_VERSION) return NULL;
  lmo_model* m = (lmo_model*)calloc(1, sizeof(lmo_model));
  m->blob = (double*)malloc(sizeof(double) * (size_t)n);
  memcpy(m->blob, blob, sizeof(double) * (size_t)n);
  const double* p = m->blob;
  m->nbody = (int)p[LMH_NBODY]; m->nv = (int)p[LMH_NV]; m->ngeom = (int)p[LMH_NGEOM]; m->nu = (int)p[LMH_NU];
  m->cone = (int)p[LMH_CONE]; m->integrator = (int)p[LMH_INTEGRATOR]; m->iterations = (int)p[LMH_ITERATIONS];
  m->timestep = p[LMH_TIMESTEP]; m->impratio = p[LMH_IMPRATIO]; m->tolerance = p[LMH_TOLERANCE];
  m->gravity[0] = p[LMH_GRAV_X]; m->gravity[1] = p[LMH_GRAV_Y]; m->gravity[2] = p[LMH_GRAV_Z];
  m->meaninertia = p[LMH_MEANINERTIA];
  if (m->nbody > LMO_MAXBODY || m->nv > LMO_MAXV || m->ngeom > LMO_MAXGEOM) { free(m->blob); free(m); return NULL; }
  m->nsite = (int)p[LMH_NSITE]; m->ntendon = (int)p[LMH_NTENDON]; m->nwrap = (int)p[LMH_NWRAP]; m->na = (int)p[LMH_NA];
  if (m->nu > LMO_MAXU) { free(m->blob); free(m); return NULL; }
  int nb = m->nbody, nv = m->nv, ng = m->ngeom, nu = m->nu;
  p += LMH_HEADER_SIZE;
#define TAKE(field, count) m->field = p; p += (count)
  TAKE(body_parent, nb); TAKE(body_pos, 3*nb); TAKE(body_quat, 4*nb); TAKE(body_mass, nb); TAKE(body_ipos, 3*nb);
  TAKE(body_inertia, 9*nb); TAKE(body_jntadr, nb); TAKE(body_jntnum, nb); TAKE(body_weldid, nb);
  TAKE(body_invweight0, 2*nb);
  TAKE(jnt_type, nv); TAKE(jnt_body, nv); TAKE(jnt_pos, 3*nv); TAKE(jnt_axis, 3*nv); TAKE(jnt_limited, nv);
  TAKE(jnt_range, 2*nv); TAKE(jnt_stiffness, nv); TAKE(jnt_margin, nv); TAKE(jnt_solref, 2*nv); TAKE(jnt_solimp, 5*nv);
  TAKE(dof_damping, nv); TAKE(dof_armature, nv); TAKE(dof_frictionloss, nv); TAKE(dof_solref, 2*nv);
  TAKE(dof_solimp, 5*nv); TAKE(dof_parent, nv); TAKE(dof_invweight0, nv);
  TAKE(geom_type, ng); TAKE(geom_body, ng); TAKE(geom_pos, 3*ng); TAKE(geom_quat, 4*ng); TAKE(geom_size, 3*ng);
  TAKE(geom_contype, ng); TAKE(geom_conaffinity, ng); TAKE(geom_condim, ng); TAKE(geom_priority, ng);
  TAKE(geom_friction, 3*ng); TAKE(geom_solmix, ng); TAKE(geom_solref, 2*ng); TAKE(geom_solimp, 5*ng);
  TAKE(geom_margin, ng); TAKE(geom_gap, ng);
  TAKE(act_dof, nu); TAKE(act_gear, nu); TAKE(act_ctrlrange, 2*nu); TAKE(act_ctrllimited, nu);
  TAKE(site_body, m->nsite); TAKE(site_pos, 3*m->nsite); TAKE(tendon_adr, m->ntendon); TAKE(tendon_num, m->ntendon);
  TAKE(wrap_site, m->nwrap);
  TAKE(act_kind, nu); TAKE(act_tendon, nu); TAKE(act_dynprm, 3*nu); TAKE(act_gainprm, 9*nu); TAKE(act_lengthrange, 2*nu);
  TAKE(act_biasprm, 3*nu); TAKE(act_forcerange, 2*nu); TAKE(act_forcelimited, nu);
  const int nhull = (int)m->blob[LMH_NHULLVERT];
  const double *hull_adr, *hull_num, *hull_vert;
  hull_adr = p; p += ng; hull_num = p; p += ng; hull_vert = p; p += 3 * nhull;
  const int nnbr = (int)m->blob[LMH_NHULLNBR];
  const double* hull_nbr_adr = p; p += nhull + 1;
  const double* hull_nbr = p; p += nnbr;
  /* version 6: the point the convex-convex collider (MPR) takes as a geom's centre, in the frame of the geom's body: the geom
     frame origin; for a mesh geom the mesh's centre of mass (where the engine's compiler puts the geom frame) */
  m->geom_center = p; p += 3 * ng;
#undef TAKE
  if (p - m->blob != n) { free(m->blob); free(m); return NULL; }
  /* convex hulls that come with the model (mesh geoms): the same as lmo_set_mesh per geom */
  for (int g = 0; g < ng; g++) if (IDX(hull_num, g) > 0 && IDX(hull_adr, g) >= 0) {
    m->mesh_nvert[g] = IDX(hull_num, g);
    m->mesh_vert[g] = (double*)malloc(sizeof(double) * 3 * m->mesh_nvert[g]);
    memcpy(m->mesh_vert[g], hull_vert + 3 * IDX(hull_adr, g), sizeof(double) * 3 * m->mesh_nvert[g]);
    /* the hull's vertex graph, per geom; a further plane-hull contact keeps 0.3 x (bounding-capsule radius + half length)
       away from the contacts already found */
    const int nvg = m->mesh_nvert[g], a0 = IDX(hull_adr, g), e0 = IDX(hull_nbr_adr, a0), e1 = IDX(hull_nbr_adr, a0 + nvg);
    if (e1 > e0) {
      m->mesh_nbr_adr[g] = (int*)malloc(sizeof(int) * (nvg + 1));
      m->mesh_nbr[g] = (int*)malloc(sizeof(int) * (e1 - e0 + 1));
      for (int i = 0; i <= nvg; i++) m->mesh_nbr_adr[g][i] = IDX(hull_nbr_adr, a0 + i) - e0;
      for (int e = e0; e < e1; e++) m->mesh_nbr[g][e - e0] = IDX(hull_nbr, e);
      m->mesh_tol[g] = 0.3 * (IDX2(m->geom_size, 3*g) + IDX2(m->geom_size, 3*g + 1));
    }
  }

  for (int b = 1; b < nb; b++) {
    int a = b;
    while (a != 0) {
      for (int k = 0; k < IDX(m->body_jntnum, a); k++) m->affects[b][IDX(m->body_jntadr, a) + k] = 1;
      a = IDX(m->body_parent, a);
    }
  }
  /* candidate pairs: different weld group, not parent/child weld groups (unless one is the world),
     contype/conaffinity compatible (MuJoCo's geom filter) */
  m->npair = 0;
  for (int g1 = 0; g1 < ng; g1++)
    for (int g2 = g1 + 1; g2 < ng; g2++) {
      int b1 = IDX(m->geom_body, g1), b2 = IDX(m->geom_body, g2);
      int w1 = IDX(m->body_weldid, b1), w2 = IDX(m->body_weldid, b2);
      if (w1 == w2) continue;
      int pw1 = IDX(m->body_weldid, IDX(m->body_parent, w1)), pw2 = IDX(m->body_weldid, IDX(m->body_parent, w2));
      if (w1 != 0 && w2 != 0 && (w1 == pw2 || w2 == pw1)) continue;
      int ct1 = IDX(m->geom_contype, g1), ca1 = IDX(m->geom_conaffinity, g1);
      int ct2 = IDX(m->geom_contype, g2), ca2 = IDX(m->geom_conaffinity, g2);
      if (!((ct1 & ca2) || (ct2 & ca1))) continue;
      if (m->npair >= LMO_MAXPAIR) { free(m->blob); free(m); return NULL; }
      /* order so that the lower geom type comes first (plane first), like MuJoCo's collision table */
      int a = g1, c = g2;
      if (IDX(m->geom_type, a) > IDX(m->geom_type, c)) { a = g2; c = g1; }
      m->pair_g1[m->npair] = a; m->pair_g2[m->npair] = c; m->npair++;
    }
  return m;
}

void lmo_model_destroy(lmo_model* m) {
  if (!m) return;
  for (int g = 0; g < LMO_MAXGEOM; g++) { free(m->mesh_vert[g]); free(m->mesh_nbr_adr[g]); free(m->mesh_nbr[g]); }
  free(m->blob); free(m);
}
void lmo_set_option(lmo_model* m, int what, double value) {
  if (what == 0) m->disable_self_collision = (int)value;
  if (what == 1) m->iterations = (int)value;
  if (what == 2) m->tolerance = value;
  if (what == 3) m->skip_pair_counter = (int)value;
  if (what == 4) m->disable_ccd = (int)value;
  if (what == 5) m->disable_native = (int)value;
}

/* attach the convex hull of mesh geom g (nv hull vertices [nv][3] in the frame of the geom's body): the geom then collides
   with planes (one contact at the support vertex); without a hull a mesh geom is proximity-only */
/* the hull's vertex graph of mesh geom g (adr[nv + 1], nbr[adr[nv]], neighbours nearest first) and the distance `tol`: further
   plane-hull contacts at neighbours of the support vertex that penetrate and lie at least `tol` away from every contact already
   found, at most 3 of them (tests that attach hulls with lmo_set_mesh) */
int lmo_set_mesh_graph(lmo_model* m, int g, const int* adr, const int* nbr, double tol) {
  if (g < 0 || g >= m->ngeom || m->mesh_nvert[g] <= 0) return 1;
  int nv = m->mesh_nvert[g];
  free(m->mesh_nbr_adr[g]); free(m->mesh_nbr[g]);
  m->mesh_nbr_adr[g] = (int*)malloc(sizeof(int) * (nv + 1)); memcpy(m->mesh_nbr_adr[g], adr, sizeof(int) * (nv + 1));
  m->mesh_nbr[g] = (int*)malloc(sizeof(int) * (adr[nv] + 1)); memcpy(m->mesh_nbr[g], nbr, sizeof(int) * adr[nv]);
  m->mesh_tol[g] = tol;
  return 0;
}

int lmo_mesh_nvert(const lmo_model* m, int g) { return (g < 0 || g >= m->ngeom) ? 0 : m->mesh_nvert[g]; }

int lmo_set_mesh(lmo_model* m, int g, int nv, const double* vert) {
  if (g < 0 || g >= m->ngeom || nv <= 0) return 1;
  free(m->mesh_vert[g]);
  free(m->mesh_nbr_adr[g]); free(m->mesh_nbr[g]); m->mesh_nbr_adr[g] = NULL; m->mesh_nbr[g] = NULL;     /* new vertices: the old graph is void */
  m->mesh_nvert[g] = nv;
  m->mesh_vert[g] = (double*)malloc(sizeof(double) * 3 * nv); memcpy(m->mesh_vert[g], vert, sizeof(double) * 3 * nv);
  return 0;
}
int lmo_nv(const lmo_model* m) { return m->nv; }
int lmo_nu(const lmo_model* m) { return m->nu; }
int lmo_na(const lmo_model* m) { return m->na; }

/* ------------------------------------------------------------------------------------------ */
/* work data                                                                                   */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  /* kinematics */
  double xpos[LMO_MAXBODY][3], xquat[LMO_MAXBODY][4], xmat[LMO_MAXBODY][9], xipos[LMO_MAXBODY][3];
  double iw[LMO_MAXBODY][9];                       /* world-frame inertia about COM */
  double xanchor[LMO_MAXV][3], xaxis[LMO_MAXV][3];
  double gpos[LMO_MAXGEOM][3], gmat[LMO_MAXGEOM][9], gcen[LMO_MAXGEOM][3];
  /* dynamics */
  double M[LMO_MAXV * LMO_MAXV], L[LMO_MAXV * LMO_MAXV];
  double bias[LMO_MAXV], passive[LMO_MAXV], actuator[LMO_MAXV], smooth[LMO_MAXV], qacc_smooth[LMO_MAXV];
  /* contacts */
  int ncon;
  lmo_contact con[LMO_MAXCON];
  /* constraints */
  int nefc;
  int type[LMO_MAXEFC], id[LMO_MAXEFC];            /* row type; dof / contact id */
  double J[LMO_MAXEFC * LMO_MAXV];
  double pos[LMO_MAXEFC], margin[LMO_MAXEFC], diagApprox[LMO_MAXEFC], R[LMO_MAXEFC], D[LMO_MAXEFC];
  double K[LMO_MAXEFC], B[LMO_MAXEFC], imp[LMO_MAXEFC], floss[LMO_MAXEFC];
  double vel[LMO_MAXEFC], aref[LMO_MAXEFC], force[LMO_MAXEFC];
  int state[LMO_MAXEFC];
  double qfrc_constraint[LMO_MAXV], qacc[LMO_MAXV];
  double act_dot[LMO_MAXU], actuator_force[LMO_MAXU], actuator_length[LMO_MAXU], actuator_velocity[LMO_MAXU];
  int solver_iter;
  int unhandled_pairs;
  int convex_contacts; double max_self_depth;     /* per forward pass: MPR contacts, deepest non-floor penetration */
  int native_contacts;                            /* per forward pass: contacts of the native box / cylinder colliders */
  int own_contacts, own_face_contacts;            /* of those: capsule-box / box-box contacts (own constructions), and the box-box FACE case among them */
} work;

enum { ROW_FRICTION = 0, ROW_LIMIT = 1, ROW_CONTACT_PLAIN = 2, ROW_CONTACT_PYR = 3, ROW_CONTACT_ELL = 4 };
enum { ST_QUADRATIC = 0, ST_SATISFIED, ST_LINEARNEG, ST_LINEARPOS, ST_CONE };

/* ------------------------------------------------------------------------------------------ */
/* position stage                                                                              */
/* ------------------------------------------------------------------------------------------ */
static void kinematics(const lmo_model* m, const double* qpos, work* w) {
  memset(w->xpos[0], 0, sizeof(double) * 3);
  w->xquat[0][0] = 1; w->xquat[0][1] = w->xquat[0][2] = w->xquat[0][3] = 0;
  quat2mat(w->xmat[0], w->xquat[0]);
  memset(w->xipos[0], 0, sizeof(double) * 3);
  for (int i = 1; i < m->nbody; i++) {
    int p = IDX(m->body_parent, i);
    double pos[3], quat[4], r[9], t[3];
    mulmat3(t, w->xmat[p], m->body_pos + 3 * i);
    add3(pos, w->xpos[p], t);
    quat_mul(quat, w->xquat[p], m->body_quat + 4 * i);
    for (int k = 0; k < IDX(m->body_jntnum, i); k++) {
      int j = IDX(m->body_jntadr, i) + k;
      quat2mat(r, quat);
      mulmat3(t, r, m->jnt_pos + 3 * j);
      add3(w->xanchor[j], pos, t);
      mulmat3(w->xaxis[j], r, m->jnt_axis + 3 * j);
      if (IDX(m->jnt_type, j) == LM_JNT_SLIDE) {
        addscl3(pos, w->xaxis[j], qpos[j]);
      } else {
        double s = sin(0.5 * qpos[j]), ql[4], qn[4];
        ql[0] = cos(0.5 * qpos[j]); ql[1] = m->jnt_axis[3*j] * s; ql[2] = m->jnt_axis[3*j+1] * s; ql[3] = m->jnt_axis[3*j+2] * s;
        quat_mul(qn, quat, ql);
        memcpy(quat, qn, sizeof(qn));
        quat2mat(r, quat);
        mulmat3(t, r, m->jnt_pos + 3 * j);
        sub3(pos, w->xanchor[j], t);
      }
    }
    double n = sqrt(quat[0]*quat[0]+quat[1]*quat[1]+quat[2]*quat[2]+quat[3]*quat[3]);
    for (int k = 0; k < 4; k++) quat[k] /= n;
    copy3(w->xpos[i], pos); memcpy(w->xquat[i], quat, sizeof(quat));
    quat2mat(w->xmat[i], quat);
    mulmat3(t, w->xmat[i], m->body_ipos + 3 * i);
    add3(w->xipos[i], pos, t);
    /* iw = R I R^T */
    const double* I = m->body_inertia + 9 * i; const double* R = w->xmat[i];
    double RI[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
      double s = 0; for (int c = 0; c < 3; c++) s += R[3*a+c] * I[3*c+b]; RI[3*a+b] = s; }
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
      double s = 0; for (int c = 0; c < 3; c++) s += RI[3*a+c] * R[3*b+c]; w->iw[i][3*a+b] = s; }
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = IDX(m->geom_body, g);
    double t[3], q[4];
    mulmat3(t, w->xmat[b], m->geom_pos + 3 * g);
    add3(w->gpos[g], w->xpos[b], t);
    quat_mul(q, w->xquat[b], m->geom_quat + 4 * g);
    quat2mat(w->gmat[g], q);
    mulmat3(t, w->xmat[b], m->geom_center + 3 * g);
    add3(w->gcen[g], w->xpos[b], t);
  }
}

/* translational (jp, 3 x nv) and rotational (jr, 3 x nv) Jacobian of a point attached to body b */
static void jac_point(const lmo_model* m, const work* w, int b, const double* point, double* jp, double* jr) {
  int nv = m->nv;
  memset(jp, 0, sizeof(double) * 3 * nv);
  if (jr) memset(jr, 0, sizeof(double) * 3 * nv);
  for (int d = 0; d < nv; d++) {
    if (!m->affects[b][d]) continue;
    const double* ax = w->xaxis[d];
    if (IDX(m->jnt_type, d) == LM_JNT_SLIDE) {
      jp[d] = ax[0]; jp[nv + d] = ax[1]; jp[2*nv + d] = ax[2];
    } else {
      double r[3], c[3];
      sub3(r, point, w->xanchor[d]);
      cross3(c, ax, r);
      jp[d] = c[0]; jp[nv + d] = c[1]; jp[2*nv + d] = c[2];
      if (jr) { jr[d] = ax[0]; jr[nv + d] = ax[1]; jr[2*nv + d] = ax[2]; }
    }
  }
}

static void mass_matrix(const lmo_model* m, work* w) {
  int nv = m->nv;
  double jp[3 * LMO_MAXV], jr[3 * LMO_MAXV], t[3 * LMO_MAXV];
  memset(w->M, 0, sizeof(double) * nv * nv);
  for (int b = 1; b < m->nbody; b++) {
    double mass = m->body_mass[b];
    jac_point(m, w, b, w->xipos[b], jp, jr);
    /* t = Iw * jr */
    for (int a = 0; a < 3; a++) for (int d = 0; d < nv; d++)
      t[a*nv + d] = w->iw[b][3*a] * jr[d] + w->iw[b][3*a+1] * jr[nv + d] + w->iw[b][3*a+2] * jr[2*nv + d];
    for (int i = 0; i < nv; i++) {
      if (!m->affects[b][i]) continue;
      for (int j = 0; j < nv; j++) {
        if (!m->affects[b][j]) continue;
        double s = 0;
        for (int a = 0; a < 3; a++) s += mass * jp[a*nv + i] * jp[a*nv + j] + jr[a*nv + i] * t[a*nv + j];
        w->M[i*nv + j] += s;
      }
    }
  }
  for (int i = 0; i < nv; i++) w->M[i*nv + i] += m->dof_armature[i];
}

/* dense Cholesky A = L L^T (lower), returns 0 on success */
static int cholesky(double* L, const double* A, int n) {
  memcpy(L, A, sizeof(double) * n * n);
  for (int j = 0; j < n; j++) {
    double s = L[j*n + j];
    for (int k = 0; k < j; k++) s -= L[j*n + k] * L[j*n + k];
    if (s <= 0) return 1;
    double d = sqrt(s);
    L[j*n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double t = L[i*n + j];
      for (int k = 0; k < j; k++) t -= L[i*n + k] * L[j*n + k];
      L[i*n + j] = t / d;
    }
  }
  return 0;
}
static void chol_solve(const double* L, int n, double* x) {
  for (int i = 0; i < n; i++) { double s = x[i]; for (int k = 0; k < i; k++) s -= L[i*n + k] * x[k]; x[i] = s / L[i*n + i]; }
  for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= L[k*n + i] * x[k]; x[i] = s / L[i*n + i]; }
}

/* ------------------------------------------------------------------------------------------ */
/* collision                                                                                   */
/* ------------------------------------------------------------------------------------------ */
static void make_frame(double* frame) {
  /* frame[0..2] = normal; complete to a right-handed orthonormal frame (rows) */
  normalize3(frame);
  double* y = frame + 3; double* z = frame + 6;
  y[0] = y[1] = y[2] = 0;
  if (frame[1] < 0.5 && frame[1] > -0.5) y[1] = 1; else y[2] = 1;
  double t = dot3(frame, y);
  addscl3(y, frame, -t);
  normalize3(y);
  cross3(z, frame, y);
}

static void contact_params(const lmo_model* m, int g1, int g2, lmo_contact* c) {
  int p1 = IDX(m->geom_priority, g1), p2 = IDX(m->geom_priority, g2);
  const double *f1 = m->geom_friction + 3*g1, *f2 = m->geom_friction + 3*g2;
  double fr[3];
  if (p1 != p2) {
    int g = p1 > p2 ? g1 : g2;
    c->dim = IDX(m->geom_condim, g);
    memcpy(c->solref, m->geom_solref + 2*g, 2 * sizeof(double));
    memcpy(c->solimp, m->geom_solimp + 5*g, 5 * sizeof(double));
    memcpy(fr, m->geom_friction + 3*g, 3 * sizeof(double));
  } else {
    int d1 = IDX(m->geom_condim, g1), d2 = IDX(m->geom_condim, g2);
    c->dim = d1 > d2 ? d1 : d2;
    double s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2], mix;
    if (s1 >= MINVAL && s2 >= MINVAL) mix = s1 / (s1 + s2);
    else if (s1 < MINVAL && s2 < MINVAL) mix = 0.5;
    else if (s1 < MINVAL) mix = 0.0; else mix = 1.0;
    const double *r1 = m->geom_solref + 2*g1, *r2 = m->geom_solref + 2*g2;
    if (r1[0] > 0 && r2[0] > 0) for (int k = 0; k < 2; k++) c->solref[k] = mix * r1[k] + (1 - mix) * r2[k];
    else for (int k = 0; k < 2; k++) c->solref[k] = r1[k] < r2[k] ? r1[k] : r2[k];
    for (int k = 0; k < 5; k++) c->solimp[k] = mix * m->geom_solimp[5*g1 + k] + (1 - mix) * m->geom_solimp[5*g2 + k];
    for (int k = 0; k < 3; k++) fr[k] = f1[k] > f2[k] ? f1[k] : f2[k];
  }
  c->friction[0] = fr[0]; c->friction[1] = fr[0]; c->friction[2] = fr[1]; c->friction[3] = fr[2]; c->friction[4] = fr[2];
  double mg1 = m->geom_margin[g1], mg2 = m->geom_margin[g2], gp1 = m->geom_gap[g1], gp2 = m->geom_gap[g2];
  c->margin = mg1 > mg2 ? mg1 : mg2;
  c->includemargin = c->margin - (gp1 > gp2 ? gp1 : gp2);
  c->geom1 = g1; c->geom2 = g2;
}

static int add_contact(work* w, const lmo_contact* tmpl, double dist, const double* pos, const double* normal,
                       const double* yaxis_hint) {
  if (w->ncon >= LMO_MAXCON) return 0;
  lmo_contact* c = &w->con[w->ncon++];
  *c = *tmpl;
  c->dist = dist;
  copy3(c->pos, pos);
  copy3(c->frame, normal);
  make_frame(c->frame);
  (void)yaxis_hint;
  return 1;
}

/* closest points between two segments p1 + s*d1 (|s|<=h1), p2 + t*d2 (|t|<=h2); d unit */
static void segment_closest(const double* p1, const double* d1, double h1, const double* p2, const double* d2,
                            double h2, double* s_out, double* t_out) {
  double r[3]; sub3(r, p1, p2);
  double b = dot3(d1, d2), c = dot3(d1, r), f = dot3(d2, r);
  double den = 1 - b * b, s, t;
  if (den > 1e-12) {
    s = (b * f - c) / den;
    if (s < -h1) s = -h1; else if (s > h1) s = h1;
  } else s = 0;
  t = b * s + f;
  if (t < -h2) { t = -h2; s = b * t - c; if (s < -h1) s = -h1; else if (s > h1) s = h1; }
  else if (t > h2) { t = h2; s = b * t - c; if (s < -h1) s = -h1; else if (s > h1) s = h1; }
  *s_out = s; *t_out = t;
}

static void sphere_sphere(work* w, const lmo_contact* tm, const double* c1, double r1, const double* c2, double r2) {
  double n[3]; sub3(n, c2, c1);
  double d = norm3(n);
  double dist = d - r1 - r2;
  if (dist >= tm->margin) return;
  if (d < MINVAL) { n[0] = 1; n[1] = 0; n[2] = 0; } else { n[0] /= d; n[1] /= d; n[2] /= d; }
  double pos[3] = { c1[0] + n[0] * (r1 + 0.5 * dist), c1[1] + n[1] * (r1 + 0.5 * dist), c1[2] + n[2] * (r1 + 0.5 * dist) };
  add_contact(w, tm, dist, pos, n, NULL);
}

/* largest gap between two oriented boxes over the 15 candidate separating axes (<= true distance;
   negative when the boxes overlap). R are row-major world rotations: column k = box axis k. */
static double box_box_gap(const double* p1, const double* R1, const double* s1,
                          const double* p2, const double* R2, const double* s2) {
  double d[3], best = -1e30; sub3(d, p2, p1);
  double ax[15][3]; int na = 0;
  for (int k = 0; k < 3; k++) { ax[na][0] = R1[k]; ax[na][1] = R1[3+k]; ax[na][2] = R1[6+k]; na++; }
  for (int k = 0; k < 3; k++) { ax[na][0] = R2[k]; ax[na][1] = R2[3+k]; ax[na][2] = R2[6+k]; na++; }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double c[3]; cross3(c, ax[i], ax[3+j]);
    double n = norm3(c); if (n < 1e-9) continue;
    ax[na][0] = c[0]/n; ax[na][1] = c[1]/n; ax[na][2] = c[2]/n; na++;
  }
  for (int a = 0; a < na; a++) {
    double r1 = 0, r2 = 0;
    for (int k = 0; k < 3; k++) {
      r1 += s1[k] * fabs(ax[a][0]*R1[k] + ax[a][1]*R1[3+k] + ax[a][2]*R1[6+k]);
      r2 += s2[k] * fabs(ax[a][0]*R2[k] + ax[a][1]*R2[3+k] + ax[a][2]*R2[6+k]);
    }
    double gap = fabs(dot3(d, ax[a])) - r1 - r2;
    if (gap > best) best = gap;
  }
  return best;
}

/* ------------------------------------------------------------------------------------------ */
/* The engine's NATIVE colliders for box / cylinder pairs (mujoco 2.3.7 engine_collision_box.c,     */
/* engine_collision_primitive.c: mjc_SphereBox, mjc_SphereCylinder, mjc_CapsuleBox, mjc_BoxBox) —   */
/* third party, not under /root/reference, not installed. Sphere-box and sphere-cylinder are fully    */
/* determined by the geometry (closest feature, midpoint, signed distance) and restated as such.     */
/* Capsule-box and box-box are NOT restatements of the engine's case analysis (mjc_CapsuleBox /     */
/* mjc_BoxBox decide which contacts of a manifold are kept; their source is not available here):      */
/* they are this repository's own constructions with the engine's contact conventions, the same ones  */
/* the device carries (csrc/lm_core.h nat_*), so device-vs-oracle agreement is circular for these two */
/* pair types and only the golden rows speak — capsule: the closest point of the axis                */
/* segment as a sphere against the box + a second contact at the far end when that is within the     */
/* margin too; boxes: separating-axis search over the 15 axes, then either the incident face clipped  */
/* against the reference face (<= 8 contacts) or the closest points of the two edges — NOT pinned by  */
/* any golden row of the reference (none has such a contact): "parity unpinned" for these two pair    */
/* types. tests/test_oracle_golden.py checks every contact against brute-force geometry.              */
/* out: rows of (dist, pos 3, normal 3); normal from geom 1 to geom 2. R: row-major, columns = axes.  */
/* ------------------------------------------------------------------------------------------ */
static int nat_sphere_box(const double* c, double r, const double* pb, const double* Rb, const double* sb, double margin, double (*out)[7]) {
  double rel[3], ctr[3], cl[3], d[3];
  sub3(rel, c, pb);
  for (int k = 0; k < 3; k++) ctr[k] = Rb[k] * rel[0] + Rb[3 + k] * rel[1] + Rb[6 + k] * rel[2];
  for (int k = 0; k < 3; k++) { cl[k] = ctr[k] < -sb[k] ? -sb[k] : (ctr[k] > sb[k] ? sb[k] : ctr[k]); d[k] = cl[k] - ctr[k]; }
  double dist = norm3(d);
  if (dist - r >= margin) return 0;
  double nl[3] = {0, 0, 0}, pl[3];
  if (dist <= MINVAL) {          /* centre inside the box: out through the nearest face */
    double closest = 2 * (sb[0] + sb[1] + sb[2]); int kk = 0;
    for (int i = 0; i < 6; i++) { double fd = fabs(((i % 2) ? 1.0 : -1.0) * sb[i / 2] - ctr[i / 2]); if (closest > fd) { closest = fd; kk = i; } }
    nl[kk / 2] = (kk % 2) ? -1.0 : 1.0;
    for (int k = 0; k < 3; k++) pl[k] = ctr[k] + nl[k] * (r - closest) * 0.5;
    dist = -closest;
  } else {
    for (int k = 0; k < 3; k++) { nl[k] = d[k] / dist; pl[k] = 0.5 * (cl[k] + ctr[k] + nl[k] * r); }
  }
  out[0][0] = dist - r;
  for (int k = 0; k < 3; k++) {
    out[0][1 + k] = pb[k] + Rb[3 * k] * pl[0] + Rb[3 * k + 1] * pl[1] + Rb[3 * k + 2] * pl[2];
    out[0][4 + k] = Rb[3 * k] * nl[0] + Rb[3 * k + 1] * nl[1] + Rb[3 * k + 2] * nl[2];
  }
  return 1;
}

/* sphere (centre c1, radius r1) against sphere (c2, r2): the engine's mjraw_SphereSphere */
static int nat_sphere_sphere(const double* c1, double r1, const double* c2, double r2, double margin, double (*out)[7]) {
  double n[3]; sub3(n, c2, c1);
  double d = norm3(n), dist = d - r1 - r2;
  if (dist >= margin) return 0;
  if (d < MINVAL) { n[0] = 1; n[1] = 0; n[2] = 0; } else { n[0] /= d; n[1] /= d; n[2] /= d; }
  out[0][0] = dist;
  for (int k = 0; k < 3; k++) { out[0][1 + k] = c1[k] + n[k] * (r1 + 0.5 * dist); out[0][4 + k] = n[k]; }
  return 1;
}

static int nat_sphere_cylinder(const double* c, double r, const double* pc, const double* Rc, const double* sc, double margin, double (*out)[7]) {
  const double radius = sc[0], height = sc[1];
  double axis[3] = { Rc[2], Rc[5], Rc[8] }, vec[3], ap[3], pp[3];
  sub3(vec, c, pc);
  const double x = dot3(vec, axis);
  for (int k = 0; k < 3; k++) { ap[k] = axis[k] * x; pp[k] = vec[k] - ap[k]; }
  const double pp2 = dot3(pp, pp);
  int side = fabs(x) < height, cap = pp2 < radius * radius;
  if (side && cap) {            /* centre inside the cylinder: out through the nearer of wall and cap */
    if (height - fabs(x) < radius - sqrt(pp2)) side = 0; else cap = 0;
  }
  if (side) {                   /* against the wall: a sphere on the axis */
    double tgt[3]; add3(tgt, pc, ap);
    return nat_sphere_sphere(c, r, tgt, radius, margin, out);
  }
  if (cap) {                    /* against a cap: plane-sphere, the normal turned from the sphere to the cylinder */
    const double sg = (x > 0) ? 1.0 : -1.0;
    double n[3] = { sg * axis[0], sg * axis[1], sg * axis[2] }, pcap[3], rel[3];
    for (int k = 0; k < 3; k++) pcap[k] = pc[k] + n[k] * height;
    sub3(rel, c, pcap);
    const double cd = dot3(rel, n);
    if (cd - r >= margin) return 0;
    out[0][0] = cd - r;
    for (int k = 0; k < 3; k++) { out[0][1 + k] = c[k] + n[k] * (-0.5 * (cd - r) - r); out[0][4 + k] = -n[k]; }
    return 1;
  }
  /* against the rim: a point */
  double tgt[3];
  const double sc_ = radius / sqrt(pp2 > MINVAL ? pp2 : MINVAL);
  for (int k = 0; k < 3; k++) tgt[k] = pc[k] + axis[k] * (x > 0 ? height : -height) + pp[k] * sc_;
  return nat_sphere_sphere(c, r, tgt, 0.0, margin, out);
}

/* capsule (centre pc, axis = third column of Rc, half length sc[1], radius sc[0]) against a box */
static int nat_capsule_box(const double* pc, const double* Rc, const double* sc, const double* pb, const double* Rb, const double* sb,
                           double margin, double (*out)[7]) {
  const double r = sc[0], half = sc[1];
  double axw[3] = { Rc[2], Rc[5], Rc[8] }, rel[3], p[3], a[3];
  sub3(rel, pc, pb);
  for (int k = 0; k < 3; k++) {
    p[k] = Rb[k] * rel[0] + Rb[3 + k] * rel[1] + Rb[6 + k] * rel[2];
    a[k] = (Rb[k] * axw[0] + Rb[3 + k] * axw[1] + Rb[6 + k] * axw[2]) * half;
  }
  /* g'(t) = (P(t) - clamp(P(t))) . a is non-decreasing along the segment P(t) = p + t a: its leftmost non-negative point is the
     (leftmost) closest point of the axis to the box */
#define NAT_GP(t, res) do { double s_ = 0; for (int k_ = 0; k_ < 3; k_++) { const double P_ = p[k_] + (t) * a[k_]; \
    const double c_ = P_ < -sb[k_] ? -sb[k_] : (P_ > sb[k_] ? sb[k_] : P_); s_ += (P_ - c_) * a[k_]; } res = s_; } while (0)
  double g0, g1, t1;
  NAT_GP(-1.0, g0); NAT_GP(1.0, g1);
  if (g0 >= 0) t1 = -1.0;
  else if (g1 < 0) t1 = 1.0;
  else {
    double lo = -1.0, hi = 1.0;
    for (int it = 0; it < 48; it++) { const double mid = 0.5 * (lo + hi); double gm; NAT_GP(mid, gm); if (gm < 0) lo = mid; else hi = mid; }
    t1 = hi;
  }
#undef NAT_GP
  int n = 0;
  double c1[3];
  for (int k = 0; k < 3; k++) c1[k] = pc[k] + axw[k] * half * t1;
  n += nat_sphere_box(c1, r, pb, Rb, sb, margin, out + n);
  /* the far end of the axis, when the capsule lies along the box (both within the margin): a second contact */
  const double t2 = (t1 <= 0) ? 1.0 : -1.0;
  if (n > 0 && fabs(t2 - t1) * half > 1e-9) {
    double c2[3];
    for (int k = 0; k < 3; k++) c2[k] = pc[k] + axw[k] * half * t2;
    n += nat_sphere_box(c2, r, pb, Rb, sb, margin, out + n);
  }
  return n;
}

/* which branch the last nat_box_box call of this thread took: 1 = the incident face clipped against the reference face (the FACE case:
   this repository's own construction, approximate), 0 = an edge pair (exact where the geometry leaves no choice) */
static __thread int g_boxbox_face_case;

/* box against box */
static int nat_box_box(const double* p1, const double* R1, const double* s1, const double* p2, const double* R2, const double* s2,
                       double margin, double (*out)[7]) {
  double A[3][3], B[3][3], d[3];
  for (int k = 0; k < 3; k++) for (int j = 0; j < 3; j++) { A[k][j] = R1[3 * j + k]; B[k][j] = R2[3 * j + k]; }   /* axis k as a vector */
  sub3(d, p2, p1);
  double best_face = -1e300, best_edge = -1e300, nf[3] = {0, 0, 0}, ne[3] = {0, 0, 0};
  int face = 0, ei = 0, ej = 0;
  for (int f = 0; f < 6; f++) {
    const double* n = (f < 3) ? A[f] : B[f - 3];
    double rA = 0, rB = 0;
    for (int k = 0; k < 3; k++) { rA += s1[k] * fabs(dot3(n, A[k])); rB += s2[k] * fabs(dot3(n, B[k])); }
    const double pr = dot3(d, n), gap = fabs(pr) - rA - rB;
    if (gap > best_face) { best_face = gap; face = f; const double sg = pr >= 0 ? 1.0 : -1.0; for (int k = 0; k < 3; k++) nf[k] = sg * n[k]; }
  }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double n[3]; cross3(n, A[i], B[j]);
    const double l = norm3(n);
    if (l < 1e-6) continue;
    for (int k = 0; k < 3; k++) n[k] /= l;
    double rA = 0, rB = 0;
    for (int k = 0; k < 3; k++) { rA += s1[k] * fabs(dot3(n, A[k])); rB += s2[k] * fabs(dot3(n, B[k])); }
    const double pr = dot3(d, n), gap = fabs(pr) - rA - rB;
    if (gap > best_edge) { best_edge = gap; ei = i; ej = j; const double sg = pr >= 0 ? 1.0 : -1.0; for (int k = 0; k < 3; k++) ne[k] = sg * n[k]; }
  }
  g_boxbox_face_case = 0;
  if (best_face >= margin || best_edge >= margin) return 0;
  /* an edge pair decides only when it separates clearly better than every face (5 % + 1 um) */
  if (best_edge > best_face + 0.05 * fabs(best_face) + 1e-6) {
    double pa[3], pb_[3];
    copy3(pa, p1); copy3(pb_, p2);
    for (int k = 0; k < 3; k++) if (k != ei) addscl3(pa, A[k], (dot3(ne, A[k]) > 0 ? 1.0 : -1.0) * s1[k]);
    for (int k = 0; k < 3; k++) if (k != ej) addscl3(pb_, B[k], (dot3(ne, B[k]) > 0 ? -1.0 : 1.0) * s2[k]);
    double s, t;
    segment_closest(pa, A[ei], s1[ei], pb_, B[ej], s2[ej], &s, &t);
    double ca[3], cb[3];
    copy3(ca, pa); addscl3(ca, A[ei], s); copy3(cb, pb_); addscl3(cb, B[ej], t);
    out[0][0] = best_edge;
    for (int k = 0; k < 3; k++) { out[0][1 + k] = 0.5 * (ca[k] + cb[k]); out[0][4 + k] = ne[k]; }
    return 1;
  }
  /* face contact: the incident face of the other box clipped against the reference face */
  g_boxbox_face_case = 1;
  const int ref1 = face < 3, kr = face % 3;
  const double *pr_ = ref1 ? p1 : p2, *sr = ref1 ? s1 : s2, *pi_ = ref1 ? p2 : p1, *si = ref1 ? s2 : s1;
  double (*Ar)[3] = ref1 ? A : B, (*Ai)[3] = ref1 ? B : A;
  double nr[3];
  for (int k = 0; k < 3; k++) nr[k] = ref1 ? nf[k] : -nf[k];           /* out of the reference face, towards the incident box */
  int ji = 0; double bj = -1;
  for (int j = 0; j < 3; j++) { const double v = fabs(dot3(nr, Ai[j])); if (v > bj) { bj = v; ji = j; } }
  const double sgi = dot3(nr, Ai[ji]) > 0 ? -1.0 : 1.0;                  /* the incident face looks back at the reference box */
  const int u = (ji + 1) % 3, v = (ji + 2) % 3, ur = (kr + 1) % 3, vr = (kr + 2) % 3;
  double poly[16][3], tmp[16][3];
  int np_ = 4;
  for (int q = 0; q < 4; q++) {
    const double su = (q == 0 || q == 3) ? 1.0 : -1.0, sv = (q < 2) ? 1.0 : -1.0;
    double w[3];
    for (int k = 0; k < 3; k++) w[k] = pi_[k] + sgi * si[ji] * Ai[ji][k] + su * si[u] * Ai[u][k] + sv * si[v] * Ai[v][k] - pr_[k];
    poly[q][0] = dot3(w, Ar[ur]); poly[q][1] = dot3(w, Ar[vr]); poly[q][2] = dot3(w, nr) - sr[kr];     /* (x, y) on the reference face, height above it */
  }
  for (int side = 0; side < 4; side++) {
    const int c = side >> 1; const double sg = (side & 1) ? -1.0 : 1.0, lim = sr[c ? vr : ur];
    int nn = 0;
    for (int q = 0; q < np_; q++) {
      const double* P = poly[q]; const double* Q = poly[(q + 1) % np_];
      const double dp = lim - sg * P[c], dq = lim - sg * Q[c];
      if (dp >= 0) { copy3(tmp[nn], P); nn++; }
      if ((dp >= 0) != (dq >= 0)) { const double f = dp / (dp - dq); for (int k = 0; k < 3; k++) tmp[nn][k] = P[k] + f * (Q[k] - P[k]); nn++; }
    }
    np_ = nn;
    for (int q = 0; q < np_; q++) copy3(poly[q], tmp[q]);
    if (np_ == 0) break;
  }
  int n = 0;
  for (int q = 0; q < np_ && n < 8; q++) {
    const double h = poly[q][2];
    if (h >= margin) continue;
    out[n][0] = h;
    for (int k = 0; k < 3; k++) {
      out[n][1 + k] = pr_[k] + poly[q][0] * Ar[ur][k] + poly[q][1] * Ar[vr][k] + (sr[kr] + 0.5 * h) * nr[k];
      out[n][4 + k] = nf[k];
    }
    n++;
  }
  return n;
}

/* the pair's native collider; types in the engine's order (t1 <= t2). Returns the number of contacts, -1 = no native collider */
static int native_pair(int t1, const double* p1, const double* R1, const double* s1, int t2, const double* p2, const double* R2,
                       const double* s2, double margin, double (*out)[7]) {
  if (t1 == LM_GEOM_SPHERE && t2 == LM_GEOM_BOX) return nat_sphere_box(p1, s1[0], p2, R2, s2, margin, out);
  if (t1 == LM_GEOM_SPHERE && t2 == LM_GEOM_CYLINDER) return nat_sphere_cylinder(p1, s1[0], p2, R2, s2, margin, out);
  if (t1 == LM_GEOM_CAPSULE && t2 == LM_GEOM_BOX) return nat_capsule_box(p1, R1, s1, p2, R2, s2, margin, out);
  if (t1 == LM_GEOM_BOX && t2 == LM_GEOM_BOX) return nat_box_box(p1, R1, s1, p2, R2, s2, margin, out);
  return -1;
}

int lmo_test_native_pair(int t1, const double* p1, const double* R1, const double* s1, int t2, const double* p2, const double* R2,
                         const double* s2, double margin, double* out /* [8][7] */) {
  double buf[8][7];
  const int n = native_pair(t1, p1, R1, s1, t2, p2, R2, s2, margin, buf);
  for (int i = 0; i < n && i < 8; i++) memcpy(out + 7 * i, buf[i], sizeof(buf[i]));
  return n;
}

static double rbound(int type, const double* size) {
  switch (type) {
    case LM_GEOM_SPHERE: return size[0];
    case LM_GEOM_CAPSULE: return size[0] + size[1];
    case LM_GEOM_CYLINDER: return sqrt(size[0]*size[0] + size[1]*size[1]);
    case LM_GEOM_BOX: return norm3(size);
    case LM_GEOM_MESH: return size[0] + size[1];   /* proximity-only geom: bounding capsule (loco_mujoco_amd/mjcf.py) */
    default: return 0;
  }
}

/* bounding capsule (centre, unit axis, half length, radius) of a geom: sphere / capsule exact, cylinder and box conservative */
static void bounding_capsule(int type, const double* size, const double* pos, const double* R, double* c, double* a,
                             double* half, double* rad) {
  copy3(c, pos);
  int k = 2;
  if (type == LM_GEOM_BOX) { k = 0; if (size[1] > size[k]) k = 1; if (size[2] > size[k]) k = 2; }
  a[0] = R[k]; a[1] = R[3 + k]; a[2] = R[6 + k];
  switch (type) {
    case LM_GEOM_SPHERE: *half = 0; *rad = size[0]; break;
    case LM_GEOM_BOX: { double o2 = 0; for (int j = 0; j < 3; j++) if (j != k) o2 += size[j] * size[j]; *half = size[k]; *rad = sqrt(o2); break; }
    default: *half = size[1]; *rad = size[0]; break;      /* capsule, cylinder, mesh (its bounding capsule) */
  }
}

/* ------------------------------------------------------------------------------------------ */
/* distance between two convex geoms (GJK), used ONLY to decide whether a pair WITHOUT a restated   */
/* collider is within its contact margin — i.e. whether the engine would have produced a contact   */
/* here that this restatement lacks (`unhandled_pairs`). No contact is generated from it.          */
/* Core shapes: sphere = point, capsule = segment, box = 8 corners, cylinder = two discs, mesh =     */
/* its hull vertices (body frame); spheres and capsules add their radius afterwards.                */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int type; const double* pos; const double* R; const double* size; const double* verts; int nverts;
                 const double* bpos; const double* bmat; } cvx;

static double cvx_radius(const cvx* g) { return (g->type == LM_GEOM_SPHERE || g->type == LM_GEOM_CAPSULE) ? g->size[0] : 0.0; }

static void cvx_support(const cvx* g, const double* dir, double* out) {
  if (g->type == LM_GEOM_MESH) {
    /* vertices in the BODY frame: direction into that frame, best vertex, back to the world */
    double dl[3] = { g->bmat[0]*dir[0] + g->bmat[3]*dir[1] + g->bmat[6]*dir[2], g->bmat[1]*dir[0] + g->bmat[4]*dir[1] + g->bmat[7]*dir[2],
                     g->bmat[2]*dir[0] + g->bmat[5]*dir[1] + g->bmat[8]*dir[2] };
    int best = 0; double db = -1e300;
    for (int i = 0; i < g->nverts; i++) { double d = dot3(g->verts + 3*i, dl); if (d > db) { db = d; best = i; } }
    mulmat3(out, g->bmat, g->verts + 3*best); add3(out, out, g->bpos);
    return;
  }
  double ax[3] = { g->R[2], g->R[5], g->R[8] };
  copy3(out, g->pos);
  switch (g->type) {
    case LM_GEOM_SPHERE: break;
    case LM_GEOM_CAPSULE: addscl3(out, ax, dot3(ax, dir) >= 0 ? g->size[1] : -g->size[1]); break;
    case LM_GEOM_BOX:
      for (int k = 0; k < 3; k++) { double e[3] = { g->R[k], g->R[3 + k], g->R[6 + k] }; addscl3(out, e, dot3(e, dir) >= 0 ? g->size[k] : -g->size[k]); }
      break;
    case LM_GEOM_CYLINDER: {
      double da = dot3(ax, dir), perp[3] = { dir[0] - da*ax[0], dir[1] - da*ax[1], dir[2] - da*ax[2] };
      double np_ = norm3(perp);
      addscl3(out, ax, da >= 0 ? g->size[1] : -g->size[1]);
      if (np_ > 1e-14) addscl3(out, perp, g->size[0] / np_);
      break;
    }
    default: break;
  }
}

/* closest point to the origin on the simplex (1..4 points); reduces the simplex to the supporting face; returns 1 if the
   origin lies inside a tetrahedron (the shapes overlap) */
static int simplex_closest(double (*S)[3], int* n, double* v) {
  if (*n == 1) { copy3(v, S[0]); return 0; }
  double best = 1e300, bv[3] = {0, 0, 0}; int bidx[4], bn = 0;
  /* points */
  for (int i = 0; i < *n; i++) { double d = dot3(S[i], S[i]); if (d < best) { best = d; copy3(bv, S[i]); bidx[0] = i; bn = 1; } }
  /* edges */
  for (int i = 0; i < *n; i++) for (int j = i + 1; j < *n; j++) {
    double e[3]; sub3(e, S[j], S[i]);
    double ee = dot3(e, e); if (ee < 1e-300) continue;
    double t = -dot3(S[i], e) / ee; if (t <= 0 || t >= 1) continue;
    double q[3]; copy3(q, S[i]); addscl3(q, e, t);
    double d = dot3(q, q); if (d < best) { best = d; copy3(bv, q); bidx[0] = i; bidx[1] = j; bn = 2; }
  }
  /* triangles */
  for (int i = 0; i < *n; i++) for (int j = i + 1; j < *n; j++) for (int k = j + 1; k < *n; k++) {
    double e1[3], e2[3], nn[3]; sub3(e1, S[j], S[i]); sub3(e2, S[k], S[i]); cross3(nn, e1, e2);
    double n2 = dot3(nn, nn); if (n2 < 1e-300) continue;
    double t = dot3(S[i], nn) / n2, q[3] = { t*nn[0], t*nn[1], t*nn[2] };      /* projection of the origin on the plane */
    /* inside test by barycentric coordinates */
    double r[3]; sub3(r, q, S[i]);
    double d11 = dot3(e1, e1), d12 = dot3(e1, e2), d22 = dot3(e2, e2), r1 = dot3(r, e1), r2 = dot3(r, e2), det = d11*d22 - d12*d12;
    if (fabs(det) < 1e-300) continue;
    double a = (d22*r1 - d12*r2) / det, b = (d11*r2 - d12*r1) / det;
    if (a <= 0 || b <= 0 || a + b >= 1) continue;
    double d = dot3(q, q); if (d < best) { best = d; copy3(bv, q); bidx[0] = i; bidx[1] = j; bidx[2] = k; bn = 3; }
  }
  if (*n == 4) {
    /* origin inside the tetrahedron? same side of every face as the opposite vertex */
    int inside = 1;
    for (int f = 0; f < 4 && inside; f++) {
      int a = (f + 1) & 3, b = (f + 2) & 3, c = (f + 3) & 3;
      double e1[3], e2[3], nn[3]; sub3(e1, S[b], S[a]); sub3(e2, S[c], S[a]); cross3(nn, e1, e2);
      double so = -dot3(S[a], nn), sd; double df[3]; sub3(df, S[f], S[a]); sd = dot3(df, nn);
      if (fabs(sd) < 1e-300) { inside = 0; break; }
      if ((so > 0) != (sd > 0) && so != 0) inside = 0;
    }
    if (inside) { v[0] = v[1] = v[2] = 0; return 1; }
  }
  double T[4][3];
  for (int i = 0; i < bn; i++) copy3(T[i], S[bidx[i]]);
  for (int i = 0; i < bn; i++) copy3(S[i], T[i]);
  *n = bn; copy3(v, bv);
  return 0;
}

/* distance between the two geoms' surfaces (negative or zero: they overlap), to about 1e-9 relative */
static double convex_distance(const cvx* A, const cvx* B) {
  double S[4][3], v[3], dir[3], a[3], b[3];
  int n = 0;
  sub3(dir, B->pos, A->pos); if (dot3(dir, dir) < 1e-24) { dir[0] = 1; dir[1] = dir[2] = 0; }
  cvx_support(A, dir, a); double nd[3] = { -dir[0], -dir[1], -dir[2] }; cvx_support(B, nd, b);
  sub3(S[0], a, b); n = 1; copy3(v, S[0]);
  for (int it = 0; it < 64; it++) {
    double vv = dot3(v, v);
    if (vv < 1e-24) return -(cvx_radius(A) + cvx_radius(B));                     /* cores touch */
    double mv[3] = { -v[0], -v[1], -v[2] }, wpt[3];
    cvx_support(A, mv, a); cvx_support(B, v, b); sub3(wpt, a, b);
    if (vv - dot3(v, wpt) <= 1e-12 * vv) break;                                   /* no progress possible: v is the closest point */
    int dup = 0;
    for (int i = 0; i < n; i++) { double df[3]; sub3(df, S[i], wpt); if (dot3(df, df) < 1e-28) dup = 1; }
    if (dup) break;
    copy3(S[n], wpt); n++;
    if (simplex_closest(S, &n, v)) return -(cvx_radius(A) + cvx_radius(B));      /* origin enclosed: the cores overlap */
  }
  return sqrt(dot3(v, v)) - cvx_radius(A) - cvx_radius(B);
}

/* test hook (tests/test_oracle_golden.py): GJK distance of two vertex clouds' hulls, world frame */
double lmo_test_hull_distance(const double* va, int na, const double* vb, int nb) {
  static const double zero[3] = {0, 0, 0}, eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, sz[3] = {0, 0, 0};
  double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
  for (int i = 0; i < na; i++) for (int k = 0; k < 3; k++) ca[k] += va[3*i + k] / na;
  for (int i = 0; i < nb; i++) for (int k = 0; k < 3; k++) cb[k] += vb[3*i + k] / nb;
  cvx A = { LM_GEOM_MESH, ca, eye, sz, va, na, zero, eye }, B = { LM_GEOM_MESH, cb, eye, sz, vb, nb, zero, eye };
  return convex_distance(&A, &B);
}


/* ------------------------------------------------------------------------------------------ */
/* convex-convex narrow phase: the engine's mjc_Convex = libccd's ccdMPRPenetration (Minkowski     */
/* Portal Refinement, G. Snethen; libccd src/mpr.c as vendored by MuJoCo 2.3.7) driven by          */
/* MuJoCo's support / centre callbacks (engine_collision_convex.c: mjccd_support, mjccd_center).   */
/* Restated from the published algorithm: the third-party sources are not under /root/reference.   */
/* Pinned by the reference's golden rollouts with bone-mesh contacts (HumanoidTorque.walk rows     */
/* 19-28, UnitreeH1.{walk,run}: tests/test_oracle_golden.py) — an iterative method, so to a stated */
/* tolerance (the engine's mpr_tolerance 1e-6, mpr_iterations 50), not to the last bit.            */
/*  - centre of a geom = its frame origin (mesh: the mesh's centre of mass, where the compiler     */
/*    puts the geom frame); support of shape A - B in direction d = sup_A(d) - sup_B(-d);          */
/*  - both shapes are inflated by margin/2 along the (unit) search direction, the contact distance */
/*    is margin - depth, the normal points from geom 1 to geom 2, the position is the mean of the  */
/*    two witness points (barycentric coordinates of the origin in the final portal).              */
/* ------------------------------------------------------------------------------------------ */
#define CCD_EPS 2.220446049250313e-16
#define MPR_TOLERANCE 1e-6
#define MPR_ITERATIONS 50
typedef struct { double v[3], v1[3], v2[3]; } mpr_sup;
typedef struct { const cvx* g; const double* center; double halfmargin; } mpr_obj;

static int ccd_is_zero(double x) { return fabs(x) < CCD_EPS; }
static int ccd_eq(double a_, double b_) {
  double ab = fabs(a_ - b_);
  if (ab < CCD_EPS) return 1;
  double a = fabs(a_), b = fabs(b_);
  return b > a ? ab < CCD_EPS * b : ab < CCD_EPS * a;
}
static int ccd_vec_eq(const double* a, const double* b) { return ccd_eq(a[0], b[0]) && ccd_eq(a[1], b[1]) && ccd_eq(a[2], b[2]); }
static void ccd_normalize(double* v) { double n = sqrt(dot3(v, v)); v[0] /= n; v[1] /= n; v[2] /= n; }   /* libccd: no zero guard */
static double sgn(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0); }

/* mjccd_support: support point of ONE geom in world direction dir (unit), in its local frame first */
static void mj_support(const mpr_obj* o, const double* dir, double* out) {
  const cvx* g = o->g;
  if (g->type == LM_GEOM_MESH) {
    double dl[3] = { g->bmat[0]*dir[0] + g->bmat[3]*dir[1] + g->bmat[6]*dir[2], g->bmat[1]*dir[0] + g->bmat[4]*dir[1] + g->bmat[7]*dir[2],
                     g->bmat[2]*dir[0] + g->bmat[5]*dir[1] + g->bmat[8]*dir[2] };
    int best = 0; double db = -1e10;
    for (int i = 0; i < g->nverts; i++) { double d = dot3(g->verts + 3*i, dl); if (d > db) { db = d; best = i; } }
    mulmat3(out, g->bmat, g->verts + 3*best); add3(out, out, g->bpos);
  } else {
    const double* R = g->R; const double* sz = g->size;
    double dl[3] = { R[0]*dir[0] + R[3]*dir[1] + R[6]*dir[2], R[1]*dir[0] + R[4]*dir[1] + R[7]*dir[2], R[2]*dir[0] + R[5]*dir[1] + R[8]*dir[2] };
    double res[3] = {0, 0, 0};
    switch (g->type) {
      case LM_GEOM_SPHERE: for (int k = 0; k < 3; k++) res[k] = dl[k] * sz[0]; break;
      case LM_GEOM_CAPSULE: for (int k = 0; k < 3; k++) res[k] = dl[k] * sz[0]; res[2] += sgn(dl[2]) * sz[1]; break;
      case LM_GEOM_CYLINDER: {
        double t = sqrt(dl[0]*dl[0] + dl[1]*dl[1]);
        if (t > MINVAL) { res[0] = dl[0] / t * sz[0]; res[1] = dl[1] / t * sz[0]; }
        res[2] = sgn(dl[2]) * sz[1];
        break;
      }
      case LM_GEOM_BOX: for (int k = 0; k < 3; k++) res[k] = sgn(dl[k]) * sz[k]; break;
      default: break;
    }
    mulmat3(out, R, res); add3(out, out, g->pos);
  }
  addscl3(out, dir, o->halfmargin);
}
static void mpr_support(const mpr_obj* A, const mpr_obj* B, const double* dir, mpr_sup* s) {
  double nd[3] = { -dir[0], -dir[1], -dir[2] };
  mj_support(A, dir, s->v1); mj_support(B, nd, s->v2); sub3(s->v, s->v1, s->v2);
}
static void mpr_portal_dir(const mpr_sup* P, double* dir) {
  double a[3], b[3]; sub3(a, P[2].v, P[1].v); sub3(b, P[3].v, P[1].v); cross3(dir, a, b); ccd_normalize(dir);
}
static int mpr_reach_tolerance(const mpr_sup* P, const mpr_sup* v4, const double* dir) {
  double dv1 = dot3(P[1].v, dir), dv2 = dot3(P[2].v, dir), dv3 = dot3(P[3].v, dir), dv4 = dot3(v4->v, dir);
  double d = fmin(fmin(dv4 - dv1, dv4 - dv2), dv4 - dv3);
  return ccd_eq(d, MPR_TOLERANCE) || d < MPR_TOLERANCE;
}
static void mpr_expand_portal(mpr_sup* P, const mpr_sup* v4) {
  double c[3]; cross3(c, v4->v, P[0].v);
  if (dot3(P[1].v, c) > 0) { if (dot3(P[2].v, c) > 0) P[1] = *v4; else P[3] = *v4; }
  else { if (dot3(P[3].v, c) > 0) P[2] = *v4; else P[1] = *v4; }
}
static double point_segment_dist2(const double* P, const double* x0, const double* b, double* witness) {
  double d[3], a[3]; sub3(d, b, x0); sub3(a, x0, P);
  double t = -dot3(a, d) / dot3(d, d), dist;
  if (t < 0 || ccd_is_zero(t)) { dist = dot3(a, a); if (witness) copy3(witness, x0); }            /* |x0 - P|^2 */
  else if (t > 1 || ccd_eq(t, 1)) { double e[3]; sub3(e, b, P); dist = dot3(e, e); if (witness) copy3(witness, b); }
  else {
    if (witness) { copy3(witness, d); for (int k = 0; k < 3; k++) witness[k] = witness[k] * t + x0[k]; double e[3]; sub3(e, witness, P); dist = dot3(e, e); }
    else { double e[3] = { a[0] + t*d[0], a[1] + t*d[1], a[2] + t*d[2] }; dist = dot3(e, e); }
  }
  return dist;
}
static double point_tri_dist2(const double* P, const double* x0, const double* B, const double* Cc, double* witness) {
  double d1[3], d2[3], a[3]; sub3(d1, B, x0); sub3(d2, Cc, x0); sub3(a, x0, P);
  double v = dot3(d1, d1), w = dot3(d2, d2), p = dot3(a, d1), q = dot3(a, d2), r = dot3(d1, d2);
  double d = w * v - r * r, s, t, dist;
  if (ccd_is_zero(d)) s = t = -1; else { s = (q * r - w * p) / d; t = (-s * r - q) / w; }
  if ((ccd_is_zero(s) || s > 0) && (ccd_eq(s, 1) || s < 1) && (ccd_is_zero(t) || t > 0) && (ccd_eq(t, 1) || t < 1) && (ccd_eq(t + s, 1) || t + s < 1)) {
    for (int k = 0; k < 3; k++) witness[k] = x0[k] + d1[k] * s + d2[k] * t;
    double e[3]; sub3(e, witness, P); dist = dot3(e, e);
  } else {
    double w2[3], d2_;
    dist = point_segment_dist2(P, x0, B, witness);
    d2_ = point_segment_dist2(P, x0, Cc, w2); if (d2_ < dist) { dist = d2_; copy3(witness, w2); }
    d2_ = point_segment_dist2(P, B, Cc, w2); if (d2_ < dist) { dist = d2_; copy3(witness, w2); }
  }
  return dist;
}
/* barycentric position of the origin in the portal tetrahedron -> mean of the witness points on the two shapes */
static void mpr_find_pos(const mpr_sup* P, double* pos) {
  double dir[3], vec[3], b[4], sum;
  mpr_portal_dir(P, dir);
  cross3(vec, P[1].v, P[2].v); b[0] = dot3(vec, P[3].v);
  cross3(vec, P[3].v, P[2].v); b[1] = dot3(vec, P[0].v);
  cross3(vec, P[0].v, P[1].v); b[2] = dot3(vec, P[3].v);
  cross3(vec, P[2].v, P[1].v); b[3] = dot3(vec, P[0].v);
  sum = b[0] + b[1] + b[2] + b[3];
  if (ccd_is_zero(sum) || sum < 0) {
    b[0] = 0;
    cross3(vec, P[2].v, P[3].v); b[1] = dot3(vec, dir);
    cross3(vec, P[3].v, P[1].v); b[2] = dot3(vec, dir);
    cross3(vec, P[1].v, P[2].v); b[3] = dot3(vec, dir);
    sum = b[1] + b[2] + b[3];
  }
  double inv = 1.0 / sum, p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0};
  for (int i = 0; i < 4; i++) { addscl3(p1, P[i].v1, b[i]); addscl3(p2, P[i].v2, b[i]); }
  for (int k = 0; k < 3; k++) pos[k] = 0.5 * (p1[k] * inv + p2[k] * inv);
}
/* returns 0 and (depth, dir, pos) when the (inflated) shapes overlap, -1 otherwise */
static int mpr_penetration(const mpr_obj* A, const mpr_obj* B, double* depth, double* pdir, double* pos) {
  static const double origin[3] = {0, 0, 0};
  mpr_sup P[4], v4; double dir[3], va[3], vb[3], dot; int res = 0;
  /* ---- portal discovery */
  copy3(P[0].v1, A->center); copy3(P[0].v2, B->center); sub3(P[0].v, P[0].v1, P[0].v2);
  if (ccd_vec_eq(P[0].v, origin)) P[0].v[0] += CCD_EPS * 10.0;
  for (int k = 0; k < 3; k++) dir[k] = -P[0].v[k];
  ccd_normalize(dir);
  mpr_support(A, B, dir, &P[1]);
  dot = dot3(P[1].v, dir);
  if (ccd_is_zero(dot) || dot < 0) return -1;
  cross3(dir, P[0].v, P[1].v);
  if (ccd_is_zero(dot3(dir, dir))) res = ccd_vec_eq(P[1].v, origin) ? 1 : 2;
  else {
    ccd_normalize(dir);
    mpr_support(A, B, dir, &P[2]);
    dot = dot3(P[2].v, dir);
    if (ccd_is_zero(dot) || dot < 0) return -1;
    sub3(va, P[1].v, P[0].v); sub3(vb, P[2].v, P[0].v); cross3(dir, va, vb); ccd_normalize(dir);
    if (dot3(dir, P[0].v) > 0) { mpr_sup t = P[1]; P[1] = P[2]; P[2] = t; for (int k = 0; k < 3; k++) dir[k] = -dir[k]; }
    int size = 3;
    while (size < 4) {
      mpr_support(A, B, dir, &P[3]);
      dot = dot3(P[3].v, dir);
      if (ccd_is_zero(dot) || dot < 0) return -1;
      int cont = 0;
      cross3(va, P[1].v, P[3].v); dot = dot3(va, P[0].v);
      if (dot < 0 && !ccd_is_zero(dot)) { P[2] = P[3]; cont = 1; }
      if (!cont) {
        cross3(va, P[3].v, P[2].v); dot = dot3(va, P[0].v);
        if (dot < 0 && !ccd_is_zero(dot)) { P[1] = P[3]; cont = 1; }
      }
      if (cont) { sub3(va, P[1].v, P[0].v); sub3(vb, P[2].v, P[0].v); cross3(dir, va, vb); ccd_normalize(dir); }
      else size = 4;
    }
  }
  if (res == 1) {                                        /* touching contact on v1 */
    *depth = 0; pdir[0] = pdir[1] = pdir[2] = 0;
    for (int k = 0; k < 3; k++) pos[k] = 0.5 * (P[1].v1[k] + P[1].v2[k]);
    return 0;
  }
  if (res == 2) {                                        /* origin on the segment v0-v1: depth = |v1|, direction = v1 */
    for (int k = 0; k < 3; k++) pos[k] = 0.5 * (P[1].v1[k] + P[1].v2[k]);
    copy3(pdir, P[1].v); *depth = sqrt(dot3(pdir, pdir)); ccd_normalize(pdir);
    return 0;
  }
  if (getenv("LMO_MPR_TRACE")) fprintf(stderr, " o discovered: v0 %.6f %.6f %.6f | v1 %.6f %.6f %.6f | v2 %.6f %.6f %.6f | v3 %.6f %.6f %.6f\n", P[0].v[0],P[0].v[1],P[0].v[2],P[1].v[0],P[1].v[1],P[1].v[2],P[2].v[0],P[2].v[1],P[2].v[2],P[3].v[0],P[3].v[1],P[3].v[2]);
  /* ---- portal refinement: until the portal's outward side holds the origin */
  for (;;) {
    mpr_portal_dir(P, dir);
    dot = dot3(dir, P[1].v);
    if (ccd_is_zero(dot) || dot > 0) break;              /* the portal encapsulates the origin */
    mpr_support(A, B, dir, &v4);
    dot = dot3(v4.v, dir);
    if (!(ccd_is_zero(dot) || dot > 0) || mpr_reach_tolerance(P, &v4, dir)) return -1;
    mpr_expand_portal(P, &v4);
  }
  /* ---- penetration: push the portal to the surface of the Minkowski difference */
  for (unsigned long it = 0;; it++) {
    mpr_portal_dir(P, dir);
    mpr_support(A, B, dir, &v4);
    if (getenv("LMO_MPR_TRACE")) fprintf(stderr, "  o pen it %lu dir %.6f %.6f %.6f v4 %.6f %.6f %.6f dv4 %.8f dv1 %.8f\n", it, dir[0], dir[1], dir[2], v4.v[0], v4.v[1], v4.v[2], dot3(v4.v, dir), dot3(P[1].v, dir));
    if (mpr_reach_tolerance(P, &v4, dir) || it > MPR_ITERATIONS) {
      *depth = sqrt(point_tri_dist2(origin, P[1].v, P[2].v, P[3].v, pdir));
      if (ccd_is_zero(pdir[0]) && ccd_is_zero(pdir[1]) && ccd_is_zero(pdir[2])) copy3(pdir, dir);
      ccd_normalize(pdir);
      mpr_find_pos(P, pos);
      return 0;
    }
    mpr_expand_portal(P, &v4);
  }
}

/* pairs the engine's collision table routes to mjc_Convex (types in the table's order: sphere < capsule < cylinder < box < mesh) */
static int engine_uses_ccd(int t1, int t2) {
  if (t2 == LM_GEOM_MESH) return t1 != LM_GEOM_PLANE;
  if (t1 == LM_GEOM_CAPSULE && t2 == LM_GEOM_CYLINDER) return 1;
  if (t1 == LM_GEOM_CYLINDER && (t2 == LM_GEOM_CYLINDER || t2 == LM_GEOM_BOX)) return 1;
  return 0;
}
/* mjc_fixNormal: a sphere or a capsule in the pair replaces the MPR direction by the direction from its centre / axis to the
   contact point (both round: the normalised difference of the two) */
static void fix_normal(const cvx* A, const cvx* B, const double* pos, double* normal) {
  double n[2][3]; int done[2] = {0, 0};
  for (int i = 0; i < 2; i++) {
    const cvx* g = i ? B : A;
    if (g->type == LM_GEOM_SPHERE) { sub3(n[i], pos, g->pos); done[i] = 1; }
    else if (g->type == LM_GEOM_CAPSULE) {
      double ax[3] = { g->R[2], g->R[5], g->R[8] }, rel[3]; sub3(rel, pos, g->pos);
      double t = dot3(rel, ax); if (t < -g->size[1]) t = -g->size[1]; else if (t > g->size[1]) t = g->size[1];
      for (int k = 0; k < 3; k++) n[i][k] = rel[k] - t * ax[k];
      done[i] = 1;
    }
    if (done[i]) normalize3(n[i]);
  }
  if (done[0] && done[1]) { sub3(normal, n[0], n[1]); normalize3(normal); }
  else if (done[0]) copy3(normal, n[0]);
  else if (done[1]) { for (int k = 0; k < 3; k++) normal[k] = -n[1][k]; }
}

static void collide(const lmo_model* m, work* w) {
  w->ncon = 0; w->unhandled_pairs = 0; w->convex_contacts = 0; w->max_self_depth = 0; w->native_contacts = 0; w->own_contacts = 0; w->own_face_contacts = 0;
  for (int pi = 0; pi < m->npair; pi++) {
    int g1 = m->pair_g1[pi], g2 = m->pair_g2[pi];
    int t1 = IDX(m->geom_type, g1), t2 = IDX(m->geom_type, g2);
    const double *p1 = w->gpos[g1], *R1 = w->gmat[g1], *s1 = m->geom_size + 3*g1;
    const double *p2 = w->gpos[g2], *R2 = w->gmat[g2], *s2 = m->geom_size + 3*g2;
    double rb2 = rbound(t2, s2);
    /* margin-less bounding-sphere prune first (see below): most pairs end here, before their contact parameters are mixed */
    {
      double rel[3]; sub3(rel, p2, p1);
      if (t1 == LM_GEOM_PLANE) { double n[3] = { R1[2], R1[5], R1[8] }; if (dot3(rel, n) - rb2 > 0) continue; }
      else {
        if (m->disable_self_collision) continue;
        double rr = rbound(t1, s1) + rb2;
        if (dot3(rel, rel) > rr * rr) continue;
      }
    }
    lmo_contact tm; memset(&tm, 0, sizeof(tm));
    contact_params(m, g1, g2, &tm);
    double margin = tm.margin;
    if (t1 == LM_GEOM_PLANE) {
      double n[3] = { R1[2], R1[5], R1[8] };           /* plane normal = its z axis */
      double rel[3];
      /* bounding-sphere prune WITHOUT the contact margin. Pinned empirically: golden rows 0,1,3,8 of
         UnitreeA1.simple are only reproduced (to 1e-12) if a foot sphere at 0 < dist < margin yields no
         contact while tilted capsules at 0 < dist < margin do (their bounding sphere still crosses the
         plane) — i.e. MuJoCo 2.3.7 prunes on the margin-less bounding sphere before the narrow phase. */
      sub3(rel, p2, p1);
      if (dot3(rel, n) - rb2 > 0) continue;
      if (t2 == LM_GEOM_SPHERE) {
        sub3(rel, p2, p1);
        double dist = dot3(rel, n) - s2[0];
        if (dist < margin) {
          double pos[3]; copy3(pos, p2); addscl3(pos, n, -(s2[0] + 0.5 * dist));
          add_contact(w, &tm, dist, pos, n, NULL);
        }
      } else if (t2 == LM_GEOM_CAPSULE) {
        double ax[3] = { R2[2], R2[5], R2[8] };
        for (int sgn = 1; sgn >= -1; sgn -= 2) {
          double c[3]; copy3(c, p2); addscl3(c, ax, sgn * s2[1]);
          sub3(rel, c, p1);
          double dist = dot3(rel, n) - s2[0];
          if (dist < margin) {
            double pos[3]; copy3(pos, c); addscl3(pos, n, -(s2[0] + 0.5 * dist));
            add_contact(w, &tm, dist, pos, n, ax);
          }
        }
      } else if (t2 == LM_GEOM_BOX) {
        /* corners in bit order (x: bit 0, y: bit 1, z: bit 2); only corners on the plane side of the box
           centre; at most the first 4 found */
        sub3(rel, p2, p1);
        double dist0 = dot3(rel, n);
        int cnt = 0;
        for (int k = 0; k < 8 && cnt < 4; k++) {
          double loc[3] = { (k & 1 ? s2[0] : -s2[0]), (k & 2 ? s2[1] : -s2[1]), (k & 4 ? s2[2] : -s2[2]) };
          double c[3]; mulmat3(c, R2, loc);
          double ldist = dot3(n, c);
          if (dist0 + ldist > margin || ldist > 0) continue;
          double dist = dist0 + ldist;
          double pos[3]; add3(pos, c, p2); addscl3(pos, n, -0.5 * dist);
          add_contact(w, &tm, dist, pos, n, NULL);
          cnt++;
        }
      } else if (t2 == LM_GEOM_CYLINDER) {
        /* disk-edge construction: deepest rim point of each cap + two more points on the near cap */
        double ax[3] = { R2[2], R2[5], R2[8] };
        sub3(rel, p2, p1);
        double dist0 = dot3(rel, n);
        double prjaxis = dot3(n, ax);
        if (prjaxis > 0) { ax[0] = -ax[0]; ax[1] = -ax[1]; ax[2] = -ax[2]; prjaxis = -prjaxis; }
        double vec[3] = { ax[0] * prjaxis - n[0], ax[1] * prjaxis - n[1], ax[2] * prjaxis - n[2] };
        double len = norm3(vec);
        if (len < 1e-12) { vec[0] = R2[0] * s2[0]; vec[1] = R2[3] * s2[0]; vec[2] = R2[6] * s2[0]; }
        else { for (int k = 0; k < 3; k++) vec[k] *= s2[0] / len; }
        double prjvec = dot3(vec, n);
        double axs[3] = { ax[0] * s2[1], ax[1] * s2[1], ax[2] * s2[1] };
        prjaxis *= s2[1];
        double d1 = dist0 + prjaxis + prjvec;
        if (d1 < margin) {
          double pos[3]; add3(pos, p2, vec); add3(pos, pos, axs); addscl3(pos, n, -0.5 * d1);
          add_contact(w, &tm, d1, pos, n, NULL);
          double d2 = dist0 - prjaxis + prjvec;
          if (d2 < margin) {
            double q[3]; add3(q, p2, vec); sub3(q, q, axs); addscl3(q, n, -0.5 * d2);
            add_contact(w, &tm, d2, q, n, NULL);
          }
          double d3 = dist0 + prjaxis - 0.5 * prjvec;
          if (d3 < margin) {
            double v1[3]; cross3(v1, vec, ax);
            normalize3(v1);
            for (int k = 0; k < 3; k++) v1[k] *= s2[0] * sqrt(3.0) * 0.5;
            for (int sgn = 1; sgn >= -1; sgn -= 2) {
              double q[3]; add3(q, p2, axs); addscl3(q, vec, -0.5); addscl3(q, v1, sgn); addscl3(q, n, -0.5 * d3);
              add_contact(w, &tm, d3, q, n, NULL);
            }
          }
        }
      } else if (t2 == LM_GEOM_MESH && m->mesh_nvert[g2] > 0) {
        /* plane vs convex hull (mjc_PlaneConvex): ONE contact, at the hull's support vertex in the direction of -normal.
           Pinned by the UnitreeH1 golden rows: ten rows with a foot on the ground — flat on it, more than 100 hull vertices
           below the plane, or on an edge — are reproduced to 1e-7 with exactly this, and with nothing that adds neighbouring
           or further penetrating vertices (tests/test_oracle_golden.py). */
        const int b2 = IDX(m->geom_body, g2), nvm = m->mesh_nvert[g2];
        const double* V = m->mesh_vert[g2];
        int best = -1; double dbest = 1e300;
        for (int i = 0; i < nvm; i++) {
          double wv[3]; mulmat3(wv, w->xmat[b2], V + 3*i); add3(wv, wv, w->xpos[b2]); sub3(wv, wv, p1);
          double d = dot3(wv, n);
          if (d < dbest) { dbest = d; best = i; }
        }
        if (dbest < margin) {
          double wv[3], pos[3];
          mulmat3(wv, w->xmat[b2], V + 3*best); add3(wv, wv, w->xpos[b2]);
          copy3(pos, wv); addscl3(pos, n, -0.5 * dbest);
          add_contact(w, &tm, dbest, pos, n, NULL);
          if (m->mesh_nbr_adr[g2]) {
            /* further contacts at the hull-graph neighbours of the support vertex (the engine's "up to 3 more contacts from
               mesh"): penetrating, nearest first, none closer than tol to the SUPPORT contact. Reverse-engineered on the
               UnitreeH1 golden rows (profiles/r2_ab_probes.md §9, profiles/r3_notes.md §2): walk row 23 is reproduced to 1e-7 only
               with the toe vertex + two heel vertices 11 mm apart from each other (170 mm from the toe) while toe-side
               neighbours 23 mm away stay out - so the tolerance is measured against the first contact, not between the
               extra ones. Against the round-2 rule (tolerance against every contact found): +2 rows reproduced, none lost */
            double cp[4][3]; int nc = 1; copy3(cp[0], pos);
            for (int e = m->mesh_nbr_adr[g2][best]; e < m->mesh_nbr_adr[g2][best + 1] && nc < 4; e++) {
              const int j = m->mesh_nbr[g2][e];
              double wj[3], rj[3]; mulmat3(wj, w->xmat[b2], V + 3*j); add3(wj, wj, w->xpos[b2]); sub3(rj, wj, p1);
              const double dj = dot3(rj, n);
              if (dj > margin) continue;
              double pj[3]; copy3(pj, wj); addscl3(pj, n, -0.5 * dj);
              int close = 0;
              { double df[3]; sub3(df, pj, cp[0]); if (norm3(df) < m->mesh_tol[g2]) close = 1; }
              if (close) continue;
              add_contact(w, &tm, dj, pj, n, NULL); copy3(cp[nc], pj); nc++;
            }
          }
        }
      } else if (t2 == LM_GEOM_MESH) {
        /* no convex-hull collider (not restated): count the bounding capsule coming within reach */
        double ax[3] = { R2[2], R2[5], R2[8] };
        sub3(rel, p2, p1);
        if (dot3(rel, n) - s2[1] * fabs(dot3(ax, n)) - s2[0] < margin) { w->unhandled_pairs++; if (getenv("LMO_DEBUG")) fprintf(stderr, "plane-mesh-nohull %d\n", g2); }
      } else { w->unhandled_pairs++; if (getenv("LMO_DEBUG")) fprintf(stderr, "plane-other %d type %d\n", g2, t2); }
      continue;
    }
    if (m->disable_self_collision) continue;
    {
      /* same margin-less bounding-sphere prune for non-plane pairs (unpinned: no golden row has one) */
      double rel[3]; sub3(rel, p2, p1);
      if (norm3(rel) - rbound(t1, s1) - rb2 > 0) continue;
    }
    if (t1 == LM_GEOM_SPHERE && t2 == LM_GEOM_SPHERE) {
      sphere_sphere(w, &tm, p1, s1[0], p2, s2[0]);
    } else if (t1 == LM_GEOM_SPHERE && t2 == LM_GEOM_CAPSULE) {
      double ax[3] = { R2[2], R2[5], R2[8] }, rel[3]; sub3(rel, p1, p2);
      double t = dot3(rel, ax); if (t < -s2[1]) t = -s2[1]; else if (t > s2[1]) t = s2[1];
      double c[3]; copy3(c, p2); addscl3(c, ax, t);
      sphere_sphere(w, &tm, p1, s1[0], c, s2[0]);
    } else if (t1 == LM_GEOM_CAPSULE && t2 == LM_GEOM_CAPSULE) {
      double a1[3] = { R1[2], R1[5], R1[8] }, a2[3] = { R2[2], R2[5], R2[8] }, s, t;
      segment_closest(p1, a1, s1[1], p2, a2, s2[1], &s, &t);
      double c1[3], c2[3]; copy3(c1, p1); addscl3(c1, a1, s); copy3(c2, p2); addscl3(c2, a2, t);
      sphere_sphere(w, &tm, c1, s1[0], c2, s2[0]);
    } else if (!m->disable_ccd && engine_uses_ccd(t1, t2) && (t1 != LM_GEOM_MESH || m->mesh_nvert[g1] > 0) && (t2 != LM_GEOM_MESH || m->mesh_nvert[g2] > 0)) {
      /* mjc_Convex: one contact from MPR on the two shapes inflated by margin / 2 each */
      const int b1 = IDX(m->geom_body, g1), b2 = IDX(m->geom_body, g2);
      cvx A = { t1, p1, R1, s1, m->mesh_vert[g1], m->mesh_nvert[g1], w->xpos[b1], w->xmat[b1] };
      cvx B = { t2, p2, R2, s2, m->mesh_vert[g2], m->mesh_nvert[g2], w->xpos[b2], w->xmat[b2] };
      mpr_obj oa = { &A, w->gcen[g1], 0.5 * margin }, ob = { &B, w->gcen[g2], 0.5 * margin };
      double depth, dir[3], pos[3];
      if (mpr_penetration(&oa, &ob, &depth, dir, pos) == 0 && !(dir[0] == 0 && dir[1] == 0 && dir[2] == 0)) {
        fix_normal(&A, &B, pos, dir);
        add_contact(w, &tm, margin - depth, pos, dir, NULL);
        w->convex_contacts++;
      }
    } else if (!m->disable_native && ((t1 == LM_GEOM_SPHERE && (t2 == LM_GEOM_BOX || t2 == LM_GEOM_CYLINDER)) || (t1 == LM_GEOM_CAPSULE && t2 == LM_GEOM_BOX)
                                      || (t1 == LM_GEOM_BOX && t2 == LM_GEOM_BOX))) {
      /* the engine's native box / cylinder colliders (restated above) */
      double buf[8][7];
      const int nc = native_pair(t1, p1, R1, s1, t2, p2, R2, s2, margin, buf);
      for (int i = 0; i < nc; i++) add_contact(w, &tm, buf[i][0], buf[i] + 1, buf[i] + 4, NULL);
      w->native_contacts += nc > 0 ? nc : 0;
      /* of those: contacts whose manifold is this repository's own construction, not the engine's case analysis (capsule-box, box-box),
         and among them the box-box FACE case (the approximate one) */
      if (nc > 0 && t2 == LM_GEOM_BOX && (t1 == LM_GEOM_BOX || t1 == LM_GEOM_CAPSULE)) {
        w->own_contacts += nc;
        if (t1 == LM_GEOM_BOX && g_boxbox_face_case) w->own_face_contacts += nc;
      }
    } else if (m->skip_pair_counter) {
      /* timing runs: the pair has no collider here, and whether the engine would have a contact is not asked */
    } else if (t1 == LM_GEOM_MESH || t2 == LM_GEOM_MESH
               || ((t1 == LM_GEOM_CYLINDER || t2 == LM_GEOM_CYLINDER || t1 == LM_GEOM_BOX || t2 == LM_GEOM_BOX) && !(t1 == LM_GEOM_BOX && t2 == LM_GEOM_BOX))) {
      /* a pair the engine collides through libccd or a native box collider, not restated: COUNTED when the exact distance of
         the two convex shapes (hull vertices for meshes; GJK) is below the contact margin — the engine would have a contact
         here. A mesh without hull vertices falls back to its bounding capsule. */
      const int b1 = IDX(m->geom_body, g1), b2 = IDX(m->geom_body, g2);
      double c1s[3] = { s1[0], s1[1], s1[2] }, c2s[3] = { s2[0], s2[1], s2[2] };
      cvx A = { t1, p1, R1, c1s, m->mesh_vert[g1], m->mesh_nvert[g1], w->xpos[b1], w->xmat[b1] };
      cvx B = { t2, p2, R2, c2s, m->mesh_vert[g2], m->mesh_nvert[g2], w->xpos[b2], w->xmat[b2] };
      if (t1 == LM_GEOM_MESH && m->mesh_nvert[g1] == 0) A.type = LM_GEOM_CAPSULE;      /* bounding capsule (r, h) in size[0], size[1] */
      if (t2 == LM_GEOM_MESH && m->mesh_nvert[g2] == 0) B.type = LM_GEOM_CAPSULE;
      if (convex_distance(&A, &B) < margin + 1e-9) { w->unhandled_pairs++; if (getenv("LMO_DEBUG")) fprintf(stderr, "cvx %d %d types %d %d\n", g1, g2, t1, t2); }
    } else if (t1 == LM_GEOM_BOX && t2 == LM_GEOM_BOX) {
      /* box-box contacts are not restated; the pair is only COUNTED, and only when no separating axis
         (3 + 3 face normals, 9 edge cross products) keeps the boxes more than `margin` apart */
      if (box_box_gap(p1, R1, s1, p2, R2, s2) < margin) { w->unhandled_pairs++; if (getenv("LMO_DEBUG")) fprintf(stderr, "boxbox %d %d\n", g1, g2); }
    } else {
      /* box / cylinder vs other non-plane geoms (the engine: native capsule-box / sphere-box colliders, libccd for
         everything with a cylinder): not restated. The pair is COUNTED when the geoms' bounding capsules (cylinder (r, h) ->
         capsule (r, h); box -> capsule along its longest edge, radius = half diagonal of the cross-section) come within
         the margin — conservative: no count means the engine has no contact there either, so a test that sees a zero
         count compares against complete physics. */
      double ca[3], aa[3], ha, ra, cb[3], ab[3], hb, rb, sa, ta;
      bounding_capsule(t1, s1, p1, R1, ca, aa, &ha, &ra);
      bounding_capsule(t2, s2, p2, R2, cb, ab, &hb, &rb);
      segment_closest(ca, aa, ha, cb, ab, hb, &sa, &ta);
      double c1[3], c2[3], dd[3]; copy3(c1, ca); addscl3(c1, aa, sa); copy3(c2, cb); addscl3(c2, ab, ta);
      sub3(dd, c2, c1);
      if (norm3(dd) - ra - rb < margin) { w->unhandled_pairs++; if (getenv("LMO_DEBUG")) fprintf(stderr, "capsule-bound %d %d types %d %d\n", g1, g2, t1, t2); }
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* constraint assembly                                                                         */
/* ------------------------------------------------------------------------------------------ */
static double impedance(const double* solimp_in, double pos, double margin) {
  double si[5];
  si[0] = fmin(MAXIMP, fmax(MINIMP, solimp_in[0])); si[1] = fmin(MAXIMP, fmax(MINIMP, solimp_in[1]));
  si[2] = fmax(0.0, solimp_in[2]); si[3] = fmin(MAXIMP, fmax(MINIMP, solimp_in[3])); si[4] = fmax(1.0, solimp_in[4]);
  if (si[0] == si[1] || si[2] <= MINVAL) return 0.5 * (si[0] + si[1]);
  double x = (pos - margin) / si[2];
  if (x < 0) x = -x;
  if (x >= 1) return si[1];
  if (x <= 0) return si[0];
  double y;
  if (si[4] == 1) y = x;
  else if (x <= si[3]) y = pow(x, si[4]) / pow(si[3], si[4] - 1);
  else y = 1 - pow(1 - x, si[4]) / pow(1 - si[3], si[4] - 1);
  return si[0] + y * (si[1] - si[0]);
}

static void add_row(const lmo_model* m, work* w, int type, int id, const double* jrow, double pos, double margin,
                    double floss, double diagApprox, const double* solref, const double* solimp, int is_friction_row) {
  int i = w->nefc++;
  int nv = m->nv;
  memcpy(w->J + i * nv, jrow, sizeof(double) * nv);
  w->type[i] = type; w->id[i] = id; w->pos[i] = pos; w->margin[i] = margin; w->floss[i] = floss;
  w->diagApprox[i] = diagApprox;
  double imp = impedance(solimp, pos, margin);
  w->imp[i] = imp;
  w->R[i] = fmax(MINVAL, (1 - imp) * diagApprox / imp);
  {
    double dmax = fmin(MAXIMP, fmax(MINIMP, solimp[1]));
    if (solref[0] > 0) {
      double tc = fmax(solref[0], 2 * m->timestep), dr = solref[1];
      w->K[i] = 1 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr);
      w->B[i] = 2 / fmax(MINVAL, dmax * tc);
    } else { w->K[i] = -solref[0] / fmax(MINVAL, dmax * dmax); w->B[i] = -solref[1] / fmax(MINVAL, dmax); }
    /* friction-loss rows and the friction dimensions of elliptic contacts have no position term */
    if (is_friction_row) w->K[i] = 0;
  }
}

static void make_constraints(const lmo_model* m, const double* qpos, work* w) {
  int nv = m->nv;
  double row[LMO_MAXV];
  w->nefc = 0;
  /* friction loss */
  for (int d = 0; d < nv; d++) {
    if (m->dof_frictionloss[d] <= 0) continue;
    memset(row, 0, sizeof(double) * nv); row[d] = 1;
    add_row(m, w, ROW_FRICTION, d, row, 0, 0, m->dof_frictionloss[d], m->dof_invweight0[d], m->dof_solref + 2*d,
            m->dof_solimp + 5*d, 1);
  }
  /* joint limits */
  for (int j = 0; j < nv; j++) {
    if (!IDX(m->jnt_limited, j)) continue;
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->jnt_range[2*j + (side + 1) / 2] - qpos[j]);
      if (dist < m->jnt_margin[j]) {
        memset(row, 0, sizeof(double) * nv); row[j] = -side;
        add_row(m, w, ROW_LIMIT, j, row, dist, m->jnt_margin[j], 0, m->dof_invweight0[j], m->jnt_solref + 2*j,
                m->jnt_solimp + 5*j, 0);
      }
    }
  }
  /* contacts */
  double jp1[3 * LMO_MAXV], jr1[3 * LMO_MAXV], jp2[3 * LMO_MAXV], jr2[3 * LMO_MAXV], jc[6 * LMO_MAXV];
  for (int ci = 0; ci < w->ncon; ci++) {
    lmo_contact* c = &w->con[ci];
    c->efc_address = -1;
    if (c->dist >= c->includemargin) continue;     /* inside the gap: detected but not solved */
    int b1 = IDX(m->geom_body, c->geom1), b2 = IDX(m->geom_body, c->geom2);
    jac_point(m, w, b1, c->pos, jp1, jr1);
    jac_point(m, w, b2, c->pos, jp2, jr2);
    /* rotate the Jacobian difference into the contact frame: rows 0..2 translational, 3..5 rotational */
    for (int r = 0; r < 3; r++) for (int d = 0; d < nv; d++) {
      double st = 0, sr = 0;
      for (int a = 0; a < 3; a++) {
        st += c->frame[3*r + a] * (jp2[a*nv + d] - jp1[a*nv + d]);
        sr += c->frame[3*r + a] * (jr2[a*nv + d] - jr1[a*nv + d]);
      }
      jc[r*nv + d] = st; jc[(3 + r)*nv + d] = sr;
    }
    double tran = m->body_invweight0[2*b1] + m->body_invweight0[2*b2];
    double rot = m->body_invweight0[2*b1 + 1] + m->body_invweight0[2*b2 + 1];
    int dim = c->dim;
    c->efc_address = w->nefc;
    if (dim == 1) {
      add_row(m, w, ROW_CONTACT_PLAIN, ci, jc, c->dist, c->includemargin, 0, tran, c->solref, c->solimp, 0);
    } else if (m->cone == LM_CONE_PYRAMIDAL) {
      int first = w->nefc;
      for (int k = 1; k < dim; k++) for (int sgn = 1; sgn >= -1; sgn -= 2) {
        double fk = c->friction[k - 1];
        for (int d = 0; d < nv; d++) row[d] = jc[d] + sgn * fk * jc[k*nv + d];
        double dA = tran + fk * fk * (k < 3 ? tran : rot);
        add_row(m, w, ROW_CONTACT_PYR, ci, row, c->dist, c->includemargin, 0, dA, c->solref, c->solimp, 0);
      }
      /* all edges of the pyramid share one regulariser, Rpy = 2 mu^2 R(first edge). Pinned (for mu = 1, condim 3)
         by the Atlas golden rollout: every contact-loaded row matches to 1e-16 with the factor, to ~1e-4 without. */
      c->mu = c->friction[0];
      double Rpy = 2 * c->mu * c->mu * w->R[first];
      for (int i = first; i < w->nefc; i++) w->R[i] = Rpy;
    } else {
      int first = w->nefc;
      for (int k = 0; k < dim; k++)
        add_row(m, w, ROW_CONTACT_ELL, ci, jc + k*nv, k == 0 ? c->dist : 0, k == 0 ? c->includemargin : 0, 0,
                k < 3 ? tran : rot, c->solref, c->solimp, k > 0);
      /* friction-row regularisation: R_1 = R_0/impratio, R_j * mu_j^2 constant; cone mu of the
         regularised cone */
      w->R[first + 1] = w->R[first] / fmax(MINVAL, m->impratio);
      c->mu = c->friction[0] * sqrt(w->R[first + 1] / w->R[first]);
      for (int k = 1; k < dim - 1; k++)
        w->R[first + k + 1] = w->R[first + 1] * c->friction[0] * c->friction[0] / (c->friction[k] * c->friction[k]);
    }
  }
  for (int i = 0; i < w->nefc; i++) w->D[i] = 1 / w->R[i];
}

/* ------------------------------------------------------------------------------------------ */
/* velocity stage: bias forces by a world-frame Newton-Euler recursion                         */
/* ------------------------------------------------------------------------------------------ */
static void rne_bias(const lmo_model* m, const double* qvel, work* w) {
  int nb = m->nbody, nv = m->nv;
  static const double zero3[3] = {0, 0, 0};
  double om[LMO_MAXBODY][3], al[LMO_MAXBODY][3], vc[LMO_MAXBODY][3], ac[LMO_MAXBODY][3];
  memcpy(om[0], zero3, sizeof(zero3)); memcpy(al[0], zero3, sizeof(zero3)); memcpy(vc[0], zero3, sizeof(zero3));
  ac[0][0] = -m->gravity[0]; ac[0][1] = -m->gravity[1]; ac[0][2] = -m->gravity[2];
  for (int i = 1; i < nb; i++) {
    int p = IDX(m->body_parent, i);
    double o[3], a[3], v[3], acc[3], c[3], r[3], t[3], t2[3];
    copy3(o, om[p]); copy3(a, al[p]); copy3(v, vc[p]); copy3(acc, ac[p]); copy3(c, w->xipos[p]);
#define TRANSPORT(target) do { sub3(r, (target), c); cross3(t, o, r); add3(v, v, t); cross3(t2, o, t); add3(acc, acc, t2); \
                               cross3(t, a, r); add3(acc, acc, t); copy3(c, (target)); } while (0)
    for (int k = 0; k < IDX(m->body_jntnum, i); k++) {
      int j = IDX(m->body_jntadr, i) + k;
      const double* u = w->xaxis[j];
      if (IDX(m->jnt_type, j) == LM_JNT_HINGE) {
        TRANSPORT(w->xanchor[j]);
        cross3(t, o, u);                       /* uses omega before the joint */
        addscl3(a, t, qvel[j]);
        addscl3(o, u, qvel[j]);
      } else {
        cross3(t, o, u);
        addscl3(acc, t, 2 * qvel[j]);
        addscl3(v, u, qvel[j]);
      }
    }
    TRANSPORT(w->xipos[i]);
#undef TRANSPORT
    copy3(om[i], o); copy3(al[i], a); copy3(vc[i], v); copy3(ac[i], acc);
  }
  memset(w->bias, 0, sizeof(double) * nv);
  for (int b = 1; b < nb; b++) {
    double f[3], tau[3], Iw_om[3], t[3];
    for (int k = 0; k < 3; k++) f[k] = m->body_mass[b] * ac[b][k];
    mulmat3(tau, w->iw[b], al[b]);
    mulmat3(Iw_om, w->iw[b], om[b]);
    cross3(t, om[b], Iw_om);
    add3(tau, tau, t);
    for (int d = 0; d < nv; d++) {
      if (!m->affects[b][d]) continue;
      const double* u = w->xaxis[d];
      if (IDX(m->jnt_type, d) == LM_JNT_SLIDE) w->bias[d] += dot3(u, f);
      else {
        double r[3], c[3]; sub3(r, w->xipos[b], w->xanchor[d]); cross3(c, u, r);
        w->bias[d] += dot3(c, f) + dot3(u, tau);
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* constraint cost / forces                                                                    */
/* ------------------------------------------------------------------------------------------ */
/* evaluates the constraint cost at jar; optionally forces/states. Returns cost. */
static double constraint_update(const lmo_model* m, work* w, const double* jar, double* force, int* state) {
  double cost = 0;
  for (int i = 0; i < w->nefc; i++) {
    double D = w->D[i], R = w->R[i];
    switch (w->type[i]) {
      case ROW_FRICTION: {
        double f = w->floss[i], Rf = R * f;
        if (jar[i] <= -Rf) { if (force) { force[i] = f; state[i] = ST_LINEARNEG; } cost += -0.5 * Rf * f - f * jar[i]; }
        else if (jar[i] >= Rf) { if (force) { force[i] = -f; state[i] = ST_LINEARPOS; } cost += -0.5 * Rf * f + f * jar[i]; }
        else { if (force) { force[i] = -D * jar[i]; state[i] = ST_QUADRATIC; } cost += 0.5 * D * jar[i] * jar[i]; }
        break;
      }
      case ROW_LIMIT: case ROW_CONTACT_PLAIN: case ROW_CONTACT_PYR:
        if (jar[i] < 0) { if (force) { force[i] = -D * jar[i]; state[i] = ST_QUADRATIC; } cost += 0.5 * D * jar[i] * jar[i]; }
        else if (force) { force[i] = 0; state[i] = ST_SATISFIED; }
        break;
      case ROW_CONTACT_ELL: {
        const lmo_contact* c = &w->con[w->id[i]];
        int dim = c->dim;
        double mu = c->mu, U[6];
        U[0] = jar[i] * mu;
        double T2 = 0;
        for (int j = 1; j < dim; j++) { U[j] = jar[i + j] * c->friction[j - 1]; T2 += U[j] * U[j]; }
        double N = U[0], T = sqrt(T2);
        if (N >= mu * T || (T <= 0 && N >= 0)) {
          if (force) for (int j = 0; j < dim; j++) { force[i + j] = 0; state[i + j] = ST_SATISFIED; }
        } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
          for (int j = 0; j < dim; j++) {
            cost += 0.5 * w->D[i + j] * jar[i + j] * jar[i + j];
            if (force) { force[i + j] = -w->D[i + j] * jar[i + j]; state[i + j] = ST_QUADRATIC; }
          }
        } else {
          double Dm = D / fmax(MINVAL, mu * mu * (1 + mu * mu));
          double NmT = N - mu * T;
          cost += 0.5 * Dm * NmT * NmT;
          if (force) {
            force[i] = -Dm * NmT * mu; state[i] = ST_CONE;
            for (int j = 1; j < dim; j++) { force[i + j] = -force[i] / T * U[j] * c->friction[j - 1]; state[i + j] = ST_CONE; }
          }
        }
        i += dim - 1;
        break;
      }
    }
  }
  return cost;
}

/* derivatives of the total cost along qacc + alpha*search; jar/jv constraint-space, quad = Gauss part */
static void line_eval(const lmo_model* m, const work* w, const double* jar, const double* jv, const double quad[3],
                      double alpha, double* d1, double* d2) {
  double g = quad[1] + alpha * quad[2], h = quad[2];
  for (int i = 0; i < w->nefc; i++) {
    double D = w->D[i], R = w->R[i];
    double x = jar[i] + alpha * jv[i];
    switch (w->type[i]) {
      case ROW_FRICTION: {
        double f = w->floss[i], Rf = R * f;
        if (x <= -Rf) g += -f * jv[i];
        else if (x >= Rf) g += f * jv[i];
        else { g += D * x * jv[i]; h += D * jv[i] * jv[i]; }
        break;
      }
      case ROW_LIMIT: case ROW_CONTACT_PLAIN: case ROW_CONTACT_PYR:
        if (x < 0) { g += D * x * jv[i]; h += D * jv[i] * jv[i]; }
        break;
      case ROW_CONTACT_ELL: {
        const lmo_contact* c = &w->con[w->id[i]];
        int dim = c->dim;
        double mu = c->mu;
        double N = x * mu, Np = jv[i] * mu, UU = 0, UV = 0, VV = 0;
        for (int j = 1; j < dim; j++) {
          double u = (jar[i + j] + alpha * jv[i + j]) * c->friction[j - 1], v = jv[i + j] * c->friction[j - 1];
          UU += u * u; UV += u * v; VV += v * v;
        }
        double T = sqrt(UU);
        if (N >= mu * T || (T <= 0 && N >= 0)) {
        } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
          for (int j = 0; j < dim; j++) {
            double xj = jar[i + j] + alpha * jv[i + j];
            g += w->D[i + j] * xj * jv[i + j]; h += w->D[i + j] * jv[i + j] * jv[i + j];
          }
        } else {
          double Dm = D / fmax(MINVAL, mu * mu * (1 + mu * mu));
          double Tp = UV / T, Tpp = VV / T - UV * UV / (T * T * T);
          double NmT = N - mu * T, NmTp = Np - mu * Tp;
          g += Dm * NmT * NmTp;
          h += Dm * (NmTp * NmTp - NmT * mu * Tpp);
        }
        i += dim - 1;
        break;
      }
    }
  }
  *d1 = g; *d2 = h;
}

/* Newton solver on the primal problem: min_a 0.5 (a-a0)^T M (a-a0) + s(J a - aref) */
static void solve_constraints(const lmo_model* m, work* w, const double* warmstart) {
  int nv = m->nv, nefc = w->nefc;
  if (nefc == 0) { memcpy(w->qacc, w->qacc_smooth, sizeof(double) * nv); memset(w->qfrc_constraint, 0, sizeof(double) * nv); w->solver_iter = 0; return; }
  double jar[LMO_MAXEFC], jv[LMO_MAXEFC], Ma[LMO_MAXV], grad[LMO_MAXV], search[LMO_MAXV], Mv[LMO_MAXV];
  double H[LMO_MAXV * LMO_MAXV], LH[LMO_MAXV * LMO_MAXV];
  double* qacc = w->qacc;

  /* warm start: whichever of (warmstart, qacc_smooth) has the lower total cost */
  double best_cost = 0;
  for (int cand = 0; cand < 2; cand++) {
    const double* a = cand == 0 ? w->qacc_smooth : warmstart;
    if (!a) continue;
    for (int i = 0; i < nefc; i++) { double s = -w->aref[i]; for (int d = 0; d < nv; d++) s += w->J[i*nv + d] * a[d]; jar[i] = s; }
    double cost = constraint_update(m, w, jar, NULL, NULL);
    for (int i = 0; i < nv; i++) { double s = 0; for (int j = 0; j < nv; j++) s += w->M[i*nv + j] * (a[j] - w->qacc_smooth[j]); cost += 0.5 * s * (a[i] - w->qacc_smooth[i]); }
    if (cand == 0 || cost < best_cost) { best_cost = cost; memcpy(qacc, a, sizeof(double) * nv); }
  }

  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  int iter = 0;
  double cost = 0;
  for (iter = 0; iter < m->iterations; iter++) {
    for (int i = 0; i < nefc; i++) { double s = -w->aref[i]; for (int d = 0; d < nv; d++) s += w->J[i*nv + d] * qacc[d]; jar[i] = s; }
    for (int i = 0; i < nv; i++) { double s = 0; for (int j = 0; j < nv; j++) s += w->M[i*nv + j] * qacc[j]; Ma[i] = s; }
    cost = constraint_update(m, w, jar, w->force, w->state);
    for (int i = 0; i < nv; i++) cost += 0.5 * (Ma[i] - w->smooth[i]) * (qacc[i] - w->qacc_smooth[i]);
    for (int d = 0; d < nv; d++) { double s = 0; for (int i = 0; i < nefc; i++) s += w->J[i*nv + d] * w->force[i]; w->qfrc_constraint[d] = s; }
    double gnorm = 0;
    for (int d = 0; d < nv; d++) { grad[d] = Ma[d] - w->smooth[d] - w->qfrc_constraint[d]; gnorm += grad[d] * grad[d]; }
    gnorm = sqrt(gnorm);
    if (scale * gnorm < m->tolerance) break;

    /* Hessian = M + J^T diag(D_active) J + cone blocks */
    memcpy(H, w->M, sizeof(double) * nv * nv);
    for (int i = 0; i < nefc; i++) {
      if (w->state[i] == ST_QUADRATIC) {
        const double* Ji = w->J + i*nv; double D = w->D[i];
        for (int a = 0; a < nv; a++) { if (Ji[a] == 0) continue; for (int b = 0; b < nv; b++) H[a*nv + b] += D * Ji[a] * Ji[b]; }
      } else if (w->state[i] == ST_CONE) {
        const lmo_contact* c = &w->con[w->id[i]];
        int dim = c->dim; double mu = c->mu;
        double U[6], Sc[6], hc[36];
        Sc[0] = mu; U[0] = jar[i] * mu; double T2 = 0;
        for (int j = 1; j < dim; j++) { Sc[j] = c->friction[j - 1]; U[j] = jar[i + j] * Sc[j]; T2 += U[j] * U[j]; }
        double T = sqrt(T2), N = U[0];
        double Dm = w->D[i] / fmax(MINVAL, mu * mu * (1 + mu * mu)), g = N - mu * T;
        hc[0] = Dm;
        for (int j = 1; j < dim; j++) { hc[j] = hc[j*dim] = -Dm * mu * U[j] / T; }
        for (int j = 1; j < dim; j++) for (int k = 1; k < dim; k++) {
          double tt = U[j] * U[k] / (T * T);
          hc[j*dim + k] = Dm * mu * mu * tt - Dm * g * mu * ((j == k ? 1.0 : 0.0) - tt) / T;
        }
        for (int j = 0; j < dim; j++) for (int k = 0; k < dim; k++) hc[j*dim + k] *= Sc[j] * Sc[k];
        for (int j = 0; j < dim; j++) for (int k = 0; k < dim; k++) {
          const double *Jj = w->J + (i + j)*nv, *Jk = w->J + (i + k)*nv; double hjk = hc[j*dim + k];
          for (int a = 0; a < nv; a++) { if (Jj[a] == 0) continue; for (int b = 0; b < nv; b++) H[a*nv + b] += hjk * Jj[a] * Jk[b]; }
        }
        i += dim - 1;
      }
    }
    if (cholesky(LH, H, nv)) break;
    for (int d = 0; d < nv; d++) search[d] = -grad[d];
    chol_solve(LH, nv, search);

    /* exact line search along `search` */
    for (int i = 0; i < nefc; i++) { double s = 0; for (int d = 0; d < nv; d++) s += w->J[i*nv + d] * search[d]; jv[i] = s; }
    for (int i = 0; i < nv; i++) { double s = 0; for (int j = 0; j < nv; j++) s += w->M[i*nv + j] * search[j]; Mv[i] = s; }
    double quad[3] = {0, 0, 0};
    for (int d = 0; d < nv; d++) { quad[1] += search[d] * (Ma[d] - w->smooth[d]); quad[2] += search[d] * Mv[d]; }
    double lo = 0, hi = -1, d1, d2, alpha;
    line_eval(m, w, jar, jv, quad, 0, &d1, &d2);
    if (d1 >= 0 || d2 <= 0) break;
    double dlo = d1;
    alpha = -d1 / d2;
    for (int ls = 0; ls < 100; ls++) {
      line_eval(m, w, jar, jv, quad, alpha, &d1, &d2);
      if (fabs(d1) < 1e-14 * fmax(1.0, fabs(dlo))) break;
      if (d1 < 0) lo = alpha; else hi = alpha;
      double next = alpha - d1 / d2;
      if (hi > 0 && !(next > lo && next < hi)) next = 0.5 * (lo + hi);
      if (hi < 0 && next <= lo) next = 2 * alpha + 1e-12;
      if (fabs(next - alpha) <= 1e-16 * fabs(alpha)) { alpha = next; break; }
      alpha = next;
    }
    double moved = 0;
    for (int d = 0; d < nv; d++) { qacc[d] += alpha * search[d]; moved += fabs(alpha * search[d]); }
    if (moved == 0) break;
  }
  w->solver_iter = iter;
  /* final forces at the returned qacc */
  for (int i = 0; i < nefc; i++) { double s = -w->aref[i]; for (int d = 0; d < nv; d++) s += w->J[i*nv + d] * qacc[d]; jar[i] = s; }
  constraint_update(m, w, jar, w->force, w->state);
  for (int d = 0; d < nv; d++) { double s = 0; for (int i = 0; i < nefc; i++) s += w->J[i*nv + d] * w->force[i]; w->qfrc_constraint[d] = s; }
}

/* ------------------------------------------------------------------------------------------ */
/* forward dynamics, integrators                                                               */
/* ------------------------------------------------------------------------------------------ */
/* ---- muscle model (MuJoCo 2.3.7 mju_muscleGain / mju_muscleBias / mju_muscleDynamics, restated) ---- */
/* Active force-length curve exactly as the reference's engine (MuJoCo 2.3.7) evaluates it. Its branch chain is
   "lmin <= L <= a", else "L <= 1", else "L <= b", else "L <= lmax", else 0 — so a muscle SHORTER than lmin falls into
   the second branch and gets 1 - 0.5((1-L)/(1-a))^2, which is negative there (about -1 just below lmin = 0.5).
   Pinned by HumanoidMuscle.walk golden rows 15-18 and 22-27 (glut_max3_r / peri_r shorten below 0.5): with FL = 0
   below lmin those rows miss by 0.4-0.8 rad/s, with the fall-through they match to 1e-15. L > lmax is unpinned. */
static double muscle_gain_length(double L, double lmin, double lmax) {
  double a = 0.5 * (lmin + 1), b = 0.5 * (1 + lmax), x;
  if (L >= lmin && L <= a) { x = (L - lmin) / fmax(MINVAL, a - lmin); return 0.5 * x * x; }
  if (L <= 1) { x = (1 - L) / fmax(MINVAL, 1 - a); return 1 - 0.5 * x * x; }
  if (L <= b) { x = (L - 1) / fmax(MINVAL, b - 1); return 1 - 0.5 * x * x; }
  if (L <= lmax) { x = (lmax - L) / fmax(MINVAL, lmax - b); return 0.5 * x * x; }
  return 0;
}
static double muscle_gain(double len, double vel, const double* lengthrange, const double* prm) {
  double range0 = prm[0], range1 = prm[1], force = prm[2], lmin = prm[4], lmax = prm[5], vmax = prm[6], fvmax = prm[8];
  double L0 = (lengthrange[1] - lengthrange[0]) / fmax(MINVAL, range1 - range0);
  double L = range0 + (len - lengthrange[0]) / fmax(MINVAL, L0);
  double V = vel / fmax(MINVAL, L0 * vmax);
  double FL = muscle_gain_length(L, lmin, lmax), FV, y = fvmax - 1;
  if (V <= -1) FV = 0;
  else if (V <= 0) FV = (V + 1) * (V + 1);
  else if (V <= y) FV = fvmax - (y - V) * (y - V) / fmax(MINVAL, y);
  else FV = fvmax;
  return -force * FL * FV;
}
static double muscle_bias(double len, const double* lengthrange, const double* prm) {
  double range0 = prm[0], range1 = prm[1], force = prm[2], lmax = prm[5], fpmax = prm[7];
  double L0 = (lengthrange[1] - lengthrange[0]) / fmax(MINVAL, range1 - range0);
  double L = range0 + (len - lengthrange[0]) / fmax(MINVAL, L0);
  double b = 0.5 * (1 + lmax), x;
  if (L <= 1) return 0;
  if (L <= b) { x = (L - 1) / fmax(MINVAL, b - 1); return -force * fpmax * 0.5 * x * x; }
  x = (L - b) / fmax(MINVAL, b - 1); return -force * fpmax * (0.5 + x);
}
static double muscle_dynamics(double ctrl, double act, const double* prm) {
  double c = ctrl < 0 ? 0 : (ctrl > 1 ? 1 : ctrl), a = act < 0 ? 0 : (act > 1 ? 1 : act);
  double tau_act = prm[0] * (0.5 + 1.5 * a), tau_deact = prm[1] / (0.5 + 1.5 * a);
  double dctrl = c - act, tau;
  if (prm[2] < MINVAL) tau = dctrl > 0 ? tau_act : tau_deact;
  else {                                      /* quintic sigmoid blend over the width prm[2] */
    double x = dctrl / prm[2] + 0.5, sg;
    if (x <= 0) sg = 0; else if (x >= 1) sg = 1; else sg = x * x * x * (3 * x * (2 * x - 5) + 10);
    tau = tau_deact + (tau_act - tau_deact) * sg;
  }
  return dctrl / fmax(MINVAL, tau);
}

/* spatial tendon through sites: length and moment arms (d length / d qpos) */
static double tendon_length(const lmo_model* m, const work* w, int t, double* moment /* nv */) {
  int nv = m->nv, adr = IDX(m->tendon_adr, t), num = IDX(m->tendon_num, t);
  double len = 0;
  memset(moment, 0, sizeof(double) * nv);
  for (int i = 0; i + 1 < num; i++) {
    int s0 = IDX(m->wrap_site, adr + i), s1 = IDX(m->wrap_site, adr + i + 1);
    int b0 = IDX(m->site_body, s0), b1 = IDX(m->site_body, s1);
    double p0[3], p1[3], d[3];
    mulmat3(p0, w->xmat[b0], m->site_pos + 3*s0); for (int k = 0; k < 3; k++) p0[k] += w->xpos[b0][k];
    mulmat3(p1, w->xmat[b1], m->site_pos + 3*s1); for (int k = 0; k < 3; k++) p1[k] += w->xpos[b1][k];
    sub3(d, p1, p0);
    double seg = norm3(d);
    len += seg;
    if (b0 == b1 || seg < MINVAL) continue;
    for (int k = 0; k < 3; k++) d[k] /= seg;
    for (int dof = 0; dof < nv; dof++) {
      /* velocity of p1 minus velocity of p0 per unit qvel[dof], projected on the segment direction */
      double j = 0;
      for (int side = 0; side < 2; side++) {
        int b = side ? b1 : b0; const double* p = side ? p1 : p0;
        if (!m->affects[b][dof]) continue;
        double jv[3];
        if (IDX(m->jnt_type, dof) == LM_JNT_HINGE) { double r[3]; sub3(r, p, w->xanchor[dof]); cross3(jv, w->xaxis[dof], r); }
        else copy3(jv, w->xaxis[dof]);
        j += (side ? 1.0 : -1.0) * dot3(d, jv);
      }
      moment[dof] += j;
    }
  }
  return len;
}

static void forward(const lmo_model* m, const double* qpos, const double* qvel, const double* ctrl,
                    const double* act, const double* warmstart, work* w) {
  int nv = m->nv;
  kinematics(m, qpos, w);
  mass_matrix(m, w);
  cholesky(w->L, w->M, nv);
  collide(m, w);
  for (int ci = 0; ci < w->ncon; ci++)
    if (IDX(m->geom_type, w->con[ci].geom1) != LM_GEOM_PLANE && -w->con[ci].dist > w->max_self_depth) w->max_self_depth = -w->con[ci].dist;
  make_constraints(m, qpos, w);
  /* velocity stage */
  for (int d = 0; d < nv; d++) w->passive[d] = -m->jnt_stiffness[d] * qpos[d] - m->dof_damping[d] * qvel[d];
  for (int i = 0; i < w->nefc; i++) {
    double s = 0; for (int d = 0; d < nv; d++) s += w->J[i*nv + d] * qvel[d];
    w->vel[i] = s;
    w->aref[i] = -w->B[i] * s - w->K[i] * w->imp[i] * (w->pos[i] - w->margin[i]);
  }
  rne_bias(m, qvel, w);
  /* actuation */
  memset(w->actuator, 0, sizeof(double) * nv);
  for (int a = 0, ia = 0; a < m->nu; a++) {
    double c = ctrl[a];
    if (IDX(m->act_ctrllimited, a)) { if (c < m->act_ctrlrange[2*a]) c = m->act_ctrlrange[2*a]; if (c > m->act_ctrlrange[2*a + 1]) c = m->act_ctrlrange[2*a + 1]; }
    if (IDX(m->act_kind, a) == LM_ACT_MOTOR) { w->actuator[IDX(m->act_dof, a)] += m->act_gear[a] * c; w->actuator_force[a] = c; continue; }
    if (IDX(m->act_kind, a) == LM_ACT_POSITION) {
      /* affine servo on a joint (mj_fwdActuation: force = gain*ctrl + bias, bias affine in actuator length / velocity,
         clamped to forcerange; joint transmission: length = gear*q, moment = gear) */
      const int d = IDX(m->act_dof, a);
      const double gear = m->act_gear[a], len = gear * qpos[d], vel = gear * qvel[d];
      double f = m->act_gainprm[9*a] * c + m->act_biasprm[3*a] + m->act_biasprm[3*a + 1] * len + m->act_biasprm[3*a + 2] * vel;
      if (IDX(m->act_forcelimited, a)) { if (f < m->act_forcerange[2*a]) f = m->act_forcerange[2*a]; if (f > m->act_forcerange[2*a + 1]) f = m->act_forcerange[2*a + 1]; }
      w->actuator[d] += gear * f; w->actuator_force[a] = f; w->actuator_length[a] = len; w->actuator_velocity[a] = vel;
      continue;
    }
    /* muscle on a tendon: length/velocity through the gear, force = gain(len, vel) * act + bias(len) */
    double moment[LMO_MAXV], gear = m->act_gear[a];
    double len = gear * tendon_length(m, w, IDX(m->act_tendon, a), moment), vel = 0;
    for (int d = 0; d < nv; d++) { moment[d] *= gear; vel += moment[d] * qvel[d]; }
    double av = act ? act[ia] : 0.0;
    double f = muscle_gain(len, vel, m->act_lengthrange + 2*a, m->act_gainprm + 9*a) * av
             + muscle_bias(len, m->act_lengthrange + 2*a, m->act_gainprm + 9*a);
    w->act_dot[ia] = muscle_dynamics(c, av, m->act_dynprm + 3*a);
    w->actuator_force[a] = f; w->actuator_length[a] = len; w->actuator_velocity[a] = vel;
    for (int d = 0; d < nv; d++) w->actuator[d] += moment[d] * f;
    ia++;
  }
  for (int d = 0; d < nv; d++) { w->smooth[d] = w->passive[d] - w->bias[d] + w->actuator[d]; w->qacc_smooth[d] = w->smooth[d]; }
  chol_solve(w->L, nv, w->qacc_smooth);
  solve_constraints(m, w, warmstart);
}

static void euler(const lmo_model* m, double* qpos, double* qvel, work* w) {
  int nv = m->nv;
  double h = m->timestep, acc[LMO_MAXV];
  int damped = 0;
  for (int d = 0; d < nv; d++) if (m->dof_damping[d] > 0) damped = 1;
  if (!damped) memcpy(acc, w->qacc, sizeof(double) * nv);
  else {
    double A[LMO_MAXV * LMO_MAXV], LA[LMO_MAXV * LMO_MAXV];
    memcpy(A, w->M, sizeof(double) * nv * nv);
    for (int d = 0; d < nv; d++) { A[d*nv + d] += h * m->dof_damping[d]; acc[d] = w->smooth[d] + w->qfrc_constraint[d]; }
    cholesky(LA, A, nv);
    chol_solve(LA, nv, acc);
  }
  for (int d = 0; d < nv; d++) qvel[d] += h * acc[d];
  for (int d = 0; d < nv; d++) qpos[d] += h * qvel[d];
}

int lmo_step(const lmo_model* m, double* qpos, double* qvel, const double* ctrl, double* warmstart, int nsub,
             lmo_stats* stats) {
  if (m->na > 0) return -1;          /* models with activation states go through lmo_step_act */
  return lmo_step_act(m, qpos, qvel, NULL, ctrl, warmstart, nsub, stats);
}

static int step_impl(const lmo_model* m, double* qpos, double* qvel, double* act, const double* ctrl, double* warmstart,
                     int nsub, lmo_stats* stats, work* w);

int lmo_step_act(const lmo_model* m, double* qpos, double* qvel, double* act, const double* ctrl, double* warmstart,
                 int nsub, lmo_stats* stats) {
  if (m->na > 0 && (!act || m->integrator != LM_INT_EULER)) return -1;
  work* w = (work*)malloc(sizeof(work));
  int rc = step_impl(m, qpos, qvel, act, ctrl, warmstart, nsub, stats, w);
  free(w);
  return rc;
}

/* One substep, then the contact list of its LAST forward pass with the contact-frame force of every contact
   (normal, tangent 1, tangent 2) — what the reference reads after each intermediate step when use_foot_forces is on
   (mushroom-rl _get_collision_force -> mj_contactForce; reference base.py:623-631,667-679). For RK4 the engine's data
   hold the fourth stage's evaluation when mj_step returns (unpinned: no golden rollout has foot forces).
   out rows: geom1, geom2, f_normal, f_t1, f_t2. */
int lmo_step_contact_forces(const lmo_model* m, double* qpos, double* qvel, double* act, const double* ctrl,
                            double* warmstart, double* out, int max_con, int* ncon) {
  if (m->na > 0 && (!act || m->integrator != LM_INT_EULER)) return -1;
  work* w = (work*)malloc(sizeof(work));
  int rc = step_impl(m, qpos, qvel, act, ctrl, warmstart, 1, NULL, w);
  int n = w->ncon < max_con ? w->ncon : max_con;
  for (int i = 0; i < n; i++) {
    const lmo_contact* c = &w->con[i];
    double* o = out + 5 * i;
    o[0] = c->geom1; o[1] = c->geom2; o[2] = o[3] = o[4] = 0;
    if (c->efc_address < 0) continue;
    const double* f = w->force + c->efc_address;
    if (m->cone == LM_CONE_ELLIPTIC || c->dim == 1) {
      for (int k = 0; k < c->dim && k < 3; k++) o[2 + k] = f[k];
    } else {                       /* pyramid edges n+mu t1, n-mu t1, n+mu t2, n-mu t2 (condim 3) */
      o[2] = f[0] + f[1] + f[2] + f[3];
      o[3] = c->friction[0] * (f[0] - f[1]);
      o[4] = c->friction[1] * (f[2] - f[3]);
    }
  }
  *ncon = n;
  free(w);
  return rc;
}

static int step_impl(const lmo_model* m, double* qpos, double* qvel, double* act, const double* ctrl, double* warmstart,
                     int nsub, lmo_stats* stats, work* w) {
  int nv = m->nv;
  if (stats) memset(stats, 0, sizeof(*stats));
  for (int s = 0; s < nsub; s++) {
    if (m->integrator == LM_INT_EULER) {
      forward(m, qpos, qvel, ctrl, act, warmstart, w);
      if (stats) { stats->convex_contacts += w->convex_contacts; stats->native_contacts += w->native_contacts; stats->own_contacts += w->own_contacts; stats->own_face_contacts += w->own_face_contacts; if (w->max_self_depth > stats->max_self_depth) stats->max_self_depth = w->max_self_depth; }
      if (warmstart) memcpy(warmstart, w->qacc, sizeof(double) * nv);
      euler(m, qpos, qvel, w);
      for (int i = 0; i < m->na; i++) act[i] += m->timestep * w->act_dot[i];      /* explicit Euler on activations */
    } else {
      /* classical RK4 on (qpos,qvel); every stage is a full forward pass, no implicit damping */
      static const double A[3] = {0.5, 0.5, 1.0}, Bw[4] = {1.0/6, 1.0/3, 1.0/3, 1.0/6};
      double h = m->timestep, q0[LMO_MAXV], v0[LMO_MAXV], X[LMO_MAXV], V[LMO_MAXV], dq[LMO_MAXV], dv[LMO_MAXV];
      memcpy(q0, qpos, sizeof(double) * nv); memcpy(v0, qvel, sizeof(double) * nv);
      memset(dq, 0, sizeof(dq)); memset(dv, 0, sizeof(dv));
      memcpy(X, q0, sizeof(double) * nv); memcpy(V, v0, sizeof(double) * nv);
      for (int st = 0; st < 4; st++) {
        forward(m, X, V, ctrl, NULL, warmstart, w);
        if (stats) { stats->convex_contacts += w->convex_contacts; stats->native_contacts += w->native_contacts; stats->own_contacts += w->own_contacts; stats->own_face_contacts += w->own_face_contacts; if (w->max_self_depth > stats->max_self_depth) stats->max_self_depth = w->max_self_depth; }
        if (st == 0 && warmstart) memcpy(warmstart, w->qacc, sizeof(double) * nv);
        for (int d = 0; d < nv; d++) { dq[d] += Bw[st] * V[d]; dv[d] += Bw[st] * w->qacc[d]; }
        if (st < 3) for (int d = 0; d < nv; d++) { double vn = v0[d] + h * A[st] * w->qacc[d]; X[d] = q0[d] + h * A[st] * V[d]; V[d] = vn; }
        if (stats && st == 0) { stats->ncon = w->ncon; stats->nefc = w->nefc; }
      }
      for (int d = 0; d < nv; d++) { qpos[d] = q0[d] + h * dq[d]; qvel[d] = v0[d] + h * dv[d]; }
    }
    if (stats) {
      stats->ncon = w->ncon; stats->nefc = w->nefc; stats->solver_iter_total += w->solver_iter;
      if (w->solver_iter > stats->solver_iter_max) stats->solver_iter_max = w->solver_iter;
      stats->unhandled_pairs += w->unhandled_pairs;
    }
  }
  return 0;
}

/* stage-level dump for parity tests: one forward pass at (qpos,qvel,ctrl) */
int lmo_forward(const lmo_model* m, const double* qpos, const double* qvel, const double* ctrl, const double* warmstart,
                lmo_forward_out* out) {
  return lmo_forward_act(m, qpos, qvel, NULL, ctrl, warmstart, out);
}

int lmo_forward_act(const lmo_model* m, const double* qpos, const double* qvel, const double* act, const double* ctrl,
                    const double* warmstart, lmo_forward_out* out) {
  work* w = (work*)malloc(sizeof(work));
  int nv = m->nv;
  forward(m, qpos, qvel, ctrl, act, warmstart, w);
  if (out->actuator_force) memcpy(out->actuator_force, w->actuator_force, sizeof(double) * m->nu);
  if (out->actuator_length) memcpy(out->actuator_length, w->actuator_length, sizeof(double) * m->nu);
  if (out->M) memcpy(out->M, w->M, sizeof(double) * nv * nv);
  if (out->bias) memcpy(out->bias, w->bias, sizeof(double) * nv);
  if (out->passive) memcpy(out->passive, w->passive, sizeof(double) * nv);
  if (out->actuator) memcpy(out->actuator, w->actuator, sizeof(double) * nv);
  if (out->qacc_smooth) memcpy(out->qacc_smooth, w->qacc_smooth, sizeof(double) * nv);
  if (out->qacc) memcpy(out->qacc, w->qacc, sizeof(double) * nv);
  if (out->qfrc_constraint) memcpy(out->qfrc_constraint, w->qfrc_constraint, sizeof(double) * nv);
  if (out->xpos) memcpy(out->xpos, w->xpos, sizeof(double) * 3 * m->nbody);
  if (out->xmat) memcpy(out->xmat, w->xmat, sizeof(double) * 9 * m->nbody);
  if (out->geom_xpos) memcpy(out->geom_xpos, w->gpos, sizeof(double) * 3 * m->ngeom);
  out->ncon = w->ncon; out->nefc = w->nefc; out->solver_iter = w->solver_iter; out->unhandled_pairs = w->unhandled_pairs;
  int nc = w->ncon < out->max_con ? w->ncon : out->max_con;
  if (out->contacts) memcpy(out->contacts, w->con, sizeof(lmo_contact) * nc);
  int ne = w->nefc < out->max_efc ? w->nefc : out->max_efc;
  if (out->efc_J) memcpy(out->efc_J, w->J, sizeof(double) * ne * nv);
  if (out->efc_aref) memcpy(out->efc_aref, w->aref, sizeof(double) * ne);
  if (out->efc_R) memcpy(out->efc_R, w->R, sizeof(double) * ne);
  if (out->efc_force) memcpy(out->efc_force, w->force, sizeof(double) * ne);
  if (out->efc_type) memcpy(out->efc_type, w->type, sizeof(int) * ne);
  free(w);
  return 0;
}
