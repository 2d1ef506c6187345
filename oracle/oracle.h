/*
 * oracle.h — C interface of the fp64 CPU oracle (TEST INFRASTRUCTURE ONLY; see oracle.c header).
 * Loaded with ctypes by oracle/pyoracle.py from tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg. Never linked into the product library.
 */
#ifndef LMO_ORACLE_H
#define LMO_ORACLE_H

#define LMO_MAXBODY 64
#define LMO_MAXV 32
#define LMO_MAXU 128
#define LMO_MAXGEOM 160
#define LMO_MAXPAIR 8192
#define LMO_MAXCON 96
#define LMO_MAXEFC 400

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lmo_model lmo_model;

typedef struct {
  double dist, pos[3], frame[9], includemargin, margin, friction[5], solref[2], solimp[5], mu;
  int dim, geom1, geom2, efc_address;
} lmo_contact;

typedef struct {
  int ncon, nefc, solver_iter_total, solver_iter_max, unhandled_pairs;
  int convex_contacts;        /* contacts from the convex-convex collider (MPR), summed over the forward passes */
  double max_self_depth;      /* deepest penetration (-dist) of a contact between two bodies of the robot over all forward passes */
  int native_contacts;        /* contacts of the native box / cylinder colliders (sphere-box, sphere-cylinder, capsule-box, box-box) */
  int own_contacts;           /* of those: capsule-box and box-box contacts — this repository's own manifold constructions (oracle.c nat_*) */
  int own_face_contacts;      /* of those: contacts of the box-box FACE case (incident face clipped against the reference face: approximate) */
} lmo_stats;

typedef struct {
  /* caller-provided buffers (any may be NULL) */
  double *M, *bias, *passive, *actuator, *qacc_smooth, *qacc, *qfrc_constraint, *xpos, *xmat, *geom_xpos;
  lmo_contact* contacts; int max_con;
  double *efc_J, *efc_aref, *efc_R, *efc_force; int* efc_type; int max_efc;
  double *actuator_force, *actuator_length;   /* [nu] (length: muscles only) */
  /* outputs */
  int ncon, nefc, solver_iter, unhandled_pairs;
} lmo_forward_out;

lmo_model* lmo_model_create(const double* blob, long n);
void lmo_model_destroy(lmo_model* m);
/* what: 0 = disable self collision (value!=0), 1 = solver iterations, 2 = solver tolerance */
int lmo_set_mesh(lmo_model* m, int g, int nv, const double* vert);
/* the vertex graph of mesh geom g's hull (further plane-hull contacts at the neighbours of the support vertex): comes with the
   model blob; settable for hulls attached with lmo_set_mesh, and clearable (all-zero adr) for A/B tests */
int lmo_set_mesh_graph(lmo_model* m, int g, const int* adr, const int* nbr, double tol);
int lmo_mesh_nvert(const lmo_model* m, int g);
void lmo_set_option(lmo_model* m, int what, double value);
/* one native box / cylinder collider on its own (tests): geom types as in lm_layout.h, R row-major; out [8][7] = (dist, pos, normal
   from geom 1 to geom 2); returns the number of contacts, -1 = the pair type has no native collider */
int lmo_test_native_pair(int t1, const double* p1, const double* R1, const double* s1, int t2, const double* p2, const double* R2,
                         const double* s2, double margin, double* out);
int lmo_nv(const lmo_model* m);
int lmo_nu(const lmo_model* m);
int lmo_na(const lmo_model* m);   /* activation states (muscles) */

/* advance (qpos,qvel) by nsub physics substeps under constant ctrl; warmstart (nv) may be NULL */
int lmo_step(const lmo_model* m, double* qpos, double* qvel, const double* ctrl, double* warmstart, int nsub,
             lmo_stats* stats);
/* same with activation states act[na] (advanced in place; explicit Euler); required when lmo_na() > 0 */
int lmo_step_act(const lmo_model* m, double* qpos, double* qvel, double* act, const double* ctrl, double* warmstart,
                 int nsub, lmo_stats* stats);
/* one substep + contact-frame forces of its last forward pass: out[max_con][5] = geom1, geom2, f_n, f_t1, f_t2 */
int lmo_step_contact_forces(const lmo_model* m, double* qpos, double* qvel, double* act, const double* ctrl,
                            double* warmstart, double* out, int max_con, int* ncon);
/* one forward-dynamics pass with intermediate results */
int lmo_forward(const lmo_model* m, const double* qpos, const double* qvel, const double* ctrl,
                const double* warmstart, lmo_forward_out* out);
int lmo_forward_act(const lmo_model* m, const double* qpos, const double* qvel, const double* act, const double* ctrl,
                    const double* warmstart, lmo_forward_out* out);

#ifdef __cplusplus
}
#endif
#endif
