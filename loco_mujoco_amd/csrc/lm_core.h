// lm_core.h — one physics substep of the "root + chains" model, written for ONE LANE = ONE CHAIN.
//
// The 4 lanes of a quad simulate one environment: lane c owns chain c (A1: one leg), every lane carries a
// replica of the 6 root dofs. Cross-lane traffic is nothing but 4-lane sums (Q::sum), which on gfx950 are two
// DPP quad_perm adds — no LDS, no barriers. All link quantities are spatial vectors about the root-frame
// origin O in world axes, so neither Jacobians nor per-link transforms are ever stored:
//   * joint-space inertia  M_ij = S_i . (Ic_k S_j)           (composite rigid body, Ic about O)
//   * bias forces          spatial Newton-Euler about O       (gravity as base acceleration)
//   * contact Jacobians    column j of a floor contact at r = p - O is (v_j + w_j x r, w_j) re-labelled by the
//                          axis-aligned contact frame; rebuilt from the 9 twists whenever needed
//   * M and the Newton Hessian H = M + J^T W J share the arrow sparsity {chain block, chain-root coupling,
//     root block}: every lane eliminates its own chain block, the 6x6 root Schur complement is a 21-float
//     quad sum, its Cholesky is replicated.
//   * two departures from the arrow, each handled as a correction around it: a self-contact between two chains adds a
//     cross block between two lanes (arrow_factor_x: the pair is eliminated together); two chains that share their first
//     link (a torso with an arm on either side) tie the two copies of its dof in every solve (tie_shared_dof).
// Semantics restated: MuJoCo 2.3.7 mj_step (third party; the reference reaches it through mushroom-rl's
// MuJoCo.step): soft constraints with solref/solimp impedance, friction-loss,
// joint-limit, pyramidal and elliptic-cone contact rows (floor: sphere, capsule, cylinder, box, convex hull; sphere /
// capsule pairs between links), Newton on the convex primal problem with exact line search, semi-implicit Euler with
// implicit joint damping or RK4, spatial tendons and muscles.
//
// This header has no HIP dependency: the includer defines LM_DEV (function qualifier) and supplies the quad
// policy Q {sum(float), any(bool), kRep, kPoints, rep(), rep_bcast(x, r), rep_sum(x), fence(), peer(lmem, ls, i, dl), peer_write(lmem, ls, i, dl, v), env_ballot(bool),
// quad_read(x, lane), quad_sync()}. csrc/lm_step.h instantiates it with DPP / ds_bpermute intrinsics (the step kernel),
// tests/emu/emu.cpp with OS threads.
#pragma once
#include <math.h>
#include <type_traits>
#include "../../include/lm_layout.h"

#define LM_PAIR_PAD 0.03f     // metres added to the reach of the link-pair list (lowering.PAIR_PAD)
#ifndef LM_DEV_COLD
#define LM_DEV_COLD LM_DEV    // the includer may make the rare, register-hungry helpers real functions (lm_step.h: noinline)
#endif
LM_DEV unsigned lm_f2u(float x) { return __builtin_bit_cast(unsigned, x); }
LM_DEV float lm_u2f(unsigned x) { return __builtin_bit_cast(float, x); }
#ifndef LM_GLD64
// 64-bit words of the convex collider's warm-start cache (global memory, written by one lane of an environment in one forward pass and read
// by another in the next): the step kernel (lm_step.h) makes them agent-scope relaxed atomics so that a read sees the last write, not a
// stale line of the CU's L1; the emulator keeps the plain forms
#define LM_GLD64(p) (*reinterpret_cast<const unsigned long long*>(p))
#define LM_GST64(p, v) (*reinterpret_cast<unsigned long long*>(p) = (v))
#endif
#ifndef LM_LMEM_T
#define LM_LMEM_T float       // element type of lane memory (A/B probe: `volatile float`)
#endif

namespace lm {

struct V3 { float x, y, z; };
struct alignas(16) F4 { float x, y, z, w; };      // one 128-bit load
LM_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
LM_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
LM_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
LM_DEV V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
LM_DEV float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
LM_DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

struct Sp { V3 w, v; };   // spatial motion (angular, linear at O)  |  spatial force (moment about O, force)
LM_DEV Sp sp0() { Sp s; s.w = v3(0, 0, 0); s.v = v3(0, 0, 0); return s; }
LM_DEV Sp operator+(Sp a, Sp b) { Sp s; s.w = a.w + b.w; s.v = a.v + b.v; return s; }
LM_DEV Sp operator*(float k, Sp a) { Sp s; s.w = k * a.w; s.v = k * a.v; return s; }
LM_DEV float spdot(Sp m, Sp f) { return dot(m.w, f.w) + dot(m.v, f.v); }   // motion . force

struct M3 { float a[9]; };     // row-major
LM_DEV V3 mul(const M3& m, V3 v) {
  return v3(m.a[0] * v.x + m.a[1] * v.y + m.a[2] * v.z, m.a[3] * v.x + m.a[4] * v.y + m.a[5] * v.z,
            m.a[6] * v.x + m.a[7] * v.y + m.a[8] * v.z);
}
LM_DEV M3 mul(const M3& p, const M3& q) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.a[3 * i + j] = p.a[3 * i] * q.a[j] + p.a[3 * i + 1] * q.a[3 + j] + p.a[3 * i + 2] * q.a[6 + j];
  return r;
}
// sin and cos together: one Cody-Waite reduction by pi/2 (joint angles are a few turns at most), then the two
// float32 minimax polynomials on [-pi/4, pi/4] (~1 ulp there). About 25 instructions for the pair, where the
// library's separate sinf + cosf (full-range reduction each) inline to ~270.
LM_DEV void sincos_small(float x, float& s, float& c) {
  const float k = rintf(x * 0.636619772367581f);
  float r = fmaf(-k, 1.5707962513e+0f, x);
  r = fmaf(-k, 7.5497894159e-8f, r);
  const float z = r * r;
  const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z, fmaf(-0.5f, z, 1.0f));
  const int q = (int)k;
  const float ss = (q & 1) ? pc : ps, cc = (q & 1) ? ps : pc;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
}
// R <- Rot(unit axis u, angle) * R
LM_DEV void rotate_world(M3& R, V3 u, float angle) {
  float s, c;
  sincos_small(angle, s, c);
  const float c1 = 1.0f - c;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    V3 col = v3(R.a[j], R.a[3 + j], R.a[6 + j]);
    V3 uc = cross(u, col);
    V3 uuc = cross(u, uc);
    col = col + s * uc + c1 * uuc;
    R.a[j] = col.x; R.a[3 + j] = col.y; R.a[6 + j] = col.z;
  }
}

// spatial inertia about O: mass, first moment h = m (c - O), rotational inertia about O (symmetric 6)
struct SpI { float m; V3 h; float xx, yy, zz, xy, xz, yz; };
LM_DEV SpI spi0() { SpI I; I.m = 0; I.h = v3(0, 0, 0); I.xx = I.yy = I.zz = I.xy = I.xz = I.yz = 0; return I; }
LM_DEV SpI operator+(const SpI& a, const SpI& b) {
  SpI I; I.m = a.m + b.m; I.h = a.h + b.h; I.xx = a.xx + b.xx; I.yy = a.yy + b.yy; I.zz = a.zz + b.zz;
  I.xy = a.xy + b.xy; I.xz = a.xz + b.xz; I.yz = a.yz + b.yz; return I;
}
LM_DEV V3 imul(const SpI& I, V3 w) {
  return v3(I.xx * w.x + I.xy * w.y + I.xz * w.z, I.xy * w.x + I.yy * w.y + I.yz * w.z, I.xz * w.x + I.yz * w.y + I.zz * w.z);
}
// momentum of inertia I under motion S: (angular about O, linear)
LM_DEV Sp apply(const SpI& I, Sp S) {
  Sp f; f.v = I.m * S.v + cross(S.w, I.h); f.w = imul(I, S.w) + cross(I.h, S.v); return f;
}
// body inertia (m, com offset r from O, world inertia about com as symmetric 6) -> spatial inertia about O
LM_DEV SpI make_spi(float m, V3 r, const float* Iw) {
  SpI I; I.m = m; I.h = m * r;
  float rr = dot(r, r);
  I.xx = Iw[0] + m * (rr - r.x * r.x); I.yy = Iw[1] + m * (rr - r.y * r.y); I.zz = Iw[2] + m * (rr - r.z * r.z);
  I.xy = Iw[3] - m * r.x * r.y; I.xz = Iw[4] - m * r.x * r.z; I.yz = Iw[5] - m * r.y * r.z;
  return I;
}
// R diag/sym(I_local) R^T as symmetric 6 (xx,yy,zz,xy,xz,yz)
LM_DEV void rotate_inertia(const M3& R, const float* Il, float* Iw) {
  float L[9] = {Il[0], Il[3], Il[4], Il[3], Il[1], Il[5], Il[4], Il[5], Il[2]};
  float T[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) T[3 * i + j] = R.a[3 * i] * L[j] + R.a[3 * i + 1] * L[3 + j] + R.a[3 * i + 2] * L[6 + j];
  auto e = [&](int i, int j) { return T[3 * i] * R.a[3 * j] + T[3 * i + 1] * R.a[3 * j + 1] + T[3 * i + 2] * R.a[3 * j + 2]; };
  Iw[0] = e(0, 0); Iw[1] = e(1, 1); Iw[2] = e(2, 2); Iw[3] = e(0, 1); Iw[4] = e(0, 2); Iw[5] = e(1, 2);
}

struct Params {
  float h;            // timestep
  V3 g;               // gravity
  int iterations;     // Newton iteration cap
  float tolerance;    // stop when scale*|grad| < tolerance (float32-appropriate)
  float scale;        // 1 / (meaninertia * nv)
  float ls_tol;       // line search: stop when |phi'| < ls_tol * |phi'(0)|
  int ls_iters;       // line search iteration cap
  float ls_noise;     // float32 noise floor of phi' relative to the sum of |terms|
  float ls_grid[3];   // four-point line search: first-round step lengths as fractions of the Newton step (besides 1)
  int ablate;         // profiling only: bitmask of solver regions to skip (0 in production)
  int nv;
  int integrator;     // LM_INT_EULER (0) | LM_INT_RK4 (1)
  int cone;           // 0 pyramidal | 1 elliptic
  int act_position;   // 1: the chain joints' actuators are position servos (torque = clamp(kp*ctrl - kp*q, force range))
  int off_runsup;     // collider-less geoms of the root body (constant-table tail, LM_H_OFF_RUNSUP; the chains' lists start at LM_C_OFF_*)
  const float* meshv; // hull vertices of the mesh colliders (global memory, 4 floats per vertex, link frame; the 4th: start of the
                      // vertex' neighbour list in meshn)
  const float* meshn; // hull-vertex graph: neighbour lists (indices into the geom's hull, nearest first, -1 ends a list)
  const float* gpt;   // geom-pair table (global memory): records of LM_GPAIR_SIZE floats, read when a BODY pair is within reach
  const float* meshadj; // adjacency blocks of the hull vertices (global memory, 4 floats per entry): hill climbing of the convex-pair collider
  const float* bpt;   // body-pair table (global memory): records of LM_BP_SIZE floats, read when a link pair is within reach
  const float* gt;    // geom table (global memory): full geom records [geom][field][chain], read when a geom is within reach of the floor
  const float* cmg;   // the constant table in GLOBAL memory: the six-link kernels read their link-pair lists from there (they do not fit
                      // beside the lane memory in the workgroup's LDS share: lowering.py ends H_CM_USED before them)
  int root_xyz;       // the root's translations are slides along +x, +y, +z in a root frame that is the world's (solve(): ROOT_XYZ)
  int root_limited;   // some root dof is `limited` (lowering.py keeps a root limit only when it can become active): the regular kernels of
                      // the families without root limit rows look at the root positions every pass (forward: ROOT_LIM)
};

struct Counters { int solver_iters; int overflow; int unhandled; int ncon; int ls_evals; int ls_capped; int it_max;
  int selfprox;      // forward passes x geom pairs without a collider (box / cylinder against something) within the margin
  int selfcon;       // self-contacts simulated, summed over the forward passes
  int natown;        // of those: contacts of box-box and capsule-box pairs — this code's OWN manifold construction, not the engine's case
                     // analysis (nat_box_box / nat_capsule_box below): counted so that a rollout says how much of it rests on them
  int pair_passes;   // forward passes in which the self-collision detection ran (diagnostics)
  int need_full;     // a convex pair came within reach in a kernel compiled WITHOUT the convex collider (PM == 2): the control step is
                     // abandoned and replayed by the full kernel (lm_step.h)
  int peak_slots, peak_q, peak_res;   // largest number of contact slots / queued convex pairs / pair results of this lane's chain in a pass (the
                     // replay kernel decides with them whether the environment fits the regular kernel again, lm_step.h)
  float grf[4][3];   // sums of the contact-frame force (normal, t1, t2) of the chain's foot-force groups (2; 4 in the six-link kernels)
#ifdef LM_TIMERS
  long long t[16];
  long long m[16];   // [8..]: passes with detection, passes, geom pairs tested of kind 0 / 1 / 2, hits of kind 0 / 1 / 2;  [0..7] convex collider: calls, ended at the one-direction test, no contact, contact, support pairs, hill steps, refinement iterations, rounds
#endif
};
#ifdef LM_TIMERS
#define LM_TICK(i) do { long long now_ = LM_CLOCK(); cnt.t[i] += now_ - tick_; tick_ = now_; } while (0)
#define LM_TICK_INIT() long long tick_ = LM_CLOCK()
#else
#define LM_TICK(i) do {} while (0)
#define LM_TICK_INIT() do {} while (0)
#endif

// optional stage dump for parity tests (one environment): written by the owning lanes
struct Debug {
  float* M;        // [nv*nv]
  float* bias; float* smooth; float* qacc_smooth; float* qacc; float* qfrc_constraint;   // [nv]
};

constexpr float kMinVal = 1e-15f;
LM_DEV int tri(int i, int j) { return i * (i + 1) / 2 + j; }   // lower-triangular index, j <= i

// LM_POW01(x, p): x^p for 0 < x <= 1 (solimp power other than the special-cased 1 and 2)
LM_DEV float impedance(const float* s /*clipped solimp[5]*/, int stride, float pos, float margin) {
  float s0 = s[0], s1 = s[stride], s2 = s[2 * stride], s3 = s[3 * stride], s4 = s[4 * stride];
  if (s0 == s1 || s2 <= kMinVal) return 0.5f * (s0 + s1);
  float x = fabsf((pos - margin) / s2);
  if (x >= 1.0f) return s1;
  if (x <= 0.0f) return s0;
  float y;
  if (s4 == 1.0f) y = x;
  else if (s4 == 2.0f) y = (x <= s3) ? x * x / s3 : 1.0f - (1.0f - x) * (1.0f - x) / (1.0f - s3);
  else if (x <= s3) y = LM_POW01(x, s4) / LM_POW01(s3, s4 - 1.0f);
  else y = 1.0f - LM_POW01(1.0f - x, s4) / LM_POW01(1.0f - s3, s4 - 1.0f);
  return s0 + y * (s1 - s0);
}

// ---- contact slot record (one floor contact of this lane's chain), kept in lane-local memory --------------
// On the GPU this is an LDS array indexed [field][lane] (stride = lanes per workgroup, conflict-free); slots are
// walked with ordinary loops so that only one contact's working set is in registers at a time.
enum { SL_LINK = 0, SL_DIM, SL_MU, SL_RX, SL_RY, SL_RZ, SL_D, SL_FR = SL_D + 6, SL_AREF = SL_FR + 5, SL_JAR = SL_AREF + 6,
       SL_JV = SL_JAR + 6, SL_ZONE = SL_JV + 6, SL_GRF /* force group of the chain (0/1) or -1 */,
       // line-search coefficients of an elliptic contact, prepared once per Newton iteration (the engine's PrimalPrepare): with
       // x(alpha) = jar + alpha jv the cone's normal part N = N0 + alpha Np and tangential norm T^2 = UU + 2 alpha UV + alpha^2 VV are
       // polynomials in alpha, the quadratic ("bottom") zone's derivative is A + alpha B: [N0, Np, UU, UV, VV, A, B, Dm, mu]
       SL_PREP, SL_SIZE = SL_PREP + 9 };
// pair extension of a slot record (kernels with self-collisions, PAIRS): SL_PART = 0 for a floor contact, else
// sign * (1 + partner lane * 8 + partner link), partner link 7 = the root body; sign = +1 when this lane's body carries the
// contact's SECOND geom (the normal points from geom 1 to geom 2). Then the unit normal, world axes.
enum { SL_PART = SL_SIZE, SL_NX, SL_NY, SL_NZ, SL_SIZE_PAIRS };       // (the tangents follow from the normal: make_frame)
// per-environment joint parameters (domain randomisation): replaces the table's damping / stiffness / frictionloss
// per-environment parameters of the kernels compiled with DR: joint damping / stiffness / frictionloss in registers, and the
// environment's MODEL VARIANT (lowering.variant_tables): `inr` = its inertial record [LM_IR_SIZE][LM_NCHAIN] in global memory
// (null: the batch has no variants), `gt` / `gpt` = its geom table and geom-pair table
template <int MC> struct DofPrm {
  float floss_r[6], floss_c[MC];                 // read inside the line search: registers
  const float* damp; const float* stiff; long long stride;   // damping / stiffness of dof d at [d * stride]: read where they are used
                                                             // (twice per forward pass) — 22 registers fewer to carry through the solver
  const float* inr; const float* gt; const float* gpt;
  float rfl_r[6], rfl_c[MC];       // friction-loss regularisers of the variant (read inside the line search: kept in registers)
  float* mprc;                     // the environment's warm-start cache of the convex collider: kMprCacheFloats floats per geom-pair record
                                   // (mpr_convex_pair), or null (set by every caller, DR or not)
};

// lane-memory map: [NS slot records][Mcc, Mcr, Mrr][root twists 6x6][chain twists MCx6][link images MCx6][link frames MCx18][muscle act NM][muscle ctrl NM][link bounding-sphere centres MCx3 (PAIRS)]
// compact slot record of the kernels compiled for condim-3 pyramids only (CONE == 0): one D, four edge rows — 21 floats
// instead of 37, which is what lets the 8-slot humanoid family keep four workgroups per CU (LDS: 160 KB / 4)
enum { SLC_D = SL_D, SLC_AREF = SLC_D + 1, SLC_JAR = SLC_AREF + 4, SLC_JV = SLC_JAR + 4, SLC_ZONE = SLC_JV + 4, SLC_GRF, SLC_SIZE,
       SLC_SIZE_PAIRS = SLC_SIZE + 4 };      // + the pair extension (SL_PART, normal) behind the compact record
template <int MC, int NS, int NM = 0, bool PAIRS = false, bool COMPACT = false> struct LaneMem {
  static constexpr int kSlot = COMPACT ? (PAIRS ? (int)SLC_SIZE_PAIRS : (int)SLC_SIZE) : (PAIRS ? (int)SL_SIZE_PAIRS : (int)SL_SIZE);
  static constexpr int kSlots = 0;
  static constexpr int kMcc = NS * kSlot;
  static constexpr int kMcr = kMcc + MC * (MC + 1) / 2;
  static constexpr int kMrr = kMcr + MC * 6;
  static constexpr int kSr = kMrr + 21;
  static constexpr int kSc = kSr + 36;
  static constexpr int kAl = kSc + MC * 6;       // link images of the current joint-space vector (MC x 6)
  static constexpr int kFrame = kAl + MC * 6;    // link frames for the collision pass: position 3, rotation 9, velocity 6
  static constexpr int kAct = kFrame + MC * 18;    // muscle activations of this lane's chain (NM), then their controls (NM)
  static constexpr int kCtrl = kAct + NM;
  static constexpr int kBS = kCtrl + NM;         // world centres of the links' bounding spheres (self-collision broad phase)
  // BIG kernels (NS > 8: the replay kernels that take over a control step in which a regular kernel ran out of contact slots,
  // lm_step.h): one slot per possible contact of the chain in any state the robots reach, and a queue / result list of the pair
  // pass of their own (the regular kernels keep theirs in the part of lane memory that holds M and the twists later in the pass)
  static constexpr bool kBig = NS > 8;
  static constexpr int kQCap = kBig ? 128 : ((MC >= 5) ? 24 : 8);    // convex pairs per chain and pass
  static constexpr int kRCap = kBig ? 64 : ((NS < 8) ? NS : 8);      // contacts per chain and pass
  // (the six-link kernels compute the sphere centres from the link frames, like the detection-only ones: 18 floats per column that —
  // with their prune records and link groups read from the table's copy in global memory — put the family back at FOUR workgroups per CU)
  static constexpr int kLists = kBS + ((PAIRS && MC != 6) ? MC * 3 : 0);
  // the ROOT twists of the kernels that build their inertias behind the pair pass (forward: DEFER): the root is replicated in the four
  // chain lanes of an environment, so its 36 numbers are STRIPED over the four columns (element i in column i & 3, field i >> 2) —
  // 9 floats per column outside the part of lane memory the pair pass's work lists overlay, written before the pass
  static constexpr bool kStripe = PAIRS && MC >= 5;
  static constexpr int kRootS = kLists + ((PAIRS && kBig) ? kQCap + 8 * kRCap : 0);
  static constexpr int kSize = kRootS + (kStripe ? 9 : 0);
  // device layout: lanes are grouped by 16 ([field][16 lanes] per group, so every field offset is a compile-time
  // constant = an immediate in the ds_read/ds_write); kGroup = floats per group, its kSize part padded to 1 mod 4
  // so that the four groups of a wave start 16 banks apart
  static constexpr int kPadded = kSize + ((5 - kSize % 4) % 4);
  static constexpr int kGroup = kPadded * 16;
};
// the lane memory of a kernel compiled for cone CONE (-1 run-time, 0 pyramids of condim 3 only, 1 elliptic)
template <int MC, int NS, int NM, bool PAIRS, int CONE> using LaneMemFor = LaneMem<MC, NS, NM, PAIRS, (CONE == 0)>;

// Elliptic-cone contact: cost/force/Hessian in the contact frame at jar[0..5] (rows beyond dim are ignored
// because their D is 0). Dj = D of row j, fr = friction coefficients of rows 1..5, mu = regularised cone mu.
// zone: 0 satisfied, 1 quadratic, 2 cone
struct ConeEval { int zone; float cost; float f[6]; };

template <bool WANT_FORCE>
LM_DEV ConeEval cone_eval(const float* jar, const float* Dj, const float* fr, float mu, int dim) {
  ConeEval e; e.cost = 0; e.zone = 0;
#pragma unroll
  for (int j = 0; j < 6; j++) e.f[j] = 0;
  float N = jar[0] * mu, T2 = 0, U[6];
  U[0] = N;
#pragma unroll
  for (int j = 1; j < 6; j++) { U[j] = (j < dim) ? jar[j] * fr[j - 1] : 0.0f; T2 = fmaf(U[j], U[j], T2); }
  float T = sqrtf(T2);
  if (N >= mu * T || (T <= 0 && N >= 0)) return e;
  if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
    e.zone = 1;
#pragma unroll
    for (int j = 0; j < 6; j++) if (j < dim) { e.cost = fmaf(0.5f * Dj[j] * jar[j], jar[j], e.cost); if (WANT_FORCE) e.f[j] = -Dj[j] * jar[j]; }
    return e;
  }
  e.zone = 2;
  float Dm = Dj[0] / fmaxf(kMinVal, mu * mu * (1.0f + mu * mu));
  float NmT = N - mu * T;
  e.cost = 0.5f * Dm * NmT * NmT;
  if (WANT_FORCE) {
    e.f[0] = -Dm * NmT * mu;
    float k = -e.f[0] / T;
#pragma unroll
    for (int j = 1; j < 6; j++) if (j < dim) e.f[j] = k * U[j] * fr[j - 1];
  }
  return e;
}

// 6x6 symmetric Hessian (lower tri, 21) of the cone cost w.r.t. jar
LM_DEV void cone_hessian(const float* jar, const float* Dj, const float* fr, float mu, int dim, int zone, float* Hc) {
#pragma unroll
  for (int i = 0; i < 21; i++) Hc[i] = 0;
  if (zone == 1) {
#pragma unroll
    for (int j = 0; j < 6; j++) if (j < dim) Hc[tri(j, j)] = Dj[j];
    return;
  }
  if (zone != 2) return;
  float Sc[6], U[6], T2 = 0;
  Sc[0] = mu; U[0] = jar[0] * mu;
#pragma unroll
  for (int j = 1; j < 6; j++) { Sc[j] = (j < dim) ? fr[j - 1] : 0.0f; U[j] = jar[j] * Sc[j]; T2 = fmaf(U[j], U[j], T2); }
  float T = sqrtf(T2), iT = 1.0f / T;
  float Dm = Dj[0] / fmaxf(kMinVal, mu * mu * (1.0f + mu * mu)), g = U[0] - mu * T;
  Hc[0] = Dm * Sc[0] * Sc[0];
#pragma unroll
  for (int j = 1; j < 6; j++) Hc[tri(j, 0)] = -Dm * mu * U[j] * iT * Sc[j] * Sc[0];
#pragma unroll
  for (int j = 1; j < 6; j++)
#pragma unroll
    for (int k = 1; k <= j; k++) {
      float tt = U[j] * U[k] * iT * iT;
      Hc[tri(j, k)] = (Dm * mu * mu * tt - Dm * g * mu * ((j == k ? 1.0f : 0.0f) - tt) * iT) * Sc[j] * Sc[k];
    }
}

// first and second derivative of the cone cost along jar + alpha*jv
LM_DEV void cone_line(const float* jar, const float* jv, float alpha, const float* Dj, const float* fr, float mu, int dim,
                      float& d1, float& d2) {
  float x0 = fmaf(alpha, jv[0], jar[0]);
  float N = x0 * mu, Np = jv[0] * mu, UU = 0, UV = 0, VV = 0;
#pragma unroll
  for (int j = 1; j < 6; j++) if (j < dim) {
    float u = fmaf(alpha, jv[j], jar[j]) * fr[j - 1], v = jv[j] * fr[j - 1];
    UU = fmaf(u, u, UU); UV = fmaf(u, v, UV); VV = fmaf(v, v, VV);
  }
  float T = sqrtf(UU);
  if (N >= mu * T || (T <= 0 && N >= 0)) return;
  if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
#pragma unroll
    for (int j = 0; j < 6; j++) if (j < dim) { float xj = fmaf(alpha, jv[j], jar[j]); d1 = fmaf(Dj[j] * xj, jv[j], d1); d2 = fmaf(Dj[j] * jv[j], jv[j], d2); }
    return;
  }
  float Dm = Dj[0] / fmaxf(kMinVal, mu * mu * (1.0f + mu * mu));
  float Tp = UV / T, Tpp = VV / T - UV * UV / (T * T * T);
  float NmT = N - mu * T, NmTp = Np - mu * Tp;
  d1 = fmaf(Dm * NmT, NmTp, d1);
  d2 += Dm * (NmTp * NmTp - NmT * mu * Tpp);
}

// ---- pyramidal friction cone (condim 3): 4 edge rows x_r = j_n +- mu j_t1 | j_n +- mu j_t2, all with the same D.
// A row is active (quadratic) iff x_r < 0; cost 0.5 D x_r^2, edge force -D x_r >= 0.
LM_DEV void pyr_rows(const float* j3, float mu, float* x) {
  x[0] = fmaf(mu, j3[1], j3[0]); x[1] = fmaf(-mu, j3[1], j3[0]); x[2] = fmaf(mu, j3[2], j3[0]); x[3] = fmaf(-mu, j3[2], j3[0]);
}
// contact-frame force (n, t1, t2) and active mask from the 4 row residuals
LM_DEV unsigned pyr_force(const float* x, float D, float mu, float* f3, float& cost) {
  float e[4]; unsigned act = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) { float xm = fminf(x[r], 0.0f); e[r] = -D * xm; cost = fmaf(0.5f * D * xm, xm, cost); if (x[r] < 0.0f) act |= 1u << r; }
  f3[0] = (e[0] + e[1]) + (e[2] + e[3]); f3[1] = mu * (e[0] - e[1]); f3[2] = mu * (e[2] - e[3]);
  return act;
}
// 3x3 contact-frame Hessian as the leading block of a 6x6 lower triangle
LM_DEV void pyr_hessian(unsigned act, float D, float mu, float* Hc) {
#pragma unroll
  for (int i = 0; i < 21; i++) Hc[i] = 0;
  float a0 = (act & 1u) ? 1.0f : 0.0f, a1 = (act & 2u) ? 1.0f : 0.0f, a2 = (act & 4u) ? 1.0f : 0.0f, a3 = (act & 8u) ? 1.0f : 0.0f;
  Hc[tri(0, 0)] = D * ((a0 + a1) + (a2 + a3));
  Hc[tri(1, 0)] = D * mu * (a0 - a1); Hc[tri(2, 0)] = D * mu * (a2 - a3);
  Hc[tri(1, 1)] = D * mu * mu * (a0 + a1); Hc[tri(2, 2)] = D * mu * mu * (a2 + a3);
}

// ---- arrow-structured factorisation -------------------------------------------------------------------------
// H = [Hcc (MC x MC, per lane), Hcr (MC x 6, per lane); Hrr (6x6): `Hrr_rep` replicated part + quad-sum of
// `Hrr_part`]. Overwrites: Hcc -> Lcc (lower Cholesky), Hcr -> W = Lcc^-1 Hcr, Lrr <- chol(Hrr - sum W^T W).
template <class Q, int MC>
LM_DEV void arrow_factor(float* Hcc, float (*Hcr)[6], const float* Hrr_rep, const float* Hrr_part, float* Lrr) {
#pragma unroll
  for (int j = 0; j < MC; j++) {
    float s = Hcc[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) s = fmaf(-Hcc[tri(j, k)], Hcc[tri(j, k)], s);
    float d = sqrtf(fmaxf(s, 1e-30f)), id = 1.0f / d;
    Hcc[tri(j, j)] = d;
#pragma unroll
    for (int i = j + 1; i < MC; i++) {
      float t = Hcc[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) t = fmaf(-Hcc[tri(i, k)], Hcc[tri(j, k)], t);
      Hcc[tri(i, j)] = t * id;
    }
  }
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int i = 0; i < MC; i++) {
      float t = Hcr[i][r];
#pragma unroll
      for (int k = 0; k < i; k++) t = fmaf(-Hcc[tri(i, k)], Hcr[k][r], t);
      Hcr[i][r] = t / Hcc[tri(i, i)];
    }
  float S[21];
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int b = 0; b <= a; b++) {
      float t = Hrr_part[tri(a, b)];
#pragma unroll
      for (int k = 0; k < MC; k++) t = fmaf(-Hcr[k][a], Hcr[k][b], t);
      S[tri(a, b)] = Q::sum(t) + Hrr_rep[tri(a, b)];
    }
#pragma unroll
  for (int j = 0; j < 6; j++) {
    float s = S[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) s = fmaf(-Lrr[tri(j, k)], Lrr[tri(j, k)], s);
    float d = sqrtf(fmaxf(s, 1e-30f)), id = 1.0f / d;
    Lrr[tri(j, j)] = d;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      float t = S[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) t = fmaf(-Lrr[tri(i, k)], Lrr[tri(j, k)], t);
      Lrr[tri(i, j)] = t * id;
    }
  }
}

// solve H x = g in place (xc chain part, xr root part) with the factors from arrow_factor
template <class Q, int MC>
LM_DEV void arrow_solve(const float* Lcc, const float (*W)[6], const float* Lrr, float* xc, float* xr) {
#pragma unroll
  for (int i = 0; i < MC; i++) {
    float t = xc[i];
#pragma unroll
    for (int k = 0; k < i; k++) t = fmaf(-Lcc[tri(i, k)], xc[k], t);
    xc[i] = t / Lcc[tri(i, i)];
  }
#pragma unroll
  for (int r = 0; r < 6; r++) {
    float t = 0;
#pragma unroll
    for (int k = 0; k < MC; k++) t = fmaf(W[k][r], xc[k], t);
    xr[r] -= Q::sum(t);
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float t = xr[i];
#pragma unroll
    for (int k = 0; k < i; k++) t = fmaf(-Lrr[tri(i, k)], xr[k], t);
    xr[i] = t / Lrr[tri(i, i)];
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    float t = xr[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) t = fmaf(-Lrr[tri(k, i)], xr[k], t);
    xr[i] = t / Lrr[tri(i, i)];
  }
#pragma unroll
  for (int i = 0; i < MC; i++) {
#pragma unroll
    for (int r = 0; r < 6; r++) xc[i] = fmaf(-W[i][r], xr[r], xc[i]);
  }
#pragma unroll
  for (int i = MC - 1; i >= 0; i--) {
    float t = xc[i];
#pragma unroll
    for (int k = i + 1; k < MC; k++) t = fmaf(-Lcc[tri(k, i)], xc[k], t);
    xc[i] = t / Lcc[tri(i, i)];
  }
}

// ---- two chains that share their first link (a torso joint with an arm chain on either side: lowering.py, C_DUPROLE) ----------
// Both lanes carry the link — the owner (role +1) with its mass, joint parameters and geoms, the other (role -1) a massless copy —
// so that every lane's chain stays serial and the matrices keep their arrow structure. The two copies x_a, x_b of the shared
// dof are ONE coordinate: every solve H x = r becomes the equality-constrained one (c = e_a - e_b):
//     H x + c lambda = r,  c.x = 0   =>   x = H^-1 r - lambda H^-1 c,  lambda = (c . H^-1 r) / (c . H^-1 c)
// i.e. a second triangular solve with the same factors. In the independent coordinates that is exactly (P^T H P)^-1 P^T r.
template <class Q, int MC>
LM_DEV void tie_shared_dof(const float* Lcc, const float (*W)[6], const float* Lrr, float* xc, float* xr, int role) {
  float zc[MC], zr[6];
#pragma unroll
  for (int k = 0; k < MC; k++) zc[k] = 0.0f;
#pragma unroll
  for (int i = 0; i < 6; i++) zr[i] = 0.0f;
  zc[0] = (float)role;
  arrow_solve<Q, MC>(Lcc, W, Lrr, zc, zr);
  const float cx = Q::sum((float)role * xc[0]), cz = Q::sum((float)role * zc[0]);
  const float lam = cx / cz;
#pragma unroll
  for (int k = 0; k < MC; k++) xc[k] = fmaf(-lam, zc[k], xc[k]);
#pragma unroll
  for (int i = 0; i < 6; i++) xr[i] = fmaf(-lam, zr[i], xr[i]);
  const float xa = Q::sum(role > 0 ? xc[0] : 0.0f);      // bit-identical copies from here on
  if (role < 0) xc[0] = xa;
}

// ---- arrow factorisation with ONE cross block per lane (self-collisions between two chains) --------------------------
// A contact between links of chains a < b couples their blocks: H_ab = X (MC x MC, kept by lane a). Elimination order
// a, b, (other chains), root:  Y = L_a^-1 X;  H_bb -= Y^T Y;  H_br -= Y^T W_a;  then b factors what is left. `role`: 0 = no
// cross block on this lane, 1 = lane a (X: in the cross block, out Y), 2 = lane b (X: out the partner's Y); `partner` = the
// other lane of the pair. `any_cross` is quad-uniform; without it this is arrow_factor. The lanes of a quad run the same
// instructions: the second pass (lane b) costs every lane of the quad one more chain-block factorisation.
template <class Q, int MC>
LM_DEV void arrow_factor_x(float* Hcc, float (*Hcr)[6], const float* Hrr_rep, const float* Hrr_part, float* Lrr,
                           float (*X)[MC], int role, int partner, bool any_cross) {
  for (int pass = 0; pass < (any_cross ? 2 : 1); pass++) {
    if (pass == 1) {
      float Ya[MC][MC], Wa[MC][6];
#pragma unroll
      for (int k = 0; k < MC; k++) {
#pragma unroll
        for (int j = 0; j < MC; j++) Ya[k][j] = Q::quad_read(X[k][j], partner);
#pragma unroll
        for (int r = 0; r < 6; r++) Wa[k][r] = Q::quad_read(Hcr[k][r], partner);
      }
      if (role == 2) {
#pragma unroll
        for (int i = 0; i < MC; i++) {
#pragma unroll
          for (int j = 0; j <= i; j++) {
            float t = Hcc[tri(i, j)];
#pragma unroll
            for (int k = 0; k < MC; k++) t = fmaf(-Ya[k][i], Ya[k][j], t);
            Hcc[tri(i, j)] = t;
          }
#pragma unroll
          for (int r = 0; r < 6; r++) {
            float t = Hcr[i][r];
#pragma unroll
            for (int k = 0; k < MC; k++) t = fmaf(-Ya[k][i], Wa[k][r], t);
            Hcr[i][r] = t;
          }
        }
#pragma unroll
        for (int k = 0; k < MC; k++)
#pragma unroll
          for (int j = 0; j < MC; j++) X[k][j] = Ya[k][j];
      }
    }
    const bool mine = (role == 2) ? (pass == 1) : (pass == 0);
    if (mine) {
#pragma unroll
      for (int j = 0; j < MC; j++) {
        float sd = Hcc[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; k++) sd = fmaf(-Hcc[tri(j, k)], Hcc[tri(j, k)], sd);
        float d = sqrtf(fmaxf(sd, 1e-30f)), id = 1.0f / d;
        Hcc[tri(j, j)] = d;
#pragma unroll
        for (int i = j + 1; i < MC; i++) {
          float t = Hcc[tri(i, j)];
#pragma unroll
          for (int k = 0; k < j; k++) t = fmaf(-Hcc[tri(i, k)], Hcc[tri(j, k)], t);
          Hcc[tri(i, j)] = t * id;
        }
      }
#pragma unroll
      for (int i = 0; i < MC; i++) {
        const float idg = 1.0f / Hcc[tri(i, i)];
#pragma unroll
        for (int r = 0; r < 6; r++) {
          float t = Hcr[i][r];
#pragma unroll
          for (int k = 0; k < i; k++) t = fmaf(-Hcc[tri(i, k)], Hcr[k][r], t);
          Hcr[i][r] = t * idg;
        }
        if (role == 1) {
#pragma unroll
          for (int j = 0; j < MC; j++) {
            float t = X[i][j];
#pragma unroll
            for (int k = 0; k < i; k++) t = fmaf(-Hcc[tri(i, k)], X[k][j], t);
            X[i][j] = t * idg;
          }
        }
      }
    }
  }
  float S[21];
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int b = 0; b <= a; b++) {
      float t = Hrr_part[tri(a, b)];
#pragma unroll
      for (int k = 0; k < MC; k++) t = fmaf(-Hcr[k][a], Hcr[k][b], t);
      S[tri(a, b)] = Q::sum(t) + Hrr_rep[tri(a, b)];
    }
#pragma unroll
  for (int j = 0; j < 6; j++) {
    float sd = S[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) sd = fmaf(-Lrr[tri(j, k)], Lrr[tri(j, k)], sd);
    float d = sqrtf(fmaxf(sd, 1e-30f)), id = 1.0f / d;
    Lrr[tri(j, j)] = d;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      float t = S[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) t = fmaf(-Lrr[tri(i, k)], Lrr[tri(j, k)], t);
      Lrr[tri(i, j)] = t * id;
    }
  }
}

// solve with the factors of arrow_factor_x (Y in X for both lanes of a pair)
template <class Q, int MC>
LM_DEV void arrow_solve_x(const float* Lcc, const float (*W)[6], const float* Lrr, const float (*Y)[MC], int role, int partner,
                          bool any_cross, float* xc, float* xr) {
  // forward: a (and the chains without a cross block), then b with its right-hand side reduced by Y^T y_a
  for (int pass = 0; pass < (any_cross ? 2 : 1); pass++) {
    if (pass == 1) {
      float t[MC];
#pragma unroll
      for (int k = 0; k < MC; k++) t[k] = Q::quad_read(xc[k], partner);
      if (role == 2) {
#pragma unroll
        for (int i = 0; i < MC; i++)
#pragma unroll
          for (int k = 0; k < MC; k++) xc[i] = fmaf(-Y[k][i], t[k], xc[i]);
      }
    }
    const bool mine = (role == 2) ? (pass == 1) : (pass == 0);
    if (mine) {
#pragma unroll
      for (int i = 0; i < MC; i++) {
        float t = xc[i];
#pragma unroll
        for (int k = 0; k < i; k++) t = fmaf(-Lcc[tri(i, k)], xc[k], t);
        xc[i] = t / Lcc[tri(i, i)];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 6; r++) {
    float t = 0;
#pragma unroll
    for (int k = 0; k < MC; k++) t = fmaf(W[k][r], xc[k], t);
    xr[r] -= Q::sum(t);
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float t = xr[i];
#pragma unroll
    for (int k = 0; k < i; k++) t = fmaf(-Lrr[tri(i, k)], xr[k], t);
    xr[i] = t / Lrr[tri(i, i)];
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    float t = xr[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) t = fmaf(-Lrr[tri(k, i)], xr[k], t);
    xr[i] = t / Lrr[tri(i, i)];
  }
  // backward: b (and the chains without a cross block), then a with Y x_b taken off
  for (int pass = 0; pass < (any_cross ? 2 : 1); pass++) {
    float t[MC];
#pragma unroll
    for (int k = 0; k < MC; k++) t[k] = 0.0f;
    if (pass == 1) {
#pragma unroll
      for (int k = 0; k < MC; k++) t[k] = Q::quad_read(xc[k], partner);
    }
    const bool mine = (role == 1) ? (pass == 1) : (pass == 0);
    if (mine) {
#pragma unroll
      for (int i = 0; i < MC; i++) {
#pragma unroll
        for (int r = 0; r < 6; r++) xc[i] = fmaf(-W[i][r], xr[r], xc[i]);
        if (role == 1) {
#pragma unroll
          for (int j = 0; j < MC; j++) xc[i] = fmaf(-Y[i][j], t[j], xc[i]);
        }
      }
#pragma unroll
      for (int i = MC - 1; i >= 0; i--) {
        float tt = xc[i];
#pragma unroll
        for (int k = i + 1; k < MC; k++) tt = fmaf(-Lcc[tri(k, i)], xc[k], tt);
        xc[i] = tt / Lcc[tri(i, i)];
      }
    }
  }
}

// ---- arrow factorisation with cross blocks between ANY chains of the quad (self-collisions) --------------------------------
// A contact between links of chains a < b couples their blocks: H_ab = X (MC x MC), kept by the LOWER lane a in slot b - a - 1 of
// its X array. The lanes are eliminated in order 0, 1, ...: lane a factors what is left of its block, Y_ab = L_a^-1 X_ab for
// every lane b above it, and the lanes above take off Y_ab^T Y_ab (own block), Y_ab^T W_a (root coupling) and — when a is coupled
// with two lanes b < d — the fill-in Y_ab^T Y_ad on X_bd. `adj` = the quad's coupling bits (bit 4 i + j: chains i and j are
// coupled; quad-uniform, fill-in added here), so uncoupled pairs cost nothing. Afterwards every lane holds the Y blocks of ALL its
// partners: the upper ones where their X was, a copy of the lower partner p's in slot NX - 1 - p, and the solve only exchanges
// vectors. Exact for any coupling pattern (a folded-up humanoid couples all three of its chains); NX >= number of chains - 1.
// The loops over lanes / cross-block slots: rolled (run-time slot indices: X lives in private memory = scratch) in the kernels with three
// cross-block slots; UNROLLED for the humanoids' two (round 6: every index of X is a compile-time constant, the 50 floats are registers —
// the cross blocks were 50 scratch dwords read and written inside the Hessian's slot loop and the factorisation of every coupled solve)
#ifdef LM_ARROW_ROLLED
#define LM_ARROW_LOOP _Pragma("nounroll")
#else
#ifdef LM_ARROW_SIX_ROLLED          // (A/B probe: the six-link kernels' loops rolled, as before the end of round 6)
#define LM_ARROW_LOOP _Pragma("unroll (NX <= 2 ? 4 : 1)")
#else
#define LM_ARROW_LOOP _Pragma("unroll ((NX <= 2 || MC == 6) ? 4 : 1)")
#endif
#endif
template <class Q, int MC, int NX>
LM_DEV void arrow_factor_g(float* Hcc, float (*Hcr)[6], const float* Hrr_rep, const float* Hrr_part, float* Lrr,
                           float (*X)[MC][MC], int c, int& adj) {
LM_ARROW_LOOP
  for (int a = 0; a < NX + 1; a++) {
    const int row = (adj >> (4 * a)) & 15;
    if (c == a || (a == NX && c > NX)) {              // (a lane beyond the last one that can hold a cross block: plain factorisation)
#pragma unroll
      for (int j = 0; j < MC; j++) {
        float sd = Hcc[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; k++) sd = fmaf(-Hcc[tri(j, k)], Hcc[tri(j, k)], sd);
        float d = sqrtf(fmaxf(sd, 1e-30f)), id = 1.0f / d;
        Hcc[tri(j, j)] = d;
#pragma unroll
        for (int i = j + 1; i < MC; i++) {
          float t = Hcc[tri(i, j)];
#pragma unroll
          for (int k = 0; k < j; k++) t = fmaf(-Hcc[tri(i, k)], Hcc[tri(j, k)], t);
          Hcc[tri(i, j)] = t * id;
        }
      }
#pragma unroll
      for (int i = 0; i < MC; i++) {
        const float idg = 1.0f / Hcc[tri(i, i)];
#pragma unroll
        for (int r = 0; r < 6; r++) {
          float t = Hcr[i][r];
#pragma unroll
          for (int k = 0; k < i; k++) t = fmaf(-Hcc[tri(i, k)], Hcr[k][r], t);
          Hcr[i][r] = t * idg;
        }
#pragma unroll
        for (int x = 0; x < NX; x++) if (x < NX - a && c == a) {      // my blocks with the lanes ABOVE me (the other slots hold copies from below)
#pragma unroll
          for (int j = 0; j < MC; j++) {
            float t = X[x][i][j];
#pragma unroll
            for (int k = 0; k < i; k++) t = fmaf(-Hcc[tri(i, k)], X[x][k][j], t);
            X[x][i][j] = t * idg;
          }
        }
      }
    }
    if ((row >> (a + 1)) == 0) continue;              // lane a is coupled with no lane above it (quad-uniform)
    float Wa[MC][6];
#pragma unroll
    for (int k = 0; k < MC; k++)
#pragma unroll
      for (int r = 0; r < 6; r++) Wa[k][r] = Q::quad_read(Hcr[k][r], a);
LM_ARROW_LOOP
    for (int x1 = 0; x1 < NX; x1++) {
      const int b = a + 1 + x1;
      if (b > NX || !((row >> b) & 1)) continue;
      float Y1[MC][MC];
#pragma unroll
      for (int k = 0; k < MC; k++)
#pragma unroll
        for (int j = 0; j < MC; j++) Y1[k][j] = Q::quad_read(X[x1][k][j], a);
      if (c == b) {
#pragma unroll
        for (int i = 0; i < MC; i++) {
#pragma unroll
          for (int j = 0; j <= i; j++) {
            float t = Hcc[tri(i, j)];
#pragma unroll
            for (int k = 0; k < MC; k++) t = fmaf(-Y1[k][i], Y1[k][j], t);
            Hcc[tri(i, j)] = t;
          }
#pragma unroll
          for (int r = 0; r < 6; r++) {
            float t = Hcr[i][r];
#pragma unroll
            for (int k = 0; k < MC; k++) t = fmaf(-Y1[k][i], Wa[k][r], t);
            Hcr[i][r] = t;
          }
        }
        // keep a copy of the lower partner's Y for the solves
#pragma unroll
        for (int k = 0; k < MC; k++)
#pragma unroll
          for (int j = 0; j < MC; j++) X[NX - 1 - a][k][j] = Y1[k][j];
      }
LM_ARROW_LOOP
      for (int x2 = x1 + 1; x2 < NX; x2++) {
        const int d = a + 1 + x2;
        if (d > NX || !((row >> d) & 1)) continue;
        float Y2[MC][MC];
#pragma unroll
        for (int k = 0; k < MC; k++)
#pragma unroll
          for (int j = 0; j < MC; j++) Y2[k][j] = Q::quad_read(X[x2][k][j], a);
        if (c == b) {                              // fill-in on X_bd (slot d - b - 1 of lane b)
#pragma unroll
          for (int i = 0; i < MC; i++)
#pragma unroll
            for (int j = 0; j < MC; j++) {
              float t = 0.0f;
#pragma unroll
              for (int k = 0; k < MC; k++) t = fmaf(Y1[k][i], Y2[k][j], t);
#pragma unroll
              for (int xs = 0; xs < NX; xs++) if (xs == d - b - 1) X[xs][i][j] -= t;
            }
        }
        adj |= (1 << (4 * b + d)) | (1 << (4 * d + b));
      }
    }
  }
  float S[21];
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int b = 0; b <= a; b++) {
      float t = Hrr_part[tri(a, b)];
#pragma unroll
      for (int k = 0; k < MC; k++) t = fmaf(-Hcr[k][a], Hcr[k][b], t);
      S[tri(a, b)] = Q::sum(t) + Hrr_rep[tri(a, b)];
    }
#pragma unroll
  for (int j = 0; j < 6; j++) {
    float sd = S[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) sd = fmaf(-Lrr[tri(j, k)], Lrr[tri(j, k)], sd);
    float d = sqrtf(fmaxf(sd, 1e-30f)), id = 1.0f / d;
    Lrr[tri(j, j)] = d;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      float t = S[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) t = fmaf(-Lrr[tri(i, k)], Lrr[tri(j, k)], t);
      Lrr[tri(i, j)] = t * id;
    }
  }
}

// solve with the factors of arrow_factor_g (`adj` as it left the coupling bits, fill-in included)
template <class Q, int MC, int NX>
LM_DEV void arrow_solve_g(const float* Lcc, const float (*W)[6], const float* Lrr, const float (*Y)[MC][MC], int c, int adj,
                          float* xc, float* xr) {
  // forward, lanes in order: y_a = L_a^-1 g_a, then every coupled lane b above takes Y_ab^T y_a off its right-hand side
LM_ARROW_LOOP
  for (int a = 0; a < NX + 1; a++) {
    if (c == a) {
#pragma unroll
      for (int i = 0; i < MC; i++) {
        float t = xc[i];
#pragma unroll
        for (int k = 0; k < i; k++) t = fmaf(-Lcc[tri(i, k)], xc[k], t);
        xc[i] = t / Lcc[tri(i, i)];
      }
    }
    const int row = (adj >> (4 * a)) & 15;
    if ((row >> (a + 1)) == 0) continue;
    float t[MC];
#pragma unroll
    for (int k = 0; k < MC; k++) t[k] = Q::quad_read(xc[k], a);
    if (c > a && ((row >> c) & 1)) {
#pragma unroll
      for (int xs = 0; xs < NX; xs++) if (xs == NX - 1 - a) {
#pragma unroll
        for (int i = 0; i < MC; i++)
#pragma unroll
          for (int k = 0; k < MC; k++) xc[i] = fmaf(-Y[xs][k][i], t[k], xc[i]);
      }
    }
  }
  // lanes NX + 1 .. 3 (if any) hold no cross block: their forward solve
  if (c > NX) {
#pragma unroll
    for (int i = 0; i < MC; i++) {
      float t = xc[i];
#pragma unroll
      for (int k = 0; k < i; k++) t = fmaf(-Lcc[tri(i, k)], xc[k], t);
      xc[i] = t / Lcc[tri(i, i)];
    }
  }
#pragma unroll
  for (int r = 0; r < 6; r++) {
    float t = 0;
#pragma unroll
    for (int k = 0; k < MC; k++) t = fmaf(W[k][r], xc[k], t);
    xr[r] -= Q::sum(t);
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float t = xr[i];
#pragma unroll
    for (int k = 0; k < i; k++) t = fmaf(-Lrr[tri(i, k)], xr[k], t);
    xr[i] = t / Lrr[tri(i, i)];
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    float t = xr[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) t = fmaf(-Lrr[tri(k, i)], xr[k], t);
    xr[i] = t / Lrr[tri(i, i)];
  }
  // backward, lanes in reverse order: x_a = L_a^-T (y_a - W_a x_r - sum over coupled lanes b above of Y_ab x_b)
  if (c > NX) {
#pragma unroll
    for (int i = 0; i < MC; i++)
#pragma unroll
      for (int r = 0; r < 6; r++) xc[i] = fmaf(-W[i][r], xr[r], xc[i]);
#pragma unroll
    for (int i = MC - 1; i >= 0; i--) {
      float tt = xc[i];
#pragma unroll
      for (int k = i + 1; k < MC; k++) tt = fmaf(-Lcc[tri(k, i)], xc[k], tt);
      xc[i] = tt / Lcc[tri(i, i)];
    }
  }
LM_ARROW_LOOP
  for (int a = NX; a >= 0; a--) {
    if (c == a) {
#pragma unroll
      for (int i = 0; i < MC; i++) {
#pragma unroll
        for (int r = 0; r < 6; r++) xc[i] = fmaf(-W[i][r], xr[r], xc[i]);
      }
#pragma unroll
      for (int i = MC - 1; i >= 0; i--) {
        float tt = xc[i];
#pragma unroll
        for (int k = i + 1; k < MC; k++) tt = fmaf(-Lcc[tri(k, i)], xc[k], tt);
        xc[i] = tt / Lcc[tri(i, i)];
      }
    }
    // lane a's solution goes to the coupled lanes BELOW it: they take Y_pa x_a off before their own back substitution
    const int row = (adj >> (4 * a)) & 15;
    if ((row & ((1 << a) - 1)) == 0) continue;
    float t[MC];
#pragma unroll
    for (int k = 0; k < MC; k++) t[k] = Q::quad_read(xc[k], a);
    if (c < a && ((row >> c) & 1)) {
#pragma unroll
      for (int xs = 0; xs < NX; xs++) if (xs == a - c - 1) {
#pragma unroll
        for (int i = 0; i < MC; i++)
#pragma unroll
          for (int j = 0; j < MC; j++) xc[i] = fmaf(-Y[xs][i][j], t[j], xc[i]);
      }
    }
  }
}

// contact-frame components (n=+z, t1=+y, t2=-x; then the same for rotation) of motion S at r
LM_DEV void contact_rows(Sp S, V3 r, float* out) {
  V3 u = S.v + cross(S.w, r);
  out[0] = u.z; out[1] = u.y; out[2] = -u.x; out[3] = S.w.z; out[4] = S.w.y; out[5] = -S.w.x;
}
// spatial force about O of contact-frame force f[6] applied at r
LM_DEV Sp contact_wrench(const float* f, V3 r) {
  V3 lin = v3(-f[2], f[1], f[0]), tor = v3(-f[5], f[4], f[3]);
  Sp F; F.v = lin; F.w = tor + cross(r, lin); return F;
}

// the same for a contact frame (n, t1, t2) in general position (self-collisions)
LM_DEV void frame_rows(Sp S, V3 r, V3 n, V3 t1, V3 t2, float* out) {
  V3 u = S.v + cross(S.w, r);
  out[0] = dot(n, u); out[1] = dot(t1, u); out[2] = dot(t2, u); out[3] = dot(n, S.w); out[4] = dot(t1, S.w); out[5] = dot(t2, S.w);
}
LM_DEV Sp frame_wrench(const float* f, V3 r, V3 n, V3 t1, V3 t2) {
  V3 lin = f[0] * n + f[1] * t1 + f[2] * t2, tor = f[3] * n + f[4] * t1 + f[5] * t2;
  Sp F; F.v = lin; F.w = tor + cross(r, lin); return F;
}
// tangents of a contact frame from its unit normal (engine: mju_makeFrame): the second axis starts from +y, or +z when the
// normal is within 60 degrees of +-y, and is made orthogonal to the normal; third = first x second
LM_DEV void make_frame(V3 n, V3& t1, V3& t2) {
  V3 y = (n.y < 0.5f && n.y > -0.5f) ? v3(0, 1, 0) : v3(0, 0, 1);
  y = y + (-dot(n, y)) * n;
  t1 = (1.0f / sqrtf(fmaxf(dot(y, y), 1e-30f))) * y;
  t2 = cross(n, t1);
}
// closest points of two segments p1 + s d1 (|s| <= h1), p2 + t d2 (|t| <= h2), unit directions (a sphere is h = 0)
LM_DEV void segment_closest(V3 p1, V3 d1, float h1, V3 p2, V3 d2, float h2, float& s_out, float& t_out) {
  const V3 r = p1 - p2;
  const float b = dot(d1, d2), cc = dot(d1, r), f = dot(d2, r), den = 1.0f - b * b;
  float s = 0.0f;
  if (den > 1e-12f) s = fminf(fmaxf((b * f - cc) / den, -h1), h1);
  float t = b * s + f;
  if (t < -h2) { t = -h2; s = fminf(fmaxf(b * t - cc, -h1), h1); }
  else if (t > h2) { t = h2; s = fminf(fmaxf(b * t - cc, -h1), h1); }
  s_out = s; t_out = t;
}

// ---- convex pairs (geom-pair kind 2): the engine's general convex collider = libccd's Minkowski Portal Refinement driven by the
// engine's support / centre callbacks (oracle/oracle.c: mpr_penetration is the float64 restatement this follows step by step).
// One support call site: the phases of the algorithm are a small state machine around it. Both shapes are inflated by
// margin / 2 along the search direction; result: normal from geom 1 to geom 2, contact point (relative to O) midway between the
// two witness points, distance = margin - depth. The support search of a hull climbs its vertex graph (adjacency blocks).
// INLINED into the step kernel (LM_DEV_COLD = forceinline, lm_step.h): float64 arithmetic, a private portal array and ~100 live values
// of its own cost every kernel that contains it 500-1000 B of scratch per lane — but as a real function (noinline: -1.5 % on the
// quadruped's bench rollout) kernels that CALL it at the register ceiling came out wrong under one build setting or another
// (profiles/r3_notes.md §4; -DLM_MPR_CALL reproduces it).
struct MprOut { float nx, ny, nz, px, py, pz, dist; int found; };
#ifdef LM_TIMERS
#define LM_MPR_COUNT(i, n) (mc[i] += (n))
#define LM_MPR_ARG , long long* mc
#else
#define LM_MPR_COUNT(i, n) do {} while (0)
#define LM_MPR_ARG
#endif
// mode 0: the one-direction separation test only (found = 1: not separated along it, the portal search has to decide);
// mode 1: the portal search without that test (the caller ran it: stage A / stage B of the work queue).
// WARM START (round 6). A forward pass meets nearly the configuration of the pass before it (RK4: four per substep), and the collider's
// time is the number of DEPENDENT hull fetches of its slowest lane (2.5 hill steps per support, 6.7 supports per call; 46 % of the calls
// end at the one-direction test, 12 % without contact behind the portal search). `ce` = this geom pair's record in the environment's
// cache (global memory, kMprCacheFloats floats, null: none; zeroed = empty):
//   [0..2] a direction that separated the two shapes when the pair was last seen, [3] 1 if there is one. Mode 0 tests IT first, then
//          the direction between the bounding capsules; mode 1 leaves the direction its search ended without contact at. Any direction
//          along which the inflated shapes are apart proves "no contact" (that is what the one-direction test rests on), and the test
//          is evaluated on the CURRENT frames: what the cache holds decides how fast the answer comes, never which;
//   [4 + 2 p, 5 + 2 p] the hull vertices (adjacency block + 1, 0 = none) at which the supports of the call's p-th direction ended:
//          p = 0 the cached direction, 1 the capsule direction, 2.. the first directions of the portal search. The hill climb of the
//          same direction one pass later starts there and usually ends there (one fetch instead of 2.5). A hull's support vertex does
//          not depend on where the climb starts (greedy ascent on a convex vertex graph), so the results are those of the cold search.
#ifndef LM_MPR_CACHE_HINTS
#define LM_MPR_CACHE_HINTS 14
#endif
constexpr int kMprCacheHints = LM_MPR_CACHE_HINTS, kMprCacheFloats = 4 + 2 * kMprCacheHints;
template <bool PAIRED>
LM_DEV_COLD MprOut mpr_convex_pair(const float* meshadj, const float* rec, bool g1own, V3 po_, M3 Ro_, V3 pp_, M3 Rp_, V3 O, float pmargin, int mode, float* ce LM_MPR_ARG) {
  LM_MPR_COUNT(0, mode == 0 ? 1 : 0);
  MprOut out; out.nx = out.ny = out.nz = out.px = out.py = out.pz = out.dist = 0.0f; out.found = 0;
    // FLOAT64 inside: the portal search takes hundreds of sign decisions on differences of nearly equal support values; in
    // float32 their rounding alone sends it down another path (a cylinder against a hull: normals 1e-2 rad apart from one
    // evaluation to the next of the same state, where the float64 collider does not move) — MI355X runs float64 vector code at
    // half rate, and the search is bound by the latency of the hull-vertex fetches anyway. Inputs (link frames, float32 hull
    // vertices) and outputs are float32.
    struct D3 { double x, y, z; };
    auto d3 = [](double x, double y, double z) -> D3 { D3 r; r.x = x; r.y = y; r.z = z; return r; };
    auto dsub = [&](D3 a, D3 b) -> D3 { return d3(a.x - b.x, a.y - b.y, a.z - b.z); };
    auto dadd = [&](D3 a, D3 b) -> D3 { return d3(a.x + b.x, a.y + b.y, a.z + b.z); };
    auto dscl = [&](double k, D3 a) -> D3 { return d3(k * a.x, k * a.y, k * a.z); };
    auto ddot = [](D3 a, D3 b) -> double { return a.x * b.x + a.y * b.y + a.z * b.z; };
    auto dcross = [&](D3 a, D3 b) -> D3 { return d3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); };
    auto dunit = [&](D3 a) -> D3 { const double n = sqrt(ddot(a, a)); return (n > 0.0) ? dscl(1.0 / n, a) : a; };
    auto up = [&](V3 a) -> D3 { return d3((double)a.x, (double)a.y, (double)a.z); };
    auto rot = [&](const M3& Rm, D3 v) -> D3 {          // R v
      return d3((double)Rm.a[0] * v.x + (double)Rm.a[1] * v.y + (double)Rm.a[2] * v.z, (double)Rm.a[3] * v.x + (double)Rm.a[4] * v.y + (double)Rm.a[5] * v.z,
                (double)Rm.a[6] * v.x + (double)Rm.a[7] * v.y + (double)Rm.a[8] * v.z);
    };
    auto rotT = [&](const M3& Rm, D3 v) -> D3 {         // R^T v
      return d3((double)Rm.a[0] * v.x + (double)Rm.a[3] * v.y + (double)Rm.a[6] * v.z, (double)Rm.a[1] * v.x + (double)Rm.a[4] * v.y + (double)Rm.a[7] * v.z,
                (double)Rm.a[2] * v.x + (double)Rm.a[5] * v.y + (double)Rm.a[8] * v.z);
    };
    const D3 pw[2] = {up((g1own ? po_ : pp_) - O), up((g1own ? pp_ : po_) - O)};
    const M3* Rw[2] = {g1own ? &Ro_ : &Rp_, g1own ? &Rp_ : &Ro_};
    const double hmg = 0.5 * (double)pmargin, eps = 2.220446049250313e-16;
    auto is_zero = [&](double x) -> bool { return fabs(x) < eps; };
    auto sgn = [](double x) -> double { return (x > 0.0) ? 1.0 : ((x < 0.0) ? -1.0 : 0.0); };
    int hint[2] = {-1, -1};            // where the hill climbing of either hull starts: its previous support vertex
    // one shape's support (the sequential form: one copy of the climbing loop, run for either shape)
    auto support1 = [&](int which, D3 d) -> D3 {
      const M3& Rl = *Rw[which];
      const float* cap = rec + (which ? LM_GP_P2 : LM_GP_P1);        // bounding capsule: centre 3, axis 3, half length, radius
      const float* x = rec + (which ? LM_GP_X2 : LM_GP_X1);
      const D3 dl = rotT(Rl, d);
      const int type = (int)x[LM_GX_TYPE];
      D3 loc = d3(0, 0, 0);
      if (type == LM_GEOM_MESH) {
        // hill climbing on the hull's vertex graph from the vertex the previous search of this geom ended at (the engine's own
        // support search for meshes with a graph): a handful of steps x ~6 neighbours instead of a scan over every vertex.
        // meshadj: per vertex a block [x y z degree][neighbour x y z, neighbour's block]...: one step = one contiguous block,
        // its header and first eight neighbours fetched together (one memory round trip per step)
        const F4* A = reinterpret_cast<const F4*>(meshadj);
        int cur = hint[which];
        if (cur < 0) {                  // first search of this pair: from the hull's extreme vertex along the dominant axis of the direction
          const double ax_ = fabs(dl.x), ay_ = fabs(dl.y), az_ = fabs(dl.z);
          const int k = (ax_ >= ay_ && ax_ >= az_) ? 0 : ((ay_ >= az_) ? 1 : 2);
          const double comp = (k == 0) ? dl.x : ((k == 1) ? dl.y : dl.z);
          cur = (int)x[LM_GX_E0 + 2 + 2 * k + ((comp < 0.0) ? 1 : 0)];
        }
        double best = -1.0e300;
#pragma nounroll
        for (int step = 0; step < 256; step++) {
          LM_MPR_COUNT(5, 1);
          const F4 h = A[cur];
          F4 e[8];
#pragma unroll
          for (int j = 0; j < 8; j++) e[j] = A[cur + 1 + j];          // (beyond the block's end for a lower degree: ignored; the table is padded)
          const int deg = (int)h.w;
          if (step == 0) { best = dl.x * (double)h.x + dl.y * (double)h.y + dl.z * (double)h.z; loc = d3(h.x, h.y, h.z); }
          int nxt = cur;
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const double dd = dl.x * (double)e[j].x + dl.y * (double)e[j].y + dl.z * (double)e[j].z;
            if (j < deg && dd > best) { best = dd; nxt = (int)e[j].w; loc = d3(e[j].x, e[j].y, e[j].z); }
          }
#pragma nounroll
          for (int j = 8; j < deg; j++) {
            const F4 ej = A[cur + 1 + j];
            const double dd = dl.x * (double)ej.x + dl.y * (double)ej.y + dl.z * (double)ej.z;
            if (dd > best) { best = dd; nxt = (int)ej.w; loc = d3(ej.x, ej.y, ej.z); }
          }
          if (nxt == cur) break;
          cur = nxt;
        }
        hint[which] = cur;
      } else {
        const D3 ctr = d3(cap[0], cap[1], cap[2]), ax = d3(cap[3], cap[4], cap[5]);
        if (type == LM_GEOM_BOX) {
          const D3 ex = d3(x[LM_GX_E0 + 3], x[LM_GX_E0 + 4], x[LM_GX_E0 + 5]), ey = d3(x[LM_GX_E0 + 6], x[LM_GX_E0 + 7], x[LM_GX_E0 + 8]);
          const D3 ez = dcross(ex, ey);
          loc = dadd(dadd(ctr, dscl(sgn(ddot(dl, ex)) * (double)x[LM_GX_E0], ex)), dadd(dscl(sgn(ddot(dl, ey)) * (double)x[LM_GX_E0 + 1], ey), dscl(sgn(ddot(dl, ez)) * (double)x[LM_GX_E0 + 2], ez)));
        } else if (type == LM_GEOM_CYLINDER) {
          const double da = ddot(dl, ax);
          const D3 perp = dsub(dl, dscl(da, ax));
          const double t = sqrt(ddot(perp, perp));
          loc = dadd(ctr, dscl(sgn(da) * (double)cap[6], ax));
          if (t > 1e-15) loc = dadd(loc, dscl((double)cap[7] / t, perp));
        } else loc = dadd(dadd(ctr, dscl((double)cap[7], dl)), dscl(sgn(ddot(dl, ax)) * (double)cap[6], ax));          // sphere (half length 0), capsule
      }
      return dadd(dadd(pw[which], rot(Rl, loc)), dscl(hmg, d));
    };
    // The supports of BOTH shapes for one direction: shape 0 along d, shape 1 along -d. A hull's support is a hill climb on its vertex
    // graph (the engine's own search for meshes with a graph; meshadj: per vertex a block [x y z degree][neighbour x y z, neighbour's
    // block]...: one step = header + first eight neighbours, one 144-byte fetch). The two hulls climb IN THE SAME LOOP: the fetches of
    // a step of either are in flight together, so a support pair costs max(steps) memory round trips, not their sum (the loads are
    // unconditional — a hull that has arrived re-reads its block — to keep them out of divergent branches).
    auto support_pair = [&](D3 d, D3* sp_, int ch0 = -1, int ch1 = -1) {
      if constexpr (!PAIRED) {
        // (the quadruped's kernel: its convex pairs are primitives; anything else in this function costs its bench rollout 2-3 % through
        // the register allocation of the kernel around it — this branch is the round's first version, untouched)
#pragma nounroll
        for (int w = 0; w < 2; w++) sp_[w] = support1(w, (w == 0) ? d : dscl(-1.0, d));
      } else {
        const float* const xg0 = rec + LM_GP_X1; const float* const xg1 = rec + LM_GP_X2;
        const int type0 = (int)xg0[LM_GX_TYPE], type1 = (int)xg1[LM_GX_TYPE];
        // support of a primitive (box, cylinder, capsule, sphere) in its link frame
        auto prim_support = [&](const float* cap, const float* x, int type, D3 dl) -> D3 {
          const D3 ctr = d3(cap[0], cap[1], cap[2]), ax = d3(cap[3], cap[4], cap[5]);
          if (type == LM_GEOM_BOX) {
            const D3 ex = d3(x[LM_GX_E0 + 3], x[LM_GX_E0 + 4], x[LM_GX_E0 + 5]), ey = d3(x[LM_GX_E0 + 6], x[LM_GX_E0 + 7], x[LM_GX_E0 + 8]);
            const D3 ez = dcross(ex, ey);
            return dadd(dadd(ctr, dscl(sgn(ddot(dl, ex)) * (double)x[LM_GX_E0], ex)), dadd(dscl(sgn(ddot(dl, ey)) * (double)x[LM_GX_E0 + 1], ey), dscl(sgn(ddot(dl, ez)) * (double)x[LM_GX_E0 + 2], ez)));
          }
          if (type == LM_GEOM_CYLINDER) {
            const double da = ddot(dl, ax);
            const D3 perp = dsub(dl, dscl(da, ax));
            const double t = sqrt(ddot(perp, perp));
            D3 loc = dadd(ctr, dscl(sgn(da) * (double)cap[6], ax));
            if (t > 1e-15) loc = dadd(loc, dscl((double)cap[7] / t, perp));
            return loc;
          }
          return dadd(dadd(ctr, dscl((double)cap[7], dl)), dscl(sgn(ddot(dl, ax)) * (double)cap[6], ax));          // sphere (half length 0), capsule
        };
        // where the climb of a hull starts: the vertex its previous search ended at, or (first search of this pair) the hull's extreme
        // vertex along the dominant axis of the direction
        auto start_vertex = [&](const float* x, int hint, D3 dl) -> int {
          if (hint >= 0) return hint;
          const double ax_ = fabs(dl.x), ay_ = fabs(dl.y), az_ = fabs(dl.z);
          const int k = (ax_ >= ay_ && ax_ >= az_) ? 0 : ((ay_ >= az_) ? 1 : 2);
          const double comp = (k == 0) ? dl.x : ((k == 1) ? dl.y : dl.z);
          return (int)x[LM_GX_E0 + 2 + 2 * k + ((comp < 0.0) ? 1 : 0)];
        };
      D3& s0 = sp_[0]; D3& s1 = sp_[1];
      const D3 dm = dscl(-1.0, d);
      const D3 dl0 = rotT(*Rw[0], d), dl1 = rotT(*Rw[1], dm);
      const bool mesh0 = type0 == LM_GEOM_MESH, mesh1 = type1 == LM_GEOM_MESH;
      D3 loc0 = d3(0, 0, 0), loc1 = d3(0, 0, 0);
      if (!mesh0) loc0 = prim_support(rec + LM_GP_P1, xg0, type0, dl0);
      if (!mesh1) loc1 = prim_support(rec + LM_GP_P2, xg1, type1, dl1);
      if (mesh0 || mesh1) {
#if defined(LM_TIMERS) && defined(LM_MPR_CLOCK)
        const long long tc0_ = LM_CLOCK();
#endif
        const F4* A = reinterpret_cast<const F4*>(meshadj);
        int cur0 = mesh0 ? start_vertex(xg0, (ch0 >= 0) ? ch0 : hint[0], dl0) : 0, cur1 = mesh1 ? start_vertex(xg1, (ch1 >= 0) ? ch1 : hint[1], dl1) : 0;
        bool go0 = mesh0, go1 = mesh1;
        double best0 = -1.0e300, best1 = -1.0e300;
#pragma nounroll
        for (int step = 0; step < 256 && (go0 || go1); step++) {
          LM_MPR_COUNT(5, (go0 ? 1 : 0) + (go1 ? 1 : 0));
          const F4 h0 = A[cur0], h1 = A[cur1];
          F4 e0[8], e1[8];
#pragma unroll
          for (int j = 0; j < 8; j++) { e0[j] = A[cur0 + 1 + j]; e1[j] = A[cur1 + 1 + j]; }     // (beyond a block's end for a lower degree: ignored; the table is padded)
          if (go0) {
            const int deg = (int)h0.w;
            if (step == 0) { best0 = dl0.x * (double)h0.x + dl0.y * (double)h0.y + dl0.z * (double)h0.z; loc0 = d3(h0.x, h0.y, h0.z); }
            int nxt = cur0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const double dd = dl0.x * (double)e0[j].x + dl0.y * (double)e0[j].y + dl0.z * (double)e0[j].z;
              if (j < deg && dd > best0) { best0 = dd; nxt = (int)e0[j].w; loc0 = d3(e0[j].x, e0[j].y, e0[j].z); }
            }
#pragma nounroll
            for (int j = 8; j < deg; j++) {
              const F4 ej = A[cur0 + 1 + j];
              const double dd = dl0.x * (double)ej.x + dl0.y * (double)ej.y + dl0.z * (double)ej.z;
              if (dd > best0) { best0 = dd; nxt = (int)ej.w; loc0 = d3(ej.x, ej.y, ej.z); }
            }
            if (nxt == cur0) go0 = false; else cur0 = nxt;
          }
          if (go1) {
            const int deg = (int)h1.w;
            if (step == 0) { best1 = dl1.x * (double)h1.x + dl1.y * (double)h1.y + dl1.z * (double)h1.z; loc1 = d3(h1.x, h1.y, h1.z); }
            int nxt = cur1;
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const double dd = dl1.x * (double)e1[j].x + dl1.y * (double)e1[j].y + dl1.z * (double)e1[j].z;
              if (j < deg && dd > best1) { best1 = dd; nxt = (int)e1[j].w; loc1 = d3(e1[j].x, e1[j].y, e1[j].z); }
            }
#pragma nounroll
            for (int j = 8; j < deg; j++) {
              const F4 ej = A[cur1 + 1 + j];
              const double dd = dl1.x * (double)ej.x + dl1.y * (double)ej.y + dl1.z * (double)ej.z;
              if (dd > best1) { best1 = dd; nxt = (int)ej.w; loc1 = d3(ej.x, ej.y, ej.z); }
            }
            if (nxt == cur1) go1 = false; else cur1 = nxt;
          }
        }
        if (mesh0) hint[0] = cur0;
        if (mesh1) hint[1] = cur1;
#if defined(LM_TIMERS) && defined(LM_MPR_CLOCK)
        if (mode == 1) mc[10] += LM_CLOCK() - tc0_;          // (probe: cycles in the hull climbs of the portal search; [13]: the search as a whole; [11] its support calls; [14] its directions)
#endif
      }
      s0 = dadd(dadd(pw[0], rot(*Rw[0], loc0)), dscl(hmg, d));
      s1 = dadd(dadd(pw[1], rot(*Rw[1], loc1)), dscl(hmg, dm));
      }
    };
    // the portal: points 1..3 as (v, v1), point 0 = v0; `expand` replaces the point with a run-time number. One private array PV[3][6]
    // (scratch: every direction of the search stores six doubles and loads eighteen). Round 6 measured the alternatives on
    // HumanoidTorque.run, same box each: everything in registers (36 selects of doubles per `put`, -DLM_MPR_PV_REGS) 12.86 ms per control
    // step against 12.60 with the array; the points v in nine double registers and only their witness points v1 in the array
    // (-DLM_MPR_PV_SPLIT) 12.84 against 11.75 — 18 more live registers in the portal loop cost the kernel around it more than the
    // loads they spare. The array stays.
#if !defined(LM_MPR_PV_SPLIT)
    double PV[3][6];
    auto pv = [&](int q) -> D3 { return d3(PV[q - 1][0], PV[q - 1][1], PV[q - 1][2]); };
    auto pv1 = [&](int q) -> D3 { return d3(PV[q - 1][3], PV[q - 1][4], PV[q - 1][5]); };
#ifndef LM_MPR_PV_REGS
    auto put = [&](int q, D3 v, D3 v1) {
      PV[q - 1][0] = v.x; PV[q - 1][1] = v.y; PV[q - 1][2] = v.z; PV[q - 1][3] = v1.x; PV[q - 1][4] = v1.y; PV[q - 1][5] = v1.z;
    };
#else
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 6; j++) PV[i][j] = 0.0;
    auto put = [&](int q, D3 v, D3 v1) {
      const double nw[6] = {v.x, v.y, v.z, v1.x, v1.y, v1.z};
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) PV[i][j] = (q - 1 == i) ? nw[j] : PV[i][j];
    };
#endif
#else
    D3 PA = d3(0, 0, 0), PB = d3(0, 0, 0), PC = d3(0, 0, 0);
    double PW[3][3];
    auto pv = [&](int q) -> D3 { return (q == 1) ? PA : ((q == 2) ? PB : PC); };
    auto pv1 = [&](int q) -> D3 { return d3(PW[q - 1][0], PW[q - 1][1], PW[q - 1][2]); };
    auto put = [&](int q, D3 v, D3 v1) {
      PA = (q == 1) ? v : PA; PB = (q == 2) ? v : PB; PC = (q == 3) ? v : PC;
      PW[q - 1][0] = v1.x; PW[q - 1][1] = v1.y; PW[q - 1][2] = v1.z;
    };
#endif
    const float* x1 = rec + LM_GP_X1; const float* x2 = rec + LM_GP_X2;
    const D3 c1 = dadd(pw[0], rot(*Rw[0], d3(x1[LM_GX_CX], x1[LM_GX_CY], x1[LM_GX_CZ])));
    const D3 c2 = dadd(pw[1], rot(*Rw[1], d3(x2[LM_GX_CX], x2[LM_GX_CY], x2[LM_GX_CZ])));
    D3 v0 = dsub(c1, c2);
    if (is_zero(v0.x) && is_zero(v0.y) && is_zero(v0.z)) v0.x += 10.0 * eps;
    auto portal_dir = [&]() -> D3 { const D3 a1 = pv(1); return dunit(dcross(dsub(pv(2), a1), dsub(pv(3), a1))); };
    auto expand = [&](D3 v4, D3 v41) {
      const D3 cr = dcross(v4, v0);
      int q;
      if (ddot(pv(1), cr) > 0.0) q = (ddot(pv(2), cr) > 0.0) ? 1 : 3;
      else q = (ddot(pv(3), cr) > 0.0) ? 2 : 1;
      put(q, v4, v41);
    };
    if (mode == 0) {
      // one-direction separation test first: along the line between the closest points of the two bounding capsules. Shapes
      // (inflated by margin / 2 each) that are apart along ANY direction do not overlap - the portal search below would say so
      // after five or six support searches, this says it after two, and most queued pairs end here
      const V3 p1f = (g1own ? po_ : pp_) - O, p2f = (g1own ? pp_ : po_) - O;
      const V3 cA = p1f + mul(*Rw[0], v3(rec[LM_GP_P1], rec[LM_GP_P1 + 1], rec[LM_GP_P1 + 2])), aA = mul(*Rw[0], v3(rec[LM_GP_A1], rec[LM_GP_A1 + 1], rec[LM_GP_A1 + 2]));
      const V3 cB = p2f + mul(*Rw[1], v3(rec[LM_GP_P2], rec[LM_GP_P2 + 1], rec[LM_GP_P2 + 2])), aB = mul(*Rw[1], v3(rec[LM_GP_A2], rec[LM_GP_A2 + 1], rec[LM_GP_A2 + 2]));
      float sa, ta;
      segment_closest(cA, aA, rec[LM_GP_H1], cB, aB, rec[LM_GP_H2], sa, ta);
      const V3 dsep = (cB + ta * aB) - (cA + sa * aA);
      // hints of the two directions of this mode (one 16-byte word of the cache record), the cached direction (another)
      unsigned long long hw0 = 0ull, hw1 = 0ull, cw0 = 0ull, cw1 = 0ull;
      if constexpr (PAIRED) if (ce) { cw0 = LM_GLD64(ce); cw1 = LM_GLD64(ce + 2); hw0 = LM_GLD64(ce + 4); hw1 = LM_GLD64(ce + 6); }
      auto lo_ = [](unsigned long long w) -> int { return (int)(unsigned)(w & 0xffffffffull) - 1; };
      auto hi_ = [](unsigned long long w) -> int { return (int)(unsigned)(w >> 32) - 1; };
      auto pack_ = [&]() -> unsigned long long { return (unsigned long long)(unsigned)(hint[0] + 1) | ((unsigned long long)(unsigned)(hint[1] + 1) << 32); };
      if constexpr (PAIRED) if (ce && (unsigned)(cw1 >> 32) != 0u) {
        const D3 dc = dunit(d3((double)lm_u2f((unsigned)cw0), (double)lm_u2f((unsigned)(cw0 >> 32)), (double)lm_u2f((unsigned)cw1)));
        D3 sp[2];
        support_pair(dc, sp, lo_(hw0), hi_(hw0));
        LM_MPR_COUNT(4, 1);
        LM_GST64(ce + 4, pack_());
        if (ddot(dsub(sp[0], sp[1]), dc) < 0.0) { LM_MPR_COUNT(1, 1); return out; }
      }
      if (dot(dsep, dsep) > 1e-12f) {
        const D3 du = dunit(up(dsep));
        D3 sp[2];
        support_pair(du, sp, lo_(hw1), hi_(hw1));
        LM_MPR_COUNT(4, 1);
        if constexpr (PAIRED) if (ce) LM_GST64(ce + 6, pack_());
        if (ddot(dsub(sp[0], sp[1]), du) < 0.0) {
          LM_MPR_COUNT(1, 1);
          if constexpr (PAIRED) if (ce) {       // the capsule direction separates: the direction to try first next time
            LM_GST64(ce, (unsigned long long)lm_f2u((float)du.x) | ((unsigned long long)lm_f2u((float)du.y) << 32));
            LM_GST64(ce + 2, (unsigned long long)lm_f2u((float)du.z) | (1ull << 32));
          }
          return out;
        }
      }
      out.found = 1;
      return out;
    }
#if defined(LM_TIMERS) && defined(LM_MPR_CLOCK)
    const long long tp0_ = LM_CLOCK();
    struct ClockOut_ { long long* m; long long t0; __device__ ~ClockOut_() { m[13] += LM_CLOCK() - t0; } } clock_out_{mc, tp0_};
#endif
    D3 dir = dunit(dscl(-1.0, v0));
    int stage = 0, iter = 0, result = 0;                 // result: 1 contact from the portal, 2 origin on the segment v0-v1, -1 none
    int nsupport = 0; (void)nsupport;
    // the cache's hints for the first directions of the search: the word of direction `guard + 1` is fetched while direction `guard`
    // climbs (two registers instead of the whole record across the loop)
    constexpr int kSeq = kMprCacheHints - 2;
    unsigned long long hnext = 0ull;
    if constexpr (PAIRED) if (ce) hnext = LM_GLD64(ce + 8);
#pragma nounroll
    for (int guard = 0; guard < 192 && result == 0; guard++) {
      nsupport++;
      LM_MPR_COUNT(4, 1);
      D3 sup[2];
#if defined(LM_TIMERS) && defined(LM_MPR_CLOCK)
      const long long ts0_ = LM_CLOCK();
      mc[14] += 1;
#endif
      if constexpr (PAIRED) {
        const unsigned long long hw = hnext;
        hnext = 0ull;
        if (ce && guard + 1 < kSeq) hnext = LM_GLD64(ce + 8 + 2 * (guard + 1));
        support_pair(dir, sup, (int)(unsigned)(hw & 0xffffffffull) - 1, (int)(unsigned)(hw >> 32) - 1);
        if (ce && guard < kSeq) LM_GST64(ce + 8 + 2 * guard, (unsigned long long)(unsigned)(hint[0] + 1) | ((unsigned long long)(unsigned)(hint[1] + 1) << 32));
      } else support_pair(dir, sup);
#if defined(LM_TIMERS) && defined(LM_MPR_CLOCK)
      mc[11] += LM_CLOCK() - ts0_;
#endif
      const D3 sv = dsub(sup[0], sup[1]);
      const double dt = ddot(sv, dir);
      if (stage == 0) {
        put(1, sv, sup[0]);
        if (is_zero(dt) || dt < 0.0) { result = -1; break; }
        dir = dcross(v0, sv);
        if (is_zero(ddot(dir, dir))) { result = (is_zero(sv.x) && is_zero(sv.y) && is_zero(sv.z)) ? -1 : 2; break; }     // touching at v1: no normal | origin on v0-v1
        dir = dunit(dir);
        stage = 1;
      } else if (stage == 1) {
        if (is_zero(dt) || dt < 0.0) { result = -1; break; }
        put(2, sv, sup[0]);
        dir = dunit(dcross(dsub(pv(1), v0), dsub(sv, v0)));
        if (ddot(dir, v0) > 0.0) { const D3 a = pv(1), a1 = pv1(1); put(1, sv, sup[0]); put(2, a, a1); dir = dscl(-1.0, dir); }
        stage = 2;
      } else if (stage == 2) {
        if (is_zero(dt) || dt < 0.0) { result = -1; break; }
        put(3, sv, sup[0]);
        bool cont = false;
        double tp = ddot(dcross(pv(1), sv), v0);
        if (tp < 0.0 && !is_zero(tp)) { put(2, sv, sup[0]); cont = true; }
        else {
          tp = ddot(dcross(sv, pv(2)), v0);
          if (tp < 0.0 && !is_zero(tp)) { put(1, sv, sup[0]); cont = true; }
        }
        if (cont) dir = dunit(dcross(dsub(pv(1), v0), dsub(pv(2), v0)));
        else {
          dir = portal_dir();
          const double de = ddot(dir, pv(1));
          stage = (is_zero(de) || de > 0.0) ? 4 : 3;          // the portal already holds the origin: straight to the penetration phase
        }
      } else {
        // reach of the new support point beyond the portal along dir
        const double reach = fmin(fmin(dt - ddot(pv(1), dir), dt - ddot(pv(2), dir)), dt - ddot(pv(3), dir));
#ifdef LM_MPR_TRACE
        printf("  d stage %d it %d dir %.6f %.6f %.6f v4 %.6f %.6f %.6f dv4 %.8f dv1 %.8f reach %.3g\n", stage, iter, dir.x, dir.y, dir.z, sv.x, sv.y, sv.z, dt, ddot(pv(1), dir), reach);
#endif
        if (stage == 3) {
          if (!(is_zero(dt) || dt > 0.0) || reach <= 1e-6) { result = -1; break; }
          expand(sv, sup[0]);
          dir = portal_dir();
          const double de = ddot(dir, pv(1));
          if (is_zero(de) || de > 0.0) stage = 4;
        } else {
          if (reach <= 1e-6 || iter > 50) { result = 1; break; }
          expand(sv, sup[0]);
          dir = portal_dir();
          iter++;
        }
      }
    }
#ifdef LM_PAIR_TRACE
    printf("   mpr result %d stage %d iter %d supports %d types %d %d\n", result, stage, iter, nsupport, (int)rec[LM_GP_X1], (int)rec[LM_GP_X2]);
#endif
    LM_MPR_COUNT(6, iter);
    if constexpr (PAIRED) if (ce) {
      // no contact: the search ended at a direction the shapes are apart along (or as good as: the next pass tests it before it trusts it);
      // a contact: no direction to try first
      if (result < 0) LM_GST64(ce, (unsigned long long)lm_f2u((float)dir.x) | ((unsigned long long)lm_f2u((float)dir.y) << 32));
      LM_GST64(ce + 2, (unsigned long long)lm_f2u((float)dir.z) | ((result < 0) ? (1ull << 32) : 0ull));
    }
    if (result <= 0) { LM_MPR_COUNT(2, 1); return out; }
    LM_MPR_COUNT(3, 1);
    double depth; D3 pdir, pos;
    if (result == 2) {
      const D3 v1 = pv(1), s1 = pv1(1);
      depth = sqrt(ddot(v1, v1)); pdir = dunit(v1);
      pos = dscl(0.5, dadd(s1, dsub(s1, v1)));
    } else {
      // closest point of the portal triangle to the origin (libccd: ccdVec3PointTriDist2) -> depth and direction
      const D3 a = pv(1), b = pv(2), cc = pv(3);
      const D3 d1 = dsub(b, a), d2 = dsub(cc, a);
      const double v = ddot(d1, d1), w = ddot(d2, d2), pq = ddot(a, d1), qq = ddot(a, d2), r = ddot(d1, d2);
      const double det = w * v - r * r;
      double sb = -1.0, tb = -1.0;
      if (!is_zero(det)) { sb = (qq * r - w * pq) / det; tb = (-sb * r - qq) / w; }
      auto ccd_eq1 = [&](double x) -> bool { const double ab = fabs(x - 1.0); return ab < eps || ab < eps * fmax(fabs(x), 1.0); };
      D3 wit;
      if ((is_zero(sb) || sb > 0.0) && (ccd_eq1(sb) || sb < 1.0) && (is_zero(tb) || tb > 0.0) && (ccd_eq1(tb) || tb < 1.0) && (ccd_eq1(tb + sb) || tb + sb < 1.0))
        wit = dadd(a, dadd(dscl(sb, d1), dscl(tb, d2)));
      else {
        auto seg = [&](D3 x0, D3 x1e, D3& wout) -> double {
          const D3 dd = dsub(x1e, x0);
          const double t = -ddot(x0, dd) / ddot(dd, dd);
          if (t < 0.0 || is_zero(t)) wout = x0;
          else if (t > 1.0 || ccd_eq1(t)) wout = x1e;
          else wout = dadd(x0, dscl(t, dd));
          return ddot(wout, wout);
        };
        D3 w2;
        double best = seg(a, b, wit);
        double d = seg(a, cc, w2); if (d < best) { best = d; wit = w2; }
        d = seg(b, cc, w2); if (d < best) { best = d; wit = w2; }
      }
      depth = sqrt(ddot(wit, wit));
      pdir = (is_zero(wit.x) && is_zero(wit.y) && is_zero(wit.z)) ? dir : dunit(wit);
      // barycentric coordinates of the origin in the portal tetrahedron -> witness points on the two shapes
      const D3 p0 = v0;
      double bc[4];
      bc[0] = ddot(dcross(a, b), cc); bc[1] = ddot(dcross(cc, b), p0); bc[2] = ddot(dcross(p0, a), cc); bc[3] = ddot(dcross(b, a), p0);
      double sum = bc[0] + bc[1] + bc[2] + bc[3];
      if (is_zero(sum) || sum < 0.0) {
        const D3 pd = portal_dir();
        bc[0] = 0.0; bc[1] = ddot(dcross(b, cc), pd); bc[2] = ddot(dcross(cc, a), pd); bc[3] = ddot(dcross(a, b), pd);
        sum = bc[1] + bc[2] + bc[3];
      }
      const double inv = 1.0 / sum;
      D3 q1 = dscl(bc[0], c1), q2 = dscl(bc[0], c2);
#pragma unroll
      for (int q = 1; q < 4; q++) { const D3 s1 = pv1(q), vv = pv(q); q1 = dadd(q1, dscl(bc[q], s1)); q2 = dadd(q2, dscl(bc[q], dsub(s1, vv))); }
      pos = dscl(0.5, dadd(dscl(inv, q1), dscl(inv, q2)));
    }
    // the engine's mjc_fixNormal: a sphere / capsule in the pair takes the direction from its centre line to the contact point
    {
      D3 nn[2]; bool have[2] = {false, false};
#pragma unroll
      for (int w = 0; w < 2; w++) {
        const float* x = rec + (w ? LM_GP_X2 : LM_GP_X1);
        const int type = (int)x[LM_GX_TYPE];
        if (type == LM_GEOM_SPHERE || type == LM_GEOM_CAPSULE) {
          const float* cap = rec + (w ? LM_GP_P2 : LM_GP_P1);
          const D3 ctr = dadd(pw[w], rot(*Rw[w], d3(cap[0], cap[1], cap[2]))), ax = rot(*Rw[w], d3(cap[3], cap[4], cap[5]));
          const D3 rel = dsub(pos, ctr);
          const double t = fmin(fmax(ddot(rel, ax), -(double)cap[6]), (double)cap[6]);
          nn[w] = dunit(dsub(rel, dscl(t, ax))); have[w] = true;
        }
      }
      if (have[0] && have[1]) pdir = dunit(dsub(nn[0], nn[1]));
      else if (have[0]) pdir = nn[0];
      else if (have[1]) pdir = dscl(-1.0, nn[1]);
    }
  out.nx = (float)pdir.x; out.ny = (float)pdir.y; out.nz = (float)pdir.z; out.px = (float)pos.x; out.py = (float)pos.y; out.pz = (float)pos.z;
  out.dist = (float)((double)pmargin - depth); out.found = 1;
  return out;
}

// ---- native pairs (geom-pair kind 1): pairs the engine sends to its own colliders for a box or a cylinder against a sphere / capsule /
// box (mjc_SphereBox, mjc_SphereCylinder, mjc_CapsuleBox, mjc_BoxBox of the third-party mujoco==2.3.7). Sphere-box and sphere-cylinder
// are determined by the geometry (closest feature) and are restatements. CAPSULE-BOX and BOX-BOX are NOT the engine's case analysis:
// the engine's source is not available here, and these two are this code's own constructions with the engine's contact conventions
// (normal from geom 1 to geom 2, point midway, dist = signed gap, at most 2 / 8 contacts) — capsule-box: the closest point of the axis
// segment (bisection) as a sphere against the box, a second contact at the far end of the axis when that end is within the margin;
// box-box: a 15-axis separation test, an edge-edge contact when an edge axis separates best by more than 5 % + 1e-6, else the incident
// face clipped against the reference face (Sutherland-Hodgman). Exactly the engine's result where the geometry leaves no choice: the
// box-box EDGE case is pinned to 1e-14 by the reference's golden rollout HumanoidTorque4Ages.run.all rows 9-10; the FACE case is off by
// 5.7e-3 at HumanoidMuscle4Ages.run.all row 33 (tests/test_oracle_golden.py holds it to 1e-2): approximate. oracle/oracle.c nat_* carries
// the same constructions in float64, so device-vs-oracle agreement says nothing about these two against the engine; their contacts are
// counted (Counters::natown -> lm_stats.own_manifold_contacts). A pair can have several
// contacts (capsule-box 2, box-box 8): `sub` selects one, `ncon` says how many there are — the caller runs the collider once per
// contact instead of keeping eight results alive in a kernel that has no registers to spare.
struct NatGeom { int type; V3 c, ax; float half, rad; V3 ex, ey, ez, hs; };
LM_DEV NatGeom nat_geom(const float* rec, bool second, V3 pl, const M3& Rl, V3 O) {
  NatGeom g;
  const float* cap = rec + (second ? LM_GP_P2 : LM_GP_P1);       // centre 3, axis 3, half length, radius (exact for sphere / capsule / cylinder)
  const float* x = rec + (second ? LM_GP_X2 : LM_GP_X1);
  g.type = (int)x[LM_GX_TYPE];
  g.c = pl + mul(Rl, v3(cap[0], cap[1], cap[2])) - O;
  g.ax = mul(Rl, v3(cap[3], cap[4], cap[5]));
  g.half = cap[6]; g.rad = cap[7];
  g.ex = mul(Rl, v3(x[LM_GX_E0 + 3], x[LM_GX_E0 + 4], x[LM_GX_E0 + 5])); g.ey = mul(Rl, v3(x[LM_GX_E0 + 6], x[LM_GX_E0 + 7], x[LM_GX_E0 + 8]));
  g.ez = cross(g.ex, g.ey);
  g.hs = v3(x[LM_GX_E0], x[LM_GX_E0 + 1], x[LM_GX_E0 + 2]);
  return g;
}
struct NatOut { float dist; V3 n, p; bool found; };
LM_DEV NatOut nat_sphere_sphere(V3 c1, float r1, V3 c2, float r2, float margin) {
  NatOut o; o.found = false; o.dist = 0.0f; o.n = v3(1, 0, 0); o.p = c1;
  const V3 dv = c2 - c1;
  const float d = sqrtf(dot(dv, dv)), dist = d - r1 - r2;
  if (dist >= margin) return o;
  o.n = (d < 1e-15f) ? v3(1, 0, 0) : (1.0f / d) * dv;
  o.dist = dist; o.p = c1 + (r1 + 0.5f * dist) * o.n; o.found = true;
  return o;
}
LM_DEV NatOut nat_sphere_box(V3 c, float r, const NatGeom& B, float margin) {
  NatOut o; o.found = false; o.dist = 0.0f; o.n = v3(1, 0, 0); o.p = c;
  const V3 rel = c - B.c;
  const float ctr[3] = {dot(B.ex, rel), dot(B.ey, rel), dot(B.ez, rel)}, hs[3] = {B.hs.x, B.hs.y, B.hs.z};
  float cl[3], d[3], d2 = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; k++) { cl[k] = fminf(fmaxf(ctr[k], -hs[k]), hs[k]); d[k] = cl[k] - ctr[k]; d2 = fmaf(d[k], d[k], d2); }
  float dist = sqrtf(d2);
  if (dist - r >= margin) return o;
  float nl[3] = {0.0f, 0.0f, 0.0f}, pl[3];
  if (dist <= 1e-15f) {               // centre inside the box: out through the nearest face
    float closest = 2.0f * (hs[0] + hs[1] + hs[2]); int kk = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) { const float fd = fabsf(((i & 1) ? 1.0f : -1.0f) * hs[i >> 1] - ctr[i >> 1]); if (closest > fd) { closest = fd; kk = i; } }
#pragma unroll
    for (int k = 0; k < 3; k++) nl[k] = (k == (kk >> 1)) ? ((kk & 1) ? -1.0f : 1.0f) : 0.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) pl[k] = ctr[k] + nl[k] * (r - closest) * 0.5f;
    dist = -closest;
  } else {
    const float id = 1.0f / dist;
#pragma unroll
    for (int k = 0; k < 3; k++) { nl[k] = d[k] * id; pl[k] = 0.5f * (cl[k] + ctr[k] + nl[k] * r); }
  }
  o.dist = dist - r;
  o.n = nl[0] * B.ex + nl[1] * B.ey + nl[2] * B.ez;
  o.p = B.c + pl[0] * B.ex + pl[1] * B.ey + pl[2] * B.ez;
  o.found = true;
  return o;
}
LM_DEV NatOut nat_sphere_cylinder(V3 c, float r, const NatGeom& C, float margin) {
  const float radius = C.rad, height = C.half;
  const V3 vec = c - C.c;
  const float x = dot(vec, C.ax);
  const V3 ap = x * C.ax, pp = vec - ap;
  const float pp2 = dot(pp, pp);
  bool side = fabsf(x) < height, cap = pp2 < radius * radius;
  if (side && cap) { if (height - fabsf(x) < radius - sqrtf(pp2)) side = false; else cap = false; }
  if (side) return nat_sphere_sphere(c, r, C.c + ap, radius, margin);
  if (cap) {
    NatOut o; o.found = false; o.dist = 0.0f; o.n = v3(1, 0, 0); o.p = c;
    const V3 n = ((x > 0.0f) ? 1.0f : -1.0f) * C.ax;
    const float cd = dot(c - (C.c + height * n), n);
    if (cd - r >= margin) return o;
    o.dist = cd - r; o.p = c + (-0.5f * (cd - r) - r) * n; o.n = (-1.0f) * n; o.found = true;
    return o;
  }
  const float sc = radius / sqrtf(fmaxf(pp2, 1e-15f));
  return nat_sphere_sphere(c, r, C.c + ((x > 0.0f) ? height : -height) * C.ax + sc * pp, 0.0f, margin);
}
LM_DEV NatOut nat_capsule_box(const NatGeom& K, const NatGeom& B, float margin, int sub, int& ncon) {
  const V3 rel = K.c - B.c;
  const float p[3] = {dot(B.ex, rel), dot(B.ey, rel), dot(B.ez, rel)}, hs[3] = {B.hs.x, B.hs.y, B.hs.z};
  const float a[3] = {dot(B.ex, K.ax) * K.half, dot(B.ey, K.ax) * K.half, dot(B.ez, K.ax) * K.half};
  // g'(t) = (P(t) - clamp(P(t))) . a, non-decreasing along the axis segment: its leftmost non-negative point is the closest point
  auto gp = [&](float t) -> float {
    float s_ = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) { const float P = fmaf(t, a[k], p[k]); s_ = fmaf(P - fminf(fmaxf(P, -hs[k]), hs[k]), a[k], s_); }
    return s_;
  };
  float t1;
  if (gp(-1.0f) >= 0.0f) t1 = -1.0f;
  else if (gp(1.0f) < 0.0f) t1 = 1.0f;
  else {
    float lo = -1.0f, hi = 1.0f;
#pragma nounroll
    for (int it = 0; it < 30; it++) { const float mid = 0.5f * (lo + hi); if (gp(mid) < 0.0f) lo = mid; else hi = mid; }
    t1 = hi;
  }
  NatOut o = nat_sphere_box(K.c + (K.half * t1) * K.ax, K.rad, B, margin);
  ncon = o.found ? 1 : 0;
  const float t2 = (t1 <= 0.0f) ? 1.0f : -1.0f;
  if (o.found && fabsf(t2 - t1) * K.half > 1e-9f) {
    const NatOut o2 = nat_sphere_box(K.c + (K.half * t2) * K.ax, K.rad, B, margin);
    if (o2.found) { ncon = 2; if (sub == 1) return o2; }
  }
  if (sub != 0) o.found = false;
  return o;
}
LM_DEV NatOut nat_box_box(const NatGeom& A, const NatGeom& B, float margin, int sub, int& ncon) {
  NatOut o; o.found = false; o.dist = 0.0f; o.n = v3(1, 0, 0); o.p = A.c;
  ncon = 0;
  const V3 Aa[3] = {A.ex, A.ey, A.ez}, Ba[3] = {B.ex, B.ey, B.ez};
  const float s1[3] = {A.hs.x, A.hs.y, A.hs.z}, s2[3] = {B.hs.x, B.hs.y, B.hs.z};
  const V3 d = B.c - A.c;
  float best_face = -3.0e38f, best_edge = -3.0e38f;
  V3 nf = v3(0, 0, 0), ne = v3(0, 0, 0);
  int face = 0, ei = 0, ej = 0;
  auto gap_of = [&](V3 n, float& pr) -> float {
    float rA = 0.0f, rB = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) { rA = fmaf(s1[k], fabsf(dot(n, Aa[k])), rA); rB = fmaf(s2[k], fabsf(dot(n, Ba[k])), rB); }
    pr = dot(d, n);
    return fabsf(pr) - rA - rB;
  };
#pragma unroll
  for (int f = 0; f < 6; f++) {
    const V3 n = (f < 3) ? Aa[f % 3] : Ba[f % 3];
    float pr; const float gap = gap_of(n, pr);
    if (gap > best_face) { best_face = gap; face = f; nf = ((pr >= 0.0f) ? 1.0f : -1.0f) * n; }
  }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      V3 n = cross(Aa[i], Ba[j]);
      const float l = sqrtf(dot(n, n));
      if (l < 1e-6f) continue;
      n = (1.0f / l) * n;
      float pr; const float gap = gap_of(n, pr);
      if (gap > best_edge) { best_edge = gap; ei = i; ej = j; ne = ((pr >= 0.0f) ? 1.0f : -1.0f) * n; }
    }
  if (best_face >= margin || best_edge >= margin) return o;
  if (best_edge > best_face + 0.05f * fabsf(best_face) + 1e-6f) {       // an edge pair decides only when it separates clearly better than every face
    V3 pa = A.c, pb = B.c, ai = Aa[0], bj = Ba[0];
    float hi_ = s1[0], hj_ = s2[0];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (k != ei) pa = pa + (((dot(ne, Aa[k]) > 0.0f) ? 1.0f : -1.0f) * s1[k]) * Aa[k]; else { ai = Aa[k]; hi_ = s1[k]; }
      if (k != ej) pb = pb + (((dot(ne, Ba[k]) > 0.0f) ? -1.0f : 1.0f) * s2[k]) * Ba[k]; else { bj = Ba[k]; hj_ = s2[k]; }
    }
    float sa, ta;
    segment_closest(pa, ai, hi_, pb, bj, hj_, sa, ta);
    ncon = 1;
    if (sub == 0) { o.dist = best_edge; o.n = ne; o.p = 0.5f * ((pa + sa * ai) + (pb + ta * bj)); o.found = true; }
    return o;
  }
  // face contact: the incident face of the other box clipped against the reference face
  const bool ref1 = face < 3; const int kr = face % 3;
  const V3* Ar = ref1 ? Aa : Ba; const V3* Ai = ref1 ? Ba : Aa;
  const float* sr = ref1 ? s1 : s2; const float* si = ref1 ? s2 : s1;
  const V3 prc = ref1 ? A.c : B.c, pic = ref1 ? B.c : A.c;
  const V3 nr = ref1 ? nf : (-1.0f) * nf;
  V3 arn = Ar[0], aru = Ar[1], arv = Ar[2]; float srn = sr[0], sru = sr[1], srv = sr[2];
  if (kr == 1) { arn = Ar[1]; aru = Ar[2]; arv = Ar[0]; srn = sr[1]; sru = sr[2]; srv = sr[0]; }
  else if (kr == 2) { arn = Ar[2]; aru = Ar[0]; arv = Ar[1]; srn = sr[2]; sru = sr[0]; srv = sr[1]; }
  (void)arn;
  int ji = 0; float bj_ = -1.0f;
#pragma unroll
  for (int j = 0; j < 3; j++) { const float v = fabsf(dot(nr, Ai[j])); if (v > bj_) { bj_ = v; ji = j; } }
  V3 ain = Ai[0], aiu = Ai[1], aiv = Ai[2]; float sin_ = si[0], siu = si[1], siv = si[2];
  if (ji == 1) { ain = Ai[1]; aiu = Ai[2]; aiv = Ai[0]; sin_ = si[1]; siu = si[2]; siv = si[0]; }
  else if (ji == 2) { ain = Ai[2]; aiu = Ai[0]; aiv = Ai[1]; sin_ = si[2]; siu = si[0]; siv = si[1]; }
  const float sgi = (dot(nr, ain) > 0.0f) ? -1.0f : 1.0f;           // the incident face looks back at the reference box
  float poly[16][3], tmp[16][3];
  int np_ = 4;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const float su = (q == 0 || q == 3) ? 1.0f : -1.0f, sv = (q < 2) ? 1.0f : -1.0f;
    const V3 w = pic + (sgi * sin_) * ain + (su * siu) * aiu + (sv * siv) * aiv - prc;
    poly[q][0] = dot(w, aru); poly[q][1] = dot(w, arv); poly[q][2] = dot(w, nr) - srn;        // (x, y) on the reference face, height above it
  }
#pragma nounroll
  for (int side = 0; side < 4; side++) {
    const int cc = side >> 1; const float sg = (side & 1) ? -1.0f : 1.0f, lim = cc ? srv : sru;
    int nn = 0;
#pragma nounroll
    for (int q = 0; q < np_; q++) {
      const int q2 = (q + 1 == np_) ? 0 : q + 1;
      const float dp = lim - sg * poly[q][cc], dq = lim - sg * poly[q2][cc];
      if (dp >= 0.0f) { tmp[nn][0] = poly[q][0]; tmp[nn][1] = poly[q][1]; tmp[nn][2] = poly[q][2]; nn++; }
      if ((dp >= 0.0f) != (dq >= 0.0f)) {
        const float f = dp / (dp - dq);
#pragma unroll
        for (int k = 0; k < 3; k++) tmp[nn][k] = fmaf(f, poly[q2][k] - poly[q][k], poly[q][k]);
        nn++;
      }
    }
    np_ = nn;
#pragma nounroll
    for (int q = 0; q < np_; q++) { poly[q][0] = tmp[q][0]; poly[q][1] = tmp[q][1]; poly[q][2] = tmp[q][2]; }
    if (np_ == 0) break;
  }
#pragma nounroll
  for (int q = 0; q < np_ && ncon < 8; q++) {
    const float h = poly[q][2];
    if (h >= margin) continue;
    if (ncon == sub) { o.dist = h; o.n = nf; o.p = prc + poly[q][0] * aru + poly[q][1] * arv + (srn + 0.5f * h) * nr; o.found = true; }
    ncon++;
  }
  return o;
}
// the pair's collider (geom 1 / geom 2 in the engine's type order), and a cheap LOWER BOUND of the pair's distance for the reach
// test and the detection slack (sphere pairs: the exact distance; capsule-box: the box's three face normals as separating axes;
// box-box: the 15 axes)
// BOXBOX: the kernel family carries the box-box collider (the humanoids: one foot box on the other; the quadruped's kernels leave its
// clipping arrays out — its model has no box pair, the lowering keeps one counted-only if a model of that family ever has)
// CAPBOX: ... and the capsule-box collider (the quadruped's REGULAR kernels leave it to the replay kernel: a leg capsule reaches a
// trunk box only far beyond the joint limits, and the collider's live values cost the regular kernel 500 bytes of scratch per lane)
template <bool BOXBOX, bool CAPBOX>
LM_DEV NatOut native_contact(const NatGeom& G1, const NatGeom& G2, float margin, int sub, int& ncon) {
  if constexpr (BOXBOX) { if (G1.type == LM_GEOM_BOX) return nat_box_box(G1, G2, margin, sub, ncon); }
  if constexpr (CAPBOX) { if (G1.type == LM_GEOM_CAPSULE) return nat_capsule_box(G1, G2, margin, sub, ncon); }
  NatOut o = (G2.type == LM_GEOM_BOX) ? nat_sphere_box(G1.c, G1.rad, G2, margin) : nat_sphere_cylinder(G1.c, G1.rad, G2, margin);
  ncon = o.found ? 1 : 0;
  if (sub != 0) o.found = false;
  return o;
}
template <bool BOXBOX>
LM_DEV float native_lower_bound(const NatGeom& G1, const NatGeom& G2) {
  if (G1.type == LM_GEOM_SPHERE && G2.type == LM_GEOM_CYLINDER) {
    const V3 vec = G1.c - G2.c;
    const float x = dot(vec, G2.ax);
    const V3 pp = vec - x * G2.ax;
    const float dr = fmaxf(sqrtf(dot(pp, pp)) - G2.rad, 0.0f), dh = fmaxf(fabsf(x) - G2.half, 0.0f);
    return sqrtf(dr * dr + dh * dh) - G1.rad;
  }
  const V3 rel = G1.c - G2.c;
  if (BOXBOX && G1.type == LM_GEOM_BOX) {
    const V3 Aa[3] = {G1.ex, G1.ey, G1.ez}, Ba[3] = {G2.ex, G2.ey, G2.ez};
    const float s1[3] = {G1.hs.x, G1.hs.y, G1.hs.z}, s2[3] = {G2.hs.x, G2.hs.y, G2.hs.z};
    float gap = -3.0e38f;
    auto test_axis = [&](V3 n) {
      float r1 = 0.0f, r2 = 0.0f;
#pragma unroll
      for (int k = 0; k < 3; k++) { r1 += s1[k] * fabsf(dot(n, Aa[k])); r2 += s2[k] * fabsf(dot(n, Ba[k])); }
      gap = fmaxf(gap, fabsf(dot(rel, n)) - r1 - r2);
    };
#pragma unroll
    for (int k = 0; k < 3; k++) { test_axis(Aa[k]); test_axis(Ba[k]); }
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
      for (int l = 0; l < 3; l++) {
        const V3 cr = cross(Aa[k], Ba[l]);
        const float n2 = dot(cr, cr);
        if (n2 > 1e-12f) test_axis((1.0f / sqrtf(n2)) * cr);
      }
    return gap;
  }
  // sphere (half = 0) or capsule against a box: the box's face normals as separating axes
  const float p[3] = {dot(G2.ex, rel), dot(G2.ey, rel), dot(G2.ez, rel)}, hs[3] = {G2.hs.x, G2.hs.y, G2.hs.z};
  const float a[3] = {dot(G2.ex, G1.ax) * G1.half, dot(G2.ey, G1.ax) * G1.half, dot(G2.ez, G1.ax) * G1.half};
  float gap = -3.0e38f;
#pragma unroll
  for (int k = 0; k < 3; k++) gap = fmaxf(gap, fabsf(p[k]) - hs[k] - fabsf(a[k]));
  return gap - G1.rad;
}

// ---- the substep ---------------------------------------------------------------------------------------------
// cm: constant table (LDS), c: chain id of this lane. State in/out: root (replicated) + chain.
// actr/actc: actuator forces (already gear*clamped ctrl) per root / chain dof.
// lmem/ls: lane-local scratch of LaneMem<MC,NS>::kSize floats, element i at lmem[i*ls].
// forward dynamics at (q, v): constrained acceleration in war/wac (in: warm start, out: qacc). With EULER the state
// is also advanced by one semi-implicit Euler step (implicit joint damping), otherwise q, v are left untouched.
// CONE: LM_CONE_PYRAMIDAL / LM_CONE_ELLIPTIC compiles the other cone's code out; -1 reads it from P.cone
// NM > 0: the chain's muscles (table `mt`, lm_layout.h MT_*/MU_*) act on the chain dofs; their activations and
// controls live in lane memory (kAct/kCtrl, filled by the caller) and are advanced here when EULER.
// DR: joint damping / stiffness / frictionloss come from `dp` (per environment) instead of the constant table.
// PM: 0 no self-collisions, 1 the pair pass with the convex collider, 2 the pair pass WITHOUT it (a convex pair within reach sets
// cnt.need_full: the quadruped's kernels, whose regular gaits never touch one), 3 DETECTION ONLY: the broad / mid phase of the pair
// pass without any of its contact machinery — a geom pair within reach sets cnt.need_full and the control step goes to the
// family's replay kernel (PM 1). The six-link family's regular kernels: UnitreeG1's 255 link pairs and the cross blocks of 6 x 6
// chains do not fit beside its lane memory, and its gaits have no self-contact
template <class Q, int MC, int NS, bool EULER, int CONE = -1, int NM = 0, int DR = 0, int PM = 0>
LM_DEV void forward(const float* cm, int c, const Params& P, float* qr, float* vr, float* qc, float* vc,
                    float* war, float* wac, const float* actr, const float* actc, LM_LMEM_T* lmem, int ls,
                    Counters& cnt, const Debug* dbg, const float* mt = nullptr, const DofPrm<MC>* dp = nullptr,
                    bool want_grf = false, float* pair_slack = nullptr) {
  // `oz` is an opaque zero (LM_OPAQUE_ZERO, refreshed per loop iteration): constant-table reads are indexed
  // through it so that the compiler re-reads them from LDS where they are used instead of hoisting hundreds of
  // loop-invariant constants into registers across the Newton / line-search loops (that is what spilled).
  constexpr bool PAIRS = PM == 1 || PM == 2, NOMPR = PM == 2, DETECT = PM != 0, DETECT_ONLY = PM == 3;
#ifdef LM_NO_DEFER
  constexpr bool DEFER = false;
#else
  constexpr bool DEFER = PAIRS && MC >= 5;        // (the quadruped's kernel — three links, no scratch — keeps its kinematics in one loop)
#endif
  // the link-pair lists: in the constant table (LDS) — the six-link kernels' in its global copy (Params::cmg)
#define LPE(off, i, f) ((MC == 6) ? P.cmg[(off) + (i) * LM_LP_SIZE + (f)] : cm[oz + (off) + (i) * LM_LP_SIZE + (f)])
#ifdef LM_A1_CAPBOX_INLINE
  constexpr bool kCapBox = true;
#else
  constexpr bool kCapBox = MC >= 5 || NS > 8 || CONE < 0;      // every humanoid kernel, the quadruped's replay kernel and its forward-only (debug) kernel
#endif
  int oz = LM_OPAQUE_ZERO();
  const bool pyramidal = (CONE < 0) ? (P.cone == 0) : (CONE == 0);
  // CONE == LM_CONE_PYRAMIDAL promises that every contact of the model is a condim-3 pyramid (checked by the launcher):
  // the elliptic code paths then compile out
#define PYR3(dim) (CONE == 0 || (pyramidal && (dim) == 3))
  const float* rb = cm + LM_CM_ROOT;
#define RD(k, f) rb[oz + LM_R_DOFS + (k) * LM_D_SIZE + (f)]
#define CH(f) cm[oz + LM_CM_CHAINS + (f) * LM_NCHAIN + c]
#define LK(k, f) CH(LM_C_LINKS + (k) * LM_LINK_SIZE + (f))
#define LX(k, f) LK(k, LM_D_SIZE + (f))
  // DR: 0 = the table's joint parameters, 1 = per-environment damping / stiffness / frictionloss, 2 = + the environment's MODEL
  // VARIANT: inertial numbers, armature, invweights and the termination scale come from its record in global memory, the
  // geom tables from its variant (a compile-time level: the run-time choice cost every DR kernel 700 B of scratch)
  constexpr bool inr_on = DR == 2;
  const float* gtp = inr_on ? dp->gt : P.gt;
  const float* gptp = inr_on ? dp->gpt : P.gpt;
  float* const mprc = (PAIRS && MC >= 5 && dp) ? dp->mprc : nullptr;       // (the quadruped's convex pairs are primitives: nothing to warm up)
#define INR(i) dp->inr[(i) * LM_NCHAIN + c]
#define LXI(k, j) (inr_on ? INR((k) * LM_IR_LINK + (j)) : LX(k, LM_L_MASS + (j)))        /* j: mass, com xyz, inertia xx yy zz xy xz yz */
#define LKV(k, j, f) (inr_on ? INR((k) * LM_IR_LINK + 10 + (j)) : LK(k, f))               /* j: 0 armature, 1 invweight (2 = friction-loss R: dp->rfl_*) */
#define RBI(j) (inr_on ? INR(LM_IR_ROOT + (j)) : rb[LM_R_MASS + (j)])
#define RDV(i, j, f) (inr_on ? INR(LM_IR_ROOT_DOF + 3 * (i) + (j)) : RD(i, f))
  const float nscale = inr_on ? INR(LM_IR_SCALE) : P.scale;
#define GE(g, f) gtp[((g) * LM_G_SIZE + (f)) * LM_NCHAIN + c]
  // the chain's tail lists (prune records, collider-less geoms, geom groups, link pairs): contiguous per chain, start in the chain block
// (six-link kernels: the prune records and the link groups are read from the constant table's copy in GLOBAL memory, like their link-pair
// lists — lowering.py ends H_CM_USED in front of them: 1.8 KB of LDS that decide between three and four workgroups per CU)
#define GP(g, f) ((MC == 6) ? P.cmg[(int)CH(LM_C_OFF_PRUNE) + (g) * LM_P_SIZE + (f)] : cm[oz + (int)CH(LM_C_OFF_PRUNE) + (g) * LM_P_SIZE + (f)])
#define CU(i, f) cm[oz + (int)CH(LM_C_OFF_CUNSUP) + (i) * LM_U_SIZE + (f)]
#define SL(s, f) lmem[((s) * LMm::kSlot + (f)) * ls]
#define PEER(dl, i) Q::peer(lmem, ls, (i), (dl))
#define DAMP_R(i) (DR ? dp->damp[(long long)(int)RD(i, LM_D_DOF) * dp->stride] : RD(i, LM_D_DAMP))
#define STIFF_R(i) (DR ? dp->stiff[(long long)(int)RD(i, LM_D_DOF) * dp->stride] : RD(i, LM_D_STIFF))
#define FLOSS_R(i) (DR ? dp->floss_r[i] : RD(i, LM_D_FLOSS))
#define DUPK(k) (MC == 6 && duprole < 0 && (k) == 0)       /* the copy of a shared link carries no joint parameters */
#define DAMP_C(k) (DR ? (DUPK(k) ? 0.0f : dp->damp[(long long)(int)LK(k, LM_D_DOF) * dp->stride]) : LK(k, LM_D_DAMP))
#define STIFF_C(k) (DR ? (DUPK(k) ? 0.0f : dp->stiff[(long long)(int)LK(k, LM_D_DOF) * dp->stride]) : LK(k, LM_D_STIFF))
#define FLOSS_C(k) (DR ? (DUPK(k) ? 0.0f : dp->floss_c[k]) : LK(k, LM_D_FLOSS))
  const float w0 = (c == 0) ? 1.0f : 0.0f;    // root rows are replicated in all lanes, counted once
  const int nl = (int)CH(LM_C_NLINKS);
  // chains that share their first link (compiled for the six-link family only): +1 owner, -1 massless copy, see tie_shared_dof
  const int duprole = (MC == 6) ? (int)CH(LM_C_DUPROLE) : 0;
  const bool anydup = (MC == 6) && Q::sum(fabsf((float)duprole)) > 0.0f;
  LM_TICK_INIT();

  // ================= position stage: kinematics, twists, inertias, contacts =================
  M3 R; V3 p;
#pragma unroll
  for (int i = 0; i < 9; i++) R.a[i] = rb[LM_R_R0 + i];
  p = v3(rb[LM_R_TX], rb[LM_R_TY], rb[LM_R_TZ]);
  V3 ru[6], ra[6];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    float t = RD(k, LM_D_TYPE);
    V3 al = v3(RD(k, LM_D_PX), RD(k, LM_D_PY), RD(k, LM_D_PZ));
    ra[k] = p + mul(R, al);
    ru[k] = mul(R, v3(RD(k, LM_D_AX), RD(k, LM_D_AY), RD(k, LM_D_AZ)));
    if (t != 0.0f) { rotate_world(R, ru[k], qr[k]); p = ra[k] - mul(R, al); }
    else p = p + qr[k] * ru[k];
  }
  const V3 O = p;
  Sp Sr[6];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    if (RD(k, LM_D_TYPE) != 0.0f) { Sr[k].w = ru[k]; Sr[k].v = cross(ru[k], O - ra[k]); }
    else { Sr[k].w = v3(0, 0, 0); Sr[k].v = ru[k]; }
  }
  // root body inertia about O
  SpI Iroot = spi0();
  auto root_inertia = [&]() {
    float Iw[6], Il[6] = {RBI(4), RBI(5), RBI(6), RBI(7), RBI(8), RBI(9)};
    rotate_inertia(R, Il, Iw);
    Iroot = make_spi(RBI(0), mul(R, v3(RBI(1), RBI(2), RBI(3))), Iw);
  };
  // root velocity / acceleration recursion (replicated), base acceleration = -gravity
  Sp Vroot = sp0(), Aroot; Aroot.w = v3(0, 0, 0); Aroot.v = v3(-P.g.x, -P.g.y, -P.g.z);
  using LMm = LaneMemFor<MC, NS, NM, PAIRS, CONE>;
#define LMEM(i) lmem[(i) * ls]
  if constexpr (!DEFER) {
    root_inertia();
#pragma unroll
    for (int k = 0; k < 6; k++) {
      Sp Sd; Sd.w = cross(Vroot.w, Sr[k].w); Sd.v = cross(Vroot.w, Sr[k].v) + cross(Vroot.v, Sr[k].w);
      Vroot = Vroot + vr[k] * Sr[k];
      Aroot = Aroot + vr[k] * Sd;
    }
  } else {
    // DEFER: the root's twists go to lane memory here (striped, LaneMem::kRootS) and come back behind the pair pass, where the
    // acceleration recursion and the root inertia are built; only the root velocity is needed by the collision passes
#pragma unroll
    for (int k = 0; k < 6; k++) {
      Vroot = Vroot + vr[k] * Sr[k];
      const float s6[6] = {Sr[k].w.x, Sr[k].w.y, Sr[k].w.z, Sr[k].v.x, Sr[k].v.y, Sr[k].v.z};
#pragma unroll
      for (int j = 0; j < 6; j++) if (((k * 6 + j) & 3) == c) LMEM(LMm::kRootS + ((k * 6 + j) >> 2)) = s6[j];
    }
    Q::quad_sync();
  }
  // a root twist from lane memory: striped over the environment's four columns (DEFER), or this lane's own copy
  auto ldSr = [&](int r) -> Sp {
    Sp S;
    if constexpr (DEFER) {
      float s6[6];
#pragma unroll
      for (int j = 0; j < 6; j++) s6[j] = Q::peer(lmem, ls, LMm::kRootS + ((r * 6 + j) >> 2), ((r * 6 + j) & 3) - c);
      S.w = v3(s6[0], s6[1], s6[2]); S.v = v3(s6[3], s6[4], s6[5]);
    } else {
      const int b_ = LMm::kSr + r * 6;
      S.w = v3(LMEM(b_), LMEM(b_ + 1), LMEM(b_ + 2)); S.v = v3(LMEM(b_ + 3), LMEM(b_ + 4), LMEM(b_ + 5));
    }
    return S;
  };
  // collider-less root geoms that reach the floor are counted, not simulated
  {
    int nu = (int)rb[LM_R_NUNSUP];
    for (int i = c; i < nu; i += 4) {            // the quad's lanes share the list
      const float* u = rb + P.off_runsup + i * LM_U_SIZE;
      const float sz = O.z + R.a[6] * u[1] + R.a[7] * u[2] + R.a[8] * u[3];
      if (sz - u[4] < u[5]) cnt.unhandled++;
    }
  }

  // chain: kinematics + velocity recursion + link inertias; floor contacts are recorded into the slots.
  // Everything that must survive into the solver (M, twists) is parked in lane memory so that the Newton loop
  // keeps only small vectors in registers.
  // slot field offsets of THIS kernel's record layout (shadow the namespace-scope enumerators of the full layout)
  constexpr bool kCompactSlots = (CONE == 0);
  constexpr int SL_D = kCompactSlots ? (int)SLC_D : (int)lm::SL_D, SL_FR = kCompactSlots ? (int)SLC_AREF : (int)lm::SL_FR;
  constexpr int SL_AREF = kCompactSlots ? (int)SLC_AREF : (int)lm::SL_AREF, SL_JAR = kCompactSlots ? (int)SLC_JAR : (int)lm::SL_JAR;
  constexpr int SL_JV = kCompactSlots ? (int)SLC_JV : (int)lm::SL_JV, SL_ZONE = kCompactSlots ? (int)SLC_ZONE : (int)lm::SL_ZONE;
  constexpr int SL_GRF = kCompactSlots ? (int)SLC_GRF : (int)lm::SL_GRF;
  constexpr int SL_PREP = kCompactSlots ? 0 : (int)lm::SL_PREP;          // (no elliptic contacts in the kernels with compact slot records)
  constexpr int SL_PART = kCompactSlots ? (int)SLC_SIZE : (int)lm::SL_PART;
  constexpr int SL_NX = SL_PART + 1, SL_NY = SL_PART + 2, SL_NZ = SL_PART + 3;
  (void)SL_FR;
  float bias_c[MC], bias_r[6];
  float a0r[6], a0c[MC];
  float sm_r[6], sm_c[MC];
  int nslot = 0;
  int pair_mask_out = 0;           // partner lanes of this lane's cross-chain contact slots (PAIRS)
  int nrootslot = 0, npairslot = 0; // slots of root-body geoms / of self-contacts held by this lane
  float pair_speed = 0.0f;          // largest speed of a point of this chain's links against the root body (PAIRS)
  int nfloor = 0;                   // slots [0, nfloor) are floor contacts, [nfloor, nslot) self-contacts: the loops over the
                                    // slots run the lean floor code first and the general-frame code for the rest
  {
    Sp Sc[MC];
    float Mcc[MC * (MC + 1) / 2], Mcr[MC][6], Mrr[21];
    Sp Vc[MC], Ac[MC];
    SpI Ic[MC];
    {
      M3 Rk = R; V3 pk = O;
      Sp V = Vroot, A = Aroot;
      const int nun = (int)CH(LM_C_NUNSUP);
#pragma unroll
      for (int k = 0; k < MC; k++) {
        if (k < nl) {
          M3 T;
#pragma unroll
          for (int i = 0; i < 9; i++) T.a[i] = LX(k, LM_L_R0 + i);
          pk = pk + mul(Rk, v3(LX(k, LM_L_TX), LX(k, LM_L_TY), LX(k, LM_L_TZ)));
          Rk = mul(Rk, T);
          float t = LK(k, LM_D_TYPE);
          V3 al = v3(LK(k, LM_D_PX), LK(k, LM_D_PY), LK(k, LM_D_PZ));
          V3 aw = pk + mul(Rk, al);
          V3 uw = mul(Rk, v3(LK(k, LM_D_AX), LK(k, LM_D_AY), LK(k, LM_D_AZ)));
          Sp Sk;
          if (t != 0.0f) { rotate_world(Rk, uw, qc[k]); pk = aw - mul(Rk, al); Sk.w = uw; Sk.v = cross(uw, O - aw); }
          else { pk = pk + qc[k] * uw; Sk.w = v3(0, 0, 0); Sk.v = uw; }
          if constexpr (!DEFER) {
            Sp Sd; Sd.w = cross(V.w, Sk.w); Sd.v = cross(V.w, Sk.v) + cross(V.v, Sk.w);
            A = A + vc[k] * Sd;
          }
          V = V + vc[k] * Sk;
          if constexpr (!DEFER) {
            Sc[k] = Sk; Vc[k] = V; Ac[k] = A;
            float Il[6], Iw[6];
#pragma unroll
            for (int i = 0; i < 6; i++) Il[i] = LXI(k, 4 + i);
            rotate_inertia(Rk, Il, Iw);
            Ic[k] = make_spi(LXI(k, 0), pk + mul(Rk, v3(LXI(k, 1), LXI(k, 2), LXI(k, 3))) - O, Iw);
          }
          LMEM(LMm::kFrame + k * 18 + 0) = pk.x; LMEM(LMm::kFrame + k * 18 + 1) = pk.y; LMEM(LMm::kFrame + k * 18 + 2) = pk.z;
#pragma unroll
          for (int i = 0; i < 9; i++) LMEM(LMm::kFrame + k * 18 + 3 + i) = Rk.a[i];
          LMEM(LMm::kFrame + k * 18 + 12) = V.w.x; LMEM(LMm::kFrame + k * 18 + 13) = V.w.y; LMEM(LMm::kFrame + k * 18 + 14) = V.w.z;
          LMEM(LMm::kFrame + k * 18 + 15) = V.v.x; LMEM(LMm::kFrame + k * 18 + 16) = V.v.y; LMEM(LMm::kFrame + k * 18 + 17) = V.v.z;
          if (DETECT) {
            // self-collision broad phase: world centre of the link's bounding sphere, and the speed of the link's points
            // against the root body, |v_c| + |w| r with the twist relative to the root (the detection's travel bound)
            const V3 cw = pk + mul(Rk, v3(LX(k, LM_L_BSX), LX(k, LM_L_BSY), LX(k, LM_L_BSZ)));
            if (PAIRS && MC != 6) { LMEM(LMm::kBS + k * 3 + 0) = cw.x; LMEM(LMm::kBS + k * 3 + 1) = cw.y; LMEM(LMm::kBS + k * 3 + 2) = cw.z; }      // (detection only: from the frames, bs_centre)
            const V3 wr = V.w - Vroot.w;
            const V3 vcr = V.v - Vroot.v + cross(wr, cw - O);
            pair_speed = fmaxf(pair_speed, sqrtf(dot(vcr, vcr)) + sqrtf(dot(wr, wr)) * LX(k, LM_L_BSR));
          }
        } else if constexpr (!DEFER) { Sc[k] = sp0(); Vc[k] = V; Ac[k] = A; Ic[k] = spi0(); }
      }
      LM_TICK(11);     // kinematics, twists, link inertias
  // floor contacts of this chain's geoms (plane z = 0, normal +z), each in the frame of its link. The replicas of a
  // small-batch environment take every kRep-th geom; in each pass they exchange how many contacts they found, so that
  // the slot records land in lane memory in the same order (geom, then candidate point) as without replicas.
      int n_overflow = 0, n_unhandled = 0;
      // one floor contact of geom g (link k, twist V) at (px, py), `dist` above the plane -> slot record
      auto emit_floor_slot = [&](int slot, int g, int k, const Sp& V, float margin, float px, float py, float dist) {
          // contact point (midway between the surfaces) relative to O; row parameters
          V3 cp = v3(px, py, 0.5f * dist) - O;
          float imp = impedance(&GE(g, LM_G_S0), LM_NCHAIN, dist, margin);
          float D0 = imp / fmaxf(kMinVal, (1.0f - imp) * GE(g, LM_G_TRAN));
          float vel[6];
          contact_rows(V, cp, vel);
          const float B = GE(g, LM_G_B), Kr = GE(g, LM_G_K) * imp * (dist - margin), mu = GE(g, LM_G_MU);
          const int dim = (int)GE(g, LM_G_DIM);
          SL(slot, SL_LINK) = (float)k; SL(slot, SL_DIM) = (float)dim; SL(slot, SL_MU) = mu;
          SL(slot, SL_GRF) = GE(g, LM_G_GRF);
          if (PAIRS) SL(slot, SL_PART) = 0.0f;
          SL(slot, SL_RX) = cp.x; SL(slot, SL_RY) = cp.y; SL(slot, SL_RZ) = cp.z;
          SL(slot, SL_D) = D0;
          if (PYR3(dim)) {
            float xv[4];
            pyr_rows(vel, mu, xv);
#pragma unroll
            for (int r = 0; r < 4; r++) SL(slot, SL_AREF + r) = -B * xv[r] - Kr;
          } else {
#pragma unroll
            for (int j2 = 1; j2 < 6; j2++) { SL(slot, SL_D + j2) = (j2 < dim) ? D0 / GE(g, LM_G_RR1 + j2 - 1) : 0.0f; SL(slot, SL_FR + j2 - 1) = GE(g, LM_G_F0 + j2 - 1); }
#pragma unroll
            for (int j2 = 0; j2 < 6; j2++) SL(slot, SL_AREF + j2) = -B * vel[j2] - ((j2 == 0) ? Kr : 0.0f);
          }
      };
      // two levels: the geoms of a link (or of the root body: link -1, this lane's share) form a group with a bounding sphere;
      // a link high above the floor costs one test per pass
      const int ngroups = (int)CH(LM_C_NLGROUP), off_lgroup_c = (int)CH(LM_C_OFF_LGROUP);
      for (int gi = 0; gi < ngroups; gi++) {
#define LG(f) ((MC == 6) ? P.cmg[off_lgroup_c + gi * LM_LG_SIZE + (f)] : cm[oz + off_lgroup_c + gi * LM_LG_SIZE + (f)])
        const int glink = (int)LG(0), gfirst = (int)LG(1), gend = gfirst + (int)LG(2);
        {
          const int fbg = LMm::kFrame + (glink < 0 ? 0 : glink) * 18;
          const V3 gc = v3(LG(3), LG(4), LG(5));
          const float zg = (glink < 0) ? O.z + R.a[6] * gc.x + R.a[7] * gc.y + R.a[8] * gc.z
                                       : LMEM(fbg + 2) + LMEM(fbg + 9) * gc.x + LMEM(fbg + 10) * gc.y + LMEM(fbg + 11) * gc.z;
          if (zg - LG(6) > 0.0f) continue;
        }
#undef LG
      for (int g0 = gfirst; g0 < gend; g0 += Q::kRep) {
        const int g = g0 + Q::rep();
        const int ng = gend;
        // ---- candidate points of my geom: sphere centre | the two capsule end centres | the box corners below the box
        // centre in bit order, at most 4 contacts per box
        int made = 0;
        float cx[4], cy[4], cd[4];
        int k = 0;
        Sp V = sp0();
        float margin = 0.0f;
        if (g < ng && (int)GP(g, 5) != LM_GEOM_MESH) {
          k = (int)GP(g, 0);               // link of the geom; -1: a geom of the root body, dealt to this chain's lane
          const int fb = LMm::kFrame + (k < 0 ? 0 : k) * 18;
          const V3 gl = v3(GP(g, 1), GP(g, 2), GP(g, 3));
          // margin-less bounding-sphere prune on the height of the geom centre (third row of the link rotation); only a
          // geom that passes it reads its full record from the geom table
          const float zc = (k < 0) ? O.z + R.a[6] * gl.x + R.a[7] * gl.y + R.a[8] * gl.z
                                   : LMEM(fb + 2) + LMEM(fb + 9) * gl.x + LMEM(fb + 10) * gl.y + LMEM(fb + 11) * gl.z;
          if (!(zc - GP(g, 4) > 0.0f)) {
            V3 pk = O;
            M3 Rk = R;
            V = Vroot;
            if (k >= 0) {
              pk = v3(LMEM(fb), LMEM(fb + 1), LMEM(fb + 2));
#pragma unroll
              for (int i = 0; i < 9; i++) Rk.a[i] = LMEM(fb + 3 + i);
              V.w = v3(LMEM(fb + 12), LMEM(fb + 13), LMEM(fb + 14)); V.v = v3(LMEM(fb + 15), LMEM(fb + 16), LMEM(fb + 17));
            }
            const V3 ctr = pk + mul(Rk, gl);
            const float rad = GE(g, LM_G_RADIUS), half = GE(g, LM_G_HALF);
            margin = GE(g, LM_G_MARGIN);
            const int gtype = (int)GE(g, LM_G_TYPE);
            const V3 ax = mul(Rk, v3(GE(g, LM_G_AX), GE(g, LM_G_AY), GE(g, LM_G_AZ)));
            M3 Rg;
            if (gtype == LM_GEOM_BOX || gtype == LM_GEOM_CYLINDER) {
              M3 Gr;
#pragma unroll
              for (int i = 0; i < 9; i++) Gr.a[i] = GE(g, LM_G_R0 + i);
              Rg = mul(Rk, Gr);
            }
            // plane vs cylinder (engine: mjc_PlaneCylinder): deepest rim point of the cap facing the plane, the rim point
            // below it on the other cap, and two more points of the near cap's rim at +-120 degrees
            V3 cax = ax, cvec = v3(0, 0, 0), cv1 = v3(0, 0, 0);
            if (gtype == LM_GEOM_CYLINDER) {
              float prj = cax.z;
              if (prj > 0.0f) { cax = -1.0f * cax; prj = -prj; }
              cvec = v3(cax.x * prj, cax.y * prj, cax.z * prj - 1.0f);
              const float l2 = dot(cvec, cvec);
              if (l2 >= 1e-30f) cvec = (rad / sqrtf(l2)) * cvec;
              else cvec = rad * v3(Rg.a[0], Rg.a[3], Rg.a[6]);        // disk parallel to the plane: the geom's x axis
              cax = half * cax;
              V3 side = cross(cvec, cax);
              cv1 = (rad * 0.8660254037844386f / sqrtf(fmaxf(dot(side, side), 1e-30f))) * side;
            }
            const int npt = (gtype == LM_GEOM_CAPSULE) ? 2 : ((gtype == LM_GEOM_BOX) ? 8 : ((gtype == LM_GEOM_CYLINDER) ? 4 : 1));
            for (int e = 0; e < npt; e++) {
              V3 sc = ctr; float rad_e = rad;
              if (gtype == LM_GEOM_CAPSULE) sc = ctr + ((e == 0) ? half : -half) * ax;
              else if (gtype == LM_GEOM_BOX) {
                V3 off = mul(Rg, v3((e & 1) ? GE(g, LM_G_SX) : -GE(g, LM_G_SX), (e & 2) ? GE(g, LM_G_SY) : -GE(g, LM_G_SY),
                                    (e & 4) ? GE(g, LM_G_SZ) : -GE(g, LM_G_SZ)));
                if (off.z > 0.0f || made >= 4) continue;
                sc = ctr + off; rad_e = 0.0f;
              } else if (gtype == LM_GEOM_CYLINDER) {
                rad_e = 0.0f;
                if (e == 0) sc = ctr + cvec + cax;
                else if (made == 0) break;                             // the deepest point is out of reach: no contact at all
                else if (e == 1) sc = ctr + cvec + (-1.0f) * cax;
                else sc = ctr + cax + (-0.5f) * cvec + ((e == 2) ? 1.0f : -1.0f) * cv1;
              }
              const float dist = sc.z - rad_e;
              if (dist >= margin) continue;
#pragma unroll
              for (int j = 0; j < 4; j++) if (made == j) { cx[j] = sc.x; cy[j] = sc.y; cd[j] = dist; }
              made++;
            }
          }
        }
        // ---- where do my contacts go: after those of the replicas with a smaller geom index in this pass
        int before = 0, total = made;
        if (Q::kRep > 1) {
          total = 0;
#pragma unroll
          for (int r = 0; r < Q::kRep; r++) { const int n_r = (int)Q::rep_bcast((float)made, r); if (r < Q::rep()) before += n_r; total += n_r; }
        }
        for (int j = 0; j < made; j++) {
          const int slot = nslot + before + j;
          if (slot >= NS) { n_overflow++; continue; }
          float px = cx[0], py = cy[0], dist = cd[0];
#pragma unroll
          for (int q = 1; q < 4; q++) if (j == q) { px = cx[q]; py = cy[q]; dist = cd[q]; }
          emit_floor_slot(slot, g, k, V, margin, px, py, dist);
        }
        nslot += total;
      }
      LM_TICK(12);     // broad phase + primitive colliders of the group
      // ---- convex meshes of the group (plane vs hull: a contact at the support vertex — pinned by the UnitreeH1 golden rows,
      // DESIGN.md — and up to three more at its hull-graph neighbours, below): every replica takes the same geom and a quarter of its hull vertices (mesh-vertex table, global
      // memory, link frame); the lowest vertex wins, ties go to the first in the table like a sequential search
      for (int g = gfirst; g < gend; g++) {
        if ((int)GP(g, 5) != LM_GEOM_MESH) continue;
        const int k = (int)GP(g, 0);
        const int fb = LMm::kFrame + (k < 0 ? 0 : k) * 18;
        const V3 gl = v3(GP(g, 1), GP(g, 2), GP(g, 3));
        V3 pk = O; M3 Rk = R; Sp V = Vroot;
        if (k >= 0) {
          pk = v3(LMEM(fb), LMEM(fb + 1), LMEM(fb + 2));
#pragma unroll
          for (int i = 0; i < 9; i++) Rk.a[i] = LMEM(fb + 3 + i);
          V.w = v3(LMEM(fb + 12), LMEM(fb + 13), LMEM(fb + 14)); V.v = v3(LMEM(fb + 15), LMEM(fb + 16), LMEM(fb + 17));
        }
        if (pk.z + Rk.a[6] * gl.x + Rk.a[7] * gl.y + Rk.a[8] * gl.z - GP(g, 4) > 0.0f) continue;      // bounding sphere above the floor
        const int v0 = (int)GE(g, LM_G_SX), nvert = (int)GE(g, LM_G_SY);
        const V3 nl = v3(Rk.a[6], Rk.a[7], Rk.a[8]);       // the plane normal in the link frame (third row of the rotation)
        float best = 3.0e38f; int ibest = 0x7fffffff;
        for (int i = Q::rep(); i < nvert; i += Q::kRep) {
          const float* vp = P.meshv + 4 * (v0 + i);
          const float d = nl.x * vp[0] + nl.y * vp[1] + nl.z * vp[2];
          if (d < best) { best = d; ibest = i; }
        }
        if (Q::kRep > 1) {
          float gb = best; int gi = ibest;
#pragma unroll
          for (int r = 0; r < Q::kRep; r++) {
            const float bd = Q::rep_bcast(best, r); const int bi = (int)Q::rep_bcast((float)ibest, r);
            if (bd < gb || (bd == gb && bi < gi)) { gb = bd; gi = bi; }
          }
          best = gb; ibest = gi;
        }
        if (ibest >= nvert) continue;
        const float* vp = P.meshv + 4 * (v0 + ibest);
        const V3 sv = pk + mul(Rk, v3(vp[0], vp[1], vp[2]));
        const float margin = GE(g, LM_G_MARGIN);
        if (!(sv.z < margin)) continue;
        if (nslot >= NS) { n_overflow += (Q::rep() == 0) ? 1 : 0; continue; }
        emit_floor_slot(nslot, g, k, V, margin, sv.x, sv.y, sv.z);
        nslot++;
        // further contacts at the hull-graph neighbours of the support vertex (the engine's "up to 3 more contacts from mesh",
        // DESIGN.md §2 item 10): penetrating, nearest first, none closer than G_SZ to the SUPPORT contact. Every replica
        // walks the (short) list itself, so all of them write the same slots.
        {
          const float tol2 = GE(g, LM_G_SZ) * GE(g, LM_G_SZ);
          const V3 cp0 = v3(sv.x, sv.y, 0.5f * sv.z);
          int nc = 1;
          for (int e = (int)vp[3]; nc < 4; e++) {
            const int j = (int)P.meshn[e];
            if (j < 0) break;
            const float* vj = P.meshv + 4 * (v0 + j);
            const V3 wj = pk + mul(Rk, v3(vj[0], vj[1], vj[2]));
            if (wj.z > margin) continue;
            const V3 dj = v3(wj.x, wj.y, 0.5f * wj.z) - cp0;
            if (dot(dj, dj) < tol2) continue;
            nc++;
            if (nslot >= NS) { n_overflow += (Q::rep() == 0) ? 1 : 0; continue; }
            emit_floor_slot(nslot, g, k, V, margin, wj.x, wj.y, wj.z);
            nslot++;
          }
        }
      }
      }
      if (nslot > NS) nslot = NS;
      Q::fence();                // slot records written by one replica are read by all of them from here on
      for (int i = Q::rep(); i < nun; i += Q::kRep) {
        const int fb = LMm::kFrame + (int)CU(i, 0) * 18;
        const float sz = LMEM(fb + 2) + LMEM(fb + 9) * CU(i, 1) + LMEM(fb + 10) * CU(i, 2) + LMEM(fb + 11) * CU(i, 3);
        if (sz - CU(i, 4) < CU(i, 5)) n_unhandled++;
      }
      LM_TICK(13);     // hull colliders (and the group loop's tail)
      if (Q::kRep > 1) { n_overflow = (int)Q::rep_sum((float)n_overflow); n_unhandled = (int)Q::rep_sum((float)n_unhandled); }
      cnt.overflow += n_overflow; cnt.unhandled += n_unhandled;
    }
    // ======== self-collisions (PAIRS): geom pairs between two links of the chains, or a link and the root body ========
    // Per forward pass (when due, below): link pairs by their bounding spheres (one list entry per pair, in one of its two lanes),
    // their body pairs by bounding capsules (body-pair table, global memory), their geom pairs by the closest points of the two
    // bounding-capsule segments (geom-pair table): a sphere / capsule pair closer than its margin is a closed-form contact, a pair
    // the engine sends to its general convex collider goes through MPR, a pair with one of the engine's native box colliders is
    // counted. The lane that tested a cross-chain pair hands the contact to the partner lane; each records it as a slot of its own
    // ("mirror" slots: same point, frame and parameters, opposite sign); every replica records all of them (identical words to
    // identical addresses). The pass itself is described where its work lists are declared.
    int pair_mask = 0;               // partner lanes of this lane's cross-chain slots
    // The pass is SKIPPED while it provably cannot find anything: `pair_slack` is the smallest gap (distance minus margin)
    // over all pairs of this quad at the last detection, less the distance the links can have travelled since — per substep
    // h x (bound on the speed of my links against the root + the largest such bound in the quad); with semi-implicit Euler
    // the positions of a substep move by exactly h x the velocities this pass starts from. In a normal gait only the
    // trunk-thigh pairs are a few centimetres apart: one detection every 5-10 substeps.
    nfloor = nslot;
    bool detect = DETECT, first_detect = true;
    float gap_min = 3.0e38f;
    if (DETECT) {
      const float s_own = pair_speed;
      const float s_quad = fmaxf(fmaxf(Q::quad_read(s_own, 0), Q::quad_read(s_own, 1)), fmaxf(Q::quad_read(s_own, 2), Q::quad_read(s_own, 3)));
      first_detect = !pair_slack || *pair_slack == 0.0f;        // nothing known yet (fresh state: the caller starts the slack at 0)
      // Euler: the positions of this pass lie h x (this pass's velocities) from the next pass's. RK4 evaluates its stages at
      // q0 + a h v(previous stage) and ends at q0 + h/6 (v1 + 2 v2 + 2 v3 + v4): two consecutive evaluation points are at most
      // 1.5 h x (the largest speed bound among the passes involved) apart — all of them lie in this substep or the one before,
      // whose maxima are carried in pair_slack[1] (this substep so far) and [2] (the previous one, moved there by substep()); the
      // factor 2 leaves room for the second-order terms.
      float travel = P.h * (s_own + s_quad);
      if (!EULER && pair_slack) {
        pair_slack[1] = fmaxf(pair_slack[1], s_own + s_quad);
        travel = 2.0f * P.h * fmaxf(pair_slack[1], pair_slack[2]);
      }
      float slack = (pair_slack ? *pair_slack : 0.0f) - travel;
      // WAVE-uniform: when one environment of the wave has to detect, its wave mates detect with it — they would wait for it anyway
      // (one instruction stream), and their slack is refreshed for free, so the wave as a whole detects about as often as its
      // neediest environment instead of whenever ANY of the four is due. A detection that was not due finds nothing (that is what
      // the slack guarantees): the states are bitwise the same, whatever the composition of the wave.
      detect = Q::any(slack <= 0.0f);
      if (pair_slack) *pair_slack = slack;
    }
#ifdef LM_NO_DETECT
    detect = false;
#endif
#ifdef LM_TIMERS
    if (DETECT && c == 0 && Q::rep() == 0) { cnt.m[9]++; if (detect) cnt.m[8]++; }
#endif
    if (DETECT && detect) {
      if (c == 0 && Q::rep() == 0) cnt.pair_passes++;
      Q::quad_sync();                // the peers' frames and sphere centres are read below
      const V3 rootc = O + mul(R, v3(rb[LM_R_BSX], rb[LM_R_BSY], rb[LM_R_BSZ]));
      const int nlp = (int)CH(LM_C_NLPAIR), off_lpair_c = (int)CH(LM_C_OFF_LPAIR);
      int n_over = 0;
      // ---- The pass is a sequence of FLAT work lists, each dealt to all lanes of the environment (4 chains x kRep replicas), because
      // every level costs a round trip to global memory (body-pair table, geom-pair table, hull vertices) and nested loops pay them one
      // after the other:
      //   1. link pairs (per chain lane, LDS only): bounding spheres of the two links -> list W of the entries in reach. A cross-chain
      //      pair is listed by the lane of its FIRST link only; the partner takes the results over at the end (mirror slots).
      //   2. body pairs of W's entries: one bounding capsule per body -> list S of the body pairs within the largest margin.
      //   3. geom pairs of S's body pairs: bounding capsules of the two geoms (closest points of the two segments). Within the margin:
      //      a closed-form contact (kind 0) goes to the chain's result list R, a pair without a collider (kind 1) is counted, a convex
      //      pair (kind 2) goes to the chain's queue.
      //   4. the queue: a one-direction separation test per pair, survivors compacted; then MPR (the engine's general convex collider =
      //      libccd's Minkowski Portal Refinement, oracle/oracle.c: mpr_penetration is the float64 restatement it follows step by step).
      //      In both a round costs what its slowest lane costs, hence cheap and dear work in separate rounds. Contacts -> R.
      //   5. every replica records its chain's results as slots, then the mirrors of the other chains' results with a link of its own.
      // The lanes agree on list positions through ballots of their flags (lists are filled in work order, chain by chain). W, S, the
      // queue and R sit in the part of lane memory that holds M, the twists and the link images later in the pass. Lists that fill up
      // (W, S) make the pass run in chunks of entries; a full queue or result list drops contacts, counted in `overflow`.
      constexpr int kQueue = LMm::kQCap;                              // convex pairs per chain and pass
      constexpr int kWcap = (MC >= 5) ? 24 : 12;                      // link-pair entries per chain and chunk
      constexpr int kScap = 24;                                       // body pairs in reach per chain and chunk (an entry has at most 24)
      constexpr int kRcap = LMm::kRCap;                               // contacts per chain and pass (a chain has NS slots)
      constexpr int kW1 = LMm::kMcc, kS1 = kW1 + kWcap;
      constexpr int kQItem = LMm::kBig ? LMm::kLists : kS1 + 2 * kScap, kRes = kQItem + kQueue;
      static_assert(!DETECT || kS1 + 2 * kScap + (LMm::kBig ? 0 : kQueue + 8 * kRcap) <= LMm::kFrame, "the work lists of the pair pass must fit the dead part of lane memory");
      constexpr int kW = 4 * Q::kRep;                                 // lanes of one environment
      const int me = Q::rep() * 4 + c;
      int base[5];
      auto set_bases = [&](const int* n) { base[0] = 0; for (int cs = 0; cs < 4; cs++) base[cs + 1] = base[cs] + n[cs]; };
      auto chain_of = [&](int g) -> int { return (g >= base[1] ? 1 : 0) + (g >= base[2] ? 1 : 0) + (g >= base[3] ? 1 : 0); };
      // bits [lo, hi) of a ballot: the lanes that work on chain cs's units in this round
      using mask_t = typename Q::mask_t;                              // one bit per lane of the environment (kW <= 16: 32 bits; the whole wave: 64)
      constexpr int kMaskBits = 8 * (int)sizeof(mask_t);
      auto mbelow = [&](int i) -> mask_t { return (i >= kMaskBits) ? ~(mask_t)0 : (((mask_t)1 << i) - (mask_t)1); };      // bits [0, i)
      auto mcount = [&](mask_t m) -> int { return (sizeof(mask_t) == 8) ? __builtin_popcountll((unsigned long long)m) : __builtin_popcount((unsigned)m); };
      auto chain_bits = [&](int cs, int round) -> mask_t {
        int lo = base[cs] - round * kW, hi = base[cs + 1] - round * kW;
        lo = (lo < 0) ? 0 : lo; hi = (hi > kW) ? kW : hi;
        return (hi > lo) ? (mbelow(hi) & ~mbelow(lo)) : (mask_t)0;
      };
      auto share4 = [&](int x, int* out) {
#pragma unroll
        for (int cs = 0; cs < 4; cs++) out[cs] = (int)Q::quad_read((float)x, cs);
      };
      int nres_of[4] = {0, 0, 0, 0}, nq_of[4] = {0, 0, 0, 0};
      float gap_of[4] = {3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};        // smallest clearance seen for a pair of chain cs (this lane's share)
      auto gap_note = [&](int cs, float g) {
#pragma unroll
        for (int k = 0; k < 4; k++) if (k == cs) gap_of[k] = fminf(gap_of[k], g);
      };
      auto off_lpair_of = [&](int cs) -> int { return (cs == c) ? off_lpair_c : (int)cm[oz + LM_CM_CHAINS + LM_C_OFF_LPAIR * LM_NCHAIN + cs]; };
      // entry i of chain cs's link-pair list, seen from lane c: `dlo` / `dl` = where the lane memory of the entry's own / partner chain
      // sits relative to mine (the work queue hands a pair to ANY lane of the environment; the collection loop uses cs = c, dlo = 0)
      struct EntryCtx { int ka, kb, lb, own_q, dl, dlo; bool same_lane; V3 po, pp; M3 Ro, Rp; Sp Vo, Vp; };
      auto entry_ctx_of = [&](int cs, int i, EntryCtx& E) {
        const int off = (cs == c) ? off_lpair_c : (int)cm[oz + LM_CM_CHAINS + LM_C_OFF_LPAIR * LM_NCHAIN + cs];
        const int code = (int)LPE(off, i, 0);
        E.ka = code & 7; E.kb = (code >> 3) & 7; E.lb = (code >> 6) & 3; E.own_q = (code >> 8) & 1; E.dl = E.lb - c; E.dlo = cs - c;
        E.same_lane = E.kb != 7 && E.lb == cs;            // two links of one chain: the entry (and its slots) live in that lane only
      };
      auto entry_ctx = [&](int i, EntryCtx& E) { entry_ctx_of(c, i, E); };
      auto entry_frames = [&](EntryCtx& E) {               // own / partner link frame and velocity
        E.pp = O; E.Rp = R; E.Vp = Vroot;
        {
          const int fb = LMm::kFrame + E.ka * 18, dlo = E.dlo;
          E.po = v3(PEER(dlo, fb), PEER(dlo, fb + 1), PEER(dlo, fb + 2));
#pragma unroll
          for (int j = 0; j < 9; j++) E.Ro.a[j] = PEER(dlo, fb + 3 + j);
          E.Vo.w = v3(PEER(dlo, fb + 12), PEER(dlo, fb + 13), PEER(dlo, fb + 14)); E.Vo.v = v3(PEER(dlo, fb + 15), PEER(dlo, fb + 16), PEER(dlo, fb + 17));
        }
        if (E.kb != 7) {
          const int fb = LMm::kFrame + E.kb * 18, dl = E.dl;
          E.pp = v3(PEER(dl, fb), PEER(dl, fb + 1), PEER(dl, fb + 2));
#pragma unroll
          for (int j = 0; j < 9; j++) E.Rp.a[j] = PEER(dl, fb + 3 + j);
          E.Vp.w = v3(PEER(dl, fb + 12), PEER(dl, fb + 13), PEER(dl, fb + 14)); E.Vp.v = v3(PEER(dl, fb + 15), PEER(dl, fb + 16), PEER(dl, fb + 17));
        }
      };
      // one self-contact -> slot record (every replica writes the same words)
      auto emit_pair_slot = [&](const EntryCtx& E, const float* rec, bool g1own, V3 nrm, V3 cp, float dist) {
        const float pmargin = rec[LM_GP_MARGIN];
        if (nslot >= NS) { n_over++; return; }
        V3 t1, t2;
        make_frame(nrm, t1, t2);
        const int slot = nslot++;
        SL(slot, SL_LINK) = (float)E.ka; SL(slot, SL_GRF) = -1.0f;
        SL(slot, SL_PART) = (g1own ? -1.0f : 1.0f) * (float)(1 + ((E.kb == 7) ? 0 : E.lb) * 8 + E.kb);
        SL(slot, SL_NX) = nrm.x; SL(slot, SL_NY) = nrm.y; SL(slot, SL_NZ) = nrm.z;
        SL(slot, SL_RX) = cp.x; SL(slot, SL_RY) = cp.y; SL(slot, SL_RZ) = cp.z;
        if (E.kb != 7 && !E.same_lane) pair_mask |= 1 << E.lb;
        // relative velocity of body 2 against body 1 at the contact point, in the contact frame
        Sp Vrel = g1own ? (E.Vp + (-1.0f) * E.Vo) : (E.Vo + (-1.0f) * E.Vp);
        float vel[6];
        frame_rows(Vrel, cp, nrm, t1, t2, vel);
        const float imp = impedance(rec + LM_GP_S0, 1, dist, pmargin);
        const float D0 = imp / fmaxf(kMinVal, (1.0f - imp) * rec[LM_GP_TRAN]);
        const float Bp = rec[LM_GP_B], Kr = rec[LM_GP_K] * imp * (dist - pmargin);
        const int dim = (int)rec[LM_GP_DIM];
        const float mu = rec[LM_GP_MU];
        SL(slot, SL_DIM) = (float)dim; SL(slot, SL_MU) = mu; SL(slot, SL_D) = D0;
        if (PYR3(dim)) {
          // pyramid edges in the slot's own frame (a frictionless condim-1 pair arrives as mu = 0: four coinciding edges)
          float xv[4];
          pyr_rows(vel, mu, xv);
#pragma unroll
          for (int r = 0; r < 4; r++) SL(slot, SL_AREF + r) = -Bp * xv[r] - Kr;
        } else {
#pragma unroll
          for (int j2 = 1; j2 < 6; j2++) { SL(slot, SL_D + j2) = (j2 < dim) ? D0 / rec[LM_GP_RR1 + j2 - 1] : 0.0f; SL(slot, SL_FR + j2 - 1) = rec[LM_GP_F0 + j2 - 1]; }
#pragma unroll
          for (int j2 = 0; j2 < 6; j2++) SL(slot, SL_AREF + j2) = -Bp * vel[j2] - ((j2 == 0) ? Kr : 0.0f);
        }
        if (Q::rep() == 0 && (g1own || E.kb == 7 || E.same_lane)) {
          cnt.selfcon++;
          if ((int)rec[LM_GP_KIND] == 1 && (int)rec[LM_GP_X2 + LM_GX_TYPE] == LM_GEOM_BOX &&
              ((int)rec[LM_GP_X1 + LM_GX_TYPE] == LM_GEOM_BOX || (int)rec[LM_GP_X1 + LM_GX_TYPE] == LM_GEOM_CAPSULE)) cnt.natown++;
        }
      };
      // ---- 4. the convex pairs of the queues
      auto flush_queue = [&]() {
        Q::fence(); Q::quad_sync();
        int n_of[4];
#pragma unroll
        for (int cs = 0; cs < 4; cs++) n_of[cs] = (nq_of[cs] < kQueue) ? nq_of[cs] : kQueue;
        set_bases(n_of);
#pragma nounroll
        for (int stage = 0; stage < 2; stage++) {
          int kept[4] = {0, 0, 0, 0};
          const int T = base[4], nrounds = (T + kW - 1) / kW;
#pragma nounroll
          for (int round = 0; round < nrounds; round++) {
#ifdef LM_TIMERS
            cnt.m[7]++;
#endif
            const int g = round * kW + me;
            int cs = 0, t = 0, kind = 2, ncon_item = 1;
            float raw = 0.0f;
            EntryCtx E;
            const float* rec = gptp;
            bool g1own = false;
            if (g < T) {
              cs = chain_of(g); t = g - base[cs];
              raw = PEER(cs - c, kQItem + t);
              const int item = (int)raw;
              entry_ctx_of(cs, item >> 16, E);
              entry_frames(E);
              rec = gptp + (item & 65535) * LM_GPAIR_SIZE;
              g1own = ((int)rec[LM_GP_G1Q] == E.own_q);
              kind = (int)rec[LM_GP_KIND];
            }
            // a native pair (kind 1) can have several contacts: one per pass of this loop, the collider run again for each
            // (wave-uniform trip count; convex pairs and the separation stage take one pass)
#pragma nounroll
            for (int sub = 0;; sub++) {
              MprOut mo; mo.found = 0; mo.nx = mo.ny = mo.nz = mo.px = mo.py = mo.pz = mo.dist = 0.0f;
              if (g < T) {
                if (kind == 1) {
                  if (stage == 0) mo.found = (sub == 0) ? 1 : 0;          // no separation stage: the reach test was the tight bound already
                  else {
                    const NatGeom G1 = nat_geom(rec, false, g1own ? E.po : E.pp, g1own ? E.Ro : E.Rp, O);
                    const NatGeom G2 = nat_geom(rec, true, g1own ? E.pp : E.po, g1own ? E.Rp : E.Ro, O);
#if defined(LM_TIMERS) && defined(LM_MPR_CLOCK)
                    const long long tn0_ = LM_CLOCK();
#endif
                    const NatOut no = native_contact<(MC >= 5), kCapBox>(G1, G2, rec[LM_GP_MARGIN], sub, ncon_item);
#if defined(LM_TIMERS) && defined(LM_MPR_CLOCK)
                    cnt.m[12] += LM_CLOCK() - tn0_; cnt.m[15] += 1;          // (probe: cycles / runs of the native colliders)
#endif
                    mo.found = no.found ? 1 : 0; mo.dist = no.dist;
                    mo.nx = no.n.x; mo.ny = no.n.y; mo.nz = no.n.z; mo.px = no.p.x; mo.py = no.p.y; mo.pz = no.p.z;
                  }
                } else if (sub == 0) {
#ifndef LM_NO_MPR
                  if constexpr (!NOMPR) {
#ifdef LM_TIMERS
                    mo = mpr_convex_pair<(MC >= 5)>(P.meshadj, rec, g1own, E.po, E.Ro, E.pp, E.Rp, O, rec[LM_GP_MARGIN], stage, mprc ? mprc + (long long)(((int)raw) & 65535) * kMprCacheFloats : nullptr, cnt.m);
#else
                    mo = mpr_convex_pair<(MC >= 5)>(P.meshadj, rec, g1own, E.po, E.Ro, E.pp, E.Rp, O, rec[LM_GP_MARGIN], stage, mprc ? mprc + (long long)(((int)raw) & 65535) * kMprCacheFloats : nullptr);
#endif
                  }
#endif
                }
              }
              const bool found = mo.found != 0;
              const mask_t fm = Q::env_ballot(found);          // (also the point between this round's reads of the queue and its writes)
              if (found) {
                const int k = ((stage == 0) ? kept[cs] : nres_of[cs] + kept[cs]) + mcount(fm & chain_bits(cs, round) & mbelow(me));
                if (stage == 0) Q::peer_write(lmem, ls, kQItem + k, cs - c, raw);       // survivor: compacted in place (k <= t)
                else if (DETECT_ONLY) cnt.need_full = 1;      // a contact: the replay kernel's business
                else if (k < kRcap) {
                  const int rb_ = kRes + 8 * k, dlw = cs - c;
                  Q::peer_write(lmem, ls, rb_, dlw, raw); Q::peer_write(lmem, ls, rb_ + 1, dlw, mo.dist);
                  Q::peer_write(lmem, ls, rb_ + 2, dlw, mo.nx); Q::peer_write(lmem, ls, rb_ + 3, dlw, mo.ny); Q::peer_write(lmem, ls, rb_ + 4, dlw, mo.nz);
                  Q::peer_write(lmem, ls, rb_ + 5, dlw, mo.px); Q::peer_write(lmem, ls, rb_ + 6, dlw, mo.py); Q::peer_write(lmem, ls, rb_ + 7, dlw, mo.pz);
                }
              }
#pragma unroll
              for (int c2 = 0; c2 < 4; c2++) kept[c2] += mcount(fm & chain_bits(c2, round));
              Q::fence(); Q::quad_sync();
              if (!Q::any(stage == 1 && g < T && kind == 1 && sub + 1 < ncon_item)) break;
            }
          }
          if (stage == 0) { for (int c2 = 0; c2 < 4; c2++) n_of[c2] = kept[c2]; set_bases(n_of); }
          else for (int c2 = 0; c2 < 4; c2++) nres_of[c2] += kept[c2];
        }
      };
      // ---- 5. results -> slots: my chain's, then the mirrors of the others' cross-chain results with a link of mine
      auto emit_results = [&]() {
        Q::fence(); Q::quad_sync();
        // ADMISSION is decided once for the environment, the same way in every lane: a contact between two chains takes a slot in BOTH
        // lanes or in neither. (Each lane checking only its own room recorded half a contact when the partner was full — a force on one
        // body without its reaction: momentum out of nothing. Found with tools/probes/r3/find_nonfinite.py: tangled quadrupeds with
        // more self-contacts than slots ended non-finite, 2 in 12 M env-steps under the random policy, where the fp64 oracle stays sane.)
        int free_of[4];
        share4(NS - nslot, free_of);
        auto room = [&](int k_) -> int { return (k_ == 0) ? free_of[0] : ((k_ == 1) ? free_of[1] : ((k_ == 2) ? free_of[2] : free_of[3])); };
        auto take = [&](int k_) {
#pragma unroll
          for (int j = 0; j < 4; j++) if (j == k_) free_of[j]--;
        };
#pragma nounroll
        for (int cs = 0; cs < 4; cs++) {
          const int nres = (nres_of[cs] < kRcap) ? nres_of[cs] : kRcap, dls = cs - c;
#pragma nounroll
          for (int k = 0; k < nres; k++) {
            const int item = (int)PEER(dls, kRes + 8 * k);
            EntryCtx E;
            entry_ctx_of(cs, item >> 16, E);
            const int partner = (E.kb != 7 && !E.same_lane) ? E.lb : -1;
            if (room(cs) <= 0 || (partner >= 0 && room(partner) <= 0)) {       // no room on one side: the whole contact is dropped, counted once
              if (cs == c) n_over++;
              continue;
            }
            take(cs);
            if (partner >= 0) take(partner);
            if (cs != c) {
              if (partner != c) continue;                                        // not a pair with a link of mine
              const int ka_o = E.ka;
              E.ka = E.kb; E.kb = ka_o; E.lb = cs; E.own_q = 1 - E.own_q; E.dl = dls; E.dlo = 0; E.same_lane = false;       // my view of it
            }
            entry_frames(E);
            const float* rec = gptp + (item & 65535) * LM_GPAIR_SIZE;
            const float dist = PEER(dls, kRes + 8 * k + 1);
            const V3 nrm = v3(PEER(dls, kRes + 8 * k + 2), PEER(dls, kRes + 8 * k + 3), PEER(dls, kRes + 8 * k + 4));
            const V3 cp = v3(PEER(dls, kRes + 8 * k + 5), PEER(dls, kRes + 8 * k + 6), PEER(dls, kRes + 8 * k + 7));
#ifdef LM_PAIR_TRACE
            if (Q::rep() == 0) printf("   contact lane %d (list of lane %d) entry %d rec %d kind %d dist %.8f nrm %.6f %.6f %.6f pos %.6f %.6f %.6f\n", c, cs, item >> 16, item & 65535, (int)rec[LM_GP_KIND], dist, nrm.x, nrm.y, nrm.z, cp.x + O.x, cp.y + O.y, cp.z + O.z);
#endif
            emit_pair_slot(E, rec, ((int)rec[LM_GP_G1Q] == E.own_q), nrm, cp, dist);
          }
        }
        Q::fence(); Q::quad_sync();
      };
      // geometry of a geom pair: closest points of the two capsule segments
      struct PairGeom { V3 c1, c2, q1, dq; float r1, r2, dd, dist; bool g1own; };
      auto pair_geom = [&](const EntryCtx& E, const float* rec) -> PairGeom {
        PairGeom G;
        G.g1own = ((int)rec[LM_GP_G1Q] == E.own_q);        // geom 1 sits on the entry's own link
        const V3 p1 = G.g1own ? E.po : E.pp, p2 = G.g1own ? E.pp : E.po;
        const M3& R1 = G.g1own ? E.Ro : E.Rp; const M3& R2 = G.g1own ? E.Rp : E.Ro;
        G.c1 = p1 + mul(R1, v3(rec[LM_GP_P1], rec[LM_GP_P1 + 1], rec[LM_GP_P1 + 2]));
        const V3 a1 = mul(R1, v3(rec[LM_GP_A1], rec[LM_GP_A1 + 1], rec[LM_GP_A1 + 2]));
        G.c2 = p2 + mul(R2, v3(rec[LM_GP_P2], rec[LM_GP_P2 + 1], rec[LM_GP_P2 + 2]));
        const V3 a2 = mul(R2, v3(rec[LM_GP_A2], rec[LM_GP_A2 + 1], rec[LM_GP_A2 + 2]));
        G.r1 = rec[LM_GP_R1]; G.r2 = rec[LM_GP_R2];
        float sa, ta;
        segment_closest(G.c1, a1, rec[LM_GP_H1], G.c2, a2, rec[LM_GP_H2], sa, ta);
        G.q1 = G.c1 + sa * a1;
        G.dq = G.c2 + ta * a2 - G.q1;
        G.dd = sqrtf(dot(G.dq, G.dq)); G.dist = G.dd - G.r1 - G.r2;
        return G;
      };
#if defined(LM_TIMERS) && defined(LM_PAIR_PHASES)
      long long ph_ = LM_CLOCK();
#define LM_PHASE(i) do { long long n_ = LM_CLOCK(); cnt.m[i] += n_ - ph_; ph_ = n_; } while (0)
#else
#define LM_PHASE(i) do {} while (0)
#endif
      // ---- 1a. link pairs of my chain, DEALT TO THE REPLICAS (an entry costs two dependent LDS round trips: its code, then the two
      // sphere centres; measured 32 k cycles per detection for the quadruped with every lane walking all entries): replica r
      // tests entries r, r + kRep, ... and keeps bit i / kRep of its mask for an entry in reach; the masks are exchanged
      // (exact as floats: <= 64 / kRep bits... 16 with four replicas), every replica then builds the same lists from them.
      // (six-link chains: up to 128 entries per lane, a second mask word)
      constexpr bool kWide = MC == 6;
      constexpr int kEB = kWide ? 128 : 64;                          // an entry of W = entry index + kEB * its body pairs
      unsigned long long reach_r = 0ull, reach_r2 = 0ull;
      // world centre of a link's bounding sphere: kept in lane memory by the kernels with the full pair pass, from the link frame
      // otherwise (detection only: no room for it beside the six-link lane memory)
      auto bs_centre = [&](int dl, int lane, int k) -> V3 {
        if constexpr (PAIRS && MC != 6) return v3(PEER(dl, LMm::kBS + k * 3), PEER(dl, LMm::kBS + k * 3 + 1), PEER(dl, LMm::kBS + k * 3 + 2));
        else {
          const int fb = LMm::kFrame + k * 18;
          M3 Rl;
#pragma unroll
          for (int j = 0; j < 9; j++) Rl.a[j] = PEER(dl, fb + 3 + j);
          const int lf = LM_CM_CHAINS + (LM_C_LINKS + k * LM_LINK_SIZE + LM_D_SIZE + LM_L_BSX) * LM_NCHAIN + lane;
          return v3(PEER(dl, fb), PEER(dl, fb + 1), PEER(dl, fb + 2)) + mul(Rl, v3(cm[oz + lf], cm[oz + lf + LM_NCHAIN], cm[oz + lf + 2 * LM_NCHAIN]));
        }
      };
      {
        // branch-free body, unrolled: the LDS reads of several entries are in flight together
        float gmy = 3.0e38f, gpart[4] = {3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};
#pragma unroll 4
        for (int i = Q::rep(); i < nlp; i += Q::kRep) {
          const int code = (int)LPE(off_lpair_c, i, 0);
          const float thr2 = LPE(off_lpair_c, i, 2);
          const int ka = code & 7, kb = (code >> 3) & 7, lb = (code >> 6) & 3, own_q = (code >> 8) & 1, dl = lb - c;
          const int kbs = (kb == 7) ? 0 : kb, dls = (kb == 7) ? 0 : dl;
          const V3 ca = bs_centre(0, c, ka);
          const V3 cbp = bs_centre(dls, (kb == 7) ? c : lb, kbs);
          const V3 cb = (kb == 7) ? rootc : cbp;
          const V3 dc = cb - ca;
          const float d2c = dot(dc, dc);
          const bool mine = true; (void)own_q;                  // (every link pair is listed ONCE, in one of its two lanes: lowering._self_collision_tables)
          const bool inr = d2c < thr2;
          // (the list's reach is LM_PAIR_PAD beyond touching: a pruned link pair is at least that far from any contact)
          const float gp_ = sqrtf(d2c) - sqrtf(thr2) + LM_PAIR_PAD;
          const bool note = mine && !inr, notep = note && kb != 7 && lb != c;
          gmy = note ? fminf(gmy, gp_) : gmy;
#pragma unroll
          for (int k = 0; k < 4; k++) gpart[k] = (notep && k == lb) ? fminf(gpart[k], gp_) : gpart[k];
          const int bit_ = i / Q::kRep;
          if (kWide && bit_ >= 64) reach_r2 |= (mine && inr) ? (1ull << (bit_ - 64)) : 0ull;
          else reach_r |= (mine && inr) ? (1ull << bit_) : 0ull;
        }
        gap_note(c, gmy);
#pragma unroll
        for (int k = 0; k < 4; k++) gap_of[k] = fminf(gap_of[k], gpart[k]);
      }
      // every replica's mask -> ONE mask of my chain's entries in reach, bit = entry (exact as floats: 16 bits with four replicas)
      unsigned long long reach_all = reach_r, reach_all2 = reach_r2;
      if (Q::kRep > 1) {
        reach_all = 0ull; reach_all2 = 0ull;
#pragma unroll
        for (int r = 0; r < Q::kRep; r++) {
          unsigned m_ = (unsigned)Q::rep_bcast((float)((unsigned)reach_r & 0xffffu), r);
          if (kWide) m_ |= (unsigned)Q::rep_bcast((float)(((unsigned)reach_r >> 16) & 0xffffu), r) << 16;      // entries 64 .. 127: bits 16 .. 31 of a replica's mask
#pragma nounroll
          while (m_) {
            const int b_ = __builtin_ctz(m_), e_ = Q::kRep * b_ + r;
            if (kWide && e_ >= 64) reach_all2 |= 1ull << (e_ - 64); else reach_all |= 1ull << e_;
            m_ &= m_ - 1u;
          }
        }
      }
      LM_PHASE(15);
      int n_prox = 0;
#pragma nounroll
      for (;;) {
        // ---- 1b. the next chunk of my chain's entries in reach (every replica builds the same list)
        int nw = 0, nunits = 0;
#pragma nounroll
        while ((reach_all != 0ull || (kWide && reach_all2 != 0ull)) && nw < kWcap) {
          const bool lo_ = reach_all != 0ull;
          const int i = lo_ ? __builtin_ctzll(reach_all) : 64 + __builtin_ctzll(reach_all2);
          const int nbp = (int)LPE(off_lpair_c, i, 1) >> 16;
          if (nunits + nbp > kScap) break;                   // its body pairs would not fit S: next chunk (an entry alone always fits)
          if (lo_) reach_all &= reach_all - 1ull; else reach_all2 &= reach_all2 - 1ull;
          LMEM(kW1 + nw) = (float)(i + kEB * nbp);
          nw++; nunits += nbp;
        }
        if (!(Q::sum((nw > 0) ? 1.0f : 0.0f) > 0.0f)) break;          // no chain of the environment has entries in reach left (quad-uniform)
        LM_PHASE(10);
        int nw_of[4], nu_of[4];
        share4(nw, nw_of); share4(nunits, nu_of);
#ifdef LM_PAIR_TRACE
        if (me == 0) printf("  chunk: entries %d %d %d %d body pairs %d %d %d %d (of %d entries)\n", nw_of[0], nw_of[1], nw_of[2], nw_of[3], nu_of[0], nu_of[1], nu_of[2], nu_of[3], nlp);
#endif
        Q::fence(); Q::quad_sync();
        // ---- 2. body pairs: one bounding capsule per body; those within the largest margin of their geom pairs go on, the others
        // hold the next detection back by their clearance
        int ns_of[4] = {0, 0, 0, 0};
        set_bases(nu_of);
        {
          const int T = base[4], nrounds = (T + kW - 1) / kW;
#pragma nounroll
          for (int round = 0; round < nrounds; round++) {
            const int g = round * kW + me;
            bool hitb = false;
            int cs = 0;
            float s0 = 0.0f, s1 = 0.0f;
            if (g < T) {
              cs = chain_of(g);
              int u = g - base[cs], i = 0, jb = 0;
#pragma nounroll
              for (int t = 0; t < nw_of[cs]; t++) {
                const int w = (int)PEER(cs - c, kW1 + t), nb = w / kEB;
                if (u < nb) { i = w & (kEB - 1); jb = u; break; }
                u -= nb;
              }
              EntryCtx E;
              entry_ctx_of(cs, i, E);
              entry_frames(E);
              const int bfirst = (int)LPE(off_lpair_of(cs), i, 1) & 65535;
              const float* br = P.bpt + (bfirst + jb) * LM_BP_SIZE;
              const float* bo = br + (E.own_q ? LM_BP_P2 : LM_BP_P1); const float* bq = br + (E.own_q ? LM_BP_P1 : LM_BP_P2);
              const V3 co = E.po + mul(E.Ro, v3(bo[0], bo[1], bo[2])), ao = mul(E.Ro, v3(bo[3], bo[4], bo[5]));
              const V3 cq = E.pp + mul(E.Rp, v3(bq[0], bq[1], bq[2])), aq = mul(E.Rp, v3(bq[3], bq[4], bq[5]));
              float sa, ta;
              segment_closest(co, ao, bo[6], cq, aq, bq[6], sa, ta);
              const V3 dq = cq + ta * aq - (co + sa * ao);
              const float bgap = sqrtf(dot(dq, dq)) - bo[7] - bq[7] - br[LM_BP_MARGIN];
              if (bgap < 0.0f) { hitb = true; s0 = (float)(i * 32 + jb + 4096 * (int)br[LM_BP_N]); s1 = br[LM_BP_FIRST]; }
              else { gap_note(cs, bgap); if (E.kb != 7 && !E.same_lane) gap_note(E.lb, bgap); }
            }
            const mask_t sm = Q::env_ballot(hitb);
            if (hitb) {
              const int k = ns_of[cs] + mcount(sm & chain_bits(cs, round) & mbelow(me));
              Q::peer_write(lmem, ls, kS1 + 2 * k, cs - c, s0); Q::peer_write(lmem, ls, kS1 + 2 * k + 1, cs - c, s1);
            }
#pragma unroll
            for (int c2 = 0; c2 < 4; c2++) ns_of[c2] += mcount(sm & chain_bits(c2, round));
          }
        }
        Q::fence(); Q::quad_sync();
        LM_PHASE(11);
        // ---- 3. geom pairs of the body pairs in reach
        int nv = 0, nv_of[4];
#pragma nounroll
        for (int t = 0; t < ns_of[c]; t++) nv += (int)LMEM(kS1 + 2 * t) >> 12;
        share4(nv, nv_of);
#ifdef LM_PAIR_TRACE
        if (me == 0) printf("  body pairs in reach %d %d %d %d geom pairs %d %d %d %d\n", ns_of[0], ns_of[1], ns_of[2], ns_of[3], nv_of[0], nv_of[1], nv_of[2], nv_of[3]);
#endif
        set_bases(nv_of);
        {
          const int T = base[4], nrounds = (T + kW - 1) / kW;
#pragma nounroll
          for (int round = 0; round < nrounds; round++) {
            const int g = round * kW + me;
            bool has_res = false, want_q = false, is_prox = false;
            int cs = 0;
            float code = 0.0f, rdist = 0.0f;
            V3 rn = v3(0, 0, 0), rp = v3(0, 0, 0);
            if (g < T) {
              cs = chain_of(g);
              int v = g - base[cs], i = 0, first = 0;
#pragma nounroll
              for (int t = 0; t < ns_of[cs]; t++) {
                const int w = (int)PEER(cs - c, kS1 + 2 * t), np_ = w >> 12;
                if (v < np_) { i = (w & 4095) >> 5; first = (int)PEER(cs - c, kS1 + 2 * t + 1); break; }
                v -= np_;
              }
              const float* rec = gptp + (first + v) * LM_GPAIR_SIZE;
              const int kind = (int)rec[LM_GP_KIND];
              {
                EntryCtx E;
                entry_ctx_of(cs, i, E);
                entry_frames(E);
                const PairGeom G = pair_geom(E, rec);
                const float pmargin = rec[LM_GP_MARGIN];
                const bool is_cross = E.kb != 7 && !E.same_lane;
#ifdef LM_PAIR_TRACE
                if (G.dist < 0.004f) printf("   pair lane %d entry %d rec %d kind %d dist %.7f\n", cs, i, first + v, kind, G.dist);
#endif
#if defined(LM_TIMERS) && !defined(LM_PAIR_PHASES)
#ifndef LM_MPR_CLOCK
                cnt.m[10 + kind]++; if (G.dist < pmargin) cnt.m[13 + kind]++;
#endif
#endif
                float clearance = G.dist - pmargin;               // what holds the next detection back: a LOWER bound of the pair's distance
                bool in_reach = G.dist < pmargin;
                if (in_reach) {
                  // the engine's mid phase: bounding spheres of the two geoms WITHOUT the margin (pinned for plane pairs by the golden
                  // rollouts, restated the same way for geom pairs by the oracle): two foot spheres 0 < dist < margin apart make no contact
                  const V3 cc = G.c2 - G.c1;
                  const float rb1 = (rec[LM_GP_X1 + LM_GX_RBOUND] > 0.0f) ? rec[LM_GP_X1 + LM_GX_RBOUND] : rec[LM_GP_H1] + G.r1;
                  const float rb2 = (rec[LM_GP_X2 + LM_GX_RBOUND] > 0.0f) ? rec[LM_GP_X2 + LM_GX_RBOUND] : rec[LM_GP_H2] + G.r2;
                  if (sqrtf(dot(cc, cc)) - rb1 - rb2 > 0.0f) in_reach = false;
                }
                if (in_reach && kind == 1) {
                  // a native pair (box / cylinder against a sphere, capsule or box): the bounding capsule of a box is loose — the
                  // tight bound decides (and is what the slack remembers: a trunk box centimetres from a thigh does not call for
                  // a detection in every pass)
                  const NatGeom G1 = nat_geom(rec, false, G.g1own ? E.po : E.pp, G.g1own ? E.Ro : E.Rp, O);
                  const NatGeom G2 = nat_geom(rec, true, G.g1own ? E.pp : E.po, G.g1own ? E.Rp : E.Ro, O);
                  const float lb = native_lower_bound<(MC >= 5)>(G1, G2);
                  clearance = fmaxf(clearance, lb - pmargin);
                  in_reach = lb < pmargin;
                }
                gap_note(cs, clearance);
                if (is_cross) gap_note(E.lb, clearance);
                if constexpr (DETECT_ONLY) {
                  // this kernel has no contact machinery for geom pairs: a CONTACT hands the control step to the replay kernel
                  // (lm_step.h). Hull pairs of neighbouring links sit inside each other's bounding capsules for good: they go
                  // through the colliders like everywhere else, and only what those find counts (flush_queue)
                  if (in_reach) {
                    if (kind == 3) is_prox = true;
                    else if (kind == 0) cnt.need_full = 1;
                    else { want_q = true; code = (float)(i * 65536 + first + v); }
                  }
                } else
                if (in_reach && kind == 3) is_prox = true;              // a pair without a collider (a mesh that came without a hull): counted
                else if (!kCapBox && in_reach && kind == 1 && (int)rec[LM_GP_X1 + LM_GX_TYPE] == LM_GEOM_CAPSULE) cnt.need_full = 1;     // capsule against a box: the replay kernel's
                else if (in_reach && kind != 0) {
                  // convex pairs (the engine's MPR) and native pairs: queued, their colliders run side by side in flush_queue
                  if constexpr (NOMPR) { if (kind == 2) cnt.need_full = 1; else { want_q = true; code = (float)(i * 65536 + first + v); } }
                  else { want_q = true; code = (float)(i * 65536 + first + v); }
                } else if (in_reach) {
                  has_res = true; code = (float)(i * 65536 + first + v);
                  rdist = G.dist;
                  rn = (G.dd < 1e-15f) ? v3(1, 0, 0) : (1.0f / G.dd) * G.dq;
                  rp = G.q1 + (G.r1 + 0.5f * rdist) * rn - O;
                }
              }
            }
            const mask_t rm = Q::env_ballot(has_res), qm = Q::env_ballot(want_q), pm = Q::env_ballot(is_prox);
            const mask_t below = chain_bits(cs, round) & mbelow(me);
            if (has_res) {
              const int k = nres_of[cs] + mcount(rm & below);
              if (k < kRcap) {
                const int rb_ = kRes + 8 * k, dlw = cs - c;
                Q::peer_write(lmem, ls, rb_, dlw, code); Q::peer_write(lmem, ls, rb_ + 1, dlw, rdist);
                Q::peer_write(lmem, ls, rb_ + 2, dlw, rn.x); Q::peer_write(lmem, ls, rb_ + 3, dlw, rn.y); Q::peer_write(lmem, ls, rb_ + 4, dlw, rn.z);
                Q::peer_write(lmem, ls, rb_ + 5, dlw, rp.x); Q::peer_write(lmem, ls, rb_ + 6, dlw, rp.y); Q::peer_write(lmem, ls, rb_ + 7, dlw, rp.z);
              }
            }
            if (want_q) {
              const int k = nq_of[cs] + mcount(qm & below);
              if (k < kQueue) Q::peer_write(lmem, ls, kQItem + k, cs - c, code);
            }
#pragma unroll
            for (int c2 = 0; c2 < 4; c2++) { const mask_t cb_ = chain_bits(c2, round); nres_of[c2] += mcount(rm & cb_); nq_of[c2] += mcount(qm & cb_); }
            n_prox += mcount(pm);
          }
        }
        Q::fence(); Q::quad_sync();
      }
      LM_PHASE(12);
      if (c == 0 && Q::rep() == 0) cnt.selfprox += n_prox;
#ifdef LM_PAIR_TRACE
      if (me == 0) printf("  queued %d %d %d %d results %d %d %d %d gaps %.5f %.5f %.5f %.5f\n", nq_of[0], nq_of[1], nq_of[2], nq_of[3], nres_of[0], nres_of[1], nres_of[2], nres_of[3], gap_of[0], gap_of[1], gap_of[2], gap_of[3]);
#endif
      LM_TICK(14);              // self-collisions: broad / mid / narrow-phase tests
      flush_queue();            // every lane of the wave arrives here together: the queued pairs of all of them run side by side
      LM_TICK(15);              // self-collisions: convex pairs (MPR)
      LM_PHASE(13);
      if constexpr (PAIRS) emit_results();
      // contacts beyond what the queue / the result list of my chain hold: dropped, counted
      if (nq_of[c] > kQueue) n_over += nq_of[c] - kQueue;
      if (nres_of[c] > kRcap) n_over += nres_of[c] - kRcap;
      cnt.peak_q = (nq_of[c] > cnt.peak_q) ? nq_of[c] : cnt.peak_q; cnt.peak_res = (nres_of[c] > cnt.peak_res) ? nres_of[c] : cnt.peak_res;
      // the smallest clearance of any pair of my chain, whichever lane of the environment looked at it
      {
        float gmine = 3.0e38f;
#pragma unroll
        for (int cs = 0; cs < 4; cs++) {
          float gq = gap_of[cs];
          gq = fminf(fminf(Q::quad_read(gq, 0), Q::quad_read(gq, 1)), fminf(Q::quad_read(gq, 2), Q::quad_read(gq, 3)));
          if (Q::kRep > 1) {
            float gm_ = gq;
#pragma unroll
            for (int r = 0; r < Q::kRep; r++) gm_ = fminf(gm_, Q::rep_bcast(gq, r));
            gq = gm_;
          }
          if (cs == c) gmine = gq;
        }
        gap_min = gmine;
      }
      LM_PHASE(14);
      if (Q::rep() == 0) cnt.overflow += n_over;
      if (pair_slack) *pair_slack = gap_min;
      Q::fence();
    }
    cnt.ncon += nslot;
    cnt.peak_slots = (nslot > cnt.peak_slots) ? nslot : cnt.peak_slots;
    LM_TICK(0);
    pair_mask_out = pair_mask;
    for (int s2 = 0; s2 < nslot; s2++) { if ((int)SL(s2, SL_LINK) < 0) nrootslot++; if (PAIRS && SL(s2, SL_PART) != 0.0f) npairslot++; }

    // DEFER (the kernels with a pair pass): twists, velocity / acceleration recursion and link inertias are built HERE, behind the
    // collision passes, from the link frames those passes read out of lane memory anyway — held in registers across the pair pass
    // (140 values per lane for five links) they were what the compiler spilled around the float64 collider (round 6:
    // tools/probes/r6/spill_flow.py — 146 of the kernel's 478 scratch dwords were stored in the kinematics loop and loaded here). A
    // hinge leaves its own axis and anchor where they were, so both follow from the frame BEHIND the joint: u = R_k axis,
    // a = p_k + R_k anchor.
    if constexpr (DEFER) {
      root_inertia();
      {
        Sp V = sp0();
#pragma unroll
        for (int k = 0; k < 6; k++) {
          Sr[k] = ldSr(k);
          Sp Sd; Sd.w = cross(V.w, Sr[k].w); Sd.v = cross(V.w, Sr[k].v) + cross(V.v, Sr[k].w);
          V = V + vr[k] * Sr[k];
          Aroot = Aroot + vr[k] * Sd;
        }
      }
      Sp V = Vroot, A = Aroot;
#pragma unroll
      for (int k = 0; k < MC; k++) {
        if (k < nl) {
          const int fb = LMm::kFrame + k * 18;
          const V3 pk = v3(LMEM(fb), LMEM(fb + 1), LMEM(fb + 2));
          M3 Rk;
#pragma unroll
          for (int i = 0; i < 9; i++) Rk.a[i] = LMEM(fb + 3 + i);
          const V3 uw = mul(Rk, v3(LK(k, LM_D_AX), LK(k, LM_D_AY), LK(k, LM_D_AZ)));
          if (LK(k, LM_D_TYPE) != 0.0f) {
            const V3 aw = pk + mul(Rk, v3(LK(k, LM_D_PX), LK(k, LM_D_PY), LK(k, LM_D_PZ)));
            Sc[k].w = uw; Sc[k].v = cross(uw, O - aw);
          } else { Sc[k].w = v3(0, 0, 0); Sc[k].v = uw; }
          Sp Sd; Sd.w = cross(V.w, Sc[k].w); Sd.v = cross(V.w, Sc[k].v) + cross(V.v, Sc[k].w);
          A = A + vc[k] * Sd;
          V.w = v3(LMEM(fb + 12), LMEM(fb + 13), LMEM(fb + 14)); V.v = v3(LMEM(fb + 15), LMEM(fb + 16), LMEM(fb + 17));      // = V + vc[k] S_k, as the kinematics loop left it
          Vc[k] = V; Ac[k] = A;
          float Il[6], Iw[6];
#pragma unroll
          for (int i = 0; i < 6; i++) Il[i] = LXI(k, 4 + i);
          rotate_inertia(Rk, Il, Iw);
          Ic[k] = make_spi(LXI(k, 0), pk + mul(Rk, v3(LXI(k, 1), LXI(k, 2), LXI(k, 3))) - O, Iw);
        } else { Sc[k] = sp0(); Vc[k] = V; Ac[k] = A; Ic[k] = spi0(); }
      }
    }
    // ======== inertia matrix (composite rigid body about O) and bias (spatial Newton-Euler) ========
    SpI comp = spi0();
    Sp Fsuf = sp0();
#pragma unroll
    for (int k = MC - 1; k >= 0; k--) {
      // body force of link k:  I A + V x* (I V)
      Sp mom = apply(Ic[k], Vc[k]);
      Sp F = apply(Ic[k], Ac[k]);
      F.w = F.w + cross(Vc[k].w, mom.w) + cross(Vc[k].v, mom.v);
      F.v = F.v + cross(Vc[k].w, mom.v);
      Fsuf = Fsuf + F;
      bias_c[k] = spdot(Sc[k], Fsuf);
      comp = comp + Ic[k];
      Sp L = apply(comp, Sc[k]);
#pragma unroll
      for (int j = 0; j <= k; j++) Mcc[tri(k, j)] = spdot(Sc[j], L);
#pragma unroll
      for (int r = 0; r < 6; r++) Mcr[k][r] = spdot(Sr[r], L);
      Mcc[tri(k, k)] += (k < nl) ? LKV(k, 0, LM_D_ARM) : 1.0f;
    }
    // whole-robot composite and force: root body + sum over the 4 chains
    SpI tot = Iroot;
    tot.m += Q::sum(comp.m);
    tot.h = tot.h + v3(Q::sum(comp.h.x), Q::sum(comp.h.y), Q::sum(comp.h.z));
    tot.xx += Q::sum(comp.xx); tot.yy += Q::sum(comp.yy); tot.zz += Q::sum(comp.zz);
    tot.xy += Q::sum(comp.xy); tot.xz += Q::sum(comp.xz); tot.yz += Q::sum(comp.yz);
    Sp mom = apply(Iroot, Vroot);
    Sp F = apply(Iroot, Aroot);
    F.w = F.w + cross(Vroot.w, mom.w) + cross(Vroot.v, mom.v);
    F.v = F.v + cross(Vroot.w, mom.v);
    F.w = F.w + v3(Q::sum(Fsuf.w.x), Q::sum(Fsuf.w.y), Q::sum(Fsuf.w.z));
    F.v = F.v + v3(Q::sum(Fsuf.v.x), Q::sum(Fsuf.v.y), Q::sum(Fsuf.v.z));
#pragma unroll
    for (int i = 0; i < 6; i++) {
      Sp L = apply(tot, Sr[i]);
#pragma unroll
      for (int j = 0; j <= i; j++) Mrr[tri(i, j)] = spdot(Sr[j], L);
      Mrr[tri(i, i)] += RDV(i, 0, LM_D_ARM);
      bias_r[i] = spdot(Sr[i], F);
    }
    LM_TICK(1);

    // ======== muscles: tendon length/velocity through the site path, force = gain(L,V)*act + bias(L), moment arms ========
    float musc[MC];
#pragma unroll
    for (int k = 0; k < MC; k++) musc[k] = 0.0f;
    if (NM > 0) {
      Sp Fl[MC];                                   // tendon forces as wrenches about O, per link
#pragma unroll
      for (int k = 0; k < MC; k++) Fl[k] = sp0();
      const int m0 = (int)mt[oz + c], nm = (int)mt[oz + LM_NCHAIN + c];
      // the replicas of a small-batch environment (Q::kRep quads) share the chain's muscles between them: muscle i
      // (its force and its activation state in the shared lane memory) belongs to replica i mod kRep
      for (int i = Q::rep(); i < nm; i += Q::kRep) {
        const float* rec = mt + oz + LM_MT_HEAD + (m0 + i) * LM_MU_SIZE;
        const float* site = mt + oz + LM_MT_SITES + 4 * (int)rec[LM_MU_SITE_ADR];
        const int nsite = (int)rec[LM_MU_SITE_NUM];
        Sp Ul[MC];                                 // wrench per unit tendon force
#pragma unroll
        for (int k = 0; k < MC; k++) Ul[k] = sp0();
        float len = 0.0f, vel = 0.0f;
        V3 pp = v3(0, 0, 0), vp = v3(0, 0, 0);
        int lp = -2;
        for (int s = 0; s < nsite; s++) {
          const int li = (int)site[4 * s];
          const V3 loc = v3(site[4 * s + 1], site[4 * s + 2], site[4 * s + 3]);
          V3 p, w, vl;
          if (li < 0) { p = O + mul(R, loc); w = Vroot.w; vl = Vroot.v; }
          else {
            const int fb = LMm::kFrame + li * 18;
            M3 Rl;
#pragma unroll
            for (int j = 0; j < 9; j++) Rl.a[j] = LMEM(fb + 3 + j);
            p = v3(LMEM(fb), LMEM(fb + 1), LMEM(fb + 2)) + mul(Rl, loc);
            w = v3(LMEM(fb + 12), LMEM(fb + 13), LMEM(fb + 14)); vl = v3(LMEM(fb + 15), LMEM(fb + 16), LMEM(fb + 17));
          }
          const V3 r = p - O;
          const V3 pv = vl + cross(w, r);          // velocity of the path point (twists are about O)
          if (s > 0) {
            const V3 d = p - pp;
            const float seg = sqrtf(dot(d, d));
            len += seg;
            if (li != lp && seg > 1e-12f) {        // a segment inside one body has no moment arm
              const V3 u = (1.0f / seg) * d;
              vel += dot(u, pv - vp);
              // +u at this point on link li, -u at the previous point on link lp (root points move no chain dof)
              Sp Wc; Wc.w = cross(r, u); Wc.v = u;
              Sp Wp; Wp.w = cross(pp - O, u); Wp.v = u;
#pragma unroll
              for (int k = 0; k < MC; k++) {
                if (li == k) Ul[k] = Ul[k] + Wc;
                if (lp == k) Ul[k] = Ul[k] + (-1.0f) * Wp;
              }
            }
          }
          pp = p; vp = pv; lp = li;
        }
        const float gear = rec[LM_MU_GEAR];
        const float L = fmaf(fmaf(gear, len, -rec[LM_MU_LR0]), rec[LM_MU_INV_L0], rec[LM_MU_RANGE0]);
        const float Vn = gear * vel * rec[LM_MU_INV_L0VMAX];
        const float lmin = rec[LM_MU_LMIN], lmax = rec[LM_MU_LMAX], fvmax = rec[LM_MU_FVMAX];
        // force-length curve with the engine's branch order (a muscle shorter than lmin lands in the second branch)
        const float a = 0.5f * (lmin + 1.0f), b = 0.5f * (1.0f + lmax);
        float FL, x;
        if (L >= lmin && L <= a) { x = (L - lmin) / fmaxf(kMinVal, a - lmin); FL = 0.5f * x * x; }
        else if (L <= 1.0f) { x = (1.0f - L) / fmaxf(kMinVal, 1.0f - a); FL = 1.0f - 0.5f * x * x; }
        else if (L <= b) { x = (L - 1.0f) / fmaxf(kMinVal, b - 1.0f); FL = 1.0f - 0.5f * x * x; }
        else if (L <= lmax) { x = (lmax - L) / fmaxf(kMinVal, lmax - b); FL = 0.5f * x * x; }
        else FL = 0.0f;
        const float y = fvmax - 1.0f;
        float FV;
        if (Vn <= -1.0f) FV = 0.0f;
        else if (Vn <= 0.0f) FV = (Vn + 1.0f) * (Vn + 1.0f);
        else if (Vn <= y) FV = fvmax - (y - Vn) * (y - Vn) / fmaxf(kMinVal, y);
        else FV = fvmax;
        float FP;
        if (L <= 1.0f) FP = 0.0f;
        else if (L <= b) { x = (L - 1.0f) / fmaxf(kMinVal, b - 1.0f); FP = rec[LM_MU_FPMAX] * 0.5f * x * x; }
        else { x = (L - b) / fmaxf(kMinVal, b - 1.0f); FP = rec[LM_MU_FPMAX] * (0.5f + x); }
        const float act = LMEM(LMm::kAct + i), ctrl = LMEM(LMm::kCtrl + i);
        const float force = -rec[LM_MU_FORCE] * fmaf(FL * FV, act, FP) * gear;
#pragma unroll
        for (int k = 0; k < MC; k++) Fl[k] = Fl[k] + force * Ul[k];
        if (EULER) {
          // activation dynamics: time constants depend on the activation, hard switch between rise and decay
          const float cc = fminf(fmaxf(ctrl, 0.0f), 1.0f), ac_ = fminf(fmaxf(act, 0.0f), 1.0f);
          const float dctrl = cc - act;
          const float tau = (dctrl > 0.0f) ? rec[LM_MU_TAU_ACT] * (0.5f + 1.5f * ac_) : rec[LM_MU_TAU_DEACT] / (0.5f + 1.5f * ac_);
          LMEM(LMm::kAct + i) = fmaf(P.h, dctrl / fmaxf(kMinVal, tau), act);
        }
      }
      Sp Fs = sp0();
#pragma unroll
      for (int k = MC - 1; k >= 0; k--) {
        if (Q::kRep > 1) {      // symmetric butterfly: every replica ends up with the bit-identical total
          Fl[k].w = v3(Q::rep_sum(Fl[k].w.x), Q::rep_sum(Fl[k].w.y), Q::rep_sum(Fl[k].w.z));
          Fl[k].v = v3(Q::rep_sum(Fl[k].v.x), Q::rep_sum(Fl[k].v.y), Q::rep_sum(Fl[k].v.z));
        }
        Fs = Fs + Fl[k]; musc[k] = spdot(Sc[k], Fs);
      }
    }

    // ======== smooth forces, unconstrained acceleration ========
#pragma unroll
    for (int i = 0; i < 6; i++) sm_r[i] = -STIFF_R(i) * qr[i] - DAMP_R(i) * vr[i] - bias_r[i] + actr[i];
#pragma unroll
    for (int k = 0; k < MC; k++) {
      // position servos (model-wide switch): actc = kp * ctrl, D_GEAR = kp; the engine clamps gain*ctrl + bias as a whole
      const float act_k = (P.act_position && k < nl) ? fminf(fmaxf(fmaf(-LK(k, LM_D_GEAR), qc[k], actc[k]), LK(k, LM_D_FLO)), LK(k, LM_D_FHI)) : actc[k];
      sm_c[k] = (k < nl) ? (-STIFF_C(k) * qc[k] - DAMP_C(k) * vc[k] - bias_c[k] + act_k + musc[k]) : 0.0f;
    }
    // park M and the twists in lane memory
#pragma unroll
    for (int i = 0; i < MC * (MC + 1) / 2; i++) LMEM(LMm::kMcc + i) = Mcc[i];
#pragma unroll
    for (int k = 0; k < MC; k++)
#pragma unroll
      for (int r = 0; r < 6; r++) LMEM(LMm::kMcr + k * 6 + r) = Mcr[k][r];
#pragma unroll
    for (int i = 0; i < 21; i++) LMEM(LMm::kMrr + i) = Mrr[i];
    if constexpr (!DEFER) {
#pragma unroll
      for (int i = 0; i < 6; i++) {
        LMEM(LMm::kSr + i * 6 + 0) = Sr[i].w.x; LMEM(LMm::kSr + i * 6 + 1) = Sr[i].w.y; LMEM(LMm::kSr + i * 6 + 2) = Sr[i].w.z;
        LMEM(LMm::kSr + i * 6 + 3) = Sr[i].v.x; LMEM(LMm::kSr + i * 6 + 4) = Sr[i].v.y; LMEM(LMm::kSr + i * 6 + 5) = Sr[i].v.z;
      }
    }
#pragma unroll
    for (int k = 0; k < MC; k++) {
      LMEM(LMm::kSc + k * 6 + 0) = Sc[k].w.x; LMEM(LMm::kSc + k * 6 + 1) = Sc[k].w.y; LMEM(LMm::kSc + k * 6 + 2) = Sc[k].w.z;
      LMEM(LMm::kSc + k * 6 + 3) = Sc[k].v.x; LMEM(LMm::kSc + k * 6 + 4) = Sc[k].v.y; LMEM(LMm::kSc + k * 6 + 5) = Sc[k].v.z;
    }
    {
      float Lrr[21], zero21[21];
#pragma unroll
      for (int i = 0; i < 21; i++) zero21[i] = 0;
      arrow_factor<Q, MC>(Mcc, Mcr, Mrr, zero21, Lrr);       // M's register copy is consumed here
#pragma unroll
      for (int i = 0; i < 6; i++) a0r[i] = sm_r[i];
#pragma unroll
      for (int k = 0; k < MC; k++) a0c[k] = sm_c[k];
      arrow_solve<Q, MC>(Mcc, Mcr, Lrr, a0c, a0r);
      if (anydup) tie_shared_dof<Q, MC>(Mcc, Mcr, Lrr, a0c, a0r, duprole);
    }
  }
  // cross-chain contacts: who is coupled with whom (quad-uniform bits: 4 i + j = chains i and j share a contact). The
  // factorisation eliminates the lanes in order and carries the cross blocks along (arrow_factor_g): exact for any pattern.
  bool any_pair = false;
  int adj0 = 0;
  constexpr int NX = PAIRS ? ((MC <= 3 || MC == 6) ? 3 : 2) : 1;         // cross blocks a lane may hold: chains above it (quadruped and six-link robots 4 chains, humanoids 3)
  if (PAIRS) {
    adj0 = (int)(Q::sum((float)(pair_mask_out << (4 * c))) + 0.5f);
    // symmetric closure (a lane that ran out of slots may lack its mirror of a contact its partner holds)
    int sym = adj0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) if ((adj0 >> (4 * i + j)) & 1) sym |= 1 << (4 * j + i);
    adj0 = sym;
    any_pair = adj0 != 0;
    if (NX < 3 && (adj0 >> 12) != 0) any_pair = false;         // (a fourth chain of a five-link humanoid has no cross-block storage: never lowered with pairs)
  }
  auto ldS = [&](int base) -> Sp {       // twist from lane memory
    Sp S; S.w = v3(LMEM(base), LMEM(base + 1), LMEM(base + 2)); S.v = v3(LMEM(base + 3), LMEM(base + 4), LMEM(base + 5)); return S;
  };

  oz = LM_OPAQUE_ZERO();
  // ================= unit constraint rows: friction loss, joint limits =================
  float fl_aref_r[6], fl_aref_c[MC];     // friction-loss reference accelerations
  float lim_s_c[MC], lim_D_c[MC], lim_aref_c[MC];   // active limit: sign (+1 lower, -1 upper, 0 none)
  // limit rows of the ROOT dofs (replicated in the four lanes, counted once like the root's friction-loss rows): compiled into the
  // muscle families (HumanoidMuscle's pelvis joints are `limited`, humanoid_muscle.xml: measured free there), into EVERY family's
  // replay kernel and into the run-time-cone kernels. The regular kernels of the other families sit at the register ceiling and no
  // robot of the path needs the rows there (lowering.py keeps a root limit only when it can become active): they only LOOK — a
  // limited root dof beyond its range hands the control step to the family's replay kernel (lm_step.h), from the untouched state; the
  // rows would have been inactive in every pass before that one, so the replay follows the same trajectory up to it.
  constexpr bool ROOT_LIM = NM > 0 || NS > 8 || CONE < 0;
  // the quadruped family (three links per chain, elliptic cones compiled in): its models have the root's translations as slides along
  // +x, +y, +z in this order, in a root frame that is the world's (checked when a model is given that family: lm_kernels.hip family_of)
  constexpr bool ROOT_XYZ = MC == 3 && CONE == 1;
  if constexpr (!ROOT_LIM) {
    if (P.root_limited) {        // (a scalar branch: no robot of the path takes it)
#pragma unroll
      for (int i = 0; i < 6; i++) if (RD(i, LM_D_LIMITED) != 0.0f && (qr[i] < RD(i, LM_D_LO) || qr[i] > RD(i, LM_D_HI))) cnt.need_full = 1;
    }
  }
  constexpr int NRL = ROOT_LIM ? 6 : 1;
  float lim_s_r[NRL], lim_D_r[NRL], lim_aref_r[NRL];
#pragma unroll
  for (int i = 0; i < NRL; i++) { lim_s_r[i] = 0; lim_D_r[i] = 0; lim_aref_r[i] = 0; }
  if constexpr (ROOT_LIM) {
#pragma unroll
    for (int i = 0; i < 6; i++) if (RD(i, LM_D_LIMITED) != 0.0f) {
      const float dlo = qr[i] - RD(i, LM_D_LO), dhi = RD(i, LM_D_HI) - qr[i];
      const float sgn = (dlo < 0.0f) ? 1.0f : ((dhi < 0.0f) ? -1.0f : 0.0f);
      if (sgn != 0.0f) {
        const float dist = (sgn > 0) ? dlo : dhi;
        const float imp = impedance(&RD(i, LM_D_LIM_S0), 1, dist, 0.0f);
        const float Rl = fmaxf(kMinVal, (1.0f - imp) * RDV(i, 1, LM_D_INVW) / imp);
        lim_s_r[i] = sgn; lim_D_r[i] = 1.0f / Rl;
        lim_aref_r[i] = -RD(i, LM_D_LIM_B) * (sgn * vr[i]) - RD(i, LM_D_LIM_K) * imp * dist;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) fl_aref_r[i] = -RD(i, LM_D_FLOSS_B) * vr[i];
#pragma unroll
  for (int k = 0; k < MC; k++) {
    fl_aref_c[k] = 0; lim_s_c[k] = 0; lim_D_c[k] = 0; lim_aref_c[k] = 0;
    if (k < nl) {
      fl_aref_c[k] = -LK(k, LM_D_FLOSS_B) * vc[k];
      if (LK(k, LM_D_LIMITED) != 0.0f) {
        float dlo = qc[k] - LK(k, LM_D_LO), dhi = LK(k, LM_D_HI) - qc[k];
        float sgn = (dlo < 0.0f) ? 1.0f : ((dhi < 0.0f) ? -1.0f : 0.0f);
        if (sgn != 0.0f) {
          float dist = (sgn > 0) ? dlo : dhi;
          float imp = impedance(&LK(k, LM_D_LIM_S0), LM_NCHAIN, dist, 0.0f);
          float Rl = fmaxf(kMinVal, (1.0f - imp) * LKV(k, 1, LM_D_INVW) / imp);
          lim_s_c[k] = sgn; lim_D_c[k] = 1.0f / Rl;
          lim_aref_c[k] = -LK(k, LM_D_LIM_B) * (sgn * vc[k]) - LK(k, LM_D_LIM_K) * imp * dist;
        }
      }
    }
  }
  LM_TICK(2);

  // ================= constraint solve: Newton on the primal problem =================
  // y = M x (M from lane memory); yr is fully summed (replicated)
  auto mulM = [&](const float* xr, const float* xc, float* yr, float* yc) {
#pragma unroll
    for (int r = 0; r < 6; r++) yr[r] = 0;
#pragma unroll
    for (int k = 0; k < MC; k++) {
      float t = 0;
#pragma unroll
      for (int r = 0; r < 6; r++) { float m = LMEM(LMm::kMcr + k * 6 + r); t = fmaf(m, xr[r], t); yr[r] = fmaf(m, xc[k], yr[r]); }
#pragma unroll
      for (int j = 0; j < MC; j++) t = fmaf(LMEM(LMm::kMcc + ((j <= k) ? tri(k, j) : tri(j, k))), xc[j], t);
      yc[k] = t;
    }
#pragma unroll
    for (int r = 0; r < 6; r++) {
      float t = Q::sum(yr[r]);
#pragma unroll
      for (int j = 0; j < 6; j++) t = fmaf(LMEM(LMm::kMrr + ((j <= r) ? tri(r, j) : tri(j, r))), xr[j], t);
      yr[r] = t;
    }
  };
  // spatial image of joint-space vector x on every link of this chain: Al[k] = sum_root x_r S_r + sum_{j<=k} x_j S_j
  // (kept in lane memory: a register array indexed by the contact's link number ends up in scratch)
  Sp img_root = sp0();             // root image of the vector last handed to link_images (contacts of root geoms, trunk pairs)
  auto link_images = [&](const float* xr, const float* xc) {
    Sp A = sp0();
#pragma unroll
    for (int r = 0; r < 6; r++) A = A + xr[r] * ldSr(r);
    img_root = A;
#pragma unroll
    for (int k = 0; k < MC; k++) {
      A = A + xc[k] * ldS(LMm::kSc + k * 6);
      LMEM(LMm::kAl + k * 6 + 0) = A.w.x; LMEM(LMm::kAl + k * 6 + 1) = A.w.y; LMEM(LMm::kAl + k * 6 + 2) = A.w.z;
      LMEM(LMm::kAl + k * 6 + 3) = A.v.x; LMEM(LMm::kAl + k * 6 + 4) = A.v.y; LMEM(LMm::kAl + k * 6 + 5) = A.v.z;
    }
    if (PAIRS && any_pair) Q::quad_sync();     // mirror slots read the partner lane's images
  };
  auto pick = [&](int link) -> Sp { return (link < 0) ? img_root : ldS(LMm::kAl + (link < 0 ? 0 : link) * 6); };
  // ---- slots of self-contacts (PAIRS): the row space is the motion of body 2 against body 1, the frame is the slot's own
  auto slot_sign = [&](int s) -> float { return PAIRS ? SL(s, SL_PART) : 0.0f; };                 // 0: floor contact
  auto slot_frame = [&](int s, V3& n, V3& t1, V3& t2) {
    n = v3(SL(s, SL_NX), SL(s, SL_NY), SL(s, SL_NZ)); make_frame(n, t1, t2);
  };
  auto peer_twist = [&](int dl, int base) -> Sp {
    Sp S; S.w = v3(PEER(dl, base), PEER(dl, base + 1), PEER(dl, base + 2)); S.v = v3(PEER(dl, base + 3), PEER(dl, base + 4), PEER(dl, base + 5)); return S;
  };
  // contact-frame components of the current link images for slot s
  auto slot_rows_of_images = [&](int s, float* out) {
    const int link = (int)SL(s, SL_LINK);
    const V3 rc = v3(SL(s, SL_RX), SL(s, SL_RY), SL(s, SL_RZ));
    const float part = slot_sign(s);
    if (PAIRS && part != 0.0f) {
      const int code = (int)fabsf(part) - 1, pl = code & 7, dl = (code >> 3) - c;
      const Sp B = (pl == 7) ? img_root : peer_twist(dl, LMm::kAl + pl * 6);
      const float sg = (part > 0.0f) ? 1.0f : -1.0f;
      const Sp A = sg * (pick(link) + (-1.0f) * B);
      V3 n, t1, t2;
      slot_frame(s, n, t1, t2);
      frame_rows(A, rc, n, t1, t2, out);
    } else contact_rows(pick(link), rc, out);
  };
  auto aligned_from = [&](int lo, int first, int step) -> int { int r = (first - lo) % step; if (r < 0) r += step; return lo + r; };
  // a contact between two chains has a slot in both lanes: each of them carries half of its cost
  auto slot_weight = [&](int s) -> float {
    if (!PAIRS) return 1.0f;
    const float part = slot_sign(s);
    const int code = (int)fabsf(part) - 1;          // mirrored: the partner is a link of ANOTHER chain (not the root body, not my own chain)
    return (part != 0.0f && (code & 7) != 7 && (code >> 3) != c) ? 0.5f : 1.0f;
  };
  auto friction_cost = [&](float x, float f, float Rr) -> float {
    if (f <= 0.0f) return 0.0f;
    float Rf = Rr * f;
    if (x <= -Rf) return -0.5f * Rf * f - f * x;
    if (x >= Rf) return -0.5f * Rf * f + f * x;
    return 0.5f * x * x / Rr;
  };
  // lane-partial constraint cost at acceleration (xr, xc) (root rows weighted so that the quad sum counts them once)
  // `dst`: where the row residuals J x - aref of the contact slots are left (SL_JAR | SL_JV as scratch): the Newton loop starts from
  // the residuals of the point the warm start picks instead of building them a third time
  auto cost_at = [&](const float* xr, const float* xc, auto dst) -> float {
    constexpr int DST = decltype(dst)::value;
    // with replicas: the unit rows are counted by replica 0, the contact slots are dealt round-robin (callers add
    // the parts up with Q::rep_sum)
    float cost = 0, cr = 0;
    if (Q::rep() == 0) {
#pragma unroll
      for (int i = 0; i < 6; i++) cr += friction_cost(xr[i] - fl_aref_r[i], FLOSS_R(i), (inr_on ? dp->rfl_r[i] : RD(i, LM_D_FLOSS_R)));
      if constexpr (ROOT_LIM) {
#pragma unroll
        for (int i = 0; i < 6; i++) { const float x = lim_s_r[i] * xr[i] - lim_aref_r[i]; if (lim_s_r[i] != 0.0f && x < 0.0f) cr += 0.5f * lim_D_r[i] * x * x; }
      }
      cost = w0 * cr;
#pragma unroll
      for (int k = 0; k < MC; k++) if (k < nl) {
        cost += friction_cost(xc[k] - fl_aref_c[k], FLOSS_C(k), (inr_on ? dp->rfl_c[k] : LK(k, LM_D_FLOSS_R)));
        float x = lim_s_c[k] * xc[k] - lim_aref_c[k];
        if (lim_s_c[k] != 0.0f && x < 0.0f) cost += 0.5f * lim_D_c[k] * x * x;
      }
    }
    if (nslot > 0 || any_pair) {
      link_images(xr, xc);
      auto cost_slot = [&](int s, auto is_pair) {
        constexpr bool IP = decltype(is_pair)::value;
        float Dj[6], fr[5], jar[6];
        if constexpr (IP) slot_rows_of_images(s, jar);
        else contact_rows(pick((int)SL(s, SL_LINK)), v3(SL(s, SL_RX), SL(s, SL_RY), SL(s, SL_RZ)), jar);
        const int dim = (int)SL(s, SL_DIM);
        if (PYR3(dim)) {
          float x[4], f3[3], cs = 0.0f;
          pyr_rows(jar, SL(s, SL_MU), x);
#pragma unroll
          for (int r = 0; r < 4; r++) { x[r] -= SL(s, SL_AREF + r); SL(s, DST + r) = x[r]; }
          pyr_force(x, SL(s, SL_D), SL(s, SL_MU), f3, cs);
          cost += (IP ? slot_weight(s) : 1.0f) * cs;
        } else {
#pragma unroll
          for (int j = 0; j < 6; j++) { jar[j] -= SL(s, SL_AREF + j); SL(s, DST + j) = jar[j]; Dj[j] = SL(s, SL_D + j); }
#pragma unroll
          for (int j = 0; j < 5; j++) fr[j] = SL(s, SL_FR + j);
          cost += (IP ? slot_weight(s) : 1.0f) * cone_eval<false>(jar, Dj, fr, SL(s, SL_MU), dim).cost;
        }
      };
      for (int s = Q::rep(); s < nfloor; s += Q::kRep) cost_slot(s, std::false_type{});
      if constexpr (PAIRS) {
        if (nslot > nfloor) for (int s = aligned_from(nfloor, Q::rep(), Q::kRep); s < nslot; s += Q::kRep) cost_slot(s, std::true_type{});
      }
      if (PAIRS && any_pair) Q::quad_sync();        // ... before the next link_images overwrites them
    }
    return cost;
  };

  float ar[6], ac[MC];
  {
    // warm start: the better of (previous qacc, qacc_smooth)
    float cost_smooth = Q::rep_sum(Q::sum(cost_at(a0r, a0c, std::integral_constant<int, (int)SL_JAR>{})));
    float yr[6], yc[MC], gauss = 0, gr = 0;
    mulM(war, wac, yr, yc);
#pragma unroll
    for (int k = 0; k < MC; k++) gauss += 0.5f * (yc[k] - sm_c[k]) * (wac[k] - a0c[k]);
#pragma unroll
    for (int i = 0; i < 6; i++) gr += 0.5f * (yr[i] - sm_r[i]) * (war[i] - a0r[i]);
    float cost_warm = Q::rep_sum(Q::sum(cost_at(war, wac, std::integral_constant<int, (int)SL_JV>{}))) + Q::sum(gauss + w0 * gr);
    bool use_warm = cost_warm < cost_smooth;
    if (use_warm) {
      // the row residuals of the chosen point into SL_JAR (every replica those of the slots it was dealt above)
      for (int s = Q::rep(); s < nslot; s += Q::kRep) {
        if (PYR3((int)SL(s, SL_DIM))) {
#pragma unroll
          for (int r = 0; r < 4; r++) SL(s, SL_JAR + r) = SL(s, SL_JV + r);
        } else {
#pragma unroll
          for (int j = 0; j < 6; j++) SL(s, SL_JAR + j) = SL(s, SL_JV + j);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) ar[i] = use_warm ? war[i] : a0r[i];
#pragma unroll
    for (int k = 0; k < MC; k++) ac[k] = use_warm ? wac[k] : a0c[k];
  }

  LM_TICK(3);
  // Newton bookkeeping (round 5; the engine's own solver keeps Jaref and Ma the same way): the row residuals J a - aref of the contact
  // slots and M a are MOVED by alpha x (J s, M s) after a line search instead of being rebuilt from the link images in every iteration,
  // the first iteration starts from the residuals the warm-start comparison left behind, and the line search does not evaluate alpha = 0
  // (phi'(0) = g . s, phi''(0) = s . H s = -g . s along the Newton direction). Same iteration counts, same parity; quadruped bench
  // rollout -9.6 %, Atlas + DR -12 %, Talos -8 %, HumanoidMuscle -7 %, HumanoidTorque -6 % (profiles/r5_notes.md §7). Carrying M a costs
  // the six-link kernels more in spills than it saves (UnitreeG1 +5 %): they recompute it. LM_JAR_RECOMPUTE / LM_MA_RECOMPUTE /
  // LM_LS_EVAL0: the bookkeeping of rounds 1-4, for A/B builds.
#ifdef LM_JAR_RECOMPUTE
  constexpr bool kJarIncr = false, kMaIncr = false;
#elif defined(LM_MA_RECOMPUTE)
  constexpr bool kJarIncr = true, kMaIncr = false;
#else
  constexpr bool kJarIncr = true, kMaIncr = MC < 6;
#endif
  float qf_r[6], qf_c[MC];      // constraint forces in joint space
#pragma unroll
  for (int i = 0; i < 6; i++) qf_r[i] = 0;
#pragma unroll
  for (int k = 0; k < MC; k++) qf_c[k] = 0;
  bool has_rows = false;
#pragma unroll
  for (int i = 0; i < 6; i++) has_rows = has_rows || (FLOSS_R(i) > 0.0f) || (ROOT_LIM && lim_s_r[ROOT_LIM ? i : 0] != 0.0f);
#pragma unroll
  for (int k = 0; k < MC; k++) has_rows = has_rows || (k < nl && (FLOSS_C(k) > 0.0f || lim_s_c[k] != 0.0f));
  has_rows = has_rows || nslot > 0;
  bool done = !(Q::sum(has_rows ? 1.0f : 0.0f) > 0.0f);   // quad-uniform: nothing to solve in this environment
  int iters = 0;

  float Mar[6], Mac[MC];          // M a, moved by alpha x M s after a line search like the row residuals
  for (int it = 0; it <= P.iterations; it++) {
    if (!Q::any(!done)) break;
    oz = LM_OPAQUE_ZERO();
    if (!done) {
      // ---- gradient at the current point
      float jfr_r[6], jfr_c[MC], jlim_c[MC];     // jar of the unit rows
      if (kMaIncr && it > 0) {}
      else if (!(P.ablate & 16)) mulM(ar, ac, Mar, Mac);
      else {
#pragma unroll
        for (int i = 0; i < 6; i++) Mar[i] = ar[i];
#pragma unroll
        for (int k = 0; k < MC; k++) Mac[k] = ac[k];
      }
      float fu_r[6];
      // friction-loss rows: force = -clamp(x/R, -f, f); quadratic zone (Hessian 1/R) iff |x| < R f
      float ff_r[6], iR_r[6], ff_c[MC], iR_c[MC];
      unsigned act_fr_r = 0, act_fr_c = 0, act_lim = 0, act_lim_r = 0;   // rows in their quadratic zone (Hessian)
      float jlim_r[NRL];
#pragma unroll
      for (int i = 0; i < 6; i++) {
        ff_r[i] = FLOSS_R(i); const float Rr = (inr_on ? dp->rfl_r[i] : RD(i, LM_D_FLOSS_R));
        iR_r[i] = (ff_r[i] > 0.0f) ? 1.0f / Rr : 0.0f;
        const float x = ar[i] - fl_aref_r[i];
        jfr_r[i] = x;
        fu_r[i] = -fminf(fmaxf(x * iR_r[i], -ff_r[i]), ff_r[i]);
        if (fabsf(x) < Rr * ff_r[i]) act_fr_r |= 1u << i;
        if constexpr (ROOT_LIM) {
          jlim_r[i] = lim_s_r[i] * ar[i] - lim_aref_r[i];
          if (jlim_r[i] < 0.0f && lim_s_r[i] != 0.0f) { fu_r[i] -= lim_s_r[i] * lim_D_r[i] * jlim_r[i]; act_lim_r |= 1u << i; }
        }
      }
#pragma unroll
      for (int k = 0; k < MC; k++) {
        ff_c[k] = (k < nl) ? FLOSS_C(k) : 0.0f; const float Rr = (k < nl) ? (inr_on ? dp->rfl_c[k] : LK(k, LM_D_FLOSS_R)) : 1.0f;
        iR_c[k] = (ff_c[k] > 0.0f) ? 1.0f / Rr : 0.0f;
        jfr_c[k] = ac[k] - fl_aref_c[k];
        jlim_c[k] = lim_s_c[k] * ac[k] - lim_aref_c[k];
        float fu = -fminf(fmaxf(jfr_c[k] * iR_c[k], -ff_c[k]), ff_c[k]);
        if (fabsf(jfr_c[k]) < Rr * ff_c[k]) act_fr_c |= 1u << k;
        if (jlim_c[k] < 0.0f && lim_s_c[k] != 0.0f) { fu -= lim_s_c[k] * lim_D_c[k] * jlim_c[k]; act_lim |= 1u << k; }
        qf_c[k] = fu;
      }
      Sp Fl[MC];                 // contact wrench sums per link
#pragma unroll
      for (int k = 0; k < MC; k++) Fl[k] = sp0();
      // replicated small-batch layout: when some lane of the wave holds several contacts, the replicas of an environment
      // take every kRep-th slot each (forces here, Hessian blocks below) and add their parts up with the symmetric
      // butterfly, so that all replicas continue with bit-identical numbers
      // decided per ENVIRONMENT (quad), not per wave: the arithmetic of an environment must not depend on which other
      // environments share its wave (ragged batches and shards of any size agree bitwise)
      const bool split = Q::kRep > 1 && Q::sum((nslot > 1) ? 1.0f : 0.0f) > 0.0f;
      const int s_first = split ? Q::rep() : 0, s_step = split ? Q::kRep : 1;
      Q::fence();
      Sp Frt = sp0();              // wrench of the contacts of root geoms held by this lane
      Sp Fp[PAIRS ? MC : 1];       // wrenches of the self-contacts per link: they act on the chain dofs only (the opposite
#pragma unroll                     // wrench on the other body cancels them on the root)
      for (int k = 0; k < (PAIRS ? MC : 1); k++) Fp[k] = sp0();
      // The row residuals J a - aref of the contact slots are built from the link images of a in the FIRST iteration only; after a line
      // search they move by alpha x (J s), which the line search has in SL_JV (below) — the engine's own Newton does the same
      // (Jaref += alpha Jv). LM_JAR_RECOMPUTE: rounds 1-4 (rebuilt in every iteration), for A/B builds.

      const bool fresh_rows = !kJarIncr;          // (the first iteration starts from the residuals the warm start left: cost_at)
      if ((nslot > 0 || any_pair) && !(P.ablate & 32)) {
        if (fresh_rows) link_images(ar, ac);
        auto grad_slot = [&](int s, auto is_pair) {
          constexpr bool IP = decltype(is_pair)::value;
          float Dj[6], fr[5], jar[6];
          const int link = (int)SL(s, SL_LINK);
          const V3 rc = v3(SL(s, SL_RX), SL(s, SL_RY), SL(s, SL_RZ));
          if (fresh_rows) {
            if constexpr (IP) slot_rows_of_images(s, jar);
            else contact_rows(pick(link), rc, jar);
          }
          const int dim = (int)SL(s, SL_DIM);
          float fc[6];
          int zone;
          if (PYR3(dim)) {
            float x[4], dummy = 0;
            if (fresh_rows) {
              pyr_rows(jar, SL(s, SL_MU), x);
#pragma unroll
              for (int r = 0; r < 4; r++) { x[r] -= SL(s, SL_AREF + r); SL(s, SL_JAR + r) = x[r]; }
            } else {
#pragma unroll
              for (int r = 0; r < 4; r++) x[r] = SL(s, SL_JAR + r);
            }
            zone = (int)pyr_force(x, SL(s, SL_D), SL(s, SL_MU), fc, dummy);
            fc[3] = fc[4] = fc[5] = 0.0f;
          } else {
#pragma unroll
            for (int j = 0; j < 6; j++) {
              if (fresh_rows) { jar[j] -= SL(s, SL_AREF + j); SL(s, SL_JAR + j) = jar[j]; }
              else jar[j] = SL(s, SL_JAR + j);
              Dj[j] = SL(s, SL_D + j);
            }
#pragma unroll
            for (int j = 0; j < 5; j++) fr[j] = SL(s, SL_FR + j);
            ConeEval e = cone_eval<true>(jar, Dj, fr, SL(s, SL_MU), dim);
            zone = e.zone;
#pragma unroll
            for (int j = 0; j < 6; j++) fc[j] = e.f[j];
          }
          SL(s, SL_ZONE) = (float)zone;
          if (zone) {
            if constexpr (IP) {
              V3 n, t1, t2;
              slot_frame(s, n, t1, t2);
              const float part = slot_sign(s);
              const Sp Fw = ((part > 0.0f) ? 1.0f : -1.0f) * frame_wrench(fc, rc, n, t1, t2);
              const int pcode = (int)fabsf(part) - 1, plink = ((pcode >> 3) == c && (pcode & 7) != 7) ? (pcode & 7) : -1;   // partner link of my own chain
#pragma unroll
              for (int k = 0; k < MC; k++) { if (link == k) Fp[k] = Fp[k] + Fw; if (plink == k) Fp[k] = Fp[k] + (-1.0f) * Fw; }
            } else {
              Sp Fw = contact_wrench(fc, rc);
              if (link < 0) Frt = Frt + Fw;
#pragma unroll
              for (int k = 0; k < MC; k++) if (link == k) Fl[k] = Fl[k] + Fw;
            }
          }
        };
        for (int s = s_first; s < nfloor; s += s_step) grad_slot(s, std::false_type{});
        if constexpr (PAIRS) {
          if (nslot > nfloor) for (int s = aligned_from(nfloor, s_first, s_step); s < nslot; s += s_step) grad_slot(s, std::true_type{});
        }
        if (PAIRS && any_pair) Q::quad_sync();
      }
      Q::fence();                  // SL_JAR / SL_ZONE of a slot are written by the replica that owns it
      if (split) {
#pragma unroll
        for (int k = 0; k < MC; k++) {
          Fl[k].w = v3(Q::rep_sum(Fl[k].w.x), Q::rep_sum(Fl[k].w.y), Q::rep_sum(Fl[k].w.z));
          Fl[k].v = v3(Q::rep_sum(Fl[k].v.x), Q::rep_sum(Fl[k].v.y), Q::rep_sum(Fl[k].v.z));
          if (PAIRS && npairslot > 0) {       // the replicas of a lane agree on its slot list: a uniform branch for them
            Fp[k].w = v3(Q::rep_sum(Fp[k].w.x), Q::rep_sum(Fp[k].w.y), Q::rep_sum(Fp[k].w.z));
            Fp[k].v = v3(Q::rep_sum(Fp[k].v.x), Q::rep_sum(Fp[k].v.y), Q::rep_sum(Fp[k].v.z));
          }
        }
        if (nrootslot > 0) {
          Frt.w = v3(Q::rep_sum(Frt.w.x), Q::rep_sum(Frt.w.y), Q::rep_sum(Frt.w.z));
          Frt.v = v3(Q::rep_sum(Frt.v.x), Q::rep_sum(Frt.v.y), Q::rep_sum(Frt.v.z));
        }
      }
      Sp Fsum = sp0(), Psum = sp0();
#pragma unroll
      for (int k = MC - 1; k >= 0; k--) {
        Fsum = Fsum + Fl[k];
        if (PAIRS) Psum = Psum + Fp[k];
        qf_c[k] += spdot(ldS(LMm::kSc + k * 6), PAIRS ? Fsum + Psum : Fsum);
      }
      Fsum = Fsum + Frt;           // what the root dofs feel: the chain's floor contacts and the root geoms' contacts
      float gc[MC], gr_[6], g2 = 0;
#pragma unroll
      for (int k = 0; k < MC; k++) { gc[k] = Mac[k] - sm_c[k] - qf_c[k]; g2 = fmaf(gc[k], gc[k], g2); }
#pragma unroll
      for (int i = 0; i < 6; i++) {
        qf_r[i] = fu_r[i] + ((P.ablate & 64) ? 0.0f : Q::sum(spdot(ldSr(i), Fsum)));
        gr_[i] = Mar[i] - sm_r[i] - qf_r[i];
      }
      float gnorm2 = Q::sum(g2);
#pragma unroll
      for (int i = 0; i < 6; i++) gnorm2 = fmaf(gr_[i], gr_[i], gnorm2);
      LM_TICK(4);
      if (nscale * sqrtf(gnorm2) < P.tolerance || it == P.iterations) done = true;
      else {
        // ---- Hessian H = M + J^T W J (arrow blocks), factor, Newton direction
        float Hcc[MC * (MC + 1) / 2], Hcr[MC][6], Hpart[21], Hrep[21];
        float Xc[NX][PAIRS ? MC : 1][PAIRS ? MC : 1];  // cross blocks H_ab of this chain a with the chains b above it (slot b - a - 1)
#pragma unroll
        for (int x = 0; x < NX; x++)
#pragma unroll
          for (int i = 0; i < (PAIRS ? MC : 1); i++)
#pragma unroll
            for (int j = 0; j < (PAIRS ? MC : 1); j++) Xc[x][i][j] = 0.0f;
        // when the slots are split over the replicas, only replica 0 starts from M (+ unit-row terms); the butterfly
        // sum below then gives every replica M + all contact blocks
        const float own = (split && Q::rep() != 0) ? 0.0f : 1.0f;
#pragma unroll
        for (int i = 0; i < MC * (MC + 1) / 2; i++) Hcc[i] = own * LMEM(LMm::kMcc + i);
#pragma unroll
        for (int i = 0; i < 21; i++) Hrep[i] = LMEM(LMm::kMrr + i);
#pragma unroll
        for (int k = 0; k < MC; k++) {
#pragma unroll
          for (int r = 0; r < 6; r++) Hcr[k][r] = own * LMEM(LMm::kMcr + k * 6 + r);
          if (act_fr_c & (1u << k)) Hcc[tri(k, k)] += own * iR_c[k];
          if (act_lim & (1u << k)) Hcc[tri(k, k)] += own * lim_D_c[k];
        }
#pragma unroll
        for (int i = 0; i < 21; i++) Hpart[i] = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) if (act_fr_r & (1u << i)) Hrep[tri(i, i)] += iR_r[i];
        if constexpr (ROOT_LIM) {
#pragma unroll
          for (int i = 0; i < 6; i++) if (act_lim_r & (1u << i)) Hrep[tri(i, i)] += lim_D_r[i];
        }
        auto hess_slot = [&](int s, auto is_pair) {
          constexpr bool IP = decltype(is_pair)::value;
          oz = LM_OPAQUE_ZERO();
          const int zone = (int)SL(s, SL_ZONE);
          if (zone == 0) return;
          float Dj[6], fr[5], Hc[21], jar[6];
          const int dim = (int)SL(s, SL_DIM);
          if (PYR3(dim)) pyr_hessian((unsigned)zone, SL(s, SL_D), SL(s, SL_MU), Hc);
          else {
#pragma unroll
            for (int j = 0; j < 6; j++) { jar[j] = SL(s, SL_JAR + j); Dj[j] = SL(s, SL_D + j); }
#pragma unroll
            for (int j = 0; j < 5; j++) fr[j] = SL(s, SL_FR + j);
            cone_hessian(jar, Dj, fr, SL(s, SL_MU), dim, zone, Hc);
          }
          const V3 rc = v3(SL(s, SL_RX), SL(s, SL_RY), SL(s, SL_RZ));
          const int link = (int)SL(s, SL_LINK);
          float Jc[6 + MC][6];
          float Jp[PAIRS ? MC : 1][6];      // the PARTNER chain's columns of a cross-chain contact (lower lane of the pair only)
          bool cross = false;
          int xslot = 0;                   // slot of the cross block this contact feeds (partner lane - my lane - 1)
          if constexpr (IP) {
            const float part = slot_sign(s);
            // self-contact: the row space is the motion of body 2 against body 1 -> the root columns vanish, my chain's
            // columns carry my sign, the partner chain's columns the opposite one
            const int code = (int)fabsf(part) - 1, pl = code & 7, pc = code >> 3, dl = pc - c;
            const float sg = (part > 0.0f) ? 1.0f : -1.0f;
            V3 n, t1, t2;
            slot_frame(s, n, t1, t2);
            const bool own_pair = pl != 7 && pc == c;          // both bodies in my chain: the joints up to the nearer one cancel
#pragma unroll
            for (int k = 0; k < MC; k++) {
              const float coef = ((k <= link) ? 1.0f : 0.0f) - ((own_pair && k <= pl) ? 1.0f : 0.0f);
              if (coef != 0.0f) frame_rows((sg * coef) * ldS(LMm::kSc + k * 6), rc, n, t1, t2, Jc[6 + k]);
              else {
#pragma unroll
                for (int j = 0; j < 6; j++) Jc[6 + k][j] = 0;
              }
            }
            cross = pl != 7 && !own_pair && pc > c && any_pair;   // the lower lane of the pair keeps the cross block
            xslot = pc - c - 1;
            if (cross) {
#pragma unroll
              for (int k = 0; k < MC; k++) {
                if (k <= pl) frame_rows((-sg) * peer_twist(dl, LMm::kSc + k * 6), rc, n, t1, t2, Jp[k]);
                else {
#pragma unroll
                  for (int j = 0; j < 6; j++) Jp[k][j] = 0;
                }
              }
            }
          } else {
#pragma unroll
            for (int r = ROOT_XYZ ? 3 : 0; r < 6; r++) if (r >= 3 || !(MC >= 5 && P.root_xyz)) contact_rows(ldSr(r), rc, Jc[r]);
#pragma unroll
            for (int k = 0; k < MC; k++) {
              if (k <= link) contact_rows(ldS(LMm::kSc + k * 6), rc, Jc[6 + k]);
              else {
#pragma unroll
                for (int j = 0; j < 6; j++) Jc[6 + k][j] = 0;
              }
            }
          }
          // J^T Hc J over the columns [root 6 | chain MC]. Two shortcuts (round 5): a SELF-contact has no root columns (the loop starts at
          // the chain's), and in the quadruped family (ROOT_XYZ) the root's three translation columns of a floor contact are the unit
          // rows -e2, +e1, +e0 of the contact frame (contact_rows of the slides along x, y, z): their products are picked out of Hc
          // and out of Hc J instead of multiplied — 342 instead of 594 multiply-adds per condim-6 slot
          auto accumulate = [&](auto nr_tag, auto unit_tag) {
            constexpr int NR = decltype(nr_tag)::value;
            constexpr bool UNIT = !IP && decltype(unit_tag)::value;
            constexpr int A0 = IP ? 6 : (UNIT ? 3 : 0);
            constexpr int ui[3] = {2, 1, 0};
            constexpr float us[3] = {-1.0f, 1.0f, 1.0f};
            if constexpr (UNIT) {
#pragma unroll
              for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b <= a; b++) Hpart[tri(a, b)] += (us[a] * us[b]) * Hc[(ui[a] <= ui[b]) ? tri(ui[b], ui[a]) : tri(ui[a], ui[b])];
            }
#pragma unroll
            for (int a = A0; a < 6 + MC; a++) {
              float t[NR];
#pragma unroll
              for (int i = 0; i < NR; i++) {
                float acc = 0;
#pragma unroll
                for (int j = 0; j < NR; j++) acc = fmaf(Hc[(j <= i) ? tri(i, j) : tri(j, i)], Jc[a][j], acc);
                t[i] = acc;
              }
              if constexpr (UNIT) {
#pragma unroll
                for (int b = 0; b < 3; b++) {
                  const float d = us[b] * t[ui[b]];
                  if (a < 6) Hpart[tri(a, b)] += d;
                  else Hcr[a - 6][b] += d;
                }
              }
#pragma unroll
              for (int b = A0; b <= a; b++) {
                float d = 0;
#pragma unroll
                for (int j = 0; j < NR; j++) d = fmaf(t[j], Jc[b][j], d);
                if (a < 6) Hpart[tri(a, b)] += d;
                else if (b < 6) Hcr[a - 6][b] += d;
                else Hcc[tri(a - 6, b - 6)] += d;
              }
              if constexpr (IP) if (cross && a >= 6) {
#pragma unroll
                for (int b = 0; b < MC; b++) {
                  float d = 0;
#pragma unroll
                  for (int j = 0; j < NR; j++) d = fmaf(t[j], Jp[b][j], d);
#pragma unroll
                  for (int xs = 0; xs < NX; xs++) if (xs == xslot) Xc[xs][a - 6][b] += d;
                }
              }
            }
          };
          if constexpr (IP || ROOT_XYZ || MC < 5) {
            if (CONE != 0 && dim > 3) accumulate(std::integral_constant<int, 6>{}, std::integral_constant<bool, ROOT_XYZ>{});
            else accumulate(std::integral_constant<int, 3>{}, std::integral_constant<bool, ROOT_XYZ>{});
          } else {
            // the humanoid families serve robots with either kind of root (Atlas, Talos, UnitreeH1 / G1: slides along x, y, z in a root
            // frame that is the world's; the two humanoids: a turned pelvis frame): a uniform branch on the model's flag
            if (P.root_xyz) {
              if (CONE != 0 && dim > 3) accumulate(std::integral_constant<int, 6>{}, std::true_type{});
              else accumulate(std::integral_constant<int, 3>{}, std::true_type{});
            } else {
              if (CONE != 0 && dim > 3) accumulate(std::integral_constant<int, 6>{}, std::false_type{});
              else accumulate(std::integral_constant<int, 3>{}, std::false_type{});
            }
          }
        };
        if (!(P.ablate & 2)) {
          for (int s = s_first; s < nfloor; s += s_step) hess_slot(s, std::false_type{});
          if constexpr (PAIRS) {
            if (nslot > nfloor) for (int s = aligned_from(nfloor, s_first, s_step); s < nslot; s += s_step) hess_slot(s, std::true_type{});
          }
        }
        if (PAIRS && any_pair) Q::quad_sync();
        if (split) {
          if (PAIRS && any_pair) {
#pragma unroll
            for (int x = 0; x < NX; x++)
#pragma unroll
              for (int i = 0; i < MC; i++)
#pragma unroll
                for (int j = 0; j < MC; j++) Xc[x][i][j] = Q::rep_sum(Xc[x][i][j]);
          }
#pragma unroll
          for (int i = 0; i < MC * (MC + 1) / 2; i++) Hcc[i] = Q::rep_sum(Hcc[i]);
#pragma unroll
          for (int k = 0; k < MC; k++)
#pragma unroll
            for (int r = 0; r < 6; r++) Hcr[k][r] = Q::rep_sum(Hcr[k][r]);
#pragma unroll
          for (int i = 0; i < 21; i++) Hpart[i] = Q::rep_sum(Hpart[i]);
        }
        LM_TICK(5);
        float Lr[21];
        float sr[6], sc[MC];
#pragma unroll
        for (int i = 0; i < 6; i++) sr[i] = -gr_[i];
#pragma unroll
        for (int k = 0; k < MC; k++) sc[k] = -gc[k];
        if (!(P.ablate & 4)) {
          bool coupled = false;
          if constexpr (PAIRS) {
            coupled = any_pair;
            if (coupled) {
              int adj = adj0;
              arrow_factor_g<Q, MC, NX>(Hcc, Hcr, Hrep, Hpart, Lr, Xc, c, adj);
              arrow_solve_g<Q, MC, NX>(Hcc, Hcr, Lr, Xc, c, adj, sc, sr);
              if (anydup) {
                // the two copies of a shared first link's dof stay ONE coordinate (tie_shared_dof), with the coupled factors
                float zc[MC], zr[6];
#pragma unroll
                for (int k = 0; k < MC; k++) zc[k] = 0.0f;
#pragma unroll
                for (int i = 0; i < 6; i++) zr[i] = 0.0f;
                zc[0] = (float)duprole;
                arrow_solve_g<Q, MC, NX>(Hcc, Hcr, Lr, Xc, c, adj, zc, zr);
                const float lam = Q::sum((float)duprole * sc[0]) / Q::sum((float)duprole * zc[0]);
#pragma unroll
                for (int k = 0; k < MC; k++) sc[k] = fmaf(-lam, zc[k], sc[k]);
#pragma unroll
                for (int i = 0; i < 6; i++) sr[i] = fmaf(-lam, zr[i], sr[i]);
                const float xa = Q::sum(duprole > 0 ? sc[0] : 0.0f);
                if (duprole < 0) sc[0] = xa;
              }
            }
          }
          if (!coupled) {
            arrow_factor<Q, MC>(Hcc, Hcr, Hrep, Hpart, Lr);
            arrow_solve<Q, MC>(Hcc, Hcr, Lr, sc, sr);
            if (anydup) tie_shared_dof<Q, MC>(Hcc, Hcr, Lr, sc, sr, duprole);
          }
        }

        LM_TICK(6);
        // ---- Newton decrement: lambda^2 = -g.s estimates twice the remaining cost gap
        float q1 = 0, q1r = 0;
#pragma unroll
        for (int k = 0; k < MC; k++) q1 = fmaf(sc[k], gc[k], q1);
#pragma unroll
        for (int i = 0; i < 6; i++) q1r = fmaf(sr[i], gr_[i], q1r);
        const float dec = -Q::sum(q1 + w0 * q1r);
#ifdef LM_LS_TRACE
        if (c == 0) printf(" newton it %d scaled decrement %.4g (tol %.3g) nslot %d\n", iters, nscale * dec, P.tolerance, nslot);
#endif
        if (!(nscale * dec >= P.tolerance)) done = true;        // converged (also catches NaN)
        else {
          iters++;
          // ---- exact line search along (sr, sc)
          float jv_r[6], jv_c[MC], jvlim_c[MC];
#pragma unroll
          for (int i = 0; i < 6; i++) jv_r[i] = sr[i];
#pragma unroll
          for (int k = 0; k < MC; k++) { jv_c[k] = sc[k]; jvlim_c[k] = lim_s_c[k] * sc[k]; }
          if (nslot > 0 || any_pair) {
            link_images(sr, sc);
            auto jv_slot = [&](int s, auto is_pair) {
              constexpr bool IP = decltype(is_pair)::value;
              float jv[6];
              if constexpr (IP) slot_rows_of_images(s, jv);
              else contact_rows(pick((int)SL(s, SL_LINK)), v3(SL(s, SL_RX), SL(s, SL_RY), SL(s, SL_RZ)), jv);
              if (PYR3((int)SL(s, SL_DIM))) {
                float xv[4];
                pyr_rows(jv, SL(s, SL_MU), xv);
#pragma unroll
                for (int r = 0; r < 4; r++) SL(s, SL_JV + r) = xv[r];
              } else {
#pragma unroll
                for (int j = 0; j < 6; j++) SL(s, SL_JV + j) = jv[j];
              }
            };
            // (by the replica that owns the slot in the gradient — rounds 1-4: every replica wrote every slot's rows; the line search
            // reads them behind the fence below)
            for (int s = s_first; s < nfloor; s += s_step) jv_slot(s, std::false_type{});
            if constexpr (PAIRS) {
              if (nslot > nfloor) for (int s = aligned_from(nfloor, s_first, s_step); s < nslot; s += s_step) jv_slot(s, std::true_type{});
            }
            if (PAIRS && any_pair) Q::quad_sync();
            if constexpr (CONE != 0) {
              // elliptic contacts: the line search's polynomials, prepared by the replica that owns the slot in the gradient
              for (int s = s_first; s < nslot; s += s_step) {
                const int dim = (int)SL(s, SL_DIM);
                if (PYR3(dim)) continue;
                const float mu = SL(s, SL_MU), x0 = SL(s, SL_JAR), v0 = SL(s, SL_JV), D0 = SL(s, SL_D);
                float UU = 0, UV = 0, VV = 0, A = D0 * x0 * v0, B = D0 * v0 * v0;
#pragma unroll
                for (int j = 1; j < 6; j++) if (j < dim) {
                  const float xj = SL(s, SL_JAR + j), vj = SL(s, SL_JV + j), fj = SL(s, SL_FR + j - 1), Dj_ = SL(s, SL_D + j);
                  const float u = xj * fj, v = vj * fj;
                  UU = fmaf(u, u, UU); UV = fmaf(u, v, UV); VV = fmaf(v, v, VV);
                  A = fmaf(Dj_ * xj, vj, A); B = fmaf(Dj_ * vj, vj, B);
                }
                SL(s, SL_PREP + 0) = x0 * mu; SL(s, SL_PREP + 1) = v0 * mu; SL(s, SL_PREP + 2) = UU; SL(s, SL_PREP + 3) = UV; SL(s, SL_PREP + 4) = VV;
                SL(s, SL_PREP + 5) = A; SL(s, SL_PREP + 6) = B; SL(s, SL_PREP + 7) = D0 / fmaxf(kMinVal, mu * mu * (1.0f + mu * mu)); SL(s, SL_PREP + 8) = mu;
              }
            }
          }
          Q::fence();
          float Mvr[6], Mvc[MC];
          mulM(sr, sc, Mvr, Mvc);
          float q2 = 0, q2r = 0, g1 = 0, g1r = 0;    // Gauss part: phi'(a) = g1 + a*q2 with g1 = s.(Ma - f_smooth)
#pragma unroll
          for (int k = 0; k < MC; k++) { g1 = fmaf(sc[k], Mac[k] - sm_c[k], g1); q2 = fmaf(sc[k], Mvc[k], q2); }
#pragma unroll
          for (int i = 0; i < 6; i++) { g1r = fmaf(sr[i], Mar[i] - sm_r[i], g1r); q2r = fmaf(sr[i], Mvr[i], q2r); }
          g1 = Q::sum(g1 + w0 * g1r); q2 = Q::sum(q2 + w0 * q2r);
          LM_TICK(7);
          // phi'(alpha), phi''(alpha) and `mag` = sum of |terms| of phi' (its float32 noise floor is ~1e-6*mag)
          auto line = [&](float alpha, float& d1, float& d2, float& mag) {
            float a1 = 0, a2 = 0, r1 = 0, r2 = 0, am = 0, rm = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) {
              const float x = fmaf(alpha, jv_r[i], jfr_r[i]);
              const float t = jv_r[i] * fminf(fmaxf(x * iR_r[i], -ff_r[i]), ff_r[i]);
              r1 += t; rm += fabsf(t);
              r2 = fmaf((fabsf(x) * iR_r[i] < ff_r[i]) ? iR_r[i] : 0.0f, jv_r[i] * jv_r[i], r2);
              if constexpr (ROOT_LIM) {
                const float jvl = lim_s_r[i] * jv_r[i];
                const float xl = fminf(fmaf(alpha, jvl, jlim_r[i]), 0.0f);       // lim_D_r = 0 when no limit is active
                const float tl = lim_D_r[i] * xl * jvl;
                r1 += tl; rm += fabsf(tl);
                r2 = fmaf((xl < 0.0f) ? lim_D_r[i] : 0.0f, jvl * jvl, r2);
              }
            }
#pragma unroll
            for (int k = 0; k < MC; k++) {
              const float x = fmaf(alpha, jv_c[k], jfr_c[k]);
              float t = jv_c[k] * fminf(fmaxf(x * iR_c[k], -ff_c[k]), ff_c[k]);
              a2 = fmaf((fabsf(x) * iR_c[k] < ff_c[k]) ? iR_c[k] : 0.0f, jv_c[k] * jv_c[k], a2);
              const float xl = fminf(fmaf(alpha, jvlim_c[k], jlim_c[k]), 0.0f);       // lim_D_c = 0 when no limit is active
              t = fmaf(lim_D_c[k] * xl, jvlim_c[k], t);
              a2 = fmaf((xl < 0.0f) ? lim_D_c[k] : 0.0f, jvlim_c[k] * jvlim_c[k], a2);
              a1 += t; am += fabsf(t);
#ifdef LM_LS_TRACE
              if (getenv("LM_ROWS")) printf("      lane %d row %d alpha %.6g t %.6g x %.5g iR %.4g f %.3g limD %.4g xl %.5g\n", c, k, alpha, t, x, iR_c[k], ff_c[k], lim_D_c[k], xl);
#endif
            }
            for (int s = 0; s < nslot; s++) {
              float Dj[6], fr[5], jar[6], jv[6];
              const int dim = (int)SL(s, SL_DIM);
              float c1 = 0, c2 = 0;
              if (PYR3(dim)) {
                const float D = SL(s, SL_D);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                  const float xv = SL(s, SL_JV + r), x = fminf(fmaf(alpha, xv, SL(s, SL_JAR + r)), 0.0f);
                  c1 = fmaf(D * x, xv, c1); c2 = fmaf((x < 0.0f) ? D : 0.0f, xv * xv, c2);
                }
              } else {
#ifdef LM_LS_CONE_DIRECT
#pragma unroll
                for (int j = 0; j < 6; j++) { jar[j] = SL(s, SL_JAR + j); jv[j] = SL(s, SL_JV + j); Dj[j] = SL(s, SL_D + j); }
#pragma unroll
                for (int j = 0; j < 5; j++) fr[j] = SL(s, SL_FR + j);
                cone_line(jar, jv, alpha, Dj, fr, SL(s, SL_MU), dim, c1, c2);
#else
                // from the prepared polynomials (cone_line is the same arithmetic on the rows themselves)
                (void)Dj; (void)fr; (void)jar; (void)jv;
                const float N0 = SL(s, SL_PREP + 0), Np = SL(s, SL_PREP + 1), UU0 = SL(s, SL_PREP + 2), UV0 = SL(s, SL_PREP + 3), VV = SL(s, SL_PREP + 4);
                const float mu = SL(s, SL_PREP + 8);
                const float N = fmaf(alpha, Np, N0), uv = fmaf(alpha, VV, UV0);
                const float T = sqrtf(fmaxf(fmaf(alpha, UV0 + uv, UU0), 0.0f));
                if (N >= mu * T || (T <= 0 && N >= 0)) {}
                else if (mu * N + T <= 0 || (T <= 0 && N < 0)) { const float B = SL(s, SL_PREP + 6); c1 = fmaf(alpha, B, SL(s, SL_PREP + 5)); c2 = B; }
                else {
                  const float Dm = SL(s, SL_PREP + 7);
                  const float Tp = uv / T, Tpp = VV / T - uv * uv / (T * T * T);
                  const float NmT = N - mu * T, NmTp = Np - mu * Tp;
                  c1 = Dm * NmT * NmTp;
                  c2 = Dm * (NmTp * NmTp - NmT * mu * Tpp);
                }
#endif
              }
#ifdef LM_LS_TRACE
              if (getenv("LM_ROWS")) printf("      lane %d slot %d alpha %.6g c1 %.6g c2 %.6g\n", c, s, alpha, c1, c2);
#endif
              if (PAIRS) { const float wgt = slot_weight(s); c1 *= wgt; c2 *= wgt; }      // mirror slots: half each
              a1 += c1; a2 += c2; am += fabsf(c1);
            }
            d1 = g1 + alpha * q2 + Q::sum(a1 + w0 * r1);
            d2 = q2 + Q::sum(a2 + w0 * r2);
            mag = fabsf(g1) + fabsf(alpha * q2) + Q::sum(am + w0 * rm);
          };
          // Root of the increasing phi'. It is piecewise smooth with very different slopes: saturating friction
          // rows give S-shapes, a stiff contact crossing its sticking sliver gives a near-jump.
          float d1, d2, mag, alpha = 0, lo = 0, hi = -1.0f, dlo, dhi = 0.0f;
          // phi'(0) = g . s = -dec and, along the Newton direction H s = -g, phi''(0) = s . H s = dec: no evaluation at 0
          // (LM_LS_EVAL0: rounds 1-4 evaluated the line at alpha = 0, for A/B builds)
#ifdef LM_LS_EVAL0
          line(0.0f, d1, d2, mag);
#else
          d1 = -dec; d2 = dec; mag = 0.0f; (void)mag;
#endif
#ifdef LM_LS_TRACE
          if (getenv("LM_SCAN")) { for (float aa = 1e-7f; aa < 2.0f; aa *= 3.0f) { float x1, x2, xm; line(aa, x1, x2, xm); if (c == 0) printf("    scan alpha %.3g d1 %.6g d2 %.6g\n", aa, x1, x2); } line(0.0f, d1, d2, mag); }
#endif
          bool ls_done = !(d1 < 0.0f && d2 > 0.0f);
          const float dref = fabsf(d1);
          dlo = d1;
          if (Q::kPoints > 1) {
            // ---- FOUR POINTS PER ROUND. With kRep = 4 the environment is replicated over the 4 quads of a 16-lane row
            // (same instruction stream, idle lanes otherwise) and every replica evaluates ONE of the four step lengths,
            // so a round costs one evaluation; with kRep = 1 (CPU emulator) the same lane evaluates all four.
            // Round 1 spreads the Newton step geometrically (the root of a step that activates constraints is often
            // 10-100x shorter than the Newton step); later rounds put Newton from both bracket ends, the secant and the
            // midpoint inside the bracket.
            float d2lo = d2, d2hi = 0.0f, best_abs = 3.0e38f;
            float cand[4];
            const float aN = ls_done ? 0.0f : -d1 / d2;
            cand[0] = aN; cand[1] = P.ls_grid[0] * aN; cand[2] = P.ls_grid[1] * aN; cand[3] = P.ls_grid[2] * aN;
            for (int round = 0; round < ((P.ablate & 8) ? 0 : (P.ls_iters + 1) / 2); round++) {
              if (!Q::any(!ls_done)) break;
              if (!ls_done) {
                float cd1[4], cd2[4], cmg[4];
                if (Q::kRep == 1) {
#pragma unroll
                  for (int j = 0; j < 4; j++) line(cand[j], cd1[j], cd2[j], cmg[j]);
                } else {
                  const int r = Q::rep() & 3;       // (sixteen replicas: four groups evaluate the same four points)
                  const float mine = (r == 0) ? cand[0] : ((r == 1) ? cand[1] : ((r == 2) ? cand[2] : cand[3]));
                  float x1, x2, xm;
                  line(mine, x1, x2, xm);
#pragma unroll
                  for (int j = 0; j < 4; j++) { cd1[j] = Q::rep_bcast(x1, j); cd2[j] = Q::rep_bcast(x2, j); cmg[j] = Q::rep_bcast(xm, j); }
                }
                if (c == 0) cnt.ls_evals++;
#ifdef LM_LS_TRACE
                if (c == 0) for (int j = 0; j < 4; j++) printf("  ls round %d alpha %.9g d1 %.6g d2 %.6g lo %.9g hi %.9g dref %.4g\n", round, cand[j], cd1[j], cd2[j], lo, hi, dref);
#endif
#pragma unroll
                for (int j = 0; j < 4; j++) {
                  const float ad = fabsf(cd1[j]);
                  if (ad < fmaxf(P.ls_tol * dref, P.ls_noise * cmg[j]) && ad < best_abs) { best_abs = ad; alpha = cand[j]; ls_done = true; }
                }
                if (!ls_done) {
#pragma unroll
                  for (int j = 0; j < 4; j++) {
                    if (cd1[j] < 0.0f) { if (cand[j] > lo) { lo = cand[j]; dlo = cd1[j]; d2lo = cd2[j]; } }
                    else if (hi < 0.0f || cand[j] < hi) { hi = cand[j]; dhi = cd1[j]; d2hi = cd2[j]; }
                  }
                  if (hi < 0.0f) {            // still descending at the longest step tried: look further out
                    cand[0] = 2.0f * lo; cand[1] = 4.0f * lo; cand[2] = 8.0f * lo; cand[3] = 16.0f * lo;
                    alpha = lo;
                  } else {
                    const float w = hi - lo;
                    alpha = (lo * dhi - hi * dlo) / (dhi - dlo);       // secant point: the answer if the rounds run out
                    if (!(alpha > lo && alpha < hi)) alpha = 0.5f * (lo + hi);
                    if (w <= 1e-4f * hi) ls_done = true;
                    else {
                      const float in_lo = lo + 0.01f * w, in_hi = hi - 0.01f * w;
                      float g0 = lo - dlo / d2lo, g1 = hi - dhi / d2hi;
                      if (!(g0 > in_lo && g0 < in_hi)) g0 = lo + 0.25f * w;
                      if (!(g1 > in_lo && g1 < in_hi)) g1 = lo + 0.75f * w;
                      cand[0] = g0; cand[1] = g1; cand[2] = fminf(fmaxf(alpha, in_lo), in_hi); cand[3] = 0.5f * (lo + hi);
                    }
                  }
                }
              }
            }
          } else {
          // ---- ONE POINT AT A TIME (full waves, no idle lanes): Newton from the current point; once the root is
          // bracketed: Newton if it lands inside the bracket, else the secant through the bracket ends, and a bisection
          // whenever the previous step failed to halve the bracket (that is what finds the slivers).
          float w_prev = 3.0e38f;
          if (!ls_done) alpha = -d1 / d2;
          for (int lsi = 0; lsi < ((P.ablate & 8) ? 0 : P.ls_iters); lsi++) {
            if (!Q::any(!ls_done)) break;
            if (!ls_done) {
              line(alpha, d1, d2, mag);
              if (c == 0) cnt.ls_evals++;
#ifdef LM_LS_TRACE
              if (c == 0) printf("  ls %d alpha %.9g d1 %.6g d2 %.6g mag %.4g lo %.9g hi %.9g dref %.4g\n", lsi, alpha, d1, d2, mag, lo, hi, dref);
#endif
              if (fabsf(d1) < fmaxf(P.ls_tol * dref, P.ls_noise * mag)) ls_done = true;
              else {
                if (d1 < 0.0f) { lo = alpha; dlo = d1; } else { hi = alpha; dhi = d1; }
                float next = alpha - d1 / d2;
                if (hi > 0.0f) {
                  const float w = hi - lo, in_lo = lo + 0.01f * w, in_hi = hi - 0.01f * w;
                  if (!(next > in_lo && next < in_hi)) next = (lo * dhi - hi * dlo) / (dhi - dlo);
                  if (!(next > in_lo && next < in_hi) || w > 0.5f * w_prev) next = 0.5f * (lo + hi);
                  if (w <= 1e-4f * hi) ls_done = true;
                  w_prev = w;
                } else if (next <= lo) next = 2.0f * alpha;
                alpha = next;
              }
            }
          }
          }
          if (c == 0 && !ls_done) cnt.ls_capped++;
          LM_TICK(8);
          if (kMaIncr) {
#pragma unroll
            for (int i = 0; i < 6; i++) Mar[i] = fmaf(alpha, Mvr[i], Mar[i]);
#pragma unroll
            for (int k = 0; k < MC; k++) Mac[k] = fmaf(alpha, Mvc[k], Mac[k]);
          }
          if (kJarIncr && nslot > 0) {
            // every replica moves the rows of the slots it owns in the gradient (the same words whoever writes them)
            for (int s = s_first; s < nslot; s += s_step) {
              if (PYR3((int)SL(s, SL_DIM))) {
#pragma unroll
                for (int r = 0; r < 4; r++) SL(s, SL_JAR + r) = fmaf(alpha, SL(s, SL_JV + r), SL(s, SL_JAR + r));
              } else {
#pragma unroll
                for (int j = 0; j < 6; j++) SL(s, SL_JAR + j) = fmaf(alpha, SL(s, SL_JV + j), SL(s, SL_JAR + j));
              }
            }
          }
#pragma unroll
          for (int i = 0; i < 6; i++) ar[i] = fmaf(alpha, sr[i], ar[i]);
#pragma unroll
          for (int k = 0; k < MC; k++) ac[k] = fmaf(alpha, sc[k], ac[k]);
          if (alpha == 0.0f) done = true;
        }
      }
    }
  }
  if (want_grf) {
    // foot forces (reference base.py:623-631,667-679): contact-frame force of the FIRST contact of each force group,
    // from the residuals of the last gradient evaluation (= the solution)
    int seen = 0;
    for (int s = 0; s < nslot; s++) {
      const int gq = (int)SL(s, SL_GRF);
      if (gq < 0 || ((seen >> gq) & 1)) continue;
      seen |= 1 << gq;
      const int dim = (int)SL(s, SL_DIM);
      float f[6];
      if (PYR3(dim)) {
        float x[4], dummy = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) x[r] = SL(s, SL_JAR + r);
        pyr_force(x, SL(s, SL_D), SL(s, SL_MU), f, dummy);
      } else {
        float jar[6], Dj[6], fr[5];
#pragma unroll
        for (int j = 0; j < 6; j++) { jar[j] = SL(s, SL_JAR + j); Dj[j] = SL(s, SL_D + j); }
#pragma unroll
        for (int j = 0; j < 5; j++) fr[j] = SL(s, SL_FR + j);
        ConeEval e = cone_eval<true>(jar, Dj, fr, SL(s, SL_MU), dim);
#pragma unroll
        for (int j = 0; j < 3; j++) f[j] = (j < dim) ? e.f[j] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 3; j++) {
        if (gq == 0) cnt.grf[0][j] += f[j];
        else if (MC < 6 || gq == 1) cnt.grf[1][j] += f[j];
        else if (gq == 2) cnt.grf[2][j] += f[j];
        else cnt.grf[3][j] += f[j];
      }
    }
  }
  cnt.solver_iters += (c == 0) ? iters : 0;
  if (c == 0 && iters > cnt.it_max) cnt.it_max = iters;

  if (dbg) {
    const int nv = P.nv;
#pragma unroll
    for (int k = 0; k < MC; k++) if (k < nl) {
      int d = (int)LK(k, LM_D_DOF);
      dbg->bias[d] = bias_c[k]; dbg->smooth[d] = sm_c[k]; dbg->qacc_smooth[d] = a0c[k]; dbg->qacc[d] = ac[k]; dbg->qfrc_constraint[d] = qf_c[k];
#pragma unroll
      for (int j = 0; j <= k; j++) { int e = (int)LK(j, LM_D_DOF); dbg->M[d * nv + e] = dbg->M[e * nv + d] = LMEM(LMm::kMcc + tri(k, j)); }
#pragma unroll
      for (int r = 0; r < 6; r++) { int e = (int)RD(r, LM_D_DOF); dbg->M[d * nv + e] = dbg->M[e * nv + d] = LMEM(LMm::kMcr + k * 6 + r); }
    }
    if (c == 0) {
#pragma unroll
      for (int i = 0; i < 6; i++) {
        int d = (int)RD(i, LM_D_DOF);
        dbg->bias[d] = bias_r[i]; dbg->smooth[d] = sm_r[i]; dbg->qacc_smooth[d] = a0r[i]; dbg->qacc[d] = ar[i]; dbg->qfrc_constraint[d] = qf_r[i];
#pragma unroll
        for (int j = 0; j <= i; j++) { int e = (int)RD(j, LM_D_DOF); dbg->M[d * nv + e] = dbg->M[e * nv + d] = LMEM(LMm::kMrr + tri(i, j)); }
      }
    }
  }

  LM_TICK(10);      // lockstep wait: this environment converged, others of the wave still iterate
  oz = LM_OPAQUE_ZERO();
  // ================= integrate: semi-implicit Euler, joint damping implicit =================
  // (M + h diag(damping)) qacc' = qfrc_smooth + qfrc_constraint ; qvel += h qacc' ; qpos += h qvel
#pragma unroll
  for (int i = 0; i < 6; i++) war[i] = ar[i];
#pragma unroll
  for (int k = 0; k < MC; k++) wac[k] = ac[k];
  if (EULER) {
    float Hcc[MC * (MC + 1) / 2], Hcr[MC][6], Hrep[21], Lr[21], zero21[21];
#pragma unroll
    for (int i = 0; i < MC * (MC + 1) / 2; i++) Hcc[i] = LMEM(LMm::kMcc + i);
#pragma unroll
    for (int i = 0; i < 21; i++) { Hrep[i] = LMEM(LMm::kMrr + i); zero21[i] = 0; }
#pragma unroll
    for (int k = 0; k < MC; k++) {
#pragma unroll
      for (int r = 0; r < 6; r++) Hcr[k][r] = LMEM(LMm::kMcr + k * 6 + r);
      if (k < nl) Hcc[tri(k, k)] += P.h * DAMP_C(k);
    }
#pragma unroll
    for (int i = 0; i < 6; i++) Hrep[tri(i, i)] += P.h * DAMP_R(i);
    arrow_factor<Q, MC>(Hcc, Hcr, Hrep, zero21, Lr);
    float xr[6], xc[MC];
#pragma unroll
    for (int i = 0; i < 6; i++) xr[i] = sm_r[i] + qf_r[i];
#pragma unroll
    for (int k = 0; k < MC; k++) xc[k] = sm_c[k] + qf_c[k];
    arrow_solve<Q, MC>(Hcc, Hcr, Lr, xc, xr);
    if (anydup) tie_shared_dof<Q, MC>(Hcc, Hcr, Lr, xc, xr, duprole);
#pragma unroll
    for (int i = 0; i < 6; i++) { vr[i] = fmaf(P.h, xr[i], vr[i]); qr[i] = fmaf(P.h, vr[i], qr[i]); }
#pragma unroll
    for (int k = 0; k < MC; k++) if (k < nl) { vc[k] = fmaf(P.h, xc[k], vc[k]); qc[k] = fmaf(P.h, vc[k], qc[k]); }
  }
  LM_TICK(9);
#undef PYR3
#undef DAMP_R
#undef STIFF_R
#undef FLOSS_R
#undef DAMP_C
#undef DUPK
#undef STIFF_C
#undef FLOSS_C
#undef RD
#undef CH
#undef LK
#undef LX
#undef GE
#undef GP
#undef CU
#undef SL
#undef PEER
#undef LPE
#undef LMEM
}

// one physics substep with the model's integrator. RK4: classical 4-stage scheme on (qpos, qvel), every stage a full
// forward pass incl. collision detection and constraint solve, no implicit damping (MuJoCo mj_RungeKutta semantics).
template <class Q, int MC, int NS, bool RK4, int CONE = -1, int NM = 0, int DR = 0, int PM = 0>
LM_DEV void substep(const float* cm, int c, const Params& P, float* qr, float* vr, float* qc, float* vc,
                    float* war, float* wac, const float* actr, const float* actc, LM_LMEM_T* lmem, int ls,
                    Counters& cnt, const Debug* dbg, const float* mt = nullptr, const DofPrm<MC>* dp = nullptr,
                    bool want_grf = false, float* pair_slack = nullptr) {
  static_assert(!(RK4 && NM > 0), "muscle activations are only advanced by the Euler integrator");
  if (!RK4) { forward<Q, MC, NS, true, CONE, NM, DR, PM>(cm, c, P, qr, vr, qc, vc, war, wac, actr, actc, lmem, ls, cnt, dbg, mt, dp, want_grf, pair_slack); return; }
  float q0r[6], v0r[6], q0c[MC], v0c[MC], dqr[6], dvr[6], dqc[MC], dvc[MC];
#pragma unroll
  for (int i = 0; i < 6; i++) { q0r[i] = qr[i]; v0r[i] = vr[i]; dqr[i] = 0; dvr[i] = 0; }
#pragma unroll
  for (int k = 0; k < MC; k++) { q0c[k] = qc[k]; v0c[k] = vc[k]; dqc[k] = 0; dvc[k] = 0; }
#pragma nounroll
  for (int st = 0; st < 4; st++) {
    forward<Q, MC, NS, false, CONE, 0, DR, PM>(cm, c, P, qr, vr, qc, vc, war, wac, actr, actc, lmem, ls, cnt, (st == 0) ? dbg : nullptr, nullptr, dp,
                                           want_grf && st == 3, pair_slack);   // the engine's data hold the 4th stage when mj_step returns
    const float b = (st == 0 || st == 3) ? (1.0f / 6.0f) : (1.0f / 3.0f);
    const float a = (st == 2) ? 1.0f : 0.5f;            // tableau entry A[st+1][st]
#pragma unroll
    for (int i = 0; i < 6; i++) {
      dqr[i] = fmaf(b, vr[i], dqr[i]); dvr[i] = fmaf(b, war[i], dvr[i]);
      if (st < 3) { qr[i] = fmaf(P.h * a, vr[i], q0r[i]); vr[i] = fmaf(P.h * a, war[i], v0r[i]); }
    }
#pragma unroll
    for (int k = 0; k < MC; k++) {
      dqc[k] = fmaf(b, vc[k], dqc[k]); dvc[k] = fmaf(b, wac[k], dvc[k]);
      if (st < 3) { qc[k] = fmaf(P.h * a, vc[k], q0c[k]); vc[k] = fmaf(P.h * a, wac[k], v0c[k]); }
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) { qr[i] = fmaf(P.h, dqr[i], q0r[i]); vr[i] = fmaf(P.h, dvr[i], v0r[i]); }
#pragma unroll
  for (int k = 0; k < MC; k++) { qc[k] = fmaf(P.h, dqc[k], q0c[k]); vc[k] = fmaf(P.h, dvc[k], v0c[k]); }
  if (PM != 0 && pair_slack) { pair_slack[2] = pair_slack[1]; pair_slack[1] = 0.0f; }      // the speed memory of the detection's travel bound (forward)
}

}  // namespace lm
