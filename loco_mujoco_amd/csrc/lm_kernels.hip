// lm_kernels.hip — the C-ABI (include/locohip.h) of the batched LocoEnv.step(): models, batches, state transfer and the
// launches of the step kernels. The kernels themselves live in lm_step.h / lm_core.h and are compiled per family in
// lm_family.hip (one object per family and part, so the library builds in parallel).
#include "lm_step.h"
#include "lm_compile.h"
#include <memory>

using lmk::KArgs; using lmk::Task; using lmk::DevStats; using lmk::LaunchCtx;

namespace {
thread_local std::string g_err;
thread_local const char* g_launch_err = nullptr;
int fail(const std::string& m) { g_err = m; return 1; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

}  // namespace

// ================================================================================================================
// C-ABI
// ================================================================================================================
struct lm_model {
  int device;
  float* d_cm;
  float* d_gt;               // geom table: full records of the geoms with a collider
  float* d_mt;               // muscle table (models with muscles)
  float* d_gpt;              // geom-pair table of the self-collision path
  int n_gpt_floats;          // its size (0: the model has no self-collision pairs)
  float* d_meshv;            // hull vertices of the mesh colliders
  float* d_meshn;            // their neighbour lists (hull vertex graph)
  float* d_bpt;              // body-pair table of the self-collision mid phase
  float* d_meshadj;          // adjacency blocks of the hull vertices (convex-pair collider)
  std::vector<float> nominal;  // [3][nv] damping | stiffness | frictionloss of the model
  lm::Params P; Task T;
  int nroot;
  std::vector<int> root_dofs;
  bool root_limited;         // a root dof with an active-able limit (lm_batch_set_replay)
  bool root_xyz;             // the root's first three dofs are slides along +x, +y, +z in a root frame that is the world's (lm_core.h ROOT_XYZ)
};

struct lm_batch {
  lm_model* m;
  int N;
  float *qpos, *qvel, *warm, *goal, *action, *obs, *reward, *table;
  float* act;                // muscle activations [na][N]
  float* dofprm;             // per-environment joint parameters [3][nv][N] (allocated by lm_set_dof_params)
  float* drspec;             // redraw rules [3][nv][3] (lm_set_dof_randomization)
  unsigned char* done;
  unsigned char* flags;      // validity flags of the last control step per environment (lm_get_flags)
  int* ep_step; unsigned* ep_count;
  DevStats* stats;
  int table_rows; unsigned long long seed; long long env_offset; int auto_reset, horizon; unsigned step_index;
  int epb, nblocks;
  unsigned long long* timers;
  unsigned long long* tline;     // LM_TIMERS builds: time line of the last launch (lm_step.h KArgs::tline)
  hipStream_t stream;
  lm_stats acc;            // host-side accumulation (double)
  hipEvent_t ev0, ev1;
  hipEvent_t ev_ext;       // orders the library's stream behind a launch on a caller's stream (lm_step_device)
  float *vrec, *vgt, *vgpt; int* var; int nvar, gpt_floats, var_rows; bool dofprm_of_variants;   // model variants (lm_set_model_variants, lm_set_variant_rows)
  // the model compiler on the device (lm_set_model_compiler, lm_compile.hip): its program, per environment the restart flag, the draw
  // counter and the values drawn; the variant tables above then hold ONE slot per environment
  int* mc_ib; double* mc_db; int mc_ndraw; unsigned char *vdirty, *mc_mask; unsigned* vgen; double* vdraws; unsigned long long mc_seed;
  float* scr; int* scr_idx; size_t scr_cap;   // staging for masked uploads (rows of the masked environments only)
  // speculate / replay (lm_step.h): list of the environments whose control step left the regular kernel's capacity, its control
  // words, the fused step at which each left; `replay` = 0 switches the mechanism off (contacts beyond the slots are then dropped)
  int *replay_list, *replay_ctl, *stall; int replay;
  int* premark;                              // prediction (lm_step.h KArgs::premark): environments that start their next control step in the replay kernel
  float *hq, *hv, *hw; int* hsub;            // resume (lm_step.h KArgs::hq): substep-start states of the control steps that leave the regular kernel
  unsigned char* replay_mark;
  // the replay kernel's pollers run beside the regular launch on `stream2`, forked from / joined into the launch stream with the two
  // events; `h_hint` (pinned) receives the number of abandoned control steps of a completed launch: how many pollers the next one gets
  int stat_pre_off, nstat; int* h_hint; int hint, hint_seen; int epoch;
  hipStream_t stream2; hipEvent_t ev_fork, ev_join, ev_done[2];
  float* slack;              // detection slack + speed memory of the self-collision pass, [3][4][N] (lm_step.h KArgs::slack)
  // the float64 host surface (lm_step_pinned): pinned action staging, the pinned ring of result sets [obs f64 | reward f64 | done]
  float* h_act; unsigned char* h_out64[LM_PINNED_SLOTS]; int* d_perm; size_t out64_bytes;
  int* env_map; int n_active;    // active list (lm_batch_set_active): device copy of the environment ids, their number (N: all)
  float* mprc; int mprc_pairs;   // warm-start cache of the convex collider, [N][mprc_pairs][lm::kMprCacheFloats] (lm_step.h KArgs::mprc), or null
};
// which kernel family serves a model (lm_family.hip): the quadruped family gets a specialised step kernel
// <3 links, 6 slots, Euler, elliptic, self-collisions>; the humanoid families (five- and six-link chains) are compiled for
// condim-3 pyramids only (T.all_pyr3, checked when the model is created): the elliptic code compiles out and the contact
// slots are compact. Everything else: generic kernels, cone read at run time, plain layout only.
// A/B switches of the probe builds (tools/probes: `make EXTRA=-DLM_PROBES ...`). The shipped library reads NO environment variable:
// tests/test_abi_exports.py checks that `getenv` is not among its undefined symbols.
#ifdef LM_PROBES
#define LM_PROBE_ENV(name) getenv(name)
#else
#define LM_PROBE_ENV(name) ((const char*)nullptr)
#endif

static int family_of(const lm_batch* b) {
  const Task& T = b->m->T;
  const bool big = T.max_links > 3, six = T.max_links > 5, rk4 = b->m->P.integrator == LM_INT_RK4, few = T.max_contacts <= 4;
  static const bool generic = LM_PROBE_ENV("LM_GENERIC_KERNELS") != nullptr;      // A/B: run-time cone for the humanoids
  const bool pyr3 = T.all_pyr3 && !generic;
  if (six) return (!rk4 && T.na == 0 && pyr3) ? 7 : -1;      // (with or without self-collision tables: its regular kernels detect, its replay kernel collides)
  // five-link humanoids whose lowering carries self-collision tables (bone hulls, link meshes, cylinders): the pair families
  if (big && T.npair > 0 && pyr3) return rk4 ? (T.na == 0 ? 8 : 6) : (T.na == 0 ? 9 : 10);
  // (the quadruped family's Hessian takes the root's translation columns as unit rows: lm_core.h ROOT_XYZ; another root -> generic kernels)
  if (!big && !rk4 && T.na == 0 && b->m->P.cone == LM_CONE_ELLIPTIC && b->m->root_xyz) return 0;
  // (families 1 / 3 — four contact slots per chain — are gone: since round 4 every five-link humanoid without muscles runs in the
  // eight-slot families, where a fifth contact on a leg does not abandon the control step)
  if (big && rk4 && T.na == 0 && pyr3) return 2;
  if (big && !rk4 && T.na == 0 && pyr3) return 4;
  if (big && !rk4 && T.na > 0 && few && pyr3) return 5;
  return 6;
}

static bool family_has_replicas(const lm_batch* b) { return family_of(b) != 6; }
static bool family_has_pairs(int fam) { return fam == 0 || fam == 7 || fam == 8 || fam == 9 || fam == 10; }

// (Round 5, tried and dropped: a one-workgroup GATE kernel in front of the regular launch that waits until the launch's pollers are
// resident, and a higher priority for their stream. Launched first on their own stream the pollers lose the race for the chip in 7
// launches of 10 and start when the first regular workgroups retire, 4.4 ms into a HumanoidTorque launch; with the gate they are
// resident at once — and every poller then keeps one regular workgroup waiting for those 4.4 ms instead (the kernels' 512 registers
// allow one wave per SIMD, and 4096 environments are exactly one wave per SIMD): the step time is the same, 16.0 ms either way.
// profiles/r5_notes.md §4.)
template <bool FWD>
static void launch_variant(lm_batch* b, const KArgs& a) {
  static const bool no_replicas = LM_PROBE_ENV("LM_NO_REPLICAS") != nullptr;                  // A/B switch
  static const lmk::family_fn table[lmk::LMK_NFAMILY][3] = {
      {lmk::launch_f0p0, lmk::launch_f0p1, lmk::launch_f0p2}, {nullptr, nullptr, nullptr},
      {lmk::launch_f2p0, lmk::launch_f2p1, lmk::launch_f2p2}, {nullptr, nullptr, nullptr},
      {lmk::launch_f4p0, lmk::launch_f4p1, lmk::launch_f4p2}, {lmk::launch_f5p0, lmk::launch_f5p1, lmk::launch_f5p2},
      {lmk::launch_f6p0, lmk::launch_f6p1, lmk::launch_f6p2}, {lmk::launch_f7p0, lmk::launch_f7p1, lmk::launch_f7p2},
      {lmk::launch_f8p0, lmk::launch_f8p1, lmk::launch_f8p2}, {lmk::launch_f9p0, lmk::launch_f9p1, lmk::launch_f9p2},
      {lmk::launch_f10p0, lmk::launch_f10p1, lmk::launch_f10p2}};
  const int fam = family_of(b);
  if (fam < 0) { g_launch_err = "chains of six links are compiled for Euler, condim-3 pyramids, no muscles only"; return; }
  const LaunchCtx L = {b->stream, b->n_active, b->epb};
  if (b->n_active <= 0) return;            // an empty active list: nothing to run
  if (fam == 6) {
    if (b->m->T.na > 0) { g_launch_err = "muscle models need the <5 links, <=4 contacts per chain, Euler> family"; return; }
    if (b->dofprm) { g_launch_err = "per-environment joint parameters are not compiled for this model family"; return; }
    table[6][b->m->P.integrator == LM_INT_RK4 ? 1 : 0](L, a, FWD ? lmk::LMK_FWD : lmk::LMK_REP1);
    return;
  }
  // the layout: replicated (4 quads per environment, workgroups of <= 4 environments), per-environment joint
  // parameters (domain randomisation), fused rollouts (replicated layout only), or plain
  int kind;
  const bool rep = b->epb <= 4 && !no_replicas;
  if (FWD) kind = lmk::LMK_FWD;
  else if (a.nfused > 1) kind = b->nvar > 0 ? lmk::LMK_FUSED_DRV : (b->dofprm ? lmk::LMK_FUSED_DR : lmk::LMK_FUSED);
  else if (b->nvar > 0) kind = rep ? lmk::LMK_DRV_REP4 : lmk::LMK_DRV_REP1;
  else if (b->dofprm) kind = rep ? lmk::LMK_DR_REP4 : lmk::LMK_DR_REP1;
  else kind = rep ? lmk::LMK_REP4 : lmk::LMK_REP1;
  const int big = b->nvar > 0 ? lmk::LMK_BIG_DRV : (b->dofprm ? lmk::LMK_BIG_DR : lmk::LMK_BIG);
  KArgs r = a;
  r.reg_grid = (b->n_active + b->epb - 1) / b->epb; r.epoch = b->epoch; r.host_hint = b->h_hint;
  const bool replay = !FWD && a.replay_list;
  bool pollers = false;
  if (replay) {
    // How many control steps did recent launches abandon? (h_hint: pinned host memory, written by the drain pass of a launch that
    // has completed by now — a hint, read without waiting for anything.) Launches that abandon some get pollers: replay workgroups on the second stream, launched
    // BEFORE the regular kernel, that take the abandoned environments over while the launch is still running (lm_step.h). A batch
    // whose robots stay inside the regular kernel (the quadruped's bench rollout) launches none.
    // (a rollout queues hundreds of launches before the first has run: no news = no change; news = the latest count, decaying slowly)
    const int last = b->h_hint[0], seen = b->h_hint[1];
    if (seen != b->hint_seen) { b->hint_seen = seen; b->hint = last > b->hint ? last : (b->hint > 0 ? b->hint - 1 : 0); }
    // how many: the abandoned steps of a launch scatter around the recent count like a Poisson variable (HumanoidTorque.run: 12.6 +- 3.5
    // per launch) — an entry without a poller of its own waits for one to finish a whole hard control step. pollers = kPollMul x
    // recent count + kPollAdd, at most kPollCap (round 5: profiles/r5_notes.md §4)
    static int poll_mul = lmk::kPollMul, poll_add = lmk::kPollAdd, poll_cap = lmk::kPollCap;
    static const bool poll_env = [] { if (const char* v = LM_PROBE_ENV("LM_POLLERS")) sscanf(v, "%d,%d,%d", &poll_mul, &poll_add, &poll_cap); return true; }();
    (void)poll_env;
    int want = a.replay_all ? lmk::kReplayGrid : (b->hint > 0 ? (poll_mul * b->hint) / 2 + poll_add : 0);
    if (want > poll_cap && !a.replay_all) want = poll_cap;
    if (want > lmk::kReplayGrid) want = lmk::kReplayGrid;
    if (b->replay >= 3) want = 0;
    if (b->h_hint[2] > 0) want = 0;        // a poller timed out once: they do not overlap with the regular kernel here (serialised kernels) — drain pass only from now on
    if (want > 0) {
      KArgs p = r;
      p.drain = 0; p.stats_off = b->stat_pre_off;
      // not further ahead than one launch: the pollers start once the launch BEFORE the previous one is complete (they are then resident
      // while the previous launch tails off and wait for its drain pass on the device). Without this a host that queues hundreds of
      // launches ahead of the device would start them long before their launch: they would wait out their time-out and leave
      if (b->epoch >= 2 && hipStreamWaitEvent(b->stream2, b->ev_done[b->epoch & 1], 0) != hipSuccess) { g_launch_err = "stream wait failed"; return; }
      const LaunchCtx L2 = {b->stream2, b->N, want};
      if (!table[fam][0](L2, p, big) && !table[fam][1](L2, p, big) && !table[fam][2](L2, p, big)) { g_launch_err = "no replay kernel in the family"; return; }
      if (hipEventRecord(b->ev_join, b->stream2) != hipSuccess) { g_launch_err = "stream join failed"; return; }
      pollers = true;

    }
  }
  // A failure from here on leaves pollers in flight that wait for a regular launch which will not come: they give up after their
  // time-out. Wait for them, and put the control words back to "no launch in progress" (the epoch did not advance: nothing drained),
  // so that the batch can be launched again or destroyed safely.
  auto bail = [&](const char* msg) {
    g_launch_err = msg;
    if (pollers) {
      (void)hipStreamSynchronize(b->stream2);
      (void)hipStreamSynchronize(b->stream);
      (void)hipMemset(b->replay_ctl, 0, sizeof(int) * 4);
    }
  };
  if (!table[fam][0](L, r, kind) && !table[fam][1](L, r, kind) && !table[fam][2](L, r, kind)) { bail("no kernel of this kind in the family"); return; }
  if (replay) {
    // the drain pass, behind the regular launch AND the pollers: whatever is still listed; resets the control words. An empty
    // list costs a few microseconds (its workgroups read a word and leave)
    if (pollers && hipStreamWaitEvent(b->stream, b->ev_join, 0) != hipSuccess) { bail("stream join failed"); return; }
    r.drain = 1; r.stats_off = 0;
    const LaunchCtx L3 = {b->stream, b->N, lmk::kReplayGrid};
    if (!table[fam][0](L3, r, big) && !table[fam][1](L3, r, big) && !table[fam][2](L3, r, big)) { bail("no replay kernel in the family"); return; }
    if (hipEventRecord(b->ev_done[b->epoch & 1], b->stream) != hipSuccess) { bail("event record failed"); return; }
    b->epoch++;
  }
}

extern "C" {

const char* lm_last_error(void) { return g_err.c_str(); }

int lm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void lm_model_destroy(lm_model* m);

int lm_model_create(const double* cmod, size_t n, int device, lm_model** out) {
  if (!cmod || n < LM_HEADER_SIZE + LM_CM_SIZE + LM_GT_SIZE) return fail("chain model too short");
  if ((unsigned)cmod[LM_H_MAGIC] != (unsigned)LM_LMC_MAGIC) return fail("bad chain-model magic");
  if ((int)cmod[LM_H_CM_SIZE] != LM_CM_SIZE || (int)cmod[LM_H_GT_SIZE] != LM_GT_SIZE) return fail("chain-model table size mismatch (regenerate include/lm_layout.h)");
  if ((int)cmod[LM_H_MAXLINKS] > LM_MAXC) return fail("chains longer than 6 links are not supported");
  if ((int)cmod[LM_HEADER_SIZE + LM_R_NDOF] != 6) return fail("root body must have 6 dofs");
  const int n_muscle = (int)cmod[LM_H_NMUSCLE];
  if (n_muscle < 0 || n_muscle > LM_MT_MAXMUS) return fail("bad muscle count");
  if (n_muscle > 0 && n < (size_t)(LM_HEADER_SIZE + LM_CM_SIZE + LM_GT_SIZE + LM_MT_SIZE)) return fail("chain model lacks the muscle table");
  if (n_muscle > 0 && (int)cmod[LM_H_INTEGRATOR] != LM_INT_EULER) return fail("muscles need the Euler integrator");
  HIPCHK(hipSetDevice(device));
  std::unique_ptr<lm_model, void (*)(lm_model*)> guard(new lm_model(), lm_model_destroy);      // freed on every error path
  lm_model* m = guard.get();
  m->device = device;
  std::vector<float> cm(LM_CM_SIZE);
  for (int i = 0; i < LM_CM_SIZE; i++) cm[i] = (float)cmod[LM_HEADER_SIZE + i];
  if (LM_PROBE_ENV("LM_NO_PAIRS")) for (int c = 0; c < LM_NCHAIN; c++) cm[LM_CM_CHAINS + LM_C_NLPAIR * LM_NCHAIN + c] = 0.0f;      // A/B: self-collision broad phase off
  // (limited root joints: limit rows in the muscle families, the run-time-cone kernels and every family's replay kernel; the other
  // regular kernels hand a control step with a root dof beyond its range to the replay kernel — lm_core.h ROOT_LIM)
  m->root_limited = false;
  for (int i = 0; i < 6; i++) if (cm[LM_R_DOFS + i * LM_D_SIZE + LM_D_LIMITED] != 0.0f) m->root_limited = true;
  m->root_xyz = cm[LM_R_NDOF] == 6.0f;
  for (int i = 0; i < 9; i++) if (cm[LM_R_R0 + i] != ((i % 4 == 0) ? 1.0f : 0.0f)) m->root_xyz = false;
  for (int i = 0; i < 3; i++) {
    const float* d = cm.data() + LM_R_DOFS + i * LM_D_SIZE;
    if (d[LM_D_TYPE] != 0.0f) m->root_xyz = false;                       // (0 = slide: mjcf.JNT_SLIDE)
    for (int k = 0; k < 3; k++) if (d[LM_D_AX + k] != ((k == i) ? 1.0f : 0.0f)) m->root_xyz = false;
  }
  HIPCHK(hipMalloc(&m->d_cm, sizeof(float) * LM_CM_SIZE));
  HIPCHK(hipMemcpy(m->d_cm, cm.data(), sizeof(float) * LM_CM_SIZE, hipMemcpyHostToDevice));
  {
    std::vector<float> gt(LM_GT_SIZE);
    for (int i = 0; i < LM_GT_SIZE; i++) gt[i] = (float)cmod[LM_HEADER_SIZE + LM_CM_SIZE + i];
    HIPCHK(hipMalloc(&m->d_gt, sizeof(float) * LM_GT_SIZE));
    HIPCHK(hipMemcpy(m->d_gt, gt.data(), sizeof(float) * LM_GT_SIZE, hipMemcpyHostToDevice));
  }
  m->d_mt = nullptr;
  if (n_muscle > 0) {
    std::vector<float> mt(LM_MT_SIZE);
    for (int i = 0; i < LM_MT_SIZE; i++) mt[i] = (float)cmod[LM_HEADER_SIZE + LM_CM_SIZE + LM_GT_SIZE + i];
    for (int c = 0; c < LM_NCHAIN; c++) if ((int)mt[LM_NCHAIN + c] > LM_MAXMUS) { return fail("too many muscles on one chain"); }
    HIPCHK(hipMalloc(&m->d_mt, sizeof(float) * LM_MT_SIZE));
    HIPCHK(hipMemcpy(m->d_mt, mt.data(), sizeof(float) * LM_MT_SIZE, hipMemcpyHostToDevice));
  }
  Task& T = m->T;
  T.na = n_muscle;
  {
    const int nv = (int)cmod[LM_H_NV];
    m->nominal.assign((size_t)3 * nv, 0.0f);
    auto put = [&](const float* blk, int stride) {
      const int d = (int)blk[LM_D_DOF * stride];
      if (d < 0 || d >= nv) return;
      m->nominal[d] = blk[LM_D_DAMP * stride]; m->nominal[nv + d] = blk[LM_D_STIFF * stride]; m->nominal[2 * nv + d] = blk[LM_D_FLOSS * stride];
    };
    for (int i = 0; i < 6; i++) put(cm.data() + LM_R_DOFS + i * LM_D_SIZE, 1);
    for (int c = 0; c < LM_NCHAIN; c++) {
      const int nl = (int)cm[LM_CM_CHAINS + LM_C_NLINKS * LM_NCHAIN + c];
      for (int k = 0; k < nl; k++) put(cm.data() + LM_CM_CHAINS + (LM_C_LINKS + k * LM_LINK_SIZE) * LM_NCHAIN + c, LM_NCHAIN);
    }
  }
  T.nv = (int)cmod[LM_H_NV]; T.nu = (int)cmod[LM_H_NU]; T.nobs = (int)cmod[LM_H_NOBS]; T.ngoal = (int)cmod[LM_H_NGOAL];
  T.nsub = (int)cmod[LM_H_NSUBSTEPS]; T.reward_type = (int)cmod[LM_H_REWARD_TYPE];
  T.n_chains = (int)cmod[LM_H_NCHAINS]; T.max_links = (int)cmod[LM_H_MAXLINKS]; T.ngrf = (int)cmod[LM_H_NGRF];
  T.max_contacts = (int)cmod[LM_H_MAXCONTACTS];
  T.npair = (int)cmod[LM_H_NGPAIR];
  {
    // every geom with a device collider is a condim-3 contact under pyramidal cones?
    T.all_pyr3 = (int)cmod[LM_H_CONE] == LM_CONE_PYRAMIDAL;
    for (int c = 0; c < LM_NCHAIN && T.all_pyr3; c++) {
      const int ng = (int)cmod[LM_HEADER_SIZE + LM_CM_CHAINS + LM_C_NGEOMS * LM_NCHAIN + c];
      for (int g = 0; g < ng; g++) if ((int)cmod[LM_HEADER_SIZE + LM_CM_SIZE + (g * LM_G_SIZE + LM_G_DIM) * LM_NCHAIN + c] != 3) T.all_pyr3 = 0;
    }
  }
  T.cm_used = ((int)cmod[LM_H_CM_USED] + 63) & ~63;          // keeps lane memory 256-byte aligned behind the table
  if (T.cm_used <= 0 || T.cm_used > ((LM_CM_SIZE + 63) & ~63)) { return fail("bad constant-table extent"); }
  if (T.ngoal > 4) { return fail("more than 4 goal entries"); }
  for (int i = 0; i < 8; i++) T.rp[i] = (float)cmod[LM_H_REWARD_P0 + i];
  lm::Params& P = m->P;
  P.h = (float)cmod[LM_H_TIMESTEP];
  P.g = lm::V3{(float)cmod[LM_H_GX], (float)cmod[LM_H_GY], (float)cmod[LM_H_GZ]};
  P.iterations = (int)cmod[LM_H_ITERATIONS];
  P.tolerance = 1e-6f;      // float32 stand-in for MuJoCo's 1e-8 (the gradient itself carries ~1e-6 relative noise)
  P.nv = T.nv;
  P.integrator = (int)cmod[LM_H_INTEGRATOR]; P.cone = (int)cmod[LM_H_CONE]; P.act_position = (int)cmod[LM_H_ACTMODE];
  P.scale = 1.0f / ((float)cmod[LM_H_MEANINERTIA] * (float)T.nv);
  P.off_runsup = (int)cmod[LM_H_OFF_RUNSUP];
  P.gt = m->d_gt;
  P.cmg = m->d_cm;
  {
    const size_t ngp = (size_t)cmod[LM_H_NGPAIR], off = (size_t)cmod[LM_H_OFF_GPT];
    if (ngp > 0 && n < off + ngp * LM_GPAIR_SIZE) return fail("chain model lacks the geom-pair table");
    std::vector<float> gpt(ngp * LM_GPAIR_SIZE + 1, 0.0f);
    for (size_t i = 0; i < ngp * LM_GPAIR_SIZE; i++) gpt[i] = (float)cmod[off + i];
    HIPCHK(hipMalloc(&m->d_gpt, sizeof(float) * gpt.size()));
    HIPCHK(hipMemcpy(m->d_gpt, gpt.data(), sizeof(float) * gpt.size(), hipMemcpyHostToDevice));
    P.gpt = m->d_gpt;
    m->n_gpt_floats = (int)(ngp * LM_GPAIR_SIZE);
  }
  {
    const size_t nmv = (size_t)cmod[LM_H_NMESHV], off = (size_t)cmod[LM_H_OFF_MESHV];
    if (nmv > 0 && n < off + 4 * nmv) return fail("chain model lacks the mesh-vertex table");
    std::vector<float> mv(4 * nmv + 4, 0.0f);
    for (size_t i = 0; i < 4 * nmv; i++) mv[i] = (float)cmod[off + i];
    HIPCHK(hipMalloc(&m->d_meshv, sizeof(float) * mv.size()));
    HIPCHK(hipMemcpy(m->d_meshv, mv.data(), sizeof(float) * mv.size(), hipMemcpyHostToDevice));
    P.meshv = m->d_meshv;
  }
  {
    const size_t nmn = (size_t)cmod[LM_H_NMESHN], off = (size_t)cmod[LM_H_OFF_MESHN];
    if (nmn > 0 && n < off + nmn) return fail("chain model lacks the hull-vertex neighbour table");
    std::vector<float> mn(nmn + 1, -1.0f);
    for (size_t i = 0; i < nmn; i++) mn[i] = (float)cmod[off + i];
    HIPCHK(hipMalloc(&m->d_meshn, sizeof(float) * mn.size()));
    HIPCHK(hipMemcpy(m->d_meshn, mn.data(), sizeof(float) * mn.size(), hipMemcpyHostToDevice));
    P.meshn = m->d_meshn;
  }
  {
    const size_t nbp = (size_t)cmod[LM_H_NBPAIR], off = (size_t)cmod[LM_H_OFF_BPT];
    if (nbp > 0 && n < off + nbp * LM_BP_SIZE) return fail("chain model lacks the body-pair table");
    std::vector<float> bp(nbp * LM_BP_SIZE + 1, 0.0f);
    for (size_t i = 0; i < nbp * LM_BP_SIZE; i++) bp[i] = (float)cmod[off + i];
    HIPCHK(hipMalloc(&m->d_bpt, sizeof(float) * bp.size()));
    HIPCHK(hipMemcpy(m->d_bpt, bp.data(), sizeof(float) * bp.size(), hipMemcpyHostToDevice));
    P.bpt = m->d_bpt;
  }
  {
    const size_t na = (size_t)cmod[LM_H_NMESHADJ], off = (size_t)cmod[LM_H_OFF_MESHADJ];
    if (na > 0 && n < off + 4 * na) return fail("chain model lacks the hull adjacency blocks");
    std::vector<float> ma(4 * na + 64, 0.0f);                  // padded: a step of the hill climbing fetches eight entries at once
    for (size_t i = 0; i < 4 * na; i++) ma[i] = (float)cmod[off + i];
    HIPCHK(hipMalloc(&m->d_meshadj, sizeof(float) * ma.size()));
    HIPCHK(hipMemcpy(m->d_meshadj, ma.data(), sizeof(float) * ma.size(), hipMemcpyHostToDevice));
    P.meshadj = m->d_meshadj;
  }
  P.ls_tol = 1e-2f; P.ls_iters = 12; P.ls_noise = 2e-6f; P.ablate = 0;
  P.root_limited = m->root_limited ? 1 : 0;
  P.root_xyz = m->root_xyz ? 1 : 0;
  P.ls_grid[0] = 0.25f; P.ls_grid[1] = 0.0625f; P.ls_grid[2] = 0.015625f;
  if (const char* v = LM_PROBE_ENV("LM_LS_GRID")) sscanf(v, "%f,%f,%f", &P.ls_grid[0], &P.ls_grid[1], &P.ls_grid[2]);   // A/B knob
  if (const char* v = LM_PROBE_ENV("LM_LS_NOISE")) P.ls_noise = (float)atof(v);
  if (const char* v = LM_PROBE_ENV("LM_ABLATE")) P.ablate = atoi(v);
  if (const char* v = LM_PROBE_ENV("LM_TOLERANCE")) P.tolerance = (float)atof(v);          // tuning knobs for A/B probes
  if (const char* v = LM_PROBE_ENV("LM_LS_TOL")) P.ls_tol = (float)atof(v);
  if (const char* v = LM_PROBE_ENV("LM_LS_ITERS")) P.ls_iters = atoi(v);
  *out = guard.release();
  return 0;
}

void lm_model_destroy(lm_model* m) {
  if (!m) return;
  if (m->d_cm) (void)hipFree(m->d_cm);
  if (m->d_gt) (void)hipFree(m->d_gt);
  if (m->d_gpt) (void)hipFree(m->d_gpt);
  if (m->d_meshv) (void)hipFree(m->d_meshv);
  if (m->d_meshn) (void)hipFree(m->d_meshn);
  if (m->d_bpt) (void)hipFree(m->d_bpt);
  if (m->d_meshadj) (void)hipFree(m->d_meshadj);
  if (m->d_mt) (void)hipFree(m->d_mt);
  delete m;
}

int lm_model_dims(const lm_model* m, lm_dims* out) {
  out->nq = m->T.nv; out->nv = m->T.nv; out->nu = m->T.nu; out->nobs = m->T.nobs; out->ngoal = m->T.ngoal;
  out->n_substeps = m->T.nsub; out->n_chains = m->T.n_chains; out->max_chain_dofs = m->T.max_links; out->na = m->T.na;
  return 0;
}

void lm_batch_destroy(lm_batch* b);

static int batch_alloc(lm_batch* b) {
  lm_model* m = b->m;
  const int N = b->N, nv = m->T.nv;
  HIPCHK(hipMalloc(&b->qpos, sizeof(float) * nv * N)); HIPCHK(hipMalloc(&b->qvel, sizeof(float) * nv * N));
  HIPCHK(hipMalloc(&b->warm, sizeof(float) * nv * N)); HIPCHK(hipMalloc(&b->goal, sizeof(float) * 4 * N));
  HIPCHK(hipMalloc(&b->action, sizeof(float) * m->T.nu * N)); HIPCHK(hipMalloc(&b->obs, sizeof(float) * m->T.nobs * N));
  HIPCHK(hipMalloc(&b->reward, sizeof(float) * N)); HIPCHK(hipMalloc(&b->done, N));
  HIPCHK(hipMalloc(&b->flags, N)); HIPCHK(hipMemset(b->flags, 0, N));
  HIPCHK(hipMalloc(&b->ep_step, sizeof(int) * N)); HIPCHK(hipMalloc(&b->ep_count, sizeof(unsigned) * N));
  if (m->T.na > 0) { HIPCHK(hipMalloc(&b->act, sizeof(float) * m->T.na * N)); HIPCHK(hipMemset(b->act, 0, sizeof(float) * m->T.na * N)); }
  b->stat_pre_off = b->nblocks; b->nstat = b->nblocks + lmk::kReplayGrid;      // the concurrent replay kernel adds into slots of its own
  HIPCHK(hipMalloc(&b->stats, sizeof(DevStats) * b->nstat));
  HIPCHK(hipHostMalloc((void**)&b->h_hint, sizeof(int) * 4, hipHostMallocDefault)); b->h_hint[0] = 0; b->h_hint[1] = 0; b->h_hint[2] = 0; b->h_hint[3] = 0;
  HIPCHK(hipMalloc(&b->replay_list, sizeof(int) * N)); HIPCHK(hipMalloc(&b->stall, sizeof(int) * N)); HIPCHK(hipMalloc(&b->replay_ctl, sizeof(int) * 8));
  HIPCHK(hipMemset(b->replay_list, 0, sizeof(int) * N)); HIPCHK(hipMemset(b->stall, 0, sizeof(int) * N)); HIPCHK(hipMemset(b->replay_ctl, 0, sizeof(int) * 8));
  HIPCHK(hipMalloc(&b->replay_mark, N)); HIPCHK(hipMemset(b->replay_mark, 0, N));
  HIPCHK(hipMalloc(&b->hq, sizeof(float) * nv * N)); HIPCHK(hipMalloc(&b->hv, sizeof(float) * nv * N)); HIPCHK(hipMalloc(&b->hw, sizeof(float) * nv * N));
  HIPCHK(hipMalloc(&b->hsub, sizeof(int) * N)); HIPCHK(hipMemset(b->hsub, 0, sizeof(int) * N));
  HIPCHK(hipMalloc(&b->premark, sizeof(int) * N)); HIPCHK(hipMemset(b->premark, 0, sizeof(int) * N));
  HIPCHK(hipMalloc(&b->slack, sizeof(float) * 12 * N)); HIPCHK(hipMemset(b->slack, 0, sizeof(float) * 12 * N));
  // the convex collider's warm-start cache: one record per environment and geom-pair record of the models whose hull pairs run through
  // it in kernels with five or more links per chain (128 B each: HumanoidTorque 692 pairs -> 88 KB per environment, 363 MB at 4096)
  b->mprc = nullptr; b->mprc_pairs = 0;
  if (m->d_meshadj && m->n_gpt_floats > 0 && m->T.max_links > 3) {
    b->mprc_pairs = m->n_gpt_floats / LM_GPAIR_SIZE;
    const size_t bytes = sizeof(float) * lm::kMprCacheFloats * (size_t)b->mprc_pairs * (size_t)N;
    HIPCHK(hipMalloc(&b->mprc, bytes)); HIPCHK(hipMemset(b->mprc, 0, bytes));
  }
  HIPCHK(hipMemset(b->qpos, 0, sizeof(float) * nv * N)); HIPCHK(hipMemset(b->qvel, 0, sizeof(float) * nv * N));
  HIPCHK(hipMemset(b->warm, 0, sizeof(float) * nv * N)); HIPCHK(hipMemset(b->goal, 0, sizeof(float) * 4 * N));
  HIPCHK(hipMemset(b->ep_step, 0, sizeof(int) * N)); HIPCHK(hipMemset(b->ep_count, 0, sizeof(unsigned) * N));
  HIPCHK(hipMemset(b->stats, 0, sizeof(DevStats) * b->nstat));
  HIPCHK(hipMalloc(&b->timers, sizeof(unsigned long long) * (32 + 32 * (size_t)b->nblocks))); HIPCHK(hipMemset(b->timers, 0, sizeof(unsigned long long) * (32 + 32 * (size_t)b->nblocks)));
#ifdef LM_TIMERS
  HIPCHK(hipMalloc(&b->tline, sizeof(unsigned long long) * (4 * (size_t)N + 2 * (size_t)b->nblocks))); HIPCHK(hipMemset(b->tline, 0, sizeof(unsigned long long) * (4 * (size_t)N + 2 * (size_t)b->nblocks)));
#endif
  HIPCHK(hipStreamCreate(&b->stream)); HIPCHK(hipStreamCreate(&b->stream2));
  HIPCHK(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&b->ev_done[0], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&b->ev_done[1], hipEventDisableTiming));
  HIPCHK(hipEventCreate(&b->ev0)); HIPCHK(hipEventCreate(&b->ev1)); HIPCHK(hipEventCreateWithFlags(&b->ev_ext, hipEventDisableTiming));
  return 0;
}

int lm_batch_create(lm_model* m, int n_envs, lm_batch** out) {
  if (n_envs <= 0) return fail("n_envs must be positive");
  HIPCHK(hipSetDevice(m->device));
  lm_batch* b = new lm_batch();
  memset(b, 0, sizeof(*b));
  b->m = m; b->N = n_envs; b->n_active = n_envs; b->env_map = nullptr;
  // Four environments per workgroup: with the replicated layout that is one full wave (4 envs x 4 replicas x 4 chains).
  // Larger batches simply run more workgroups back to back (wider workgroups without replicas were 30-50 % slower at
  // every size, profiles/r1_ab_probes.md); lm_batch_set_layout selects the plain layout.
  {
    int epb = n_envs < 4 ? n_envs : 4;
    const char* ov = LM_PROBE_ENV("LM_ENVS_PER_BLOCK");
    if (ov && atoi(ov) >= 1 && atoi(ov) <= 16) epb = atoi(ov);
    b->epb = epb;
  }
  b->nblocks = (n_envs + b->epb - 1) / b->epb;
  b->replay = 1;
  {
    // a model with self-collision tables needs a kernel family with the pair pass: anything else would silently not simulate them
    const int fam = family_of(b);
    if (m->T.npair > 0 && !family_has_pairs(fam)) {
      delete b;
      return fail("the model carries self-collision tables but its kernel family has no pair pass (RK4 with muscles, or a cone / condim the pair families are not compiled for)");
    }
  }
  if (batch_alloc(b)) { lm_batch_destroy(b); return 1; }     // g_err holds the failed call; nothing leaks
  *out = b;
  return 0;
}

/* speculate / replay (lm_step.h): on (default) = a control step that needs more contact slots, longer pair lists or — the
   quadruped — the convex collider is replayed by the family's big kernel instead of dropping contacts; off = the regular kernels
   alone (contacts beyond the slots are dropped and counted: the behaviour of rounds 1-3, kept for A/B measurements). */
int lm_batch_set_replay(lm_batch* b, int enabled) {
  if (!b) return fail("null batch");
  // 2 (tests): every control step goes through the replay kernel; 3 / 4 = 1 / 2 without pollers: the replay kernel only as the pass
  // behind the regular launch (profilers that run one kernel at a time would leave the pollers waiting for their time-out)
  if (!enabled && b->m->T.na == 0 && family_of(b) >= 0 && family_of(b) != 6) {
    // the regular kernels of the families without muscles have no limit rows for the root dofs: without the replay kernel a root dof
    // beyond its range would run without its row (flagged per step, but wrong physics) — refuse rather than offer that
    if (b->m->root_limited) return fail("this model has a limited root joint whose limit rows live in the replay kernel: replay cannot be switched off");
  }
  b->replay = (enabled >= 2 && enabled <= 4) ? enabled : (enabled ? 1 : 0);
  return 0;
}

/* launch geometry: environments per workgroup. 4 (the default) = the replicated layout, one wave = 4 environments x 4 replicas x 4
   chains; 8 or 16 = the plain layout, a workgroup of 8 / 16 quads without replicas (the only other layout the families are compiled
   for; slower at every batch size measured, profiles/r1_ab_probes.md, kept for very large batches and as a cross-check of the
   replicas' protocol). Statistics slots were allocated for the default: only coarser geometries are accepted. */
int lm_batch_set_layout(lm_batch* b, int envs_per_workgroup) {
  if (!b) return fail("null batch");
  const int def = b->N < 4 ? b->N : 4;
  if (envs_per_workgroup == 4) envs_per_workgroup = def;      // the advertised default, also for a batch of fewer than four environments
  if (envs_per_workgroup != def && envs_per_workgroup != 8 && envs_per_workgroup != 16) return fail("environments per workgroup: 4 (replicated layout), 8 or 16 (plain layout)");
  if (envs_per_workgroup > def && !family_has_replicas(b)) return fail("the generic kernel family has one layout only");
  b->epb = envs_per_workgroup;
  b->nblocks = (b->N + b->epb - 1) / b->epb;
  return 0;
}

void lm_batch_destroy(lm_batch* b) {
  if (!b) return;
  hipSetDevice(b->m->device);
  if (b->stream) hipStreamSynchronize(b->stream);
  void* bufs[] = {b->qpos, b->qvel, b->warm, b->goal, b->action, b->obs, b->reward, b->done, b->flags, b->ep_step, b->ep_count, b->stats,
                  b->table, b->act, b->dofprm, b->drspec, b->timers, b->scr, b->scr_idx, b->replay_list, b->replay_ctl, b->stall, b->replay_mark, b->slack, b->mprc, b->env_map, b->hq, b->hv, b->hw, b->hsub, b->premark, b->tline,
                  b->vrec, b->vgt, b->vgpt, b->var, b->mc_ib, b->mc_db, b->vdirty, b->mc_mask, b->vgen, b->vdraws};
  for (void* p : bufs) if (p) (void)hipFree(p);
  if (b->d_perm) (void)hipFree(b->d_perm);
  if (b->h_act) (void)hipHostFree(b->h_act);
  for (int i = 0; i < LM_PINNED_SLOTS; i++) if (b->h_out64[i]) (void)hipHostFree(b->h_out64[i]);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  if (b->ev_ext) (void)hipEventDestroy(b->ev_ext);
  if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
  if (b->ev_join) (void)hipEventDestroy(b->ev_join);
  if (b->ev_done[0]) (void)hipEventDestroy(b->ev_done[0]);
  if (b->ev_done[1]) (void)hipEventDestroy(b->ev_done[1]);
  if (b->stream2) { (void)hipStreamSynchronize(b->stream2); (void)hipStreamDestroy(b->stream2); }
  if (b->h_hint) (void)hipHostFree(b->h_hint);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
}

// masked uploads move only the masked environments: compact rows + their indices go up, a small kernel scatters them into
// the [dim][N] arrays (reference counterpart: the per-environment reset of LocoEnv.reset, environments/base.py:344-373)
__global__ void scatter_rows(float* __restrict__ dst, const float* __restrict__ rows, const int* __restrict__ idx,
                             int n, int dim, int N) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * dim) return;
  const int k = t / dim, d = t - k * dim;
  dst[(size_t)d * N + idx[k]] = rows ? rows[t] : 0.0f;
}
__global__ void zero_ints(int* __restrict__ dst, const int* __restrict__ idx, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) dst[idx[t]] = 0;
}

// indices of the masked environments, uploaded once per call into b->scr_idx; staging sized for `dim` floats per row
static int mask_indices(lm_batch* b, const uint8_t* mask, int dim, std::vector<int>& idx) {
  idx.clear();
  for (int e = 0; e < b->N; e++) if (mask[e]) idx.push_back(e);
  const size_t need = (size_t)std::max<size_t>(idx.size(), 1) * (size_t)std::max(dim, 1);
  if (need > b->scr_cap) {
    HIPCHK(hipStreamSynchronize(b->stream));
    if (b->scr) HIPCHK(hipFree(b->scr));
    b->scr = nullptr; b->scr_cap = 0;
    HIPCHK(hipMalloc(&b->scr, sizeof(float) * need));
    b->scr_cap = need;
  }
  if (!b->scr_idx) HIPCHK(hipMalloc(&b->scr_idx, sizeof(int) * b->N));
  if (!idx.empty()) HIPCHK(hipMemcpyAsync(b->scr_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return 0;
}

// rows == nullptr: zero the masked rows
static int scatter_masked(lm_batch* b, float* dev, const float* host_aos, int dim, const std::vector<int>& idx) {
  const int n = (int)idx.size();
  if (n == 0 || dim == 0) return 0;
  if (host_aos) {
    std::vector<float> rows((size_t)n * dim);
    for (int k = 0; k < n; k++) memcpy(&rows[(size_t)k * dim], host_aos + (size_t)idx[k] * dim, sizeof(float) * dim);
    HIPCHK(hipMemcpyAsync(b->scr, rows.data(), sizeof(float) * n * dim, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));     // `rows` is pageable host memory that dies with this frame
  }
  const int threads = 256, blocks = (n * dim + threads - 1) / threads;
  hipLaunchKernelGGL(scatter_rows, dim3(blocks), dim3(threads), 0, b->stream, dev, host_aos ? b->scr : nullptr, b->scr_idx, n, dim, b->N);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(b->stream));
  return 0;
}

static int upload_soa(lm_batch* b, float* dev, const float* host_aos, int dim, const uint8_t* mask) {
  const int N = b->N;
  if (mask) {
    std::vector<int> idx;
    if (mask_indices(b, mask, dim, idx)) return 1;
    return scatter_masked(b, dev, host_aos, dim, idx);
  }
  std::vector<float> soa((size_t)dim * N);
  for (int e = 0; e < N; e++)
    for (int d = 0; d < dim; d++) soa[(size_t)d * N + e] = host_aos[(size_t)e * dim + d];
  HIPCHK(hipMemcpyAsync(dev, soa.data(), sizeof(float) * dim * N, hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return 0;
}

int lm_set_state(lm_batch* b, const float* qpos, const float* qvel, const uint8_t* mask) {
  HIPCHK(hipSetDevice(b->m->device));
  const int N = b->N, nv = b->m->T.nv, na = b->m->T.na;
  if (mask) {
    // positions, velocities, and the cleared warm start / activations / step counter of the masked environments only
    std::vector<int> idx;
    if (mask_indices(b, mask, nv, idx)) return 1;
    if (scatter_masked(b, b->qpos, qpos, nv, idx)) return 1;
    if (scatter_masked(b, b->qvel, qvel, nv, idx)) return 1;
    if (scatter_masked(b, b->warm, nullptr, nv, idx)) return 1;
    if (b->act && scatter_masked(b, b->act, nullptr, na, idx)) return 1;
    if (scatter_masked(b, b->slack, nullptr, 12, idx)) return 1;           // new positions: the self-collision detection is due
    if (!idx.empty()) {
      const int n = (int)idx.size();
      hipLaunchKernelGGL(zero_ints, dim3((n + 255) / 256), dim3(256), 0, b->stream, b->ep_step, b->scr_idx, n);
      hipLaunchKernelGGL(zero_ints, dim3((n + 255) / 256), dim3(256), 0, b->stream, b->premark, b->scr_idx, n);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(b->stream));
    }
    return 0;
  }
  if (upload_soa(b, b->qpos, qpos, nv, nullptr)) return 1;
  if (upload_soa(b, b->qvel, qvel, nv, nullptr)) return 1;
  if (b->act) HIPCHK(hipMemsetAsync(b->act, 0, sizeof(float) * na * N, b->stream));
  HIPCHK(hipMemsetAsync(b->warm, 0, sizeof(float) * nv * N, b->stream));
  HIPCHK(hipMemsetAsync(b->ep_step, 0, sizeof(int) * N, b->stream));
  HIPCHK(hipMemsetAsync(b->premark, 0, sizeof(int) * N, b->stream));
  HIPCHK(hipMemsetAsync(b->slack, 0, sizeof(float) * 12 * N, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return 0;
}

int lm_get_state(lm_batch* b, float* qpos, float* qvel) {
  HIPCHK(hipSetDevice(b->m->device));
  const int N = b->N, nv = b->m->T.nv;
  std::vector<float> soa((size_t)nv * N);
  for (int pass = 0; pass < 2; pass++) {
    float* dst = pass == 0 ? qpos : qvel;
    if (!dst) continue;
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipMemcpy(soa.data(), pass == 0 ? b->qpos : b->qvel, sizeof(float) * nv * N, hipMemcpyDeviceToHost));
    for (int e = 0; e < N; e++) for (int d = 0; d < nv; d++) dst[(size_t)e * nv + d] = soa[(size_t)d * N + e];
  }
  return 0;
}

int lm_set_dof_params(lm_batch* b, const float* damping, const float* stiffness, const float* frictionloss, const uint8_t* mask) {
  HIPCHK(hipSetDevice(b->m->device));
  const int N = b->N, nv = b->m->T.nv;
  if (!b->dofprm) {
    // fail HERE, not at the first launch: the generic kernels (and six-link RK4 models, which have no kernel at all) are not
    // compiled for per-environment joint parameters / model variants
    if (family_of(b) < 0 || family_of(b) == 6)
      return fail("per-environment joint parameters and model variants are not compiled for this model's kernel family (generic kernels)");
    std::vector<float> init((size_t)3 * nv * N);
    for (int p = 0; p < 3; p++) for (int d = 0; d < nv; d++) for (int e = 0; e < N; e++) init[((size_t)p * nv + d) * N + e] = b->m->nominal[(size_t)p * nv + d];
    HIPCHK(hipMalloc(&b->dofprm, sizeof(float) * 3 * nv * N));
    HIPCHK(hipMemcpy(b->dofprm, init.data(), sizeof(float) * 3 * nv * N, hipMemcpyHostToDevice));
  }
  const float* src[3] = {damping, stiffness, frictionloss};
  if (damping || stiffness || frictionloss) b->dofprm_of_variants = false;      // the caller's own values: they outlive the variant pool
  for (int p = 0; p < 3; p++) {
    if (!src[p]) continue;
    for (size_t i = 0; i < (size_t)N * nv; i++) if (!(src[p][i] >= 0.0f) && (!mask || mask[i / nv])) return fail("joint parameters must be non-negative");
    if (upload_soa(b, b->dofprm + (size_t)p * nv * N, src[p], nv, mask)) return 1;
  }
  return 0;
}

int lm_get_dof_params(lm_batch* b, float* damping, float* stiffness, float* frictionloss) {
  HIPCHK(hipSetDevice(b->m->device));
  const int N = b->N, nv = b->m->T.nv;
  float* dst[3] = {damping, stiffness, frictionloss};
  std::vector<float> soa((size_t)nv * N);
  HIPCHK(hipStreamSynchronize(b->stream));
  for (int p = 0; p < 3; p++) {
    if (!dst[p]) continue;
    if (b->dofprm) HIPCHK(hipMemcpy(soa.data(), b->dofprm + (size_t)p * nv * N, sizeof(float) * nv * N, hipMemcpyDeviceToHost));
    for (int e = 0; e < N; e++) for (int d = 0; d < nv; d++) dst[p][(size_t)e * nv + d] = b->dofprm ? soa[(size_t)d * N + e] : b->m->nominal[(size_t)p * nv + d];
  }
  return 0;
}

int lm_set_dof_randomization(lm_batch* b, const float* spec) {
  HIPCHK(hipSetDevice(b->m->device));
  const int nv = b->m->T.nv;
  if (!spec) { if (b->drspec) { (void)hipFree(b->drspec); b->drspec = nullptr; } return 0; }
  for (int i = 0; i < 3 * nv; i++) { const int k = (int)spec[3 * i]; if (k < 0 || k > 3) return fail("bad randomisation kind"); }
  if (!b->dofprm && lm_set_dof_params(b, nullptr, nullptr, nullptr, nullptr)) return 1;
  if (!b->drspec) HIPCHK(hipMalloc(&b->drspec, sizeof(float) * 9 * nv));
  HIPCHK(hipMemcpy(b->drspec, spec, sizeof(float) * 9 * nv, hipMemcpyHostToDevice));
  return 0;
}

static void compile_models(lm_batch* b, const unsigned char* mask, int all);

int lm_set_model_variants(lm_batch* b, const float* records, const float* geom_tables, const float* pair_tables,
                          int pair_floats, int n_variants) {
  HIPCHK(hipSetDevice(b->m->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  for (float** p : {&b->vrec, &b->vgt, &b->vgpt}) if (*p) { (void)hipFree(*p); *p = nullptr; }
  b->nvar = 0; b->gpt_floats = 0;
  if (b->mc_ib) { (void)hipFree(b->mc_ib); (void)hipFree(b->mc_db); b->mc_ib = nullptr; b->mc_db = nullptr; }      // a pool replaces the compiler
  if (n_variants <= 0) {       // pool removed: the joint-parameter rows go with it when this call had created them
    if (b->dofprm_of_variants && b->dofprm && !b->drspec) { (void)hipFree(b->dofprm); b->dofprm = nullptr; }
    b->dofprm_of_variants = false;
    return 0;
  }
  if (!records || !geom_tables) return fail("model variants need inertial records and geom tables");
  if ((pair_tables != nullptr) != (b->m->n_gpt_floats > 0) || (pair_tables && pair_floats != b->m->n_gpt_floats))
    return fail("geom-pair tables of the variants do not match the model's");
  if (!b->dofprm) {                                  // the kernels with per-environment parameters
    if (lm_set_dof_params(b, nullptr, nullptr, nullptr, nullptr)) return 1;
    b->dofprm_of_variants = true;
  }
  const size_t nr = (size_t)n_variants * LM_IR_SIZE * LM_NCHAIN, ng = (size_t)n_variants * LM_GT_SIZE, np_ = (size_t)n_variants * pair_floats;
  HIPCHK(hipMalloc(&b->vrec, sizeof(float) * nr)); HIPCHK(hipMemcpy(b->vrec, records, sizeof(float) * nr, hipMemcpyHostToDevice));
  HIPCHK(hipMalloc(&b->vgt, sizeof(float) * ng)); HIPCHK(hipMemcpy(b->vgt, geom_tables, sizeof(float) * ng, hipMemcpyHostToDevice));
  if (pair_tables) { HIPCHK(hipMalloc(&b->vgpt, sizeof(float) * np_)); HIPCHK(hipMemcpy(b->vgpt, pair_tables, sizeof(float) * np_, hipMemcpyHostToDevice)); }
  if (!b->var) HIPCHK(hipMalloc(&b->var, sizeof(int) * b->N));
  HIPCHK(hipMemset(b->var, 0, sizeof(int) * b->N));
  HIPCHK(hipMemset(b->slack, 0, sizeof(float) * 12 * b->N));
  b->nvar = n_variants; b->gpt_floats = pair_floats;
  return 0;
}

int lm_set_variant_index(lm_batch* b, const int32_t* index, const uint8_t* mask) {
  HIPCHK(hipSetDevice(b->m->device));
  if (b->nvar <= 0) return fail("the batch has no model variants");
  if (b->mc_ib) return fail("the model compiler is on: every environment owns its slot (lm_compile_models draws a new model)");
  std::vector<int> cur(b->N);
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(cur.data(), b->var, sizeof(int) * b->N, hipMemcpyDeviceToHost));
  for (int e = 0; e < b->N; e++) {
    if (mask && !mask[e]) continue;
    if (index[e] < 0 || index[e] >= b->nvar) return fail("variant index out of range");
    cur[e] = index[e];
  }
  HIPCHK(hipMemcpy(b->var, cur.data(), sizeof(int) * b->N, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(b->slack, 0, sizeof(float) * 12 * b->N));       // another model: whatever the pair pass knew is void
  return 0;
}

int lm_set_variant_rows(lm_batch* b, int rows_per_variant) {
  if (rows_per_variant < 0) return fail("rows_per_variant must be >= 0");
  if (rows_per_variant > 0 && b->mc_ib) return fail("the model compiler is on: the model does not follow the reset-table row");
  if (rows_per_variant > 0 && (b->nvar <= 0 || b->table_rows != b->nvar * rows_per_variant))
    return fail("the reset table must hold n_variants blocks of rows_per_variant rows (set the variants and the table first)");
  b->var_rows = rows_per_variant;
  return 0;
}

int lm_get_variant_index(lm_batch* b, int32_t* index) {
  HIPCHK(hipSetDevice(b->m->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  if (b->nvar <= 0) { for (int e = 0; e < b->N; e++) index[e] = 0; return 0; }
  HIPCHK(hipMemcpy(index, b->var, sizeof(int) * b->N, hipMemcpyDeviceToHost));
  return 0;
}

int lm_set_model_compiler(lm_batch* b, const int32_t* index, long long n_index, const double* data, long long n_data,
                          const float* record, const float* geom_table, const float* pair_table, int pair_floats, uint64_t seed) {
  HIPCHK(hipSetDevice(b->m->device));
  if (!index || !data || n_index < lmc::kIntHead || n_data < lmc::kDblHead) return fail("the model compiler needs its program");
  if ((unsigned)index[0] != lmc::kMagic) return fail("not a model-compiler program (lowering.model_compiler_tables)");
  const int nv = index[1], nrb = index[2], ngs = index[3], nd = index[4], nslot = index[5], nrec = index[6], ncon = index[7], nbody = index[8];
  if (nv != b->m->T.nv) return fail("the model-compiler program is of another model (nv)");
  if (nv > lmc::kMaxNv || nrb > lmc::kMaxRbody || ngs > lmc::kMaxGslot || nd > lmc::kMaxDraw || nslot > lmc::kMaxSlot || nbody > lmc::kMaxBody || nd < 1)
    return fail("the model-compiler program is beyond the device compiler's tables (lm_compile.h)");
  const long long want_i = lmc::kIntHead + (long long)nd * lmc::kDrawInts + (long long)nrb * lmc::kRbInts + 2ll * nrec + (long long)ncon * lmc::kConInts;
  const long long want_d = lmc::kDblHead + (long long)nd * lmc::kDrawDbls + (long long)nrb * lmc::kRbDbls + (long long)nbody * 6 * nv + (long long)nv * nv + 2ll * nv +
                           (long long)nslot * 10 + 3ll * ngs;
  if (want_i != n_index || want_d != n_data) return fail("the model-compiler program has the wrong size");
  const int* recops = index + lmc::kIntHead + nd * lmc::kDrawInts + nrb * lmc::kRbInts;
  for (int i = 0; i < nrec; i++)
    if (recops[2 * i] < 0 || recops[2 * i] >= LM_IR_SIZE * LM_NCHAIN || recops[2 * i + 1] < 0 || recops[2 * i + 1] >= 3 * nv + 1 + nslot * 10)
      return fail("the model-compiler program writes outside the inertial record");
  const int* conops = recops + 2 * nrec;
  for (int i = 0; i < ncon; i++) {
    const int* op = conops + i * lmc::kConInts;
    const long long last = (long long)op[1] + 12ll * op[2], cap = op[0] == 0 ? (long long)LM_GT_SIZE : (long long)pair_floats;
    if (op[0] < 0 || op[0] > 1 || op[1] < 0 || op[2] < 1 || last >= cap || op[4] < 0 || op[4] >= ngs || op[5] < 0 || op[5] >= ngs ||
        op[6] < 0 || op[6] >= nbody || op[7] < 0 || op[7] >= nbody)
      return fail("the model-compiler program writes outside the contact tables");
  }
  // what the draws and the drawn bodies index (the kernel sizes its LDS tables by the header's counts and indexes the body Jacobians by
  // these numbers: a malformed program must not get past this point — round-5 advisor; the Python binding checks the same)
  const int* draws = index + lmc::kIntHead;
  for (int i = 0; i < nd; i++) {
    const int kind = draws[4 * i], target = draws[4 * i + 1], idx = draws[4 * i + 2], comp = draws[4 * i + 3];
    const int lim_i = target == 0 ? nv : (target >= 1 && target <= 3 ? nrb : (target == 4 ? ngs : -1)), lim_c = (target == 0 || target == 1) ? 1 : 3;
    if (kind < 1 || kind > 3 || lim_i < 0 || idx < 0 || idx >= lim_i || comp < 0 || comp >= lim_c) return fail("the model-compiler program has a draw outside its table");
  }
  const int* rbody = draws + nd * lmc::kDrawInts;
  for (int i = 0; i < nrb; i++) {
    const int body = rbody[4 * i], kind = rbody[4 * i + 1], slot = rbody[4 * i + 2], has_sv = rbody[4 * i + 3];
    if (body <= 0 || body >= nbody || kind < 1 || kind > 2 || slot < 0 || slot >= nslot || has_sv < 0 || has_sv > 1) return fail("the model-compiler program has a drawn body outside the model");
  }
  // ... and the nominal tables, BEFORE anything of the batch's current variant state is torn down
  if (!record || !geom_table) return fail("the model compiler needs the nominal inertial record and geom table");
  if ((pair_table != nullptr) != (b->m->n_gpt_floats > 0) || (pair_table && pair_floats != b->m->n_gpt_floats))
    return fail("the geom-pair table does not match the model's");
  // the tables: one slot per environment, every slot starts as the nominal model
  if (lm_set_model_variants(b, nullptr, nullptr, nullptr, 0, 0)) return 1;
  if (!b->dofprm) {
    if (lm_set_dof_params(b, nullptr, nullptr, nullptr, nullptr)) return 1;
    b->dofprm_of_variants = true;
  }
  const int N = b->N;
  const size_t nr = (size_t)LM_IR_SIZE * LM_NCHAIN, ng = (size_t)LM_GT_SIZE, np_ = (size_t)(pair_table ? pair_floats : 0);
  float* nominal = nullptr;
  HIPCHK(hipMalloc(&nominal, sizeof(float) * (nr + ng + np_)));
  HIPCHK(hipMemcpy(nominal, record, sizeof(float) * nr, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(nominal + nr, geom_table, sizeof(float) * ng, hipMemcpyHostToDevice));
  if (np_) HIPCHK(hipMemcpy(nominal + nr + ng, pair_table, sizeof(float) * np_, hipMemcpyHostToDevice));
  HIPCHK(hipMalloc(&b->vrec, sizeof(float) * nr * N)); HIPCHK(hipMalloc(&b->vgt, sizeof(float) * ng * N));
  if (np_) HIPCHK(hipMalloc(&b->vgpt, sizeof(float) * np_ * N));
  lmc::replicate(b->vrec, nominal, (long long)nr, N, b->stream);
  lmc::replicate(b->vgt, nominal + nr, (long long)ng, N, b->stream);
  if (np_) lmc::replicate(b->vgpt, nominal + nr + ng, (long long)np_, N, b->stream);
  if (!b->var) HIPCHK(hipMalloc(&b->var, sizeof(int) * N));
  lmc::iota(b->var, N, b->stream);
  HIPCHK(hipMalloc(&b->mc_ib, sizeof(int) * n_index)); HIPCHK(hipMemcpy(b->mc_ib, index, sizeof(int) * n_index, hipMemcpyHostToDevice));
  HIPCHK(hipMalloc(&b->mc_db, sizeof(double) * n_data)); HIPCHK(hipMemcpy(b->mc_db, data, sizeof(double) * n_data, hipMemcpyHostToDevice));
  for (void** p : {(void**)&b->vdirty, (void**)&b->mc_mask, (void**)&b->vgen, (void**)&b->vdraws}) if (*p) { (void)hipFree(*p); *p = nullptr; }
  HIPCHK(hipMalloc(&b->vdirty, N)); HIPCHK(hipMemset(b->vdirty, 0, N));
  HIPCHK(hipMalloc(&b->mc_mask, N));
  HIPCHK(hipMalloc(&b->vgen, sizeof(unsigned) * N)); HIPCHK(hipMemset(b->vgen, 0, sizeof(unsigned) * N));
  HIPCHK(hipMalloc(&b->vdraws, sizeof(double) * (size_t)N * nd));
  b->mc_ndraw = nd; b->mc_seed = seed; b->nvar = N; b->gpt_floats = (int)np_; b->var_rows = 0;
  compile_models(b, nullptr, 1);                       // every environment starts on a model of its own
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipFree(nominal));
  return 0;
}

int lm_compile_models(lm_batch* b, const uint8_t* mask) {
  HIPCHK(hipSetDevice(b->m->device));
  if (!b->mc_ib) return fail("the batch has no model compiler (lm_set_model_compiler)");
  if (mask) HIPCHK(hipMemcpyAsync(b->mc_mask, mask, b->N, hipMemcpyHostToDevice, b->stream));
  compile_models(b, mask ? b->mc_mask : nullptr, mask ? 0 : 1);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(b->stream));
  return 0;
}

int lm_get_model_draws(lm_batch* b, double* draws, uint32_t* generation) {
  HIPCHK(hipSetDevice(b->m->device));
  if (!b->mc_ib) return fail("the batch has no model compiler (lm_set_model_compiler)");
  HIPCHK(hipStreamSynchronize(b->stream));
  if (draws) HIPCHK(hipMemcpy(draws, b->vdraws, sizeof(double) * (size_t)b->N * b->mc_ndraw, hipMemcpyDeviceToHost));
  if (generation) HIPCHK(hipMemcpy(generation, b->vgen, sizeof(unsigned) * b->N, hipMemcpyDeviceToHost));
  return 0;
}

int lm_get_model_tables(lm_batch* b, int env, float* record, float* geom_table, float* pair_table) {
  HIPCHK(hipSetDevice(b->m->device));
  if (b->nvar <= 0) return fail("the batch has no model variants");
  if (env < 0 || env >= b->N) return fail("environment out of range");
  HIPCHK(hipStreamSynchronize(b->stream));
  int var = 0;
  HIPCHK(hipMemcpy(&var, b->var + env, sizeof(int), hipMemcpyDeviceToHost));
  if (record) HIPCHK(hipMemcpy(record, b->vrec + (size_t)var * LM_IR_SIZE * LM_NCHAIN, sizeof(float) * LM_IR_SIZE * LM_NCHAIN, hipMemcpyDeviceToHost));
  if (geom_table) HIPCHK(hipMemcpy(geom_table, b->vgt + (size_t)var * LM_GT_SIZE, sizeof(float) * LM_GT_SIZE, hipMemcpyDeviceToHost));
  if (pair_table && b->vgpt) HIPCHK(hipMemcpy(pair_table, b->vgpt + (size_t)var * b->gpt_floats, sizeof(float) * b->gpt_floats, hipMemcpyDeviceToHost));
  return 0;
}

int lm_set_activation(lm_batch* b, const float* act, const uint8_t* mask) {
  HIPCHK(hipSetDevice(b->m->device));
  if (!b->act) return fail("model has no activation states");
  return upload_soa(b, b->act, act, b->m->T.na, mask);
}

int lm_get_activation(lm_batch* b, float* act) {
  HIPCHK(hipSetDevice(b->m->device));
  if (!b->act) return fail("model has no activation states");
  const int N = b->N, na = b->m->T.na;
  std::vector<float> soa((size_t)na * N);
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(soa.data(), b->act, sizeof(float) * na * N, hipMemcpyDeviceToHost));
  for (int e = 0; e < N; e++) for (int d = 0; d < na; d++) act[(size_t)e * na + d] = soa[(size_t)d * N + e];
  return 0;
}

int lm_set_goal(lm_batch* b, const float* goal, const uint8_t* mask) {
  HIPCHK(hipSetDevice(b->m->device));
  if (b->m->T.ngoal == 0) return 0;
  return upload_soa(b, b->goal, goal, b->m->T.ngoal, mask);
}

static KArgs make_args(lm_batch* b) {
  KArgs a;
  memset(&a, 0, sizeof(a));
  a.cm = b->m->d_cm; a.mt = b->m->d_mt; a.act = b->act; a.dofprm = b->dofprm; a.drspec = b->drspec;
  a.vrec = b->nvar > 0 ? b->vrec : nullptr; a.vgt = b->vgt; a.vgpt = b->vgpt; a.var = b->var; a.nvar = b->nvar; a.gpt_floats = b->gpt_floats; a.var_rows = b->var_rows; a.vdirty = b->mc_ib ? b->vdirty : nullptr; a.qpos = b->qpos; a.qvel = b->qvel; a.warm = b->warm; a.goal = b->goal;
  a.ep_step = b->ep_step; a.ep_count = b->ep_count; a.flags = b->flags; a.slack = b->slack; a.mprc = b->mprc; a.mprc_pairs = b->mprc_pairs; a.env_map = b->env_map; a.n_active = b->n_active;
  a.table = b->table; a.table_rows = b->table_rows; a.seed = b->seed; a.env_offset = b->env_offset;
  a.auto_reset = b->auto_reset; a.horizon = b->horizon; a.step_index = b->step_index;
  a.N = b->N; a.P = b->m->P; a.T = b->m->T; a.stats = b->stats;
  a.epb = b->epb; a.timers = b->timers; a.tline = b->tline; a.nfused = 1;
  // speculate / replay: every family but the generic one has a replay kernel
  if (b->replay && family_of(b) >= 0 && family_of(b) != 6) { a.replay_list = b->replay_list; a.replay_ctl = b->replay_ctl; a.stall = b->stall; a.replay_mark = b->replay_mark;
    static const bool no_resume = LM_PROBE_ENV("LM_NO_RESUME") != nullptr;       // A/B: restart abandoned control steps from their own state (round 4)
    if (!no_resume) { a.hq = b->hq; a.hv = b->hv; a.hw = b->hw; }
    a.hsub = b->hsub;
    static const bool no_premark = LM_PROBE_ENV("LM_NO_PREMARK") != nullptr;       // A/B: every control step starts in the regular kernel (round 4)
    if (!no_premark) {
      a.premark = b->premark;
      // the regular kernels' capacity per chain (lm_family.hip / lm_core.h LaneMem: contact slots, queued convex pairs, pair results)
      const Task& T = b->m->T;
      const int fam = family_of(b);
      a.reg_ns = fam == 0 ? 6 : (fam == 5 ? 4 : 8); a.reg_q = T.max_links >= 5 ? 24 : 8; a.reg_r = a.reg_ns < 8 ? a.reg_ns : 8;
    } a.replay_all = b->replay == 2 || b->replay == 4; }
  static const bool no_xcd_map = LM_PROBE_ENV("LM_NO_XCD_MAP") != nullptr;
  a.xcd_map = no_xcd_map ? 0 : 1;
  return a;
}

static void compile_models(lm_batch* b, const unsigned char* mask, int all) {
  lmc::Args c;
  c.ib = b->mc_ib; c.db = b->mc_db; c.N = b->N; c.seed = b->mc_seed; c.env_offset = b->env_offset; c.dirty = b->vdirty; c.mask = mask; c.all = all;
  c.gen = b->vgen; c.vrec = b->vrec; c.vgt = b->vgt; c.vgpt = b->vgpt; c.gpt_floats = b->gpt_floats; c.slack = b->slack; c.draws = b->vdraws;
  lmc::launch(c, b->stream);
}

static void launch_step(lm_batch* b, const KArgs& a) {
  // the per-thread HIP error state is shared with whoever else uses HIP in this process (PyTorch probes peers, pointer
  // attributes ...): drop what they left behind so that the check after the launch reports OUR launch
  (void)hipGetLastError();
  g_launch_err = nullptr;
  launch_variant<false>(b, a);
  // the environments that restarted an episode in this launch get their fresh model before the next one (stream order)
  if (b->mc_ib && b->auto_reset && b->table_rows > 0) compile_models(b, nullptr, 0);
}

static int drain_stats(lm_batch* b) {
  std::vector<DevStats> s(b->nstat);
  HIPCHK(hipMemcpyAsync(s.data(), b->stats, sizeof(DevStats) * b->nstat, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipMemsetAsync(b->stats, 0, sizeof(DevStats) * b->nstat, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  for (const DevStats& x : s) {
    b->acc.env_steps += x.env_steps; b->acc.episodes += x.episodes; b->acc.reward_sum += x.reward_sum;
    b->acc.nan_resets += x.nan_resets; b->acc.solver_iters += x.solver_iters; b->acc.overflow_contacts += x.overflow;
    b->acc.unhandled_geoms += x.unhandled; b->acc.linesearch_evals += x.ls_evals; b->acc.linesearch_capped += x.ls_capped; b->acc.steps_with_8plus_iters += x.it_ge8;
    b->acc.self_proximity += x.selfprox; b->acc.self_contacts += x.selfcon; b->acc.replayed_env_steps += x.replayed; b->acc.own_manifold_contacts += x.natown;
  }
  return 0;
}

int lm_batch_set_active(lm_batch* b, const int32_t* env_ids, int count) {
  HIPCHK(hipSetDevice(b->m->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  if (!env_ids) { b->n_active = b->N; if (b->env_map) { HIPCHK(hipFree(b->env_map)); b->env_map = nullptr; } return 0; }
  if (family_of(b) == 0) return fail("active lists are not compiled into the quadruped's kernels (lm_step.h: the indirection costs the bench kernel 0.9 %)");
  if (count < 0 || count > b->N) return fail("active list: more entries than environments");
  std::vector<char> seen((size_t)b->N, 0);
  for (int i = 0; i < count; i++) {
    if (env_ids[i] < 0 || env_ids[i] >= b->N) return fail("active list: environment id out of range");
    if (seen[env_ids[i]]) return fail("active list: an environment is listed twice");
    seen[env_ids[i]] = 1;
  }
  if (!b->env_map) HIPCHK(hipMalloc(&b->env_map, sizeof(int) * (size_t)b->N));
  if (count > 0) HIPCHK(hipMemcpy(b->env_map, env_ids, sizeof(int) * (size_t)count, hipMemcpyHostToDevice));
  b->n_active = count;
  return 0;
}

#ifndef LM_TOOLCHAIN
#define LM_TOOLCHAIN "unknown (built outside csrc/Makefile)"
#endif
const char* lm_toolchain(void) { return LM_TOOLCHAIN; }

int lm_step(lm_batch* b, const float* action, float* obs, float* reward, uint8_t* done) {
  HIPCHK(hipSetDevice(b->m->device));
  const int N = b->N; const Task& T = b->m->T;
  KArgs a = make_args(b);
  if (action) { HIPCHK(hipMemcpyAsync(b->action, action, sizeof(float) * T.nu * N, hipMemcpyHostToDevice, b->stream)); a.action = b->action; a.action_mode = 0; }
  else a.action_mode = 1;
  a.obs = b->obs; a.reward = b->reward; a.done = b->done;
  launch_step(b, a);
  if (g_launch_err) return fail(g_launch_err);
  HIPCHK(hipGetLastError());
  b->step_index++;
  if (obs) HIPCHK(hipMemcpyAsync(obs, b->obs, sizeof(float) * T.nobs * N, hipMemcpyDeviceToHost, b->stream));
  if (reward) HIPCHK(hipMemcpyAsync(reward, b->reward, sizeof(float) * N, hipMemcpyDeviceToHost, b->stream));
  if (done) HIPCHK(hipMemcpyAsync(done, b->done, N, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return 0;
}

// ---- the float64 host surface (include/locohip.h lm_step_pinned)
__global__ void pack_out64_kernel(const float* __restrict__ obs, const float* __restrict__ reward, const unsigned char* __restrict__ done,
                                  const int* __restrict__ perm, int N, int nobs, double* __restrict__ o64, double* __restrict__ r64,
                                  unsigned char* __restrict__ d8) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, total = N * nobs;
  if (i < total) {
    const int e = i / nobs, j = i - e * nobs;
    o64[i] = (double)obs[e * nobs + (perm ? perm[j] : j)];
  }
  if (i < N) { r64[i] = (double)reward[i]; d8[i] = done[i]; }
}

static int pinned_alloc(lm_batch* b) {
  if (b->h_act) return 0;
  const size_t N = (size_t)b->N, nobs = (size_t)b->m->T.nobs;
  b->out64_bytes = sizeof(double) * (N * nobs + N) + N;
  HIPCHK(hipHostMalloc((void**)&b->h_act, sizeof(float) * N * (size_t)b->m->T.nu, hipHostMallocDefault));
  for (int i = 0; i < LM_PINNED_SLOTS; i++) {
    HIPCHK(hipHostMalloc((void**)&b->h_out64[i], b->out64_bytes, hipHostMallocDefault));
    memset(b->h_out64[i], 0, b->out64_bytes);
  }
  return 0;
}

int lm_pinned_slot(lm_batch* b, int slot, double** obs, double** reward, uint8_t** done) {
  HIPCHK(hipSetDevice(b->m->device));
  if (slot < 0 || slot >= LM_PINNED_SLOTS) return fail("pinned slot out of range");
  if (pinned_alloc(b)) return 1;
  const size_t N = (size_t)b->N, nobs = (size_t)b->m->T.nobs;
  double* base = reinterpret_cast<double*>(b->h_out64[slot]);
  if (obs) *obs = base;
  if (reward) *reward = base + N * nobs;
  if (done) *done = reinterpret_cast<uint8_t*>(base + N * nobs + N);
  return 0;
}

int lm_set_obs_order(lm_batch* b, const int32_t* perm, int n) {
  HIPCHK(hipSetDevice(b->m->device));
  if (b->d_perm) { HIPCHK(hipStreamSynchronize(b->stream)); HIPCHK(hipFree(b->d_perm)); b->d_perm = nullptr; }
  if (!perm) return 0;
  const int nobs = b->m->T.nobs;
  if (n != nobs) return fail("observation order: one entry per observation column");
  for (int j = 0; j < n; j++) if (perm[j] < 0 || perm[j] >= nobs) return fail("observation order: column out of range");
  HIPCHK(hipMalloc(&b->d_perm, sizeof(int) * n));
  HIPCHK(hipMemcpy(b->d_perm, perm, sizeof(int) * n, hipMemcpyHostToDevice));
  return 0;
}

int lm_step_pinned(lm_batch* b, const double* action, int slot) {
  HIPCHK(hipSetDevice(b->m->device));
  if (slot < 0 || slot >= LM_PINNED_SLOTS) return fail("pinned slot out of range");
  if (!action) return fail("lm_step_pinned needs an action (policy-free rollouts: lm_rollout)");
  if (pinned_alloc(b)) return 1;
  const int N = b->N; const Task& T = b->m->T;
  const size_t na = (size_t)N * T.nu;
  for (size_t i = 0; i < na; i++) b->h_act[i] = (float)action[i];
  KArgs a = make_args(b);
  // the step kernel reads the action out of the pinned staging buffer itself and the conversion kernel writes the pinned slot itself
  // (both mapped into the device's address space): no copy is queued on either side of the launch. Measured on one box against an
  // H2D copy in front and a D2H copy behind (tools/probes/r6/surface.py, 4096 quadrupeds): 1.258 against 1.274 ms per LocoEnv.step()
  a.action = b->h_act;
  a.action_mode = 0;
  a.obs = b->obs; a.reward = b->reward; a.done = b->done;
  launch_step(b, a);
  if (g_launch_err) return fail(g_launch_err);
  b->step_index++;
  double* o64 = reinterpret_cast<double*>(b->h_out64[slot]);
  const int total = N * T.nobs, threads = 256;
  hipLaunchKernelGGL(pack_out64_kernel, dim3((total + threads - 1) / threads), dim3(threads), 0, b->stream, b->obs, b->reward, b->done,
                     b->d_perm, N, T.nobs, o64, o64 + (size_t)N * T.nobs, reinterpret_cast<unsigned char*>(o64 + (size_t)N * T.nobs + N));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(b->stream));
  return 0;
}

int lm_step_device(lm_batch* b, const float* d_action, float* d_obs, float* d_reward, uint8_t* d_done, void* stream, int sync) {
  HIPCHK(hipSetDevice(b->m->device));
  KArgs a = make_args(b);
  if (d_action) { a.action = d_action; a.action_mode = 0; } else a.action_mode = 1;
  a.obs = d_obs ? d_obs : b->obs; a.reward = d_reward ? d_reward : b->reward; a.done = d_done ? d_done : b->done;
  hipStream_t own = b->stream;
  hipStream_t used = stream ? (hipStream_t)stream : own;       // the caller's stream (e.g. torch's current stream)
  // a launch on a foreign stream is ordered on BOTH sides against the library's own stream: it waits for what the
  // library has queued (state uploads, earlier steps), and whatever the library queues next (lm_get_state, lm_get_stats,
  // lm_rollout ...) waits for it
  if (used != own) { HIPCHK(hipEventRecord(b->ev_ext, own)); HIPCHK(hipStreamWaitEvent(used, b->ev_ext, 0)); }
  b->stream = used;
  launch_step(b, a);
  b->stream = own;
  if (g_launch_err) return fail(g_launch_err);
  HIPCHK(hipGetLastError());
  if (used != own) { HIPCHK(hipEventRecord(b->ev_ext, used)); HIPCHK(hipStreamWaitEvent(own, b->ev_ext, 0)); }
  b->step_index++;
  if (sync) HIPCHK(hipStreamSynchronize(used));
  return 0;
}

int lm_set_reset_table(lm_batch* b, const float* rows, int n_rows, uint64_t seed, int64_t global_env_offset) {
  HIPCHK(hipSetDevice(b->m->device));
  const Task& T = b->m->T;
  const size_t w = 2 * T.nv + T.ngoal;
  if (n_rows <= 0) return fail("empty reset table");
  if (b->table) { HIPCHK(hipFree(b->table)); b->table = nullptr; }
  HIPCHK(hipMalloc(&b->table, sizeof(float) * w * n_rows));
  HIPCHK(hipMemcpy(b->table, rows, sizeof(float) * w * n_rows, hipMemcpyHostToDevice));
  b->table_rows = n_rows; b->seed = seed; b->env_offset = global_env_offset;
  b->var_rows = 0;                 // a new table: the variant no longer follows the row until lm_set_variant_rows says so
  return 0;
}

int lm_set_auto_reset(lm_batch* b, int enabled, int horizon) {
  if (enabled && b->table_rows <= 0) return fail("auto reset needs a reset table (lm_set_reset_table)");
  b->auto_reset = enabled; b->horizon = horizon;
  return 0;
}

int lm_rollout_fused(lm_batch* b, int n_steps, int steps_per_launch, int action_mode, uint64_t seed, lm_stats* stats) {
  HIPCHK(hipSetDevice(b->m->device));
  if (action_mode != 0 && action_mode != 1) return fail("action_mode must be 0 (zero) or 1 (uniform random)");
  if (steps_per_launch < 1) return fail("steps_per_launch must be >= 1");
  static const bool no_replicas = LM_PROBE_ENV("LM_NO_REPLICAS") != nullptr;
  if (b->epb > 4 || no_replicas || !family_has_replicas(b)) steps_per_launch = 1;     // no fused kernels for the full-wave layout
  if (b->mc_ib) steps_per_launch = 1;       // a restart inside a launch needs its fresh model before the episode's first step
  KArgs a = make_args(b);
  a.action = nullptr; a.action_mode = action_mode == 0 ? 1 : 2;   // kernel: 1 = zero action, 2 = random
  a.seed = b->seed ^ (seed * 0x9E3779B97F4A7C15ull);
  a.obs = b->obs; a.reward = b->reward; a.done = b->done;
  HIPCHK(hipEventRecord(b->ev0, b->stream));
  for (int s = 0; s < n_steps; s += steps_per_launch) {
    a.nfused = (n_steps - s < steps_per_launch) ? n_steps - s : steps_per_launch;
    a.step_index = b->step_index; b->step_index += (unsigned)a.nfused;
    launch_step(b, a);
    if (g_launch_err) return fail(g_launch_err);
  }
  HIPCHK(hipEventRecord(b->ev1, b->stream));
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventSynchronize(b->ev1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
  if (drain_stats(b)) return 1;
  b->acc.kernel_ms += ms;
  if (stats) { *stats = b->acc; stats->kernel_ms = ms; }
  return 0;
}

int lm_rollout(lm_batch* b, int n_steps, int action_mode, uint64_t seed, lm_stats* stats) {
  return lm_rollout_fused(b, n_steps, 1, action_mode, seed, stats);
}

int lm_forward_debug(lm_batch* b, const float* action, lm_forward_out* out) {
  HIPCHK(hipSetDevice(b->m->device));
  const int N = b->N; const Task& T = b->m->T; const int nv = T.nv;
  KArgs a = make_args(b);
  a.stats = nullptr; a.replay_list = nullptr;
  if (action) { HIPCHK(hipMemcpy(b->action, action, sizeof(float) * T.nu * N, hipMemcpyHostToDevice)); a.action = b->action; a.action_mode = 0; }
  else a.action_mode = 1;
  float* buf; int* ibuf;
  const size_t per = (size_t)nv * nv + 5 * nv;
  HIPCHK(hipMalloc(&buf, sizeof(float) * per * N)); HIPCHK(hipMalloc(&ibuf, sizeof(int) * 2 * N));
  HIPCHK(hipMemset(buf, 0, sizeof(float) * per * N));
  a.dM = buf; a.dbias = buf + (size_t)nv * nv * N; a.dsmooth = a.dbias + (size_t)nv * N; a.dqacc_smooth = a.dsmooth + (size_t)nv * N;
  a.dqacc = a.dqacc_smooth + (size_t)nv * N; a.dqfrc = a.dqacc + (size_t)nv * N; a.dncon = ibuf; a.diter = ibuf + N;
  (void)hipGetLastError();
  g_launch_err = nullptr;
  launch_variant<true>(b, a);
  if (g_launch_err) return fail(g_launch_err);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(b->stream));
  auto get = [&](float* dst, const float* src, size_t n) -> int { if (dst) HIPCHK(hipMemcpy(dst, src, sizeof(float) * n, hipMemcpyDeviceToHost)); return 0; };
  if (get(out->M, a.dM, (size_t)nv * nv * N) || get(out->qfrc_bias, a.dbias, (size_t)nv * N) || get(out->qfrc_smooth, a.dsmooth, (size_t)nv * N) ||
      get(out->qacc_smooth, a.dqacc_smooth, (size_t)nv * N) || get(out->qacc, a.dqacc, (size_t)nv * N) || get(out->qfrc_constraint, a.dqfrc, (size_t)nv * N)) return 1;
  if (out->ncon) HIPCHK(hipMemcpy(out->ncon, a.dncon, sizeof(int) * N, hipMemcpyDeviceToHost));
  if (out->solver_iter) HIPCHK(hipMemcpy(out->solver_iter, a.diter, sizeof(int) * N, hipMemcpyDeviceToHost));
  HIPCHK(hipFree(buf)); HIPCHK(hipFree(ibuf));
  return 0;
}

int lm_get_flags(lm_batch* b, uint8_t* out) {
  HIPCHK(hipSetDevice(b->m->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(out, b->flags, b->N, hipMemcpyDeviceToHost));
  return 0;
}

int lm_get_replay_marks(lm_batch* b, uint8_t* out, int reset) {
  if (!b) return fail("null batch");
  HIPCHK(hipSetDevice(b->m->device));
  // copy and clear ON the library's stream: it is ordered behind every launch (also one on a caller's stream: ev_ext) and behind the
  // replay kernel's pollers (ev_join), which write the marks
  if (out) HIPCHK(hipMemcpyAsync(out, b->replay_mark, b->N, hipMemcpyDeviceToHost, b->stream));
  if (reset) HIPCHK(hipMemsetAsync(b->replay_mark, 0, b->N, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return 0;
}

int lm_get_stats(lm_batch* b, lm_stats* out, int reset) {
  HIPCHK(hipSetDevice(b->m->device));
  if (drain_stats(b)) return 1;
  if (out) *out = b->acc;
  if (reset) memset(&b->acc, 0, sizeof(b->acc));
  return 0;
}

#ifdef LM_TIMERS      // (the shipped library exports nothing that include/locohip.h does not declare: tests/test_abi_exports.py)
/* profiling builds (-DLM_TIMERS): cycles spent per solver region, summed over workgroups (not part of the ABI header) */
int lm_debug_timers(lm_batch* b, unsigned long long* out16) {
  HIPCHK(hipSetDevice(b->m->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(out16, b->timers, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(b->timers, 0, sizeof(unsigned long long) * 16));
  return 0;
}

/* profiling builds: per workgroup of the LAST launch [nblocks][16] = cycles, then per environment (4) solver iterations,
   contact slots (summed over passes), line-search evaluations */
int lm_debug_wg_records(lm_batch* b, unsigned long long* out, int nblocks) {
  HIPCHK(hipSetDevice(b->m->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  if (nblocks != b->nblocks) return fail("nblocks mismatch");
  HIPCHK(hipMemcpy(out, b->timers + 16, sizeof(unsigned long long) * 16 * (size_t)nblocks, hipMemcpyDeviceToHost));
  return 0;
}

/* profiling builds: per workgroup of the LAST launch [nblocks][16] = cycles per solver region */
int lm_debug_wg_regions(lm_batch* b, unsigned long long* out, int nblocks) {
  HIPCHK(hipSetDevice(b->m->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  if (nblocks != b->nblocks) return fail("nblocks mismatch");
  HIPCHK(hipMemcpy(out, b->timers + 16 + 16 * (size_t)nblocks, sizeof(unsigned long long) * 16 * (size_t)nblocks, hipMemcpyDeviceToHost));
  return 0;
}

/* profiling builds: wall-clock time line of the last launch(es) since the last call: [N][4] + [nblocks][2] (lm_step.h KArgs::tline), cleared */
int lm_debug_timeline(lm_batch* b, unsigned long long* out) {
  HIPCHK(hipSetDevice(b->m->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  const size_t n = 4 * (size_t)b->N + 2 * (size_t)b->nblocks;
  HIPCHK(hipMemcpy(out, b->tline, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(b->tline, 0, sizeof(unsigned long long) * n));
  return 0;
}

/* profiling builds: counters of the pair pass and the convex collider since the last call (16 values, summed over all lanes) */
int lm_debug_mpr_counters(lm_batch* b, unsigned long long* out8 /* 16 values */) {
  HIPCHK(hipSetDevice(b->m->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(out8, b->timers + 16 + 32 * (size_t)b->nblocks, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(b->timers + 16 + 32 * (size_t)b->nblocks, 0, sizeof(unsigned long long) * 16));
  return 0;
}

#endif

int lm_sync(lm_batch* b) {
  HIPCHK(hipSetDevice(b->m->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  return 0;
}

}  // extern "C"
