// TEST TOOLING ONLY — CPU emulation of one quad of GPU lanes, so that the device code in
// loco_mujoco_amd/csrc/lm_core.h can be debugged against the fp64 oracle in a container without a GPU.
// Four OS threads play the four lanes; the quad sum (two DPP adds on gfx950) becomes a barrier + the same
// (x0+x1)+(x2+x3) association. Nothing in the product loads this file; it is built by tests/test_emu_core.py.
#include <atomic>
#include <barrier>
#include <chrono>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <thread>
#include <type_traits>
#include <vector>
#define LM_DEV inline
#define LM_OPAQUE_ZERO() 0
#define LM_POW01(x, p) exp2f((p) * log2f(x))
#include "../../loco_mujoco_amd/csrc/lm_core.h"

#ifndef EMU_SIX_PAIRS
#define EMU_SIX_PAIRS 1
#endif
#ifndef EMU_LS_POINTS
#define EMU_LS_POINTS 1
#endif
#ifndef EMU_REP
#define EMU_REP 1
#endif
namespace {
// EMU_REP = 4 plays the replicated small-batch layout: 16 OS threads per environment = 4 replicas x 4 chain lanes. On the
// GPU the replicas of a lane share one column of lane memory and run in lock step. Here every replica has a PRIVATE copy
// of its chain's lane memory and the copies are reconciled only inside Q::fence(): whatever a replica changed since the
// previous fence is handed to the others (two replicas changing the same word to different values is an error). A
// hand-over that the device code does not bracket with a fence therefore reads stale data here and fails the parity
// tests — that is how the fence placement is checked without a GPU.
constexpr int kThreads = 4 * EMU_REP;
std::barrier<> g_bar(kThreads);      // everybody: any(), fence(), start / end of an environment
// the device's cross-lane operations have different participants: a quad sum joins the 4 chain lanes of ONE replica, the
// replica operations join the EMU_REP replicas of ONE chain lane (lanes with fewer geoms / links skip them altogether)
struct QuadBarrier { std::barrier<> b{4}; void arrive_and_wait() { b.arrive_and_wait(); } };
QuadBarrier g_bar_rep[EMU_REP];      // one per replica (quad)
std::barrier<> g_bar_lane[4] = {std::barrier<>(EMU_REP), std::barrier<>(EMU_REP), std::barrier<>(EMU_REP), std::barrier<>(EMU_REP)};
float g_buf[EMU_REP][4], g_rbuf[EMU_REP][4];
int g_ibuf[kThreads];
float* g_lmem[EMU_REP][4];          // private lane memory of (replica, chain)
std::vector<float> g_snap[4];       // per chain: lane memory as of the last fence
int g_lmem_size = 0, g_conflicts = 0;
thread_local int t_lane = 0, t_rep = 0;
int g_ops[64][5];                   // per thread: calls of sum, any, rep_bcast, rep_sum, fence (EMU_WATCHDOG diagnostics)
#define OPC(k) (g_ops[t_rep * 4 + t_lane][k]++)
template <int POINTS, int REP>
struct QuadThreadsT {
  static constexpr int kRep = REP, kPoints = (REP >= 4) ? 4 : POINTS;
  using mask_t = std::conditional_t<(REP > 8), unsigned long long, unsigned>;       // one bit per thread of the environment (lm_step.h QuadDppT)
  static int rep() { return t_rep; }
  static float rep_bcast(float x, int r) {
    if (REP == 1) return x;
    OPC(2);
    g_rbuf[t_rep][t_lane] = x; g_bar_lane[t_lane].arrive_and_wait();
    float y = g_rbuf[r][t_lane];
    g_bar_lane[t_lane].arrive_and_wait(); return y;
  }
  static float rep_sum(float x) {       // the device's association: four replicas (DPP row rotations) replica^2 then replica^1; sixteen: the
    if (REP == 1) return x;             // ds_bpermute butterfly replica^1, ^2, ^4, ^8
    OPC(3);
    float y = x;
    constexpr int kStages = (REP >= 16) ? 4 : (REP >= 8 ? 3 : (REP >= 4 ? 2 : 1));
    for (int k = 0; k < kStages; k++) {
      const int bit = (REP == 4) ? (2 >> k) : (1 << k);
      g_rbuf[t_rep][t_lane] = y; g_bar_lane[t_lane].arrive_and_wait();
      const float o = g_rbuf[t_rep ^ bit][t_lane];
      g_bar_lane[t_lane].arrive_and_wait();
      y = y + o;
    }
    return y;
  }
  static void fence() {
    if (REP == 1) return;
    OPC(4);
    g_bar.arrive_and_wait();
    if (t_lane == 0 && t_rep == 0) {
      for (int c = 0; c < 4; c++) for (int i = 0; i < g_lmem_size; i++) {
        const float old = g_snap[c][i];
        bool have = false; float nv = old;
        for (int r = 0; r < REP; r++) {
          const float v = g_lmem[r][c][i];
          if (memcmp(&v, &old, 4) == 0) continue;
          if (have && memcmp(&v, &nv, 4) != 0) g_conflicts++;
          have = true; nv = v;
        }
        if (have) { for (int r = 0; r < REP; r++) g_lmem[r][c][i] = nv; g_snap[c][i] = nv; }
      }
    }
    g_bar.arrive_and_wait();
  }
  static float sum(float x) {
    OPC(0);
    g_buf[t_rep][t_lane] = x; g_bar_rep[t_rep].arrive_and_wait();
    float s = (g_buf[t_rep][0] + g_buf[t_rep][1]) + (g_buf[t_rep][2] + g_buf[t_rep][3]);
    g_bar_rep[t_rep].arrive_and_wait(); return s;
  }
  // the lanes of one quad (replica): peers' lane memory, values of a named lane, a join before / after peer reads
  static void quad_sync() { g_bar_rep[t_rep].arrive_and_wait(); }
  static float peer(const float* /*lmem*/, int /*ls*/, int i, int dl) { return g_lmem[t_rep][t_lane + dl][i]; }
  static void peer_write(float* /*lmem*/, int /*ls*/, int i, int dl, float v) { g_lmem[t_rep][t_lane + dl][i] = v; }
  static mask_t env_ballot(bool b) {      // bit 4 * replica + chain of every thread of the environment
    g_ibuf[t_rep * 4 + t_lane] = b; g_bar.arrive_and_wait();
    mask_t r = 0;
    for (int i = 0; i < kThreads; i++) r |= (mask_t)(g_ibuf[i] ? 1 : 0) << i;
    g_bar.arrive_and_wait(); return r;
  }
  static float quad_read(float x, int src) {
    g_buf[t_rep][t_lane] = x; g_bar_rep[t_rep].arrive_and_wait();
    float y = g_buf[t_rep][src];
    g_bar_rep[t_rep].arrive_and_wait(); return y;
  }
  static bool any(bool b) {
    OPC(1);
    g_ibuf[t_rep * 4 + t_lane] = b; g_bar.arrive_and_wait();
    bool r = false;
    for (int i = 0; i < kThreads; i++) r = r || g_ibuf[i];
    g_bar.arrive_and_wait(); return r;
  }
};
using QuadThreads = QuadThreadsT<EMU_LS_POINTS, EMU_REP>;
// EMU_DR compiles the per-environment joint-parameter path (DR = true); the parameters come from emu_set_dof_params
// ([3][n][nv]: damping, stiffness, frictionloss) or, when none are set, from the constant table (must change nothing)
#ifdef EMU_DR
constexpr int kEmuDR = EMU_DR;         // 1: per-environment joint parameters, 2: + a model variant (emu_set_model_variant)
#else
constexpr int kEmuDR = 0;
#endif
const double* g_dofprm = nullptr;
// one model variant for every environment (lowering.variant_tables): inertial record, geom table, geom-pair table (or null)
const float* g_vrec = nullptr; const float* g_vgt = nullptr; const float* g_vgpt = nullptr;
}  // namespace

// EMU_PYRAMID_ONLY compiles the humanoid families like the library does (condim-3 pyramids only, elliptic code out)
#ifdef EMU_PYRAMID_ONLY
template <int MC> constexpr int kEmuCone = (MC >= 5) ? 0 : -1;
#else
template <int MC> constexpr int kEmuCone = -1;
#endif

// chain_model: float64 [HEADER + CM]; state arrays [n][nv] double in/out; ctrl [n][nu] (already un-normalised)
// act: muscle activations [n][na] double in/out (NM > 0 only)
// PM: the pair pass of the kernel (lm_core.h forward: 0 none, 1 with the convex collider, 2 without it).
// leave / only: the device's speculate / replay protocol (lm_step.h). With `leave` a control step that runs out of contact slots or
// list space, or meets a convex pair without having the collider, stores NOTHING and sets leave[e]; with `only` the environments
// whose flag is 0 are skipped (the replay pass of the big instantiation over the flagged ones).
template <int MC, int NS, bool RK4, int NM = 0, int PM = 0>
static int emu_run_t(const double* chain_model, int n, double* qpos, double* qvel, double* warm, const double* action,
                     int nsub, int debug_env, float* dbgM, float* dbg5, int* counters, double* act = nullptr,
                     int* leave = nullptr, const int* only = nullptr) {
  constexpr bool PAIRS = PM == 1 || PM == 2;      // (3: detection only — the lane memory of a kernel without the pair pass)
  const double* H = chain_model;
  const int nv = (int)H[LM_H_NV], nu = (int)H[LM_H_NU];
  std::vector<float> cm(LM_CM_SIZE);
  for (int i = 0; i < LM_CM_SIZE; i++) cm[i] = (float)H[LM_HEADER_SIZE + i];
  const int na = (int)H[LM_H_NMUSCLE];
  std::vector<float> mt(NM > 0 ? LM_MT_SIZE : 1);
  if (NM > 0) for (int i = 0; i < LM_MT_SIZE; i++) mt[i] = (float)H[LM_HEADER_SIZE + LM_CM_SIZE + LM_GT_SIZE + i];
  std::vector<float> gt(LM_GT_SIZE);
  for (int i = 0; i < LM_GT_SIZE; i++) gt[i] = (float)H[LM_HEADER_SIZE + LM_CM_SIZE + i];
  lm::Params P;
  P.h = (float)H[LM_H_TIMESTEP]; P.g = lm::v3((float)H[LM_H_GX], (float)H[LM_H_GY], (float)H[LM_H_GZ]);
  P.iterations = (int)H[LM_H_ITERATIONS]; P.tolerance = 1e-6f; P.nv = nv;
  P.scale = 1.0f / ((float)H[LM_H_MEANINERTIA] * nv);
  P.ls_tol = 1e-2f; P.ls_iters = 12; P.ls_noise = 2e-6f; P.ablate = 0;
  P.root_xyz = cm[LM_R_NDOF] == 6.0f;        // as lm_kernels.hip lm_model_create
  for (int i = 0; i < 9; i++) if (cm[LM_R_R0 + i] != ((i % 4 == 0) ? 1.0f : 0.0f)) P.root_xyz = 0;
  for (int i = 0; i < 3; i++) {
    const float* d = cm.data() + LM_R_DOFS + i * LM_D_SIZE;
    if (d[LM_D_TYPE] != 0.0f) P.root_xyz = 0;
    for (int k = 0; k < 3; k++) if (d[LM_D_AX + k] != ((k == i) ? 1.0f : 0.0f)) P.root_xyz = 0;
  }
  P.root_limited = 0;
  for (int i = 0; i < 6; i++) if (cm[LM_R_DOFS + i * LM_D_SIZE + LM_D_LIMITED] != 0.0f) P.root_limited = 1;
  P.ls_grid[0] = 0.25f; P.ls_grid[1] = 0.0625f; P.ls_grid[2] = 0.015625f;
  if (const char* v = getenv("LM_LS_GRID")) sscanf(v, "%f,%f,%f", &P.ls_grid[0], &P.ls_grid[1], &P.ls_grid[2]);   // A/B knob
  P.integrator = (int)H[LM_H_INTEGRATOR]; P.cone = (int)H[LM_H_CONE]; P.act_position = (int)H[LM_H_ACTMODE];
  P.off_runsup = (int)H[LM_H_OFF_RUNSUP]; P.gt = gt.data(); P.cmg = cm.data();
  std::vector<float> gpt((size_t)H[LM_H_NGPAIR] * LM_GPAIR_SIZE + 1);
  for (size_t i = 0; i + 1 < gpt.size(); i++) gpt[i] = (float)H[(size_t)H[LM_H_OFF_GPT] + i];
  P.gpt = gpt.data();
  std::vector<float> meshv(4 * (size_t)H[LM_H_NMESHV] + 4);
  for (size_t i = 0; i + 4 < meshv.size(); i++) meshv[i] = (float)H[(size_t)H[LM_H_OFF_MESHV] + i];
  P.meshv = meshv.data();
  std::vector<float> meshn((size_t)H[LM_H_NMESHN] + 1, -1.0f);
  for (size_t i = 0; i + 1 < meshn.size(); i++) meshn[i] = (float)H[(size_t)H[LM_H_OFF_MESHN] + i];
  P.meshn = meshn.data();
  std::vector<float> bpt((size_t)H[LM_H_NBPAIR] * LM_BP_SIZE + 1);
  for (size_t i = 0; i + 1 < bpt.size(); i++) bpt[i] = (float)H[(size_t)H[LM_H_OFF_BPT] + i];
  P.bpt = bpt.data();
  std::vector<float> madj(4 * (size_t)H[LM_H_NMESHADJ] + 64 + 4, 0.0f);
  float* madj_al = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(madj.data()) + 15) & ~uintptr_t(15));
  for (size_t i = 0; i < 4 * (size_t)H[LM_H_NMESHADJ]; i++) madj_al[i] = (float)H[(size_t)H[LM_H_OFF_MESHADJ] + i];
  P.meshadj = madj_al;
  int cnt_tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto lane_main = [&](int t) {
    const int c = t & 3;
    t_lane = c; t_rep = t >> 2;
    const float* rb = cm.data();
    for (int e = 0; e < n; e++) {
      if (only && !only[e]) continue;
      float qr[6], vr[6], war[6], actr[6], qc[MC], vc[MC], wac[MC], actc[MC];
      int dr[6], dc[MC];
      const int nl = (int)cm[LM_CM_CHAINS + LM_C_NLINKS * LM_NCHAIN + c];
      auto actuate = [&](const float* blk, int stride) -> float {
        int k = (int)blk[LM_D_ACT * stride];
        if (k < 0) return 0.0f;
        float ctrl = (float)action[e * nu + k] * blk[LM_D_ACT_DELTA * stride] + blk[LM_D_ACT_MEAN * stride];
        ctrl = fminf(fmaxf(ctrl, blk[LM_D_CTRL_LO * stride]), blk[LM_D_CTRL_HI * stride]);
        return blk[LM_D_GEAR * stride] * ctrl;
      };
      for (int i = 0; i < 6; i++) {
        const float* blk = rb + LM_R_DOFS + i * LM_D_SIZE;
        dr[i] = (int)blk[LM_D_DOF];
        qr[i] = (float)qpos[e * nv + dr[i]]; vr[i] = (float)qvel[e * nv + dr[i]]; war[i] = (float)warm[e * nv + dr[i]];
        actr[i] = actuate(blk, 1);
      }
      for (int k = 0; k < MC; k++) {
        qc[k] = vc[k] = wac[k] = actc[k] = 0; dc[k] = -1;
        if (k < nl) {
          const float* blk = cm.data() + LM_CM_CHAINS + (LM_C_LINKS + k * LM_LINK_SIZE) * LM_NCHAIN + c;
          dc[k] = (int)blk[LM_D_DOF * LM_NCHAIN];
          qc[k] = (float)qpos[e * nv + dc[k]]; vc[k] = (float)qvel[e * nv + dc[k]]; wac[k] = (float)warm[e * nv + dc[k]];
          actc[k] = actuate(blk, LM_NCHAIN);
        }
      }
      lm::DofPrm<MC> dofp = {};
      std::vector<float> dof_ds;
      if (kEmuDR) {
        auto prm = [&](int which, int dof, float nominal) -> float {
          return g_dofprm ? (float)g_dofprm[((size_t)which * n + e) * nv + dof] : nominal;
        };
        // damping | stiffness of this environment by dof index (the kernel reads them where they are used)
        std::vector<float>& ds = dof_ds;
        ds.assign(2 * (size_t)nv, 0.0f);
        for (int i = 0; i < 6; i++) {
          const float* blk = rb + LM_R_DOFS + i * LM_D_SIZE;
          ds[dr[i]] = prm(0, dr[i], blk[LM_D_DAMP]); ds[nv + dr[i]] = prm(1, dr[i], blk[LM_D_STIFF]); dofp.floss_r[i] = prm(2, dr[i], blk[LM_D_FLOSS]);
        }
        for (int k = 0; k < MC; k++) if (k < nl) {
          const float* blk = cm.data() + LM_CM_CHAINS + (LM_C_LINKS + k * LM_LINK_SIZE) * LM_NCHAIN + c;
          ds[dc[k]] = prm(0, dc[k], blk[LM_D_DAMP * LM_NCHAIN]); ds[nv + dc[k]] = prm(1, dc[k], blk[LM_D_STIFF * LM_NCHAIN]);
          dofp.floss_c[k] = prm(2, dc[k], blk[LM_D_FLOSS * LM_NCHAIN]);
        }
        dofp.damp = ds.data(); dofp.stiff = ds.data() + nv; dofp.stride = 1;
        dofp.inr = g_vrec; dofp.gt = g_vrec ? g_vgt : P.gt; dofp.gpt = (g_vrec && g_vgpt) ? g_vgpt : P.gpt;
        if (g_vrec) {
          for (int i = 0; i < 6; i++) dofp.rfl_r[i] = g_vrec[(LM_IR_ROOT_DOF + 3 * i + 2) * LM_NCHAIN + c];
          for (int k = 0; k < MC; k++) dofp.rfl_c[k] = g_vrec[(k * LM_IR_LINK + 12) * LM_NCHAIN + c];
        }
      }
      lm::Counters cnt = {};
      using LMm = lm::LaneMemFor<MC, NS, NM, PAIRS, kEmuCone<MC>>;
      float lmem[LMm::kSize];
      // rep = 1: lane memory starts uninitialised (MemorySanitizer sees reads of never-written words); replicated: every
      // private copy and the snapshot start as the same quiet NaN, so such a read poisons the result instead
      const float never = __builtin_nanf("0xbad");
      if (EMU_REP > 1) for (int i = 0; i < LMm::kSize; i++) lmem[i] = never;
      g_lmem[t_rep][c] = lmem;
      if (t_rep == 0) g_snap[c].assign(LMm::kSize, never);
      g_lmem_size = LMm::kSize;
      g_bar.arrive_and_wait();
      if (NM > 0) {          // this lane's muscles: activation state and control (un-normalised, clamped) into lane memory
        const int m0 = (int)mt[c], nm = (int)mt[LM_NCHAIN + c];
        for (int i = 0; i < nm; i++) {
          const float* rec = mt.data() + LM_MT_HEAD + (m0 + i) * LM_MU_SIZE;
          const int k = (int)rec[LM_MU_ACT];
          float ctrl = (k >= 0) ? (float)action[e * nu + k] * rec[LM_MU_ACT_DELTA] + rec[LM_MU_ACT_MEAN] : 0.0f;
          lmem[LMm::kCtrl + i] = fminf(fmaxf(ctrl, rec[LM_MU_CTRL_LO]), rec[LM_MU_CTRL_HI]);
          lmem[LMm::kAct + i] = (float)act[e * na + (int)rec[LM_MU_STATE]];
        }
        if (EMU_REP > 1 && t_rep == 0) for (int i = 0; i < LMm::kSize; i++) g_snap[c][i] = lmem[i];     // all replicas start identical
        g_bar.arrive_and_wait();
      }
      lm::Debug dbg = {dbgM, dbg5, dbg5 + nv, dbg5 + 2 * nv, dbg5 + 3 * nv, dbg5 + 4 * nv};
      float pair_slack[3] = {0.0f, 0.0f, 0.0f};
      for (int s = 0; s < nsub; s++)
        lm::substep<QuadThreads, MC, NS, RK4, (PAIRS && MC <= 3) ? 1 : kEmuCone<MC>, NM, kEmuDR, PM>(cm.data(), c, P, qr, vr, qc, vc, war, wac, actr, actc, lmem, 1, cnt,
                                                      (e == debug_env && s == 0 && dbgM && t_rep == 0) ? &dbg : nullptr, mt.data(), &dofp, false, pair_slack);
      QuadThreads::fence();          // like the kernel before it stores: the activations were updated by their owner replicas
      static int acc[64][9];
      {
        int* A = acc[t_rep * 4 + c];
        A[0] = cnt.solver_iters; A[1] = cnt.overflow; A[2] = cnt.unhandled; A[3] = cnt.ncon; A[4] = cnt.ls_evals; A[5] = cnt.ls_capped; A[6] = cnt.selfprox; A[7] = cnt.selfcon;
        A[8] = cnt.need_full;
      }
      if (getenv("EMU_PAIR_TRACE") && c == 0 && t_rep == 0) fprintf(stderr, "env %d: pair detection passes %d, slack at the end %.4f\n", e, cnt.pair_passes, pair_slack[0]);
      g_bar.arrive_and_wait();
      // the environment-wide vote of the step kernel: did the control step stay inside this instantiation's capacity?
      bool left = false;
      if (leave) for (int t2 = 0; t2 < kThreads; t2++) left = left || acc[t2][1] > 0 || acc[t2][8] > 0;
      if (!left && t_rep == 0) {
        if (NM > 0) {
          const int m0 = (int)mt[c], nm = (int)mt[LM_NCHAIN + c];
          for (int i = 0; i < nm; i++) act[e * na + (int)mt[LM_MT_HEAD + (m0 + i) * LM_MU_SIZE + LM_MU_STATE]] = lmem[LMm::kAct + i];
        }
        if (c == 0) for (int i = 0; i < 6; i++) { qpos[e * nv + dr[i]] = qr[i]; qvel[e * nv + dr[i]] = vr[i]; warm[e * nv + dr[i]] = war[i]; }
        for (int k = 0; k < MC; k++) if (dc[k] >= 0) { qpos[e * nv + dc[k]] = qc[k]; qvel[e * nv + dc[k]] = vc[k]; warm[e * nv + dc[k]] = wac[k]; }
      }
      g_bar.arrive_and_wait();
      if (c == 0 && t_rep == 0) {
        if (!left) for (int l = 0; l < 4; l++) for (int j = 0; j < 8; j++) cnt_tot[j] += acc[l][j];
        if (leave) leave[e] = left ? 1 : 0;
      }
      g_bar.arrive_and_wait();
    }
  };
  g_conflicts = 0;
  memset(g_ops, 0, sizeof(g_ops));
  std::atomic<bool> finished{false};
  std::thread watchdog([&] {            // EMU_WATCHDOG=seconds: on a hang print how many cross-lane calls every thread made
    const char* w = getenv("EMU_WATCHDOG");
    if (!w) return;
    for (int i = 0; i < atoi(w) * 10 && !finished; i++) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    if (finished) return;
    for (int t = 0; t < kThreads; t++)
      fprintf(stderr, "emu hang: replica %d lane %d: sum %d any %d rep_bcast %d rep_sum %d fence %d\n", t >> 2, t & 3, g_ops[t][0], g_ops[t][1], g_ops[t][2], g_ops[t][3], g_ops[t][4]);
    _exit(3);
  });
  std::vector<std::thread> ths;
  for (int t = 1; t < kThreads; t++) ths.emplace_back(lane_main, t);
  lane_main(0);
  for (auto& th : ths) th.join();
  finished = true; watchdog.join();
  if (g_conflicts) { fprintf(stderr, "emu: %d lane-memory words were changed to different values by two replicas between fences\n", g_conflicts); return -2; }
  if (counters) memcpy(counters, cnt_tot, sizeof(cnt_tot));
  return 0;
}

extern "C" void emu_set_dof_params(const double* p) { g_dofprm = p; }
extern "C" void emu_set_model_variant(const float* rec, const float* gt, const float* gpt) { g_vrec = rec; g_vgt = gt; g_vgpt = gpt; }

// same family selection as the library's launch_variant(); BIGK = the family's replay instantiation (lm_step.h launch_family: 128
// slots per chain, the convex collider)
template <bool BIGK>
static int emu_dispatch(const double* chain_model, int n, double* qpos, double* qvel, double* warm, const double* action,
                        int nsub, int debug_env, float* dbgM, float* dbg5, int* counters, double* act, int* leave, const int* only) {
  constexpr int N4 = BIGK ? 128 : 4, N8 = BIGK ? 128 : 8, N6 = BIGK ? 128 : 6;
#define EMU_ARGS chain_model, n, qpos, qvel, warm, action, nsub, debug_env, dbgM, dbg5, counters
  const bool rk4 = (int)chain_model[LM_H_INTEGRATOR] == LM_INT_RK4;
  const bool big = (int)chain_model[LM_H_MAXLINKS] > 3, few = (int)chain_model[LM_H_MAXCONTACTS] <= 4;
  if ((int)chain_model[LM_H_MAXLINKS] > 5)
    return (!rk4 && (int)chain_model[LM_H_NMUSCLE] == 0) ? emu_run_t<6, N8, false, 0, BIGK ? 1 : EMU_SIX_PAIRS>(EMU_ARGS, nullptr, leave, only) : -1;      // (lm_family.hip LM_SIX_PAIRS: 1 the whole pair pass, 3 detection only | the replay kernel's pair pass)
  // the five-link humanoids with self-collision tables (bone hulls, UnitreeH1's cylinders and meshes): the pair families, 8 slots
  if ((int)chain_model[LM_H_NGPAIR] > 0 && big) {
    if ((int)chain_model[LM_H_NMUSCLE] > 0) return (act && !rk4) ? emu_run_t<5, N8, false, LM_MAXMUS, 1>(EMU_ARGS, act, leave, only) : -1;
    return rk4 ? emu_run_t<5, N8, true, 0, 1>(EMU_ARGS, nullptr, leave, only) : emu_run_t<5, N8, false, 0, 1>(EMU_ARGS, nullptr, leave, only);
  }
  if ((int)chain_model[LM_H_NMUSCLE] > 0)
    return (act && big && !rk4 && few) ? emu_run_t<5, N4, false, LM_MAXMUS>(EMU_ARGS, act, leave, only) : -1;
  // the quadruped (LM_A1_PAIRS = 2 in lm_family.hip: the regular kernels leave the convex collider to the replay kernel)
  if (!big && !rk4 && (int)chain_model[LM_H_CONE] == LM_CONE_ELLIPTIC) return emu_run_t<3, N6, false, 0, BIGK ? 1 : 2>(EMU_ARGS, nullptr, leave, only);
  if (!big && !rk4) return emu_run_t<3, 5, false>(EMU_ARGS, nullptr, BIGK ? nullptr : leave, only);          // (generic family: no replay kernel)
  if (!big && rk4) return emu_run_t<3, 4, true>(EMU_ARGS, nullptr, BIGK ? nullptr : leave, only);
  if (!rk4 && few) return emu_run_t<5, N4, false>(EMU_ARGS, nullptr, leave, only);
  if (!rk4) return emu_run_t<5, N8, false>(EMU_ARGS, nullptr, leave, only);
  if (few) return emu_run_t<5, N4, true>(EMU_ARGS, nullptr, leave, only);
  return emu_run_t<5, N8, true>(EMU_ARGS, nullptr, leave, only);
#undef EMU_ARGS
}

// replay = 0: the regular instantiation alone (contacts beyond its slots are dropped and counted, a convex pair of the quadruped is
// NOT simulated); 1: speculate / replay like the library. replayed[n] (may be NULL): which environments the big instantiation ran.
extern "C" int emu_run2(const double* chain_model, int n, double* qpos, double* qvel, double* warm, const double* action,
                        int nsub, int debug_env, float* dbgM, float* dbg5, int* counters, double* act, int replay, int* replayed) {
  if (replay == 2) {          // tests: the big instantiation for EVERY environment
    std::vector<int> all(n, 1);
    return emu_dispatch<true>(chain_model, n, qpos, qvel, warm, action, nsub, debug_env, dbgM, dbg5, counters, act, nullptr, all.data());
  }
  if (!replay) return emu_dispatch<false>(chain_model, n, qpos, qvel, warm, action, nsub, debug_env, dbgM, dbg5, counters, act, nullptr, nullptr);
  std::vector<int> leave(n, 0);
  int cnt1[8] = {0}, cnt2[8] = {0};
  int rc = emu_dispatch<false>(chain_model, n, qpos, qvel, warm, action, nsub, debug_env, dbgM, dbg5, cnt1, act, leave.data(), nullptr);
  if (rc) return rc;
  bool any = false;
  for (int e = 0; e < n; e++) any = any || leave[e];
  if (any) {
    // the abandoned control steps, from the untouched state, in the big instantiation (whatever it still drops is final)
    rc = emu_dispatch<true>(chain_model, n, qpos, qvel, warm, action, nsub, debug_env, dbgM, dbg5, cnt2, act, nullptr, leave.data());
    if (rc) return rc;
  }
  if (counters) for (int j = 0; j < 8; j++) counters[j] = cnt1[j] + cnt2[j];
  if (replayed) for (int e = 0; e < n; e++) replayed[e] = leave[e];
  return 0;
}

extern "C" int emu_run(const double* chain_model, int n, double* qpos, double* qvel, double* warm, const double* action,
                       int nsub, int debug_env, float* dbgM, float* dbg5 /*bias,smooth,qacc_smooth,qacc,qfrc_c: 5*nv*/,
                       int* counters /*8*/, double* act /* [n][na] muscle activations, may be NULL without muscles */) {
  return emu_run2(chain_model, n, qpos, qvel, warm, action, nsub, debug_env, dbgM, dbg5, counters, act, 1, nullptr);
}
