// TEST TOOLING ONLY — CPU emulation of one quad of GPU lanes, so that the device code in
// loco_mujoco_amd/csrc/lm_core.h can be debugged against the fp64 oracle in a container without a GPU.
// Four OS threads play the four lanes; the quad sum (two DPP adds on gfx950) becomes a barrier + the same
// (x0+x1)+(x2+x3) association. Nothing in the product loads this file; it is built by tests/test_emu_core.py.
#include <barrier>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#define LM_DEV inline
#define LM_OPAQUE_ZERO() 0
#define LM_POW01(x, p) exp2f((p) * log2f(x))
#include "../../loco_mujoco_amd/csrc/lm_core.h"

namespace {
std::barrier<> g_bar(4);
float g_buf[4];
int g_ibuf[4];
thread_local int t_lane = 0;
// POINTS = 4 runs the four-points-per-round line search of the replicated GPU layout with one replica (the lane
// evaluates the four step lengths itself): same decisions, same results, no extra lanes needed on the CPU
template <int POINTS>
struct QuadThreadsT {
  static constexpr int kRep = 1, kPoints = POINTS;
  static int rep() { return 0; }
  static float rep_bcast(float x, int) { return x; }
  static float rep_sum(float x) { return x; }
  static void fence() {}
  static float sum(float x) {
    g_buf[t_lane] = x; g_bar.arrive_and_wait();
    float s = (g_buf[0] + g_buf[1]) + (g_buf[2] + g_buf[3]);
    g_bar.arrive_and_wait(); return s;
  }
  static bool any(bool b) {
    g_ibuf[t_lane] = b; g_bar.arrive_and_wait();
    bool r = g_ibuf[0] | g_ibuf[1] | g_ibuf[2] | g_ibuf[3];
    g_bar.arrive_and_wait(); return r;
  }
};
#ifndef EMU_LS_POINTS
#define EMU_LS_POINTS 1
#endif
using QuadThreads = QuadThreadsT<EMU_LS_POINTS>;
// EMU_DR compiles the per-environment joint-parameter path (DR = true); the parameters come from emu_set_dof_params
// ([3][n][nv]: damping, stiffness, frictionloss) or, when none are set, from the constant table (must change nothing)
#ifdef EMU_DR
constexpr bool kEmuDR = true;
#else
constexpr bool kEmuDR = false;
#endif
const double* g_dofprm = nullptr;
}  // namespace

// EMU_PYRAMID_ONLY compiles the humanoid families like the library does (condim-3 pyramids only, elliptic code out)
#ifdef EMU_PYRAMID_ONLY
template <int MC> constexpr int kEmuCone = (MC == 5) ? 0 : -1;
#else
template <int MC> constexpr int kEmuCone = -1;
#endif

// chain_model: float64 [HEADER + CM]; state arrays [n][nv] double in/out; ctrl [n][nu] (already un-normalised)
// act: muscle activations [n][na] double in/out (NM > 0 only)
template <int MC, int NS, bool RK4, int NM = 0>
static int emu_run_t(const double* chain_model, int n, double* qpos, double* qvel, double* warm, const double* action,
                     int nsub, int debug_env, float* dbgM, float* dbg5, int* counters, double* act = nullptr) {
  const double* H = chain_model;
  const int nv = (int)H[LM_H_NV], nu = (int)H[LM_H_NU];
  std::vector<float> cm(LM_CM_SIZE);
  for (int i = 0; i < LM_CM_SIZE; i++) cm[i] = (float)H[LM_HEADER_SIZE + i];
  const int na = (int)H[LM_H_NMUSCLE];
  std::vector<float> mt(NM > 0 ? LM_MT_SIZE : 1);
  if (NM > 0) for (int i = 0; i < LM_MT_SIZE; i++) mt[i] = (float)H[LM_HEADER_SIZE + LM_CM_SIZE + i];
  lm::Params P;
  P.h = (float)H[LM_H_TIMESTEP]; P.g = lm::v3((float)H[LM_H_GX], (float)H[LM_H_GY], (float)H[LM_H_GZ]);
  P.iterations = (int)H[LM_H_ITERATIONS]; P.tolerance = 1e-6f; P.nv = nv;
  P.scale = 1.0f / ((float)H[LM_H_MEANINERTIA] * nv);
  P.ls_tol = 1e-2f; P.ls_iters = 12; P.ls_noise = 2e-6f; P.ablate = 0;
  P.ls_grid[0] = 0.25f; P.ls_grid[1] = 0.0625f; P.ls_grid[2] = 0.015625f;
  if (const char* v = getenv("LM_LS_GRID")) sscanf(v, "%f,%f,%f", &P.ls_grid[0], &P.ls_grid[1], &P.ls_grid[2]);   // A/B knob
  P.integrator = (int)H[LM_H_INTEGRATOR]; P.cone = (int)H[LM_H_CONE]; P.act_position = (int)H[LM_H_ACTMODE];
  int cnt_tot[6] = {0, 0, 0, 0, 0, 0};
  auto lane_main = [&](int c) {
    t_lane = c;
    const float* rb = cm.data();
    for (int e = 0; e < n; e++) {
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
      if (kEmuDR) {
        auto prm = [&](int which, int dof, float nominal) -> float {
          return g_dofprm ? (float)g_dofprm[((size_t)which * n + e) * nv + dof] : nominal;
        };
        for (int i = 0; i < 6; i++) {
          const float* blk = rb + LM_R_DOFS + i * LM_D_SIZE;
          dofp.damp_r[i] = prm(0, dr[i], blk[LM_D_DAMP]); dofp.stiff_r[i] = prm(1, dr[i], blk[LM_D_STIFF]); dofp.floss_r[i] = prm(2, dr[i], blk[LM_D_FLOSS]);
        }
        for (int k = 0; k < MC; k++) if (k < nl) {
          const float* blk = cm.data() + LM_CM_CHAINS + (LM_C_LINKS + k * LM_LINK_SIZE) * LM_NCHAIN + c;
          dofp.damp_c[k] = prm(0, dc[k], blk[LM_D_DAMP * LM_NCHAIN]); dofp.stiff_c[k] = prm(1, dc[k], blk[LM_D_STIFF * LM_NCHAIN]);
          dofp.floss_c[k] = prm(2, dc[k], blk[LM_D_FLOSS * LM_NCHAIN]);
        }
      }
      lm::Counters cnt = {};
      using LMm = lm::LaneMem<MC, NS, NM>;
      float lmem[LMm::kSize];
      if (NM > 0) {          // this lane's muscles: activation state and control (un-normalised, clamped) into lane memory
        const int m0 = (int)mt[c], nm = (int)mt[LM_NCHAIN + c];
        for (int i = 0; i < nm; i++) {
          const float* rec = mt.data() + LM_MT_HEAD + (m0 + i) * LM_MU_SIZE;
          const int k = (int)rec[LM_MU_ACT];
          float ctrl = (k >= 0) ? (float)action[e * nu + k] * rec[LM_MU_ACT_DELTA] + rec[LM_MU_ACT_MEAN] : 0.0f;
          lmem[LMm::kCtrl + i] = fminf(fmaxf(ctrl, rec[LM_MU_CTRL_LO]), rec[LM_MU_CTRL_HI]);
          lmem[LMm::kAct + i] = (float)act[e * na + (int)rec[LM_MU_STATE]];
        }
      }
      lm::Debug dbg = {dbgM, dbg5, dbg5 + nv, dbg5 + 2 * nv, dbg5 + 3 * nv, dbg5 + 4 * nv};
      for (int s = 0; s < nsub; s++)
        lm::substep<QuadThreads, MC, NS, RK4, kEmuCone<MC>, NM, kEmuDR>(cm.data(), c, P, qr, vr, qc, vc, war, wac, actr, actc, lmem, 1, cnt,
                                                      (e == debug_env && s == 0 && dbgM) ? &dbg : nullptr, mt.data(), &dofp);
      if (NM > 0) {
        const int m0 = (int)mt[c], nm = (int)mt[LM_NCHAIN + c];
        for (int i = 0; i < nm; i++) act[e * na + (int)mt[LM_MT_HEAD + (m0 + i) * LM_MU_SIZE + LM_MU_STATE]] = lmem[LMm::kAct + i];
      }
      g_bar.arrive_and_wait();
      if (c == 0) for (int i = 0; i < 6; i++) { qpos[e * nv + dr[i]] = qr[i]; qvel[e * nv + dr[i]] = vr[i]; warm[e * nv + dr[i]] = war[i]; }
      for (int k = 0; k < MC; k++) if (dc[k] >= 0) { qpos[e * nv + dc[k]] = qc[k]; qvel[e * nv + dc[k]] = vc[k]; warm[e * nv + dc[k]] = wac[k]; }
      static int acc[4][6];
      acc[c][0] = cnt.solver_iters; acc[c][1] = cnt.overflow; acc[c][2] = cnt.unhandled; acc[c][3] = cnt.ncon; acc[c][4] = cnt.ls_evals; acc[c][5] = cnt.ls_capped;
      g_bar.arrive_and_wait();
      if (c == 0) for (int l = 0; l < 4; l++) for (int j = 0; j < 6; j++) cnt_tot[j] += acc[l][j];
      g_bar.arrive_and_wait();
    }
  };
  std::thread t1(lane_main, 1), t2(lane_main, 2), t3(lane_main, 3);
  lane_main(0);
  t1.join(); t2.join(); t3.join();
  if (counters) memcpy(counters, cnt_tot, sizeof(cnt_tot));
  return 0;
}

extern "C" void emu_set_dof_params(const double* p) { g_dofprm = p; }

extern "C" int emu_run(const double* chain_model, int n, double* qpos, double* qvel, double* warm, const double* action,
                       int nsub, int debug_env, float* dbgM, float* dbg5 /*bias,smooth,qacc_smooth,qacc,qfrc_c: 5*nv*/,
                       int* counters /*6*/, double* act /* [n][na] muscle activations, may be NULL without muscles */) {
  // same family selection as the library's launch_variant()
  const bool rk4 = (int)chain_model[LM_H_INTEGRATOR] == LM_INT_RK4;
  const bool big = (int)chain_model[LM_H_MAXLINKS] > 3, few = (int)chain_model[LM_H_MAXCONTACTS] <= 4;
  if ((int)chain_model[LM_H_NMUSCLE] > 0)
    return (act && big && !rk4 && few) ? emu_run_t<5, 4, false, LM_MAXMUS>(chain_model, n, qpos, qvel, warm, action, nsub, debug_env, dbgM, dbg5, counters, act) : -1;
  if (!big && !rk4) return emu_run_t<3, 4, false>(chain_model, n, qpos, qvel, warm, action, nsub, debug_env, dbgM, dbg5, counters);
  if (!big && rk4) return emu_run_t<3, 4, true>(chain_model, n, qpos, qvel, warm, action, nsub, debug_env, dbgM, dbg5, counters);
  if (!rk4 && few) return emu_run_t<5, 4, false>(chain_model, n, qpos, qvel, warm, action, nsub, debug_env, dbgM, dbg5, counters);
  if (!rk4) return emu_run_t<5, 8, false>(chain_model, n, qpos, qvel, warm, action, nsub, debug_env, dbgM, dbg5, counters);
  if (few) return emu_run_t<5, 4, true>(chain_model, n, qpos, qvel, warm, action, nsub, debug_env, dbgM, dbg5, counters);
  return emu_run_t<5, 8, true>(chain_model, n, qpos, qvel, warm, action, nsub, debug_env, dbgM, dbg5, counters);
}
