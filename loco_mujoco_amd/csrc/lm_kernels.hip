// lm_kernels.hip — gfx950 kernels and the C-ABI (include/locohip.h) of the batched LocoEnv.step().
//
// Mapping: one 4-lane quad = one environment, lane c = chain c (lm_core.h). 64-thread workgroups = one
// wave = 16 environments, so 4096 environments are 256 workgroups — one per CU, each wave alone on a SIMD
// with the full 512-VGPR budget (the working set of the Newton solve lives in registers; the constant
// model table sits in LDS and is read with quad-broadcast addresses). State is SoA [dof][env] in HBM:
// a wave's 16 environments read 16 consecutive floats per dof. Per control step the kernel moves
// 4*(2nq+2nv+nu+nobs+2) + 8nv bytes per environment — the path is VALU/latency bound, not
// HBM bound, so everything between the state load and the state store happens in registers.
//
// One launch = one control step = n_substeps physics steps + observation + reward (on the previous
// observation) + termination + optional device-side episode reset.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <string>
#include <vector>

#define LM_DEV __device__ __forceinline__
// an integer the optimiser cannot see through (always 0): see lm_core.h `oz`
__device__ __forceinline__ int lm_opaque_zero() { int z = 0; asm volatile("" : "+v"(z)); return z; }
#define LM_OPAQUE_ZERO() lm_opaque_zero()
#define LM_POW01(x, p) __builtin_amdgcn_exp2f((p) * __builtin_amdgcn_logf(x))   // v_exp_f32(p * v_log_f32(x))
#define LM_CLOCK() ((long long)__builtin_readcyclecounter())
#include "lm_core.h"
#include "../../include/locohip.h"

namespace {

// ---- quad policy on gfx950: DPP quad_perm butterflies, no LDS ------------------------------------------------
// REP = 4: the environment is replicated over the four quads of a 16-lane row (lanes that would idle in small batches);
// the replicas run the same instruction stream and split the four step lengths of a line-search round between them.
template <int REP>
struct QuadDppT {
  static constexpr int kRep = REP, kPoints = (REP == 4) ? 4 : 1;
  static __device__ __forceinline__ int rep() { return (threadIdx.x >> 2) & (REP - 1); }
  static __device__ __forceinline__ float rep_bcast(float x, int r) {
    if (REP == 1) return x;
    const int src = (int)((__lane_id() & ~12u) | ((unsigned)r << 2));
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(x)));
  }
  // sum over the replicas: butterfly over lane^4 and lane^8, the same association in every replica
  static __device__ __forceinline__ float rep_sum(float x) {
    if (REP == 1) return x;
    const unsigned l = __lane_id();
    float y = x + __int_as_float(__builtin_amdgcn_ds_bpermute((int)((l ^ 4u) << 2), __float_as_int(x)));
    return y + __int_as_float(__builtin_amdgcn_ds_bpermute((int)((l ^ 8u) << 2), __float_as_int(y)));
  }
  static __device__ __forceinline__ float sum(float x) {
    // quad_perm:[1,0,3,2] = 0xB1, quad_perm:[2,3,0,1] = 0x4E
    float y = x + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));
    return y + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(y), 0x4E, 0xF, 0xF, true));
  }
  static __device__ __forceinline__ bool any(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
  // replicas hand records to each other through the lane memory they share (contact slots, row states). The lanes of
  // a wave run in lock step and the LDS executes a wave's instructions in order, so no hardware wait is needed, but
  // the COMPILER must not move or forward lane-memory accesses across the hand-over point.
  static __device__ __forceinline__ void fence() {
    if (REP > 1) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
  }
};
using QuadDpp = QuadDppT<1>;

thread_local std::string g_err;
thread_local const char* g_launch_err = nullptr;
int fail(const std::string& m) { g_err = m; return 1; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct Task {
  int nv, nu, nobs, ngoal, nsub, reward_type, n_chains, max_links, na, ngrf, cm_used, max_contacts, all_pyr3;
  float rp[8];
};

struct DevStats { float env_steps, episodes, reward_sum, nan_resets, solver_iters, overflow, unhandled, ls_evals, ls_capped, it_ge8, pad0, pad1; };

struct KArgs {
  const float* cm;          // constant table [LM_CM_SIZE]
  const float* mt;          // muscle table [LM_MT_SIZE] or null
  float* act;               // muscle activations, SoA [na][N], or null
  float* dofprm;            // per-environment joint damping | stiffness | frictionloss, SoA [3][nv][N], or null
  const float* drspec;      // their redraw rule at an episode restart [3][nv][3] = (kind, a, b), or null
  float* qpos; float* qvel; float* warm; float* goal;   // SoA [dim][N]
  int* ep_step; unsigned* ep_count;
  const float* action;      // [N][nu] or null
  float* obs; float* reward; unsigned char* done;       // [N][nobs], [N], [N] (may be null)
  const float* table; int table_rows;                   // reset rows [K][nq+nv+ngoal]
  unsigned long long seed; long long env_offset;
  int auto_reset, horizon, action_mode; unsigned step_index;
  int nfused;               // control steps per launch (policy-free rollouts; 1 for lm_step*)
  int xcd_map;              // 1: XCD-aware workgroup -> environment mapping (see step_kernel)
  int N;
  int epb;                  // environments per workgroup (workgroup = 4*epb threads)
  lm::Params P; Task T;
  DevStats* stats;
  unsigned long long* timers;   // LM_TIMERS builds: cycle counters per solver region (lane 0 of each workgroup)
  // debug (forward only)
  float* dM; float* dbias; float* dsmooth; float* dqacc_smooth; float* dqacc; float* dqfrc; int* dncon; int* diter;
};

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

template <int MC, int NS, bool RK4, bool FORWARD_ONLY, int CONE = -1, int NM = 0, bool DR = false, int REP = 1, bool FUSED = false>
__global__ __launch_bounds__(64) void step_kernel(KArgs a) {
  using QuadDpp = QuadDppT<REP>;
  extern __shared__ float dyn_lds[];                       // [constant model table (used part)] [lane memory]
  float* cm = dyn_lds;
  __shared__ float mt[NM > 0 ? LM_MT_SIZE : 1];            // muscle records + tendon paths (muscle variant only)
  if (NM > 0) for (int i = threadIdx.x; i < LM_MT_SIZE; i += blockDim.x) mt[i] = a.mt[i];
  __shared__ float blk_stats[12];
  float* lane_mem = dyn_lds + a.T.cm_used;                 // per 16 lanes: contact slot records, M, twists as [field][lane] (LaneMem<MC,NS>::kGroup floats)
  for (int i = threadIdx.x; i < a.T.cm_used && i < LM_CM_SIZE; i += blockDim.x) cm[i] = a.cm[i];
  for (int i = threadIdx.x; i < 12; i += blockDim.x) blk_stats[i] = 0.0f;
  __syncthreads();
  const int c = threadIdx.x & 3;
  const int e_local = threadIdx.x / (4 * REP);               // REP quads per environment (replicas), see QuadDppT
  // XCD-aware workgroup -> environment mapping. The dispatcher deals consecutive workgroups round-robin to the 8 XCDs (own
  // L2 each), while neighbouring environments share 64-byte lines of the SoA state arrays ([dof][N]: 4 environments of a
  // workgroup use 16 B of a line). Handing every XCD a CONTIGUOUS range of environments keeps each line inside one L2:
  // workgroup b runs on XCD b % 8 and takes the (b / 8)-th group of that XCD's range (LM_NO_XCD_MAP: A/B switch).
  int wg = blockIdx.x;
  if (a.xcd_map) {
    const int nb = gridDim.x, x = wg & 7, per = nb >> 3, rem = nb & 7;
    wg = x * per + (x < rem ? x : rem) + (wg >> 3);
  }
  const int e_raw = wg * a.epb + e_local;
  // padding quads of the last workgroup recompute env N-1; they and the replicas 1..REP-1 store nothing
  const bool valid = e_raw < a.N && QuadDpp::rep() == 0;
  const int e = (e_raw < a.N) ? e_raw : a.N - 1;
  const int N = a.N, nv = a.T.nv;
  const float* rb = cm + LM_CM_ROOT;
#define RD(k, f) rb[LM_R_DOFS + (k) * LM_D_SIZE + (f)]
#define LK(k, f) cm[LM_CM_CHAINS + (LM_C_LINKS + (k) * LM_LINK_SIZE + (f)) * LM_NCHAIN + c]
  const int nl = (int)cm[LM_CM_CHAINS + LM_C_NLINKS * LM_NCHAIN + c];

  // Policy-free rollouts run `nfused` control steps in one launch: every environment advances on its own, no device-wide
  // join between control steps (a launch otherwise ends with its slowest environment). Each control step reloads its
  // state from global memory exactly like a launch of its own would (the lanes of an environment hand it to each
  // other there), so the results are bitwise those of `nfused` single-step launches.
  // (FUSED is a template parameter: the loop around the single-step kernels costs them 4-10 % in SGPR pressure)
  for (int fused = 0; fused < (FUSED ? a.nfused : 1); fused++) {
  if (FUSED && fused > 0) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  const unsigned step_index = a.step_index + (unsigned)fused;
  // ---- load state (root replicated in the 4 lanes: same address -> one transaction)
  float qr[6], vr[6], war[6], qc[MC], vc[MC], wac[MC], goal[4];
  int dr[6], dc[MC];
#pragma unroll
  for (int i = 0; i < 6; i++) { dr[i] = (int)RD(i, LM_D_DOF); qr[i] = a.qpos[dr[i] * N + e]; vr[i] = a.qvel[dr[i] * N + e]; war[i] = a.warm[dr[i] * N + e]; }
#pragma unroll
  for (int k = 0; k < MC; k++) {
    dc[k] = (k < nl) ? (int)LK(k, LM_D_DOF) : 0;
    qc[k] = (k < nl) ? a.qpos[dc[k] * N + e] : 0.0f; vc[k] = (k < nl) ? a.qvel[dc[k] * N + e] : 0.0f; wac[k] = (k < nl) ? a.warm[dc[k] * N + e] : 0.0f;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) goal[i] = (i < a.T.ngoal) ? a.goal[i * N + e] : 0.0f;
  lm::DofPrm<MC> dofp;
  if (DR) {
    const long long pn = (long long)nv * N;
#pragma unroll
    for (int i = 0; i < 6; i++) { dofp.damp_r[i] = a.dofprm[dr[i] * N + e]; dofp.stiff_r[i] = a.dofprm[pn + dr[i] * N + e]; dofp.floss_r[i] = a.dofprm[2 * pn + dr[i] * N + e]; }
#pragma unroll
    for (int k = 0; k < MC; k++) {
      dofp.damp_c[k] = (k < nl) ? a.dofprm[dc[k] * N + e] : 0.0f; dofp.stiff_c[k] = (k < nl) ? a.dofprm[pn + dc[k] * N + e] : 0.0f;
      dofp.floss_c[k] = (k < nl) ? a.dofprm[2 * pn + dc[k] * N + e] : 0.0f;
    }
  }

  // ---- reward on the PREVIOUS observation (reference utils/reward.py:73,110-115)
  auto src = [&](float code) -> float {
    int s = (int)code;
    float v = 0;
    if (s >= LM_SRC_ROOT_QPOS) {
#pragma unroll
      for (int i = 0; i < 6; i++) if (s - LM_SRC_ROOT_QPOS == i) v = qr[i];
    } else if (s >= LM_SRC_GOAL) {
#pragma unroll
      for (int i = 0; i < 4; i++) if (s - LM_SRC_GOAL == i) v = goal[i];
    } else {
#pragma unroll
      for (int i = 0; i < 6; i++) if (s == i) v = vr[i];
    }
    return v;
  };
  float reward = 0.0f;
  if (a.T.reward_type == 1) { float d = src(a.T.rp[0]) - a.T.rp[1]; reward = expf(-d * d); }
  else if (a.T.reward_type == 2) {
    float gv = src(a.T.rp[4]);
    float dx = src(a.T.rp[0]) - gv * src(a.T.rp[2]), dy = src(a.T.rp[1]) - gv * src(a.T.rp[3]);
    reward = expf(-5.0f * sqrtf(dx * dx + dy * dy));
  }

  // ---- actuation: action in [-1,1] -> ctrl (reference base.py:606-621) -> clamp -> gear
  const long long gid = a.env_offset + e;
  auto actuate = [&](float kf, float delta, float mean, float lo, float hi, float gear) -> float {
    int k = (int)kf;
    if (k < 0) return 0.0f;
    float act = 0.0f;
    if (a.action_mode == 0 && a.action) act = a.action[(long long)e * a.T.nu + k];
    else if (a.action_mode == 2) {
      unsigned long long r = mix64(a.seed ^ mix64((unsigned long long)gid * 0x100000001B3ull + step_index) ^ (unsigned long long)(k + 1) * 0xD6E8FEB86659FD93ull);
      act = (float)(r >> 40) * (2.0f / 16777216.0f) - 1.0f;
    }
    float ctrl = fminf(fmaxf(fmaf(act, delta, mean), lo), hi);
    return gear * ctrl;
  };
  float actr[6], actc[MC];
#pragma unroll
  for (int i = 0; i < 6; i++) actr[i] = actuate(RD(i, LM_D_ACT), RD(i, LM_D_ACT_DELTA), RD(i, LM_D_ACT_MEAN), RD(i, LM_D_CTRL_LO), RD(i, LM_D_CTRL_HI), RD(i, LM_D_GEAR));
#pragma unroll
  for (int k = 0; k < MC; k++) actc[k] = (k < nl) ? actuate(LK(k, LM_D_ACT), LK(k, LM_D_ACT_DELTA), LK(k, LM_D_ACT_MEAN), LK(k, LM_D_CTRL_LO), LK(k, LM_D_CTRL_HI), LK(k, LM_D_GEAR)) : 0.0f;

  // ---- physics
  lm::Counters cnt = {};
  using LMm = lm::LaneMem<MC, NS, NM>;
  const int lm_lane = e_local * 4 + c;                        // replicas share their environment's lane memory (same values)
  float* lmem = lane_mem + (lm_lane >> 4) * LMm::kGroup + (lm_lane & 15);
  constexpr int ls = 16;
  if (NM > 0) {
    // this lane's muscles: activation state and un-normalised, clamped control into lane memory
    const int m0 = (int)mt[c], nm = (int)mt[LM_NCHAIN + c];
    for (int i = 0; i < nm; i++) {
      const float* rec = mt + LM_MT_HEAD + (m0 + i) * LM_MU_SIZE;
      const int k = (int)rec[LM_MU_ACT];
      float u = 0.0f;
      if (k >= 0) {
        if (a.action_mode == 0 && a.action) u = a.action[(long long)e * a.T.nu + k];
        else if (a.action_mode == 2) {
          unsigned long long r = mix64(a.seed ^ mix64((unsigned long long)gid * 0x100000001B3ull + step_index) ^ (unsigned long long)(k + 1) * 0xD6E8FEB86659FD93ull);
          u = (float)(r >> 40) * (2.0f / 16777216.0f) - 1.0f;
        }
      }
      lmem[(LMm::kCtrl + i) * ls] = fminf(fmaxf(fmaf(u, rec[LM_MU_ACT_DELTA], rec[LM_MU_ACT_MEAN]), rec[LM_MU_CTRL_LO]), rec[LM_MU_CTRL_HI]);
      lmem[(LMm::kAct + i) * ls] = a.act[(long long)(int)rec[LM_MU_STATE] * N + e];
    }
  }
  if (FORWARD_ONLY) {
    if (!valid) return;
    lm::Debug dbg = {a.dM + (long long)e * nv * nv, a.dbias + e * nv, a.dsmooth + e * nv, a.dqacc_smooth + e * nv, a.dqacc + e * nv, a.dqfrc + e * nv};
    lm::forward<QuadDpp, MC, NS, false, -1, NM>(cm, c, a.P, qr, vr, qc, vc, war, wac, actr, actc, lmem, ls, cnt, &dbg, mt);
    int ncon = (int)(QuadDpp::sum((float)cnt.ncon) + 0.5f);
    if (c == 0 && valid) { a.dncon[e] = ncon; a.diter[e] = cnt.solver_iters; }
    return;
  }
  for (int s = 0; s < a.T.nsub; s++)
    lm::substep<QuadDpp, MC, NS, RK4, CONE, NM, DR>(cm, c, a.P, qr, vr, qc, vc, war, wac, actr, actc, lmem, ls, cnt, nullptr, mt, &dofp, a.T.ngrf > 0);

  QuadDpp::fence();          // the stores below read lane memory that other replicas wrote (muscle activations)

  // ---- termination (reference _has_fallen via per-dof bounds), non-finite guard
  float bad = 0.0f, viol = 0.0f;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    if (!(fabsf(qr[i]) < 1e30f) || !(fabsf(vr[i]) < 1e30f)) bad = 1.0f;
    if (qr[i] < RD(i, LM_D_TERM_QLO) || qr[i] > RD(i, LM_D_TERM_QHI) || vr[i] < RD(i, LM_D_TERM_VLO) || vr[i] > RD(i, LM_D_TERM_VHI)) viol = 1.0f;
  }
#pragma unroll
  for (int k = 0; k < MC; k++) if (k < nl) {
    if (!(fabsf(qc[k]) < 1e30f) || !(fabsf(vc[k]) < 1e30f)) bad = 1.0f;
    if (qc[k] < LK(k, LM_D_TERM_QLO) || qc[k] > LK(k, LM_D_TERM_QHI) || vc[k] < LK(k, LM_D_TERM_VLO) || vc[k] > LK(k, LM_D_TERM_VHI)) viol = 1.0f;
  }
  const bool nonfinite = QuadDpp::sum(bad) > 0.0f;
  const bool absorbing = QuadDpp::sum(viol) > 0.0f || nonfinite;
  int step_no = a.ep_step[e] + 1;
  const bool trunc = a.horizon > 0 && step_no >= a.horizon;
  float episodes = 0.0f;
  bool zero_act = false;                     // a restarted episode starts with zero muscle activation (mj_resetData)
  if (absorbing || trunc) {
    episodes = 1.0f;
    if (a.auto_reset && a.table_rows > 0) {
      // restart from a trajectory sample (reference trajectory.py:236-273 + base.py:478-497), counter-based RNG
      unsigned ec = a.ep_count[e] + 1;
      unsigned long long r = mix64(a.seed ^ mix64((unsigned long long)gid * 2ull + 1ull) ^ ((unsigned long long)ec << 32));
      const float* row = a.table + (long long)(r % (unsigned long long)a.table_rows) * (2 * nv + a.T.ngoal);
#pragma unroll
      for (int i = 0; i < 6; i++) { qr[i] = row[dr[i]]; vr[i] = row[nv + dr[i]]; war[i] = 0.0f; }
#pragma unroll
      for (int k = 0; k < MC; k++) if (k < nl) { qc[k] = row[dc[k]]; vc[k] = row[nv + dc[k]]; wac[k] = 0.0f; }
#pragma unroll
      for (int i = 0; i < 4; i++) if (i < a.T.ngoal) goal[i] = row[2 * nv + i];
      if (c == 0 && valid) {
        a.ep_count[e] = ec;
        for (int i = 0; i < a.T.ngoal; i++) a.goal[i * N + e] = goal[i];
      }
      step_no = 0;
      zero_act = true;
      if (DR && a.drspec && valid) {
        // new episode, new joint parameters (reference base.py:183-185): counter-based draws keyed like the state draw
        auto redraw = [&](int dof, int p) {
          const float* sp = a.drspec + ((long long)p * nv + dof) * 3;
          const int kind = (int)sp[0];
          if (kind == 0) return;
          const unsigned long long r = mix64(a.seed ^ mix64((unsigned long long)gid * 2ull + 1ull) ^ ((unsigned long long)ec << 32) ^ (unsigned long long)(dof * 3 + p + 1) * 0xD6E8FEB86659FD93ull);
          const float u1 = ((float)(r >> 40) + 0.5f) * (1.0f / 16777216.0f), u2 = (float)((r >> 16) & 0xFFFFFFull) * (1.0f / 16777216.0f);
          float v;
          if (kind == 2) v = sp[1] + (sp[2] - sp[1]) * u1;                          // U(a, b)
          else {
            const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);        // N(a, b), kind 1 clipped at 0
            v = fmaf(sp[2], z, sp[1]);
            if (kind == 1) v = fmaxf(v, 0.0f);
          }
          a.dofprm[((long long)p * nv + dof) * N + e] = v;
        };
        for (int p = 0; p < 3; p++) {
          if (c == 0) for (int i = 0; i < 6; i++) redraw(dr[i], p);
          for (int k = 0; k < MC; k++) if (k < nl) redraw(dc[k], p);
        }
      }
    } else if (nonfinite) {
      zero_act = true;
      // no reset table: park the environment at rest in its last finite configuration is impossible; zero it
#pragma unroll
      for (int i = 0; i < 6; i++) { qr[i] = 0.0f; vr[i] = 0.0f; war[i] = 0.0f; }
#pragma unroll
      for (int k = 0; k < MC; k++) { qc[k] = 0.0f; vc[k] = 0.0f; wac[k] = 0.0f; }
    }
  }

  // ---- store state, observation [qpos[idx], qvel[idx], goal], reward, done
  if (valid) {
  if (c == 0) {
#pragma unroll
    for (int i = 0; i < 6; i++) { a.qpos[dr[i] * N + e] = qr[i]; a.qvel[dr[i] * N + e] = vr[i]; a.warm[dr[i] * N + e] = war[i]; }
    a.ep_step[e] = step_no;
    if (a.reward) a.reward[e] = reward;
    if (a.done) a.done[e] = absorbing ? 1 : 0;
  }
#pragma unroll
  for (int k = 0; k < MC; k++) if (k < nl) { a.qpos[dc[k] * N + e] = qc[k]; a.qvel[dc[k] * N + e] = vc[k]; a.warm[dc[k] * N + e] = wac[k]; }
  if (NM > 0) {
    const int m0 = (int)mt[c], nm = (int)mt[LM_NCHAIN + c];
    for (int i = 0; i < nm; i++) {
      const float v = lmem[(LMm::kAct + i) * ls];
      a.act[(long long)(int)mt[LM_MT_HEAD + (m0 + i) * LM_MU_SIZE + LM_MU_STATE] * N + e] = zero_act ? 0.0f : ((fabsf(v) < 1e30f) ? v : 0.0f);
    }
  }
  if (a.obs) {
    float* o = a.obs + (long long)e * a.T.nobs;
    if (c == 0) {
#pragma unroll
      for (int i = 0; i < 6; i++) { int iq = (int)RD(i, LM_D_QOBS), iv = (int)RD(i, LM_D_VOBS); if (iq >= 0) o[iq] = qr[i]; if (iv >= 0) o[iv] = vr[i]; }
      for (int i = 0; i < a.T.ngoal; i++) o[a.T.nobs - a.T.ngrf - a.T.ngoal + i] = goal[i];
    }
#pragma unroll
    for (int k = 0; k < MC; k++) if (k < nl) { int iq = (int)LK(k, LM_D_QOBS), iv = (int)LK(k, LM_D_VOBS); if (iq >= 0) o[iq] = qc[k]; if (iv >= 0) o[iv] = vc[k]; }
    if (a.T.ngrf > 0) {
      // mean contact-frame foot force over the control step's substeps, in kN (reference base.py:596-599: mean_grf / 1000);
      // an episode that restarts in this step reports zeros like the reference's fresh running mean
      const float scale = (step_no == 0 && episodes > 0.0f) ? 0.0f : 1.0f / (1000.0f * (float)a.T.nsub);
      const int o0 = (int)cm[LM_CM_CHAINS + LM_C_GRF_OBS0 * LM_NCHAIN + c], o1 = (int)cm[LM_CM_CHAINS + LM_C_GRF_OBS1 * LM_NCHAIN + c];
#pragma unroll
      for (int j = 0; j < 3; j++) { if (o0 >= 0) o[o0 + j] = cnt.grf[0][j] * scale; if (o1 >= 0) o[o1 + j] = cnt.grf[1][j] * scale; }
    }
  }
  }

  // ---- statistics: LDS adds inside the workgroup, one plain read-modify-write per workgroup slot (no global atomics)
  if (a.stats) {
    if (valid) {
      if (c == 0) {
        atomicAdd(&blk_stats[0], 1.0f); atomicAdd(&blk_stats[1], episodes); atomicAdd(&blk_stats[2], reward);
        atomicAdd(&blk_stats[3], nonfinite ? 1.0f : 0.0f); atomicAdd(&blk_stats[4], (float)cnt.solver_iters);
        atomicAdd(&blk_stats[7], (float)cnt.ls_evals); atomicAdd(&blk_stats[8], (float)cnt.ls_capped);
        atomicAdd(&blk_stats[9], cnt.it_max >= 8 ? 1.0f : 0.0f);
      }
      if (cnt.overflow) atomicAdd(&blk_stats[5], (float)cnt.overflow);
      if (cnt.unhandled) atomicAdd(&blk_stats[6], (float)cnt.unhandled);
    }
#ifdef LM_TIMERS
    if (threadIdx.x == 0) for (int i = 0; i < 12; i++) atomicAdd(&a.timers[i], (unsigned long long)cnt.t[i]);
#endif
  }
  }  // fused control steps
  if (a.stats) {
    __syncthreads();
    for (int i = threadIdx.x; i < 12; i += blockDim.x) {
      float* dst = reinterpret_cast<float*>(a.stats + blockIdx.x) + i;
      *dst += blk_stats[i];
    }
  }
#undef RD
#undef LK
}

}  // namespace

// ================================================================================================================
// C-ABI
// ================================================================================================================
struct lm_model {
  int device;
  float* d_cm;
  float* d_mt;               // muscle table (models with muscles)
  std::vector<float> nominal;  // [3][nv] damping | stiffness | frictionloss of the model
  lm::Params P; Task T;
  int nroot;
  std::vector<int> root_dofs;
};

struct lm_batch {
  lm_model* m;
  int N;
  float *qpos, *qvel, *warm, *goal, *action, *obs, *reward, *table;
  float* act;                // muscle activations [na][N]
  float* dofprm;             // per-environment joint parameters [3][nv][N] (allocated by lm_set_dof_params)
  float* drspec;             // redraw rules [3][nv][3] (lm_set_dof_randomization)
  unsigned char* done;
  int* ep_step; unsigned* ep_count;
  DevStats* stats;
  int table_rows; unsigned long long seed; long long env_offset; int auto_reset, horizon; unsigned step_index;
  int epb, nblocks;
  unsigned long long* timers;
  hipStream_t stream;
  lm_stats acc;            // host-side accumulation (double)
  hipEvent_t ev0, ev1;
};

// kernel variants: MC = links per chain the code is unrolled for, NS = contact slots per chain, RK4 = integrator,
// CONE = friction cone compiled in (the quadruped family gets a specialised step kernel <3,5,Euler,elliptic>;
// everything else reads the cone at run time)
template <class K>
static void launch_one(K kernel, dim3 grid, dim3 block, size_t lane_floats, lm_batch* b, const KArgs& a) {
  // the workgroup's LDS = constant table (the part the model uses) + lane memory, both dynamic; opt in to more than
  // the default 64 KB cap
  const size_t bytes = sizeof(float) * ((size_t)a.T.cm_used + lane_floats);
  hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  hipLaunchKernelGGL(kernel, grid, block, bytes, b->stream, a);
}

// one robot family = (links per chain MC, contact slots per chain NS, integrator, compiled-in cone, muscles per chain NM).
// Picks the layout: replicated (4 quads per environment, small workgroups of <= 4 environments), per-environment joint
// parameters (domain randomisation), or plain.
template <int MC, int NS, bool RK4, int CONE, int NM, bool FWD>
static void launch_family(lm_batch* b, const KArgs& a) {
  static const bool no_replicas = getenv("LM_NO_REPLICAS") != nullptr;                  // A/B switch
  const dim3 grid((b->N + b->epb - 1) / b->epb);
  using LMm = lm::LaneMem<MC, NS, NM>;
  if (FWD) {
    launch_one(step_kernel<MC, NS, RK4, true, -1, NM, false, 1>, grid, dim3(4 * b->epb), (size_t)LMm::kGroup * ((4 * b->epb + 15) / 16), b, a);
  } else if (a.nfused > 1) {
    // fused rollouts exist for the replicated layout only (lm_rollout_fused falls back to single steps otherwise)
    if (b->dofprm) launch_one(step_kernel<MC, NS, RK4, false, CONE, NM, true, 4, true>, grid, dim3(16 * b->epb), (size_t)LMm::kGroup, b, a);
    else launch_one(step_kernel<MC, NS, RK4, false, CONE, NM, false, 4, true>, grid, dim3(16 * b->epb), (size_t)LMm::kGroup, b, a);
  } else if (b->dofprm && b->epb <= 4 && !no_replicas) {
    launch_one(step_kernel<MC, NS, RK4, false, CONE, NM, true, 4>, grid, dim3(16 * b->epb), (size_t)LMm::kGroup, b, a);
  } else if (b->dofprm) {
    launch_one(step_kernel<MC, NS, RK4, false, CONE, NM, true, 1>, grid, dim3(4 * b->epb), (size_t)LMm::kGroup * ((4 * b->epb + 15) / 16), b, a);
  } else if (b->epb <= 4 && !no_replicas) {
    launch_one(step_kernel<MC, NS, RK4, false, CONE, NM, false, 4>, grid, dim3(16 * b->epb), (size_t)LMm::kGroup, b, a);
  } else {
    launch_one(step_kernel<MC, NS, RK4, false, CONE, NM, false, 1>, grid, dim3(4 * b->epb), (size_t)LMm::kGroup * ((4 * b->epb + 15) / 16), b, a);
  }
}

template <bool FWD>
static void launch_variant(lm_batch* b, const KArgs& a) {
  const Task& T = b->m->T;
  const bool big = T.max_links > 3, rk4 = b->m->P.integrator == LM_INT_RK4, few = T.max_contacts <= 4;
  static const bool generic = getenv("LM_GENERIC_KERNELS") != nullptr;      // A/B: run-time cone for the humanoids
  const bool pyr3 = T.all_pyr3 && !generic;
#ifdef LM_PROBE_TALOS_ONLY          // tools/probes: a library with one kernel family builds in seconds (compiler A/B)
  if (big && !rk4 && T.na == 0 && few && pyr3) launch_family<5, 4, false, LM_CONE_PYRAMIDAL, 0, FWD>(b, a);
  else g_launch_err = "probe build: Talos family only";
#else
  if (!big && !rk4 && T.na == 0 && b->m->P.cone == LM_CONE_ELLIPTIC) launch_family<3, 5, false, LM_CONE_ELLIPTIC, 0, FWD>(b, a);  // quadruped: thigh (2) + calf (2) + foot (1) contacts per leg
  // the humanoid families are compiled for condim-3 pyramids only (T.all_pyr3, checked when the model is created): the
  // elliptic code compiles out, no scratch (was 470 B per lane). NB: sensitive to the optimisation level, see the Makefile.
  else if (big && rk4 && T.na == 0 && few && pyr3) launch_family<5, 4, true, LM_CONE_PYRAMIDAL, 0, FWD>(b, a);   // one box foot per leg
  else if (big && rk4 && T.na == 0 && pyr3) launch_family<5, 8, true, LM_CONE_PYRAMIDAL, 0, FWD>(b, a);          // Atlas: two boxes per foot
  else if (big && !rk4 && T.na == 0 && few && pyr3) launch_family<5, 4, false, LM_CONE_PYRAMIDAL, 0, FWD>(b, a);         // Talos (Euler)
  else if (big && !rk4 && T.na == 0 && pyr3) launch_family<5, 8, false, LM_CONE_PYRAMIDAL, 0, FWD>(b, a);                // Talos carrying a box
  else if (big && !rk4 && T.na > 0 && few && pyr3) launch_family<5, 4, false, LM_CONE_PYRAMIDAL, LM_MAXMUS, FWD>(b, a);   // muscle humanoid
  else if (T.na > 0) g_launch_err = "muscle models need the <5 links, <=4 contacts per chain, Euler> family";
  // generic fallbacks (cone read at run time; no replicated / randomised variants are compiled for them)
  else if (b->dofprm) g_launch_err = "per-environment joint parameters are not compiled for this model family";
  else {
    const dim3 grid((b->N + b->epb - 1) / b->epb), block(4 * b->epb);
    const size_t groups = (block.x + 15) / 16;
    if (!big && !rk4) launch_one(step_kernel<3, 4, false, FWD, -1>, grid, block, (size_t)lm::LaneMem<3, 4>::kGroup * groups, b, a);
    else if (!big) launch_one(step_kernel<3, 4, true, FWD, -1>, grid, block, (size_t)lm::LaneMem<3, 4>::kGroup * groups, b, a);
    else if (!rk4) launch_one(step_kernel<5, 8, false, FWD, -1>, grid, block, (size_t)lm::LaneMem<5, 8>::kGroup * groups, b, a);
    else launch_one(step_kernel<5, 8, true, FWD, -1>, grid, block, (size_t)lm::LaneMem<5, 8>::kGroup * groups, b, a);
  }
#endif
}

extern "C" {

const char* lm_last_error(void) { return g_err.c_str(); }

int lm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int lm_model_create(const double* cmod, size_t n, int device, lm_model** out) {
  if (!cmod || n < LM_HEADER_SIZE + LM_CM_SIZE) return fail("chain model too short");
  if ((unsigned)cmod[LM_H_MAGIC] != (unsigned)LM_LMC_MAGIC) return fail("bad chain-model magic");
  if ((int)cmod[LM_H_CM_SIZE] != LM_CM_SIZE) return fail("chain-model table size mismatch (regenerate include/lm_layout.h)");
  if ((int)cmod[LM_H_MAXLINKS] > 5) return fail("chains longer than 5 links are not supported");
  if ((int)cmod[LM_HEADER_SIZE + LM_R_NDOF] != 6) return fail("root body must have 6 dofs");
  const int n_muscle = (int)cmod[LM_H_NMUSCLE];
  if (n_muscle < 0 || n_muscle > LM_MT_MAXMUS) return fail("bad muscle count");
  if (n_muscle > 0 && n < (size_t)(LM_HEADER_SIZE + LM_CM_SIZE + LM_MT_SIZE)) return fail("chain model lacks the muscle table");
  if (n_muscle > 0 && (int)cmod[LM_H_INTEGRATOR] != LM_INT_EULER) return fail("muscles need the Euler integrator");
  HIPCHK(hipSetDevice(device));
  lm_model* m = new lm_model();
  m->device = device;
  std::vector<float> cm(LM_CM_SIZE);
  for (int i = 0; i < LM_CM_SIZE; i++) cm[i] = (float)cmod[LM_HEADER_SIZE + i];
  for (int i = 0; i < 6; i++) {
    const float* blk = cm.data() + LM_R_DOFS + i * LM_D_SIZE;
    if (blk[LM_D_LIMITED] != 0.0f) { delete m; return fail("limited root joints are not supported"); }
  }
  HIPCHK(hipMalloc(&m->d_cm, sizeof(float) * LM_CM_SIZE));
  HIPCHK(hipMemcpy(m->d_cm, cm.data(), sizeof(float) * LM_CM_SIZE, hipMemcpyHostToDevice));
  m->d_mt = nullptr;
  if (n_muscle > 0) {
    std::vector<float> mt(LM_MT_SIZE);
    for (int i = 0; i < LM_MT_SIZE; i++) mt[i] = (float)cmod[LM_HEADER_SIZE + LM_CM_SIZE + i];
    for (int c = 0; c < LM_NCHAIN; c++) if ((int)mt[LM_NCHAIN + c] > LM_MAXMUS) { delete m; return fail("too many muscles on one chain"); }
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
  {
    // every geom with a device collider is a condim-3 contact under pyramidal cones?
    T.all_pyr3 = (int)cmod[LM_H_CONE] == LM_CONE_PYRAMIDAL;
    for (int c = 0; c < LM_NCHAIN && T.all_pyr3; c++) {
      const int ng = (int)cmod[LM_HEADER_SIZE + LM_CM_CHAINS + LM_C_NGEOMS * LM_NCHAIN + c];
      for (int g = 0; g < ng; g++) if ((int)cmod[LM_HEADER_SIZE + LM_CM_CHAINS + (LM_C_GEOMS + g * LM_G_SIZE + LM_G_DIM) * LM_NCHAIN + c] != 3) T.all_pyr3 = 0;
    }
  }
  T.cm_used = ((int)cmod[LM_H_CM_USED] + 63) & ~63;          // keeps lane memory 256-byte aligned behind the table
  if (T.cm_used <= 0 || T.cm_used > ((LM_CM_SIZE + 63) & ~63)) { delete m; return fail("bad constant-table extent"); }
  if (T.ngoal > 4) { delete m; return fail("more than 4 goal entries"); }
  for (int i = 0; i < 8; i++) T.rp[i] = (float)cmod[LM_H_REWARD_P0 + i];
  lm::Params& P = m->P;
  P.h = (float)cmod[LM_H_TIMESTEP];
  P.g = lm::V3{(float)cmod[LM_H_GX], (float)cmod[LM_H_GY], (float)cmod[LM_H_GZ]};
  P.iterations = (int)cmod[LM_H_ITERATIONS];
  P.tolerance = 1e-6f;      // float32 stand-in for MuJoCo's 1e-8 (the gradient itself carries ~1e-6 relative noise)
  P.nv = T.nv;
  P.integrator = (int)cmod[LM_H_INTEGRATOR]; P.cone = (int)cmod[LM_H_CONE]; P.act_position = (int)cmod[LM_H_ACTMODE];
  P.scale = 1.0f / ((float)cmod[LM_H_MEANINERTIA] * (float)T.nv);
  P.ls_tol = 1e-2f; P.ls_iters = 12; P.ls_noise = 2e-6f; P.ablate = 0;
  P.ls_grid[0] = 0.25f; P.ls_grid[1] = 0.0625f; P.ls_grid[2] = 0.015625f;
  if (const char* v = getenv("LM_LS_GRID")) sscanf(v, "%f,%f,%f", &P.ls_grid[0], &P.ls_grid[1], &P.ls_grid[2]);   // A/B knob
  if (const char* v = getenv("LM_LS_NOISE")) P.ls_noise = (float)atof(v);
  if (const char* v = getenv("LM_ABLATE")) P.ablate = atoi(v);
  if (const char* v = getenv("LM_TOLERANCE")) P.tolerance = (float)atof(v);          // tuning knobs for A/B probes
  if (const char* v = getenv("LM_LS_TOL")) P.ls_tol = (float)atof(v);
  if (const char* v = getenv("LM_LS_ITERS")) P.ls_iters = atoi(v);
  *out = m;
  return 0;
}

void lm_model_destroy(lm_model* m) {
  if (!m) return;
  (void)hipFree(m->d_cm);
  if (m->d_mt) (void)hipFree(m->d_mt);
  delete m;
}

int lm_model_dims(const lm_model* m, lm_dims* out) {
  out->nq = m->T.nv; out->nv = m->T.nv; out->nu = m->T.nu; out->nobs = m->T.nobs; out->ngoal = m->T.ngoal;
  out->n_substeps = m->T.nsub; out->n_chains = m->T.n_chains; out->max_chain_dofs = m->T.max_links; out->na = m->T.na;
  return 0;
}

int lm_batch_create(lm_model* m, int n_envs, lm_batch** out) {
  if (n_envs <= 0) return fail("n_envs must be positive");
  HIPCHK(hipSetDevice(m->device));
  lm_batch* b = new lm_batch();
  memset(b, 0, sizeof(*b));
  b->m = m; b->N = n_envs;
  // Four environments per workgroup: with the replicated layout that is one full wave (4 envs x 4 replicas x 4 chains),
  // and a CU's 160 KB of LDS holds four such workgroups = one wave per SIMD. Larger batches simply run more workgroups
  // back to back (measured: 4096 envs 1.32 ms, 16384 envs 4.0 ms, 65536 envs 13.8 ms per control step for UnitreeA1;
  // wider workgroups without replicas were 30-50 % slower at every size). LM_ENVS_PER_BLOCK overrides.
  {
    int epb = n_envs < 4 ? n_envs : 4;
    const char* ov = getenv("LM_ENVS_PER_BLOCK");
    if (ov && atoi(ov) >= 1 && atoi(ov) <= 16) epb = atoi(ov);
    b->epb = epb;
  }
  const int N = n_envs, nv = m->T.nv;
  HIPCHK(hipMalloc(&b->qpos, sizeof(float) * nv * N)); HIPCHK(hipMalloc(&b->qvel, sizeof(float) * nv * N));
  HIPCHK(hipMalloc(&b->warm, sizeof(float) * nv * N)); HIPCHK(hipMalloc(&b->goal, sizeof(float) * 4 * N));
  HIPCHK(hipMalloc(&b->action, sizeof(float) * m->T.nu * N)); HIPCHK(hipMalloc(&b->obs, sizeof(float) * m->T.nobs * N));
  HIPCHK(hipMalloc(&b->reward, sizeof(float) * N)); HIPCHK(hipMalloc(&b->done, N));
  HIPCHK(hipMalloc(&b->ep_step, sizeof(int) * N)); HIPCHK(hipMalloc(&b->ep_count, sizeof(unsigned) * N));
  b->act = nullptr; b->dofprm = nullptr; b->drspec = nullptr;
  if (m->T.na > 0) { HIPCHK(hipMalloc(&b->act, sizeof(float) * m->T.na * N)); HIPCHK(hipMemset(b->act, 0, sizeof(float) * m->T.na * N)); }
  b->nblocks = (n_envs + b->epb - 1) / b->epb;
  HIPCHK(hipMalloc(&b->stats, sizeof(DevStats) * b->nblocks));
  HIPCHK(hipMemset(b->qpos, 0, sizeof(float) * nv * N)); HIPCHK(hipMemset(b->qvel, 0, sizeof(float) * nv * N));
  HIPCHK(hipMemset(b->warm, 0, sizeof(float) * nv * N)); HIPCHK(hipMemset(b->goal, 0, sizeof(float) * 4 * N));
  HIPCHK(hipMemset(b->ep_step, 0, sizeof(int) * N)); HIPCHK(hipMemset(b->ep_count, 0, sizeof(unsigned) * N));
  HIPCHK(hipMemset(b->stats, 0, sizeof(DevStats) * b->nblocks));
  HIPCHK(hipMalloc(&b->timers, sizeof(unsigned long long) * 16)); HIPCHK(hipMemset(b->timers, 0, sizeof(unsigned long long) * 16));
  HIPCHK(hipStreamCreate(&b->stream));
  HIPCHK(hipEventCreate(&b->ev0)); HIPCHK(hipEventCreate(&b->ev1));
  *out = b;
  return 0;
}

void lm_batch_destroy(lm_batch* b) {
  if (!b) return;
  hipSetDevice(b->m->device);
  hipStreamSynchronize(b->stream);
  (void)hipFree(b->qpos); (void)hipFree(b->qvel); (void)hipFree(b->warm); (void)hipFree(b->goal); (void)hipFree(b->action); (void)hipFree(b->obs);
  (void)hipFree(b->reward); (void)hipFree(b->done); (void)hipFree(b->ep_step); (void)hipFree(b->ep_count); (void)hipFree(b->stats); (void)hipFree(b->table); if (b->act) (void)hipFree(b->act); if (b->dofprm) (void)hipFree(b->dofprm); if (b->drspec) (void)hipFree(b->drspec);
  (void)hipEventDestroy(b->ev0); (void)hipEventDestroy(b->ev1); (void)hipStreamDestroy(b->stream);
  delete b;
}

static int upload_soa(lm_batch* b, float* dev, const float* host_aos, int dim, const uint8_t* mask) {
  const int N = b->N;
  std::vector<float> soa((size_t)dim * N);
  if (mask) HIPCHK(hipMemcpyAsync(soa.data(), dev, sizeof(float) * dim * N, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  for (int e = 0; e < N; e++) {
    if (mask && !mask[e]) continue;
    for (int d = 0; d < dim; d++) soa[(size_t)d * N + e] = host_aos[(size_t)e * dim + d];
  }
  HIPCHK(hipMemcpyAsync(dev, soa.data(), sizeof(float) * dim * N, hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return 0;
}

int lm_set_state(lm_batch* b, const float* qpos, const float* qvel, const uint8_t* mask) {
  HIPCHK(hipSetDevice(b->m->device));
  const int N = b->N, nv = b->m->T.nv;
  if (upload_soa(b, b->qpos, qpos, nv, mask)) return 1;
  if (upload_soa(b, b->qvel, qvel, nv, mask)) return 1;
  std::vector<float> z((size_t)nv * N, 0.0f);
  std::vector<int> zs(N, 0);
  if (mask) {
    // clear warm start / step counter only for the masked environments
    std::vector<float> w((size_t)nv * N);
    std::vector<int> st(N);
    HIPCHK(hipMemcpy(w.data(), b->warm, sizeof(float) * nv * N, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(st.data(), b->ep_step, sizeof(int) * N, hipMemcpyDeviceToHost));
    for (int e = 0; e < N; e++) if (mask[e]) { st[e] = 0; for (int d = 0; d < nv; d++) w[(size_t)d * N + e] = 0.0f; }
    HIPCHK(hipMemcpy(b->warm, w.data(), sizeof(float) * nv * N, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->ep_step, st.data(), sizeof(int) * N, hipMemcpyHostToDevice));
    if (b->act) {
      const int na = b->m->T.na;
      std::vector<float> av((size_t)na * N);
      HIPCHK(hipMemcpy(av.data(), b->act, sizeof(float) * na * N, hipMemcpyDeviceToHost));
      for (int e = 0; e < N; e++) if (mask[e]) for (int d = 0; d < na; d++) av[(size_t)d * N + e] = 0.0f;
      HIPCHK(hipMemcpy(b->act, av.data(), sizeof(float) * na * N, hipMemcpyHostToDevice));
    }
  } else {
    if (b->act) HIPCHK(hipMemset(b->act, 0, sizeof(float) * b->m->T.na * N));
    HIPCHK(hipMemcpy(b->warm, z.data(), sizeof(float) * nv * N, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->ep_step, zs.data(), sizeof(int) * N, hipMemcpyHostToDevice));
  }
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
    std::vector<float> init((size_t)3 * nv * N);
    for (int p = 0; p < 3; p++) for (int d = 0; d < nv; d++) for (int e = 0; e < N; e++) init[((size_t)p * nv + d) * N + e] = b->m->nominal[(size_t)p * nv + d];
    HIPCHK(hipMalloc(&b->dofprm, sizeof(float) * 3 * nv * N));
    HIPCHK(hipMemcpy(b->dofprm, init.data(), sizeof(float) * 3 * nv * N, hipMemcpyHostToDevice));
  }
  const float* src[3] = {damping, stiffness, frictionloss};
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
  a.cm = b->m->d_cm; a.mt = b->m->d_mt; a.act = b->act; a.dofprm = b->dofprm; a.drspec = b->drspec; a.qpos = b->qpos; a.qvel = b->qvel; a.warm = b->warm; a.goal = b->goal;
  a.ep_step = b->ep_step; a.ep_count = b->ep_count;
  a.table = b->table; a.table_rows = b->table_rows; a.seed = b->seed; a.env_offset = b->env_offset;
  a.auto_reset = b->auto_reset; a.horizon = b->horizon; a.step_index = b->step_index;
  a.N = b->N; a.P = b->m->P; a.T = b->m->T; a.stats = b->stats;
  a.epb = b->epb; a.timers = b->timers; a.nfused = 1;
  static const bool no_xcd_map = getenv("LM_NO_XCD_MAP") != nullptr;
  a.xcd_map = no_xcd_map ? 0 : 1;
  return a;
}

static bool family_has_replicas(const lm_batch* b) {      // mirrors launch_variant: the generic fallbacks have no replicated kernels
  const Task& T = b->m->T;
  const bool big = T.max_links > 3, rk4 = b->m->P.integrator == LM_INT_RK4, few = T.max_contacts <= 4;
  const bool pyr3 = T.all_pyr3 && getenv("LM_GENERIC_KERNELS") == nullptr;
  if (!big && !rk4 && T.na == 0 && b->m->P.cone == LM_CONE_ELLIPTIC) return true;
  if (big && T.na == 0 && pyr3) return true;
  if (big && !rk4 && T.na > 0 && few && pyr3) return true;
  return false;
}

static void launch_step(lm_batch* b, const KArgs& a) {
  // the per-thread HIP error state is shared with whoever else uses HIP in this process (PyTorch probes peers, pointer
  // attributes ...): drop what they left behind so that the check after the launch reports OUR launch
  (void)hipGetLastError();
  g_launch_err = nullptr;
  launch_variant<false>(b, a);
}

static int drain_stats(lm_batch* b) {
  std::vector<DevStats> s(b->nblocks);
  HIPCHK(hipMemcpyAsync(s.data(), b->stats, sizeof(DevStats) * b->nblocks, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipMemsetAsync(b->stats, 0, sizeof(DevStats) * b->nblocks, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  for (const DevStats& x : s) {
    b->acc.env_steps += x.env_steps; b->acc.episodes += x.episodes; b->acc.reward_sum += x.reward_sum;
    b->acc.nan_resets += x.nan_resets; b->acc.solver_iters += x.solver_iters; b->acc.overflow_contacts += x.overflow;
    b->acc.unhandled_geoms += x.unhandled; b->acc.linesearch_evals += x.ls_evals; b->acc.linesearch_capped += x.ls_capped; b->acc.steps_with_8plus_iters += x.it_ge8;
  }
  return 0;
}

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

int lm_step_device(lm_batch* b, const float* d_action, float* d_obs, float* d_reward, uint8_t* d_done, void* stream, int sync) {
  HIPCHK(hipSetDevice(b->m->device));
  KArgs a = make_args(b);
  if (d_action) { a.action = d_action; a.action_mode = 0; } else a.action_mode = 1;
  a.obs = d_obs ? d_obs : b->obs; a.reward = d_reward ? d_reward : b->reward; a.done = d_done ? d_done : b->done;
  hipStream_t own = b->stream;
  if (stream) b->stream = (hipStream_t)stream;          // run on the caller's stream (e.g. torch's current stream)
  launch_step(b, a);
  hipStream_t used = b->stream;
  b->stream = own;
  if (g_launch_err) return fail(g_launch_err);
  HIPCHK(hipGetLastError());
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
  static const bool no_replicas = getenv("LM_NO_REPLICAS") != nullptr;
  if (b->epb > 4 || no_replicas || !family_has_replicas(b)) steps_per_launch = 1;     // no fused kernels for the full-wave layout
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
  a.stats = nullptr;
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

int lm_get_stats(lm_batch* b, lm_stats* out, int reset) {
  HIPCHK(hipSetDevice(b->m->device));
  if (drain_stats(b)) return 1;
  if (out) *out = b->acc;
  if (reset) memset(&b->acc, 0, sizeof(b->acc));
  return 0;
}

/* profiling builds (-DLM_TIMERS): cycles spent per solver region, summed over workgroups (not part of the ABI header) */
int lm_debug_timers(lm_batch* b, unsigned long long* out16) {
  HIPCHK(hipSetDevice(b->m->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(out16, b->timers, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(b->timers, 0, sizeof(unsigned long long) * 16));
  return 0;
}

int lm_sync(lm_batch* b) {
  HIPCHK(hipSetDevice(b->m->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  return 0;
}

}  // extern "C"
