// lm_step.h — the gfx950 step kernel (one launch = one control step of a batch) and its launch helpers.
//
// Mapping: one 4-lane quad = one environment, lane c = chain c (lm_core.h). The default ("replicated") layout runs
// every environment on the 4 quads of a 16-lane row: a workgroup is ONE wave = 4 environments x 4 replicas x 4 chains;
// the replicas share their environment's lane memory in LDS and split the line-search step lengths, contact slots,
// geoms and muscles between them. The plain layout (REP = 1) packs 16 environments into a wave. State is
// SoA [dof][env] in HBM; per control step the kernel moves 4*(2nq+2nv+nu+nobs+2) + 8nv bytes per environment — the
// path is VALU-issue bound, not HBM bound, so everything between the state load and the state store
// happens in registers and LDS.
//
// One launch = one control step = n_substeps physics steps + observation + reward (on the previous observation) +
// termination + optional device-side episode reset. This header is compiled once per kernel family and part
// (lm_family.hip, -DLM_FAMILY=k -DLM_PART=p) so that the families build in parallel; lm_kernels.hip holds the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <string>
#include <vector>
#include <type_traits>

#define LM_DEV __device__ __forceinline__
// The convex-pair collider is INLINED into the step kernel. As a real function (noinline) it kept its float64 registers and its private
// portal array out of the quadruped's kernel (-1.5 % on the bench rollout, which never calls it), but in round 3 kernels that contained
// the CALL and sat at the register ceiling came out wrong under one build setting or another (profiles/r3_notes.md §4). Round 5
// (profiles/r5_notes.md §5): on the current source -DLM_MPR_CALL is right with all three scheduler settings, and no faster for the
// humanoids either (HumanoidTorque.run 15.9 / 16.2 / 16.5 ms against 16.0 inlined: the kernels spill MORE with the call, 668-742 VGPRs
// against 578) — the defect behind those failures is a code-generation one that comes and goes with the source (see csrc/Makefile).
#ifdef LM_MPR_CALL
#define LM_DEV_COLD __device__ __attribute__((noinline))
#else
#define LM_DEV_COLD __device__ __forceinline__
#endif
// an integer the optimiser cannot see through (always 0): see lm_core.h `oz`
__device__ __forceinline__ int lm_opaque_zero() { int z = 0; asm volatile("" : "+v"(z)); return z; }
#define LM_OPAQUE_ZERO() lm_opaque_zero()
// words of the convex collider's warm-start cache (lm_core.h mpr_convex_pair): agent-scope relaxed atomics — past the CU's L1, which the
// lane that wrote the word one forward pass earlier does not share a coherent view with by default (loads could hit a stale line)
#define LM_GLD64(p) __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LM_GST64(p, v) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LM_POW01(x, p) __builtin_amdgcn_exp2f((p) * __builtin_amdgcn_logf(x))   // v_exp_f32(p * v_log_f32(x))
#define LM_CLOCK() ((long long)__builtin_readcyclecounter())
#include "lm_core.h"
#include "../../include/locohip.h"

namespace lmk {

// ---- quad policy on gfx950: DPP quad_perm butterflies, no LDS ------------------------------------------------
// REP = 4: the environment is replicated over the four quads of a 16-lane row (lanes that would idle in small batches);
// the replicas run the same instruction stream and split the four step lengths of a line-search round between them.
template <int REP>
struct QuadDppT {
  static constexpr int kRep = REP, kPoints = (REP >= 4) ? 4 : 1;
  // one bit per lane of an environment (bit 4 * replica + chain): 16 lanes with four replicas, the whole wave with sixteen
  using mask_t = typename std::conditional<(REP > 8), unsigned long long, unsigned>::type;
  static __device__ __forceinline__ int rep() { return (threadIdx.x >> 2) & (REP - 1); }
  static __device__ __forceinline__ float rep_bcast(float x, int r) {
    if (REP == 1) return x;
    const int src = (int)((__lane_id() & ~(unsigned)(4 * (REP - 1))) | ((unsigned)r << 2));
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(x)));
  }
  // sum over the replicas, the same association in every replica. Four replicas: the row rotations of DPP — x + ror8(x), then
  // y + ror4(y): every replica adds the same two pair sums (a + b = b + a bit for bit), one VALU instruction per stage (rounds 1-4
  // and the sixteen-replica layout: ds_bpermute butterflies over lane^4, lane^8 (, lane^16, lane^32) through the LDS crossbar)
  static __device__ __forceinline__ float rep_sum(float x) {
    if (REP == 1) return x;
#ifndef LM_REP_SUM_BPERMUTE
    if (REP == 4) {
      // row_ror:8 = 0x128, row_ror:4 = 0x124 (rotation within the 16-lane row = within the environment)
      const float y4 = x + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x128, 0xF, 0xF, true));
      return y4 + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(y4), 0x124, 0xF, 0xF, true));
    }
#endif
    const unsigned l = __lane_id();
    float y = x + __int_as_float(__builtin_amdgcn_ds_bpermute((int)((l ^ 4u) << 2), __float_as_int(x)));
    y = y + __int_as_float(__builtin_amdgcn_ds_bpermute((int)((l ^ 8u) << 2), __float_as_int(y)));
    if (REP > 4) {
      y = y + __int_as_float(__builtin_amdgcn_ds_bpermute((int)((l ^ 16u) << 2), __float_as_int(y)));
      y = y + __int_as_float(__builtin_amdgcn_ds_bpermute((int)((l ^ 32u) << 2), __float_as_int(y)));
    }
    return y;
  }
  static __device__ __forceinline__ float sum(float x) {
    // quad_perm:[1,0,3,2] = 0xB1, quad_perm:[2,3,0,1] = 0x4E
#ifdef LM_UPDATE_DPP
    float y = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false));
    return y + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(y), 0x4E, 0xF, 0xF, false));
#else
    float y = x + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));
    return y + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(y), 0x4E, 0xF, 0xF, true));
#endif
  }
  static __device__ __forceinline__ bool any(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
  // the four lanes of a quad (self-collisions between two chains): the partner lane's column of lane memory is `dl` floats
  // away ([field][16 lanes] groups); a value of a named lane of the quad; and the point where lane memory written by the
  // quad's other lanes becomes readable — lock step and the in-order LDS make that a compiler-only fence, like fence()
  static __device__ __forceinline__ float peer(const LM_LMEM_T* lmem, int ls, int i, int dl) { return lmem[i * ls + dl]; }
  static __device__ __forceinline__ void peer_write(LM_LMEM_T* lmem, int ls, int i, int dl, float v) { lmem[i * ls + dl] = v; }
  // one bit per lane of my environment (bit 4 * replica + chain): the wave's ballot, my environment's part of it
  static __device__ __forceinline__ mask_t env_ballot(bool b) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(b);
    if (REP == 16) return (mask_t)m;                     // the wave is ONE environment
    return (REP == 4) ? (mask_t)((m >> (__lane_id() & 48u)) & 0xffffull) : (mask_t)((m >> (__lane_id() & 60u)) & 0xfull);
  }
  static __device__ __forceinline__ float quad_read(float x, int src) {
    const int lane = (int)((__lane_id() & ~3u) | (unsigned)src);
    return __int_as_float(__builtin_amdgcn_ds_bpermute(lane << 2, __float_as_int(x)));
  }
  static __device__ __forceinline__ void quad_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
  // replicas hand records to each other through the lane memory they share (contact slots, row states). The lanes of
  // a wave run in lock step and the LDS executes a wave's instructions in order, so no hardware wait is needed, but
  // the COMPILER must not move or forward lane-memory accesses across the hand-over point.
  static __device__ __forceinline__ void fence() {
    if (REP > 1) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
  }
};
using QuadDpp = QuadDppT<1>;

struct Task {
  int nv, nu, nobs, ngoal, nsub, reward_type, n_chains, max_links, na, ngrf, cm_used, max_contacts, all_pyr3, npair;
  float rp[8];
};

struct DevStats { float env_steps, episodes, reward_sum, nan_resets, solver_iters, overflow, unhandled, ls_evals, ls_capped, it_ge8, selfprox, selfcon, replayed, natown, pad_[2]; };
constexpr int kNStats = 14;

struct KArgs {
  const float* cm;          // constant table [LM_CM_SIZE]
  const float* mt;          // muscle table [LM_MT_SIZE] or null
  float* act;               // muscle activations, SoA [na][N], or null
  float* dofprm;            // per-environment joint damping | stiffness | frictionloss, SoA [3][nv][N], or null
  const float* drspec;      // their redraw rule at an episode restart [3][nv][3] = (kind, a, b), or null
  // model variants (lm_set_model_variants): inertial records [nvar][LM_IR_SIZE][4], geom tables [nvar][LM_GT_SIZE], geom-pair
  // tables [nvar][gpt_floats] (or null), and the variant of every environment [N] (redrawn at a device-side restart)
  const float* vrec; const float* vgt; const float* vgpt; int* var; int nvar, gpt_floats;
  int var_rows;             // > 0: the reset table is nvar blocks of var_rows rows, the variant follows the row (lm_set_variant_rows)
  unsigned char* vdirty;    // [N] or null: the model compiler is on (lm_set_model_compiler: slot e belongs to environment e) — a device-side
                            // restart asks for a FRESH model here instead of drawing an index; lm_compile.hip writes it after the launch
  float* qpos; float* qvel; float* warm; float* goal;   // SoA [dim][N]
  int* ep_step; unsigned* ep_count;
  const float* action;      // [N][nu] or null
  float* obs; float* reward; unsigned char* done;       // [N][nobs], [N], [N] (may be null)
  unsigned char* flags;     // [N] per control step: 1 contact dropped (slots full) | 2 self pair without a collider in reach | 4 collider-less geom at the floor
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
  unsigned long long* tline;    // LM_TIMERS builds: wall-clock (100 MHz) time line of the LAST launch: per environment [4] = listed for the replay
                                // kernel | taken by a replay workgroup | done there | substep it resumes at; then per regular workgroup [2] = start | end
  // speculate / replay (see step_kernel): environments whose control step left the regular kernel's capacity
  int* replay_list;             // [N] environment id + 1 per entry (0 = empty / taken), appended by the regular kernel (null: no replay, drops are final)
  int* replay_ctl;              // [0] entries appended, [1] tickets handed out (pollers), [2] regular workgroups that are through, [3] workgroups of the
                                // drain pass that are through (the last one resets [0..3]), [5] the epoch: launches whose drain pass is complete,
                                // [7] pollers that left by their time-out, ever (-> host_hint[2]: the host stops launching pollers)
  int* stall;                   // [N] the fused control step at which the environment left the regular kernel (0 for single-step launches)
  // RESUME (round 5): the state at the START of the substep in which the control step left the regular kernel's capacity, SoA [nv][N]
  // each (positions, velocities, warm start), and that substep's number [N] (0: from the control step's own state). The replay kernel
  // continues from there instead of running the whole control step again: a robot that runs out of contact slots in substep 8 of 10
  // used to cost 0.8 + 1.0 control steps, and the launch ends with it (profiles/r5_notes.md §3-4)
  float* hq; float* hv; float* hw; int* hsub;
  // PREDICTION (round 5): a robot folded on the floor stays beyond the regular kernel's capacity for a few control steps. The replay
  // kernel leaves a mark [N] when the control step it just ran needed more than the regular kernel holds (reg_ns slots / reg_q queued
  // pairs / reg_r pair results per chain) and the episode goes on; the regular kernel hands a marked environment over BEFORE its first
  // substep instead of finding out again at the end of it (1.5 ms into the launch, and the launch ends with these robots)
  int* premark; int reg_ns, reg_q, reg_r;
  int replay_all;               // tests (lm_batch_set_replay(b, 2)): EVERY control step is abandoned and run by the replay kernel
  unsigned char* replay_mark;   // [N] sticky: the replay kernel ran (part of) this environment's control steps since the marks were last cleared
  int reg_grid;                 // workgroups of the regular launch (the pollers leave when all of them are through)
  int epoch;                    // number of this launch among the batch's launches with a replay pass (the pollers wait for replay_ctl[5] to reach it)
  int* host_hint;               // pinned host words: the drain pass leaves {entries of its launch, its epoch + 1} there
  int drain;                    // replay kernel: 1 = the pass behind the regular launch (takes whatever is still listed, resets the control words)
  int stats_off;                // first statistics slot of this launch (the concurrent replay kernel has a range of its own)
  // self-collision detection (lm_core.h): per chain lane the clearance left since the last detection and the speed memory of its
  // travel bound, SoA [3][4][N]; kept from one control step to the next (zeroed by state uploads and restarts: "detect now")
  float* slack;
  // warm-start cache of the convex collider (lm_core.h mpr_convex_pair): lm::kMprCacheFloats floats per environment and geom-pair
  // record, [N][mprc_pairs][16], or null (no hull pairs / a family without the collider in its regular kernels)
  float* mprc; int mprc_pairs;
  // ACTIVE LIST (lm_batch_set_active): the launch runs `n_active` environments, slot s of the grid = environment env_map[s] (null:
  // environment s; n_active = N for a batch without a list). N stays the stride of every state array: an environment keeps its place.
  // Several models that share one set of environment ids (the humanoid's four sizes, a size drawn per episode) each step their own.
  const int* env_map; int n_active;
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

// SPECULATE / REPLAY. The regular kernels are sized for what a robot does in its gaits: NS contact slots per chain, a short queue
// for convex pairs, and — the quadruped's (PM == 2) — no convex-pair collider at all. A control step that needs more (a chain with
// more simultaneous contacts than slots, a full queue or result list of the pair pass, a convex pair within reach of a kernel
// without the collider) is ABANDONED at the end of the substep that found out: nothing of it is stored, the environment's lanes sit
// the rest of the control step out, and the environment is appended to `replay_list`. The REPLAY kernel (the same code compiled with
// NS > 8: a slot for every contact, long lists, the collider; ONE environment per workgroup) runs that control step — and, in a fused
// rollout, the rest of the launch's control steps — from the untouched state. It is launched twice per step (lm_kernels.hip):
//   * as POLLERS, a few workgroups on a second stream, BEFORE the regular kernel and only when recent launches had abandoned steps:
//     they wait for entries (a ticket each) and work them off while the regular launch is still running — a replay behind the
//     launch costs a whole control step's latency for a handful of environments, beside it next to nothing. Their stream does not
//     wait for the previous launch: they become resident while that one tails off and wait for its drain pass on the device (epoch);
//   * as the DRAIN pass behind the regular launch, on its stream: whatever is still listed (no pollers; more entries than they got
//     through; a poller that gave up waiting), and the reset of the control words.
// An entry is taken with an atomic exchange (0 = taken), so no environment is run twice or left behind whichever pass finds it.
// The engine the reference calls never drops a contact (humanoid_torque.xml:19 njmax 1000 / nconmax 400): neither does this.
// Whatever exceeds even the replay kernel's capacity is dropped, counted and flagged as before.
// Next entry for a replay workgroup (thread 0): the environment id + 1, or 0 = leave.
__device__ __forceinline__ int replay_next(const KArgs& a, int& cursor) {
  if (a.drain) {
    const int tail = a.replay_ctl[0];                  // complete: the producers finished before this pass started
    for (; cursor < tail; cursor += (int)gridDim.x) {
      if (a.replay_list[cursor] == 0) continue;
      const int v = atomicExch(&a.replay_list[cursor], 0);
      if (v) { cursor += (int)gridDim.x; return v; }
    }
    return 0;
  }
  const long long t0 = (long long)wall_clock64();      // 100 MHz
  // The pollers are launched without waiting for the previous launch (gated on its completion they would start AFTER the regular
  // kernel had taken every SIMD): they are resident early, and wait here until the previous launch's drain pass has reset the
  // control words (epoch)
  while (__hip_atomic_load(&a.replay_ctl[5], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
    if ((long long)wall_clock64() - t0 > 50000000ll) { atomicAdd(&a.replay_ctl[7], 1); return 0; }
    __builtin_amdgcn_s_sleep(64);
  }
  const int ticket = atomicAdd(&a.replay_ctl[1], 1);
  for (;;) {
    if (ticket < a.N && __hip_atomic_load(&a.replay_list[ticket], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0) {
      const int v = atomicExch(&a.replay_list[ticket], 0);
      if (v) return v;
    }
    // every regular workgroup is through and my ticket is beyond the last entry: nothing will come any more
    if (__hip_atomic_load(&a.replay_ctl[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= a.reg_grid &&
        ticket >= __hip_atomic_load(&a.replay_ctl[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) return 0;
    // (0.5 s: never wait for ever — the drain pass takes what is left; counted: the host stops launching pollers after the first
    // time-out, which means that they do not run beside the regular kernel here — a profiler or a runtime that serialises kernels)
    if ((long long)wall_clock64() - t0 > 50000000ll) { atomicAdd(&a.replay_ctl[7], 1); return 0; }
    __builtin_amdgcn_s_sleep(64);
  }
}

template <int MC, int NS, bool RK4, bool FORWARD_ONLY, int CONE = -1, int NM = 0, int DR = 0, int REP = 1, bool FUSED = false, int PM = 0>
__global__ __launch_bounds__(64) void step_kernel(KArgs a) {
  using QuadDpp = QuadDppT<REP>;
  constexpr bool PAIRS = PM != 0;
  constexpr bool REPLAY = NS > 8;        // (always compiled with FUSED: a replayed environment finishes the launch's control steps here)
  // REPLAY, drain pass: most launches leave nothing on the list — be counted and leave before the tables are copied (the last
  // workgroup through resets the control words for the next launch)
  if (REPLAY && a.drain) {
    bool work = false;
    const int tail = a.replay_ctl[0];
    for (int i = (int)blockIdx.x; i < tail && !work; i += (int)gridDim.x) work = a.replay_list[i] != 0;
    if (!work) {
      if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&a.replay_ctl[3], 1) == (int)gridDim.x - 1) { if (a.host_hint) { a.host_hint[0] = tail; a.host_hint[1] = a.epoch + 1; a.host_hint[2] = a.replay_ctl[7]; } a.replay_ctl[0] = 0; a.replay_ctl[1] = 0; a.replay_ctl[2] = 0; a.replay_ctl[3] = 0; __threadfence(); __hip_atomic_store(&a.replay_ctl[5], a.epoch + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
      }
      return;
    }
  }
#ifdef LM_TIMERS
  if (!REPLAY && !FORWARD_ONLY && a.tline && threadIdx.x == 0) a.tline[4 * (long long)a.N + 2 * blockIdx.x] = wall_clock64();
#endif
  extern __shared__ float dyn_lds[];                       // [constant model table (used part)] [lane memory]
  float* cm = dyn_lds;
  __shared__ float mt[NM > 0 ? LM_MT_SIZE : 1];            // muscle records + tendon paths (muscle variant only)
  if (NM > 0) for (int i = threadIdx.x; i < LM_MT_SIZE; i += blockDim.x) mt[i] = a.mt[i];
  __shared__ float blk_stats[kNStats];
  float* lane_mem = dyn_lds + a.T.cm_used;                 // per 16 lanes: contact slot records, M, twists as [field][lane] (LaneMem<MC,NS>::kGroup floats)
  for (int i = threadIdx.x; i < a.T.cm_used && i < LM_CM_SIZE; i += blockDim.x) cm[i] = a.cm[i];
  for (int i = threadIdx.x; i < kNStats; i += blockDim.x) blk_stats[i] = 0.0f;
  __syncthreads();
  const int c = threadIdx.x & 3;
  const int e_local = threadIdx.x / (4 * REP);               // REP quads per environment (replicas), see QuadDppT
  // REPLAY: the workgroup takes one listed environment after the other (replay_next)
  int cursor = (int)blockIdx.x;
  for (int item = 0; REPLAY || item < 1; item++) {
  int entry = 0;
  if (REPLAY) {
    if (threadIdx.x == 0) entry = replay_next(a, cursor);
    entry = __shfl(entry, 0, 64);
    if (entry <= 0) break;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // the state the regular kernel stored before it listed the environment
#ifdef LM_TIMERS
    if (a.tline && threadIdx.x == 0) a.tline[4 * (entry - 1) + 1] = wall_clock64();
#endif
  }
  // XCD-aware workgroup -> environment mapping. The dispatcher deals consecutive workgroups round-robin to the 8 XCDs (own
  // L2 each), while neighbouring environments share 64-byte lines of the SoA state arrays ([dof][N]: 4 environments of a
  // workgroup use 16 B of a line). Handing every XCD a CONTIGUOUS range of environments keeps each line inside one L2:
  // workgroup b runs on XCD b % 8 and takes the (b / 8)-th group of that XCD's range (LM_NO_XCD_MAP: A/B switch).
  int wg = blockIdx.x;
  if (!REPLAY && a.xcd_map) {
    const int nb = gridDim.x, x = wg & 7, per = nb >> 3, rem = nb & 7;
    wg = x * per + (x < rem ? x : rem) + (wg >> 3);
  }
  // (the quadruped's kernels — the bench line's — are compiled WITHOUT the active list: the indirection costs them 0.9 %, 1.212 against
  // 1.200 ms per control step in four alternating runs on one box, and a quadruped batch has one model; lm_batch_set_active refuses it)
#if defined(LM_NO_ENV_MAP) || (defined(LM_FAMILY) && LM_FAMILY == 0)
  int e_raw = wg * a.epb + e_local;
  bool in_range = e_raw < a.N;
  if (!in_range) e_raw = a.N - 1;
#else
  const int slot_ = wg * a.epb + e_local;
  bool in_range = slot_ < a.n_active;
  // (the padding quads of the last workgroup recompute the last active environment)
  int e_raw = in_range ? slot_ : a.n_active - 1;
  if (!REPLAY && a.env_map) e_raw = a.env_map[e_raw];
#endif
  int first_step = 0;               // REPLAY: the fused control step at which the environment left the regular kernel
  if (REPLAY) { in_range = true; e_raw = entry - 1; first_step = a.stall[e_raw]; }
  // padding quads (REPLAY: none — the entry is the environment); they and the replicas 1..REP-1 store nothing
  const bool valid0 = in_range && QuadDpp::rep() == 0;
  const int e = e_raw;
  bool gone = false;                // this environment left the launch: abandoned here and handed to the replay kernel
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
  bool valid = valid0 && !gone && (!REPLAY || fused >= first_step);
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
  dofp.mprc = a.mprc ? a.mprc + (long long)e * a.mprc_pairs * lm::kMprCacheFloats : nullptr;
  if (DR) {
    const long long pn = (long long)nv * N;
#pragma unroll
    for (int i = 0; i < 6; i++) dofp.floss_r[i] = a.dofprm[2 * pn + dr[i] * N + e];
#pragma unroll
    for (int k = 0; k < MC; k++) dofp.floss_c[k] = (k < nl) ? a.dofprm[2 * pn + dc[k] * N + e] : 0.0f;
    dofp.damp = a.dofprm + e; dofp.stiff = a.dofprm + pn + e; dofp.stride = N;
    // the environment's model variant (inertial record, geom tables); redrawn below when the episode restarts
    dofp.inr = nullptr; dofp.gt = a.P.gt; dofp.gpt = a.P.gpt;
    if (DR == 2) {
      const int var = a.var[e];
      dofp.inr = a.vrec + (long long)var * (LM_IR_SIZE * LM_NCHAIN);
      dofp.gt = a.vgt + (long long)var * LM_GT_SIZE;
      dofp.gpt = a.vgpt ? a.vgpt + (long long)var * a.gpt_floats : a.P.gpt;
#pragma unroll
      for (int i = 0; i < 6; i++) dofp.rfl_r[i] = dofp.inr[(LM_IR_ROOT_DOF + 3 * i + 2) * LM_NCHAIN + c];
#pragma unroll
      for (int k = 0; k < MC; k++) dofp.rfl_c[k] = dofp.inr[(k * LM_IR_LINK + 12) * LM_NCHAIN + c];
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
  // (the lane memory of a DETECTION-ONLY kernel, PM == 3 — the -DLM_SIX_PAIRS=3 variant of the six-link family, lm_family.hip — is the
  // layout without the pair extension: forward() has always used that one, this alias and the launcher now size the workgroup's LDS by it
  // too, 59 floats per column less. The shipped six-link kernels carry the full pair pass, PM == 1: 42.7 KB per workgroup, three per CU)
  using LMm = lm::LaneMemFor<MC, NS, NM, (PM == 1 || PM == 2), CONE>;
  const int lm_lane = e_local * 4 + c;                        // replicas share their environment's lane memory (same values)
  // REPLAY: ONE environment per workgroup and four lane-memory columns instead of sixteen — the whole LDS of the workgroup for one
  // robot's contact slots (128 per chain: more than the engine's own nconmax of 400 per robot, humanoid_torque.xml:19)
  constexpr int ls = REPLAY ? 4 : 16;
  LM_LMEM_T* lmem = REPLAY ? lane_mem + c : lane_mem + (lm_lane >> 4) * LMm::kGroup + (lm_lane & 15);
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
    lm::forward<QuadDpp, MC, NS, false, -1, NM, 0, PM>(cm, c, a.P, qr, vr, qc, vc, war, wac, actr, actc, lmem, ls, cnt, &dbg, mt);
    int ncon = (int)(QuadDpp::sum((float)cnt.ncon) + 0.5f);
    if (c == 0 && valid) { a.dncon[e] = ncon; a.diter[e] = cnt.solver_iters; }
    return;
  }
  // self-collision detection is skipped while the clearance found by the last one cannot have been used up (lm_core.h): that
  // knowledge survives the control step (a fresh state starts at 0 = "detect now")
  float pair_slack[3] = {0.0f, 0.0f, 0.0f};
  if (PAIRS && a.slack) {
#pragma unroll
    for (int j = 0; j < 3; j++) pair_slack[j] = a.slack[((long long)j * 4 + c) * N + e];
  }
  // Did the control step stay inside the kernel's capacity? Checked after every substep: if not it is abandoned — nothing of it is
  // stored, its lanes sit the rest out (the wave's other environments are no longer held up by the robot that needs the big kernel,
  // usually the slowest of them) and the environment is listed for the replay kernel at once: a poller may already be waiting.
  // Any lane of the environment may have seen it (the pair pass deals its tests to all replicas): an environment-wide vote.
  // RESUME: the families whose control steps are long and do run out of capacity under a random policy (the humanoids). Not with
  // muscles (their activation states advance in lane memory) and not with foot-force observations (running sums over the substeps):
  // those restart from the control step's own state as before, and so does the quadruped (rare, and its kernel is the bench line's)
  constexpr bool RESUME = MC >= 5 && NM == 0 && !FORWARD_ONLY;
  constexpr bool PREDICT = MC >= 5;      // (KArgs::premark: the humanoids' folded robots; the quadruped's kernel stays as it was)
  const bool resume_on = RESUME && a.hq != nullptr && a.T.ngrf == 0;
  int s0 = 0;                       // REPLAY: the substep at which this control step is taken over
  if (REPLAY && RESUME && resume_on && fused == first_step && in_range) {
    s0 = a.hsub[e];
    if (s0 > 0) {
      // the state at the start of that substep, as the regular kernel left it (reward, actuation and the episode bookkeeping above
      // and below belong to the control step and use its own state / counters)
#pragma unroll
      for (int i = 0; i < 6; i++) { qr[i] = a.hq[dr[i] * N + e]; vr[i] = a.hv[dr[i] * N + e]; war[i] = a.hw[dr[i] * N + e]; }
#pragma unroll
      for (int k = 0; k < MC; k++) if (k < nl) { qc[k] = a.hq[dc[k] * N + e]; vc[k] = a.hv[dc[k] * N + e]; wac[k] = a.hw[dc[k] * N + e]; }
      pair_slack[0] = pair_slack[1] = pair_slack[2] = 0.0f;       // "detect now": the slack of the control step's start says nothing here
    }
  }
  for (int s = s0; s < a.T.nsub; s++) {
    if (!REPLAY && a.replay_list && !gone) {
      const bool leave = (s == 0 && (a.replay_all || (PREDICT && a.premark && a.premark[e] != 0))) || QuadDpp::env_ballot(cnt.overflow > 0 || cnt.need_full > 0) != 0u;
      if (leave) {
        if (FUSED) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // what this wave stored in the launch's earlier control steps
        if (valid && c == 0) {
          a.stall[e] = fused;
          // substep s - 1 is the one that did not fit: its start state was stored at the top of that iteration (below)
          if (a.hsub) a.hsub[e] = (resume_on && s >= 2) ? s - 1 : 0;
#ifdef LM_TIMERS
          if (a.tline) { a.tline[4 * e] = wall_clock64(); a.tline[4 * e + 3] = (resume_on && s >= 2) ? s - 1 : 0; }
#endif
          __threadfence();
          const int k = atomicAdd(&a.replay_ctl[0], 1);
          atomicExch(&a.replay_list[k], e + 1);
        }
        gone = true; valid = false;
      } else if (RESUME && resume_on && s >= 1 && valid) {
        // the start state of substep s, for a replay kernel that may have to take the control step over from here
        if (c == 0) {
#pragma unroll
          for (int i = 0; i < 6; i++) { a.hq[dr[i] * N + e] = qr[i]; a.hv[dr[i] * N + e] = vr[i]; a.hw[dr[i] * N + e] = war[i]; }
        }
#pragma unroll
        for (int k = 0; k < MC; k++) if (k < nl) { a.hq[dc[k] * N + e] = qc[k]; a.hv[dc[k] * N + e] = vc[k]; a.hw[dc[k] * N + e] = wac[k]; }
      }
    }
    if (REPLAY ? fused >= first_step : !gone)        // (the replay kernel: the control steps before the one it takes over are the regular kernel's)
      lm::substep<QuadDpp, MC, NS, RK4, CONE, NM, DR, PM>(cm, c, a.P, qr, vr, qc, vc, war, wac, actr, actc, lmem, ls, cnt, nullptr, mt, &dofp, a.T.ngrf > 0, pair_slack);
  }

  QuadDpp::fence();          // the stores below read lane memory that other replicas wrote (muscle activations)

  // ---- the last substep's verdict (see the loop above)
  if (!REPLAY && a.replay_list && !gone) {
    const bool leave = QuadDpp::env_ballot(cnt.overflow > 0 || cnt.need_full > 0) != 0u;
    if (leave) {
      if (FUSED) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (valid && c == 0) {
        a.stall[e] = fused;
        if (a.hsub) a.hsub[e] = (resume_on && a.T.nsub >= 2) ? a.T.nsub - 1 : 0;      // the LAST substep did not fit
#ifdef LM_TIMERS
        if (a.tline) { a.tline[4 * e] = wall_clock64(); a.tline[4 * e + 3] = (resume_on && a.T.nsub >= 2) ? a.T.nsub - 1 : 0; }
#endif
        __threadfence();
        const int k = atomicAdd(&a.replay_ctl[0], 1);
        atomicExch(&a.replay_list[k], e + 1);
      }
      gone = true; valid = false;
    }
  }

  // ---- validity flags of this control step: where the device left its collision model (tests and statistics)
  bool need_unsim = false;
  {
    // (with the replay switched off — lm_batch_set_replay(b, 0), an A/B mode — what the regular kernel leaves to the replay kernel is not
    // simulated at all: a convex pair of the quadruped within reach, a root dof of the muscle humanoid beyond its limit. Flagged and
    // counted like a pair without a collider)
    const float f_need = (!REPLAY && !a.replay_list && QuadDpp::env_ballot(cnt.need_full > 0) != 0u) ? 1.0f : 0.0f;      // (any replica may have seen it)
    const float f_over = QuadDpp::sum((float)cnt.overflow), f_prox = QuadDpp::sum((float)cnt.selfprox) + f_need, f_unh = QuadDpp::sum((float)cnt.unhandled);
    need_unsim = f_need > 0.0f;
    if (a.flags && c == 0 && valid) a.flags[e] = (unsigned char)((f_over > 0.0f ? 1 : 0) | (f_prox > 0.0f ? 2 : 0) | (f_unh > 0.0f ? 4 : 0));
  }

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
  // done byte: bit 0 = absorbing state; bit 1 = the episode ended in this step on the device's side (restarted from the
  // reset table — the observation written below is then the first of the NEW episode — or the horizon was reached)
  // Without device-side restarts bit 1 is set in the ONE step that reaches the horizon, not in every later one.
  const bool restarts = a.auto_reset && a.table_rows > 0;
  const unsigned char done_byte = (unsigned char)((absorbing ? 1 : 0) | (((restarts && (trunc || absorbing)) || (!restarts && step_no == a.horizon)) ? 2 : 0));
  if (absorbing || trunc) {
    // without device-side restarts an episode that runs past its horizon (or stays absorbed) is counted once
    episodes = (a.auto_reset && a.table_rows > 0) || step_no == a.horizon || (absorbing && !trunc) ? 1.0f : 0.0f;
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
      if (DR == 2 && (a.vdirty || a.nvar > 1) && c == 0 && valid) {        // (model compiler: one slot per environment — a batch of ONE has nvar == 1)
        // new episode, new model variant (reference base.py:183-185: a freshly randomised model per reset)
        const unsigned long long rv = mix64(a.seed ^ mix64((unsigned long long)gid * 2ull + 1ull) ^ ((unsigned long long)ec << 32) ^ 0xA24BAED4963EE407ull);
        if (a.vdirty) a.vdirty[e] = 1;
        else a.var[e] = (a.var_rows > 0) ? (int)((r % (unsigned long long)a.table_rows) / (unsigned long long)a.var_rows)      // the model of the row drawn above
                                    : (int)(rv % (unsigned long long)a.nvar);
      }
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
            const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);        // N(a, b)
            v = fmaxf(fmaf(sp[2], z, sp[1]), 0.0f);      // both normal kinds are clipped at 0 (a joint parameter cannot be negative)
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
    if (a.done) a.done[e] = done_byte;
  }
#pragma unroll
  for (int k = 0; k < MC; k++) if (k < nl) { a.qpos[dc[k] * N + e] = qc[k]; a.qvel[dc[k] * N + e] = vc[k]; a.warm[dc[k] * N + e] = wac[k]; }
  if (PAIRS && a.slack) {
    const bool fresh = step_no == 0 || nonfinite;       // the episode restarted (or the state was zeroed): nothing is known about its pairs
#pragma unroll
    for (int j = 0; j < 3; j++) a.slack[((long long)j * 4 + c) * N + e] = fresh ? 0.0f : pair_slack[j];
  }
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
      if (MC == 6) {          // UnitreeG1: four force points per foot
        const int o2 = (int)cm[LM_CM_CHAINS + LM_C_GRF_OBS2 * LM_NCHAIN + c], o3 = (int)cm[LM_CM_CHAINS + LM_C_GRF_OBS3 * LM_NCHAIN + c];
#pragma unroll
        for (int j = 0; j < 3; j++) { if (o2 >= 0) o[o2 + j] = cnt.grf[2][j] * scale; if (o3 >= 0) o[o3 + j] = cnt.grf[3][j] * scale; }
      }
    }
  }
  }

  // ---- REPLAY: will the next control step of this environment need this kernel again? (see KArgs::premark)
  if (REPLAY && MC >= 5 && a.premark) {
    const bool big = QuadDpp::env_ballot(cnt.peak_slots > a.reg_ns || cnt.peak_q > a.reg_q || cnt.peak_res > a.reg_r) != 0u;
    if (valid && c == 0) a.premark[e] = (big && step_no != 0 && !nonfinite) ? 1 : 0;
  }

  // ---- statistics: LDS adds inside the workgroup, one plain read-modify-write per workgroup slot (no global atomics)
  if (a.stats) {
    if (valid) {
      if (c == 0) {
        atomicAdd(&blk_stats[0], 1.0f); atomicAdd(&blk_stats[1], episodes); atomicAdd(&blk_stats[2], reward);
        atomicAdd(&blk_stats[3], nonfinite ? 1.0f : 0.0f); atomicAdd(&blk_stats[4], (float)cnt.solver_iters);
        atomicAdd(&blk_stats[7], (float)cnt.ls_evals); atomicAdd(&blk_stats[8], (float)cnt.ls_capped);
        atomicAdd(&blk_stats[9], cnt.it_max >= 8 ? 1.0f : 0.0f);
        if (need_unsim) atomicAdd(&blk_stats[10], 1.0f);
      }
      if (cnt.overflow) atomicAdd(&blk_stats[5], (float)cnt.overflow);
      if (cnt.unhandled) atomicAdd(&blk_stats[6], (float)cnt.unhandled);
      if (PAIRS && cnt.selfprox) atomicAdd(&blk_stats[10], (float)cnt.selfprox);

      if (PAIRS && cnt.selfcon) atomicAdd(&blk_stats[11], (float)cnt.selfcon);
      if (PAIRS && cnt.natown) atomicAdd(&blk_stats[13], (float)cnt.natown);
      if (REPLAY && c == 0) { atomicAdd(&blk_stats[12], 1.0f); if (a.replay_mark) a.replay_mark[e] = 1; }
    }
#ifdef LM_TIMERS
    if (threadIdx.x == 0) for (int i = 0; i < 16; i++) atomicAdd(&a.timers[i], (unsigned long long)cnt.t[i]);
    for (int i = 0; i < 16; i++) if (cnt.m[i]) atomicAdd(&a.timers[16 + 32 * (long long)gridDim.x + i], (unsigned long long)cnt.m[i]);        // every lane
    {
      unsigned long long* rec = a.timers + 16 + 16 * (long long)wg;       // wg: the workgroup after the XCD mapping (environments 4 wg .. 4 wg + 3)
      const float ncon_env = QuadDpp::sum((float)cnt.ncon);
      if (threadIdx.x == 0) { long long tot = 0; for (int i = 0; i < 16; i++) tot += cnt.t[i]; rec[0] = (unsigned long long)tot; }
      if (threadIdx.x == 0) for (int i = 0; i < 16; i++) a.timers[16 + 16 * (long long)gridDim.x + 16 * (long long)wg + i] = (unsigned long long)cnt.t[i];
      if (c == 0 && QuadDpp::rep() == 0 && e_local < 4) {
        rec[1 + e_local] = (unsigned long long)cnt.solver_iters; rec[5 + e_local] = (unsigned long long)ncon_env;
        rec[9 + e_local] = (unsigned long long)cnt.ls_evals; rec[13 + (e_local & 1)] = (unsigned long long)(absorbing ? 1 : 0);
      }
    }
#endif
  }
  }  // fused control steps
#ifdef LM_TIMERS
  if (REPLAY && a.tline && threadIdx.x == 0) a.tline[4 * e_raw + 2] = wall_clock64();
#endif
  }  // REPLAY: list entries of this workgroup
  if (a.stats) {
    __syncthreads();
    for (int i = threadIdx.x; i < kNStats; i += blockDim.x) {
      float* dst = reinterpret_cast<float*>(a.stats + a.stats_off + blockIdx.x) + i;
      *dst += blk_stats[i];
    }
  }
#ifdef LM_TIMERS
  if (!REPLAY && !FORWARD_ONLY && a.tline && threadIdx.x == 0) a.tline[4 * (long long)a.N + 2 * blockIdx.x + 1] = wall_clock64();
#endif
  if (!REPLAY && !FORWARD_ONLY && a.replay_list) {
    // this regular workgroup is through: the pollers leave when all are (replay_next)
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence(); atomicAdd(&a.replay_ctl[2], 1); }
  }
  if (REPLAY && a.drain) {
    // the last workgroup of the drain pass resets the control words for the next launch (the list itself is all zeros again:
    // every entry was taken) and leaves the number of entries for the host
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const int tail = a.replay_ctl[0];
      if (atomicAdd(&a.replay_ctl[3], 1) == (int)gridDim.x - 1) { if (a.host_hint) { a.host_hint[0] = tail; a.host_hint[1] = a.epoch + 1; a.host_hint[2] = a.replay_ctl[7]; } a.replay_ctl[0] = 0; a.replay_ctl[1] = 0; a.replay_ctl[2] = 0; a.replay_ctl[3] = 0; __threadfence(); __hip_atomic_store(&a.replay_ctl[5], a.epoch + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
    }
  }
#undef RD
#undef LK
}

// ---- launch helpers ---------------------------------------------------------------------------------------------
struct LaunchCtx { hipStream_t stream; int N, epb; };

// kernel kinds of one family (picked by the host, lm_kernels.hip::launch_variant)
enum { LMK_FWD = 0, LMK_REP4, LMK_REP1, LMK_DR_REP4, LMK_DR_REP1, LMK_FUSED, LMK_FUSED_DR, LMK_DRV_REP4, LMK_DRV_REP1, LMK_FUSED_DRV,
       LMK_BIG, LMK_BIG_DR, LMK_BIG_DRV /* the replay kernels, one per part */, LMK_NKINDS };
constexpr int LMK_NFAMILY = 11;     // 0 quadruped, 2 humanoid RK4 8 slots, 4 Euler 8 slots, 5 muscles, 6 generic, 7 six-link chains (Euler, 8 slots), (1 / 3: the four-slot humanoid families, dropped in round 5)
                                    // 8 / 9 / 10 = five-link humanoids WITH self-collisions (8 slots): RK4 | Euler | Euler + muscles
// Replicas of the replay kernels' ONE environment per workgroup. 4 (shipped): the regular kernels' arithmetic exactly — a control step
// comes out bitwise the same from either kernel. 16 (-DLM_REPLAY_REP=16): the whole wave for the environment, everything that is dealt
// over replicas / lanes dealt four times wider. Measured in round 5 (profiles/r5_notes.md §3): a hard HumanoidTorque costs 10.4 ms with
// sixteen replicas and 10.2 ms with four — what it waits for is the latency of single convex pairs and of spilled registers, not lanes.
#ifndef LM_REPLAY_REP
#define LM_REPLAY_REP 4
#endif
constexpr int kReplayRep = LM_REPLAY_REP;
constexpr int kReplayGrid = 128;    // workgroups of the replay kernel's drain pass (each walks the list with this stride), and the most pollers
constexpr int kPollMul = 2, kPollAdd = 2, kPollCap = 32;      // pollers of a launch = kPollMul / 2 x (recently abandoned steps) + kPollAdd, at most kPollCap (lm_kernels.hip)

template <class K>
static void launch_one(K kernel, dim3 grid, dim3 block, size_t lane_floats, const LaunchCtx& L, const KArgs& a) {
  // the workgroup's LDS = constant table (the part the model uses) + lane memory, both dynamic; opt in to more than
  // the default 64 KB cap
  const size_t bytes = sizeof(float) * ((size_t)a.T.cm_used + lane_floats);
  hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  hipLaunchKernelGGL(kernel, grid, block, bytes, L.stream, a);
}

// one robot family = (links per chain MC, contact slots per chain NS, integrator, compiled-in cone, muscles per chain NM, pair
// pass PM of the regular kernels: 0 none, 1 with the convex collider, 2 without — the replay kernel has it)
template <int MC, int NS, bool RK4, int CONE, int NM, int PART, int PM = 0>
static bool launch_family(const LaunchCtx& L, const KArgs& a, int kind) {
  const dim3 grid((L.N + L.epb - 1) / L.epb);
  using LMm = lm::LaneMemFor<MC, NS, NM, (PM == 1 || PM == 2), CONE>;       // (as the kernel's own alias: detection-only kernels carry no pair extension)
  const size_t plain = (size_t)LMm::kGroup * ((4 * L.epb + 15) / 16), rep = (size_t)LMm::kGroup;
  // the replay kernel: 128 contact slots per chain (one environment per workgroup: its whole LDS), the convex collider,
  // fused (it finishes the launch's control steps of its environments)
  constexpr int NSB = 128, PMB = (PM != 0) ? 1 : 0;
  using LMb = lm::LaneMemFor<MC, NSB, NM, (PM != 0), CONE>;
  if (kind == LMK_BIG || kind == LMK_BIG_DR || kind == LMK_BIG_DRV) {
    if (kind != LMK_BIG + PART) return false;
    KArgs b = a;
    b.epb = 1; b.xcd_map = 0;
    // L.epb carries the number of workgroups asked for here (pollers: a few; the drain pass: kReplayGrid). The statistics slots are
    // one per workgroup of the REGULAR launch (+ kReplayGrid for the pollers): not more workgroups than that
    const int ngroups = (int)((L.N + a.epb - 1) / a.epb), want = L.epb;
    launch_one(step_kernel<MC, NSB, RK4, false, CONE, NM, PART, kReplayRep, true, PMB>, dim3(ngroups < want ? ngroups : want), dim3(4 * kReplayRep), (size_t)LMb::kPadded * 4, L, b);
    return true;
  }
  if constexpr (PART == 0) {
    // the forward-only (debug) kernel reads the cone at run time: full slot records (and always carries the convex collider)
    if (kind == LMK_FWD) launch_one(step_kernel<MC, NS, RK4, true, -1, NM, 0, 1, false, PMB>, grid, dim3(4 * L.epb),
                                    (size_t)lm::LaneMemFor<MC, NS, NM, (PM != 0), -1>::kGroup * ((4 * L.epb + 15) / 16), L, a);
    else if (kind == LMK_REP4) launch_one(step_kernel<MC, NS, RK4, false, CONE, NM, 0, 4, false, PM>, grid, dim3(16 * L.epb), rep, L, a);
    else if (kind == LMK_REP1) launch_one(step_kernel<MC, NS, RK4, false, CONE, NM, 0, 1, false, PM>, grid, dim3(4 * L.epb), plain, L, a);
    else return false;
  } else if constexpr (PART == 1) {
    if (kind == LMK_DR_REP4) launch_one(step_kernel<MC, NS, RK4, false, CONE, NM, 1, 4, false, PM>, grid, dim3(16 * L.epb), rep, L, a);
    else if (kind == LMK_DR_REP1) launch_one(step_kernel<MC, NS, RK4, false, CONE, NM, 1, 1, false, PM>, grid, dim3(4 * L.epb), plain, L, a);
    else if (kind == LMK_FUSED) launch_one(step_kernel<MC, NS, RK4, false, CONE, NM, 0, 4, true, PM>, grid, dim3(16 * L.epb), rep, L, a);
    else if (kind == LMK_FUSED_DR) launch_one(step_kernel<MC, NS, RK4, false, CONE, NM, 1, 4, true, PM>, grid, dim3(16 * L.epb), rep, L, a);
    else return false;
  } else {
    // per-environment joint parameters AND model variants (lm_set_model_variants)
    if (kind == LMK_DRV_REP4) launch_one(step_kernel<MC, NS, RK4, false, CONE, NM, 2, 4, false, PM>, grid, dim3(16 * L.epb), rep, L, a);
    else if (kind == LMK_DRV_REP1) launch_one(step_kernel<MC, NS, RK4, false, CONE, NM, 2, 1, false, PM>, grid, dim3(4 * L.epb), plain, L, a);
    else if (kind == LMK_FUSED_DRV) launch_one(step_kernel<MC, NS, RK4, false, CONE, NM, 2, 4, true, PM>, grid, dim3(16 * L.epb), rep, L, a);
    else return false;
  }
  return true;
}

// defined in the lm_family.hip objects; false = this family/part has no kernel of that kind
typedef bool (*family_fn)(const LaunchCtx&, const KArgs&, int kind);
bool launch_f0p0(const LaunchCtx&, const KArgs&, int); bool launch_f0p1(const LaunchCtx&, const KArgs&, int); bool launch_f0p2(const LaunchCtx&, const KArgs&, int);
bool launch_f2p0(const LaunchCtx&, const KArgs&, int); bool launch_f2p1(const LaunchCtx&, const KArgs&, int); bool launch_f2p2(const LaunchCtx&, const KArgs&, int);
bool launch_f4p0(const LaunchCtx&, const KArgs&, int); bool launch_f4p1(const LaunchCtx&, const KArgs&, int); bool launch_f4p2(const LaunchCtx&, const KArgs&, int);
bool launch_f5p0(const LaunchCtx&, const KArgs&, int); bool launch_f5p1(const LaunchCtx&, const KArgs&, int); bool launch_f5p2(const LaunchCtx&, const KArgs&, int);
bool launch_f6p0(const LaunchCtx&, const KArgs&, int); bool launch_f6p1(const LaunchCtx&, const KArgs&, int); bool launch_f6p2(const LaunchCtx&, const KArgs&, int);
bool launch_f7p0(const LaunchCtx&, const KArgs&, int); bool launch_f7p1(const LaunchCtx&, const KArgs&, int); bool launch_f7p2(const LaunchCtx&, const KArgs&, int);
bool launch_f8p0(const LaunchCtx&, const KArgs&, int); bool launch_f8p1(const LaunchCtx&, const KArgs&, int); bool launch_f8p2(const LaunchCtx&, const KArgs&, int);
bool launch_f9p0(const LaunchCtx&, const KArgs&, int); bool launch_f9p1(const LaunchCtx&, const KArgs&, int); bool launch_f9p2(const LaunchCtx&, const KArgs&, int);
bool launch_f10p0(const LaunchCtx&, const KArgs&, int); bool launch_f10p1(const LaunchCtx&, const KArgs&, int); bool launch_f10p2(const LaunchCtx&, const KArgs&, int);

}  // namespace lmk
