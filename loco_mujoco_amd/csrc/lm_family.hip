// lm_family.hip — one kernel family (LM_FAMILY) and part (LM_PART) of the step kernels; see lm_step.h.
// The library links one object per family and part (parts 0..2 of the families 0, 2, 4, 5 and 7..10, parts 0..1 of the generic family 6) so that `make -j`
// builds them in parallel.
#include "lm_step.h"

namespace lmk {
#define LM_CAT2(a, b, c, d) a##b##c##d
#define LM_CAT(a, b, c, d) LM_CAT2(a, b, c, d)

bool LM_CAT(launch_f, LM_FAMILY, p, LM_PART)(const LaunchCtx& L, const KArgs& a, int kind) {
#if LM_FAMILY == 0      // quadruped: thigh (2) + calf (2) + foot (1) floor contacts per leg + one for a self-contact, elliptic cones
#ifndef LM_A1_NS
#define LM_A1_NS 6
#endif
#ifndef LM_A1_PAIRS
// 2: the pair pass WITHOUT the inlined convex collider in the regular kernels: a control step that brings a box / cylinder pair within
// reach is abandoned and run by the family's replay kernel (lm_step.h). Measured at the end of round 4, same box, two runs each
// (tools/probes/r4/ab_a1_variants.sh): 1.558 ms per control step of the bench rollout against 1.629 ms with the collider inlined (1):
// -4.4 % — the native box / cylinder colliders had taken the kernel's scratch from 608 to 896 bytes per lane. The bench rollout
// abandons no control step, a random policy two environments per launch (taken over by the pollers beside the launch). Other
// variants of the same A/B: max-ILP scheduler +3.0 %, iterative-minreg +9.0 %, -O2 +2.0 %, five slots +1.0 %.
#define LM_A1_PAIRS 2
#endif
  return launch_family<3, LM_A1_NS, false, LM_CONE_ELLIPTIC, 0, LM_PART, LM_A1_PAIRS>(L, a, kind);
#elif LM_FAMILY == 2    // five-link humanoids, RK4 (Atlas: two boxes per foot)
  return launch_family<5, 8, true, LM_CONE_PYRAMIDAL, 0, LM_PART>(L, a, kind);
#elif LM_FAMILY == 4    // five-link humanoids, Euler (Talos, the carry tasks)
  return launch_family<5, 8, false, LM_CONE_PYRAMIDAL, 0, LM_PART>(L, a, kind);
#elif LM_FAMILY == 5    // muscle humanoid
  return launch_family<5, 4, false, LM_CONE_PYRAMIDAL, LM_MAXMUS, LM_PART>(L, a, kind);
#elif LM_FAMILY == 7    // UnitreeG1 (two 6-link legs: four 1 mm spheres per foot, two arms that share the torso link), UnitreeH1 with its arms.
#ifndef LM_SIX_PAIRS
// 1: the whole pair pass in the regular kernels (lane memory 32.8 KB + 9.4 KB of constants: THREE workgroups per CU — a batch of 4096
// runs its last quarter of workgroups behind the first finishers). 3: detection only (39.2 KB, four per CU; a self-contact hands the
// control step to the replay kernel): right for gaits — but robots that stumble under a random policy touch themselves in a quarter of
// their control steps, and 1100 replays per launch at one environment per workgroup cost 129 ms per step (measured, round 4).
#define LM_SIX_PAIRS 1
#endif
  return launch_family<6, 8, false, LM_CONE_PYRAMIDAL, 0, LM_PART, LM_SIX_PAIRS>(L, a, kind);
#elif LM_FAMILY == 8    // HumanoidTorque with its bone hulls colliding (RK4): floor + self-contacts in eight slots
  return launch_family<5, 8, true, LM_CONE_PYRAMIDAL, 0, LM_PART, 1>(L, a, kind);
#elif LM_FAMILY == 9    // UnitreeH1: hip-yaw cylinders and link meshes colliding (Euler)
  return launch_family<5, 8, false, LM_CONE_PYRAMIDAL, 0, LM_PART, 1>(L, a, kind);
#elif LM_FAMILY == 10   // HumanoidMuscle with its bone hulls colliding
  return launch_family<5, 8, false, LM_CONE_PYRAMIDAL, LM_MAXMUS, LM_PART, 1>(L, a, kind);
#elif LM_PART == 2
  return false;          // the generic family has no kernels with per-environment parameters
#else
  // generic fallbacks (cone read at run time, plain layout only); kind = LMK_FWD or LMK_REP1; `a.T.max_links`, the
  // integrator pick the instance. PART 0: Euler, PART 1: RK4
  const dim3 grid((L.N + L.epb - 1) / L.epb), block(4 * L.epb);
  const size_t groups = (block.x + 15) / 16;
  const bool big = a.T.max_links > 3;
  constexpr bool RK = LM_PART == 1;
  if (kind == LMK_FWD) {
    if (!big) launch_one(step_kernel<3, 4, RK, true, -1>, grid, block, (size_t)lm::LaneMem<3, 4>::kGroup * groups, L, a);
    else launch_one(step_kernel<5, 8, RK, true, -1>, grid, block, (size_t)lm::LaneMem<5, 8>::kGroup * groups, L, a);
  } else if (kind == LMK_REP1) {
    if (!big) launch_one(step_kernel<3, 4, RK, false, -1>, grid, block, (size_t)lm::LaneMem<3, 4>::kGroup * groups, L, a);
    else launch_one(step_kernel<5, 8, RK, false, -1>, grid, block, (size_t)lm::LaneMem<5, 8>::kGroup * groups, L, a);
  } else return false;
  return true;
#endif
}

}  // namespace lmk
