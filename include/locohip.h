/*
 * locohip.h — C-ABI of the MI355X batched locomotion step ("liblocohip.so").
 *
 * The reference has no C boundary on this path: LocoEnv.step() is mushroom-rl's MuJoCo.step, which
 * calls into the MuJoCo C library through its pybind module. The entry points
 * below are what a loco-mujoco maintainer would bind with ctypes to replace, for
 * a whole batch of environments at once:
 *
 *   lm_model_create     <- MjModel.from_xml_string + MultiMuJoCo.__init__ bookkeeping (model upload)
 *                          (/root/reference/loco_mujoco/environments/base.py:109-126;
 *                           obs/action spec unitreeA1.py:778-854)
 *   lm_batch_create     <- mujoco.MjData(model), one per environment (base.py:185)
 *   lm_set_state        <- LocoEnv.set_sim_state: data.joint(name).qpos/qvel = value (base.py:478-497)
 *                          after mj_resetData (base.py:180): also clears the solver warm start
 *   lm_get_state        <- data.qpos / data.qvel reads (ObservationHelper._build_obs, base.py:202)
 *   lm_set_dof_params / lm_set_dof_randomization / lm_set_model_variants <- DomainRandomizationHandler.get_randomized_model
 *                          (utils/domain_randomization.py:219-227): joint damping/stiffness/frictionloss per environment
 *   lm_set_model_compiler / lm_compile_models / lm_get_model_draws / lm_get_model_tables <- the reference's re-compile of a
 *                          randomised XML at every reset (base.py:183-185, utils/domain_randomization.py:219-227,386-514: armature,
 *                          Inertial mass / diaginertia / fullinertia, Geoms friction), drawn and compiled on the device
 *   lm_set/get_activation <- data.act (muscle activation state, humanoids.py:320 HumanoidMuscle; integrated by mj_step)
 *   lm_set_goal         <- per-episode goal written into the observation
 *                          (unitreeA1.py:288-291 set_goal, :454-476 _create_observation)
 *   lm_step             <- MuJoCo.step: _preprocess_action (base.py:606-621) -> ctrl ->
 *                          mujoco.mj_step(model, data, n_substeps) -> _create_observation
 *                          (base.py:584-604, unitreeA1.py:454-476) -> is_absorbing/_has_fallen
 *                          (base.py:243-255, unitreeA1.py:503-536) -> reward (base.py:170-176,
 *                          utils/reward.py:66-117, evaluated on the previous observation)
 *   lm_set_reset_table  <- Trajectory.reset_trajectory + set_sim_state for finished episodes
 *                          (utils/trajectory.py:236-273, base.py:178-203), done on the device
 *   lm_rollout          <- the user's `for step in range(n): env.step(a)` loop
 *                          (tests/test_environments.py:15-38), kept on the device for benchmarking
 *   lm_rollout_fused    <- the same loop with several control steps per kernel launch (policy-free only)
 *   lm_forward_debug    <- mujoco.mj_forward (base.py:362) with intermediate results, for parity tests
 *
 * All arrays at the boundary are caller-owned HOST buffers, row-major [n_envs][dim], float32.
 * Device memory lives behind the opaque handles. Functions return 0 on success, non-zero on error;
 * lm_last_error() returns a message for the calling thread. One handle = one GPU; handles are not
 * thread-safe (one host thread per handle), calls on a handle are stream-ordered.
 */
#ifndef LOCOHIP_H
#define LOCOHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lm_model lm_model;
typedef struct lm_batch lm_batch;

typedef struct {
  int nq, nv, nu, nobs, ngoal, n_substeps, n_chains, max_chain_dofs;
  int na;                  /* activation states per environment (muscles); 0 for torque-driven models */
} lm_dims;

typedef struct {
  double env_steps;        /* control steps executed, summed over environments */
  double episodes;         /* episodes finished (absorbing or horizon) */
  double reward_sum;       /* sum of rewards */
  double nan_resets;       /* environments reset because their state became non-finite */
  double solver_iters;     /* Newton iterations, summed over environments and substeps */
  double overflow_contacts;/* contacts dropped because a per-chain contact slot budget was exceeded */
  double unhandled_geoms;  /* substeps in which a geom without a device collider came within margin */
  double linesearch_evals; /* line-search function evaluations after the first, summed like solver_iters */
  double linesearch_capped;/* line searches that ran into the iteration cap */
  double steps_with_8plus_iters; /* env-steps in which some substep needed >= 8 Newton iterations */
  double kernel_ms;        /* HIP-event time of the step kernels of this call (rollout only) */
  double self_proximity;   /* forward passes x geom pairs WITHOUT a collider (a box / cylinder against another geom of the robot)
                              within the contact margin: the state was outside the validated collision domain */
  double self_contacts;    /* self-contacts simulated (sphere / capsule pairs in closed form, the engine's native box / cylinder colliders,
                              convex pairs of two links through MPR), summed over the forward passes */
  double replayed_env_steps; /* env-steps that left the regular kernel's capacity (contact slots, pair lists, convex collider) and were
                              run by the family's replay kernel instead (lm_batch_set_replay); part of env_steps */
  double own_manifold_contacts; /* of self_contacts: contacts of box-box and capsule-box pairs. Their manifold construction is this
                              library's own (the engine's mjc_BoxBox / mjc_CapsuleBox case analysis is not restated: csrc/lm_core.h nat_*):
                              exact where the geometry leaves no choice (pinned edge-edge case), approximate for face contacts */
} lm_stats;

typedef struct {
  /* all optional (NULL = skip); [n_envs][...] row-major float32 */
  float* M;            /* [nv*nv]  joint-space inertia */
  float* qfrc_bias;    /* [nv] */
  float* qfrc_smooth;  /* [nv]  passive - bias + actuator */
  float* qacc_smooth;  /* [nv] */
  float* qacc;         /* [nv]  after the constraint solve */
  float* qfrc_constraint; /* [nv] */
  int* ncon;           /* [1] active contacts */
  int* solver_iter;    /* [1] */
} lm_forward_out;

int lm_device_count(void);
const char* lm_last_error(void);

/* chain_model: float64 array laid out as in lm_layout.h (header H_* + constant table), produced by the
   host-side mini-compiler + lowering (loco_mujoco_amd/mjcf.py, lowering.py) from the MJCF model and the
   task description (observation/action spec, termination bounds, reward). */
int lm_model_create(const double* chain_model, size_t n, int device, lm_model** out);
void lm_model_destroy(lm_model* m);
int lm_model_dims(const lm_model* m, lm_dims* out);

int lm_batch_create(lm_model* m, int n_envs, lm_batch** out);
void lm_batch_destroy(lm_batch* b);
/* Launch geometry (no counterpart in the reference: its step is one MjData at a time, base.py:185). envs_per_workgroup = 4: one wave
   per 4 environments, each replicated over 4 quads (default); 8 or 16: plain layout without replicas. Same physics either way. */
int lm_batch_set_layout(lm_batch* b, int envs_per_workgroup);
/* Speculate / replay (no counterpart in the reference: the engine it calls sizes its contact buffers for everything,
   environments/data/humanoid/humanoid_torque.xml:19 njmax 1000 / nconmax 400, data/atlas/atlas.xml:23). The regular step kernels hold a few
   contact slots per chain; a control step that needs more is abandoned unstored and run by the family's replay kernel (a slot for
   every contact, long pair lists, the convex collider): as a few polling workgroups beside the regular launch (second stream; only
   when recent launches abandoned steps) and as a pass behind it. enabled = 1 (default); 0 = regular kernels only: contacts beyond
   the slots are dropped and counted in overflow_contacts, and what the regular kernels leave to the replay kernel altogether (the
   quadruped's convex pairs and capsule-box pairs, root-joint limits of the muscle humanoid) is not simulated: flag bit 2 and
   self_proximity say when (A/B measurements); 2 = every control step through the replay kernel
   (tests); 3 / 4 = 1 / 2 without the pollers (for profilers that run one kernel at a time: pollers would wait out their 0.5 s). */
int lm_batch_set_replay(lm_batch* b, int enabled);
/* one byte per environment: 1 = the replay kernel ran at least one control step of this environment since the marks were last
   cleared (reset = 1 clears them after the copy; out may be NULL). Diagnostics of THIS path, like lm_get_flags. */
int lm_get_replay_marks(lm_batch* b, uint8_t* out, int reset);

/* mask: [n_envs] bytes, NULL = all environments. Setting a state clears that env's warm start. */
int lm_set_state(lm_batch* b, const float* qpos, const float* qvel, const uint8_t* mask);
int lm_get_state(lm_batch* b, float* qpos, float* qvel);
int lm_set_goal(lm_batch* b, const float* goal, const uint8_t* mask);
/* muscle activations [n_envs][na] (data.act of the reference: mj_resetData zeroes it, base.py:180, so lm_set_state
   and device-side episode restarts zero it too; these two exist for checkpointing and for parity tests) */
/* per-environment joint damping / stiffness / frictionloss [n_envs][nv] (NULL = leave as is) — what the reference's
   domain randomisation changes per episode by re-compiling the model (utils/domain_randomization.py:299-383,
   base.py:183-185). The first call allocates the arrays (initialised with the model's values) and switches the batch
   to the kernel variant that reads them. */
int lm_set_dof_params(lm_batch* b, const float* damping, const float* stiffness, const float* frictionloss,
                      const uint8_t* mask);
int lm_get_dof_params(lm_batch* b, float* damping, float* stiffness, float* frictionloss);
/* redraw rule used when the device restarts an episode (lm_set_auto_reset): spec[3][nv][3] = (kind, a, b) per parameter
   (damping, stiffness, frictionloss) and dof; kind 0 keep, 1 max(N(a,b),0), 2 U(a,b), 3 N(a,b). NULL disables. */
int lm_set_dof_randomization(lm_batch* b, const float* spec);
/* model variants: what the reference's domain randomisation changes by re-compiling the model with other inertial / armature /
   geom-friction numbers (utils/domain_randomization.py:386-514 set_geom_conf / set_inertial_conf, base.py:183-185). The host
   lowers n_variants randomised models (lowering.variant_tables) and hands over, per variant, the inertial record
   [LM_IR_SIZE][4], the geom table [LM_GT_SIZE] and the geom-pair table [pair_floats] (NULL when the model has none). Every
   environment holds the index of its variant (0 after this call); a device-side restart redraws it uniformly. Switches the
   batch to the kernel variant with per-environment parameters, like lm_set_dof_params. n_variants = 0 removes the pool. */
int lm_set_model_variants(lm_batch* b, const float* records, const float* geom_tables, const float* pair_tables,
                          int pair_floats, int n_variants);
int lm_set_variant_index(lm_batch* b, const int32_t* index, const uint8_t* mask);
int lm_get_variant_index(lm_batch* b, int32_t* index);
/* several MODELS of one environment as variants (the reference's MultiMuJoCo draws a model per episode, base.py:186-190): the
   reset table holds n_variants blocks of rows_per_variant rows (each block with its model's constants in the goal columns);
   a device-side restart from row i then puts the environment on variant i / rows_per_variant. 0 = independent uniform redraw. */
int lm_set_variant_rows(lm_batch* b, int rows_per_variant);
/* the model compiler on the device: a FRESHLY randomised model per environment and episode, like the reference's re-compile at
   every reset() (base.py:183-185, utils/domain_randomization.py:219-227,386-514). `index` / `data` = the program of
   lowering.model_compiler_tables (the draws of the randomisation rules, the model's Jacobians at qpos0, where every derived
   value lands); `record` / `geom_table` / `pair_table` = the nominal model's tables (lm_set_model_variants' layout, ONE set).
   The batch then holds one slot per environment; this call and lm_compile_models (the host-side reset; mask NULL = everyone)
   draw and compile on the device, and so does every device-side restart (lm_set_auto_reset) before the episode's first step:
   armature / mass / diaginertia / fullinertia / geom friction drawn with a counter-based generator keyed by (seed, global
   environment id, models this environment has had), M(qpos0) -> dof_invweight0 / body_invweight0 / meaninertia in float64.
   Fused launches fall back to one control step per launch. lm_set_model_variants replaces the compiler by a pool again. */
int lm_set_model_compiler(lm_batch* b, const int32_t* index, long long n_index, const double* data, long long n_data,
                          const float* record, const float* geom_table, const float* pair_table, int pair_floats, uint64_t seed);
int lm_compile_models(lm_batch* b, const uint8_t* mask);
/* what the compiler drew for the CURRENT model of every environment, [N][n_draw] in the order of the program's draws, and how
   many models each environment has had (either may be NULL): the host (and the oracle) rebuild that model from these */
int lm_get_model_draws(lm_batch* b, double* draws, uint32_t* generation);
/* the tables environment `env` runs on (its variant's slot; any of the three may be NULL) */
int lm_get_model_tables(lm_batch* b, int env, float* record, float* geom_table, float* pair_table);
int lm_set_activation(lm_batch* b, const float* act, const uint8_t* mask);
int lm_get_activation(lm_batch* b, float* act);

/* one control step for every environment. action in [-1,1] (normalised, base.py:606-621).
   obs [n_envs][nobs], reward [n_envs], done [n_envs] may each be NULL. Synchronous.
   The done byte: bit 0 (value 1) = absorbing state (the reference's `absorbing`: is_absorbing(obs), base.py:274-282);
   bit 1 (value 2) = the episode ended in this step on the device's side — it was restarted from the reset table (the
   observation written is then the first of the NEW episode), or, without device-side restarts, this is the step that reached
   the horizon. `done & 1` is what the reference's step() returns; `done != 0` mixes truncation into it. */
int lm_step(lm_batch* b, const float* action, float* obs, float* reward, uint8_t* done);

/* the same step with DEVICE pointers (action [n_envs][nu], obs [n_envs][nobs], reward [n_envs], done [n_envs]; any may be
   NULL) for training loops that keep policy inputs/outputs on the GPU: no PCIe traffic. `stream` = a hipStream_t to run
   on (NULL: the library's own stream); with sync = 0 the call returns after the launch. The caller orders its own work
   against that stream. */
int lm_step_device(lm_batch* b, const float* d_action, float* d_obs, float* d_reward, uint8_t* d_done, void* stream, int sync);

/* ACTIVE LIST: from now on every step / rollout of this batch runs only the listed environments (ids in [0, n_envs), each at most
   once; count 0: nothing runs), in one launch of ceil(count / environments per workgroup) workgroups; the others keep their state, and
   what the step writes for them (observation, reward, done) is left as it was. NULL: all environments again. Host buffers keep their
   [n_envs][...] shape — an environment keeps its row. For environments that share ids across several models and change model per
   episode (the reference's MultiMuJoCo with models that differ in geometry, base.py:186-190: HumanoidTorque4Ages "all"): one batch per
   model, each stepping the environments currently of its size. Random numbers stay keyed by the global environment id. Compiled into every
   kernel family but the quadruped's (one model per batch there; the indirection costs its bench kernel 0.9 %): refused for it. */
int lm_batch_set_active(lm_batch* b, const int32_t* env_ids, int count);

/* LocoEnv.step()'s host surface in ONE call (reference gymnasium.py:47-65 -> base.py step(): numpy float64 action in, float64
   observation / reward and the absorbing flag out — the library's counterpart of the dtype conversions and copies the Python
   layer did around lm_step, environments/base.py). The results land in PINNED host memory owned by the batch: a ring of
   LM_PINNED_SLOTS result sets, so that what step t returned stays intact while steps t+1 .. t+LM_PINNED_SLOTS-1 run.
     lm_pinned_slot   pointers to result set `slot` (allocated at the first call): obs [n_envs][nobs] float64 in the column order of
                      lm_set_obs_order, reward [n_envs] float64, done [n_envs] (the done byte of lm_step)
     lm_set_obs_order column j of the float64 observation = column perm[j] of the kernel's (the reference's observation order where
                      it differs from the device's, base.py _obs_perm); NULL: the kernel's order
     lm_step_pinned   one control step: the action (float64, pageable or not) is converted into a pinned staging buffer, which the step
                      kernel reads itself; a small kernel behind it converts observation / reward to float64 (applying the order)
                      straight into the pinned slot — no copy is queued on either side of the launch; synchronous */
#define LM_PINNED_SLOTS 4
int lm_pinned_slot(lm_batch* b, int slot, double** obs, double** reward, uint8_t** done);
int lm_set_obs_order(lm_batch* b, const int32_t* perm, int n);
int lm_step_pinned(lm_batch* b, const double* action, int slot);

/* device-side episode handling: rows = [qpos(nq) | qvel(nv) | goal(ngoal)]; when enabled, an
   environment whose step ended absorbing (or reached `horizon` control steps, 0 = never) restarts
   from a row drawn with a counter-based RNG keyed by (seed, global env id, episode count) and the
   observation returned for that step is the fresh one. */
int lm_set_reset_table(lm_batch* b, const float* rows, int n_rows, uint64_t seed, int64_t global_env_offset);
int lm_set_auto_reset(lm_batch* b, int enabled, int horizon);

/* n_steps control steps entirely on the device. action_mode 0: zero action; 1: a ~ U(-1,1)^nu from
   the counter-based RNG. Accumulates into *stats (may be NULL). */
int lm_rollout(lm_batch* b, int n_steps, int action_mode, uint64_t seed, lm_stats* stats);

/* The same rollout with `steps_per_launch` control steps per kernel launch: every environment advances on its own, without
   the device-wide join that ends each single-step launch with its slowest environment. Bitwise the same states,
   observations and statistics as lm_rollout (which is steps_per_launch = 1); only for the policy-free action modes, a
   policy in the loop needs lm_step / lm_step_device. */
int lm_rollout_fused(lm_batch* b, int n_steps, int steps_per_launch, int action_mode, uint64_t seed, lm_stats* stats);

/* Validity flags of the LAST control step, one byte per environment: 1 = a contact was dropped (the chain's contact slots were
 * full), 2 = two geoms of the robot WITHOUT a pair collider (a box or cylinder against another geom) came within the contact
 * margin, 4 = a geom without a floor collider (mesh without hull) reached the floor. 0 = the step stayed inside the collision model
 * that the parity tests validate. No counterpart in the reference (MuJoCo collides everything): statistics of THIS path. */
int lm_get_flags(lm_batch* b, uint8_t* out);

/* one forward-dynamics pass at the current state with `action`, without advancing it */
int lm_forward_debug(lm_batch* b, const float* action, lm_forward_out* out);

/* the compiler the library was built with and its optimisation level ("HIP version: ...;AMD clang version ... | -Os"), recorded by
   csrc/Makefile: no counterpart in the reference; the kernels sit at the 512-register ceiling, where one code-generation defect of the
   toolchain was met (csrc/Makefile) — a library built by another compiler should be re-validated (tests/test_abi_exports.py) */
const char* lm_toolchain(void);

int lm_get_stats(lm_batch* b, lm_stats* out, int reset);
int lm_sync(lm_batch* b);

#ifdef __cplusplus
}
#endif
#endif
