/* so101.h — C ABI of libso101_hip.so: the MI355X (gfx950) batched step of the SO100/SO101
 * hand-over environments.
 *
 * The reference (tuul-ai/so101_sim) has no FFI for this path: its boundary is the Python API
 *   so101_sim/task_suite.py:103-155   create_task_env(...) -> dm_control composer.Environment
 *   env.reset() / env.step(action)    (dm_control loop; hooks in so101_sim/tasks/base/so100_task.py:266-320,
 *                                      so101_sim/tasks/so100_hand_over.py:238-323)
 * and all arithmetic happens inside mujoco.mj_step (third party).  Each entry point below names the
 * reference call it stands in for.  The Python shim so101_sim_amd/env.py binds these with ctypes
 * (INTEGRATION.md shows the binding a reference maintainer would add).
 *
 * Conventions
 *  - return 0 on success, negative so101_status otherwise; no C++ exception crosses the boundary;
 *    so101_last_error() gives a message for the last failing call on that handle (or on create).
 *  - every array argument is a DEVICE pointer owned by the caller (tensor.data_ptr()); the library
 *    never frees caller memory and keeps bound pointers until so101_destroy / the next bind.
 *  - state arrays are struct-of-arrays with the env index fastest: qpos[nq][N], qvel[nv][N], ...
 *    so that lanes reading one scalar for consecutive envs coalesce.
 *  - `hip_stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); calls are
 *    asynchronous with respect to the host.
 *  - a handle is bound to one device (every entry point makes that device current for the duration of the call
 *    and restores the caller's) and is not thread-safe.
 *  - bound state buffers must stay allocated until so101_destroy or the next so101_bind_state returns: the reset
 *    prefetch reads them asynchronously on an internal stream (both calls drain it).
 */
#ifndef SO101_H_
#define SO101_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SO101_ABI_VERSION 10     /* 10: so101_tree_last_plan; 9: so101_tree_config gains the observation delays, so101_tree_bind_physics_state; 8: the general-tree engine (so101_tree_*) */
#define SO101_OBS_DIM 18      /* joints_pos(6, delayed) | undelayed_joints_pos(6) | commanded_joints_pos(6) */
#define SO101_ACT_DIM 6
#define SO101_SOLVER_PGS 0
#define SO101_SOLVER_NEWTON 1
#define SO101_RING_DEPTH 5    /* joints_pos delay: 0.1 s = 5 control steps (so100_task.py:81,196-198) */

typedef enum {
  SO101_OK = 0,
  SO101_ERR_ARG = -1,        /* NULL/invalid argument */
  SO101_ERR_MODEL = -2,      /* blob rejected (magic/version/dims outside the compiled limits) */
  SO101_ERR_HIP = -3,        /* HIP runtime error (message holds hipGetErrorString) */
  SO101_ERR_STATE = -4       /* call order violated (e.g. step before bind_state) */
} so101_status;

typedef struct so101_sim so101_sim; /* opaque */

/* Caller-owned device buffers holding the per-env state (struct-of-arrays, N fastest). */
typedef struct {
  float* qpos;         /* [nq][N]  generalized positions          (physics.data.qpos) */
  float* qvel;         /* [nv][N]  generalized velocities         (physics.data.qvel) */
  float* ctrl;         /* [nu][N]  actuator controls, UNclamped   (physics.data.ctrl) */
  float* warmstart;    /* [nv][N]  previous qacc (solver warm start, data.qacc_warmstart) */
  float* obs_ring;     /* [SO101_RING_DEPTH][6][N] joints_pos delay line */
  float* ep_return;    /* [N] running sum of rewards of the current episode */
  int32_t* step_count; /* [N] control steps since reset */
  int32_t* episode;    /* [N] episode counter (keys the reset RNG) */
  const float* mass_scale; /* [2][N] per-env scale of the free props' mass and inertia (object, container): domain
                          randomisation beyond the reference, which randomises poses only (so100_hand_over.py:37-55);
                          NULL = 1.0 everywhere.  Read at every step and reset; after changing values call
                          so101_configure (cached initial states were settled with the old masses). */
} so101_buffers;

/* Tunables that the reference fixes through its MJCF / dm_control arguments. */
typedef struct {
  float action_offset[SO101_ACT_DIM]; /* calibration homing offsets added to every action
                                         (scripts/so101_calibration.py:62-88; zeros when the file is absent) */
  int32_t last_step;       /* control step on which physics.time() >= time_limit first holds
                              (task_suite.py:151; fp64 accumulation replayed on the host) */
  int32_t n_substeps;      /* control_timestep / physics timestep = 10 (task_suite.py:41) */
  int32_t solver_iterations; /* solver iteration cap; <=0 keeps the model's (100) */
  float solver_tolerance;  /* solver early-exit tolerance; <0 keeps the model's (1e-8) */
  int32_t settle_max_substeps; /* PropPlacer settle budget, 1000 = 2.0 s (so100_hand_over.py:222-229) */
  int32_t terminate_on_success; /* SO100Task terminate_episode (so100_task.py:120,297-302) */
  uint64_t env_id_base;    /* global index of env 0 of this handle (multi-GPU sharding) */
  int32_t solver;          /* SO101_SOLVER_NEWTON (default: the reference's scene sets no <option solver>, so MuJoCo's
                              default Newton applies) or SO101_SOLVER_PGS (the solver BASELINE.json's north_star names) */
  int32_t prefetch_resets; /* 1 (default): settle the next episode's initial state of every env ahead of time on an
                              internal low-priority stream, so that auto-resets inside so101_step cost a copy;
                              0: always settle inside the call.  Results are identical either way. */
  int32_t pipeline;        /* how so101_step is launched; same device functions, bit-identical results:
                              1 (default) launch chains: per env slice, every substep is a narrowphase launch (one
                                wavefront per candidate pair) followed by a solve launch (one wavefront per env);
                              0 one fused launch, one wavefront per env;
                              2 per-env chained: cost order, prologue and ONE persistent launch whose wavefronts pull
                                narrowphase chunks and per-env solve items from device-side queues, every env advancing
                                on its own dependencies (csrc/so101_chain.hpp);
                              3 merged launches: like 1, with the narrowphase of substep s + 1 pulled as chunks inside
                                the solve launch of substep s (half the launch boundaries).
                              2 and 3 are correct and measured slower on MI355X: wavefronts that run narrowphase and
                              solver code out of step refetch the 240 KB of code from L2 (profiles/README.md). */
  int32_t groups;          /* pipelined step: the envs, sorted by the solver time of their previous step, are cut into 1..8
                              slices whose launch chains run on separate internal streams and fill each other's tails;
                              0 (default) = 4 when GPU_MAX_HW_QUEUES >= 6 was set before HIP initialised, else 3 (the
                              runtime maps streams onto that many hardware queues; chains that share one serialise);
                              results do not depend on it */
  int32_t use_graph;       /* 1 (default): the launch sequence of the pipelined step is captured once into a HIP graph and
                              replayed with one hipGraphLaunch per step (re-captured when configuration, bound buffers,
                              pool or the step's I/O pointers change); 0: plain launches.  Same kernels either way. */
  int32_t chain_waves;     /* pipeline = 2: persistent wavefronts of the step kernel; 0 (default) = 2048 (2 per SIMD on
                              256 CUs), capped at 2 * N + 6 */
} so101_config;

int so101_version(void);

/* Compile-time limits of this build, for the host shim's sanity checks. */
int so101_max_contacts(void);

/* Replaces: composer.Environment construction + MuJoCo model compile (task_suite.py:148-155).
 * model_blob: f32 blob from so101_sim_amd/model/blob.py (host memory, copied).  Allocates the
 * device copy of the model and internal scratch only. */
int so101_create(const void* model_blob, size_t blob_bytes, int n_envs, int hip_device, uint64_t seed,
                 so101_sim** out);
void so101_destroy(so101_sim* sim);

int so101_default_config(so101_config* cfg);
int so101_configure(so101_sim* sim, const so101_config* cfg);
int so101_bind_state(so101_sim* sim, const so101_buffers* buffers);

/* Optional observables physics_state = concat(qpos, qvel) (shape (38,), so100_task.py:366-368; enabled by the reference iff
 * image_observation_enabled, :203) and delayed_physics_state, the same delayed by 0.3 s = 15 control steps and padded with the
 * episode's first value (:204-210, task_suite.py:154).  ring [15][38][N] is the delay line (caller-owned like obs_ring),
 * physics_state / delayed [N][38] are rewritten by every so101_step / so101_reset / so101_begin_episode.  All NULL = off
 * (default).  The pointers must stay valid like the bound state buffers. */
#define SO101_PHYSICS_STATE_DIM 38
#define SO101_PHYSICS_STATE_DELAY 15
int so101_bind_physics_state(so101_sim* sim, float* ring, float* physics_state, float* delayed_physics_state);

/* Replaces: env.reset() -> task.initialize_episode (so100_task.py:304-320, so100_hand_over.py:320-323):
 * arm qpos/qvel = 0, ctrl = home + offsets, object/container placement drawn from the counter RNG keyed
 * (seed, env_id, episode), container rejection-sampled against collisions, props settled with the arm
 * held, delay line filled with the reset value.  mask: [N] bytes, nonzero = reset that env; NULL = all. */
int so101_reset(so101_sim* sim, const uint8_t* mask, void* hip_stream);

/* Reset pool: with pool_size > 0 every reset (so101_reset and the auto-reset inside so101_step) starts the episode
 * from entry floor(u * pool_size) of the caller's pool, u = the counter RNG at (seed, global env id, episode, draw
 * 1000), instead of placement + settle: scripted pre-grasp states for contact-heavy workloads
 * (examples/automated_lerobot_dataset_generator.py:180-205 scripts such states on the reference), checkpoints.
 * qpos[20][K], qvel[18][K], ctrl[6][K] are device pointers that must outlive their use; pool_size = 0 restores the
 * reference's reset. */
int so101_set_reset_pool(so101_sim* sim, const float* qpos, const float* qvel, const float* ctrl, int pool_size);

/* Settled-state store: the 1000-substep PropPlacer settle of so100_hand_over.py:222-229 dominates short episodes, and its
 * result is a pure function of (model, seed, global env id, episode, mass scale, solver settings).
 * so101_compute_settled() runs placement + settle for episodes first_episode .. first_episode + n_episodes - 1 of
 * every env into caller-owned device tables qpos[n_episodes][20][N], qvel[..][18][N], warmstart[..][18][N],
 * flags[n_episodes][N] without touching the envs; the caller may keep them on disk across runs (the Python face
 * does: BatchedEnvironment.save_settled_cache / load_settled_cache).  so101_set_settled_store() hands such tables
 * back: resets of those episodes copy the entry (bit-identical to settling again) and the background prefetch skips
 * them.  n_episodes = 0 detaches the store.  The tables must outlive their use. */
int so101_compute_settled(so101_sim* sim, int first_episode, int n_episodes, float* qpos, float* qvel, float* warmstart,
                          int32_t* flags, void* hip_stream);
int so101_set_settled_store(so101_sim* sim, const float* qpos, const float* qvel, const float* warmstart,
                            const int32_t* flags, int first_episode, int n_episodes);

/* PropPlacer settle only (so100_hand_over.py:222-229, settle_physics=True): from the state in the bound buffers, arm
 * held, props integrated until |qvel| < 1e-3 and |qacc| < 1e-2 or settle_max_substeps are used.  For callers that draw
 * the placements themselves - the single-env Python facade draws them from numpy.random.RandomState in the reference's
 * order (object xyz, object yaw, container xyz per attempt), so that a seed reproduces the reference's episode; follow
 * with so101_begin_episode. */
int so101_settle(so101_sim* sim, void* hip_stream);

/* Starts an episode from whatever state the caller wrote into the bound buffers (checkpoint restore,
 * known-answer tests): ctrl = home + offsets, delay line filled with the current joints_pos,
 * step_count/ep_return cleared, no placement, no settle.  The reference's equivalent is
 * physics.set_state(...) followed by the observation updater's reset (dm_control). */
int so101_begin_episode(so101_sim* sim, void* hip_stream);

/* Replaces: env.step(action): before_step (so100_task.py:266-287), n_substeps x mj_step, observables
 * (so100_task.py:323-368 with the delays of :189-210), get_reward (so100_hand_over.py:238-275),
 * get_discount / termination (so100_task.py:292-302) and the time limit (task_suite.py:151).
 * Envs whose previous step was LAST are reset first and report FIRST (dm_control auto-reset).
 *   action    [N][6]  float32, row-major as the caller's (N,6) tensor
 *   obs       [N][SO101_OBS_DIM]
 *   reward    [N]  in {0,1}      discount [N] in {0,1}      step_type [N]: 0 FIRST, 1 MID, 2 LAST */
int so101_step(so101_sim* sim, const float* action, float* obs, float* reward, float* discount,
               uint8_t* step_type, void* hip_stream);

/* Physics only: n_substeps of mj_step on every env with the bound ctrl; no task logic.
 * freeze_arm != 0 restores the arm's qpos/qvel after every substep (PropPlacer settle). */
int so101_physics(so101_sim* sim, int n_substeps, int freeze_arm, void* hip_stream);

/* Reward of the current state (SO100HandOver.get_reward, overlap mode) -> reward[N]. */
int so101_reward(so101_sim* sim, float* reward, void* hip_stream);

/* Episode returns for logging (copied from the bound ep_return). */
int so101_get_returns(so101_sim* sim, float* out, void* hip_stream);

/* Diagnostics of the most recent substep, per env: [N][SO101_DIAG_DIM] int32
 *   0 ncon, 1 nefc, 2 solver iterations, 3 broadphase candidates, 4 overflow flags,
 *   5 collision time, 6 constraint+solver time, 7 whole-call time of this env's wavefront (10 ns ticks, summed over
 *   the substeps of the last call; words 5 and 7 are zero unless the library was built with -DSO101_DEBUG_CLOCKS). */
#define SO101_DIAG_DIM 8
int so101_get_diag(so101_sim* sim, int32_t* out, void* hip_stream);

/* Event counters since so101_create (or the last call with clear != 0): out is a DEVICE pointer to
 * SO101_NEVENTS uint64.  Counter b counts env-steps (b <= 3) or env-resets (b = 4, 5) on which flag b was raised:
 *   0 broadphase candidate list overflowed (candidates dropped)      1 contact list overflowed (contacts dropped)
 *   2 (retired in round 5, always 0: arm-link contacts beyond the LDS pool recompute their Jacobian instead of being dropped)
 *                                                                     3 physics diverged: NaN or |x| > 1e10 in the
 *     state (mj_checkPos/Vel/Acc); the episode ends with reward 0 / discount 0 like a dm_control PhysicsError
 *     (task_suite.py:153 raise_exception_on_physics_error=False)
 *   4 reset: the container placer's 20 attempts all collided (dm_control's PropPlacer raises RuntimeError there;
 *     here the last sample is kept and the event is counted)         5 reset: the settle did not converge within
 *     settle_max_substeps (dm_control warns, examples/so101_rl_breakdown.ipynb:50-55)
 *   6 chained steps (pipeline = 2) ended by the watchdog (scheduler protocol error).  The abort is sticky: that step's results and
 *     those of EVERY later chained step of the handle are invalid - each counts here - until so101_configure selects another step path
 *   7 more than so101_max_contacts() contacts in a substep: the env kept ONE contact per touching geom pair (the first of its patch) for
 *     that substep instead of cutting the list; counter 1 then only counts substeps whose touching PAIRS alone exceeded the limit
 * The same bits appear per env in diag word 4 for the most recent substep. */
#define SO101_NEVENTS 8
int so101_get_events(so101_sim* sim, uint64_t* out, int clear, void* hip_stream);

/* Stage dump of one forward pass (no integration) for parity tests against the oracle:
 * out is [N][SO101_DEBUG_DIM] float32; layout documented in csrc/so101_device.hpp (DBG_*). */
#define SO101_DEBUG_DIM 2048
int so101_debug_forward(so101_sim* sim, float* out, void* hip_stream);

/* Diagnostics of the pipelined step's last narrowphase launch (device buffers, any may be NULL):
 * ncand[N] (count | overflow << 16), cand[N][256] (geom1 | geom2 << 16), ticks[N][256] (10 ns per candidate; profiling builds
 * only, SO101_ERR_STATE otherwise), conres[48 N][24] (contact count, normal, 5 x (dist, position) per candidate, at the
 * candidates' work-list positions of the last substep - 48 records per env of a slice, mean use 12). */
/* (profiling builds, -DSO101_DEBUG_CLOCKS; zeros otherwise) Stage clocks of k_pipe_solve's second-to-last substep: stage[N][8] = smooth dynamics, contact gather, constraint
 * rows, solver, integrate, next broadphase (10 ns ticks), ncon, solver iterations. */
int so101_debug_stages(so101_sim* sim, uint32_t* stage, void* hip_stream);

int so101_debug_candidates(so101_sim* sim, int32_t* ncand, uint32_t* cand, uint32_t* ticks, float* conres, void* hip_stream);

/* Facts about the handle for logs and tests (host value, synchronises `hip_stream` for the device-side ones):
 *   SO101_INFO_GRAPH_ACTIVE   1 when the last so101_step replayed a captured HIP graph
 *   SO101_INFO_STEP_PATH      0 fused, 1 launch chains, 2 per-env chained: what the last so101_step ran
 *   SO101_INFO_CHAINS         launch chains of the last pipeline = 1 step
 *   SO101_INFO_HW_QUEUES      GPU_MAX_HW_QUEUES as the library read it (4 when unset)
 *   SO101_INFO_SCHED_ABORTS   times the chained step's watchdog ended a launch (a protocol error; also event 6)
 *   SO101_INFO_SCRATCH_BYTES  library-owned device memory of this handle */
#define SO101_INFO_GRAPH_ACTIVE 0
#define SO101_INFO_STEP_PATH 1
#define SO101_INFO_CHAINS 2
#define SO101_INFO_HW_QUEUES 3
#define SO101_INFO_SCHED_ABORTS 4
#define SO101_INFO_SCRATCH_BYTES 5
long long so101_get_info(so101_sim* sim, int what, void* hip_stream);

/* Where the wavefronts of the chained step's persistent kernel spent their time since the last clear: out = 16 uint64 in
 * HOST memory (the call synchronises `hip_stream`): [0] claiming work that was found, [1] looking for work that was not,
 * [2] narrowphase chunks, [3] solve items (10 ns ticks, summed over wavefronts); [4] chunks, [5] solve items, [6] idle
 * rounds, [7] wavefronts that ran, [8] their lifetimes (ticks). */
int so101_debug_chain_stats(so101_sim* sim, uint64_t* out, int clear, void* hip_stream);

const char* so101_last_error(const so101_sim* sim);

/* ---- general-tree engine (csrc/so101_tree.hpp): models outside the SO100 topology - the ALOHA hand-over scenes of
 * so101_sim/tasks/hand_over.py (two 8-dof arms with slide fingers coupled by joint equalities, position actuators, joint
 * damping; reference model so101_sim/assets/aloha/aloha_pbr.xml).  The blob is the f32 blob of
 * so101_sim_amd.model.scenes.compile_aloha_scene / compile_dining_scene; limits: 32 bodies, 16 actuators and, per build, 32 dofs /
 * 128 geoms / 64 contacts / 384 rows (ALOHA hand-over) or 64 dofs / 256 geoms / 128 contacts / 768 rows (Dining).
 * Physics entry points only so far (what dm_control's physics.step() does for these scenes); same conventions as above:
 * device pointers, env-fastest struct-of-arrays state [dim][n_envs], asynchronous on `hip_stream`. */
typedef struct so101_tree so101_tree;
int so101_tree_create(const void* model_blob, size_t blob_bytes, int n_envs, int hip_device, so101_tree** out);
void so101_tree_destroy(so101_tree* sim);
/* dims[16] = nq, nv, nu, nbody, ngeom, floats per env of so101_tree_debug_forward, contact capacity; then the layout of that row for the
 * handle's build - offsets of bias, qacc_smooth, qacc, body positions, mass matrix, contacts, normal forces, the row stride of the mass
 * matrix - and the build itself (32 or 64: so101_tree_create takes the 32-dof / 128-geom / 64-contact build whenever the model fits it,
 * otherwise the 64-dof / 256-geom / 128-contact one of the Dining scenes) */
int so101_tree_dims(const so101_tree* sim, int* dims);
/* What the last so101_tree_step of this handle enqueued, as the library counted it (not a host-side copy of the rule): out[4] = env slices,
 * kernel launches, memsets, path (0: no step yet, 1: the single kernel k_tree_step, 2: the launch chain).  bench.py reports it as evidence. */
int so101_tree_last_plan(const so101_tree* sim, int* out);
int so101_tree_bind_state(so101_tree* sim, float* qpos, float* qvel, float* ctrl, float* warmstart);
/* solver_iterations <= 0 / solver_tolerance < 0 keep the model's (100, 1e-8) */
int so101_tree_configure(so101_tree* sim, int solver_iterations, float solver_tolerance);
/* n_substeps of mj_step (forward dynamics with contacts and the Newton solver, Euler with implicit joint damping) on the
 * bound state, ctrl as bound */
int so101_tree_physics(so101_tree* sim, int n_substeps, void* hip_stream);
/* forward dynamics at the bound state without integrating; out[n_envs][dims[5]] floats: counts (contacts, rows, solver
 * iterations, candidates, flags, scalar rows) at 0, then bias, qacc_smooth, qacc, body positions, the mass matrix, contacts [.][10]
 * (position, normal, distance, geom1, geom2, condim) and normal forces at the offsets so101_tree_dims reports (32-dof build: 8, 40, 72,
 * 104, 200 [32][32], 1224, 1864) */
int so101_tree_debug_forward(so101_tree* sim, float* out, void* hip_stream);
/* out[n_envs][8] int32 of the last so101_tree_physics / so101_tree_step / so101_tree_reset: contacts, rows, solver iterations,
 * candidates, flags (1 candidate overflow, 2 contact overflow, 4 row overflow: contacts dropped, 8 physics diverged, 16 container
 * placement rejected 20 times, 32 settle budget used up); after so101_tree_step word 5 holds the state of the contact-sequence reward */
int so101_tree_get_diag(so101_tree* sim, int32_t* out, void* hip_stream);
/* ---- env layer of the hand-over scenes on this engine (HandOverBanana / HandOverPen of the reference's task_suite.py:60-61):
 * before_step with the gripper unit conversion (aloha2_task.py:316-349), n_substeps of physics, the observables of
 * aloha2_task.py:386-444 with their 0.1 s delay, the overlap reward and termination of hand_over.py:246-284 / aloha2_task.py:355-367,
 * reset = home pose + prop placement + settle (aloha2_task.py:369-383, hand_over.py:208-236).  Caller-owned per-env arrays
 * (device, env-fastest): ring_pos [joints_delay_steps][npos][n_envs], ring_vel [joints_delay_steps][nvel][n_envs] (npos = nu = 14,
 * nvel = 16 for ALOHA; the delay as configured, 5 by default; with a delay of 0 the pointers are still required but never touched),
 * ep_return, step_count, episode.  Observation row of so101_tree_step, so101_tree_obs_dim() = 3 npos + 2 nvel floats per env:
 *   joints_pos (delayed) | joints_vel (delayed) | undelayed_joints_pos | undelayed_joints_vel | commanded_joints_pos
 * step_type 0 FIRST (the call after a LAST resets the env and ignores the action), 1 MID, 2 LAST; a physics error ends the
 * episode with reward 0 and discount 0.  Reset happens inside the step call: a copy when the settled-state store or the reset prefetch
 * (prefetch_resets) holds the episode, placement + settle in place otherwise. */
typedef struct {
  int n_substeps;            /* physics steps per control step (10) */
  int last_step;             /* control step on which the time limit ends the episode */
  int settle_max_substeps;   /* settle budget of a reset (1000) */
  int terminate_on_success;
  int solver_iterations;     /* <= 0: the model's */
  float solver_tolerance;    /* < 0: the model's */
  uint64_t seed, env_id_base;
  int reward_mode;           /* 0 overlap boxes (HandOver's default; Dining 'bbox'), 1 contact sequence (reward_based_on_overlap = False,
                                hand_over.py:286-338), 2 object touches receptacle (Dining 'contact', dining_place_in_container.py:126-154) */
  int reward_requires_handover;   /* mode 1: start the sequence at "right gripper touches the object" instead of at its last state */
  int joints_delay_steps;    /* delay of joints_pos / joints_vel in CONTROL steps (joints_observation_delay_secs / control_timestep,
                                aloha2_task.py:153-155,236-243); < 0: the reference's default 5 (0.1 s); 0: undelayed; at most 64 */
  int physics_delay_steps;   /* delay of delayed_physics_state (image_observation_delay_secs, aloha2_task.py:157-159,244-251); < 0: 15 (0.3 s) */
  int prefetch_resets;       /* != 0: the settled initial state of every env's NEXT episode is computed on an internal low-priority stream beside
                                the stepping kernels (library-owned cache + a second per-env scratch); a reset that finds its entry copies it, one
                                that does not settles in place - the same bits either way.  so101_tree_configure_env / so101_tree_destroy join it */
  int pipeline;              /* != 0: so101_tree_step runs as a launch chain - per substep one launch with a wavefront per candidate pair of the whole
                                batch (the narrowphase) and one with a wavefront per env (everything else) - instead of one kernel; the same bits
                                either way (the contact rewards add one narrowphase launch on the post-step state).  At most 63 substeps; above, the single kernel */
} so101_tree_config;
int so101_tree_obs_dim(const so101_tree* sim);
int so101_tree_bind_env(so101_tree* sim, float* ring_pos, float* ring_vel, float* ep_return, int32_t* step_count, int32_t* episode);
int so101_tree_configure_env(so101_tree* sim, const so101_tree_config* cfg);
/* physics_state / delayed_physics_state (aloha2_task.py:244-251,441-444: qpos | qvel, and its copy of physics_delay_steps control
 * steps ago, padded with the reset state) written by so101_tree_step / so101_tree_reset / so101_tree_begin_episode: caller-owned
 * device buffers ring [physics_delay_steps][nq + nv][n_envs] (env-fastest), physics_state [n_envs][nq + nv], delayed [n_envs][nq + nv];
 * all NULL switches the outputs off.  Replaces the host-side line of rounds 3 (reference test: aloha2_task_test.py:136-173). */
int so101_tree_bind_physics_state(so101_tree* sim, float* ring, float* physics_state, float* delayed);
int so101_tree_reset(so101_tree* sim, const uint8_t* mask, void* hip_stream);
/* Settled-state store, as so101_compute_settled / so101_set_settled_store above: placement + settle of episodes first_episode ..
 * first_episode + count - 1 of every env into caller-owned device tables qpos [count][nq][n_envs], qvel / warmstart [count][nv][n_envs],
 * flags [count][n_envs] (one launch per episode at the width of the machine), and their use by the resets of those episodes - copies,
 * bit-identical to settling in place.  The tables depend on seed, env_id_base, n_envs and the solver settings; count = 0 detaches. */
int so101_tree_compute_settled(so101_tree* sim, int first_episode, int count, float* qpos, float* qvel, float* warmstart, int32_t* flags, void* hip_stream);
int so101_tree_set_settled_store(so101_tree* sim, int first_episode, int count, const float* qpos, const float* qvel, const float* warmstart, const int32_t* flags);
/* PropPlacer(settle_physics=True) alone, for callers that draw the placements themselves: physics steps on the bound state with the
 * one-dof joints held until the props rest (|qvel| < 1e-3, |qacc| < 1e-2) or settle_max_substeps is used up (diag flag 32) */
int so101_tree_settle(so101_tree* sim, void* hip_stream);
/* adopt the bound state (qpos, qvel, ctrl) as the post-reset state of a new episode: delay lines filled with it, counters cleared */
int so101_tree_begin_episode(so101_tree* sim, void* hip_stream);
int so101_tree_step(so101_tree* sim, const float* action /*[n_envs][nu]*/, float* obs, float* reward, float* discount, uint8_t* step_type,
                    void* hip_stream);
const char* so101_tree_last_error(const so101_tree* sim);

#ifdef __cplusplus
}
#endif
#endif /* SO101_H_ */
