// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under so101_sim_amd/ may include, link or call this.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
//
// CPU fp64, single-env restatement of the hot path of tuul-ai/so101_sim (lock-step env.step()).
// The arithmetic of that path lives in third-party packages that are NOT vendored in the reference
// tree: mujoco>=3.3.3, dm-control>=1.0.31 (reference requirements.txt:2,7; no lock file).  This file
// therefore restates MuJoCo's published computation pipeline (mj_step: kinematics, CRBA, RNE,
// actuation, collision, constraint assembly, Newton / PGS, semi-implicit Euler) as driven by the reference
// call sites:
//   * 10 substeps per control step         so101_sim/task_suite.py:41 (+ scene_pbr.xml:4 default dt)
//   * options multiccd / noslip=0          so101_sim/tasks/base/so100_task.py:151-152
//   * action -> ctrl                       so101_sim/tasks/base/so100_task.py:266-287
//   * observables and delays               so101_sim/tasks/base/so100_task.py:189-210,323-368
//   * reward (OOBB 6-axis SAT)             so101_sim/tasks/so100_hand_over.py:238-275,
//                                          so101_sim/utils/oobb_utils.py:114-273,
//                                          so101_sim/utils/success_detector_utils.py:19-28
//   * discount / termination               so101_sim/tasks/base/so100_task.py:292-302
//   * reset placement + settle             so101_sim/tasks/so100_hand_over.py:37-55,208-229,320-323
//
// PINNING STATUS: the reference holds no numeric test for this path.  The oracle is pinned by
//   (1) the known-answer vectors captured in the reference's executed notebooks (tests/golden/kat*.json;
//       so101_rl.ipynb:219-240, examples/so101_rl_breakdown.ipynb:274-288,352-363): the free-space arm step (KAT-1) to
//       <=1e-8 relative; the notebook's resting pose of banana and bowl is an equilibrium of this contact model (net
//       vertical acceleration < 0.5 % of g, height after one control step within 1e-5 m: tests/test_oracle_golden.py);
//   (2) an algorithm-independent definition of what mujoco's GJK / EPA returns for a penetrating pair - the minimum
//       translation, brute-forced from support functions in oracle/geomcheck.py: on the twelve contact-rich fixture states
//       89 % of the contacting pairs report a depth within 2 % of it, 95.5 % within 25 %, worst 1.59x (deep arm-on-arm and
//       arm-on-hull penetrations; tests/test_contact_geometry.py holds the numbers for oracle and kernel alike).
// Beyond that the contact phase is "parity unpinned": MuJoCo cannot run in this image, so no contact FORCE of the reference
// is available to compare with.
// Documented deviations from mujoco 3.3.x: non-flat convex pairs use the libccd-style MPR penetration query (MuJoCo's narrow
// phase before the native GJK / EPA became default) - flat-face pairs are closed form with up to 5 patch contacts
// (multiccd, so100_task.py:151).  Solver: Newton by default (mujoco's default; the reference scene sets no <option solver>),
// PGS (the one BASELINE.json's north_star names) selectable - both restated here and parity-tested.
#pragma once
#include <cstdint>
#include <cstddef>

extern "C" {
typedef struct orc_sim orc_sim;

// model_blob: f64 blob written by so101_sim_amd/model/blob.py
orc_sim* orc_create(const void* model_blob, size_t bytes);
void orc_destroy(orc_sim*);
int orc_nq(const orc_sim*);
int orc_nv(const orc_sim*);
int orc_nu(const orc_sim*);

// solver knobs: iterations<=0 keeps the model's (100), tolerance<0 keeps the model's (1e-8)
void orc_set_solver(orc_sim*, int iterations, double tolerance);
void orc_set_collision(orc_sim*, int enable);
/* per-env domain randomisation of the free props' mass (object, container): mass, inertia scale by s, body_invweight0 by 1/s */
void orc_set_mass_scale(orc_sim*, const double* scale /*[nfree]*/);
/* solver-parity tests: forward() uses this contact list instead of its own narrowphase; rows [n][9] = pos3 normal3
 * dist geom1 geom2, n = 0 restores the narrowphase */
void orc_inject_contacts(orc_sim*, int n, const double* rows);
// 1 = Newton (default: mujoco's default solver, mj_solNewton restated), 0 = PGS (BASELINE north_star)
void orc_set_solver_type(orc_sim*, int type);
// penetration query of non-flat convex pairs: 1 (default; what the kernels run) = MPR's final tetrahedron expanded by EPA to the nearest
// face of the Minkowski difference (the minimum translation, as mujoco >= 3.3's native GJK / EPA reports it); 0 = MPR's own portal depth
// (the -DSO101_MPR option of the kernels)
void orc_set_narrowphase(orc_sim*, int mode);
// hull against hull: 1 (default; what the kernels run) = up to NCPP contacts on flat features (hull_patch in so101_oracle.cpp), 0 = the single
// EPA contact of rounds 1-4
void orc_set_hull_multicontact(orc_sim*, int on);
/* the kernels' contact capacity (so101_max_contacts()) mirrored: a substep with more contacts keeps one per touching geom pair; 0 = no limit (default) */
void orc_set_contact_capacity(orc_sim*, int capacity);
int orc_contacts_reduced(const orc_sim*);      /* bit 0: the last collision pass reduced the list, bit 1: it was cut as well */
int orc_epa_iterations(const orc_sim*);
int orc_ls_evals(const orc_sim*);

// state access (one env)
void orc_set_state(orc_sim*, const double* qpos, const double* qvel, const double* warm);
void orc_get_state(const orc_sim*, double* qpos, double* qvel, double* warm);
void orc_set_ctrl(orc_sim*, const double* ctrl);

// physics: nsub x { forward ; Euler }.  freeze_arm != 0 holds the hinge dofs static (settle phase).  Returns 1 when the
// physics diverged (mj_check*: the state has been reset to qpos0)
int orc_substeps(orc_sim*, int nsub, int freeze_arm);
// forward only (fills derived quantities below)
void orc_forward(orc_sim*, int freeze_arm);

// derived quantities of the last forward
int orc_ncon(const orc_sim*);
int orc_nefc(const orc_sim*);
int orc_solver_iter(const orc_sim*);
void orc_get_M(const orc_sim*, double* M /*nv*nv*/);
void orc_get_bias(const orc_sim*, double* bias /*nv*/);
void orc_get_qacc(const orc_sim*, double* qacc /*nv*/, double* qacc_smooth /*nv*/);
void orc_get_actuator_force(const orc_sim*, double* f /*nu*/);
// contact k: out[0..2]=pos, [3..5]=normal, [6]=dist, [7]=geom1, [8]=geom2, [9]=dim
void orc_get_contact(const orc_sim*, int k, double* out10);
void orc_get_efc_force(const orc_sim*, double* f /*nefc*/);
void orc_get_body_pose(const orc_sim*, int body, double* pos3, double* quat4);
double orc_max_prop_qacc(const orc_sim*);

// task layer ------------------------------------------------------------------------------
// reward in {0,1}: overlap mode of SO100HandOver.get_reward on the current state
double orc_reward(orc_sim*);
// SAT on explicit boxes (oobb_utils.overlap_oobb_oobb); each box = pos3, quat4, half3
int orc_overlap_oobb(const double* box0, const double* box1);

// whole-env semantics (mirror of the batched HIP env; one env)
typedef struct {
  double offsets[6];        // calibration homing offsets added to the action (so101_calibration.py:62-88)
  int last_step;            // control step index on which time>=time_limit first holds
  int settle_max_substeps;  // 1000 (PropPlacer max_settle_physics_time 2.0 s)
  uint64_t seed;            // base seed
  uint64_t env_id;          // global env index (keys the RNG stream)
} orc_env_cfg;
void orc_env_config(orc_sim*, const orc_env_cfg*);
// reset: samples placement with the counter RNG keyed (seed, env_id, episode), settles.
void orc_env_reset(orc_sim*);
// step: action[6] -> obs[18] (joints_pos delayed 5 | undelayed_joints_pos | commanded_joints_pos),
// reward, discount, step_type (0 FIRST,1 MID,2 LAST).  Auto-resets when called after LAST.
void orc_env_step(orc_sim*, const double* action, double* obs18, double* reward, double* discount,
                  int* step_type);
// adopt the current state as the post-reset state of a new episode (for known-answer tests)
void orc_env_begin(orc_sim*);
void orc_env_obs(const orc_sim*, double* obs18);
int orc_env_step_count(const orc_sim*);
double orc_env_return(const orc_sim*);

// uniform(0,1) of the shared counter RNG (so tests can pin the GPU's reset draws)
double orc_rng_uniform(uint64_t seed, uint64_t env_id, uint64_t episode, uint32_t draw);

// bench.py cpu_baseline: the handover workload on `n_threads` host threads with `envs_per_thread` envs each.  Times the
// reset (placement + settle of every env) and, per solver setting k (iterations[k], tolerance[k]; Newton), `reps`
// repetitions of `steps` control steps of every env under uniform random actions.  Returns env-steps per repetition.
long long orc_bench_rollout(const void* blob, size_t bytes, int n_threads, int envs_per_thread, int steps, int reps, int n_settings,
                            const int* iterations, const double* tolerance, uint64_t seed, double* reset_seconds,
                            double* rep_seconds /*[n_settings][reps]*/);
}
