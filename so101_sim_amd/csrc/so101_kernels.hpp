// Fused kernels, one 64-thread workgroup (one wavefront) per environment.  Every kernel that contains the constraint
// solver is a template on the solver (1 = Newton, 0 = PGS) and is instantiated in its own translation unit
// (tu_*.hip), so that the Newton kernels carry no PGS code and the library builds in parallel.
#pragma once
#include "so101_env.hpp"

// Fills the cache of settled initial states for every env whose next episode is not in it yet.  A few persistent
// waves pull env indices from a queue so that the stepping kernels keep most of the machine.
template <int SOLVER>
__global__ void __launch_bounds__(64, 2) k_prepare(const DevModel* m, StepParams P, DevBuffers B, PrepBuffers C, int ahead) {
  __shared__ EnvLDS L;
  int lane = wave_lane(), N = P.n_envs;
  for (;;) {
    // one launch per look-ahead (so101_hip.hip launch_prepare): the NEXT episode of every env first, then the one after it - an
    // env whose physics diverges needs its next entry at once, the second one only at the reset after that.  Within a launch
    // an env belongs to one wavefront (two entries of an env share nothing but the env's episode counter).
    int e = 0;
    if (lane == 0) e = atomicAdd(C.cursor + ahead, 1);
    e = wave_uniform_i(e);
    if (ahead == 0 || C.scan_count <= 0) { if (e >= N) break; }
    else { if (e >= C.scan_count) break; e = (int)(((unsigned int)C.scan_first + (unsigned int)e) % (unsigned int)N); }
    // an entry is only overwritten when its episode lies behind the env's counter, i.e. after env_reset() has consumed it
    // (the counter is bumped after the entry has been read)
    int next = __atomic_load_n(&B.episode[e], __ATOMIC_ACQUIRE);
    {
      int target = next + ahead;
      size_t slot = (unsigned int)target & 1u;
      if (__atomic_load_n(&C.tag[slot * N + e], __ATOMIC_ACQUIRE) == target) continue;
      if ((unsigned int)target - (unsigned int)C.store_first < (unsigned int)C.store_count) continue;     // in the settled-state store
      if (lane == 0) { L.overflow = 0; L.t_collision = 0; L.t_solve = 0; }
      load_env_constants(L, B, e, N);
      env_settle<SOLVER>(m, L, P, e, (unsigned int)target);
      wave_sync();
      if (lane < NQ) C.qpos[(slot * NQ + lane) * N + e] = L.qpos[lane];
      if (lane < NV) { C.qvel[(slot * NV + lane) * N + e] = L.qvel[lane]; C.warm[(slot * NV + lane) * N + e] = L.warm[lane]; }
      if (lane == 0) C.flags[slot * N + e] = L.overflow;
      __threadfence();
      wave_sync();
      if (lane == 0) __atomic_store_n(&C.tag[slot * N + e], target, __ATOMIC_RELEASE);
      wave_sync();
    }
  }
}

// Placement + settle of episodes first .. first + count - 1 of every env into a caller-owned table (the settled-state
// store of PrepBuffers); does not touch the envs' state.  One wave per (episode, env).
template <int SOLVER>
__global__ void __launch_bounds__(64, 2) k_settle_table(const DevModel* m, StepParams P, DevBuffers B, int first, float* qpos, float* qvel,
                                                     float* warm, int* flags) {
  __shared__ EnvLDS L;
  int lane = wave_lane(), N = P.n_envs;
  int e = (int)(blockIdx.x % (unsigned int)N);
  size_t k = blockIdx.x / (unsigned int)N;
  if (lane == 0) { L.overflow = 0; L.t_collision = 0; L.t_solve = 0; }
  load_env_constants(L, B, e, N);
  env_settle<SOLVER>(m, L, P, e, (unsigned int)first + (unsigned int)k);
  wave_sync();
  if (lane < NQ) qpos[(k * NQ + lane) * N + e] = L.qpos[lane];
  if (lane < NV) { qvel[(k * NV + lane) * N + e] = L.qvel[lane]; warm[(k * NV + lane) * N + e] = L.warm[lane]; }
  if (lane == 0) flags[k * N + e] = L.overflow;
}

// __launch_bounds__(64, 2): two waves per SIMD => at most 256 VGPRs
template <int SOLVER>
__global__ void __launch_bounds__(64, 2) k_reset(const DevModel* m, StepParams P, DevBuffers B, PrepBuffers C, EventBuffers E,
                                              const unsigned char* mask, unsigned char* need_reset, int* diag) {
  __shared__ EnvLDS L;
  int e = blockIdx.x;
  if (mask && !mask[e]) return;
  if (wave_lane() == 0) { L.overflow = 0; L.t_collision = 0; L.t_solve = 0; L.t_begin = (unsigned int)SO101_CLOCK(); }
  env_reset<SOLVER>(m, L, P, B, C, e);
  store_state(L, B, e, P.n_envs);
  store_diag(L, diag, e);
  count_events(L, E, e);
  if (wave_lane() == 0) need_reset[e] = 0;
}

// PropPlacer settle only (dm_control initializers.PropPlacer(settle_physics=True), so100_hand_over.py:222-229) from the
// bound state: the placements were drawn by the caller (the single-env facade draws them from numpy's RandomState in
// the reference's order, so that a seed reproduces the reference's episode).  Arm held, props integrated until
// |qvel| < 1e-3 and |qacc| < 1e-2 or the budget is used; flags as in env_settle().
template <int SOLVER>
__global__ void __launch_bounds__(64, 2) k_settle(const DevModel* m, StepParams P, DevBuffers B, EventBuffers E, int* diag) {
  __shared__ EnvLDS L;
  int e = blockIdx.x, lane = wave_lane();
  load_state(L, B, e, P.n_envs);
  if (lane < NARM) { L.arm0_q[lane] = L.qpos[lane]; L.arm0_v[lane] = L.qvel[lane]; }
  wave_sync();
  bool settled = P.settle_max == 0;
  for (int k = 0; k < P.settle_max && !settled; k++) {
    if (substep<SOLVER>(m, L, P.iterations, P.tolerance, true, 7)) break;
    float mv = 0.f, ma = 0.f;
    if (lane >= NARM && lane < NV) { mv = fabsf(L.qvel[lane]); ma = fabsf(L.qacc[lane]); }
    mv = wave_max_f(mv); ma = wave_max_f(ma);
    settled = mv < 1e-3f && ma < 1e-2f;
  }
  if (!settled && lane == 0) L.overflow |= 32;
  wave_sync();
  store_state(L, B, e, P.n_envs);
  store_diag(L, diag, e);
  count_events(L, E, e);
}

template <int SOLVER>
__global__ void __launch_bounds__(64, 2) k_step(const DevModel* m, StepParams P, DevBuffers B, PrepBuffers C, EventBuffers E,
                                             const float* action, float* obs, float* reward, float* discount,
                                             unsigned char* step_type, unsigned char* need_reset, int* diag) {
  __shared__ EnvLDS L;
  int e = blockIdx.x, lane = wave_lane(), N = P.n_envs;
  if (need_reset[e]) {
    if (lane == 0) { L.overflow = 0; L.t_collision = 0; L.t_solve = 0; L.t_begin = (unsigned int)SO101_CLOCK(); }
    env_reset<SOLVER>(m, L, P, B, C, e);
    store_state(L, B, e, N);
    store_diag(L, diag, e);
    count_events(L, E, e);
    write_first(L, e, obs, reward, discount, step_type, need_reset);
    return;
  }
  int sc = B.step_count[e] + 1;
  load_state(L, B, e, N);
  // before_step: ctrl = action + homing offsets, unclamped (so100_task.py:266-287)
  if (lane < NU) L.ctrl[lane] = action[(size_t)e * NU + lane] + P.action_offset[lane];
  wave_sync();
  bool diverged = false;
  for (int s = 0; s < P.n_substeps && !diverged; s++) diverged = substep<SOLVER>(m, L, P.iterations, P.tolerance, false, 7);
  finish_step(m, L, P, B, e, sc, diverged, obs, reward, discount, step_type, need_reset, diag, E);
}

template <int SOLVER>
__global__ void __launch_bounds__(64, 2) k_physics(const DevModel* m, StepParams P, DevBuffers B, int nsub, int freeze, int* diag) {
  __shared__ EnvLDS L;
  int e = blockIdx.x, lane = wave_lane();
  load_state(L, B, e, P.n_envs);
  if (lane < NARM) { L.arm0_q[lane] = L.qpos[lane]; L.arm0_v[lane] = L.qvel[lane]; }
  wave_sync();
  for (int s = 0; s < nsub; s++) substep<SOLVER>(m, L, P.iterations, P.tolerance, freeze != 0, P.phases);
  store_state(L, B, e, P.n_envs);
  store_diag(L, diag, e);
}

DEV void twist_to_qacc(EnvLDS& L) { forward_accelerations(L); }

template <int SOLVER>
__global__ void __launch_bounds__(64) k_debug_forward(const DevModel* m, StepParams P, DevBuffers B, float* out) {
  __shared__ EnvLDS L;
  int e = blockIdx.x, lane = wave_lane();
  float* o = out + (size_t)e * DBG_DIM;
  for (int i = lane; i < DBG_DIM; i += WAVE) o[i] = 0.f;
  load_state(L, B, e, P.n_envs);
  kinematics(m, L);
  crba_arm(m, L);
  smooth_dynamics(m, L);
  twist_to_qacc(L);
  if (lane < NV) o[DBG_SMOOTH + lane] = L.qacc[lane];
  if (lane < 36) { o[DBG_M + lane] = L.Marm[lane / 6][lane % 6]; o[DBG_MINV + lane] = L.Minv[lane / 6][lane % 6]; }
  if (lane < NARM) o[DBG_BIAS + lane] = L.bias[lane];
  if (lane < 24) o[DBG_XPOS + lane] = L.xpos[lane / 3][lane % 3];
  wave_sync();
  collision(m, L);
  make_constraints(m, L);
  if constexpr (SOLVER == 1) solve_newton(m, L, P.iterations, P.tolerance); else solve_pgs(m, L, P.iterations, P.tolerance);
  twist_to_qacc(L);
  if (lane < NV) o[DBG_QACC + lane] = L.qacc[lane];
  if (lane == 0) {
    o[DBG_COUNTS + 0] = (float)L.ncon; o[DBG_COUNTS + 1] = (float)L.nrow; o[DBG_COUNTS + 2] = (float)L.iters;
    o[DBG_COUNTS + 3] = (float)L.ncand; o[DBG_COUNTS + 4] = (float)L.overflow;
    for (int k = 0; k < L.ncon; k++) {
      const Contact& c = L.con[k];
      float* oc = o + DBG_CON + 10 * k;
      oc[0] = c.pos[0]; oc[1] = c.pos[1]; oc[2] = c.pos[2]; oc[3] = c.frame[0]; oc[4] = c.frame[1]; oc[5] = c.frame[2];
      oc[6] = c.dist; oc[7] = (float)c.g1; oc[8] = (float)c.g2; oc[9] = (float)c.dim;
      for (int j = 0; j < 6; j++) o[DBG_FORCE + 6 * k + j] = c.f[j];
    }
    for (int k = 0; k < L.nrow; k++) o[DBG_ROWF + k] = L.row[k].f;
  }
  float r = task_reward(m, L);
  if (lane == 0) o[DBG_REWARD] = r;
}
