// Env-level device functions shared by the fused and the pipelined kernels: state load/store, reset + settle,
// the end-of-step task logic (observables, reward, termination).  One environment per wavefront.
#pragma once
#include "so101_device.hpp"

// per-env constants that are not part of the integrated state
DEV void load_env_constants(EnvLDS& L, const DevBuffers& B, int e, int N) {
  int lane = wave_lane();
  if (lane < NFREE) L.fscale[lane] = B.mass_scale ? B.mass_scale[(size_t)lane * N + e] : 1.f;
}

DEV void load_state(EnvLDS& L, const DevBuffers& B, int e, int N) {
  int lane = wave_lane();
  load_env_constants(L, B, e, N);
  if (lane < NQ) L.qpos[lane] = B.qpos[(size_t)lane * N + e];
  if (lane < NV) { L.qvel[lane] = B.qvel[(size_t)lane * N + e]; L.warm[lane] = B.warm[(size_t)lane * N + e]; }
  if (lane < NU) L.ctrl[lane] = B.ctrl[(size_t)lane * N + e];
  if (lane == 0) { L.overflow = 0; L.t_collision = 0; L.t_solve = 0; L.t_begin = (unsigned int)SO101_CLOCK(); }
  wave_sync();
}

DEV void store_state(const EnvLDS& L, const DevBuffers& B, int e, int N) {
  int lane = wave_lane();
  if (lane < NQ) B.qpos[(size_t)lane * N + e] = L.qpos[lane];
  if (lane < NV) { B.qvel[(size_t)lane * N + e] = L.qvel[lane]; B.warm[(size_t)lane * N + e] = L.warm[lane]; }
  if (lane < NU) B.ctrl[(size_t)lane * N + e] = L.ctrl[lane];
}

// physics_state observable (qpos | qvel) and its 15-control-step delay line, when the caller bound them.  `fill`: an episode
// starts here - every slot of the line takes the current state (the observable's INITIAL_VALUE padding, task_suite.py:154) and
// both outputs report it; otherwise `sc` = control steps since reset including this one: read the value of step sc - 15,
// store this step's.
DEV void physics_state_obs(const EnvLDS& L, const DevBuffers& B, int e, int N, bool fill, int sc) {
  if (!B.ps_ring) return;
  int lane = wave_lane();
  if (lane < PS_DIM) {
    float v = lane < NQ ? L.qpos[lane] : L.qvel[lane - NQ], delayed = v;
    if (fill) {
      for (int r = 0; r < PS_DELAY; r++) B.ps_ring[((size_t)r * PS_DIM + lane) * N + e] = v;
    } else {
      size_t ri = ((size_t)((sc - 1) % PS_DELAY) * PS_DIM + lane) * N + e;
      delayed = B.ps_ring[ri];
      B.ps_ring[ri] = v;
    }
    B.ps_out[(size_t)e * PS_DIM + lane] = v;
    B.ps_delayed[(size_t)e * PS_DIM + lane] = delayed;
  }
}

// diag words (include/so101.h): 0 ncon, 1 nefc, 2 solver iterations, 3 broadphase candidates, 4 flags of the LAST
// substep; 5-7 stage clocks (debug builds); the sticky flag word lives in so101_sim::flags (see store_flags)
DEV void store_diag(const EnvLDS& L, int* diag, int e) {
  if (wave_lane() == 0 && diag) {
    int nefc = L.nrow;
    for (int k = 0; k < L.ncon; k++) nefc += L.con[k].dim;
    diag[8 * e + 0] = L.ncon; diag[8 * e + 1] = nefc; diag[8 * e + 2] = L.iters; diag[8 * e + 3] = L.ncand;
    diag[8 * e + 4] = L.overflow;
    diag[8 * e + 5] = (int)L.t_collision; diag[8 * e + 6] = (int)L.t_solve;
    diag[8 * e + 7] = (int)((unsigned int)SO101_CLOCK() - L.t_begin);
  }
}

// adds this env's flag word (ORed with what earlier substeps of the control step left in E.flags) to the counters
template <bool AG = false>       // AG: E.flags[e] was last written by another wavefront of this launch (wave.hpp)
DEV void count_events(const EnvLDS& L, const EventBuffers& E, int e) {
  if (wave_lane() == 0) {
    int f = (AG ? ld_agent(&E.flags[e]) : E.flags[e]) | L.overflow;
    if (f) {
      if constexpr (AG) st_agent(&E.flags[e], 0); else E.flags[e] = 0;
      for (int b = 0; b < SO101_NEVENTS; b++) if ((f >> b) & 1) atomicAdd(&E.events[b], 1ull);
    }
  }
}

// env.reset(), part 1: SO100Task.initialize_episode + SO100HandOver placers + settle (so100_task.py:304-320,
// so100_hand_over.py:208-229,320-323).  Leaves the settled state of `episode` in LDS; touches no HBM state.
// Flags (L.overflow): 16 = the container placer used all 20 attempts and the last sample still collides (dm_control
// raises RuntimeError there), 32 = the settle budget ran out before |qvel| < 1e-3 and |qacc| < 1e-2 (dm_control
// warns, examples/so101_rl_breakdown.ipynb:50-55), 8 = the settle diverged.
template <int SOLVER>
DEV void env_settle(const DevModel* m, EnvLDS& L, const StepParams& P, int e, unsigned int episode) {
  int lane = wave_lane();
  unsigned long long env_id = P.env_id_base + (unsigned long long)e;
  if (lane < NQ) L.qpos[lane] = 0.f;
  if (lane < NV) { L.qvel[lane] = 0.f; L.warm[lane] = 0.f; }
  if (lane < NU) L.ctrl[lane] = m->home_ctrl[lane] + P.action_offset[lane];
  if (lane < NARM) { L.arm0_q[lane] = 0.f; L.arm0_v[lane] = 0.f; }
  wave_sync();
  // draws: object xyz, object yaw, container xyz (+3 per rejection)
  if (lane == 0) {
    float* qo = &L.qpos[NARM]; float* qc = &L.qpos[NARM + 7];
    for (int k = 0; k < 3; k++) qo[k] = m->obj_lo[k] + rng_uniform(P.seed, env_id, episode, k) * (m->obj_hi[k] - m->obj_lo[k]);
    float yaw = m->obj_yaw[0] + rng_uniform(P.seed, env_id, episode, 3) * (m->obj_yaw[1] - m->obj_yaw[0]);
    float sn, cs; sincos_f(0.5f * yaw, &sn, &cs);
    qo[3] = cs; qo[4] = 0.f; qo[5] = 0.f; qo[6] = sn;
    qc[3] = 1.f; qc[4] = 0.f; qc[5] = 0.f; qc[6] = 0.f;
  }
  wave_sync();
  bool placed = false;
  for (int attempt = 0; attempt < 20 && !placed; attempt++) {
    if (lane < 3) L.qpos[NARM + 7 + lane] = m->con_lo[lane] + rng_uniform(P.seed, env_id, episode, 4 + 3 * attempt + lane) * (m->con_hi[lane] - m->con_lo[lane]);
    wave_sync();
    kinematics(m, L);
    collision(m, L);
    bool hit = false;
    for (int k = 0; k < L.ncon; k++) if (L.con[k].d1 == NARM + 1 || L.con[k].d2 == NARM + 1) hit = true;
    wave_sync();
    placed = !hit;
  }
  if (!placed && lane == 0) L.overflow |= 16;
  // settle: arm restored after every substep; stop when |qvel|<1e-3 and |qacc|<1e-2 over the prop dofs
  bool settled = P.settle_max == 0;
  for (int k = 0; k < P.settle_max && !settled; k++) {
    if (substep<SOLVER>(m, L, P.iterations, P.tolerance, true, 7)) break;     // diverged: flag 8 is already set
    float mv = 0.f, ma = 0.f;
    if (lane >= NARM && lane < NV) { mv = fabsf(L.qvel[lane]); ma = fabsf(L.qacc[lane]); }
    mv = wave_max_f(mv); ma = wave_max_f(ma);
    settled = mv < 1e-3f && ma < 1e-2f;
  }
  if (!settled && lane == 0) L.overflow |= 32;
  wave_sync();
}

// env.reset(): takes the settled state of the next episode from the cache or computes it, then starts the
// episode: delay line padded with the reset value (task_suite.py:154 INITIAL_VALUE), counters cleared.
template <int SOLVER>
DEV void env_reset(const DevModel* m, EnvLDS& L, const StepParams& P, const DevBuffers& B, const PrepBuffers& C, int e) {
  int lane = wave_lane(), N = P.n_envs;
  unsigned int episode = (unsigned int)B.episode[e];
  load_env_constants(L, B, e, N);
  const size_t slot = episode & 1u;
  bool cached = C.tag && __atomic_load_n(&C.tag[slot * N + e], __ATOMIC_ACQUIRE) == (int)episode;
  if (C.pool_size > 0) {
    // reset pool: the episode starts from a caller-provided state (pre-grasp pools, checkpoints) instead of
    // placement + settle; the entry is a pure function of (seed, global env id, episode)
    float u = rng_uniform(P.seed, P.env_id_base + (unsigned long long)e, episode, 1000u);
    int k = (int)(u * (float)C.pool_size);
    k = k < C.pool_size - 1 ? k : C.pool_size - 1;
    size_t K = (size_t)C.pool_size;
    if (lane < NQ) L.qpos[lane] = C.pool_qpos[(size_t)lane * K + k];
    if (lane < NV) { L.qvel[lane] = C.pool_qvel[(size_t)lane * K + k]; L.warm[lane] = 0.f; }
    if (lane < NU) L.ctrl[lane] = C.pool_ctrl[(size_t)lane * K + k];
    if (lane == 0) { L.ncon = 0; L.nrow = 0; L.iters = 0; L.ncand = 0; }
    wave_sync();
  } else if (episode - (unsigned int)C.store_first < (unsigned int)C.store_count) {
    size_t k = episode - (unsigned int)C.store_first, o = (size_t)e;
    if (lane < NQ) L.qpos[lane] = C.store_qpos[(k * NQ + lane) * N + o];
    if (lane < NV) { L.qvel[lane] = C.store_qvel[(k * NV + lane) * N + o]; L.warm[lane] = C.store_warm[(k * NV + lane) * N + o]; }
    if (lane < NU) L.ctrl[lane] = m->home_ctrl[lane] + P.action_offset[lane];
    if (lane == 0) { L.ncon = 0; L.nrow = 0; L.iters = 0; L.ncand = 0; L.overflow |= C.store_flags[k * N + o]; }
    wave_sync();
  } else if (cached) {
    if (lane < NQ) L.qpos[lane] = C.qpos[(slot * NQ + lane) * N + e];
    if (lane < NV) { L.qvel[lane] = C.qvel[(slot * NV + lane) * N + e]; L.warm[lane] = C.warm[(slot * NV + lane) * N + e]; }
    if (lane < NU) L.ctrl[lane] = m->home_ctrl[lane] + P.action_offset[lane];
    if (lane == 0) { L.ncon = 0; L.nrow = 0; L.iters = 0; L.ncand = 0; L.overflow |= C.flags[slot * N + e]; }
    wave_sync();
  } else {
    env_settle<SOLVER>(m, L, P, e, episode);
  }
  if (lane < NARM) {
    for (int r = 0; r < 5; r++) B.ring[((size_t)r * NARM + lane) * N + e] = L.qpos[lane];
  }
  physics_state_obs(L, B, e, N, true, 0);
  // the cache entry has been read completely before the episode counter tells k_prepare() to refill it
  __threadfence();
  wave_sync();
  if (lane == 0) { B.step_count[e] = 0; B.ep_return[e] = 0.f; __atomic_store_n(&B.episode[e], (int)(episode + 1u), __ATOMIC_RELEASE); }
}

// FIRST time step of an env that was just reset inside a step call (dm_control auto-reset: the call after LAST
// resets and reports FIRST; the action is ignored)
DEV void write_first(const EnvLDS& L, int e, float* obs, float* reward, float* discount, unsigned char* step_type,
                     unsigned char* need_reset) {
  int lane = wave_lane();
  if (lane < NARM) {
    obs[(size_t)e * 18 + lane] = L.qpos[lane];
    obs[(size_t)e * 18 + 6 + lane] = L.qpos[lane];
    obs[(size_t)e * 18 + 12 + lane] = L.ctrl[lane];
  }
  if (lane == 0) { reward[e] = 0.f; discount[e] = 1.f; step_type[e] = 0; need_reset[e] = 0; }
}

// End of a control step: observables with the 5-step joints_pos delay line (so100_task.py:189-210,323-368), reward
// (so100_hand_over.py:238-275), discount / termination (so100_task.py:292-302), time limit (task_suite.py:151).
// `sc` = control steps since reset including this one.  Needs the post-step state in LDS.
template <bool AG = false>
DEV void finish_step(const DevModel* m, EnvLDS& L, const StepParams& P, const DevBuffers& B, int e, int sc, bool diverged,
                     float* obs, float* reward, float* discount, unsigned char* step_type, unsigned char* need_reset,
                     int* diag, const EventBuffers& E) {
  int lane = wave_lane(), N = P.n_envs;
  kinematics(m, L);     // position-dependent quantities of the post-step state (legacy step2/step1 order)
  // joints_pos delay line: read the value of control step k-5, then store step k
  int slot = (sc - 1) % 5;
  if (lane < NARM) {
    size_t ri = ((size_t)slot * NARM + lane) * N + e;
    float delayed = B.ring[ri];
    B.ring[ri] = L.qpos[lane];
    obs[(size_t)e * 18 + lane] = delayed;
    obs[(size_t)e * 18 + 6 + lane] = L.qpos[lane];
    obs[(size_t)e * 18 + 12 + lane] = L.ctrl[lane];
  }
  physics_state_obs(L, B, e, N, false, sc);
  float r = diverged ? 0.f : task_reward(m, L);
  // physics error (dm_control): reward 0, discount 0, episode terminates
  bool success = (P.terminate_on_success && r >= 1.f) || diverged, timeout = sc >= P.last_step;
  store_state(L, B, e, N);
  store_diag(L, diag, e);
  if (lane == 0) {
    reward[e] = r; discount[e] = success ? 0.f : 1.f;
    unsigned char st = (success || timeout) ? 2 : 1;
    step_type[e] = st; need_reset[e] = st == 2;
    B.step_count[e] = sc; B.ep_return[e] += r;
  }
  count_events<AG>(L, E, e);
}
