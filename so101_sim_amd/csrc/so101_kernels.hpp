// __global__ entry points: one 64-thread workgroup (one wavefront) per environment.
#pragma once
#include "so101_device.hpp"

struct DevBuffers {
  float *qpos, *qvel, *ctrl, *warm, *ring, *ep_return;
  int *step_count, *episode;
};

// Library-owned cache of settled initial states.  The settled state of an episode is a pure function of
// (seed, global env id, episode index, config), so it can be computed ahead of time: k_prepare() fills the
// cache on a side stream while the envs are stepping and env_reset() consumes an entry when its tag matches the
// episode that is about to start; otherwise env_reset() settles in place.  Either way the result is the same
// bits, only the time at which the work is done differs.
struct PrepBuffers {
  float *qpos, *qvel, *warm;   // [NQ|NV|NV][n_envs]
  int *tag;                    // episode index the entry belongs to, -1 = empty
  int *cursor;                 // work-queue head of k_prepare
};

// debug dump layout (floats) of so101_debug_forward
#define DBG_M 0          // 36  arm mass matrix
#define DBG_MINV 36      // 36
#define DBG_BIAS 72      // 6
#define DBG_SMOOTH 78    // 18  qacc_smooth (generalized)
#define DBG_QACC 96      // 18  qacc after the solve
#define DBG_COUNTS 114   // ncon, nrow, iters, ncand, overflow
#define DBG_XPOS 120     // 8*3 dynamic body positions
#define DBG_CON 144      // MAXCON * 10: pos3 normal3 dist g1 g2 dim
#define DBG_FORCE 464    // MAXCON * 6
#define DBG_ROWF 656     // MAXROW1
#define DBG_REWARD 672

DEV void load_state(EnvLDS& L, const DevBuffers& B, int e, int N) {
  int lane = wave_lane();
  if (lane < NQ) L.qpos[lane] = B.qpos[(size_t)lane * N + e];
  if (lane < NV) { L.qvel[lane] = B.qvel[(size_t)lane * N + e]; L.warm[lane] = B.warm[(size_t)lane * N + e]; }
  if (lane < NU) L.ctrl[lane] = B.ctrl[(size_t)lane * N + e];
  if (lane == 0) { L.overflow = 0; L.t_collision = 0; L.t_solve = 0; L.t_begin = (unsigned int)wall_clock64(); }
  wave_sync();
}

DEV void store_state(const EnvLDS& L, const DevBuffers& B, int e, int N) {
  int lane = wave_lane();
  if (lane < NQ) B.qpos[(size_t)lane * N + e] = L.qpos[lane];
  if (lane < NV) { B.qvel[(size_t)lane * N + e] = L.qvel[lane]; B.warm[(size_t)lane * N + e] = L.warm[lane]; }
  if (lane < NU) B.ctrl[(size_t)lane * N + e] = L.ctrl[lane];
}

DEV void store_diag(const EnvLDS& L, int* diag, int e) {
  if (wave_lane() == 0 && diag) {
    int nefc = L.nrow;
    for (int k = 0; k < L.ncon; k++) nefc += L.con[k].dim;
    diag[8 * e + 0] = L.ncon; diag[8 * e + 1] = nefc; diag[8 * e + 2] = L.iters; diag[8 * e + 3] = L.ncand;
    diag[8 * e + 4] = L.overflow;
    diag[8 * e + 5] = (int)L.t_collision; diag[8 * e + 6] = (int)L.t_solve;
    diag[8 * e + 7] = (int)((unsigned int)wall_clock64() - L.t_begin);
  }
}

DEV void twist_to_qacc(EnvLDS& L) {
  int lane = wave_lane();
  if (lane < NARM) L.qacc[lane] = L.qacc_arm[lane];
  if (lane >= 32 && lane < 32 + NFREE) {
    int f = lane - 32, b = NARM + f;
    float r[3] = {L.xipos[b][0] - L.xpos[b][0], L.xipos[b][1] - L.xpos[b][1], L.xipos[b][2] - L.xpos[b][2]};
    float al[3] = {L.facc[f][3], L.facc[f][4], L.facc[f][5]}, ww[3] = {L.fvel[f][3], L.fvel[f][4], L.fvel[f][5]};
    float t1[3], t2[3], ab[3];
    cross3(t1, al, r); cross3(t2, ww, r); cross3(t2, ww, t2);
    matTvec3(ab, L.xmat[b], al);
#pragma unroll
    for (int i = 0; i < 3; i++) { L.qacc[NARM + 6 * f + i] = L.facc[f][i] - t1[i]; L.qacc[NARM + 6 * f + 3 + i] = ab[i]; }
  }
  wave_sync();
}

// env.reset(), part 1: SO100Task.initialize_episode + SO100HandOver placers + settle (so100_task.py:304-320,
// so100_hand_over.py:208-229,320-323).  Leaves the settled state of `episode` in LDS; touches no HBM state.
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
    qo[3] = cosf(0.5f * yaw); qo[4] = 0.f; qo[5] = 0.f; qo[6] = sinf(0.5f * yaw);
    qc[3] = 1.f; qc[4] = 0.f; qc[5] = 0.f; qc[6] = 0.f;
  }
  wave_sync();
  for (int attempt = 0; attempt < 20; attempt++) {
    if (lane < 3) L.qpos[NARM + 7 + lane] = m->con_lo[lane] + rng_uniform(P.seed, env_id, episode, 4 + 3 * attempt + lane) * (m->con_hi[lane] - m->con_lo[lane]);
    wave_sync();
    kinematics(m, L);
    collision(m, L);
    bool hit = false;
    for (int k = 0; k < L.ncon; k++) if (L.con[k].d1 == NARM + 1 || L.con[k].d2 == NARM + 1) hit = true;
    wave_sync();
    if (!hit) break;
  }
  // settle: arm restored after every substep; stop when |qvel|<1e-3 and |qacc|<1e-2 over the prop dofs
  for (int k = 0; k < P.settle_max; k++) {
    substep(m, L, P.iterations, P.tolerance, true, 7, P.solver);
    float mv = 0.f, ma = 0.f;
    if (lane >= NARM && lane < NV) { mv = fabsf(L.qvel[lane]); ma = fabsf(L.qacc[lane]); }
    mv = wave_max_f(mv); ma = wave_max_f(ma);
    if (mv < 1e-3f && ma < 1e-2f) break;
  }
}

// env.reset(): takes the settled state of the next episode from the cache or computes it, then starts the
// episode: delay line padded with the reset value (task_suite.py:154 INITIAL_VALUE), counters cleared.
DEV void env_reset(const DevModel* m, EnvLDS& L, const StepParams& P, const DevBuffers& B, const PrepBuffers& C, int e) {
  int lane = wave_lane(), N = P.n_envs;
  unsigned int episode = (unsigned int)B.episode[e];
  bool cached = C.tag && __atomic_load_n(&C.tag[e], __ATOMIC_ACQUIRE) == (int)episode;
  if (cached) {
    if (lane < NQ) L.qpos[lane] = C.qpos[(size_t)lane * N + e];
    if (lane < NV) { L.qvel[lane] = C.qvel[(size_t)lane * N + e]; L.warm[lane] = C.warm[(size_t)lane * N + e]; }
    if (lane < NU) L.ctrl[lane] = m->home_ctrl[lane] + P.action_offset[lane];
    if (lane == 0) { L.ncon = 0; L.nrow = 0; L.iters = 0; L.ncand = 0; }
    wave_sync();
  } else {
    env_settle(m, L, P, e, episode);
  }
  if (lane < NARM) {
    for (int r = 0; r < 5; r++) B.ring[((size_t)r * NARM + lane) * N + e] = L.qpos[lane];
  }
  // the cache entry has been read completely before the episode counter tells k_prepare() to refill it
  __threadfence();
  wave_sync();
  if (lane == 0) { B.step_count[e] = 0; B.ep_return[e] = 0.f; __atomic_store_n(&B.episode[e], (int)(episode + 1u), __ATOMIC_RELEASE); }
}

// Fills the cache for every env whose next episode is not in it yet.  A few persistent waves pull env indices
// from a queue so that the stepping kernels keep most of the machine.
__global__ void __launch_bounds__(64, 2) k_prepare(const DevModel* m, StepParams P, DevBuffers B, PrepBuffers C) {
  __shared__ EnvLDS L;
  int lane = wave_lane(), N = P.n_envs;
  for (;;) {
    int e = 0;
    if (lane == 0) e = atomicAdd(C.cursor, 1);
    e = wave_uniform_i(e);
    if (e >= N) break;
    int target = __atomic_load_n(&B.episode[e], __ATOMIC_ACQUIRE);
    if (__atomic_load_n(&C.tag[e], __ATOMIC_ACQUIRE) == target) continue;
    if (lane == 0) { L.overflow = 0; L.t_collision = 0; L.t_solve = 0; }
    env_settle(m, L, P, e, (unsigned int)target);
    wave_sync();
    if (lane < NQ) C.qpos[(size_t)lane * N + e] = L.qpos[lane];
    if (lane < NV) { C.qvel[(size_t)lane * N + e] = L.qvel[lane]; C.warm[(size_t)lane * N + e] = L.warm[lane]; }
    __threadfence();
    wave_sync();
    if (lane == 0) __atomic_store_n(&C.tag[e], target, __ATOMIC_RELEASE);
    wave_sync();
  }
}

// __launch_bounds__(64, 2): two waves per SIMD => at most 256 VGPRs; measured 234 -> 204 ms per control step on
// the 4096-env random-action workload against the unconstrained allocation (256 VGPR + 75 AGPR, one wave per SIMD).
__global__ void __launch_bounds__(64, 2) k_reset(const DevModel* m, StepParams P, DevBuffers B, PrepBuffers C, const unsigned char* mask,
                                              unsigned char* need_reset, int* diag) {
  __shared__ EnvLDS L;
  int e = blockIdx.x;
  if (mask && !mask[e]) return;
  if (wave_lane() == 0) { L.overflow = 0; L.t_collision = 0; L.t_solve = 0; L.t_begin = (unsigned int)wall_clock64(); }
  env_reset(m, L, P, B, C, e);
  store_state(L, B, e, P.n_envs);
  store_diag(L, diag, e);
  if (wave_lane() == 0) need_reset[e] = 0;
}

__global__ void __launch_bounds__(64) k_begin(const DevModel* m, StepParams P, DevBuffers B, unsigned char* need_reset) {
  int e = blockIdx.x, lane = wave_lane(), N = P.n_envs;
  if (lane < NARM) {
    float q = B.qpos[(size_t)lane * N + e];
    for (int r = 0; r < 5; r++) B.ring[((size_t)r * NARM + lane) * N + e] = q;
    B.ctrl[(size_t)lane * N + e] = m->home_ctrl[lane] + P.action_offset[lane];
  }
  if (lane == 0) { B.step_count[e] = 0; B.ep_return[e] = 0.f; need_reset[e] = 0; }
}

__global__ void __launch_bounds__(64, 2) k_step(const DevModel* m, StepParams P, DevBuffers B, PrepBuffers C, const float* action, float* obs,
                                             float* reward, float* discount, unsigned char* step_type,
                                             unsigned char* need_reset, int* diag) {
  __shared__ EnvLDS L;
  int e = blockIdx.x, lane = wave_lane(), N = P.n_envs;
  if (need_reset[e]) {
    // dm_control auto-reset: the call after LAST resets and reports FIRST; the action is ignored
    if (lane == 0) { L.overflow = 0; L.t_collision = 0; L.t_solve = 0; L.t_begin = (unsigned int)wall_clock64(); }
    env_reset(m, L, P, B, C, e);
    store_state(L, B, e, N);
    store_diag(L, diag, e);
    if (lane < NARM) {
      obs[(size_t)e * 18 + lane] = L.qpos[lane];
      obs[(size_t)e * 18 + 6 + lane] = L.qpos[lane];
      obs[(size_t)e * 18 + 12 + lane] = L.ctrl[lane];
    }
    if (lane == 0) { reward[e] = 0.f; discount[e] = 1.f; step_type[e] = 0; need_reset[e] = 0; }
    return;
  }
  int sc = B.step_count[e] + 1;
  load_state(L, B, e, N);
  // before_step: ctrl = action + homing offsets, unclamped (so100_task.py:266-287)
  if (lane < NU) L.ctrl[lane] = action[(size_t)e * NU + lane] + P.action_offset[lane];
  wave_sync();
  bool diverged = false;
  for (int s = 0; s < P.n_substeps && !diverged; s++) diverged = substep(m, L, P.iterations, P.tolerance, false, 7, P.solver);
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
}

#include "so101_pipeline.hpp"

__global__ void __launch_bounds__(64, 2) k_physics(const DevModel* m, StepParams P, DevBuffers B, int nsub, int freeze, int* diag) {
  __shared__ EnvLDS L;
  int e = blockIdx.x, lane = wave_lane();
  load_state(L, B, e, P.n_envs);
  if (lane < NARM) { L.arm0_q[lane] = L.qpos[lane]; L.arm0_v[lane] = L.qvel[lane]; }
  wave_sync();
  for (int s = 0; s < nsub; s++) substep(m, L, P.iterations, P.tolerance, freeze != 0, P.phases, P.solver);
  store_state(L, B, e, P.n_envs);
  store_diag(L, diag, e);
}

__global__ void __launch_bounds__(64) k_reward(const DevModel* m, StepParams P, DevBuffers B, float* reward) {
  __shared__ EnvLDS L;
  int e = blockIdx.x;
  load_state(L, B, e, P.n_envs);
  kinematics(m, L);
  float r = task_reward(m, L);
  if (wave_lane() == 0) reward[e] = r;
}

__global__ void __launch_bounds__(64) k_debug_forward(const DevModel* m, StepParams P, DevBuffers B, float* out) {
  __shared__ EnvLDS L;
  int e = blockIdx.x, lane = wave_lane();
  float* o = out + (size_t)e * 1024;
  for (int i = lane; i < 1024; i += WAVE) o[i] = 0.f;
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
  make_constraints(m, L, P.solver == 0);
  if (P.solver == 1) solve_newton(m, L, P.iterations, P.tolerance); else solve_pgs(m, L, P.iterations, P.tolerance);
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
