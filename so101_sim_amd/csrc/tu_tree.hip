// General-tree engine (so101_tree.hpp): kernels and the so101_tree_* entry points of include/so101.h.
// Kernels and their launches live in one translation unit (no relocatable device code).
#include <hip/hip_runtime.h>
#include "so101_blob.hpp"
#include "../../include/so101.h"
#include "so101_tree.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

// so101_tree_debug_forward layout (floats per env; dims[7..14] of so101_tree_dims report it): counts, bias, qacc_smooth, qacc [TV each],
// body positions [TB][3], mass matrix [TV][TV], contacts [TCON][10] (pos3 normal3 dist geom1 geom2 dim), normal forces [TCON]
#undef TDBG_COUNTS
#undef TDBG_BIAS
#undef TDBG_QSM
#undef TDBG_QACC
#undef TDBG_XPOS
#undef TDBG_M
#undef TDBG_CON
#undef TDBG_FORCE
#undef TDBG_DIM
#define TDBG_COUNTS 0        // ncon, nrow, iters, ncand, flags, nscalar
#define TDBG_BIAS 8
#define TDBG_QSM (TDBG_BIAS + TV)
#define TDBG_QACC (TDBG_QSM + TV)
#define TDBG_XPOS (TDBG_QACC + TV)     // [TB][3]
#define TDBG_M (TDBG_XPOS + 3 * TB)    // [TV][TV]
#define TDBG_CON (TDBG_M + TV * TV)    // [TCON][10]
#define TDBG_FORCE (TDBG_CON + 10 * TCON)
#define TDBG_DIM (((TDBG_FORCE + TCON) + 63) / 64 * 64)

// entry points of this build: so101_tree32_* / so101_tree64_*; the public so101_tree_* of include/so101.h dispatch on the handle
// (csrc/tu_tree_api.hip)
#undef TAPI
#undef TAPI_CAT
#undef TAPI_CAT2
#define TAPI_CAT2(v, name) so101_tree##v##_##name
#define TAPI_CAT(v, name) TAPI_CAT2(v, name)
#define TAPI(name) TAPI_CAT(TREE_VARIANT, name)

#ifndef TREE_SOLVE_OCC
#if TREE_VARIANT == 64
#define TREE_SOLVE_OCC 1     // (LDS allows two envs per CU: registers are free; the Hessian's four accumulator tiles want them)
#else
#define TREE_SOLVE_OCC 2
#endif
#endif

namespace TREE_NS {
using tree::TreePipe;

__global__ void __launch_bounds__(64) k_tree_physics(const TreeModel* tm, const DevModel* gm, TreeBuffers B, int N, int nsub, int iterations, float tolerance) {
  BLOCK_SHARED(TreeLDS, L);
  int e = blockIdx.x, lane = wave_lane();
  if (lane == 0) L.flags = 0;
  tree::load_state(tm, L, B, e, N);
  TreeScratch G = tree::scratch_of(B, e);
  int bad = 0;
  for (int s = 0; s < nsub; s++) {
    tree::forward(tm, gm, L, G, iterations, tolerance);
    tree::euler(tm, L);
    // mj_checkPos / Vel / Acc: NaN or beyond 1e10 ends the episode (the caller sees flag 8 and resets the env)
    bool ok = true;
    if (lane < tm->nq) ok = ok && fabsf(L.qpos[lane]) <= 1e10f;
    if (lane < tm->nv) ok = ok && fabsf(L.qvel[lane]) <= 1e10f && fabsf(L.qacc[lane]) <= 1e10f;
    if (wave_ballot(!ok) != 0ull) { bad = 1; break; }
  }
  if (lane == 0 && bad) L.flags |= 8;
  tree::store_state(tm, L, B, e, N);
  if (lane == 0 && B.diag) { int* d = B.diag + 8 * e; d[0] = L.ncon; d[1] = L.nrow; d[2] = L.iters; d[3] = L.ncand; d[4] = L.flags; }
}

__global__ void __launch_bounds__(64) k_tree_forward(const TreeModel* tm, const DevModel* gm, TreeBuffers B, int N, int iterations, float tolerance, float* out, int phases) {
  BLOCK_SHARED(TreeLDS, L);
  int e = blockIdx.x, lane = wave_lane();
  if (lane == 0) L.flags = 0;
#ifdef TREE_PROF
  if (lane < 32) L.prof[lane] = 0u;
#endif
  tree::load_state(tm, L, B, e, N);
  TreeScratch G = tree::scratch_of(B, e);
  tree::forward(tm, gm, L, G, iterations, tolerance, phases);
  float* o = out + (size_t)e * TDBG_DIM;
  int nv = tm->nv, nb = tm->nbody;
  if (lane == 0) { o[0] = (float)L.ncon; o[1] = (float)L.nrow; o[2] = (float)L.iters; o[3] = (float)L.ncand; o[4] = (float)L.flags; o[5] = (float)L.nscalar; }
  if (lane < nv) {
    o[TDBG_BIAS + lane] = L.bias[lane]; o[TDBG_QSM + lane] = L.qsm[lane]; o[TDBG_QACC + lane] = L.qacc[lane];
    for (int r = 0; r < nv; r++) o[TDBG_M + r * TV + lane] = L.M[r][lane];
  }
  if (lane < nb) for (int k = 0; k < 3; k++) o[TDBG_XPOS + 3 * lane + k] = L.xpos[lane][k];
#ifdef TREE_PROF
  wave_sync();
  if (lane < 16) o[TDBG_M + lane] = (float)L.prof[lane];         // (profiling variant: the phase clocks take the first slots of the mass-matrix dump)
#endif
  for (int ci = lane; ci < L.ncon; ci += WAVE) {
    const TCon& c = L.con[ci];
    float* q = o + TDBG_CON + 10 * ci;
    for (int k = 0; k < 3; k++) { q[k] = c.pos[k]; q[3 + k] = c.frame[k]; }
    q[6] = c.dist; q[7] = (float)c.g1; q[8] = (float)c.g2; q[9] = (float)c.dim;
    o[TDBG_FORCE + ci] = L.nrow > 0 ? G.ef[c.row] : 0.f;
  }
}

// env.reset() of the envs whose mask byte is set (NULL: all), then the FIRST observation is what the next step call reports
__global__ void __launch_bounds__(64) k_tree_reset(const TreeModel* tm, const DevModel* gm, TreeTask T, TreeBuffers B, TreeEnvBuffers E, TreeStore S, const unsigned char* mask) {
  BLOCK_SHARED(TreeLDS, L);
  int e = blockIdx.x, lane = wave_lane(), N = T.n_envs;
  if (mask && !mask[e]) return;
  if (lane == 0) L.flags = 0;
  wave_sync();
  TreeScratch G = tree::scratch_of(B, e);
  tree::env_reset(tm, gm, T, L, G, B, E, S, e);
  tree::store_state(tm, L, B, e, N);
  if (lane < tm->nu) B.ctrl[(size_t)lane * N + e] = L.ctrl[lane];
  if (lane == 0) { E.need_reset[e] = 0; if (B.diag) { int* d = B.diag + 8 * e; d[0] = L.ncon; d[1] = L.nrow; d[2] = L.iters; d[3] = L.ncand; d[4] = L.flags; } }
}

// placement + settle of episode `episode` of every env into row k of caller-owned tables (so101_tree_compute_settled): what env_reset()
// would compute for that episode, without touching any env state
__global__ void __launch_bounds__(64) k_tree_settle_table(const TreeModel* tm, const DevModel* gm, TreeTask T, TreeBuffers B, unsigned int episode, int k,
                                                          float* qpos, float* qvel, float* warm, int* flags) {
  BLOCK_SHARED(TreeLDS, L);
  int e = blockIdx.x, lane = wave_lane(), N = T.n_envs;
  if (lane == 0) L.flags = 0;
  wave_sync();
  TreeScratch G = tree::scratch_of(B, e);
  tree::env_settle(tm, gm, T, L, G, e, episode);
  size_t kk = (size_t)k;
  if (lane < tm->nq) qpos[(kk * tm->nq + lane) * N + e] = L.qpos[lane];
  if (lane < tm->nv) { qvel[(kk * tm->nv + lane) * N + e] = L.qvel[lane]; warm[(kk * tm->nv + lane) * N + e] = L.warm[lane]; }
  if (lane == 0) flags[kk * N + e] = L.flags;
}

// Reset prefetch: the settled state of the NEXT episode of every env whose cache entry does not hold it yet (TreeStore, so101_tree.hpp), into
// the library's cache, on a low-priority stream beside the stepping kernels; its own per-env scratch (B.scratch here is the second one)
__global__ void __launch_bounds__(64) k_tree_prepare(const TreeModel* tm, const DevModel* gm, TreeTask T, TreeBuffers B, const int* episode_of, TreeStore S) {
  BLOCK_SHARED(TreeLDS, L);
  int e = blockIdx.x, lane = wave_lane(), N = T.n_envs;
  unsigned int episode = (unsigned int)ld_agent(&episode_of[e]);
  if (ld_agent(&S.ctag[e]) == episode + 1u) return;                                      // the entry is current
  if (S.qpos && episode - (unsigned int)S.first < (unsigned int)S.count) return;          // the caller's settled-state store covers it
  if (lane == 0) L.flags = 0;
  wave_sync();
  TreeScratch G = tree::scratch_of(B, e);
  tree::env_settle(tm, gm, T, L, G, e, episode);
  if (lane == 0) st_agent(&S.ctag[e], 0u);                                                 // (no reader may take a half-written entry for the old episode)
  drain_stores();
  if (lane < tm->nq) st_agent(&S.cq[(size_t)lane * N + e], L.qpos[lane]);
  if (lane < tm->nv) { st_agent(&S.cv[(size_t)lane * N + e], L.qvel[lane]); st_agent(&S.cw[(size_t)lane * N + e], L.warm[lane]); }
  if (lane == 0) st_agent(&S.cf[e], L.flags);
  drain_stores();
  wave_sync();
  if (lane == 0) st_agent(&S.ctag[e], episode + 1u);
}

// settle the bound state with the one-dof joints held (dm_control's PropPlacer(settle_physics=True) alone, for callers that draw the
// placements themselves): until |qvel| < 1e-3 and |qacc| < 1e-2 over the props' dofs or the budget ends (flag 32)
__global__ void __launch_bounds__(64) k_tree_settle(const TreeModel* tm, const DevModel* gm, TreeTask T, TreeBuffers B) {
  BLOCK_SHARED(TreeLDS, L);
  int e = blockIdx.x, lane = wave_lane(), N = T.n_envs;
  if (lane == 0) L.flags = 0;
  tree::load_state(tm, L, B, e, N);
  TreeScratch G = tree::scratch_of(B, e);
  float q0 = lane < tm->njnt ? L.qpos[lane] : 0.f, v0 = lane < tm->njnt ? L.qvel[lane] : 0.f;
  bool settled = T.settle_max == 0;
  for (int k = 0; k < T.settle_max && !settled; k++) {
    tree::forward(tm, gm, L, G, T.iterations, T.tolerance);
    tree::euler(tm, L);
    bool ok = true;
    if (lane < tm->nq) ok = ok && fabsf(L.qpos[lane]) <= 1e10f;
    if (lane < tm->nv) ok = ok && fabsf(L.qvel[lane]) <= 1e10f && fabsf(L.qacc[lane]) <= 1e10f;
    if (wave_ballot(!ok) != 0ull) { if (lane == 0) L.flags |= 8; break; }
    if (lane < tm->njnt) { L.qpos[lane] = q0; L.qvel[lane] = v0; }
    wave_sync();
    float mv = 0.f, ma = 0.f;
    if (lane >= tm->njnt && lane < tm->nv) { mv = fabsf(L.qvel[lane]); ma = fabsf(L.qacc[lane]); }
    mv = wave_max_f(mv); ma = wave_max_f(ma);
    settled = mv < 1e-3f && ma < 1e-2f;
  }
  if (!settled && lane == 0) L.flags |= 32;
  tree::store_state(tm, L, B, e, N);
  if (lane == 0 && B.diag) { int* d = B.diag + 8 * e; d[0] = L.ncon; d[1] = L.nrow; d[2] = L.iters; d[3] = L.ncand; d[4] = L.flags; }
}

// adopt the bound state as the post-reset state of a new episode (known-answer tests, checkpoints): delay lines filled with it,
// counters cleared, no reset pending
__global__ void __launch_bounds__(64) k_tree_begin(const TreeModel* tm, TreeTask T, TreeBuffers B, TreeEnvBuffers E) {
  BLOCK_SHARED(TreeLDS, L);
  int e = blockIdx.x, lane = wave_lane(), N = T.n_envs;
  tree::load_state(tm, L, B, e, N);
  tree::fill_delay_lines(tm, T, L, E, e);
  if (lane == 0) { E.step_count[e] = 0; E.ep_return[e] = 0.f; E.need_reset[e] = 0; E.success_state[e] = T.requires_handover ? 0 : 2; }
}

// end of a control step of one env (after the last substep): observation, reward / discount / termination, state and counters out
// CONTACTS: where the contact rewards take the contacts of the post-step state from - 0: not compiled (reward mode 0 only), 1: collision() in
// place (single kernel), 2: gathered from the launch chain's records (k_tree_pipe_finish)
template <int CONTACTS>
__device__ __forceinline__ void tree_finish_step(const TreeModel* tm, const DevModel* gm, const TreeTask& T, TreeLDS& L, TreeScratch& G, const TreeBuffers& B,
                                                 const TreeEnvBuffers& E, int e, bool diverged, float* obs, float* reward, float* discount, unsigned char* step_type,
                                                 const TreePipe* P = nullptr) {
  int lane = wave_lane(), N = T.n_envs;
  if (diverged) {         // mj_check*: the data is reset, dm_control ends the episode with reward 0 and discount 0
    if (lane < tm->nq) L.qpos[lane] = 0.f;
    if (lane < tm->nv) { L.qvel[lane] = 0.f; L.warm[lane] = 0.f; }
    wave_sync();
    if (lane > 0 && lane < tm->nbody && tm->body_jnttype[lane] == TJ_FREE) L.qpos[tm->body_qposadr[lane] + 3] = 1.f;
    if (lane == 0) L.flags |= 8;
    wave_sync();
  }
  int sc = E.step_count[e] + 1;
  tree::kinematics(tm, L);
  tree::write_obs(T, L, E, e, sc, false, obs);
  tree::write_physics_state(tm, T, L, E, e, sc);
  // dm_control's Environment.step asks the task three times: get_reward, get_discount (-> should_terminate_episode -> get_reward) and
  // should_terminate_episode (-> get_reward), aloha2_task.py:353-367.  For the overlap and touching rewards the three answers are the
  // same; the contact sequence (hand_over.py:286-338) advances its state machine on EVERY call, so reward, discount and termination may
  // each see a different state (0 -> 1 -> 2 and even the success within one control step).  Contacts: physics.data.contact after
  // physics.step(), whose legacy step ends with mj_step1 - the contacts of the integrated state, recomputed here.
  float r = 0.f, r_disc = 0.f, r_term = 0.f;
  if (!diverged) {
    if (CONTACTS == 0 || T.reward_mode == 0) r = r_disc = r_term = tree::task_reward(tm, T, L);
    else if (T.reward_mode == 2) {
      if constexpr (CONTACTS == 2) tree::gather_contacts(tm, gm, L, *P, e); else tree::collision(tm, gm, L);
      r = r_disc = r_term = tree::task_reward_touching(tm, T, L);
    } else {
      if constexpr (CONTACTS == 2) tree::gather_contacts(tm, gm, L, *P, e); else tree::collision(tm, gm, L);
      int st = E.success_state[e];
      r = tree::task_reward_contacts(tm, T, L, G, &st);
      if (T.terminate_on_success) { r_disc = tree::task_reward_contacts(tm, T, L, G, &st); r_term = tree::task_reward_contacts(tm, T, L, G, &st); }
      if (lane == 0) E.success_state[e] = st;
    }
  }
  bool disc0 = (T.terminate_on_success && r_disc >= 1.f) || diverged, success = (T.terminate_on_success && r_term >= 1.f) || diverged, timeout = sc >= T.last_step;
  tree::store_state(tm, L, B, e, N);
  if (lane < tm->nu) B.ctrl[(size_t)lane * N + e] = L.ctrl[lane];
  if (lane == 0) {
    reward[e] = r; discount[e] = disc0 ? 0.f : 1.f;
    unsigned char st = (success || timeout) ? 2 : 1;
    step_type[e] = st; E.need_reset[e] = st == 2;
    E.step_count[e] = sc; E.ep_return[e] += r;
    if (B.diag) { int* d = B.diag + 8 * e; d[0] = L.ncon; d[1] = L.nrow; d[2] = L.iters; d[3] = L.ncand; d[4] = L.flags; d[5] = E.success_state[e]; }
  }
}

// start of a control step of one env: an env whose last step was LAST resets and reports FIRST (returns true: done for this call);
// the others load their state and apply the action
__device__ __forceinline__ bool tree_begin_step(const TreeModel* tm, const DevModel* gm, const TreeTask& T, TreeLDS& L, TreeScratch& G, const TreeBuffers& B,
                                                const TreeEnvBuffers& E, const TreeStore& S, int e, const float* action, float* obs, float* reward, float* discount,
                                                unsigned char* step_type) {
  int lane = wave_lane(), N = T.n_envs;
  if (E.need_reset[e]) {
    tree::env_reset(tm, gm, T, L, G, B, E, S, e);
    tree::kinematics(tm, L);
    tree::write_obs(T, L, E, e, 0, true, obs);
    tree::store_state(tm, L, B, e, N);
    if (lane < tm->nu) B.ctrl[(size_t)lane * N + e] = L.ctrl[lane];
    if (lane == 0) { reward[e] = 0.f; discount[e] = 1.f; step_type[e] = 0; E.need_reset[e] = 0; }
    if (lane == 0 && B.diag) { int* d = B.diag + 8 * e; d[0] = L.ncon; d[1] = L.nrow; d[2] = L.iters; d[3] = L.ncand; d[4] = L.flags; }
    return true;
  }
  tree::load_state(tm, L, B, e, N);
  // before_step (aloha2_task.py:316-349): joint targets as given, grippers from follower units to the sim's ctrl range; no clipping
  if (lane < tm->nu) {
    float a = action[(size_t)e * tm->nu + lane];
    if (T.act_is_gripper[lane]) a = tree::convert_gripper(a, T.grip[4], T.grip[5], T.grip[2], T.grip[3]);
    L.ctrl[lane] = a;
  }
  wave_sync();
  return false;
}

// (The single-kernel step: N = 1, steps of more than 63 substeps, and the reference the launch chain below is tested against.  One wavefront per
// SIMD: 458 unified registers, the hull caches and the EPA polytope of the inlined narrowphase among them.  __launch_bounds__(64, 2) was measured
// in round 4 - 256 VGPRs, 770 spilled, 864 B of scratch per lane: 171 k against 175 k env-steps/s at 4096 envs.)
// one control step of every env: dm_control's Environment.step - an env whose last step was LAST resets and reports FIRST
// (the action is ignored), the others apply the action, run n_substeps and report observation, reward, discount, step type
__global__ void __launch_bounds__(64) k_tree_step(const TreeModel* tm, const DevModel* gm, TreeTask T, TreeBuffers B, TreeEnvBuffers E, TreeStore S, const float* action,
                                                  float* obs, float* reward, float* discount, unsigned char* step_type) {
  BLOCK_SHARED(TreeLDS, L);
  int e = blockIdx.x, lane = wave_lane(), N = T.n_envs;
  if (lane == 0) L.flags = 0;
  wave_sync();
  TreeScratch G = tree::scratch_of(B, e);
  if (tree_begin_step(tm, gm, T, L, G, B, E, S, e, action, obs, reward, discount, step_type)) return;
  bool diverged = false;
  for (int s = 0; s < T.n_substeps && !diverged; s++) {
    tree::forward(tm, gm, L, G, T.iterations, T.tolerance);
    tree::euler(tm, L);
    bool ok = true;
    if (lane < tm->nq) ok = ok && fabsf(L.qpos[lane]) <= 1e10f;
    if (lane < tm->nv) ok = ok && fabsf(L.qvel[lane]) <= 1e10f && fabsf(L.qacc[lane]) <= 1e10f;
    diverged = wave_ballot(!ok) != 0ull;
  }
  tree_finish_step<1>(tm, gm, T, L, G, B, E, e, diverged, obs, reward, discount, step_type);
}

#ifdef TREE_PROF
__device__ unsigned long long g_tprof[33];     // (profiling variant: phase clocks of k_tree_pipe_solve summed over env-substeps, printed by destroy)
#endif
// ---- the control step as a launch chain (so101_tree.hpp, "the narrowphase in a launch of its own"): k_tree_pipe_begin, then per substep
// k_tree_narrow and k_tree_pipe_solve.  Reward mode 0 (overlap boxes) only: the contact rewards need the contacts of the post-step state.
__global__ void __launch_bounds__(64) k_tree_pipe_begin(const TreeModel* tm, const DevModel* gm, TreeTask T, TreeBuffers B, TreeEnvBuffers E, TreeStore S, TreePipe P,
                                                        const float* action, float* obs, float* reward, float* discount, unsigned char* step_type, int e0) {
  BLOCK_SHARED(TreeLDS, L);
  int e = e0 + blockIdx.x, lane = wave_lane(), N = T.n_envs;
  if (lane == 0) L.flags = 0;
  wave_sync();
  TreeScratch G = tree::scratch_of(B, e);
  if (tree_begin_step(tm, gm, T, L, G, B, E, S, e, action, obs, reward, discount, step_type)) {
    if (lane == 0) { P.active[e] = 0; P.ncand[e] = 0; }
    return;
  }
  if (lane < tm->nu) B.ctrl[(size_t)lane * N + e] = L.ctrl[lane];
  if (lane == 0) { P.active[e] = 1; P.pflags[e] = 0; }
  tree::kinematics(tm, L);
  tree::publish(tm, gm, L, P, e, N, 0);
}

// one wavefront per candidate pair of the whole batch: persistent wavefronts take two work items per fetch
// (round 5: the hulls of a pair are staged in LDS - so101_device.hpp, HullLDS - instead of held in 48 VGPRs; with the hull patches inlined the register
//  cache left this kernel 55 spilled VGPRs and 184 B of scratch per lane)
__global__ void __launch_bounds__(64, 2) k_tree_narrow(const TreeModel* tm, const DevModel* gm, TreePipe P, int N, int s) {
  __shared__ __attribute__((aligned(16))) float pool[3 * 2 * HULL_LDS_MAX];
  int lane = wave_lane();
  const int nwork = ldc(&P.counters[2 * s]);
  const unsigned int* list = P.work + (size_t)(s & 1) * P.work_cap;
  for (;;) {
    int i0 = 0;
    if (lane == 0) i0 = atomicAdd(&P.counters[2 * s + 1], 2);
    i0 = wave_uniform_i(i0);
    if (i0 >= nwork) break;
    unsigned int wl = 0, cl = 0;
    if (lane < 2 && i0 + lane < nwork) { wl = list[i0 + lane]; cl = P.cand[wl]; }
#pragma unroll 1
    for (int j = 0; j < 2; j++) {
      if (i0 + j >= nwork) break;
      unsigned int w = (unsigned int)__builtin_amdgcn_readlane((int)wl, j), c = (unsigned int)__builtin_amdgcn_readlane((int)cl, j);
      int e = (int)(w / TCAND), g1 = (int)(c & 0xffffu), g2 = (int)(c >> 16);
      int b1 = wave_uniform_i(tm->geom_body[g1]), b2 = wave_uniform_i(tm->geom_body[g2]);
      const float* p1 = P.pose + ((size_t)e * TB + b1) * 12; const float* p2 = P.pose + ((size_t)e * TB + b2) * 12;
      GeomW G1, G2;
      load_geom_at(gm, g1, p1, p1 + 3, G1); load_geom_at(gm, g2, p2, p2 + 3, G2);
      PairContacts pc;
      HullLDS H1{pool, hull_lds_slots(G1.type, G1.vnum)}, H2{pool + 3 * HULL_LDS_MAX, hull_lds_slots(G2.type, G2.vnum)};
      wave_sync();                                     // (the scans of the previous pair are done)
      hull_load(gm, G1, H1); hull_load(gm, G2, H2);
      wave_sync();
      narrow_pair_cached<HullLDS, G64>(gm, G1, G2, ldc(ldc(&gm->geom_rbound) + g1), ldc(ldc(&gm->geom_rbound) + g2), H1, H2, pc);
      if (lane == 0) {
        float* r = P.rec + (size_t)w * TREC;
        r[0] = (float)__popc(pc.valid); r[1] = pc.nrm[0]; r[2] = pc.nrm[1]; r[3] = pc.nrm[2];
        int o = 4;                                     // valid slots are written compactly, in slot order
#pragma unroll
        for (int q = 0; q < NCPP; q++)
          if ((pc.valid >> q) & 1u) { r[o] = pc.dist[q]; r[o + 1] = pc.pos[q][0]; r[o + 2] = pc.pos[q][1]; r[o + 3] = pc.pos[q][2]; o += 4; }
      }
    }
  }
}

__global__ void __launch_bounds__(64, TREE_SOLVE_OCC) k_tree_pipe_solve(const TreeModel* tm, const DevModel* gm, TreeTask T, TreeBuffers B, TreeEnvBuffers E, TreePipe P, int s, int last,
                                                                       float* obs, float* reward, float* discount, unsigned char* step_type, int e0) {
  BLOCK_SHARED(TreeLDS, L);
  int e = e0 + blockIdx.x, lane = wave_lane(), N = T.n_envs;
  int act = P.active[e];
  if (act == 0) return;
  if (lane == 0) L.flags = P.pflags[e];
#ifdef TREE_PROF
  if (lane < 32) L.prof[lane] = 0u;
  wave_sync();
#endif
  TPROF_T0();
  tree::load_state(tm, L, B, e, N);
  TPROF(16)
  TreeScratch G = tree::scratch_of(B, e);
  bool diverged = act == 2;
  if (!diverged) {
    tree::forward_smooth(tm, L);
    TPROF(12)
    tree::gather_contacts(tm, gm, L, P, e);
    TPROF(21)
    tree::forward_constrained(tm, L, G, T.iterations, T.tolerance);
    TPROF(13)
    tree::euler(tm, L);
    bool ok = true;
    if (lane < tm->nq) ok = ok && fabsf(L.qpos[lane]) <= 1e10f;
    if (lane < tm->nv) ok = ok && fabsf(L.qvel[lane]) <= 1e10f && fabsf(L.qacc[lane]) <= 1e10f;
    diverged = wave_ballot(!ok) != 0ull;
    if (diverged && lane == 0) P.active[e] = 2;
    TPROF(22)
  }
  if (!last) {
    if (act == 1) {
      tree::store_state(tm, L, B, e, N);
      if (lane == 0) { P.pflags[e] = L.flags; P.pdiag[4 * e] = L.nrow; P.pdiag[4 * e + 1] = L.iters; P.pdiag[4 * e + 2] = L.ncon; P.pdiag[4 * e + 3] = L.ncand; }
      TPROF(23)
      if (!diverged) {
        tree::kinematics(tm, L);
        TPROF(24)
        tree::publish(tm, gm, L, P, e, N, s + 1);
        TPROF(25)
      }
      else if (lane == 0) P.ncand[e] = 0;
    }
#ifdef TREE_PROF
    wave_sync();
    if (lane < 32) atomicAdd(&g_tprof[lane], (unsigned long long)L.prof[lane]);
    if (lane == 0) atomicAdd(&g_tprof[32], 1ull);
#endif
    return;
  }
  if (act == 2) {
    // diverged in an earlier substep: this launch computed nothing, the diagnostics are those of the last substep that was
    if (lane == 0) { L.nrow = P.pdiag[4 * e]; L.iters = P.pdiag[4 * e + 1]; L.ncon = P.pdiag[4 * e + 2]; L.ncand = P.pdiag[4 * e + 3]; }
    wave_sync();
  }
  tree_finish_step<0>(tm, gm, T, L, G, B, E, e, diverged, obs, reward, discount, step_type);
}

// end of a control step whose reward needs the contacts of the post-step state (reward modes 1 and 2): the last k_tree_pipe_solve published that
// state's candidates like a substep's, k_tree_narrow has turned them into records
__global__ void __launch_bounds__(64, 2) k_tree_pipe_finish(const TreeModel* tm, const DevModel* gm, TreeTask T, TreeBuffers B, TreeEnvBuffers E, TreePipe P,
                                                            float* obs, float* reward, float* discount, unsigned char* step_type, int e0) {
  BLOCK_SHARED(TreeLDS, L);
  int e = e0 + blockIdx.x, lane = wave_lane(), N = T.n_envs;
  int act = P.active[e];
  if (act == 0) return;
  if (lane == 0) { L.flags = P.pflags[e]; L.nrow = P.pdiag[4 * e]; L.iters = P.pdiag[4 * e + 1]; L.ncon = P.pdiag[4 * e + 2]; L.ncand = P.pdiag[4 * e + 3]; }
  tree::load_state(tm, L, B, e, N);
  TreeScratch G = tree::scratch_of(B, e);
  tree_finish_step<2>(tm, gm, T, L, G, B, E, e, act == 2, obs, reward, discount, step_type, &P);
}

// ==================================================================================================== host side
struct TreeHandle {
  int n_envs = 0, device = 0;
  TreeModel hm{};
  DevModel hg{};
  TreeModel* dm = nullptr;
  DevModel* dg = nullptr;
  TreeBuffers buf{};
  bool bound = false, env_bound = false, has_task = false;
  TreeTask task{};
  TreeEnvBuffers env{};
  TreeStore store{};
  int iterations = 0; float tolerance = 0.f;
  // reset prefetch: second per-env scratch, low-priority stream, "previous launch still running" bookkeeping
  float* scratch2 = nullptr;
  float *cq = nullptr, *cv = nullptr, *cw = nullptr; int* cf = nullptr; unsigned int* ctag = nullptr;
  hipStream_t prep_stream = nullptr;
  hipEvent_t prep_done = nullptr, main_ev = nullptr;
  bool prep_pending = false, prefetch = false;
  // launch chain of the control step (so101_tree_config.pipeline): hand-off buffers, allocated when first asked for
  TreePipe pipe{};
  bool pipeline = false;
  static constexpr int MAXSLICES = 4;
  hipStream_t slice_stream[MAXSLICES] = {};      // env slices whose chains overlap (one's narrowphase beside another's solve)
  hipEvent_t slice_begin = nullptr, slice_done[MAXSLICES] = {};
  int plan[4] = {0, 0, 0, 0};          // what the last so101_tree_step enqueued: env slices, kernel launches, memsets, path (0 none yet, 1 single kernel, 2 launch chain)
  std::vector<void*> owned;
  std::string err;
};

namespace {
thread_local std::string g_tree_error;

// every entry point that touches HIP runs with the handle's device current and restores the caller's device on exit
struct TreeDeviceGuard {
  int prev = -1, dev;
  bool ok;
  explicit TreeDeviceGuard(TreeHandle* s) : dev(s->device) {
    ok = hipGetDevice(&prev) == hipSuccess && (prev == dev || hipSetDevice(dev) == hipSuccess);
    if (!ok) s->err = "hipSetDevice: cannot make the handle's device current";
  }
  ~TreeDeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};
#define TREE_GUARD(s) TreeDeviceGuard guard_(s); if (!guard_.ok) return SO101_ERR_HIP

void tq2m(float* m, const float* q) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = 1 - 2 * (y * y + z * z); m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = 1 - 2 * (x * x + z * z); m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = 1 - 2 * (x * x + y * y);
}
bool t_ok(TreeHandle* s, hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  s->err = std::string(what) + ": " + hipGetErrorString(e);
  return false;
}
template <class T>
bool t_upload(TreeHandle* s, const std::vector<T>& v, const T** out) {
  void* p = nullptr;
  if (!t_ok(s, hipMalloc(&p, (v.size() ? v.size() : 1) * sizeof(T)), "hipMalloc(tree model)")) return false;
  s->owned.push_back(p);
  if (!v.empty() && !t_ok(s, hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy(tree model)")) return false;
  *out = (const T*)p;
  return true;
}

int tree_build(TreeHandle* s, const BlobView& b) {
  auto fail = [&](const std::string& msg) { s->err = msg; return (int)SO101_ERR_MODEL; };
  for (const char* n : {"nq", "nv", "nu", "nbody", "ngeom", "narm", "nfree", "npair", "nvert", "neq", "opt_iterations", "opt_mpr_iterations", "opt_cone_elliptic",
                        "opt_timestep", "opt_impratio", "opt_tolerance", "opt_mpr_tolerance", "stat_meaninertia"})
    if (b.count(n) < 1) return fail(std::string("blob entry missing (not a general-tree model?): ") + n);
  TreeModel& M = s->hm;
  M.nq = b.I("nq")[0]; M.nv = b.I("nv")[0]; M.nu = b.I("nu")[0]; M.nbody = b.I("nbody")[0]; M.ngeom = b.I("ngeom")[0];
  M.npair = b.I("npair")[0]; M.njnt = b.I("narm")[0]; M.nfree = b.I("nfree")[0]; M.neq = b.I("neq")[0];
  int nvert = b.I("nvert")[0];
  if (M.nq > TQ || M.nv > TV || M.nu > TU || M.nbody > TB || M.ngeom > TGEOM || M.njnt > TJ || M.neq > TE || M.nq < 0 || M.nv < 0 || M.nbody < 1)
    return fail("model dimensions outside the general-tree builds (32 bodies, 64 dofs, 64 positions, 16 actuators, 256 geoms)");
  size_t nb = M.nbody, ng = M.ngeom, nv = M.nv, nj = M.njnt, nu = M.nu, ne = M.neq, np = M.npair;
  struct Need { const char* name; size_t count; };
  const Need arrays[] = {
    {"body_parent", nb}, {"body_jnttype", nb}, {"body_qposadr", nb}, {"body_dofadr", nb}, {"body_pos", 3 * nb}, {"body_quat", 4 * nb}, {"body_ipos", 3 * nb},
    {"body_iquat", 4 * nb}, {"body_mass", nb}, {"body_inertia", 3 * nb}, {"body_invweight0", 2 * nb}, {"arm_body", nj}, {"dof_body", nv}, {"dof_armature", nv},
    {"dof_damping", nv}, {"dof_frictionloss", nv}, {"dof_invweight0", nv}, {"dof_solref", 2 * nj}, {"dof_solimp", 5 * nj}, {"jnt_axis", 3 * nj}, {"jnt_range", 2 * nj},
    {"jnt_limited", nj}, {"jnt_solref", 2 * nj}, {"jnt_solimp", 5 * nj}, {"jnt_actfrclimited", nj}, {"jnt_actfrcrange", 2 * nj}, {"act_dof", nu}, {"act_gain", nu},
    {"act_bias", 3 * nu}, {"act_ctrlrange", 2 * nu}, {"act_forcerange", 2 * nu}, {"act_ctrllimited", nu}, {"act_forcelimited", nu}, {"eq_dof", 2 * ne},
    {"eq_qposadr", 2 * ne}, {"eq_polycoef", 5 * ne}, {"eq_solref", 2 * ne}, {"eq_solimp", 5 * ne}, {"opt_gravity", 3}, {"geom_type", ng}, {"geom_body", ng},
    {"geom_condim", ng}, {"geom_vertadr", ng}, {"geom_vertnum", ng}, {"geom_pos", 3 * ng}, {"geom_quat", 4 * ng}, {"geom_size", 3 * ng}, {"geom_friction", 3 * ng},
    {"geom_solref", 2 * ng}, {"geom_solimp", 5 * ng}, {"geom_center", 3 * ng}, {"geom_aabb", 6 * ng}, {"geom_solmix", ng}, {"geom_priority", ng}, {"geom_rbound", ng},
    {"geom_margin", ng}, {"geom_gap", ng}, {"mesh_vert", 3 * (size_t)nvert}, {"pair_geom", 2 * np}};
  for (const Need& a : arrays) if (b.count(a.name) < a.count) return fail(std::string("blob entry missing or too short: ") + a.name);
  auto in_range = [&](const char* name, size_t limit) { for (int v : b.I(name)) if (v < 0 || (size_t)v >= limit) return false; return true; };
  if (!in_range("body_parent", nb) || !in_range("geom_body", nb) || !in_range("pair_geom", ng) || !in_range("arm_body", nb) || !in_range("dof_body", nb) ||
      !in_range("act_dof", nv) || !in_range("eq_dof", nv) || !in_range("eq_qposadr", M.nq)) return fail("blob index array out of range");
  auto gva = b.I("geom_vertadr"), gvn = b.I("geom_vertnum");
  for (size_t g = 0; g < ng; g++) if (gvn[g] > 0 && (gva[g] < 0 || (size_t)gva[g] + (size_t)gvn[g] > (size_t)nvert)) return fail("geom vertex range outside mesh_vert");
  for (float v : b.F("geom_margin")) if (v != 0.f) return fail("geom margin must be 0");
  for (float v : b.F("geom_gap")) if (v != 0.f) return fail("geom gap must be 0");

  M.elliptic = b.I("opt_cone_elliptic")[0]; M.iterations = b.I("opt_iterations")[0];
  M.dt = b.F("opt_timestep")[0]; M.impratio = b.F("opt_impratio")[0]; M.tolerance = b.F("opt_tolerance")[0]; M.meaninertia = b.F("stat_meaninertia")[0];
  auto grav = b.F("opt_gravity"); for (int k = 0; k < 3; k++) M.grav[k] = grav[k];
  auto parent = b.I("body_parent"), jt = b.I("body_jnttype"), qadr = b.I("body_qposadr"), dadr = b.I("body_dofadr"), armb = b.I("arm_body");
  auto bpos = b.F("body_pos"), bquat = b.F("body_quat"), ipos = b.F("body_ipos"), iquat = b.F("body_iquat"), mass = b.F("body_mass"), inertia = b.F("body_inertia"),
       invw = b.F("body_invweight0");
  M.maxdepth = 0;
  for (int i = 0; i < M.nbody; i++) {
    if (i > 0 && parent[i] >= i) return fail("bodies must be numbered parents first");
    M.body_parent[i] = parent[i]; M.body_jnttype[i] = jt[i]; M.body_qposadr[i] = qadr[i]; M.body_dofadr[i] = dadr[i]; M.body_jnt[i] = -1;
    M.body_depth[i] = i == 0 ? 0 : M.body_depth[parent[i]] + 1;
    M.body_anc[i] = (i == 0 ? 0u : M.body_anc[parent[i]]) | (1u << i);
    M.body_dofs[i] = i == 0 ? 0ull : M.body_dofs[parent[i]];
    if (M.body_depth[i] > M.maxdepth) M.maxdepth = M.body_depth[i];
    for (int k = 0; k < 3; k++) { M.body_pos[i][k] = bpos[3 * i + k]; M.body_ipos[i][k] = ipos[3 * i + k]; M.body_inertia[i][k] = inertia[3 * i + k]; }
    for (int k = 0; k < 4; k++) { M.body_quat[i][k] = bquat[4 * i + k]; M.body_iquat[i][k] = iquat[4 * i + k]; }
    M.body_mass[i] = mass[i]; M.body_invweight0[i][0] = invw[2 * i]; M.body_invweight0[i][1] = invw[2 * i + 1];
    int ndof = jt[i] == TJ_FREE ? 6 : (jt[i] == TJ_HINGE || jt[i] == TJ_SLIDE ? 1 : 0);
    if (ndof && (dadr[i] < 0 || dadr[i] + ndof > M.nv)) return fail("body dof address out of range");
    if (ndof && (qadr[i] < 0 || qadr[i] + (ndof == 6 ? 7 : 1) > M.nq)) return fail("body qpos address out of range");
    for (int k = 0; k < ndof; k++) M.body_dofs[i] |= 1ull << (dadr[i] + k);
  }
  auto dbody = b.I("dof_body");
  auto arma = b.F("dof_armature"), damp = b.F("dof_damping"), floss = b.F("dof_frictionloss"), dinv = b.F("dof_invweight0"), dsr = b.F("dof_solref"), dsi = b.F("dof_solimp");
  auto jaxis = b.F("jnt_axis"), jrange = b.F("jnt_range"), jsr = b.F("jnt_solref"), jsi = b.F("jnt_solimp"), jfr = b.F("jnt_actfrcrange");
  auto jlim = b.I("jnt_limited"), jfl = b.I("jnt_actfrclimited");
  for (int j = 0; j < M.njnt; j++) {
    int body = armb[j];
    if (jt[body] != TJ_HINGE && jt[body] != TJ_SLIDE) return fail("arm_body entry without a one-dof joint");
    M.jnt_body[j] = body; M.body_jnt[body] = j; M.jnt_limited[j] = jlim[j]; M.jnt_actfrclimited[j] = jfl[j];
    for (int k = 0; k < 3; k++) M.jnt_axis[j][k] = jaxis[3 * j + k];
    for (int k = 0; k < 2; k++) { M.jnt_range[j][k] = jrange[2 * j + k]; M.jnt_solref[j][k] = jsr[2 * j + k]; M.jnt_actfrcrange[j][k] = jfr[2 * j + k]; }
    for (int k = 0; k < 5; k++) M.jnt_solimp[j][k] = jsi[5 * j + k];
  }
  M.nfric = 0; M.any_damping = 0;
  for (int d = 0; d < M.nv; d++) {
    int body = dbody[d];
    M.dof_body[d] = body; M.dof_jnt[d] = M.body_jnt[body];
    M.dof_armature[d] = arma[d]; M.dof_damping[d] = damp[d]; M.dof_frictionloss[d] = floss[d]; M.dof_invweight0[d] = dinv[d];
    if (damp[d] > 0.f) M.any_damping = 1;
    int j = M.dof_jnt[d];
    M.dof_solref[d][0] = j >= 0 ? dsr[2 * j] : 0.02f; M.dof_solref[d][1] = j >= 0 ? dsr[2 * j + 1] : 1.f;
    const float defimp[5] = {0.9f, 0.95f, 0.001f, 0.5f, 2.f};
    for (int k = 0; k < 5; k++) M.dof_solimp[d][k] = j >= 0 ? dsi[5 * j + k] : defimp[k];
    if (floss[d] > 0.f) { if (M.nfric >= TFR) return fail("too many dofs with frictionloss"); M.fric_dof[M.nfric++] = d; }
  }
  auto adof = b.I("act_dof"), acl = b.I("act_ctrllimited"), afl = b.I("act_forcelimited");
  auto again = b.F("act_gain"), abias = b.F("act_bias"), acr = b.F("act_ctrlrange"), afr = b.F("act_forcerange");
  for (int a = 0; a < M.nu; a++) {
    M.act_dof[a] = adof[a]; M.act_ctrllimited[a] = acl[a]; M.act_forcelimited[a] = afl[a]; M.act_gain[a] = again[a];
    int body = dbody[adof[a]];
    if (jt[body] == TJ_FREE) return fail("actuator on a free body");
    M.act_qposadr[a] = qadr[body];
    for (int k = 0; k < 3; k++) M.act_bias[a][k] = abias[3 * a + k];
    for (int k = 0; k < 2; k++) { M.act_ctrlrange[a][k] = acr[2 * a + k]; M.act_forcerange[a][k] = afr[2 * a + k]; }
    for (int a2 = 0; a2 < a; a2++) if (adof[a2] == adof[a]) return fail("two actuators on one dof");
  }
  auto edof = b.I("eq_dof"), eqa = b.I("eq_qposadr");
  auto epc = b.F("eq_polycoef"), esr = b.F("eq_solref"), esi = b.F("eq_solimp");
  for (int e = 0; e < M.neq; e++) {
    for (int k = 0; k < 2; k++) { M.eq_dof[e][k] = edof[2 * e + k]; M.eq_qposadr[e][k] = eqa[2 * e + k]; M.eq_solref[e][k] = esr[2 * e + k]; }
    for (int k = 0; k < 5; k++) { M.eq_polycoef[e][k] = epc[5 * e + k]; M.eq_solimp[e][k] = esi[5 * e + k]; }
  }
  if (M.neq + M.nfric > 32) return fail("too many scalar constraint rows");

  // geometry: the tables the shared narrowphase reads (DevModel), every geom relative to its body
  DevModel& G = s->hg;
  G.ngeom = M.ngeom; G.npair = M.npair; G.nvert = nvert; G.iterations = M.iterations; G.mpr_iter = b.I("opt_mpr_iterations")[0];
  G.dt = M.dt; G.mpr_tol = b.F("opt_mpr_tolerance")[0]; G.impratio = M.impratio; G.tolerance = M.tolerance; G.meaninertia = M.meaninertia;
  auto gquat = b.F("geom_quat");
  std::vector<float> gmat(9 * ng);
  for (size_t g = 0; g < ng; g++) tq2m(&gmat[9 * g], &gquat[4 * g]);
  auto mv = b.F("mesh_vert");
  std::vector<float> vx(nvert), vy(nvert), vz(nvert);
  for (int i = 0; i < nvert; i++) { vx[i] = mv[3 * i]; vy[i] = mv[3 * i + 1]; vz[i] = mv[3 * i + 2]; }
  auto gtype = b.I("geom_type"), pairs = b.I("pair_geom");
  std::vector<unsigned int> packed(np);
  for (size_t p = 0; p < np; p++) {
    int g1 = pairs[2 * p], g2 = pairs[2 * p + 1];
    if (gtype[g1] > gtype[g2]) std::swap(g1, g2);
    packed[p] = (unsigned)g1 | ((unsigned)g2 << 8) | (gtype[g1] == G_PLANE ? 1u << 16 : 0u);
  }
  // support-bound tables of the hulls (so101_model.hpp DevModel::hull_sbt): in double, rounded up to float
  std::vector<float> sbt((size_t)ng * SBT_DIM, 0.f);
  for (size_t g = 0; g < ng; g++) {
    if (gtype[g] != G_MESH) continue;
    for (int face = 0; face < 6; face++) {
      int ax = face / 2; double sg = (face & 1) ? -1.0 : 1.0;
      for (int iu = 0; iu < SBT_GRID; iu++)
        for (int iv = 0; iv < SBT_GRID; iv++) {
          const double step = 2.0 / (SBT_GRID - 1);
          double c[3]; c[ax] = sg; c[(ax + 1) % 3] = -1.0 + step * iu; c[(ax + 2) % 3] = -1.0 + step * iv;
          double best = -1e300;
          for (int k = gva[g]; k < gva[g] + gvn[g]; k++) best = std::max(best, (double)mv[3 * k] * c[0] + (double)mv[3 * k + 1] * c[1] + (double)mv[3 * k + 2] * c[2]);
          float f = (float)best;
          if ((double)f < best) f = std::nextafterf(f, 3.0e38f);
          sbt[g * SBT_DIM + (face * SBT_GRID + iu) * SBT_GRID + iv] = f;
        }
    }
  }
  G.hull_sbt = nullptr;
  bool ok = (getenv("SO101_NO_SBT") != nullptr || t_upload(s, sbt, &G.hull_sbt)) &&
            t_upload(s, gtype, &G.geom_type) && t_upload(s, b.I("geom_body"), &G.geom_dyn) && t_upload(s, b.I("geom_condim"), &G.geom_condim) &&
            t_upload(s, gva, &G.geom_vertadr) && t_upload(s, gvn, &G.geom_vertnum) && t_upload(s, b.F("geom_pos"), &G.geom_pos) && t_upload(s, gmat, &G.geom_mat) &&
            t_upload(s, b.F("geom_size"), &G.geom_size) && t_upload(s, b.F("geom_friction"), &G.geom_friction) && t_upload(s, b.F("geom_solref"), &G.geom_solref) &&
            t_upload(s, b.F("geom_solimp"), &G.geom_solimp) && t_upload(s, b.F("geom_center"), &G.geom_center) && t_upload(s, b.F("geom_aabb"), &G.geom_aabb) &&
            t_upload(s, b.F("geom_rbound"), &G.geom_rbound) && t_upload(s, vx, &G.vx) && t_upload(s, vy, &G.vy) && t_upload(s, vz, &G.vz) &&
            t_upload(s, pairs, &G.pair) && t_upload(s, packed, &G.pair_packed) && t_upload(s, b.I("geom_body"), &M.geom_body) &&
            t_upload(s, b.F("geom_solmix"), &M.geom_solmix) && t_upload(s, b.I("geom_priority"), &M.geom_priority);
  M.geom_class = nullptr;
  if (ok && b.count("task_geom_class") >= ng) ok = t_upload(s, b.I("task_geom_class"), &M.geom_class);
  if (!ok) return SO101_ERR_HIP;
  // task layer (hand-over scenes): absent from the bare-arm blob
  TreeTask& T = s->task;
  T.n_substeps = 10; T.last_step = 1 << 30; T.settle_max = 1000; T.terminate_on_success = 1;
  s->has_task = b.count("task_object_body") >= 1 && b.I("task_object_body")[0] > 0 && b.count("task_obs_qposadr") >= 1;
  if (s->has_task) {
    size_t nbox = b.count("task_nbox") ? (size_t)b.I("task_nbox")[0] : 99;
    size_t npos = b.count("task_obs_qposadr");
    const Need tneed[] = {{"task_container_body", 1}, {"task_box_pos", 3 * nbox}, {"task_box_half", 3 * nbox}, {"task_obj_pos_lo", 3}, {"task_obj_pos_hi", 3},
                          {"task_obj_yaw", 2}, {"task_con_pos_lo", 3}, {"task_con_pos_hi", 3}, {"task_home_ctrl", nu}, {"task_home_qpos", nj},
                          {"task_obs_is_gripper", npos}, {"task_act_is_gripper", nu}, {"task_gripper_limits", 6}, {"body_bvh_aabb", 6 * nb}};
    if (nbox > 2 || npos > TU || npos > 64) return fail("task dimensions out of range");
    for (const Need& a : tneed) if (b.count(a.name) < a.count) return fail(std::string("blob entry missing or too short: ") + a.name);
    T.obj_body = b.I("task_object_body")[0]; T.con_body = b.I("task_container_body")[0]; T.nbox = (int)nbox;
    T.dist_threshold = b.count("task_dist_threshold") ? b.F("task_dist_threshold")[0] : 0.f;
    if (T.obj_body <= 0 || T.obj_body >= M.nbody || T.con_body <= 0 || T.con_body >= M.nbody || jt[T.obj_body] != TJ_FREE || jt[T.con_body] != TJ_FREE)
      return fail("task bodies must be free bodies");
    T.npos = (int)npos; T.nvel = M.njnt;
    auto oq = b.I("task_obs_qposadr"), og = b.I("task_obs_is_gripper"), ag = b.I("task_act_is_gripper");
    for (int k = 0; k < T.npos; k++) { if (oq[k] < 0 || oq[k] >= M.njnt) return fail("task_obs_qposadr out of range"); T.obs_qposadr[k] = oq[k]; T.obs_is_gripper[k] = og[k]; }
    if (T.npos != M.nu) return fail("one commanded position per actuator expected");
    for (int k = 0; k < M.nu; k++) T.act_is_gripper[k] = ag[k];
    auto gl = b.F("task_gripper_limits"); for (int k = 0; k < 6; k++) T.grip[k] = gl[k];
    auto bp = b.F("task_box_pos"), bh = b.F("task_box_half"), bvh = b.F("body_bvh_aabb");
    for (int k = 0; k < T.nbox; k++) for (int i = 0; i < 3; i++) { T.box_pos[k][i] = bp[3 * k + i]; T.box_half[k][i] = bh[3 * k + i]; }
    for (int i = 0; i < 6; i++) T.obj_bvh[i] = bvh[6 * T.obj_body + i];
    auto lo = b.F("task_obj_pos_lo"), hi = b.F("task_obj_pos_hi"), yaw = b.F("task_obj_yaw"), clo = b.F("task_con_pos_lo"), chi = b.F("task_con_pos_hi");
    for (int i = 0; i < 3; i++) { T.obj_lo[i] = lo[i]; T.obj_hi[i] = hi[i]; T.con_lo[i] = clo[i]; T.con_hi[i] = chi[i]; }
    T.obj_yaw[0] = yaw[0]; T.obj_yaw[1] = yaw[1];
    T.kind = b.count("task_kind") ? b.I("task_kind")[0] : 0;
    if (T.kind == 1) {                          // Dining (tasks/base/dining.py): six props, six regions
      if (b.count("task_prop_bodies") < 6 || b.count("task_region_lo") < 18 || b.count("task_region_hi") < 18) return fail("dining blob: task_prop_bodies / task_region_lo / task_region_hi missing");
      auto pbod = b.I("task_prop_bodies"); auto rlo = b.F("task_region_lo"), rhi = b.F("task_region_hi");
      for (int k = 0; k < 6; k++) {
        if (pbod[k] < 1 || pbod[k] >= M.nbody || jt[pbod[k]] != TJ_FREE) return fail("dining blob: task_prop_bodies must name free bodies");
        T.prop_body[k] = pbod[k];
        for (int i = 0; i < 3; i++) { T.region_lo[k][i] = rlo[3 * k + i]; T.region_hi[k][i] = rhi[3 * k + i]; }
      }
    } else if (T.kind != 0) return fail("unknown task_kind");
    auto hq = b.F("task_home_qpos"), hc = b.F("task_home_ctrl");
    for (int k = 0; k < M.njnt; k++) T.home_qpos[k] = hq[k];
    for (int k = 0; k < M.nu; k++) T.home_ctrl[k] = hc[k];
  }
  void* p = nullptr;
  if (!t_ok(s, hipMalloc(&p, sizeof(TreeModel)), "hipMalloc(TreeModel)")) return SO101_ERR_HIP;
  s->owned.push_back(p); s->dm = (TreeModel*)p;
  if (!t_ok(s, hipMemcpy(p, &M, sizeof(TreeModel), hipMemcpyHostToDevice), "hipMemcpy(TreeModel)")) return SO101_ERR_HIP;
  if (!t_ok(s, hipMalloc(&p, sizeof(DevModel)), "hipMalloc(DevModel)")) return SO101_ERR_HIP;
  s->owned.push_back(p); s->dg = (DevModel*)p;
  if (!t_ok(s, hipMemcpy(p, &G, sizeof(DevModel), hipMemcpyHostToDevice), "hipMemcpy(DevModel)")) return SO101_ERR_HIP;
  return SO101_OK;
}
}  // namespace

static TreeTask task_now(TreeHandle* s) { TreeTask T = s->task; T.n_envs = s->n_envs; T.iterations = s->iterations; T.tolerance = s->tolerance; return T; }
// the store the kernels see: the caller's settled-state tables plus, with the prefetch on, the library's cache of next-episode states
static TreeStore store_now(TreeHandle* s) {
  TreeStore S = s->store;
  S.cq = S.cv = S.cw = nullptr; S.cf = nullptr; S.ctag = nullptr;
  if (s->prefetch && s->ctag) { S.cq = s->cq; S.cv = s->cv; S.cw = s->cw; S.cf = s->cf; S.ctag = s->ctag; }
  return S;
}
static bool tree_prefetch_setup(TreeHandle* s) {          // lazily: second scratch, cache, low-priority stream
  if (s->ctag) return true;
  size_t n = (size_t)s->n_envs;
  auto alloc = [&](void** out, size_t bytes) {
    void* p = nullptr;
    if (!t_ok(s, hipMalloc(&p, bytes), "hipMalloc(prefetch)")) return false;
    s->owned.push_back(p); *out = p;
    return t_ok(s, hipMemset(p, 0, bytes), "hipMemset(prefetch)");
  };
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  return alloc((void**)&s->scratch2, n * T_SCRATCH * sizeof(float)) && alloc((void**)&s->cq, n * s->hm.nq * sizeof(float)) &&
         alloc((void**)&s->cv, n * s->hm.nv * sizeof(float)) && alloc((void**)&s->cw, n * s->hm.nv * sizeof(float)) && alloc((void**)&s->cf, n * sizeof(int)) &&
         t_ok(s, hipStreamCreateWithPriority(&s->prep_stream, hipStreamNonBlocking, lo), "hipStreamCreateWithPriority") &&
         t_ok(s, hipEventCreateWithFlags(&s->prep_done, hipEventDisableTiming), "hipEventCreate") &&
         t_ok(s, hipEventCreateWithFlags(&s->main_ev, hipEventDisableTiming), "hipEventCreate") && alloc((void**)&s->ctag, n * sizeof(unsigned int));
}
static bool tree_pipe_setup(TreeHandle* s) {             // lazily: the hand-off buffers of the launch chain
  if (s->pipe.pose) return true;
  size_t n = (size_t)s->n_envs;
  auto alloc = [&](void** out, size_t bytes) {
    void* p = nullptr;
    if (!t_ok(s, hipMalloc(&p, bytes), "hipMalloc(pipeline)")) return false;
    s->owned.push_back(p); *out = p;
    return t_ok(s, hipMemset(p, 0, bytes), "hipMemset(pipeline)");
  };
  TreePipe& P = s->pipe;
  void* pose = nullptr;
  bool ok = alloc((void**)&P.cand, n * TCAND * sizeof(unsigned int)) && alloc((void**)&P.ncand, n * sizeof(int)) && alloc((void**)&P.rec, n * TCAND * TREC * sizeof(float)) &&
            alloc((void**)&P.work, 2 * n * TCAND * sizeof(unsigned int)) && alloc((void**)&P.counters, TreeHandle::MAXSLICES * 2 * TPIPE_MAXSUB * sizeof(int)) && alloc((void**)&P.active, n) &&
            alloc((void**)&P.pflags, n * sizeof(int)) && alloc((void**)&P.pdiag, 4 * n * sizeof(int)) && alloc(&pose, n * TB * 12 * sizeof(float));
  for (int g = 0; ok && g < TreeHandle::MAXSLICES; g++)
    ok = t_ok(s, hipStreamCreateWithFlags(&s->slice_stream[g], hipStreamNonBlocking), "hipStreamCreate") &&
         t_ok(s, hipEventCreateWithFlags(&s->slice_done[g], hipEventDisableTiming), "hipEventCreate");
  ok = ok && t_ok(s, hipEventCreateWithFlags(&s->slice_begin, hipEventDisableTiming), "hipEventCreate");
  if (ok) P.pose = (float*)pose;          // (last: marks the set as complete)
  return ok;
}
static bool drain_tree_prepare(TreeHandle* s) {
  s->prep_pending = false;
  return !s->prep_stream || t_ok(s, hipStreamSynchronize(s->prep_stream), "hipStreamSynchronize(prefetch)");
}
// k_tree_prepare behind whatever `stream` holds now, unless the previous launch is still running (it takes every env whose next episode is
// missing, so a skipped launch only delays the refill)
static void launch_tree_prepare(TreeHandle* s, hipStream_t stream) {
  if (!s->prefetch || !s->ctag || s->task.settle_max == 0) return;
  if (s->prep_pending) {
    if (hipEventQuery(s->prep_done) != hipSuccess) { (void)hipGetLastError(); return; }
    s->prep_pending = false;
  }
  if (hipEventRecord(s->main_ev, stream) != hipSuccess || hipStreamWaitEvent(s->prep_stream, s->main_ev, 0) != hipSuccess) return;
  TreeBuffers B2 = s->buf; B2.scratch = s->scratch2;
  hipLaunchKernelGGL(k_tree_prepare, dim3(s->n_envs), dim3(64), 0, s->prep_stream, s->dm, s->dg, task_now(s), B2, (const int*)s->env.episode, store_now(s));
  if (hipGetLastError() == hipSuccess && hipEventRecord(s->prep_done, s->prep_stream) == hipSuccess) s->prep_pending = true;
}


extern "C" {

int TAPI(create)(const void* blob, size_t bytes, int n_envs, int hip_device, TreeHandle** out) {
  if (!out) return SO101_ERR_ARG;
  *out = nullptr;
  if (!blob || n_envs <= 0) { g_tree_error = "so101_tree_create: bad argument"; return SO101_ERR_ARG; }
  BlobView b;
  if (!b.parse(blob, bytes, g_tree_error)) return SO101_ERR_MODEL;
  TreeHandle* s = new TreeHandle();
  s->n_envs = n_envs; s->device = hip_device;
  int rc = SO101_OK;
  TreeDeviceGuard guard(s);                       // (the caller's current device is restored on every return path)
  if (!guard.ok) rc = SO101_ERR_HIP;
  if (rc == SO101_OK) rc = tree_build(s, b);
  if (rc == SO101_OK) {
    void* p = nullptr;
    if (!t_ok(s, hipMalloc(&p, (size_t)n_envs * T_SCRATCH * sizeof(float)), "hipMalloc(scratch)")) rc = SO101_ERR_HIP;
    else { s->owned.push_back(p); s->buf.scratch = (float*)p; }
  }
  if (rc == SO101_OK) {
    void* p = nullptr;
    if (!t_ok(s, hipMalloc(&p, (size_t)n_envs * 8 * sizeof(int)), "hipMalloc(diag)")) rc = SO101_ERR_HIP;
    else { s->owned.push_back(p); s->buf.diag = (int*)p; (void)hipMemset(p, 0, (size_t)n_envs * 8 * sizeof(int)); }
  }
  if (rc == SO101_OK) {
    void* p = nullptr;
    if (!t_ok(s, hipMalloc(&p, (size_t)n_envs), "hipMalloc(need_reset)")) rc = SO101_ERR_HIP;
    else { s->owned.push_back(p); s->env.need_reset = (unsigned char*)p; (void)hipMemset(p, 1, (size_t)n_envs); }
  }
  if (rc == SO101_OK) {
    void* p = nullptr;
    if (!t_ok(s, hipMalloc(&p, (size_t)n_envs * sizeof(int)), "hipMalloc(success_state)")) rc = SO101_ERR_HIP;
    else { s->owned.push_back(p); s->env.success_state = (int*)p; (void)hipMemset(p, 0, (size_t)n_envs * sizeof(int)); }
  }
  if (rc != SO101_OK) { g_tree_error = s->err; for (void* p : s->owned) (void)hipFree(p); delete s; return rc; }
  s->iterations = s->hm.iterations; s->tolerance = s->hm.tolerance;
  *out = s;
  return SO101_OK;
}

void TAPI(destroy)(TreeHandle* s) {
  if (!s) return;
  {
    TreeDeviceGuard guard(s);
    (void)hipDeviceSynchronize();                 // nothing of this handle may still be running on buffers that are about to go
#ifdef TREE_PROF
    {
      unsigned long long h[33] = {};
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tprof), sizeof(h));
      if (h[32]) {
        static const char* nm[32] = {"cost: M x + jar", "cost: blocks", "cost: gradient", "cost: H = M", "cost: H scalar rows", "cost: H contacts", "chol factor", "chol solve",
                                     "ls setup (J search)", "line search", "make_constraints", "newton total", "SMOOTH HALF", "CONSTRAINED HALF", "", "newton loop top (incl. cost)",
                                     "load_state", "kinematics", "crba + factor", "rne", "smooth", "gather_contacts", "euler + check", "store_state", "kinematics (next)", "publish (next broadphase)",
                                     "", "", "", "", "", ""};
        fprintf(stderr, "TREE_PROF k_tree_pipe_solve, %llu env-substeps, mean us each:\n", h[32]);
        for (int k = 0; k < 32; k++) if (nm[k][0]) fprintf(stderr, "  %-30s %8.2f\n", nm[k], (double)h[k] * 1e-2 / (double)h[32]);
      }
    }
#endif
    if (s->prep_stream) (void)hipStreamDestroy(s->prep_stream);
    if (s->prep_done) (void)hipEventDestroy(s->prep_done);
    if (s->main_ev) (void)hipEventDestroy(s->main_ev);
    for (int g = 0; g < TreeHandle::MAXSLICES; g++) { if (s->slice_stream[g]) (void)hipStreamDestroy(s->slice_stream[g]); if (s->slice_done[g]) (void)hipEventDestroy(s->slice_done[g]); }
    if (s->slice_begin) (void)hipEventDestroy(s->slice_begin);
    for (void* p : s->owned) (void)hipFree(p);
  }
  delete s;
}

const char* TAPI(last_error)(const TreeHandle* s) { return s ? s->err.c_str() : g_tree_error.c_str(); }

int TAPI(dims)(const TreeHandle* s, int* dims /* [16]: nq nv nu nbody ngeom debug_dim max_contacts, then the debug layout (offsets of bias, qacc_smooth, qacc, xpos, M, contacts, forces; row stride of M) and the build (32 / 64) */) {
  if (!s || !dims) return SO101_ERR_ARG;
  dims[0] = s->hm.nq; dims[1] = s->hm.nv; dims[2] = s->hm.nu; dims[3] = s->hm.nbody; dims[4] = s->hm.ngeom; dims[5] = TDBG_DIM; dims[6] = TCON;
  dims[7] = TDBG_BIAS; dims[8] = TDBG_QSM; dims[9] = TDBG_QACC; dims[10] = TDBG_XPOS; dims[11] = TDBG_M; dims[12] = TDBG_CON; dims[13] = TDBG_FORCE; dims[14] = TV; dims[15] = TREE_VARIANT;
  return SO101_OK;
}

int TAPI(last_plan)(const TreeHandle* s, int* out /* [4] */) {
  if (!s || !out) return SO101_ERR_ARG;
  for (int i = 0; i < 4; i++) out[i] = s->plan[i];
  return SO101_OK;
}

int TAPI(bind_state)(TreeHandle* s, float* qpos, float* qvel, float* ctrl, float* warmstart) {
  if (!s || !qpos || !qvel || !ctrl || !warmstart) { if (s) s->err = "so101_tree_bind_state: NULL buffer"; return SO101_ERR_ARG; }
  s->buf.qpos = qpos; s->buf.qvel = qvel; s->buf.ctrl = ctrl; s->buf.warm = warmstart;
  s->bound = true;
  return SO101_OK;
}

int TAPI(configure)(TreeHandle* s, int solver_iterations, float solver_tolerance) {
  if (!s) return SO101_ERR_ARG;
  s->iterations = solver_iterations > 0 ? solver_iterations : s->hm.iterations;
  s->tolerance = solver_tolerance >= 0.f ? solver_tolerance : s->hm.tolerance;
  return SO101_OK;
}

int TAPI(physics)(TreeHandle* s, int n_substeps, void* stream) {
  if (!s || n_substeps < 0) return SO101_ERR_ARG;
  if (!s->bound) { s->err = "so101_tree_physics before so101_tree_bind_state"; return SO101_ERR_STATE; }
  TREE_GUARD(s);
  hipLaunchKernelGGL(k_tree_physics, dim3(s->n_envs), dim3(64), 0, (hipStream_t)stream, s->dm, s->dg, s->buf, s->n_envs, n_substeps, s->iterations, s->tolerance);
  return t_ok(s, hipGetLastError(), "k_tree_physics") ? SO101_OK : SO101_ERR_HIP;
}

int TAPI(debug_forward)(TreeHandle* s, float* out, void* stream) {
  if (!s || !out) return SO101_ERR_ARG;
  if (!s->bound) { s->err = "so101_tree_debug_forward before so101_tree_bind_state"; return SO101_ERR_STATE; }
  TREE_GUARD(s);
  static const int phases = getenv("SO101_TREE_PHASES") ? atoi(getenv("SO101_TREE_PHASES")) : 0x7f;      // timing runs only
  hipLaunchKernelGGL(k_tree_forward, dim3(s->n_envs), dim3(64), 0, (hipStream_t)stream, s->dm, s->dg, s->buf, s->n_envs, s->iterations, s->tolerance, out, phases);
  return t_ok(s, hipGetLastError(), "k_tree_forward") ? SO101_OK : SO101_ERR_HIP;
}

// ---- env layer (hand-over scenes)
int TAPI(obs_dim)(const TreeHandle* s) { return s && s->has_task ? 3 * s->task.npos + 2 * s->task.nvel : 0; }

int TAPI(bind_env)(TreeHandle* s, float* ring_pos, float* ring_vel, float* ep_return, int32_t* step_count, int32_t* episode) {
  if (!s || !ring_pos || !ring_vel || !ep_return || !step_count || !episode) { if (s) s->err = "so101_tree_bind_env: NULL buffer"; return SO101_ERR_ARG; }
  if (!s->has_task) { s->err = "so101_tree_bind_env: the model carries no task (bare-arm blob)"; return SO101_ERR_STATE; }
  s->env.ring_pos = ring_pos; s->env.ring_vel = ring_vel; s->env.ep_return = ep_return; s->env.step_count = step_count; s->env.episode = episode;
  s->env_bound = true;
  return SO101_OK;
}

int TAPI(bind_physics_state)(TreeHandle* s, float* ring, float* physics_state, float* delayed) {
  if (!s) return SO101_ERR_ARG;
  if (!s->has_task) { s->err = "so101_tree_bind_physics_state: the model carries no task (bare-arm blob)"; return SO101_ERR_STATE; }
  bool all = ring && physics_state && delayed, none = !ring && !physics_state && !delayed;
  if (!all && !none) { s->err = "so101_tree_bind_physics_state: all three buffers or none"; return SO101_ERR_ARG; }
  s->env.ps_ring = ring; s->env.ps_out = physics_state; s->env.ps_delayed = delayed;
  return SO101_OK;
}

int TAPI(configure_env)(TreeHandle* s, const so101_tree_config* c) {
  if (!s || !c) return SO101_ERR_ARG;
  if (c->n_substeps < 1 || c->n_substeps > 1000 || c->last_step < 1 || c->settle_max_substeps < 0) { s->err = "so101_tree_configure_env: value out of range"; return SO101_ERR_ARG; }
  TreeTask& T = s->task;
  T.n_substeps = c->n_substeps; T.last_step = c->last_step; T.settle_max = c->settle_max_substeps; T.terminate_on_success = c->terminate_on_success;
  T.seed = c->seed; T.env_id_base = c->env_id_base;
  if (c->reward_mode < 0 || c->reward_mode > 2) { s->err = "so101_tree_configure_env: reward_mode must be 0, 1 or 2"; return SO101_ERR_ARG; }
  if (c->reward_mode != 0 && !s->hm.geom_class) { s->err = "so101_tree_configure_env: the model blob carries no task_geom_class (contact-sequence reward)"; return SO101_ERR_STATE; }
  T.reward_mode = c->reward_mode; T.requires_handover = c->reward_requires_handover;
  // observation delays in control steps; negative = the reference's defaults (0.1 s and 0.3 s at a 0.02 s control step)
  int jd = c->joints_delay_steps < 0 ? T_RING_DEFAULT : c->joints_delay_steps, pd = c->physics_delay_steps < 0 ? T_PS_DEFAULT : c->physics_delay_steps;
  if (jd > T_DELAY_MAX || pd > T_DELAY_MAX) { s->err = "so101_tree_configure_env: observation delay above 64 control steps"; return SO101_ERR_ARG; }
  T.jdelay = jd; T.pdelay = pd;
  // reset prefetch: whatever the cache holds was settled under the previous configuration
  {
    TREE_GUARD(s);
    if (!drain_tree_prepare(s)) return SO101_ERR_HIP;
    s->prefetch = c->prefetch_resets != 0;
    if (s->prefetch && !tree_prefetch_setup(s)) return SO101_ERR_HIP;
    s->pipeline = c->pipeline != 0;
    if (s->pipeline && !tree_pipe_setup(s)) return SO101_ERR_HIP;
    if (s->ctag && !t_ok(s, hipMemset(s->ctag, 0, (size_t)s->n_envs * sizeof(unsigned int)), "hipMemset(prefetch)")) return SO101_ERR_HIP;
  }
  return TAPI(configure)(s, c->solver_iterations, c->solver_tolerance);
}


int TAPI(reset)(TreeHandle* s, const uint8_t* mask, void* stream) {
  if (!s) return SO101_ERR_ARG;
  if (!s->bound || !s->env_bound) { s->err = "so101_tree_reset before so101_tree_bind_state / so101_tree_bind_env"; return SO101_ERR_STATE; }
  TREE_GUARD(s);
  hipLaunchKernelGGL(k_tree_reset, dim3(s->n_envs), dim3(64), 0, (hipStream_t)stream, s->dm, s->dg, task_now(s), s->buf, s->env, store_now(s), mask);
  if (!t_ok(s, hipGetLastError(), "k_tree_reset")) return SO101_ERR_HIP;
  launch_tree_prepare(s, (hipStream_t)stream);
  return SO101_OK;
}

int TAPI(step)(TreeHandle* s, const float* action, float* obs, float* reward, float* discount, uint8_t* step_type, void* stream) {
  if (!s || !action || !obs || !reward || !discount || !step_type) { if (s) s->err = "so101_tree_step: NULL argument"; return SO101_ERR_ARG; }
  if (!s->bound || !s->env_bound) { s->err = "so101_tree_step before so101_tree_bind_state / so101_tree_bind_env"; return SO101_ERR_STATE; }
  TREE_GUARD(s);
  hipStream_t st = (hipStream_t)stream;
  TreeTask T = task_now(s);
  const bool post = T.reward_mode != 0;          // contact rewards: one more narrowphase launch, on the post-step state
  if (s->pipeline && s->pipe.pose && T.n_substeps + (post ? 1 : 0) <= TPIPE_MAXSUB) {
    // launch chain: prologue, then per substep the narrowphase of every candidate pair of the batch and the rest of the substep per env.
    // Env slices on their own streams: the narrowphase of one (latency-bound, two wavefronts per SIMD) runs beside the solve of another
    // (four envs per CU by its LDS footprint); the slices share nothing but the model.  Measured, 1 / 2 / 3 / 4 slices:
    // ALOHA (4096 envs) 206 / 232 / 228 / 231 k, Dining (1024 envs) 33.2 / 44.6 / - / 47.0 k env-steps/s.
    static const int slices_env = getenv("SO101_TREE_SLICES") ? atoi(getenv("SO101_TREE_SLICES")) : 0;          // (kernel experiments)
    const int G = s->n_envs < 128 ? 1 : (slices_env >= 1 && slices_env <= TreeHandle::MAXSLICES ? slices_env : (s->n_envs < 512 ? 2 : 4));
    if (G > 1 && !t_ok(s, hipEventRecord(s->slice_begin, st), "hipEventRecord")) return SO101_ERR_HIP;
    s->plan[0] = G; s->plan[1] = G * (1 + 2 * T.n_substeps + (post ? 2 : 0)); s->plan[2] = G; s->plan[3] = 2;
    for (int g = 0; g < G; g++) {
      int e0 = (int)((long long)s->n_envs * g / G), ng = (int)((long long)s->n_envs * (g + 1) / G) - e0;
      hipStream_t gs = G == 1 ? st : s->slice_stream[g];
      TreePipe P = s->pipe;
      P.counters = s->pipe.counters + 2 * TPIPE_MAXSUB * g;
      P.work = s->pipe.work + (size_t)2 * e0 * TCAND;
      P.work_cap = (unsigned int)ng * TCAND;
      static const int nwq_env = getenv("SO101_TREE_NARROW_WAVES_Q") ? atoi(getenv("SO101_TREE_NARROW_WAVES_Q")) : 0;      // (kernel experiments: quarter waves per env)
      int nw = (int)((long long)ng * (nwq_env > 0 ? nwq_env : 8) / 4); nw = nw < 1 ? 1 : (nw < 4096 ? nw : 4096);
      if (G > 1 && !t_ok(s, hipStreamWaitEvent(gs, s->slice_begin, 0), "hipStreamWaitEvent")) return SO101_ERR_HIP;
      if (!t_ok(s, hipMemsetAsync(P.counters, 0, 2 * TPIPE_MAXSUB * sizeof(int), gs), "hipMemsetAsync(pipeline)")) return SO101_ERR_HIP;
      hipLaunchKernelGGL(k_tree_pipe_begin, dim3(ng), dim3(64), 0, gs, s->dm, s->dg, T, s->buf, s->env, store_now(s), P, action, obs, reward, discount, step_type, e0);
      for (int k = 0; k < T.n_substeps; k++) {
        hipLaunchKernelGGL(k_tree_narrow, dim3(nw), dim3(64), 0, gs, s->dm, s->dg, P, s->n_envs, k);
        hipLaunchKernelGGL(k_tree_pipe_solve, dim3(ng), dim3(64), 0, gs, s->dm, s->dg, T, s->buf, s->env, P, k, (int)(!post && k == T.n_substeps - 1), obs, reward, discount, step_type, e0);
      }
      if (post) {
        hipLaunchKernelGGL(k_tree_narrow, dim3(nw), dim3(64), 0, gs, s->dm, s->dg, P, s->n_envs, T.n_substeps);
        hipLaunchKernelGGL(k_tree_pipe_finish, dim3(ng), dim3(64), 0, gs, s->dm, s->dg, T, s->buf, s->env, P, obs, reward, discount, step_type, e0);
      }
      if (!t_ok(s, hipGetLastError(), "k_tree_pipe_solve")) return SO101_ERR_HIP;
      if (G > 1 && !(t_ok(s, hipEventRecord(s->slice_done[g], gs), "hipEventRecord") && t_ok(s, hipStreamWaitEvent(st, s->slice_done[g], 0), "hipStreamWaitEvent"))) return SO101_ERR_HIP;
    }
  } else {
    s->plan[0] = 1; s->plan[1] = 1; s->plan[2] = 0; s->plan[3] = 1;
    hipLaunchKernelGGL(k_tree_step, dim3(s->n_envs), dim3(64), 0, st, s->dm, s->dg, T, s->buf, s->env, store_now(s), action, obs, reward, discount, step_type);
    if (!t_ok(s, hipGetLastError(), "k_tree_step")) return SO101_ERR_HIP;
  }
  launch_tree_prepare(s, (hipStream_t)stream);
  return SO101_OK;
}

int TAPI(compute_settled)(TreeHandle* s, int first_episode, int count, float* qpos, float* qvel, float* warmstart, int32_t* flags, void* stream) {
  if (!s || !qpos || !qvel || !warmstart || !flags || first_episode < 0 || count <= 0) { if (s) s->err = "so101_tree_compute_settled: bad argument"; return SO101_ERR_ARG; }
  if (!s->has_task) { s->err = "so101_tree_compute_settled: the model carries no task"; return SO101_ERR_STATE; }
  TREE_GUARD(s);
  for (int k = 0; k < count; k++) {      // one launch per episode: the per-env scratch is used by one wavefront at a time
    hipLaunchKernelGGL(k_tree_settle_table, dim3(s->n_envs), dim3(64), 0, (hipStream_t)stream, s->dm, s->dg, task_now(s), s->buf, (unsigned int)(first_episode + k), k,
                       qpos, qvel, warmstart, flags);
    if (!t_ok(s, hipGetLastError(), "k_tree_settle_table")) return SO101_ERR_HIP;
  }
  return SO101_OK;
}

int TAPI(set_settled_store)(TreeHandle* s, int first_episode, int count, const float* qpos, const float* qvel, const float* warmstart, const int32_t* flags) {
  if (!s) return SO101_ERR_ARG;
  if (count > 0 && (!qpos || !qvel || !warmstart || !flags || first_episode < 0)) { s->err = "so101_tree_set_settled_store: bad argument"; return SO101_ERR_ARG; }
  s->store = TreeStore{};
  if (count > 0) { s->store.qpos = qpos; s->store.qvel = qvel; s->store.warm = warmstart; s->store.flags = flags; s->store.first = first_episode; s->store.count = count; }
  return SO101_OK;
}

int TAPI(settle)(TreeHandle* s, void* stream) {
  if (!s) return SO101_ERR_ARG;
  if (!s->bound) { s->err = "so101_tree_settle before so101_tree_bind_state"; return SO101_ERR_STATE; }
  TREE_GUARD(s);
  hipLaunchKernelGGL(k_tree_settle, dim3(s->n_envs), dim3(64), 0, (hipStream_t)stream, s->dm, s->dg, task_now(s), s->buf);
  return t_ok(s, hipGetLastError(), "k_tree_settle") ? SO101_OK : SO101_ERR_HIP;
}

int TAPI(begin_episode)(TreeHandle* s, void* stream) {
  if (!s) return SO101_ERR_ARG;
  if (!s->bound || !s->env_bound) { s->err = "so101_tree_begin_episode before so101_tree_bind_state / so101_tree_bind_env"; return SO101_ERR_STATE; }
  TREE_GUARD(s);
  hipLaunchKernelGGL(k_tree_begin, dim3(s->n_envs), dim3(64), 0, (hipStream_t)stream, s->dm, task_now(s), s->buf, s->env);
  return t_ok(s, hipGetLastError(), "k_tree_begin") ? SO101_OK : SO101_ERR_HIP;
}

int TAPI(get_diag)(TreeHandle* s, int* out /* [n_envs][8] device or host-visible memory */, void* stream) {
  if (!s || !out) return SO101_ERR_ARG;
  TREE_GUARD(s);
  return t_ok(s, hipMemcpyAsync(out, s->buf.diag, (size_t)s->n_envs * 8 * sizeof(int), hipMemcpyDefault, (hipStream_t)stream), "hipMemcpyAsync(diag)") ? SO101_OK : SO101_ERR_HIP;
}

}  // extern "C"
}  // namespace TREE_NS
