// Pipelined control step: the narrowphase leaves the env's wavefront.
//
// In the fused k_step one wavefront walks its env's candidate pairs one after the other, so a launch lasts as long
// as its most crowded env (props inside each other: 100+ candidate pairs against a mean of 16).  Here every substep
// is two launches:
//   k_narrow      one wavefront per CANDIDATE PAIR of the whole batch (persistent waves pulling from a work list),
//                 needs no LDS and half the registers of the fused kernel -> the machine stays full and balanced;
//   k_pipe_solve  one wavefront per env: smooth dynamics, gathers its contacts in candidate order, constraint rows,
//                 solver, Euler step, and the broadphase of the NEXT substep (which refills the work list).
// k_pipe_begin does the per-call prologue (auto-reset, before_step, first broadphase).  State makes a round trip
// through HBM per substep (~300 B/env) plus body poses, candidates and contact records (~1 KB/env).
// All stages call the same device functions as the fused path; contact order (= candidate order) is preserved.
//
// Env slices: a launch ends when its slowest env does (a Newton solve that needs 15 iterations instead of 3), and
// the next launch of the chain cannot start before that; with ~2.2 rounds of resident waves per launch that tail is
// about half of every k_pipe_solve.  k_order sorts the envs by the solver time of their previous control step and
// the sorted order is cut into up to three slices (default: n/4 most expensive, 3n/8, 3n/8) whose launch chains run
// on separate streams, so that one chain's tail is filled by the other chains' kernels: 365 k -> 452 k env-steps/s
// at 4096 envs.  Four chains are slower than one (264-276 k).  Every slice has its own work lists and counters;
// everything else is indexed by the global env index.  Results never depend on the slicing (each env is advanced
// by the same code on the same data).
#pragma once

#define MAXSUB 32

struct PipeBuffers {
  float* pose;            // [N][NDYN][12] xpos, xmat of the dynamic bodies
  unsigned int* cand;     // [N][MAXCAND]  geom1 | geom2 << 16, in pair-list order
  int* ncand;             // [N]           count | broadphase overflow flag << 16
  unsigned int* work;     // [2][work_cap] env * MAXCAND + k, double buffered over substeps (per env group)
  int* counters;          // [MAXSUB][2]   work items, cursor
  float* conres;          // [N][MAXCAND][8] dist, normal, position, valid
  unsigned char* active;  // [N] 0 not stepping in this call (auto-reset), 1 stepping, 2 diverged
  unsigned int* stage;    // [N][8] k_pipe_solve stage clocks of the last substep (10 ns ticks), diagnostics
  unsigned int* cost;     // [N] solver time of the env in its last substep (ticks): scheduling hint only
  int* order;             // [N] per group: env indices sorted by decreasing cost, see k_order
  unsigned int work_cap;  // capacity of one work list = envs of the group * MAXCAND
  unsigned int* ticks;    // [N][MAXCAND] narrowphase time of each candidate of the last substep (10 ns ticks), diagnostics
};

// hands the candidates in L.cand (and the poses the narrowphase needs) to substep s
DEV void publish_candidates(const EnvLDS& L, const PipeBuffers& W, int e, int N, int s) {
  int lane = wave_lane(), ncand = L.ncand;
  for (int i = lane; i < NDYN * 12; i += WAVE) {
    int b = i / 12, j = i % 12;
    W.pose[(size_t)e * (NDYN * 12) + i] = j < 3 ? L.xpos[b][j] : L.xmat[b][j - 3];
  }
  int base = 0;
  if (lane == 0) {
    base = ncand ? atomicAdd(&W.counters[2 * s], ncand) : 0;
    W.ncand[e] = ncand | ((L.overflow & 1) << 16);
  }
  base = wave_bcast_i(base, 0);
  unsigned int* list = W.work + (size_t)(s & 1) * W.work_cap;
  for (int k = lane; k < ncand; k += WAVE) {
    unsigned int w = (unsigned int)e * MAXCAND + k;
    W.cand[w] = (unsigned int)L.cand[k][0] | ((unsigned int)L.cand[k][1] << 16);
    list[base + k] = w;
  }
}

// contacts of this env for the current substep, in candidate order, truncated at MAXCON like the fused loop
DEV void gather_contacts(const DevModel* m, EnvLDS& L, const PipeBuffers& W, int e) {
  int lane = wave_lane();
  int info = W.ncand[e], ncand = info & 0xffff, ncon = 0;
  for (int k0 = 0; k0 < ncand; k0 += WAVE) {
    int k = k0 + lane;
    size_t w = (size_t)e * MAXCAND + k;
    bool valid = k < ncand && W.conres[w * 8 + 7] != 0.f;
    unsigned long long mask = wave_ballot(valid);
    int idx = ncon + wave_prefix(mask);
    if (valid && idx < MAXCON) {
      const float* r = W.conres + w * 8;
      unsigned int c = W.cand[w];
      float nrm[3] = {r[1], r[2], r[3]}, pos[3] = {r[4], r[5], r[6]};
      contact_init(m, L.con[idx], (int)(c & 0xffffu), (int)(c >> 16), r[0], nrm, pos);
    }
    ncon += __popcll(mask);
  }
  if (lane == 0) {
    L.ncand = ncand; L.narmcon = 0;
    L.overflow |= info >> 16;
    if (ncon > MAXCON) L.overflow |= 2;
    L.ncon = ncon > MAXCON ? MAXCON : ncon;
  }
  wave_sync();
}

__global__ void __launch_bounds__(64, 2) k_pipe_begin(const DevModel* m, StepParams P, DevBuffers B, PrepBuffers C, PipeBuffers W,
                                                   const float* action, float* obs, float* reward, float* discount,
                                                   unsigned char* step_type, unsigned char* need_reset, int* diag, int e0) {
  __shared__ EnvLDS L;
  int e = wave_uniform_i(W.order[e0 + blockIdx.x]), lane = wave_lane(), N = P.n_envs;
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
    if (lane == 0) { reward[e] = 0.f; discount[e] = 1.f; step_type[e] = 0; need_reset[e] = 0; W.active[e] = 0; W.ncand[e] = 0; }
    return;
  }
  load_state(L, B, e, N);
  // before_step: ctrl = action + homing offsets, unclamped (so100_task.py:266-287)
  if (lane < NU) { float c = action[(size_t)e * NU + lane] + P.action_offset[lane]; L.ctrl[lane] = c; B.ctrl[(size_t)lane * N + e] = c; }
  wave_sync();
  kinematics(m, L);
  broadphase(m, L);
  publish_candidates(L, W, e, N, 0);
  if (lane == 0) W.active[e] = 1;
}

// One wavefront per candidate pair.  No LDS; the two geoms (and the first 512 vertices of their hulls) live in
// registers.  Work items are taken NARROW_CHUNK at a time: one atomic and one dependent pair of loads per chunk
// instead of per item (that chain costs ~3 us, an MPR query on two boxes ~7 us).
#define NARROW_CHUNK 4
__global__ void __launch_bounds__(64, 2) k_narrow(const DevModel* m, int N, PipeBuffers W, int s) {
  int lane = wave_lane();
  int nwork = W.counters[2 * s];
  const unsigned int* list = W.work + (size_t)(s & 1) * W.work_cap;
  for (;;) {
    int i0 = 0;
    if (lane == 0) i0 = atomicAdd(&W.counters[2 * s + 1], NARROW_CHUNK);
    i0 = wave_uniform_i(i0);
    if (i0 >= nwork) break;
    unsigned int wl = 0, cl = 0;
    if (lane < NARROW_CHUNK && i0 + lane < nwork) { wl = list[i0 + lane]; cl = W.cand[wl]; }
    // not unrolled: four inlined copies of the MPR query are ~130 KB of code, more than the instruction cache holds
#pragma unroll 1
    for (int j = 0; j < NARROW_CHUNK; j++) {
      if (i0 + j >= nwork) break;
      unsigned long long t0 = wall_clock64();
      unsigned int w = (unsigned int)__builtin_amdgcn_readlane((int)wl, j), c = (unsigned int)__builtin_amdgcn_readlane((int)cl, j);
      int e = (int)(w / MAXCAND), g1 = (int)(c & 0xffffu), g2 = (int)(c >> 16);
      const float* pose = W.pose + (size_t)e * (NDYN * 12);
      int d1 = ldc(ldc(&m->geom_dyn) + g1), d2 = ldc(ldc(&m->geom_dyn) + g2);
      const float* p1 = pose + 12 * (d1 < 0 ? 0 : d1); const float* p2 = pose + 12 * (d2 < 0 ? 0 : d2);
      GeomW G1, G2;
      load_geom_at(m, g1, p1, p1 + 3, G1); load_geom_at(m, g2, p2, p2 + 3, G2);
      float dist, nrm[3], pos[3];
      bool ok = narrow_pair<HullCache>(m, G1, G2, &dist, nrm, pos);
      if (lane == 0) {
        float* r = W.conres + (size_t)w * 8;
        r[0] = dist; r[1] = nrm[0]; r[2] = nrm[1]; r[3] = nrm[2]; r[4] = pos[0]; r[5] = pos[1]; r[6] = pos[2]; r[7] = ok ? 1.f : 0.f;
        W.ticks[w] = (unsigned int)(wall_clock64() - t0);
      }
    }
  }
}

// Longest-processing-time-first launch order for k_pipe_solve.  A launch ends with its slowest env (Newton iteration
// counts: mean 2.7, max ~19) and workgroups are dispatched in index order, so envs that were expensive in the
// previous control step go first: counting sort of the group's envs by log2(cost), descending.  The order only
// changes WHEN an env is processed, never its result.  One workgroup for the whole batch.
__global__ void __launch_bounds__(1024) k_order(const unsigned int* cost, int* order, int e0, int ng) {   // e0 = 0, ng = N
  __shared__ int hist[32], start[32];
  int t = threadIdx.x;
  if (t < 32) hist[t] = 0;
  __syncthreads();
  for (int i = t; i < ng; i += 1024) {
    unsigned int c = cost[e0 + i];
    int b = c ? __clz((int)c) : 31;                      // large cost -> small bucket index (factor-of-two buckets)
    atomicAdd(&hist[b], 1);
  }
  __syncthreads();
  if (t == 0) { int acc = 0; for (int b = 0; b < 32; b++) { start[b] = acc; acc += hist[b]; } }
  __syncthreads();
  for (int i = t; i < ng; i += 1024) {
    unsigned int c = cost[e0 + i];
    int b = c ? __clz((int)c) : 31;
    order[e0 + atomicAdd(&start[b], 1)] = e0 + i;
  }
}

__global__ void __launch_bounds__(64, 2) k_pipe_solve(const DevModel* m, StepParams P, DevBuffers B, PipeBuffers W, int s, int last,
                                                   float* obs, float* reward, float* discount, unsigned char* step_type,
                                                   unsigned char* need_reset, int* diag, int e0) {
  __shared__ EnvLDS L;
  int e = wave_uniform_i(W.order[e0 + blockIdx.x]), lane = wave_lane(), N = P.n_envs;
  int act = W.active[e];
  if (act == 0) return;
  int sc = B.step_count[e] + 1;
  unsigned long long c0 = wall_clock64(), c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0;
  load_state(L, B, e, N);
  bool diverged = act == 2;
  if (!diverged) {
    forward_smooth(m, L);
    c1 = wall_clock64();
    gather_contacts(m, L, W, e);
    c2 = wall_clock64();
    make_constraints(m, L, P.solver == 0);
    c3 = wall_clock64();
    if (P.solver == 1) solve_newton(m, L, P.iterations, P.tolerance); else solve_pgs(m, L, P.iterations, P.tolerance);
    c4 = wall_clock64();
    forward_accelerations(L);
    if (lane == 0) { L.t_solve += (unsigned int)(c4 - c2); W.cost[e] = (unsigned int)(c4 - c2); }
    euler(m, L);
    diverged = check_divergence(L);
    if (diverged && lane == 0) W.active[e] = 2;
    c5 = wall_clock64();
  } else {
    if (lane == 0) { L.ncon = 0; L.nrow = 0; L.iters = 0; L.ncand = 0; L.overflow = 8; }
    wave_sync();
  }
  if (!last) {
    store_state(L, B, e, N);
    if (!diverged) {
      kinematics(m, L);
      broadphase(m, L);
      publish_candidates(L, W, e, N, s + 1);
    } else if (lane == 0) W.ncand[e] = 0;
    if (lane == 0) {
      unsigned int* st = W.stage + (size_t)e * 8;
      st[0] = (unsigned int)(c1 - c0); st[1] = (unsigned int)(c2 - c1); st[2] = (unsigned int)(c3 - c2); st[3] = (unsigned int)(c4 - c3);
      st[4] = (unsigned int)(c5 - c4); st[5] = (unsigned int)(wall_clock64() - c5); st[6] = (unsigned int)L.ncon; st[7] = (unsigned int)L.iters;
    }
    return;
  }
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
