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

#include "so101_env.hpp"

// the env's integrated state as one contiguous 256-byte record (substep round trips of the pipelined step)
DEV void store_state_aos(const EnvLDS& L, const PipeBuffers& W, int e) {
  int lane = wave_lane();
  float v = 0.f;
  if (lane < NQ) v = L.qpos[lane];
  else if (lane < NQ + NV) v = L.qvel[lane - NQ];
  else if (lane < NQ + 2 * NV) v = L.warm[lane - NQ - NV];
  else if (lane < NQ + 2 * NV + NU) v = L.ctrl[lane - NQ - 2 * NV];
  W.state[(size_t)e * STATE_AOS + lane] = v;
}
DEV void load_state_aos(EnvLDS& L, const DevBuffers& B, const PipeBuffers& W, int e, int N) {
  int lane = wave_lane();
  load_env_constants(L, B, e, N);
  float v = W.state[(size_t)e * STATE_AOS + lane];
  if (lane < NQ) L.qpos[lane] = v;
  else if (lane < NQ + NV) L.qvel[lane - NQ] = v;
  else if (lane < NQ + 2 * NV) L.warm[lane - NQ - NV] = v;
  else if (lane < NQ + 2 * NV + NU) L.ctrl[lane - NQ - 2 * NV] = v;
  if (lane == 0) { L.overflow = 0; L.t_collision = 0; L.t_solve = 0; L.t_begin = (unsigned int)SO101_CLOCK(); }
  wave_sync();
}

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

// contacts of this env for the current substep, in candidate order (a pair's contacts stay together, in the order the
// narrowphase produced them), truncated at MAXCON like the fused loop.  lane = candidate.
DEV void gather_contacts(const DevModel* m, EnvLDS& L, const PipeBuffers& W, int e) {
  int lane = wave_lane();
  int info = W.ncand[e], ncand = info & 0xffff, ncon = 0;
  for (int k0 = 0; k0 < ncand; k0 += WAVE) {
    int k = k0 + lane;
    size_t w = (size_t)e * MAXCAND + k;
    const float* r = W.conres + w * CONRES_DIM;
    int cnt = k < ncand ? (int)r[0] : 0;
    // exclusive prefix of the per-candidate contact counts (at most NCPP each): one ballot per possible count bit
    int idx = ncon, total = 0;
#pragma unroll
    for (int b = 0; b < 3; b++) {
      unsigned long long mask = wave_ballot((cnt >> b) & 1);
      idx += wave_prefix(mask) << b;
      total += __popcll(mask) << b;
    }
    if (cnt > 0) {
      unsigned int c = W.cand[w];
      float nrm[3] = {r[1], r[2], r[3]};
      for (int j = 0; j < cnt; j++) {
        if (idx + j < MAXCON) {
          float pos[3] = {r[5 + 4 * j], r[6 + 4 * j], r[7 + 4 * j]};
          contact_init(m, L.con[idx + j], (int)(c & 0xffffu), (int)(c >> 16), r[4 + 4 * j], nrm, pos);
        }
      }
    }
    ncon += total;
  }
  if (lane == 0) {
    L.ncand = ncand; L.narmcon = 0;
    L.overflow |= info >> 16;
    if (ncon > MAXCON) L.overflow |= 2;
    L.ncon = ncon > MAXCON ? MAXCON : ncon;
  }
  wave_sync();
}
