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

// Memory policy AG (all functions below): false = the data was produced by an EARLIER launch (launch chains, pipeline = 1);
// true = it was produced by another wavefront of THIS launch (per-env chaining, pipeline = 2, so101_chain.hpp): every
// store and load of handed-off bytes is then an agent-scope (sc1) access, see wave.hpp.
template <bool AG, class T> DEV T pld(const T* p) { if constexpr (AG) return ld_agent(p); else return *p; }
template <bool AG, class T> DEV void pst(T* p, T v) { if constexpr (AG) st_agent(p, v); else *p = v; }

// the env's integrated state as one contiguous 256-byte record (substep round trips of the pipelined step)
template <bool AG = false>
DEV void store_state_aos(const EnvLDS& L, const PipeBuffers& W, int e) {
  int lane = wave_lane();
  float v = 0.f;
  if (lane < NQ) v = L.qpos[lane];
  else if (lane < NQ + NV) v = L.qvel[lane - NQ];
  else if (lane < NQ + 2 * NV) v = L.warm[lane - NQ - NV];
  else if (lane < NQ + 2 * NV + NU) v = L.ctrl[lane - NQ - 2 * NV];
  pst<AG>(&W.state[(size_t)e * STATE_AOS + lane], v);
}
template <bool AG = false>
DEV void load_state_aos(EnvLDS& L, const DevBuffers& B, const PipeBuffers& W, int e, int N) {
  int lane = wave_lane();
  load_env_constants(L, B, e, N);
  float v = pld<AG>(&W.state[(size_t)e * STATE_AOS + lane]);
  if (lane < NQ) L.qpos[lane] = v;
  else if (lane < NQ + NV) L.qvel[lane - NQ] = v;
  else if (lane < NQ + 2 * NV) L.warm[lane - NQ - NV] = v;
  else if (lane < NQ + 2 * NV + NU) L.ctrl[lane - NQ - 2 * NV] = v;
  if (lane == 0) { L.overflow = 0; L.t_collision = 0; L.t_solve = 0; L.t_begin = (unsigned int)SO101_CLOCK(); }
  wave_sync();
}

// Hands the candidates in L.cand to the narrowphase launch of substep s as self-contained WORK ITEMS (round 5).
//
// Until round 4 the work list held one word per candidate and k_narrow followed it: list -> W.cand -> geom_dyn -> the geom tables and the
// body pose -> the hull.  Every arrow is a dependent memory round trip (0.4 - 0.8 us each on this machine), five to seven of them per
// candidate pair in front of 6 - 11 us of work; the counters said the wavefronts of k_narrow spent half their cycles parked on them.  Now the
// wavefront that HAS the env's poses in LDS (k_pipe_begin / k_pipe_solve, right after the broadphase) writes, lane = candidate, everything a
// pair's query needs into ONE 192-byte item - the record position of the result, and both geoms in world coordinates (type, hull address and
// size, sizes, orientation, position, centre, bounding radius: load_geom_at(), the same expressions on the same numbers as the fused path) -
// and k_narrow reads its items with one coalesced load per item: atomic -> items -> hulls -> work.
//
// Two orders.  RECORDS: the contact record of candidate k of env e lives at position cbase[e] + k of the slice's pool (contiguous per env, in
// candidate order: gather_contacts reads them like that).  ITEMS: longest-processing-time first (round 4) - pairs without a box or the plane
// (hull against hull / capsule / cylinder: MPR + EPA, 13-48 us) fill the item array from its front, the others (a flat face against a hull:
// closed form, 6-11 us) from its end, and k_narrow walks front first; a launch of persistent wavefronts then ends on cheap items instead of
// on an EPA chunk taken last.  The order only changes WHEN a pair is processed, never its result.  The item array has the capacity of the
// record pool (W.conres_cap): a candidate without a record has no item either (dropped from the solve, counted as a candidate overflow).
DEV bool heavy_pair(const DevModel* m, int g1, int g2) {
  int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
  return t1 != G_PLANE && t1 != G_BOX && t2 != G_PLANE && t2 != G_BOX;
}
// one geom of an item (one lane): world-frame GeomW by load_geom_at() with per-lane loads (policy G16: plain loads, no uniformity claims)
DEV void item_put_geom(const DevModel* m, const EnvLDS& L, int g, unsigned int* it, GeomW& G) {
  int d = m->geom_dyn[g];
  load_geom_at<G16>(m, g, L.xpos[d < 0 ? 0 : d], L.xmat[d < 0 ? 0 : d], G);
  it[ITEM_G_TYPE] = (unsigned int)G.type; it[ITEM_G_VADR] = (unsigned int)G.vadr; it[ITEM_G_VNUM] = (unsigned int)G.vnum;
#pragma unroll
  for (int i = 0; i < 3; i++) { it[ITEM_G_SIZE + i] = __float_as_uint(G.size[i]); it[ITEM_G_P + i] = __float_as_uint(G.p[i]); it[ITEM_G_C + i] = __float_as_uint(G.c[i]); }
#pragma unroll
  for (int i = 0; i < 9; i++) it[ITEM_G_R + i] = __float_as_uint(G.R[i]);
  it[ITEM_G_RBOUND] = __float_as_uint(m->geom_rbound[g]);
}
DEV void publish_candidates(const DevModel* m, const EnvLDS& L, const PipeBuffers& W, int e, int N, int s) {
  int lane = wave_lane(), ncand = L.ncand;
  for (int i = lane; i < NDYN * 12; i += WAVE) {          // (the poses: k_pipe_solve's kinematics_from_pose reads them back)
    int b = i / 12, j = i % 12;
    W.pose[(size_t)e * (NDYN * 12) + i] = j < 3 ? L.xpos[b][j] : L.xmat[b][j - 3];
  }
  int base = 0;
  if (lane == 0) base = ncand ? atomicAdd(&W.counters[2 * s], ncand) : 0;
  base = wave_bcast_i(base, 0);
  // the contact records of this substep live at the candidates' record positions: what does not fit the slice's pool is dropped from
  // the solve (no item is written for it) and counted as a candidate overflow
  int room = (int)W.conres_cap - base, keep = ncand < room ? ncand : (room > 0 ? room : 0);
  int nheavy = 0;
  for (int k0 = 0; k0 < keep; k0 += WAVE) { int k = k0 + lane; nheavy += __popcll(wave_ballot(k < keep && heavy_pair(m, L.cand[k][0], L.cand[k][1]))); }
  int hb = 0, lb = 0;
  if (lane == 0) {
    W.cbase[e] = base;
    W.ncand[e] = keep | (((L.overflow & 1) | (keep < ncand ? 1 : 0)) << 16);
    hb = nheavy ? atomicAdd(&W.counters[2 * MAXSUB + 2 * s], nheavy) : 0;
    lb = keep - nheavy ? atomicAdd(&W.counters[2 * MAXSUB + 2 * s + 1], keep - nheavy) : 0;
  }
  hb = wave_bcast_i(hb, 0); lb = wave_bcast_i(lb, 0);
  int hseen = 0, lseen = 0;
  for (int k0 = 0; k0 < keep; k0 += WAVE) {
    int k = k0 + lane;
    bool in = k < keep;
    int g1 = L.cand[in ? k : 0][0], g2 = L.cand[in ? k : 0][1];
    bool heavy = in && heavy_pair(m, g1, g2);
    unsigned long long mh = wave_ballot(heavy), ml = wave_ballot(in && !heavy);
    if (in) {
      unsigned int w = (unsigned int)e * MAXCAND + k;
      W.cand[w] = (unsigned int)g1 | ((unsigned int)g2 << 16);
      unsigned int pos = heavy ? (unsigned int)(hb + hseen + wave_prefix(mh)) : W.conres_cap - 1u - (unsigned int)(lb + lseen + wave_prefix(ml));
      unsigned int it[ITEM_WORDS];
      it[0] = (unsigned int)(base + k); it[1] = w;
      GeomW G1, G2;
      item_put_geom(m, L, g1, it + ITEM_GEOM0, G1); item_put_geom(m, L, g2, it + ITEM_GEOM1, G2);
      // round 6: a flat face against a hull - the cell of the hull's support-vertex lists that the query's first support direction falls into, so
      // that k_narrow can fetch those few vertices right behind the item instead of staging the whole hull (so101_device.hpp HullSub); word 46 = first
      // entry, word 47 = entries (1 .. HL_MAX) | cell << 8, or 0: none
      it[46] = 0u; it[47] = 0u;
      if (m->hl_off) {
        int cell = light_first_cell(G1, G2);
        if (cell >= 0) {
          const unsigned int* o = m->hl_off + (size_t)g2 * (HL_CELLS + 1) + cell;
          unsigned int a = o[0], cnt = o[1] - a;
          if (cnt >= 1u && cnt <= (unsigned int)HL_MAX) { it[46] = a; it[47] = cnt | ((unsigned int)cell << 8); }
        }
      }
      uint4* dst = (uint4*)(W.items + (size_t)pos * ITEM_WORDS);
#pragma unroll
      for (int q = 0; q < ITEM_WORDS / 4; q++) dst[q] = make_uint4(it[4 * q], it[4 * q + 1], it[4 * q + 2], it[4 * q + 3]);
    }
    hseen += __popcll(mh); lseen += __popcll(ml);
  }
}

// contacts of this env for the current substep, in candidate order (a pair's contacts stay together, in the order the
// narrowphase produced them), truncated at MAXCON like the fused loop.  lane = candidate.
template <bool AG = false>
DEV void gather_contacts(const DevModel* m, EnvLDS& L, const PipeBuffers& W, int e) {
  int lane = wave_lane();
  int info = pld<AG>(&W.ncand[e]), ncand = info & 0xffff, ncon = 0;
  // more contacts than MAXCON: one contact per pair (the first of its patch) for this substep, as collision() does in the fused step
  int total = 0;
  for (int k0 = 0; k0 < ncand; k0 += WAVE) {
    int k = k0 + lane;
    size_t w = (size_t)e * MAXCAND + k;
    const float* r = W.conres + (W.conres_cap ? (size_t)(W.cbase[e] + k) : w) * CONRES_DIM;
    int cnt = k < ncand ? (int)pld<AG>(&r[0]) : 0;
#pragma unroll
    for (int b = 0; b < 3; b++) total += __popcll(wave_ballot((cnt >> b) & 1)) << b;
  }
  const bool reduced = total > MAXCON;
  for (int k0 = 0; k0 < ncand; k0 += WAVE) {
    int k = k0 + lane;
    size_t w = (size_t)e * MAXCAND + k;
    const float* r = W.conres + (W.conres_cap ? (size_t)(W.cbase[e] + k) : w) * CONRES_DIM;
    int cnt = k < ncand ? (int)pld<AG>(&r[0]) : 0;
    if (reduced && cnt > 1) cnt = 1;
    // exclusive prefix of the per-candidate contact counts (at most NCPP each): one ballot per possible count bit
    int idx = ncon, total = 0;
#pragma unroll
    for (int b = 0; b < 3; b++) {
      unsigned long long mask = wave_ballot((cnt >> b) & 1);
      idx += wave_prefix(mask) << b;
      total += __popcll(mask) << b;
    }
    if (cnt > 0) {
      unsigned int c = pld<AG>(&W.cand[w]);
      float nrm[3] = {pld<AG>(&r[1]), pld<AG>(&r[2]), pld<AG>(&r[3])};
      for (int j = 0; j < cnt; j++) {
        if (idx + j < MAXCON) {
          float pos[3] = {pld<AG>(&r[5 + 4 * j]), pld<AG>(&r[6 + 4 * j]), pld<AG>(&r[7 + 4 * j])};
          contact_init(m, L.con[idx + j], (int)(c & 0xffffu), (int)(c >> 16), pld<AG>(&r[4 + 4 * j]), nrm, pos);
        }
      }
    }
    ncon += total;
  }
  if (lane == 0) {
    L.ncand = ncand; L.narmcon = 0;
    L.overflow |= info >> 16;
    if (reduced) L.overflow |= 128;
    if (ncon > MAXCON) L.overflow |= 2;
    L.ncon = ncon > MAXCON ? MAXCON : ncon;
  }
  wave_sync();
}

// One substep of one env on the solve side of the pipeline: smooth dynamics from the published poses, contact gather,
// constraint rows, Newton, Euler step; then, unless this was the last substep, the integrated state goes back to its
// record and the next substep's broadphase runs (candidates left in L.cand for the caller to publish); the last substep
// ends with the task logic (finish_step).  Returns true when candidates for substep s + 1 are in L (false: last
// substep, or the env has diverged - then W.ncand[e] = 0 has been stored).  Shared by k_pipe_solve and k_chain.
template <bool AG>
DEV bool pipe_solve_env(const DevModel* m, EnvLDS& L, const StepParams& P, const DevBuffers& B, const EventBuffers& E, const PipeBuffers& W,
                        int e, int s, int last, int act, const SolveIO& io, unsigned long long* ts = nullptr /* [4] wall-clock stamps */) {
  int lane = wave_lane(), N = P.n_envs;
  if (ts) ts[0] = ts[1] = ts[2] = ts[3] = wall_clock64();
  int sc = B.step_count[e] + 1;
  unsigned long long c0 = SO101_CLOCK(), c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0;
  load_state_aos<AG>(L, B, W, e, N);
  bool diverged = act == 2;
  if (!diverged) {
    // (the poses of this state were published for the narrowphase by the kernel / wavefront that produced it)
    kinematics_from_pose<AG>(m, L, W.pose + (size_t)e * (NDYN * 12));
    crba_arm(m, L);
    smooth_dynamics(m, L);
    c1 = SO101_CLOCK();
    gather_contacts<AG>(m, L, W, e);
    c2 = SO101_CLOCK();
    if (ts) ts[1] = wall_clock64();
    unsigned long long t_solve0 = wall_clock64();            // scheduling hint of k_order: always measured
    make_constraints(m, L);
    c3 = SO101_CLOCK();
    solve_newton(m, L, P.iterations, P.tolerance);
    c4 = SO101_CLOCK();
    forward_accelerations(L);
    if (lane == 0) { unsigned int dt = (unsigned int)(wall_clock64() - t_solve0); L.t_solve += dt; W.cost[e] = dt; }
    euler(m, L);
    diverged = check_divergence(L);
    if (diverged && lane == 0) { if constexpr (AG) st_agent8(&W.active[e], (unsigned char)2); else W.active[e] = 2; }
    c5 = SO101_CLOCK();
    if (ts) ts[2] = wall_clock64();
  } else {
    if (lane == 0) { L.ncon = 0; L.nrow = 0; L.iters = 0; L.ncand = 0; L.overflow = 8; }
    wave_sync();
  }
  int ncon_solved = L.ncon, iters_solved = L.iters;      // (the next broadphase clears the counts)
  if (!last) {
    store_state_aos<AG>(L, W, e);
    if (!diverged) {
      unsigned long long q0 = SO101_CLOCK();
      kinematics(m, L);
      unsigned long long q1 = SO101_CLOCK();
      broadphase(m, L);
#ifdef SO101_DEBUG_CLOCKS
      if (lane == 0) { L.nw.prof[8] = (unsigned int)(q0 - c5); L.nw.prof[9] = (unsigned int)(q1 - q0); }
#endif
    } else if (lane == 0) pst<AG>(&W.ncand[e], 0);
    if (lane == 0) {
      if (L.overflow) pst<AG>(&E.flags[e], pld<AG>(&E.flags[e]) | L.overflow);      // rare; summed into the event counters by finish_step()
#ifdef SO101_DEBUG_CLOCKS
      for (int k = 0; k < 16; k++) W.ticks[(size_t)e * MAXCAND + 240 + k] = L.nw.prof[k];      // (solver / broadphase phases; slots of candidates 240+ are idle)
#endif
      if (SO101_CLOCKS_ON) {
        unsigned int* st = W.stage + (size_t)e * 8;
        st[0] = (unsigned int)(c1 - c0); st[1] = (unsigned int)(c2 - c1); st[2] = (unsigned int)(c3 - c2); st[3] = (unsigned int)(c4 - c3);
        st[4] = (unsigned int)(c5 - c4); st[5] = (unsigned int)(SO101_CLOCK() - c5); st[6] = (unsigned int)ncon_solved; st[7] = (unsigned int)iters_solved;
      }
    }
    if (ts) ts[3] = wall_clock64();
    return !diverged;
  }
  if (ts) ts[3] = wall_clock64();
  finish_step<AG>(m, L, P, B, e, sc, diverged, io.obs, io.reward, io.discount, io.step_type, io.need_reset, io.diag, E);
  return false;
}
