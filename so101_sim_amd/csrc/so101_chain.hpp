// Per-env chaining of the pipelined control step (pipeline = 2): ONE persistent launch per control step in which
// every env advances through its substeps on its own dependencies, instead of 20 launch boundaries per chain at
// which every env waits for the slowest one of its slice.
//
// Work items:
//   narrow chunk   up to NARROW_CHUNK candidate pairs of one env (k_narrow's body): any wavefront
//   solve          one substep of one env (k_pipe_solve's body): any wavefront
// Dependencies of env e in substep s:   chunks(e, s) --all done--> solve(e, s) --publishes--> chunks(e, s + 1)
//   * solve(e, s) stores the state record, poses and candidates of e, sets pending[e] = s + 1 << 16 | number of
//     chunks, and pushes the chunks into the narrow queue of e's class;
//   * the wavefront that finishes a chunk stores its contact records and decrements pending[e]; whoever brings the
//     count to zero pushes solve(e, s + 1) into the solve queue of e's class (an env without candidates goes on
//     solving in the same wavefront);
//   * the last substep ends with finish_step() and adds one to the done counter; wavefronts leave when it reaches N.
// Classes: envs that were among the most expensive eighth of the previous control step (k_order) use their own queue
// pair, which every wavefront looks at first - their ten substeps are the critical path of the step.
//
// Queues: bounded multi-producer multi-consumer rings of 8-byte granules {item, ticket + 1}, fetch-add only.  A producer
// reserves tickets with one atomic add on the tail, announces them with one add on the `avail` count and writes the
// granules; a consumer takes an entitlement from `avail` (and gives it back if there was none), then a ticket from the
// head, and reads the granule of that ticket (its producer is at most two store instructions away).  Nobody waits for
// work that may never come: a wavefront that finds every `avail` count at zero goes round its loop (solve hi, narrow
// hi, solve lo, narrow lo, sleep), so the scheme cannot deadlock whatever the dispatch order or residency of the
// wavefronts is.  Tickets run on across control steps (32 bits, wrap-safe: a stale granule can only alias after 2^32
// pushes without its slot being rewritten); head == tail and avail == 0 at the end of every step.
//
// Memory discipline (wave.hpp, MI355X_MICROARCH.md "inter-workgroup visibility"): every byte handed from one wavefront
// to another inside the launch - state record, poses, candidates, contact records, flags, queue granules - is written
// with agent-scope (sc1, write-through) stores, drained with s_waitcnt vmcnt(0) before the atomic / granule that
// announces it, and read with agent-scope loads (L1 bypassed).  No cache-wide fences.
//
// Watchdog: a wavefront that has found no work for CHAIN_WATCHDOG_TICKS of the 100 MHz wall clock raises the abort flag;
// every wavefront leaves when it sees the flag.  A protocol error therefore ends the launch instead of hanging the GPU;
// the host reports it (so101_step returns SO101_ERR_STATE from the next call on, so101_get_info).
#pragma once
#include "so101_pipeline.hpp"

#ifdef SO101_EMU
#define CHAIN_WATCHDOG_TICKS 12000000000ull    // lane-thread emulation: 120 s
#else
#define CHAIN_WATCHDOG_TICKS 300000000ull      // 3 s at 100 MHz
#endif

struct ChainQ { unsigned int* ctl; unsigned long long* slot; unsigned int mask; };
#define QC_HEAD 0          // tickets handed to consumers
#define QC_AVAIL 16        // (int) items announced by producers minus items claimed by consumers
#define QC_TAIL 32         // tickets handed to producers          (each on its own 64-byte line)
DEV ChainQ chain_queue(const ChainQueues& C, int q) { ChainQ Q; Q.ctl = C.qctl + 64 * q; Q.slot = C.qslot[q]; Q.mask = C.qmask[q]; return Q; }
// the queue of env e's class: selects between two compile-time indices (a run-time index would put the struct on the stack)
DEV ChainQ chain_queue_of(const ChainQueues& C, int type, int e) {
  int hi = wave_uniform_i((int)C.cls[e]);
  ChainQ a = chain_queue(C, type), b = chain_queue(C, type + 1), Q;
  Q.ctl = hi ? b.ctl : a.ctl; Q.slot = hi ? b.slot : a.slot; Q.mask = hi ? b.mask : a.mask;
  return Q;
}

// One lane: take an item if one has been announced.  Fetch-adds only: an entitlement from the `avail` count (given back when
// there was none), then a ticket from the head - an entitled consumer's ticket always belongs to an item whose producer has
// at least reserved it, so the wait for the granule is bounded by that producer's two store instructions.  (The first
// version took the head with a compare-and-swap: correct, and 180 times slower than the launch chains at 4096 envs - a
// successful CAS needs the previous one's result, so the queue handed out one item per memory round trip.)
DEV bool q_pop_lane(const ChainQ& Q, unsigned int* item, unsigned int* abort_flag) {
  int old = (int)atom_add_agent(&Q.ctl[QC_AVAIL], 0xffffffffu);
  if (old <= 0) { atom_add_agent(&Q.ctl[QC_AVAIL], 1u); return false; }
  unsigned int h = atom_add_agent(&Q.ctl[QC_HEAD], 1u);
  unsigned long long t0 = 0;
  for (unsigned int spin = 0;; spin++) {
    unsigned long long v = ld_agent64(&Q.slot[h & Q.mask]);
    if ((unsigned int)(v >> 32) == h + 1u) { *item = (unsigned int)v; return true; }
    // watchdog on the wall clock (never seen on the GPU; ends the launch instead of hanging it)
    if ((spin & 1023u) == 1023u) {
      unsigned long long now = wall_clock64();
      if (t0 == 0) t0 = now;
      else if (now - t0 > CHAIN_WATCHDOG_TICKS) { atom_add_agent(abort_flag, 1u); return false; }
    }
  }
}
// one lane: push one item (the payload it announces has been drained by the caller)
DEV void q_push_lane(const ChainQ& Q, unsigned int item) {
  unsigned int t = atom_add_agent(&Q.ctl[QC_TAIL], 1u);
  atom_add_agent(&Q.ctl[QC_AVAIL], 1u);
  st_agent64(&Q.slot[t & Q.mask], (unsigned long long)item | ((unsigned long long)(t + 1u) << 32));
}

// narrow item: env << 8 | chunk index << 2 | pairs in the chunk - 1;  solve item: env | substep << 24
DEV unsigned int narrow_item(int e, int chunk, int cnt) { return ((unsigned int)e << 8) | ((unsigned int)chunk << 2) | (unsigned int)(cnt - 1); }
DEV unsigned int solve_item(int e, int s) { return (unsigned int)e | ((unsigned int)s << 24); }

// Hands the candidates in L.cand (and the poses) of env e to substep s through the queues.  Returns the number of
// chunks pushed; 0 = no candidate: the caller goes on with solve(e, s) itself.
DEV int publish_chain(const EnvLDS& L, const PipeBuffers& W, const ChainQueues& C, int e, int s) {
  int lane = wave_lane(), ncand = L.ncand;
  for (int i = lane; i < NDYN * 12; i += WAVE) {
    int b = i / 12, j = i % 12;
    st_agent(&W.pose[(size_t)e * (NDYN * 12) + i], j < 3 ? L.xpos[b][j] : L.xmat[b][j - 3]);
  }
  for (int k = lane; k < ncand; k += WAVE)
    st_agent(&W.cand[(size_t)e * MAXCAND + k], (unsigned int)L.cand[k][0] | ((unsigned int)L.cand[k][1] << 16));
  int nch = (ncand + NARROW_CHUNK - 1) / NARROW_CHUNK;
  if (lane == 0) {
    st_agent(&W.ncand[e], ncand | ((L.overflow & 1) << 16));
    st_agent(&C.pending[e], ((unsigned int)s << 16) | (unsigned int)nch);
  }
  drain_stores();                                 // payload and pending count are at the coherence point before any chunk can be seen
  if (nch == 0) return 0;                         // (the caller goes on with this env itself: its loads follow the drain)
  ChainQ Q = chain_queue_of(C, Q_NARROW, e);
  unsigned int base = 0;
  if (lane == 0) { base = atom_add_agent(&Q.ctl[QC_TAIL], (unsigned int)nch); atom_add_agent(&Q.ctl[QC_AVAIL], (unsigned int)nch); }
  base = (unsigned int)wave_uniform_i((int)base);
  for (int k = lane; k < nch; k += WAVE) {
    int cnt = ncand - NARROW_CHUNK * k; cnt = cnt < NARROW_CHUNK ? cnt : NARROW_CHUNK;
    unsigned int t = base + (unsigned int)k;
    st_agent64(&Q.slot[t & Q.mask], (unsigned long long)narrow_item(e, k, cnt) | ((unsigned long long)(t + 1u) << 32));
  }
  return nch;
}

// k_narrow's body for one chunk: the pairs' contact records, then the env's pending count; the wavefront that brings it
// to zero hands the env to the solve queue.
// CHAINED = false (merged launches, pipeline = 3): the records are read by the NEXT launch - plain stores, no pending count.
template <bool CHAINED>
DEV void narrow_chunk(const DevModel* m, const PipeBuffers& W, const ChainQueues* Cp, unsigned int item) {
  int lane = wave_lane();
  unsigned long long ta = CHAINED ? wall_clock64() : 0ull;
  int e = (int)(item >> 8), k0 = (int)((item >> 2) & 63u) * NARROW_CHUNK, cnt = (int)(item & 3u) + 1;
  unsigned int cl = 0;
  if (lane < cnt) cl = ld_agent(&W.cand[(size_t)e * MAXCAND + k0 + lane]);
  // the env's body poses: 96 floats over the lanes (two loads), handed to the geoms below with v_readlane
  const float* pose = W.pose + (size_t)e * (NDYN * 12);
  float pv0 = ld_agent(&pose[lane]), pv1 = lane < NDYN * 12 - WAVE ? ld_agent(&pose[WAVE + lane]) : 0.f;
  unsigned int c_first = (unsigned int)__builtin_amdgcn_readlane((int)cl, 0);     // (waits for the loads)
  unsigned long long tb = CHAINED ? wall_clock64() + (c_first & 0u) : 0ull;
#pragma unroll 1
  for (int j = 0; j < cnt; j++) {
    unsigned int c = (unsigned int)__builtin_amdgcn_readlane((int)cl, j);
    int g1 = (int)(c & 0xffffu), g2 = (int)(c >> 16);
    int d1 = ldc(ldc(&m->geom_dyn) + g1), d2 = ldc(ldc(&m->geom_dyn) + g2);
    float p1[12], p2[12];
    int o1 = 12 * (d1 < 0 ? 0 : d1), o2 = 12 * (d2 < 0 ? 0 : d2);
#pragma unroll
    for (int i = 0; i < 12; i++) {
      int i1 = o1 + i, i2 = o2 + i;
      p1[i] = i1 < WAVE ? wave_get_f(pv0, i1 & 63) : wave_get_f(pv1, i1 & 63);
      p2[i] = i2 < WAVE ? wave_get_f(pv0, i2 & 63) : wave_get_f(pv1, i2 & 63);
    }
    GeomW G1, G2;
    load_geom_at(m, g1, p1, p1 + 3, G1); load_geom_at(m, g2, p2, p2 + 3, G2);
    PairContacts pc;
    narrow_pair<HullCache>(m, G1, G2, g1, g2, pc);
    // record: count, normal, then the valid slots compactly in slot order - lane i stores word i
    float rec[CONRES_DIM];
    rec[0] = (float)__popc(pc.valid); rec[1] = pc.nrm[0]; rec[2] = pc.nrm[1]; rec[3] = pc.nrm[2];
#pragma unroll
    for (int i = 4; i < CONRES_DIM; i++) rec[i] = 0.f;
    int o = 4;
#pragma unroll
    for (int q = 0; q < NCPP; q++)
      if ((pc.valid >> q) & 1u) {
#pragma unroll
        for (int t = 0; t < NCPP; t++)            // (o = 4 + 4 t for some t <= q: selects instead of a dynamic register index)
          if (o == 4 + 4 * t) { rec[4 + 4 * t] = pc.dist[q]; rec[5 + 4 * t] = pc.pos[q][0]; rec[6 + 4 * t] = pc.pos[q][1]; rec[7 + 4 * t] = pc.pos[q][2]; }
        o += 4;
      }
    float mine = 0.f;
#pragma unroll
    for (int i = 0; i < CONRES_DIM; i++) mine = lane == i ? rec[i] : mine;
    if (lane < CONRES_DIM) pst<CHAINED>(&W.conres[((size_t)e * MAXCAND + k0 + j) * CONRES_DIM + lane], mine);
  }
  if constexpr (!CHAINED) return;
  const ChainQueues& C = *Cp;
  unsigned long long tc = wall_clock64();
  drain_stores();
  ChainQ QS = chain_queue_of(C, Q_SOLVE, e);
  unsigned int old = 0;
  if (lane == 0) {
    old = atom_add_agent(&C.pending[e], 0xffffffffu);           // -1
    if ((old & 0xffffu) == 1u) q_push_lane(QS, solve_item(e, (int)(old >> 16)));
  }
  old = (unsigned int)wave_uniform_i((int)old);
  unsigned long long td = wall_clock64() + (old & 0u);
  if (lane == 0) { atomicAdd(&C.stats[9], tb - ta); atomicAdd(&C.stats[10], tc - tb); atomicAdd(&C.stats[11], td - tc); }
}

DEV void chain_narrow(const DevModel* m, const PipeBuffers& W, const ChainQueues& C, unsigned int item) { narrow_chunk<true>(m, W, &C, item); }

// ---- merged launches (pipeline = 3) -------------------------------------------------------------------------------------------
// The launch chains of pipeline = 1 with the narrowphase folded into the solve launches: the wavefront that has solved
// substep s of its env publishes the env's candidates as chunks into the CHAIN's queue and then, instead of leaving, pulls
// chunks of substep s + 1 (anybody's) until every env of the launch has published and the queue is empty.  The contact
// records are consumed by the next launch, so only the solve -> narrowphase hand-off (poses, candidates, queue granules)
// happens inside a launch (agent-scope accesses, wave.hpp).  Narrowphase work fills the tail of the solves (a launch lasts
// as long as its slowest Newton solve) and half of the launch boundaries disappear; the wavefronts of a launch still start
// together and run the same code at about the same time, which the per-env chained kernel (k_chain) does not - its
// desynchronised wavefronts refetch the 240 KB of code from L2 4.3 times as often and stall on it (profiles/README.md).
// At most MERGED_LINGER wavefronts stay to poll for the chunks of the last solves; the others leave when they find the
// queue empty, which frees their slots for the other chains' launches.
#define MERGED_LINGER 48
DEV ChainQ merged_queue(const PipeBuffers& W) { ChainQ Q; Q.ctl = W.mq_ctl; Q.slot = W.mq_slot; Q.mask = W.mq_mask; return Q; }

// candidates of env e (in L) -> chunks in the chain's queue; poses and candidates with agent-scope stores (read by chunk
// workers of this launch) - the next launch's solve reads the same bytes with plain loads after the launch boundary
DEV void publish_merged(const EnvLDS& L, const PipeBuffers& W, int e) {
  int lane = wave_lane(), ncand = L.ncand;
  for (int i = lane; i < NDYN * 12; i += WAVE) {
    int b = i / 12, j = i % 12;
    st_agent(&W.pose[(size_t)e * (NDYN * 12) + i], j < 3 ? L.xpos[b][j] : L.xmat[b][j - 3]);
  }
  for (int k = lane; k < ncand; k += WAVE)
    st_agent(&W.cand[(size_t)e * MAXCAND + k], (unsigned int)L.cand[k][0] | ((unsigned int)L.cand[k][1] << 16));
  int nch = (ncand + NARROW_CHUNK - 1) / NARROW_CHUNK;
  if (lane == 0) W.ncand[e] = ncand | ((L.overflow & 1) << 16);
  if (nch == 0) return;
  drain_stores();
  ChainQ Q = merged_queue(W);
  unsigned int base = 0;
  if (lane == 0) { base = atom_add_agent(&Q.ctl[QC_TAIL], (unsigned int)nch); atom_add_agent(&Q.ctl[QC_AVAIL], (unsigned int)nch); }
  base = (unsigned int)wave_uniform_i((int)base);
  for (int k = lane; k < nch; k += WAVE) {
    int cnt = ncand - NARROW_CHUNK * k; cnt = cnt < NARROW_CHUNK ? cnt : NARROW_CHUNK;
    unsigned int t = base + (unsigned int)k;
    st_agent64(&Q.slot[t & Q.mask], (unsigned long long)narrow_item(e, k, cnt) | ((unsigned long long)(t + 1u) << 32));
  }
}

// this wavefront has passed its publish point in launch `launch` of the chain (whether or not it had anything to publish)
DEV void merged_published(const PipeBuffers& W, int launch) {
  drain_stores();
  if (wave_lane() == 0) atom_add_agent(&W.mq_pub[launch], 1u);
}

// pull chunks until every env of the launch has published and the queue is empty
DEV void merged_helper(const DevModel* m, const PipeBuffers& W, int launch, int ng) {
  int lane = wave_lane();
  ChainQ Q = merged_queue(W);
  bool lingering = false;
  unsigned long long t0 = wall_clock64();
  for (;;) {
    unsigned int item = 0; int got = 0, all = 0;
    if (lane == 0) {
      all = ld_agent(&W.mq_pub[launch]) >= (unsigned int)ng ? 1 : 0;       // read BEFORE the count: no push can follow a complete publish count
      if ((int)ld_agent(&Q.ctl[QC_AVAIL]) > 0 && q_pop_lane(Q, &item, W.mq_pub + 127)) got = 1;
    }
    got = wave_uniform_i(got);
    if (got) { narrow_chunk<false>(m, W, nullptr, (unsigned int)wave_uniform_i((int)item)); continue; }
    int stop = wave_uniform_i(all);
    if (!stop && !lingering) {
      int over = 0;
      if (lane == 0) over = atom_add_agent(&W.mq_pub[64 + launch], 1u) >= (unsigned int)MERGED_LINGER ? 1 : 0;
      stop = wave_uniform_i(over);
      lingering = true;
    }
    if (!stop && wall_clock64() - t0 > CHAIN_WATCHDOG_TICKS) stop = 1;      // (never seen: a launch cannot outlive its solves by seconds)
    if (stop) break;
    idle_sleep();
  }
}

// k_pipe_solve's body for env e from substep s on: as long as a substep yields no candidate the same wavefront goes on
DEV void chain_solve(const DevModel* m, EnvLDS& L, const StepParams& P, const DevBuffers& B, const EventBuffers& E, const PipeBuffers& W,
                     const ChainQueues& C, unsigned int item, const SolveIO& io) {
  int e = (int)(item & 0xffffffu), s = (int)(item >> 24);
  for (;;) {
    int last = s == P.n_substeps - 1;
    int act = (int)ld_agent8(&W.active[e]);
    unsigned long long ts[4];
    bool more = pipe_solve_env<true>(m, L, P, B, E, W, e, s, last, act, io, ts);
    if (wave_lane() == 0) { atomicAdd(&C.stats[12], ts[1] - ts[0]); atomicAdd(&C.stats[13], ts[2] - ts[1]); atomicAdd(&C.stats[14], ts[3] - ts[2]); }
    unsigned long long tp = wall_clock64();
    if (last) {
      drain_stores();
      if (wave_lane() == 0) atom_add_agent(&C.chain_ctl[0], 1u);
      return;
    }
    if (more) {
      int nch = publish_chain(L, W, C, e, s + 1);
      if (wave_lane() == 0) atomicAdd(&C.stats[15], wall_clock64() - tp);
      if (nch > 0) return;
    }
    else drain_stores();        // diverged: state record and ncand = 0 stored; the env idles through its remaining substeps here
    wave_sync();
    s++;
  }
}
