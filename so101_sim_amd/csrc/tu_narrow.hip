// Translation unit: the narrowphase kernel of the pipelined control step (see so101_pipeline.hpp).
#include "so101_pipeline.hpp"
#include "so101_launch.hpp"

// One wavefront per candidate pair (policy G64 of so101_device.hpp), persistent wavefronts pulling chunks of work items.
//
// Round 5: memory round trips per CHUNK instead of five to seven per pair.  A chunk is 2 (heavy region) or 4 (light region) self-contained
// items (so101_pipeline.hpp: both geoms in world coordinates, hull addresses, the record position); a wavefront takes
//   1. the chunk (one atomic),
//   2. its items, one coalesced 192-byte load each, lane = word (the fields are handed out with v_readlane: wave-uniform values in SGPRs
//      instead of 64 copies in a VGPR each),
//   3. EVERY hull of the chunk into LDS at once (up to HULL_POOL vertex slots of the workgroup's 20 KB share; loads of all hulls in flight
//      together), and then works through the pairs without touching memory again until the records are written.  A hull that did not fit
//      is staged when its pair's turn comes (over the slots of the pairs already done).
// Support scans read the staged hulls with ds_read_b128 (so101_device.hpp, HullLDS); the register hull cache of rounds 2-4 (48 VGPRs, 43
// more spilled to scratch: 14 MB written per launch) is gone from this kernel, the kernel has no scratch.
//
// Measured alternative (kept as policy G16, bit-identical results): one pair per DPP row of 16 lanes, four pairs per
// wavefront.  It is SLOWER (4096-env bench 490-509 k against 635 k env-steps/s): ~70 % of a query's instructions are
// the lane-parallel hull scans, not the uniform portal math.
#ifndef NARROW_WAVES
#define NARROW_WAVES 2
#endif
#ifndef HULL_POOL
#define HULL_POOL 1536           // vertex slots of the LDS hull pool (18 KB): three 512-slot or six 256-slot hulls
#endif

DEV float item_f(unsigned int word, int i) { return __uint_as_float((unsigned int)__builtin_amdgcn_readlane((int)word, i)); }
DEV int item_i(unsigned int word, int i) { return __builtin_amdgcn_readlane((int)word, i); }
DEV void item_geom(unsigned int word, int o, GeomW& G, float& rbound) {
  G.type = item_i(word, o + ITEM_G_TYPE); G.vadr = item_i(word, o + ITEM_G_VADR); G.vnum = item_i(word, o + ITEM_G_VNUM);
#pragma unroll
  for (int i = 0; i < 3; i++) { G.size[i] = item_f(word, o + ITEM_G_SIZE + i); G.p[i] = item_f(word, o + ITEM_G_P + i); G.c[i] = item_f(word, o + ITEM_G_C + i); }
#pragma unroll
  for (int i = 0; i < 9; i++) G.R[i] = item_f(word, o + ITEM_G_R + i);
  rbound = item_f(word, o + ITEM_G_RBOUND);
}

__global__ void __launch_bounds__(64, NARROW_WAVES) k_narrow(const DevModel* m, int N, PipeBuffers W, int s) {
  __shared__ __attribute__((aligned(16))) float pool[3 * HULL_POOL];
  int lane = wave_lane();
  // Scalar loads on purpose.  The counts share their cache line with the cursor every wave of this launch does atomics on; when the
  // compiler picked a plain vector load here the whole launch ran 26 % longer at an identical instruction count (round 3); a scalar or a
  // non-temporal load does not.  The counts are final before this kernel starts.
  // heavy items (no box, no plane: MPR + EPA) fill the item array from its front, the others from its end (publish_candidates): front first
  const int nheavy = ldc(&W.counters[2 * MAXSUB + 2 * s]);
  const int nwork = nheavy + ldc(&W.counters[2 * MAXSUB + 2 * s + 1]);
  int last_i0 = 0;
  for (;;) {
    int i0 = 0;
    // work items per fetch: launch-time numbers (W.narrow_chunk, at most NARROW_CHUNK each).  Smaller chunks balance the tail of a launch
    // (a chunk of four EPA pairs is 45-190 us against a launch of ~100 us alone), larger ones save atomics.  Two sizes: W.narrow_chunk & 15
    // pairs per fetch while this wavefront's LAST fetch started in the heavy region, W.narrow_chunk >> 4 once it has seen the light region.
    const int chunk = last_i0 < nheavy ? (int)(W.narrow_chunk & 15u) : (int)(W.narrow_chunk >> 4);
    if (lane == 0) i0 = atomicAdd(&W.counters[2 * s + 1], chunk);
    i0 = wave_uniform_i(i0);
    if (i0 >= nwork) break;
    last_i0 = i0;
    const int cnt = nwork - i0 < chunk ? nwork - i0 : chunk;
    // ---- the items: word `lane` of item j in it[j]
    unsigned int it[NARROW_CHUNK];
#pragma unroll
    for (int j = 0; j < NARROW_CHUNK; j++) {
      it[j] = 0u;
      if (j < cnt) {
        int i = i0 + j;
        unsigned int pos = i < nheavy ? (unsigned int)i : W.conres_cap - 1u - (unsigned int)(i - nheavy);
        if (lane < ITEM_WORDS) it[j] = W.items[(size_t)pos * ITEM_WORDS + lane];
      }
    }
    // ---- LDS slots of the chunk's hulls, in item order (geom 1, geom 2); -1 = does not fit, staged at its pair's turn.  The hulls are
    // cut into blocks of 256 slots; lane b of the four table registers describes block b (source vertex, vertices left, first LDS float,
    // slots per coordinate of its hull), and the blocks are fetched three at a time: 36 loads in flight, two round trips for a full pool
    int off[2 * NARROW_CHUNK], used = 0, nb = 0;
    int tsrc = 0, tcnt = 0, tdst = 0, tn = 0;
    wave_sync();                                       // (the scans of the previous chunk are done)
#pragma unroll
    for (int j = 0; j < NARROW_CHUNK; j++) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        off[2 * j + h] = -1;
        if (j < cnt) {
          int o = h ? ITEM_GEOM1 : ITEM_GEOM0;
          int vadr = item_i(it[j], o + ITEM_G_VADR), vnum = item_i(it[j], o + ITEM_G_VNUM);
          int n = hull_lds_slots(item_i(it[j], o + ITEM_G_TYPE), vnum);
          if (n && used + n <= HULL_POOL) {
            off[2 * j + h] = used;
            for (int k = 0; k < n; k += 256) {
              bool me = lane == nb;
              tsrc = me ? vadr + k : tsrc; tcnt = me ? vnum - k : tcnt; tdst = me ? 3 * used + k : tdst; tn = me ? n : tn;
              nb++;
            }
            used += n;
          }
        }
      }
    }
    for (int b0 = 0; b0 < nb; b0 += 3) {
      HullStage<4> T[3];
#pragma unroll
      for (int u = 0; u < 3; u++)
        if (b0 + u < nb) hull_stage_issue<4>(m, __builtin_amdgcn_readlane(tsrc, b0 + u), __builtin_amdgcn_readlane(tcnt, b0 + u), 0, T[u]);
#pragma unroll
      for (int u = 0; u < 3; u++)
        if (b0 + u < nb) { HullLDS H{pool + __builtin_amdgcn_readlane(tdst, b0 + u), __builtin_amdgcn_readlane(tn, b0 + u)}; hull_stage_store<4>(H, 0, 0, T[u]); }
    }
    wave_sync();
    // not unrolled: four inlined copies of the query are ~130 KB of code, more than the instruction cache holds
#pragma unroll 1
    for (int j = 0; j < cnt; j++) {
      unsigned long long t0 = SO101_CLOCK();
      unsigned int word = it[0]; int o1 = off[0], o2 = off[1];
#pragma unroll
      for (int q = 1; q < NARROW_CHUNK; q++) if (j == q) { word = it[q]; o1 = off[2 * q]; o2 = off[2 * q + 1]; }
      unsigned int rec = (unsigned int)item_i(word, 0), w = (unsigned int)item_i(word, 1);
      GeomW G1, G2; float rb1, rb2;
      item_geom(word, ITEM_GEOM0, G1, rb1); item_geom(word, ITEM_GEOM1, G2, rb2);
      HullLDS H1{pool, 0}, H2{pool, 0};
      int n1 = hull_lds_slots(G1.type, G1.vnum), n2 = hull_lds_slots(G2.type, G2.vnum);
      if (o1 >= 0) { H1.p = pool + 3 * o1; H1.n = n1; }
      if (o2 >= 0) { H2.p = pool + 3 * o2; H2.n = n2; }
      if ((n1 && o1 < 0) || (n2 && o2 < 0)) {
        // late staging over the slots of the pairs already done (every earlier pair of the chunk is finished; this pair's own staged hull,
        // if any, is staged again behind it - two 512-slot hulls always fit)
        wave_sync();
        H1.p = pool; H1.n = n1; H2.p = pool + 3 * n1; H2.n = n2;
        hull_load(m, G1, H1); hull_load(m, G2, H2);
        wave_sync();
        // (the later pairs of the chunk lost their staged hulls)
#pragma unroll
        for (int q = 0; q < 2 * NARROW_CHUNK; q++) off[q] = -1;
      }
      PairContacts pc;
#ifdef SO101_DEBUG_CLOCKS
      int e = (int)(w / MAXCAND);
      unsigned int* nprof = W.ticks + (size_t)e * MAXCAND + 224;      // per-env sums: [0] fetch, [1] hull load, [2] face scan, [3] MPR, [4] rest
      if (lane == 0) atomicAdd(&nprof[0], (unsigned int)(SO101_CLOCK() - t0));
      unsigned long long t1 = SO101_CLOCK();
      narrow_pair_cached<HullLDS>(m, G1, G2, rb1, rb2, H1, H2, pc, nprof);
      if (lane == 0) atomicAdd(&nprof[5], (unsigned int)(SO101_CLOCK() - t1));
#else
#ifdef NARROW_DIAG_REPEAT      // diagnostic builds only (scripts/build_variant.py): the query NARROW_DIAG_REPEAT times, same result - what the step time owes to the narrowphase
#pragma unroll 1
      for (int rep = 1; rep < NARROW_DIAG_REPEAT; rep++) { narrow_pair_cached<HullLDS>(m, G1, G2, rb1, rb2, H1, H2, pc); wave_sync(); }
#endif
      narrow_pair_cached<HullLDS>(m, G1, G2, rb1, rb2, H1, H2, pc);
#endif
      if (lane == 0) {
        float* r = W.conres + (size_t)rec * CONRES_DIM;
        r[0] = (float)__popc(pc.valid); r[1] = pc.nrm[0]; r[2] = pc.nrm[1]; r[3] = pc.nrm[2];
        int o = 4;                                     // valid slots are written compactly, in slot order
#pragma unroll
        for (int q = 0; q < NCPP; q++)
          if ((pc.valid >> q) & 1u) { r[o] = pc.dist[q]; r[o + 1] = pc.pos[q][0]; r[o + 2] = pc.pos[q][1]; r[o + 3] = pc.pos[q][2]; o += 4; }
        if (SO101_CLOCKS_ON) W.ticks[w] = (unsigned int)(SO101_CLOCK() - t0);
      }
    }
  }
}

namespace so101 {
void launch_narrow(int waves, hipStream_t st, const DevModel* m, int n_envs, const PipeBuffers& W, int substep) {
  hipLaunchKernelGGL(k_narrow, dim3(waves), dim3(64), 0, st, m, n_envs, W, substep);
}
}  // namespace so101
