// Translation unit: the narrowphase kernel of the pipelined control step (see so101_pipeline.hpp).
// (everything derived from the lane index is loop invariant in a persistent kernel; LLVM hoists it in front of the work loop and keeps it
// live across everything: see wave.hpp.  With the opaque lane index k_narrow needs 212 VGPRs instead of 246 at two waves per SIMD.)
#define SO101_OPAQUE_LANE
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
// Two instances, both at two wavefronts per SIMD (no scratch, a pool of 1536 vertex slots): ROWS = false for batches up to 8192 envs, ROWS =
// true - with the row pass for the light region of the list, below - for larger ones.  Measured, round 5, env-steps/s at 4096 / 32768 envs:
// no row pass 737 k (light pairs two per fetch) / 996 k; row pass 722 k / 1078 k; the register-cache kernel of round 4: 727 k / 1038 k.  Three
// wavefronts per SIMD (168 VGPRs, 1024 slots) gave 738 k at 4096 envs before the hull-against-hull patches existed and 656 k with them (137
// spilled VGPRs instead of 41); NARROW_WAVES_SMALL = 3 builds that instance.
#ifndef NARROW_WAVES_SMALL
#define NARROW_WAVES_SMALL 2
#endif
template <bool ROWS> struct NarrowCfg { static constexpr int waves = ROWS ? 2 : NARROW_WAVES_SMALL, pool = (ROWS || NARROW_WAVES_SMALL == 2) ? 1536 : 1024; };

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

template <bool ROWS>
__global__ void __launch_bounds__(64, NarrowCfg<ROWS>::waves) k_narrow(const DevModel* m, int N, PipeBuffers W, int s) {
  constexpr int HULL_POOL = NarrowCfg<ROWS>::pool;          // vertex slots of the LDS hull pool: 256 or 512 per staged hull
  __shared__ __attribute__((aligned(16))) float pool[3 * HULL_POOL];
  // the row pass: every row reads its own item.  (Shares its storage with the values the full-wave query parks across MPR / EPA -
  // narrow_park_store() of so101_device.hpp -: the row pass runs the closed forms only and parks nothing, and its items are dead before
  // the full-wave loop starts.)
  unsigned int* row_items = (unsigned int*)narrow_park_store();
  static_assert(NARROW_CHUNK * ITEM_WORDS <= NARROW_PARK_WORDS, "the row items must fit the park area");
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
    const int chunk = last_i0 < nheavy ? (int)(W.narrow_chunk & 15u) : (int)((W.narrow_chunk >> 4) & 15u);
    if (lane == 0) i0 = atomicAdd(&W.counters[2 * s + 1], chunk);
    i0 = wave_uniform_i(i0);
    if (i0 >= nwork) break;
    last_i0 = i0;
#ifdef SO101_PRIO_EXP      // (kernel experiment: heavy pairs issue ahead of the SIMD's other wavefront)
    if (i0 < nheavy) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#endif
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
          if (!ROWS && item_i(it[j], 47) != 0) n = 0;          // (a fast item: its hull is only staged if the fast path hands the pair back, below)
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
    // ---- ROW PASS (light region of the list: a flat face - the plane, a box - against anything): four pairs at once, one per DPP row of 16
    // lanes (policy G16), closed forms only.  The wavefront is issue-bound (doubling the queries of every pair lengthened the control step by
    // 48 %; a third wavefront per SIMD changed nothing), and two thirds of a light pair's instructions are wave-uniform arithmetic issued
    // for 64 lanes on one pair's numbers: here one instruction stream serves four pairs, the hull scans cost the same per pair (16 lanes x
    // four times the steps), the reductions stay inside a row (four DPP steps, no v_readlane).  Rows whose pair needs the iterative query
    // (no exact face: ~10 %) report it and are served by the whole wavefront below.  Same vertices, same arithmetic: same contacts.
    unsigned int todo = (1u << cnt) - 1u;              // pairs the full-wave loop below still has to do
    if constexpr (ROWS) if (i0 >= nheavy) {
      unsigned long long tr0 = SO101_CLOCK();
#pragma unroll
      for (int j = 0; j < NARROW_CHUNK; j++) if (lane < ITEM_WORDS) row_items[j * ITEM_WORDS + lane] = it[j];
      wave_sync();
      const int row = lane >> 4;
      bool settled = false;
      if (row < cnt) {
        const unsigned int* I = row_items + row * ITEM_WORDS;
        GeomW G1, G2;
        auto geom = [&](int o, GeomW& G) {
          G.type = (int)I[o + ITEM_G_TYPE]; G.vadr = (int)I[o + ITEM_G_VADR]; G.vnum = (int)I[o + ITEM_G_VNUM];
#pragma unroll
          for (int i = 0; i < 3; i++) { G.size[i] = __uint_as_float(I[o + ITEM_G_SIZE + i]); G.p[i] = __uint_as_float(I[o + ITEM_G_P + i]); G.c[i] = __uint_as_float(I[o + ITEM_G_C + i]); }
#pragma unroll
          for (int i = 0; i < 9; i++) G.R[i] = __uint_as_float(I[o + ITEM_G_R + i]);
        };
        geom(ITEM_GEOM0, G1); geom(ITEM_GEOM1, G2);
        float rb1 = __uint_as_float(I[ITEM_GEOM0 + ITEM_G_RBOUND]), rb2 = __uint_as_float(I[ITEM_GEOM1 + ITEM_G_RBOUND]);
        int o1 = off[0], o2 = off[1];
#pragma unroll
        for (int q = 1; q < NARROW_CHUNK; q++) if (row == q) { o1 = off[2 * q]; o2 = off[2 * q + 1]; }
        int n1 = hull_lds_slots(G1.type, G1.vnum), n2 = hull_lds_slots(G2.type, G2.vnum);
        // (a hull that found no room in the pool is scanned in memory by its row: n = 0)
        HullLDS H1{pool + 3 * (o1 < 0 ? 0 : o1), o1 < 0 ? 0 : n1}, H2{pool + 3 * (o2 < 0 ? 0 : o2), o2 < 0 ? 0 : n2};
        PairContacts pc;
        settled = narrow_pair_cached<HullLDS, G16, true>(m, G1, G2, rb1, rb2, H1, H2, pc);
        if (settled && (lane & 15) == 0) {
          float* r = W.conres + (size_t)I[0] * CONRES_DIM;
          r[0] = (float)__popc(pc.valid); r[1] = pc.nrm[0]; r[2] = pc.nrm[1]; r[3] = pc.nrm[2];
          int o = 4;
#pragma unroll
          for (int q = 0; q < NCPP; q++)
            if ((pc.valid >> q) & 1u) { r[o] = pc.dist[q]; r[o + 1] = pc.pos[q][0]; r[o + 2] = pc.pos[q][1]; r[o + 3] = pc.pos[q][2]; o += 4; }
        }
      }
      unsigned long long sm = wave_ballot(settled);
#ifdef SO101_EMU_ROWSTATS
      if (lane == 0) fprintf(stderr, "row pass: cnt %d settled mask %llx i0 %d nheavy %d nwork %d\n", cnt, sm, i0, nheavy, nwork);
#endif
      todo = 0u;
#pragma unroll
      for (int j = 0; j < NARROW_CHUNK; j++) if (j < cnt && !((sm >> (16 * j)) & 1ull)) todo |= 1u << j;
#ifdef SO101_DEBUG_CLOCKS
      if (lane == 0) {          // per-env sums (scripts/gpu_narrow_ticks.py): [10] row-pass ticks, [11] rows attempted, [12] rows settled
        unsigned int* rp = W.ticks + (size_t)(item_i(it[0], 1) / MAXCAND) * MAXCAND + 224;
        atomicAdd(&rp[10], (unsigned int)(SO101_CLOCK() - tr0)); atomicAdd(&rp[11], (unsigned int)cnt); atomicAdd(&rp[12], (unsigned int)(cnt - __popc(todo)));
      }
#endif
    }
    wave_sync();
    // not unrolled: four inlined copies of the query are ~130 KB of code, more than the instruction cache holds
#pragma unroll 1
    for (int j = 0; j < cnt; j++) {
      if (!((todo >> j) & 1u)) continue;
      unsigned long long t0 = SO101_CLOCK();
      unsigned int word = it[0]; int o1 = off[0], o2 = off[1];
#pragma unroll
      for (int q = 1; q < NARROW_CHUNK; q++) if (j == q) { word = it[q]; o1 = off[2 * q]; o2 = off[2 * q + 1]; }
      unsigned int rec = (unsigned int)item_i(word, 0), w = (unsigned int)item_i(word, 1);
      GeomW G1, G2; float rb1, rb2;
      item_geom(word, ITEM_GEOM0, G1, rb1); item_geom(word, ITEM_GEOM1, G2, rb2);
      // ---- fast path (round 6): a plane / box face against a hull whose work item names a cell of the hull's support-vertex lists.  Those entries
      // (two per lane at most) ARE the hull for the query's first support direction and its patch samples: one coalesced load instead of staging
      // 256-512 vertices in LDS, one or two vertices per lane instead of four to sixteen in every scan.  Same arithmetic on the same floats, the same
      // winners: the record is the one the full query writes (the fused step scans the whole hull and the identity tests compare the two).  A pair
      // the first face does not settle goes through the full query below.
      if constexpr (!ROWS) {
        unsigned int fw = (unsigned int)item_i(word, 47);
        if (fw != 0u && ldc(&m->hl_entry) != nullptr && light_first_cell(G1, G2) == (int)(fw >> 8)) {
          const int fcnt = (int)(fw & 255u);
          const float* E = ldc(&m->hl_entry) + 4 * (size_t)(unsigned int)item_i(word, 46);
          HullSub S1, S2;
#pragma unroll
          for (int q = 0; q < 2; q++) {
            int k = lane + WAVE * q;
            bool in = k < fcnt;
            float4 ev; ev.x = 0.f; ev.y = 0.f; ev.z = 0.f; ev.w = 0.f;
            if (in) ev = *(const float4*)(E + 4 * (size_t)k);
            S2.x[q] = ev.x; S2.y[q] = ev.y; S2.z[q] = ev.z; S2.i[q] = in ? __float_as_int(ev.w) : 0x7fffffff;
            S1.x[q] = 0.f; S1.y[q] = 0.f; S1.z[q] = 0.f; S1.i[q] = 0x7fffffff;
          }
          PairContacts fpc;
          if (narrow_pair_cached<HullSub, G64, true, true>(m, G1, G2, rb1, rb2, S1, S2, fpc)) {
            if (lane == 0) {
              float* r = W.conres + (size_t)rec * CONRES_DIM;
              r[0] = (float)__popc(fpc.valid); r[1] = fpc.nrm[0]; r[2] = fpc.nrm[1]; r[3] = fpc.nrm[2];
              int o = 4;
#pragma unroll
              for (int q = 0; q < NCPP; q++)
                if ((fpc.valid >> q) & 1u) { r[o] = fpc.dist[q]; r[o + 1] = fpc.pos[q][0]; r[o + 2] = fpc.pos[q][1]; r[o + 3] = fpc.pos[q][2]; o += 4; }
              if (SO101_CLOCKS_ON) W.ticks[w] = ((unsigned int)(SO101_CLOCK() - t0) & 0x0fffffffu) | ((unsigned int)__popc(fpc.valid) << 28);
            }
            continue;
          }
        }
      }
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
        if (SO101_CLOCKS_ON) W.ticks[w] = ((unsigned int)(SO101_CLOCK() - t0) & 0x0fffffffu) | ((unsigned int)__popc(pc.valid) << 28);      // (profiling builds: 10 ns ticks | contacts << 28)
      }
    }
  }
}

namespace so101 {
void launch_narrow(int waves, hipStream_t st, const DevModel* m, int n_envs, const PipeBuffers& W, int substep) {
  // W.narrow_chunk bit 8: the instance with the row pass (large batches; set by the host, so101_hip.hip)
  if (W.narrow_chunk & 256u) hipLaunchKernelGGL(k_narrow<true>, dim3(waves), dim3(64), 0, st, m, n_envs, W, substep);
  else hipLaunchKernelGGL(k_narrow<false>, dim3(waves), dim3(64), 0, st, m, n_envs, W, substep);
}
}  // namespace so101
