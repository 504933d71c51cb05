// Translation unit: the narrowphase kernel of the pipelined control step (see so101_pipeline.hpp).
#include "so101_pipeline.hpp"
#include "so101_launch.hpp"

// One wavefront per candidate pair (policy G64 of so101_device.hpp).  No LDS; the two geoms (and the first 512 vertices
// of their hulls) live in registers.  Work items are taken NARROW_CHUNK at a time: one atomic and one dependent pair
// of loads per chunk instead of per item (that chain costs ~3 us, an MPR query on two boxes ~7 us).
//
// Measured alternative (kept as policy G16, bit-identical results): one pair per DPP row of 16 lanes, four pairs per
// wavefront.  It is SLOWER (4096-env bench 490-509 k against 635 k env-steps/s): ~70 % of a query's instructions are
// the lane-parallel hull scans, not the uniform portal math, and a row caches only 128 vertices of a hull in registers
// (the arm links have 400-525), so every support call of a big hull goes back to L2.
#ifndef NARROW_WAVES
#define NARROW_WAVES 2
#endif
#ifndef NARROW_CACHE
#define NARROW_CACHE HullLDS
#endif
__global__ void __launch_bounds__(64, NARROW_WAVES) k_narrow(const DevModel* m, int N, PipeBuffers W, int s) {
  int lane = wave_lane();
  // Scalar load on purpose.  The count shares its cache line with the cursor every wave of this launch does atomics on; when the
  // compiler picked a plain vector load here (any unrelated edit at the top of the kernel flips its choice) the whole launch
  // ran 26 % longer at an identical instruction count (measured: 287 us against 227 us per launch, 582 k against 685 k
  // env-steps/s); a scalar or a non-temporal load does not.  The count is final before this kernel starts.
  int nwork = ldc(&W.counters[2 * s]);
  // heavy items (no box, no plane: MPR + EPA) fill the list from its front, the others from its end (publish_candidates): front first
  const int nheavy = ldc(&W.counters[2 * MAXSUB + 2 * s]);
  const unsigned int* list = W.work + (size_t)(s & 1) * W.work_cap;
  int last_i0 = 0;
  for (;;) {
    int i0 = 0;
    // work items per fetch: launch-time numbers (W.narrow_chunk, at most NARROW_CHUNK each).  Smaller chunks balance the tail of a launch
    // (a chunk of four EPA pairs is 45-190 us against a launch of ~100 us alone), larger ones save atomics.  Measured in round 4 at 4096
    // envs, one size: 1 / 2 / 3 / 4 / 8 pairs -> 509 / 678 / 688 / 661 / 584 k env-steps/s (32 768 envs: 2 / 4 -> 912 / 962 k); with the
    // heavy-first list order and two sizes (heavy / light): 3/3 707 k, 2/3 712 k, 2/4 720 k, 1/4 692 k.  The host picks 2/4 up to 8192
    // envs and 4/4 above.
    // (two sizes: W.narrow_chunk & 15 pairs per fetch while this wavefront's LAST fetch started in the heavy region of the list,
    //  W.narrow_chunk >> 4 once it has seen the light region - no extra read of the cursor's cache line)
    const int chunk = last_i0 < nheavy ? (int)(W.narrow_chunk & 15u) : (int)(W.narrow_chunk >> 4);
    if (lane == 0) i0 = atomicAdd(&W.counters[2 * s + 1], chunk);
    i0 = wave_uniform_i(i0);
    if (i0 >= nwork) break;
    last_i0 = i0;
    unsigned int wl = 0, cl = 0;
    int rb = 0;
    if (lane < chunk && i0 + lane < nwork) {
      int i = i0 + lane;
      wl = list[i < nheavy ? (unsigned int)i : W.work_cap - 1u - (unsigned int)(i - nheavy)];
      cl = W.cand[wl];
      rb = W.cbase[wl / MAXCAND];                 // the record of candidate k of env e sits at cbase[e] + k
    }
    // not unrolled: four inlined copies of the MPR query are ~130 KB of code, more than the instruction cache holds
#pragma unroll 1
    for (int j = 0; j < chunk; j++) {
      if (i0 + j >= nwork) break;
      unsigned long long t0 = SO101_CLOCK();
      unsigned int w = (unsigned int)__builtin_amdgcn_readlane((int)wl, j), c = (unsigned int)__builtin_amdgcn_readlane((int)cl, j);
      unsigned int rec = (unsigned int)__builtin_amdgcn_readlane(rb, j) + w % MAXCAND;
      if (rec >= W.conres_cap) continue;          // no room for this candidate's contact record (counted by its env)
      int e = (int)(w / MAXCAND), g1 = (int)(c & 0xffffu), g2 = (int)(c >> 16);
      const float* pose = W.pose + (size_t)e * (NDYN * 12);
      int d1 = ldc(ldc(&m->geom_dyn) + g1), d2 = ldc(ldc(&m->geom_dyn) + g2);
      const float* p1 = pose + 12 * (d1 < 0 ? 0 : d1); const float* p2 = pose + 12 * (d2 < 0 ? 0 : d2);
      GeomW G1, G2;
      load_geom_at(m, g1, p1, p1 + 3, G1); load_geom_at(m, g2, p2, p2 + 3, G2);
      PairContacts pc;
#ifdef SO101_DEBUG_CLOCKS
      unsigned int* nprof = W.ticks + (size_t)e * MAXCAND + 224;      // per-env sums: [0] fetch, [1] hull load, [2] face scan, [3] MPR, [4] rest
      if (lane == 0) atomicAdd(&nprof[0], (unsigned int)(SO101_CLOCK() - t0));
      unsigned long long t1 = SO101_CLOCK();
      narrow_pair<NARROW_CACHE>(m, G1, G2, g1, g2, pc, nprof);
      if (lane == 0) atomicAdd(&nprof[5], (unsigned int)(SO101_CLOCK() - t1));
#else
      narrow_pair<NARROW_CACHE>(m, G1, G2, g1, g2, pc);
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
