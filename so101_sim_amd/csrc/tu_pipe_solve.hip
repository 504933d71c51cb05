// Translation unit: the per-substep kernels of the pipelined control step (see so101_pipeline.hpp).
#include "so101_pipeline.hpp"
#include "so101_launch.hpp"

// Longest-processing-time-first launch order for k_pipe_solve.  A launch ends with its slowest env (Newton iteration
// counts: mean 2.7, max ~19) and workgroups are dispatched in index order, so envs that were expensive in the
// previous control step go first: counting sort of the group's envs by log2(cost), descending.  The order only
// changes WHEN an env is processed, never its result.  One workgroup for the whole batch.
__global__ void __launch_bounds__(1024) k_order(const unsigned int* cost, int* order, unsigned char* cls, int e0, int ng, int deal) {   // e0 = 0, ng = N
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
    int pos = atomicAdd(&start[b], 1);
    // deal > 1 (round 5): the sorted envs are DEALT to `deal` equal slices - sorted position p goes to slice p % deal, place p / deal - so that every
    // launch chain gets the same mix of expensive and cheap envs, each slice still most expensive first.  With the sorted order cut into
    // contiguous quarters one chain held all the long Newton solves and heavy pairs: 729 -> 738 k env-steps/s at 4096 envs (first window
    // 779 -> 800 k; 1500 steps 627 -> 634 k on the device clock; 32 768 envs unchanged)
    order[e0 + (deal > 1 ? (pos % deal) * (ng / deal) + pos / deal : pos)] = e0 + i;
    cls[e0 + i] = pos < ng / 8 ? 1 : 0;        // the expensive eighth: served first by the chained step (so101_chain.hpp)
  }
}

__global__ void __launch_bounds__(64, 2) k_pipe_solve(const DevModel* m, StepParams P, DevBuffers B, EventBuffers E, PipeBuffers W, int s, int last,
                                                   float* obs, float* reward, float* discount, unsigned char* step_type,
                                                   unsigned char* need_reset, int* diag, int e0) {
  __shared__ EnvLDS L;
  int e = wave_uniform_i(W.order[e0 + blockIdx.x]);
#ifdef SO101_PRIO_EXP      // (kernel experiment: the envs that were expensive in the previous step - first in the launch order - issue ahead of their SIMD's other wavefront)
  if (blockIdx.x < gridDim.x / SO101_PRIO_EXP) __builtin_amdgcn_s_setprio(3);
#endif
  int act = W.active[e];
  if (act == 0) return;
  SolveIO io{obs, reward, discount, step_type, need_reset, diag};
  if (pipe_solve_env<false>(m, L, P, B, E, W, e, s, last, act, io)) {
    unsigned long long q2 = SO101_CLOCK();
    publish_candidates(m, L, W, e, P.n_envs, s + 1);
#ifdef SO101_DEBUG_CLOCKS
    if (wave_lane() == 0) W.ticks[(size_t)e * MAXCAND + 240 + 13] = (unsigned int)(SO101_CLOCK() - q2);
#endif
  }
}

namespace so101 {
void launch_order(hipStream_t st, const unsigned int* cost, int* order, unsigned char* cls, int n_envs, int deal) {
  hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, st, cost, order, cls, 0, n_envs, deal);
}
void launch_pipe_solve(int n_group, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, const EventBuffers& E,
                       const PipeBuffers& W, int substep, int last, const StepIO& io, unsigned char* need_reset, int* diag, int e0) {
  hipLaunchKernelGGL(k_pipe_solve, dim3(n_group), dim3(64), 0, st, m, P, B, E, W, substep, last, io.obs, io.reward, io.discount, io.step_type,
                     need_reset, diag, e0);
}
}  // namespace so101
