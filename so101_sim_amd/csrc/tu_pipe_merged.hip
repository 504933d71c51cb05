// Translation unit: the launches of the merged pipeline (pipeline = 3, see so101_chain.hpp "merged launches").
// Built only with -DSO101_EXPERIMENTAL_PIPELINES (python -m so101_sim_amd.build --experimental -> libso101_hip_exp.so): this step path was built, proven
// bit-identical to the fused step and measured slower than the launch chains; the default library does not carry it.
#ifdef SO101_EXPERIMENTAL_PIPELINES
#include "so101_chain.hpp"
#include "so101_launch.hpp"

// launch `launch` of a chain: s >= 0: solve substep s of this wavefront's env, publish its candidates for s + 1, then help with
// narrowphase chunks; s < 0: chunks only (the candidates of substep 0, published by k_pipe_begin)
__global__ void __launch_bounds__(64, 2) k_pipe_merged(const DevModel* m, StepParams P, DevBuffers B, EventBuffers E, PipeBuffers W, int s, int last, int launch,
                                                    SolveIO io, int e0, int ng) {
  BLOCK_SHARED(EnvLDS, L);
  if (s >= 0) {
    int e = wave_uniform_i(W.order[e0 + blockIdx.x]);
    int act = W.active[e];
    if (act != 0 && pipe_solve_env<false>(m, L, P, B, E, W, e, s, last, act, io)) publish_merged(L, W, e);
    if (last) return;
    merged_published(W, launch);
  }
  merged_helper(m, W, launch, s >= 0 ? ng : 0);
}

namespace so101 {
void launch_pipe_merged(int n_group, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, const EventBuffers& E,
                        const PipeBuffers& W, int substep, int last, int launch, const StepIO& io, unsigned char* need_reset, int* diag, int e0) {
  SolveIO sio{io.obs, io.reward, io.discount, io.step_type, need_reset, diag};
  SO101_LAUNCH_CONCURRENT(k_pipe_merged, dim3(n_group), dim3(64), st, m, P, B, E, W, substep, last, launch, sio, e0, n_group);
}
}  // namespace so101

#else
#include "so101_launch.hpp"
namespace so101 { void launch_pipe_merged(int, hipStream_t, const DevModel*, const StepParams&, const DevBuffers&, const EventBuffers&, const PipeBuffers&, int, int, int, const StepIO&, unsigned char*, int*, int) {} }
#endif
