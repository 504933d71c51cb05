// Translation unit: fused control step (Newton).
#include "so101_kernels.hpp"
#include "so101_launch.hpp"

namespace so101 {
void launch_step(int solver, int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B,
                 const PrepBuffers& C, const EventBuffers& E, const StepIO& io, unsigned char* need_reset, int* diag) {
  if (solver == 0) { launch_step_pgs(n_envs, st, m, P, B, C, E, io, need_reset, diag); return; }
  hipLaunchKernelGGL(k_step<1>, dim3(n_envs), dim3(64), 0, st, m, P, B, C, E, io.action, io.obs, io.reward, io.discount, io.step_type,
                     need_reset, diag);
}
}  // namespace so101
