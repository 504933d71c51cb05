// Translation unit: PGS instantiations of the fused step and of env.reset().  PGS (the solver BASELINE.json's
// north_star names) runs the fused path only: the pipelined step and the reset prefetch are Newton kernels.
#include "so101_kernels.hpp"
#include "so101_launch.hpp"

namespace so101 {
void launch_step_pgs(int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, const PrepBuffers& C,
                     const EventBuffers& E, const StepIO& io, unsigned char* need_reset, int* diag) {
  hipLaunchKernelGGL(k_step<0>, dim3(n_envs), dim3(64), 0, st, m, P, B, C, E, io.action, io.obs, io.reward, io.discount, io.step_type,
                     need_reset, diag);
}
void launch_settle_pgs(int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, const EventBuffers& E, int* diag) {
  hipLaunchKernelGGL(k_settle<0>, dim3(n_envs), dim3(64), 0, st, m, P, B, E, diag);
}
void launch_settle_table_pgs(int n_envs, int first, int count, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B,
                             float* qpos, float* qvel, float* warm, int* flags) {
  hipLaunchKernelGGL(k_settle_table<0>, dim3((unsigned int)n_envs * (unsigned int)count), dim3(64), 0, st, m, P, B, first, qpos, qvel, warm, flags);
}
void launch_reset_pgs(int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, const PrepBuffers& C,
                      const EventBuffers& E, const unsigned char* mask, unsigned char* need_reset, int* diag) {
  hipLaunchKernelGGL(k_reset<0>, dim3(n_envs), dim3(64), 0, st, m, P, B, C, E, mask, need_reset, diag);
}
}  // namespace so101
