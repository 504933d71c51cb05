// Translation unit: env.reset() and the reset prefetch (Newton).
#include "so101_kernels.hpp"
#include "so101_launch.hpp"

namespace so101 {
void launch_reset(int solver, int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B,
                  const PrepBuffers& C, const EventBuffers& E, const unsigned char* mask, unsigned char* need_reset, int* diag) {
  if (solver == 0) { launch_reset_pgs(n_envs, st, m, P, B, C, E, mask, need_reset, diag); return; }
  hipLaunchKernelGGL(k_reset<1>, dim3(n_envs), dim3(64), 0, st, m, P, B, C, E, mask, need_reset, diag);
}
void launch_settle(int solver, int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, const EventBuffers& E, int* diag) {
  if (solver == 0) { launch_settle_pgs(n_envs, st, m, P, B, E, diag); return; }
  hipLaunchKernelGGL(k_settle<1>, dim3(n_envs), dim3(64), 0, st, m, P, B, E, diag);
}
void launch_settle_table(int solver, int n_envs, int first, int count, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B,
                         float* qpos, float* qvel, float* warm, int* flags) {
  if (solver == 0) { launch_settle_table_pgs(n_envs, first, count, st, m, P, B, qpos, qvel, warm, flags); return; }
  hipLaunchKernelGGL(k_settle_table<1>, dim3((unsigned int)n_envs * (unsigned int)count), dim3(64), 0, st, m, P, B, first, qpos, qvel, warm, flags);
}
void launch_prepare(int waves, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, const PrepBuffers& C) {
  for (int ahead = 0; ahead < 2; ahead++) hipLaunchKernelGGL(k_prepare<1>, dim3(waves), dim3(64), 0, st, m, P, B, C, ahead);
}
}  // namespace so101
