// Translation unit: PGS instantiations of so101_physics and so101_debug_forward.
#include "so101_kernels.hpp"
#include "so101_launch.hpp"

namespace so101 {
void launch_physics_pgs(int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, int nsub, int freeze, int* diag) {
  hipLaunchKernelGGL(k_physics<0>, dim3(n_envs), dim3(64), 0, st, m, P, B, nsub, freeze, diag);
}
void launch_debug_forward_pgs(int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, float* out) {
  hipLaunchKernelGGL(k_debug_forward<0>, dim3(n_envs), dim3(64), 0, st, m, P, B, out);
}
}  // namespace so101
