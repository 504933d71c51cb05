// Translation unit: physics-only stepping, the parity-test stage dump, reward, episode bookkeeping (Newton).
#include "so101_kernels.hpp"
#include "so101_launch.hpp"

__global__ void __launch_bounds__(64) k_begin(const DevModel* m, StepParams P, DevBuffers B, unsigned char* need_reset) {
  int e = blockIdx.x, lane = wave_lane(), N = P.n_envs;
  if (lane < NARM) {
    float q = B.qpos[(size_t)lane * N + e];
    for (int r = 0; r < 5; r++) B.ring[((size_t)r * NARM + lane) * N + e] = q;
    B.ctrl[(size_t)lane * N + e] = m->home_ctrl[lane] + P.action_offset[lane];
  }
  if (lane == 0) { B.step_count[e] = 0; B.ep_return[e] = 0.f; need_reset[e] = 0; }
  if (B.ps_ring && lane < PS_DIM) {           // physics_state delay line: padded with the state the episode starts from
    float v = lane < NQ ? B.qpos[(size_t)lane * N + e] : B.qvel[(size_t)(lane - NQ) * N + e];
    for (int r = 0; r < PS_DELAY; r++) B.ps_ring[((size_t)r * PS_DIM + lane) * N + e] = v;
    B.ps_out[(size_t)e * PS_DIM + lane] = v; B.ps_delayed[(size_t)e * PS_DIM + lane] = v;
  }
}

__global__ void __launch_bounds__(64) k_reward(const DevModel* m, StepParams P, DevBuffers B, float* reward) {
  __shared__ EnvLDS L;
  int e = blockIdx.x;
  load_state(L, B, e, P.n_envs);
  kinematics(m, L);
  float r = task_reward(m, L);
  if (wave_lane() == 0) reward[e] = r;
}

namespace so101 {
void launch_physics(int solver, int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B,
                    int nsub, int freeze, int* diag) {
  if (solver == 0) { launch_physics_pgs(n_envs, st, m, P, B, nsub, freeze, diag); return; }
  hipLaunchKernelGGL(k_physics<1>, dim3(n_envs), dim3(64), 0, st, m, P, B, nsub, freeze, diag);
}
void launch_debug_forward(int solver, int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, float* out) {
  if (solver == 0) { launch_debug_forward_pgs(n_envs, st, m, P, B, out); return; }
  hipLaunchKernelGGL(k_debug_forward<1>, dim3(n_envs), dim3(64), 0, st, m, P, B, out);
}
void launch_begin(int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, unsigned char* need_reset) {
  hipLaunchKernelGGL(k_begin, dim3(n_envs), dim3(64), 0, st, m, P, B, need_reset);
}
void launch_reward(int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, float* reward) {
  hipLaunchKernelGGL(k_reward, dim3(n_envs), dim3(64), 0, st, m, P, B, reward);
}
}  // namespace so101
