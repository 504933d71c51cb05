// Translation unit: prologue of the pipelined control step.
#ifdef SO101_EXPERIMENTAL_PIPELINES
#include "so101_chain.hpp"
#else
#include "so101_pipeline.hpp"
#endif
#include "so101_launch.hpp"

__global__ void __launch_bounds__(64, 2) k_pipe_begin(const DevModel* m, StepParams P, DevBuffers B, PrepBuffers C, EventBuffers E, PipeBuffers W,
                                                   const float* action, float* obs, float* reward, float* discount,
                                                   unsigned char* step_type, unsigned char* need_reset, int* diag, int e0,
                                                   const ChainParams* chain /* pipeline = 2: the chained step's queues; else NULL */) {
  __shared__ EnvLDS L;
  int e = wave_uniform_i(W.order[e0 + blockIdx.x]), lane = wave_lane(), N = P.n_envs;
  if (need_reset[e]) {
    if (lane == 0) { L.overflow = 0; L.t_collision = 0; L.t_solve = 0; L.t_begin = (unsigned int)SO101_CLOCK(); }
    env_reset<1>(m, L, P, B, C, e);
    store_state(L, B, e, N);
    store_diag(L, diag, e);
    count_events(L, E, e);
    write_first(L, e, obs, reward, discount, step_type, need_reset);
    if (lane == 0) {
      W.active[e] = 0; W.ncand[e] = 0;
#ifdef SO101_EXPERIMENTAL_PIPELINES
      if (chain) atom_add_agent(ldc(&chain->Q.chain_ctl), 1u);       // not stepping in this call: done as far as k_chain is concerned
#endif
    }
    return;
  }
  load_state(L, B, e, N);
  // before_step: ctrl = action + homing offsets, unclamped (so100_task.py:266-287)
  if (lane < NU) L.ctrl[lane] = action[(size_t)e * NU + lane] + P.action_offset[lane];
  wave_sync();
  store_state_aos(L, W, e);
  kinematics(m, L);
  broadphase(m, L);
  if (lane == 0) W.active[e] = 1;
#ifdef SO101_EXPERIMENTAL_PIPELINES
  if (chain) {
    // (all of this becomes visible to k_chain at the launch boundary; the queue protocol is the same as inside it)
    ChainQueues Q = ldc_obj(&chain->Q);
    ChainQ QS = chain_queue_of(Q, Q_SOLVE, e);
    if (publish_chain(L, W, Q, e, 0) == 0 && lane == 0) q_push_lane(QS, solve_item(e, 0));
  } else if (W.mq_ctl) publish_merged(L, W, e);          // pipeline = 3: chunks for the chain's first (narrowphase only) launch
  else
#endif
  publish_candidates(m, L, W, e, N, 0);
}

namespace so101 {
void launch_pipe_begin(int n_group, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, const PrepBuffers& C,
                       const EventBuffers& E, const PipeBuffers& W, const StepIO& io, unsigned char* need_reset, int* diag, int e0,
                       const ChainParams* chain) {
  hipLaunchKernelGGL(k_pipe_begin, dim3(n_group), dim3(64), 0, st, m, P, B, C, E, W, io.action, io.obs, io.reward, io.discount, io.step_type,
                     need_reset, diag, e0, chain);
}
}  // namespace so101
