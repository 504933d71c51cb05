// Host-side launchers, one per kernel.  Each is defined in the translation unit that holds the kernel (tu_*.hip):
// without relocatable device code a kernel can only be launched from the file it is compiled in, and splitting the
// kernels over several files lets the library build in parallel.  `solver`: 1 = Newton, 0 = PGS (include/so101.h).
#pragma once
#include <hip/hip_runtime.h>
#include "so101_model.hpp"

namespace so101 {

struct StepIO {            // per-call arrays of so101_step (device pointers)
  const float* action; float* obs; float* reward; float* discount; unsigned char* step_type;
};

// fused kernels -------------------------------------------------------------------------------------------------
void launch_reset(int solver, int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B,
                  const PrepBuffers& C, const EventBuffers& E, const unsigned char* mask, unsigned char* need_reset, int* diag);
void launch_settle(int solver, int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, const EventBuffers& E, int* diag);
void launch_settle_table(int solver, int n_envs, int first, int count, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B,
                         float* qpos, float* qvel, float* warm, int* flags);
void launch_prepare(int waves, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, const PrepBuffers& C);   // Newton
void launch_step(int solver, int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B,
                 const PrepBuffers& C, const EventBuffers& E, const StepIO& io, unsigned char* need_reset, int* diag);
void launch_physics(int solver, int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B,
                    int nsub, int freeze, int* diag);
void launch_debug_forward(int solver, int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, float* out);
void launch_begin(int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, unsigned char* need_reset);
void launch_reward(int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, float* reward);

// pipelined step (Newton) -----------------------------------------------------------------------------------------
void launch_order(hipStream_t st, const unsigned int* cost, int* order, unsigned char* cls, int n_envs, int deal = 1);   // deal: equal slices the sorted envs are dealt to (1 = plain sorted order)
void launch_pipe_begin(int n_group, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, const PrepBuffers& C,
                       const EventBuffers& E, const PipeBuffers& W, const StepIO& io, unsigned char* need_reset, int* diag, int e0,
                       const ChainParams* chain = nullptr);
void launch_narrow(int waves, hipStream_t st, const DevModel* m, int n_envs, const PipeBuffers& W, int substep);
void launch_pipe_solve(int n_group, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, const EventBuffers& E,
                       const PipeBuffers& W, int substep, int last, const StepIO& io, unsigned char* need_reset, int* diag, int e0);

// merged launches (pipeline = 3): solve of substep s + narrowphase chunks of substep s + 1 in one launch per chain
void launch_pipe_merged(int n_group, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, const EventBuffers& E,
                        const PipeBuffers& W, int substep, int last, int launch, const StepIO& io, unsigned char* need_reset, int* diag, int e0);

// per-env chained step (so101_chain.hpp): k_order + k_pipe_begin + ONE persistent launch
void launch_chain(int waves, hipStream_t st, const ChainParams* params /* device memory */);

// the PGS instantiations live in their own files
void launch_reset_pgs(int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, const PrepBuffers& C,
                      const EventBuffers& E, const unsigned char* mask, unsigned char* need_reset, int* diag);
void launch_settle_pgs(int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, const EventBuffers& E, int* diag);
void launch_settle_table_pgs(int n_envs, int first, int count, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B,
                             float* qpos, float* qvel, float* warm, int* flags);
void launch_step_pgs(int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, const PrepBuffers& C,
                     const EventBuffers& E, const StepIO& io, unsigned char* need_reset, int* diag);
void launch_physics_pgs(int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, int nsub, int freeze, int* diag);
void launch_debug_forward_pgs(int n_envs, hipStream_t st, const DevModel* m, const StepParams& P, const DevBuffers& B, float* out);

}  // namespace so101
