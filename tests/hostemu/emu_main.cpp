// TEST HARNESS ONLY — builds the product's C ABI (so101_hip.hip) for the CPU lane-thread emulation.
#include <hip/hip_runtime.h>
thread_local emu_idx threadIdx;
thread_local emu_idx blockIdx;
pthread_barrier_t emu_barrier;
pthread_barrier_t emu_row_barrier[4];
float emu_xchg_f[64];
int emu_xchg_i[64];
unsigned long long emu_xchg_u;
#include "../../so101_sim_amd/csrc/so101_hip.hip"
#include "../../so101_sim_amd/csrc/tu_step.hip"
#include "../../so101_sim_amd/csrc/tu_reset.hip"
#include "../../so101_sim_amd/csrc/tu_misc.hip"
#include "../../so101_sim_amd/csrc/tu_pipe_begin.hip"
#include "../../so101_sim_amd/csrc/tu_pipe_solve.hip"
#include "../../so101_sim_amd/csrc/tu_pgs_a.hip"
#include "../../so101_sim_amd/csrc/tu_pgs_b.hip"
