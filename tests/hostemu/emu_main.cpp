// TEST HARNESS ONLY — builds the product's C ABI (so101_hip.hip) for the CPU lane-thread emulation.
#include <hip/hip_runtime.h>
thread_local emu_idx threadIdx;
thread_local emu_idx blockIdx;
thread_local EmuBlock* emu_blk;
#include "../../so101_sim_amd/csrc/so101_hip.hip"
#include "../../so101_sim_amd/csrc/tu_step.hip"
#include "../../so101_sim_amd/csrc/tu_reset.hip"
#include "../../so101_sim_amd/csrc/tu_misc.hip"
#include "../../so101_sim_amd/csrc/tu_pipe_begin.hip"
#include "../../so101_sim_amd/csrc/tu_pipe_solve.hip"
#include "../../so101_sim_amd/csrc/tu_narrow.hip"
#include "../../so101_sim_amd/csrc/tu_chain.hip"
#include "../../so101_sim_amd/csrc/tu_pipe_merged.hip"
#include "../../so101_sim_amd/csrc/tu_pgs_a.hip"
#include "../../so101_sim_amd/csrc/tu_pgs_b.hip"
#include "../../so101_sim_amd/csrc/tu_tree.hip"
#include "../../so101_sim_amd/csrc/tu_tree64.hip"
#include "../../so101_sim_amd/csrc/tu_tree_api.hip"
