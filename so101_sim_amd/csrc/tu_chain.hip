// Translation unit: the persistent kernel of the per-env chained control step (so101_chain.hpp).
// Built only with -DSO101_EXPERIMENTAL_PIPELINES (python -m so101_sim_amd.build --experimental -> libso101_hip_exp.so): this step path was built, proven
// bit-identical to the fused step and measured slower than the launch chains; the default library does not carry it.
#ifdef SO101_EXPERIMENTAL_PIPELINES
#define SO101_OPAQUE_LANE 1
#include "so101_chain.hpp"
#include "so101_launch.hpp"

// Every wavefront of the grid loops: solve item of the expensive class, narrow chunk of the expensive class, solve item,
// narrow chunk, else look at the done counter / abort flag and sleep.  2 waves per SIMD (the narrowphase keeps both hulls
// in registers), 20 KB of LDS per wavefront for the solve items.
//
// The launch parameters live in a device-memory block (ChainParams) instead of the kernel argument segment, and every
// phase loads the part it needs through a laundered pointer: as kernel arguments the ~70 pointers and scalars stayed in
// SGPRs across the whole loop, and the narrowphase (252 VGPRs on its own) had no room left for the registers their spills
// need (first build: 116 spilled VGPRs, 448 B of scratch per lane).
// The narrowphase chunk as a real function call: its 252 VGPRs (both hulls in registers) are allocated on their own.  Inlined
// next to the solver, whose ~430 spilled SGPRs reserve VGPRs for the WHOLE kernel, it spilled 116-168 VGPRs to scratch.
// Nothing but the parameter pointer and the clock is live across the call.
#ifdef CHAIN_NARROW_CALL
SO101_NOINLINE __device__ void chain_narrow_call(const ChainParams* cp_in, unsigned int item_in) {
  const ChainParams* cp = uniform_ptr(cp_in);
  unsigned int item = (unsigned int)wave_uniform_i((int)item_in);
  chain_narrow(ldc(&cp->m), ldc_obj(&cp->W), ldc_obj(&cp->Q), item);
}
#endif

__global__ void __launch_bounds__(64, 2) k_chain(const ChainParams* cp0) {
  BLOCK_SHARED(EnvLDS, L);
  int lane = wave_lane();
  unsigned long long t_idle = wall_clock64(), t_start = t_idle;
  // where this wavefront's time went (10 ns ticks; summed into Q.stats when it leaves)
  unsigned int st_pop = 0, st_idle = 0, st_narrow = 0, st_solve = 0, n_narrow = 0, n_solve = 0, n_idle = 0;
  // Role: the kind of item this wavefront looks for FIRST (it takes the other kind when its own queues are empty).  Wavefronts
  // that share an instruction cache should run the same code: role from the CU id (HW_REG_HW_ID: cu_id 11:8, sh_id 12, se_id 15:13).
  int solve_first = 1;
  {
    const ChainParams* cp = cp0; LAUNDER_UNIFORM(cp);
    int mode = ldc(&cp->Q.role_mode);
#ifndef SO101_EMU
    unsigned int hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    unsigned int cu = (hw >> 8) & 15u, se = (hw >> 13) & 7u;
    if (mode == 1) solve_first = (int)(cu & 1u);
    else if (mode == 2) solve_first = (int)((cu >> 1) & 1u);
    else if (mode == 3) solve_first = (int)(se & 1u);
    else if (mode == 4) solve_first = (int)((cu >> 2) & 1u);
    else if (mode == 5) solve_first = 0;
#endif
  }
  for (;;) {
    unsigned long long t0 = wall_clock64();
    unsigned int item = 0;
    int kind = -1;
    {
      const ChainParams* cp = cp0; LAUNDER_UNIFORM(cp);
      ChainQueues Q = ldc_obj(&cp->Q);
      if (lane == 0) {
        // the four `avail` counts in one round trip, then at most one claim (solve hi, narrow hi, solve lo, narrow lo)
        int a3 = (int)ld_agent(&Q.qctl[64 * 3 + QC_AVAIL]), a1 = (int)ld_agent(&Q.qctl[64 * 1 + QC_AVAIL]);
        int a2 = (int)ld_agent(&Q.qctl[64 * 2 + QC_AVAIL]), a0 = (int)ld_agent(&Q.qctl[64 * 0 + QC_AVAIL]);
        unsigned int* ab = Q.chain_ctl + 32;
        if (solve_first) {
          if (a3 > 0 && q_pop_lane(chain_queue(Q, 3), &item, ab)) kind = 1;
          else if (a2 > 0 && q_pop_lane(chain_queue(Q, 2), &item, ab)) kind = 1;
          else if (a1 > 0 && q_pop_lane(chain_queue(Q, 1), &item, ab)) kind = 0;
          else if (a0 > 0 && q_pop_lane(chain_queue(Q, 0), &item, ab)) kind = 0;
        } else {
          if (a1 > 0 && q_pop_lane(chain_queue(Q, 1), &item, ab)) kind = 0;
          else if (a0 > 0 && q_pop_lane(chain_queue(Q, 0), &item, ab)) kind = 0;
          else if (a3 > 0 && q_pop_lane(chain_queue(Q, 3), &item, ab)) kind = 1;
          else if (a2 > 0 && q_pop_lane(chain_queue(Q, 2), &item, ab)) kind = 1;
        }
      }
      kind = wave_uniform_i(kind);
      item = (unsigned int)wave_uniform_i((int)item);
    }
    if (kind == 1) {
      const ChainParams* cp = cp0; LAUNDER_UNIFORM(cp);
      unsigned long long t1 = wall_clock64();
      chain_solve(ldc(&cp->m), L, ldc_obj(&cp->P), ldc_obj(&cp->B), ldc_obj(&cp->E), ldc_obj(&cp->W), ldc_obj(&cp->Q), item, ldc_obj(&cp->io));
      wave_sync();
      t_idle = wall_clock64();
      st_pop += (unsigned int)(t1 - t0); st_solve += (unsigned int)(t_idle - t1); n_solve++;
      continue;
    }
    if (kind == 0) {
      unsigned long long t1 = wall_clock64();
#ifdef CHAIN_NARROW_CALL
      chain_narrow_call(cp0, item);
#else
      const ChainParams* cp = cp0; LAUNDER_UNIFORM(cp);
      chain_narrow(ldc(&cp->m), ldc_obj(&cp->W), ldc_obj(&cp->Q), item);
#endif
      t_idle = wall_clock64();
      st_pop += (unsigned int)(t1 - t0); st_narrow += (unsigned int)(t_idle - t1); n_narrow++;
      continue;
    }
    int stop = 0;
    {
      const ChainParams* cp = cp0; LAUNDER_UNIFORM(cp);
      unsigned int* ctl = ldc(&cp->Q.chain_ctl);
      if (lane == 0) {
        // ctl[32]: abort word, STICKY across steps (after a watchdog abort the queues and pending counts are inconsistent: every later
        // chained step of this handle ends at its first idle round and its results are invalid until the handle is reconfigured to
        // another step path); ctl[33]: cleared by the host before every launch, so that EVERY aborted step counts once in events[6]
        if (ld_agent(&ctl[0]) >= (unsigned int)ldc(&cp->P.n_envs)) stop = 1;
        else if (ld_agent(&ctl[32]) != 0u) {
          if (atom_add_agent(&ctl[33], 1u) == 0u) atomicAdd(ldc(&cp->E.events) + 6, 1ull);
          stop = 1;
        } else if (wall_clock64() - t_idle > CHAIN_WATCHDOG_TICKS) {
          if (atom_add_agent(&ctl[32], 1u) == 0u && atom_add_agent(&ctl[33], 1u) == 0u) atomicAdd(ldc(&cp->E.events) + 6, 1ull);
          stop = 1;
        }
      }
    }
    if (wave_uniform_i(stop)) break;
    {
      const ChainParams* cp = cp0; LAUNDER_UNIFORM(cp);
      int ns = ldc(&cp->Q.idle_sleeps);
      for (int k = 0; k < ns; k++) idle_sleep();
    }
    st_idle += (unsigned int)(wall_clock64() - t0); n_idle++;
  }
  {
    const ChainParams* cp = cp0; LAUNDER_UNIFORM(cp);
    unsigned long long* S = ldc(&cp->Q.stats);
    if (lane == 0) {
      atomicAdd(&S[CS_T_POP], (unsigned long long)st_pop); atomicAdd(&S[CS_T_IDLE], (unsigned long long)st_idle);
      atomicAdd(&S[CS_T_NARROW], (unsigned long long)st_narrow); atomicAdd(&S[CS_T_SOLVE], (unsigned long long)st_solve);
      atomicAdd(&S[CS_N_NARROW], (unsigned long long)n_narrow); atomicAdd(&S[CS_N_SOLVE], (unsigned long long)n_solve);
      atomicAdd(&S[CS_N_IDLE], (unsigned long long)n_idle); atomicAdd(&S[CS_WAVES], 1ull);
      atomicAdd(&S[CS_T_LIFE], wall_clock64() - t_start);
    }
  }
}

namespace so101 {
void launch_chain(int waves, hipStream_t st, const ChainParams* params) {
#ifdef SO101_EMU
  waves = waves < 3 ? waves : 3;           // OS threads: three wavefronts alive at once exercise every hand-off
#endif
  SO101_LAUNCH_CONCURRENT(k_chain, dim3(waves), dim3(64), st, params);
}
}  // namespace so101

#else
#include "so101_launch.hpp"
namespace so101 { void launch_chain(int, hipStream_t, const ChainParams*) {} }
#endif
