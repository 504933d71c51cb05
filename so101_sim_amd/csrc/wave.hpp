// Wave-level primitives for gfx950 (64-lane wavefronts).  One environment == one wavefront == one
// 64-thread workgroup, so "wave" and "block" coincide and __syncthreads() is a single-wave barrier.
#ifndef SO101_WAVE_HPP_
#define SO101_WAVE_HPP_
#include <hip/hip_runtime.h>

#define WAVE 64

__device__ __forceinline__ int wave_lane() { return threadIdx.x; }
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// DPP row/bank operations keep reductions in the VALU (no LDS crossbar round trip).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

__device__ __forceinline__ float wave_max_f(float v) {
  v = fmaxf(v, __shfl_xor(v, 32));
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 8));
  v = fmaxf(v, __shfl_xor(v, 4));
  v = fmaxf(v, __shfl_xor(v, 2));
  v = fmaxf(v, __shfl_xor(v, 1));
  return v;
}

__device__ __forceinline__ float wave_sum_f(float v) {
  v += __shfl_xor(v, 32);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __ballot(p); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int wave_prefix(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

__device__ __forceinline__ float wave_bcast_f(float v, int src) { return __shfl(v, src); }
__device__ __forceinline__ int wave_bcast_i(int v, int src) { return __shfl(v, src); }

// (max value, smallest index attaining it) over all lanes; every lane gets the result
__device__ __forceinline__ void wave_argmax(float& val, int& idx) {
  float mx = wave_max_f(val);
  int cand = (val == mx) ? idx : 0x7fffffff;
  cand = min(cand, __shfl_xor(cand, 32));
  cand = min(cand, __shfl_xor(cand, 16));
  cand = min(cand, __shfl_xor(cand, 8));
  cand = min(cand, __shfl_xor(cand, 4));
  cand = min(cand, __shfl_xor(cand, 2));
  cand = min(cand, __shfl_xor(cand, 1));
  val = mx;
  idx = cand;
}
#endif  // SO101_WAVE_HPP_
