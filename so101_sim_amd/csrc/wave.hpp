// Wave-level primitives for gfx950 (64-lane wavefronts).  One environment == one wavefront == one
// 64-thread workgroup, so "wave" and "block" coincide and __syncthreads() is a single-wave barrier.
//
// Reductions use DPP (data-parallel primitives: the cross-lane operand is fetched inside the VALU, no LDS
// crossbar round trip as with ds_bpermute/__shfl) and v_readlane for the final broadcast.
#ifndef SO101_WAVE_HPP_
#define SO101_WAVE_HPP_
#include <hip/hip_runtime.h>

#define WAVE 64
#define BLOCK_SHARED(T, name) __shared__ T name
#define SO101_LAUNCH_CONCURRENT(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)

// SO101_OPAQUE_LANE (tu_chain.hip): the lane index comes out of a volatile asm, once per call site.  In the persistent
// kernel everything derived from threadIdx.x is invariant with respect to the work loop; LLVM hoisted hundreds of lane
// predicates and lane-derived addresses of BOTH phases in front of the loop and kept them live across it (first build: 168
// spilled VGPRs, 720 B of scratch per lane).  An asm result cannot be hoisted or merged, so lane-derived values live where
// the source computes them; two VALU instructions per call site.
__device__ __forceinline__ int wave_lane() {
#ifdef SO101_OPAQUE_LANE
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  __builtin_assume(l >= 0 && l < 64);
  return l;
#else
  return threadIdx.x;
#endif
}
__device__ __forceinline__ void wave_sync() { __syncthreads(); }
// Exchange through LDS only, inside the one wavefront of the workgroup: a wavefront's LDS instructions execute in order, so a ds_read behind a
// ds_write sees it; what is needed is that the compiler keeps that order.  Unlike wave_sync() (fence + s_barrier: s_waitcnt vmcnt(0)) this does not
// wait for global loads that are still in flight - the point of it (software-pipelined fetches in so101_tree.hpp).  NOT for data handed over through
// global memory.
__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// DPP controls (GFX9 encoding)
#define DPP_QUAD_XOR1 0xB1        // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E        // quad_perm [2,3,0,1]
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float v) {
  int x = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
}

// all-lanes max: butterflies inside each row of 16, then row broadcasts; lane 63 ends with the total
__device__ __forceinline__ float wave_max_f(float v) {
  v = fmaxf(v, dpp_f<DPP_QUAD_XOR1>(v));
  v = fmaxf(v, dpp_f<DPP_QUAD_XOR2>(v));
  v = fmaxf(v, dpp_f<DPP_ROW_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
  v = fmaxf(v, dpp_f<DPP_ROW_BCAST15, 0xA>(v));
  v = fmaxf(v, dpp_f<DPP_ROW_BCAST31, 0xC>(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ int wave_min_i(int v) {
  v = min(v, dpp_i<DPP_QUAD_XOR1>(v));
  v = min(v, dpp_i<DPP_QUAD_XOR2>(v));
  v = min(v, dpp_i<DPP_ROW_HALF_MIRROR>(v));
  v = min(v, dpp_i<DPP_ROW_MIRROR>(v));
  v = min(v, dpp_i<DPP_ROW_BCAST15, 0xA>(v));
  v = min(v, dpp_i<DPP_ROW_BCAST31, 0xC>(v));
  return __builtin_amdgcn_readlane(v, 63);
}

// all-lanes sum, every lane gets the result: butterflies inside each row of 16 with DPP (xor 1, xor 2, half-row
// mirror, row mirror: both partners of every step add the same two values, so all 16 lanes of a row end with
// identical bits), then the four row sums are combined as (r0 + r1) + (r2 + r3) through v_readlane.  No LDS crossbar
// (ds_bpermute) round trips; the emulation harness reproduces the same order.  Must be called with all lanes active.
__device__ __forceinline__ float wave_sum_f(float v) {
  v += dpp_f<DPP_QUAD_XOR1>(v);
  v += dpp_f<DPP_QUAD_XOR2>(v);
  v += dpp_f<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_f<DPP_ROW_MIRROR>(v);
  float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}

// The same sum when the caller knows (wave-uniform `row0_only`) that lanes 16-63 contribute +0.0: their row sums are
// +0.0, so (r0 + 0) + (0 + 0) == r0 + 0 bit for bit and three v_readlane + two adds are skipped.  The Newton solver's
// sums over contacts qualify whenever an env has at most 16 contacts (most envs, most of the time).
__device__ __forceinline__ float wave_sum_rows_f(float v, bool row0_only) {
  v += dpp_f<DPP_QUAD_XOR1>(v);
  v += dpp_f<DPP_QUAD_XOR2>(v);
  v += dpp_f<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_f<DPP_ROW_MIRROR>(v);
  float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  if (row0_only) return r0 + 0.f;
  float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __ballot(p); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int wave_prefix(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// Load from memory that no kernel ever writes (the model tables): the constant address space lets the compiler use
// scalar loads (s_load, SGPR result, scalar branches on it) whenever the address is wave-uniform, even in kernels
// that also store to global memory.
template <class T>
__device__ __forceinline__ T ldc(const T* p) { return *(const __attribute__((address_space(4))) T*)(unsigned long long)p; }

// a whole (trivially copyable) object from memory that no kernel writes, through scalar loads
template <class T>
__device__ __forceinline__ T ldc_obj(const T* p) {
  T out;
  // (typed source pointer: with a void* the copy has alignment 1 and becomes VECTOR loads)
  __builtin_memcpy(&out, (const __attribute__((address_space(4))) T*)(unsigned long long)p, sizeof(T));
  return out;
}
// a pointer that is the same in every lane but arrived in vector registers (argument of a non-inlined device function)
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  unsigned long long v = (unsigned long long)p;
  unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)v), hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(v >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}
#define SO101_NOINLINE __attribute__((noinline))
// makes the compiler forget what it knows about a wave-uniform pointer: loads through it are issued where the source has them
// instead of being hoisted to the top of the kernel and kept in registers across everything in between
#define LAUNDER_UNIFORM(p) asm volatile("" : "+s"(p))

// tells the compiler that v is the same in every lane (keeps it in an SGPR: scalar loads, scalar branches)
__device__ __forceinline__ int wave_uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

// value of lane `src` (wave-uniform src): v_readlane, the result lives in an SGPR
__device__ __forceinline__ float wave_get_f(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); }

__device__ __forceinline__ int wave_get_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }       // (wave-uniform src)
__device__ __forceinline__ float wave_bcast_f(float v, int src) { return __shfl(v, src); }
__device__ __forceinline__ int wave_bcast_i(int v, int src) { return __shfl(v, src); }

// (max value, smallest index attaining it) over all lanes; every lane gets the result
__device__ __forceinline__ void wave_argmax(float& val, int& idx) {
  float mx = wave_max_f(val);
  int cand = (val == mx) ? idx : 0x7fffffff;
  idx = wave_min_i(cand);
  val = mx;
}

// argmax that also carries a 3-vector payload held by each lane (the winning lane's payload is broadcast
// with v_readlane, no memory access): used by the hull support function
__device__ __forceinline__ void wave_argmax3(float& val, int& idx, float& x, float& y, float& z) {
  float mx = wave_max_f(val);
  int cand = (val == mx) ? idx : 0x7fffffff;
  int best = wave_min_i(cand);
  unsigned long long who = __ballot(cand == best);
  int src = who ? (int)__builtin_ctzll(who) : 0;
  x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src));
  y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), src));
  z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), src));
  val = mx;
  idx = best;
}
// argmax with a 3-vector payload inside each DPP row of 16 lanes (four independent groups per wavefront): four
// butterfly steps (xor 1, xor 2, half-row mirror, row mirror) carry (value, index, payload) and keep the better of the
// two partners - larger value, smaller index on ties - so all 16 lanes of a row end with the row's winner.  Rows may be
// individually inactive (EXEC), the exchanges never leave the row.
template <int CTRL>
__device__ __forceinline__ void row_argmax3_step(float& val, int& idx, float& x, float& y, float& z) {
  float pv = dpp_f<CTRL>(val), px = dpp_f<CTRL>(x), py = dpp_f<CTRL>(y), pz = dpp_f<CTRL>(z);
  int pi = dpp_i<CTRL>(idx);
  bool take = pv > val || (pv == val && pi < idx);
  val = take ? pv : val; idx = take ? pi : idx; x = take ? px : x; y = take ? py : y; z = take ? pz : z;
}
__device__ __forceinline__ void row_argmax3(float& val, int& idx, float& x, float& y, float& z) {
  row_argmax3_step<DPP_QUAD_XOR1>(val, idx, x, y, z);
  row_argmax3_step<DPP_QUAD_XOR2>(val, idx, x, y, z);
  row_argmax3_step<DPP_ROW_HALF_MIRROR>(val, idx, x, y, z);
  row_argmax3_step<DPP_ROW_MIRROR>(val, idx, x, y, z);
}
// ---- matrix cores: D (32 x 32, f32) += A (32 x 2) * B (2 x 32), v_mfma_f32_32x32x2_f32.  Lane l supplies A[l % 32][l / 32] and
// B[l / 32][l % 32]; it holds D[8 (r / 4) + 4 (l / 32) + r % 4][l % 32] in element r of the accumulator.
typedef float mfma_acc16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ mfma_acc16 mfma_32x32x2(float a, float b, mfma_acc16 acc) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0); }
// nothing is scheduled across this point: keeps unrolled, mutually independent blocks from being interleaved into one
// register-hungry stream
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// value of lane ^ 32 (the other half of the wavefront)
__device__ __forceinline__ float wave_xor32_f(float v) { return __shfl_xor(v, 32); }

// ---- agent-scope memory operations: hand-offs between wavefronts INSIDE one launch (so101_chain.hpp) ------------------
// Per-XCD L2s are not coherent with each other and a CU's vector L1 is never refreshed by another CU's stores
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility").  Bytes that one wavefront hands
// to another within a launch are therefore written with `sc1` (write-through) stores, drained with s_waitcnt vmcnt(0)
// before the flag / queue slot / counter that announces them, and read with `sc1` loads (L1 bypassed) - every store and
// every load of those bytes, no exceptions.  The relaxed agent-scope atomics below compile to exactly those forms
// (global_load/store ... sc1); no cache-wide release / acquire fence is needed with this discipline.
template <class T>
__device__ __forceinline__ T ld_agent(const T* p) {
  static_assert(sizeof(T) == 4, "ld_agent: 4-byte types");
#ifdef SO101_CHAIN_PLAIN      // EXPERIMENT ONLY (wrong results): what the sc1 accesses cost
  return *(const volatile T*)p;
#endif
  int v = __hip_atomic_load((const int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_bit_cast(T, v);
}
template <class T>
__device__ __forceinline__ void st_agent(T* p, T v) {
  static_assert(sizeof(T) == 4, "st_agent: 4-byte types");
#ifdef SO101_CHAIN_PLAIN
  *(volatile T*)p = v; return;
#endif
  __hip_atomic_store((int*)p, __builtin_bit_cast(int, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_agent64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned char ld_agent8(const unsigned char* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent8(unsigned char* p, unsigned char v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every store this wavefront has issued has reached the coherence point (on gfx9 vmcnt counts stores as well as loads)
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned int atom_add_agent(unsigned int* p, unsigned int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned int atom_cas_agent(unsigned int* p, unsigned int expect, unsigned int desired) {
  __hip_atomic_compare_exchange_strong(p, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return expect;                       // the value found: == the caller's `expect` when the exchange happened
}
__device__ __forceinline__ void idle_sleep() { __builtin_amdgcn_s_sleep(32); }      // ~2 k cycles off the issue port
#endif  // SO101_WAVE_HPP_
