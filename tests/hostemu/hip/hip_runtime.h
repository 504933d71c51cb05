// TEST HARNESS ONLY — a minimal stand-in for <hip/hip_runtime.h> so that the product's kernel source
// (so101_sim_amd/csrc/*.hpp, so101_hip.hip) can be compiled with g++ and executed on the CPU with one
// OS thread per lane.  Used by tests/ to debug kernel logic without a GPU; never shipped, never
// loaded by the product package (so101_sim_amd/native.py only loads libso101_hip.so).
#pragma once
#include <pthread.h>
#include <time.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct emu_idx { unsigned x, y, z; };
extern thread_local emu_idx threadIdx;
extern thread_local emu_idx blockIdx;
// Per-block context: barrier, cross-lane exchange buffers, "LDS".  Blocks normally run one after the other
// (emu_launch); the persistent scheduler kernel (k_chain) needs several blocks alive at once (emu_launch_concurrent),
// so nothing block-local may be a process-wide global.
struct EmuBlock {
  pthread_barrier_t barrier;
  pthread_barrier_t row_barrier[4];      // one per DPP row of 16 lanes: rows of a wave may diverge (k_narrow)
  float xchg_f[64];
  int xchg_i[64];
  float pv[64], px[64], py[64], pz[64];
  int pi[64];
  alignas(64) unsigned char lds[98304];  // BLOCK_SHARED storage of kernels that may run concurrently
};
extern thread_local EmuBlock* emu_blk;
#define emu_barrier (emu_blk->barrier)
#define emu_row_barrier (emu_blk->row_barrier)
#define emu_xchg_f (emu_blk->xchg_f)
#define emu_xchg_i (emu_blk->xchg_i)

inline void __syncthreads() { pthread_barrier_wait(&emu_barrier); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned int v) { return __builtin_popcount(v); }
using std::max;
using std::min;

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? 0 : 1; }
inline hipError_t hipFree(void* p) { free(p); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
// streams and events: launches are synchronous here, so these only have to exist
typedef void* hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return 0; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (void*)1; return 0; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)1; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (void*)1; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventQuery(hipEvent_t) { return 0; }
// graphs: capture is refused here, so101_step falls back to plain launches
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum { hipStreamCaptureModeThreadLocal = 1 };
inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return 1; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*) { return 1; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return 1; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 1; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return 0; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return 0; }
inline unsigned long long wall_clock64() {        // 100 MHz like the device's s_memrealtime
  timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return (unsigned long long)ts.tv_sec * 100000000ull + (unsigned long long)ts.tv_nsec / 10ull;
}
// vector types and bit casts the kernels use
struct float4 { float x, y, z, w; };
struct uint4 { unsigned int x, y, z, w; };
inline uint4 make_uint4(unsigned int x, unsigned int y, unsigned int z, unsigned int w) { return uint4{x, y, z, w}; }
inline unsigned int __float_as_uint(float f) { unsigned int u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned int u) { float f; memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
// v_readlane: value of lane `l`
inline int __builtin_amdgcn_readlane(int v, int l) { emu_xchg_i[threadIdx.x] = v; __syncthreads(); int r = emu_xchg_i[l]; __syncthreads(); return r; }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline hipError_t hipGetLastError() { return 0; }

inline EmuBlock* emu_block_new(unsigned threads) {
  EmuBlock* b = new EmuBlock();
  pthread_barrier_init(&b->barrier, nullptr, threads);
  if (threads == 64) for (int r = 0; r < 4; r++) pthread_barrier_init(&b->row_barrier[r], nullptr, 16);
  return b;
}
inline void emu_block_free(EmuBlock* b, unsigned threads) {
  pthread_barrier_destroy(&b->barrier);
  if (threads == 64) for (int r = 0; r < 4; r++) pthread_barrier_destroy(&b->row_barrier[r]);
  delete b;
}
template <typename K, typename... A>
void emu_launch(K kernel, dim3 grid, dim3 block, A... args) {
  for (unsigned b = 0; b < grid.x; b++) {
    EmuBlock* blk = emu_block_new(block.x);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < block.x; t++)
      th.emplace_back([=]() { threadIdx = {t, 0, 0}; blockIdx = {b, 0, 0}; emu_blk = blk; kernel(args...); });
    for (auto& x : th) x.join();
    emu_block_free(blk, block.x);
  }
}
// all blocks of the grid alive at once (persistent kernels whose blocks hand work to each other)
template <typename K, typename... A>
void emu_launch_concurrent(K kernel, dim3 grid, dim3 block, A... args) {
  std::vector<EmuBlock*> blks;
  std::vector<std::thread> th;
  for (unsigned b = 0; b < grid.x; b++) {
    EmuBlock* blk = emu_block_new(block.x);
    blks.push_back(blk);
    for (unsigned t = 0; t < block.x; t++)
      th.emplace_back([=]() { threadIdx = {t, 0, 0}; blockIdx = {b, 0, 0}; emu_blk = blk; kernel(args...); });
  }
  for (auto& x : th) x.join();
  for (EmuBlock* blk : blks) emu_block_free(blk, block.x);
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch(kernel, grid, block, __VA_ARGS__)
#define SO101_LAUNCH_CONCURRENT(kernel, grid, block, stream, ...) emu_launch_concurrent(kernel, grid, block, __VA_ARGS__)
