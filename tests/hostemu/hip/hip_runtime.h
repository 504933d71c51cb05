// TEST HARNESS ONLY — a minimal stand-in for <hip/hip_runtime.h> so that the product's kernel source
// (so101_sim_amd/csrc/*.hpp, so101_hip.hip) can be compiled with g++ and executed on the CPU with one
// OS thread per lane.  Used by tests/ to debug kernel logic without a GPU; never shipped, never
// loaded by the product package (so101_sim_amd/native.py only loads libso101_hip.so).
#pragma once
#include <pthread.h>
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
extern pthread_barrier_t emu_barrier;
extern pthread_barrier_t emu_row_barrier[4];      // one per DPP row of 16 lanes: rows of a wave may diverge (k_narrow)
extern float emu_xchg_f[64];
extern int emu_xchg_i[64];
extern unsigned long long emu_xchg_u;

inline void __syncthreads() { pthread_barrier_wait(&emu_barrier); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned int v) { return __builtin_popcount(v); }
using std::max;
using std::min;

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToDevice = 3 };
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
inline unsigned long long wall_clock64() { return 0ull; }
// v_readlane: value of lane `l`
inline int __builtin_amdgcn_readlane(int v, int l) { emu_xchg_i[threadIdx.x] = v; __syncthreads(); int r = emu_xchg_i[l]; __syncthreads(); return r; }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline hipError_t hipGetLastError() { return 0; }

template <typename K, typename... A>
void emu_launch(K kernel, dim3 grid, dim3 block, A... args) {
  for (unsigned b = 0; b < grid.x; b++) {
    pthread_barrier_init(&emu_barrier, nullptr, block.x);
    if (block.x == 64) for (int r = 0; r < 4; r++) pthread_barrier_init(&emu_row_barrier[r], nullptr, 16);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < block.x; t++)
      th.emplace_back([=]() { threadIdx = {t, 0, 0}; blockIdx = {b, 0, 0}; kernel(args...); });
    for (auto& x : th) x.join();
    pthread_barrier_destroy(&emu_barrier);
    if (block.x == 64) for (int r = 0; r < 4; r++) pthread_barrier_destroy(&emu_row_barrier[r]);
  }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch(kernel, grid, block, __VA_ARGS__)
