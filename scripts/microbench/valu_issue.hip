// How many cycles does one wave64 VALU instruction cost a SIMD of gfx950?  (Round 6: the VALU roofline of bench.py / the verdict's "27 % VALU issue"
// assume 2 cycles - 32 lanes per clock -; the datasheet's 157.3 TFLOP/s fp32 vector peak fits either 32 lanes x 1 or 16 lanes x packed 2.)
// A wavefront runs N independent v_fma_f32 (eight accumulators, no dependence closer than eight instructions) between two s_memtime reads;
// 1, 2 or 4 wavefronts share a SIMD (workgroups of 256 / 512 / 1024 threads, one workgroup per CU).  Cycles per instruction as each wavefront
// sees it:  16 lanes / clock -> 4, 8, 16;   32 lanes / clock with a 4-cycle issue limit per wavefront -> 4, 4, 8.
// The same for v_pk_fma_f32 (two fp32 per lane and instruction).
// Measured on MI355X (profiles/r06_valu_issue.txt): v_fma_f32 5.2 ticks per instruction for a lone wavefront and for two per SIMD (44.5 -> 102 TFLOP/s),
// 7.8-9.9 for four (103.5 TFLOP/s: saturated); v_pk_fma_f32 5.1 / 8.9 / 13-17 (104 / 120 / 125 TFLOP/s).  So: a wavefront issues one VALU instruction
// every ~5 cycles at best, TWO wavefronts per SIMD are needed to saturate plain fp32 (a SIMD retires a wave64 instruction in ~2.2 of its cycles:
// 32 lanes per clock), and the sustained plain-fma rate of the whole device is 0.81 T wave-instructions/s - 66 % of the 157.3 TFLOP/s datasheet
// peak (the shader clock under this load, from ticks and event time: 1.5 - 2.0 GHz over the six runs, not the 2.4 GHz the peak is quoted at).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue scripts/microbench/valu_issue.hip && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP 64
template <int PACKED>
__global__ void k_issue(unsigned long long* out, int iters, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  float b0 = seed, b1 = seed + 1, b2 = seed + 2, b3 = seed + 3, b4 = seed + 4, b5 = seed + 5, b6 = seed + 6, b7 = seed + 7;
  float m = 1.0000001f, c = 1e-9f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7}, pm = {m, m}, pc = {c, c};
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();       // s_memtime
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++) {
      if constexpr (PACKED) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));
      } else {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x + p0.y + p7.y;
  if (threadIdx.x % 64 == 0) {
    unsigned int hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    out[2 * w] = t1 - t0 + (s == 12345.678f ? 1 : 0);
    out[2 * w + 1] = hw;
  }
}

template <int PACKED>
void run(int threads, int iters) {
  int blocks = 256, waves = blocks * threads / 64;
  unsigned long long* d; hipMalloc(&d, sizeof(unsigned long long) * 2 * waves);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_issue<PACKED>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.f);        // warm-up
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_issue<PACKED>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * waves); hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost);
  std::vector<double> cpi;
  for (int w = 0; w < waves; w++) cpi.push_back((double)h[2 * w] / ((double)iters * REP));
  std::sort(cpi.begin(), cpi.end());
  double n_inst = (double)waves * iters * REP;
  // whole-device rate: wave64 instructions per second and what that is in fp32 FMA TFLOP/s (x 64 lanes x 2 flop, x 2 when packed)
  double rate = n_inst / (ms * 1e-3);
  printf("%-13s %4d threads/workgroup (%d wavefronts per SIMD): s_memtime ticks per instruction as a wavefront sees it: median %.2f (p10 %.2f, p90 %.2f); kernel %.3f ms -> %.1f G wave-instructions/s = %.1f TFLOP/s\n",
         PACKED ? "v_pk_fma_f32" : "v_fma_f32", threads, threads / 256, cpi[waves / 2], cpi[waves / 10], cpi[waves * 9 / 10], ms, rate * 1e-9, rate * 64 * 2 * (PACKED ? 2 : 1) * 1e-12);
  hipFree(d);
}

int main() {
  for (int t : {256, 512, 1024}) run<0>(t, 4000);
  for (int t : {256, 512, 1024}) run<1>(t, 4000);
  return 0;
}
