// Micro-benchmark: the Newton solver's Hessian assembly  H = sum_k J_k' Hc_k J_k  (18 x 18, lane k owns contact k) as
//   A  the product path's way: per-lane 6x6 products, then one cross-lane DPP sum per Hessian entry (so101_newton.hpp), and
//   B  with matrix cores: the lanes' Jacobian rows J (6 x 18) and W = Hc J staged through LDS into the operand layout of
//      v_mfma_f32_32x32x2_f32 (K = constraint row), 3 n MFMAs for n contacts, result rows back to lane a = row a.
// One wavefront per workgroup, 20 KB of LDS per workgroup and __launch_bounds__(64, 2) like k_pipe_solve (2 waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_hessian mfma_hessian.hip     Run: ./mfma_hessian
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../so101_sim_amd/csrc/wave.hpp"

#define NVS 18
typedef float floatx16 __attribute__((ext_vector_type(16)));

struct Con { float J[12][6]; float Hc[21]; int g0, g1; };

__device__ __forceinline__ int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

__device__ __forceinline__ void make_contact(Con& C, int lane, int ncon, unsigned seed) {
  unsigned s = seed * 747796405u + (unsigned)lane * 2891336453u + 1u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) * (1.f / 65536.f) - 0.5f; };
  bool on = lane < ncon;
#pragma unroll
  for (int c = 0; c < 12; c++)
#pragma unroll
    for (int j = 0; j < 6; j++) C.J[c][j] = on ? rnd() : 0.f;
  // symmetric positive block Hessian
#pragma unroll
  for (int k = 0; k < 21; k++) C.Hc[k] = on ? 0.1f * rnd() : 0.f;
#pragma unroll
  for (int i = 0; i < 6; i++) C.Hc[tri(i, i)] = on ? 1.f + rnd() : 0.f;
  int t = lane % 3;                                  // group pairs (0,1), (0,2), (1,2)
  C.g0 = t == 2 ? 1 : 0; C.g1 = t == 0 ? 1 : 2;
  if (!on) { C.g0 = -1; C.g1 = -1; }
}

// ---- A: the product path (general instance, <= 16 contacts: single-row sums)
__device__ __forceinline__ void hessian_dpp(const Con& C, int lane, float* h) {
  auto csum = [&](float v) -> float { return wave_sum_rows_f(v, true); };
#pragma unroll
  for (int b = 0; b < NVS; b++) h[b] = 0.f;
  // wave-uniform guards as in so101_newton.hpp (they also keep the sums in separate basic blocks: one giant block spills)
  bool actG[3], actX[3];
  bool on = C.g0 >= 0;
#pragma unroll
  for (int G = 0; G < 3; G++) actG[G] = __ballot(on && (C.g0 == G || C.g1 == G)) != 0ull;
  actX[0] = __ballot(on && C.g0 == 0 && C.g1 == 1) != 0ull;
  actX[1] = __ballot(on && C.g0 == 0 && C.g1 == 2) != 0ull;
  actX[2] = __ballot(on && C.g0 == 1 && C.g1 == 2) != 0ull;
#pragma unroll
  for (int b = 0; b < 6; b++) {
    float W0[6], W1[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int j = 0; j < 6; j++) { float hc = C.Hc[tri(i, j)]; s0 += hc * C.J[b][j]; s1 += hc * C.J[6 + b][j]; }
      W0[i] = s0; W1[i] = s1;
    }
#pragma unroll
    for (int a = 0; a <= b; a++) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int j = 0; j < 6; j++) { s0 += C.J[a][j] * W0[j]; s1 += C.J[6 + a][j] * W1[j]; }
#pragma unroll
      for (int G = 0; G < 3; G++) {
        if (actG[G]) {
          float tot = csum((C.g0 == G) ? s0 : ((C.g1 == G) ? s1 : 0.f));
          if (lane == 6 * G + a) h[6 * G + b] += tot;
          if (a != b && lane == 6 * G + b) h[6 * G + a] += tot;
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 6; a++) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 6; j++) s += C.J[a][j] * W1[j];
#pragma unroll
      for (int p = 0; p < 3; p++) {
        const int G = p == 2 ? 1 : 0, G2 = p == 0 ? 1 : 2;
        if (actX[p]) {
          float tot = csum((C.g0 == G && C.g1 == G2) ? s : 0.f);
          if (lane == 6 * G + a) h[6 * G2 + b] += tot;
          if (lane == 6 * G2 + b) h[6 * G + a] += tot;
        }
      }
    }
  }
}

// ---- B: matrix cores.  stage: [6 ncon][18] rows of J and of W = Hc J in LDS
#define ROWS 96
__device__ __forceinline__ void hessian_mfma(const Con& C, int lane, int ncon, float* Jl, float* Wl, float* h) {
  if (lane < ncon) {
    // rows 6 lane .. 6 lane + 5: zero the group this contact does not touch, then the two slots at their groups' columns
    int gz = 3 - C.g0 - C.g1;
#pragma unroll
    for (int j = 0; j < 6; j++) {
      float* jd = Jl + (6 * lane + j) * NVS; float* wd = Wl + (6 * lane + j) * NVS;
#pragma unroll
      for (int q = 0; q < 6; q++) {
        float w0 = 0.f, w1 = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) { float hc = C.Hc[tri(j, i)]; w0 += hc * C.J[q][i]; w1 += hc * C.J[6 + q][i]; }
        jd[6 * C.g0 + q] = C.J[q][j]; jd[6 * C.g1 + q] = C.J[6 + q][j]; jd[6 * gz + q] = 0.f;
        wd[6 * C.g0 + q] = w0; wd[6 * C.g1 + q] = w1; wd[6 * gz + q] = 0.f;
      }
    }
  }
  __syncthreads();
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  int col = lane & 31, half = lane >> 5;
  bool live = col < NVS;
  int off = half * NVS + (live ? col : 0);
  for (int t = 0; t < 3 * ncon; t++) {
    float a = Jl[off + 2 * NVS * t], b = Wl[off + 2 * NVS * t];
    a = live ? a : 0.f; b = live ? b : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  // lane l holds column l % 32, rows 8 blk + 4 (l / 32) + r % 4 in register r = 4 blk + r % 4; H is symmetric: row a = column a
  float other[8];
#pragma unroll
  for (int r = 0; r < 8; r++) other[r] = __shfl_xor(acc[r], 32);
#pragma unroll
  for (int r = 0; r < 4; r++) { h[r] = acc[r]; h[4 + r] = other[r]; h[8 + r] = acc[4 + r]; h[12 + r] = other[4 + r]; }
  h[16] = acc[8]; h[17] = acc[9];
  __syncthreads();
}

template <int MODE>
__global__ void __launch_bounds__(64, 2) k_bench(float* out, int ncon, int reps) {
  __shared__ float lds[5120];                 // 20 KB like EnvLDS; J rows at 0, W rows at ROWS * NVS
  int lane = threadIdx.x;
  Con C; make_contact(C, lane, ncon, blockIdx.x);
  float h[NVS], acc = 0.f, first[NVS];
#pragma unroll 1
  for (int r = 0; r < reps; r++) {
    if (MODE == 0) hessian_dpp(C, lane, h); else hessian_mfma(C, lane, ncon, lds, lds + ROWS * NVS, h);
    float t = 0.f;
#pragma unroll
    for (int b = 0; b < NVS; b++) { t += h[b]; if (r == 0) first[b] = h[b]; }
    acc += t;
    C.Hc[0] += 1e-6f * acc;                   // makes every assembly depend on the previous one
  }
  if (lane < NVS)
    for (int b = 0; b < NVS; b++) out[((size_t)blockIdx.x * NVS + lane) * NVS + b] = first[b] + (b == 0 ? 1e-20f * acc : 0.f);
}

int main() {
  const int blocks = 16384, reps = 200;
  float* out[2];
  for (int m = 0; m < 2; m++) hipMalloc(&out[m], sizeof(float) * blocks * NVS * NVS);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int ncon : {4, 9, 16}) {
    float ms[2];
    for (int m = 0; m < 2; m++) {
      for (int pass = 0; pass < 2; pass++) {
        hipEventRecord(e0);
        if (m == 0) hipLaunchKernelGGL(k_bench<0>, dim3(blocks), dim3(64), 0, 0, out[0], ncon, reps);
        else hipLaunchKernelGGL(k_bench<1>, dim3(blocks), dim3(64), 0, 0, out[1], ncon, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[m], e0, e1);
      }
    }
    std::vector<float> a((size_t)blocks * NVS * NVS), b(a.size());
    hipMemcpy(a.data(), out[0], a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), out[1], b.size() * 4, hipMemcpyDeviceToHost);
    double md = 0, mx = 0;
    for (size_t i = 0; i < a.size(); i++) { md = std::max(md, (double)fabsf(a[i] - b[i])); mx = std::max(mx, (double)fabsf(a[i])); }
    // wave-time of one Hessian: 2048 resident waves work through blocks x reps assemblies
    double us[2] = {ms[0] * 1e3 * 2048.0 / ((double)blocks * reps), ms[1] * 1e3 * 2048.0 / ((double)blocks * reps)};
    printf("ncon %2d: DPP sums %.2f us per Hessian and wavefront, MFMA %.2f us (x%.2f); max |difference| %.3g of max |H| %.3g\n", ncon, us[0], us[1], us[0] / us[1], md, mx);
  }
  return 0;
}
