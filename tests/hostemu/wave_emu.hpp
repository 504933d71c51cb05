// TEST HARNESS ONLY — lane-thread emulation of so101_sim_amd/csrc/wave.hpp (pre-included with -include so
// that the product header's include guard skips the gfx950 implementation).
#ifndef SO101_WAVE_HPP_
#define SO101_WAVE_HPP_
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sched.h>
#define WAVE 64
#define SO101_EMU 1
// block-shared storage of kernels whose blocks may be alive concurrently under emulation (k_chain)
#define BLOCK_SHARED(T, name) static_assert(sizeof(T) <= sizeof(EmuBlock::lds), "LDS"); T& name = *reinterpret_cast<T*>(emu_blk->lds)
inline int wave_lane() { return threadIdx.x; }
inline void wave_sync() { __syncthreads(); }
inline void wave_lds_sync() { __syncthreads(); }
inline float wave_max_f(float v) {
  emu_xchg_f[threadIdx.x] = v; __syncthreads();
  float m = emu_xchg_f[0]; for (int i = 1; i < 64; i++) m = std::fmax(m, emu_xchg_f[i]);
  __syncthreads(); return m;
}
inline float wave_sum_f(float v) {
  emu_xchg_f[threadIdx.x] = v; __syncthreads();
  // same order as the device version so that rounding matches: xor 1, xor 2, half-row mirror, row mirror inside each
  // row of 16, then (r0 + r1) + (r2 + r3)
  float t[64], u[64];
  for (int i = 0; i < 64; i++) t[i] = emu_xchg_f[i];
  for (int i = 0; i < 64; i++) u[i] = t[i] + t[i ^ 1];
  for (int i = 0; i < 64; i++) t[i] = u[i] + u[i ^ 2];
  for (int i = 0; i < 64; i++) u[i] = t[i] + t[(i & ~7) | (7 - (i & 7))];
  for (int i = 0; i < 64; i++) t[i] = u[i] + u[(i & ~15) | (15 - (i & 15))];
  float r = (t[0] + t[16]) + (t[32] + t[48]);
  __syncthreads(); return r;
}
inline float wave_sum_rows_f(float v, bool row0_only) {
  emu_xchg_f[threadIdx.x] = v; __syncthreads();
  float t[64], u[64];
  for (int i = 0; i < 64; i++) t[i] = emu_xchg_f[i];
  for (int i = 0; i < 64; i++) u[i] = t[i] + t[i ^ 1];
  for (int i = 0; i < 64; i++) t[i] = u[i] + u[i ^ 2];
  for (int i = 0; i < 64; i++) u[i] = t[i] + t[(i & ~7) | (7 - (i & 7))];
  for (int i = 0; i < 64; i++) t[i] = u[i] + u[(i & ~15) | (15 - (i & 15))];
  float full = (t[0] + t[16]) + (t[32] + t[48]), r = row0_only ? t[0] + 0.f : full;
  // the shortcut is only legal when it changes nothing: checked on every call under emulation
  if (row0_only && std::memcmp(&r, &full, sizeof r) != 0) { std::fprintf(stderr, "wave_sum_rows_f: row-0 shortcut differs from the full sum\n"); std::abort(); }
  __syncthreads(); return r;
}
inline unsigned long long wave_ballot(bool p) {
  emu_xchg_i[threadIdx.x] = p; __syncthreads();
  unsigned long long m = 0; for (int i = 0; i < 64; i++) if (emu_xchg_i[i]) m |= 1ull << i;
  __syncthreads(); return m;
}
inline int wave_prefix(unsigned long long mask) { return __builtin_popcountll(mask & ((1ull << threadIdx.x) - 1ull)); }
// readfirstlane: every lane gets lane 0's value
inline int wave_uniform_i(int v) { emu_xchg_i[threadIdx.x] = v; __syncthreads(); int r = emu_xchg_i[0]; __syncthreads(); return r; }
template <class T> inline T ldc(const T* p) { return *p; }
template <class T> inline T ldc_obj(const T* p) { return *p; }
#define LAUNDER_UNIFORM(p) (void)(p)
template <class T> inline T* uniform_ptr(T* p) { return p; }
#define SO101_NOINLINE
inline float wave_get_f(float v, int src) { emu_xchg_f[threadIdx.x] = v; __syncthreads(); float r = emu_xchg_f[src]; __syncthreads(); return r; }
inline float wave_bcast_f(float v, int src) { emu_xchg_f[threadIdx.x] = v; __syncthreads(); float r = emu_xchg_f[src]; __syncthreads(); return r; }
inline int wave_get_i(int v, int src) { emu_xchg_i[threadIdx.x] = v; __syncthreads(); int r = emu_xchg_i[src]; __syncthreads(); return r; }
inline int wave_bcast_i(int v, int src) { emu_xchg_i[threadIdx.x] = v; __syncthreads(); int r = emu_xchg_i[src]; __syncthreads(); return r; }
inline void wave_argmax(float& val, int& idx) {
  emu_xchg_f[threadIdx.x] = val; emu_xchg_i[threadIdx.x] = idx; __syncthreads();
  float m = emu_xchg_f[0]; for (int i = 1; i < 64; i++) m = std::fmax(m, emu_xchg_f[i]);
  int b = 0x7fffffff; for (int i = 0; i < 64; i++) if (emu_xchg_f[i] == m) b = std::min(b, emu_xchg_i[i]);
  __syncthreads(); val = m; idx = b;
}
inline int wave_min_i(int v) {
  emu_xchg_i[threadIdx.x] = v; __syncthreads();
  int m = emu_xchg_i[0]; for (int i = 1; i < 64; i++) m = std::min(m, emu_xchg_i[i]);
  __syncthreads(); return m;
}
inline void wave_argmax3(float& val, int& idx, float& x, float& y, float& z) {
  float *px = emu_blk->px, *py = emu_blk->py, *pz = emu_blk->pz;
  emu_xchg_f[threadIdx.x] = val; emu_xchg_i[threadIdx.x] = idx; px[threadIdx.x] = x; py[threadIdx.x] = y; pz[threadIdx.x] = z;
  __syncthreads();
  float m = emu_xchg_f[0]; for (int i = 1; i < 64; i++) m = std::fmax(m, emu_xchg_f[i]);
  int b = 0x7fffffff, src = 0;
  for (int i = 0; i < 64; i++) if (emu_xchg_f[i] == m) b = std::min(b, emu_xchg_i[i]);
  bool found = false;
  for (int i = 0; i < 64 && !found; i++) if (((emu_xchg_f[i] == m) ? emu_xchg_i[i] : 0x7fffffff) == b) { src = i; found = true; }
  float ox = px[src], oy = py[src], oz = pz[src];
  __syncthreads();
  val = m; idx = b; x = ox; y = oy; z = oz;
}
// row-local argmax with payload: the 16 lanes of a DPP row agree on (max value, smallest index, winner's payload).
// Rows of a wave may diverge (k_narrow runs one geom pair per row), so the exchange synchronises the row only.
inline void row_argmax3(float& val, int& idx, float& x, float& y, float& z) {
  float *pv = emu_blk->pv, *px = emu_blk->px, *py = emu_blk->py, *pz = emu_blk->pz;
  int* pi = emu_blk->pi;
  int t = threadIdx.x, row = t >> 4, r0 = t & ~15, src = r0;
  pv[t] = val; pi[t] = idx; px[t] = x; py[t] = y; pz[t] = z;
  pthread_barrier_wait(&emu_row_barrier[row]);
  for (int i = r0 + 1; i < r0 + 16; i++)
    if (pv[i] > pv[src] || (pv[i] == pv[src] && pi[i] < pi[src])) src = i;
  float ov = pv[src], ox = px[src], oy = py[src], oz = pz[src]; int oi = pi[src];
  pthread_barrier_wait(&emu_row_barrier[row]);
  val = ov; idx = oi; x = ox; y = oy; z = oz;
}
// matrix cores (wave.hpp): D += A (32 x 2) * B (2 x 32) in the lane layout of v_mfma_f32_32x32x2_f32
struct mfma_acc16 { float v[16]; float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };
inline mfma_acc16 mfma_32x32x2(float a, float b, mfma_acc16 acc) {
  int l = threadIdx.x;
  emu_blk->px[l] = a; emu_blk->py[l] = b; __syncthreads();
  for (int r = 0; r < 16; r++) {
    int i = 8 * (r / 4) + 4 * (l / 32) + r % 4, j = l % 32;
    acc.v[r] = std::fma(emu_blk->px[i + 32], emu_blk->py[j + 32], std::fma(emu_blk->px[i], emu_blk->py[j], acc.v[r]));
  }
  __syncthreads();
  return acc;
}
#define SCHED_FENCE() (void)0
inline float wave_xor32_f(float v) { emu_xchg_f[threadIdx.x] = v; __syncthreads(); float r = emu_xchg_f[threadIdx.x ^ 32]; __syncthreads(); return r; }
// agent-scope memory operations (wave.hpp): sequentially consistent host atomics
template <class T> inline T ld_agent(const T* p) { int v = __atomic_load_n((const int*)p, __ATOMIC_SEQ_CST); T o; std::memcpy(&o, &v, 4); return o; }
template <class T> inline void st_agent(T* p, T v) { int x; std::memcpy(&x, &v, 4); __atomic_store_n((int*)p, x, __ATOMIC_SEQ_CST); }
inline unsigned long long ld_agent64(const unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
inline void st_agent64(unsigned long long* p, unsigned long long v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
inline unsigned char ld_agent8(const unsigned char* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
inline void st_agent8(unsigned char* p, unsigned char v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
inline void drain_stores() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline unsigned int atom_add_agent(unsigned int* p, unsigned int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned int atom_cas_agent(unsigned int* p, unsigned int expect, unsigned int desired) {
  __atomic_compare_exchange_n(p, &expect, desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return expect;
}
inline void idle_sleep() { sched_yield(); }
#endif
