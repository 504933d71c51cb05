// Device code of the batched SO100 hand-over step: ONE ENVIRONMENT PER WAVEFRONT (64 lanes).
//
// Stages per physics substep (MuJoCo mj_step order; the reference drives 10 of them per env.step,
// so101_sim/task_suite.py:41):
//   kinematics -> CRBA (arm 6x6) + inverse -> RNE bias -> actuation -> smooth acceleration
//   -> geom AABBs -> broadphase over the statically filtered pair list -> MPR narrowphase with
//   wave-parallel hull support -> constraint rows (dof frictionloss, joint limits, elliptic contacts)
//   -> PGS in velocity space -> semi-implicit Euler.
// Lane use: "uniform" code is executed identically by all lanes on wave-uniform values; "lane-
// parallel" code maps lanes to dofs / geoms / pairs / contacts / hull vertices.  All cross-lane data
// flows through LDS (EnvLDS) between wave_sync() points or through the helpers in wave.hpp.
//
// Free bodies are carried in centre-of-mass twist coordinates inside the solver (inverse inertia is
// then 1/m and a symmetric 3x3), and mapped back to MuJoCo's (origin velocity, body-frame angular
// velocity) coordinates for integration; the PGS force iterates are invariant to that change of
// velocity coordinates.
#pragma once
#include "so101_model.hpp"
#include "wave.hpp"

#define DEV __device__ __forceinline__
// Stage clocks (100 MHz s_memrealtime ticks) for scripts/gpu_*.py: compiled in only with -DSO101_DEBUG_CLOCKS
// (python -m so101_sim_amd.build --clocks).  Production builds read the clock twice per solve (the scheduling hint
// of k_order) and nowhere else.
#ifdef SO101_DEBUG_CLOCKS
#define SO101_CLOCKS_ON 1
#define SO101_CLOCK() wall_clock64()
#else
#define SO101_CLOCKS_ON 0
#define SO101_CLOCK() 0ull
#endif
#ifndef SO101_COLLINEAR_REL
#define SO101_COLLINEAR_REL 1e-3f      // mpr_penetration: relative bound of the "origin on the v0-v1 segment" test, sin(angle) (kernel experiments: -DSO101_COLLINEAR_REL=...)
#endif
#define MINVAL_F 1e-15f
#define MINIMP_F 1e-4f
#define MAXIMP_F 0.9999f
#define EPS_F 1.1920929e-7f

// ------------------------------------------------------------------ small math
DEV float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
DEV void cross3(float* o, const float* a, const float* b) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
DEV float normalize3(float* a) {
  float n = sqrtf(dot3(a, a));
  if (n < MINVAL_F) { a[0] = 1.f; a[1] = 0.f; a[2] = 0.f; return 0.f; }
  float inv = 1.f / n;
  a[0] *= inv; a[1] *= inv; a[2] *= inv;
  return n;
}
DEV void matvec3(float* o, const float* m, const float* v) {
  float x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  float y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  float z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
DEV void matTvec3(float* o, const float* m, const float* v) {
  float x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  float y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  float z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
DEV void matmul3(float* o, const float* a, const float* b) {
  float t[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
#pragma unroll
  for (int i = 0; i < 9; i++) o[i] = t[i];
}
DEV void quat2mat(float* m, const float* q) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = 1.f - 2.f * (y * y + z * z); m[1] = 2.f * (x * y - w * z); m[2] = 2.f * (x * z + w * y);
  m[3] = 2.f * (x * y + w * z); m[4] = 1.f - 2.f * (x * x + z * z); m[5] = 2.f * (y * z - w * x);
  m[6] = 2.f * (x * z - w * y); m[7] = 2.f * (y * z + w * x); m[8] = 1.f - 2.f * (x * x + y * y);
}
DEV void mulquat(float* o, const float* a, const float* b) {
  float w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  float x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  float y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  float z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
DEV void normquat(float* q) {
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL_F) { q[0] = 1.f; q[1] = q[2] = q[3] = 0.f; return; }
  float inv = 1.f / n;
  q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}
// sin and cos of one angle: Cody-Waite reduction by pi/2 (three constants, exact products through fma) and the
// cephes minimax polynomials on [-pi/4, pi/4]; <= 2 ulp for |x| < 1e4 rad.  The libm sinf/cosf expand to ~220
// instructions each (large-argument path), and the kinematics needs six pairs per substep.
DEV void sincos_f(float x, float* sn, float* cs) {
  float k = rintf(x * 0.63661977236758134308f);
  float r = fmaf(-k, 1.57079625129699707031f, x);
  r = fmaf(-k, 7.54978941586159635335e-08f, r);
  r = fmaf(-k, 5.39030285815811905290e-15f, r);
  float z = r * r;
  float s = fmaf(r * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  float c = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), fmaf(-0.5f, z, 1.f));
  int q = (int)k;
  float s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
}
DEV void rotvecquat(float* o, const float* v, const float* q) {
  float m[9]; quat2mat(m, q); matvec3(o, m, v);
}
DEV void mat2quat(float* q, const float* m) {
  float t = m[0] + m[4] + m[8];
  if (t > 0.f) {
    float s = sqrtf(t + 1.f) * 2.f; q[0] = 0.25f * s; q[1] = (m[7] - m[5]) / s; q[2] = (m[2] - m[6]) / s; q[3] = (m[3] - m[1]) / s;
  } else if (m[0] > m[4] && m[0] > m[8]) {
    float s = sqrtf(1.f + m[0] - m[4] - m[8]) * 2.f; q[0] = (m[7] - m[5]) / s; q[1] = 0.25f * s; q[2] = (m[1] + m[3]) / s; q[3] = (m[2] + m[6]) / s;
  } else if (m[4] > m[8]) {
    float s = sqrtf(1.f + m[4] - m[0] - m[8]) * 2.f; q[0] = (m[2] - m[6]) / s; q[1] = (m[1] + m[3]) / s; q[2] = 0.25f * s; q[3] = (m[5] + m[7]) / s;
  } else {
    float s = sqrtf(1.f + m[8] - m[0] - m[4]) * 2.f; q[0] = (m[3] - m[1]) / s; q[1] = (m[2] + m[6]) / s; q[2] = (m[5] + m[7]) / s; q[3] = 0.25f * s;
  }
  normquat(q);
}
// symmetric 3x3 packed as xx yy zz xy xz yz
DEV void symvec3(float* o, const float* s, const float* v) {
  float x = s[0] * v[0] + s[3] * v[1] + s[4] * v[2];
  float y = s[3] * v[0] + s[1] * v[1] + s[5] * v[2];
  float z = s[4] * v[0] + s[5] * v[1] + s[2] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
// o = R * S * R^T for symmetric S
DEV void rotsym(float* o, const float* R, const float* s) {
  float t[9];   // t = R*S
#pragma unroll
  for (int i = 0; i < 3; i++) {
    t[3 * i + 0] = R[3 * i] * s[0] + R[3 * i + 1] * s[3] + R[3 * i + 2] * s[4];
    t[3 * i + 1] = R[3 * i] * s[3] + R[3 * i + 1] * s[1] + R[3 * i + 2] * s[5];
    t[3 * i + 2] = R[3 * i] * s[4] + R[3 * i + 1] * s[5] + R[3 * i + 2] * s[2];
  }
  o[0] = t[0] * R[0] + t[1] * R[1] + t[2] * R[2];
  o[1] = t[3] * R[3] + t[4] * R[4] + t[5] * R[5];
  o[2] = t[6] * R[6] + t[7] * R[7] + t[8] * R[8];
  o[3] = t[0] * R[3] + t[1] * R[4] + t[2] * R[5];
  o[4] = t[0] * R[6] + t[1] * R[7] + t[2] * R[8];
  o[5] = t[3] * R[6] + t[4] * R[7] + t[5] * R[8];
}
DEV int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// world inertia of every dynamic body (lane-parallel)
DEV void world_inertias(const DevModel* m, EnvLDS& L) {
  int lane = wave_lane();
  if (lane < NDYN) {
    const float* Ib = lane < NARM ? m->arm_Ib[lane] : m->free_Ib[lane - NARM];
    float sc = lane < NARM ? 1.f : L.fscale[lane - NARM];      // per-env prop mass scale: mass and inertia scale together
    float o[6]; rotsym(o, L.xmat[lane], Ib);
#pragma unroll
    for (int i = 0; i < 6; i++) L.Iw[lane][i] = sc * o[i];
    if (lane >= NARM) {
      int f = lane - NARM;
      float oi[6]; rotsym(oi, L.xmat[lane], m->free_Ibinv[f]);
      float isc = 1.f / sc;
#pragma unroll
      for (int i = 0; i < 6; i++) L.fIinv[f][i] = isc * oi[i];
      L.fmass[f] = sc * m->free_mass[f];
      L.fminv[f] = 1.f / L.fmass[f];
    }
  }
  wave_sync();
}

// ------------------------------------------------------------------ kinematics (uniform)
DEV void kinematics(const DevModel* m, EnvLDS& L) {
  int lane = wave_lane();
  // arm chain: every lane walks the 6 links on uniform values; lane 0 publishes to LDS
  float xp[3] = {m->base_pos[0], m->base_pos[1], m->base_pos[2]};
  float xq[4] = {m->base_quat[0], m->base_quat[1], m->base_quat[2], m->base_quat[3]};
  float R[9]; quat2mat(R, xq);
#pragma unroll
  for (int k = 0; k < NARM; k++) {
    float t[3]; matvec3(t, R, m->arm_pos[k]);
    xp[0] += t[0]; xp[1] += t[1]; xp[2] += t[2];
    mulquat(xq, xq, m->arm_quat[k]);
    float sn, cs; sincos_f(0.5f * L.qpos[k], &sn, &cs);
    float jq[4] = {cs, m->arm_axis[k][0] * sn, m->arm_axis[k][1] * sn, m->arm_axis[k][2] * sn};
    mulquat(xq, xq, jq);
    normquat(xq);
    quat2mat(R, xq);
    float ax[3]; matvec3(ax, R, m->arm_axis[k]);
    float ip[3]; matvec3(ip, R, m->arm_ipos[k]);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 3; i++) { L.xpos[k][i] = xp[i]; L.axis[k][i] = ax[i]; L.xipos[k][i] = xp[i] + ip[i]; }
#pragma unroll
      for (int i = 0; i < 9; i++) L.xmat[k][i] = R[i];
    }
  }
  // free bodies: lanes 0..NFREE-1
  if (lane < NFREE) {
    int f = lane, b = NARM + f;
    const float* q = &L.qpos[NARM + 7 * f];
    float fq[4] = {q[3], q[4], q[5], q[6]};
    normquat(fq);
    float Rf[9]; quat2mat(Rf, fq);
    float ip[3]; matvec3(ip, Rf, m->free_ipos[f]);
#pragma unroll
    for (int i = 0; i < 3; i++) { L.xpos[b][i] = q[i]; L.xipos[b][i] = q[i] + ip[i]; }
#pragma unroll
    for (int i = 0; i < 9; i++) L.xmat[b][i] = Rf[i];
  }
  wave_sync();
  world_inertias(m, L);
}

// Kinematics of a state whose body poses are already known (the pipelined step publishes them for the narrowphase at
// the end of the previous kernel): poses from global memory, then the derived quantities one body per lane - joint axes,
// COM positions, world inertias - with the same expressions kinematics() uses, so the bits are the same.  Replaces the
// serial walk down the arm (every lane on identical values, ~900 dependent instructions) by ~50.
template <bool AG = false>       // AG: the poses were stored by another wavefront of this launch (wave.hpp, agent-scope loads)
DEV void kinematics_from_pose(const DevModel* m, EnvLDS& L, const float* pose /* [NDYN][12]: xpos, xmat */) {
  int lane = wave_lane();
  for (int i = lane; i < NDYN * 12; i += WAVE) {
    int b = i / 12, j = i % 12;
    float v = AG ? ld_agent(&pose[i]) : pose[i];
    if (j < 3) L.xpos[b][j] = v; else L.xmat[b][j - 3] = v;
  }
  wave_sync();
  if (lane < NARM) {
    int k = lane;
    float R[9];
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = L.xmat[k][i];
    float ax[3]; matvec3(ax, R, m->arm_axis[k]);
    float ip[3]; matvec3(ip, R, m->arm_ipos[k]);
#pragma unroll
    for (int i = 0; i < 3; i++) { L.axis[k][i] = ax[i]; L.xipos[k][i] = L.xpos[k][i] + ip[i]; }
  } else if (lane < NDYN) {
    int f = lane - NARM, b = lane;
    float Rf[9];
#pragma unroll
    for (int i = 0; i < 9; i++) Rf[i] = L.xmat[b][i];
    float ip[3]; matvec3(ip, Rf, m->free_ipos[f]);
#pragma unroll
    for (int i = 0; i < 3; i++) L.xipos[b][i] = L.xpos[b][i] + ip[i];
  }
  wave_sync();
  world_inertias(m, L);
}

// ------------------------------------------------------------------ CRBA + inverse of the arm block
DEV void crba_arm(const DevModel* m, EnvLDS& L) {
  int lane = wave_lane();
  // composite (mass, COM, inertia about COM) of the sub-chain k..5 — uniform backward pass
  float mc[NARM], Cc[NARM][3], Ic[NARM][6];
  float cm = 0.f, cC[3] = {0.f, 0.f, 0.f}, cI[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = NARM - 1; k >= 0; k--) {
    float m1 = m->arm_mass[k], mt = m1 + cm;
    float c1[3] = {L.xipos[k][0], L.xipos[k][1], L.xipos[k][2]};
    float C[3];
#pragma unroll
    for (int i = 0; i < 3; i++) C[i] = (m1 * c1[i] + cm * cC[i]) / mt;
    float d1[3] = {c1[0] - C[0], c1[1] - C[1], c1[2] - C[2]};
    float d2[3] = {cC[0] - C[0], cC[1] - C[1], cC[2] - C[2]};
    float dd1 = dot3(d1, d1), dd2 = dot3(d2, d2);
    float I[6];
    I[0] = L.Iw[k][0] + cI[0] + m1 * (dd1 - d1[0] * d1[0]) + cm * (dd2 - d2[0] * d2[0]);
    I[1] = L.Iw[k][1] + cI[1] + m1 * (dd1 - d1[1] * d1[1]) + cm * (dd2 - d2[1] * d2[1]);
    I[2] = L.Iw[k][2] + cI[2] + m1 * (dd1 - d1[2] * d1[2]) + cm * (dd2 - d2[2] * d2[2]);
    I[3] = L.Iw[k][3] + cI[3] - m1 * d1[0] * d1[1] - cm * d2[0] * d2[1];
    I[4] = L.Iw[k][4] + cI[4] - m1 * d1[0] * d1[2] - cm * d2[0] * d2[2];
    I[5] = L.Iw[k][5] + cI[5] - m1 * d1[1] * d1[2] - cm * d2[1] * d2[2];
    cm = mt;
#pragma unroll
    for (int i = 0; i < 3; i++) cC[i] = C[i];
#pragma unroll
    for (int i = 0; i < 6; i++) cI[i] = I[i];
    mc[k] = cm;
#pragma unroll
    for (int i = 0; i < 3; i++) Cc[k][i] = cC[i];
#pragma unroll
    for (int i = 0; i < 6; i++) Ic[k][i] = cI[i];
  }
  // M[j][k], k <= j: lane e computes entry e of the packed lower triangle (21 entries) from the composites, which lane 0
  // parks in LDS (the Newton scratch is idle here).  Every lane used to compute all 21 entries on identical values.
  float* park = &L.nw.H[0][0];                       // [NARM][10]: mass, COM, inertia of sub-chain j..5
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < NARM; j++) {
      park[10 * j] = mc[j];
#pragma unroll
      for (int i = 0; i < 3; i++) park[10 * j + 1 + i] = Cc[j][i];
#pragma unroll
      for (int i = 0; i < 6; i++) park[10 * j + 4 + i] = Ic[j][i];
    }
  }
  wave_sync();
  if (lane < NARM * (NARM + 1) / 2) {
    int j = 0;
#pragma unroll
    for (int t = 1; t < NARM; t++) j += lane >= t * (t + 1) / 2 ? 1 : 0;
    int k = lane - j * (j + 1) / 2;
    const float* pj = park + 10 * j;
    float mcj = pj[0], Ccj[3] = {pj[1], pj[2], pj[3]}, Icj[6] = {pj[4], pj[5], pj[6], pj[7], pj[8], pj[9]};
    float ak[3] = {L.axis[k][0], L.axis[k][1], L.axis[k][2]};
    float aj[3] = {L.axis[j][0], L.axis[j][1], L.axis[j][2]};
    float rk[3] = {Ccj[0] - L.xpos[k][0], Ccj[1] - L.xpos[k][1], Ccj[2] - L.xpos[k][2]};
    float rj[3] = {Ccj[0] - L.xpos[j][0], Ccj[1] - L.xpos[j][1], Ccj[2] - L.xpos[j][2]};
    float hl[3]; cross3(hl, ak, rk);
    hl[0] *= mcj; hl[1] *= mcj; hl[2] *= mcj;
    float ha[3]; symvec3(ha, Icj, ak);
    float t3[3]; cross3(t3, rj, hl);
    float v = aj[0] * (ha[0] + t3[0]) + aj[1] * (ha[1] + t3[1]) + aj[2] * (ha[2] + t3[2]);
    if (j == k) v += m->armature[j];
    L.Marm[j][k] = v; L.Marm[k][j] = v;
  }
  wave_sync();
  float Mfull[NARM][NARM];
#pragma unroll
  for (int j = 0; j < NARM; j++)
#pragma unroll
    for (int k = 0; k <= j; k++) { float v = L.Marm[j][k]; Mfull[j][k] = v; Mfull[k][j] = v; }
  // Cholesky (uniform) then lane c solves for column c of the inverse
  float Lc[NARM][NARM];
#pragma unroll
  for (int j = 0; j < NARM; j++) {
    float d = Mfull[j][j];
#pragma unroll
    for (int k = 0; k < j; k++) d -= Lc[j][k] * Lc[j][k];
    d = sqrtf(d);
    Lc[j][j] = d;
    float inv = 1.f / d;
#pragma unroll
    for (int i = j + 1; i < NARM; i++) {
      float v = Mfull[i][j];
#pragma unroll
      for (int k = 0; k < j; k++) v -= Lc[i][k] * Lc[j][k];
      Lc[i][j] = v * inv;
    }
  }
  if (lane < NARM) {
    float y[NARM], x[NARM];
#pragma unroll
    for (int i = 0; i < NARM; i++) {
      float v = (i == lane) ? 1.f : 0.f;
#pragma unroll
      for (int k = 0; k < i; k++) v -= Lc[i][k] * y[k];
      y[i] = v / Lc[i][i];
    }
#pragma unroll
    for (int i = NARM - 1; i >= 0; i--) {
      float v = y[i];
#pragma unroll
      for (int k = i + 1; k < NARM; k++) v -= Lc[k][i] * x[k];
      x[i] = v / Lc[i][i];
    }
#pragma unroll
    for (int i = 0; i < NARM; i++) L.Minv[i][lane] = x[i];
  }
  wave_sync();
}

// ------------------------------------------------------------------ RNE bias, actuation, smooth acceleration
DEV void smooth_dynamics(const DevModel* m, EnvLDS& L) {
  int lane = wave_lane();
  // --- arm: recursive Newton-Euler with zero joint acceleration and gravity as base acceleration
  float w[3] = {0.f, 0.f, 0.f}, al[3] = {0.f, 0.f, 0.f}, ao[3] = {-m->grav[0], -m->grav[1], -m->grav[2]};
  float pp[3] = {m->base_pos[0], m->base_pos[1], m->base_pos[2]};
  float fk[NARM][3], nk[NARM][3];
#pragma unroll
  for (int k = 0; k < NARM; k++) {
    float d[3] = {L.xpos[k][0] - pp[0], L.xpos[k][1] - pp[1], L.xpos[k][2] - pp[2]};
    float t1[3], t2[3];
    cross3(t1, al, d); cross3(t2, w, d); cross3(t2, w, t2);
    ao[0] += t1[0] + t2[0]; ao[1] += t1[1] + t2[1]; ao[2] += t1[2] + t2[2];
    float a[3] = {L.axis[k][0], L.axis[k][1], L.axis[k][2]};
    float qd = L.qvel[k];
    float wa[3]; cross3(wa, w, a);
    al[0] += wa[0] * qd; al[1] += wa[1] * qd; al[2] += wa[2] * qd;
    w[0] += a[0] * qd; w[1] += a[1] * qd; w[2] += a[2] * qd;
    float r[3] = {L.xipos[k][0] - L.xpos[k][0], L.xipos[k][1] - L.xpos[k][1], L.xipos[k][2] - L.xpos[k][2]};
    cross3(t1, al, r); cross3(t2, w, r); cross3(t2, w, t2);
    float mass = m->arm_mass[k];
    float F[3] = {mass * (ao[0] + t1[0] + t2[0]), mass * (ao[1] + t1[1] + t2[1]), mass * (ao[2] + t1[2] + t2[2])};
    float Iw_[3], Ia[3], N[3], rF[3];
    symvec3(Iw_, L.Iw[k], w); symvec3(Ia, L.Iw[k], al);
    cross3(N, w, Iw_); cross3(rF, r, F);
#pragma unroll
    for (int i = 0; i < 3; i++) { fk[k][i] = F[i]; nk[k][i] = Ia[i] + N[i] + rF[i]; pp[i] = L.xpos[k][i]; }
  }
  float bias[NARM];
#pragma unroll
  for (int k = NARM - 1; k >= 0; k--) {
    bias[k] = L.axis[k][0] * nk[k][0] + L.axis[k][1] * nk[k][1] + L.axis[k][2] * nk[k][2];
    if (k > 0) {
      float dd[3] = {L.xpos[k][0] - L.xpos[k - 1][0], L.xpos[k][1] - L.xpos[k - 1][1], L.xpos[k][2] - L.xpos[k - 1][2]};
      float t[3]; cross3(t, dd, fk[k]);
#pragma unroll
      for (int i = 0; i < 3; i++) { fk[k - 1][i] += fk[k][i]; nk[k - 1][i] += nk[k][i] + t[i]; }
    }
  }
  // --- actuation (lane = actuator = dof for this model): clamp ctrl, affine bias, clamp force
  if (lane < NARM) {
    float c = L.ctrl[lane];
    if (m->ctrllimited[lane]) c = fminf(fmaxf(c, m->ctrlrange[lane][0]), m->ctrlrange[lane][1]);
    float force = m->act_gain[lane] * c + m->act_bias[lane][0] + m->act_bias[lane][1] * L.qpos[lane] + m->act_bias[lane][2] * L.qvel[lane];
    if (m->forcelimited[lane]) force = fminf(fmaxf(force, m->forcerange[lane][0]), m->forcerange[lane][1]);
    float b = 0.f;
#pragma unroll
    for (int k = 0; k < NARM; k++) if (k == lane) b = bias[k];
    L.bias[lane] = b;
    L.tau[lane] = force - b;
  }
  wave_sync();
  if (lane < NARM) {
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < NARM; c++) v += L.Minv[lane][c] * L.tau[c];
    L.qacc_arm[lane] = v;
  }
  // --- free bodies in COM-twist coordinates: a_com = g, alpha = Iinv (-w x I w)
  if (lane >= 32 && lane < 32 + NFREE) {
    int f = lane - 32, b = NARM + f;
    const float* qv = &L.qvel[NARM + 6 * f];
    float wb[3] = {qv[3], qv[4], qv[5]}, ww[3];
    matvec3(ww, L.xmat[b], wb);
    float r[3] = {L.xipos[b][0] - L.xpos[b][0], L.xipos[b][1] - L.xpos[b][1], L.xipos[b][2] - L.xpos[b][2]};
    float wr[3]; cross3(wr, ww, r);
    float Iw_[3]; symvec3(Iw_, L.Iw[b], ww);
    float g[3]; cross3(g, ww, Iw_);
    g[0] = -g[0]; g[1] = -g[1]; g[2] = -g[2];
    float alp[3]; symvec3(alp, L.fIinv[f], g);
    // Solver coordinates: (a~, alpha) with a~ = qacc_lin + alpha x r, i.e. the COM acceleration WITHOUT the
    // centripetal term w x (w x r).  J*qacc in MuJoCo's generalized coordinates (which neglects Jdot*qvel)
    // is then e.(a~ + alpha x (p - com)) exactly, and force updates stay (1/m, Iinv).
    float cen[3]; cross3(cen, ww, wr);
#pragma unroll
    for (int i = 0; i < 3; i++) {
      L.fvel[f][i] = qv[i] + wr[i]; L.fvel[f][3 + i] = ww[i];
      L.facc[f][i] = m->grav[i] - cen[i]; L.facc[f][3 + i] = alp[i];
    }
  }
  wave_sync();
}

// ------------------------------------------------------------------ geometry
// Lane-group policies of the narrowphase.  The query of one geom pair is "uniform" code (every lane of the group
// computes the same portal) around a lane-parallel hull scan.  G64: the whole wavefront works on one pair (fused
// kernels: the env's wave walks its candidates).  G16: one pair per DPP row of 16 lanes, four pairs per wavefront
// (k_narrow): the uniform part is issued once for four pairs, the hull scan takes four times as many steps per pair;
// rows diverge freely (each row's reductions are row-local DPP butterflies, loads become vector loads with a
// row-uniform address).  Both pick the same support vertex (max dot, smallest index), hence bit-identical contacts.
struct G64 {
  static constexpr int N = 64;
  DEV static int sub() { return wave_lane(); }
  DEV static void argmax3(float& val, int& idx, float& x, float& y, float& z) { wave_argmax3(val, idx, x, y, z); }
  DEV static void argmax(float& val, int& idx) { wave_argmax(val, idx); }
  template <class T> DEV static T ld(const T* p) { return ldc(p); }
  DEV static int uni(int v) { return wave_uniform_i(v); }
};
struct G16 {
  static constexpr int N = 16;
  DEV static int sub() { return wave_lane() & 15; }
  DEV static void argmax3(float& val, int& idx, float& x, float& y, float& z) { row_argmax3(val, idx, x, y, z); }
  DEV static void argmax(float& val, int& idx) { float x = 0.f, y = 0.f, z = 0.f; row_argmax3(val, idx, x, y, z); }
  template <class T> DEV static T ld(const T* p) { return *p; }
  DEV static int uni(int v) { return v; }
};

struct GeomW { int type, vadr, vnum; float size[3], R[9], p[3], c[3]; };

// xp/xm: world position and orientation of the geom's dynamic body (ignored for static geoms)
template <class GP = G64>
DEV void load_geom_at(const DevModel* m, int g, const float* xp, const float* xm, GeomW& G) {
  g = GP::uni(g);
  G.type = GP::ld(ldc(&m->geom_type) + g); G.vadr = GP::ld(ldc(&m->geom_vertadr) + g); G.vnum = GP::ld(ldc(&m->geom_vertnum) + g);
  const float* gp = ldc(&m->geom_pos) + 3 * g; const float* gm = ldc(&m->geom_mat) + 9 * g;
  const float* gc = ldc(&m->geom_center) + 3 * g; const float* gs = ldc(&m->geom_size) + 3 * g;
#pragma unroll
  for (int i = 0; i < 3; i++) G.size[i] = GP::ld(gs + i);
  int d = GP::ld(ldc(&m->geom_dyn) + g);
  float lp[3] = {GP::ld(gp), GP::ld(gp + 1), GP::ld(gp + 2)}, lm[9], lc[3] = {GP::ld(gc), GP::ld(gc + 1), GP::ld(gc + 2)};
#pragma unroll
  for (int i = 0; i < 9; i++) lm[i] = GP::ld(gm + i);
  // static geoms go through the same arithmetic with an identity pose (exact: 1*a + 0*b + 0*c == a), so that the
  // geom stays in registers instead of becoming a stack object selected by the branch
  float X[9], P0[3];
#pragma unroll
  for (int i = 0; i < 9; i++) X[i] = d < 0 ? (i % 4 == 0 ? 1.f : 0.f) : xm[i];
#pragma unroll
  for (int i = 0; i < 3; i++) P0[i] = d < 0 ? 0.f : xp[i];
  float t[3]; matvec3(t, X, lp);
#pragma unroll
  for (int i = 0; i < 3; i++) G.p[i] = P0[i] + t[i];
  matmul3(G.R, X, lm);
  float cw[3]; matvec3(cw, G.R, lc);
#pragma unroll
  for (int i = 0; i < 3; i++) G.c[i] = G.p[i] + cw[i];
}

DEV void load_geom(const DevModel* m, const EnvLDS& L, int g, GeomW& G) {
  g = wave_uniform_i(g);
  int d = ldc(ldc(&m->geom_dyn) + g);
  load_geom_at(m, g, L.xpos[d < 0 ? 0 : d], L.xmat[d < 0 ? 0 : d], G);
}

// Hull vertices of one geom held in registers for the duration of a narrowphase query: lane l keeps vertices
// l, l+64, ... (HULL_K of them; 64 * 8 = 512 covers all but four hulls of the SO100 scenes, the rest of those is scanned in memory).  An MPR
// query evaluates ~20-30 support points per geom; reading the hull once instead of once per support call removes
// the vertex traffic (205 -> ~10 vector loads per candidate pair).  Only k_narrow can afford the registers; the fused
// kernels use NoCache and scan memory.  Both variants visit the vertices in the same order with the same
// arithmetic, so they return the same vertex.
#ifndef HULL_K
#define HULL_K 8
#endif
struct HullCache { float x[HULL_K], y[HULL_K], z[HULL_K]; };
struct NoCache {};

template <class GP = G64>
DEV void hull_load(const DevModel* m, const GeomW& G, HullCache& H) {
  // every slot is written (slots beyond the hull, and the caches of primitives, hold zeros): the caches are moved
  // around with selects later, and a select over a never-written register is undefined behaviour for the compiler
#pragma unroll
  for (int j = 0; j < HULL_K; j++) { H.x[j] = 0.f; H.y[j] = 0.f; H.z[j] = 0.f; }
  if (G.type != G_MESH) return;
  int lane = GP::sub();
  const float* x = ldc(&m->vx) + G.vadr; const float* y = ldc(&m->vy) + G.vadr; const float* z = ldc(&m->vz) + G.vadr;
#pragma unroll
  for (int j = 0; j < HULL_K; j++) {
    if (GP::N * j >= G.vnum) break;
    int i = lane + GP::N * j;
    bool v = i < G.vnum;
    H.x[j] = v ? x[i] : 0.f; H.y[j] = v ? y[i] : 0.f; H.z[j] = v ? z[i] : 0.f;
  }
}
template <class GP = G64>
DEV void hull_load(const DevModel*, const GeomW&, NoCache&) {}

// The same cache in LDS (k_narrow, one wavefront per workgroup: round 5).  The register cache above costs 48 VGPRs for the two geoms of a
// pair across the whole query - with the EPA polytope inlined k_narrow needed 256 + 43 spilled - while the kernel used 256 B of its 20 KB LDS
// share.  Here the first HULL_LDS_MAX vertices of a hull are staged by the kernel (so101_narrow.hpp: all hulls of a work-item chunk at once,
// behind ONE memory round trip) as x[n] | y[n] | z[n] with n = 256 or 512 slots, and a support scan reads them four vertices per lane and
// instruction (ds_read_b128: lane l takes vertices 256 J + 4 l + 0..3, in increasing index order, so "largest dot product, smallest index"
// picks the vertex every other variant picks).
#define HULL_LDS_MAX 512
struct HullLDS { float* p; int n; };      // p: [3][n] floats in LDS (16-byte aligned), n: slots per coordinate (0: nothing staged)
DEV int hull_lds_slots(int type, int vnum) { return type != G_MESH ? 0 : (vnum <= 256 ? 256 : HULL_LDS_MAX); }
// the loads of one hull (issued, not waited for) and their LDS stores: split so that a caller can issue the loads of several hulls first
template <int NJ> struct HullStage { float x[NJ], y[NJ], z[NJ]; };
template <int NJ>
DEV void hull_stage_issue(const DevModel* m, int vadr, int vnum, int j0, HullStage<NJ>& T) {
  int lane = wave_lane();
  const float* x = ldc(&m->vx) + vadr; const float* y = ldc(&m->vy) + vadr; const float* z = ldc(&m->vz) + vadr;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    int i = lane + WAVE * (j0 + j);
    bool v = i < vnum;
    T.x[j] = v ? x[i] : 0.f; T.y[j] = v ? y[i] : 0.f; T.z[j] = v ? z[i] : 0.f;
  }
}
template <int NJ>
DEV void hull_stage_store(const HullLDS& H, int vnum, int j0, const HullStage<NJ>& T) {
  int lane = wave_lane();
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    int i = lane + WAVE * (j0 + j);
    if (WAVE * (j0 + j) < H.n) { H.p[i] = T.x[j]; H.p[H.n + i] = T.y[j]; H.p[2 * H.n + i] = T.z[j]; }
  }
}
// one hull into its LDS slots (loads of a 256-slot block in flight together)
DEV void hull_stage(const DevModel* m, int vadr, int vnum, const HullLDS& H) {
  for (int j0 = 0; WAVE * j0 < H.n; j0 += 4) {
    HullStage<4> T;
    hull_stage_issue<4>(m, vadr, vnum, j0, T);
    hull_stage_store<4>(H, vnum, j0, T);
  }
}
template <class GP = G64>
DEV void hull_load(const DevModel* m, const GeomW& G, HullLDS& H) { if (H.n) hull_stage(m, G.vadr, G.vnum, H); }
// uniform values parked in LDS across a phase that does not need them (k_narrow: the candidate face across the iterative query, the portal
// across the EPA expansion): the register allocator otherwise keeps them in VGPRs, 64 copies of each, or spills them to scratch memory
#define NARROW_PARK_WORDS 192
DEV float* narrow_park_store() { __shared__ __attribute__((aligned(16))) float park[NARROW_PARK_WORDS]; return park; }

DEV void select_geom(bool first, const GeomW& A, const GeomW& B, GeomW& o) {
  o.type = first ? A.type : B.type; o.vadr = first ? A.vadr : B.vadr; o.vnum = first ? A.vnum : B.vnum;
#pragma unroll
  for (int i = 0; i < 3; i++) { o.size[i] = first ? A.size[i] : B.size[i]; o.p[i] = first ? A.p[i] : B.p[i]; o.c[i] = first ? A.c[i] : B.c[i]; }
#pragma unroll
  for (int i = 0; i < 9; i++) o.R[i] = first ? A.R[i] : B.R[i];
}
DEV void select_hull(bool first, const HullCache& A, const HullCache& B, HullCache& o) {
#pragma unroll
  for (int j = 0; j < HULL_K; j++) { o.x[j] = first ? A.x[j] : B.x[j]; o.y[j] = first ? A.y[j] : B.y[j]; o.z[j] = first ? A.z[j] : B.z[j]; }
}
DEV void select_hull(bool, const NoCache&, const NoCache&, NoCache&) {}
DEV void select_hull(bool first, const HullLDS& A, const HullLDS& B, HullLDS& o) { o.p = first ? A.p : B.p; o.n = first ? A.n : B.n; }
// A SUBSET of a hull in registers (round 6, k_narrow's fast path for a flat face against a hull): the entries of one cell of the hull's
// support-vertex lists (DevModel::hl_entry), two per lane at most, in increasing index order (entry l, then entry l + 64); slots beyond the
// list carry the index 0x7fffffff.  Valid for the directions of that cell (widened by 4e-3 rad) only: there the largest dot product over the
// subset is the largest over the hull, attained by the same vertices - support() and support_multi() return the same point bit for bit.
struct HullSub { float x[2], y[2], z[2]; int i[2]; };
template <class C> struct is_hull_sub { static constexpr bool value = false; };
template <> struct is_hull_sub<HullSub> { static constexpr bool value = true; };
DEV void select_hull(bool first, const HullSub& A, const HullSub& B, HullSub& o) {
#pragma unroll
  for (int j = 0; j < 2; j++) { o.x[j] = first ? A.x[j] : B.x[j]; o.y[j] = first ? A.y[j] : B.y[j]; o.z[j] = first ? A.z[j] : B.z[j]; o.i[j] = first ? A.i[j] : B.i[j]; }
}
// cell of the cube map a direction (any length, geom frame) falls into: face 2 a + (negative), then HL_GRID x HL_GRID along the axes a + 1, a + 2
DEV int hl_cell(const float* dl) {
  float a0 = fabsf(dl[0]), a1 = fabsf(dl[1]), a2 = fabsf(dl[2]);
  int ax = a0 >= a1 ? (a0 >= a2 ? 0 : 2) : (a1 >= a2 ? 1 : 2);
  float dm = ax == 0 ? dl[0] : (ax == 1 ? dl[1] : dl[2]);
  float du = ax == 0 ? dl[1] : (ax == 1 ? dl[2] : dl[0]);
  float dv = ax == 0 ? dl[2] : (ax == 1 ? dl[0] : dl[1]);
  float inv = 1.f / fmaxf(fabsf(dm), 1e-20f);
  float gu = fminf(fmaxf((du * inv + 1.f) * (0.5f * HL_GRID), 0.f), (float)HL_GRID), gv = fminf(fmaxf((dv * inv + 1.f) * (0.5f * HL_GRID), 0.f), (float)HL_GRID);
  int iu = (int)gu; iu = iu > HL_GRID - 1 ? HL_GRID - 1 : iu;
  int iv = (int)gv; iv = iv > HL_GRID - 1 ? HL_GRID - 1 : iv;
  return ((2 * ax + (dm < 0.f ? 1 : 0)) * HL_GRID + iu) * HL_GRID + iv;
}
template <class C> struct is_hull_lds { static constexpr bool value = false; };
template <> struct is_hull_lds<HullLDS> { static constexpr bool value = true; };

// support point (world) of G in world direction dir; wave-parallel over hull vertices for meshes
template <class Cache, class GP = G64>
DEV void support(const DevModel* m, const GeomW& G, const float* dir, float* out, const Cache& H) {
  float dl[3]; matTvec3(dl, G.R, dir);
  float loc[3] = {0.f, 0.f, 0.f};
  if (G.type == G_MESH) {
    int lane = GP::sub();
    // each lane scans vertices lane, lane+64, ... (coalesced SoA loads) and keeps its best vertex in registers;
    // the wave-level argmax then broadcasts the winner with v_readlane (no second memory access, and a
    // non-finite direction of a diverged state can never index out of range)
    float best = -3.0e38f, bx = 0.f, by = 0.f, bz = 0.f; int bi = 0x7fffffff;
    const float* x = ldc(&m->vx) + G.vadr; const float* y = ldc(&m->vy) + G.vadr; const float* z = ldc(&m->vz) + G.vadr;
    int first = lane;
    if constexpr (is_hull_sub<Cache>::value) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        int i = H.i[k];
        float d = H.x[k] * dl[0] + H.y[k] * dl[1] + H.z[k] * dl[2];
        if (i < G.vnum && d > best) { best = d; bi = i; bx = H.x[k]; by = H.y[k]; bz = H.z[k]; }
      }
      first = G.vnum;                                  // (nothing else to scan)
    } else if constexpr (is_hull_lds<Cache>::value) {
      const float4* X4 = (const float4*)H.p; const float4* Y4 = (const float4*)(H.p + H.n); const float4* Z4 = (const float4*)(H.p + 2 * H.n);
      auto block = [&](int J) {
        float4 xv = X4[GP::N * J + lane], yv = Y4[GP::N * J + lane], zv = Z4[GP::N * J + lane];
        float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w}, zs[4] = {zv.x, zv.y, zv.z, zv.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          int i = 4 * GP::N * J + 4 * lane + k;
          float d = xs[k] * dl[0] + ys[k] * dl[1] + zs[k] * dl[2];
          if (i < G.vnum && d > best) { best = d; bi = i; bx = xs[k]; by = ys[k]; bz = zs[k]; }
        }
      };
      if constexpr (GP::N == WAVE) {
#pragma unroll
        for (int J = 0; J < HULL_LDS_MAX / (4 * WAVE); J++) {
          if (4 * WAVE * J >= G.vnum || 4 * WAVE * J >= H.n) break;
          block(J);
        }
      } else {
        // a row of 16 lanes per pair (k_narrow's row pass): 64 vertices per step, trip count per row
        const int lim = G.vnum < H.n ? G.vnum : H.n;
#pragma unroll 1
        for (int J = 0; 4 * GP::N * J < lim; J++) block(J);
      }
      first = lane + H.n;
    } else if constexpr (sizeof(Cache) >= sizeof(HullCache)) {
#pragma unroll
      for (int j = 0; j < HULL_K; j++) {
        if (GP::N * j >= G.vnum) break;
        int i = lane + GP::N * j;
        float X = H.x[j], Y = H.y[j], Z = H.z[j];
        float d = X * dl[0] + Y * dl[1] + Z * dl[2];
        if (i < G.vnum && d > best) { best = d; bi = i; bx = X; by = Y; bz = Z; }
      }
      first = lane + GP::N * HULL_K;
    }
#pragma unroll 4
    for (int i = first; i < G.vnum; i += GP::N) {
      float X = x[i], Y = y[i], Z = z[i];
      float d = X * dl[0] + Y * dl[1] + Z * dl[2];
      if (d > best) { best = d; bi = i; bx = X; by = Y; bz = Z; }
    }
    GP::argmax3(best, bi, bx, by, bz);
    loc[0] = bx; loc[1] = by; loc[2] = bz;
  } else if (G.type == G_BOX) {
#pragma unroll
    for (int i = 0; i < 3; i++) loc[i] = dl[i] >= 0.f ? G.size[i] : -G.size[i];
  } else if (G.type == G_CAPSULE) {
    float n = sqrtf(dot3(dl, dl));
    if (n > MINVAL_F) { float s = G.size[0] / n; loc[0] = s * dl[0]; loc[1] = s * dl[1]; loc[2] = s * dl[2]; }
    loc[2] += dl[2] >= 0.f ? G.size[1] : -G.size[1];
  } else if (G.type == G_CYLINDER) {
    float n = sqrtf(dl[0] * dl[0] + dl[1] * dl[1]);
    if (n > MINVAL_F) { float s = G.size[0] / n; loc[0] = s * dl[0]; loc[1] = s * dl[1]; }
    loc[2] = dl[2] >= 0.f ? G.size[1] : -G.size[1];
  } else if (G.type == G_SPHERE) {
    float n = sqrtf(dot3(dl, dl));
    if (n > MINVAL_F) { float s = G.size[0] / n; loc[0] = s * dl[0]; loc[1] = s * dl[1]; loc[2] = s * dl[2]; }
  }
  float w[3]; matvec3(w, G.R, loc);
  out[0] = G.p[0] + w[0]; out[1] = G.p[1] + w[1]; out[2] = G.p[2] + w[2];
}

struct MV { float v[3], a[3], b[3]; };

template <class Cache, class GP = G64>
DEV void mdsupport(const DevModel* m, const GeomW& G1, const GeomW& G2, const float* dir, const float* org, MV& o,
                   const Cache& H1, const Cache& H2) {
  float nd[3] = {-dir[0], -dir[1], -dir[2]};
  support<Cache, GP>(m, G1, dir, o.a, H1);
  support<Cache, GP>(m, G2, nd, o.b, H2);
#pragma unroll
  for (int i = 0; i < 3; i++) { o.a[i] -= org[i]; o.b[i] -= org[i]; o.v[i] = o.a[i] - o.b[i]; }
}

DEV bool isz(float x) { return fabsf(x) < EPS_F; }

// origin to segment P0-P1: squared distance, closest point and the parameter t (weight of P1)
DEV float seg_origin(const float* P0, const float* P1, float* wt, float* tout) {
  float dd[3] = {P1[0] - P0[0], P1[1] - P0[1], P1[2] - P0[2]};
  float t = -dot3(P0, dd) / fmaxf(dot3(dd, dd), 1e-30f);
  t = fminf(fmaxf(t, 0.f), 1.f);
  wt[0] = P0[0] + t * dd[0]; wt[1] = P0[1] + t * dd[1]; wt[2] = P0[2] + t * dd[2];
  *tout = t;
  return dot3(wt, wt);
}

// squared distance of the origin to triangle (x0,B,C); wit = closest point, bw = its barycentric weights
DEV float origin_tri_dist2(const float* x0, const float* B, const float* C, float* wit, float* bw) {
  float d1[3] = {B[0] - x0[0], B[1] - x0[1], B[2] - x0[2]}, d2[3] = {C[0] - x0[0], C[1] - x0[1], C[2] - x0[2]};
  float v = dot3(d1, d1), w = dot3(d2, d2), p = dot3(x0, d1), q = dot3(x0, d2), r = dot3(d1, d2);
  float den = w * v - r * r, sp = -1.f, tp = -1.f;
  if (fabsf(den) > 1e-30f) { sp = (q * r - w * p) / den; tp = (-sp * r - q) / w; }
  if ((isz(sp) || sp > 0.f) && (isz(sp - 1.f) || sp < 1.f) && (isz(tp) || tp > 0.f) && (isz(tp - 1.f) || tp < 1.f) &&
      (isz(tp + sp - 1.f) || tp + sp < 1.f)) {
    wit[0] = x0[0] + sp * d1[0] + tp * d2[0]; wit[1] = x0[1] + sp * d1[1] + tp * d2[1]; wit[2] = x0[2] + sp * d1[2] + tp * d2[2];
    bw[0] = 1.f - sp - tp; bw[1] = sp; bw[2] = tp;
    return dot3(wit, wit);
  }
  float w1[3], w2[3], w3[3], t1, t2, t3;
  float e1 = seg_origin(x0, B, w1, &t1), e2 = seg_origin(x0, C, w2, &t2), e3 = seg_origin(B, C, w3, &t3);
  float best = e1; wit[0] = w1[0]; wit[1] = w1[1]; wit[2] = w1[2]; bw[0] = 1.f - t1; bw[1] = t1; bw[2] = 0.f;
  if (e2 < best) { best = e2; wit[0] = w2[0]; wit[1] = w2[1]; wit[2] = w2[2]; bw[0] = 1.f - t2; bw[1] = 0.f; bw[2] = t2; }
  if (e3 < best) { best = e3; wit[0] = w3[0]; wit[1] = w3[1]; wit[2] = w3[2]; bw[0] = 0.f; bw[1] = 1.f - t3; bw[2] = t3; }
  return best;
}

#define MVCOPY(dst, src) do { _Pragma("unroll") for (int _i = 0; _i < 3; _i++) { (dst).v[_i] = (src).v[_i]; (dst).a[_i] = (src).a[_i]; (dst).b[_i] = (src).b[_i]; } } while (0)

// Interior point of a geom for the MPR origin ray.  For primitives it is the point of the primitive closest to
// `target` (the other geom's centre), pulled slightly inside, so that the ray follows the local penetration
// direction: with the fixed geometric centre of a large flat box (the 1.0 x 0.8 m table top) the ray is nearly
// parallel to the contact face and MPR's depth estimate becomes erratic (EPA, which mujoco >= 3.3 uses, has no such
// dependence).  Hulls keep their centre of mass.
DEV void interior_point(const GeomW& G, const float* target, float* out) {
  if (G.type == G_MESH || G.type == G_SPHERE || G.type == G_PLANE) { out[0] = G.c[0]; out[1] = G.c[1]; out[2] = G.c[2]; return; }
  float rel[3] = {target[0] - G.p[0], target[1] - G.p[1], target[2] - G.p[2]}, t[3];
  matTvec3(t, G.R, rel);
  if (G.type == G_BOX) {
#pragma unroll
    for (int i = 0; i < 3; i++) { float lim = G.size[i] - fminf(1e-3f, 0.5f * G.size[i]); t[i] = fminf(fmaxf(t[i], -lim), lim); }
  } else if (G.type == G_CYLINDER) {
    float rmax = G.size[0] - fminf(1e-3f, 0.5f * G.size[0]), rho = sqrtf(t[0] * t[0] + t[1] * t[1]);
    if (rho > rmax) { float sc = rmax / rho; t[0] *= sc; t[1] *= sc; }
    float lim = G.size[1] - fminf(1e-3f, 0.5f * G.size[1]);
    t[2] = fminf(fmaxf(t[2], -lim), lim);
  } else {            // capsule: closest point of the axis segment
    t[0] = 0.f; t[1] = 0.f; t[2] = fminf(fmaxf(t[2], -G.size[1]), G.size[1]);
  }
  float w[3]; matvec3(w, G.R, t);
  out[0] = G.p[0] + w[0]; out[1] = G.p[1] + w[1]; out[2] = G.p[2] + w[2];
}

// EPA (expanding polytope) from the tetrahedron the MPR query ends with - its interior point v0 and the portal v1 v2 v3, which
// contains the origin: the face nearest to the origin is pushed out along its normal until the support point in that direction lies on
// it.  That face is a face of the Minkowski difference and its distance the MINIMUM translation separating the geoms - what mujoco >=
// 3.3's native GJK / EPA reports (the reference enables multiccd on top, so100_task.py:151); MPR's portal is only some face of an inner
// approximation (DESIGN.md section 4 measures the difference against the brute-forced minimum).  Wave-parallel: lane k keeps vertex k
// (with its witness points) and face k (vertex indices, unit normal, distance) in registers, at most 64 of each; one expansion =
// wave-argmin over the faces, one support pair, a visibility ballot, a scan of the visible faces' edges for the horizon, and new faces
// in the freed lanes.  Entirely wave-uniform control flow (G64 policy only).
#define EPA_MAX_EXPANSIONS 30     // 4 + 2 x 30 faces fill the 64 face lanes
struct EpaFace { int a, b, c; float n[3], d; bool alive; };

DEV void epa_make_face(bool doit, int a, int b, int c, float vx, float vy, float vz, EpaFace& F) {
  // (every lane takes part in the exchanges; only `doit` lanes keep the result)
  float A[3] = {wave_bcast_f(vx, a), wave_bcast_f(vy, a), wave_bcast_f(vz, a)};
  float B[3] = {wave_bcast_f(vx, b), wave_bcast_f(vy, b), wave_bcast_f(vz, b)};
  float C[3] = {wave_bcast_f(vx, c), wave_bcast_f(vy, c), wave_bcast_f(vz, c)};
  float e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, e2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]}, n[3];
  cross3(n, e1, e2);
  float len = sqrtf(dot3(n, n));
  bool ok = len > 1e-14f;
  float inv = ok ? 1.f / len : 0.f;
  n[0] *= inv; n[1] *= inv; n[2] *= inv;
  float d = dot3(n, A);
  // (no per-face flip: the winding is consistent by construction - the first tetrahedron is oriented as a whole, a new face takes its
  //  horizon edge in the direction the removed face had it - and a per-face sign test turns a face the origin lies ON inside out)
  if (doit) {
    F.a = a; F.b = b; F.c = c;
    F.n[0] = n[0]; F.n[1] = n[1]; F.n[2] = n[2];
    F.d = ok ? d : 3.0e38f; F.alive = ok;
  }
}

template <class Cache, class GP>
DEV bool epa_expand(const DevModel* m, const GeomW& G1, const GeomW& G2, const float* org, const MV& v0, const MV& v1, const MV& v2, const MV& v3,
                    float tol, float* depth, float* dir, float* pos, const Cache& H1, const Cache& H2) {
  __shared__ unsigned int epa_list[WAVE];                 // horizon edges of one expansion, in the order the new faces take them
  int lane = wave_lane();
  float vx = 0.f, vy = 0.f, vz = 0.f, ax = 0.f, ay = 0.f, az = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
#define EPA_PUT(k, X) if (lane == (k)) { vx = X.v[0]; vy = X.v[1]; vz = X.v[2]; ax = X.a[0]; ay = X.a[1]; az = X.a[2]; bx = X.b[0]; by = X.b[1]; bz = X.b[2]; }
  EPA_PUT(0, v0) EPA_PUT(1, v1) EPA_PUT(2, v2) EPA_PUT(3, v3)
  int nv = 4, nf = 4;
  EpaFace F; F.a = F.b = F.c = 0; F.n[0] = F.n[1] = F.n[2] = 0.f; F.d = 3.0e38f; F.alive = false;
  {
    // faces (1 2 3), (0 2 1), (0 3 2), (0 1 3): consistently wound; outward when (v2 - v1) x (v3 - v1) points away from v0, else all four swapped
    float e1[3] = {v2.v[0] - v1.v[0], v2.v[1] - v1.v[1], v2.v[2] - v1.v[2]}, e2[3] = {v3.v[0] - v1.v[0], v3.v[1] - v1.v[1], v3.v[2] - v1.v[2]}, nn[3];
    cross3(nn, e1, e2);
    float to0[3] = {v0.v[0] - v1.v[0], v0.v[1] - v1.v[1], v0.v[2] - v1.v[2]};
    bool swap = dot3(nn, to0) > 0.f;
    int a0 = lane == 0 ? 1 : 0, b0 = lane == 0 ? 2 : (lane == 1 ? 2 : (lane == 2 ? 3 : 1)), c0 = lane == 0 ? 3 : (lane == 1 ? 1 : (lane == 2 ? 2 : 3));
    epa_make_face(lane < 4, lane < 4 ? a0 : 0, lane < 4 ? (swap ? c0 : b0) : 0, lane < 4 ? (swap ? b0 : c0) : 0, vx, vy, vz, F);
  }
  int best = 0; float bd = 0.f, bn[3] = {0.f, 0.f, 0.f};
  bool converged = false;
  for (int it = 0; it <= EPA_MAX_EXPANSIONS; it++) {
    float key = F.alive ? -F.d : -3.0e38f; int idx = lane;
    wave_argmax(key, idx);
    if (key <= -3.0e38f) return false;
    best = idx; bd = -key;
    bn[0] = wave_get_f(F.n[0], best); bn[1] = wave_get_f(F.n[1], best); bn[2] = wave_get_f(F.n[2], best);
    MV w;
    mdsupport<Cache, GP>(m, G1, G2, bn, org, w, H1, H2);
    float reach = dot3(bn, w.v) - bd;
    bool dup = lane < nv && fabsf(vx - w.v[0]) + fabsf(vy - w.v[1]) + fabsf(vz - w.v[2]) < 1e-9f;
#ifdef SO101_EPA_TRACE
    { bool anydup = wave_ballot(dup) != 0ull; if (lane == 0) fprintf(stderr, "  epa it %d best %d bd %.7g reach %.3g nv %d nf %d dup %d w %.6f %.6f %.6f\n", it, best, bd, reach, nv, nf, (int)anydup, w.v[0], w.v[1], w.v[2]); }
#endif
    if (reach <= tol) { converged = true; break; }
    // (a support point the polytope already has, although the face claims room beyond it: rounding has made the polytope inconsistent)
    if (wave_ballot(dup) != 0ull || it == EPA_MAX_EXPANSIONS || nv >= WAVE || nf + 2 > WAVE) break;
    int wi = nv;
    EPA_PUT(wi, w)
    nv++;
    bool vis = F.alive && (F.n[0] * w.v[0] + F.n[1] * w.v[1] + F.n[2] * w.v[2] - F.d > 0.f);
    unsigned long long vmask = wave_ballot(vis);
    int nvis = __popcll(vmask);
    int packed = F.a | (F.b << 8) | (F.c << 16) | ((vis ? 1 : 0) << 24);
    // horizon: an edge of a visible face whose reversed edge belongs to no other visible face
    bool s0 = false, s1 = false, s2 = false;
    for (unsigned long long rest = vmask; rest != 0ull; rest &= rest - 1ull) {      // the visible faces only (a handful of the polytope's)
      int j = (int)__builtin_ctzll(rest);
      int pj = wave_get_i(packed, j);                     // (v_readlane: j is wave-uniform)
      int ja = pj & 255, jb = (pj >> 8) & 255, jc = (pj >> 16) & 255;
#define EPA_REV(x, y) ((ja == (y) && jb == (x)) || (jb == (y) && jc == (x)) || (jc == (y) && ja == (x)))
      bool other = j != lane;
      s0 = s0 || (other && EPA_REV(F.a, F.b)); s1 = s1 || (other && EPA_REV(F.b, F.c)); s2 = s2 || (other && EPA_REV(F.c, F.a));
    }
    bool h0 = vis && !s0, h1 = vis && !s1, h2 = vis && !s2;
    unsigned long long m0 = wave_ballot(h0), m1 = wave_ballot(h1), m2 = wave_ballot(h2);
    int K = __popcll(m0) + __popcll(m1) + __popcll(m2);
#ifdef SO101_EPA_TRACE
    if (lane == 0) fprintf(stderr, "     nvis %d K %d\n", nvis, K);
#endif
    int base = wave_prefix(m0) + wave_prefix(m1) + wave_prefix(m2);
    wave_sync();
    if (h0) epa_list[base] = (unsigned int)(F.a | (F.b << 8));
    if (h1) epa_list[base + (h0 ? 1 : 0)] = (unsigned int)(F.b | (F.c << 8));
    if (h2) epa_list[base + (h0 ? 1 : 0) + (h1 ? 1 : 0)] = (unsigned int)(F.c | (F.a << 8));
    wave_sync();
    int extra = K > nvis ? K - nvis : 0;
    if (nf + extra > WAVE) extra = WAVE - nf;
    bool fresh = lane >= nf && lane < nf + extra;
    int r = vis ? wave_prefix(vmask) : (nvis + lane - nf);
    bool make = (vis || fresh) && r < K;
    unsigned int e = epa_list[make ? r : 0];
    if (vis) { F.alive = false; F.d = 3.0e38f; }
    epa_make_face(make, make ? (int)(e & 255u) : 0, make ? (int)(e >> 8) : 0, make ? wi : 0, vx, vy, vz, F);
    nf += extra;
  }
#undef EPA_PUT
#undef EPA_REV
  // a polytope that has not reached the surface after EPA_MAX_EXPANSIONS (a 0.6 mm sphere deep inside a mesh: the difference is
  // curved everywhere) is an INNER bound, its nearest face too shallow: the caller falls back to MPR's own answer
  if (!converged) return false;
  *depth = bd; dir[0] = bn[0]; dir[1] = bn[1]; dir[2] = bn[2];
  // Witness face.  A flat facet of the Minkowski difference (an edge against an edge, a face against an edge) is triangulated by the
  // polytope; its triangles are coplanar up to rounding, so WHICH of them has the smallest plane distance is decided by the last bit,
  // and the projection of the origin may lie in a neighbour of the winner (a 6 cm hull edge across the 1 m table edge: the clamped
  // barycentric weights of the wrong sliver put the contact 1.2 cm away).  Among the faces coplanar with the nearest one (plane
  // distance within tol, normal within 1e-5) the witness is therefore interpolated on the one that contains the projection best
  // (largest smallest barycentric weight; lane = face, one pass).  Depth and normal stay those of the nearest face.
  {
    float p0[3] = {bd * bn[0], bd * bn[1], bd * bn[2]};
    float fA[3] = {wave_bcast_f(vx, F.a), wave_bcast_f(vy, F.a), wave_bcast_f(vz, F.a)}, fB[3] = {wave_bcast_f(vx, F.b), wave_bcast_f(vy, F.b), wave_bcast_f(vz, F.b)},
          fC[3] = {wave_bcast_f(vx, F.c), wave_bcast_f(vy, F.c), wave_bcast_f(vz, F.c)};
    float g1[3], g2[3], gp[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { g1[i] = fB[i] - fA[i]; g2[i] = fC[i] - fA[i]; gp[i] = p0[i] - fA[i]; }
    float q11 = dot3(g1, g1), q12 = dot3(g1, g2), q22 = dot3(g2, g2), s1 = dot3(gp, g1), s2 = dot3(gp, g2), qden = q11 * q22 - q12 * q12;
    bool okf = qden > 1e-30f;
    float ub = okf ? (q22 * s1 - q12 * s2) / qden : 0.f, uc = okf ? (q11 * s2 - q12 * s1) / qden : 0.f;
    float score = fminf(1.f - ub - uc, fminf(ub, uc));
    bool elig = F.alive && okf && F.d - bd <= tol && F.n[0] * bn[0] + F.n[1] * bn[1] + F.n[2] * bn[2] >= 1.f - 1e-5f;
    float key = elig ? score : -3.0e38f; int widx = lane;
    wave_argmax(key, widx);
    if (key > -3.0e38f) best = widx;
  }
  int ia = wave_bcast_i(F.a, best), ib = wave_bcast_i(F.b, best), ic = wave_bcast_i(F.c, best);
  float A[3] = {wave_bcast_f(vx, ia), wave_bcast_f(vy, ia), wave_bcast_f(vz, ia)}, B[3] = {wave_bcast_f(vx, ib), wave_bcast_f(vy, ib), wave_bcast_f(vz, ib)},
        C[3] = {wave_bcast_f(vx, ic), wave_bcast_f(vy, ic), wave_bcast_f(vz, ic)};
  float p[3] = {bd * bn[0], bd * bn[1], bd * bn[2]}, e1[3], e2[3], ep[3];
#pragma unroll
  for (int i = 0; i < 3; i++) { e1[i] = B[i] - A[i]; e2[i] = C[i] - A[i]; ep[i] = p[i] - A[i]; }
  float d11 = dot3(e1, e1), d12 = dot3(e1, e2), d22 = dot3(e2, e2), r1 = dot3(ep, e1), r2 = dot3(ep, e2), den = d11 * d22 - d12 * d12;
  float wb = den > 1e-30f ? (d22 * r1 - d12 * r2) / den : 0.f, wc = den > 1e-30f ? (d11 * r2 - d12 * r1) / den : 0.f;
  wb = fminf(fmaxf(wb, 0.f), 1.f); wc = fminf(fmaxf(wc, 0.f), 1.f - wb);
  float wa = 1.f - wb - wc;
  float PA[3] = {wave_bcast_f(ax, ia), wave_bcast_f(ay, ia), wave_bcast_f(az, ia)}, PB[3] = {wave_bcast_f(ax, ib), wave_bcast_f(ay, ib), wave_bcast_f(az, ib)},
        PC[3] = {wave_bcast_f(ax, ic), wave_bcast_f(ay, ic), wave_bcast_f(az, ic)};
  float QA[3] = {wave_bcast_f(bx, ia), wave_bcast_f(by, ia), wave_bcast_f(bz, ia)}, QB[3] = {wave_bcast_f(bx, ib), wave_bcast_f(by, ib), wave_bcast_f(bz, ib)},
        QC[3] = {wave_bcast_f(bx, ic), wave_bcast_f(by, ic), wave_bcast_f(bz, ic)};
#pragma unroll
  for (int i = 0; i < 3; i++) pos[i] = 0.5f * ((wa * PA[i] + wb * PB[i] + wc * PC[i]) + (wa * QA[i] + wb * QB[i] + wc * QC[i])) + org[i];
  return true;
}

// MPR penetration query (XenoCollide / libccd ccdMPRPenetration).  Entirely wave-uniform control flow.
template <class Cache, class GP = G64>
DEV bool mpr_penetration(const DevModel* m, const GeomW& G1, const GeomW& G2, float* depth, float* dir, float* pos,
                         const Cache& H1, const Cache& H2) {
  const float mpr_tol = ldc(&m->mpr_tol); const int mpr_iter = ldc(&m->mpr_iter);
  float org[3], c2[3];
  interior_point(G1, G2.c, org);
  interior_point(G2, org, c2);
  MV v0, v1, v2, v3, v4;
#pragma unroll
  for (int i = 0; i < 3; i++) { v0.a[i] = 0.f; v0.b[i] = c2[i] - org[i]; v0.v[i] = -v0.b[i]; }
  if (isz(v0.v[0]) && isz(v0.v[1]) && isz(v0.v[2])) v0.v[0] += 1e-5f;
  float d[3] = {-v0.v[0], -v0.v[1], -v0.v[2]};
  normalize3(d);
  mdsupport<Cache, GP>(m, G1, G2, d, org, v1, H1, H2);
  float dt = dot3(v1.v, d);
  if (isz(dt) || dt < 0.f) return false;
  cross3(d, v0.v, v1.v);
  float dn = sqrtf(dot3(d, d));
  // "origin on the v0-v1 segment": v0 and v1 collinear.  libccd's absolute test |v0 x v1| < eps, with the fp32 epsilon 1.2e-7, fires far from
  // collinearity when the vectors are short - the nudged ray of two coinciding interior points is 1e-5 long, so any support point within 9
  // degrees of it passed for "on the ray" and the pair got the distance to that support point as its depth: a wrist hull whose centre lies
  // inside the static puck reported 76 mm sideways where the minimum translation (and the fp64 oracle, whose epsilon is 1e-10 as in MuJoCo's
  // double-precision build of libccd) says 45 mm through the cap (round 6: seed 3 of test_failure_rates_on_the_headline_workload, env 23;
  // tests/golden/probe_outlier_states.json).  Hence also a RELATIVE bound, sin(angle) < 1e-3: for |v0| |v1| >= 1.2e-4 m^2 - centimetre-sized
  // vectors, every pair whose interior points are apart - the absolute test is the tighter one and decides as before (measured: with 1e-4 the
  // symmetric finger pairs of the ALOHA grippers, whose support points ARE on the ray up to fp32 rounding, went through the full portal search
  // and EPA instead of this exit - the same contacts within the parity tolerances, ALOHA 274 -> 259 k env-steps/s; gpurun_out g16).
  if (dn < fminf(EPS_F, SO101_COLLINEAR_REL * sqrtf(dot3(v0.v, v0.v) * dot3(v1.v, v1.v)))) {
    if (isz(v1.v[0]) && isz(v1.v[1]) && isz(v1.v[2])) {     // touching contact
      *depth = 0.f; dir[0] = dir[1] = dir[2] = 0.f;
#pragma unroll
      for (int i = 0; i < 3; i++) pos[i] = 0.5f * (v1.a[i] + v1.b[i]) + org[i];
      return true;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) { pos[i] = 0.5f * (v1.a[i] + v1.b[i]) + org[i]; dir[i] = v1.v[i]; }
    *depth = normalize3(dir);                                 // origin on the v0-v1 segment
    return true;
  }
  normalize3(d);
  mdsupport<Cache, GP>(m, G1, G2, d, org, v2, H1, H2);
  dt = dot3(v2.v, d);
  if (isz(dt) || dt < 0.f) return false;
  float va[3], vb[3];
#pragma unroll
  for (int i = 0; i < 3; i++) { va[i] = v1.v[i] - v0.v[i]; vb[i] = v2.v[i] - v0.v[i]; }
  cross3(d, va, vb); normalize3(d);
  if (dot3(d, v0.v) > 0.f) {
    MV t; MVCOPY(t, v1); MVCOPY(v1, v2); MVCOPY(v2, t);
    d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2];
  }
  bool have3 = false;
  for (int guard = 0; guard < 100 && !have3; guard++) {
    mdsupport<Cache, GP>(m, G1, G2, d, org, v3, H1, H2);
    dt = dot3(v3.v, d);
    if (isz(dt) || dt < 0.f) return false;
    bool cont = false;
    cross3(va, v1.v, v3.v);
    dt = dot3(va, v0.v);
    if (dt < 0.f && !isz(dt)) { MVCOPY(v2, v3); cont = true; }
    if (!cont) {
      cross3(va, v3.v, v2.v);
      dt = dot3(va, v0.v);
      if (dt < 0.f && !isz(dt)) { MVCOPY(v1, v3); cont = true; }
    }
    if (cont) {
#pragma unroll
      for (int i = 0; i < 3; i++) { va[i] = v1.v[i] - v0.v[i]; vb[i] = v2.v[i] - v0.v[i]; }
      cross3(d, va, vb); normalize3(d);
    } else have3 = true;
  }
  if (!have3) return false;
  // refine the portal until it encloses the origin ray, then push it to the surface
  bool inside = false;
  for (int it = 0; it < 200; it++) {
#pragma unroll
    for (int i = 0; i < 3; i++) { va[i] = v2.v[i] - v1.v[i]; vb[i] = v3.v[i] - v1.v[i]; }
    cross3(d, va, vb); normalize3(d);
    if (!inside) {
      dt = dot3(d, v1.v);
      if (isz(dt) || dt > 0.f) {                                        // portal encapsules origin: start penetration phase
#ifndef SO101_MPR     // (default; -DSO101_MPR = build.py --mpr, libso101_hip_mpr.so, keeps MPR's own answer): EPA takes over here - the tetrahedron v0 v1 v2 v3 contains the
        if constexpr (GP::N == WAVE) {          // origin from now on, and MPR's own refinement of the portal towards the surface is work EPA does anyway
          if constexpr (is_hull_lds<Cache>::value) {
            float* pk = narrow_park_store() + 64;
            if (wave_lane() == 0) {
#pragma unroll
              for (int i = 0; i < 3; i++) { pk[i] = v0.v[i]; pk[3 + i] = v0.a[i]; pk[6 + i] = v0.b[i]; pk[9 + i] = v1.v[i]; pk[12 + i] = v1.a[i]; pk[15 + i] = v1.b[i];
                                            pk[18 + i] = v2.v[i]; pk[21 + i] = v2.a[i]; pk[24 + i] = v2.b[i]; pk[27 + i] = v3.v[i]; pk[30 + i] = v3.a[i]; pk[33 + i] = v3.b[i]; }
            }
            wave_sync();
          }
          if (epa_expand<Cache, GP>(m, G1, G2, org, v0, v1, v2, v3, mpr_tol, depth, dir, pos, H1, H2)) return true;
          if constexpr (is_hull_lds<Cache>::value) {
            wave_sync();
            const float* pk = narrow_park_store() + 64;
#pragma unroll
            for (int i = 0; i < 3; i++) { v0.v[i] = pk[i]; v0.a[i] = pk[3 + i]; v0.b[i] = pk[6 + i]; v1.v[i] = pk[9 + i]; v1.a[i] = pk[12 + i]; v1.b[i] = pk[15 + i];
                                          v2.v[i] = pk[18 + i]; v2.a[i] = pk[21 + i]; v2.b[i] = pk[24 + i]; v3.v[i] = pk[27 + i]; v3.a[i] = pk[30 + i]; v3.b[i] = pk[33 + i]; }
          }
        }
#endif
        inside = true; it = -1; continue;
      }
    }
    mdsupport<Cache, GP>(m, G1, G2, d, org, v4, H1, H2);
    float dv1 = dot3(v1.v, d), dv2 = dot3(v2.v, d), dv3 = dot3(v3.v, d), dv4 = dot3(v4.v, d);
    float dm = fminf(fminf(dv4 - dv1, dv4 - dv2), dv4 - dv3);
    bool reached = isz(dm - mpr_tol) || dm < mpr_tol;
    if (!inside) {
      if (!(isz(dv4) || dv4 > 0.f)) return false;     // cannot encapsule origin
      if (reached || it > 100) return false;
    } else if (reached || it > mpr_iter) {
      float pd[3], bw[3];
      float d2 = origin_tri_dist2(v1.v, v2.v, v3.v, pd, bw);
      *depth = sqrtf(d2);
      if (isz(pd[0]) && isz(pd[1]) && isz(pd[2])) { *depth = 0.f; dir[0] = d[0]; dir[1] = d[1]; dir[2] = d[2]; }
      else { dir[0] = pd[0]; dir[1] = pd[1]; dir[2] = pd[2]; normalize3(dir); }
      // contact position: midpoint of the two witness points of the closest point on the portal (the witness pair
      // GJK/EPA reports); libccd's origin-ray weights are path dependent for deep penetrations
#pragma unroll
      for (int i = 0; i < 3; i++) {
        float p1 = bw[0] * v1.a[i] + bw[1] * v2.a[i] + bw[2] * v3.a[i];
        float p2 = bw[0] * v1.b[i] + bw[1] * v2.b[i] + bw[2] * v3.b[i];
        pos[i] = 0.5f * (p1 + p2) + org[i];
      }
      return true;
    }
    // expand portal
    float v4v0[3]; cross3(v4v0, v4.v, v0.v);
    float t1 = dot3(v1.v, v4v0);
    if (t1 > 0.f) {
      float t2 = dot3(v2.v, v4v0);
      if (t2 > 0.f) MVCOPY(v1, v4); else MVCOPY(v3, v4);
    } else {
      float t3 = dot3(v3.v, v4v0);
      if (t3 > 0.f) MVCOPY(v2, v4); else MVCOPY(v1, v4);
    }
  }
  return false;
}

DEV void make_frame(float* fr) {
  float* x = fr; float* y = fr + 3; float* z = fr + 6;
  y[0] = 0.f; y[1] = 0.f; y[2] = 0.f;
  if (x[1] < 0.5f && x[1] > -0.5f) y[1] = 1.f; else y[2] = 1.f;
  float t = dot3(x, y);
  y[0] -= t * x[0]; y[1] -= t * x[1]; y[2] -= t * x[2];
  normalize3(y);
  cross3(z, x, y);
}

// ------------------------------------------------------------------ collision driver
// oriented box of geom g in the world: axes R (columns), centre c, half extents h (the geom-frame box that the model
// compiler put around the hull / primitive)
DEV void geom_obb(const DevModel* m, const EnvLDS& L, int g, float* R, float* c, float* h) {
  const float* ab = m->geom_aabb + 6 * g;
  const float* gp = m->geom_pos + 3 * g; const float* gm = m->geom_mat + 9 * g;
  int d = m->geom_dyn[g];
  float lm[9], lp[3] = {gp[0], gp[1], gp[2]}, p[3];
#pragma unroll
  for (int i = 0; i < 9; i++) lm[i] = gm[i];
  if (d < 0) {
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = lm[i];
#pragma unroll
    for (int i = 0; i < 3; i++) p[i] = lp[i];
  } else {
    matmul3(R, L.xmat[d], lm);
    float t[3]; matvec3(t, L.xmat[d], lp);
#pragma unroll
    for (int i = 0; i < 3; i++) p[i] = L.xpos[d][i] + t[i];
  }
  float lc[3] = {ab[0], ab[1], ab[2]}, cw[3];
  matvec3(cw, R, lc);
#pragma unroll
  for (int i = 0; i < 3; i++) { c[i] = p[i] + cw[i]; h[i] = ab[3 + i]; }
}

// Upper bound of a hull's support function h(dl) = max_v v . dl (dl: unit direction in the geom frame) from its support-bound table
// (so101_model.hpp DevModel::hull_sbt): bilinear over the four grid points around dl / |dl|_inf on the cube face, times |dl|_inf.
DEV float sbt_bound(const float* T, const float* dl) {
  float a0 = fabsf(dl[0]), a1 = fabsf(dl[1]), a2 = fabsf(dl[2]);
  int ax = a0 >= a1 ? (a0 >= a2 ? 0 : 2) : (a1 >= a2 ? 1 : 2);
  float dm = ax == 0 ? dl[0] : (ax == 1 ? dl[1] : dl[2]);
  float du = ax == 0 ? dl[1] : (ax == 1 ? dl[2] : dl[0]);
  float dv = ax == 0 ? dl[2] : (ax == 1 ? dl[0] : dl[1]);
  float mm = fmaxf(fabsf(dm), 1e-20f), inv = 1.f / mm;
  const float gs = 0.5f * (float)(SBT_GRID - 1);
  float gu = fminf(fmaxf((du * inv + 1.f) * gs, 0.f), (float)(SBT_GRID - 1)), gv = fminf(fmaxf((dv * inv + 1.f) * gs, 0.f), (float)(SBT_GRID - 1));
  int iu = (int)gu; iu = iu > SBT_GRID - 2 ? SBT_GRID - 2 : iu;
  int iv = (int)gv; iv = iv > SBT_GRID - 2 ? SBT_GRID - 2 : iv;
  float fu = gu - (float)iu, fv = gv - (float)iv;
  const float* F = T + ((2 * ax + (dm < 0.f ? 1 : 0)) * SBT_GRID + iu) * SBT_GRID + iv;
  float h = (1.f - fu) * ((1.f - fv) * F[0] + fv * F[1]) + fu * ((1.f - fv) * F[SBT_GRID] + fv * F[SBT_GRID + 1]);
  return h * mm;
}
// lowest extent of mesh geom g (world rotation R, world origin p) along the world unit direction f, from below: min_x (x . f) >= this
DEV float sbt_lowest(const DevModel* m, int g, const float* R, const float* p, const float* f) {
  float dl[3] = {-(R[0] * f[0] + R[3] * f[1] + R[6] * f[2]), -(R[1] * f[0] + R[4] * f[1] + R[7] * f[2]), -(R[2] * f[0] + R[5] * f[1] + R[8] * f[2])};
  return dot3(p, f) - sbt_bound(m->hull_sbt + (size_t)g * SBT_DIM, dl) - 2e-6f;
}

// Round 6: separating directions beyond the oriented boxes' fifteen, for a pair (g1, g2 = a hull) that passed them.  A: world rotation of g1, ca /
// a: centre and half extents of its oriented box, B / cb: the hull's, t = A' (cb - ca).  A third of the candidates that reached the narrowphase
// ended in "no intersection" (3-8 us of a wavefront each): a hull whose ORIENTED BOX dips below a box face although no vertex does, the static puck
// and capsule of the scene under the props' pieces, arm links near props.  The hull's extent along a direction comes from its support-bound
// table - a few per cent of its size above the truth instead of the box's tens of per cent.  Conservative: a pair dropped here has a separating
// plane, so no contact changes (rollouts are bit-identical with and without the tables: scripts/gpu_sbt_ab.py).
DEV bool sbt_separated(const DevModel* m, int g1, int g2, const float* A, const float* ca, const float* a, const float* B, const float* cb, const float* t) {
  const float gap = 1e-6f;
  const int t1 = m->geom_type[g1];
  const float* lc2 = m->geom_aabb + 6 * g2;
  float pb[3];                                 // the hull's geom origin: its box centre minus the rotated local centre
#pragma unroll
  for (int i = 0; i < 3; i++) pb[i] = cb[i] - (B[3 * i] * lc2[0] + B[3 * i + 1] * lc2[1] + B[3 * i + 2] * lc2[2]);
  bool sep = false;
  if (t1 == G_BOX) {
    // the three box faces on the hull's side: the hull's lowest point along the face normal against the face
#pragma unroll
    for (int i = 0; i < 3; i++) {
      float sg = t[i] >= 0.f ? 1.f : -1.f;
      float f[3] = {sg * A[i], sg * A[3 + i], sg * A[6 + i]};
      sep = sep || sbt_lowest(m, g2, B, pb, f) - dot3(ca, f) > a[i] + gap;
    }
    return sep;
  }
  // a sphere / capsule / cylinder / another hull: the centre-to-centre direction, the primitive's axis and the radial direction from that axis
  // as candidate separating directions - the primitive's extent in closed form, a hull's from its table
  const float* lc1 = m->geom_aabb + 6 * g1;
  float pa[3];
#pragma unroll
  for (int i = 0; i < 3; i++) pa[i] = ca[i] - (A[3 * i] * lc1[0] + A[3 * i + 1] * lc1[1] + A[3 * i + 2] * lc1[2]);
  const float r1 = m->geom_size[3 * g1], hl1 = m->geom_size[3 * g1 + 1];
  float az[3] = {A[2], A[5], A[8]};                     // the primitive's axis (capsule, cylinder)
  // highest extent of geom 1 along the unit direction d (world): max_x x . d <= this
  auto top1 = [&](const float* d) -> float {
    float along = fabsf(dot3(az, d));
    if (t1 == G_SPHERE) return dot3(pa, d) + r1;
    if (t1 == G_CAPSULE) return dot3(pa, d) + r1 + hl1 * along;
    if (t1 == G_CYLINDER) return dot3(pa, d) + r1 * sqrtf(fmaxf(1.f - along * along, 0.f)) + hl1 * along;
    float nd[3] = {-d[0], -d[1], -d[2]};
    return -sbt_lowest(m, g1, A, pa, nd);               // (a hull: max x . d = - min x . (-d))
  };
  float dirs[3][3]; int nd_ = 1;
  { float w[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]}; float n = sqrtf(dot3(w, w)); float inv = n > 1e-9f ? 1.f / n : 0.f; dirs[0][0] = w[0] * inv; dirs[0][1] = w[1] * inv; dirs[0][2] = w[2] * inv; if (!(n > 1e-9f)) nd_ = 0; }
#pragma unroll
  for (int q = 1; q < 3; q++) { dirs[q][0] = 0.f; dirs[q][1] = 0.f; dirs[q][2] = 0.f; }
  bool has[3] = {nd_ == 1, false, false};
  if (t1 == G_CAPSULE || t1 == G_CYLINDER) {
    float w[3] = {cb[0] - pa[0], cb[1] - pa[1], cb[2] - pa[2]};
    float s_ = dot3(w, az), sg = s_ >= 0.f ? 1.f : -1.f;
    dirs[1][0] = sg * az[0]; dirs[1][1] = sg * az[1]; dirs[1][2] = sg * az[2];
    float rr[3] = {w[0] - s_ * az[0], w[1] - s_ * az[1], w[2] - s_ * az[2]}; float n = sqrtf(dot3(rr, rr)); float inv = n > 1e-9f ? 1.f / n : 0.f;
    dirs[2][0] = rr[0] * inv; dirs[2][1] = rr[1] * inv; dirs[2][2] = rr[2] * inv;
    has[1] = true; has[2] = n > 1e-9f;
  }
#pragma unroll
  for (int q = 0; q < 3; q++)
    if (has[q]) sep = sep || sbt_lowest(m, g2, B, pb, dirs[q]) - top1(dirs[q]) > gap + 2e-6f;
  return sep;
}
// the plane (point pp, unit normal n) against a hull (world rotation B, box centre cb): its lowest point along the normal is above the plane
DEV bool sbt_plane_separated(const DevModel* m, int g2, const float* pp, const float* n, const float* B, const float* cb) {
  const float* lc = m->geom_aabb + 6 * g2;
  float pb[3];
#pragma unroll
  for (int i = 0; i < 3; i++) pb[i] = cb[i] - (B[3 * i] * lc[0] + B[3 * i + 1] * lc[1] + B[3 * i + 2] * lc[2]);
  return sbt_lowest(m, g2, B, pb, n) - dot3(pp, n) > 1e-6f;
}

// Second broadphase pass, lane = candidate: separating-axis test of the two geoms' ORIENTED boxes (15 axes).  The world
// axis-aligned box of a long tilted link overlaps many hulls it is nowhere near; the oriented box is tight.  A pair whose
// oriented boxes are more than 1e-6 m apart cannot touch, so dropping it here changes no contact - it only spares the
// narrowphase a query that would end in "no intersection" (a wavefront's work for a few microseconds; here it costs one
// lane a few hundred instructions).  Plane pairs pass untouched.  The candidate list is compacted in place, order kept.
DEV void obb_filter(const DevModel* m, EnvLDS& L) {
  int lane = wave_lane();
  int ncand = L.ncand, nout = 0;
  for (int k0 = 0; k0 < ncand; k0 += WAVE) {
    int k = k0 + lane;
    bool keep = false; int g1 = 0, g2 = 0;
    if (k < ncand) {
      g1 = L.cand[k][0]; g2 = L.cand[k][1];
      keep = true;
      if (m->geom_type[g1] != G_PLANE) {
        float A[9], ca[3], a[3], B[9], cb[3], b[3];
        geom_obb(m, L, g1, A, ca, a); geom_obb(m, L, g2, B, cb, b);
        // B in A's frame: Rm = A' B, t = A' (cb - ca)
        float Rm[3][3], Ab[3][3], dv[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]}, t[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
          t[i] = A[i] * dv[0] + A[3 + i] * dv[1] + A[6 + i] * dv[2];
#pragma unroll
          for (int j = 0; j < 3; j++) {
            Rm[i][j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
            Ab[i][j] = fabsf(Rm[i][j]) + 1e-6f;
          }
        }
        const float gap = 1e-6f;
        bool sep = false;
#pragma unroll
        for (int i = 0; i < 3; i++) sep = sep || fabsf(t[i]) > a[i] + b[0] * Ab[i][0] + b[1] * Ab[i][1] + b[2] * Ab[i][2] + gap;
#pragma unroll
        for (int j = 0; j < 3; j++)
          sep = sep || fabsf(t[0] * Rm[0][j] + t[1] * Rm[1][j] + t[2] * Rm[2][j]) > a[0] * Ab[0][j] + a[1] * Ab[1][j] + a[2] * Ab[2][j] + b[j] + gap;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) {
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            float ra = a[i1] * Ab[i2][j] + a[i2] * Ab[i1][j], rb = b[j1] * Ab[i][j2] + b[j2] * Ab[i][j1];
            sep = sep || fabsf(t[i2] * Rm[i1][j] - t[i1] * Rm[i2][j]) > ra + rb + gap;
          }
        keep = !sep;
        if (keep && m->hull_sbt && m->geom_type[g2] == G_MESH) keep = !sbt_separated(m, g1, g2, A, ca, a, B, cb, t);
      } else if (m->hull_sbt && m->geom_type[g2] == G_MESH) {
        // the plane against a hull: the same bound along the plane's normal
        float A[9], ca[3], a[3], B[9], cb[3], b[3];
        geom_obb(m, L, g1, A, ca, a); geom_obb(m, L, g2, B, cb, b);
        float pp[3] = {ca[0], ca[1], ca[2]}, n[3] = {A[2], A[5], A[8]};
        if (m->geom_dyn[g1] < 0) { pp[0] = m->geom_pos[3 * g1]; pp[1] = m->geom_pos[3 * g1 + 1]; pp[2] = m->geom_pos[3 * g1 + 2]; }
        keep = !sbt_plane_separated(m, g2, pp, n, B, cb);
      }
    }
    unsigned long long mask = wave_ballot(keep);
    int idx = nout + wave_prefix(mask);
    wave_sync();                                   // every lane has read its candidate before the slots are rewritten
    if (keep) { L.cand[idx][0] = (unsigned short)g1; L.cand[idx][1] = (unsigned short)g2; }
    nout += __popcll(mask);
  }
  wave_sync();
  if (lane == 0) L.ncand = nout;
  wave_sync();
}

// Broadphase: world boxes of all geoms, then the statically filtered pair list is tested lane-parallel; survivors
// are appended to L.cand in pair order.  The pair words of a chunk (PAIR_CHUNK x 64 pairs) are fetched into registers
// with one burst of loads before any of them is used (the loop used to pay one L2 round trip per 64 pairs), and a
// geom's box is two 16-byte LDS reads.
#define PAIR_CHUNK 32
#ifdef SO101_DEBUG_CLOCKS
#define BPROF(k) { unsigned long long pn_ = SO101_CLOCK(); if (wave_lane() == 0) L.nw.prof[k] = (unsigned int)(pn_ - bp_); bp_ = pn_; }
#else
#define BPROF(k)
#endif
DEV void broadphase(const DevModel* m, EnvLDS& L) {
  int lane = wave_lane();
#ifdef SO101_DEBUG_CLOCKS
  unsigned long long bp_ = SO101_CLOCK();
#endif
  for (int g = lane; g < m->ngeom; g += WAVE) {
    const float* ab = m->geom_aabb + 6 * g;
    const float* gp = m->geom_pos + 3 * g; const float* gm = m->geom_mat + 9 * g;
    int d = m->geom_dyn[g];
    float R[9], p[3];
    if (d < 0) {
#pragma unroll
      for (int i = 0; i < 9; i++) R[i] = gm[i];
#pragma unroll
      for (int i = 0; i < 3; i++) p[i] = gp[i];
    } else {
      float lm[9], lp[3] = {gp[0], gp[1], gp[2]};
#pragma unroll
      for (int i = 0; i < 9; i++) lm[i] = gm[i];
      matmul3(R, L.xmat[d], lm);
      float t[3]; matvec3(t, L.xmat[d], lp);
#pragma unroll
      for (int i = 0; i < 3; i++) p[i] = L.xpos[d][i] + t[i];
    }
    float c[3] = {ab[0], ab[1], ab[2]}, h[3] = {ab[3], ab[4], ab[5]}, cw[3];
    matvec3(cw, R, c);
    float blo[3], bhi[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      float e = fabsf(R[3 * i]) * h[0] + fabsf(R[3 * i + 1]) * h[1] + fabsf(R[3 * i + 2]) * h[2];
      blo[i] = p[i] + cw[i] - e; bhi[i] = p[i] + cw[i] + e;
    }
    if (m->geom_type[g] == G_PLANE) {
      // A plane's "box" is the half space it bounds when its normal is a world axis (the floor): another box reaches the
      // plane iff it overlaps that half space - the same test as "lowest corner below the plane", without a special case
      // in the pair loop.  Any other plane gets all of space (its pairs all go on to the narrowphase).
      float n[3] = {R[2], R[5], R[8]};
#pragma unroll
      for (int i = 0; i < 3; i++) {
        bool axis = fabsf(n[i]) == 1.f;
        blo[i] = (axis && n[i] < 0.f) ? p[i] : -3.0e38f;
        bhi[i] = (axis && n[i] > 0.f) ? p[i] : 3.0e38f;
      }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) { L.aabb[i][g] = blo[i]; L.aabb[3 + i][g] = bhi[i]; }
  }
  if (lane == 0) { L.ncand = 0; L.ncon = 0; L.narmcon = 0; }
  wave_sync();
  BPROF(10)
  int base = 0;
  const unsigned int* pairs = ldc(&m->pair_packed);
  const int npair = ldc(&m->npair);
  for (int c0 = 0; c0 < npair; c0 += PAIR_CHUNK * WAVE) {
    unsigned int w[PAIR_CHUNK];
#pragma unroll
    for (int i = 0; i < PAIR_CHUNK; i++) {
      int p = c0 + i * WAVE + lane;
      w[i] = p < npair ? pairs[p] : 0xffffffffu;
    }
    // Phase 1, reads only: the box tests of the whole chunk, one result bit per pair word.  (With the candidate stores in
    // the same loop every iteration waited for its own LDS round trips - the stores may alias the boxes as far as the
    // compiler knows -: 13.5 us per env-substep for 30 iterations; the candidate order is the same either way.)
    unsigned int hits = 0u;
    BPROF(14)
    // straight-line: no branch inside, so the LDS reads of neighbouring pair words overlap (a padding word tests geom 0
    // against itself and is masked out)
#pragma unroll
    for (int i0 = 0; i0 < PAIR_CHUNK; i0 += 4) {
      if (c0 + i0 * WAVE >= npair) break;            // four pair words per basic block (padding words test geom 0 against itself and are masked out)
#pragma unroll
      for (int i = i0; i < i0 + 4; i++) {
        bool valid = w[i] != 0xffffffffu;
        int g1 = valid ? (int)(w[i] & 0xffu) : 0, g2 = valid ? (int)((w[i] >> 8) & 0xffu) : 0;
        float l1[3] = {L.aabb[0][g1], L.aabb[1][g1], L.aabb[2][g1]}, h1[3] = {L.aabb[3][g1], L.aabb[4][g1], L.aabb[5][g1]};
        float l2[3] = {L.aabb[0][g2], L.aabb[1][g2], L.aabb[2][g2]}, h2[3] = {L.aabb[3][g2], L.aabb[4][g2], L.aabb[5][g2]};
        unsigned int sep = (unsigned int)(l1[0] > h2[0]) | (unsigned int)(l2[0] > h1[0]) | (unsigned int)(l1[1] > h2[1]) |
                           (unsigned int)(l2[1] > h1[1]) | (unsigned int)(l1[2] > h2[2]) | (unsigned int)(l2[2] > h1[2]);
        hits |= (valid && sep == 0u) ? (1u << i) : 0u;
      }
    }
    BPROF(15)
    // Phase 2, stores only: survivors appended in pair order
#pragma unroll
    for (int i = 0; i < PAIR_CHUNK; i++) {
      if (c0 + i * WAVE >= npair) break;
      bool hit = (hits >> i) & 1u;
      unsigned long long mask = wave_ballot(hit);
      if (mask == 0ull) continue;
      int idx = base + wave_prefix(mask);
      if (hit && idx < MAXCAND) { L.cand[idx][0] = (unsigned short)(w[i] & 0xffu); L.cand[idx][1] = (unsigned short)((w[i] >> 8) & 0xffu); }
      base += __popcll(mask);
    }
  }
  if (lane == 0) { L.ncand = base < MAXCAND ? base : MAXCAND; if (base > MAXCAND) L.overflow |= 1; }
  wave_sync();
  BPROF(11)
  obb_filter(m, L);
  BPROF(12)
}

// ---- multi-contact for flat faces ("multiccd", so101_sim/tasks/base/so100_task.py:151) --------------------------
// MuJoCo's convex-pair multi-contact re-runs the penetration query on configurations tilted by +-1e-3 rad about the
// two tangent axes and keeps results farther apart than 1e-3 of the smaller bounding radius; its native-ccd path
// clips the aligned faces of box / mesh pairs.  Both sample the extreme points of a flat contact patch.  Here that is
// done in closed form whenever one geom presents a flat REFERENCE FACE - the plane, or the box face / cylinder cap along
// which the pair is shallowest (narrow_pair) - and the other (INCIDENT) geom is sampled through its support function:
//   a_0 = support(-f), a_k = support(-f + eps s_k), s_k = (+-u +- v)/sqrt(2) along the face axes, eps = 1e-3.
// A sample becomes a contact when it is below the face plane, inside the face rectangle / disc and farther than
// 1e-3 min(rbound) from the contacts already accepted; all contacts of the pair share the normal +-f.  When a_0 does
// not qualify, the single MPR contact stays.  Other convex pairs keep one contact.  The supports are wave-parallel,
// the control flow is wave-uniform.  (The test oracle restates the same rule in fp64.)
#define FACE_DEPTH_REL 1e-2f
#define FACE_DEPTH_ABS 1e-6f
#define PATCH_EPS 1e-3f
#define PATCH_DUP 1e-3f

// The five sample points of a flat contact patch in ONE pass over the hull: a_0 = support(-f) and
// a_k = support(-f + eps s_k), s_k = (+-u +- v)/sqrt(2) (see face_patch).  Five separate support calls scan a hull five
// times; the banana's hulls have a thousand vertices, more than the register cache holds, so each scan went back to L2
// twice.  Here every vertex is read once and scored against the five directions; the five winners (max dot, smallest
// index - the same vertex support() would return) are fetched afterwards.
struct Patch5 { float p[NCPP][3]; };

// support points of G in NCPP world directions d[k] (unit), one pass over the hull
template <class Cache, class GP = G64>
DEV void support_multi(const DevModel* m, const GeomW& G, const float (*d)[3], Patch5& P, const Cache& H);

template <class Cache, class GP = G64>
DEV void support_patch(const DevModel* m, const GeomW& G, const float* f, const float* u, const float* v, Patch5& P, const Cache& H) {
  float d[NCPP][3];
#pragma unroll
  for (int k = 0; k < NCPP; k++) {
    float su = (k == 1 || k == 4) ? 1.f : -1.f, sv = (k == 1 || k == 2) ? 1.f : -1.f;
    float e = k == 0 ? 0.f : PATCH_EPS * 0.70710678f;
#pragma unroll
    for (int i = 0; i < 3; i++) d[k][i] = -f[i] + e * (su * u[i] + sv * v[i]);
    normalize3(d[k]);
  }
  support_multi<Cache, GP>(m, G, d, P, H);
}

template <class Cache, class GP>
DEV void support_multi(const DevModel* m, const GeomW& G, const float (*d)[3], Patch5& P, const Cache& H) {
  if (G.type != G_MESH) {
#pragma unroll
    for (int k = 0; k < NCPP; k++) support<Cache, GP>(m, G, d[k], P.p[k], H);
    return;
  }
  float dl[NCPP][3], best[NCPP];
  int bi[NCPP];
#pragma unroll
  for (int k = 0; k < NCPP; k++) { matTvec3(dl[k], G.R, d[k]); best[k] = -3.0e38f; bi[k] = 0x7fffffff; }
  int lane = GP::sub();
  const float* x = ldc(&m->vx) + G.vadr; const float* y = ldc(&m->vy) + G.vadr; const float* z = ldc(&m->vz) + G.vadr;
  int first = lane;
  if constexpr (is_hull_sub<Cache>::value) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      int i = H.i[q];
#pragma unroll
      for (int k = 0; k < NCPP; k++) {
        float s = H.x[q] * dl[k][0] + H.y[q] * dl[k][1] + H.z[q] * dl[k][2];
        if (i < G.vnum && s > best[k]) { best[k] = s; bi[k] = i; }
      }
    }
    first = G.vnum;
  } else if constexpr (is_hull_lds<Cache>::value) {
    const float4* X4 = (const float4*)H.p; const float4* Y4 = (const float4*)(H.p + H.n); const float4* Z4 = (const float4*)(H.p + 2 * H.n);
    auto block = [&](int J) {
      float4 xv = X4[GP::N * J + lane], yv = Y4[GP::N * J + lane], zv = Z4[GP::N * J + lane];
      float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w}, zs[4] = {zv.x, zv.y, zv.z, zv.w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        int i = 4 * GP::N * J + 4 * lane + q;
#pragma unroll
        for (int k = 0; k < NCPP; k++) {
          float s = xs[q] * dl[k][0] + ys[q] * dl[k][1] + zs[q] * dl[k][2];
          if (i < G.vnum && s > best[k]) { best[k] = s; bi[k] = i; }
        }
      }
    };
    if constexpr (GP::N == WAVE) {
#pragma unroll
      for (int J = 0; J < HULL_LDS_MAX / (4 * WAVE); J++) {
        if (4 * WAVE * J >= G.vnum || 4 * WAVE * J >= H.n) break;
        block(J);
      }
    } else {
      const int lim = G.vnum < H.n ? G.vnum : H.n;
#pragma unroll 1
      for (int J = 0; 4 * GP::N * J < lim; J++) block(J);
    }
    first = lane + H.n;
  } else if constexpr (sizeof(Cache) >= sizeof(HullCache)) {
#pragma unroll
    for (int j = 0; j < HULL_K; j++) {
      if (GP::N * j >= G.vnum) break;
      int i = lane + GP::N * j;
      float X = H.x[j], Y = H.y[j], Z = H.z[j];
#pragma unroll
      for (int k = 0; k < NCPP; k++) {
        float s = X * dl[k][0] + Y * dl[k][1] + Z * dl[k][2];
        if (i < G.vnum && s > best[k]) { best[k] = s; bi[k] = i; }
      }
    }
    first = lane + GP::N * HULL_K;
  }
#pragma unroll 4
  for (int i = first; i < G.vnum; i += GP::N) {
    float X = x[i], Y = y[i], Z = z[i];
#pragma unroll
    for (int k = 0; k < NCPP; k++) {
      float s = X * dl[k][0] + Y * dl[k][1] + Z * dl[k][2];
      if (s > best[k]) { best[k] = s; bi[k] = i; }
    }
  }
  // the five reductions first, then the five winners' coordinates in one burst of loads (fetching each winner right
  // after its reduction put five L2 round trips in series), then the transforms
#pragma unroll
  for (int k = 0; k < NCPP; k++) GP::argmax(best[k], bi[k]);
  float loc[NCPP][3];
#pragma unroll
  for (int k = 0; k < NCPP; k++) {
    int w = (unsigned int)bi[k] < (unsigned int)G.vnum ? bi[k] : 0;       // (a non-finite direction selects nothing)
    if constexpr (is_hull_lds<Cache>::value) {
      if (w < H.n) { loc[k][0] = H.p[w]; loc[k][1] = H.p[H.n + w]; loc[k][2] = H.p[2 * H.n + w]; continue; }      // (the staged copy: same floats)
    }
    loc[k][0] = x[w]; loc[k][1] = y[w]; loc[k][2] = z[w];
  }
#pragma unroll
  for (int k = 0; k < NCPP; k++) {
    float wv[3];
    matvec3(wv, G.R, loc[k]);
    P.p[k][0] = G.p[0] + wv[0]; P.p[k][1] = G.p[1] + wv[1]; P.p[k][2] = G.p[2] + wv[2];
  }
}

// contacts of one geom pair: slot k holds patch sample k (or the single MPR contact in slot 0) when bit k of `valid` is
// set - fixed slots instead of a compacted list: "store at the running count" is register indexing, i.e. scratch memory
struct PairContacts { unsigned int valid; float nrm[3], dist[NCPP], pos[NCPP][3]; };

// inside the face outline (rectangle hu x hv, or disc of radius hu when hv < 0) by at least `margin`
DEV bool inside_margin(const float* rel, const float* u, const float* v, float hu, float hv, float margin) {
  float pu = dot3(rel, u), pv = dot3(rel, v), ru = hu - margin;
  if (hv >= 0.f) return fabsf(pu) <= ru && fabsf(pv) <= hv - margin;
  return ru >= 0.f && pu * pu + pv * pv <= ru * ru;
}

DEV bool inside_face(const float* rel, const float* u, const float* v, float hu, float hv) {
  if (hu < 0.f) return true;                                          // unbounded plane
  float pu = dot3(rel, u), pv = dot3(rel, v);
  return hv >= 0.f ? (fabsf(pu) <= hu && fabsf(pv) <= hv) : (pu * pu + pv * pv <= hu * hu);      // rectangle / disc
}



DEV bool face_patch(const Patch5& P, const float* f, const float* c, const float* u, const float* v, float hu, float hv,
                    float dup_tol, PairContacts& out) {
  out.valid = 0u;
#pragma unroll
  for (int k = 0; k < NCPP; k++) {
    const float* p = P.p[k];
    float rel[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
    float dist = dot3(rel, f);
    bool ok = dist < 0.f;
    ok = ok && inside_face(rel, u, v, hu, hv);
    if (k == 0 && !ok) return false;
    float cp[3] = {p[0] - 0.5f * dist * f[0], p[1] - 0.5f * dist * f[1], p[2] - 0.5f * dist * f[2]};
#pragma unroll
    for (int j = 0; j < k; j++) {
      float dd[3] = {cp[0] - out.pos[j][0], cp[1] - out.pos[j][1], cp[2] - out.pos[j][2]};
      if (((out.valid >> j) & 1u) && sqrtf(dot3(dd, dd)) < dup_tol) ok = false;
    }
    out.dist[k] = dist; out.pos[k][0] = cp[0]; out.pos[k][1] = cp[1]; out.pos[k][2] = cp[2];
    if (ok) out.valid |= 1u << k;
  }
  return true;
}

// flat face number `axis` of box / cylinder G on the side that `toward` (world, any length) points to (box: local
// x/y/z; cylinder: only axis 2, the cap): outward normal f, centre c, in-plane axes u/v with half extents (hv < 0: disc
// of radius hu).  Returns false when the geom has no such face.
DEV bool flat_face(const GeomW& G, int axis, const float* toward, float* f, float* c, float* u, float* v, float* hu, float* hv, float* half) {
  float loc[3]; matTvec3(loc, G.R, toward);
  if (G.type == G_CYLINDER) {
    if (axis != 2) return false;
    float sg = loc[2] >= 0.f ? 1.f : -1.f;
#pragma unroll
    for (int k = 0; k < 3; k++) { f[k] = sg * G.R[3 * k + 2]; u[k] = G.R[3 * k]; v[k] = G.R[3 * k + 1]; c[k] = G.p[k] + f[k] * G.size[1]; }
    *hu = G.size[0]; *hv = -1.f; *half = G.size[1];
    return true;
  }
  // axis picks as 0/1 weights (exact arithmetic; chains of selects on the index get turned into indexed loads of a
  // stack copy of the geom, i.e. scratch memory)
  float w0 = axis == 0 ? 1.f : 0.f, w1 = axis == 1 ? 1.f : 0.f, w2 = axis == 2 ? 1.f : 0.f;
  float li = w0 * loc[0] + w1 * loc[1] + w2 * loc[2];
  float sg = li >= 0.f ? 1.f : -1.f;
  // u axis = (axis + 1) % 3 -> weights (w2, w0, w1); v axis = (axis + 2) % 3 -> weights (w1, w2, w0)
  float si = w0 * G.size[0] + w1 * G.size[1] + w2 * G.size[2];
  *half = si;
  *hu = w2 * G.size[0] + w0 * G.size[1] + w1 * G.size[2];
  *hv = w1 * G.size[0] + w2 * G.size[1] + w0 * G.size[2];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    f[k] = sg * (w0 * G.R[3 * k] + w1 * G.R[3 * k + 1] + w2 * G.R[3 * k + 2]);
    u[k] = w2 * G.R[3 * k] + w0 * G.R[3 * k + 1] + w1 * G.R[3 * k + 2];
    v[k] = w1 * G.R[3 * k] + w2 * G.R[3 * k + 1] + w0 * G.R[3 * k + 2];
    c[k] = G.p[k] + f[k] * si;
  }
  return true;
}

// The direction (world, not normalised) of the FIRST support query that narrow_pair_cached() makes on the hull G2 of a light pair - the plane's normal
// negated, or the outward normal, negated, of the box face scan_faces() visits first (same expressions as there) - and the cell of G2's support-vertex
// lists it falls into.  Shared by the wavefront that writes the work item and the one that serves it.  -1: no such query (not a plane / box against a hull).
DEV int light_first_cell(const GeomW& G1, const GeomW& G2) {
  if (G2.type != G_MESH) return -1;
  float f[3];
  if (G1.type == G_PLANE) { f[0] = G1.R[2]; f[1] = G1.R[5]; f[2] = G1.R[8]; }
  else if (G1.type == G_BOX) {
    float toward[3] = {G2.c[0] - G1.c[0], G2.c[1] - G1.c[1], G2.c[2] - G1.c[2]};
    float loc[3]; matTvec3(loc, G1.R, toward);
    float lb0 = G1.size[0] - fabsf(loc[0]), lb1 = G1.size[1] - fabsf(loc[1]), lb2 = G1.size[2] - fabsf(loc[2]);
    int axis = 0; float bl = lb0;
    if (lb1 < bl) { axis = 1; bl = lb1; }
    if (lb2 < bl) { axis = 2; bl = lb2; }
    if (!(bl < 3.0e38f)) return -1;                      // (non-finite bounds: the full query)
    float w0 = axis == 0 ? 1.f : 0.f, w1 = axis == 1 ? 1.f : 0.f, w2 = axis == 2 ? 1.f : 0.f;
    float li = w0 * loc[0] + w1 * loc[1] + w2 * loc[2], sg = li >= 0.f ? 1.f : -1.f;
#pragma unroll
    for (int k = 0; k < 3; k++) f[k] = sg * (w0 * G1.R[3 * k] + w1 * G1.R[3 * k + 1] + w2 * G1.R[3 * k + 2]);
  } else return -1;
  float nf[3] = {-f[0], -f[1], -f[2]}, dl[3];
  matTvec3(dl, G2.R, nf);
  if (!(fabsf(dl[0]) + fabsf(dl[1]) + fabsf(dl[2]) > 0.5f)) return -1;      // (a diverged pose)
  return hl_cell(dl);
}

#ifdef SO101_DEBUG_CLOCKS
#define QPROF(k) { unsigned long long qn_ = SO101_CLOCK(); if (prof && wave_lane() == 0) atomicAdd(&prof[k], (unsigned int)(qn_ - qp_)); qp_ = qn_; }
#else
#define QPROF(k)
#endif
struct FaceRef { float f[3], c[3], u[3], v[3], hu, hv, depth; int side; bool exact, separated; Patch5 P; };

// Flat-face scan of GR (box / cylinder) against the incident geom GI, before any iterative query.  For every flat face
// on the side of GI's centre, a0 = GI's deepest point below the face plane, d0 its depth:
//  * d0 <= 0: the face plane separates the pair - no contact, exactly (R.separated);
//  * a0 inside the face outline with a lateral margin >= d0, and d0 <= the half thickness behind the face: a0 is a point
//    of the box at distance d0 from the box's boundary, so no translation shorter than d0 separates the pair and the
//    translation d0 along the face normal does - minimum penetration depth d0 along the face normal, EXACTLY, and no
//    iterative query is needed (R.exact: props resting on the table top, finger pads, the static puck);
//  * a0 merely inside the outline (d0 <= half thickness): a CANDIDATE; the shallowest one is kept in R and later wins
//    over MPR's answer when it is not deeper (narrow_pair).
// ONE_FACE (k_narrow's fast path, HullSub): only the face visited first - the incident hull's subset is valid for that face's normal alone;
// a pair that face does not settle (neither separated nor exact) is handed back to the full query.
template <class Cache, class GP = G64, bool ONE_FACE = false>
DEV void scan_faces(const DevModel* m, const GeomW& GR, const GeomW& GI, const Cache& HI, int side, FaceRef& R, unsigned int* prof = nullptr) {
  if (GR.type != G_BOX && GR.type != G_CYLINDER) return;
#ifdef SO101_DEBUG_CLOCKS
  unsigned long long qp_ = SO101_CLOCK();
#endif
  float toward[3] = {GI.c[0] - GR.c[0], GI.c[1] - GR.c[1], GI.c[2] - GR.c[2]};
  // Visiting order: increasing depth of the incident's centre below the face plane (= half extent along the axis minus
  // |centre offset along it|, a lower bound of d0), i.e. the face the incident geom sticks out of first - for a prop on
  // the table the top face.  The scan ends at the first EXACT face: the geoms then share the point a0, so no other face
  // plane separates them, and another exact face would give the same depth.  (Saves two of three hull scans for every
  // hull resting on the table; the result does not depend on the order otherwise.)
  float loc[3]; matTvec3(loc, GR.R, toward);
  float lb0 = GR.size[0] - fabsf(loc[0]), lb1 = GR.size[1] - fabsf(loc[1]), lb2 = GR.size[2] - fabsf(loc[2]);
  unsigned int done = 0u;
#pragma unroll 1
  for (int it = 0; it < (ONE_FACE ? 1 : 3); it++) {
    if (R.separated || R.exact) break;
    int axis = 0; float bl = 3.0e38f;
    if (!(done & 1u)) { axis = 0; bl = lb0; }
    if (!(done & 2u) && lb1 < bl) { axis = 1; bl = lb1; }
    if (!(done & 4u) && lb2 < bl) { axis = 2; bl = lb2; }
    if (bl == 3.0e38f) axis = (done & 1u) ? ((done & 2u) ? 2 : 1) : 0;          // (non-finite bounds: plain index order)
    done |= 1u << axis;
    float f[3], c[3], u[3], v[3], hu, hv, half;
    if (!flat_face(GR, axis, toward, f, c, u, v, &hu, &hv, &half)) continue;
    float cr[3] = {c[0] - GI.c[0], c[1] - GI.c[1], c[2] - GI.c[2]};
    if (dot3(cr, f) > half) continue;                  // d0 >= depth of the incident's centre > half thickness
    float nf[3] = {-f[0], -f[1], -f[2]}, a0[3];
    QPROF(6)
    support<Cache, GP>(m, GI, nf, a0, HI);
    QPROF(7)
    float rel[3] = {a0[0] - c[0], a0[1] - c[1], a0[2] - c[2]};
    float d0 = -dot3(rel, f);
    if (!(d0 > 0.f)) { R.separated = true; continue; }
    if (d0 > half || !inside_face(rel, u, v, hu, hv)) continue;
    bool ex = inside_margin(rel, u, v, hu, hv, d0);
    if (!ex && !(d0 < R.depth)) continue;
    R.exact = ex; R.depth = d0; R.side = side; R.hu = hu; R.hv = hv;
#pragma unroll
    for (int k = 0; k < 3; k++) { R.f[k] = f[k]; R.c[k] = c[k]; R.u[k] = u[k]; R.v[k] = v[k]; }
    // the face is (so far) the reference face: its five patch samples in one more pass over the incident hull
    // (taking them in the same pass that finds a0 for the face visited first measured no faster: 1.257 M vs 1.261 M
    // env-steps/s at 32768 envs - the passes over the hull are not what a candidate's 5-10 us go into)
    Patch5 P;
    QPROF(8)
    support_patch<Cache, GP>(m, GI, f, u, v, P, HI);
#pragma unroll
    for (int k = 0; k < NCPP; k++) { R.P.p[k][0] = P.p[k][0]; R.P.p[k][1] = P.p[k][1]; R.P.p[k][2] = P.p[k][2]; }
    QPROF(9)
  }
}

// Hull against hull (round 5; the reference runs with multiccd, so100_task.py:151, aloha2_task.py:197): behind the EPA contact (slot 0) the
// extreme points of whatever flat feature each hull presents along the contact normal n (geom 1 -> geom 2).  With w1 = pos + depth/2 n on
// geom 1's surface and w2 = pos - depth/2 n on geom 2's:
//   b_k = support_2(-n + eps s_k), a_k = support_1(+n + eps s_k), k = 1..4 (the tilted samples of support_patch);
//   b_k is a contact when it lies below geom 1's supporting plane (through w1) and, with r the unit lateral direction from w1 to b_k,
//   r . (b_k - w1) <= r . (support_1(n + eps r) - w1) + 1e-6 - inside the extent of geom 1's feature in that direction, again by a tilted
//   support (a vertex or a curved patch has extent 0: nothing beyond the EPA contact survives); samples laterally closer than dup_tol to
//   w1 are skipped (they would repeat the EPA contact); a_k likewise against geom 2 at w2.
// Accepted in the order b_1..b_4, a_1..a_4 while farther than dup_tol from those already accepted, NCPP in all; normal n for all, distance
// = the sample's signed distance to the other hull's plane, position = the midpoint.  Four more passes over the hulls (two of samples, two
// of extents), only for mesh pairs that EPA found in contact.  (The test suite holds an fp64 restatement of the rule.)
template <class Cache, class GP = G64>
DEV void hull_patch(const DevModel* m, const GeomW& G1, const GeomW& G2, const Cache& H1, const Cache& H2, const float* n, float depth, const float* pos,
                    float dup_tol, PairContacts& out) {
  float fr[9] = {n[0], n[1], n[2], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  make_frame(fr);
  const float* u = fr + 3; const float* v = fr + 6;
  float nn[3] = {-n[0], -n[1], -n[2]};
  float w1[3], w2[3];
#pragma unroll
  for (int i = 0; i < 3; i++) { w1[i] = pos[i] + 0.5f * depth * n[i]; w2[i] = pos[i] - 0.5f * depth * n[i]; }
  int count = 1;                                       // (slot 0: the EPA contact, already in `out`)
#pragma unroll 1
  for (int side = 0; side < 2; side++) {
    if (count >= NCPP) break;
    const float sg = side == 0 ? -1.f : 1.f;
    GeomW GS, GO; Cache HS, HO;
    select_geom(side == 0, G2, G1, GS); select_geom(side == 0, G1, G2, GO);
    select_hull(side == 0, H2, H1, HS); select_hull(side == 0, H1, H2, HO);
    float wo[3] = {side == 0 ? w1[0] : w2[0], side == 0 ? w1[1] : w2[1], side == 0 ? w1[2] : w2[2]};
    Patch5 S;
    support_patch<Cache, GP>(m, GS, side == 0 ? n : nn, u, v, S, HS);        // slots 1..4: support(sg n + eps s_k)
    float dist[4], rl[4], de[NCPP][3], r[4][3];
    bool cand[4];
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float* p = S.p[k + 1];
      float rel[3] = {p[0] - wo[0], p[1] - wo[1], p[2] - wo[2]};
      float h = dot3(rel, n);
      dist[k] = -sg * h;
      cand[k] = dist[k] < 0.f;
      r[k][0] = rel[0] - h * n[0]; r[k][1] = rel[1] - h * n[1]; r[k][2] = rel[2] - h * n[2];
      rl[k] = normalize3(r[k]);
      // (a sample laterally closer than dup_tol to the EPA witness would only repeat the EPA contact: a vertex or an edge end - skipped
      //  before the extent pass, which is then not run at all for a hull that presents a vertex)
      cand[k] = cand[k] && rl[k] >= dup_tol;
      bool ext = cand[k];
      any = any || ext;
#pragma unroll
      for (int i = 0; i < 3; i++) de[k + 1][i] = ext ? -sg * n[i] + PATCH_EPS * r[k][i] : -sg * n[i];
      normalize3(de[k + 1]);
    }
#pragma unroll
    for (int i = 0; i < 3; i++) de[0][i] = -sg * n[i];
    Patch5 E;
#pragma unroll
    for (int k = 0; k < NCPP; k++) { E.p[k][0] = wo[0]; E.p[k][1] = wo[1]; E.p[k][2] = wo[2]; }
    if (any) support_multi<Cache, GP>(m, GO, de, E, HO);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      bool ok = cand[k] && count < NCPP;
      if (ok) {
        const float* e = E.p[k + 1];
        float ext = r[k][0] * (e[0] - wo[0]) + r[k][1] * (e[1] - wo[1]) + r[k][2] * (e[2] - wo[2]);
        ok = rl[k] <= ext + 1e-6f;
      }
      const float* p = S.p[k + 1];
      float cp[3] = {p[0] + sg * 0.5f * dist[k] * n[0], p[1] + sg * 0.5f * dist[k] * n[1], p[2] + sg * 0.5f * dist[k] * n[2]};
#pragma unroll
      for (int j = 0; j < NCPP; j++) {
        float dd[3] = {cp[0] - out.pos[j][0], cp[1] - out.pos[j][1], cp[2] - out.pos[j][2]};
        if (((out.valid >> j) & 1u) && sqrtf(dot3(dd, dd)) < dup_tol) ok = false;
      }
      if (ok) {
#pragma unroll
        for (int t = 1; t < NCPP; t++)                 // (slot = count: selects instead of a dynamic register index)
          if (count == t) { out.dist[t] = dist[k]; out.pos[t][0] = cp[0]; out.pos[t][1] = cp[1]; out.pos[t][2] = cp[2]; }
        out.valid |= 1u << count;
        count++;
      }
    }
  }
}

// Narrowphase of one candidate pair (geom types ordered): up to NCPP contacts sharing one normal (geom1 -> geom2),
// each with its penetration distance (< 0) and position.
// narrow_pair_cached: the caches H1 / H2 are ready (k_narrow stages them in LDS), rb1 / rb2 = the geoms' bounding radii
// FACES_ONLY (k_narrow's row pass, policy G16: four pairs per wavefront, one per DPP row): the plane and flat-face closed forms only; returns
// false when the pair needs the iterative query (MPR / EPA), which the caller then runs with the whole wavefront.  Otherwise returns true.
template <class Cache, class GP = G64, bool FACES_ONLY = false, bool ONE_FACE = false>
DEV bool narrow_pair_cached(const DevModel* m, const GeomW& G1, const GeomW& G2, float rb1, float rb2, const Cache& H1, const Cache& H2, PairContacts& out,
                            unsigned int* prof = nullptr) {
  static_assert(!ONE_FACE || FACES_ONLY, "the one-face scan has no iterative query behind it");
#ifdef SO101_DEBUG_CLOCKS
  unsigned long long qp_ = SO101_CLOCK();
#endif
  out.valid = 0u; out.nrm[0] = out.nrm[1] = out.nrm[2] = 0.f;
#pragma unroll
  for (int j = 0; j < NCPP; j++) { out.dist[j] = 0.f; out.pos[j][0] = out.pos[j][1] = out.pos[j][2] = 0.f; }
  if (G1.type == G_PLANE) {
    float fr[9] = {G1.R[2], G1.R[5], G1.R[8], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    make_frame(fr);
    Patch5 P;
    support_patch<Cache, GP>(m, G2, fr, fr + 3, fr + 6, P, H2);
    face_patch(P, fr, G1.p, fr + 3, fr + 6, -1.f, -1.f, PATCH_DUP * rb2, out);
    out.nrm[0] = fr[0]; out.nrm[1] = fr[1]; out.nrm[2] = fr[2];
    return true;
  }
  FaceRef best;
  best.depth = 3.0e38f; best.side = -1; best.hu = 0.f; best.hv = 0.f; best.exact = false; best.separated = false;
#pragma unroll
  for (int k = 0; k < 3; k++) { best.f[k] = 0.f; best.c[k] = 0.f; best.u[k] = 0.f; best.v[k] = 0.f; }
#pragma unroll
  for (int k = 0; k < NCPP; k++) { best.P.p[k][0] = 0.f; best.P.p[k][1] = 0.f; best.P.p[k][2] = 0.f; }
  scan_faces<Cache, GP, ONE_FACE>(m, G1, G2, H2, 0, best, prof);
  if constexpr (!ONE_FACE) { if (!best.separated && !best.exact) scan_faces<Cache, GP>(m, G2, G1, H1, 1, best, prof); }
  QPROF(2)
  if (best.separated) return true;
  float depth = 0.f, nrm[3] = {0.f, 0.f, 0.f}, pos[3] = {0.f, 0.f, 0.f};
  if constexpr (FACES_ONLY) { if (!best.exact) return false; }
  else if (!best.exact) {
    if constexpr (is_hull_lds<Cache>::value) {
      float* pk = narrow_park_store();
      if (wave_lane() == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { pk[k] = best.f[k]; pk[3 + k] = best.c[k]; pk[6 + k] = best.u[k]; pk[9 + k] = best.v[k]; }
        pk[12] = best.hu; pk[13] = best.hv; pk[14] = best.depth; pk[15] = __int_as_float(best.side);
#pragma unroll
        for (int k = 0; k < NCPP; k++) { pk[16 + 3 * k] = best.P.p[k][0]; pk[17 + 3 * k] = best.P.p[k][1]; pk[18 + 3 * k] = best.P.p[k][2]; }
      }
      wave_sync();
    }
    // MPR's depth is the depth along ITS final portal normal, which for a thin plate (finger pad) against a hull can be
    // an oblique direction ten times deeper than the plate's face normal: the shallowest face candidate wins when it
    // is not deeper (1 % + 1e-6 m slack: for a face contact both are the same number)
    bool ok = mpr_penetration<Cache, GP>(m, G1, G2, &depth, nrm, pos, H1, H2);
    if (!ok || !(depth > 0.f)) return true;
    if constexpr (is_hull_lds<Cache>::value) {
      wave_sync();
      const float* pk = narrow_park_store();
#pragma unroll
      for (int k = 0; k < 3; k++) { best.f[k] = pk[k]; best.c[k] = pk[3 + k]; best.u[k] = pk[6 + k]; best.v[k] = pk[9 + k]; }
      best.hu = pk[12]; best.hv = pk[13]; best.depth = pk[14]; best.side = __float_as_int(pk[15]);
#pragma unroll
      for (int k = 0; k < NCPP; k++) { best.P.p[k][0] = pk[16 + 3 * k]; best.P.p[k][1] = pk[17 + 3 * k]; best.P.p[k][2] = pk[18 + 3 * k]; }
    }
    if (best.side >= 0 && !(best.depth <= depth * (1.f + FACE_DEPTH_REL) + FACE_DEPTH_ABS)) best.side = -1;
  }
  QPROF(3)
  int ref = best.side;
  bool patched = false;
  if (ref >= 0) patched = face_patch(best.P, best.f, best.c, best.u, best.v, best.hu, best.hv, PATCH_DUP * fminf(rb1, rb2), out);
  QPROF(13)
  const float* f = best.f;
  if (patched) {
    float sg = ref == 0 ? 1.f : -1.f;
    out.nrm[0] = sg * f[0]; out.nrm[1] = sg * f[1]; out.nrm[2] = sg * f[2];
  } else if (best.exact) {
    out.valid = 0u;
  } else {
    out.valid = 1u; out.dist[0] = -depth;
#pragma unroll
    for (int k = 0; k < 3; k++) { out.nrm[k] = nrm[k]; out.pos[0][k] = pos[k]; }
#if !defined(SO101_NO_HULL_PATCH) && !defined(SO101_MPR)      // (the MPR option keeps the single contact: its portal normal is no face normal of the Minkowski difference; NO_HULL_PATCH: kernel experiments)
    if constexpr (!FACES_ONLY) {
      // (inlined.  Measured in k_narrow, round 5, env-steps/s at 4096 envs: at three wavefronts per SIMD the patch code costs 96 more spilled
      //  VGPRs on the common path - 656 k against 738 k without it -, out of line behind a call with its arguments in LDS 537 k; at two
      //  wavefronts per SIMD nothing spills: 730 k)
      if (G1.type == G_MESH && G2.type == G_MESH) hull_patch<Cache, GP>(m, G1, G2, H1, H2, nrm, depth, pos, PATCH_DUP * fminf(rb1, rb2), out);
    }
#endif
  }
  QPROF(14)
  return true;
}

template <class Cache, class GP = G64>
DEV void narrow_pair(const DevModel* m, const GeomW& G1, const GeomW& G2, int g1, int g2, PairContacts& out, unsigned int* prof = nullptr) {
  static_assert(!is_hull_lds<Cache>::value, "the LDS cache is staged by its kernel: narrow_pair_cached");
#ifdef SO101_DEBUG_CLOCKS
  unsigned long long qp_ = SO101_CLOCK();
#endif
  Cache H1, H2;
  hull_load<GP>(m, G1, H1); hull_load<GP>(m, G2, H2);
  QPROF(1)
  float rb1 = GP::ld(ldc(&m->geom_rbound) + g1), rb2 = GP::ld(ldc(&m->geom_rbound) + g2);
  narrow_pair_cached<Cache, GP>(m, G1, G2, rb1, rb2, H1, H2, out, prof);
}

// Contact record of an accepted pair (one lane): frame, body indices, mixed friction / solref / solimp
DEV void contact_init(const DevModel* m, Contact& c, int g1, int g2, float dist, const float* nrm, const float* pos) {
  c.dist = dist;
  float fr[9] = {nrm[0], nrm[1], nrm[2], 0, 0, 0, 0, 0, 0};
  make_frame(fr);
#pragma unroll
  for (int i = 0; i < 9; i++) c.frame[i] = fr[i];
#pragma unroll
  for (int i = 0; i < 3; i++) c.pos[i] = pos[i];
  c.d1 = m->geom_dyn[g1]; c.d2 = m->geom_dyn[g2]; c.g1 = g1; c.g2 = g2;
  int cd1 = m->geom_condim[g1], cd2 = m->geom_condim[g2];
  c.dim = cd1 > cd2 ? cd1 : cd2;
#pragma unroll
  for (int i = 0; i < 3; i++) c.fric[i] = fmaxf(m->geom_friction[3 * g1 + i], m->geom_friction[3 * g2 + i]);
  // solref / solimp mixed with equal weights (solmix = 1 on every geom of these scenes); stash in aref/f
  c.aref[0] = 0.5f * (m->geom_solref[2 * g1] + m->geom_solref[2 * g2]);
  c.aref[1] = 0.5f * (m->geom_solref[2 * g1 + 1] + m->geom_solref[2 * g2 + 1]);
#pragma unroll
  for (int i = 0; i < 5; i++) c.f[i] = 0.5f * (m->geom_solimp[5 * g1 + i] + m->geom_solimp[5 * g2 + i]);
  c.armslot = -1;
}

// Fused collision stage: the wave walks its own candidate list.  (The pipelined step hands the candidates to
// k_narrow instead, one wavefront per candidate.)
// More contacts than MAXCON (round 5; until round 4 the tail of the list was cut off): the env goes over to ONE contact per geom pair - the
// first of each pair's patch: the deepest point of a flat patch, the EPA contact of a hull pair - for this substep, so that no touching pair
// loses its contact; flag 128 (event 7, "contacts_reduced").  Only when the touching PAIRS alone exceed MAXCON is the list cut (flag 2,
// "contact_overflow").  The launch chains' gather_contacts() applies the same rule from the pairs' counts, so both step paths keep the
// same contacts in the same order.
DEV int reduce_contacts(EnvLDS& L, int ncon) {
  int lane = wave_lane(), out = 0;
  wave_sync();
  for (int j = 0; j < ncon; j++) {
    bool first = j == 0 || L.con[j].g1 != L.con[j - 1].g1 || L.con[j].g2 != L.con[j - 1].g2;      // (a pair's contacts are consecutive)
    if (first) {
      if (out != j && lane < (int)(sizeof(Contact) / 4)) ((int*)&L.con[out])[lane] = ((const int*)&L.con[j])[lane];
      out++;
    }
    wave_sync();
  }
  return out;
}
DEV void collision(const DevModel* m, EnvLDS& L) {
  int lane = wave_lane();
  broadphase(m, L);
  int ncand = L.ncand, ncon = 0;
  bool full = false, reduced = false;
  for (int k = 0; k < ncand && !full; k++) {
    int g1 = L.cand[k][0], g2 = L.cand[k][1];
    GeomW G1, G2;
    load_geom(m, L, g1, G1); load_geom(m, L, g2, G2);
    PairContacts pc;
    narrow_pair<NoCache>(m, G1, G2, g1, g2, pc);
    int cnt = __popc(pc.valid);
    if (!reduced && ncon + cnt > MAXCON) {
      if (lane == 0) L.overflow |= 128;
      ncon = reduce_contacts(L, ncon);
      reduced = true;
    }
    bool taken = false;                                // (reduced: the pair's first contact only)
#pragma unroll
    for (int j = 0; j < NCPP; j++) {
      if (((pc.valid >> j) & 1u) && !full && !(reduced && taken)) {
        if (ncon >= MAXCON) { if (lane == 0) L.overflow |= 2; full = true; }
        else { if (lane == 0) contact_init(m, L.con[ncon], g1, g2, pc.dist[j], pc.nrm, pc.pos[j]); ncon++; taken = true; }
      }
    }
  }
  if (lane == 0) L.ncon = ncon;
  wave_sync();
}

// ------------------------------------------------------------------ constraint rows
// x^power for the solimp sigmoid with a power other than MuJoCo's default 2: ONE out-of-line copy (the inlined
// ocml powf is ~1100 instructions and impedance() is expanded at three call sites)
static __device__ __attribute__((noinline)) float impedance_pow(float x, float power) { return powf(x, power); }

DEV float impedance(const float* solimp, float pos) {
  float dmin = fminf(fmaxf(solimp[0], MINIMP_F), MAXIMP_F), dmax = fminf(fmaxf(solimp[1], MINIMP_F), MAXIMP_F);
  float width = fmaxf(solimp[2], 0.f), mid = fminf(fmaxf(solimp[3], MINIMP_F), MAXIMP_F), power = fmaxf(solimp[4], 1.f);
  if (dmin == dmax || width <= MINVAL_F) return 0.5f * (dmin + dmax);
  float x = fabsf(pos) / width;
  if (x >= 1.f) return dmax;
  if (x <= 0.f) return dmin;
  float y;
  if (power == 1.f) y = x;
  else if (power == 2.f) y = x <= mid ? x * x / mid : 1.f - (1.f - x) * (1.f - x) / (1.f - mid);
  else if (x <= mid) y = impedance_pow(x, power) / impedance_pow(mid, power - 1.f);
  else y = 1.f - impedance_pow(1.f - x, power) / impedance_pow(1.f - mid, power - 1.f);
  return dmin + y * (dmax - dmin);
}

DEV void kb_from_solref(const DevModel* m, const float* solref_in, const float* solimp, float* K, float* B) {
  float s0 = solref_in[0], s1 = solref_in[1];
  float dmax = fminf(fmaxf(solimp[1], MINIMP_F), MAXIMP_F);
  if (s0 > 0.f) {
    s0 = fmaxf(s0, 2.f * m->dt);     // refsafe
    *K = 1.f / fmaxf(MINVAL_F, dmax * dmax * s0 * s0 * s1 * s1);
    *B = 2.f / fmaxf(MINVAL_F, dmax * s0);
  } else {
    *K = -s0 / fmaxf(MINVAL_F, dmax * dmax);
    *B = -s1 / fmaxf(MINVAL_F, dmax);
  }
}

// Jacobian row of contact axis `u` (translational or rotational) against arm link `link`: out[d], d<=link
DEV void arm_jac_row(const EnvLDS& L, int link, const float* p, const float* u, bool rot, float* out) {
#pragma unroll
  for (int d = 0; d < NARM; d++) {
    float v = 0.f;
    if (d <= link) {
      const float* a = L.axis[d];
      if (rot) v = dot3(u, a);
      else {
        float r[3] = {p[0] - L.xpos[d][0], p[1] - L.xpos[d][1], p[2] - L.xpos[d][2]}, t[3];
        cross3(t, a, r);
        v = dot3(u, t);
      }
    }
    out[d] = v;
  }
}

// row j (0-2 translational, 3-5 rotational, along the contact frame's axes) of an arm-link contact's Jacobian in the six arm dofs:
// J(link of geom2) - J(link of geom1); either side may be static or a free body (link -1: a zero row)
DEV void arm_contact_row(const EnvLDS& L, const Contact& c, int j, float* Jd) {
  int l1 = (c.d1 >= 0 && c.d1 < NARM) ? c.d1 : -1, l2 = (c.d2 >= 0 && c.d2 < NARM) ? c.d2 : -1;
  float Jr[NARM], J1[NARM];
  arm_jac_row(L, l2, c.pos, &c.frame[3 * (j % 3)], j >= 3, Jr);
  arm_jac_row(L, l1, c.pos, &c.frame[3 * (j % 3)], j >= 3, J1);
#pragma unroll
  for (int q = 0; q < NARM; q++) Jd[q] = Jr[q] - J1[q];
}

// Constraint rows of the current contacts: scalar rows (dof frictionloss, joint limits), and per contact the
// regularisers R, the cone parameter mu, the reference accelerations and - for contacts that touch an arm link - the
// joint-space Jacobian rows in the LDS pool.  What only PGS needs (the diagonal blocks of A) is built by solve_pgs().
DEV void make_constraints(const DevModel* m, EnvLDS& L) {
  int lane = wave_lane();
  // ---- scalar rows: frictionloss (dof order) then active joint limits (joint order).  Lane d < 6 proposes dof d's
  // frictionloss row, lane 8 + 2 h + side the limit row of joint h; the active ones are compacted in lane order, which is
  // the list order above.  (One lane used to build the list in a loop: six impedance / solref evaluations in series.)
  {
    bool active = false;
    Row1 r; r.dof = 0; r.sign = 0.f; r.R = 1.f; r.aref = 0.f; r.floss = 0.f; r.f = 0.f; r.Ainv = 0.f; r.pad = 0.f;
    if (lane < NARM) {
      int d = lane;
      if (m->frictionloss[d] > 0.f) {
        float imp = impedance(m->dof_solimp, 0.f), K, B;
        kb_from_solref(m, m->dof_solref, m->dof_solimp, &K, &B);
        r.dof = d; r.sign = 1.f; r.floss = m->frictionloss[d];
        r.R = fmaxf(MINVAL_F, (1.f - imp) * m->dof_invweight0[d] / imp);
        r.aref = -B * L.qvel[d];
        r.f = 0.f; r.Ainv = 1.f / (L.Minv[d][d] + r.R);
        active = true;
      }
    } else if (lane >= 8 && lane < 8 + 2 * NARM) {
      int h = (lane - 8) >> 1, side = (lane - 8) & 1;
      if (m->limited[h]) {
        float q = L.qpos[h];
        float pos = side == 0 ? q - m->range[h][0] : m->range[h][1] - q;
        if (pos < 0.f) {
          float imp = impedance(m->jnt_solimp, pos), K, B;
          kb_from_solref(m, m->jnt_solref, m->jnt_solimp, &K, &B);
          r.dof = h; r.sign = side == 0 ? 1.f : -1.f; r.floss = 0.f;
          r.R = fmaxf(MINVAL_F, (1.f - imp) * m->dof_invweight0[h] / imp);
          r.aref = -B * r.sign * L.qvel[h] - K * imp * pos;
          r.f = 0.f; r.Ainv = 1.f / (L.Minv[h][h] + r.R);
          active = true;
        }
      }
    }
    unsigned long long mask = wave_ballot(active);
    if (active) L.row[wave_prefix(mask)] = r;
    // arm-pool slots for contacts that touch an arm link (in contact order): lane = contact
    bool arm = false;
    if (lane < L.ncon) { const Contact& c = L.con[lane]; arm = (c.d1 >= 0 && c.d1 < NARM) || (c.d2 >= 0 && c.d2 < NARM); }
    unsigned long long amask = wave_ballot(arm);
    if (arm) {
      int slot = wave_prefix(amask);
      // slots beyond the LDS pool keep their number: such a contact's Jacobian is not parked in the pool but computed again by the lane
      // that owns the contact when the Newton solver loads it (conreg_load: the same lane, the same expressions, the same bits), so NO arm
      // contact is ever dropped (round 5; until round 4 the tail of the pool was cut off and counted).  The PGS kernels stop at 32 contacts
      // and never get here.
      L.con[lane].armslot = slot;
    }
    int narm = __popcll(amask);
    if (lane == 0) {
      L.nrow = __popcll(mask);
      L.narmcon = narm < MAXARMCON ? narm : MAXARMCON;
    }
  }
  wave_sync();
  // ---- contact rows: lane = contact
  int ncon = L.ncon;
  if (lane < ncon) {
    Contact& c = L.con[lane];
    float solref[2] = {c.aref[0], c.aref[1]}, solimp[5] = {c.f[0], c.f[1], c.f[2], c.f[3], c.f[4]};
    float imp = impedance(solimp, c.dist), K, B;
    kb_from_solref(m, solref, solimp, &K, &B);
    float tran = 0.f;      // body_invweight0 of a free prop scales with 1 / mass scale
    if (c.d1 >= 0) tran += m->dyn_invweight0[c.d1][0] / (c.d1 >= NARM ? L.fscale[c.d1 - NARM] : 1.f);
    if (c.d2 >= 0) tran += m->dyn_invweight0[c.d2][0] / (c.d2 >= NARM ? L.fscale[c.d2 - NARM] : 1.f);
    float R0 = fmaxf(MINVAL_F, (1.f - imp) * tran / imp);
    float R1 = R0 / fmaxf(MINVAL_F, m->impratio);
    float mu0 = c.fric[0];
    c.R[0] = R0; c.R[1] = R1;
    c.R[2] = fmaxf(MINVAL_F, R1 * mu0 * mu0 / (c.fric[1] * c.fric[1]));
    c.R[3] = fmaxf(MINVAL_F, R1 * mu0 * mu0 / (c.fric[2] * c.fric[2]));
    c.mu = mu0 * sqrtf(R1 / R0);
    // row velocities J qvel: translational rows j<3 use frame[j] at the contact point, rotational rows frame[j-3]
    float vel[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int side = 0; side < 2; side++) {
      int d = side == 0 ? c.d1 : c.d2;
      float sgn = side == 0 ? -1.f : 1.f;
      if (d >= NARM) {
        int f = d - NARM;
        float r[3] = {c.pos[0] - L.xipos[d][0], c.pos[1] - L.xipos[d][1], c.pos[2] - L.xipos[d][2]};
        float wr[3]; cross3(wr, &L.fvel[f][3], r);
        float pv[3] = {L.fvel[f][0] + wr[0], L.fvel[f][1] + wr[1], L.fvel[f][2] + wr[2]};      // velocity of the contact point
#pragma unroll
        for (int j = 0; j < 3; j++) { vel[j] += sgn * dot3(&c.frame[3 * j], pv); vel[3 + j] += sgn * dot3(&c.frame[3 * j], &L.fvel[f][3]); }
      }
    }
    if (c.armslot >= 0) {
      // Arm part, ONCE per contact: J = J(link of geom2) - J(link of geom1) in the 6 arm dofs (either side may
      // be static or a free body; for arm-arm self-collision both contribute), kept dof-major in this contact's
      // slot of the LDS pool.
      const bool pooled = c.armslot < MAXARMCON;
      ArmCon& ac = L.armcon[pooled ? c.armslot : 0];
#pragma unroll
      for (int j = 0; j < 6; j++) {
        float Jd[NARM];
        arm_contact_row(L, c, j, Jd);
        float vj = 0.f;
#pragma unroll
        for (int q = 0; q < NARM; q++) { if (pooled) ac.Jt[q][j] = Jd[q]; vj += Jd[q] * L.qvel[q]; }
        vel[j] += vj;
      }
    }
#pragma unroll
    for (int j = 0; j < 6; j++) {
      c.aref[j] = -B * vel[j] - (j == 0 ? K * imp * c.dist : 0.f);
      c.f[j] = 0.f;
    }
  }
  wave_sync();
}

#include "so101_solver.hpp"
#include "so101_newton.hpp"

// ------------------------------------------------------------------ forward + Euler
// `phases` is a profiling aid of so101_physics (StepParams.phases, 7 = everything): bit0 collision, bit1 constraint
// rows + solve, bit2 solve iterations.
DEV void forward_smooth(const DevModel* m, EnvLDS& L) {
  kinematics(m, L);
  crba_arm(m, L);
  smooth_dynamics(m, L);
}

// solver coordinates -> MuJoCo's generalized accelerations
DEV void forward_accelerations(EnvLDS& L) {
  int lane = wave_lane();
  if (lane < NARM) L.qacc[lane] = L.qacc_arm[lane];
  if (lane >= 32 && lane < 32 + NFREE) {
    int f = lane - 32, b = NARM + f;
    float r[3] = {L.xipos[b][0] - L.xpos[b][0], L.xipos[b][1] - L.xpos[b][1], L.xipos[b][2] - L.xpos[b][2]};
    float al[3] = {L.facc[f][3], L.facc[f][4], L.facc[f][5]}, ww[3] = {L.fvel[f][3], L.fvel[f][4], L.fvel[f][5]};
    float t1[3], t2[3], ab[3];
    cross3(t1, al, r); cross3(t2, ww, r); cross3(t2, ww, t2);
    matTvec3(ab, L.xmat[b], al);
#pragma unroll
    for (int i = 0; i < 3; i++) { L.qacc[NARM + 6 * f + i] = L.facc[f][i] - t1[i]; L.qacc[NARM + 6 * f + 3 + i] = ab[i]; }
  }
  wave_sync();
}


// constraint rows + solve for the contacts in L.con, then back to MuJoCo's generalized accelerations.
// SOLVER is a compile-time choice (SO101_SOLVER_NEWTON / SO101_SOLVER_PGS): every kernel exists once per solver, so
// the Newton kernels carry no PGS code (and no PGS-only constraint data) and vice versa.
template <int SOLVER>
DEV void forward_constrained(const DevModel* m, EnvLDS& L, int max_iter, float tolerance, int phases) {
  if (phases & 2) {
    make_constraints(m, L);
    if constexpr (SOLVER == 1) solve_newton(m, L, (phases & 4) ? max_iter : 0, tolerance);
    else solve_pgs(m, L, (phases & 4) ? max_iter : 0, tolerance);
  }
  forward_accelerations(L);
}

template <int SOLVER>
DEV void forward(const DevModel* m, EnvLDS& L, int max_iter, float tolerance, int phases = 7) {
  forward_smooth(m, L);
  unsigned long long t0 = SO101_CLOCK();
  if (phases & 1) collision(m, L);
  else { if (wave_lane() == 0) { L.ncand = 0; L.ncon = 0; L.narmcon = 0; } wave_sync(); }
  unsigned long long t1 = SO101_CLOCK();
  forward_constrained<SOLVER>(m, L, max_iter, tolerance, phases);
  if (SO101_CLOCKS_ON && wave_lane() == 0) { L.t_collision += (unsigned int)(t1 - t0); L.t_solve += (unsigned int)(SO101_CLOCK() - t1); }
}

DEV void euler(const DevModel* m, EnvLDS& L) {
  int lane = wave_lane();
  float dt = m->dt;
  if (lane < NV) { L.qvel[lane] += dt * L.qacc[lane]; L.warm[lane] = L.qacc[lane]; }
  wave_sync();
  if (lane < NARM) L.qpos[lane] += dt * L.qvel[lane];
  if (lane >= 32 && lane < 32 + NFREE) {
    int f = lane - 32;
    float* q = &L.qpos[NARM + 7 * f]; const float* v = &L.qvel[NARM + 6 * f];
    q[0] += dt * v[0]; q[1] += dt * v[1]; q[2] += dt * v[2];
    float w[3] = {v[3], v[4], v[5]};
    float ang = dt * normalize3(w);
    float sn, cs; sincos_f(0.5f * ang, &sn, &cs);
    float dq[4] = {cs, w[0] * sn, w[1] * sn, w[2] * sn};
    float qq[4] = {q[3], q[4], q[5], q[6]};
    mulquat(qq, qq, dq);
    normquat(qq);
    q[3] = qq[0]; q[4] = qq[1]; q[5] = qq[2]; q[6] = qq[3];
  }
  wave_sync();
}

// mj_checkPos / mj_checkVel / mj_checkAcc: a NaN or |x| > 1e10 anywhere in the state means the simulation
// diverged; MuJoCo resets the data to qpos0, dm_control (raise_exception_on_physics_error=False,
// so101_sim/task_suite.py:153) ends the episode with reward 0 and discount 0.  Returns true when diverged.
DEV bool check_divergence(EnvLDS& L) {
  int lane = wave_lane();
  bool bad = false;
  if (lane < NQ) { float x = L.qpos[lane]; bad = bad || !(fabsf(x) <= 1e10f); }
  if (lane < NV) { float x = L.qvel[lane], y = L.qacc[lane]; bad = bad || !(fabsf(x) <= 1e10f) || !(fabsf(y) <= 1e10f); }
  bool any = wave_ballot(bad) != 0ull;
  if (any) {
    wave_sync();
    if (lane < NQ) L.qpos[lane] = (lane == NARM + 3 || lane == NARM + 10) ? 1.f : 0.f;
    if (lane < NV) { L.qvel[lane] = 0.f; L.warm[lane] = 0.f; L.qacc[lane] = 0.f; }
    if (lane == 0) L.overflow |= 8;
    wave_sync();
  }
  return any;
}

template <int SOLVER>
DEV bool substep(const DevModel* m, EnvLDS& L, int max_iter, float tolerance, bool freeze_arm, int phases = 7) {
  forward<SOLVER>(m, L, max_iter, tolerance, phases);
  euler(m, L);
  if (check_divergence(L)) return true;
  if (freeze_arm) {   // dm_control JointStaticIsolator: non-prop joints restored after every step
    int lane = wave_lane();
    if (lane < NARM) { L.qpos[lane] = L.arm0_q[lane]; L.qvel[lane] = L.arm0_v[lane]; }
    wave_sync();
  }
  return false;
}

// ------------------------------------------------------------------ reward (uniform): so100_hand_over.py:238-275
struct BoxW { float pos[3], quat[4], half[3]; };

DEV bool overlap_aabb_oobb(const float* half0, const BoxW& b) {
  float R[9]; quat2mat(R, b.quat);
  // 6 face axes only, strict inequalities (oobb_utils.py:223-246); projections of the 8 corners reduce to centre +- extent
  bool sep = false;
#pragma unroll
  for (int a = 0; a < 6; a++) {
    float ax[3];
    if (a < 3) { ax[0] = a == 0; ax[1] = a == 1; ax[2] = a == 2; }
    else { ax[0] = R[a - 3]; ax[1] = R[3 + a - 3]; ax[2] = R[6 + a - 3]; }
    float mx0 = -3e38f, mn0 = 3e38f, mx1 = -3e38f, mn1 = 3e38f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      float sx = (i & 1) ? 1.f : -1.f, sy = (i & 2) ? 1.f : -1.f, sz = (i & 4) ? 1.f : -1.f;
      float av[3] = {sx * half0[0], sy * half0[1], sz * half0[2]};
      float lv[3] = {sx * b.half[0], sy * b.half[1], sz * b.half[2]}, ov[3];
      matvec3(ov, R, lv);
      ov[0] += b.pos[0]; ov[1] += b.pos[1]; ov[2] += b.pos[2];
      float p0 = dot3(av, ax), p1 = dot3(ov, ax);
      mx0 = fmaxf(mx0, p0); mn0 = fminf(mn0, p0); mx1 = fmaxf(mx1, p1); mn1 = fminf(mn1, p1);
    }
    if (mx0 < mn1 || mn0 > mx1) sep = true;
  }
  return !sep;
}

DEV bool overlap_oobb_oobb(const BoxW& b0, const BoxW& b1) {
  float inv[4] = {b0.quat[0], -b0.quat[1], -b0.quat[2], -b0.quat[3]};
  float dp[3] = {b1.pos[0] - b0.pos[0], b1.pos[1] - b0.pos[1], b1.pos[2] - b0.pos[2]};
  BoxW r;
  rotvecquat(r.pos, dp, inv);
  mulquat(r.quat, inv, b1.quat);
  r.half[0] = b1.half[0]; r.half[1] = b1.half[1]; r.half[2] = b1.half[2];
  return overlap_aabb_oobb(b0.half, r);
}

// requires kinematics() of the current qpos to be in LDS
DEV float task_reward(const DevModel* m, const EnvLDS& L) {
  // any_props_moving: linear part only, >= 1e-3 (success_detector_utils.py:22-28)
#pragma unroll
  for (int f = 0; f < NFREE; f++) {
    const float* v = &L.qvel[NARM + 6 * f];
    float mx = fmaxf(fabsf(v[0]), fmaxf(fabsf(v[1]), fabsf(v[2])));
    if (mx >= 1e-3f) return 0.f;
  }
  int ob = NARM + 0, cb = NARM + 1;
  BoxW o;
  float im[9], xim[9];
  quat2mat(im, m->free_iquat[0]);
  matmul3(xim, L.xmat[ob], im);
  mat2quat(o.quat, xim);
  float ctr[3]; rotvecquat(ctr, m->free_bvh[0], o.quat);
#pragma unroll
  for (int i = 0; i < 3; i++) { o.pos[i] = ctr[i] + L.xipos[ob][i]; o.half[i] = m->free_bvh[0][3 + i]; }
  const float* cq_ = &L.qpos[NARM + 7 + 3];
  float cq[4] = {cq_[0], cq_[1], cq_[2], cq_[3]};
  normquat(cq);
  for (int k = 0; k < m->nbox; k++) {
    BoxW cw;
    float r[3]; rotvecquat(r, m->box_pos[k], cq);
#pragma unroll
    for (int i = 0; i < 3; i++) { cw.pos[i] = L.xpos[cb][i] + r[i]; cw.half[i] = m->box_half[k][i]; }
    float ident[4] = {1.f, 0.f, 0.f, 0.f};
    mulquat(cw.quat, cq, ident);
    if (!overlap_oobb_oobb(o, cw)) return 0.f;
  }
  return 1.f;
}

// ------------------------------------------------------------------ counter RNG (Philox4x32-10), 24-bit uniforms
DEV float rng_uniform(unsigned long long seed, unsigned long long env, unsigned int episode, unsigned int draw) {
  unsigned int c0 = (unsigned int)env, c1 = (unsigned int)(env >> 32), c2 = episode, c3 = draw;
  unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; r++) {
    unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned int)p1, n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned int)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}
