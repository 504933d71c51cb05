// Newton constraint solver (mujoco's default, mj_solNewton) for one env per wavefront.
//
//   min_x  0.5 (x - x_s)' M (x - x_s) + sum_blocks s_b(J_b x - aref_b)
//
// in the solver coordinates x = [arm qacc (6) | object twist (6) | container twist (6)], where M is block diagonal:
// the 6x6 arm matrix and, per free body, (m I3, I_world).
//
// Layout: lane k < ncon owns contact k for the whole solve - its Jacobian (12 local columns: the one or two coordinate
// groups the contact touches), regularisers and reference accelerations live in that lane's registers (ConReg); lane
// r < nrow additionally owns scalar row r (dof frictionloss / joint limit).  Everything that couples blocks is a
// wave-level sum over lanes (DPP adds, no LDS, no barrier):
//   gradient  g_d      = (M (x - x_s))_d - sum_k (J_k' f_k)_d                      18 sums
//   Hessian   H_(a,b)  = M_(a,b) + sum_k (J_k' Hc_k J_k)_(a,b)                      <= 171 sums, lane a keeps row a
// followed by the register-resident Cholesky (lane i owns row i of H, pivot rows broadcast with v_readlane).  The
// per-contact LDS staging of the previous version (three barrier rounds per contact and iteration) is gone; the only
// LDS traffic left is the 18-vector x / search direction that every lane reads, and one transposition of the factor.
// The minimiser is unique, so results agree with the fp64 oracle at solution level (not iterate level).
#pragma once

struct ConReg {
  float J[12][6];            // J[c][j]: row j, local column c (0-5: group g0, 6-11: group g1)
  float Dj[6], fr[5], aref[6], mu;
  int dim, g0, g1;           // coordinate groups: 0 arm, 1 object, 2 container, -1 none; g0 < g1 when both are present
};

// block cost s(r), force = -ds/dr and (optionally) the symmetric block Hessian d2s/dr2 (packed lower triangle)
DEV float contact_cost(const ConReg& c, const float* r, float* force, float* Hc, bool want_h, int* zone) {
  int dim = c.dim;
  if (want_h) {
#pragma unroll
    for (int k = 0; k < 21; k++) Hc[k] = 0.f;
  }
  float mu = c.mu, U[6], T = 0.f;
  U[0] = r[0] * mu;
#pragma unroll
  for (int j = 1; j < 6; j++) { U[j] = (j < dim) ? r[j] * c.fr[j - 1] : 0.f; T += U[j] * U[j]; }
  T = sqrtf(T);
  float N = U[0];
  if (dim == 0 || (N >= mu * T) || (T <= 0.f && N >= 0.f)) {           // top zone (or dropped contact): free
#pragma unroll
    for (int j = 0; j < 6; j++) force[j] = 0.f;
    *zone = 0;
    return 0.f;
  }
  if ((mu * N + T <= 0.f) || (T <= 0.f && N < 0.f)) {                  // bottom zone: quadratic in every row
    float cost = 0.f;
#pragma unroll
    for (int j = 0; j < 6; j++) {
      float D = (j < dim) ? c.Dj[j] : 0.f;
      force[j] = -D * r[j]; cost += 0.5f * D * r[j] * r[j];
      if (want_h) Hc[j * (j + 1) / 2 + j] = D;
    }
    *zone = 1;
    return cost;
  }
  float Dm = c.Dj[0] / fmaxf(mu * mu * (1.f + mu * mu), MINVAL_F), sN = N - mu * T, iT = 1.f / T;   // middle zone
  force[0] = -Dm * sN * mu;
#pragma unroll
  for (int j = 1; j < 6; j++) force[j] = (j < dim) ? -force[0] * iT * U[j] * c.fr[j - 1] : 0.f;
  if (want_h) {
    float a = Dm * mu * mu, b = Dm * sN * mu;
    Hc[0] = a;
#pragma unroll
    for (int k = 1; k < 6; k++) {
      float wk = (k < dim) ? U[k] * c.fr[k - 1] * iT : 0.f;          // mu_k u_k / T
      Hc[k * (k + 1) / 2] = -a * wk;
#pragma unroll
      for (int l = 1; l <= k; l++) {
        float wl = (l < dim) ? U[l] * c.fr[l - 1] * iT : 0.f;
        float diag = (k == l && k < dim) ? c.fr[k - 1] * c.fr[k - 1] * iT : 0.f;
        Hc[k * (k + 1) / 2 + l] = a * wk * wl - b * (diag - wk * wl * iT);
      }
    }
  }
  *zone = 2;
  return 0.5f * Dm * sN * sN;
}

// first and second derivative of the block cost along the line r + alpha dr (closed form per zone; the zone is that
// of the point itself, as in mj_solNewton's line search)
DEV void contact_line(const ConReg& c, const float* r, const float* dr, float* d1, float* d2) {
  int dim = c.dim;
  float mu = c.mu, N = r[0] * mu, dN = dr[0] * mu, TT = 0.f, UdU = 0.f, dUdU = 0.f;
#pragma unroll
  for (int j = 1; j < 6; j++) {
    float u = (j < dim) ? r[j] * c.fr[j - 1] : 0.f, du = (j < dim) ? dr[j] * c.fr[j - 1] : 0.f;
    TT += u * u; UdU += u * du; dUdU += du * du;
  }
  float T = sqrtf(TT);
  *d1 = 0.f; *d2 = 0.f;
  if (dim == 0 || (N >= mu * T) || (T <= 0.f && N >= 0.f)) return;
  if ((mu * N + T <= 0.f) || (T <= 0.f && N < 0.f)) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < 6; j++) { float D = (j < dim) ? c.Dj[j] : 0.f; a += D * r[j] * dr[j]; b += D * dr[j] * dr[j]; }
    *d1 = a; *d2 = b;
    return;
  }
  float Dm = c.Dj[0] / fmaxf(mu * mu * (1.f + mu * mu), MINVAL_F), sN = N - mu * T, iT = 1.f / T;
  float dT = UdU * iT, ddT = (dUdU - dT * dT) * iT, dS = dN - mu * dT;
  *d1 = Dm * sN * dS;
  *d2 = Dm * (dS * dS - sN * mu * ddT);
}

DEV float row_cost(const Row1& r, float jar, float* force, float* h) {
  float D = 1.f / r.R;
  if (r.floss > 0.f) {
    float rf = r.R * r.floss;
    if (jar <= -rf) { *force = r.floss; *h = 0.f; return r.floss * (-0.5f * rf - jar); }
    if (jar >= rf) { *force = -r.floss; *h = 0.f; return r.floss * (-0.5f * rf + jar); }
    *force = -D * jar; *h = D; return 0.5f * D * jar * jar;
  }
  if (jar < 0.f) { *force = -D * jar; *h = D; return 0.5f * D * jar * jar; }
  *force = 0.f; *h = 0.f; return 0.f;
}

// the contact's Jacobian and parameters, from the LDS record into the lane's registers
// coordinate groups a contact touches (0 arm, 1 object, 2 container; -1 none; g0 < g1 when both are present)
DEV void contact_groups(const Contact& c, int* g0, int* g1) {
  int a = c.armslot >= 0 ? 0 : 3, f1 = c.d1 >= NARM ? c.d1 - NARM + 1 : 3, f2 = c.d2 >= NARM ? c.d2 - NARM + 1 : 3;   // 3 = none
  int lo = min(a, min(f1, f2)), hi = max(a == 3 ? -1 : a, max(f1 == 3 ? -1 : f1, f2 == 3 ? -1 : f2));
  *g0 = lo == 3 ? -1 : lo;
  *g1 = hi > lo ? hi : -1;
}

template <bool SINGLE>
DEV void conreg_load(const EnvLDS& L, const Contact& c, ConReg& r) {
  r.dim = c.dim; r.mu = c.mu;
  r.fr[0] = c.fric[0]; r.fr[1] = c.fric[0]; r.fr[2] = c.fric[1]; r.fr[3] = c.fric[2]; r.fr[4] = c.fric[2];
  r.Dj[0] = 1.f / c.R[0]; r.Dj[1] = 1.f / c.R[1]; r.Dj[2] = r.Dj[1]; r.Dj[3] = 1.f / c.R[2]; r.Dj[4] = 1.f / c.R[3]; r.Dj[5] = r.Dj[4];
#pragma unroll
  for (int j = 0; j < 6; j++) r.aref[j] = c.aref[j];
  contact_groups(c, &r.g0, &r.g1);
#pragma unroll
  for (int s = 0; s < (SINGLE ? 1 : 2); s++) {
    int g = s == 0 ? r.g0 : r.g1;
    if (g == 0) {
      if (c.armslot < MAXARMCON) {
        const ArmCon& ac = L.armcon[c.armslot];
#pragma unroll
        for (int q = 0; q < 6; q++)
#pragma unroll
          for (int j = 0; j < 6; j++) r.J[6 * s + q][j] = ac.Jt[q][j];
      } else {
        // beyond the LDS pool (more than MAXARMCON arm-link contacts in this env: props wedged under the arm): the rows again, by the
        // expressions make_constraints() used for this contact's reference accelerations
#pragma unroll 1
        for (int j = 0; j < 6; j++) {
          float Jd[NARM];
          arm_contact_row(L, c, j, Jd);
#pragma unroll
          for (int q = 0; q < 6; q++) {
#pragma unroll
            for (int jj = 0; jj < 6; jj++) if (jj == j) r.J[6 * s + q][jj] = Jd[q];
          }
        }
      }
    } else if (g > 0) {
      int d = NARM + g - 1;
      float sgn = (c.d2 == d) ? 1.f : -1.f;
      float rr[3] = {c.pos[0] - L.xipos[d][0], c.pos[1] - L.xipos[d][1], c.pos[2] - L.xipos[d][2]};
#pragma unroll
      for (int j = 0; j < 3; j++) {
        float u[3] = {sgn * c.frame[3 * j], sgn * c.frame[3 * j + 1], sgn * c.frame[3 * j + 2]}, t[3];
        cross3(t, rr, u);
#pragma unroll
        for (int q = 0; q < 3; q++) {
          r.J[6 * s + q][j] = u[q]; r.J[6 * s + 3 + q][j] = t[q];
          r.J[6 * s + q][3 + j] = 0.f; r.J[6 * s + 3 + q][3 + j] = u[q];
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 6; q++)
#pragma unroll
        for (int j = 0; j < 6; j++) r.J[6 * s + q][j] = 0.f;
    }
  }
}

// M * v in solver coordinates, element `lane` (lane < NVS)
DEV float mass_times(const DevModel* m, const EnvLDS& L, const float* v /*LDS*/, int lane) {
  if (lane < NARM) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NARM; c++) s += L.Marm[lane][c] * v[c];
    return s;
  }
  int f = (lane - NARM) / 6, k = (lane - NARM) % 6;
  if (k < 3) return L.fmass[f] * v[lane];
  const float* I = L.Iw[NARM + f];
  const float* w = &v[NARM + 6 * f + 3];
  float o[3]; symvec3(o, I, w);
  return k == 3 ? o[0] : (k == 4 ? o[1] : o[2]);
}

// entry (a, b) of the block-diagonal M
DEV float mass_entry(const DevModel* m, const EnvLDS& L, int a, int b) {
  if (a < NARM) return b < NARM ? L.Marm[a][b] : 0.f;
  if (b < NARM || (a - NARM) / 6 != (b - NARM) / 6) return 0.f;
  int f = (a - NARM) / 6, i = (a - NARM) % 6, j = (b - NARM) % 6;
  if (i < 3 && j < 3) return i == j ? L.fmass[f] : 0.f;
  if (i >= 3 && j >= 3) {
    int p = i - 3, q = j - 3;
    return L.Iw[NARM + f][p == q ? p : p + q + 2];        // packed xx yy zz xy xz yz: (0,1)->3, (0,2)->4, (1,2)->5
  }
  return 0.f;
}

// SINGLE: no contact couples two coordinate groups (g1 < 0 everywhere), so H = M + J' Hc J is block diagonal - three
// independent 6x6 systems.  Slot 1 of every ConReg is never touched (half the Jacobian registers and products), lane
// i keeps only the six entries of row i inside its block, and the three blocks are factorised and solved side by side:
// six pivot steps instead of eighteen on the dependent chain.
template <bool ROW0, bool SINGLE>
DEV void solve_newton_impl(const DevModel* m, EnvLDS& L, int max_iter, float tolerance) {
  int lane = wave_lane();
  int nrow = L.nrow, ncon = L.ncon;
  NewtonScratch& W = L.nw;
#ifdef SO101_DEBUG_CLOCKS
  unsigned long long pc = SO101_CLOCK();
  if (lane < 8) W.prof[lane] = 0u;
#define NPROF(k) { unsigned long long pn = SO101_CLOCK(); if (lane == 0) W.prof[k] += (unsigned int)(pn - pc); pc = pn; }
#else
#define NPROF(k)
#endif
  if (lane == 0) L.iters = 0;
  if (nrow + ncon == 0) { wave_sync(); return; }
  bool has_con = lane < ncon, has_row = lane < nrow;
  // sums over contact / scalar-row lanes only: with at most 16 of each, lanes 16-63 add exact zeros (wave.hpp)
  auto csum = [&](float v) -> float { return wave_sum_rows_f(v, ROW0); };
  ConReg C;
  // lanes without a contact carry an all-zero block (every field is read by the wave-wide arithmetic below)
  C.dim = 0; C.g0 = -1; C.g1 = -1; C.mu = 0.f;
#pragma unroll
  for (int c = 0; c < 12; c++)
#pragma unroll
    for (int j = 0; j < 6; j++) C.J[c][j] = 0.f;
#pragma unroll
  for (int j = 0; j < 6; j++) { C.Dj[j] = 0.f; C.aref[j] = 0.f; }
#pragma unroll
  for (int j = 0; j < 5; j++) C.fr[j] = 0.f;
  if (has_con) conreg_load<SINGLE>(L, L.con[lane], C);
#ifdef SO101_EMU_TRACE
  if (has_con) fprintf(stderr, "  con %d dim %d g %d %d R %.6g %.6g %.6g %.6g aref %.6g %.6g %.6g %.6g %.6g %.6g mu %.6g\n", lane, C.dim, C.g0, C.g1,
                       L.con[lane].R[0], L.con[lane].R[1], L.con[lane].R[2], L.con[lane].R[3], C.aref[0], C.aref[1], C.aref[2], C.aref[3], C.aref[4], C.aref[5], C.mu);
#endif
  Row1 rreg; rreg.dof = 0; rreg.sign = 0.f; rreg.R = 1.f; rreg.aref = 0.f; rreg.floss = 0.f; rreg.f = 0.f; rreg.Ainv = 0.f; rreg.pad = 0.f;
  if (has_row) rreg = L.row[lane];
  // which coordinate groups / group pairs any contact touches (wave-uniform): sums over absent blocks are skipped
  bool anyG[3], anyX[3];
#pragma unroll
  for (int G = 0; G < 3; G++) anyG[G] = wave_ballot(has_con && (C.g0 == G || C.g1 == G)) != 0ull;
  anyX[0] = wave_ballot(has_con && C.g0 == 0 && C.g1 == 1) != 0ull;
  anyX[1] = wave_ballot(has_con && C.g0 == 0 && C.g1 == 2) != 0ull;
  anyX[2] = wave_ballot(has_con && C.g0 == 1 && C.g1 == 2) != 0ull;
  // x_s (smooth) and the warm start in solver coordinates; x lives in W.x
  if (lane < NARM) { W.xs[lane] = L.qacc_arm[lane]; W.xw[lane] = L.warm[lane]; }
  if (lane >= 32 && lane < 32 + NFREE) {
    int f = lane - 32, b = NARM + f;
    const float* wq = &L.warm[NARM + 6 * f];
    float wb[3] = {wq[3], wq[4], wq[5]}, alp[3];
    matvec3(alp, L.xmat[b], wb);
    float r[3] = {L.xipos[b][0] - L.xpos[b][0], L.xipos[b][1] - L.xpos[b][1], L.xipos[b][2] - L.xpos[b][2]}, t1[3];
    cross3(t1, alp, r);
#pragma unroll
    for (int i = 0; i < 3; i++) {
      W.xs[NARM + 6 * f + i] = L.facc[f][i]; W.xs[NARM + 6 * f + 3 + i] = L.facc[f][3 + i];
      W.xw[NARM + 6 * f + i] = wq[i] + t1[i]; W.xw[NARM + 6 * f + 3 + i] = alp[i];
    }
  }
  wave_sync();
  // J * v of this lane's contact for an LDS vector v in solver coordinates
  auto block_jx = [&](const float* x, float* out6) {
    int o0 = C.g0 < 0 ? 0 : 6 * C.g0, o1 = C.g1 < 0 ? 0 : 6 * C.g1;
#pragma unroll
    for (int j = 0; j < 6; j++) out6[j] = 0.f;
#pragma unroll
    for (int q = 0; q < 6; q++) {
      float x0 = x[o0 + q], x1 = x[o1 + q];
#pragma unroll
      for (int j = 0; j < 6; j++) { if constexpr (SINGLE) out6[j] += C.J[q][j] * x0; else out6[j] += C.J[q][j] * x0 + C.J[6 + q][j] * x1; }
    }
  };
  // total cost at x; leaves this lane's residuals (jar, rjar), forces and block Hessians in registers
  float jar[6] = {0, 0, 0, 0, 0, 0}, force[6] = {0, 0, 0, 0, 0, 0}, Hc[21], rjar = 0.f, rforce = 0.f, rh = 0.f, mxd = 0.f;
  int zone = 0;
#pragma unroll
  for (int k = 0; k < 21; k++) Hc[k] = 0.f;
  auto eval_cost = [&](const float* x, bool want_h) -> float {
    if (lane < NVS) W.tmp[lane] = x[lane] - W.xs[lane];
    wave_sync();
    float part = 0.f;
    if (lane < NVS) { mxd = mass_times(m, L, W.tmp, lane); part = 0.5f * mxd * W.tmp[lane]; }
    if (has_con) {
      block_jx(x, jar);
#pragma unroll
      for (int j = 0; j < 6; j++) jar[j] -= C.aref[j];
      part += contact_cost(C, jar, force, Hc, want_h, &zone);
    }
    if (has_row) {
      rjar = rreg.sign * x[rreg.dof] - rreg.aref;
      part += row_cost(rreg, rjar, &rforce, &rh);
    }
    float tot = wave_sum_f(part);
    wave_sync();
    return tot;
  };
  // warm start: the better of the previous qacc and the unconstrained acceleration
  float cw = eval_cost(W.xw, false), cs = eval_cost(W.xs, false);
  if (lane < NVS) W.x[lane] = (cw < cs) ? W.xw[lane] : W.xs[lane];
  wave_sync();
  float scale = 1.f / (m->meaninertia * (float)NV);
  float cost = eval_cost(W.x, true), dec_prev = 0.f;
  // The factor of the scaled Hessian (lane a: row a of L) and the scaling survive from one iteration to the next: while
  // no block changes its zone and no contact sits in the middle zone of its cone (the only zone whose Hessian depends on
  // x), H is the same matrix and is neither assembled nor factorised again - resting props take their two or three
  // iterations on one factorisation.
  constexpr int HW = SINGLE ? 6 : NVS;              // entries of row `lane` kept in registers
  const int grp = lane / 6, sub = lane % 6;         // SINGLE: block and row inside the block
  // value of v at lane 6 * (own block) + b
  auto gget = [&](float v, int b) -> float {
    float a0 = wave_get_f(v, b), a1 = wave_get_f(v, 6 + b), a2 = wave_get_f(v, 12 + b);
    return grp == 0 ? a0 : (grp == 1 ? a1 : a2);
  };
  float h[HW], mxs = 1.f;
#pragma unroll
  for (int b = 0; b < HW; b++) h[b] = 0.f;
  int zone_prev = -1;
  bool rquad_prev = false;
  int it = 0;
  NPROF(0)
  for (; it < max_iter; it++) {
    // ---- gradient g = M (x - x_s) - J' f, lane d keeps g_d
    float jl[SINGLE ? 6 : 12];
#pragma unroll
    for (int c = 0; c < (SINGLE ? 6 : 12); c++) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 6; j++) s += C.J[c][j] * force[j];
      jl[c] = s;
    }
    float grad = mxd;
#pragma unroll
    for (int d = 0; d < NVS; d++) {
      const int G = d / 6, q = d % 6;
      if (anyG[G] || G == 0) {
        float v;
        if constexpr (SINGLE) v = (C.g0 == G) ? jl[q] : 0.f; else v = (C.g0 == G) ? jl[q] : ((C.g1 == G) ? jl[6 + q] : 0.f);
        if (G == 0) v += (has_row && rreg.dof == q) ? rreg.sign * rforce : 0.f;
        float tot = csum(v);
        if (lane == d) grad -= tot;
      }
    }
    NPROF(1)
    // ---- Hessian H = M + sum_blocks J' Hc J: lane a < NVS keeps row a in registers
    bool rquad = rh != 0.f;
    bool same = wave_ballot((has_con && (zone != zone_prev || zone == 2)) || (has_row && rquad != rquad_prev)) == 0ull;
    zone_prev = zone; rquad_prev = rquad;
    if constexpr (SINGLE) {
    if (!(it > 0 && same)) {
#pragma unroll
      for (int b = 0; b < 6; b++) h[b] = lane < NVS ? mass_entry(m, L, lane, 6 * grp + b) : 0.f;
#pragma unroll
      for (int q = 0; q < NARM; q++) {                       // scalar rows: J = +-e_dof, Hc = D when quadratic
        float tot = csum((has_row && rreg.dof == q) ? rh : 0.f);
        if (lane == q) h[q] += tot;
      }
      bool actG[3];
      {
        bool on = has_con && zone != 0;
#pragma unroll
        for (int G = 0; G < 3; G++) actG[G] = wave_ballot(on && C.g0 == G) != 0ull;
      }
      if (actG[0] || actG[1] || actG[2]) {
#pragma unroll
        for (int b = 0; b < 6; b++) {
          float W0[6];
#pragma unroll
          for (int i = 0; i < 6; i++) {
            float s0 = 0.f;
#pragma unroll
            for (int j = 0; j < 6; j++) s0 += Hc[i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i] * C.J[b][j];
            W0[i] = s0;
          }
#pragma unroll
          for (int a = 0; a <= b; a++) {
            float s0 = 0.f;
#pragma unroll
            for (int j = 0; j < 6; j++) s0 += C.J[a][j] * W0[j];
#pragma unroll
            for (int G = 0; G < 3; G++) {
              if (actG[G]) {
                float tot = csum((C.g0 == G) ? s0 : 0.f);
                if (lane == 6 * G + a) h[b] += tot;
                if (a != b && lane == 6 * G + b) h[a] += tot;
              }
            }
          }
        }
      }
      // diagonal scaling, then the three 6x6 Cholesky factorisations side by side (lane 6 G + i: row i of block G)
      float dg = 1.f;
#pragma unroll
      for (int b = 0; b < 6; b++) dg = (lane < NVS && sub == b) ? h[b] : dg;
      mxs = 1.f / sqrtf(fmaxf(dg, 1e-30f));
#pragma unroll
      for (int b = 0; b < 6; b++) h[b] *= mxs * gget(mxs, b);
#pragma unroll
      for (int j = 0; j < 6; j++) {
        float d = sqrtf(fmaxf(gget(h[j], j), 1e-7f));
        float l = (sub == j) ? d : h[j] / d;
        h[j] = l;
#pragma unroll
        for (int k = j + 1; k < 6; k++) h[k] -= l * gget(l, k);
      }
      if (lane < NVS) {
#pragma unroll
        for (int b = 0; b < 6; b++) W.H[lane][b] = h[b];
      }
      wave_sync();
      NPROF(3)
    }   // (assembly + factorisation)
    }
    float y = 0.f;
    if constexpr (SINGLE) {
      y = lane < NVS ? -grad * mxs : 0.f;
#pragma unroll
      for (int i = 0; i < 6; i++) {            // forward substitution, the three blocks in lock step
        float yi = gget(y, i) / gget(h[i], i);
        if (sub == i) y = yi;
        else if (sub > i) y -= h[i] * yi;
      }
      float t[6];
#pragma unroll
      for (int i = 0; i < 6; i++) t[i] = lane < NVS ? W.H[6 * grp + i][sub] : 0.f;
#pragma unroll
      for (int i = 5; i >= 0; i--) {           // L' x = y
        float xi = gget(y, i) / gget(h[i], i);
        if (sub == i) y = xi;
        else if (sub < i) y -= t[i] * xi;
      }
    }
    if constexpr (!SINGLE) {
    if (!(it > 0 && same)) {
#pragma unroll
    for (int b = 0; b < NVS; b++) h[b] = lane < NVS ? mass_entry(m, L, lane, b) : 0.f;
#pragma unroll
    for (int q = 0; q < NARM; q++) {                       // scalar rows: J = +-e_dof, Hc = D when quadratic
      float tot = csum((has_row && rreg.dof == q) ? rh : 0.f);
      if (lane == q) h[q] += tot;
    }
    bool actG[3], actX[3];
    {
      bool on = has_con && zone != 0;
#pragma unroll
      for (int G = 0; G < 3; G++) actG[G] = wave_ballot(on && (C.g0 == G || C.g1 == G)) != 0ull;
      actX[0] = wave_ballot(on && C.g0 == 0 && C.g1 == 1) != 0ull;
      actX[1] = wave_ballot(on && C.g0 == 0 && C.g1 == 2) != 0ull;
      actX[2] = wave_ballot(on && C.g0 == 1 && C.g1 == 2) != 0ull;
    }
#ifndef SO101_DPP_HESSIAN
    // Contact part on the matrix cores:  sum_k J_k' Hc_k J_k = Jt W  with the constraint rows as the K dimension, W = Hc J.
    // Six contacts at a time stage their 36 rows of J and W (18 coordinates each) in LDS - in the storage of the geom
    // boxes / arm-contact pool, which nobody reads during the solve -, 18 v_mfma_f32_32x32x2_f32 accumulate them, and the
    // accumulator's column a (H is symmetric) goes to lane a.  With every group pair coupled the 171 cross-lane sums below cost
    // 17.8 us per assembly and wavefront on a full machine, this path 3.1-5.4 us for 4-16 contacts
    // (scripts/microbench/mfma_hessian.hip, profiles/README.md) - and coupled envs are the ones whose solves last longest.
    if (actG[0] || actG[1] || actG[2]) {
      constexpr int HB = 6;                                  // contacts per batch
      float* Jl = &L.aabb[0][0];
      float* Wl = Jl + 6 * HB * NVS;
      static_assert(sizeof(float) * 2 * 6 * HB * NVS <= sizeof(ArmCon) * MAXARMCON, "Hessian staging must fit the collision scratch (union in EnvLDS)");
      mfma_acc16 acc;
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] = 0.f;
      const int col = lane & 31, half = lane >> 5;
      const bool live = col < NVS;
      const int off = half * NVS + (live ? col : 0);
      for (int k0 = 0; k0 < ncon; k0 += HB) {
        int k1 = k0 + HB < ncon ? k0 + HB : ncon;
        if (lane >= k0 && lane < k1) {
          // rows 6 (lane - k0) ..: the group this contact does not touch is zero, the two slots go to their groups' columns
          // (a contact with one group: slot 1 is all zeros and lands on a group of its own)
          int ga = C.g0 < 0 ? 0 : C.g0, gb = C.g1 < 0 ? (ga == 0 ? 1 : 0) : C.g1, gz = 3 - ga - gb;
          float* jd = Jl + 6 * (lane - k0) * NVS; float* wd = Wl + 6 * (lane - k0) * NVS;
#pragma unroll
          for (int q = 0; q < 6; q++) {
            // column q of both slots: W0 = Hc J[:, q of slot 0], W1 = Hc J[:, q of slot 1]  (the products of the sums' path)
            float W0[6], W1[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
              float s0 = 0.f, s1 = 0.f;
#pragma unroll
              for (int j = 0; j < 6; j++) { float hc = Hc[i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i]; s0 += hc * C.J[q][j]; s1 += hc * C.J[6 + q][j]; }
              W0[i] = s0; W1[i] = s1;
            }
#pragma unroll
            for (int j = 0; j < 6; j++) {
              jd[j * NVS + 6 * ga + q] = C.J[q][j]; jd[j * NVS + 6 * gb + q] = C.J[6 + q][j]; jd[j * NVS + 6 * gz + q] = 0.f;
              wd[j * NVS + 6 * ga + q] = W0[j]; wd[j * NVS + 6 * gb + q] = W1[j]; wd[j * NVS + 6 * gz + q] = 0.f;
            }
            SCHED_FENCE();
          }
        }
        wave_sync();
        int steps = 3 * (k1 - k0);
        for (int t = 0; t < steps; t++) {
          float av = Jl[off + 2 * NVS * t], bv = Wl[off + 2 * NVS * t];
          acc = mfma_32x32x2(live ? av : 0.f, live ? bv : 0.f, acc);
        }
        wave_sync();
      }
      // lane l: column l % 32, rows 8 blk + 4 (l / 32) + r % 4 in element 4 blk + r % 4 -> lane a gets column a = row a
      float o[8];
#pragma unroll
      for (int r = 0; r < 8; r++) o[r] = wave_xor32_f(acc[r]);
      if (lane < NVS) {
#pragma unroll
        for (int r = 0; r < 4; r++) { h[r] += acc[r]; h[4 + r] += o[r]; h[8 + r] += acc[4 + r]; h[12 + r] += o[4 + r]; }
        h[16] += acc[8]; h[17] += acc[9];
      }
    }
#else
    if (actG[0] || actG[1] || actG[2]) {
#pragma unroll
      for (int b = 0; b < 6; b++) {
        // W0 = Hc J[:, b of slot 0], W1 = Hc J[:, b of slot 1]
        float W0[6], W1[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int j = 0; j < 6; j++) { float hc = Hc[i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i]; s0 += hc * C.J[b][j]; s1 += hc * C.J[6 + b][j]; }
          W0[i] = s0; W1[i] = s1;
        }
        // diagonal blocks: entries (a, b), a <= b, of slot 0 and slot 1
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
        // cross blocks: entry (a of slot 0, b of slot 1)
        if (actX[0] || actX[1] || actX[2]) {
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
    }
#endif
    NPROF(2)
    // ---- symmetric diagonal scaling  H~ = S H S, S = diag(H)^-1/2 : translational (mass ~ 4e-2) and rotational
    // (inertia ~ 1e-5) coordinates differ by ~1e3 in scale, which puts cond(H) near 1/eps_fp32; after scaling the
    // fp32 Cholesky is safe.  Solve H~ y = -S g, search = S y.
    float dg = 1.f;
#pragma unroll
    for (int b = 0; b < NVS; b++) dg = (lane == b) ? h[b] : dg;
    mxs = 1.f / sqrtf(fmaxf(dg, 1e-30f));
#pragma unroll
    for (int b = 0; b < NVS; b++) h[b] *= mxs * wave_get_f(mxs, b);
    // ---- Cholesky H~ = L L' and the two triangular solves, register resident: lane i < NVS owns row i of H~ (and
    // of L), the pivot row is broadcast with v_readlane
#pragma unroll
    for (int j = 0; j < NVS; j++) {
      float d = sqrtf(fmaxf(wave_get_f(h[j], j), 1e-7f));       // pivot floor: H~ has unit diagonal
      float l = (lane == j) ? d : h[j] / d;
      h[j] = l;
#pragma unroll
      for (int k = j + 1; k < NVS; k++) h[k] -= l * wave_get_f(l, k);     // rows i < k hold unused upper entries
    }
    // backward substitution needs column `lane` of L: one transposed round trip through LDS
    if (lane < NVS) {
#pragma unroll
      for (int b = 0; b < NVS; b++) W.H[lane][b] = h[b];
    }
    wave_sync();
    NPROF(3)
    }   // (assembly + factorisation)
    y = lane < NVS ? -grad * mxs : 0.f;
#pragma unroll
    for (int i = 0; i < NVS; i++) {          // forward substitution L y = b
      float yi = wave_get_f(y, i) / wave_get_f(h[i], i);
      if (lane == i) y = yi;
      else if (lane > i) y -= h[i] * yi;
    }
    float t[NVS];
#pragma unroll
    for (int i = 0; i < NVS; i++) t[i] = lane < NVS ? W.H[i][lane] : 0.f;
#pragma unroll
    for (int i = NVS - 1; i >= 0; i--) {     // L' x = y
      float xi = wave_get_f(y, i) / wave_get_f(h[i], i);
      if (lane == i) y = xi;
      else if (lane < i) y -= t[i] * xi;
    }
    }
    float sv = y * mxs;
    if (lane < NVS) W.search[lane] = sv;
    wave_sync();
    // ---- termination test on the Newton decrement  dec = -g' search  (twice the cost the quadratic model still
    // expects to gain; in exact arithmetic MuJoCo's test "cost improvement < tolerance" is 0.5 dec < tolerance).
    // The cost itself cannot be used in fp32: with stiff contacts (finger pads squeezing a prop: cost ~1e8) a wrong
    // angular acceleration of a 1e-5 kg m^2 body changes it by less than one ulp, while the decrement, built from the
    // per-coordinate gradient, still resolves it.  Once 0.5 dec is below the cost's fp32 resolution, this step is the
    // last one if the decrement has stalled (rounding floor) or has just collapsed by more than 100x (quadratic
    // convergence: the step about to be taken leaves an error far below the tolerance); the iteration only goes on while
    // the decrement shrinks slowly, i.e. while cone zones are still switching.
    NPROF(4)
    float dec = wave_sum_f(lane < NVS ? -grad * sv : 0.f);
    bool done = scale * 0.5f * dec < tolerance ||
                (it > 0 && 0.5f * dec < 4e-7f * fabsf(cost) && (dec > 0.5f * dec_prev || dec < 1e-2f * dec_prev));
    dec_prev = dec;
    // ---- exact line search: phi'(alpha) = 0 by safeguarded Newton
    float jv[6] = {0, 0, 0, 0, 0, 0}, rjv = 0.f;
    if (has_con) block_jx(W.search, jv);
    if (has_row) rjv = rreg.sign * W.search[rreg.dof];
    float q1p = 0.f, q2p = 0.f;
    if (lane < NVS) { float ms = mass_times(m, L, W.search, lane); q1p = sv * mxd; q2p = sv * ms; }
    float q1 = wave_sum_f(q1p), q2 = wave_sum_f(q2p);
    float alpha = 0.f, lo = 0.f, hi = -1.f, d10 = 0.f;
    for (int ls = 0; ls < 12; ls++) {
      float d1p = 0.f, d2p = 0.f;
      if (has_con) {
        float r6[6];
#pragma unroll
        for (int j = 0; j < 6; j++) r6[j] = jar[j] + alpha * jv[j];
        contact_line(C, r6, jv, &d1p, &d2p);
      }
      if (has_row) {
        float f1, h1;
        row_cost(rreg, rjar + alpha * rjv, &f1, &h1);
        d1p -= f1 * rjv; d2p += h1 * rjv * rjv;
      }
      float d1 = q1 + q2 * alpha + csum(d1p), d2 = q2 + csum(d2p);
      if (ls == 0) { d10 = fabsf(d1); if (!(d1 < 0.f)) break; }
      else {
        if (fabsf(d1) <= 1e-4f * d10) break;
        if (d1 < 0.f) lo = alpha; else hi = alpha;
      }
      float cand = alpha - d1 / fmaxf(d2, 1e-30f);
      if (hi > 0.f && (cand <= lo || cand >= hi)) cand = 0.5f * (lo + hi);
      alpha = cand;
    }
    if (lane < NVS) W.x[lane] += alpha * sv;
    wave_sync();
    NPROF(5)
#ifdef SO101_EMU_TRACE
    { float oldc = cost; float nc = eval_cost(W.x, true); if (lane == 0) fprintf(stderr, "  newton it %d cost %.9g -> %.9g dec %.4g alpha %.4g done %d\n", it, oldc, nc, dec, alpha, (int)done); }
#endif
    cost = eval_cost(W.x, !done);          // (the block Hessians are only needed if another iteration follows)
    float gnorm = scale * sqrtf(wave_sum_f(lane < NVS ? grad * grad : 0.f));   // gradient of the previous point (cheap proxy)
    NPROF(6)
    if (done || gnorm < tolerance) { it++; break; }
  }
  // constrained accelerations back to the shared island state; forces for diagnostics
  if (lane < NARM) L.qacc_arm[lane] = W.x[lane];
  if (lane >= NARM && lane < NVS) L.facc[(lane - NARM) / 6][(lane - NARM) % 6] = W.x[lane];
  if (has_con) {
#pragma unroll
    for (int j = 0; j < 6; j++) L.con[lane].f[j] = force[j];
  }
  if (has_row) L.row[lane].f = rforce;
  if (lane == 0) L.iters = it;
  wave_sync();
}

// Instances: with at most 16 contacts (and 16 scalar rows) every sum over contact lanes is a single-row sum (wave.hpp
// wave_sum_rows_f) - same bits, ~100-200 fewer v_readlane + adds per iteration; a run-time flag inside one instance was
// slower than no shortcut at all (650 k against 680 k env-steps/s: the scalar branch around each sum stops the scheduler
// from overlapping neighbouring sums).  Among those, envs whose contacts never couple two coordinate groups (props
// resting on the table, the arm in free space or on the table - most envs of the hand-over workload) take the
// block-diagonal instance.
DEV void solve_newton(const DevModel* m, EnvLDS& L, int max_iter, float tolerance) {
  int lane = wave_lane();
  bool coupled = false;
  if (lane < L.ncon) { int g0, g1; contact_groups(L.con[lane], &g0, &g1); coupled = g1 >= 0; }
  bool cross = wave_ballot(coupled) != 0ull;
#if defined(SO101_PROBE_INSTANCE) && SO101_PROBE_INSTANCE == 1
  (void)cross; solve_newton_impl<true, true>(m, L, max_iter, tolerance);
#elif defined(SO101_PROBE_INSTANCE) && SO101_PROBE_INSTANCE == 2
  (void)cross; solve_newton_impl<true, false>(m, L, max_iter, tolerance);
#elif defined(SO101_PROBE_INSTANCE) && SO101_PROBE_INSTANCE == 3
  (void)cross; solve_newton_impl<false, false>(m, L, max_iter, tolerance);
#else
  if (wave_uniform_i((int)(L.ncon <= 16 && L.nrow <= 16))) {
    if (!cross) solve_newton_impl<true, true>(m, L, max_iter, tolerance);
    else solve_newton_impl<true, false>(m, L, max_iter, tolerance);
  } else solve_newton_impl<false, false>(m, L, max_iter, tolerance);
#endif
}
