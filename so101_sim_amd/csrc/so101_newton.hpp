// Newton constraint solver (mujoco's default, mj_solNewton) for one env per wavefront.
//
//   min_x  0.5 (x - x_s)' M (x - x_s) + sum_blocks s_b(J_b x - aref_b)
//
// in the solver coordinates x = [arm qacc (6) | object twist (6) | container twist (6)], where M is block diagonal:
// the 6x6 arm matrix and, per free body, (m I3, I_world).  Unlike PGS (a long sequential chain of tiny block
// updates) every stage here is dense lane-parallel work:
//   lanes = constraint blocks : residual r = J x - aref, zone / force / block Hessian, line-search derivatives
//   lanes = matrix entries    : H = M + sum J_b' Hc_b J_b (12x12 local blocks staged through LDS), Cholesky, solves
// The minimiser is unique, so results agree with the fp64 oracle at solution level (not iterate level).
#pragma once

#define NBLK (MAXCON + MAXROW1)     // lane k < MAXCON: contact k ; lane 32 + r: scalar row r

// block cost s(r), force = -ds/dr and (optionally) the symmetric block Hessian d2s/dr2 (packed lower triangle)
DEV float contact_cost(const Contact& c, const float* r, float* force, float* Hc, bool want_h, int* zone) {
  const float fr[5] = {c.fric[0], c.fric[0], c.fric[1], c.fric[2], c.fric[2]};
  const float Dj[6] = {1.f / c.R[0], 1.f / c.R[1], 1.f / c.R[1], 1.f / c.R[2], 1.f / c.R[3], 1.f / c.R[3]};
  int dim = c.dim;
  if (want_h) {
#pragma unroll
    for (int k = 0; k < 21; k++) Hc[k] = 0.f;
  }
  float mu = c.mu, U[6], T = 0.f;
  U[0] = r[0] * mu;
#pragma unroll
  for (int j = 1; j < 6; j++) { U[j] = (j < dim) ? r[j] * fr[j - 1] : 0.f; T += U[j] * U[j]; }
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
      float D = (j < dim) ? Dj[j] : 0.f;
      force[j] = -D * r[j]; cost += 0.5f * D * r[j] * r[j];
      if (want_h) Hc[j * (j + 1) / 2 + j] = D;
    }
    *zone = 1;
    return cost;
  }
  float Dm = Dj[0] / fmaxf(mu * mu * (1.f + mu * mu), MINVAL_F), sN = N - mu * T, iT = 1.f / T;   // middle zone
  force[0] = -Dm * sN * mu;
#pragma unroll
  for (int j = 1; j < 6; j++) force[j] = (j < dim) ? -force[0] * iT * U[j] * fr[j - 1] : 0.f;
  if (want_h) {
    float a = Dm * mu * mu, b = Dm * sN * mu;
    Hc[0] = a;
#pragma unroll
    for (int k = 1; k < 6; k++) {
      float wk = (k < dim) ? U[k] * fr[k - 1] * iT : 0.f;          // mu_k u_k / T
      Hc[k * (k + 1) / 2] = -a * wk;
#pragma unroll
      for (int l = 1; l <= k; l++) {
        float wl = (l < dim) ? U[l] * fr[l - 1] * iT : 0.f;
        float diag = (k == l && k < dim) ? fr[k - 1] * fr[k - 1] * iT : 0.f;
        Hc[k * (k + 1) / 2 + l] = a * wk * wl - b * (diag - wk * wl * iT);
      }
    }
  }
  *zone = 2;
  return 0.5f * Dm * sN * sN;
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

// solver-coordinate groups a contact touches: slot 0 / slot 1 (-1 = none); 0 arm, 1 object, 2 container
DEV void contact_groups(const Contact& c, int* g0, int* g1) {
  int a = c.armslot >= 0 ? 0 : -1;
  int f1 = c.d1 >= NARM ? c.d1 - NARM + 1 : -1, f2 = c.d2 >= NARM ? c.d2 - NARM + 1 : -1;
  int first = a >= 0 ? a : (f1 >= 0 ? f1 : f2);
  int second = a >= 0 ? (f1 >= 0 ? f1 : f2) : (f1 >= 0 ? f2 : -1);
  *g0 = first; *g1 = second;
}

// entry (row j, local column q of group g) of the contact Jacobian in solver coordinates
DEV float contact_jentry(const EnvLDS& L, const Contact& c, int g, int j, int q) {
  if (g == 0) return L.armcon[c.armslot].J[j][q];
  int d = NARM + g - 1;
  float sgn = (c.d2 == d) ? 1.f : -1.f;
  int jj = j % 3, qq = q % 3;
  float u0 = c.frame[3 * jj], u1 = c.frame[3 * jj + 1], u2 = c.frame[3 * jj + 2];
  float uq = qq == 0 ? u0 : (qq == 1 ? u1 : u2);           // selects, never a lane-indexed register array
  if (j < 3) {
    if (q < 3) return sgn * uq;
    float r0 = c.pos[0] - L.xipos[d][0], r1 = c.pos[1] - L.xipos[d][1], r2 = c.pos[2] - L.xipos[d][2];
    float t0 = r1 * u2 - r2 * u1, t1 = r2 * u0 - r0 * u2, t2 = r0 * u1 - r1 * u0;      // r x u
    return sgn * (qq == 0 ? t0 : (qq == 1 ? t1 : t2));
  }
  return q < 3 ? 0.f : sgn * uq;
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
  if (k < 3) return m->free_mass[f] * v[lane];
  const float* I = L.Iw[NARM + f];
  const float* w = &v[NARM + 6 * f + 3];
  float o[3]; symvec3(o, I, w);
  return k == 3 ? o[0] : (k == 4 ? o[1] : o[2]);
}

DEV void solve_newton(const DevModel* m, EnvLDS& L, int max_iter, float tolerance) {
  int lane = wave_lane();
  int nrow = L.nrow, ncon = L.ncon;
  NewtonScratch& W = L.nw;
  if (lane == 0) L.iters = 0;
  if (nrow + ncon == 0) { wave_sync(); return; }
  bool has_con = lane < ncon, has_row = lane >= 32 && lane - 32 < nrow;
  Contact creg; Row1 rreg;
  if (has_con) creg = L.con[lane];
  if (has_row) rreg = L.row[lane - 32];
  int g0 = -1, g1 = -1;
  if (has_con) contact_groups(creg, &g0, &g1);
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
  // residual of this lane's block at the point `x` (LDS vector of NVS), and J*v for a direction
  auto block_jx = [&](const float* x, float* out6) {
    if (has_con) {
      Acc a;
#pragma unroll
      for (int q = 0; q < NARM; q++) a.arm[q] = x[q];
#pragma unroll
      for (int f = 0; f < NFREE; f++)
#pragma unroll
        for (int i = 0; i < 6; i++) a.fr[f][i] = x[NARM + 6 * f + i];
      jacc_reg(L, creg, a, out6);
    } else if (has_row) out6[0] = rreg.sign * x[rreg.dof];
  };
  // total cost at x (also leaves this lane's residual in jar); gauss part via W.mxd = M (x - x_s)
  float jar[6] = {0, 0, 0, 0, 0, 0}, force[6] = {0, 0, 0, 0, 0, 0}, Hc[21];
  auto eval_cost = [&](const float* x, bool want_h) -> float {
    if (lane < NVS) W.tmp[lane] = x[lane] - W.xs[lane];
    wave_sync();
    float part = 0.f;
    if (lane < NVS) { float mv = mass_times(m, L, W.tmp, lane); W.mxd[lane] = mv; part = 0.5f * mv * W.tmp[lane]; }
    block_jx(x, jar);
    if (has_con) {
#pragma unroll
      for (int j = 0; j < 6; j++) jar[j] -= creg.aref[j];
      int zn;
      part += contact_cost(creg, jar, force, Hc, want_h, &zn);
      if (want_h) W.zone[lane] = zn;
    } else if (has_row) {
      jar[0] -= rreg.aref;
      float h;
      part += row_cost(rreg, jar[0], &force[0], &h);
      Hc[0] = h;
      if (want_h) { W.rowf[lane - 32] = force[0]; W.rowh[lane - 32] = h; }
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
  float cost = eval_cost(W.x, true);
  int it = 0;
  for (; it < max_iter; it++) {
    // ---- gradient g = M (x - x_s) - J' f : per-block contributions in local columns, summed in block order
    if (has_con) {
#pragma unroll
      for (int col = 0; col < 12; col++) {
        int g = col < 6 ? g0 : g1;
        float s = 0.f;
        if (g >= 0) {
#pragma unroll
          for (int j = 0; j < 6; j++) s += contact_jentry(L, creg, g, j, col % 6) * force[j];
        }
        W.jtf[lane][col] = s;
      }
    }
    wave_sync();
    if (lane < NVS) {
      float gsum = W.mxd[lane];
      int grp = lane / 6, q = lane % 6;
      for (int k = 0; k < ncon; k++) {
        int a0, a1; contact_groups(L.con[k], &a0, &a1);
        if (a0 == grp) gsum -= W.jtf[k][q];
        else if (a1 == grp) gsum -= W.jtf[k][6 + q];
      }
      if (lane < NARM) for (int k = 0; k < nrow; k++) if (L.row[k].dof == lane) gsum -= L.row[k].sign * W.rowf[k];
      W.grad[lane] = gsum;
    }
    // ---- Hessian H = M + sum_blocks J' Hc J   (lower triangle kept full for simplicity)
    for (int e = lane; e < NVS * NVS; e += WAVE) {
      int a = e / NVS, b = e % NVS;
      float v = 0.f;
      if (a < NARM && b < NARM) v = L.Marm[a][b];
      else if (a >= NARM && b >= NARM && (a - NARM) / 6 == (b - NARM) / 6) {
        int f = (a - NARM) / 6, i = (a - NARM) % 6, j = (b - NARM) % 6;
        if (i < 3 && j < 3) v = (i == j) ? m->free_mass[f] : 0.f;
        else if (i >= 3 && j >= 3) {
          const float* I = L.Iw[NARM + f];
          int p = i - 3, q = j - 3;
          v = I[p == q ? p : p + q + 2];        // packed xx yy zz xy xz yz: (0,1)->3, (0,2)->4, (1,2)->5
        }
      }
      W.H[a][b] = v;
    }
    wave_sync();
    if (lane < NARM) {                       // scalar rows: J = +-e_dof, Hc = D when quadratic
      float add = 0.f;
      for (int k = 0; k < nrow; k++) if (L.row[k].dof == lane) add += W.rowh[k];
      W.H[lane][lane] += add;
    }
    for (int k = 0; k < ncon; k++) {         // contacts with a non-zero block Hessian
      if (W.zone[k] == 0) continue;
      const Contact& c = L.con[k];
      int a0, a1; contact_groups(c, &a0, &a1);
      if (lane == k) {
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
          for (int j = 0; j <= i; j++) { W.Hst[i][j] = Hc[i * (i + 1) / 2 + j]; W.Hst[j][i] = Hc[i * (i + 1) / 2 + j]; }
      }
      for (int t = lane; t < 72; t += WAVE) {
        int j = t / 12, col = t % 12, g = col < 6 ? a0 : a1;
        W.Jst[j][col] = g >= 0 ? contact_jentry(L, c, g, j, col % 6) : 0.f;
      }
      wave_sync();
      for (int t = lane; t < 72; t += WAVE) {
        int kk = t / 12, col = t % 12;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 6; j++) s += W.Hst[kk][j] * W.Jst[j][col];
        W.Wst[kk][col] = s;
      }
      wave_sync();
      int ncol = a1 >= 0 ? 12 : 6;
      for (int t = lane; t < ncol * ncol; t += WAVE) {
        int la = t / ncol, lb = t % ncol;
        float s = 0.f;
#pragma unroll
        for (int kk = 0; kk < 6; kk++) s += W.Jst[kk][la] * W.Wst[kk][lb];
        int ga = (la < 6 ? a0 : a1) * 6 + la % 6, gb = (lb < 6 ? a0 : a1) * 6 + lb % 6;
        W.H[ga][gb] += s;
      }
      wave_sync();
    }
    // ---- symmetric diagonal scaling  H~ = S H S, S = diag(H)^-1/2 : translational (mass ~ 4e-2) and rotational
    // (inertia ~ 1e-5) coordinates differ by ~1e3 in scale, which puts cond(H) near 1/eps_fp32; after scaling the
    // fp32 Cholesky is safe.  Solve H~ y = -S g, search = S y.
    if (lane < NVS) W.mxs[lane] = 1.f / sqrtf(fmaxf(W.H[lane][lane], 1e-30f));
    wave_sync();
    for (int e = lane; e < NVS * NVS; e += WAVE) { int a = e / NVS, b = e % NVS; W.H[a][b] *= W.mxs[a] * W.mxs[b]; }
    wave_sync();
    // ---- Cholesky H~ = L L' and the two triangular solves, register resident: lane i < NVS owns row i of H~ (and
    // of L), the pivot row is broadcast with v_readlane.  No LDS traffic and no barriers inside the factorisation
    // (the LDS version spent ~130 barrier rounds per Newton iteration here).
    float h[NVS];
#pragma unroll
    for (int b = 0; b < NVS; b++) h[b] = lane < NVS ? W.H[lane][b] : 0.f;
#pragma unroll
    for (int j = 0; j < NVS; j++) {
      float d = sqrtf(fmaxf(wave_get_f(h[j], j), 1e-7f));       // pivot floor: H~ has unit diagonal
      float l = (lane == j) ? d : h[j] / d;
      h[j] = l;
#pragma unroll
      for (int k = j + 1; k < NVS; k++) h[k] -= l * wave_get_f(l, k);     // rows i < k hold unused upper entries
    }
    float y = lane < NVS ? -W.grad[lane] * W.mxs[lane] : 0.f;
#pragma unroll
    for (int i = 0; i < NVS; i++) {          // forward substitution L y = b
      float yi = wave_get_f(y, i) / wave_get_f(h[i], i);
      if (lane == i) y = yi;
      else if (lane > i) y -= h[i] * yi;
    }
    // backward substitution needs column `lane` of L: one transposed round trip through LDS
    wave_sync();
    if (lane < NVS) {
#pragma unroll
      for (int b = 0; b < NVS; b++) W.H[lane][b] = h[b];
    }
    wave_sync();
    float t[NVS];
#pragma unroll
    for (int i = 0; i < NVS; i++) t[i] = lane < NVS ? W.H[i][lane] : 0.f;
#pragma unroll
    for (int i = NVS - 1; i >= 0; i--) {     // L' x = y
      float xi = wave_get_f(y, i) / wave_get_f(h[i], i);
      if (lane == i) y = xi;
      else if (lane < i) y -= t[i] * xi;
    }
    if (lane < NVS) { float sv = y * W.mxs[lane]; W.tmp[lane] = sv; W.search[lane] = sv; }
    wave_sync();
    // ---- exact line search: phi'(alpha) = 0 by safeguarded Newton
    float jv[6] = {0, 0, 0, 0, 0, 0}, jar0[6];
    block_jx(W.search, jv);
#pragma unroll
    for (int j = 0; j < 6; j++) jar0[j] = jar[j];
    float q1p = 0.f, q2p = 0.f;
    if (lane < NVS) { float ms = mass_times(m, L, W.search, lane); q1p = W.search[lane] * W.mxd[lane]; q2p = W.search[lane] * ms; }
    float q1 = wave_sum_f(q1p), q2 = wave_sum_f(q2p);
    float alpha = 0.f, lo = 0.f, hi = -1.f, d10 = 0.f;
    for (int ls = 0; ls < 12; ls++) {
      float d1p = 0.f, d2p = 0.f;
      if (has_con) {
        float r6[6], f6[6], h21[21];
#pragma unroll
        for (int j = 0; j < 6; j++) r6[j] = jar0[j] + alpha * jv[j];
        int zn; contact_cost(creg, r6, f6, h21, true, &zn);
#pragma unroll
        for (int j = 0; j < 6; j++) {
          d1p -= f6[j] * jv[j];
#pragma unroll
          for (int k = 0; k < 6; k++) d2p += jv[j] * h21[tri(j, k)] * jv[k];
        }
      } else if (has_row) {
        float f1, h1;
        row_cost(rreg, jar0[0] + alpha * jv[0], &f1, &h1);
        d1p = -f1 * jv[0]; d2p = h1 * jv[0] * jv[0];
      }
      float d1 = q1 + q2 * alpha + wave_sum_f(d1p), d2 = q2 + wave_sum_f(d2p);
      if (ls == 0) { d10 = fabsf(d1); if (!(d1 < 0.f)) break; }
      else {
        if (fabsf(d1) <= 1e-4f * d10) break;
        if (d1 < 0.f) lo = alpha; else hi = alpha;
      }
      float cand = alpha - d1 / fmaxf(d2, 1e-30f);
      if (hi > 0.f && (cand <= lo || cand >= hi)) cand = 0.5f * (lo + hi);
      alpha = cand;
    }
    if (lane < NVS) W.x[lane] += alpha * W.search[lane];
    wave_sync();
    float newcost = eval_cost(W.x, true);

    float gp = lane < NVS ? W.grad[lane] * W.grad[lane] : 0.f;       // gradient of the previous point (cheap proxy)
    float gnorm = scale * sqrtf(wave_sum_f(gp));
    float improvement = scale * (cost - newcost);
    float floor32 = 4e-7f * scale * fabsf(cost);                      // cost differences below fp32 resolution
    cost = newcost;
    if (improvement < fmaxf(tolerance, floor32) || gnorm < tolerance) { it++; break; }
  }
  // constrained accelerations back to the shared island state; forces for diagnostics
  if (lane < NARM) L.qacc_arm[lane] = W.x[lane];
  if (lane >= NARM && lane < NVS) L.facc[(lane - NARM) / 6][(lane - NARM) % 6] = W.x[lane];
  if (has_con) {
#pragma unroll
    for (int j = 0; j < 6; j++) L.con[lane].f[j] = force[j];
  }
  if (has_row) L.row[lane - 32].f = force[0];
  if (lane == 0) L.iters = it;
  wave_sync();
}
