// PGS constraint solve, island-parallel.
//
// The rows of one env split into at most three independent "islands" (arm, object, container; a contact
// between two dynamic bodies merges their islands).  Gauss-Seidel over a block-diagonal system
// decouples exactly, so each island is swept by ONE LANE in MuJoCo's row order, with the island's
// accelerations held in that lane's registers: the iteration loop has no barrier and no shared writes.
// Lanes 0..2 own the islands rooted at (arm, object, container); the wave reconverges once per iteration
// for the termination test (sum of cost improvements, MuJoCo's rule).
//
// Velocity-space form: the dual residual  A f + b  of a block is evaluated as  J·acc − aref + R f  with
// acc = qacc_smooth + M⁻¹Jᵀf carried along; the iterates equal those of MuJoCo's explicit-A PGS.
#pragma once

struct Acc { float arm[NARM]; float fr[NFREE][6]; };

DEV void acc_load(const EnvLDS& L, Acc& a) {
#pragma unroll
  for (int q = 0; q < NARM; q++) a.arm[q] = L.qacc_arm[q];
#pragma unroll
  for (int f = 0; f < NFREE; f++)
#pragma unroll
    for (int i = 0; i < 6; i++) a.fr[f][i] = L.facc[f][i];
}

// island group of a dynamic-body index: arm links -> 0, free body f -> 1+f, static -> -1
DEV int body_group(int d) { return d < 0 ? -1 : (d < NARM ? 0 : d - NARM + 1); }

// J·acc for the 6 rows of contact c (sign of each side folded in)
DEV void jacc_reg(const EnvLDS& L, const Contact& c, const Acc& a, float* jv) {
#pragma unroll
  for (int j = 0; j < 6; j++) jv[j] = 0.f;
#pragma unroll
  for (int side = 0; side < 2; side++) {
    int d = side == 0 ? c.d1 : c.d2;
    float sgn = side == 0 ? -1.f : 1.f;
    if (d >= NARM) {
      bool f1 = d > NARM;
      float al[3], aa[3];
#pragma unroll
      for (int i = 0; i < 3; i++) { al[i] = f1 ? a.fr[1][i] : a.fr[0][i]; aa[i] = f1 ? a.fr[1][3 + i] : a.fr[0][3 + i]; }
      float r[3] = {c.pos[0] - L.xipos[d][0], c.pos[1] - L.xipos[d][1], c.pos[2] - L.xipos[d][2]};
      float t[3]; cross3(t, aa, r);
      float pl[3] = {al[0] + t[0], al[1] + t[1], al[2] + t[2]};
#pragma unroll
      for (int j = 0; j < 3; j++) { jv[j] += sgn * dot3(&c.frame[3 * j], pl); jv[3 + j] += sgn * dot3(&c.frame[3 * j], aa); }
    }
  }
  if (c.armslot >= 0) {            // arm part once: the stored rows already hold J(link2) - J(link1)
    const ArmCon& ac = L.armcon[c.armslot];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < NARM; q++) v += ac.Jt[q][j] * a.arm[q];
      jv[j] += v;
    }
  }
}

// acc += M⁻¹ Jᵀ df for contact c
DEV void apply_reg(const EnvLDS& L, const Contact& c, const float* df, Acc& a) {
  float F[3], T0[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    F[i] = c.frame[i] * df[0] + c.frame[3 + i] * df[1] + c.frame[6 + i] * df[2];
    T0[i] = c.frame[i] * df[3] + c.frame[3 + i] * df[4] + c.frame[6 + i] * df[5];
  }
#pragma unroll
  for (int side = 0; side < 2; side++) {
    int d = side == 0 ? c.d1 : c.d2;
    float sgn = side == 0 ? -1.f : 1.f;
    if (d >= NARM) {
      int f = d - NARM;
      float r[3] = {c.pos[0] - L.xipos[d][0], c.pos[1] - L.xipos[d][1], c.pos[2] - L.xipos[d][2]};
      float T[3]; cross3(T, r, F);
      T[0] += T0[0]; T[1] += T0[1]; T[2] += T0[2];
      float da[3]; symvec3(da, L.fIinv[f], T);
      float mi = sgn * L.fminv[f];
      float s0 = f == 0 ? 1.f : 0.f, s1 = 1.f - s0;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        a.fr[0][i] += s0 * mi * F[i]; a.fr[0][3 + i] += s0 * sgn * da[i];
        a.fr[1][i] += s1 * mi * F[i]; a.fr[1][3 + i] += s1 * sgn * da[i];
      }
    }
  }
  if (c.armslot >= 0) {            // arm: qacc += Minv (J^T df)
    const ArmCon& ac = L.armcon[c.armslot];
    float g[NARM];
#pragma unroll
    for (int q = 0; q < NARM; q++) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 6; j++) v += ac.Jt[q][j] * df[j];
      g[q] = v;
    }
#pragma unroll
    for (int q = 0; q < NARM; q++) {
      float v = 0.f;
#pragma unroll
      for (int s = 0; s < NARM; s++) v += L.Minv[q][s] * g[s];
      a.arm[q] += v;
    }
  }
}

DEV float arm_get(const Acc& a, int d) {
  float v = a.arm[0];
#pragma unroll
  for (int q = 1; q < NARM; q++) v = (d == q) ? a.arm[q] : v;
  return v;
}

// one PGS update of scalar row r (dof frictionloss / joint limit); returns the cost decrease
DEV float row_update(const EnvLDS& L, Row1& r, Acc& a) {
  int d = r.dof; float sg = r.sign, fold = r.f;
  float res = sg * arm_get(a, d) - r.aref + r.R * fold;
  float fnew = fold - res * r.Ainv;
  if (r.floss > 0.f) fnew = fminf(fmaxf(fnew, -r.floss), r.floss);
  else if (fnew < 0.f) fnew = 0.f;
  float df = fnew - fold;
  float change = df * (0.5f * df / r.Ainv + res);
  if (change > 1e-10f) return 0.f;
#pragma unroll
  for (int q = 0; q < NARM; q++) a.arm[q] += L.Minv[q][d] * sg * df;
  r.f = fnew;
  return -change;
}

// What only PGS needs per contact, kept in the registers of the contact's lane for the whole solve: the diagonal
// block A = J Minv J' + R of the dual problem (packed lower triangle) and the spectral form D Ac D = Q diag(lam) Q'
// of its mu-scaled friction block.
struct PgsReg { float A[21], Q[25], lam[5]; };

DEV void pgs_block(const EnvLDS& L, const Contact& c, PgsReg& P) {
  int dim = c.dim;
  float A[6][6];
#pragma unroll
  for (int j = 0; j < 6; j++)
#pragma unroll
    for (int k = 0; k < 6; k++) A[j][k] = 0.f;
#pragma unroll
  for (int side = 0; side < 2; side++) {
    int d = side == 0 ? c.d1 : c.d2;
    if (d >= NARM) {
      int f = d - NARM;
      float r[3] = {c.pos[0] - L.xipos[d][0], c.pos[1] - L.xipos[d][1], c.pos[2] - L.xipos[d][2]};
      float ul[6][3], ua[6][3], Iua[6][3];
#pragma unroll
      for (int j = 0; j < 6; j++) {
        const float* u = &c.frame[3 * (j % 3)];
        if (j < 3) { ul[j][0] = u[0]; ul[j][1] = u[1]; ul[j][2] = u[2]; cross3(ua[j], r, u); }
        else { ul[j][0] = ul[j][1] = ul[j][2] = 0.f; ua[j][0] = u[0]; ua[j][1] = u[1]; ua[j][2] = u[2]; }
        symvec3(Iua[j], L.fIinv[f], ua[j]);
      }
      float mi = L.fminv[f];
#pragma unroll
      for (int j = 0; j < 6; j++)
#pragma unroll
        for (int k = 0; k <= j; k++) A[j][k] += mi * dot3(ul[j], ul[k]) + dot3(ua[j], Iua[k]);
    }
  }
  if (c.armslot >= 0) {
    const ArmCon& ac = L.armcon[c.armslot];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      float Bk[NARM];                    // column k of Minv J^T
#pragma unroll
      for (int q = 0; q < NARM; q++) {
        float v = 0.f;
#pragma unroll
        for (int s = 0; s < NARM; s++) v += L.Minv[q][s] * ac.Jt[s][k];
        Bk[q] = v;
      }
#pragma unroll
      for (int j = k; j < 6; j++) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < NARM; q++) v += ac.Jt[q][j] * Bk[q];
        A[j][k] += v;
      }
    }
  }
  const float Rj[6] = {c.R[0], c.R[1], c.R[1], c.R[2], c.R[3], c.R[3]};
#pragma unroll
  for (int j = 0; j < 6; j++) A[j][j] += Rj[j];
#pragma unroll
  for (int j = 0; j < 6; j++)
#pragma unroll
    for (int k = 0; k <= j; k++) P.A[j * (j + 1) / 2 + k] = A[j][k];
  // Spectral form of the friction block for the cone QCQP (mju_QCQP): with D = diag(mu_j) the scaled block
  // D Ac D = Q diag(lam) Q^T is decomposed ONCE per substep (cyclic Jacobi, lane = contact); every Newton step
  // on the cone multiplier inside the PGS sweep is then O(5) instead of a 5x5 Cholesky factorisation.  Rows
  // >= dim are decoupled (identity).
  const float fr5[5] = {c.fric[0], c.fric[0], c.fric[1], c.fric[2], c.fric[2]};
  int nf = dim - 1;
  float S[5][5], Qm[5][5];
#pragma unroll
  for (int i = 0; i < 5; i++)
#pragma unroll
    for (int k = 0; k < 5; k++) {
      float aik = i >= k ? A[i + 1][k + 1] : A[k + 1][i + 1];        // only the lower triangle of A is filled
      S[i][k] = (i < nf && k < nf) ? aik * fr5[i] * fr5[k] : (i == k ? 1.f : 0.f);
      Qm[i][k] = i == k ? 1.f : 0.f;
    }
  for (int sweep = 0; sweep < 6; sweep++) {
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
      for (int q = p + 1; q < 5; q++) {
        float apq = S[p][q];
        if (fabsf(apq) > 1e-30f) {
          float app = S[p][p], aqq = S[q][q];
          float tau = (aqq - app) / (2.f * apq);
          float t = (tau >= 0.f ? 1.f : -1.f) / (fabsf(tau) + sqrtf(1.f + tau * tau));
          float cs = 1.f / sqrtf(1.f + t * t), sn = t * cs;
#pragma unroll
          for (int k = 0; k < 5; k++) {
            if (k != p && k != q) {
              float skp = S[k][p], skq = S[k][q];
              float np_ = cs * skp - sn * skq, nq_ = sn * skp + cs * skq;
              S[k][p] = np_; S[p][k] = np_; S[k][q] = nq_; S[q][k] = nq_;
            }
            float qkp = Qm[k][p], qkq = Qm[k][q];
            Qm[k][p] = cs * qkp - sn * qkq; Qm[k][q] = sn * qkp + cs * qkq;
          }
          S[p][p] = app - t * apq; S[q][q] = aqq + t * apq; S[p][q] = 0.f; S[q][p] = 0.f;
        }
      }
  }
#pragma unroll
  for (int i = 0; i < 5; i++) {
    P.lam[i] = fmaxf(S[i][i], 1e-30f);
#pragma unroll
    for (int k = 0; k < 5; k++) P.Q[5 * i + k] = Qm[i][k];
  }
}

// one PGS update of an elliptic contact block; returns the cost decrease
DEV float contact_update(const EnvLDS& L, Contact& c, const PgsReg& P, Acc& a) {
  int dim = c.dim;
  if (dim == 0) return 0.f;
  float res[6], old[6], f[6];
  jacc_reg(L, c, a, res);
  const float Rj[6] = {c.R[0], c.R[1], c.R[1], c.R[2], c.R[3], c.R[3]};
  const float fr[5] = {c.fric[0], c.fric[0], c.fric[1], c.fric[2], c.fric[2]};
  float A[6][6];
#pragma unroll
  for (int j = 0; j < 6; j++)
#pragma unroll
    for (int q = 0; q <= j; q++) { float v = P.A[j * (j + 1) / 2 + q]; A[j][q] = v; A[q][j] = v; }
#pragma unroll
  for (int j = 0; j < 6; j++) { old[j] = c.f[j]; f[j] = old[j]; res[j] = (j < dim) ? res[j] - c.aref[j] + Rj[j] * old[j] : 0.f; }
  // normal / ray update
  if (f[0] < MINVAL_F) {
    f[0] -= res[0] / A[0][0];
    if (f[0] < 0.f) f[0] = 0.f;
#pragma unroll
    for (int j = 1; j < 6; j++) f[j] = 0.f;
  } else {
    float denom = 0.f, vr = 0.f;
#pragma unroll
    for (int j = 0; j < 6; j++) {
      float v1 = 0.f;
#pragma unroll
      for (int q = 0; q < 6; q++) v1 += A[j][q] * f[q];
      denom += f[j] * v1; vr += f[j] * res[j];
    }
    if (denom >= MINVAL_F) {
      float x = -vr / denom;
      if (f[0] + x * f[0] < 0.f) x = -1.f;
#pragma unroll
      for (int j = 0; j < 6; j++) f[j] += x * old[j];
    }
  }
  // friction update with the normal fixed
  if (f[0] >= MINVAL_F && dim > 1) {
    float bc[5], v[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
      float b = res[j + 1];
#pragma unroll
      for (int q = 0; q < 5; q++) b -= A[j + 1][q + 1] * old[q + 1];
      b += A[j + 1][0] * (f[0] - old[0]);
      bc[j] = (j + 1 < dim) ? b : 0.f;
    }
    // mju_QCQP: min 0.5 v'Ac v + bc'v  s.t.  sum (v_j/mu_j)^2 <= fn^2.  In the scaled variable y = v/mu and the
    // eigenbasis of D Ac D (c.Q, c.lam, built once per substep) the stationarity condition (D Ac D + la I) y = -D bc
    // reads z_i = -g_i / (lam_i + la); the Newton iteration on the multiplier la is MuJoCo's
    // (val = |y|^2 - r^2, deriv = -2 y'(P^-1)y) evaluated in O(5).  la = 0 is the unconstrained minimum.
    float g[5], z[5], r2 = f[0] * f[0], la = 0.f;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 5; j++) s += P.Q[5 * j + i] * (bc[j] * fr[j]);
      g[i] = s;
    }
    for (int iter = 0; iter < 20; iter++) {
      float val = -r2, deriv = 0.f;
#pragma unroll
      for (int i = 0; i < 5; i++) {
        float inv = 1.f / (P.lam[i] + la);
        z[i] = -g[i] * inv;
        val += z[i] * z[i];
        deriv -= 2.f * z[i] * z[i] * inv;
      }
      // fp64 MuJoCo stops at val < 1e-10 / delta < 1e-10; in fp32 the thresholds sit at the rounding level of val
      if (val <= (la == 0.f ? 1e-10f : 2e-6f * r2)) break;
      float delta = -val / deriv;
      if (!(delta > 1e-6f * la)) break;
      la += delta;
    }
#pragma unroll
    for (int j = 0; j < 5; j++) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 5; i++) s += P.Q[5 * j + i] * z[i];
      v[j] = (j + 1 < dim) ? s * fr[j] : 0.f;
    }
    if (la != 0.f) {          // constraint active: put v exactly on the cone (no drift)
      float s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 5; j++) s2 += (v[j] / fr[j]) * (v[j] / fr[j]);
      float sc = sqrtf(r2 / fmaxf(MINVAL_F, s2));
#pragma unroll
      for (int j = 0; j < 5; j++) v[j] *= sc;
    }
#pragma unroll
    for (int j = 0; j < 5; j++) f[j + 1] = v[j];
  }
  float df[6], change = 0.f;
#pragma unroll
  for (int j = 0; j < 6; j++) df[j] = f[j] - old[j];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    float v1 = 0.f;
#pragma unroll
    for (int q = 0; q < 6; q++) v1 += A[j][q] * df[q];
    change += df[j] * (0.5f * v1 + res[j]);
  }
  if (change > 1e-10f) return 0.f;      // cost went up: keep the old forces
  apply_reg(L, c, df, a);
#pragma unroll
  for (int j = 0; j < 6; j++) c.f[j] = f[j];
  return -change;
}

DEV void solve_pgs(const DevModel* m, EnvLDS& L, int max_iter, float tolerance) {
  int lane = wave_lane();
  int nrow = L.nrow, ncon = L.ncon;
  if (lane == 0) L.iters = 0;
  if (nrow + ncon == 0) { wave_sync(); return; }
  // this solver's lane layout holds MAXCON_PGS contacts; further ones are dropped and flagged
  if (ncon > MAXCON_PGS) { ncon = MAXCON_PGS; if (lane == 0) L.overflow |= 2; }
  // ---- islands (uniform): union the dynamic bodies each contact couples
  int root[3] = {0, 1, 2};
  for (int k = 0; k < ncon; k++) {
    int g1 = body_group(L.con[k].d1), g2 = body_group(L.con[k].d2);
    if (g1 >= 0 && g2 >= 0) {
      int r1 = g1 == 0 ? root[0] : (g1 == 1 ? root[1] : root[2]);
      int r2 = g2 == 0 ? root[0] : (g2 == 1 ? root[1] : root[2]);
      int lo = r1 < r2 ? r1 : r2, hi = r1 < r2 ? r2 : r1;
#pragma unroll
      for (int i = 0; i < 3; i++) if (root[i] == hi) root[i] = lo;
    }
  }
  bool owner = lane < 3 && (lane == 0 ? root[0] == 0 : (lane == 1 ? root[1] == 1 : root[2] == 2));
  auto island_of = [&](const Contact& c) {
    int g = body_group(c.d1 >= 0 ? c.d1 : c.d2);
    return g == 0 ? root[0] : (g == 1 ? root[1] : root[2]);
  };
  Acc smooth; acc_load(L, smooth);          // uniform copy of the unconstrained accelerations
  // ---- warm start: forces from the previous qacc through the primal->force map (lane = row / contact)
  {
    Acc w;
#pragma unroll
    for (int d = 0; d < NARM; d++) w.arm[d] = L.warm[d];
#pragma unroll
    for (int f = 0; f < NFREE; f++) {
      int b = NARM + f;
      const float* wq = &L.warm[NARM + 6 * f];
      float wb[3] = {wq[3], wq[4], wq[5]}, alp[3];
      matvec3(alp, L.xmat[b], wb);
      float r[3] = {L.xipos[b][0] - L.xpos[b][0], L.xipos[b][1] - L.xpos[b][1], L.xipos[b][2] - L.xpos[b][2]};
      float ww[3] = {L.fvel[f][3], L.fvel[f][4], L.fvel[f][5]};
      float t1[3], t2[3];
      cross3(t1, alp, r); cross3(t2, ww, r); cross3(t2, ww, t2);
#pragma unroll
      for (int i = 0; i < 3; i++) { w.fr[f][i] = wq[i] + t1[i]; w.fr[f][3 + i] = alp[i]; }
    }
    if (lane < nrow) {
      Row1& r = L.row[lane];
      float jar = r.sign * L.warm[r.dof] - r.aref;
      float f = -jar / r.R;
      if (r.floss > 0.f) f = fminf(fmaxf(f, -r.floss), r.floss);
      else f = jar < 0.f ? f : 0.f;
      r.f = f;
    }
    if (lane < ncon) {
      Contact& c = L.con[lane];
      float jar[6];
      jacc_reg(L, c, w, jar);
      const float Rj[6] = {c.R[0], c.R[1], c.R[1], c.R[2], c.R[3], c.R[3]};
      const float fr[5] = {c.fric[0], c.fric[0], c.fric[1], c.fric[2], c.fric[2]};
      int dim = c.dim;
#pragma unroll
      for (int j = 0; j < 6; j++) jar[j] -= c.aref[j];
      float mu = c.mu, U[6], T = 0.f;
      U[0] = jar[0] * mu;
#pragma unroll
      for (int j = 1; j < 6; j++) { U[j] = (j < dim) ? jar[j] * fr[j - 1] : 0.f; T += U[j] * U[j]; }
      T = sqrtf(T);
      float N = U[0], fo[6];
      if ((N >= mu * T) || (T <= 0.f && N >= 0.f)) {
#pragma unroll
        for (int j = 0; j < 6; j++) fo[j] = 0.f;
      } else if ((mu * N + T <= 0.f) || (T <= 0.f && N < 0.f)) {
#pragma unroll
        for (int j = 0; j < 6; j++) fo[j] = (j < dim) ? -jar[j] / Rj[j] : 0.f;
      } else {
        float Dm = (1.f / Rj[0]) / fmaxf(mu * mu * (1.f + mu * mu), MINVAL_F), NmT = N - mu * T;
        fo[0] = -Dm * NmT * mu;
#pragma unroll
        for (int j = 1; j < 6; j++) fo[j] = (j < dim) ? -fo[0] / T * U[j] * fr[j - 1] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 6; j++) c.f[j] = fo[j];
    }
  }
  wave_sync();
  // ---- accelerations produced by the warm forces: each island owner applies its own rows
  Acc a = smooth;
  if (owner) {
    if (lane == 0) {
      for (int k = 0; k < nrow; k++) {
        const Row1& r = L.row[k];
#pragma unroll
        for (int q = 0; q < NARM; q++) a.arm[q] += L.Minv[q][r.dof] * r.sign * r.f;
      }
    }
    for (int k = 0; k < ncon; k++) {
      const Contact& c = L.con[k];
      if (island_of(c) != lane) continue;
      float df[6];
#pragma unroll
      for (int j = 0; j < 6; j++) df[j] = c.f[j];
      apply_reg(L, c, df, a);
    }
    // publish the bodies this island owns: scratch[0..5] arm, [8..13] object, [16..21] container
#pragma unroll
    for (int g = 0; g < 3; g++) {
      int rg = g == 0 ? root[0] : (g == 1 ? root[1] : root[2]);
      if (rg == lane) {
#pragma unroll
        for (int i = 0; i < 6; i++) L.scratch[8 * g + i] = g == 0 ? a.arm[i] : a.fr[g - 1][i];
      }
    }
  }
  wave_sync();
  // ---- dual cost  sum f.(0.5 (A f) + b),  A f = J (acc_w - smooth) + R f,  b = J smooth - aref
  float cost = 0.f;
  {
    Acc aw;
#pragma unroll
    for (int i = 0; i < 6; i++) { aw.arm[i] = L.scratch[i]; aw.fr[0][i] = L.scratch[8 + i]; aw.fr[1][i] = L.scratch[16 + i]; }
    if (lane < nrow) {
      const Row1& r = L.row[lane];
      float jn = r.sign * L.scratch[r.dof], js = r.sign * L.qacc_arm[r.dof];
      cost += r.f * (0.5f * (jn - js + r.R * r.f) + js - r.aref);
    }
    if (lane < ncon) {
      const Contact& c = L.con[lane];
      float jn6[6], js6[6];
      jacc_reg(L, c, aw, jn6); jacc_reg(L, c, smooth, js6);
      const float Rj[6] = {c.R[0], c.R[1], c.R[1], c.R[2], c.R[3], c.R[3]};
#pragma unroll
      for (int j = 0; j < 6; j++) if (j < c.dim) cost += c.f[j] * (0.5f * (jn6[j] - js6[j] + Rj[j] * c.f[j]) + js6[j] - c.aref[j]);
    }
  }
  cost = wave_sum_f(cost);
  if (cost > 0.f) {             // worse than zero forces: cold start
    if (lane < nrow) L.row[lane].f = 0.f;
    if (lane < ncon) {
#pragma unroll
      for (int j = 0; j < 6; j++) L.con[lane].f[j] = 0.f;
    }
    a = smooth;
  }
  wave_sync();
  // ---- publish the starting accelerations (warm or smooth): L.qacc_arm / L.facc are the shared island state
  if (owner) {
#pragma unroll
    for (int g = 0; g < 3; g++) {
      int rg = g == 0 ? root[0] : (g == 1 ? root[1] : root[2]);
      if (rg == lane) {
#pragma unroll
        for (int i = 0; i < 6; i++) { if (g == 0) L.qacc_arm[i] = a.arm[i]; else L.facc[g - 1][i] = a.fr[g - 1][i]; }
      }
    }
  }
  wave_sync();
  // ---- main iteration, lane = constraint block.  Lane k < MAXCON_PGS keeps contact k (A block, friction-block
  // inverse, frame, aref, R, forces) in REGISTERS for the whole solve; lane 32+r keeps scalar row r.  A sweep
  // runs in turns: at turn t every lane whose block is the t-th of its island (MuJoCo order: scalar rows, then
  // contacts) updates at once — islands advance in parallel, blocks of one island stay sequential, so the
  // iterates are those of the plain Gauss-Seidel sweep.  Only the island accelerations travel through LDS.
  bool has_con = lane < ncon, has_row = lane >= 32 && lane - 32 < nrow;
  Contact creg;
  PgsReg preg;
  Row1 rreg;
  if (has_con) { creg = L.con[lane]; pgs_block(L, creg, preg); }
  if (has_row) rreg = L.row[lane - 32];
  int myroot = has_con ? island_of(creg) : (has_row ? root[0] : -1);
  unsigned long long below = (1ull << lane) - 1ull;
  unsigned long long mk0 = wave_ballot(has_con && myroot == 0), mk1 = wave_ballot(has_con && myroot == 1),
                     mk2 = wave_ballot(has_con && myroot == 2);
  int rows_in0 = root[0] == 0 ? nrow : 0, rows_in1 = root[0] == 1 ? nrow : 0, rows_in2 = root[0] == 2 ? nrow : 0;
  int mypos = -1;
  if (has_row) mypos = lane - 32;
  if (has_con) {
    unsigned long long mine = myroot == 0 ? mk0 : (myroot == 1 ? mk1 : mk2);
    mypos = (myroot == root[0] ? nrow : 0) + __popcll(mine & below);
  }
  int len0 = rows_in0 + __popcll(mk0), len1 = rows_in1 + __popcll(mk1), len2 = rows_in2 + __popcll(mk2);
  int turns = len0 > len1 ? len0 : len1;
  turns = turns > len2 ? turns : len2;
  float scale = 1.f / (m->meaninertia * (float)NV);
  int it = 0;
  for (; it < max_iter; it++) {
    float improvement = 0.f;
    for (int t = 0; t < turns; t++) {
      if (has_row && mypos == t) {
        Acc b;
#pragma unroll
        for (int q = 0; q < NARM; q++) b.arm[q] = L.qacc_arm[q];
        improvement += row_update(L, rreg, b);
#pragma unroll
        for (int q = 0; q < NARM; q++) L.qacc_arm[q] = b.arm[q];
      }
      if (has_con && mypos == t) {
        Acc b;
        acc_load(L, b);
        improvement += contact_update(L, creg, preg, b);
        if (creg.armslot >= 0) {
#pragma unroll
          for (int q = 0; q < NARM; q++) L.qacc_arm[q] = b.arm[q];
        }
#pragma unroll
        for (int f = 0; f < NFREE; f++) {
          if (creg.d1 == NARM + f || creg.d2 == NARM + f) {
#pragma unroll
            for (int i = 0; i < 6; i++) L.facc[f][i] = b.fr[f][i];
          }
        }
      }
      wave_sync();
    }
    improvement = wave_sum_f(improvement);
    if (improvement * scale < tolerance) { it++; break; }
  }
  // forces back to LDS (diagnostics / debug dump); accelerations are already there
  if (has_con) {
#pragma unroll
    for (int j = 0; j < 6; j++) L.con[lane].f[j] = creg.f[j];
  }
  if (has_row) L.row[lane - 32].f = rreg.f;
  if (lane == 0) L.iters = it;
  wave_sync();
}
