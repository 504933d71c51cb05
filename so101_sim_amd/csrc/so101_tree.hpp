// General-tree engine (the ALOHA hand-over scenes): a substep for models outside the SO100 topology -
// any tree of bodies with one hinge / slide joint each, free bodies, position actuators, joint damping, joint
// equalities - with every dimension read from the model at run time.
//
// One env = one wavefront, as everywhere in this library, but nothing here is specialised: lanes take bodies, dofs,
// constraint rows or contacts in turn, matrices live in LDS, the constraint Jacobian in a per-env global scratch.  It is the
// FIRST GPU path for these scenes, written for parity with the fp64 CPU restatement used by the tests (the same stages in the
// same order); the SO100 kernels of so101_device.hpp / so101_newton.hpp stay the fast path for the headline workload.
// The narrowphase (support functions, flat-face scan, MPR, patch contacts) is shared with them.
#include "so101_device.hpp"

// Two builds of this file share one library (csrc/tu_tree.hip, csrc/tu_tree64.hip; TREE_VARIANT), each inside its own namespace:
//   32 (ALOHA hand-over): 32 dofs, 40 positions, 128 geoms, 64 contacts, 384 rows - LDS 40 KB per env (25 KB + the constraint
//      rows of substeps with at most 96 of them): four envs per CU;
//   64 (Dining, SURVEY 8f-4: two arms + six free props = 52 dofs, 240 geoms, ~105 contacts while the props land): 64 dofs, 64
//      positions, 256 geoms, 128 contacts, 768 rows - LDS 70 KB per env, the Cholesky a call instead of inlined code.
// so101_tree_create picks the build from the model's dimensions (csrc/tu_tree_api.hip).
#ifndef TREE_VARIANT
#define TREE_VARIANT 32
#endif
#undef TB
#undef TV
#undef TQ
#undef TU
#undef TJ
#undef TE
#undef TFR
#undef TCON
#undef TROW
#undef TCAND
#undef TGEOM
#undef TJS
#undef TRL
#undef TREE_NS
#undef T_SCRATCH
#define TB 32          // bodies (world included)
#define TU 16          // actuators
#define TJ 24          // one-dof joints
#define TE 4           // joint equalities
#define TFR 16         // dofs with frictionloss
#if TREE_VARIANT == 64
#define TREE_NS tv64
#define TV 64          // dofs
#define TQ 64          // generalized positions
#define TCON 160       // contacts per env (round 5: 128 -> 160, hull pairs on flat features carry up to five contacts now and the six props landing after a reset went over 128)
#define TROW 960       // constraint rows per env (6 per contact at most)
#define TCAND 512      // broadphase candidates per env
#define TGEOM 256      // collision geoms
#define TJS 64         // row stride of the constraint Jacobian scratch
#define TRL 0          // constraint rows kept in LDS (none: 70 KB of LDS per env already)
#undef TREE_CHOL_DEV
#define TREE_CHOL_DEV __device__ __attribute__((noinline))      // 64 x 64 unrolled pivot steps: one copy of the code, called from the three sites
#else
#define TREE_NS tv32
#define TV 32
#define TQ 40
#define TCON 64
#define TROW 384
#define TCAND 256
#define TGEOM 128
#define TJS 32
#ifndef TREE_TRL32
#define TREE_TRL32 96
#endif
#define TRL TREE_TRL32 // constraint rows kept in LDS while a substep has no more (round 4): Jacobian + per-row vectors, 15 KB aliased with CRBA / RNE / collision scratch
#undef TREE_CHOL_DEV
#define TREE_CHOL_DEV DEV
#endif

#ifndef TREE_DIM_T
#define TREE_DIM_T unsigned char     // per-contact row counts (0-6) in LDS
#endif

namespace TREE_NS {

enum { TJ_NONE = 0, TJ_HINGE = 1, TJ_FREE = 2, TJ_SLIDE = 3 };
enum { TR_FRICTION = 0, TR_LIMIT = 1, TR_CONTACT = 2, TR_EQUALITY = 3 };

struct TreeModel {
  int nq, nv, nu, nbody, ngeom, npair, njnt, nfree, neq, nfric, maxdepth, elliptic, iterations, any_damping, pad0, pad1;
  float dt, grav[3], impratio, tolerance, meaninertia, pad2;
  int body_parent[TB], body_jnttype[TB], body_qposadr[TB], body_dofadr[TB], body_depth[TB], body_jnt[TB];
  unsigned int body_anc[TB];       // bit a: body a is this body or one of its ancestors
  unsigned long long body_dofs[TB]; // bit d: dof d moves this body
  float body_pos[TB][3], body_quat[TB][4], body_ipos[TB][3], body_iquat[TB][4], body_mass[TB], body_inertia[TB][3], body_invweight0[TB][2];
  int dof_body[TV], dof_jnt[TV];   // joint index of a one-dof joint, -1 for the dofs of a free body
  float dof_armature[TV], dof_damping[TV], dof_frictionloss[TV], dof_invweight0[TV], dof_solref[TV][2], dof_solimp[TV][5];
  int jnt_body[TJ], jnt_limited[TJ], jnt_actfrclimited[TJ];
  float jnt_axis[TJ][3], jnt_range[TJ][2], jnt_solref[TJ][2], jnt_solimp[TJ][5], jnt_actfrcrange[TJ][2];
  int act_dof[TU], act_qposadr[TU], act_ctrllimited[TU], act_forcelimited[TU];
  float act_gain[TU], act_bias[TU][3], act_ctrlrange[TU][2], act_forcerange[TU][2];
  int eq_dof[TE][2], eq_qposadr[TE][2];
  float eq_polycoef[TE][5], eq_solref[TE][2], eq_solimp[TE][5];
  int fric_dof[TFR];
  const int* geom_body;            // [ngeom]
  const float* geom_solmix;        // [ngeom]
  const int* geom_priority;        // [ngeom]
  const int* geom_class;           // [ngeom] task classes: 1 object, 2 container, 4 below the left gripper_link, 8 below the right one (or NULL)
};

struct TreeBuffers { float *qpos, *qvel, *ctrl, *warm; float* scratch; int* diag; };   // [nq|nv|nu|nv][N] env-fastest; scratch [N][T_SCRATCH]; diag [N][8]

// Per-env working set that does not fit the LDS budget (occupancy is bounded by LDS here): the constraint Jacobian, the per-row
// vectors of the solver and the contacts' Hessian blocks.  Written and read by the env's own wavefront only (through the CU's L1).
#define T_SCRATCH (TROW * TJS + 8 * TROW + TCON * 36)       // floats per env
// Row stride of the Jacobian rows kept in LDS: TJS + 1.  The passes with lane = row (residuals J x, J search, reference accelerations) read column d
// of 64 different rows at once; with a stride of TJS = 32 words all of them fall on ONE bank (64-way conflict, 64 cycles per ds_read), with 33 they
// spread over all 32 (round 5).  Rows in the global scratch keep TJS (whole 128 / 256-byte lines per row).
#undef TJL
#define TJL (TJS + 1)
struct TreeScratch {
  float* global_base;              // the env's slice of the global scratch (the pointers below may point into LDS instead, see use_row_storage)
  float* J;                        // [TROW][js]
  int js;                          // row stride of J: TJS in the global scratch, TJL in LDS
  float *eD, *eR, *earef, *efl, *ejar, *ef, *ejv;   // [TROW] each
  unsigned int* etype;             // [TROW]
  float (*Hc)[36];                 // [TCON]
};

struct TCon {                      // 32 words
  float pos[3], frame[9], dist, mu, fric[5], solref[2], solimp[5];
  int g1, g2, b1, b2, dim, row;
};

struct TreeLDS {
  float qpos[TQ], qvel[TV], ctrl[TU], warm[TV], qacc[TV], qsm[TV], bias[TV], qfrc[TV], qact[TV];
  float xpos[TB][3], xquat[TB][4], xmat[TB][9], xipos[TB][3], ximat[TB][9];
  float S[TV][6];
  union {                                                // the phases of a substep share this storage
    struct {
      float Ic[TB][36];                                  // CRBA: spatial inertia about the world origin, each body's own, then composite (in place)
      float rw[TB][3], ral[TB][3], rao[TB][3], rf[TB][3], rn[TB][3];      // RNE
    };
    struct { float aabb[6][TGEOM]; unsigned int cand[TCAND]; };   // collision: world boxes of the geoms, candidate pairs
    struct { float xJ[(TRL ? TRL : 1) * TJL]; float xv[8 * (TRL ? TRL : 1)]; };      // constraints + solver: the rows' Jacobian (row stride TJL) and per-row vectors (nrow <= TRL)
  };
  float M[TV][TV + 1];
  union { float L[TV][TV + 1]; float H[TV][TV + 1]; };  // factor of M (until qacc_smooth is known), then the Newton Hessian / M + h D
  TCon con[TCON];
  unsigned long long bdofs[TB];     // tm->body_dofs, copied by make_constraints(): the per-contact loops index it by the contact's bodies (round 5: a global load per contact before)
  float x[TV], grad[TV], search[TV], Ma[TV], tmp[TV];
  TREE_DIM_T cdim[TCON + 1];
  TREE_DIM_T hdim[TCON];           // Newton: rows of the contact's Hessian block, 0 when the block is inactive
  int ncand, ncon, nrow, nscalar, iters, flags;
#ifdef TREE_PROF
  unsigned int prof[32];            // (profiling variant, scripts/gpu_tree_prof.py: phase clocks of a substep, 10 ns ticks)
#endif
};
#ifdef TREE_PROF
#define TPROF_T0() unsigned long long tp_ = wall_clock64()
#define TPROF(k) { unsigned long long tn_ = wall_clock64(); if (wave_lane() == 0) L.prof[k] += (unsigned int)(tn_ - tp_); tp_ = tn_; }
#else
#define TPROF_T0()
#define TPROF(k)
#endif

namespace tree {

DEV TreeScratch scratch_of(const TreeBuffers& B, int e) {
  float* base = B.scratch + (size_t)e * T_SCRATCH;
  TreeScratch G;
  G.global_base = base;
  G.J = base; G.js = TJS; base += TROW * TJS;
  G.eD = base; G.eR = base + TROW; G.earef = base + 2 * TROW; G.efl = base + 3 * TROW; G.ejar = base + 4 * TROW; G.ef = base + 5 * TROW; G.ejv = base + 6 * TROW;
  G.etype = (unsigned int*)(base + 7 * TROW); base += 8 * TROW;
  G.Hc = (float (*)[36])base;
  return G;
}

// ------------------------------------------------------------------ state in / out (env-fastest struct-of-arrays)
DEV void load_state(const TreeModel* tm, TreeLDS& L, const TreeBuffers& B, int e, int N) {
  int lane = wave_lane();
  if (lane < tm->nq) L.qpos[lane] = B.qpos[(size_t)lane * N + e];
  if (lane < tm->nv) { L.qvel[lane] = B.qvel[(size_t)lane * N + e]; L.warm[lane] = B.warm[(size_t)lane * N + e]; }
  if (lane < tm->nu) L.ctrl[lane] = B.ctrl[(size_t)lane * N + e];
  wave_sync();
}
DEV void store_state(const TreeModel* tm, TreeLDS& L, const TreeBuffers& B, int e, int N) {
  int lane = wave_lane();
  wave_sync();
  if (lane < tm->nq) B.qpos[(size_t)lane * N + e] = L.qpos[lane];
  if (lane < tm->nv) { B.qvel[(size_t)lane * N + e] = L.qvel[lane]; B.warm[(size_t)lane * N + e] = L.warm[lane]; }
}

// ------------------------------------------------------------------ kinematics: lane = body, one tree level at a time
DEV void kinematics(const TreeModel* tm, TreeLDS& L) {
  int lane = wave_lane(), nb = tm->nbody;
  if (lane == 0) {
    L.xpos[0][0] = L.xpos[0][1] = L.xpos[0][2] = 0.f;
    L.xquat[0][0] = 1.f; L.xquat[0][1] = L.xquat[0][2] = L.xquat[0][3] = 0.f;
#pragma unroll
    for (int i = 0; i < 9; i++) { L.xmat[0][i] = (i % 4 == 0) ? 1.f : 0.f; L.ximat[0][i] = (i % 4 == 0) ? 1.f : 0.f; }
    L.xipos[0][0] = L.xipos[0][1] = L.xipos[0][2] = 0.f;
  }
  wave_sync();
  // the body's constants once, in front of the level loop (round 5: inside it every level began with a round trip to the model tables)
  const bool has = lane > 0 && lane < nb;
  const int b = has ? lane : 0;
  const int depth = has ? tm->body_depth[b] : -1, p = tm->body_parent[b], jt = tm->body_jnttype[b], qa = tm->body_qposadr[b];
  float bpos[3], bquat[4], bipos[3], im[9], ax[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 3; i++) { bpos[i] = tm->body_pos[b][i]; bipos[i] = tm->body_ipos[b][i]; }
#pragma unroll
  for (int i = 0; i < 4; i++) bquat[i] = tm->body_quat[b][i];
  quat2mat(im, tm->body_iquat[b]);
  if (jt == TJ_HINGE || jt == TJ_SLIDE) {
    const float* a = tm->jnt_axis[tm->body_jnt[b]];
    ax[0] = a[0]; ax[1] = a[1]; ax[2] = a[2];
  }
  const int maxdepth = tm->maxdepth;
  for (int level = 1; level <= maxdepth; level++) {
    if (depth == level) {
      float xp[3], xq[4];
      if (jt == TJ_FREE) {
        xp[0] = L.qpos[qa]; xp[1] = L.qpos[qa + 1]; xp[2] = L.qpos[qa + 2];
        xq[0] = L.qpos[qa + 3]; xq[1] = L.qpos[qa + 4]; xq[2] = L.qpos[qa + 5]; xq[3] = L.qpos[qa + 6];
        normquat(xq);
      } else {
        float t[3]; matvec3(t, L.xmat[p], bpos);
        xp[0] = L.xpos[p][0] + t[0]; xp[1] = L.xpos[p][1] + t[1]; xp[2] = L.xpos[p][2] + t[2];
        mulquat(xq, L.xquat[p], bquat);
        if (jt == TJ_HINGE) {
          float sn, cs; sincos_f(0.5f * L.qpos[qa], &sn, &cs);
          float jq[4] = {cs, ax[0] * sn, ax[1] * sn, ax[2] * sn}, o[4];
          mulquat(o, xq, jq);
          xq[0] = o[0]; xq[1] = o[1]; xq[2] = o[2]; xq[3] = o[3];
        }
        normquat(xq);
        if (jt == TJ_SLIDE) {
          float a[3]; rotvecquat(a, ax, xq);
          float q = L.qpos[qa];
          xp[0] += a[0] * q; xp[1] += a[1] * q; xp[2] += a[2] * q;
        }
      }
      float xm[9]; quat2mat(xm, xq);
      float t[3]; matvec3(t, xm, bipos);
      float xim[9]; matmul3(xim, xm, im);
#pragma unroll
      for (int i = 0; i < 3; i++) { L.xpos[b][i] = xp[i]; L.xipos[b][i] = xp[i] + t[i]; }
#pragma unroll
      for (int i = 0; i < 4; i++) L.xquat[b][i] = xq[i];
#pragma unroll
      for (int i = 0; i < 9; i++) { L.xmat[b][i] = xm[i]; L.ximat[b][i] = xim[i]; }
    }
    wave_sync();
  }
  // motion axes: S[d] = (omega ; velocity of the point at the world origin), lane = dof
  if (lane < tm->nv) {
    int d = lane, b = tm->dof_body[d], jt = tm->body_jnttype[b];
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (jt == TJ_HINGE || jt == TJ_SLIDE) {
      float a[3]; matvec3(a, L.xmat[b], tm->jnt_axis[tm->dof_jnt[d]]);
      if (jt == TJ_HINGE) { s[0] = a[0]; s[1] = a[1]; s[2] = a[2]; cross3(s + 3, L.xpos[b], a); }
      else { s[3] = a[0]; s[4] = a[1]; s[5] = a[2]; }
    } else {
      int k = d - tm->body_dofadr[b];
      if (k < 3) s[3 + k] = 1.f;
      else {
        float a[3] = {L.xmat[b][k - 3], L.xmat[b][3 + k - 3], L.xmat[b][6 + k - 3]};
        s[0] = a[0]; s[1] = a[1]; s[2] = a[2]; cross3(s + 3, L.xpos[b], a);
      }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) L.S[d][i] = s[i];
  }
  wave_sync();
}

// ------------------------------------------------------------------ Cholesky: A = L L^T, lower part in place in LDS.  Lane i takes row i into
// registers; column j of the factor travels between lanes with v_readlane (right-looking: a[i][k] -= L[i][j] L[k][j], j ascending).
// Entries beyond n are never read back.
TREE_CHOL_DEV void chol_factor(float (*A)[TV + 1], int n) {
  int lane = wave_lane(), row = lane < TV ? lane : 0;
  wave_sync();
  float a[TV];
#pragma unroll
  for (int k = 0; k < TV; k++) a[k] = A[row][k];
#pragma unroll
  for (int j = 0; j < TV; j++) {
    if (j < n) {
      float ajj = wave_get_f(a[j], j);
      float d = sqrtf(fmaxf(ajj, MINVAL_F));
      float lij = lane == j ? d : a[j] / d;
      a[j] = lij;
#pragma unroll
      for (int k = j + 1; k < TV; k++) {
        float lkj = wave_get_f(lij, k);
        a[k] -= lij * lkj;
      }
    }
  }
  if (lane < n) {
#pragma unroll
    for (int k = 0; k < TV; k++) if (k <= lane) A[lane][k] = a[k];
  }
  wave_sync();
}
// x := A^-1 x for the factor above (x in LDS, n entries).  Register resident like the factorisation: lane i takes row i of L (forward
// substitution) and column i of L (backward) into registers, the solved entries travel with v_readlane - two barriers instead of four per
// row (the LDS version cost ~110 barriers per solve at 28 dofs; round 4).  Entries beyond n are never used.
TREE_CHOL_DEV void chol_solve(float (*A)[TV + 1], int n, float* x) {
  int lane = wave_lane(), row = lane < TV ? lane : 0;
  wave_sync();
  float a[TV], t[TV];
#pragma unroll
  for (int k = 0; k < TV; k++) { a[k] = A[row][k]; t[k] = A[k][row]; }       // (row stride TV + 1: both reads are conflict-free)
  float y = lane < n ? x[row] : 0.f;
  // 1 / L[lane][lane], one division per lane (round 5; before: two divisions of broadcast values per column, 2 x TV in all)
  float dg = 1.f;
#pragma unroll
  for (int k = 0; k < TV; k++) dg = (row == k) ? a[k] : dg;
  const float inv = 1.f / dg;
#pragma unroll
  for (int j = 0; j < TV; j++) {
    if (j < n) {
      float yj = wave_get_f(y, j) * wave_get_f(inv, j);
      if (lane == j) y = yj; else if (lane > j) y -= a[j] * yj;
    }
  }
#pragma unroll
  for (int j = TV - 1; j >= 0; j--) {
    if (j < n) {
      float xj = wave_get_f(y, j) * wave_get_f(inv, j);
      if (lane == j) y = xj; else if (lane < j) y -= t[j] * xj;
    }
  }
  if (lane < n) x[lane] = y;
  wave_sync();
}

// ------------------------------------------------------------------ CRBA: composite spatial inertias about the world origin
DEV void crba(const TreeModel* tm, TreeLDS& L) {
  int lane = wave_lane(), nb = tm->nbody, nv = tm->nv;
  if (lane < nb) {
    int b = lane;
    float o[36];
#pragma unroll
    for (int i = 0; i < 36; i++) o[i] = 0.f;
    float mass = tm->body_mass[b];
    if (b > 0 && mass > 0.f) {
      const float* c = L.xipos[b]; const float* R = L.ximat[b];
      float I[9];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          float v = 0.f;
#pragma unroll
          for (int k = 0; k < 3; k++) v += R[3 * i + k] * tm->body_inertia[b][k] * R[3 * j + k];
          I[3 * i + j] = v;
        }
      float cx[9] = {0.f, -c[2], c[1], c[2], 0.f, -c[0], -c[1], c[0], 0.f}, cx2[9];
      matmul3(cx2, cx, cx);
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          o[6 * i + j] = I[3 * i + j] - mass * cx2[3 * i + j];
          o[6 * i + 3 + j] = mass * cx[3 * i + j];
          o[6 * (3 + i) + j] = -mass * cx[3 * i + j];
          o[6 * (3 + i) + 3 + j] = (i == j) ? mass : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 36; i++) L.Ic[b][i] = o[i];
  }
  wave_sync();
  {                                      // composite of body b: its own plus every descendant's, in body order; in place
    int b = lane < nb ? lane : 0;
    float o[36];
#pragma unroll
    for (int i = 0; i < 36; i++) o[i] = L.Ic[b][i];
    for (int d = b + 1; d < nb; d++)
      if ((tm->body_anc[d] >> b) & 1u) {
#pragma unroll
        for (int i = 0; i < 36; i++) o[i] += L.Ic[d][i];
      }
    wave_sync();                         // (every lane has read the bodies' own inertias before any composite replaces one)
    if (lane < nb) {
#pragma unroll
      for (int i = 0; i < 36; i++) L.Ic[b][i] = o[i];
    }
  }
  wave_sync();
  // M[r][c] = S_r' Ic[deeper body] S_c when one of the two bodies is an ancestor of the other; lane = column
  if (lane < nv) {
    int c = lane, bc = tm->dof_body[c];
    float sc[6];
#pragma unroll
    for (int i = 0; i < 6; i++) sc[i] = L.S[c][i];
    for (int r = 0; r < nv; r++) {
      int br = tm->dof_body[r], bb = -1;
      if ((tm->body_anc[br] >> bc) & 1u) bb = br; else if ((tm->body_anc[bc] >> br) & 1u) bb = bc;
      float v = 0.f;
      if (bb >= 0) {
        const float* I = L.Ic[bb];
#pragma unroll
        for (int i = 0; i < 6; i++) {
          float t = 0.f;
#pragma unroll
          for (int j = 0; j < 6; j++) t += I[6 * i + j] * sc[j];
          v += L.S[r][i] * t;
        }
      }
      if (r == c) v += tm->dof_armature[c];
      L.M[r][c] = v; L.L[r][c] = v;
    }
  }
  wave_sync();
  chol_factor(L.L, nv);
}

// ------------------------------------------------------------------ RNE bias (Coriolis / centrifugal + gravity at qacc = 0)
DEV void rne_bias(const TreeModel* tm, TreeLDS& L) {
  int lane = wave_lane(), nb = tm->nbody;
  if (lane < nb) {
#pragma unroll
    for (int k = 0; k < 3; k++) { L.rw[lane][k] = 0.f; L.ral[lane][k] = 0.f; L.rao[lane][k] = -tm->grav[k]; L.rf[lane][k] = 0.f; L.rn[lane][k] = 0.f; }
  }
  wave_sync();
  for (int level = 1; level <= tm->maxdepth; level++) {
    if (lane > 0 && lane < nb && tm->body_depth[lane] == level) {
      int b = lane, p = tm->body_parent[b], jt = tm->body_jnttype[b], da = tm->body_dofadr[b];
      float wb[3], ab[3], aob[3];
      if (jt == TJ_FREE) {
        float v3[3] = {L.qvel[da + 3], L.qvel[da + 4], L.qvel[da + 5]};
        matvec3(wb, L.xmat[b], v3);
#pragma unroll
        for (int k = 0; k < 3; k++) { ab[k] = 0.f; aob[k] = -tm->grav[k]; }
      } else {
        float d[3] = {L.xpos[b][0] - L.xpos[p][0], L.xpos[b][1] - L.xpos[p][1], L.xpos[b][2] - L.xpos[p][2]};
        float t1[3], t2[3], t3[3];
        cross3(t1, L.ral[p], d); cross3(t2, L.rw[p], d); cross3(t3, L.rw[p], t2);
#pragma unroll
        for (int k = 0; k < 3; k++) { aob[k] = L.rao[p][k] + t1[k] + t3[k]; wb[k] = L.rw[p][k]; ab[k] = L.ral[p][k]; }
        if (jt == TJ_HINGE || jt == TJ_SLIDE) {
          float a[3]; matvec3(a, L.xmat[b], tm->jnt_axis[tm->body_jnt[b]]);
          float qd = L.qvel[da], t[3]; cross3(t, L.rw[p], a);
          if (jt == TJ_HINGE) {
#pragma unroll
            for (int k = 0; k < 3; k++) { wb[k] += a[k] * qd; ab[k] += t[k] * qd; }
          } else {
#pragma unroll
            for (int k = 0; k < 3; k++) aob[k] += 2.f * t[k] * qd;
          }
        }
      }
      float mass = tm->body_mass[b];
      float F[3] = {0.f, 0.f, 0.f}, Nn[3] = {0.f, 0.f, 0.f};
      if (mass > 0.f) {
        float r[3] = {L.xipos[b][0] - L.xpos[b][0], L.xipos[b][1] - L.xpos[b][1], L.xipos[b][2] - L.xpos[b][2]};
        float t1[3], t2[3], t3[3]; cross3(t1, ab, r); cross3(t2, wb, r); cross3(t3, wb, t2);
#pragma unroll
        for (int k = 0; k < 3; k++) F[k] = mass * (aob[k] + t1[k] + t3[k]);
        const float* R = L.ximat[b];
        float lw[3], la[3], Iw[3], Ia[3];
        matTvec3(lw, R, wb); matTvec3(la, R, ab);
#pragma unroll
        for (int k = 0; k < 3; k++) { lw[k] *= tm->body_inertia[b][k]; la[k] *= tm->body_inertia[b][k]; }
        matvec3(Iw, R, lw); matvec3(Ia, R, la);
        float N3[3], rF[3]; cross3(N3, wb, Iw); cross3(rF, r, F);
#pragma unroll
        for (int k = 0; k < 3; k++) Nn[k] = Ia[k] + N3[k] + rF[k];
      }
#pragma unroll
      for (int k = 0; k < 3; k++) { L.rw[b][k] = wb[k]; L.ral[b][k] = ab[k]; L.rao[b][k] = aob[k]; L.rf[b][k] = F[k]; L.rn[b][k] = Nn[k]; }
    }
    wave_sync();
  }
  // force and moment (about the body's own origin) of the subtree of each dof's body; lane = dof
  if (lane < tm->nv) {
    int d = lane, b = tm->dof_body[d], jt = tm->body_jnttype[b];
    float F[3] = {0.f, 0.f, 0.f}, Nn[3] = {0.f, 0.f, 0.f};
    for (int c = b; c < nb; c++)
      if ((tm->body_anc[c] >> b) & 1u) {
        float dd[3] = {L.xpos[c][0] - L.xpos[b][0], L.xpos[c][1] - L.xpos[b][1], L.xpos[c][2] - L.xpos[b][2]}, t[3];
        cross3(t, dd, L.rf[c]);
#pragma unroll
        for (int k = 0; k < 3; k++) { F[k] += L.rf[c][k]; Nn[k] += L.rn[c][k] + t[k]; }
      }
    float v;
    if (jt == TJ_HINGE || jt == TJ_SLIDE) {
      float a[3]; matvec3(a, L.xmat[b], tm->jnt_axis[tm->dof_jnt[d]]);
      v = jt == TJ_HINGE ? dot3(a, Nn) : dot3(a, F);
    } else {
      int k = d - tm->body_dofadr[b];
      if (k < 3) v = F[k];
      else { float t[3]; matTvec3(t, L.xmat[b], Nn); v = t[k - 3]; }
    }
    L.bias[d] = v;
  }
  wave_sync();
}

// ------------------------------------------------------------------ actuation, passive forces, unconstrained acceleration
DEV void smooth(const TreeModel* tm, TreeLDS& L) {
  int lane = wave_lane(), nv = tm->nv;
  if (lane < nv) L.qact[lane] = 0.f;
  wave_sync();
  if (lane < tm->nu) {                              // (every actuator of these scenes drives its own dof)
    int a = lane, d = tm->act_dof[a];
    float c = L.ctrl[a];
    if (tm->act_ctrllimited[a]) c = fminf(fmaxf(c, tm->act_ctrlrange[a][0]), tm->act_ctrlrange[a][1]);
    float force = tm->act_gain[a] * c + tm->act_bias[a][0] + tm->act_bias[a][1] * L.qpos[tm->act_qposadr[a]] + tm->act_bias[a][2] * L.qvel[d];
    if (tm->act_forcelimited[a]) force = fminf(fmaxf(force, tm->act_forcerange[a][0]), tm->act_forcerange[a][1]);
    L.qact[d] = force;
  }
  wave_sync();
  if (lane < nv) {
    int d = lane, j = tm->dof_jnt[d];
    float f = L.qact[d];
    if (j >= 0 && tm->jnt_actfrclimited[j]) f = fminf(fmaxf(f, tm->jnt_actfrcrange[j][0]), tm->jnt_actfrcrange[j][1]);
    float rhs = f - L.bias[d] - tm->dof_damping[d] * L.qvel[d];
    L.qfrc[d] = rhs; L.qsm[d] = rhs;
  }
  wave_sync();
  chol_solve(L.L, nv, L.qsm);
}

// ------------------------------------------------------------------ collision
DEV void geom_pose(const TreeModel* tm, const DevModel* gm, const TreeLDS& L, int g, float* p, float* R) {
  int b = tm->geom_body[g];
  const float* gp = gm->geom_pos + 3 * g; const float* gmat = gm->geom_mat + 9 * g;
  float lp[3] = {gp[0], gp[1], gp[2]}, lm[9], t[3];
#pragma unroll
  for (int i = 0; i < 9; i++) lm[i] = gmat[i];
  matvec3(t, L.xmat[b], lp);
  p[0] = L.xpos[b][0] + t[0]; p[1] = L.xpos[b][1] + t[1]; p[2] = L.xpos[b][2] + t[2];
  matmul3(R, L.xmat[b], lm);
}

DEV void contact_params(const TreeModel* tm, const DevModel* gm, TCon& c, int g1, int g2) {      // mj_contactParam
  int cd1 = gm->geom_condim[g1], cd2 = gm->geom_condim[g2];
  c.dim = cd1 > cd2 ? cd1 : cd2;
  float f[3];
#pragma unroll
  for (int k = 0; k < 3; k++) f[k] = fmaxf(gm->geom_friction[3 * g1 + k], gm->geom_friction[3 * g2 + k]);
  c.fric[0] = c.fric[1] = f[0]; c.fric[2] = f[1]; c.fric[3] = c.fric[4] = f[2];
  float s1 = tm->geom_solmix[g1], s2 = tm->geom_solmix[g2];
  float mix = (s1 >= MINVAL_F && s2 >= MINVAL_F) ? s1 / (s1 + s2) : (s1 < MINVAL_F && s2 < MINVAL_F ? 0.5f : (s1 < MINVAL_F ? 0.f : 1.f));
  int p1 = tm->geom_priority[g1], p2 = tm->geom_priority[g2];
  if (p1 > p2) mix = 1.f; else if (p1 < p2) mix = 0.f;
#pragma unroll
  for (int k = 0; k < 2; k++) c.solref[k] = mix * gm->geom_solref[2 * g1 + k] + (1.f - mix) * gm->geom_solref[2 * g2 + k];
#pragma unroll
  for (int k = 0; k < 5; k++) c.solimp[k] = mix * gm->geom_solimp[5 * g1 + k] + (1.f - mix) * gm->geom_solimp[5 * g2 + k];
}

// candidate pairs of the current poses into L.cand, in pair-list order (world boxes, then the oriented-box filter); clears the contact count
DEV int broadphase(const TreeModel* tm, const DevModel* gm, TreeLDS& L) {
  int lane = wave_lane(), ng = tm->ngeom;
  // world boxes of the geoms
  for (int g = lane; g < ng; g += WAVE) {
    float p[3], R[9]; geom_pose(tm, gm, L, g, p, R);
    const float* c = gm->geom_aabb + 6 * g;
    float cl[3] = {c[0], c[1], c[2]}, h[3] = {c[3], c[4], c[5]}, cw[3];
    matvec3(cw, R, cl);
#pragma unroll
    for (int i = 0; i < 3; i++) {
      float e = fabsf(R[3 * i]) * h[0] + fabsf(R[3 * i + 1]) * h[1] + fabsf(R[3 * i + 2]) * h[2];
      L.aabb[i][g] = p[i] + cw[i] - e; L.aabb[3 + i][g] = p[i] + cw[i] + e;
    }
  }
  if (lane == 0) { L.ncand = 0; L.ncon = 0; }
  wave_sync();
  // candidate pairs in pair-list order.  Four blocks of 64 pairs per trip (round 5): their pair words are four loads in flight and their box
  // reads overlap; one block per trip paid a round trip to the pair list each (360 trips for the 23 016 pairs of the Dining scenes: 197 us of the
  // 720 us an env-substep took).  The ballots run block by block, so the candidate order is the list's.
  int base = 0;
  const int npair = tm->npair;
  for (int p0 = 0; p0 < npair; p0 += 4 * WAVE) {
    unsigned int pk[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { int p = p0 + u * WAVE + lane; pk[u] = p < npair ? gm->pair_packed[p] : 0xffffffffu; }   // geom1 | geom2 << 8 | plane flag << 16, types ordered
    bool hit[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      int g1 = (int)(pk[u] & 0xffu), g2 = (int)((pk[u] >> 8) & 0xffu);
      hit[u] = false;
      if (pk[u] != 0xffffffffu) {
        if ((pk[u] >> 16) & 1u) {
          float pp[3], R[9]; geom_pose(tm, gm, L, g1, pp, R);
          float nrm[3] = {R[2], R[5], R[8]}, lowest = 0.f;
#pragma unroll
          for (int k = 0; k < 3; k++) lowest += nrm[k] * ((nrm[k] >= 0.f ? L.aabb[k][g2] : L.aabb[3 + k][g2]) - pp[k]);
          hit[u] = !(lowest > 0.f);
        } else {
          // (all twelve box reads first, then one combined test: with `||` inside the loop every axis was its own LDS round trip behind a branch)
          float lo1[3], hi1[3], lo2[3], hi2[3];
#pragma unroll
          for (int k = 0; k < 3; k++) { lo1[k] = L.aabb[k][g1]; hi1[k] = L.aabb[3 + k][g1]; lo2[k] = L.aabb[k][g2]; hi2[k] = L.aabb[3 + k][g2]; }
          bool sep = false;
#pragma unroll
          for (int k = 0; k < 3; k++) sep = sep | (lo1[k] > hi2[k]) | (lo2[k] > hi1[k]);
          hit[u] = !sep;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      unsigned long long mask = wave_ballot(hit[u]);
      int idx = base + wave_prefix(mask);
      if (hit[u] && idx < TCAND) L.cand[idx] = (pk[u] & 0xffu) | (((pk[u] >> 8) & 0xffu) << 16);
      base += __popcll(mask);
    }
  }
  wave_sync();
  int ncand = base;
  if (ncand > TCAND) { ncand = TCAND; if (lane == 0) L.flags |= 1; }
  // Second pass, lane = candidate (as obb_filter of so101_device.hpp for the SO100 scenes): separating-axis test of the two geoms'
  // ORIENTED boxes (15 axes).  The world box of a long tilted arm link overlaps many hulls it is nowhere near; a pair whose oriented
  // boxes are more than 1e-6 m apart cannot touch, so dropping it changes no contact - it spares the wavefront a narrowphase query that
  // would end in "no intersection" (of the ALOHA scenes' 34 candidates per env about half).  Plane pairs pass.  Order kept.
  {
    int nout = 0;
    for (int k0 = 0; k0 < ncand; k0 += WAVE) {
      int k = k0 + lane;
      bool keep = false; unsigned int cg = 0u;
      if (k < ncand) {
        cg = L.cand[k];
        int g1 = (int)(cg & 0xffffu), g2 = (int)(cg >> 16);
        keep = true;
        if (gm->geom_type[g1] != G_PLANE) {
          float A[9], pa[3], Bm[9], pb[3];
          geom_pose(tm, gm, L, g1, pa, A); geom_pose(tm, gm, L, g2, pb, Bm);
          const float* aa = gm->geom_aabb + 6 * g1; const float* ab = gm->geom_aabb + 6 * g2;
          float ca[3], cb[3], la[3] = {aa[0], aa[1], aa[2]}, lb[3] = {ab[0], ab[1], ab[2]}, a[3] = {aa[3], aa[4], aa[5]}, b[3] = {ab[3], ab[4], ab[5]};
          matvec3(ca, A, la); matvec3(cb, Bm, lb);
          float dv[3] = {pb[0] + cb[0] - pa[0] - ca[0], pb[1] + cb[1] - pa[1] - ca[1], pb[2] + cb[2] - pa[2] - ca[2]};
          float Rm[3][3], Ab[3][3], t[3];
#pragma unroll
          for (int i = 0; i < 3; i++) {
            t[i] = A[i] * dv[0] + A[3 + i] * dv[1] + A[6 + i] * dv[2];
#pragma unroll
            for (int j = 0; j < 3; j++) { Rm[i][j] = A[i] * Bm[j] + A[3 + i] * Bm[3 + j] + A[6 + i] * Bm[6 + j]; Ab[i][j] = fabsf(Rm[i][j]) + 1e-6f; }
          }
          const float gap = 1e-6f;
          bool sep = false;
#pragma unroll
          for (int i = 0; i < 3; i++) sep = sep || fabsf(t[i]) > a[i] + b[0] * Ab[i][0] + b[1] * Ab[i][1] + b[2] * Ab[i][2] + gap;
#pragma unroll
          for (int j = 0; j < 3; j++) sep = sep || fabsf(t[0] * Rm[0][j] + t[1] * Rm[1][j] + t[2] * Rm[2][j]) > a[0] * Ab[0][j] + a[1] * Ab[1][j] + a[2] * Ab[2][j] + b[j] + gap;
#pragma unroll
          for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
              const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
              float ra = a[i1] * Ab[i2][j] + a[i2] * Ab[i1][j], rb = b[j1] * Ab[i][j2] + b[j2] * Ab[i][j1];
              sep = sep || fabsf(t[i2] * Rm[i1][j] - t[i1] * Rm[i2][j]) > ra + rb + gap;
            }
          keep = !sep;
          // round 6: further separating directions from the hulls' support-bound tables (so101_device.hpp sbt_separated)
          if (keep && gm->hull_sbt && gm->geom_type[g2] == G_MESH) {
            float cwa[3] = {pa[0] + ca[0], pa[1] + ca[1], pa[2] + ca[2]}, cwb[3] = {pb[0] + cb[0], pb[1] + cb[1], pb[2] + cb[2]};
            keep = !sbt_separated(gm, g1, g2, A, cwa, a, Bm, cwb, t);
          }
        } else if (gm->hull_sbt && gm->geom_type[g2] == G_MESH) {
          float A[9], pa[3], Bm[9], pb[3];
          geom_pose(tm, gm, L, g1, pa, A); geom_pose(tm, gm, L, g2, pb, Bm);
          const float* ab = gm->geom_aabb + 6 * g2;
          float lb[3] = {ab[0], ab[1], ab[2]}, cb[3];
          matvec3(cb, Bm, lb);
          float cwb[3] = {pb[0] + cb[0], pb[1] + cb[1], pb[2] + cb[2]}, n[3] = {A[2], A[5], A[8]};
          keep = !sbt_plane_separated(gm, g2, pa, n, Bm, cwb);
        }
      }
      unsigned long long mask = wave_ballot(keep);
      int idx = nout + wave_prefix(mask);
      wave_sync();                                 // every lane has read its candidate before the slots are rewritten
      if (keep) L.cand[idx] = cg;
      nout += __popcll(mask);
    }
    wave_sync();
    ncand = nout;
  }
  if (lane == 0) L.ncand = ncand;
  wave_sync();
  return ncand;
}

// one contact of the pair (g1, g2) into slot `slot` of the env's contact list (one lane)
DEV void write_contact(const TreeModel* tm, const DevModel* gm, TCon& c, int g1, int g2, int b1, int b2, const float* nrm, float dist, const float* pos) {
  float fr[9] = {nrm[0], nrm[1], nrm[2], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  make_frame(fr);
#pragma unroll
  for (int i = 0; i < 9; i++) c.frame[i] = fr[i];
  c.pos[0] = pos[0]; c.pos[1] = pos[1]; c.pos[2] = pos[2]; c.dist = dist;
  c.g1 = g1; c.g2 = g2; c.b1 = b1; c.b2 = b2;
  contact_params(tm, gm, c, g1, g2);
  if (!tm->elliptic) c.dim = 1;
}

DEV void collision(const TreeModel* tm, const DevModel* gm, TreeLDS& L, bool narrow = true) {      // narrow = false: timing runs (broadphase only)
  int lane = wave_lane();
  int ncand = broadphase(tm, gm, L);
  int ncon = 0;
  for (int k = 0; k < (narrow ? ncand : 0); k++) {
    unsigned int cg = L.cand[k];
    int g1 = wave_uniform_i((int)(cg & 0xffffu)), g2 = wave_uniform_i((int)(cg >> 16));
    int b1 = wave_uniform_i(tm->geom_body[g1]), b2 = wave_uniform_i(tm->geom_body[g2]);
    GeomW G1, G2;
    load_geom_at(gm, g1, L.xpos[b1], L.xmat[b1], G1); load_geom_at(gm, g2, L.xpos[b2], L.xmat[b2], G2);
    PairContacts pc;
    narrow_pair<HullCache, G64>(gm, G1, G2, g1, g2, pc);       // (registers are free here: the LDS footprint, not VGPRs, bounds occupancy)
    unsigned int valid = pc.valid;
    int n = __popc(valid);
    if (n == 0) continue;
    if (ncon + n > TCON) { if (lane == 0) L.flags |= 2; break; }
    if (lane < NCPP && ((valid >> lane) & 1u)) {
      int slot = ncon + __popc(valid & ((1u << lane) - 1u));
      float dist = 0.f, pos[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NCPP; j++) if (j == lane) { dist = pc.dist[j]; pos[0] = pc.pos[j][0]; pos[1] = pc.pos[j][1]; pos[2] = pc.pos[j][2]; }
      write_contact(tm, gm, L.con[slot], g1, g2, b1, b2, pc.nrm, dist, pos);
    }
    ncon += n;
    wave_sync();
  }
  if (lane == 0) { L.ncon = ncon; L.ncand = ncand; }
  wave_sync();
}

// ------------------------------------------------------------------ the narrowphase in a launch of its own (round 4, tu_tree.hip: k_tree_pipe_*)
// A control step as a launch chain: per substep one launch with a wavefront per CANDIDATE PAIR of the whole batch (persistent wavefronts
// pulling from a work list, as k_narrow of the SO100 step) and one with a wavefront per env for everything else.  Inside k_tree_step the env's
// wavefront walks its 18 (ALOHA) to 59 (Dining) candidates one after the other at ~12 us each while the hull caches and the EPA polytope
// keep the kernel at one wavefront per SIMD; out of it, the pairs of all envs are balanced over the machine.  Hand-off through global
// memory between launches of one stream: body poses and candidates out, contact records back (count, normal, then (dist, position) per
// contact, compact, in slot order), the env's state through its home buffers.  Same functions on the same data: bit-identical to the fused step.
#define TREC 24                  // floats per contact record
#define TPIPE_MAXSUB 64          // work-list slots per step: the substeps, plus one for the contacts of the post-step state (contact rewards)
struct TreePipe {
  float* pose;                   // [N][TB][12] xpos, xmat of every body
  unsigned int* cand;            // [N][TCAND] geom1 | geom2 << 16
  int* ncand;                    // [N]
  float* rec;                    // [N][TCAND][TREC]
  unsigned int* work;            // [2][work_cap] env * TCAND + k, double buffered over substeps (per env slice)
  unsigned int work_cap;         // envs of the slice * TCAND
  int* counters;                 // [TPIPE_MAXSUB][2] work items, cursor
  unsigned char* active;         // [N] 0 not stepping in this call (auto-reset), 1 stepping, 2 diverged
  int* pflags;                   // [N] event flags of the step so far
  int* pdiag;                    // [N][4] rows, solver iterations, contacts, candidates of the env's last substep (diagnostics of a step that ends in k_tree_pipe_finish)
};

// hands the candidates of the current poses to substep s
DEV void publish(const TreeModel* tm, const DevModel* gm, TreeLDS& L, const TreePipe& P, int e, int N, int s) {
  int lane = wave_lane(), nb = tm->nbody;
  for (int i = lane; i < nb * 12; i += WAVE) {
    int b = i / 12, j = i % 12;
    P.pose[(size_t)e * (TB * 12) + i] = j < 3 ? L.xpos[b][j] : L.xmat[b][j - 3];
  }
  int ncand = broadphase(tm, gm, L);
  int base = 0;
  if (lane == 0) { base = ncand ? atomicAdd(&P.counters[2 * s], ncand) : 0; P.ncand[e] = ncand; }
  base = wave_bcast_i(base, 0);
  unsigned int* list = P.work + (size_t)(s & 1) * P.work_cap;
  for (int k = lane; k < ncand; k += WAVE) { P.cand[(size_t)e * TCAND + k] = L.cand[k]; list[base + k] = (unsigned int)e * TCAND + (unsigned int)k; }
}

// the env's contacts of this substep from the records, in candidate order, truncated at TCON like the fused loop
DEV void gather_contacts(const TreeModel* tm, const DevModel* gm, TreeLDS& L, const TreePipe& P, int e) {
  // Round 5: the pairs' counts are read by lane = candidate and scanned with three ballots (a count has three bits), which gives every pair its first
  // slot at once; only the pairs that HAVE contacts are then visited, eight per trip, one lane per contact as in collision().  Before, one
  // wavefront walked all candidates and waited for a record's count from global memory before it looked at the next (18-59 round trips per
  // env-substep: 16 / 40 us on the hand-over / Dining scenes).  (The records are written lane = contact, not lane = pair: with the pair's lane writing
  // all of its contacts the inlined copy of write_contact() rounded the tangent frames differently from collision()'s copy - last bit, every contact
  // that is not axis-aligned - and the chain stopped being bit-identical to the fused step.)
  int lane = wave_lane(), ncand = P.ncand[e], ncon = 0;
  bool full = false;
  for (int k0 = 0; k0 < ncand && !full; k0 += WAVE) {
    int k = k0 + lane;
    int n = k < ncand ? (int)P.rec[((size_t)e * TCAND + k) * TREC] : 0;
    unsigned long long b0 = wave_ballot((n & 1) != 0), b1 = wave_ballot((n & 2) != 0), b2 = wave_ballot((n & 4) != 0);
    int slot = ncon + wave_prefix(b0) + 2 * wave_prefix(b1) + 4 * wave_prefix(b2);
    unsigned long long over = wave_ballot(n > 0 && slot + n > TCON);       // the first pair that does not fit ends the list (flag 2), as in collision()
    unsigned long long todo = wave_ballot(n > 0);
    if (over) {
      int first = (int)__builtin_ctzll(over);
      todo &= (1ull << first) - 1ull;
      ncon = wave_bcast_i(slot, first);
      full = true;
      if (lane == 0) L.flags |= 2;
    } else ncon += __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
    while (todo) {
      // up to eight pairs with contacts per trip: lane = (pair of the trip, contact of the pair)
      const int p = lane >> 3, j = lane & 7;
      int l = -1;
      unsigned long long t = todo;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        if (t) { if (q == p) l = (int)__builtin_ctzll(t); t &= t - 1ull; }
      }
      todo = t;
      int nn = wave_bcast_i(n, l < 0 ? 0 : l), ss = wave_bcast_i(slot, l < 0 ? 0 : l);
      if (l >= 0 && j < nn) {
        int kk = k0 + l;
        const float* r = P.rec + ((size_t)e * TCAND + kk) * TREC;
        unsigned int cg = P.cand[(size_t)e * TCAND + kk];
        int g1 = (int)(cg & 0xffffu), g2 = (int)(cg >> 16);
        float nrm[3] = {r[1], r[2], r[3]}, pos[3] = {r[5 + 4 * j], r[6 + 4 * j], r[7 + 4 * j]};
        write_contact(tm, gm, L.con[ss + j], g1, g2, tm->geom_body[g1], tm->geom_body[g2], nrm, r[4 + 4 * j], pos);
      }
    }
  }
  if (lane == 0) { L.ncon = ncon; L.ncand = ncand; }
  wave_sync();
}

// ------------------------------------------------------------------ constraint rows (mj_makeConstraint order: equality, dof friction, limits, contacts)
// impedance, regulariser and reference-acceleration coefficients of one row group
DEV void row_params(const TreeModel* tm, const float* solref_in, const float* solimp, float pos, float* imp, float* K, float* Bc) {
  float s0 = solref_in[0], s1 = solref_in[1];
  float dmax = fminf(fmaxf(solimp[1], MINIMP_F), MAXIMP_F);
  if (s0 > 0.f) {
    s0 = fmaxf(s0, 2.f * tm->dt);
    *K = 1.f / fmaxf(MINVAL_F, dmax * dmax * s0 * s0 * s1 * s1); *Bc = 2.f / fmaxf(MINVAL_F, dmax * s0);
  } else { *K = -s0 / fmaxf(MINVAL_F, dmax * dmax); *Bc = -s1 / fmaxf(MINVAL_F, dmax); }
  *imp = impedance(solimp, pos);
}

// Where the rows of THIS substep live: in LDS when they fit (nrow <= TRL; the ALOHA hand-over scenes have ~80), in the env's global scratch
// otherwise.  Measured in round 4 (experiment build, 3 envs per CU): with Jacobian and per-row vectors in LDS the Newton stage of an env
// takes 99 us instead of 187 - it is bound by the latency of its loads (section 8 of DESIGN.md).  The contacts' Hessian blocks stay global.
DEV void use_row_storage(TreeLDS& L, TreeScratch& G, int nrow) {
  if (TRL > 0 && nrow <= TRL) {
    G.J = L.xJ; G.js = TJL;
    G.eD = L.xv; G.eR = L.xv + TRL; G.earef = L.xv + 2 * TRL; G.efl = L.xv + 3 * TRL; G.ejar = L.xv + 4 * TRL; G.ef = L.xv + 5 * TRL; G.ejv = L.xv + 6 * TRL;
    G.etype = (unsigned int*)(L.xv + 7 * TRL);
  } else {
    float* base = G.global_base;
    G.J = base; G.js = TJS; base += TROW * TJS;
    G.eD = base; G.eR = base + TROW; G.earef = base + 2 * TROW; G.efl = base + 3 * TROW; G.ejar = base + 4 * TROW; G.ef = base + 5 * TROW; G.ejv = base + 6 * TROW;
    G.etype = (unsigned int*)(base + 7 * TROW);
  }
}

// init + (Jacobian row r) . v for a vector v in LDS, one row per lane.  Rows in the global scratch are fetched as float4 (row stride TJS: 16-byte
// aligned; a quarter of the load instructions, each of which touches 64 different lines); rows in LDS have the conflict-free stride TJL.
DEV float row_dot(const TreeScratch& G, int r, const float* v, int nv, float init) {
  float s = init;
  if (TRL == 0 || G.js == TJS) {
    const float4* p = (const float4*)(G.J + (size_t)r * TJS);
    for (int d = 0; d < nv; d += 4) {
      float4 a = p[d >> 2];
      s += a.x * v[d];
      if (d + 1 < nv) s += a.y * v[d + 1];
      if (d + 2 < nv) s += a.z * v[d + 2];
      if (d + 3 < nv) s += a.w * v[d + 3];
    }
  } else {
    const float* p = G.J + r * G.js;
    for (int d = 0; d < nv; d++) s += p[d] * v[d];
  }
  return s;
}

DEV void make_constraints(const TreeModel* tm, TreeLDS& L, TreeScratch& G) {
  int lane = wave_lane(), nv = tm->nv, neq = tm->neq, nfric = tm->nfric, njnt = tm->njnt;
  // joint limits that are violated: (joint, side) pairs in joint order
  bool lo_on = false, hi_on = false;
  float q_j = 0.f;
  if (lane < njnt && tm->jnt_limited[lane]) {
    q_j = L.qpos[tm->body_qposadr[tm->jnt_body[lane]]];
    lo_on = q_j - tm->jnt_range[lane][0] < 0.f; hi_on = tm->jnt_range[lane][1] - q_j < 0.f;
  }
  unsigned long long mlo = wave_ballot(lo_on), mhi = wave_ballot(hi_on);
  int nlim = __popcll(mlo) + __popcll(mhi);
  int nscalar = neq + nfric + nlim;
  int ncon = L.ncon;
  if (lane < TB) L.bdofs[lane] = lane < tm->nbody ? tm->body_dofs[lane] : 0ull;
  // first row of every contact
  for (int c = lane; c < ncon; c += WAVE) L.cdim[c] = L.con[c].dim;
  wave_sync();
  for (int c = lane; c < ncon; c += WAVE) { int crow = nscalar; for (int k = 0; k < c; k++) crow += L.cdim[k]; L.con[c].row = crow; }
  int nrow = nscalar, keep = 0;
  for (int k = 0; k < ncon; k++) {                     // contacts whose rows do not fit the row capacity are dropped (flag 4)
    if (nrow + L.cdim[k] > TROW) break;
    nrow += L.cdim[k]; keep++;
  }
  if (keep < ncon) { ncon = keep; if (lane == 0) { L.flags |= 4; L.ncon = keep; } }
  if (lane == 0) { L.nrow = nrow; L.nscalar = nscalar; }
  wave_sync();                              // (the collision stage's boxes and candidates, which the row storage may alias, are no longer read)
  use_row_storage(L, G, nrow);
  // zero the Jacobian rows of the scalar constraints
  for (int r = 0; r < nscalar; r++) if (lane < TJS) G.J[r * G.js + lane] = 0.f;
  wave_sync();
  // ---- scalar rows: one lane each
  {
    int r = -1, type = 0, d1 = -1, d2 = -1;
    float j1 = 0.f, j2 = 0.f, pos = 0.f, diag = 0.f, floss = 0.f;
    float solref[2] = {0.02f, 1.f}, solimp[5] = {0.9f, 0.95f, 0.001f, 0.5f, 2.f};
    if (lane < neq) {
      int e = lane; r = e; type = TR_EQUALITY;
      d1 = tm->eq_dof[e][0]; d2 = tm->eq_dof[e][1];
      const float* pc = tm->eq_polycoef[e];
      float q1 = L.qpos[tm->eq_qposadr[e][0]], q2 = L.qpos[tm->eq_qposadr[e][1]];
      float poly = pc[0] + q2 * (pc[1] + q2 * (pc[2] + q2 * (pc[3] + q2 * pc[4])));
      float deriv = pc[1] + q2 * (2.f * pc[2] + q2 * (3.f * pc[3] + q2 * 4.f * pc[4]));
      pos = q1 - poly; j1 = 1.f; j2 = -deriv;
      solref[0] = tm->eq_solref[e][0]; solref[1] = tm->eq_solref[e][1];
#pragma unroll
      for (int k = 0; k < 5; k++) solimp[k] = tm->eq_solimp[e][k];
      diag = tm->dof_invweight0[d1] + tm->dof_invweight0[d2];
    } else if (lane < neq + nfric) {
      int d = tm->fric_dof[lane - neq]; r = lane; type = TR_FRICTION;
      d1 = d; j1 = 1.f; floss = tm->dof_frictionloss[d];
      solref[0] = tm->dof_solref[d][0]; solref[1] = tm->dof_solref[d][1];
#pragma unroll
      for (int k = 0; k < 5; k++) solimp[k] = tm->dof_solimp[d][k];
      diag = tm->dof_invweight0[d];
    }
    // limit rows are taken by the joint's own lane (after the rows above, which belong to lanes < neq + nfric <= 20 < 32; the
    // two groups of lanes overlap, so the limits go through a second pass below)
    if (r >= 0) {
      float imp, K, Bc; row_params(tm, solref, solimp, pos, &imp, &K, &Bc);
      float R = fmaxf(MINVAL_F, (1.f - imp) * diag / imp);
      float vel = j1 * L.qvel[d1] + (d2 >= 0 ? j2 * L.qvel[d2] : 0.f);
      G.J[r * G.js + d1] = j1; if (d2 >= 0) G.J[r * G.js + d2] = j2;
      G.etype[r] = (unsigned int)type | ((unsigned int)d1 << 8) | ((unsigned int)(d2 >= 0 ? d2 : 0xff) << 16); G.eR[r] = R; G.eD[r] = 1.f / R; G.efl[r] = floss;
      G.earef[r] = -Bc * vel - (type == TR_FRICTION ? 0.f : K * imp * pos);
    }
  }
  if (lo_on || hi_on) {
    int j = lane, b = tm->jnt_body[j], d = tm->body_dofadr[b];
    int r0 = neq + nfric + wave_prefix(mlo) + wave_prefix(mhi);
    float solref[2] = {tm->jnt_solref[j][0], tm->jnt_solref[j][1]}, solimp[5];
#pragma unroll
    for (int k = 0; k < 5; k++) solimp[k] = tm->jnt_solimp[j][k];
    for (int side = 0; side < 2; side++) {
      if (!(side == 0 ? lo_on : hi_on)) continue;
      int r = r0 + (side == 1 && lo_on ? 1 : 0);
      float pos = side == 0 ? q_j - tm->jnt_range[j][0] : tm->jnt_range[j][1] - q_j, sg = side == 0 ? 1.f : -1.f;
      float imp, K, Bc; row_params(tm, solref, solimp, pos, &imp, &K, &Bc);
      float R = fmaxf(MINVAL_F, (1.f - imp) * tm->dof_invweight0[d] / imp);
      G.J[r * G.js + d] = sg;
      G.etype[r] = TR_LIMIT | ((unsigned int)d << 8) | (0xffu << 16); G.eR[r] = R; G.eD[r] = 1.f / R; G.efl[r] = 0.f;
      G.earef[r] = -Bc * sg * L.qvel[d] - K * imp * pos;
    }
  }
  // ---- contact rows: Jacobian by lane = dof (coalesced rows), parameters by lane = contact
  for (int c = 0; c < ncon; c++) {
    const TCon& C = L.con[c];
    int row = C.row, dim = C.dim, b1 = C.b1, b2 = C.b2;
    if (lane < TJS) {
      int d = lane;
      float jp[3] = {0.f, 0.f, 0.f}, jr[3] = {0.f, 0.f, 0.f};
      if (d < nv) {
        bool in1 = (L.bdofs[b1] >> d) & 1ull, in2 = (L.bdofs[b2] >> d) & 1ull;
        float sgn = (in2 ? 1.f : 0.f) - (in1 ? 1.f : 0.f);
        if (sgn != 0.f) {
          float t[3]; cross3(t, L.S[d], C.pos);
#pragma unroll
          for (int k = 0; k < 3; k++) { jp[k] = sgn * (L.S[d][3 + k] + t[k]); jr[k] = sgn * L.S[d][k]; }
        }
      }
      for (int j = 0; j < dim; j++) {
        const float* ax = &C.frame[3 * (j < 3 ? j : j - 3)];
        G.J[(row + j) * G.js + d] = j < 3 ? dot3(ax, jp) : dot3(ax, jr);
      }
    }
  }
  for (int ci = lane; ci < ncon; ci += WAVE) {
    TCon& C = L.con[ci];
    int row = C.row, dim = C.dim;
    float imp, K, Bc; row_params(tm, C.solref, C.solimp, C.dist, &imp, &K, &Bc);
    float tran = tm->body_invweight0[C.b1][0] + tm->body_invweight0[C.b2][0], rot = tm->body_invweight0[C.b1][1] + tm->body_invweight0[C.b2][1];
    float R0 = fmaxf(MINVAL_F, (1.f - imp) * tran / imp);
    float R[6];
    R[0] = R0;
    for (int j = 1; j < 6; j++) R[j] = fmaxf(MINVAL_F, (1.f - imp) * (j < 3 ? tran : rot) / imp);
    if (dim > 1) {
      R[1] = R[0] / fmaxf(MINVAL_F, tm->impratio);
      C.mu = C.fric[0] * sqrtf(R[1] / R[0]);
      for (int j = 2; j < dim; j++) R[j] = fmaxf(MINVAL_F, R[1] * C.fric[0] * C.fric[0] / (C.fric[j - 1] * C.fric[j - 1]));
    } else C.mu = 0.f;
    for (int j = 0; j < dim; j++) {
      G.etype[row + j] = TR_CONTACT; G.eR[row + j] = R[j]; G.eD[row + j] = 1.f / R[j]; G.efl[row + j] = 0.f;
      G.ejar[row + j] = Bc; G.ejv[row + j] = j == 0 ? K * imp * C.dist : 0.f;      // (parked until the pass below)
    }
  }
  wave_sync();
  // reference acceleration of the contact rows: aref = -B (J qvel) - K imp pos; lane = row
  for (int r = nscalar + lane; r < nrow; r += WAVE) {
    float vel = row_dot(G, r, L.qvel, nv, 0.f);
    G.earef[r] = -G.ejar[r] * vel - G.ejv[r];
  }
  wave_sync();
}

// ------------------------------------------------------------------ Newton (mj_solNewton restated)
// cost, force (= -ds/dr) and Hessian (6 x 6 storage, the leading dim x dim part is used) of one elliptic contact block at r (dim rows);
// returns the cost.  Every loop runs over the six possible rows with a guard, so that the small arrays stay in registers.
DEV float contact_block(const TCon& C, const float* D, int dim, const float* r, float* force, float* Hc, bool want_h) {
#pragma unroll
  for (int k = 0; k < 36; k++) Hc[k] = 0.f;
#pragma unroll
  for (int j = 0; j < 6; j++) force[j] = 0.f;
  if (dim == 1) {
    if (r[0] < 0.f) { force[0] = -D[0] * r[0]; Hc[0] = D[0]; return 0.5f * D[0] * r[0] * r[0]; }
    return 0.f;
  }
  float mu = C.mu, U[6], fr[6];
  U[0] = r[0] * mu; fr[0] = 0.f;
  float T = 0.f;
#pragma unroll
  for (int j = 1; j < 6; j++) { fr[j] = j < dim ? C.fric[j - 1] : 0.f; U[j] = j < dim ? r[j] * fr[j] : 0.f; T += U[j] * U[j]; }
  T = sqrtf(T);
  float N = U[0];
  if ((N >= mu * T) || (T <= 0.f && N >= 0.f)) return 0.f;
  if ((mu * N + T <= 0.f) || (T <= 0.f && N < 0.f)) {
    float cost = 0.f;
#pragma unroll
    for (int j = 0; j < 6; j++) if (j < dim) { force[j] = -D[j] * r[j]; cost += 0.5f * D[j] * r[j] * r[j]; Hc[j * 6 + j] = D[j]; }
    return cost;
  }
  float Dm = D[0] / fmaxf(mu * mu * (1.f + mu * mu), MINVAL_F), sN = N - mu * T;
  force[0] = -Dm * sN * mu;
#pragma unroll
  for (int j = 1; j < 6; j++) if (j < dim) force[j] = -force[0] / T * U[j] * fr[j];
  if (want_h) {
    Hc[0] = Dm * mu * mu;
#pragma unroll
    for (int k = 1; k < 6; k++) if (k < dim) {
      float mk = fr[k];
      float hk = -Dm * mu * mu * U[k] * mk / T;
      Hc[k] = hk; Hc[k * 6] = hk;
#pragma unroll
      for (int l = 1; l < 6; l++) if (l < dim) {
        float ml = fr[l];
        Hc[k * 6 + l] = Dm * mu * mu * mk * ml * U[k] * U[l] / (T * T) - Dm * sN * mu * mk * ml * ((k == l ? 1.f : 0.f) / T - U[k] * U[l] / (T * T * T));
      }
    }
  }
  return 0.5f * Dm * sN * sN;
}
// first and second derivative of one contact block's cost along the line r + alpha dr (closed form per zone - the zone of the point
// itself, as in mj_solNewton's line search; what the line search used to take from contact_block's full 6 x 6 Hessian)
DEV void contact_line(const TCon& C, const float* D, int dim, const float* r, const float* dr, float* d1, float* d2) {
  *d1 = 0.f; *d2 = 0.f;
  if (dim == 1) { if (r[0] < 0.f) { *d1 = D[0] * r[0] * dr[0]; *d2 = D[0] * dr[0] * dr[0]; } return; }
  float mu = C.mu, N = r[0] * mu, dN = dr[0] * mu, TT = 0.f, UdU = 0.f, dUdU = 0.f;
#pragma unroll
  for (int j = 1; j < 6; j++) {
    float fr = j < dim ? C.fric[j - 1] : 0.f;
    float u = j < dim ? r[j] * fr : 0.f, du = j < dim ? dr[j] * fr : 0.f;
    TT += u * u; UdU += u * du; dUdU += du * du;
  }
  float T = sqrtf(TT);
  if ((N >= mu * T) || (T <= 0.f && N >= 0.f)) return;
  if ((mu * N + T <= 0.f) || (T <= 0.f && N < 0.f)) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < 6; j++) if (j < dim) { a += D[j] * r[j] * dr[j]; b += D[j] * dr[j] * dr[j]; }
    *d1 = a; *d2 = b;
    return;
  }
  float Dm = D[0] / fmaxf(mu * mu * (1.f + mu * mu), MINVAL_F), sN = N - mu * T, iT = 1.f / T;
  float dT = UdU * iT, ddT = (dUdU - dT * dT) * iT, dS = dN - mu * dT;
  *d1 = Dm * sN * dS;
  *d2 = Dm * (dS * dS - sN * mu * ddT);
}

// scalar rows: cost, force, second derivative
DEV float scalar_block(int type, float D, float R, float fl, float r, float* force, float* h) {
  *h = 0.f;
  if (type == TR_EQUALITY) { *force = -D * r; *h = D; return 0.5f * D * r * r; }
  if (type == TR_FRICTION) {
    float rf = R * fl;
    if (r <= -rf) { *force = fl; return fl * (-0.5f * rf - r); }
    if (r >= rf) { *force = -fl; return fl * (-0.5f * rf + r); }
    *force = -D * r; *h = D; return 0.5f * D * r * r;
  }
  if (r < 0.f) { *force = -D * r; *h = D; return 0.5f * D * r * r; }
  *force = 0.f; return 0.f;
}

// total cost at the point L.x; with want: gradient in L.grad and Hessian in L.H (not yet factored); forces in L.ef, jar in L.ejar
DEV float total_cost(const TreeModel* tm, TreeLDS& L, const TreeScratch& G, bool want) {
  int lane = wave_lane(), nv = tm->nv, nrow = L.nrow, nscalar = L.nscalar, ncon = L.ncon;
  float part = 0.f;
  TPROF_T0();
  if (lane < nv) {
    float v = 0.f;
    for (int c = 0; c < nv; c++) v += L.M[lane][c] * L.x[c];
    L.Ma[lane] = v;
    part = 0.5f * (v - L.qfrc[lane]) * (L.x[lane] - L.qsm[lane]);
  }
  for (int r = lane; r < nrow; r += WAVE) {
    G.ejar[r] = row_dot(G, r, L.x, nv, -G.earef[r]);
  }
  wave_sync();
  TPROF(0)
  for (int r = lane; r < nscalar; r += WAVE) {
    float f, h;
    part += scalar_block((int)(G.etype[r] & 0xffu), G.eD[r], G.eR[r], G.efl[r], G.ejar[r], &f, &h);
    G.ef[r] = f; G.ejv[r] = h;        // (ejv doubles as the scalar rows' second derivative until the line search fills it)
  }
  for (int ci = lane; ci < ncon; ci += WAVE) {         // (one pass with at most 64 contacts; the 64-dof build holds up to 128)
    const TCon& C = L.con[ci];
    int dim = C.dim, row = C.row;
    float r6[6], f6[6], D6[6], Hc[36];
#pragma unroll
    for (int j = 0; j < 6; j++) { r6[j] = j < dim ? G.ejar[row + j] : 0.f; D6[j] = j < dim ? G.eD[row + j] : 0.f; }
    part += contact_block(C, D6, dim, r6, f6, Hc, want);
#pragma unroll
    for (int j = 0; j < 6; j++) if (j < dim) G.ef[row + j] = f6[j];
    if (want) {
      bool any = false;
#pragma unroll
      for (int k = 0; k < 36; k++) { G.Hc[ci][k] = Hc[k]; any = any || Hc[k] != 0.f; }
      L.hdim[ci] = any ? dim : 0;          // (0: the block is inactive, nothing to add to the Hessian)
    }
  }
  float cost = wave_sum_f(part);
  wave_sync();
  TPROF(1)
  if (!want) return cost;
  // gradient: Ma - qfrc_smooth - J' force; lane = dof
  if (lane < nv) {
    float g = L.Ma[lane] - L.qfrc[lane];
    const int js = G.js;
    int r = 0;
    for (; r + 8 <= nrow; r += 8) {                       // (eight rows' loads in flight; the subtractions keep their order)
      float a[8], f[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { a[u] = G.J[(r + u) * js + lane]; f[u] = G.ef[r + u]; }
#pragma unroll
      for (int u = 0; u < 8; u++) g -= a[u] * f[u];
    }
    for (; r < nrow; r++) g -= G.J[r * js + lane] * G.ef[r];
    L.grad[lane] = g;
  }
  TPROF(2)
  // Hessian: M + sum J' Hc J, assembled in LDS (L.H) over the dofs each row TOUCHES (round 4; before, lane = column accumulated every
  // contact's 6 x nv product in registers: 36 + 6 TV multiply-adds per lane and contact, two barriers, although a prop-on-table contact
  // moves 6 of the 28 / 52 dofs - 480 instructions per contact against 60-150 here).
  for (int i = lane; i < nv * (TV + 1); i += WAVE) (&L.H[0][0])[i] = (&L.M[0][0])[i];       // (same layout: rows 0 .. nv-1 with their padding word)
  wave_sync();
  TPROF(3)
  // Round 5: the constraint part  sum_rows J' Hc J  on the matrix cores, as in the SO100 solver (so101_newton.hpp).  With the constraint rows as the K
  // dimension the sum is  J' T,  T = blockdiag(Hc) J:  v_mfma_f32_32x32x2_f32 takes two rows per instruction - lane l supplies column l % 32 of row
  // l / 32 of both factors - and the 32 x 32 accumulator tile(s) hold H's constraint part until the end; no LDS exchange, no barrier.  Before, every
  // active contact was compacted to the dofs it touches and multiplied through LDS (three exchanges per contact, ~1.5 us each on a wavefront that has
  // its SIMD to itself: 40 of the 115 us of a hand-over env's Newton solve, 98 of 300 on the Dining scenes), the scalar rows one after the other.
  {
    constexpr int NT = (TV + 31) / 32;                      // tiles per dimension: 1 (32-dof build) or 2
    const int col = lane & 31, kk = lane >> 5;
    const int js = G.js;
    mfma_acc16 acc[NT][NT];
#pragma unroll
    for (int ti = 0; ti < NT; ti++)
#pragma unroll
      for (int tj = 0; tj < NT; tj++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[ti][tj][r] = 0.f;
    // scalar rows (equality, dof friction, limits; Hc = the row's second derivative, 0 outside its quadratic zone): two per instruction
#pragma unroll 4
    for (int r0 = 0; r0 < nscalar; r0 += 2) {
      int r = r0 + kk;
      bool on = r < nscalar;
      float h = on ? G.ejv[r] : 0.f, jv[NT];
#pragma unroll
      for (int t = 0; t < NT; t++) jv[t] = on ? G.J[r * js + 32 * t + col] : 0.f;
#pragma unroll
      for (int ti = 0; ti < NT; ti++)
#pragma unroll
        for (int tj = 0; tj < NT; tj++) acc[ti][tj] = mfma_32x32x2(jv[ti], h * jv[tj], acc[ti][tj]);
    }
    TPROF(4)
    // contact blocks, the active ones; the 6 x 6 block and the Jacobian columns of the NEXT one (the next TWO on the 64-dof build, whose rows live in
    // the global scratch: one contact's arithmetic does not cover a round trip there) are fetched while this one is multiplied
    auto next_active = [&](int c) -> int { c++; while (c < ncon && L.hdim[c] == 0) c++; return c; };
    struct Fetched { float4 h[9]; float j[NT][6]; };
    auto fetch = [&](int cc, Fetched& F) {
      int dim = L.hdim[cc], row = L.con[cc].row;
      const float4* hp = (const float4*)G.Hc[cc];
#pragma unroll
      for (int q = 0; q < 9; q++) F.h[q] = hp[q];
#pragma unroll
      for (int t = 0; t < NT; t++)
#pragma unroll
        for (int j = 0; j < 6; j++) F.j[t][j] = j < dim ? G.J[(row + j) * js + 32 * t + col] : 0.f;
    };
    auto multiply = [&](int cc, const Fetched& F) {
      const int dim = L.hdim[cc];
      float hc[36], tv[NT][6];
#pragma unroll
      for (int q = 0; q < 9; q++) { hc[4 * q] = F.h[q].x; hc[4 * q + 1] = F.h[q].y; hc[4 * q + 2] = F.h[q].z; hc[4 * q + 3] = F.h[q].w; }
#pragma unroll
      for (int t = 0; t < NT; t++)
#pragma unroll
        for (int j = 0; j < 6; j++) {                          // T = Hc J, this lane's column(s)
          float v = 0.f;
#pragma unroll
          for (int l = 0; l < 6; l++) v += hc[j * 6 + l] * F.j[t][l];
          tv[t][j] = v;
        }
#pragma unroll
      for (int s2 = 0; s2 < 3; s2++) {
        if (2 * s2 < dim) {
          float a[NT], b[NT];
#pragma unroll
          for (int t = 0; t < NT; t++) { a[t] = kk ? F.j[t][2 * s2 + 1] : F.j[t][2 * s2]; b[t] = kk ? tv[t][2 * s2 + 1] : tv[t][2 * s2]; }
#pragma unroll
          for (int ti = 0; ti < NT; ti++)
#pragma unroll
            for (int tj = 0; tj < NT; tj++) acc[ti][tj] = mfma_32x32x2(a[ti], b[tj], acc[ti][tj]);
        }
      }
    };
    if constexpr (TRL == 0) {
      Fetched A, B, cur;
      int c0 = next_active(-1), c1 = c0 < ncon ? next_active(c0) : ncon;
      if (c0 < ncon) fetch(c0, A);
      if (c1 < ncon) fetch(c1, B);
      while (c0 < ncon) {
        cur = A;
        int c2 = c1 < ncon ? next_active(c1) : ncon;
        if (c2 < ncon) fetch(c2, A);
        multiply(c0, cur);
        if (c1 >= ncon) break;
        cur = B;
        int c3 = c2 < ncon ? next_active(c2) : ncon;
        if (c3 < ncon) fetch(c3, B);
        multiply(c1, cur);
        c0 = c2; c1 = c3;
      }
    } else {
      Fetched N, cur;
      int c = next_active(-1);
      if (c < ncon) fetch(c, N);
      while (c < ncon) {
        cur = N;
        int cn = next_active(c);
        if (cn < ncon) fetch(cn, N);
        multiply(c, cur);
        c = cn;
      }
    }
    // the accumulators into H: lane l holds column l % 32, rows 8 (r / 4) + 4 (l / 32) + r % 4 of each tile
#pragma unroll
    for (int ti = 0; ti < NT; ti++)
#pragma unroll
      for (int tj = 0; tj < NT; tj++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          int hr = 32 * ti + 8 * (r / 4) + 4 * kk + (r % 4), hcol = 32 * tj + col;
          if (hr < nv && hcol < nv) L.H[hr][hcol] += acc[ti][tj][r];
        }
  }
  wave_sync();
  TPROF(5)
  return cost;
}

DEV void solve_newton(const TreeModel* tm, TreeLDS& L, const TreeScratch& G, int max_iter, float tolerance) {
  int lane = wave_lane(), nv = tm->nv, nrow = L.nrow, nscalar = L.nscalar, ncon = L.ncon;
  if (lane == 0) L.iters = 0;
  if (nrow == 0) { if (lane < nv) L.qacc[lane] = L.qsm[lane]; wave_sync(); return; }
  // warm start: the better of the previous acceleration and the unconstrained one
  if (lane < nv) L.x[lane] = L.warm[lane];
  wave_sync();
  float cw = total_cost(tm, L, G, false);
  if (lane < nv) L.x[lane] = L.qsm[lane];
  wave_sync();
  float cs = total_cost(tm, L, G, false);
  if (cw < cs) { if (lane < nv) L.x[lane] = L.warm[lane]; }
  wave_sync();
  float scale = 1.f / (tm->meaninertia * (float)(nv > 1 ? nv : 1));
  float cost = total_cost(tm, L, G, true);
  int it = 0;
  TPROF_T0();
  for (; it < max_iter; ) {
    TPROF(15)
    chol_factor(L.H, nv);
    TPROF(6)
    if (lane < nv) L.search[lane] = -L.grad[lane];
    wave_sync();
    chol_solve(L.H, nv, L.search);
    TPROF(7)
    // line search on phi(alpha) = cost(x + alpha search): safeguarded Newton on phi'
    for (int r = lane; r < nrow; r += WAVE) {
      G.ejv[r] = row_dot(G, r, L.search, nv, 0.f);
    }
    float p1 = 0.f, p2 = 0.f;
    if (lane < nv) {
      float mv = 0.f;
      for (int c = 0; c < nv; c++) mv += L.M[lane][c] * L.search[c];
      p1 = L.search[lane] * (L.Ma[lane] - L.qfrc[lane]); p2 = L.search[lane] * mv;
    }
    float q1 = wave_sum_f(p1), q2 = wave_sum_f(p2);
    wave_sync();
    TPROF(8)
    float alpha = 0.f, lo = 0.f, hi = -1.f, d10 = 0.f;
    for (int ls = 0; ls < 24; ls++) {
      float a1 = 0.f, a2 = 0.f;
      for (int r = lane; r < nscalar; r += WAVE) {
        float f, h, jv = G.ejv[r];
        scalar_block((int)(G.etype[r] & 0xffu), G.eD[r], G.eR[r], G.efl[r], G.ejar[r] + alpha * jv, &f, &h);
        a1 -= f * jv; a2 += jv * h * jv;
      }
      for (int ci = lane; ci < ncon; ci += WAVE) {
        const TCon& C = L.con[ci];
        int dim = C.dim, row = C.row;
        float r6[6], D6[6], jv6[6], c1, c2;
#pragma unroll
        for (int j = 0; j < 6; j++) { jv6[j] = j < dim ? G.ejv[row + j] : 0.f; r6[j] = j < dim ? G.ejar[row + j] + alpha * jv6[j] : 0.f; D6[j] = j < dim ? G.eD[row + j] : 0.f; }
        contact_line(C, D6, dim, r6, jv6, &c1, &c2);
        a1 += c1; a2 += c2;
      }
      float d1 = q1 + q2 * alpha + wave_sum_f(a1), d2 = q2 + wave_sum_f(a2);
      if (ls == 0) { d10 = fabsf(d1); if (!(d1 < 0.f)) break; }
      else {
        if (fabsf(d1) <= 1e-5f * d10) break;
        if (d1 < 0.f) lo = alpha; else hi = alpha;
      }
      float cand = alpha - d1 / fmaxf(d2, 1e-30f);
      if (hi > 0.f && (cand <= lo || cand >= hi)) cand = 0.5f * (lo + hi);
      alpha = cand;
    }
    if (lane < nv) L.x[lane] += alpha * L.search[lane];
    wave_sync();
    TPROF(9)
    float newcost = total_cost(tm, L, G, true);
    float improvement = scale * (cost - newcost);
    float gn = wave_sum_f(lane < nv ? L.grad[lane] * L.grad[lane] : 0.f);
    float gnorm = scale * sqrtf(gn);
    cost = newcost;
    it++;
    if (improvement < tolerance || gnorm < tolerance) break;
  }
  if (lane < nv) L.qacc[lane] = L.x[lane];
  if (lane == 0) L.iters = it;
  wave_sync();
}

// ------------------------------------------------------------------ forward dynamics and integration
// `phases`: stage mask for timing runs (so101_tree_debug_forward with SO101_TREE_PHASES set); every caller on the step path passes all
// the two halves of forward() around the collision stage (launch chain: the contacts come from the records instead)
DEV void forward_smooth(const TreeModel* tm, TreeLDS& L) {
  TPROF_T0();
  kinematics(tm, L);
  TPROF(17)
  crba(tm, L);
  TPROF(18)
  rne_bias(tm, L);
  TPROF(19)
  smooth(tm, L);
  TPROF(20)
}
DEV void forward_constrained(const TreeModel* tm, TreeLDS& L, TreeScratch& G, int max_iter, float tolerance) {
  TPROF_T0();
  make_constraints(tm, L, G);
  TPROF(10)
  solve_newton(tm, L, G, max_iter, tolerance);
  TPROF(11)
}
DEV void forward(const TreeModel* tm, const DevModel* gm, TreeLDS& L, TreeScratch& G, int max_iter, float tolerance, int phases = 0x7f) {
  kinematics(tm, L);
  if (phases & 2) crba(tm, L);
  if (phases & 4) rne_bias(tm, L);
  if (phases & 8) smooth(tm, L);
  if (phases & 16) collision(tm, gm, L, !(phases & 128)); else { if (wave_lane() == 0) { L.ncon = 0; L.ncand = 0; } wave_sync(); }
  TPROF_T0();
  if (phases & 32) make_constraints(tm, L, G); else { if (wave_lane() == 0) { L.nrow = 0; L.nscalar = 0; } wave_sync(); }
  TPROF(10)
  if (phases & 64) solve_newton(tm, L, G, max_iter, tolerance);
  TPROF(11)
}

DEV void euler(const TreeModel* tm, TreeLDS& L) {
  int lane = wave_lane(), nv = tm->nv, nb = tm->nbody;
  float dt = tm->dt;
  if (lane < nv) L.warm[lane] = L.qacc[lane];
  if (tm->any_damping) {
    // joint damping implicit in the velocity (mj_Euler): qacc' = (M + h D)^-1 M qacc
    if (lane < nv) {
      float v = 0.f;
      for (int c = 0; c < nv; c++) { v += L.M[lane][c] * L.qacc[c]; L.H[lane][c] = L.M[lane][c] + (c == lane ? dt * tm->dof_damping[lane] : 0.f); }
      L.tmp[lane] = v;
    }
    wave_sync();
    chol_factor(L.H, nv);
    chol_solve(L.H, nv, L.tmp);
  } else {
    if (lane < nv) L.tmp[lane] = L.qacc[lane];
    wave_sync();
  }
  if (lane < nv) L.qvel[lane] += dt * L.tmp[lane];
  wave_sync();
  if (lane > 0 && lane < nb) {
    int b = lane, jt = tm->body_jnttype[b], qa = tm->body_qposadr[b], d = tm->body_dofadr[b];
    if (jt == TJ_HINGE || jt == TJ_SLIDE) L.qpos[qa] += dt * L.qvel[d];
    else if (jt == TJ_FREE) {
#pragma unroll
      for (int k = 0; k < 3; k++) L.qpos[qa + k] += dt * L.qvel[d + k];
      float w[3] = {L.qvel[d + 3], L.qvel[d + 4], L.qvel[d + 5]};
      float ang = dt * normalize3(w), sn, cs; sincos_f(0.5f * ang, &sn, &cs);
      float dq[4] = {cs, w[0] * sn, w[1] * sn, w[2] * sn}, q0[4] = {L.qpos[qa + 3], L.qpos[qa + 4], L.qpos[qa + 5], L.qpos[qa + 6]}, o[4];
      mulquat(o, q0, dq); normquat(o);
#pragma unroll
      for (int k = 0; k < 4; k++) L.qpos[qa + 3 + k] = o[k];
    }
  }
  wave_sync();
}


// ================================================================== env layer of the ALOHA hand-over tasks
// (so101_sim/tasks/base/aloha2_task.py:279-444 action / observables / reset, so101_sim/tasks/hand_over.py:36-56,246-349
// placement and the overlap reward; the SO100 counterpart is so101_env.hpp)
}  // namespace tree

#define T_RING_DEFAULT 5       // joints_pos / joints_vel delay of the reference's default: 0.1 s = 5 control steps (aloha2_task.py:102)
#define T_PS_DEFAULT 15        // delayed_physics_state: 0.3 s = 15 control steps (aloha2_task.py:103)
#define T_DELAY_MAX 64
struct TreeTask {
  int npos, nvel, obj_body, con_body, nbox, n_substeps, last_step, settle_max, terminate_on_success, n_envs, iterations;
  int jdelay, pdelay;                      // observation delays in control steps (so101_tree_config): joints_pos / joints_vel, delayed_physics_state
  // kind 1 = the Dining scene (tasks/base/dining.py:162-267): six free props dropped into six table regions, the three top and the three
  // bottom regions shuffled among (plate, bowl, container) and (mug, pen, banana), a uniform yaw each, no rejection, then settled
  int kind, prop_body[6];
  float region_lo[6][3], region_hi[6][3];
  int reward_mode, requires_handover;      // 0 overlap boxes (the default), 1 contact sequence (hand_over.py:286-338)
  float dist_threshold, tolerance, grip[6];    // gripper limits: sim_qpos open, close, sim_ctrl open, close, follower open, close
  int obs_qposadr[TU], obs_is_gripper[TU], act_is_gripper[TU];
  float box_pos[2][3], box_half[2][3], obj_bvh[6];
  float obj_lo[3], obj_hi[3], obj_yaw[2], con_lo[3], con_hi[3], home_qpos[TQ], home_ctrl[TU];
  unsigned long long seed, env_id_base;
};
struct TreeEnvBuffers {
  float *ring_pos, *ring_vel, *ep_return; int *step_count, *episode; unsigned char* need_reset; int* success_state;
  float *ps_ring, *ps_out, *ps_delayed;    // so101_tree_bind_physics_state: delay line [pdelay][nq + nv][N], physics_state / delayed_physics_state [N][nq + nv]; NULL = off
};
// Settled-state store (so101_tree_set_settled_store): the results of placement + settle for episodes first .. first + count - 1 of every
// env, computed once by so101_tree_compute_settled and kept by the caller.  The settled state of an episode is a pure function of (seed,
// global env id, episode, configuration), so a reset that finds its entry copies the same bits it would have computed.
struct TreeStore {
  const float *qpos, *qvel, *warm; const int* flags; int first, count;     // [count][nq|nv|nv][N], [count][N]
  // Reset prefetch (round 4; so101_tree_config.prefetch_resets): the settled state of every env's NEXT episode, computed by k_tree_prepare on a
  // low-priority stream beside the stepping kernels into a library-owned cache [nq|nv|nv][N]; ctag[e] = 1 + the episode the entry holds
  // (0: empty).  The producer writes the entry with agent-scope stores, drains them, then the tag; the consumer reads the tag, then the entry,
  // with agent-scope loads (the two kernels may run on different XCDs, whose L2s are not coherent) and settles in place when the tag does not
  // name its episode - the same bits either way, the settled state being a pure function of (seed, env id, episode, configuration).
  float *cq, *cv, *cw; int* cf; unsigned int* ctag;
};

namespace tree {

DEV float convert_gripper(float v, float from_open, float from_close, float to_open, float to_close) {
  return (v - from_close) / (from_open - from_close) * (to_open - to_close) + to_close;
}

// requires the kinematics of the current state in LDS
DEV float task_reward(const TreeModel* tm, const TreeTask& T, const TreeLDS& L) {
  int ob = T.obj_body, cb = T.con_body;
  const float* vo = &L.qvel[tm->body_dofadr[ob]]; const float* vc = &L.qvel[tm->body_dofadr[cb]];
  if (fmaxf(fabsf(vo[0]), fmaxf(fabsf(vo[1]), fabsf(vo[2]))) >= 1e-3f) return 0.f;        // any_props_moving: linear part only
  if (fmaxf(fabsf(vc[0]), fmaxf(fabsf(vc[1]), fabsf(vc[2]))) >= 1e-3f) return 0.f;
  BoxW o;
  mat2quat(o.quat, L.ximat[ob]);
  float ctr[3]; rotvecquat(ctr, T.obj_bvh, o.quat);
#pragma unroll
  for (int i = 0; i < 3; i++) { o.pos[i] = ctr[i] + L.xipos[ob][i]; o.half[i] = T.obj_bvh[3 + i]; }
  float cq[4] = {L.xquat[cb][0], L.xquat[cb][1], L.xquat[cb][2], L.xquat[cb][3]};
  for (int k = 0; k < T.nbox; k++) {
    BoxW cw;
    float r[3]; rotvecquat(r, T.box_pos[k], cq);
#pragma unroll
    for (int i = 0; i < 3; i++) { cw.pos[i] = L.xpos[cb][i] + r[i]; cw.half[i] = T.box_half[k][i]; }
    float ident[4] = {1.f, 0.f, 0.f, 0.f};
    mulquat(cw.quat, cq, ident);
    if (!overlap_oobb_oobb(o, cw)) return 0.f;
  }
  return 1.f;
}

// reward_based_on_overlap = False (hand_over.py:286-338): a three-state sequence over the contacts of the last physics step - the right
// gripper touched the object (0 -> 1), then the left one did (1 -> 2), then the object rests touching the container within the
// distance threshold (reward 1).  Without reward_requires_handover an episode starts in state 2.  Needs the kinematics of the
// current state and the contact list of the last substep in LDS.
DEV float task_reward_contacts(const TreeModel* tm, const TreeTask& T, const TreeLDS& L, const TreeScratch& G, int* state_io) {
  int lane = wave_lane(), ob = T.obj_body, cb = T.con_body;
  const float* vo = &L.qvel[tm->body_dofadr[ob]]; const float* vc = &L.qvel[tm->body_dofadr[cb]];
  bool moving = fmaxf(fabsf(vo[0]), fmaxf(fabsf(vo[1]), fabsf(vo[2]))) >= 1e-3f || fmaxf(fabsf(vc[0]), fmaxf(fabsf(vc[1]), fabsf(vc[2]))) >= 1e-3f;
  bool t81 = false, t41 = false, t12 = false;
  for (int c = lane; c < L.ncon; c += WAVE) {
    int c1 = tm->geom_class[L.con[c].g1], c2 = tm->geom_class[L.con[c].g2];
    t81 = t81 || ((c1 & 8) && (c2 & 1)) || ((c2 & 8) && (c1 & 1));
    t41 = t41 || ((c1 & 4) && (c2 & 1)) || ((c2 & 4) && (c1 & 1));
    t12 = t12 || ((c1 & 1) && (c2 & 2)) || ((c2 & 1) && (c1 & 2));
  }
  bool right_obj = wave_ballot(t81) != 0ull, left_obj = wave_ballot(t41) != 0ull, obj_con = wave_ballot(t12) != 0ull;
  int st = *state_io;
  float r = 0.f;
  if (st == 0) { if (right_obj) st = 1; }
  else if (st == 1) { if (left_obj) st = 2; }
  else {
    float dx = L.xpos[cb][0] - L.xpos[ob][0], dy = L.xpos[cb][1] - L.xpos[ob][1];
    bool inside = sqrtf(dx * dx + dy * dy) < T.dist_threshold;
    if (!moving && obj_con && inside) r = 1.f;
  }
  *state_io = st;
  return r;
}

// reward 'contact' of the Dining tasks (dining_place_in_container.py:126-154, "put the red mug on the plate"): 1 when a geom of the
// object touches a geom of the receptacle and neither prop moves (linear velocity, success_detector_utils.py:22-28).  dm_control
// evaluates it on physics.data.contact after physics.step(), whose legacy step ends with mj_step1: the contacts of the state AFTER the
// last substep - the caller runs kinematics + collision on the integrated state before calling this.
DEV float task_reward_touching(const TreeModel* tm, const TreeTask& T, const TreeLDS& L) {
  int lane = wave_lane(), ob = T.obj_body, cb = T.con_body;
  const float* vo = &L.qvel[tm->body_dofadr[ob]]; const float* vc = &L.qvel[tm->body_dofadr[cb]];
  bool moving = fmaxf(fabsf(vo[0]), fmaxf(fabsf(vo[1]), fabsf(vo[2]))) >= 1e-3f || fmaxf(fabsf(vc[0]), fmaxf(fabsf(vc[1]), fabsf(vc[2]))) >= 1e-3f;
  bool t12 = false;
  for (int c = lane; c < L.ncon; c += WAVE) {
    int c1 = tm->geom_class[L.con[c].g1], c2 = tm->geom_class[L.con[c].g2];
    t12 = t12 || ((c1 & 1) && (c2 & 2)) || ((c2 & 1) && (c1 & 2));
  }
  return (!moving && wave_ballot(t12) != 0ull) ? 1.f : 0.f;
}

// Dining placement (dining.py:162-228): region samples (draws 0-17: three uniforms per region, top left / middle / right, bottom left /
// middle / right), the two shuffles (draws 18, 19: one of the six orders of three each - the batched envs' counter RNG; a single env
// draws numpy's shuffle on the host instead), a yaw in [-pi, pi) per prop in placer order plate, bowl, container, mug, pen, banana
// (draws 20-25).  One lane writes the 42 numbers.
DEV void dining_place(const TreeModel* tm, const TreeTask& T, TreeLDS& L, unsigned long long env_id, unsigned int episode) {
  if (wave_lane() == 0) {
    float smp[6][3];
    for (int r = 0; r < 6; r++)
      for (int k = 0; k < 3; k++) smp[r][k] = T.region_lo[r][k] + rng_uniform(T.seed, env_id, episode, 3 * r + k) * (T.region_hi[r][k] - T.region_lo[r][k]);
    const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    int pt = (int)(rng_uniform(T.seed, env_id, episode, 18) * 6.f), pb = (int)(rng_uniform(T.seed, env_id, episode, 19) * 6.f);
    pt = pt > 5 ? 5 : pt; pb = pb > 5 ? 5 : pb;
    for (int p = 0; p < 6; p++) {
      int region = p < 3 ? perm[pt][p] : 3 + perm[pb][p - 3];
      int qa = tm->body_qposadr[T.prop_body[p]];
      float yaw = T.obj_yaw[0] + rng_uniform(T.seed, env_id, episode, 20 + p) * (T.obj_yaw[1] - T.obj_yaw[0]);
      float sn, cs; sincos_f(0.5f * yaw, &sn, &cs);
      for (int k = 0; k < 3; k++) L.qpos[qa + k] = smp[region][k];
      L.qpos[qa + 3] = cs; L.qpos[qa + 4] = 0.f; L.qpos[qa + 5] = 0.f; L.qpos[qa + 6] = sn;
    }
  }
  wave_sync();
}

// env.reset(): arms at the home pose, object and container placed (container by rejection, <= 20 tries), settled with the arms held
// (aloha2_task.py:369-383, hand_over.py:208-236,340-346).  Same counter-RNG draws as the SO100 reset and the oracle.
DEV void env_settle(const TreeModel* tm, const DevModel* gm, const TreeTask& T, TreeLDS& L, TreeScratch& G, int e, unsigned int episode) {
  int lane = wave_lane();
  unsigned long long env_id = T.env_id_base + (unsigned long long)e;
  if (lane < tm->nq) L.qpos[lane] = lane < tm->njnt ? T.home_qpos[lane] : 0.f;
  if (lane < tm->nv) { L.qvel[lane] = 0.f; L.warm[lane] = 0.f; L.qacc[lane] = 0.f; }
  if (lane < tm->nu) L.ctrl[lane] = T.home_ctrl[lane];
  wave_sync();
  int qo = tm->body_qposadr[T.obj_body], qc = tm->body_qposadr[T.con_body];
  if (T.kind == 1) dining_place(tm, T, L, env_id, episode);
  if (T.kind == 0 && lane == 0) {
    for (int k = 0; k < 3; k++) L.qpos[qo + k] = T.obj_lo[k] + rng_uniform(T.seed, env_id, episode, k) * (T.obj_hi[k] - T.obj_lo[k]);
    float yaw = T.obj_yaw[0] + rng_uniform(T.seed, env_id, episode, 3) * (T.obj_yaw[1] - T.obj_yaw[0]);
    float sn, cs; sincos_f(0.5f * yaw, &sn, &cs);
    L.qpos[qo + 3] = cs; L.qpos[qo + 4] = 0.f; L.qpos[qo + 5] = 0.f; L.qpos[qo + 6] = sn;
    L.qpos[qc + 3] = 1.f; L.qpos[qc + 4] = 0.f; L.qpos[qc + 5] = 0.f; L.qpos[qc + 6] = 0.f;
  }
  wave_sync();
  bool placed = T.kind == 1;                                   // (Dining: ignore_collisions = True, dining.py:246)
  for (int attempt = 0; attempt < 20 && !placed; attempt++) {
    if (lane < 3) L.qpos[qc + lane] = T.con_lo[lane] + rng_uniform(T.seed, env_id, episode, 4 + 3 * attempt + lane) * (T.con_hi[lane] - T.con_lo[lane]);
    wave_sync();
    kinematics(tm, L);
    collision(tm, gm, L);
    bool hit = false;
    for (int k = 0; k < L.ncon; k++) if (L.con[k].b1 == T.con_body || L.con[k].b2 == T.con_body) hit = true;
    wave_sync();
    placed = !hit;
  }
  if (!placed && lane == 0) L.flags |= 16;
  float q0 = lane < tm->njnt ? L.qpos[lane] : 0.f;
  // settle until |qvel| < 1e-3 and |qacc| < 1e-2 over the props' dofs
  bool settled = T.settle_max == 0;
  for (int k = 0; k < T.settle_max && !settled; k++) {
    forward(tm, gm, L, G, T.iterations, T.tolerance);
    euler(tm, L);
    bool ok = true;
    if (lane < tm->nq) ok = ok && fabsf(L.qpos[lane]) <= 1e10f;
    if (lane < tm->nv) ok = ok && fabsf(L.qvel[lane]) <= 1e10f && fabsf(L.qacc[lane]) <= 1e10f;
    if (wave_ballot(!ok) != 0ull) { if (lane == 0) L.flags |= 8; break; }
    if (lane < tm->njnt) { L.qpos[lane] = q0; L.qvel[lane] = 0.f; }     // (one-dof joints come first in qpos / qvel; dm_control holds them)
    wave_sync();
    float mv = 0.f, ma = 0.f;
    if (lane >= tm->njnt && lane < tm->nv) { mv = fabsf(L.qvel[lane]); ma = fabsf(L.qacc[lane]); }
    mv = wave_max_f(mv); ma = wave_max_f(ma);
    settled = mv < 1e-3f && ma < 1e-2f;
  }
  if (!settled && lane == 0) L.flags |= 32;
  wave_sync();
}

// the episode starts: delay lines padded with the reset-time value (task_suite.py:154 INITIAL_VALUE); physics_state and its delayed
// copy (aloha2_task.py:244-251,441-444: qpos | qvel, delayed by image_observation_delay_secs) both report the reset state
DEV void fill_delay_lines(const TreeModel* tm, const TreeTask& T, const TreeLDS& L, const TreeEnvBuffers& E, int e) {
  int lane = wave_lane(), N = T.n_envs;
  if (lane < T.npos) {
    float v = L.qpos[T.obs_qposadr[lane]];
    if (T.obs_is_gripper[lane]) v = convert_gripper(v, T.grip[0], T.grip[1], T.grip[4], T.grip[5]);
    for (int r = 0; r < T.jdelay; r++) E.ring_pos[((size_t)r * T.npos + lane) * N + e] = v;
  }
  if (lane < T.nvel) for (int r = 0; r < T.jdelay; r++) E.ring_vel[((size_t)r * T.nvel + lane) * N + e] = L.qvel[lane];
  if (E.ps_out) {
    int D = tm->nq + tm->nv;
    for (int i = lane; i < D; i += WAVE) {
      float v = i < tm->nq ? L.qpos[i] : L.qvel[i - tm->nq];
      for (int r = 0; r < T.pdelay; r++) E.ps_ring[((size_t)r * D + i) * N + e] = v;
      E.ps_out[(size_t)e * D + i] = v;
      E.ps_delayed[(size_t)e * D + i] = v;
    }
  }
}

// physics_state / delayed_physics_state of control step sc (>= 1) from the integrated state in LDS
DEV void write_physics_state(const TreeModel* tm, const TreeTask& T, const TreeLDS& L, const TreeEnvBuffers& E, int e, int sc) {
  if (!E.ps_out) return;
  int lane = wave_lane(), N = T.n_envs, D = tm->nq + tm->nv;
  int slot = T.pdelay > 0 ? (sc - 1) % T.pdelay : 0;
  for (int i = lane; i < D; i += WAVE) {
    float v = i < tm->nq ? L.qpos[i] : L.qvel[i - tm->nq], delayed = v;
    if (T.pdelay > 0) { size_t ri = ((size_t)slot * D + i) * N + e; delayed = E.ps_ring[ri]; E.ps_ring[ri] = v; }
    E.ps_out[(size_t)e * D + i] = v;
    E.ps_delayed[(size_t)e * D + i] = delayed;
  }
}

// env.reset(): the settled state of the episode from the store when it holds it, computed otherwise; then the episode starts -
// delay lines padded with the reset-time value (task_suite.py:154 INITIAL_VALUE), counters cleared
DEV void env_reset(const TreeModel* tm, const DevModel* gm, const TreeTask& T, TreeLDS& L, TreeScratch& G, const TreeBuffers& B, const TreeEnvBuffers& E,
                   const TreeStore& S, int e) {
  int lane = wave_lane(), N = T.n_envs;
  unsigned int episode = (unsigned int)E.episode[e];
  if (S.qpos && episode - (unsigned int)S.first < (unsigned int)S.count) {
    size_t k = episode - (unsigned int)S.first;
    if (lane < tm->nq) L.qpos[lane] = S.qpos[(k * tm->nq + lane) * N + e];
    if (lane < tm->nv) { L.qvel[lane] = S.qvel[(k * tm->nv + lane) * N + e]; L.warm[lane] = S.warm[(k * tm->nv + lane) * N + e]; L.qacc[lane] = 0.f; }
    if (lane < tm->nu) L.ctrl[lane] = T.home_ctrl[lane];
    if (lane == 0) { L.ncon = 0; L.nrow = 0; L.iters = 0; L.ncand = 0; L.flags |= S.flags[k * N + e]; }
    wave_sync();
  } else if (S.ctag && ld_agent(&S.ctag[e]) == episode + 1u) {
    if (lane < tm->nq) L.qpos[lane] = ld_agent(&S.cq[(size_t)lane * N + e]);
    if (lane < tm->nv) { L.qvel[lane] = ld_agent(&S.cv[(size_t)lane * N + e]); L.warm[lane] = ld_agent(&S.cw[(size_t)lane * N + e]); L.qacc[lane] = 0.f; }
    if (lane < tm->nu) L.ctrl[lane] = T.home_ctrl[lane];
    if (lane == 0) { L.ncon = 0; L.nrow = 0; L.iters = 0; L.ncand = 0; L.flags |= ld_agent(&S.cf[e]); }
    wave_sync();
  } else {
    env_settle(tm, gm, T, L, G, e, episode);
  }
  fill_delay_lines(tm, T, L, E, e);
  // (the episode counter last, agent scope: k_tree_prepare reads it to know which episode to prepare next, and must not see the new value
  //  before this env has finished reading its cache entry)
  wave_sync();
  if (lane == 0) { E.step_count[e] = 0; E.ep_return[e] = 0.f; E.success_state[e] = T.requires_handover ? 0 : 2; st_agent(&E.episode[e], (int)(episode + 1u)); }
}

// observation row: joints_pos (delayed) | joints_vel (delayed) | undelayed_joints_pos | undelayed_joints_vel | commanded_joints_pos
DEV int obs_dim(const TreeTask& T) { return 3 * T.npos + 2 * T.nvel; }

DEV void write_obs(const TreeTask& T, const TreeLDS& L, const TreeEnvBuffers& E, int e, int sc, bool first, float* obs) {
  int lane = wave_lane(), N = T.n_envs, D = obs_dim(T);
  float* o = obs + (size_t)e * D;
  int slot = (first || T.jdelay == 0) ? 0 : (sc - 1) % T.jdelay;
  if (lane < T.npos) {
    float v = L.qpos[T.obs_qposadr[lane]], c = L.ctrl[lane];
    if (T.obs_is_gripper[lane]) { v = convert_gripper(v, T.grip[0], T.grip[1], T.grip[4], T.grip[5]); c = convert_gripper(c, T.grip[2], T.grip[3], T.grip[4], T.grip[5]); }
    size_t ri = ((size_t)slot * T.npos + lane) * N + e;
    float delayed = T.jdelay > 0 ? E.ring_pos[ri] : v;
    if (!first && T.jdelay > 0) E.ring_pos[ri] = v;
    o[lane] = delayed; o[T.npos + T.nvel + lane] = v; o[2 * T.npos + 2 * T.nvel + lane] = c;
  }
  if (lane < T.nvel) {
    float v = L.qvel[lane];
    size_t ri = ((size_t)slot * T.nvel + lane) * N + e;
    float delayed = T.jdelay > 0 ? E.ring_vel[ri] : v;
    if (!first && T.jdelay > 0) E.ring_vel[ri] = v;
    o[T.npos + lane] = delayed; o[2 * T.npos + T.nvel + lane] = v;
  }
}

}  // namespace tree

}  // namespace TREE_NS
