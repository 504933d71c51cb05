// ORACLE — TEST INFRASTRUCTURE ONLY (see so101_oracle.hpp for scope, citations and pinning status).
// fp64, one env, deliberately plain loops: every stage is a literal restatement, not an optimised
// implementation.  The HIP product path under so101_sim_amd/csrc shares no code with this file.
#include "so101_oracle.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

typedef double real;
const real MINVAL = 1e-15;   // mjMINVAL
const real MINIMP = 1e-4, MAXIMP = 0.9999;
enum { G_PLANE = 0, G_SPHERE = 1, G_CAPSULE = 2, G_CYLINDER = 3, G_BOX = 4, G_MESH = 5 };
enum { J_NONE = 0, J_HINGE = 1, J_FREE = 2, J_SLIDE = 3 };
enum { C_FRICTION = 0, C_LIMIT = 1, C_CONTACT = 2, C_EQUALITY = 3 };

// ---------------------------------------------------------------- small vector helpers
inline real dot3(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void cross3(real* o, const real* a, const real* b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
inline real norm3(const real* a) { return std::sqrt(dot3(a, a)); }
inline real normalize3(real* a) {
  real n = norm3(a);
  if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return 0; }
  a[0] /= n; a[1] /= n; a[2] /= n; return n;
}
inline void mulmatvec3(real* o, const real* m, const real* v) {   // row-major 3x3
  real x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  real y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  real z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
inline void mulmatTvec3(real* o, const real* m, const real* v) {
  real x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  real y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  real z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
inline void mulmat3(real* o, const real* a, const real* b) {
  real t[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
    t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  std::memcpy(o, t, sizeof t);
}
inline void quat2mat(real* m, const real* q) {
  real w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = 1 - 2 * (y * y + z * z); m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = 1 - 2 * (x * x + z * z); m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = 1 - 2 * (x * x + y * y);
}
inline void mulquat(real* o, const real* a, const real* b) {
  real t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
               a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
               a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
               a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
  std::memcpy(o, t, sizeof t);
}
inline void normquat(real* q) {
  real n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  for (int i = 0; i < 4; i++) q[i] /= n;
}
inline void rotvecquat(real* o, const real* v, const real* q) {   // mju_rotVecQuat
  real m[9]; quat2mat(m, q); mulmatvec3(o, m, v);
}
// mju_mat2Quat restated (branch on the largest diagonal combination)
inline void mat2quat(real* q, const real* m) {
  real t = m[0] + m[4] + m[8];
  if (t > 0) {
    real s = std::sqrt(t + 1) * 2; q[0] = 0.25 * s; q[1] = (m[7] - m[5]) / s; q[2] = (m[2] - m[6]) / s; q[3] = (m[3] - m[1]) / s;
  } else if (m[0] > m[4] && m[0] > m[8]) {
    real s = std::sqrt(1 + m[0] - m[4] - m[8]) * 2; q[0] = (m[7] - m[5]) / s; q[1] = 0.25 * s; q[2] = (m[1] + m[3]) / s; q[3] = (m[2] + m[6]) / s;
  } else if (m[4] > m[8]) {
    real s = std::sqrt(1 + m[4] - m[0] - m[8]) * 2; q[0] = (m[2] - m[6]) / s; q[1] = (m[1] + m[3]) / s; q[2] = 0.25 * s; q[3] = (m[5] + m[7]) / s;
  } else {
    real s = std::sqrt(1 + m[8] - m[0] - m[4]) * 2; q[0] = (m[3] - m[1]) / s; q[1] = (m[2] + m[6]) / s; q[2] = (m[5] + m[7]) / s; q[3] = 0.25 * s;
  }
  normquat(q);
}

// ---------------------------------------------------------------- blob reader
struct Blob {
  std::map<std::string, std::pair<const void*, uint32_t>> ent;
  std::map<std::string, int> kind;
  bool parse(const void* p, size_t bytes) {
    const uint8_t* b = (const uint8_t*)p;
    uint32_t magic, ver, rb, n;
    std::memcpy(&magic, b, 4); std::memcpy(&ver, b + 4, 4); std::memcpy(&rb, b + 8, 4); std::memcpy(&n, b + 12, 4);
    if (magic != 0x424D3153u || ver != 3 || rb != 8) return false;
    for (uint32_t k = 0; k < n; k++) {
      const uint8_t* e = b + 16 + 48 * k;
      char name[33]; std::memcpy(name, e, 32); name[32] = 0;
      uint32_t kd, cnt; uint64_t off;
      std::memcpy(&kd, e + 32, 4); std::memcpy(&cnt, e + 36, 4); std::memcpy(&off, e + 40, 8);
      if (off + (size_t)cnt * (kd ? 8 : 4) > bytes) return false;
      ent[name] = {b + off, cnt}; kind[name] = kd;
    }
    return true;
  }
  std::vector<int> I(const char* n) const {
    auto it = ent.find(n); std::vector<int> v;
    if (it == ent.end()) { std::fprintf(stderr, "oracle: blob entry %s missing\n", n); return v; }
    v.resize(it->second.second); std::memcpy(v.data(), it->second.first, 4 * v.size()); return v;
  }
  std::vector<real> R(const char* n) const {
    auto it = ent.find(n); std::vector<real> v;
    if (it == ent.end()) { std::fprintf(stderr, "oracle: blob entry %s missing\n", n); return v; }
    v.resize(it->second.second); std::memcpy(v.data(), it->second.first, 8 * v.size()); return v;
  }
  bool has(const char* n) const { return ent.find(n) != ent.end(); }
  int i(const char* n) const { return I(n)[0]; }
  real r(const char* n) const { return R(n)[0]; }
};

struct Model {
  int nq, nv, nu, nbody, ngeom, nvert, npair, narm, nfree;
  real dt, gravity[3], impratio, tolerance, mpr_tol, meaninertia;
  int iterations, mpr_iter, elliptic;
  std::vector<int> body_parent, body_jnttype, body_qposadr, body_dofadr, body_weldid, arm_body, free_body,
      jnt_limited, dof_body, act_dof, act_ctrllimited, act_forcelimited, geom_type, geom_body, geom_condim,
      geom_priority, geom_vertadr, geom_vertnum, pair_geom;
  std::vector<real> body_pos, body_quat, body_ipos, body_iquat, body_mass, body_inertia, body_invweight0,
      body_bvh_aabb, jnt_axis, jnt_range, jnt_solref, jnt_solimp, dof_solref, dof_solimp, dof_armature,
      dof_frictionloss, dof_damping, dof_invweight0, act_gain, act_bias, act_ctrlrange, act_forcerange,
      geom_pos, geom_quat, geom_size, geom_friction, geom_solref, geom_solimp, geom_solmix, geom_margin,
      geom_gap, geom_rbound, geom_center, geom_aabb, mesh_vert;
  std::vector<int> body_hinge;   // body -> index of its one-dof joint (hinge / slide) or -1
  // general-tree scenes (ALOHA; absent from the SO100 blobs): joint-level clamp of the actuator force
  // (<joint actuatorfrcrange>), joint equalities q1 - q1_0 = poly(q2 - q2_0) (aloha_pbr.xml:296-299; qpos0 = 0)
  int neq = 0; bool any_damping = false;
  std::vector<int> jnt_actfrclimited, eq_dof, eq_qposadr;
  std::vector<real> jnt_actfrcrange, eq_polycoef, eq_solref, eq_solimp, home_qpos;
  // task
  int obj_body, con_body, nbox;
  std::vector<real> box_pos, box_half, obj_lo, obj_hi, obj_yaw, con_lo, con_hi, home_ctrl;
  int task_kind = 0;                                // 1: the Dining scene (tasks/base/dining.py:162-267)
  std::vector<int> prop_bodies;                     // [6] in placer order plate, bowl, container, mug, pen, banana
  std::vector<real> region_lo, region_hi;           // [6][3] top left / middle / right, bottom left / middle / right
};

struct Contact {
  real pos[3], frame[9], dist;
  int g1, g2, dim;
  real friction[5], solref[2], solimp[5], mu;
};

struct Support { real p[3]; };

}  // namespace

struct orc_sim {
  Model m;
  // state
  std::vector<real> qpos, qvel, ctrl, warm;
  // derived
  std::vector<real> xpos, xquat, xmat, xipos, ximat, gpos, gmat, S, M, Minv, bias, qfrc_act, act_force,
      qacc_smooth, qacc;
  std::vector<Contact> con;
  // constraints
  int nefc = 0, solver_iter = 0;
  std::vector<real> J, efc_pos, efc_D, efc_R, efc_aref, efc_force, efc_floss, efc_b, AR;
  std::vector<int> efc_type, efc_id, efc_dim;
  int iterations; real tolerance; bool collide = true;
  int contact_capacity = 0;      // > 0: the kernels' capacity rule (csrc/so101_device.hpp reduce_contacts / gather_contacts): a substep with more contacts keeps ONE per touching geom
                                 // pair - the first of its patch -, and the list is cut only when the touching pairs alone exceed it.  0 (default): no limit, like MuJoCo
  bool contacts_reduced = false, contacts_cut = false;
  bool hull_multi = true;        // several contacts for hull pairs resting on flat features (hull_patch; off: the single EPA contact of rounds 1-4)
  std::vector<real> mass0, inertia0, invweight0;     // unscaled prop masses (orc_set_mass_scale)
  std::vector<Contact> injected;                      // orc_inject_contacts
  int narrow = 1;              // 1 (default, what the kernels run) = MPR portal expanded by EPA to the nearest face of the Minkowski difference (minimum translation: mujoco >= 3.3's native GJK / EPA), 0 = MPR's own depth (the -DSO101_MPR option of the kernels)
  int epa_iters = 0;           // (statistics: polytope expansions of the last forward)
  int solver = 1;              // 1 = Newton (mujoco default; the reference scene sets no solver), 0 = PGS (north_star)
  int ls_evals = 0;
  // env layer
  orc_env_cfg cfg{};
  int step_count = 0; uint64_t episode = 0; bool need_reset = true; real ep_return = 0;
  real ring[5][6]; int ring_head = 0;
  real delayed[6];
  real cmd[6];
};

namespace {

// ================================================================ kinematics
void kinematics(orc_sim* s) {
  const Model& m = s->m;
  int nb = m.nbody;
  s->xpos.assign(3 * nb, 0); s->xquat.assign(4 * nb, 0); s->xmat.assign(9 * nb, 0);
  s->xipos.assign(3 * nb, 0); s->ximat.assign(9 * nb, 0);
  s->xquat[0] = 1; s->xmat[0] = s->xmat[4] = s->xmat[8] = 1; s->ximat[0] = s->ximat[4] = s->ximat[8] = 1;
  for (int b = 1; b < nb; b++) {
    int p = m.body_parent[b];
    real* xp = &s->xpos[3 * b]; real* xq = &s->xquat[4 * b];
    if (m.body_jnttype[b] == J_FREE) {
      const real* q = &s->qpos[m.body_qposadr[b]];
      xp[0] = q[0]; xp[1] = q[1]; xp[2] = q[2];
      xq[0] = q[3]; xq[1] = q[4]; xq[2] = q[5]; xq[3] = q[6];
      normquat(xq);
    } else {
      real t[3]; mulmatvec3(t, &s->xmat[9 * p], &m.body_pos[3 * b]);
      for (int k = 0; k < 3; k++) xp[k] = s->xpos[3 * p + k] + t[k];
      mulquat(xq, &s->xquat[4 * p], &m.body_quat[4 * b]);
      if (m.body_jnttype[b] == J_HINGE) {
        int h = m.body_hinge[b];
        real ang = s->qpos[m.body_qposadr[b]];
        const real* ax = &m.jnt_axis[3 * h];
        real sn = std::sin(0.5 * ang), jq[4] = {std::cos(0.5 * ang), ax[0] * sn, ax[1] * sn, ax[2] * sn};
        mulquat(xq, xq, jq);
      }
      normquat(xq);
      if (m.body_jnttype[b] == J_SLIDE) {            // translation along the joint axis (body frame), qpos0 = 0
        real a[3]; rotvecquat(a, &m.jnt_axis[3 * m.body_hinge[b]], xq);
        real q = s->qpos[m.body_qposadr[b]];
        for (int k = 0; k < 3; k++) xp[k] += a[k] * q;
      }
    }
    quat2mat(&s->xmat[9 * b], xq);
    real t[3]; mulmatvec3(t, &s->xmat[9 * b], &m.body_ipos[3 * b]);
    for (int k = 0; k < 3; k++) s->xipos[3 * b + k] = xp[k] + t[k];
    real im[9]; quat2mat(im, &m.body_iquat[4 * b]);
    mulmat3(&s->ximat[9 * b], &s->xmat[9 * b], im);
  }
  s->gpos.assign(3 * m.ngeom, 0); s->gmat.assign(9 * m.ngeom, 0);
  for (int g = 0; g < m.ngeom; g++) {
    int b = m.geom_body[g];
    real t[3]; mulmatvec3(t, &s->xmat[9 * b], &m.geom_pos[3 * g]);
    for (int k = 0; k < 3; k++) s->gpos[3 * g + k] = s->xpos[3 * b + k] + t[k];
    real gm[9]; quat2mat(gm, &m.geom_quat[4 * g]);
    mulmat3(&s->gmat[9 * g], &s->xmat[9 * b], gm);
  }
  // motion axes S[d] = (omega ; velocity of the point at the world origin)
  s->S.assign(6 * m.nv, 0);
  for (int b = 1; b < nb; b++) {
    int d = m.body_dofadr[b];
    const real* xp = &s->xpos[3 * b];
    if (m.body_jnttype[b] == J_HINGE) {
      real a[3]; mulmatvec3(a, &s->xmat[9 * b], &m.jnt_axis[3 * m.body_hinge[b]]);
      real* r = &s->S[6 * d];
      r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; cross3(r + 3, xp, a);
    } else if (m.body_jnttype[b] == J_SLIDE) {
      real a[3]; mulmatvec3(a, &s->xmat[9 * b], &m.jnt_axis[3 * m.body_hinge[b]]);
      real* r = &s->S[6 * d];
      r[3] = a[0]; r[4] = a[1]; r[5] = a[2];
    } else if (m.body_jnttype[b] == J_FREE) {
      for (int k = 0; k < 3; k++) {
        s->S[6 * (d + k) + 3 + k] = 1;
        real* r = &s->S[6 * (d + 3 + k)];
        real a[3] = {s->xmat[9 * b + k], s->xmat[9 * b + 3 + k], s->xmat[9 * b + 6 + k]};
        r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; cross3(r + 3, xp, a);
      }
    }
  }
}

bool is_ancestor(const Model& m, int a, int d) {   // a == d or a is an ancestor of d
  while (d != 0) { if (d == a) return true; d = m.body_parent[d]; }
  return a == 0;
}

// ================================================================ CRBA: composite inertia about the world origin
void crba(orc_sim* s) {
  const Model& m = s->m;
  int nb = m.nbody, nv = m.nv;
  std::vector<real> Ic(36 * nb, 0);
  for (int b = 1; b < nb; b++) {
    real mass = m.body_mass[b];
    if (mass <= 0) continue;
    const real* c = &s->xipos[3 * b];
    const real* R = &s->ximat[9 * b];
    real I[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      real v = 0;
      for (int k = 0; k < 3; k++) v += R[3 * i + k] * m.body_inertia[3 * b + k] * R[3 * j + k];
      I[3 * i + j] = v;
    }
    real cx[9] = {0, -c[2], c[1], c[2], 0, -c[0], -c[1], c[0], 0};
    real cx2[9]; mulmat3(cx2, cx, cx);
    real* o = &Ic[36 * b];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      o[6 * i + j] = I[3 * i + j] - mass * cx2[3 * i + j];
      o[6 * i + 3 + j] = mass * cx[3 * i + j];
      o[6 * (3 + i) + j] = -mass * cx[3 * i + j];
      o[6 * (3 + i) + 3 + j] = (i == j) ? mass : 0;
    }
  }
  for (int b = nb - 1; b >= 1; b--) {
    int p = m.body_parent[b];
    for (int k = 0; k < 36; k++) Ic[36 * p + k] += Ic[36 * b + k];
  }
  s->M.assign(nv * nv, 0);
  for (int r = 0; r < nv; r++) for (int c = 0; c < nv; c++) {
    int br = m.dof_body[r], bc = m.dof_body[c], bb;
    if (is_ancestor(m, bc, br)) bb = br; else if (is_ancestor(m, br, bc)) bb = bc; else continue;
    real t[6];
    for (int i = 0; i < 6; i++) { t[i] = 0; for (int j = 0; j < 6; j++) t[i] += Ic[36 * bb + 6 * i + j] * s->S[6 * c + j]; }
    real v = 0; for (int i = 0; i < 6; i++) v += s->S[6 * r + i] * t[i];
    s->M[r * nv + c] = v;
  }
  for (int d = 0; d < nv; d++) s->M[d * nv + d] += m.dof_armature[d];   // armature on the diagonal
  // dense inverse through Cholesky (M is SPD); the oracle favours clarity over sparsity
  std::vector<real> L(s->M);
  for (int j = 0; j < nv; j++) {
    for (int k = 0; k < j; k++) for (int i = j; i < nv; i++) L[i * nv + j] -= L[i * nv + k] * L[j * nv + k];
    real d = std::sqrt(L[j * nv + j]);
    for (int i = j; i < nv; i++) L[i * nv + j] /= d;
  }
  s->Minv.assign(nv * nv, 0);
  for (int c = 0; c < nv; c++) {
    std::vector<real> y(nv, 0);
    for (int i = 0; i < nv; i++) {
      real v = (i == c) ? 1 : 0;
      for (int k = 0; k < i; k++) v -= L[i * nv + k] * y[k];
      y[i] = v / L[i * nv + i];
    }
    for (int i = nv - 1; i >= 0; i--) {
      real v = y[i];
      for (int k = i + 1; k < nv; k++) v -= L[k * nv + i] * s->Minv[k * nv + c];
      s->Minv[i * nv + c] = v / L[i * nv + i];
    }
  }
}

// ================================================================ RNE bias (Coriolis/centrifugal + gravity), qacc = 0
void rne_bias(orc_sim* s) {
  const Model& m = s->m;
  int nb = m.nbody;
  std::vector<real> w(3 * nb, 0), al(3 * nb, 0), ao(3 * nb, 0), f(3 * nb, 0), n(3 * nb, 0);
  for (int k = 0; k < 3; k++) ao[k] = -m.gravity[k];   // gravity as base acceleration
  for (int b = 1; b < nb; b++) {
    int p = m.body_parent[b];
    real* wb = &w[3 * b]; real* ab = &al[3 * b]; real* aob = &ao[3 * b];
    if (m.body_jnttype[b] == J_FREE) {
      const real* v = &s->qvel[m.body_dofadr[b]];
      mulmatvec3(wb, &s->xmat[9 * b], v + 3);         // body-frame angular velocity -> world
      for (int k = 0; k < 3; k++) { ab[k] = 0; aob[k] = -m.gravity[k]; }
    } else {
      real d[3]; for (int k = 0; k < 3; k++) d[k] = s->xpos[3 * b + k] - s->xpos[3 * p + k];
      real t1[3], t2[3];
      cross3(t1, &al[3 * p], d); cross3(t2, &w[3 * p], d); cross3(t2, &w[3 * p], t2);
      for (int k = 0; k < 3; k++) { aob[k] = ao[3 * p + k] + t1[k] + t2[k]; wb[k] = w[3 * p + k]; ab[k] = al[3 * p + k]; }
      if (m.body_jnttype[b] == J_HINGE) {
        real a[3]; mulmatvec3(a, &s->xmat[9 * b], &m.jnt_axis[3 * m.body_hinge[b]]);
        real qd = s->qvel[m.body_dofadr[b]];
        real t[3]; cross3(t, &w[3 * p], a);
        for (int k = 0; k < 3; k++) { wb[k] += a[k] * qd; ab[k] += t[k] * qd; }
      } else if (m.body_jnttype[b] == J_SLIDE) {      // origin moving along a in the rotating parent frame: Coriolis 2 w x (a qd)
        real a[3]; mulmatvec3(a, &s->xmat[9 * b], &m.jnt_axis[3 * m.body_hinge[b]]);
        real qd = s->qvel[m.body_dofadr[b]];
        real t[3]; cross3(t, &w[3 * p], a);
        for (int k = 0; k < 3; k++) aob[k] += 2 * t[k] * qd;
      }
    }
    real mass = m.body_mass[b];
    if (mass > 0) {
      real r[3]; for (int k = 0; k < 3; k++) r[k] = s->xipos[3 * b + k] - s->xpos[3 * b + k];
      real t1[3], t2[3]; cross3(t1, ab, r); cross3(t2, wb, r); cross3(t2, wb, t2);
      real F[3]; for (int k = 0; k < 3; k++) F[k] = mass * (aob[k] + t1[k] + t2[k]);
      const real* R = &s->ximat[9 * b];
      real Iw[3], Ia[3], lw[3], la[3];
      mulmatTvec3(lw, R, wb); mulmatTvec3(la, R, ab);
      for (int k = 0; k < 3; k++) { lw[k] *= m.body_inertia[3 * b + k]; la[k] *= m.body_inertia[3 * b + k]; }
      mulmatvec3(Iw, R, lw); mulmatvec3(Ia, R, la);
      real N[3]; cross3(N, wb, Iw);
      real rF[3]; cross3(rF, r, F);
      for (int k = 0; k < 3; k++) { f[3 * b + k] += F[k]; n[3 * b + k] += Ia[k] + N[k] + rF[k]; }
    }
  }
  s->bias.assign(m.nv, 0);
  for (int b = nb - 1; b >= 1; b--) {
    int p = m.body_parent[b], d = m.body_dofadr[b];
    if (m.body_jnttype[b] == J_HINGE) {
      real a[3]; mulmatvec3(a, &s->xmat[9 * b], &m.jnt_axis[3 * m.body_hinge[b]]);
      s->bias[d] = dot3(a, &n[3 * b]);
    } else if (m.body_jnttype[b] == J_SLIDE) {
      real a[3]; mulmatvec3(a, &s->xmat[9 * b], &m.jnt_axis[3 * m.body_hinge[b]]);
      s->bias[d] = dot3(a, &f[3 * b]);
    } else if (m.body_jnttype[b] == J_FREE) {
      for (int k = 0; k < 3; k++) s->bias[d + k] = f[3 * b + k];
      real t[3]; mulmatTvec3(t, &s->xmat[9 * b], &n[3 * b]);
      for (int k = 0; k < 3; k++) s->bias[d + 3 + k] = t[k];
    }
    if (p != 0) {
      real dd[3]; for (int k = 0; k < 3; k++) dd[k] = s->xpos[3 * b + k] - s->xpos[3 * p + k];
      real t[3]; cross3(t, dd, &f[3 * b]);
      for (int k = 0; k < 3; k++) { f[3 * p + k] += f[3 * b + k]; n[3 * p + k] += n[3 * b + k] + t[k]; }
    }
  }
}

// ================================================================ actuation (scene_pbr.xml:11 `general`, affine bias)
void actuation(orc_sim* s) {
  const Model& m = s->m;
  s->qfrc_act.assign(m.nv, 0); s->act_force.assign(m.nu, 0);
  for (int a = 0; a < m.nu; a++) {
    int d = m.act_dof[a];
    real c = s->ctrl[a];
    if (m.act_ctrllimited[a]) c = std::min(std::max(c, m.act_ctrlrange[2 * a]), m.act_ctrlrange[2 * a + 1]);
    // hinge joint transmission: length = qpos, velocity = qvel of the joint (arm dofs precede free bodies)
    real force = m.act_gain[a] * c + m.act_bias[3 * a] + m.act_bias[3 * a + 1] * s->qpos[d] + m.act_bias[3 * a + 2] * s->qvel[d];
    if (m.act_forcelimited[a]) force = std::min(std::max(force, m.act_forcerange[2 * a]), m.act_forcerange[2 * a + 1]);
    s->act_force[a] = force;
    s->qfrc_act[d] += force;
  }
  for (size_t h = 0; h < m.jnt_actfrclimited.size(); h++) if (m.jnt_actfrclimited[h]) {      // mj_fwdActuation: jnt_actfrcrange
    int d = m.body_dofadr[m.arm_body[h]];
    s->qfrc_act[d] = std::min(std::max(s->qfrc_act[d], m.jnt_actfrcrange[2 * h]), m.jnt_actfrcrange[2 * h + 1]);
  }
}

// ================================================================ collision
// support point of geom g in world direction dir (libccd-style support mapping per geom type)
void support(const orc_sim* s, int g, const real* dir, real* out) {
  const Model& m = s->m;
  const real* R = &s->gmat[9 * g]; const real* P = &s->gpos[3 * g];
  real dl[3]; mulmatTvec3(dl, R, dir);
  real loc[3] = {0, 0, 0};
  const real* sz = &m.geom_size[3 * g];
  switch (m.geom_type[g]) {
    case G_SPHERE: { real n = norm3(dl); if (n > MINVAL) for (int k = 0; k < 3; k++) loc[k] = sz[0] * dl[k] / n; break; }
    case G_BOX: for (int k = 0; k < 3; k++) loc[k] = dl[k] >= 0 ? sz[k] : -sz[k]; break;
    case G_CAPSULE: {
      real n = norm3(dl);
      if (n > MINVAL) for (int k = 0; k < 3; k++) loc[k] = sz[0] * dl[k] / n;
      loc[2] += dl[2] >= 0 ? sz[1] : -sz[1];
      break;
    }
    case G_CYLINDER: {
      real n = std::sqrt(dl[0] * dl[0] + dl[1] * dl[1]);
      if (n > MINVAL) { loc[0] = sz[0] * dl[0] / n; loc[1] = sz[0] * dl[1] / n; }
      loc[2] = dl[2] >= 0 ? sz[1] : -sz[1];
      break;
    }
    case G_MESH: {
      const real* v = &m.mesh_vert[3 * m.geom_vertadr[g]];
      int n = m.geom_vertnum[g], best = 0; real bv = -1e300;
      for (int i = 0; i < n; i++) { real d = dot3(v + 3 * i, dl); if (d > bv) { bv = d; best = i; } }
      for (int k = 0; k < 3; k++) loc[k] = v[3 * best + k];
      break;
    }
    default: break;
  }
  real w[3]; mulmatvec3(w, R, loc);
  for (int k = 0; k < 3; k++) out[k] = P[k] + w[k];
}

void geom_center(const orc_sim* s, int g, real* out) {
  real w[3]; mulmatvec3(w, &s->gmat[9 * g], &s->m.geom_center[3 * g]);
  for (int k = 0; k < 3; k++) out[k] = s->gpos[3 * g + k] + w[k];
}

// Minkowski-difference support  v = s1(dir) - s2(-dir)  plus the two witnesses (libccd ccd_support_t)
struct MV { real v[3], a[3], b[3]; };
void mdsupport(const orc_sim* s, int g1, int g2, const real* dir, const real* org, MV* o) {
  real nd[3] = {-dir[0], -dir[1], -dir[2]};
  support(s, g1, dir, o->a); support(s, g2, nd, o->b);
  for (int k = 0; k < 3; k++) { o->a[k] -= org[k]; o->b[k] -= org[k]; o->v[k] = o->a[k] - o->b[k]; }
}

inline bool isz(real x) { return std::fabs(x) < 1e-10; }   // CCD_EPS-style zero test

// squared distance of the origin to triangle (x0,B,C): closest point `wit` and its barycentric weights `bw`
// (libccd ccdVec3PointTriDist2 for P=0, extended with the weights)
real origin_tri_dist2(const real* x0, const real* B, const real* C, real* wit, real* bw) {
  real d1[3], d2[3], a[3];
  for (int k = 0; k < 3; k++) { d1[k] = B[k] - x0[k]; d2[k] = C[k] - x0[k]; a[k] = x0[k]; }
  real v = dot3(d1, d1), w = dot3(d2, d2), p = dot3(a, d1), q = dot3(a, d2), r = dot3(d1, d2);
  real den = w * v - r * r, sp = -1, tp = -1;
  if (!isz(den)) { sp = (q * r - w * p) / den; tp = (-sp * r - q) / w; }
  auto seg = [&](const real* P0, const real* P1, real* wt, real* tout) {   // origin to segment
    real dd[3] = {P1[0] - P0[0], P1[1] - P0[1], P1[2] - P0[2]};
    real t = -dot3(P0, dd) / std::max(dot3(dd, dd), MINVAL);
    t = std::min(std::max(t, 0.0), 1.0);
    for (int k = 0; k < 3; k++) wt[k] = P0[k] + t * dd[k];
    *tout = t;
    return dot3(wt, wt);
  };
  if ((isz(sp) || sp > 0) && (isz(sp - 1) || sp < 1) && (isz(tp) || tp > 0) && (isz(tp - 1) || tp < 1) &&
      (isz(tp + sp - 1) || tp + sp < 1)) {
    for (int k = 0; k < 3; k++) wit[k] = x0[k] + sp * d1[k] + tp * d2[k];
    bw[0] = 1 - sp - tp; bw[1] = sp; bw[2] = tp;
    return dot3(wit, wit);
  }
  real w1[3], w2[3], w3[3], t1, t2, t3;
  real e1 = seg(x0, B, w1, &t1), e2 = seg(x0, C, w2, &t2), e3 = seg(B, C, w3, &t3);
  real best = e1; std::memcpy(wit, w1, sizeof w1); bw[0] = 1 - t1; bw[1] = t1; bw[2] = 0;
  if (e2 < best) { best = e2; std::memcpy(wit, w2, sizeof w2); bw[0] = 1 - t2; bw[1] = 0; bw[2] = t2; }
  if (e3 < best) { best = e3; std::memcpy(wit, w3, sizeof w3); bw[0] = 0; bw[1] = 1 - t3; bw[2] = t3; }
  return best;
}

// Interior point of a geom for the MPR origin ray: for primitives the point closest to `target` (the other geom's
// centre) pulled slightly inside, so the ray follows the local penetration direction (with the fixed
// centre of the large flat table box MPR's depth is erratic; EPA in mujoco >= 3.3 has no such dependence).
void interior_point(const orc_sim* s, int g, const real* target, real* out) {
  const Model& m = s->m;
  int tp = m.geom_type[g];
  if (tp == G_MESH || tp == G_SPHERE || tp == G_PLANE) { geom_center(s, g, out); return; }
  const real* R = &s->gmat[9 * g]; const real* P = &s->gpos[3 * g]; const real* sz = &m.geom_size[3 * g];
  real rel[3] = {target[0] - P[0], target[1] - P[1], target[2] - P[2]}, t[3];
  mulmatTvec3(t, R, rel);
  if (tp == G_BOX) {
    for (int i = 0; i < 3; i++) { real lim = sz[i] - std::min(1e-3, 0.5 * sz[i]); t[i] = std::min(std::max(t[i], -lim), lim); }
  } else if (tp == G_CYLINDER) {
    real rmax = sz[0] - std::min(1e-3, 0.5 * sz[0]), rho = std::sqrt(t[0] * t[0] + t[1] * t[1]);
    if (rho > rmax) { real sc = rmax / rho; t[0] *= sc; t[1] *= sc; }
    real lim = sz[1] - std::min(1e-3, 0.5 * sz[1]);
    t[2] = std::min(std::max(t[2], -lim), lim);
  } else { t[0] = 0; t[1] = 0; t[2] = std::min(std::max(t[2], -sz[1]), sz[1]); }
  real w[3]; mulmatvec3(w, R, t);
  for (int k = 0; k < 3; k++) out[k] = P[k] + w[k];
}

// EPA (expanding polytope) from the tetrahedron MPR ends with - its interior point v0 and the portal v1 v2 v3, which contains the
// origin -: the face of the polytope nearest to the origin is pushed out along its normal until the support point in that direction
// lies on it (within tol).  That face is then a face of the Minkowski difference A - B, and its distance the MINIMUM translation
// that separates the geoms - what mujoco >= 3.3's native GJK / EPA reports.  Vertices carry their witness points.
struct EpaFace { int a, b, c; real n[3], d; bool alive; };
bool epa_penetration(const orc_sim* s, int g1, int g2, const real* org, const MV& v0, const MV& v1, const MV& v2, const MV& v3,
                     real tol, real* depth, real* dir, real* pos, int* iters_out) {
  std::vector<MV> V = {v0, v1, v2, v3};
  std::vector<EpaFace> F;
  auto add_face = [&](int a, int b, int c) {
    EpaFace f; f.a = a; f.b = b; f.c = c; f.alive = true;
    real e1[3], e2[3];
    for (int k = 0; k < 3; k++) { e1[k] = V[b].v[k] - V[a].v[k]; e2[k] = V[c].v[k] - V[a].v[k]; }
    cross3(f.n, e1, e2);
    real len = norm3(f.n);
    if (len < 1e-14) { f.d = 1e30; f.n[0] = f.n[1] = f.n[2] = 0; f.alive = false; F.push_back(f); return; }
    for (int k = 0; k < 3; k++) f.n[k] /= len;
    f.d = dot3(f.n, V[a].v);          // (no per-face flip: consistent winding by construction, as in the kernels)
    F.push_back(f);
  };
  {
    real e1[3], e2[3], nn[3], to0[3];
    for (int k = 0; k < 3; k++) { e1[k] = v2.v[k] - v1.v[k]; e2[k] = v3.v[k] - v1.v[k]; to0[k] = v0.v[k] - v1.v[k]; }
    cross3(nn, e1, e2);
    bool swap = dot3(nn, to0) > 0;
    if (!swap) { add_face(1, 2, 3); add_face(0, 2, 1); add_face(0, 3, 2); add_face(0, 1, 3); }
    else { add_face(1, 3, 2); add_face(0, 1, 2); add_face(0, 2, 3); add_face(0, 3, 1); }
  }
  int best = -1;
  bool converged = false;
  const int max_expansions = 30;          // as the kernels: 4 + 2 x 30 faces fill their 64 face lanes
  for (int it = 0; it <= max_expansions; it++) {
    best = -1;
    for (size_t i = 0; i < F.size(); i++) if (F[i].alive && (best < 0 || F[i].d < F[best].d)) best = (int)i;
    if (best < 0) return false;
    MV w;
    mdsupport(s, g1, g2, F[best].n, org, &w);
    real reach = dot3(F[best].n, w.v) - F[best].d;
    bool dup = false;
    for (const MV& x : V) if (std::fabs(x.v[0] - w.v[0]) + std::fabs(x.v[1] - w.v[1]) + std::fabs(x.v[2] - w.v[2]) < 1e-12) dup = true;
    if (reach <= tol) { converged = true; break; }
    if (dup || it == max_expansions) break;
    if (iters_out) (*iters_out)++;
    int wi = (int)V.size();
    V.push_back(w);
    // faces that see the new vertex go; the boundary of the hole (edges shared with a face that stays) is closed with new faces
    std::vector<std::pair<int, int>> edges;
    std::vector<char> vis(F.size(), 0);
    for (size_t i = 0; i < F.size(); i++) if (F[i].alive && dot3(F[i].n, w.v) - F[i].d > 0) vis[i] = 1;
    for (size_t i = 0; i < F.size(); i++) if (vis[i]) {
      int e[3][2] = {{F[i].a, F[i].b}, {F[i].b, F[i].c}, {F[i].c, F[i].a}};
      for (auto& ed : e) {
        bool shared_with_visible = false;
        for (size_t j = 0; j < F.size(); j++) if (j != i && vis[j]) {
          int g[3][2] = {{F[j].a, F[j].b}, {F[j].b, F[j].c}, {F[j].c, F[j].a}};
          for (auto& gd : g) if (gd[0] == ed[1] && gd[1] == ed[0]) shared_with_visible = true;
        }
        if (!shared_with_visible) edges.push_back({ed[0], ed[1]});
      }
    }
    for (size_t i = 0; i < F.size(); i++) if (vis[i]) F[i].alive = false;
    for (auto& ed : edges) add_face(ed.first, ed.second, wi);
  }
  if (!converged) return false;           // an inner bound only (a tiny sphere deep inside a mesh): the caller keeps MPR's answer
  *depth = F[best].d;
  for (int k = 0; k < 3; k++) dir[k] = F[best].n[k];
  // witness: the projection of the origin onto the face, in barycentric coordinates of the face's vertices.  A flat facet of the
  // Minkowski difference (an edge against an edge, a face against an edge) is triangulated by the polytope and its triangles are
  // coplanar up to rounding: among the faces coplanar with the nearest one (plane distance within tol, normal within 1e-5) the
  // witness is interpolated on the one that contains the projection best (largest smallest barycentric weight), as in the kernels.
  real p[3] = {F[best].d * F[best].n[0], F[best].d * F[best].n[1], F[best].d * F[best].n[2]}, e1[3], e2[3], ep[3];
  {
    int wbest = -1; real wscore = -1e30;
    for (size_t i = 0; i < F.size(); i++) {
      const EpaFace& g = F[i];
      if (!g.alive || g.d - F[best].d > tol || dot3(g.n, F[best].n) < 1 - 1e-5) continue;
      for (int k = 0; k < 3; k++) { e1[k] = V[g.b].v[k] - V[g.a].v[k]; e2[k] = V[g.c].v[k] - V[g.a].v[k]; ep[k] = p[k] - V[g.a].v[k]; }
      real q11 = dot3(e1, e1), q12 = dot3(e1, e2), q22 = dot3(e2, e2), s1 = dot3(ep, e1), s2 = dot3(ep, e2), qden = q11 * q22 - q12 * q12;
      if (!(qden > 1e-30)) continue;
      real ub = (q22 * s1 - q12 * s2) / qden, uc = (q11 * s2 - q12 * s1) / qden;
      real score = std::min(1 - ub - uc, std::min(ub, uc));
      if (score > wscore) { wscore = score; wbest = (int)i; }
    }
    if (wbest >= 0) best = wbest;
  }
  const EpaFace& f = F[best];
  for (int k = 0; k < 3; k++) { e1[k] = V[f.b].v[k] - V[f.a].v[k]; e2[k] = V[f.c].v[k] - V[f.a].v[k]; ep[k] = p[k] - V[f.a].v[k]; }
  real d11 = dot3(e1, e1), d12 = dot3(e1, e2), d22 = dot3(e2, e2), r1 = dot3(ep, e1), r2 = dot3(ep, e2), den = d11 * d22 - d12 * d12;
  real wb = den > 1e-30 ? (d22 * r1 - d12 * r2) / den : 0, wc = den > 1e-30 ? (d11 * r2 - d12 * r1) / den : 0;
  wb = std::min(std::max(wb, 0.0), 1.0); wc = std::min(std::max(wc, 0.0), 1.0 - wb);
  real wa = 1 - wb - wc;
  for (int k = 0; k < 3; k++) {
    real p1 = wa * V[f.a].a[k] + wb * V[f.b].a[k] + wc * V[f.c].a[k], p2 = wa * V[f.a].b[k] + wb * V[f.b].b[k] + wc * V[f.c].b[k];
    pos[k] = 0.5 * (p1 + p2) + org[k];
  }
  return true;
}

// MPR penetration (libccd ccdMPRPenetration restated).  Returns true when the geoms intersect and
// fills depth, dir (from g1 into g2) and pos (world).
bool mpr_penetration(const orc_sim* s, int g1, int g2, real* depth, real* dir, real* pos) {
  const Model& m = s->m;
  real org[3], c1[3] = {0, 0, 0}, c2[3], cdef2[3];
  geom_center(s, g2, cdef2);
  interior_point(s, g1, cdef2, org);          // work in a frame centred on geom1's interior point
  interior_point(s, g2, org, c2);
  for (int k = 0; k < 3; k++) c2[k] -= org[k];
  MV v0, v1, v2, v3, v4;
  for (int k = 0; k < 3; k++) { v0.a[k] = c1[k]; v0.b[k] = c2[k]; v0.v[k] = c1[k] - c2[k]; }
  if (isz(v0.v[0]) && isz(v0.v[1]) && isz(v0.v[2])) v0.v[0] += 1e-5;
  real d[3] = {-v0.v[0], -v0.v[1], -v0.v[2]}; normalize3(d);
  mdsupport(s, g1, g2, d, org, &v1);
  real dt = dot3(v1.v, d);
  if (isz(dt) || dt < 0) return false;
  cross3(d, v0.v, v1.v);
  int state = 0;   // 0: portal found, 1: touching at v1, 2: origin on segment v0-v1
  if (isz(norm3(d))) {
    state = (isz(v1.v[0]) && isz(v1.v[1]) && isz(v1.v[2])) ? 1 : 2;
  } else {
    normalize3(d);
    mdsupport(s, g1, g2, d, org, &v2);
    dt = dot3(v2.v, d);
    if (isz(dt) || dt < 0) return false;
    real va[3], vb[3];
    for (int k = 0; k < 3; k++) { va[k] = v1.v[k] - v0.v[k]; vb[k] = v2.v[k] - v0.v[k]; }
    cross3(d, va, vb); normalize3(d);
    if (dot3(d, v0.v) > 0) { std::swap(v1, v2); for (int k = 0; k < 3; k++) d[k] = -d[k]; }
    bool have3 = false;
    for (int guard = 0; guard < 100 && !have3; guard++) {
      mdsupport(s, g1, g2, d, org, &v3);
      dt = dot3(v3.v, d);
      if (isz(dt) || dt < 0) return false;
      bool cont = false;
      cross3(va, v1.v, v3.v);
      dt = dot3(va, v0.v);
      if (dt < 0 && !isz(dt)) { v2 = v3; cont = true; }
      if (!cont) {
        cross3(va, v3.v, v2.v);
        dt = dot3(va, v0.v);
        if (dt < 0 && !isz(dt)) { v1 = v3; cont = true; }
      }
      if (cont) {
        for (int k = 0; k < 3; k++) { va[k] = v1.v[k] - v0.v[k]; vb[k] = v2.v[k] - v0.v[k]; }
        cross3(d, va, vb); normalize3(d);
      } else have3 = true;
    }
    if (!have3) return false;
  }
  auto portal_dir = [&](real* o) {
    real a[3], b[3];
    for (int k = 0; k < 3; k++) { a[k] = v2.v[k] - v1.v[k]; b[k] = v3.v[k] - v1.v[k]; }
    cross3(o, a, b); normalize3(o);
  };
  auto reach_tol = [&](const MV& x, const real* dd) {
    real dv1 = dot3(v1.v, dd), dv2 = dot3(v2.v, dd), dv3 = dot3(v3.v, dd), dv4 = dot3(x.v, dd);
    real dm = std::min(std::min(dv4 - dv1, dv4 - dv2), dv4 - dv3);
    return isz(dm - m.mpr_tol) || dm < m.mpr_tol;
  };
  auto expand = [&](const MV& x) {
    real v4v0[3]; cross3(v4v0, x.v, v0.v);
    real t = dot3(v1.v, v4v0);
    if (t > 0) { t = dot3(v2.v, v4v0); if (t > 0) v1 = x; else v3 = x; }
    else { t = dot3(v3.v, v4v0); if (t > 0) v2 = x; else v1 = x; }
  };
  if (state == 1) {           // touching contact
    *depth = 0; dir[0] = dir[1] = dir[2] = 0;
    for (int k = 0; k < 3; k++) pos[k] = 0.5 * (v1.a[k] + v1.b[k]) + org[k];
    return true;
  }
  if (state == 2) {           // origin on the v0-v1 segment
    for (int k = 0; k < 3; k++) { pos[k] = 0.5 * (v1.a[k] + v1.b[k]) + org[k]; dir[k] = v1.v[k]; }
    *depth = normalize3(dir);
    return true;
  }
  // refine portal until it encapsulates the origin
  for (int guard = 0;; guard++) {
    portal_dir(d);
    dt = dot3(d, v1.v);
    if (isz(dt) || dt > 0) break;                       // portal encapsules origin
    mdsupport(s, g1, g2, d, org, &v4);
    dt = dot3(v4.v, d);
    if (!(isz(dt) || dt > 0)) return false;             // cannot encapsule origin
    if (reach_tol(v4, d) || guard > 100) return false;
    expand(v4);
  }
  if (s->narrow == 1) {       // EPA takes over as soon as the portal encloses the origin: the tetrahedron v0 v1 v2 v3 contains it from here on,
    int n_it = 0;             // and MPR's own refinement of the portal towards the surface is work EPA does anyway
    bool ok = epa_penetration(s, g1, g2, org, v0, v1, v2, v3, m.mpr_tol, depth, dir, pos, &n_it);
    const_cast<orc_sim*>(s)->epa_iters += n_it;
    if (ok) return true;
  }
  // find penetration
  for (int it = 0;; it++) {
    portal_dir(d);
    mdsupport(s, g1, g2, d, org, &v4);
    if (reach_tol(v4, d) || it > m.mpr_iter) {
      real pd[3], bw[3];
      real d2 = origin_tri_dist2(v1.v, v2.v, v3.v, pd, bw);
      *depth = std::sqrt(d2);
      if (isz(pd[0]) && isz(pd[1]) && isz(pd[2])) { *depth = 0; for (int k = 0; k < 3; k++) dir[k] = d[k]; }
      else { for (int k = 0; k < 3; k++) dir[k] = pd[k]; normalize3(dir); }
      // contact position: midpoint of the two witness points of the closest point on the portal (the witness
      // pair GJK/EPA reports; libccd's origin-ray weights are path dependent for deep penetrations)
      for (int k = 0; k < 3; k++) {
        real p1 = bw[0] * v1.a[k] + bw[1] * v2.a[k] + bw[2] * v3.a[k];
        real p2 = bw[0] * v1.b[k] + bw[1] * v2.b[k] + bw[2] * v3.b[k];
        pos[k] = 0.5 * (p1 + p2) + org[k];
      }
      return true;
    }
    expand(v4);
  }
}

void make_frame(real* fr) {   // mju_makeFrame: fr[0:3] given (unit); fill two tangents
  real* x = fr; real* y = fr + 3; real* z = fr + 6;
  y[0] = y[1] = y[2] = 0;
  if (x[1] < 0.5 && x[1] > -0.5) y[1] = 1; else y[2] = 1;
  real t = dot3(x, y);
  for (int k = 0; k < 3; k++) y[k] -= t * x[k];
  normalize3(y);
  cross3(z, x, y);
}

void world_aabb(const orc_sim* s, int g, real* lo, real* hi) {
  const Model& m = s->m;
  const real* R = &s->gmat[9 * g];
  const real* c = &m.geom_aabb[6 * g]; const real* h = c + 3;
  real cw[3]; mulmatvec3(cw, R, c);
  for (int i = 0; i < 3; i++) {
    real e = std::fabs(R[3 * i]) * h[0] + std::fabs(R[3 * i + 1]) * h[1] + std::fabs(R[3 * i + 2]) * h[2];
    lo[i] = s->gpos[3 * g + i] + cw[i] - e; hi[i] = s->gpos[3 * g + i] + cw[i] + e;
  }
}

void set_contact_params(const Model& m, Contact& c, int g1, int g2) {
  // mj_contactParam at equal priority: condim max, friction max, solref/solimp solmix-weighted
  c.dim = std::max(m.geom_condim[g1], m.geom_condim[g2]);
  real f[3];
  for (int k = 0; k < 3; k++) f[k] = std::max(m.geom_friction[3 * g1 + k], m.geom_friction[3 * g2 + k]);
  c.friction[0] = c.friction[1] = f[0]; c.friction[2] = f[1]; c.friction[3] = c.friction[4] = f[2];
  real s1 = m.geom_solmix[g1], s2 = m.geom_solmix[g2];
  real mix = (s1 >= MINVAL && s2 >= MINVAL) ? s1 / (s1 + s2) : (s1 < MINVAL && s2 < MINVAL ? 0.5 : (s1 < MINVAL ? 0.0 : 1.0));
  if (m.geom_priority[g1] > m.geom_priority[g2]) mix = 1; else if (m.geom_priority[g1] < m.geom_priority[g2]) mix = 0;
  for (int k = 0; k < 2; k++) c.solref[k] = mix * m.geom_solref[2 * g1 + k] + (1 - mix) * m.geom_solref[2 * g2 + k];
  for (int k = 0; k < 5; k++) c.solimp[k] = mix * m.geom_solimp[5 * g1 + k] + (1 - mix) * m.geom_solimp[5 * g2 + k];
}

// ---------------------------------------------------------------- multi-contact for flat faces ("multiccd")
// The reference enables mjENBL_MULTICCD (so101_sim/tasks/base/so100_task.py:151).  MuJoCo's convex-pair multi-contact
// re-runs the penetration query on configurations tilted by +-1e-3 rad about the two tangent axes and keeps results
// farther apart than 1e-3 of the smaller bounding radius (libccd path); the native-ccd path of >= 3.3 clips the
// aligned faces of box / mesh pairs.  Both amount to sampling the extreme points of the flat contact patch.  Restated
// here for the configurations in which the tilted query has a closed form: one geom presents a flat REFERENCE FACE
// (the plane, or the box face / cylinder cap along which the pair is shallowest, see collision()) and the other
// geom (any convex type, the INCIDENT geom) is sampled with its support function:
//   a_0 = support(-f)                       deepest point below the face (f = outward face normal)
//   a_k = support(-f + eps * s_k), k=1..4   s_k = (+-u +- v)/sqrt(2), u/v the face axes, eps = 1e-3 (the tilt angle)
// A sample is a contact if it lies below the face plane, inside the face rectangle / disc, and farther than
// 1e-3 * min(rbound) from the contacts (positions) already accepted.  All contacts of the pair share the normal +-f.  When a_0
// does not qualify the single MPR contact is kept.  Other convex pairs (hull-hull, cylinder, capsule) keep one contact.
constexpr real FACE_DEPTH_REL = 1e-2, FACE_DEPTH_ABS = 1e-6, PATCH_EPS = 1e-3, PATCH_DUP = 1e-3;
constexpr int NCPP = 5;

struct Patch { int n; real nrm[3], dist[NCPP], pos[NCPP][3]; };

// inside the face outline (rectangle hu x hv, or disc of radius hu when hv < 0) by at least `margin`
bool inside_margin(const real* rel, const real* u, const real* v, real hu, real hv, real margin) {
  real pu = dot3(rel, u), pv = dot3(rel, v), ru = hu - margin;
  if (hv >= 0) return std::fabs(pu) <= ru && std::fabs(pv) <= hv - margin;
  return ru >= 0 && pu * pu + pv * pv <= ru * ru;
}

bool inside_face(const real* rel, const real* u, const real* v, real hu, real hv) {
  if (hu < 0) return true;                                            // unbounded plane
  real pu = dot3(rel, u), pv = dot3(rel, v);
  return hv >= 0 ? (std::fabs(pu) <= hu && std::fabs(pv) <= hv) : (pu * pu + pv * pv <= hu * hu);
}



// reference face: outward normal f (towards the incident geom), point c on it, axes u, v with half extents hu, hv
// (hu < 0: unbounded plane); dup_tol: minimal distance between two contacts of the pair
bool face_patch(const orc_sim* s, int gI, const real* f, const real* c, const real* u, const real* v, real hu, real hv,
                real dup_tol, Patch* out) {
  out->n = 0;
  static const real su[4] = {1, -1, -1, 1}, sv[4] = {1, 1, -1, -1};
  for (int k = 0; k < NCPP; k++) {
    real d[3];
    for (int i = 0; i < 3; i++) d[i] = -f[i] + (k == 0 ? 0.0 : PATCH_EPS * (su[k - 1] * u[i] + sv[k - 1] * v[i]) * 0.70710678118654752440);
    normalize3(d);
    real p[3]; support(s, gI, d, p);
    real rel[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
    real dist = dot3(rel, f);
    bool ok = dist < 0;
    ok = ok && inside_face(rel, u, v, hu, hv);
    if (k == 0 && !ok) return false;
    real cp[3] = {p[0] - 0.5 * dist * f[0], p[1] - 0.5 * dist * f[1], p[2] - 0.5 * dist * f[2]};   // contact position
    for (int j = 0; j < out->n && ok; j++) {
      real dd[3] = {cp[0] - out->pos[j][0], cp[1] - out->pos[j][1], cp[2] - out->pos[j][2]};
      if (norm3(dd) < dup_tol) ok = false;
    }
    if (!ok) continue;
    int n = out->n++;
    for (int i = 0; i < 3; i++) out->pos[n][i] = cp[i];
    out->dist[n] = dist;
  }
  return true;
}

// Hull against hull (round 5).  MuJoCo's multiccd gives a mesh pair resting face on face (or face on edge) several contacts; until round 4
// kernel and oracle kept the single EPA contact there (a mug on a plate rocked on one point per hull pair).  EPA ends on the nearest face
// of the Minkowski difference: normal n (geom 1 -> geom 2), depth, witness midpoint pos, so w1 = pos + depth/2 n is a point of geom 1's
// surface and w2 = pos - depth/2 n one of geom 2's.  The same tilted support samples as for box faces then take the extreme points of
// whatever flat feature each hull presents along n:
//   b_k = support_2(-n + eps s_k), a_k = support_1(+n + eps s_k), k = 1..4, s_k = (+-u +- v)/sqrt(2), (u, v) = make_frame(n), eps = 1e-3.
// b_k becomes a contact when it lies below geom 1's supporting plane (through w1, normal n) and laterally inside geom 1's feature there:
// with r the unit lateral direction from w1 to b_k, r . (b_k - w1) <= r . (support_1(n + eps r) - w1) + 1e-6 - the feature's extent in
// that direction, again by a tilted support; a vertex or a curved patch has extent 0, so nothing beyond the EPA contact survives there.
// Samples laterally closer than dup_tol to w1 are skipped (they would only repeat the EPA contact).
// a_k symmetrically against geom 2's feature at w2.  Accepted in the order b_1..b_4, a_1..a_4 behind the EPA contact (slot 0) while they are
// farther than dup_tol from the contacts already accepted, up to NCPP in all; every contact carries the normal n, its distance is the
// sample's signed distance to the other hull's supporting plane and its position the midpoint between sample and plane.
void hull_patch(const orc_sim* s, int g1, int g2, const real* n, real depth, const real* pos, real dup_tol, Patch* out) {
  out->n = 1; out->dist[0] = -depth;
  for (int i = 0; i < 3; i++) out->pos[0][i] = pos[i];
  real fr[9] = {n[0], n[1], n[2], 0, 0, 0, 0, 0, 0};
  make_frame(fr);
  const real* u = fr + 3; const real* v = fr + 6;
  static const real su[4] = {1, -1, -1, 1}, sv[4] = {1, 1, -1, -1};
  real w[2][3];
  for (int i = 0; i < 3; i++) { w[0][i] = pos[i] + 0.5 * depth * n[i]; w[1][i] = pos[i] - 0.5 * depth * n[i]; }
  for (int side = 0; side < 2; side++) {              // side 0: samples of geom 2 against geom 1's feature; side 1: the other way round
    int gs = side == 0 ? g2 : g1, go = side == 0 ? g1 : g2;
    real sg = side == 0 ? -1.0 : 1.0;                // the sampled hull is probed towards the other one: -n for geom 2, +n for geom 1
    const real* wo = w[side];                         // surface point of the OTHER hull (the plane the sample is measured against)
    for (int k = 0; k < 4; k++) {
      if (out->n >= NCPP) return;
      real d[3];
      for (int i = 0; i < 3; i++) d[i] = sg * n[i] + PATCH_EPS * (su[k] * u[i] + sv[k] * v[i]) * 0.70710678118654752440;
      normalize3(d);
      real p[3]; support(s, gs, d, p);
      real rel[3] = {p[0] - wo[0], p[1] - wo[1], p[2] - wo[2]};
      real dist = -sg * dot3(rel, n);                 // geom 2's sample: (p - w1) . n;  geom 1's sample: -(p - w2) . n
      if (!(dist < 0)) continue;
      real h = dot3(rel, n), r[3] = {rel[0] - h * n[0], rel[1] - h * n[1], rel[2] - h * n[2]};
      real rl = normalize3(r);
      if (rl < dup_tol) continue;                     // laterally at the EPA witness: would repeat the EPA contact
      {
        real de[3], e[3];
        for (int i = 0; i < 3; i++) de[i] = -sg * n[i] + PATCH_EPS * r[i];
        normalize3(de);
        support(s, go, de, e);
        real ext = r[0] * (e[0] - wo[0]) + r[1] * (e[1] - wo[1]) + r[2] * (e[2] - wo[2]);
        if (!(rl <= ext + 1e-6)) continue;
      }
      real cp[3] = {p[0] + sg * 0.5 * dist * n[0], p[1] + sg * 0.5 * dist * n[1], p[2] + sg * 0.5 * dist * n[2]};
      bool ok = true;
      for (int j = 0; j < out->n && ok; j++) {
        real dd[3] = {cp[0] - out->pos[j][0], cp[1] - out->pos[j][1], cp[2] - out->pos[j][2]};
        if (norm3(dd) < dup_tol) ok = false;
      }
      if (!ok) continue;
      int q = out->n++;
      for (int i = 0; i < 3; i++) out->pos[q][i] = cp[i];
      out->dist[q] = dist;
    }
  }
}

// flat face number `axis` (box: 0..2 = local x/y/z on the side facing `toward`; cylinder: only axis 2 = the cap) of
// geom g: outward normal f, centre c, in-plane axes u/v with half extents (hv < 0: disc of radius hu).  Returns the
// cosine between f and `toward` (unit, world); the face is a candidate when that exceeds FACE_MIN_COS.
real flat_face(const orc_sim* s, int g, int axis, const real* toward, real* f, real* c, real* u, real* v, real* hu, real* hv, real* half = nullptr) {
  const real* R = &s->gmat[9 * g]; const real* P = &s->gpos[3 * g]; const real* sz = &s->m.geom_size[3 * g];
  real loc[3]; mulmatTvec3(loc, R, toward);
  if (s->m.geom_type[g] == G_CYLINDER) {
    if (axis != 2) return -1;
    real sg = loc[2] >= 0 ? 1.0 : -1.0;
    for (int k = 0; k < 3; k++) { f[k] = sg * R[3 * k + 2]; u[k] = R[3 * k]; v[k] = R[3 * k + 1]; c[k] = P[k] + f[k] * sz[1]; }
    *hu = sz[0]; *hv = -1;
    if (half) *half = sz[1];
    return 1;
  }
  int i = axis, iu = (i + 1) % 3, iv = (i + 2) % 3;
  real sg = loc[i] >= 0 ? 1.0 : -1.0;
  for (int k = 0; k < 3; k++) { f[k] = sg * R[3 * k + i]; u[k] = R[3 * k + iu]; v[k] = R[3 * k + iv]; c[k] = P[k] + f[k] * sz[i]; }
  *hu = sz[iu]; *hv = sz[iv];
  if (half) *half = sz[i];
  return 1;
}

void collision(orc_sim* s) {
  const Model& m = s->m;
  s->con.clear();
  if (!s->collide) return;
  std::vector<real> lo(3 * m.ngeom), hi(3 * m.ngeom);
  for (int g = 0; g < m.ngeom; g++) world_aabb(s, g, &lo[3 * g], &hi[3 * g]);
  for (int p = 0; p < m.npair; p++) {
    int g1 = m.pair_geom[2 * p], g2 = m.pair_geom[2 * p + 1];
    if (m.geom_type[g1] > m.geom_type[g2]) std::swap(g1, g2);   // collision table is upper-triangular in type
    Patch pt; pt.n = 0;
    if (m.geom_type[g1] == G_PLANE) {
      // plane : convex  — reference face = the plane, incident = the other geom
      const real* R = &s->gmat[9 * g1];
      real nrm[3] = {R[2], R[5], R[8]};
      // cheap reject on the world box
      real lowest = 0;
      for (int k = 0; k < 3; k++) lowest += nrm[k] * ((nrm[k] >= 0 ? lo[3 * g2 + k] : hi[3 * g2 + k]) - s->gpos[3 * g1 + k]);
      if (lowest > 0) continue;
      real fr[9] = {nrm[0], nrm[1], nrm[2], 0, 0, 0, 0, 0, 0};
      make_frame(fr);
      if (!face_patch(s, g2, nrm, &s->gpos[3 * g1], fr + 3, fr + 6, -1, -1, PATCH_DUP * m.geom_rbound[g2], &pt)) continue;
      for (int k = 0; k < 3; k++) pt.nrm[k] = nrm[k];
    } else {
      bool sep = false;
      for (int k = 0; k < 3; k++) if (lo[3 * g1 + k] > hi[3 * g2 + k] || lo[3 * g2 + k] > hi[3 * g1 + k]) sep = true;
      if (sep) continue;
      // Flat-face scan (before any iterative query).  For every flat face (box face, cylinder cap) on the side of the
      // other geom's centre, let a0 be the other geom's deepest point below the face plane, d0 its depth:
      //  * d0 <= 0: the face plane separates the pair - no contact, exactly;
      //  * a0 inside the face outline with a lateral margin >= d0, and d0 <= the half thickness behind the face: a0 is a
      //    point of the box whose distance to the box's boundary is d0, so no translation shorter than d0 separates the
      //    pair and the translation d0 along the face normal does - minimum penetration depth d0 along the face normal,
      //    EXACTLY, no iterative query needed (props resting on the table top, finger pads, the static puck);
      //  * a0 merely inside the outline (d0 <= half thickness): a CANDIDATE direction; the shallowest one is kept and
      //    wins over MPR's answer when it is not deeper (1 % + 1e-6 m slack: for a face contact both are the same
      //    number).  MPR's depth is the depth along ITS final portal normal, which for a thin plate against a hull can be
      //    an oblique direction ten times deeper than the plate's face normal; the minimum over both repairs that.
      bool separated = false, exact = false;
      real f[3], c[3], u[3], v[3], hu = 0, hv = 0, cand = 1e30;
      int ref = -1;
      // Faces of one geom are visited in order of increasing depth of the other geom's centre below the face plane
      // (half extent along the axis minus |centre offset along it|, a lower bound of d0); the scan ends at the first
      // exact face: the geoms then share the point a0, so no other face plane separates them, and another exact face
      // would give the same depth.
      for (int side = 0; side < 2 && !separated && !exact; side++) {
        int g = side == 0 ? g1 : g2, gI = side == 0 ? g2 : g1;
        if (m.geom_type[g] != G_BOX && m.geom_type[g] != G_CYLINDER) continue;
        real ci[3], cg[3]; geom_center(s, gI, ci); geom_center(s, g, cg);
        real toward[3] = {ci[0] - cg[0], ci[1] - cg[1], ci[2] - cg[2]};
        real loc[3]; mulmatTvec3(loc, &s->gmat[9 * g], toward);
        real lb[3];
        for (int k = 0; k < 3; k++) lb[k] = m.geom_size[3 * g + k] - std::fabs(loc[k]);
        int order[3] = {0, 1, 2};
        std::stable_sort(order, order + 3, [&](int a, int b) { return lb[a] < lb[b]; });
        for (int it = 0; it < 3 && !separated && !exact; it++) {
          int axis = order[it];
          real f2[3], c2[3], u2[3], v2[3], hu2, hv2, half;
          if (flat_face(s, g, axis, toward, f2, c2, u2, v2, &hu2, &hv2, &half) < 0) continue;
          real cr[3] = {c2[0] - ci[0], c2[1] - ci[1], c2[2] - ci[2]};
          if (dot3(cr, f2) > half) continue;                // d0 >= depth of the incident's centre > half thickness
          real nf[3] = {-f2[0], -f2[1], -f2[2]}, a0[3];
          support(s, gI, nf, a0);
          real rel[3] = {a0[0] - c2[0], a0[1] - c2[1], a0[2] - c2[2]};
          real d0 = -dot3(rel, f2);
          if (!(d0 > 0)) { separated = true; break; }
          if (d0 > half || !inside_face(rel, u2, v2, hu2, hv2)) continue;
          bool ex = inside_margin(rel, u2, v2, hu2, hv2, d0);
          if (!ex && !(d0 < cand)) continue;
          exact = ex; cand = d0; ref = side; hu = hu2; hv = hv2;
          for (int k = 0; k < 3; k++) { f[k] = f2[k]; c[k] = c2[k]; u[k] = u2[k]; v[k] = v2[k]; }
        }
      }
      if (separated) continue;
      real depth = 0, dir[3] = {0, 0, 0}, pos[3] = {0, 0, 0};
      if (!exact) {
        if (!mpr_penetration(s, g1, g2, &depth, dir, pos)) continue;
        if (depth <= 0) continue;               // margin 0: only penetrating contacts are kept
        if (ref >= 0 && !(cand <= depth * (1 + FACE_DEPTH_REL) + FACE_DEPTH_ABS)) ref = -1;
      }
      bool patched = false;
      if (ref >= 0) {
        int gI = ref == 0 ? g2 : g1;
        patched = face_patch(s, gI, f, c, u, v, hu, hv, PATCH_DUP * std::min(m.geom_rbound[g1], m.geom_rbound[g2]), &pt);
        for (int k = 0; k < 3; k++) pt.nrm[k] = ref == 0 ? f[k] : -f[k];
      }
      if (!patched) {
        if (exact) continue;
        pt.n = 1; pt.dist[0] = -depth;
        for (int k = 0; k < 3; k++) { pt.nrm[k] = dir[k]; pt.pos[0][k] = pos[k]; }
        if (s->hull_multi && s->narrow == 1 && m.geom_type[g1] == G_MESH && m.geom_type[g2] == G_MESH)       // (EPA only: MPR's portal normal is no face normal)
          hull_patch(s, g1, g2, dir, depth, pos, PATCH_DUP * std::min(m.geom_rbound[g1], m.geom_rbound[g2]), &pt);
      }
    }
    for (int j = 0; j < pt.n; j++) {
      Contact c;
      c.dist = pt.dist[j];
      for (int k = 0; k < 3; k++) { c.frame[k] = pt.nrm[k]; c.pos[k] = pt.pos[j][k]; }
      c.g1 = g1; c.g2 = g2;
      make_frame(c.frame);
      set_contact_params(m, c, g1, g2);
      s->con.push_back(c);
    }
  }
  // the kernels' capacity rule, mirrored on request (orc_set_contact_capacity): one lane per contact in the Newton solver = 64 slots per env
  s->contacts_reduced = s->contacts_cut = false;
  if (s->contact_capacity > 0 && (int)s->con.size() > s->contact_capacity) {
    std::vector<Contact> kept;
    for (size_t j = 0; j < s->con.size(); j++)
      if (j == 0 || s->con[j].g1 != s->con[j - 1].g1 || s->con[j].g2 != s->con[j - 1].g2) kept.push_back(s->con[j]);      // (a pair's contacts are consecutive)
    s->contacts_reduced = true;
    if ((int)kept.size() > s->contact_capacity) { kept.resize(s->contact_capacity); s->contacts_cut = true; }
    s->con.swap(kept);
  }
}

// ================================================================ constraints
void point_jacobian(const orc_sim* s, int body, const real* p, real* Jp, real* Jr) {   // 3 x nv each
  const Model& m = s->m;
  int nv = m.nv;
  std::fill(Jp, Jp + 3 * nv, 0.0); std::fill(Jr, Jr + 3 * nv, 0.0);
  for (int d = 0; d < nv; d++) {
    if (!is_ancestor(m, m.dof_body[d], body)) continue;
    const real* w = &s->S[6 * d]; const real* vo = w + 3;
    real t[3]; cross3(t, w, p);
    for (int k = 0; k < 3; k++) { Jp[k * nv + d] = vo[k] + t[k]; Jr[k * nv + d] = w[k]; }
  }
}

real impedance(const real* solimp_in, real pos) {
  real dmin = std::min(std::max(solimp_in[0], MINIMP), MAXIMP), dmax = std::min(std::max(solimp_in[1], MINIMP), MAXIMP);
  real width = std::max(solimp_in[2], 0.0), mid = std::min(std::max(solimp_in[3], MINIMP), MAXIMP), power = std::max(solimp_in[4], 1.0);
  if (dmin == dmax || width <= MINVAL) return 0.5 * (dmin + dmax);
  real x = std::fabs(pos) / width;
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  real y;
  if (power == 1) y = x;
  else if (x <= mid) y = std::pow(x, power) / std::pow(mid, power - 1);
  else y = 1 - std::pow(1 - x, power) / std::pow(1 - mid, power - 1);
  return dmin + y * (dmax - dmin);
}

void make_constraints(orc_sim* s, bool freeze_arm) {
  const Model& m = s->m;
  int nv = m.nv;
  // count rows
  struct Row { int type, id, dim; };
  std::vector<Row> rows;
  for (int e = 0; e < m.neq; e++) rows.push_back({C_EQUALITY, e, 1});      // mj_makeConstraint order: equality, friction, limit, contact
  for (int d = 0; d < nv; d++) if (m.dof_frictionloss[d] > 0) rows.push_back({C_FRICTION, d, 1});
  std::vector<std::pair<int, int>> lim;   // (hinge, side)
  for (int h = 0; h < m.narm; h++) if (m.jnt_limited[h]) {
    real q = s->qpos[m.body_qposadr[m.arm_body[h]]];
    if (q - m.jnt_range[2 * h] < 0) lim.push_back({h, 0});
    if (m.jnt_range[2 * h + 1] - q < 0) lim.push_back({h, 1});
  }
  for (size_t k = 0; k < lim.size(); k++) rows.push_back({C_LIMIT, (int)k, 1});
  for (size_t k = 0; k < s->con.size(); k++) rows.push_back({C_CONTACT, (int)k, m.elliptic ? s->con[k].dim : 1});
  int nefc = 0; for (auto& r : rows) nefc += r.dim;
  s->nefc = nefc;
  s->J.assign((size_t)nefc * nv, 0); s->efc_pos.assign(nefc, 0); s->efc_D.assign(nefc, 0); s->efc_R.assign(nefc, 0);
  s->efc_aref.assign(nefc, 0); s->efc_force.assign(nefc, 0); s->efc_floss.assign(nefc, 0); s->efc_b.assign(nefc, 0);
  s->efc_type.assign(nefc, 0); s->efc_id.assign(nefc, 0); s->efc_dim.assign(nefc, 1);
  std::vector<real> Jp(3 * nv), Jr(3 * nv), Jp2(3 * nv), Jr2(3 * nv);
  int i = 0;
  for (auto& r : rows) {
    real solref[2], solimp[5], diag[6] = {0, 0, 0, 0, 0, 0}, pos = 0;
    if (r.type == C_EQUALITY) {                         // mj_instantiateEquality, mjEQ_JOINT with two joints
      int e = r.id, d1 = m.eq_dof[2 * e], d2 = m.eq_dof[2 * e + 1];
      const real* pc = &m.eq_polycoef[5 * e];
      real q1 = s->qpos[m.eq_qposadr[2 * e]], q2 = s->qpos[m.eq_qposadr[2 * e + 1]];      // qpos0 = 0 for both
      real poly = pc[0] + q2 * (pc[1] + q2 * (pc[2] + q2 * (pc[3] + q2 * pc[4])));
      real deriv = pc[1] + q2 * (2 * pc[2] + q2 * (3 * pc[3] + q2 * 4 * pc[4]));
      pos = q1 - poly;
      s->J[(size_t)i * nv + d1] = 1; s->J[(size_t)i * nv + d2] = -deriv;
      std::memcpy(solref, &m.eq_solref[2 * e], sizeof solref); std::memcpy(solimp, &m.eq_solimp[5 * e], sizeof solimp);
      diag[0] = m.dof_invweight0[d1] + m.dof_invweight0[d2];
    } else if (r.type == C_FRICTION) {
      int d = r.id;
      s->J[(size_t)i * nv + d] = 1;
      s->efc_floss[i] = m.dof_frictionloss[d];
      int h = m.body_hinge[m.dof_body[d]];
      std::memcpy(solref, &m.dof_solref[2 * h], sizeof solref); std::memcpy(solimp, &m.dof_solimp[5 * h], sizeof solimp);
      diag[0] = m.dof_invweight0[d];
    } else if (r.type == C_LIMIT) {
      int h = lim[r.id].first, side = lim[r.id].second;
      int b = m.arm_body[h], d = m.body_dofadr[b];
      real q = s->qpos[m.body_qposadr[b]];
      pos = side == 0 ? q - m.jnt_range[2 * h] : m.jnt_range[2 * h + 1] - q;
      s->J[(size_t)i * nv + d] = side == 0 ? 1 : -1;
      std::memcpy(solref, &m.jnt_solref[2 * h], sizeof solref); std::memcpy(solimp, &m.jnt_solimp[5 * h], sizeof solimp);
      diag[0] = m.dof_invweight0[d];
    } else {
      Contact& c = s->con[r.id];
      int b1 = m.geom_body[c.g1], b2 = m.geom_body[c.g2];
      point_jacobian(s, b1, c.pos, Jp.data(), Jr.data());
      point_jacobian(s, b2, c.pos, Jp2.data(), Jr2.data());
      for (int j = 0; j < r.dim; j++) {
        const real* ax = &c.frame[3 * (j < 3 ? j : j - 3)];
        const real* A1 = j < 3 ? Jp.data() : Jr.data(); const real* A2 = j < 3 ? Jp2.data() : Jr2.data();
        for (int d = 0; d < nv; d++) {
          real v = 0;
          for (int k = 0; k < 3; k++) v += ax[k] * (A2[k * nv + d] - A1[k * nv + d]);
          s->J[(size_t)(i + j) * nv + d] = v;
        }
      }
      pos = c.dist;
      std::memcpy(solref, c.solref, sizeof solref); std::memcpy(solimp, c.solimp, sizeof solimp);
      real tran = m.body_invweight0[2 * b1] + m.body_invweight0[2 * b2];
      real rot = m.body_invweight0[2 * b1 + 1] + m.body_invweight0[2 * b2 + 1];
      for (int j = 0; j < r.dim; j++) diag[j] = j < 3 ? tran : rot;
    }
    if (freeze_arm) { /* arm is restored after every substep by the caller; nothing to mask here */ }
    // impedance, regulariser, reference acceleration (mj_makeImpedance restated)
    if (solref[0] > 0) solref[0] = std::max(solref[0], 2 * m.dt);       // refsafe
    real imp = impedance(solimp, pos);
    real dmax = std::min(std::max(solimp[1], MINIMP), MAXIMP);
    real K, B;
    if (solref[0] > 0) { K = 1 / std::max(MINVAL, dmax * dmax * solref[0] * solref[0] * solref[1] * solref[1]); B = 2 / std::max(MINVAL, dmax * solref[0]); }
    else { K = -solref[0] / std::max(MINVAL, dmax * dmax); B = -solref[1] / std::max(MINVAL, dmax); }
    for (int j = 0; j < r.dim; j++) {
      s->efc_type[i + j] = r.type; s->efc_id[i + j] = r.id; s->efc_dim[i + j] = r.dim; s->efc_pos[i + j] = pos;
      s->efc_R[i + j] = std::max(MINVAL, (1 - imp) * diag[j] / imp);
    }
    if (r.type == C_CONTACT && r.dim > 1) {
      Contact& c = s->con[r.id];
      s->efc_R[i + 1] = s->efc_R[i] / std::max(MINVAL, m.impratio);
      c.mu = c.friction[0] * std::sqrt(s->efc_R[i + 1] / s->efc_R[i]);
      for (int j = 2; j < r.dim; j++)
        s->efc_R[i + j] = std::max(MINVAL, s->efc_R[i + 1] * c.friction[0] * c.friction[0] / (c.friction[j - 1] * c.friction[j - 1]));
    }
    for (int j = 0; j < r.dim; j++) {
      real vel = 0;
      for (int d = 0; d < nv; d++) vel += s->J[(size_t)(i + j) * nv + d] * s->qvel[d];
      bool fric = (r.type == C_FRICTION) || (r.type == C_CONTACT && j > 0);
      s->efc_aref[i + j] = -B * vel - (fric ? 0.0 : K * imp * pos);
      s->efc_D[i + j] = 1 / s->efc_R[i + j];
    }
    i += r.dim;
  }
}

// QCQP:  min 0.5 x'Ax + b'x  s.t.  sum (x_i/d_i)^2 <= r^2   (mju_QCQP restated, n <= 5)
bool qcqp(real* res, const real* Ain, const real* bin, const real* d, real r, int n) {
  real A[25], b[5], P[25], y[5], z[5];
  for (int i = 0; i < n; i++) { b[i] = bin[i] * d[i]; for (int j = 0; j < n; j++) A[i * n + j] = Ain[i * n + j] * d[i] * d[j]; }
  real la = 0;
  for (int iter = 0; iter < 20; iter++) {
    for (int i = 0; i < n * n; i++) P[i] = A[i];
    for (int i = 0; i < n; i++) P[i * n + i] += la;
    // Cholesky in place (lower)
    bool ok = true;
    for (int j = 0; j < n && ok; j++) {
      real v = P[j * n + j];
      for (int k = 0; k < j; k++) v -= P[j * n + k] * P[j * n + k];
      if (v < MINVAL) { ok = false; break; }
      v = std::sqrt(v); P[j * n + j] = v;
      for (int i = j + 1; i < n; i++) {
        real w = P[i * n + j];
        for (int k = 0; k < j; k++) w -= P[i * n + k] * P[j * n + k];
        P[i * n + j] = w / v;
      }
    }
    if (!ok) { for (int i = 0; i < n; i++) res[i] = 0; return false; }
    auto solve = [&](real* out, const real* rhs) {
      real t[5];
      for (int i = 0; i < n; i++) { real v = rhs[i]; for (int k = 0; k < i; k++) v -= P[i * n + k] * t[k]; t[i] = v / P[i * n + i]; }
      for (int i = n - 1; i >= 0; i--) { real v = t[i]; for (int k = i + 1; k < n; k++) v -= P[k * n + i] * out[k]; out[i] = v / P[i * n + i]; }
    };
    real nb[5]; for (int i = 0; i < n; i++) nb[i] = -b[i];
    solve(y, nb);
    real val = -r * r; for (int i = 0; i < n; i++) val += y[i] * y[i];
    if (val < 1e-10) break;
    solve(z, y);
    real deriv = 0; for (int i = 0; i < n; i++) deriv += -2 * y[i] * z[i];
    real delta = -val / deriv;
    if (delta < 1e-10) break;
    la += delta;
  }
  for (int i = 0; i < n; i++) res[i] = y[i] * d[i];
  return la != 0;
}

// force from a primal acceleration (mj_constraintUpdate restated) — used for the PGS warm start
void constraint_update(orc_sim* s, const real* jar) {
  int i = 0;
  while (i < s->nefc) {
    int tp = s->efc_type[i], dim = s->efc_dim[i];
    if (tp == C_EQUALITY) {
      s->efc_force[i] = -s->efc_D[i] * jar[i];
    } else if (tp == C_FRICTION) {
      real f = -s->efc_D[i] * jar[i], fl = s->efc_floss[i];
      s->efc_force[i] = std::min(std::max(f, -fl), fl);
    } else if (tp == C_LIMIT || dim == 1) {
      s->efc_force[i] = jar[i] < 0 ? -s->efc_D[i] * jar[i] : 0;
    } else {
      const Contact& c = s->con[s->efc_id[i]];
      real mu = c.mu, U[6];
      U[0] = jar[i] * mu;
      real T = 0;
      for (int j = 1; j < dim; j++) { U[j] = jar[i + j] * c.friction[j - 1]; T += U[j] * U[j]; }
      T = std::sqrt(T);
      real N = U[0];
      if ((N >= mu * T) || (T <= 0 && N >= 0)) { for (int j = 0; j < dim; j++) s->efc_force[i + j] = 0; }
      else if ((mu * N + T <= 0) || (T <= 0 && N < 0)) { for (int j = 0; j < dim; j++) s->efc_force[i + j] = -s->efc_D[i + j] * jar[i + j]; }
      else {
        real Dm = s->efc_D[i] / std::max(mu * mu * (1 + mu * mu), MINVAL), NmT = N - mu * T;
        s->efc_force[i] = -Dm * NmT * mu;
        for (int j = 1; j < dim; j++) s->efc_force[i + j] = -s->efc_force[i] / T * U[j] * c.friction[j - 1];
      }
    }
    i += dim;
  }
}

void solve_pgs(orc_sim* s) {
  const Model& m = s->m;
  int nv = m.nv, n = s->nefc;
  s->solver_iter = 0;
  s->qacc = s->qacc_smooth;
  if (n == 0) return;
  // B = Minv J^T ; AR = J B + diag(R) ; b = J qacc_smooth - aref
  std::vector<real> Bm((size_t)nv * n, 0);
  for (int r = 0; r < n; r++) for (int d = 0; d < nv; d++) {
    real v = 0; for (int k = 0; k < nv; k++) v += s->Minv[d * nv + k] * s->J[(size_t)r * nv + k];
    Bm[(size_t)d * n + r] = v;
  }
  s->AR.assign((size_t)n * n, 0);
  for (int r = 0; r < n; r++) for (int c = 0; c < n; c++) {
    real v = 0; for (int d = 0; d < nv; d++) v += s->J[(size_t)r * nv + d] * Bm[(size_t)d * n + c];
    s->AR[(size_t)r * n + c] = v + (r == c ? s->efc_R[r] : 0);
  }
  for (int r = 0; r < n; r++) {
    real v = 0; for (int d = 0; d < nv; d++) v += s->J[(size_t)r * nv + d] * s->qacc_smooth[d];
    s->efc_b[r] = v - s->efc_aref[r];
  }
  // warm start from the previous qacc (mujoco warmstart(): map to forces, keep if dual cost < 0)
  {
    std::vector<real> jar(n);
    for (int r = 0; r < n; r++) {
      real v = 0; for (int d = 0; d < nv; d++) v += s->J[(size_t)r * nv + d] * s->warm[d];
      jar[r] = v - s->efc_aref[r];
    }
    constraint_update(s, jar.data());
    real cost = 0;
    for (int r = 0; r < n; r++) {
      real v = 0; for (int c = 0; c < n; c++) v += s->AR[(size_t)r * n + c] * s->efc_force[c];
      cost += s->efc_force[r] * (0.5 * v + s->efc_b[r]);
    }
    if (cost > 0) std::fill(s->efc_force.begin(), s->efc_force.end(), 0.0);
  }
  real scale = 1 / (m.meaninertia * std::max(1, nv));
  real* f = s->efc_force.data();
  for (int iter = 0; iter < s->iterations; iter++) {
    real improvement = 0;
    int i = 0;
    while (i < n) {
      int dim = s->efc_dim[i], tp = s->efc_type[i];
      real res[6], old[6], Athis[36];
      for (int j = 0; j < dim; j++) {
        real v = s->efc_b[i + j];
        for (int c = 0; c < n; c++) v += s->AR[(size_t)(i + j) * n + c] * f[c];
        res[j] = v; old[j] = f[i + j];
      }
      for (int j = 0; j < dim; j++) for (int k = 0; k < dim; k++) Athis[j * dim + k] = s->AR[(size_t)(i + j) * n + i + k];
      if (dim == 1) {
        f[i] -= res[0] / Athis[0];
        if (tp == C_FRICTION) { real fl = s->efc_floss[i]; f[i] = std::min(std::max(f[i], -fl), fl); }
        else if (tp != C_EQUALITY && f[i] < 0) f[i] = 0;
      } else {
        const Contact& c = s->con[s->efc_id[i]];
        // normal / ray update
        if (f[i] < MINVAL) {
          f[i] -= res[0] / Athis[0];
          if (f[i] < 0) f[i] = 0;
          for (int j = 1; j < dim; j++) f[i + j] = 0;
        } else {
          real v[6], v1[6], denom = 0, vr = 0;
          for (int j = 0; j < dim; j++) v[j] = f[i + j];
          for (int j = 0; j < dim; j++) { v1[j] = 0; for (int k = 0; k < dim; k++) v1[j] += Athis[j * dim + k] * v[k]; }
          for (int j = 0; j < dim; j++) { denom += v[j] * v1[j]; vr += v[j] * res[j]; }
          if (denom >= MINVAL) {
            real x = -vr / denom;
            if (f[i] + x * v[0] < 0) x = -f[i] / v[0];
            for (int j = 0; j < dim; j++) f[i + j] += x * v[j];
          }
        }
        // friction update with the normal fixed
        if (f[i] >= MINVAL) {
          int nf = dim - 1;
          real Ac[25], bc[5], v[5];
          for (int j = 0; j < nf; j++) {
            for (int k = 0; k < nf; k++) Ac[j * nf + k] = Athis[(j + 1) * dim + k + 1];
            bc[j] = res[j + 1];
            for (int k = 0; k < nf; k++) bc[j] -= Ac[j * nf + k] * old[k + 1];
            bc[j] += Athis[(j + 1) * dim] * (f[i] - old[0]);
          }
          bool active = qcqp(v, Ac, bc, c.friction, f[i], nf);
          if (active) {
            real ssq = 0; for (int j = 0; j < nf; j++) ssq += (v[j] / c.friction[j]) * (v[j] / c.friction[j]);
            real sc = std::sqrt(f[i] * f[i] / std::max(MINVAL, ssq));
            for (int j = 0; j < nf; j++) v[j] *= sc;
          }
          for (int j = 0; j < nf; j++) f[i + 1 + j] = v[j];
        }
      }
      // cost change; revert on increase
      real delta[6], change = 0;
      for (int j = 0; j < dim; j++) delta[j] = f[i + j] - old[j];
      for (int j = 0; j < dim; j++) {
        real v = 0; for (int k = 0; k < dim; k++) v += Athis[j * dim + k] * delta[k];
        change += delta[j] * (0.5 * v + res[j]);
      }
      if (change > 1e-10) { for (int j = 0; j < dim; j++) f[i + j] = old[j]; change = 0; }
      improvement -= change;
      i += dim;
    }
    s->solver_iter = iter + 1;
    if (improvement * scale < s->tolerance) break;
  }
  for (int d = 0; d < nv; d++) {
    real v = s->qacc_smooth[d];
    for (int r = 0; r < n; r++) v += Bm[(size_t)d * n + r] * f[r];
    s->qacc[d] = v;
  }
}


// ================================================================ Newton solver (mj_solNewton restated)
// Primal problem:  min_a  0.5 (a - a_s)' M (a - a_s) + sum_i s_i( (J a - aref)_i ),  a_s = qacc_smooth.
// Row costs s_i (mj_constraintUpdate): quadratic 0.5 D r^2 when active; dof frictionloss is Huber-like (quadratic
// inside |r| < R*floss, linear outside); elliptic contacts have three zones in (N, T) = (mu*r0, |mu_k r_k|):
// top (N >= mu T) free, bottom (mu N + T <= 0) fully quadratic, middle 0.5*Dm*(N - mu T)^2.
struct RowEval { real cost, g1, g2; };    // cost and its first/second derivative along a line

// cost, force (= -ds/dr) and, if Hc != nullptr, the dim x dim Hessian d2s/dr2 of one constraint block at r = jar
real block_cost(const orc_sim* s, int i, const real* jar, real* force, real* Hc) {
  int tp = s->efc_type[i], dim = s->efc_dim[i];
  if (Hc) for (int k = 0; k < dim * dim; k++) Hc[k] = 0;
  if (tp == C_EQUALITY) {                              // always active, either sign
    real D = s->efc_D[i], r = jar[0];
    force[0] = -D * r; if (Hc) Hc[0] = D; return 0.5 * D * r * r;
  }
  if (tp == C_FRICTION) {
    real D = s->efc_D[i], R = s->efc_R[i], fl = s->efc_floss[i], rf = R * fl, r = jar[0];
    if (r <= -rf) { force[0] = fl; return fl * (-0.5 * rf - r); }
    if (r >= rf) { force[0] = -fl; return fl * (-0.5 * rf + r); }
    force[0] = -D * r; if (Hc) Hc[0] = D; return 0.5 * D * r * r;
  }
  if (tp == C_LIMIT || dim == 1) {
    real D = s->efc_D[i], r = jar[0];
    if (r < 0) { force[0] = -D * r; if (Hc) Hc[0] = D; return 0.5 * D * r * r; }
    force[0] = 0; return 0;
  }
  const Contact& c = s->con[s->efc_id[i]];
  real mu = c.mu, U[6];
  U[0] = jar[0] * mu;
  real T = 0;
  for (int j = 1; j < dim; j++) { U[j] = jar[j] * c.friction[j - 1]; T += U[j] * U[j]; }
  T = std::sqrt(T);
  real N = U[0];
  if ((N >= mu * T) || (T <= 0 && N >= 0)) { for (int j = 0; j < dim; j++) force[j] = 0; return 0; }        // top
  if ((mu * N + T <= 0) || (T <= 0 && N < 0)) {                                                              // bottom
    real cost = 0;
    for (int j = 0; j < dim; j++) { real D = s->efc_D[i + j]; force[j] = -D * jar[j]; cost += 0.5 * D * jar[j] * jar[j]; if (Hc) Hc[j * dim + j] = D; }
    return cost;
  }
  real Dm = s->efc_D[i] / std::max(mu * mu * (1 + mu * mu), MINVAL), sN = N - mu * T;                        // middle
  force[0] = -Dm * sN * mu;
  for (int j = 1; j < dim; j++) force[j] = -force[0] / T * U[j] * c.friction[j - 1];
  if (Hc) {
    // d2/dr2 of 0.5*Dm*(mu r0 - mu T)^2 with T = |mu_k r_k|
    Hc[0] = Dm * mu * mu;
    for (int k = 1; k < dim; k++) {
      real mk = c.friction[k - 1];
      Hc[k] = Hc[k * dim] = -Dm * mu * mu * U[k] * mk / T;
      for (int l = 1; l < dim; l++) {
        real ml = c.friction[l - 1];
        Hc[k * dim + l] = Dm * mu * mu * mk * ml * U[k] * U[l] / (T * T)
                          - Dm * sN * mu * mk * ml * ((k == l ? 1.0 : 0.0) / T - U[k] * U[l] / (T * T * T));
      }
    }
  }
  return 0.5 * Dm * sN * sN;
}

void solve_newton(orc_sim* s) {
  const Model& m = s->m;
  int nv = m.nv, n = s->nefc;
  s->solver_iter = 0; s->ls_evals = 0;
  s->qacc = s->qacc_smooth;
  if (n == 0) return;
  const real* J = s->J.data();
  std::vector<real> qfrc_smooth(nv, 0);      // M a_s
  for (int r = 0; r < nv; r++) for (int c = 0; c < nv; c++) qfrc_smooth[r] += s->M[r * nv + c] * s->qacc_smooth[c];
  std::vector<real> jar(n), force(n), Ma(nv), grad(nv), search(nv), jv(n), H(nv * nv), Mv(nv);
  auto total_cost = [&](const std::vector<real>& a, bool want_grad_hess) {
    for (int r = 0; r < nv; r++) { Ma[r] = 0; for (int c = 0; c < nv; c++) Ma[r] += s->M[r * nv + c] * a[c]; }
    real cost = 0;
    for (int r = 0; r < nv; r++) cost += 0.5 * (Ma[r] - qfrc_smooth[r]) * (a[r] - s->qacc_smooth[r]);
    for (int i = 0; i < n; i++) { real v = -s->efc_aref[i]; for (int d = 0; d < nv; d++) v += J[(size_t)i * nv + d] * a[d]; jar[i] = v; }
    if (want_grad_hess) for (int k = 0; k < nv * nv; k++) H[k] = s->M[k];
    int i = 0;
    while (i < n) {
      int dim = s->efc_dim[i];
      real Hc[36];
      cost += block_cost(s, i, &jar[i], &force[i], want_grad_hess ? Hc : nullptr);
      if (want_grad_hess)
        for (int j = 0; j < dim; j++) for (int k = 0; k < dim; k++) {
          real h = Hc[j * dim + k];
          if (h == 0) continue;
          for (int a_ = 0; a_ < nv; a_++) { real ja = J[(size_t)(i + j) * nv + a_]; if (ja == 0) continue; for (int b_ = 0; b_ < nv; b_++) H[a_ * nv + b_] += ja * h * J[(size_t)(i + k) * nv + b_]; }
        }
      i += dim;
    }
    if (want_grad_hess)
      for (int r = 0; r < nv; r++) { real g = Ma[r] - qfrc_smooth[r]; for (int i2 = 0; i2 < n; i2++) g -= J[(size_t)i2 * nv + r] * force[i2]; grad[r] = g; }
    return cost;
  };
  // warm start: the better of qacc_warmstart and qacc_smooth
  std::vector<real> a = s->warm;
  real cw = total_cost(a, false), cs = total_cost(s->qacc_smooth, false);
  if (!(cw < cs)) a = s->qacc_smooth;
  real scale = 1 / (m.meaninertia * std::max(1, nv));
  real cost = total_cost(a, true);
  for (int iter = 0; iter < s->iterations; iter++) {
    // search = -H^-1 grad (Cholesky)
    std::vector<real> Lc(H);
    for (int j = 0; j < nv; j++) {
      for (int k = 0; k < j; k++) for (int i = j; i < nv; i++) Lc[i * nv + j] -= Lc[i * nv + k] * Lc[j * nv + k];
      real d = std::sqrt(std::max(Lc[j * nv + j], MINVAL));
      for (int i = j; i < nv; i++) Lc[i * nv + j] /= d;
    }
    std::vector<real> y(nv);
    for (int i = 0; i < nv; i++) { real v = -grad[i]; for (int k = 0; k < i; k++) v -= Lc[i * nv + k] * y[k]; y[i] = v / Lc[i * nv + i]; }
    for (int i = nv - 1; i >= 0; i--) { real v = y[i]; for (int k = i + 1; k < nv; k++) v -= Lc[k * nv + i] * search[k]; search[i] = v / Lc[i * nv + i]; }
    // exact line search on phi(alpha) = cost(a + alpha*search): safeguarded Newton on phi'
    for (int i = 0; i < n; i++) { real v = 0; for (int d = 0; d < nv; d++) v += J[(size_t)i * nv + d] * search[d]; jv[i] = v; }
    for (int r = 0; r < nv; r++) { Mv[r] = 0; for (int c = 0; c < nv; c++) Mv[r] += s->M[r * nv + c] * search[c]; }
    real q1 = 0, q2 = 0;                       // Gauss part: phi_g = c0 + q1 alpha + 0.5 q2 alpha^2
    for (int r = 0; r < nv; r++) { q1 += search[r] * (Ma[r] - qfrc_smooth[r]); q2 += search[r] * Mv[r]; }
    std::vector<real> jar0(jar);
    auto dphi = [&](real alpha, real* d2) {
      s->ls_evals++;
      real d1 = q1 + q2 * alpha; *d2 = q2;
      int i = 0;
      while (i < n) {
        int dim = s->efc_dim[i];
        real r6[6], f6[6], Hc[36];
        for (int j = 0; j < dim; j++) r6[j] = jar0[i + j] + alpha * jv[i + j];
        block_cost(s, i, r6, f6, Hc);
        for (int j = 0; j < dim; j++) { d1 -= f6[j] * jv[i + j]; for (int k = 0; k < dim; k++) *d2 += jv[i + j] * Hc[j * dim + k] * jv[i + k]; }
        i += dim;
      }
      return d1;
    };
    real lo = 0, hi = -1, alpha = 0, d2, d1 = dphi(0, &d2), d10 = std::fabs(d1);
    if (d1 < 0) {
      for (int ls = 0; ls < 60; ls++) {
        real step = -d1 / std::max(d2, MINVAL), cand = alpha + step;
        if (hi > 0 && (cand <= lo || cand >= hi)) cand = 0.5 * (lo + hi);
        alpha = cand;
        d1 = dphi(alpha, &d2);
        if (std::fabs(d1) <= 1e-12 * std::max(d10, MINVAL)) break;
        if (d1 < 0) lo = alpha; else hi = alpha;
        if (hi > 0 && hi - lo < 1e-15 * std::max(1.0, hi)) break;
      }
    }
    for (int r = 0; r < nv; r++) a[r] += alpha * search[r];
    real newcost = total_cost(a, true);
    real improvement = scale * (cost - newcost), gnorm = 0;
    for (int r = 0; r < nv; r++) gnorm += grad[r] * grad[r];
    gnorm = scale * std::sqrt(gnorm);
    cost = newcost;
    s->solver_iter = iter + 1;
    if (improvement < s->tolerance || gnorm < s->tolerance) break;
  }
  s->qacc = a;
  total_cost(a, false);
  s->efc_force = force;
}

void forward(orc_sim* s, bool freeze_arm) {
  const Model& m = s->m;
  int nv = m.nv;
  kinematics(s);
  crba(s);
  rne_bias(s);
  actuation(s);
  s->qacc_smooth.assign(nv, 0);
  for (int r = 0; r < nv; r++) {
    real v = 0;
    for (int c = 0; c < nv; c++)        // qfrc_passive = -damping * qvel (zero in the SO100 model)
      v += s->Minv[r * nv + c] * (s->qfrc_act[c] - s->bias[c] - m.dof_damping[c] * s->qvel[c]);
    s->qacc_smooth[r] = v;
  }
  if (s->injected.empty()) collision(s);
  else {                                   // solver-parity tests: the caller's contact list instead of the narrowphase
    s->con.clear();
    for (const Contact& c0 : s->injected) {
      Contact c = c0;
      make_frame(c.frame);
      set_contact_params(m, c, c.g1, c.g2);
      s->con.push_back(c);
    }
  }
  make_constraints(s, freeze_arm);
  if (s->solver == 1) solve_newton(s); else solve_pgs(s);
}

void euler(orc_sim* s) {
  const Model& m = s->m;
  real dt = m.dt;
  std::vector<real> acc(s->qacc);
  if (m.any_damping) {
    // mj_Euler with joint damping (eulerdamp enabled by default): damping implicit in velocity,
    // qacc' = (M + h diag(damping))^-1 (qfrc_smooth + qfrc_constraint) = (M + h D)^-1 M qacc
    int nv = m.nv;
    std::vector<real> A(s->M), rhs(nv, 0);
    for (int r = 0; r < nv; r++) { for (int c = 0; c < nv; c++) rhs[r] += s->M[r * nv + c] * s->qacc[c]; A[r * nv + r] += dt * m.dof_damping[r]; }
    for (int j = 0; j < nv; j++) {
      for (int k = 0; k < j; k++) for (int i = j; i < nv; i++) A[i * nv + j] -= A[i * nv + k] * A[j * nv + k];
      real d = std::sqrt(A[j * nv + j]);
      for (int i = j; i < nv; i++) A[i * nv + j] /= d;
    }
    for (int i = 0; i < nv; i++) { real v = rhs[i]; for (int k = 0; k < i; k++) v -= A[i * nv + k] * rhs[k]; rhs[i] = v / A[i * nv + i]; }
    for (int i = nv - 1; i >= 0; i--) { real v = rhs[i]; for (int k = i + 1; k < nv; k++) v -= A[k * nv + i] * acc[k]; acc[i] = v / A[i * nv + i]; }
  }
  for (int d = 0; d < m.nv; d++) s->qvel[d] += dt * acc[d];
  for (int b = 1; b < m.nbody; b++) {
    int qa = m.body_qposadr[b], d = m.body_dofadr[b];
    if (m.body_jnttype[b] == J_HINGE || m.body_jnttype[b] == J_SLIDE) s->qpos[qa] += dt * s->qvel[d];
    else if (m.body_jnttype[b] == J_FREE) {
      for (int k = 0; k < 3; k++) s->qpos[qa + k] += dt * s->qvel[d + k];
      real w[3] = {s->qvel[d + 3], s->qvel[d + 4], s->qvel[d + 5]};
      real ang = dt * normalize3(w);           // mju_quatIntegrate
      real sn = std::sin(0.5 * ang), dq[4] = {std::cos(0.5 * ang), w[0] * sn, w[1] * sn, w[2] * sn};
      mulquat(&s->qpos[qa + 3], &s->qpos[qa + 3], dq);
      normquat(&s->qpos[qa + 3]);
    }
  }
  s->warm = s->qacc;
}

// mj_checkPos/Vel/Acc: NaN or |x| > mjMAXVAL (1e10) => divergence; MuJoCo resets the data (qpos0), and
// dm_control with raise_exception_on_physics_error=False (so101_sim/task_suite.py:153) ends the episode
// with reward 0 / discount 0.
bool diverged_state(orc_sim* s) {
  const Model& m = s->m;
  bool bad = false;
  for (real x : s->qpos) bad = bad || !(std::fabs(x) <= 1e10);
  for (real x : s->qvel) bad = bad || !(std::fabs(x) <= 1e10);
  for (real x : s->qacc) bad = bad || !(std::fabs(x) <= 1e10);
  if (bad) {
    std::fill(s->qpos.begin(), s->qpos.end(), 0.0); std::fill(s->qvel.begin(), s->qvel.end(), 0.0);
    std::fill(s->warm.begin(), s->warm.end(), 0.0); std::fill(s->qacc.begin(), s->qacc.end(), 0.0);
    for (int f = 0; f < m.nfree; f++) s->qpos[m.body_qposadr[m.free_body[f]] + 3] = 1;
  }
  return bad;
}

bool substeps(orc_sim* s, int nsub, bool freeze_arm) {
  const Model& m = s->m;
  std::vector<real> q0, v0;
  if (freeze_arm) { q0.assign(s->qpos.begin(), s->qpos.begin() + m.narm); v0.assign(s->qvel.begin(), s->qvel.begin() + m.narm); }
  for (int k = 0; k < nsub; k++) {
    forward(s, freeze_arm);
    euler(s);
    if (diverged_state(s)) return true;
    if (freeze_arm) {   // dm_control JointStaticIsolator: non-prop joints restored after every step
      std::copy(q0.begin(), q0.end(), s->qpos.begin()); std::copy(v0.begin(), v0.end(), s->qvel.begin());
    }
  }
  return false;
}

// ================================================================ reward (so100_hand_over.py:238-275)
struct Box { real pos[3], quat[4], half[3]; };

bool overlap_aabb_oobb(const real* half0, const Box& b) {   // oobb_utils.overlap_aabb_oobb (:202-248)
  real av[8][3], ov[8][3];
  for (int i = 0; i < 8; i++) {
    int iz = i / 4, ixy = i % 4, ix = ixy % 2, iy = ixy / 2;
    real t[3] = {ix ? 1.0 : 0.0, iy ? 1.0 : 0.0, iz ? 1.0 : 0.0};
    real loc[3];
    for (int k = 0; k < 3; k++) { av[i][k] = -half0[k] * (1 - t[k]) + half0[k] * t[k]; loc[k] = -b.half[k] * (1 - t[k]) + b.half[k] * t[k]; }
    real r[3]; rotvecquat(r, loc, b.quat);
    for (int k = 0; k < 3; k++) ov[i][k] = b.pos[k] + r[k];
  }
  real axes[6][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int k = 0; k < 3; k++) { real e[3] = {0, 0, 0}; e[k] = 1; rotvecquat(axes[3 + k], e, b.quat); }
  for (int a = 0; a < 6; a++) {
    real mx0 = -1e300, mn0 = 1e300, mx1 = -1e300, mn1 = 1e300;
    for (int i = 0; i < 8; i++) {
      real p0 = dot3(av[i], axes[a]), p1 = dot3(ov[i], axes[a]);
      mx0 = std::max(mx0, p0); mn0 = std::min(mn0, p0); mx1 = std::max(mx1, p1); mn1 = std::min(mn1, p1);
    }
    if (mx0 < mn1 || mn0 > mx1) return false;     // strict: touching counts as overlap
  }
  return true;
}

bool overlap_oobb_oobb(const Box& b0, const Box& b1) {       // oobb_utils.overlap_oobb_oobb (:251-273)
  real inv[4] = {b0.quat[0], -b0.quat[1], -b0.quat[2], -b0.quat[3]};
  real dp[3] = {b1.pos[0] - b0.pos[0], b1.pos[1] - b0.pos[1], b1.pos[2] - b0.pos[2]};
  Box r;
  rotvecquat(r.pos, dp, inv);
  mulquat(r.quat, inv, b1.quat);
  std::memcpy(r.half, b1.half, sizeof r.half);
  return overlap_aabb_oobb(b0.half, r);
}

real reward(orc_sim* s) {
  const Model& m = s->m;
  kinematics(s);
  // any_props_moving: max |linear velocity| >= 1e-3 for either prop (success_detector_utils.py:22-28)
  int props[2] = {m.obj_body, m.con_body};
  for (int p = 0; p < 2; p++) {
    int d = m.body_dofadr[props[p]];
    real mx = std::max(std::fabs(s->qvel[d]), std::max(std::fabs(s->qvel[d + 1]), std::fabs(s->qvel[d + 2])));
    if (mx >= 1e-3) return 0.0;
  }
  // object: one box = (xipos, ximat -> quat, bvh root aabb) since the prop body has >1 geom
  Box ob;
  int b = m.obj_body;
  mat2quat(ob.quat, &s->ximat[9 * b]);
  real ctr[3]; rotvecquat(ctr, &m.body_bvh_aabb[6 * b], ob.quat);
  for (int k = 0; k < 3; k++) { ob.pos[k] = ctr[k] + s->xipos[3 * b + k]; ob.half[k] = m.body_bvh_aabb[6 * b + 3 + k]; }
  int cb = m.con_body;
  const real* cpos = &s->xpos[3 * cb]; const real* cq = &s->xquat[4 * cb];
  for (int k = 0; k < m.nbox; k++) {
    Box cw;                                       // transform_oobb (:175-199)
    real r[3]; rotvecquat(r, &m.box_pos[3 * k], cq);
    for (int j = 0; j < 3; j++) { cw.pos[j] = cpos[j] + r[j]; cw.half[j] = m.box_half[3 * k + j]; }
    real ident[4] = {1, 0, 0, 0};
    mulquat(cw.quat, cq, ident);
    if (!overlap_oobb_oobb(ob, cw)) return 0.0;
  }
  return 1.0;
}

// ================================================================ counter RNG (Philox4x32-10)
void philox(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
double rng_uniform(uint64_t seed, uint64_t env, uint64_t episode, uint32_t draw) {
  uint32_t c[4] = {(uint32_t)env, (uint32_t)(env >> 32), (uint32_t)episode, draw};
  philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return (double)(c[0] >> 8) * (1.0 / 16777216.0);   // 24-bit mantissa: exact in f32 and f64
}

void env_obs(const orc_sim* s, real* obs) {
  // joints_pos (delayed 5 control steps) | undelayed_joints_pos | commanded_joints_pos
  for (int k = 0; k < 6; k++) { obs[k] = s->delayed[k]; obs[6 + k] = s->qpos[k]; obs[12 + k] = s->cmd[k]; }
}

void env_reset(orc_sim* s) {
  const Model& m = s->m;
  uint32_t draw = 0;
  auto U = [&](real lo, real hi) { real u = rng_uniform(s->cfg.seed, s->cfg.env_id, s->episode, draw++); return lo + u * (hi - lo); };
  std::fill(s->qpos.begin(), s->qpos.end(), 0.0); std::fill(s->qvel.begin(), s->qvel.end(), 0.0);
  std::fill(s->warm.begin(), s->warm.end(), 0.0);
  if (!m.home_qpos.empty()) {                       // ALOHA: arms at HOME_QPOS, ctrl = HOME_CTRL (aloha2_task.py:369-380)
    for (int k = 0; k < m.narm; k++) s->qpos[k] = m.home_qpos[k];
    for (int k = 0; k < m.nu; k++) s->ctrl[k] = m.home_ctrl[k];
  } else
  for (int k = 0; k < 6; k++) { s->ctrl[k] = m.home_ctrl[k] + s->cfg.offsets[k]; s->cmd[k] = s->ctrl[k]; }
  int qo = m.body_qposadr[m.obj_body], qc = m.body_qposadr[m.con_body];
  if (m.task_kind == 1) {
    // Dining (dining.py:162-267): six region samples (three uniforms each), the top three regions shuffled among plate / bowl / container
    // and the bottom three among mug / pen / banana (counter RNG: one of the six orders each; a single env of the product draws numpy's
    // shuffle on the host instead), a uniform yaw per prop in placer order; collisions ignored, then settled
    real smp[6][3];
    for (int r = 0; r < 6; r++) for (int k = 0; k < 3; k++) smp[r][k] = U(m.region_lo[3 * r + k], m.region_hi[3 * r + k]);
    static const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    // (the index is taken from the 24-bit draw in single precision, as the kernels do: u * 6 rounds identically)
    int pt = (int)((float)U(0, 1) * 6.0f), pb = (int)((float)U(0, 1) * 6.0f);
    pt = std::min(pt, 5); pb = std::min(pb, 5);
    for (int p = 0; p < 6; p++) {
      int region = p < 3 ? perm[pt][p] : 3 + perm[pb][p - 3], qa = m.body_qposadr[m.prop_bodies[p]];
      real yaw = U(m.obj_yaw[0], m.obj_yaw[1]);
      for (int k = 0; k < 3; k++) s->qpos[qa + k] = smp[region][k];
      s->qpos[qa + 3] = std::cos(0.5 * yaw); s->qpos[qa + 4] = 0; s->qpos[qa + 5] = 0; s->qpos[qa + 6] = std::sin(0.5 * yaw);
    }
  } else {
  // object: position then yaw (so100_hand_over.py:209-214), collisions ignored
  for (int k = 0; k < 3; k++) s->qpos[qo + k] = U(m.obj_lo[k], m.obj_hi[k]);
  real yaw = U(m.obj_yaw[0], m.obj_yaw[1]);
  s->qpos[qo + 3] = std::cos(0.5 * yaw); s->qpos[qo + 4] = 0; s->qpos[qo + 5] = 0; s->qpos[qo + 6] = std::sin(0.5 * yaw);
  s->qpos[qc + 3] = 1;
  }
  // container: rejection-sampled until none of its geoms is in penetrating contact (<=20 tries)
  for (int attempt = 0; attempt < (m.task_kind == 1 ? 0 : 20); attempt++) {
    for (int k = 0; k < 3; k++) s->qpos[qc + k] = U(m.con_lo[k], m.con_hi[k]);
    kinematics(s); collision(s);
    bool hit = false;
    for (auto& c : s->con) if (m.geom_body[c.g1] == m.con_body || m.geom_body[c.g2] == m.con_body) hit = true;
    if (!hit) break;
  }
  // settle with the arm restored after every substep, until |qvel|<1e-3 and |qacc|<1e-2 on the props
  for (int k = 0; k < s->cfg.settle_max_substeps; k++) {
    substeps(s, 1, true);
    real mv = 0, ma = 0;
    for (int d = m.narm; d < m.nv; d++) { mv = std::max(mv, std::fabs(s->qvel[d])); ma = std::max(ma, std::fabs(s->qacc[d])); }
    if (mv < 1e-3 && ma < 1e-2) break;
  }
  for (int r = 0; r < 5; r++) for (int k = 0; k < 6; k++) s->ring[r][k] = s->qpos[k];   // INITIAL_VALUE padding
  for (int k = 0; k < 6; k++) s->delayed[k] = s->qpos[k];
  s->ring_head = 0; s->step_count = 0; s->ep_return = 0; s->need_reset = false; s->episode++;
}

}  // namespace

// ==================================================================== C API
extern "C" {

orc_sim* orc_create(const void* blob, size_t bytes) {
  Blob b;
  if (!b.parse(blob, bytes)) return nullptr;
  orc_sim* s = new orc_sim();
  Model& m = s->m;
  m.nq = b.i("nq"); m.nv = b.i("nv"); m.nu = b.i("nu"); m.nbody = b.i("nbody"); m.ngeom = b.i("ngeom");
  m.nvert = b.i("nvert"); m.npair = b.i("npair"); m.narm = b.i("narm"); m.nfree = b.i("nfree");
  m.dt = b.r("opt_timestep"); auto g = b.R("opt_gravity"); for (int k = 0; k < 3; k++) m.gravity[k] = g[k];
  m.impratio = b.r("opt_impratio"); m.tolerance = b.r("opt_tolerance"); m.iterations = b.i("opt_iterations");
  m.elliptic = b.i("opt_cone_elliptic"); m.mpr_tol = b.r("opt_mpr_tolerance"); m.mpr_iter = b.i("opt_mpr_iterations");
  m.meaninertia = b.r("stat_meaninertia");
#define LI(x) m.x = b.I(#x)
#define LR(x) m.x = b.R(#x)
  LI(body_parent); LI(body_jnttype); LI(body_qposadr); LI(body_dofadr); LI(body_weldid); LI(arm_body); LI(free_body);
  LI(jnt_limited); LI(dof_body); LI(act_dof); LI(act_ctrllimited); LI(act_forcelimited); LI(geom_type); LI(geom_body);
  LI(geom_condim); LI(geom_priority); LI(geom_vertadr); LI(geom_vertnum); LI(pair_geom);
  LR(body_pos); LR(body_quat); LR(body_ipos); LR(body_iquat); LR(body_mass); LR(body_inertia); LR(body_invweight0);
  LR(body_bvh_aabb); LR(jnt_axis); LR(jnt_range); LR(jnt_solref); LR(jnt_solimp); LR(dof_solref); LR(dof_solimp);
  LR(dof_armature); LR(dof_frictionloss); LR(dof_damping); LR(dof_invweight0); LR(act_gain); LR(act_bias);
  LR(act_ctrlrange); LR(act_forcerange); LR(geom_pos); LR(geom_quat); LR(geom_size); LR(geom_friction); LR(geom_solref);
  LR(geom_solimp); LR(geom_solmix); LR(geom_margin); LR(geom_gap); LR(geom_rbound); LR(geom_center); LR(geom_aabb);
  LR(mesh_vert);
#undef LI
#undef LR
  m.body_hinge.assign(m.nbody, -1);
  for (int h = 0; h < m.narm; h++) m.body_hinge[m.arm_body[h]] = h;
  if (b.has("neq")) {                               // general-tree scene (ALOHA)
    m.neq = b.i("neq");
    m.jnt_actfrclimited = b.I("jnt_actfrclimited"); m.jnt_actfrcrange = b.R("jnt_actfrcrange");
    m.eq_dof = b.I("eq_dof"); m.eq_qposadr = b.I("eq_qposadr"); m.eq_polycoef = b.R("eq_polycoef");
    m.eq_solref = b.R("eq_solref"); m.eq_solimp = b.R("eq_solimp");
    if (b.has("task_home_qpos")) m.home_qpos = b.R("task_home_qpos");
  }
  for (real d : m.dof_damping) m.any_damping = m.any_damping || d > 0;
  m.obj_body = b.i("task_object_body"); m.con_body = b.i("task_container_body"); m.nbox = b.i("task_nbox");
  m.box_pos = b.R("task_box_pos"); m.box_half = b.R("task_box_half"); m.obj_lo = b.R("task_obj_pos_lo");
  m.obj_hi = b.R("task_obj_pos_hi"); m.obj_yaw = b.R("task_obj_yaw"); m.con_lo = b.R("task_con_pos_lo");
  m.con_hi = b.R("task_con_pos_hi"); m.home_ctrl = b.R("task_home_ctrl");
  if (b.has("task_kind")) m.task_kind = b.i("task_kind");
  if (m.task_kind == 1) { m.prop_bodies = b.I("task_prop_bodies"); m.region_lo = b.R("task_region_lo"); m.region_hi = b.R("task_region_hi"); }
  s->qpos.assign(m.nq, 0); s->qvel.assign(m.nv, 0); s->ctrl.assign(m.nu, 0); s->warm.assign(m.nv, 0);
  s->qacc.assign(m.nv, 0); s->qacc_smooth.assign(m.nv, 0);
  for (int f = 0; f < m.nfree; f++) s->qpos[m.body_qposadr[m.free_body[f]] + 3] = 1;
  s->iterations = m.iterations; s->tolerance = m.tolerance;
  s->cfg.settle_max_substeps = 1000; s->cfg.last_step = 1 << 30;
  std::memset(s->ring, 0, sizeof s->ring); std::memset(s->cmd, 0, sizeof s->cmd); std::memset(s->delayed, 0, sizeof s->delayed);
  return s;
}
void orc_destroy(orc_sim* s) { delete s; }
int orc_nq(const orc_sim* s) { return s->m.nq; }
int orc_nv(const orc_sim* s) { return s->m.nv; }
int orc_nu(const orc_sim* s) { return s->m.nu; }
void orc_set_solver(orc_sim* s, int it, double tol) { if (it > 0) s->iterations = it; if (tol >= 0) s->tolerance = tol; }
void orc_set_collision(orc_sim* s, int e) { s->collide = e != 0; }
void orc_inject_contacts(orc_sim* s, int n, const double* rows) {     // rows [n][9]: pos3 normal3 dist geom1 geom2; n = 0 clears
  s->injected.clear();
  for (int k = 0; k < n; k++) {
    const double* r = rows + 9 * k;
    Contact c{};
    for (int i = 0; i < 3; i++) { c.pos[i] = r[i]; c.frame[i] = r[3 + i]; }
    normalize3(c.frame);
    c.dist = r[6]; c.g1 = (int)r[7]; c.g2 = (int)r[8];
    s->injected.push_back(c);
  }
}
void orc_set_mass_scale(orc_sim* s, const double* scale) {
  Model& m = s->m;
  if (s->mass0.empty()) { s->mass0 = m.body_mass; s->inertia0 = m.body_inertia; s->invweight0 = m.body_invweight0; }
  for (int f = 0; f < m.nfree; f++) {
    int b = m.free_body[f];
    m.body_mass[b] = s->mass0[b] * scale[f];
    for (int k = 0; k < 3; k++) m.body_inertia[3 * b + k] = s->inertia0[3 * b + k] * scale[f];
    for (int k = 0; k < 2; k++) m.body_invweight0[2 * b + k] = s->invweight0[2 * b + k] / scale[f];
  }
}
void orc_set_solver_type(orc_sim* s, int t) { s->solver = t; }
void orc_set_narrowphase(orc_sim* s, int mode) { s->narrow = mode; }
void orc_set_hull_multicontact(orc_sim* s, int on) { s->hull_multi = on != 0; }
void orc_set_contact_capacity(orc_sim* s, int cap) { s->contact_capacity = cap; }
int orc_contacts_reduced(const orc_sim* s) { return (s->contacts_reduced ? 1 : 0) | (s->contacts_cut ? 2 : 0); }
int orc_epa_iterations(const orc_sim* s) { return s->epa_iters; }
int orc_ls_evals(const orc_sim* s) { return s->ls_evals; }
void orc_set_state(orc_sim* s, const double* q, const double* v, const double* w) {
  if (q) std::copy(q, q + s->m.nq, s->qpos.begin());
  if (v) std::copy(v, v + s->m.nv, s->qvel.begin());
  if (w) std::copy(w, w + s->m.nv, s->warm.begin());
}
void orc_get_state(const orc_sim* s, double* q, double* v, double* w) {
  if (q) std::copy(s->qpos.begin(), s->qpos.end(), q);
  if (v) std::copy(s->qvel.begin(), s->qvel.end(), v);
  if (w) std::copy(s->warm.begin(), s->warm.end(), w);
}
void orc_set_ctrl(orc_sim* s, const double* c) { std::copy(c, c + s->m.nu, s->ctrl.begin()); }
int orc_substeps(orc_sim* s, int n, int fz) { return substeps(s, n, fz != 0) ? 1 : 0; }
void orc_forward(orc_sim* s, int fz) { forward(s, fz != 0); }
int orc_ncon(const orc_sim* s) { return (int)s->con.size(); }
int orc_nefc(const orc_sim* s) { return s->nefc; }
int orc_solver_iter(const orc_sim* s) { return s->solver_iter; }
void orc_get_M(const orc_sim* s, double* M) { std::copy(s->M.begin(), s->M.end(), M); }
void orc_get_bias(const orc_sim* s, double* b) { std::copy(s->bias.begin(), s->bias.end(), b); }
void orc_get_qacc(const orc_sim* s, double* a, double* as) {
  if (a) std::copy(s->qacc.begin(), s->qacc.end(), a);
  if (as) std::copy(s->qacc_smooth.begin(), s->qacc_smooth.end(), as);
}
void orc_get_actuator_force(const orc_sim* s, double* f) { std::copy(s->act_force.begin(), s->act_force.end(), f); }
void orc_get_contact(const orc_sim* s, int k, double* o) {
  const Contact& c = s->con[k];
  for (int i = 0; i < 3; i++) { o[i] = c.pos[i]; o[3 + i] = c.frame[i]; }
  o[6] = c.dist; o[7] = c.g1; o[8] = c.g2; o[9] = c.dim;
}
void orc_get_efc_force(const orc_sim* s, double* f) { std::copy(s->efc_force.begin(), s->efc_force.end(), f); }
void orc_get_efc_array(const orc_sim* s, int which, double* out) {
  if (which == 0) std::copy(s->efc_R.begin(), s->efc_R.end(), out);
  else if (which == 1) std::copy(s->efc_aref.begin(), s->efc_aref.end(), out);
  else if (which == 2) for (size_t i = 0; i < s->efc_type.size(); i++) out[i] = s->efc_type[i];
  else if (which == 3) std::copy(s->J.begin(), s->J.end(), out);
}
void orc_get_body_pose(const orc_sim* s, int b, double* p, double* q) {
  for (int k = 0; k < 3; k++) p[k] = s->xpos[3 * b + k];
  for (int k = 0; k < 4; k++) q[k] = s->xquat[4 * b + k];
}
double orc_max_prop_qacc(const orc_sim* s) {
  double ma = 0; for (int d = s->m.narm; d < s->m.nv; d++) ma = std::max(ma, std::fabs(s->qacc[d])); return ma;
}
double orc_reward(orc_sim* s) { return reward(s); }
int orc_overlap_oobb(const double* a, const double* b) {
  Box b0, b1;
  std::memcpy(b0.pos, a, 24); std::memcpy(b0.quat, a + 3, 32); std::memcpy(b0.half, a + 7, 24);
  std::memcpy(b1.pos, b, 24); std::memcpy(b1.quat, b + 3, 32); std::memcpy(b1.half, b + 7, 24);
  return overlap_oobb_oobb(b0, b1) ? 1 : 0;
}
void orc_env_config(orc_sim* s, const orc_env_cfg* c) { s->cfg = *c; s->need_reset = true; s->episode = 0; }
void orc_env_reset(orc_sim* s) { env_reset(s); }
void orc_env_begin(orc_sim* s) {
  for (int k = 0; k < 6; k++) { s->ctrl[k] = s->m.home_ctrl[k] + s->cfg.offsets[k]; s->cmd[k] = s->ctrl[k]; s->delayed[k] = s->qpos[k]; }
  for (int r = 0; r < 5; r++) for (int k = 0; k < 6; k++) s->ring[r][k] = s->qpos[k];
  s->ring_head = 0; s->step_count = 0; s->ep_return = 0; s->need_reset = false;
}
void orc_env_obs(const orc_sim* s, double* o) { env_obs(s, o); }
int orc_env_step_count(const orc_sim* s) { return s->step_count; }
double orc_env_return(const orc_sim* s) { return s->ep_return; }
void orc_env_step(orc_sim* s, const double* action, double* obs, double* rew, double* disc, int* st) {
  if (s->need_reset) {     // first call, or the call after LAST: reset and return FIRST
    env_reset(s);
    env_obs(s, obs); *rew = 0; *disc = 1; *st = 0;
    return;
  }
  for (int k = 0; k < 6; k++) { s->ctrl[k] = action[k] + s->cfg.offsets[k]; s->cmd[k] = s->ctrl[k]; }
  bool diverged = substeps(s, 10, false);
  s->step_count++;
  // delay ring (50 physics steps = 5 control steps): read the value of step k-5, then store step k
  for (int k = 0; k < 6; k++) { s->delayed[k] = s->ring[s->ring_head][k]; s->ring[s->ring_head][k] = s->qpos[k]; }
  s->ring_head = (s->ring_head + 1) % 5;
  real r = diverged ? 0.0 : reward(s);
  bool success = r >= 1.0 || diverged, timeout = s->step_count >= s->cfg.last_step;
  *rew = r; *disc = success ? 0.0 : 1.0; *st = (success || timeout) ? 2 : 1;
  s->ep_return += r;
  env_obs(s, obs);
  if (*st == 2) s->need_reset = true;
}
double orc_rng_uniform(uint64_t seed, uint64_t env, uint64_t ep, uint32_t draw) { return rng_uniform(seed, env, ep, draw); }

// ---- bench.py's cpu_baseline leg: the handover workload on host threads (one orc_sim per env, envs_per_thread envs per
// thread, every thread steps its envs one after the other - a CPU simulator's natural batching).  Phase 0 (timed on its
// own): placement + settle of every env.  Then, per solver setting, `reps` repetitions of `steps` control steps of
// every env with uniform random actions over the action spec (so100_task.py:232-251), all threads released together
// and the repetition's time taken when the last thread is done.  Returns the env-steps of one repetition.
namespace {
struct BenchBarrier {
  std::mutex mu; std::condition_variable cv; int n, waiting = 0; unsigned long gen = 0;
  explicit BenchBarrier(int n_) : n(n_) {}
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    unsigned long g = gen;
    if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); } else cv.wait(lk, [&] { return gen != g; });
  }
};
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

long long orc_bench_rollout(const void* blob, size_t bytes, int n_threads, int envs_per_thread, int steps, int reps, int n_settings,
                            const int* iterations, const double* tolerance, uint64_t seed, double* reset_seconds,
                            double* rep_seconds /*[n_settings][reps]*/) {
  if (n_threads <= 0 || envs_per_thread <= 0 || steps <= 0 || reps <= 0 || n_settings <= 0) return -1;
  static const double lo[6] = {-3.14159265358979, -3.14158, -3.14158, -3.14158, -3.14158, 0.0};
  static const double hi[6] = {3.14159265358979, 3.14158, 3.14158, 3.14158, 3.14158, 0.08};
  BenchBarrier bar(n_threads + 1);
  std::vector<int> failed(n_threads, 0);
  auto worker = [&](int t) {
    std::vector<orc_sim*> envs;
    for (int i = 0; i < envs_per_thread; i++) {
      orc_sim* s = orc_create(blob, bytes);
      if (!s) { failed[t] = 1; break; }
      orc_env_cfg c{};
      c.last_step = 500; c.settle_max_substeps = 1000; c.seed = seed; c.env_id = (uint64_t)t * envs_per_thread + i;
      orc_env_config(s, &c);
      orc_set_solver_type(s, 1);
      envs.push_back(s);
    }
    uint64_t lcg = 0x9E3779B97F4A7C15ull * (uint64_t)(t + 1) + seed;
    auto uni = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (double)(lcg >> 11) * (1.0 / 9007199254740992.0); };
    bar.wait();                                   // ---- reset phase
    for (orc_sim* s : envs) orc_env_reset(s);
    bar.wait();
    for (int k = 0; k < n_settings; k++) {
      for (orc_sim* s : envs) orc_set_solver(s, iterations[k], tolerance[k]);
      for (int r = 0; r < reps; r++) {
        bar.wait();
        for (int i = 0; i < steps; i++)
          for (orc_sim* s : envs) {
            double a[6], obs[18], rew, disc; int st;
            for (int j = 0; j < 6; j++) a[j] = lo[j] + (hi[j] - lo[j]) * uni();
            orc_env_step(s, a, obs, &rew, &disc, &st);
          }
        bar.wait();
      }
    }
    for (orc_sim* s : envs) orc_destroy(s);
  };
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; t++) th.emplace_back(worker, t);
  bar.wait();
  double t0 = now_s();
  bar.wait();
  if (reset_seconds) *reset_seconds = now_s() - t0;
  for (int k = 0; k < n_settings; k++)
    for (int r = 0; r < reps; r++) {
      bar.wait();
      double t1 = now_s();
      bar.wait();
      rep_seconds[k * reps + r] = now_s() - t1;
    }
  for (auto& x : th) x.join();
  for (int f : failed) if (f) return -1;
  return (long long)n_threads * envs_per_thread * steps;
}
}
