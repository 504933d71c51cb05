// libso101_hip.so — C ABI (include/so101.h) over the gfx950 kernels.  Host side: blob parsing, device
// copy of the model, launch plumbing.  No torch types, no CPU compute path: every entry point that
// does physics launches a HIP kernel or fails.
#include "so101_blob.hpp"
#include "../../include/so101.h"
#include "so101_launch.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace {

thread_local std::string g_create_error;

void h_quat2mat(float* m, const float* q) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = 1 - 2 * (y * y + z * z); m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = 1 - 2 * (x * x + z * z); m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = 1 - 2 * (x * x + y * y);
}
void h_mulquat(float* o, const float* a, const float* b) {
  float t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
  memcpy(o, t, sizeof t);
}
void h_matmul(float* o, const float* a, const float* b) {
  float t[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  memcpy(o, t, sizeof t);
}
// I = R diag(d) R^T packed xx yy zz xy xz yz
void h_inertia(float* o, const float* iquat, const float* diag, bool inverse) {
  float R[9]; h_quat2mat(R, iquat);
  float d[3] = {inverse ? 1.f / diag[0] : diag[0], inverse ? 1.f / diag[1] : diag[1], inverse ? 1.f / diag[2] : diag[2]};
  auto e = [&](int i, int j) { return R[3 * i] * d[0] * R[3 * j] + R[3 * i + 1] * d[1] * R[3 * j + 1] + R[3 * i + 2] * d[2] * R[3 * j + 2]; };
  o[0] = e(0, 0); o[1] = e(1, 1); o[2] = e(2, 2); o[3] = e(0, 1); o[4] = e(0, 2); o[5] = e(1, 2);
}

}  // namespace

struct so101_sim {
  int n_envs = 0, device = 0;
  uint64_t seed = 0;
  DevModel hm{};               // host copy (device pointers inside)
  DevModel* dm = nullptr;      // device copy
  std::vector<void*> owned;    // device allocations to free
  so101_config cfg{};
  DevBuffers buf{};
  bool bound = false;
  unsigned char* need_reset = nullptr;
  int* diag = nullptr;
  // reset prefetch: cache of settled initial states, filled by k_prepare on a low-priority side stream
  PrepBuffers prep{};
  hipStream_t prep_stream = nullptr;
  int prep_scan = 0;                       // next slice of the second look-ahead (launch_prepare)
  hipEvent_t prep_done = nullptr, main_ev = nullptr;
  bool prep_pending = false;
  int prep_waves = 0;
  float* conres_full = nullptr;            // [N][MAXCAND][CONRES_DIM] contact records of pipelines 2 and 3 (allocated on first use)
  unsigned long long* mq_slot = nullptr;   // merged launches (pipeline = 3): chunk rings, queue words and launch counters of the chains
  unsigned int *mq_ctl = nullptr, *mq_pub = nullptr;
  ChainQueues chain{};         // queues of the per-env chained step (pipeline = 2)
  unsigned char* chain_cls = nullptr;
  ChainParams* chain_params = nullptr;     // device copy of k_chain's parameter block
  ChainParams chain_host{};                // what it holds
  bool chain_params_valid = false;
  PipeBuffers pipe{};          // scratch of the pipelined step (group 0's view; the groups differ in work/counters)
  static constexpr int MAXGROUPS = 8;
  hipStream_t group_stream[MAXGROUPS] = {};
  hipEvent_t group_done[MAXGROUPS] = {};
  hipEvent_t step_begin = nullptr;
  EventBuffers ev{};           // per-env flag accumulator + global event counters (so101_get_events)
  // captured launch sequence of the pipelined step (see so101_step)
  hipGraphExec_t graph_exec = nullptr;
  so101::StepIO graph_io{};
  unsigned long long generation = 1, graph_gen = 0;     // bumped by configure / bind_state / set_reset_pool
  bool graph_failed = false;
  hipStream_t graph_failed_stream = nullptr;            // capture is retried when the caller moves to another stream
  int last_path = -1, last_chains = 0;                  // so101_get_info
  bool last_graph = false;
  size_t scratch_bytes = 0;
  std::string err;
};

namespace {

bool hip_ok(so101_sim* s, hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  s->err = std::string(what) + ": " + hipGetErrorString(e);
  return false;
}

// Every entry point that touches HIP runs with the handle's device current and restores the caller's device on exit
// (a handle is bound to one device; streams, events and allocations below belong to it).
struct DeviceGuard {
  int prev = -1, dev;
  bool ok;
  explicit DeviceGuard(so101_sim* s) : dev(s->device) {
    ok = hipGetDevice(&prev) == hipSuccess && (prev == dev || hipSetDevice(dev) == hipSuccess);
    if (!ok) s->err = "hipSetDevice: cannot make the handle's device current";
  }
  ~DeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};
#define GUARD_DEVICE(s) DeviceGuard guard_(s); if (!guard_.ok) return SO101_ERR_HIP

template <typename T>
bool upload(so101_sim* s, const std::vector<T>& v, const T** out) {
  void* p = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  if (!hip_ok(s, hipMalloc(&p, bytes), "hipMalloc(model)")) return false;
  s->owned.push_back(p);
  if (!v.empty() && !hip_ok(s, hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy(model)")) return false;
  *out = (const T*)p;
  return true;
}

StepParams make_params(const so101_sim* s) {
  StepParams P{};
  for (int k = 0; k < NU; k++) P.action_offset[k] = s->cfg.action_offset[k];
  P.last_step = s->cfg.last_step; P.n_substeps = s->cfg.n_substeps;
  P.iterations = s->cfg.solver_iterations > 0 ? s->cfg.solver_iterations : s->hm.iterations;
  P.tolerance = s->cfg.solver_tolerance >= 0.f ? s->cfg.solver_tolerance : s->hm.tolerance;
  P.settle_max = s->cfg.settle_max_substeps; P.terminate_on_success = s->cfg.terminate_on_success;
  P.n_envs = s->n_envs; P.seed = s->seed; P.env_id_base = s->cfg.env_id_base; P.solver = s->cfg.solver;
  P.phases = 7;
#ifdef SO101_DEBUG_CLOCKS
  if (const char* ph = getenv("SO101_DEBUG_PHASES")) P.phases = atoi(ph);      // profiling builds: stage mask of so101_physics
#endif
  return P;
}

int build_model(so101_sim* s, const BlobView& b) {
  auto fail = [&](const std::string& msg) { s->err = msg; return (int)SO101_ERR_MODEL; };
  const char* scalars[] = {"nq", "nv", "nu", "nbody", "ngeom", "narm", "nfree", "npair", "nvert", "opt_iterations", "opt_mpr_iterations",
                           "task_nbox", "opt_cone_elliptic", "opt_timestep", "opt_impratio", "opt_tolerance", "opt_mpr_tolerance",
                           "stat_meaninertia", "task_object_body", "task_container_body"};
  for (const char* n : scalars) if (b.count(n) < 1) return fail(std::string("blob entry missing: ") + n);
  {
    // every array the code below indexes, with the element count it relies on: a stale or foreign blob is rejected
    // here instead of being dereferenced
    size_t nbody = (size_t)b.I("nbody")[0], ngeom = (size_t)b.I("ngeom")[0], npair = (size_t)b.I("npair")[0], nvert = (size_t)b.I("nvert")[0];
    size_t nbox = (size_t)b.I("task_nbox")[0];
    if (nbody > 4096 || ngeom > 4096 || npair > (1u << 24) || nvert > (1u << 24) || nbox > 2) return fail("blob dimensions out of range");
    struct Need { const char* name; size_t count; };
    const Need arrays[] = {
      {"arm_body", NARM}, {"free_body", NFREE}, {"body_parent", nbody}, {"body_jnttype", nbody}, {"body_pos", 3 * nbody}, {"body_quat", 4 * nbody},
      {"body_ipos", 3 * nbody}, {"body_iquat", 4 * nbody}, {"body_mass", nbody}, {"body_inertia", 3 * nbody}, {"body_invweight0", 2 * nbody},
      {"body_bvh_aabb", 6 * nbody}, {"opt_gravity", 3}, {"jnt_axis", 3 * NARM}, {"jnt_range", 2 * NARM}, {"dof_armature", NARM},
      {"dof_frictionloss", NARM}, {"jnt_limited", NARM}, {"dof_invweight0", NV}, {"dof_damping", NARM}, {"jnt_solref", 2 * NARM},
      {"jnt_solimp", 5 * NARM}, {"dof_solref", 2 * NARM}, {"dof_solimp", 5 * NARM}, {"act_gain", NU}, {"act_bias", 3 * NU},
      {"act_ctrlrange", 2 * NU}, {"act_forcerange", 2 * NU}, {"act_ctrllimited", NU}, {"act_forcelimited", NU}, {"act_dof", NU},
      {"task_box_pos", 3 * nbox}, {"task_box_half", 3 * nbox}, {"task_obj_pos_lo", 3}, {"task_obj_pos_hi", 3}, {"task_obj_yaw", 2},
      {"task_con_pos_lo", 3}, {"task_con_pos_hi", 3}, {"task_home_ctrl", NU}, {"geom_type", ngeom}, {"geom_body", ngeom}, {"geom_condim", ngeom},
      {"geom_vertadr", ngeom}, {"geom_vertnum", ngeom}, {"geom_pos", 3 * ngeom}, {"geom_quat", 4 * ngeom}, {"geom_size", 3 * ngeom},
      {"geom_friction", 3 * ngeom}, {"geom_solref", 2 * ngeom}, {"geom_solimp", 5 * ngeom}, {"geom_center", 3 * ngeom}, {"geom_aabb", 6 * ngeom},
      {"geom_solmix", ngeom}, {"geom_margin", ngeom}, {"geom_gap", ngeom}, {"geom_priority", ngeom}, {"geom_rbound", ngeom}, {"mesh_vert", 3 * nvert}, {"pair_geom", 2 * npair}};
    for (const Need& a : arrays) if (b.count(a.name) < a.count) return fail(std::string("blob entry missing or too short: ") + a.name);
    auto in_range = [&](const char* name, size_t limit, bool allow_negative) {
      for (int v : b.I(name)) if ((v < 0 && !allow_negative) || (v >= 0 && (size_t)v >= limit)) return false;
      return true;
    };
    if (!in_range("arm_body", nbody, false) || !in_range("free_body", nbody, false) || !in_range("body_parent", nbody, false) ||
        !in_range("geom_body", nbody, false) || !in_range("pair_geom", ngeom, false) || !in_range("task_object_body", nbody, false) ||
        !in_range("task_container_body", nbody, false)) return fail("blob index array out of range");
    auto gva = b.I("geom_vertadr"), gvn = b.I("geom_vertnum");
    for (size_t g = 0; g < ngeom; g++) if (gvn[g] > 0 && (gva[g] < 0 || (size_t)gva[g] + (size_t)gvn[g] > nvert)) return fail("geom vertex range outside mesh_vert");
  }
  DevModel& M = s->hm;
  int nq = b.I("nq")[0], nv = b.I("nv")[0], nu = b.I("nu")[0], nbody = b.I("nbody")[0], ngeom = b.I("ngeom")[0];
  int narm = b.I("narm")[0], nfree = b.I("nfree")[0];
  if (narm != NARM || nfree != NFREE || nq != NQ || nv != NV || nu != NU)
    return fail("model topology outside this build (need a 6-hinge chain and 2 free bodies)");
  if (ngeom > MAXGEOM) return fail("too many collision geoms for this build");
  auto arm_body = b.I("arm_body"), free_body = b.I("free_body"), parent = b.I("body_parent"), jt = b.I("body_jnttype");
  auto bpos = b.F("body_pos"), bquat = b.F("body_quat"), ipos = b.F("body_ipos"), iquat = b.F("body_iquat");
  auto mass = b.F("body_mass"), inertia = b.F("body_inertia"), invw = b.F("body_invweight0"), bvh = b.F("body_bvh_aabb");
  for (int k = 1; k < NARM; k++) if (parent[arm_body[k]] != arm_body[k - 1]) return fail("arm links do not form a serial chain");
  if (jt[parent[arm_body[0]]] != 0) return fail("arm base must be static");
  for (int f = 0; f < NFREE; f++) if (parent[free_body[f]] != 0) return fail("free bodies must be children of the world");
  if (b.I("task_object_body")[0] != free_body[0] || b.I("task_container_body")[0] != free_body[1])
    return fail("free bodies must be ordered (object, container)");
  // world pose of static bodies
  std::vector<float> wpos(3 * nbody, 0.f), wquat(4 * nbody, 0.f);
  std::vector<int> is_static(nbody, 0);
  wquat[0] = 1.f; is_static[0] = 1;
  for (int i = 1; i < nbody; i++) {
    if (jt[i] != 0 || !is_static[parent[i]]) continue;
    is_static[i] = 1;
    float R[9]; h_quat2mat(R, &wquat[4 * parent[i]]);
    for (int k = 0; k < 3; k++) wpos[3 * i + k] = wpos[3 * parent[i] + k] + R[3 * k] * bpos[3 * i] + R[3 * k + 1] * bpos[3 * i + 1] + R[3 * k + 2] * bpos[3 * i + 2];
    h_mulquat(&wquat[4 * i], &wquat[4 * parent[i]], &bquat[4 * i]);
  }
  int base = parent[arm_body[0]];
  for (int k = 0; k < 3; k++) M.base_pos[k] = wpos[3 * base + k];
  for (int k = 0; k < 4; k++) M.base_quat[k] = wquat[4 * base + k];
  M.ngeom = ngeom; M.npair = b.I("npair")[0]; M.nvert = b.I("nvert")[0];
  M.iterations = b.I("opt_iterations")[0]; M.mpr_iter = b.I("opt_mpr_iterations")[0]; M.nbox = b.I("task_nbox")[0];
  if (M.nbox > 2) return fail("at most 2 overlap boxes");
  if (!b.I("opt_cone_elliptic")[0]) return fail("only elliptic cones are implemented (scene_pbr.xml:4)");
  M.dt = b.F("opt_timestep")[0]; auto g = b.F("opt_gravity"); for (int k = 0; k < 3; k++) M.grav[k] = g[k];
  M.impratio = b.F("opt_impratio")[0]; M.tolerance = b.F("opt_tolerance")[0]; M.mpr_tol = b.F("opt_mpr_tolerance")[0];
  M.meaninertia = b.F("stat_meaninertia")[0];
  auto axis = b.F("jnt_axis"), range = b.F("jnt_range"), arma = b.F("dof_armature"), floss = b.F("dof_frictionloss");
  auto limited = b.I("jnt_limited");
  auto dofw = b.F("dof_invweight0"), damping = b.F("dof_damping");
  auto jsr = b.F("jnt_solref"), jsi = b.F("jnt_solimp"), dsr = b.F("dof_solref"), dsi = b.F("dof_solimp");
  for (int k = 0; k < NARM; k++) {
    int bi = arm_body[k];
    for (int i = 0; i < 3; i++) { M.arm_pos[k][i] = bpos[3 * bi + i]; M.arm_axis[k][i] = axis[3 * k + i]; M.arm_ipos[k][i] = ipos[3 * bi + i]; }
    for (int i = 0; i < 4; i++) M.arm_quat[k][i] = bquat[4 * bi + i];
    h_inertia(M.arm_Ib[k], &iquat[4 * bi], &inertia[3 * bi], false);
    M.arm_mass[k] = mass[bi]; M.armature[k] = arma[k]; M.frictionloss[k] = floss[k];
    M.range[k][0] = range[2 * k]; M.range[k][1] = range[2 * k + 1]; M.limited[k] = limited[k];
    if (damping[k] != 0.f) return fail("joint damping is outside this build (SO100 arm has none)");
    for (int i = 0; i < 2; i++) if (jsr[2 * k + i] != jsr[i] || dsr[2 * k + i] != dsr[i]) return fail("per-joint solref must be uniform");
    for (int i = 0; i < 5; i++) if (jsi[5 * k + i] != jsi[i] || dsi[5 * k + i] != dsi[i]) return fail("per-joint solimp must be uniform");
  }
  for (int i = 0; i < 2; i++) { M.jnt_solref[i] = jsr[i]; M.dof_solref[i] = dsr[i]; }
  for (int i = 0; i < 5; i++) { M.jnt_solimp[i] = jsi[i]; M.dof_solimp[i] = dsi[i]; }
  for (int d = 0; d < NV; d++) M.dof_invweight0[d] = dofw[d];
  auto again = b.F("act_gain"), abias = b.F("act_bias"), acr = b.F("act_ctrlrange"), afr = b.F("act_forcerange");
  auto acl = b.I("act_ctrllimited"), afl = b.I("act_forcelimited"), adof = b.I("act_dof");
  for (int a = 0; a < NU; a++) {
    if (adof[a] != a) return fail("actuator a must drive dof a");
    M.act_gain[a] = again[a];
    for (int i = 0; i < 3; i++) M.act_bias[a][i] = abias[3 * a + i];
    for (int i = 0; i < 2; i++) { M.ctrlrange[a][i] = acr[2 * a + i]; M.forcerange[a][i] = afr[2 * a + i]; }
    M.ctrllimited[a] = acl[a]; M.forcelimited[a] = afl[a];
  }
  std::vector<int> dyn_of_body(nbody, -1);
  for (int k = 0; k < NARM; k++) dyn_of_body[arm_body[k]] = k;
  for (int f = 0; f < NFREE; f++) {
    int bi = free_body[f];
    dyn_of_body[bi] = NARM + f;
    M.free_mass[f] = mass[bi];
    for (int i = 0; i < 3; i++) M.free_ipos[f][i] = ipos[3 * bi + i];
    for (int i = 0; i < 4; i++) M.free_iquat[f][i] = iquat[4 * bi + i];
    h_inertia(M.free_Ib[f], &iquat[4 * bi], &inertia[3 * bi], false);
    h_inertia(M.free_Ibinv[f], &iquat[4 * bi], &inertia[3 * bi], true);
    for (int i = 0; i < 6; i++) M.free_bvh[f][i] = bvh[6 * bi + i];
  }
  for (int bi = 0; bi < nbody; bi++) if (dyn_of_body[bi] >= 0) { M.dyn_invweight0[dyn_of_body[bi]][0] = invw[2 * bi]; M.dyn_invweight0[dyn_of_body[bi]][1] = invw[2 * bi + 1]; }
  auto bp = b.F("task_box_pos"), bh = b.F("task_box_half");
  for (int k = 0; k < M.nbox; k++) for (int i = 0; i < 3; i++) { M.box_pos[k][i] = bp[3 * k + i]; M.box_half[k][i] = bh[3 * k + i]; }
  auto olo = b.F("task_obj_pos_lo"), ohi = b.F("task_obj_pos_hi"), oy = b.F("task_obj_yaw"), clo = b.F("task_con_pos_lo"), chi = b.F("task_con_pos_hi"), hc = b.F("task_home_ctrl");
  for (int i = 0; i < 3; i++) { M.obj_lo[i] = olo[i]; M.obj_hi[i] = ohi[i]; M.con_lo[i] = clo[i]; M.con_hi[i] = chi[i]; }
  M.obj_yaw[0] = oy[0]; M.obj_yaw[1] = oy[1];
  for (int i = 0; i < NU; i++) M.home_ctrl[i] = hc[i];
  // geoms: dynamic ones keep body-local frames, static ones are resolved to world frames here
  auto gtype = b.I("geom_type"), gbody = b.I("geom_body"), gcondim = b.I("geom_condim"), gva = b.I("geom_vertadr"), gvn = b.I("geom_vertnum");
  auto gpos = b.F("geom_pos"), gquat = b.F("geom_quat"), gsize = b.F("geom_size"), gfr = b.F("geom_friction"), gsr = b.F("geom_solref");
  auto gsi = b.F("geom_solimp"), gctr = b.F("geom_center"), gaabb = b.F("geom_aabb"), gmix = b.F("geom_solmix"), gmargin = b.F("geom_margin"), ggap = b.F("geom_gap");
  auto gprio = b.I("geom_priority");
  auto grb = b.F("geom_rbound");
  std::vector<int> gdyn(ngeom);
  std::vector<float> gp(3 * ngeom), gm(9 * ngeom);
  for (int i = 0; i < ngeom; i++) {
    if (gmix[i] != 1.f || gmargin[i] != 0.f || ggap[i] != 0.f || gprio[i] != 0) return fail("geom solmix/margin/gap/priority must be default");
    int bi = gbody[i];
    gdyn[i] = dyn_of_body[bi];
    float lm[9]; h_quat2mat(lm, &gquat[4 * i]);
    if (gdyn[i] >= 0) {
      memcpy(&gm[9 * i], lm, sizeof lm);
      for (int k = 0; k < 3; k++) gp[3 * i + k] = gpos[3 * i + k];
    } else {
      if (!is_static[bi]) return fail("geom on an unsupported body");
      float R[9]; h_quat2mat(R, &wquat[4 * bi]);
      h_matmul(&gm[9 * i], R, lm);
      for (int k = 0; k < 3; k++) gp[3 * i + k] = wpos[3 * bi + k] + R[3 * k] * gpos[3 * i] + R[3 * k + 1] * gpos[3 * i + 1] + R[3 * k + 2] * gpos[3 * i + 2];
    }
  }
  auto mv = b.F("mesh_vert");
  int nvert = M.nvert;
  std::vector<float> vx(nvert), vy(nvert), vz(nvert);
  for (int i = 0; i < nvert; i++) { vx[i] = mv[3 * i]; vy[i] = mv[3 * i + 1]; vz[i] = mv[3 * i + 2]; }
  auto pairs = b.I("pair_geom");
  // broadphase pair list, one word per pair: geom1 | geom2 << 8 | (geom1 is a plane) << 16, geom types ordered
  std::vector<unsigned int> packed(M.npair);
  for (int k = 0; k < M.npair; k++) {
    int g1 = pairs[2 * k], g2 = pairs[2 * k + 1];
    if (gtype[g1] > gtype[g2]) std::swap(g1, g2);
    packed[k] = (unsigned int)g1 | ((unsigned int)g2 << 8) | ((gtype[g1] == G_PLANE ? 1u : 0u) << 16);
  }
  // support-bound tables of the hulls (so101_model.hpp DevModel::hull_sbt, obb_filter): in double, rounded up to float
  std::vector<float> sbt((size_t)ngeom * SBT_DIM, 0.f);
  for (int g = 0; g < ngeom; g++) {
    if (gtype[g] != G_MESH) continue;
    for (int face = 0; face < 6; face++) {
      int ax = face / 2; double sg = (face & 1) ? -1.0 : 1.0;
      for (int iu = 0; iu < SBT_GRID; iu++)
        for (int iv = 0; iv < SBT_GRID; iv++) {
          const double step = 2.0 / (SBT_GRID - 1);
          double c[3]; c[ax] = sg; c[(ax + 1) % 3] = -1.0 + step * iu; c[(ax + 2) % 3] = -1.0 + step * iv;
          double best = -1e300;
          for (int k = gva[g]; k < gva[g] + gvn[g]; k++) best = std::max(best, (double)mv[3 * k] * c[0] + (double)mv[3 * k + 1] * c[1] + (double)mv[3 * k + 2] * c[2]);
          float f = (float)best;
          if ((double)f < best) f = std::nextafterf(f, 3.0e38f);
          sbt[(size_t)g * SBT_DIM + (face * SBT_GRID + iu) * SBT_GRID + iv] = f;
        }
    }
  }
  // support-vertex lists (so101_model.hpp DevModel::hl_entry): per hull and cube-map cell the vertices that can win a support query there
  std::vector<float> hle; std::vector<unsigned int> hlo((size_t)ngeom * (HL_CELLS + 1), 0u);
  const bool hl_off_env = getenv("SO101_NO_HL") != nullptr;          // (tests and kernel experiments: every query scans the whole hull, as until round 6)
  for (int g = 0; g < ngeom; g++) {
    unsigned int* off = &hlo[(size_t)g * (HL_CELLS + 1)];
    if (gtype[g] != G_MESH || hl_off_env) { for (int c = 0; c <= HL_CELLS; c++) off[c] = (unsigned int)(hle.size() / 4); continue; }
    const int n = gvn[g]; const float* V = &mv[3 * (size_t)gva[g]];
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int k = 0; k < n; k++) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], (double)V[3 * k + a]); hi[a] = std::max(hi[a], (double)V[3 * k + a]); }
    const double diam = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    std::vector<double> S((size_t)(HL_GRID + 1) * (HL_GRID + 1) * n);          // scores of every vertex at the grid points of one face
    std::vector<double> cn((size_t)(HL_GRID + 1) * (HL_GRID + 1));
    for (int face = 0; face < 6; face++) {
      int ax = face / 2; double sg = (face & 1) ? -1.0 : 1.0;
      for (int iu = 0; iu <= HL_GRID; iu++)
        for (int iv = 0; iv <= HL_GRID; iv++) {
          double c[3]; c[ax] = sg; c[(ax + 1) % 3] = -1.0 + 2.0 * iu / HL_GRID; c[(ax + 2) % 3] = -1.0 + 2.0 * iv / HL_GRID;
          size_t pt = (size_t)iu * (HL_GRID + 1) + iv;
          cn[pt] = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
          double* sp = &S[pt * n];
          for (int k = 0; k < n; k++) sp[k] = V[3 * k] * c[0] + V[3 * k + 1] * c[1] + V[3 * k + 2] * c[2];
        }
      for (int iu = 0; iu < HL_GRID; iu++)
        for (int iv = 0; iv < HL_GRID; iv++) {
          const size_t pts[4] = {(size_t)iu * (HL_GRID + 1) + iv, (size_t)(iu + 1) * (HL_GRID + 1) + iv, (size_t)iu * (HL_GRID + 1) + iv + 1, (size_t)(iu + 1) * (HL_GRID + 1) + iv + 1};
          int win[4]; double eps[4];
          for (int q = 0; q < 4; q++) {
            const double* sp = &S[pts[q] * n]; int w = 0;
            for (int k = 1; k < n; k++) if (sp[k] > sp[w]) w = k;
            win[q] = w; eps[q] = 4e-3 * diam * cn[pts[q]] + 1e-6;
          }
          off[(face * HL_GRID + iu) * HL_GRID + iv] = (unsigned int)(hle.size() / 4);
          for (int k = 0; k < n; k++) {
            bool keep = true;
            for (int w = 0; w < 4 && keep; w++) {            // beaten by corner winner w at ALL four corners by more than the widening: out
              bool some = false;
              for (int q = 0; q < 4; q++) some = some || S[pts[q] * n + k] >= S[pts[q] * n + win[w]] - eps[q];
              keep = some;
            }
            if (keep) { hle.push_back(V[3 * k]); hle.push_back(V[3 * k + 1]); hle.push_back(V[3 * k + 2]); float fi; unsigned int ui = (unsigned int)k; memcpy(&fi, &ui, 4); hle.push_back(fi); }
          }
        }
    }
    off[HL_CELLS] = (unsigned int)(hle.size() / 4);
  }
  M.hl_entry = nullptr; M.hl_off = nullptr;
  if (!hl_off_env && !(upload(s, hle, &M.hl_entry) && upload(s, hlo, &M.hl_off))) return SO101_ERR_HIP;
  const bool sbt_off = getenv("SO101_NO_SBT") != nullptr;          // (tests and kernel experiments, read at every so101_create: the oriented-box filter alone, as until round 5)
  M.hull_sbt = nullptr;
  bool ok = (sbt_off || upload(s, sbt, &M.hull_sbt)) &&
            upload(s, gtype, &M.geom_type) && upload(s, gdyn, &M.geom_dyn) && upload(s, gcondim, &M.geom_condim) &&
            upload(s, gva, &M.geom_vertadr) && upload(s, gvn, &M.geom_vertnum) && upload(s, gp, &M.geom_pos) &&
            upload(s, gm, &M.geom_mat) && upload(s, gsize, &M.geom_size) && upload(s, gfr, &M.geom_friction) &&
            upload(s, gsr, &M.geom_solref) && upload(s, gsi, &M.geom_solimp) && upload(s, gctr, &M.geom_center) &&
            upload(s, gaabb, &M.geom_aabb) && upload(s, grb, &M.geom_rbound) && upload(s, vx, &M.vx) && upload(s, vy, &M.vy) && upload(s, vz, &M.vz) &&
            upload(s, pairs, &M.pair) && upload(s, packed, &M.pair_packed);
  if (!ok) return SO101_ERR_HIP;
  void* dm = nullptr;
  if (!hip_ok(s, hipMalloc(&dm, sizeof(DevModel)), "hipMalloc(DevModel)")) return SO101_ERR_HIP;
  s->owned.push_back(dm);
  if (!hip_ok(s, hipMemcpy(dm, &M, sizeof(DevModel), hipMemcpyHostToDevice), "hipMemcpy(DevModel)")) return SO101_ERR_HIP;
  s->dm = (DevModel*)dm;
  return SO101_OK;
}

// Launches k_prepare behind whatever `stream` holds now, unless the previous one is still running (it picks up
// every env whose next episode is missing, so skipping a launch only delays the refill).
// (pool resets are plain copies: nothing to prefetch)
bool prefetch_on(const so101_sim* s) { return s->cfg.prefetch_resets && s->prep_stream && s->cfg.solver == SO101_SOLVER_NEWTON && s->prep.pool_size == 0; }

void launch_prepare(so101_sim* s, hipStream_t stream, bool wide = false) {
  if (!prefetch_on(s)) return;
  if (s->prep_pending) {
    if (hipEventQuery(s->prep_done) != hipSuccess) { (void)hipGetLastError(); return; }
    s->prep_pending = false;
  }
  if (hipEventRecord(s->main_ev, stream) != hipSuccess) return;
  if (hipStreamWaitEvent(s->prep_stream, s->main_ev, 0) != hipSuccess) return;
  if (hipMemsetAsync(s->prep.cursor, 0, 2 * sizeof(int), s->prep_stream) != hipSuccess) return;
  // `wide` (after an explicit so101_reset, when nothing is stepping yet): the first fill of the cache - two episodes for every
  // env - at the width of the machine (0.8 s for 4096 envs) instead of 256 wavefronts beside the first ~600 control steps, during
  // which every env whose physics diverged had to settle inside the step call (130 ms for the whole batch each time)
  int waves = wide ? (s->n_envs < 2048 ? s->n_envs : 2048) : s->prep_waves;
  // (the second look-ahead in slices - PrepBuffers::scan_first / scan_count, SO101_PREP_SLICE=<envs> - was measured in round 5: it removes the
  //  ~2 s that a device-wide synchronise right after a mass reset waits for the refill - `bench.py --steps 500` reads 388 k on the host clock
  //  against 666 k on the device's - but four or eight launches with their own tails do not finish the refill of 4096 envs within an episode
  //  beside the steps, and envs then settle inside step calls: 1500 steps 627 k -> 590 k (1024-env slices) / 503 k (512).  One launch stays.)
  PrepBuffers C = s->prep;
  static const int prep_slice = getenv("SO101_PREP_SLICE") ? atoi(getenv("SO101_PREP_SLICE")) : 0;
  if (!wide && prep_slice > 0 && s->n_envs > prep_slice) { C.scan_first = s->prep_scan; C.scan_count = prep_slice; s->prep_scan = (s->prep_scan + prep_slice) % s->n_envs; }
  else { C.scan_first = 0; C.scan_count = 0; }
  so101::launch_prepare(waves, s->prep_stream, s->dm, make_params(s), s->buf, C);
  if (hipEventRecord(s->prep_done, s->prep_stream) == hipSuccess) s->prep_pending = true;
}

bool drain_prepare(so101_sim* s) {
  if (!s->prep_stream) return true;
  s->prep_pending = false;
  return hip_ok(s, hipStreamSynchronize(s->prep_stream), "hipStreamSynchronize(prepare)");
}

// the cache the kernels see: none when the prefetch is off (PGS, or prefetch_resets = 0)
PrepBuffers prep_view(const so101_sim* s) {
  PrepBuffers C = s->prep;
  if (!prefetch_on(s)) C.tag = nullptr;
  return C;
}

template <typename T>
bool dev_alloc(so101_sim* s, T** out, size_t count, int fill, const char* what) {
  void* p = nullptr;
  size_t bytes = sizeof(T) * (count ? count : 1);
  if (!hip_ok(s, hipMalloc(&p, bytes), what)) return false;
  s->owned.push_back(p);
  s->scratch_bytes += bytes;
  if (!hip_ok(s, hipMemset(p, fill, bytes), what)) return false;
  *out = (T*)p;
  return true;
}

}  // namespace

extern "C" {

static void drop_graph(so101_sim* s);

int so101_version(void) { return SO101_ABI_VERSION; }
int so101_max_contacts(void) { return MAXCON; }

int so101_default_config(so101_config* cfg) {
  if (!cfg) return SO101_ERR_ARG;
  memset(cfg, 0, sizeof *cfg);
  cfg->last_step = 1 << 30; cfg->n_substeps = 10; cfg->solver_iterations = 0; cfg->solver_tolerance = -1.f;
  cfg->settle_max_substeps = 1000; cfg->terminate_on_success = 1; cfg->env_id_base = 0; cfg->solver = SO101_SOLVER_NEWTON;
  cfg->prefetch_resets = 1; cfg->pipeline = 1; cfg->groups = 0; cfg->use_graph = 1; cfg->chain_waves = 0;
  return SO101_OK;
}

int so101_create(const void* blob, size_t bytes, int n_envs, int device, uint64_t seed, so101_sim** out) {
  if (!blob || !out || n_envs <= 0) { g_create_error = "so101_create: bad argument"; return SO101_ERR_ARG; }
  *out = nullptr;
  so101_sim* s = new so101_sim();
  s->n_envs = n_envs; s->device = device; s->seed = seed;
  so101_default_config(&s->cfg);
  BlobView b;
  int rc = SO101_OK;
  if (!b.parse(blob, bytes, s->err)) rc = SO101_ERR_MODEL;
  std::unique_ptr<DeviceGuard> guard;
  if (rc == SO101_OK) { guard.reset(new DeviceGuard(s)); if (!guard->ok) rc = SO101_ERR_HIP; }
  if (rc == SO101_OK) rc = build_model(s, b);
  size_t n = (size_t)n_envs;
  if (rc == SO101_OK) {
    bool ok = dev_alloc(s, &s->need_reset, n, 1, "hipMalloc(need_reset)") && dev_alloc(s, &s->diag, SO101_DIAG_DIM * n, 0, "hipMalloc(diag)") &&
              dev_alloc(s, &s->ev.flags, n, 0, "hipMalloc(events)") && dev_alloc(s, &s->ev.events, (size_t)SO101_NEVENTS, 0, "hipMalloc(events)");
    if (!ok) rc = SO101_ERR_HIP;
  }
  if (rc == SO101_OK) {
    PrepBuffers& C = s->prep;
    bool ok = dev_alloc(s, &C.qpos, 2 * NQ * n, 0, "hipMalloc(prep)") && dev_alloc(s, &C.qvel, 2 * NV * n, 0, "hipMalloc(prep)") &&
              dev_alloc(s, &C.warm, 2 * NV * n, 0, "hipMalloc(prep)") && dev_alloc(s, &C.tag, 2 * n, 0xFF, "hipMalloc(prep)") &&
              dev_alloc(s, &C.cursor, (size_t)2, 0, "hipMalloc(prep)") && dev_alloc(s, &C.flags, 2 * n, 0, "hipMalloc(prep)");
    int lo = 0, hi = 0;
    ok = ok && hip_ok(s, hipDeviceGetStreamPriorityRange(&lo, &hi), "hipDeviceGetStreamPriorityRange") &&
         hip_ok(s, hipStreamCreateWithPriority(&s->prep_stream, hipStreamNonBlocking, lo), "hipStreamCreateWithPriority") &&
         hip_ok(s, hipEventCreateWithFlags(&s->prep_done, hipEventDisableTiming), "hipEventCreate") &&
         hip_ok(s, hipEventCreateWithFlags(&s->main_ev, hipEventDisableTiming), "hipEventCreate");
    // one wave per CU at most: the refill runs beside the stepping kernels, it must not crowd them out
    if (ok) s->prep_waves = n_envs < 256 ? n_envs : 256; else rc = SO101_ERR_HIP;
  }
  if (rc == SO101_OK) {
    PipeBuffers& W = s->pipe;
    bool ok = dev_alloc(s, &W.pose, NDYN * 12 * n, 0, "hipMalloc(pipe)") && dev_alloc(s, &W.cand, MAXCAND * n, 0, "hipMalloc(pipe)") &&
              dev_alloc(s, &W.ncand, n, 0, "hipMalloc(pipe)") && dev_alloc(s, &W.items, ITEM_WORDS * (CONRES_PER_ENV * n + (size_t)MAXCAND * so101_sim::MAXGROUPS), 0, "hipMalloc(pipe)") &&
              dev_alloc(s, &W.counters, (size_t)4 * MAXSUB * so101_sim::MAXGROUPS, 0, "hipMalloc(pipe)") &&
              dev_alloc(s, &W.conres, CONRES_DIM * (CONRES_PER_ENV * n + (size_t)MAXCAND * so101_sim::MAXGROUPS), 0, "hipMalloc(pipe)") && dev_alloc(s, &W.cbase, n, 0, "hipMalloc(pipe)") &&
              dev_alloc(s, &W.active, n, 0, "hipMalloc(pipe)") &&
#ifdef SO101_DEBUG_CLOCKS
              dev_alloc(s, &W.ticks, MAXCAND * n, 0, "hipMalloc(pipe)") &&
#else
              dev_alloc(s, &W.ticks, (size_t)1, 0, "hipMalloc(pipe)") &&
#endif
              dev_alloc(s, &W.stage, 8 * n, 0, "hipMalloc(pipe)") &&
              dev_alloc(s, &W.cost, n, 0, "hipMalloc(pipe)") && dev_alloc(s, &W.order, n, 0, "hipMalloc(pipe)") &&
              dev_alloc(s, &W.state, STATE_AOS * n, 0, "hipMalloc(pipe)") &&
              dev_alloc(s, &s->chain.pending, n, 0, "hipMalloc(chain)") && dev_alloc(s, &s->chain_cls, n, 0, "hipMalloc(chain)") &&
              dev_alloc(s, &s->chain.qctl, (size_t)4 * 64, 0, "hipMalloc(chain)") && dev_alloc(s, &s->chain.chain_ctl, (size_t)64, 0, "hipMalloc(chain)") &&
              dev_alloc(s, &s->chain_params, (size_t)1, 0, "hipMalloc(chain)") && dev_alloc(s, &s->chain.stats, (size_t)16, 0, "hipMalloc(chain)");
    s->chain.cls = s->chain_cls;
    s->chain.idle_sleeps = getenv("SO101_CHAIN_IDLE") ? atoi(getenv("SO101_CHAIN_IDLE")) : 8;
    s->chain.role_mode = getenv("SO101_CHAIN_ROLE") ? atoi(getenv("SO101_CHAIN_ROLE")) : 0;
    // (the queues of pipelines 2 and 3 and their full-size contact-record array are allocated when such a step is first asked
    // for: ensure_experimental_buffers())
    // (kernel experiments: SO101_GROUP_PRIO=1 gives the chain of the most expensive envs - slice 0 of the cost-sorted order - the highest
    //  stream priority and the cheapest slice the lowest)
    int plo = 0, phi = 0;
    hipDeviceGetStreamPriorityRange(&plo, &phi);
    const bool gprio_on = getenv("SO101_GROUP_PRIO") && atoi(getenv("SO101_GROUP_PRIO")) != 0;
    for (int g = 0; g < so101_sim::MAXGROUPS && ok; g++) {
      int pr = !gprio_on ? 0 : (g == 0 ? phi : (g >= 3 ? plo : (plo + phi) / 2));
      // (measured in round 5 and removed again: the chains pinned to disjoint sets of 64 CUs by hipExtStreamCreateWithCUMask - kernels of different
      //  chains then never share a CU or its instruction cache - ran at 459 k against 666 k env-steps/s without graph replay; 48 and 32 CUs 283 / 281 k)
      ok = hip_ok(s, gprio_on ? hipStreamCreateWithPriority(&s->group_stream[g], hipStreamNonBlocking, pr) : hipStreamCreateWithFlags(&s->group_stream[g], hipStreamNonBlocking), "hipStreamCreate") &&
           hip_ok(s, hipEventCreateWithFlags(&s->group_done[g], hipEventDisableTiming), "hipEventCreate");
    }
    ok = ok && hip_ok(s, hipEventCreateWithFlags(&s->step_begin, hipEventDisableTiming), "hipEventCreate");
    if (!ok) rc = SO101_ERR_HIP;
  }
  if (rc != SO101_OK) { g_create_error = s->err; so101_destroy(s); return rc; }
  *out = s;
  return SO101_OK;
}

void so101_destroy(so101_sim* s) {
  if (!s) return;
  {
    DeviceGuard guard(s);
    drop_graph(s);
    if (s->prep_stream) { (void)hipStreamSynchronize(s->prep_stream); (void)hipStreamDestroy(s->prep_stream); }
    for (int g = 0; g < so101_sim::MAXGROUPS; g++) {
      if (s->group_stream[g]) { (void)hipStreamSynchronize(s->group_stream[g]); (void)hipStreamDestroy(s->group_stream[g]); }
      if (s->group_done[g]) (void)hipEventDestroy(s->group_done[g]);
    }
    if (s->step_begin) (void)hipEventDestroy(s->step_begin);
    if (s->prep_done) (void)hipEventDestroy(s->prep_done);
    if (s->main_ev) (void)hipEventDestroy(s->main_ev);
    for (void* p : s->owned) (void)hipFree(p);
  }
  delete s;
}

int so101_configure(so101_sim* s, const so101_config* cfg) {
  if (!s || !cfg) return SO101_ERR_ARG;
  if (cfg->n_substeps <= 0 || cfg->settle_max_substeps < 0) { s->err = "so101_configure: bad substep counts"; return SO101_ERR_ARG; }
  if (cfg->solver != SO101_SOLVER_PGS && cfg->solver != SO101_SOLVER_NEWTON) { s->err = "so101_configure: unknown solver"; return SO101_ERR_ARG; }
  GUARD_DEVICE(s);
  // cached initial states were settled under the old configuration
  if (!drain_prepare(s)) return SO101_ERR_HIP;
  if (s->prep.tag && !hip_ok(s, hipMemset(s->prep.tag, 0xFF, sizeof(int) * 2 * (size_t)s->n_envs), "hipMemset(prep)")) return SO101_ERR_HIP;
  s->cfg = *cfg;
  s->generation++;
  return SO101_OK;
}

int so101_bind_state(so101_sim* s, const so101_buffers* b) {
  if (!s || !b) return SO101_ERR_ARG;
  if (!b->qpos || !b->qvel || !b->ctrl || !b->warmstart || !b->obs_ring || !b->ep_return || !b->step_count || !b->episode) {
    s->err = "so101_bind_state: NULL buffer"; return SO101_ERR_ARG;
  }
  GUARD_DEVICE(s);
  // a prefetch in flight reads the OLD episode buffer: let it finish before the pointers change, and forget what it
  // cached (the new buffers carry their own episode counters)
  if (!drain_prepare(s)) return SO101_ERR_HIP;
  if (s->bound && s->prep.tag && !hip_ok(s, hipMemset(s->prep.tag, 0xFF, sizeof(int) * 2 * (size_t)s->n_envs), "hipMemset(prep)")) return SO101_ERR_HIP;
  s->buf.qpos = b->qpos; s->buf.qvel = b->qvel; s->buf.ctrl = b->ctrl; s->buf.warm = b->warmstart; s->buf.ring = b->obs_ring;
  s->buf.ep_return = b->ep_return; s->buf.step_count = b->step_count; s->buf.episode = b->episode;
  s->buf.mass_scale = b->mass_scale;
  s->bound = true;
  s->generation++;
  return SO101_OK;
}

int so101_bind_physics_state(so101_sim* s, float* ring, float* out, float* delayed) {
  if (!s) return SO101_ERR_ARG;
  bool all = ring && out && delayed, none = !ring && !out && !delayed;
  if (!all && !none) { s->err = "so101_bind_physics_state: give all three arrays or none"; return SO101_ERR_ARG; }
  GUARD_DEVICE(s);
  if (!drain_prepare(s)) return SO101_ERR_HIP;
  s->buf.ps_ring = ring; s->buf.ps_out = out; s->buf.ps_delayed = delayed;
  s->generation++;
  return SO101_OK;
}

#define REQUIRE_BOUND(s) do { if (!(s)) return SO101_ERR_ARG; if (!(s)->bound) { (s)->err = "state buffers not bound (call so101_bind_state)"; return SO101_ERR_STATE; } } while (0)
#define LAUNCH_CHECK(s, name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { (s)->err = std::string(name) + ": " + hipGetErrorString(e_); return SO101_ERR_HIP; } } while (0)

int so101_reset(so101_sim* s, const uint8_t* mask, void* stream) {
  REQUIRE_BOUND(s);
  GUARD_DEVICE(s);
  so101::launch_reset(s->cfg.solver, s->n_envs, (hipStream_t)stream, s->dm, make_params(s), s->buf, prep_view(s), s->ev, mask, s->need_reset, s->diag);
  LAUNCH_CHECK(s, "k_reset");
  launch_prepare(s, (hipStream_t)stream, mask == nullptr);
  return SO101_OK;
}

int so101_set_reset_pool(so101_sim* s, const float* qpos, const float* qvel, const float* ctrl, int pool_size) {
  if (!s || pool_size < 0) return SO101_ERR_ARG;
  if (pool_size > 0 && (!qpos || !qvel || !ctrl)) { s->err = "so101_set_reset_pool: NULL pool array"; return SO101_ERR_ARG; }
  GUARD_DEVICE(s);
  if (!drain_prepare(s)) return SO101_ERR_HIP;
  s->prep.pool_qpos = pool_size ? qpos : nullptr; s->prep.pool_qvel = pool_size ? qvel : nullptr;
  s->prep.pool_ctrl = pool_size ? ctrl : nullptr; s->prep.pool_size = pool_size;
  s->generation++;
  return SO101_OK;
}

int so101_compute_settled(so101_sim* s, int first_episode, int n_episodes, float* qpos, float* qvel, float* warm, int32_t* flags, void* stream) {
  REQUIRE_BOUND(s);      // the bound mass_scale array (if any) takes part in the settle
  if (first_episode < 0 || n_episodes <= 0 || !qpos || !qvel || !warm || !flags) { s->err = "so101_compute_settled: bad argument"; return SO101_ERR_ARG; }
  if ((long long)n_episodes * s->n_envs > (1ll << 30)) { s->err = "so101_compute_settled: table too large for one launch"; return SO101_ERR_ARG; }
  GUARD_DEVICE(s);
  so101::launch_settle_table(s->cfg.solver, s->n_envs, first_episode, n_episodes, (hipStream_t)stream, s->dm, make_params(s), s->buf, qpos, qvel, warm, flags);
  LAUNCH_CHECK(s, "k_settle_table");
  return SO101_OK;
}

int so101_set_settled_store(so101_sim* s, const float* qpos, const float* qvel, const float* warm, const int32_t* flags, int first_episode, int n_episodes) {
  if (!s || n_episodes < 0 || first_episode < 0) return SO101_ERR_ARG;
  if (n_episodes > 0 && (!qpos || !qvel || !warm || !flags)) { s->err = "so101_set_settled_store: NULL table"; return SO101_ERR_ARG; }
  GUARD_DEVICE(s);
  if (!drain_prepare(s)) return SO101_ERR_HIP;
  s->prep.store_qpos = n_episodes ? qpos : nullptr; s->prep.store_qvel = n_episodes ? qvel : nullptr;
  s->prep.store_warm = n_episodes ? warm : nullptr; s->prep.store_flags = n_episodes ? (const int*)flags : nullptr;
  s->prep.store_first = first_episode; s->prep.store_count = n_episodes;
  s->generation++;
  return SO101_OK;
}

int so101_settle(so101_sim* s, void* stream) {
  REQUIRE_BOUND(s);
  GUARD_DEVICE(s);
  so101::launch_settle(s->cfg.solver, s->n_envs, (hipStream_t)stream, s->dm, make_params(s), s->buf, s->ev, s->diag);
  LAUNCH_CHECK(s, "k_settle");
  return SO101_OK;
}

int so101_begin_episode(so101_sim* s, void* stream) {
  REQUIRE_BOUND(s);
  GUARD_DEVICE(s);
  so101::launch_begin(s->n_envs, (hipStream_t)stream, s->dm, make_params(s), s->buf, s->need_reset);
  LAUNCH_CHECK(s, "k_begin");
  return SO101_OK;
}

// Enqueues the launch sequence of one pipelined control step on `st` (and the handle's chain streams, forked from and
// joined to `st` with events).  Also the body of the captured HIP graph.
static int enqueue_pipelined(so101_sim* s, hipStream_t st, const so101::StepIO& io) {
  StepParams P = make_params(s);
  PrepBuffers C = prep_view(s);
  // groups = 0 (default): four chains when the HIP runtime was given enough hardware queues for them, else three.
  // The runtime maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; this step uses chains + 2 streams,
  // and streams that share a queue serialise: with the default, 4 chains run at 400 k env-steps/s against 650 k for
  // 3; with GPU_MAX_HW_QUEUES=8 (so101_sim_amd sets it when imported before HIP initialises) 4 chains reach 672 k.
  // SO101_HW_QUEUES_EFFECTIVE: set by a host that knows the variable came too late for the runtime (so101_sim_amd/__init__.py)
  static const int hwq = getenv("SO101_HW_QUEUES_EFFECTIVE") ? atoi(getenv("SO101_HW_QUEUES_EFFECTIVE"))
                         : (getenv("GPU_MAX_HW_QUEUES") ? atoi(getenv("GPU_MAX_HW_QUEUES")) : 4);
  int G = s->cfg.groups > 0 ? s->cfg.groups : (hwq >= 6 ? 4 : 3);
  static bool warned = false;
  if (!warned && hwq < 6 && s->cfg.groups == 0 && s->n_envs >= 64) {
    warned = true;
    fprintf(stderr, "libso101_hip: %d hardware queues (GPU_MAX_HW_QUEUES): the pipelined step runs three launch chains instead of four "
                    "(about 3 %% slower at 4096 envs); export GPU_MAX_HW_QUEUES=8 before HIP initialises\n", hwq);
  }
  if (G > so101_sim::MAXGROUPS) G = so101_sim::MAXGROUPS;
  int n = s->n_envs;
  if (n < 64) G = 1;
  s->last_chains = G;
  // Slices of the cost-sorted env order (most expensive first), one launch chain each: 2 chains split at n/2, 3 at n/4
  // and 5n/8, 4 and more in equal parts (re-measured in round 2 at 4096 envs: 1 chain 515 k, 2 618 k, 3 653 k, 4 672 k
  // with enough hardware queues, 5-6 650 k env-steps/s; unequal 4-way splits are 1-2 % slower)
  int bounds[so101_sim::MAXGROUPS + 1];
  bounds[0] = 0;
  if (G == 2) bounds[1] = n / 2;
  if (G == 3) { bounds[1] = n / 4; bounds[2] = (5 * n) / 8; }
  if (G > 3) for (int g = 1; g < G; g++) bounds[g] = (int)((long long)n * g / G);
  bounds[G] = n;
#ifdef SO101_DEBUG_CLOCKS
  static const char* dbg_bounds = getenv("SO101_DEBUG_BOUNDS");     // profiling builds: "128,1024" = slice ends
  if (dbg_bounds) {
    G = 0; bounds[0] = 0;
    for (const char* p = dbg_bounds; *p && G < so101_sim::MAXGROUPS - 1;) { int v = atoi(p); if (v > bounds[G] && v < n) bounds[++G] = v; while (*p && *p != ',') p++; if (*p) p++; }
    bounds[++G] = n;
  }
#endif
  // (equal slices - four chains - get the sorted envs dealt out, tu_pipe_solve.hip k_order; the three unequal slices keep the plain sorted order)
  bool equal_slices = G > 1;
  for (int g = 0; g < G && equal_slices; g++) equal_slices = bounds[g + 1] - bounds[g] == n / G;
  so101::launch_order(st, s->pipe.cost, s->pipe.order, s->chain_cls, n, equal_slices && n % G == 0 ? G : 1);
  LAUNCH_CHECK(s, "k_order");
  if (G > 1 && !hip_ok(s, hipEventRecord(s->step_begin, st), "hipEventRecord")) return SO101_ERR_HIP;
  for (int g = 0; g < G; g++) {
    int e0 = bounds[g], ng = bounds[g + 1] - bounds[g];
    if (ng <= 0) continue;
    hipStream_t gs = G == 1 ? st : s->group_stream[g];
    PipeBuffers W = s->pipe;
    W.counters = s->pipe.counters + 4 * MAXSUB * g;      // [MAXSUB][2] work items / cursor, then [MAXSUB][2] heavy / light items (so101_pipeline.hpp)
    // the slice's pool of contact records: CONRES_PER_ENV per env plus a floor of MAXCAND, so that ONE env can always place every
    // candidate the broadphase may hand over (a single env or a small batch is not cut off at 48 records; which env would lose
    // records depended on the arrival order of the atomics in publish_candidates)
    W.conres = s->pipe.conres + ((size_t)e0 * CONRES_PER_ENV + (size_t)g * MAXCAND) * CONRES_DIM;
    W.conres_cap = (unsigned int)ng * CONRES_PER_ENV + MAXCAND;
    W.items = s->pipe.items + ((size_t)e0 * CONRES_PER_ENV + (size_t)g * MAXCAND) * ITEM_WORDS;      // one work item per record position of the slice
    static const int chunk_env = getenv("SO101_NARROW_CHUNK") ? atoi(getenv("SO101_NARROW_CHUNK")) : 0;          // (kernel experiments)
    static const int chunk_env_l = getenv("SO101_NARROW_CHUNK_LIGHT") ? atoi(getenv("SO101_NARROW_CHUNK_LIGHT")) : 0;
    // measured at 4096 envs (heavy / light pairs per fetch -> env-steps/s): 3/3 707 k, 1/3 684 k, 2/3 712 k, 2/4 720 k, 1/4 692 k, 2/2 679 k
    // (a handful of envs: one pair per fetch - 16 wavefronts share an env's dozen pairs, and the step is a chain of dependent launches)
    // (round 6, with the support-bound tables and the fast path for flat faces the light pairs no longer hide a second heavy pair behind the first:
    //  heavy pairs one per fetch and 1.25 wavefronts per env 763 -> 781 k env-steps/s at 4096 envs, 439 -> 461 k at 2048, each alone +1 %; at 8192
    //  envs 893 -> 880 k, so two per fetch and 1.5 wavefronts per env stay there)
    int ch = chunk_env >= 1 && chunk_env <= NARROW_CHUNK ? chunk_env : (n <= 4096 ? 1 : (n <= 8192 ? 2 : NARROW_CHUNK));
    // (round 5, work items + LDS hull pool of 1024 slots: light pairs per fetch 4 / 3 / 2 / 1 -> 718 / 736 / 740 / 604 k env-steps/s at 4096 envs - four
    //  light pairs with a 512-slot hull among them overflow the pool and stage late -; 32 768 envs, row-pass instance: 4 -> 1078 k, 2 -> 938 k)
    int cl = chunk_env_l >= 1 && chunk_env_l <= NARROW_CHUNK ? chunk_env_l : (n <= 16 ? 1 : (n <= 8192 ? 2 : NARROW_CHUNK));
    // bit 8: the row pass (four light pairs per wavefront, one per DPP row; so101 tu_narrow.hip).  Measured, round 5: 32 768 envs 996 k -> 1 067 k
    // env-steps/s (first window 1.21 -> 1.35 M); 4096 envs 715 k -> 710 k (there the step follows the critical path of its slowest slice -
    // heavy pairs, long Newton solves -, not the light pairs' instruction count), so it is on above 8192 envs
    const char* rows_var = getenv("SO101_NARROW_ROWS");            // (tests and kernel experiments: read when the step is enqueued / captured, so a handle created after a change sees it)
    const int rows_env = rows_var ? atoi(rows_var) : -1;
    const bool rows = rows_env >= 0 ? rows_env != 0 : n > 8192;
    W.narrow_chunk = (unsigned int)ch | ((unsigned int)cl << 4) | (rows ? 256u : 0u);        // heavy region | light region | row pass
    // persistent narrowphase waves (they pull work items until the list is empty): 1.5 - 2 per env of the slice, at
    // most what fills 256 CUs - a smaller narrowphase grid leaves slots to the other chains' solve kernels
    // (round 4, with the heavy-first work list: 1.5 waves per env 725 k, 2 per env 718 k, 2.5 per env 697 k env-steps/s at 4096 envs)
    static const int nw_env = getenv("SO101_NARROW_WAVES_Q") ? atoi(getenv("SO101_NARROW_WAVES_Q")) : 0;      // (kernel experiments: quarter waves per env)
    const int nw_quarters = nw_env > 0 ? nw_env : (n <= 4096 ? 5 : (n <= 8192 ? 6 : 8));
    int nw = (int)((long long)ng * nw_quarters / 4);
    // (a handful of envs - the reference's own N = 1: a step is a chain of dependent single-env launches, so the pairs of an env are spread
    //  over 16 wavefronts instead of queued on one or two)
    nw = nw < 16 ? 16 : (nw < 4096 ? nw : 4096);
    if (G > 1 && !hip_ok(s, hipStreamWaitEvent(gs, s->step_begin, 0), "hipStreamWaitEvent")) return SO101_ERR_HIP;
    if (!hip_ok(s, hipMemsetAsync(W.counters, 0, sizeof(int) * 4 * MAXSUB, gs), "hipMemsetAsync(pipe)")) return SO101_ERR_HIP;
    if (s->cfg.pipeline == 3) {
      // merged launches: the narrowphase of substep k + 1 rides in the solve launch of substep k (so101_chain.hpp)
      size_t capg = 64; while (capg < (size_t)ng * (MAXCAND / NARROW_CHUNK)) capg <<= 1;
      W.conres = s->conres_full; W.conres_cap = 0u;
      W.mq_ctl = s->mq_ctl + 64 * g; W.mq_pub = s->mq_pub + 128 * g;
      W.mq_slot = s->mq_slot + (size_t)2 * e0 * (MAXCAND / NARROW_CHUNK); W.mq_mask = (unsigned int)(capg - 1);
      if (!hip_ok(s, hipMemsetAsync(W.mq_pub, 0, sizeof(unsigned int) * 127, gs), "hipMemsetAsync(merged)")) return SO101_ERR_HIP;
      so101::launch_pipe_begin(ng, gs, s->dm, P, s->buf, C, s->ev, W, io, s->need_reset, s->diag, e0);
      so101::launch_pipe_merged(nw, gs, s->dm, P, s->buf, s->ev, W, -1, 0, 0, io, s->need_reset, s->diag, e0);
      for (int k = 0; k < P.n_substeps; k++)
        so101::launch_pipe_merged(ng, gs, s->dm, P, s->buf, s->ev, W, k, (int)(k == P.n_substeps - 1), k + 1, io, s->need_reset, s->diag, e0);
    } else {
      W.mq_ctl = nullptr;
      so101::launch_pipe_begin(ng, gs, s->dm, P, s->buf, C, s->ev, W, io, s->need_reset, s->diag, e0);
      for (int k = 0; k < P.n_substeps; k++) {
        so101::launch_narrow(nw, gs, s->dm, s->n_envs, W, k);
        so101::launch_pipe_solve(ng, gs, s->dm, P, s->buf, s->ev, W, k, (int)(k == P.n_substeps - 1), io, s->need_reset, s->diag, e0);
      }
    }
    LAUNCH_CHECK(s, "k_pipe_solve");
    if (G > 1 && !(hip_ok(s, hipEventRecord(s->group_done[g], gs), "hipEventRecord") &&
                   hip_ok(s, hipStreamWaitEvent(st, s->group_done[g], 0), "hipStreamWaitEvent"))) return SO101_ERR_HIP;
  }
  return SO101_OK;
}

// Buffers that only the experimental step paths need (pipeline 2: four work queues; pipeline 3: one chunk ring per chain; both: the
// contact records indexed [env][candidate]): allocated the first time such a step is requested, outside any capture.
static bool ensure_experimental_buffers(so101_sim* s) {
  if (s->conres_full) return true;
  size_t n = (size_t)s->n_envs;
  bool ok = dev_alloc(s, &s->conres_full, CONRES_DIM * MAXCAND * n, 0, "hipMalloc(chain)");
  // merged launches: one chunk ring per chain (capacity: the next power of two above 64 chunks per env of the chain, so a ring
  // can never wrap onto live granules), head / avail / tail words and per-launch counters per chain
  size_t cap = 64; while (cap < 2 * n * (MAXCAND / NARROW_CHUNK)) cap <<= 1;
  ok = ok && dev_alloc(s, &s->mq_slot, cap, 0, "hipMalloc(merged)") && dev_alloc(s, &s->mq_ctl, (size_t)64 * so101_sim::MAXGROUPS, 0, "hipMalloc(merged)") &&
       dev_alloc(s, &s->mq_pub, (size_t)128 * so101_sim::MAXGROUPS, 0, "hipMalloc(merged)");
  // chained step: narrow chunks (at most MAXCAND / NARROW_CHUNK outstanding per env), solve items (one per env)
  for (int q = 0; q < 4 && ok; q++) {
    size_t need = q < Q_SOLVE ? n * (MAXCAND / NARROW_CHUNK) : n, c = 64;
    while (c < need) c <<= 1;
    ok = dev_alloc(s, &s->chain.qslot[q], c, 0, "hipMalloc(chain)");
    s->chain.qmask[q] = (unsigned int)(c - 1);
  }
  if (!ok) s->conres_full = nullptr;
  return ok;
}

// The per-env chained step (pipeline = 2, csrc/so101_chain.hpp): cost order + classes, prologue, ONE persistent launch.
// k_chain reads its parameters from a device-memory block; sync_chain_params() brings it up to date on `st` BEFORE any
// capture starts (an upload from host memory must not become a graph node).
static ChainParams chain_params_now(so101_sim* s, const so101::StepIO& io) {
  ChainParams cp;
  memset(&cp, 0, sizeof cp);
  cp.m = s->dm; cp.P = make_params(s); cp.B = s->buf; cp.E = s->ev; cp.W = s->pipe; cp.Q = s->chain;
  cp.W.conres = s->conres_full; cp.W.conres_cap = 0u;
  cp.io = SolveIO{io.obs, io.reward, io.discount, io.step_type, s->need_reset, s->diag};
  return cp;
}
static bool sync_chain_params(so101_sim* s, hipStream_t st, const so101::StepIO& io) {
  ChainParams cp = chain_params_now(s, io);
  if (s->chain_params_valid && memcmp(&cp, &s->chain_host, sizeof cp) == 0) return true;
  // the previous contents may still be in use by a step in flight on another stream of the caller: drain first (rare)
  if (!hip_ok(s, hipDeviceSynchronize(), "hipDeviceSynchronize(chain params)")) return false;
  s->chain_host = cp;
  if (!hip_ok(s, hipMemcpy(s->chain_params, &s->chain_host, sizeof cp, hipMemcpyHostToDevice), "hipMemcpy(chain params)")) return false;
  s->chain_params_valid = true;
  return true;
}
static int enqueue_chained(so101_sim* s, hipStream_t st, const so101::StepIO& io) {
  StepParams P = make_params(s);
  PrepBuffers C = prep_view(s);
  int n = s->n_envs;
  so101::launch_order(st, s->pipe.cost, s->pipe.order, s->chain_cls, n);
  LAUNCH_CHECK(s, "k_order");
  if (!hip_ok(s, hipMemsetAsync(s->chain.chain_ctl, 0, sizeof(unsigned int), st), "hipMemsetAsync(chain)")) return SO101_ERR_HIP;
  // (the per-step "abort counted" word; the abort word itself, chain_ctl[32], stays set: see k_chain)
  if (!hip_ok(s, hipMemsetAsync(s->chain.chain_ctl + 33, 0, sizeof(unsigned int), st), "hipMemsetAsync(chain)")) return SO101_ERR_HIP;
  PipeBuffers W = s->pipe;
  W.conres = s->conres_full; W.conres_cap = 0u; W.mq_ctl = nullptr;
  so101::launch_pipe_begin(n, st, s->dm, P, s->buf, C, s->ev, W, io, s->need_reset, s->diag, 0, s->chain_params);
  LAUNCH_CHECK(s, "k_pipe_begin");
  // persistent wavefronts: what fills the machine at 2 per SIMD (8 per CU x 256 CUs), fewer for small batches
  int waves = s->cfg.chain_waves > 0 ? s->cfg.chain_waves : 2048;
  if (waves > 2 * n + 6) waves = 2 * n + 6;
  so101::launch_chain(waves, st, s->chain_params);
  LAUNCH_CHECK(s, "k_chain");
  return SO101_OK;
}

static void drop_graph(so101_sim* s) {
  if (s->graph_exec) { (void)hipGraphExecDestroy(s->graph_exec); s->graph_exec = nullptr; }
}

int so101_step(so101_sim* s, const float* action, float* obs, float* reward, float* discount, uint8_t* step_type, void* stream) {
  REQUIRE_BOUND(s);
  if (!action || !obs || !reward || !discount || !step_type) { s->err = "so101_step: NULL argument"; return SO101_ERR_ARG; }
  GUARD_DEVICE(s);
  hipStream_t st = (hipStream_t)stream;
  so101::StepIO io{action, obs, reward, discount, step_type};
  // the pipelined step is a Newton path; PGS (107 ms per control step at 4096 envs) runs the fused kernel
  if (s->cfg.pipeline && s->cfg.n_substeps <= MAXSUB && s->cfg.solver == SO101_SOLVER_NEWTON) {
#ifndef SO101_EXPERIMENTAL_PIPELINES
    if (s->cfg.pipeline >= 2) {
      s->err = "so101_step: pipeline 2 (per-env chaining) and 3 (merged launches) are experimental step paths that the default library does not carry: "
               "python -m so101_sim_amd.build --experimental builds libso101_hip_exp.so with them";
      return SO101_ERR_STATE;
    }
#endif
    const bool chained = s->cfg.pipeline == 2;
    s->last_path = s->cfg.pipeline;
    if (s->cfg.pipeline >= 2 && !ensure_experimental_buffers(s)) return SO101_ERR_HIP;
    if (chained && !sync_chain_params(s, st, io)) return SO101_ERR_HIP;
    // The launch sequence of a control step (launch chains: ~90 kernels, memsets and event edges over 5 streams; chained: 3
    // kernels and a memset) depends only on the configuration and the caller's pointers: it is captured ONCE into a HIP
    // graph and replayed with one call per step.  Any change of configuration, bound buffers, pool or I/O pointers
    // re-captures.  The legacy null stream cannot capture: the step then runs as plain launches, and capture is tried again
    // when the caller comes with another stream.
    if (s->cfg.use_graph && !(s->graph_failed && s->graph_failed_stream == st)) {
      bool same = s->graph_exec && s->graph_gen == s->generation && s->graph_io.action == io.action && s->graph_io.obs == io.obs &&
                  s->graph_io.reward == io.reward && s->graph_io.discount == io.discount && s->graph_io.step_type == io.step_type;
      if (!same) {
        drop_graph(s);
        hipGraph_t g = nullptr;
        bool ok = false;
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
          int rc = chained ? enqueue_chained(s, st, io) : enqueue_pipelined(s, st, io);
          hipError_t e = hipStreamEndCapture(st, &g);
          if (rc == SO101_OK && e == hipSuccess && g && hipGraphInstantiate(&s->graph_exec, g, nullptr, nullptr, 0) == hipSuccess) {
            s->graph_gen = s->generation; s->graph_io = io; ok = true;
          } else s->graph_exec = nullptr;
          if (g) (void)hipGraphDestroy(g);
        }
        s->graph_failed = !ok; s->graph_failed_stream = st;
        if (!ok) (void)hipGetLastError();
      }
      if (s->graph_exec) {
        if (!hip_ok(s, hipGraphLaunch(s->graph_exec, st), "hipGraphLaunch")) return SO101_ERR_HIP;
        s->last_graph = true;
        launch_prepare(s, st);
        return SO101_OK;
      }
    }
    s->last_graph = false;
    int rc = chained ? enqueue_chained(s, st, io) : enqueue_pipelined(s, st, io);
    if (rc != SO101_OK) return rc;
    launch_prepare(s, st);
    return SO101_OK;
  }
  s->last_path = 0; s->last_graph = false;
  so101::launch_step(s->cfg.solver, s->n_envs, st, s->dm, make_params(s), s->buf, prep_view(s), s->ev, io, s->need_reset, s->diag);
  LAUNCH_CHECK(s, "k_step");
  launch_prepare(s, st);
  return SO101_OK;
}

long long so101_get_info(so101_sim* s, int what, void* stream) {
  if (!s) return -1;
  switch (what) {
    case SO101_INFO_GRAPH_ACTIVE: return s->last_graph ? 1 : 0;
    case SO101_INFO_STEP_PATH: return s->last_path;
    case SO101_INFO_CHAINS: return s->last_chains;
    case SO101_INFO_HW_QUEUES: return getenv("SO101_HW_QUEUES_EFFECTIVE") ? atoi(getenv("SO101_HW_QUEUES_EFFECTIVE"))
                                      : (getenv("GPU_MAX_HW_QUEUES") ? atoi(getenv("GPU_MAX_HW_QUEUES")) : 4);
    case SO101_INFO_SCRATCH_BYTES: return (long long)s->scratch_bytes;
    case SO101_INFO_SCHED_ABORTS: {
      DeviceGuard guard(s);
      unsigned int v = 0;
      if (!guard.ok || !s->chain.chain_ctl) return -1;
      if (hipMemcpyAsync(&v, s->chain.chain_ctl + 32, sizeof v, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return -1;
      if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
      return (long long)v;
    }
    default: return -1;
  }
}

int so101_physics(so101_sim* s, int nsub, int freeze, void* stream) {
  REQUIRE_BOUND(s);
  if (nsub < 0) return SO101_ERR_ARG;
  GUARD_DEVICE(s);
  so101::launch_physics(s->cfg.solver, s->n_envs, (hipStream_t)stream, s->dm, make_params(s), s->buf, nsub, freeze, s->diag);
  LAUNCH_CHECK(s, "k_physics");
  return SO101_OK;
}

int so101_reward(so101_sim* s, float* reward, void* stream) {
  REQUIRE_BOUND(s);
  if (!reward) return SO101_ERR_ARG;
  GUARD_DEVICE(s);
  so101::launch_reward(s->n_envs, (hipStream_t)stream, s->dm, make_params(s), s->buf, reward);
  LAUNCH_CHECK(s, "k_reward");
  return SO101_OK;
}

int so101_get_returns(so101_sim* s, float* out, void* stream) {
  REQUIRE_BOUND(s);
  if (!out) return SO101_ERR_ARG;
  GUARD_DEVICE(s);
  if (!hip_ok(s, hipMemcpyAsync(out, s->buf.ep_return, sizeof(float) * (size_t)s->n_envs, hipMemcpyDeviceToDevice, (hipStream_t)stream), "hipMemcpyAsync(returns)"))
    return SO101_ERR_HIP;
  return SO101_OK;
}

int so101_get_diag(so101_sim* s, int32_t* out, void* stream) {
  if (!s || !out) return SO101_ERR_ARG;
  GUARD_DEVICE(s);
  if (!hip_ok(s, hipMemcpyAsync(out, s->diag, sizeof(int) * SO101_DIAG_DIM * (size_t)s->n_envs, hipMemcpyDeviceToDevice, (hipStream_t)stream), "hipMemcpyAsync(diag)"))
    return SO101_ERR_HIP;
  return SO101_OK;
}

int so101_get_events(so101_sim* s, uint64_t* out, int clear, void* stream) {
  if (!s || !out) return SO101_ERR_ARG;
  GUARD_DEVICE(s);
  hipStream_t st = (hipStream_t)stream;
  if (!hip_ok(s, hipMemcpyAsync(out, s->ev.events, sizeof(uint64_t) * SO101_NEVENTS, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync(events)")) return SO101_ERR_HIP;
  if (clear && !hip_ok(s, hipMemsetAsync(s->ev.events, 0, sizeof(uint64_t) * SO101_NEVENTS, st), "hipMemsetAsync(events)")) return SO101_ERR_HIP;
  return SO101_OK;
}

int so101_debug_forward(so101_sim* s, float* out, void* stream) {
  REQUIRE_BOUND(s);
  if (!out) return SO101_ERR_ARG;
  GUARD_DEVICE(s);
  so101::launch_debug_forward(s->cfg.solver, s->n_envs, (hipStream_t)stream, s->dm, make_params(s), s->buf, out);
  LAUNCH_CHECK(s, "k_debug_forward");
  return SO101_OK;
}

int so101_debug_stages(so101_sim* s, uint32_t* stage, void* stream) {
  if (!s || !stage) return SO101_ERR_ARG;
  GUARD_DEVICE(s);
  return hip_ok(s, hipMemcpyAsync(stage, s->pipe.stage, sizeof(int) * 8 * (size_t)s->n_envs, hipMemcpyDeviceToDevice, (hipStream_t)stream), "hipMemcpyAsync(debug)") ? SO101_OK : SO101_ERR_HIP;
}

int so101_debug_candidates(so101_sim* s, int32_t* ncand, uint32_t* cand, uint32_t* ticks, float* conres, void* stream) {
  if (!s) return SO101_ERR_ARG;
  GUARD_DEVICE(s);
  size_t n = (size_t)s->n_envs;
  hipStream_t st = (hipStream_t)stream;
  bool ok = true;
  if (ncand) ok = ok && hip_ok(s, hipMemcpyAsync(ncand, s->pipe.ncand, sizeof(int) * n, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync(debug)");
  if (cand) ok = ok && hip_ok(s, hipMemcpyAsync(cand, s->pipe.cand, sizeof(int) * MAXCAND * n, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync(debug)");
#ifdef SO101_DEBUG_CLOCKS
  if (ticks) ok = ok && hip_ok(s, hipMemcpyAsync(ticks, s->pipe.ticks, sizeof(int) * MAXCAND * n, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync(debug)");
#else
  if (ticks) { s->err = "so101_debug_candidates: per-candidate ticks exist in profiling builds only (python -m so101_sim_amd.build --clocks)"; return SO101_ERR_STATE; }
#endif
  // contact records: [conres_cap of the slices][24] at work-list positions (launch chains), see csrc/so101_model.hpp PipeBuffers
  if (conres) ok = ok && hip_ok(s, hipMemcpyAsync(conres, s->pipe.conres, sizeof(float) * CONRES_DIM * CONRES_PER_ENV * n, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync(debug)");
  return ok ? SO101_OK : SO101_ERR_HIP;
}

int so101_debug_chain_stats(so101_sim* s, uint64_t* out, int clear, void* stream) {
  if (!s || !out) return SO101_ERR_ARG;
  GUARD_DEVICE(s);
  hipStream_t st = (hipStream_t)stream;
  if (!hip_ok(s, hipMemcpyAsync(out, s->chain.stats, sizeof(uint64_t) * 16, hipMemcpyDeviceToHost, st), "hipMemcpyAsync(chain stats)")) return SO101_ERR_HIP;
  if (clear && !hip_ok(s, hipMemsetAsync(s->chain.stats, 0, sizeof(uint64_t) * 16, st), "hipMemsetAsync(chain stats)")) return SO101_ERR_HIP;
  return hip_ok(s, hipStreamSynchronize(st), "hipStreamSynchronize") ? SO101_OK : SO101_ERR_HIP;
}

const char* so101_last_error(const so101_sim* s) { return s ? s->err.c_str() : g_create_error.c_str(); }

}  // extern "C"
