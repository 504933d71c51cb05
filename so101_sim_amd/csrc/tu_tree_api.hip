// Public entry points of the general-tree engine (include/so101.h, so101_tree_*): a handle names one of the two builds of
// csrc/so101_tree.hpp - 32 dofs / 128 geoms / 64 contacts (tu_tree.hip: the ALOHA hand-over scenes) or 64 dofs / 256 geoms / 128 contacts
// (tu_tree64.hip: the Dining scenes) - chosen by so101_tree_create from the model's dimensions; every other call forwards.
#include <cstddef>
#include <cstdint>
#include <string>
#include "so101_blob.hpp"
#include "../../include/so101.h"

#define TREE_BUILD_DECL(V)                                                                                                                   \
  int so101_tree##V##_create(const void*, size_t, int, int, void**);                                                                         \
  void so101_tree##V##_destroy(void*);                                                                                                        \
  const char* so101_tree##V##_last_error(const void*);                                                                                        \
  int so101_tree##V##_dims(const void*, int*);                                                                                                \
  int so101_tree##V##_last_plan(const void*, int*);                                                                                            \
  int so101_tree##V##_bind_state(void*, float*, float*, float*, float*);                                                                      \
  int so101_tree##V##_configure(void*, int, float);                                                                                           \
  int so101_tree##V##_physics(void*, int, void*);                                                                                             \
  int so101_tree##V##_debug_forward(void*, float*, void*);                                                                                    \
  int so101_tree##V##_get_diag(void*, int32_t*, void*);                                                                                       \
  int so101_tree##V##_obs_dim(const void*);                                                                                                   \
  int so101_tree##V##_bind_env(void*, float*, float*, float*, int32_t*, int32_t*);                                                            \
  int so101_tree##V##_bind_physics_state(void*, float*, float*, float*);                                                                      \
  int so101_tree##V##_configure_env(void*, const so101_tree_config*);                                                                         \
  int so101_tree##V##_reset(void*, const uint8_t*, void*);                                                                                    \
  int so101_tree##V##_compute_settled(void*, int, int, float*, float*, float*, int32_t*, void*);                                              \
  int so101_tree##V##_set_settled_store(void*, int, int, const float*, const float*, const float*, const int32_t*);                           \
  int so101_tree##V##_settle(void*, void*);                                                                                                   \
  int so101_tree##V##_begin_episode(void*, void*);                                                                                            \
  int so101_tree##V##_step(void*, const float*, float*, float*, float*, uint8_t*, void*);

// (the builds define these with their own handle type in place of void*: same C symbol, same calling convention)
extern "C" {
TREE_BUILD_DECL(32)
TREE_BUILD_DECL(64)
}

struct so101_tree { int variant; void* impl; };

namespace {
thread_local std::string g_api_error;
}

#define FWD(name, ...) (s->variant == 64 ? so101_tree64_##name(s->impl, ##__VA_ARGS__) : so101_tree32_##name(s->impl, ##__VA_ARGS__))

extern "C" {

int so101_tree_create(const void* blob, size_t bytes, int n_envs, int hip_device, so101_tree** out) {
  if (!out) return SO101_ERR_ARG;
  *out = nullptr;
  if (!blob || n_envs <= 0) { g_api_error = "so101_tree_create: bad argument"; return SO101_ERR_ARG; }
  BlobView b;
  if (!b.parse(blob, bytes, g_api_error)) return SO101_ERR_MODEL;
  for (const char* n : {"nq", "nv", "ngeom"})
    if (b.count(n) < 1) { g_api_error = std::string("blob entry missing (not a general-tree model?): ") + n; return SO101_ERR_MODEL; }
  // the small build whenever the model fits it: half the LDS per env, twice the resident envs
  bool small = b.I("nq")[0] <= 40 && b.I("nv")[0] <= 32 && b.I("ngeom")[0] <= 128;
  void* impl = nullptr;
  int rc = small ? so101_tree32_create(blob, bytes, n_envs, hip_device, &impl) : so101_tree64_create(blob, bytes, n_envs, hip_device, &impl);
  if (rc != SO101_OK) { const char* m = small ? so101_tree32_last_error(nullptr) : so101_tree64_last_error(nullptr); g_api_error = m ? m : "?"; return rc; }
  *out = new so101_tree{small ? 32 : 64, impl};
  return SO101_OK;
}

void so101_tree_destroy(so101_tree* s) {
  if (!s) return;
  if (s->variant == 64) so101_tree64_destroy(s->impl); else so101_tree32_destroy(s->impl);
  delete s;
}

const char* so101_tree_last_error(const so101_tree* s) {
  if (!s) return g_api_error.c_str();
  return s->variant == 64 ? so101_tree64_last_error(s->impl) : so101_tree32_last_error(s->impl);
}

int so101_tree_dims(const so101_tree* s, int* dims) { return s ? FWD(dims, dims) : SO101_ERR_ARG; }
int so101_tree_last_plan(const so101_tree* s, int* out) { return s ? FWD(last_plan, out) : SO101_ERR_ARG; }
int so101_tree_bind_state(so101_tree* s, float* qpos, float* qvel, float* ctrl, float* warmstart) { return s ? FWD(bind_state, qpos, qvel, ctrl, warmstart) : SO101_ERR_ARG; }
int so101_tree_configure(so101_tree* s, int solver_iterations, float solver_tolerance) { return s ? FWD(configure, solver_iterations, solver_tolerance) : SO101_ERR_ARG; }
int so101_tree_physics(so101_tree* s, int n_substeps, void* stream) { return s ? FWD(physics, n_substeps, stream) : SO101_ERR_ARG; }
int so101_tree_debug_forward(so101_tree* s, float* out, void* stream) { return s ? FWD(debug_forward, out, stream) : SO101_ERR_ARG; }
int so101_tree_get_diag(so101_tree* s, int32_t* out, void* stream) { return s ? FWD(get_diag, out, stream) : SO101_ERR_ARG; }
int so101_tree_obs_dim(const so101_tree* s) { return s ? FWD(obs_dim) : 0; }
int so101_tree_bind_env(so101_tree* s, float* ring_pos, float* ring_vel, float* ep_return, int32_t* step_count, int32_t* episode) {
  return s ? FWD(bind_env, ring_pos, ring_vel, ep_return, step_count, episode) : SO101_ERR_ARG;
}
int so101_tree_bind_physics_state(so101_tree* s, float* ring, float* physics_state, float* delayed) { return s ? FWD(bind_physics_state, ring, physics_state, delayed) : SO101_ERR_ARG; }
int so101_tree_configure_env(so101_tree* s, const so101_tree_config* cfg) { return s ? FWD(configure_env, cfg) : SO101_ERR_ARG; }
int so101_tree_reset(so101_tree* s, const uint8_t* mask, void* stream) { return s ? FWD(reset, mask, stream) : SO101_ERR_ARG; }
int so101_tree_compute_settled(so101_tree* s, int first_episode, int count, float* qpos, float* qvel, float* warmstart, int32_t* flags, void* stream) {
  return s ? FWD(compute_settled, first_episode, count, qpos, qvel, warmstart, flags, stream) : SO101_ERR_ARG;
}
int so101_tree_set_settled_store(so101_tree* s, int first_episode, int count, const float* qpos, const float* qvel, const float* warmstart, const int32_t* flags) {
  return s ? FWD(set_settled_store, first_episode, count, qpos, qvel, warmstart, flags) : SO101_ERR_ARG;
}
int so101_tree_settle(so101_tree* s, void* stream) { return s ? FWD(settle, stream) : SO101_ERR_ARG; }
int so101_tree_begin_episode(so101_tree* s, void* stream) { return s ? FWD(begin_episode, stream) : SO101_ERR_ARG; }
int so101_tree_step(so101_tree* s, const float* action, float* obs, float* reward, float* discount, uint8_t* step_type, void* stream) {
  return s ? FWD(step, action, obs, reward, discount, step_type, stream) : SO101_ERR_ARG;
}

}  // extern "C"
