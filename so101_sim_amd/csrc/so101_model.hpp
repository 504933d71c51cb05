// Device-side model and per-env LDS layout of the SO100 hand-over step.
//
// The kernels are specialised to the topology every SO100HandOver scene has
// (so101_sim/assets/so100/scene_pbr.xml:74-146 + two free props, so100_hand_over.py:169,199):
// ONE serial chain of NARM hinge links on a static base + NFREE free bodies + static geoms.
// so101_create() rejects blobs outside these limits.
#pragma once
#include <stdint.h>

#define NARM 6
#define NFREE 2
#define NDYN (NARM + NFREE)
#define NQ (NARM + 7 * NFREE)
#define NV (NARM + 6 * NFREE)
#define NU 6
#define MAXGEOM 96
#define MAXCAND 256
#define MAXCON 64                 // contacts per env (one lane each in the Newton solver)
#define MAXCON_PGS 32             // the PGS kernels keep the lane layout contacts 0..31 | scalar rows 32..43
#define MAXARMCON 40              // contacts that touch an arm link (their joint-space Jacobian rows live in an LDS pool)
#define NCPP 5                    // contacts per candidate pair: deepest point + up to 4 samples of a flat patch
#define MAXROW1 (2 * NARM)
#define NVS NV                    // solver coordinates: arm qacc | object twist | container twist

enum { G_PLANE = 0, G_SPHERE = 1, G_CAPSULE = 2, G_CYLINDER = 3, G_BOX = 4, G_MESH = 5 };

struct DevModel {
  int ngeom, npair, nvert, iterations, mpr_iter, nbox, pad0, pad1;
  float dt, grav[3], impratio, tolerance, mpr_tol, meaninertia;
  // arm chain --------------------------------------------------------------------------
  float base_pos[3], base_quat[4];
  float arm_pos[NARM][3], arm_quat[NARM][4], arm_axis[NARM][3], arm_ipos[NARM][3];
  float arm_Ib[NARM][6];      // inertia about the COM in the body frame: xx yy zz xy xz yz
  float arm_mass[NARM], armature[NARM], frictionloss[NARM], range[NARM][2];
  int limited[NARM];
  float jnt_solref[2], jnt_solimp[5], dof_solref[2], dof_solimp[5];
  float dof_invweight0[NV];
  float act_gain[NU], act_bias[NU][3], ctrlrange[NU][2], forcerange[NU][2];
  int ctrllimited[NU], forcelimited[NU];
  // free bodies ------------------------------------------------------------------------
  float free_mass[NFREE], free_ipos[NFREE][3], free_iquat[NFREE][4];
  float free_Ib[NFREE][6], free_Ibinv[NFREE][6], free_bvh[NFREE][6];
  float dyn_invweight0[NDYN][2];
  // task --------------------------------------------------------------------------------
  float box_pos[2][3], box_half[2][3];
  float obj_lo[3], obj_hi[3], obj_yaw[2], con_lo[3], con_hi[3], home_ctrl[NU];
  // geoms (device pointers) ---------------------------------------------------------------
  const int* geom_type;
  const int* geom_dyn;        // 0..5 arm link, 6.. free body, -1 static
  const int* geom_condim;
  const int* geom_vertadr;
  const int* geom_vertnum;
  const float* geom_pos;      // [g][3]  body-local (dynamic) or world (static)
  const float* geom_mat;      // [g][9]
  const float* geom_size;     // [g][3]
  const float* geom_friction; // [g][3]
  const float* geom_solref;   // [g][2]
  const float* geom_solimp;   // [g][5]
  const float* geom_center;   // [g][3]  MPR interior point, geom frame
  const float* geom_aabb;     // [g][6]  centre, half in the geom frame
  const float* geom_rbound;   // [g]     bounding-sphere radius (0 for the plane)
  const float* vx;            // hull vertices, struct-of-arrays, geom frame
  const float* vy;
  const float* vz;
  const int* pair;            // [npair][2] statically filtered candidate geom pairs, sorted
  const unsigned int* pair_packed;   // [npair] geom1 | geom2 << 8 | plane flag << 16 (types ordered), same order
  // support-bound tables of the mesh geoms (round 6; NULL: none): hull_sbt[g][SBT_DIM] = max over the hull's vertices of v . c for the 6 x 5 x 5
  // grid points c of the cube [-1, 1]^3's faces (face 2 a + (sign < 0), u and v in steps of 0.5 along axes a + 1, a + 2), rounded up.  The support
  // function is convex and positively homogeneous, so for any direction d the bilinear interpolation over the four grid points around
  // d / |d|_inf, times |d|_inf, is an UPPER bound of h(d) - a few per cent of the hull's size above it, against the tens of per cent of an
  // oriented box around a curved shell piece.  Built at so101_create from the blob's vertices (so101_hip.hip); used by obb_filter.
  const float* hull_sbt;
  // support-vertex lists of the mesh geoms (round 6; NULL: none).  The cube-map of directions is cut into 6 x HL_GRID x HL_GRID cells; hl_off[g * (HL_CELLS
  // + 1) + c] .. [.. + c + 1] is the range of cell c's entries in hl_entry, four floats each: x, y, z of a vertex (the floats of vx / vy / vz) and its
  // index within the hull (bits).  A cell's list holds, in increasing index order, every vertex that can be the support point for SOME direction of the
  // cell widened by 4e-3 rad: a vertex v is left out only when one of the four vertices that win at the cell's corners beats it at all four corners by
  // more than the widening - then it loses everywhere in between by linearity (so101_hip.hip).  Median 10-17 of a hull's 200-1080 vertices.
  const float* hl_entry;
  const unsigned int* hl_off;
};
#ifndef SBT_GRID
#define SBT_GRID 5
#endif
#define SBT_DIM (6 * SBT_GRID * SBT_GRID)
#define HL_GRID 8
#define HL_CELLS (6 * HL_GRID * HL_GRID)
#define HL_MAX 128               // a list longer than two vertices per lane is not used (the query scans the whole hull as before)

// Per-env launch parameters that are not part of the model.
struct StepParams {
  float action_offset[NU];
  int last_step, n_substeps, iterations, settle_max, terminate_on_success, n_envs, phases, solver;
  float tolerance;
  unsigned long long seed, env_id_base;
};

struct Contact {               // 40 words
  float pos[3], frame[9], dist;
  int d1, d2, dim, armslot;   // dyn index of geom1/geom2 body (-1 static); slot in the arm pool or -1
  int g1, g2;
  float fric[3];              // tangential, torsional, rolling
  float R[4];                 // normal, tangential, torsional, rolling
  float aref[6], f[6];        // (contact_init() parks the mixed solref in aref[0..1] and solimp in f[0..4])
  float mu, pad;
};

struct ArmCon {                // arm-link contact in joint space, dof-major: Jt[q][j] = (J(link of geom2) - J(link of geom1))[row j][dof q]
  float Jt[NARM][6];
};

struct Row1 {                  // scalar rows: dof frictionloss and joint limits
  int dof; float sign, R, aref, f, floss, Ainv, pad;
};

struct NewtonScratch {          // LDS working set of solve_newton(): the vectors every lane reads, and the transposed factor
  float H[NVS][NVS + 1];
  float x[NVS], xs[NVS], xw[NVS], tmp[NVS], search[NVS];
#ifdef SO101_DEBUG_CLOCKS
  unsigned int prof[16];       // profiling builds: wall-clock ticks per solver phase (setup, gradient, Hessian, factor, solve, line search, cost, -)
                               // and, [8..], of the next substep's broadphase (state store, kinematics, geom boxes, pair list, oriented boxes, publish)
#endif
};

struct EnvLDS {
  float qpos[NQ], qvel[NV], ctrl[NU], warm[NV];
  float xpos[NDYN][3], xmat[NDYN][9], xipos[NDYN][3], axis[NARM][3], Iw[NDYN][6];
  float Marm[NARM][NARM], Minv[NARM][NARM];
  float fminv[NFREE], fIinv[NFREE][6], fvel[NFREE][6], facc[NFREE][6];   // COM-twist coordinates
  float fscale[NFREE], fmass[NFREE];      // per-env mass / inertia scale of the free props (domain randomisation), scaled mass
  float tau[NARM], bias[NARM], qacc[NV], qacc_arm[NARM];
  float arm0_q[NARM], arm0_v[NARM];
  // collision-phase scratch (geom boxes, broadphase candidates) shares storage with the arm-contact Jacobian
  // pool: the pool is first written by make_constraints(), after collision() has consumed boxes and candidates
  union {
    // world boxes component-major (lo xyz, hi xyz): lanes test consecutive geoms, so consecutive lanes read consecutive
    // banks (geom-major rows of 32 bytes were read with ds_read_b32 at a 32-byte lane stride: 8-way bank conflicts,
    // 11.5 of the 13.5 us the pair list cost per env-substep)
    struct { float aabb[6][MAXGEOM]; unsigned short cand[MAXCAND][2]; };
    ArmCon armcon[MAXARMCON];
  };
  Contact con[MAXCON];
  Row1 row[MAXROW1];
  NewtonScratch nw;
  int ncand, ncon, nrow, narmcon, iters, overflow;
  unsigned int t_collision, t_solve, t_begin;   // stage clocks, 10 ns ticks (SO101_DEBUG_CLOCKS builds)
  float scratch[64];
};

// Caller-owned per-env state (so101_buffers of include/so101.h), struct-of-arrays with the env index fastest.
struct DevBuffers {
  float *qpos, *qvel, *ctrl, *warm, *ring, *ep_return;
  int *step_count, *episode;
  const float* mass_scale;     // [NFREE][n_envs] or NULL (= 1)
  // optional (so101_bind_physics_state): the physics_state observable and its 15-step delayed copy (so100_task.py:203-210,366-368)
  float *ps_ring, *ps_out, *ps_delayed;      // [PS_DELAY][NQ + NV][n_envs], [n_envs][NQ + NV], [n_envs][NQ + NV]
};
#define PS_DELAY 15            // delayed_physics_state: 0.3 s = 15 control steps
#define PS_DIM (NQ + NV)

// Library-owned cache of settled initial states.  The settled state of an episode is a pure function of
// (seed, global env id, episode index, config), so it can be computed ahead of time: k_prepare() fills the
// cache on a side stream while the envs are stepping and env_reset() consumes an entry when its tag matches the
// episode that is about to start; otherwise env_reset() settles in place.  Either way the result is the same
// bits, only the time at which the work is done differs.
struct PrepBuffers {
  // Two entries per env (slot = episode & 1): the next episode's settled state and the one after it.  When every env
  // passes its time limit on the same control step, the whole cache is consumed at once and refilling it takes ~2 s in
  // the background; an env whose physics diverge inside that window resets a second time, and with one entry per env it
  // had to settle inside the step call (one wavefront, 1000 substeps, ~130 ms - measured 42 ms per control step over the
  // 50 steps after a mass reset).  The second entry covers that reset.
  float *qpos, *qvel, *warm;   // [2][NQ|NV|NV][n_envs]
  int *tag;                    // [2][n_envs] episode index the entry belongs to, -1 = empty
  int *cursor;                 // work-queue head of k_prepare
  int *flags;                  // [2][n_envs] flag word of the settle that produced the entry (placement / settle failures)
  // Reset pool (so101_set_reset_pool): when pool_size > 0 an episode starts from pool entry
  // floor(u * pool_size), u = rng_uniform(seed, env id, episode, draw 1000), instead of placement + settle
  const float *pool_qpos, *pool_qvel, *pool_ctrl;   // [NQ|NV|NU][pool_size]
  int pool_size;
  // Settled-state store (so101_set_settled_store): the results of placement + settle for episodes
  // store_first .. store_first + store_count - 1 of every env, computed once (so101_compute_settled) and kept across
  // runs by the caller.  Entries are what env_settle() would produce, so a reset that finds one is bit-identical.
  const float *store_qpos, *store_qvel, *store_warm;   // [store_count][NQ|NV|NV][n_envs]
  const int* store_flags;                               // [store_count][n_envs]
  int store_first, store_count;
  // k_prepare, second look-ahead (the episode after the next one): envs scan_first .. scan_first + scan_count - 1 (mod n_envs) of this launch;
  // scan_count = 0: all envs (the default: slices were measured slower, so101_hip.hip launch_prepare)
  int scan_first, scan_count;
};

// Event accounting (so101_get_events): flags[e] ORs the per-substep flag words of env e within one control step;
// events[b] counts env-steps (bits 0-3) or env-resets (bits 4-5) on which bit b was set.
//   bit 0 broadphase candidate overflow, 1 contact overflow (MAXCON), 2 arm-contact pool overflow, 3 physics
//   diverged (mj_check*: episode ends like a dm_control PhysicsError), 4 container placement rejected 20 times,
//   5 settle did not converge within its budget
struct EventBuffers {
  int* flags;                        // [n_envs]
  unsigned long long* events;        // [SO101_NEVENTS]
};
#define SO101_NEVENTS 8

#define MAXSUB 32
#define CONRES_DIM (4 + 4 * NCPP)
#define CONRES_PER_ENV 48       // compact contact-record pool: records per env of a slice (mean use 12) + MAXCAND per slice (one env can always place all its candidates); overflow drops candidates and is counted

// work items of the narrowphase launches (so101_pipeline.hpp, publish_candidates): word offsets
#define ITEM_WORDS 48            // 192 bytes: rec, env * MAXCAND + k, geom 1 at word ITEM_GEOM0, geom 2 at ITEM_GEOM1
#define ITEM_GEOM0 2
#define ITEM_GEOM1 24
#define ITEM_G_TYPE 0
#define ITEM_G_VADR 1
#define ITEM_G_VNUM 2
#define ITEM_G_SIZE 3
#define ITEM_G_R 6
#define ITEM_G_P 15
#define ITEM_G_C 18
#define ITEM_G_RBOUND 21

// Scratch of the pipelined step (so101_pipeline.hpp)
struct PipeBuffers {
  float* pose;            // [N][NDYN][12] xpos, xmat of the dynamic bodies
  unsigned int* cand;     // [N][MAXCAND]  geom1 | geom2 << 16, in pair-list order
  int* ncand;             // [N]           count | broadphase overflow flag << 16
  unsigned int* items;    // [conres_cap][ITEM_WORDS] work items of the current substep's narrowphase launch, per env slice (so101_pipeline.hpp)
  int* counters;          // [MAXSUB][2]   work items, cursor
  float* conres;          // narrowphase results: count, normal, NCPP x (dist, position) per candidate.  Launch chains: one record
                          // per work-list position of the current substep, [conres_cap][CONRES_DIM] per env slice (an env's
                          // records start at cbase[e]); conres_cap = 0 (pipelines 2, 3): [N][MAXCAND][CONRES_DIM]
  unsigned int conres_cap;
  unsigned int narrow_chunk;   // k_narrow: candidate pairs per work-item fetch (1 .. NARROW_CHUNK), set per launch by the host
  int* cbase;             // [N] work-list position of the env's first candidate in the current substep
  unsigned char* active;  // [N] 0 not stepping in this call (auto-reset), 1 stepping, 2 diverged
  unsigned int* stage;    // [N][8] k_pipe_solve stage clocks of the last substep (10 ns ticks; SO101_DEBUG_CLOCKS builds)
  unsigned int* cost;     // [N] solver time of the env in its last substep (ticks): scheduling hint only
  int* order;             // [N] per group: env indices sorted by decreasing cost, see k_order
  unsigned int* ticks;    // [N][MAXCAND] narrowphase time of each candidate (10 ns ticks; SO101_DEBUG_CLOCKS builds)
  // merged launches (pipeline = 3, so101_chain.hpp): the chain's chunk queue and per-launch counters
  unsigned int* mq_ctl;   // [64] head, avail, tail (QC_*)
  unsigned long long* mq_slot;
  unsigned int mq_mask;
  unsigned int* mq_pub;   // [128] per launch of the chain: [l] wavefronts past their publish point, [64 + l] lingering wavefronts; [127] abort
  float* state;           // [N][STATE_AOS] qpos | qvel | warm | ctrl of an env, contiguous: the substep round trips of the
                          // pipelined step (the caller's buffers are env-fastest struct-of-arrays: with one env per
                          // wavefront every scalar is its own 64-byte line; they are read once and written once per call)
};

// ---- per-env chaining (pipeline = 2, so101_chain.hpp): work queues of the persistent step kernel
struct ChainQueues {
  unsigned int* pending;  // [N] substep << 16 | narrowphase chunks of that substep still outstanding
  const unsigned char* cls;   // [N] 1 = among the most expensive eighth of the previous control step (served first), by k_order
  unsigned int* qctl;     // [4][64] per queue (narrow lo, narrow hi, solve lo, solve hi): word 0 head, word 32 tail
  unsigned long long* qslot[4];   // ring slots: item | (ticket + 1) << 32
  unsigned int qmask[4];
  unsigned int* chain_ctl;        // [0] envs that finished this control step, [32] watchdog aborts
  int idle_sleeps;                // s_sleep instructions of an idle round (each ~2 k cycles)
  int role_mode;                  // how a wavefront picks the kind of work it looks for first (tu_chain.hip)
  unsigned long long* stats;      // [16] wall-clock ticks (10 ns) and counts summed over the wavefronts of k_chain, see CS_*
};
// k_chain statistics (so101_debug_chain_stats): ticks spent claiming work that was found, looking for work that was not,
// in narrowphase chunks, in solve items; item counts; launches; wavefront lifetimes
enum { CS_T_POP = 0, CS_T_IDLE, CS_T_NARROW, CS_T_SOLVE, CS_N_NARROW, CS_N_SOLVE, CS_N_IDLE, CS_WAVES, CS_T_LIFE, CS_N };
struct SolveIO { float* obs; float* reward; float* discount; unsigned char* step_type; unsigned char* need_reset; int* diag; };
#define STATE_AOS 64
#ifndef NARROW_CHUNK
#define NARROW_CHUNK 4     // candidate pairs per narrowphase work item
#endif
#define Q_NARROW 0         // queue index = type + class
#define Q_SOLVE 2

// launch parameters of k_chain, kept in device memory (tu_chain.hip)
struct ChainParams { const DevModel* m; StepParams P; DevBuffers B; EventBuffers E; PipeBuffers W; ChainQueues Q; SolveIO io; };

// debug dump layout (floats) of so101_debug_forward
#define DBG_M 0          // 36  arm mass matrix
#define DBG_MINV 36      // 36
#define DBG_BIAS 72      // 6
#define DBG_SMOOTH 78    // 18  qacc_smooth (generalized)
#define DBG_QACC 96      // 18  qacc after the solve
#define DBG_COUNTS 114   // ncon, nrow, iters, ncand, overflow
#define DBG_XPOS 120     // 8*3 dynamic body positions
#define DBG_CON 144      // MAXCON * 10: pos3 normal3 dist g1 g2 dim
#define DBG_FORCE (DBG_CON + 10 * MAXCON)    // MAXCON * 6
#define DBG_ROWF (DBG_FORCE + 6 * MAXCON)    // MAXROW1
#define DBG_REWARD (DBG_ROWF + 16)
#define DBG_DIM 2048
