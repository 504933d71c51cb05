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
#define MAXCON 32
#define MAXARMCON MAXCON
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
  const float* vx;            // hull vertices, struct-of-arrays, geom frame
  const float* vy;
  const float* vz;
  const int* pair;            // [npair][2] statically filtered candidate geom pairs, sorted
  const unsigned int* pair_packed;   // [npair] geom1 | geom2 << 8 | plane flag << 16 (types ordered), same order
};

// Per-env launch parameters that are not part of the model.
struct StepParams {
  float action_offset[NU];
  int last_step, n_substeps, iterations, settle_max, terminate_on_success, n_envs, phases, solver;
  float tolerance;
  unsigned long long seed, env_id_base;
};

struct Contact {
  float pos[3], frame[9], dist;
  int d1, d2, dim, armslot;   // dyn index of geom1/geom2 body (-1 static); slot in the arm pool or -1
  int g1, g2;
  float fric[3];              // tangential, torsional, rolling
  float R[4];                 // normal, tangential, torsional, rolling
  float aref[6], f[6];
  float A[21];                // diagonal block of AR = J Minv J^T + R, packed lower triangle
  float Q[25], lam[5];        // eigen-decomposition D Ac D = Q diag(lam) Q^T of the mu-scaled friction block
  float mu, pad;
};

struct ArmCon {                // rows of an arm-link contact in joint space: J(link of geom2) - J(link of geom1)
  float J[6][NARM];
};

struct Row1 {                  // scalar rows: dof frictionloss and joint limits
  int dof; float sign, R, aref, f, floss, Ainv, pad;
};

struct NewtonScratch {          // working set of solve_newton()
  float H[NVS][NVS + 1];
  float x[NVS], xs[NVS], xw[NVS], tmp[NVS], mxd[NVS], grad[NVS], search[NVS], mxs[NVS];
  float Jst[6][12], Wst[6][12], Hst[6][6];
  float jtf[MAXCON][12];
  float rowf[MAXROW1], rowh[MAXROW1];
  int zone[MAXCON];
};

struct EnvLDS {
  float qpos[NQ], qvel[NV], ctrl[NU], warm[NV];
  float xpos[NDYN][3], xmat[NDYN][9], xipos[NDYN][3], axis[NARM][3], Iw[NDYN][6];
  float Marm[NARM][NARM], Minv[NARM][NARM];
  float fminv[NFREE], fIinv[NFREE][6], fvel[NFREE][6], facc[NFREE][6];   // COM-twist coordinates
  float tau[NARM], bias[NARM], qacc[NV], qacc_arm[NARM];
  float arm0_q[NARM], arm0_v[NARM];
  // collision-phase scratch (geom boxes, broadphase candidates) shares storage with the arm-contact Jacobian
  // pool: the pool is first written by make_constraints(), after collision() has consumed boxes and candidates
  union {
    struct { float aabb[MAXGEOM][6]; unsigned short cand[MAXCAND][2]; };
    ArmCon armcon[MAXARMCON];
  };
  Contact con[MAXCON];
  Row1 row[MAXROW1];
  NewtonScratch nw;
  int ncand, ncon, nrow, narmcon, iters, overflow;
  unsigned int t_collision, t_solve, t_begin;   // stage clocks, 10 ns ticks
  float scratch[64];
};
