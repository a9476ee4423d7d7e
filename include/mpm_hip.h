/* C ABI of the snow-MPM step — the second program of the reference, `./run.sh mpm`.
 *
 * The reference has no plugin / FFI interface for this path either: it is the loop body of main(), mpm.cc:1301-1436,
 * over a PointList (mpm.cc:354-1018) and a handful of OpenVDB grids.  The entry points below are what a binding of that
 * loop would call; each cites the lines it replaces.  Same library as fluid_hip.h (libfluid_hip.so), same conventions:
 * int status codes (0 = ok), no exceptions across the boundary, fluid_last_error() holds the message of the calling
 * thread's last failure, one handle = one host thread + one HIP stream, device memory owned by the handle, host buffers
 * by the caller.  There is no CPU path: every call fails with FLUID_ERR_HIP when no gfx950 device is usable.
 *
 * Grid layout of every downloaded field: dense, cells -B..B per axis (N = 2B+1), z fastest:
 *   index(x, y, z) = ((x + B) * N + (y + B)) * N + (z + B)             (the reference's triple loops, e.g. mpm.cc:1318-1331)
 * 3x3 matrices are row-major, 9 doubles; vectors are AoS xyz like std::vector<openvdb::Vec3d>.
 */
#ifndef MPM_HIP_H
#define MPM_HIP_H
#include <stdint.h>
#include "fluid_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpm_sim mpm_sim_t;

typedef struct {
    int32_t B;              /* cells -B..B                       15        mpm.cc:1023,1274 (pos.initialize(15, ..)) */
    int32_t W;              /* solid where any |c| > W            13        mpm.cc:1149-1157                           */
    int32_t device;         /* HIP device ordinal                                                                      */
    int32_t cg_max_iters;   /* 0 = 2 * 3 * numActive                        IterativeSolverBase.h:362                  */
    double dx;              /* 1.0 * factor                                 mpm.cc:1282                                */
    double gravity[3];      /* (0, -10, 0)                                  mpm.cc:1281                                */
    double youngs_modulus;  /* E = 48000                                    mpm.cc:1391                                */
    double poisson_ratio;   /* nu = 0.47                                    mpm.cc:1392                                */
    double beta;            /* 0.5 (implicitness of the velocity update)    mpm.cc:1393                                */
    double hardening;       /* 10 (the `epsilon` argument)                  mpm.cc:1395, deformHeader.h:254,276        */
    double theta_c;         /* 0.025 critical compression                   mpm.cc:1410                                */
    double theta_s;         /* 0.0075 critical stretch                      mpm.cc:1410                                */
    double max_dt;          /* 0.001 (FLIPadvect's maxTimeStep)             mpm.cc:1417                                */
    double dt0;             /* 0.001 first step                             mpm.cc:1295                                */
    double cg_tol;          /* Eigen default epsilon                        IterativeSolverBase.h:283                  */
    int32_t transpose_system; /* 1 (default): solve A^T x = b like the reference's solver object does — Eigen's
                               ConjugateGradient<SparseMatrix<double>, Lower|Upper> multiplies by the transpose of a
                               column-major matrix (ConjugateGradient.h:202-212), and A = I + beta dt^2 D^-1 K (mpm.cc:418-441,691)
                               is not symmetric.  0: solve A x = b with the matrix as assembled.                          */
    int32_t pad_;
} mpm_params_t;

/* What one pass of the loop body reports (the values the reference prints: "DT", "Max Force ...", "Error: ", "MAX ...") */
typedef struct {
    double dt_in;             /* dt used by this step                      mpm.cc:1313 */
    double dt_out;            /* dt produced by FLIPadvect for the next    mpm.cc:1418 */
    double cg_error;          /* cg.error(): |r| / |b|                     mpm.cc:1405 */
    double max_speed;         /* FLIPadvect's maxSpeed                     mpm.cc:910-927 */
    double max_grad;          /* "MAX" line                                mpm.cc:559-584 */
    double max_fp;
    double max_fe;
    double max_force[3];      /* "Max Force" line                          mpm.cc:377-417 */
    double max_mi;
    double max_force_coeff2;
    int32_t num_active;       /* grid nodes with mass > 0.1                mpm.cc:1346-1364 */
    int32_t cg_iters;         /* iterations of this library's solver (NOT the reference's count: other preconditioner) */
    int32_t any_active;       /* the "yes" flag of the Max Force line */
    int32_t cg_status;        /* how the solve ended: 1 converged (or b = 0), 2 breakdown (<p, A p> <= 0 or NaN: mpm_step_solve
                                 returns FLUID_ERR_SOLVER), 3 stopped at the iteration cap like Eigen does (x is what the last
                                 iteration left; cg_error says how far it got) */
    double ms_transfer, ms_forces, ms_solve, ms_deform, ms_advect, ms_apply_avg;   /* HIP-event times of the phases */
} mpm_step_stats_t;

/* Particle arrays (mpm_download_particles / mpm_set_state) — PointList members, mpm.cc:356-361 */
#define MPM_P_POS 0      /* positions   3 doubles */
#define MPM_P_VEL 1      /* velocities  3 doubles */
#define MPM_P_FE 2       /* FEvec       9 doubles */
#define MPM_P_FP 3       /* FPvec       9 doubles */
#define MPM_P_GRADV 4    /* gradV       9 doubles */
#define MPM_P_VOLUME 5   /* volume      1 double  */

/* Grid fields (mpm_download_field) */
#define MPM_F_CONTAINER 0   /* float   containerGrid: node mass                       mpm.cc:1341 */
#define MPM_F_SOLID 1       /* float   solidGrid (1 = solid)                          mpm.cc:1149-1157 */
#define MPM_F_OUTPUT 2      /* float   outputGrid (written to the .vdb files)         mpm.cc:1366-1380 */
#define MPM_F_INDICES 3     /* int32   unknown number or -1                           mpm.cc:1315,1346-1364 */
#define MPM_F_VEL_BEFORE 4  /* double3 velBeforeUpdate = vels after P2Gtransfer       mpm.cc:1390 */
#define MPM_F_FORCES 5      /* double3 gridForces                                     mpm.cc:1395 */
#define MPM_F_VEL 6         /* double3 vels after updateVelocity                      mpm.cc:1404 (the reference zeroes it at :1431) */

/* Reference literals of mpm.cc (see the field comments). */
int mpm_default_params(mpm_params_t* p);

/* Allocates the grids of mpm.cc:1021-1157,1297-1299 on the device; the solid shell is every cell with a |coordinate| > W. */
int mpm_create(const mpm_params_t* p, mpm_sim_t** out);
int mpm_destroy(mpm_sim_t* s);

/* PointList::add for every point, mpm.cc:471-491: kept iff |p| < B - 2 on every axis; FE = FP = identity, volume = 0,
 * the step counter returns to 0 (findVolume runs in the next step, mpm.cc:1343-1346).  The reference gives every
 * particle the velocity (0, -50, 0) (mpm.cc:484); here the caller passes the velocities (vel may be NULL for that
 * literal).  kept (may be NULL) receives the number of particles kept. */
int mpm_upload_particles(mpm_sim_t* s, int64_t n, const double* pos, const double* vel, int64_t* kept);
int64_t mpm_num_particles(const mpm_sim_t* s);
/* Overwrites FEvec / FPvec / volume (any may be NULL) and the step counter — for restarts and for parity tests that
 * re-synchronise the two sides. */
int mpm_set_state(mpm_sim_t* s, const double* FE, const double* FP, const double* volume, int32_t step_no);
int mpm_set_dt(mpm_sim_t* s, double dt);
double mpm_get_dt(const mpm_sim_t* s);

/* One pass of the loop body mpm.cc:1313-1431: interpolate (mass), P2Gtransfer, findVolume (first step), numbering,
 * output grid, populateGridForces, populateMatrices + cg.compute/solve (matrix-free here), updateVelocity,
 * updateDeformationGradient, FLIPadvect. */
int mpm_step(mpm_sim_t* s, mpm_step_stats_t* stats);
/* The same pass in two halves, so that the linear system can be inspected between them (mpm_apply_matrix,
 * mpm_download_system): mpm.cc:1313-1404 (through updateVelocity), then mpm.cc:1410-1431.  mpm_step = both. */
int mpm_step_solve(mpm_sim_t* s, mpm_step_stats_t* stats);
int mpm_step_advance(mpm_sim_t* s, mpm_step_stats_t* stats);

int mpm_download_particles(mpm_sim_t* s, int32_t what, double* out);
int mpm_download_field(mpm_sim_t* s, int32_t field, void* out);

/* Unknowns of the last mpm_step_solve. */
int32_t mpm_num_active(const mpm_sim_t* s);
/* The linear system of the last step as the reference assembles it (mpm.cc:370-444): right-hand side b and solution x,
 * 3 * num_active doubles each (unknown k holds 3k..3k+2).  `count` = the doubles each buffer holds: it must be exactly
 * 3 * mpm_num_active(s) (FLUID_ERR_ARG otherwise, nothing is written) — a caller that sized its buffers for another step's
 * system cannot be overrun. */
int mpm_download_system(mpm_sim_t* s, double* b, double* x, int64_t count);
/* y = A v (or A^T v when transpose_system = 1) for the step's matrix A = I + beta dt^2 M (mpm.cc:418-441), applied matrix-free by the solver's own
 * kernel; v and y hold 3 * num_active doubles.  Only between mpm_step_solve and mpm_step_advance (afterwards the particles
 * have moved).  For tests: column k of A is mpm_apply_matrix(e_k). */
int mpm_apply_matrix(mpm_sim_t* s, const double* v, double* y);

/* Known-answer hook: the device functions the step kernels use, on n independent 3x3 matrices (row-major, 9 doubles each).
 *   MPM_EVAL_POLAR    a = F                       -> out0 = getR(F), out1 = getS(F)                    deformHeader.h:22-36
 *   MPM_EVAL_SIGMA    a = FE, b = FP, p0 = mu0, p1 = lambda0, p2 = epsilon -> out0 = getSigma(...)     deformHeader.h:273-307
 *   MPM_EVAL_HESSIAN  a = F, b = dF, p0 = lambda, p1 = mu -> out0 = d2Psi/dF2 : dF — dPsydFdF (deformHeader.h:241-249) for the
 *                     dF that getDelFE (:107-132) builds, and for any other dF (the operator is linear in it)
 *   MPM_EVAL_CLAMP    a = (I + dt gradV) FE, b = FP, p0 = 1 - theta_c, p1 = 1 + theta_s -> out0 = FE', out1 = FP'   mpm.cc:543-555
 * No handle needed; runs on the current device. */
#define MPM_EVAL_POLAR 0
#define MPM_EVAL_SIGMA 1
#define MPM_EVAL_HESSIAN 2
#define MPM_EVAL_CLAMP 3
int mpm_eval(int32_t what, int64_t n, const double* a, const double* b, double p0, double p1, double p2, double* out0, double* out1);

/* The reference's scene (mpm.cc:1037-1052,1274-1278): the cone of voxels {(i, j, k): -W <= j <= -W + layers - 1,
 * i^2 + k^2 <= ((j + W) / 2)^2}, UniformPointScatter with points_per_voxel points per voxel and std::mt19937(seed),
 * filtered by PointList::add with boundary B.  layers = 4, points_per_voxel = 400, seed = 0, B = 15, W = 13 is the
 * reference's own.  pos == NULL: returns the number of points only.  Returns < 0 on bad arguments. */
int64_t mpm_scene_cone(int32_t B, int32_t W, int32_t layers, float points_per_voxel, uint32_t seed, double* pos);

#ifdef __cplusplus
}
#endif
#endif
