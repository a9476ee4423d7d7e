/*
 * fluid_hip.h — C ABI of the MI355X-native PIC/FLIP step (libfluid_hip.so).
 *
 * The reference (Aakash1312/Fluid-Simulation) has no plugin/FFI interface: the hot path is
 * the body of the step loop in main() (fluid.cc:1368-1507).  This header is the boundary a
 * maintainer would bind instead of that loop body; every entry point names the reference
 * lines it replaces.  Plain pointers and sizes only; no C++ or torch types; no exceptions
 * cross the boundary (int status, fluid_last_error() for the text).
 *
 * Conventions
 *   grid      N cells per axis, cell coordinate c in [lo,hi], lo = -(N/2), hi = lo+N-1
 *             (N=121 -> the reference's -60..60, fluid.cc:1159).  "W" = [lo+2,hi-2]
 *             (the reference's literal 58, fluid.cc:1264).  Dense layout, z fastest:
 *             linear = ((x-lo)*N + (y-lo))*N + (z-lo)  — the order of the reference's
 *             index numbering sweep (fluid.cc:1416-1433).
 *   particles host side AoS xyz doubles, like std::vector<openvdb::Vec3d> (fluid.cc:806-807).
 *   ownership device memory belongs to the handle; host buffers belong to the caller.
 *   threading one handle = one host thread = one HIP stream; calls are sequential.
 */
#ifndef FLUID_HIP_H
#define FLUID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fluid_sim fluid_sim_t;

/* status codes */
#define FLUID_OK 0
#define FLUID_ERR_ARG 1      /* bad argument                                   */
#define FLUID_ERR_HIP 2      /* HIP runtime error (no device, OOM, launch)     */
#define FLUID_ERR_STATE 3    /* call order (e.g. step before upload_particles) */
#define FLUID_ERR_SOLVER 4   /* PCG hit the iteration cap / broke down         */
#define FLUID_ERR_PEER 5     /* multi-GPU: another rank failed in this step; every rank returns together */

/* field ids for fluid_download_field / fluid_upload_field */
#define FLUID_FIELD_CONTAINER 0   /* float32 N^3  particle weight density (fluid.cc:1157,1413)        */
#define FLUID_FIELD_WEIGHTS 1     /* float32 N^3  P2G weights (fluid.cc:809,1108); == container here  */
#define FLUID_FIELD_VEL 2         /* float64 3*N^3, SoA planes [u|v|w] (Vec3dGrid vels, :1240)        */
#define FLUID_FIELD_VEL_BEFORE 3  /* float64 3*N^3  velBeforeUpdate (:1455)                           */
#define FLUID_FIELD_INDICES 4     /* int32   N^3   unknown numbering, -1 elsewhere (:1388,1416-1433)  */
#define FLUID_FIELD_RHS 5         /* float32 N^3   rhs grid of the last setRHS (:1234,414-479)        */
#define FLUID_FIELD_DIVER 6       /* float32 N^3   b of the last pass = diver grid (:1231,566-610)    */
#define FLUID_FIELD_PRESSURE 7    /* float64 N^3   p scattered back to cells (VectorXd p, :1474)      */
#define FLUID_FIELD_OUTPUT 8      /* float32 N^3   outputGrid written to .vdb (:1434-1448)            */
#define FLUID_FIELD_SOLID 9       /* uint8   N^3   1 = solid (:1166,1266)                             */
#define FLUID_FIELD_DIVER2 14     /* float32 N^3   b2, divergence after the update (:1477-1481)       */
#define FLUID_FIELD_SEARCH 15     /* solver dtype N^3  PCG search vector s (stencil input)            */
#define FLUID_FIELD_Q 16          /* solver dtype N^3  q = A s (stencil output)                       */
#define FLUID_FIELD_FLAGS 17      /* uint8   N^3   bit0 solid, bit1 fluid, bits2-4 diag count         */

/* solver precision */
#define FLUID_PRECISION_FP64 0    /* fp64 PCG vectors (reference arithmetic: Eigen VectorXd)          */
#define FLUID_PRECISION_FP32 1    /* fp32 PCG vectors (stencil micro-benchmark / experiments only)    */

/* preconditioner of the PCG (fluid_params.preconditioner); the reference uses Eigen IncompleteCholesky */
#define FLUID_PRECOND_MG 0        /* geometric multigrid V(2,2) cycle; iteration count independent of N  */
#define FLUID_PRECOND_JACOBI 1    /* Eigen DiagonalPreconditioner arithmetic, iteration-for-iteration Eigen's Jacobi CG */

/* start of every pressure solve (fluid_params.solve_start) */
#define FLUID_START_WARM 0        /* x0 = the previous solve's pressure (Eigen's solveWithGuess form of the same loop;
                                     the converged p is the same within cg_tol, cg_iters is NOT the reference's count);
                                     from the third pass of a step's do..while on: x0 = p_k + (1 - update_frac)(p_k - p_{k-1}),
                                     exact up to the float32 rounding of the right-hand side (same matrix, b_{k+1} = 0.9 b_k + c) */
#define FLUID_START_ZERO 1        /* x0 = 0 like the reference's cg.solve(b), fluid.cc:1474: cg_iters comparable        */

/* arithmetic of the multigrid V-cycle inside the fp64 PCG (fluid_params.mg_precision) */
#define FLUID_MG_FP32 0           /* float cycle (default: M^-1 only has to be a fixed SPD operator)     */
#define FLUID_MG_FP64 1           /* double cycle                                                        */

/* multi-GPU pressure block (fluid_params.dist_solve; fluid_create_dist only) */
#define FLUID_DIST_AUTO 0         /* decomposed solve when the active box is large, replicated otherwise */
#define FLUID_DIST_DECOMPOSED 1   /* domain-decomposed PCG with the globally coupled V-cycle             */
#define FLUID_DIST_REPLICATED 2   /* P2G fields all-reduced, the pressure block solved on every rank     */

typedef struct fluid_params {
    int32_t n;                /* cells per axis                         fluid.cc:1159 (121)      */
    int32_t device;           /* HIP device ordinal                                             */
    double dx;                /* cell size                              fluid.cc:1358 (1.0)      */
    double rho;               /* density                                fluid.cc:1471,1475 (1)   */
    double gravity[3];        /*                                        fluid.cc:1357 (0,-10,0)  */
    double max_dt;            /* maxTimeStep of FLIPadvect              fluid.cc:1490 (0.1)      */
    double outer_tol;         /* do..while(error > 0.1)                 fluid.cc:1484            */
    double update_frac;       /* velUpdate(dt/10)                       fluid.cc:1475 (0.1)      */
    double cg_tol;            /* Eigen default epsilon                  IterativeSolverBase.h:283 */
    int32_t cg_max_iters;     /* 0 = 2*numActive                        IterativeSolverBase.h:362 */
    int32_t max_outer_passes; /* 0 = unlimited (reference)                                       */
    int32_t precision;        /* FLUID_PRECISION_*                                               */
    int32_t preconditioner;   /* FLUID_PRECOND_*                                                  */
    double flip_blend;        /* 1 = pure FLIP (the reference, fluid.cc:981); b < 1 blends in the PIC gather of
                                 the reference's unused clampedCatmullRom (fluid.cc:125-207):
                                 v' = b (v + delta) + (1-b) v_pic.  Build extension.      */
    int32_t solve_start;      /* FLUID_START_*                                                   */
    int32_t mg_precision;     /* FLUID_MG_*                                                      */
    int32_t dist_solve;       /* FLUID_DIST_*                                                    */
    int32_t pad_;             /* must be 0                                                       */
} fluid_params_t;

typedef struct fluid_step_stats {
    double dt_in;             /* dt used by this step's pressure block  fluid.cc:1469-1475       */
    double dt_out;            /* dt written by FLIPadvect               fluid.cc:992-999         */
    double error;             /* last ||b-b2||/||b||                    fluid.cc:1483            */
    double max_speed;         /*                                        fluid.cc:976-991         */
    double relres;            /* last solve: sqrt(|r|^2/|b|^2)          ConjugateGradient.h:87   */
    int64_t num_active;       /* numActive                              fluid.cc:1395-1433       */
    int32_t outer_passes;     /* passes of the do..while                fluid.cc:1457-1484       */
    int32_t cg_iters;         /* PCG iterations summed over the passes                           */
    int32_t cg_iters_last;    /* iterations of the last solve                                    */
    int32_t box_lo[3];        /* active box of this step (index space, inclusive)                */
    int32_t box_hi[3];
    int32_t paths;            /* kernel forms this step took: FLUID_PATH_* bits                  */
} fluid_step_stats_t;
#define FLUID_PATH_P2G_TILES 1   /* particle -> grid in its 2 x 2-column tile form (a box whose row partials would not fit) */
#define FLUID_PATH_P2G_CROWD 16  /* particle -> grid with the cells of >= 18 particles summed on the matrix cores first (piled particles, mostly empty box) */
#define FLUID_PATH_TILE_LISTS 2  /* level-0 solver kernels over the lists of tiles that hold an unknown (mostly-air box) */
#define FLUID_PATH_DIST_DECOMPOSED 4  /* multi-GPU: window arrays, domain-decomposed PCG with the globally coupled V-cycle  */
#define FLUID_PATH_DIST_REPLICATED 8  /* multi-GPU: particles sharded, pressure block replicated on every rank              */
#define FLUID_PATH_DIST_REBALANCED 32 /* multi-GPU: the cut planes were moved at the end of this step (the window changed)      */
#define FLUID_PATH_MG_GALERKIN 128   /* the V-cycle's coarse levels were Galerkin operators by aggregation (mostly-air box whose re-discretised levels lose much of the pool) */
#define FLUID_PATH_DROPLETS 64        /* closed pockets of <= 64 unknowns (airborne droplets) were solved apart from the global system   */
#define FLUID_PATH_DROPLETS_SHORT 256 /* ... and at least one of them left its own CG by the iteration cap or a breakdown, not by the stopping rule (relres above cg_tol there) */

/* ---- lifetime ------------------------------------------------------------------------- */
/* Reference defaults (N=121, g=(0,-10,0), dx=1, rho=1, max_dt=0.1, outer_tol=0.1,
 * update_frac=0.1, cg_tol=2.2e-16, fp64).  fluid.cc:1357-1367. */
int fluid_default_params(fluid_params_t* p);
/* Allocates all device fields, sets the default solid shell (solid outside W,
 * fluid.cc:1256-1266).  Replaces the grid/PointList set-up of fluid.cc:1157-1347. */
int fluid_create(const fluid_params_t* p, fluid_sim_t** out);
int fluid_destroy(fluid_sim_t* s);
/* Text of the last error on the calling thread ("" if none). */
const char* fluid_last_error(void);
/* "libfluid_hip <version> gfx950" */
const char* fluid_version(void);

/* ---- scene ---------------------------------------------------------------------------- */
/* solid[N^3] uint8, 1 = solid (saccessor.setValue(xyz,1), fluid.cc:1266,1341).  Cells outside
 * W must be solid (the reference reads pressure(-1) otherwise) -> FLUID_ERR_ARG. */
int fluid_set_solid(fluid_sim_t* s, const uint8_t* solid);
/* PointList contents (fluid.cc:806-807,841): n particles, pos/vel = 3n doubles AoS.
 * vel may be NULL (zeros, like PointList::add). */
int fluid_upload_particles(fluid_sim_t* s, int64_t n, const double* pos, const double* vel);
/* Back to the caller in the ORIGINAL upload order. */
int fluid_download_particles(fluid_sim_t* s, double* pos, double* vel);
int64_t fluid_num_particles(fluid_sim_t* s);
/* dt carried between steps (double dt, fluid.cc:1367; written by FLIPadvect :1490). */
int fluid_set_dt(fluid_sim_t* s, double dt);
int fluid_get_dt(fluid_sim_t* s, double* dt);
/* Host-only synthetic input "water_cube_drop" (generalises fluid.cc:1176,1349 +
 * PointScatter.h:421-429): centred cube of side round(N*41/121) cells, ppc particles per cube
 * voxel at c - 0.5 + U[0,1)^3 (counter-based RNG).  pos==NULL -> returns the count only.
 * Needs no GPU. */
int64_t fluid_scene_water_cube_drop(int32_t n, int32_t ppc, uint64_t seed, double* pos);
/* Host-only: the reference's own initial particles — what
 *   fluidGrid->fill(CoordBBox(lo, hi), 0, true); std::mt19937 r(seed);
 *   UniformPointScatter<PointList, std::mt19937>(pos, points_per_volume, r)(*fluidGrid);      fluid.cc:1176,1347-1350
 * leaves in PointList::positions (openvdb/tools/PointScatter.h:143-185, tree fill + ValueOn order, libstdc++'s mt19937 /
 * uniform_int / uniform_real, g++'s right-to-left argument evaluation), with PointList::add's |p| < boundary - 2 filter
 * (fluid.cc:841; boundary <= 0: none).  The reference's scene: lo = -20, hi = 20, 10.f, seed 0, boundary 60 -> 689210
 * points for a 121^3 grid.  pos == NULL -> returns the count only; < 0 on bad arguments.  Needs no GPU. */
int64_t fluid_scene_uniform_scatter(const int32_t lo[3], const int32_t hi[3], float points_per_volume, uint32_t seed,
                                    int32_t boundary, double* pos);

/* ---- the step ------------------------------------------------------------------------- */
/* One iteration of the loop body fluid.cc:1378-1490 (everything except the .vdb write). */
int fluid_step(fluid_sim_t* s, fluid_step_stats_t* stats);

/* Per-phase entry points (parity tests drive these one by one). */
int fluid_p2g(fluid_sim_t* s);                 /* fluid.cc:1378,1384 (P2Gtransfer 1106-1148) + 1388-1413 (interpolate 843-882) */
int fluid_flags_index(fluid_sim_t* s);         /* fluid.cc:1416-1455 index sweep, output copy, velBeforeUpdate */
int fluid_rhs_div(fluid_sim_t* s, int which);  /* setRHS+setDiver: which=0 -> b (1469-1470), 1 -> b2 (1477-1480) */
int fluid_solve(fluid_sim_t* s);               /* setA,setA2,compute,solve fluid.cc:1471-1474 (matrix-free PCG) */
int fluid_vel_update(fluid_sim_t* s);          /* velUpdate(dt/10) fluid.cc:1475 */
int fluid_pressure_pass(fluid_sim_t* s, double* error); /* one do..while body fluid.cc:1457-1483 */
int fluid_flip_advect(fluid_sim_t* s);         /* FLIPadvect fluid.cc:1490 (972-1038) */
/* Stats of the phases run since the last fluid_step / fluid_p2g. */
int fluid_get_stats(fluid_sim_t* s, fluid_step_stats_t* stats);

/* ---- fields --------------------------------------------------------------------------- */
/* bytes must equal the field's size (see FLUID_FIELD_*). */
int fluid_download_field(fluid_sim_t* s, int field, void* dst, size_t bytes);
/* Upload CONTAINER (then call fluid_flags_index), VEL, VEL_BEFORE, DIVER, PRESSURE, SEARCH. */
int fluid_upload_field(fluid_sim_t* s, int field, const void* src, size_t bytes);

/* ---- the reference's unused grid / particle utilities --------------------------------------------
 * `extrapolate` (fluid.cc:705-802) and `PointList::resample` (fluid.cc:1053-1080) are dead code in the reference — the one
 * call of extrapolate, at the end of P2Gtransfer (fluid.cc:1147), is commented out and resample is never called — so fluid_step
 * does not run them; they are here for a caller that wants the reference's code path with them switched on.  Single GPU.
 * fluid_extrapolate: call after fluid_p2g; fills FLUID_FIELD_VEL (and VEL_BEFORE) of every cell inside W that P2G left without
 * a velocity with the average of its defined 26-neighbours, layer by layer; n_layers (may be NULL) = passes RUN — the "anything new"
 * counter is read back every 8 passes, so this is the number of layers that filled a cell rounded up to a multiple of 8.
 * fluid_resample: at most per_cell particles per base cell, in upload-index order; the others are parked at
 * (hi + 40, hi + 40, hi + 40) (the reference's (100, 100, 100)); only cells with x < hi - 10 (its `rx < 50`). */
int fluid_extrapolate(fluid_sim_t* s, int32_t* n_layers);
int fluid_resample(fluid_sim_t* s, int32_t per_cell, int64_t* n_parked);

/* The closed pockets of the last step's pressure system that were solved apart from the global solve (FLUID_PATH_DROPLETS;
 * kernels_droplets.hip): n_components of them; cells (may be NULL) receives 64 entries per component — the window-array cell
 * indices (ix * ny + iy) * nz + iz of its unknowns, ascending, padded with -1 — for at most cap_components components. */
int fluid_get_droplets(fluid_sim_t* s, int32_t* n_components, int64_t* cells, int32_t cap_components);

/* ---- stencil operator alone (micro-benchmark + parity of the 7-point apply) ------------ */
/* q = A s with the matrix of fluid.cc:304-412 for the current flags and dt; `reps` launches
 * timed with HIP events on the handle's stream; avg_ms = mean duration of one launch.
 * box: 0 = dense sweep over all N^3 cells, 1 = active box only. */
int fluid_stencil_apply(fluid_sim_t* s, int reps, int box, float* avg_ms);
/* The same, HBM-proof: the launches rotate over `nsets` separate copies of (s, q, flags), as many as `footprint_bytes` needs
 * (one set = (2 T + 1) N^3 bytes; at least 2), so that with a footprint well above the 256 MiB Infinity Cache no launch
 * finds its operands cached — fluid_stencil_apply's back-to-back launches over ONE set of 151 MB (fp32, 256^3) do.
 * nsets_out: the number of sets used.  q of the last launch is left in the handle's FLUID_FIELD_Q. */
int fluid_stencil_apply_hbm(fluid_sim_t* s, int reps, int box, int64_t footprint_bytes, int32_t* nsets_out, float* avg_ms);

/* ---- known-answer hooks ------------------------------------------------------------------- */
/* w[i] = spline(x[i]) (fluid.cc:22-37) evaluated by the DEVICE function the P2G / G2P kernels use (which = 0), or the
 * three-cell form those kernels call per axis, spline_at(p, round(p) - 1 + d, d) with d = which - 1 (which = 1..3: x[i]
 * is then the particle coordinate p).  Host buffers; tests compare bit for bit with the reference's own function. */
int fluid_spline_eval(int32_t device, int32_t which, int64_t n, const double* x, double* w);
/* out[0] = sum_i a[i] * b[i] over n doubles with the block-partial + fixed-order re-summation the PCG kernels use for
 * their dot products (restates the long dot-product test of openvdb/unittest/TestConjGradient.cc:212-237). */
int fluid_dot_eval(int32_t device, int64_t n, const double* a, const double* b, double* out);

/* ---- profiling ------------------------------------------------------------------------- */
/* Kernel classes timed with hipEvent pairs on the handle's stream. */
#define FLUID_PROF_PCG_SQ 0      /* fused p-update + 7-point apply + dot     */
#define FLUID_PROF_PCG_XR 1      /* fused x,r update + dots                  */
#define FLUID_PROF_P2G 2
#define FLUID_PROF_G2P 3
#define FLUID_PROF_SORT 4
#define FLUID_PROF_SOLVE 5       /* whole solve                              */
#define FLUID_PROF_MG_UP0 6  /* level-0 up leg of the V-cycle (k_mg_up)      */
#define FLUID_PROF_COUNT 7
/* Every `sample_every`-th launch of the per-iteration classes (PCG_SQ, PCG_XR, MG_UP0) and every max(1, sample_every / 8)-th
 * launch of the per-step classes (P2G, G2P, SORT, SOLVE) is bracketed by an event pair (0 = off).  An event record stalls the
 * stream by ~5-10 us: sample sparsely inside a timed region. */
int fluid_profile_enable(fluid_sim_t* s, int sample_every);
/* Resolves pending events; n_launches = launches seen, n_sampled = launches timed,
 * total_ms = sum over the timed ones, cells = sum of cells swept by the timed ones. */
int fluid_profile_read(fluid_sim_t* s, int klass, int64_t* n_launches, int64_t* n_sampled, double* total_ms, double* cells);
int fluid_profile_reset(fluid_sim_t* s);

/* ---- multi-GPU: 3-D block decomposition (one process per GPU) --------------------------------
 * The reference is single-process (no communication backend); this is new design
 * (SURVEY.md 8e).  The grid is cut into dims[0] x dims[1] x dims[2] blocks by per-axis cut planes;
 * rank r owns block (bx, by, bz) = (r / (dims[1] dims[2]), (r / dims[2]) % dims[1], r % dims[2]):
 * its cells, and the particles whose base cell (round(pos), fluid.cc:267) lies in it.
 *
 *   decomposed solve   a rank's field arrays cover its block + a 4-cell halo ring only (the
 *                      "window").  Per step: particle migration and ghost particles with the
 *                      <= 26 adjacent blocks, halo exchanges of flags / velocity / pressure /
 *                      FLIP delta, the unknown numbering from all-reduced row counts, and a
 *                      domain-decomposed PCG whose multigrid V-cycle is globally coupled: halo
 *                      exchanges of the residual and of the coarse correction inside the two
 *                      finest levels, the coarser levels gathered (one all-reduce) and solved
 *                      redundantly on every rank — the same cycle as on one GPU, so the
 *                      iteration count does not grow with the number of blocks.
 *   replicated solve   particles sharded the same way, but every rank keeps full-size arrays;
 *                      the P2G result of the active box is assembled on every rank by one SUM
 *                      all-reduce and the pressure block runs identically everywhere (bit-identical
 *                      to the one-GPU step; fallback for boxes too small to be worth exchanging).
 *
 * Transport is supplied by the caller (RCCL inside the library: fluid_rccl_comm_create; gloo in
 * the tests: the Python callbacks; an in-process transport for several blocks per process:
 * fluid_local_comm_create).  All pointers are DEVICE pointers; calls must be ordered after prior
 * work on `stream` and complete (or be stream-ordered) before later work on it.  Return 0 on
 * success.  Every rank makes the same sequence of allreduce calls; exchange calls are matched
 * pairwise (rank a lists peer b with the byte counts b lists for a, in both directions). */
#define FLUID_DT_F64 0
#define FLUID_DT_I32 1
#define FLUID_DT_I64 2
#define FLUID_DT_F32 3
#define FLUID_DT_U8 4
#define FLUID_OP_SUM 0
#define FLUID_OP_MAX 1
#define FLUID_OP_MIN 2
#define FLUID_MAX_RANKS 64
typedef struct fluid_comm {
    int32_t rank, size;
    void* ctx;
    /* Neighbour exchange: for i < n send sbytes[i] bytes from sbuf[i] to rank peer[i] and receive
     * rbytes[i] bytes from the same rank into rbuf[i] (either count may be 0).  Peers are distinct. */
    int (*exchange)(void* ctx, int32_t n, const int32_t* peer, const void* const* sbuf, const size_t* sbytes,
                    void* const* rbuf, const size_t* rbytes, void* stream);
    /* In-place all-reduce of `count` elements of dtype FLUID_DT_* with FLUID_OP_*. */
    int (*allreduce)(void* ctx, void* buf, int64_t count, int32_t dtype, int32_t op, void* stream);
} fluid_comm_t;

/* Native transport: fluid_comm_t over RCCL (grouped ncclSend/ncclRecv, ncclAllReduce on the solver's stream).
 * librccl_path: the librccl.so to dlopen ("" = by name); id128: ncclUniqueId made by rank 0 with
 * fluid_rccl_unique_id and handed to the other ranks by the launcher.  Binds to the current HIP device. */
int fluid_rccl_unique_id(const char* librccl_path, void* id128);
int fluid_rccl_comm_create(const char* librccl_path, const void* id128, int32_t rank, int32_t size, fluid_comm_t* out);
int fluid_rccl_comm_destroy(fluid_comm_t* comm);
const char* fluid_rccl_last_error(void);

/* In-process transport: `size` handles driven by `size` host threads of ONE process (several blocks per GPU: the
 * tests run 2 x 2 x 2 blocks on the one GPU of their box this way; a host that drives several GPUs from one
 * process can use it too).  Device-to-device copies between the handles' buffers, host-side rendezvous. */
int fluid_local_group_create(int32_t size, void** group);
int fluid_local_group_destroy(void* group);
/* Wakes every rank waiting inside the transport with an error: the driver thread of a rank that failed elsewhere calls it. */
int fluid_local_group_abort(void* group);
int fluid_local_comm_create(void* group, int32_t rank, fluid_comm_t* out);

typedef struct fluid_decomp {
    int32_t dims[3];          /* blocks per axis; dims[0]*dims[1]*dims[2] == comm->size                        */
    const int32_t* cuts[3];   /* cuts[a][0..dims[a]]: cuts[a][0] = 0, cuts[a][dims[a]] = n, ascending; interior
                                 cuts are multiples of 4 and every block is >= 8 cells wide (the coupled V-cycle
                                 coarsens 2 x 2 x 2 twice across block faces)                                  */
} fluid_decomp_t;
/* Like fluid_create, for rank comm->rank of comm->size. */
int fluid_create_dist(const fluid_params_t* p, const fluid_comm_t* comm, const fluid_decomp_t* decomp, fluid_sim_t** out);
/* Geometry of this handle's arrays in a decomposed run (one GPU: the whole grid): global index of the window's first
 * cell, window dims (the shape of fluid_download_field's arrays), owned block [own_lo, own_hi) in global indices. */
int fluid_window(fluid_sim_t* s, int32_t origin[3], int32_t dims[3], int32_t own_lo[3], int32_t own_hi[3]);
/* Upload THIS rank's particles (base cell inside its block) with their global ids (unique
 * across ranks; the order of fluid_download_particles_ids is the device order). */
int fluid_upload_particles_ids(fluid_sim_t* s, int64_t n, const double* pos, const double* vel, const uint32_t* ids);
/* This rank's current particles and ids; call with NULLs to get the count. */
int64_t fluid_download_particles_ids(fluid_sim_t* s, double* pos, double* vel, uint32_t* ids);
/* Cut planes for dims[0] x dims[1] x dims[2] blocks of about equal particle count per axis slab, from a host
 * particle set (host-only); cuts[a] receives dims[a]+1 values that satisfy fluid_decomp's rules. */
int fluid_partition_blocks(int32_t n, int64_t np, const double* pos, const int32_t dims[3], int32_t* cuts_x, int32_t* cuts_y,
                           int32_t* cuts_z);
/* Re-balancing of the cut planes of a decomposed run: every `every` steps (0 = never, the default) the blocks' particle counts
 * are compared and, when the fullest holds more than `ratio` x the mean (>= 1), the planes are placed anew by particle count and
 * the particles handed to their new owners; collective — every rank of the run sets the same values.  The window of this handle
 * then changes (fluid_window), fields downloaded afterwards have the new shape, and the step that did it reports
 * FLUID_PATH_DIST_REBALANCED.  fluid_dist_get_cuts: the current planes (dims[a] + 1 values per axis) and how often they moved. */
int fluid_dist_set_rebalance(fluid_sim_t* s, int32_t every, double ratio);
int fluid_dist_get_cuts(fluid_sim_t* s, int32_t* cuts_x, int32_t* cuts_y, int32_t* cuts_z, int32_t* n_rebalanced);
/* What a decomposed handle decided about itself (any pointer may be NULL): overlap = 0 not checked yet (no solve with peers so far),
 * 1 the residual's overlapped halo exchange (second stream) delivered the serial exchange's bytes on every rank and is in use,
 * 2 it did not on some rank and is switched off everywhere; cg_form = 0 two scalar all-reduces per PCG iteration, 1 Chronopoulos-Gear
 * (one); n_refused = re-balances every rank gave up together because some rank could not build its second window. */
int fluid_dist_get_info(fluid_sim_t* s, int32_t* overlap, int32_t* cg_form, int32_t* n_refused);

/* ---- OpenVDB file output (replaces file2.write(grids2) / file.write(grids), fluid.cc:1503-1504,1508) ----
 * Dense float32 N^3 arrays (z fastest, cell (0,0,0) = index coordinate (lo,lo,lo), lo = -(N/2)) written as unnamed
 * FloatGrids (Tree_float_5_4_3, background 0, voxel size 1, every cell of [lo,hi]^3 active) in OpenVDB's file format 224
 * — the grids fluid.cc:1161-1164,1434-1451 builds.  Host only: no GPU, no OpenVDB library (zlib for the ZIP flag).
 * The reference writes ONE grid into every simulation/mygrids<i>.vdb (`grids2` lives inside the loop, fluid.cc:1373) and
 * EVERY step's grid into the final mygrids.vdb (`grids`, :1366,1450,1508): the streaming form appends one grid per step. */
#define FLUID_VDB_ACTIVE_MASK 2       /* io/Compression.h:80 COMPRESS_ACTIVE_MASK                                  */
#define FLUID_VDB_ZIP_ACTIVE_MASK 3   /* COMPRESS_ZIP | COMPRESS_ACTIVE_MASK: the library's default (Compression.h:78-81) */
typedef struct fluid_vdb_writer fluid_vdb_writer_t;
int fluid_vdb_open(const char* path, int32_t n, int32_t n_grids, int32_t compression, fluid_vdb_writer_t** out);
int fluid_vdb_append(fluid_vdb_writer_t* w, const float* grid);     /* exactly n_grids times                       */
int fluid_vdb_close(fluid_vdb_writer_t* w);                         /* FLUID_ERR_ARG if fewer grids were appended  */
/* n_grids arrays in one call; fluid_write_vdb = ZIP | ACTIVE_MASK. */
int fluid_write_vdb(const char* path, int32_t n, int32_t n_grids, const float* const* grids);
int fluid_write_vdb_ex(const char* path, int32_t n, int32_t n_grids, const float* const* grids, int32_t compression);

#ifdef __cplusplus
}
#endif
#endif /* FLUID_HIP_H */
