// Internal state of a handle and the host functions shared by fluid_api.hip (one GPU) and fluid_dist.hip (decomposed runs).
#pragma once
#include "common.h"
#include "../../include/fluid_hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

int fluid_fail(int code, const std::string& msg);   // sets fluid_last_error() of the calling thread, returns code
#define HIPCHK(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess)                                                                                     \
            return fluid_fail(FLUID_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " @" + std::to_string(__LINE__)); \
    } while (0)

using namespace fl;

struct ProfClass {
    long launches = 0, sampled = 0;
    double ms = 0, cells = 0;
    std::vector<hipEvent_t> e0, e1;
    std::vector<double> pc;
};

struct fluid_sim {
    fluid_params_t prm;
    Grid g;
    size_t ncell = 0;
    hipStream_t st = nullptr;
    // grid fields
    uint8_t *solid = nullptr, *flags = nullptr;
    float *container = nullptr, *rhs = nullptr, *diver = nullptr, *diver2 = nullptr;
    double *u = nullptr, *v = nullptr, *w = nullptr, *ub = nullptr, *vb = nullptr, *wb = nullptr;
    double *dcx = nullptr, *dcy = nullptr, *dcz = nullptr, *pressure = nullptr;
    double* p_guess = nullptr;    // last solved pressure, never cleared: the multigrid PCG starts from it (solve_start = FLUID_START_ZERO: from 0 like the reference)
    bool warm = true, have_guess = false;
    // Passes of one step's do..while solve the SAME matrix for right-hand sides b_{k+1} = (1 - f) b_k + c (f = update_frac: the partial
    // velocity update takes f of the pressure gradient out of the divergence, c = what gravity puts back, the same in every pass), so
    // p_{k+1} = p_k + (1 - f)(p_k - p_{k-1}) up to the float32 rounding of b: from the third pass on the solve starts from that.
    double* p_guess2 = nullptr;   // the solution before p_guess
    long guess_step = -1, guess2_step = -1;   // step counter and pass index at which each was stored
    int guess_pass = -1, guess2_pass = -1;
    long step_counter = 0;
    bool extrapolate = true;      // FLUID_EXTRAPOLATE=0 switches it off
    // the start of this solve: guess arrays and coefficients (g2 == nullptr: plain warm start); called by both solve paths
    // ... and q = p_{k+1} - (1 - f) p_k = A^-1 c changes little from one step to the next (c is gravity's share; the walls and most of the
    // water stay where they are): kept from the last step that ran two passes (p_q, make_q), it starts the SECOND pass of a step from
    // (1 - f) p_0 + q — 23 -> 21 iterations with q = 0, -> 18-19 with the last q in the splash of the 256^3 drop.  (The FIRST pass of the
    // next step from (1 - f) p_last + q: measured, no fewer iterations.)
    double* p_q = nullptr;
    long q_step = -1;
    void start_guess(const double*& g1, const double*& g2, double& ca, double& cb) const
    {
        g1 = p_guess, g2 = nullptr, ca = 1.0, cb = 0.0;
        const int k = stats.outer_passes;   // passes completed in this step = index of this one
        const bool same_step = p_guess2 && k >= 2 && guess_step == step_counter && guess2_step == step_counter && guess_pass == k - 1 && guess2_pass == k - 2;
        const double f = 1.0 - prm.update_frac;
        if (extrapolate && same_step) {
            g2 = p_guess2, ca = 1.0 + f, cb = -f;
        } else if (extrapolate && p_q && q_step >= 0 && step_counter - q_step <= 4 && k == 1 && guess_step == step_counter && guess_pass == 0) {
            g2 = p_q, ca = f, cb = 1.0;
        }
    }
    // right after the second pass of a step (and, in a decomposed run, after its halo exchange): is q to be formed from p_guess, p_guess2?
    bool make_q() const { return warm && p_q && stats.outer_passes == 1 && guess_step == step_counter && guess_pass == 1 && guess2_step == step_counter && guess2_pass == 0; }
    // after store_pressure wrote the new solution into p_guess2's buffer: it becomes the latest
    void rotate_guess()
    {
        std::swap(p_guess, p_guess2);
        guess2_step = guess_step, guess2_pass = guess_pass;
        guess_step = step_counter, guess_pass = stats.outer_passes;
    }
    int *indices = nullptr, *scan_sums = nullptr, *ipart = nullptr;
    // pcg
    void *R = nullptr, *S[2] = {nullptr, nullptr}, *Q = nullptr, *X = nullptr, *Zmg = nullptr;
    double *pcx = nullptr, *pcy = nullptr, *pcz = nullptr;  // getVelocity(c, vels) per cell: allocated on first use, PIC blend only  // box-local layout (LBox)
    uint8_t* cntL = nullptr;
    LBox L{};
    size_t lmax = 0;
    double *part_bb = nullptr, *part_rr = nullptr, *part_rz[2] = {nullptr, nullptr}, *part_pq = nullptr, *part_err = nullptr;
    PcgState* ps = nullptr;
    PcgState* h_ps = nullptr;  // pinned, 2 slots
    hipEvent_t ev_poll[2] = {nullptr, nullptr};
    // particles
    long np = 0, cap = 0;
    Particles pa{}, pb{};
    int *key = nullptr, *slot = nullptr, *order = nullptr, *order2 = nullptr, *cell_count = nullptr, *cell_start = nullptr;
    uint32_t* spid = nullptr;   // original ids in sorted-position order (per-cell rank pass)
    double *stage_pos = nullptr, *stage_vel = nullptr;
    double* pw = nullptr;  // 9 axis weights per particle, SoA with stride cap
    double* p2g_part = nullptr;  // k_p2g_rows' three x-plane partials: 12 doubles per cell of the P2G box (grown on demand)
    size_t p2g_part_cap = 0;
    int2* p2g_crowd = nullptr;   // (cell, piece) list of k_p2g_crowd_sum, grown on demand; its count is p2g_items[1]
    size_t p2g_crowd_cap = 0;
    int* p2g_items = nullptr;    // k_p2g_rows' work list (count + int4 items), grown on demand; the count is 0 between launches
    size_t p2g_items_cap = 0;
    int max_cell = 0;            // most particles in one cell after the last sort (all ranks' cells when distributed)
    // mostly-air box (splash, settled pool): level-0 legs, SQ and XR run over compacted lists of the tiles that hold an unknown
    uint8_t* tl_flags = nullptr;  // per-tile flags (both tile shapes, one after the other)
    int *tl_mg = nullptr, *tl_sq = nullptr;
    size_t tl_cap = 0;
    int n_tl_mg = 0, n_tl_sq = 0;
    int *row_flags = nullptr, *row_pos = nullptr, *row_list = nullptr;   // XR's list of non-empty z rows (32 cells), built with the tile lists
    size_t row_cap = 0;
    int n_rows = 0;
    bool rows_on = true;          // XR over the z rows that hold an unknown (finer than the SQ tile list)
    // closed pockets (airborne droplets) of the pressure system, solved on their own (kernels_droplets.hip; FLUID_DROPLETS=0: off)
    bool drops_on = true;
    bool row_sweeps = true;       // FLUID_ROW_SWEEPS=0: the per-step box sweeps keep their cell-per-thread forms (kernels_grid.hip, k_*4)
    int* drop_ctr = nullptr;      // 64 x DROP_NCTR ints: the slot counters, then DROP_NCTR + 1 range starts of the dense numbering
    int* drop_n = nullptr;        // cells per component
    int* drop_cells = nullptr;    // 64 local-box cell indices per component
    int n_drop = 0;               // components taken out of this step's global solve
    int drop_last = -1;           // ... found by the last search (-1: none yet); below drop_min the search runs every 8th step only
    int drop_min = 4000;          // FLUID_DROPLETS_MIN (measured at 256^3: the search pays from ~4 000 droplets on)
    int drop_skipped = 0;
    bool lists_hint = false;      // the previous step's box was mostly air: build the lists before this step's flags sync
    bool lists_on = false;        // this step's solves use them
    int lists_force = -1;         // FLUID_TILE_LISTS=0|1
    int p2g_force = 0;           // FLUID_P2G_FORM=rows|tiles: 1 / 2, for experiments
    long p2g_ref_cells = 0;      // decomposed run: cells of the GLOBAL active box (0: use Rb)
    long last_num_active = 0;    // unknowns of the previous step (the same on every rank)
    StepState* ss = nullptr;
    StepState* h_ss = nullptr;  // pinned
    // boxes
    Box Pb{0, 0, 0, -1, -1, -1}, Rb{0, 0, 0, -1, -1, -1}, Sb{0, 0, 0, -1, -1, -1};
    Box dirty{0, 0, 0, -1, -1, -1};   // box holding non-zero step-field data (zeroed before the next P2G)
    int flag_x0 = 0, flag_x1 = -1;    // x planes whose flags / indices the last flags pass may have made non-trivial
    bool flags_valid = false;         // flags / indices outside [flag_x0, flag_x1] are known to be "solid or empty" / -1
    bool sorted = false, have_p2g = false, have_flags = false;
    long n_out = 0;           // particles whose base cell is off the grid (last bucket of the sorted array), from the last sort
    bool sort_hint = false;   // Pb is the bounding box of THESE particles one step ago (false after an upload)
    double dt = 0.1;
    fluid_step_stats_t stats{};
    // multigrid preconditioner (single-GPU fp64 solve)
    static constexpr int MG_MAXL = 8;
    int mg_nl = 0, mg_tail = 0;   // levels; first level handled by the single-block tail kernel
    // iteration count of the previous multigrid solve of the same kind (sizes the first unpolled batch): first pass of a step, second pass,
    // later passes (those start from the extrapolation and need about half the iterations)
    long mg_last_iters_k[3] = {0, 0, 0};
    int pass_class() const { return stats.outer_passes < 2 ? stats.outer_passes : 2; }
    int mg_csweeps = 3;           // red-black sweeps (each direction) on the coarsest level (12 -> 2 changes the PCG count by 1 in 520)
    MLevel mgl[MG_MAXL];
    uint8_t *mg_typ[MG_MAXL] = {}, *mg_cnt[MG_MAXL] = {};
    // Galerkin coarse levels by aggregation (kernels_gal.hip): mostly-air boxes, float cycle, one GPU.  FLUID_MG_GALERKIN=0|1
    int gal_mode = 1;             // FLUID_MG_GALERKIN: 0 never, 1 whichever cycle needed fewer iterations when last measured (mostly-air boxes), 2 always there
    int gal_it[2] = {-1, -1};     // first-pass iterations of the last step that used the re-discretised / the Galerkin cycle (-1: not measured)
    int gal_since_probe = 0;      // steps since the worse cycle was last looked at
    bool gal_eligible = false;    // this step could have taken the Galerkin cycle
    bool gal = false;             // ... and in use this step (set by mg_setup)
    int gal_lc = 0;               // the coarsest level of the Galerkin cycle (one block)
    double gal_dt = 0;            // dt the coefficients were built with
    double gal_wc = 1.8;          // over-correction of the piecewise-constant prolongation 
    int gal_sweeps = 3;           // red-black sweeps, each direction, on the coarsest level (2 ... 16: the same iteration counts)
    char* gal_slab = nullptr;     // per level 1..gal_lc: gd, gx, gy, gz (float) and the unknown flags
    size_t gal_slab_cap = 0;
    float* gal_c[MG_MAXL][4] = {};
    uint8_t* gal_cnt[MG_MAXL] = {};
    uint8_t* gal_tfl[MG_MAXL] = {};   // per leg tile: holds an unknown
    char *mg_u[MG_MAXL] = {}, *mg_v[MG_MAXL] = {}, *mg_f[MG_MAXL] = {}, *mg_r[MG_MAXL] = {};  // per level: u, ping-pong, rhs, residual (float or double)
    double mg_wc[4] = {1.25, 1.1, 1.0, 1.0};   // weight of the coarse correction at level 0 / level 1 / deeper kernel levels / inside the tail 
    bool mg_fp32 = true;           // the V-cycle computes and stores in float inside the double PCG (mg_precision = FLUID_MG_FP64: double)
    double* mg_part = nullptr;    // per-block partials of r.z when a level-0 launch has more blocks than the PCG kernels re-sum
    char* mg_slab = nullptr;      // one allocation behind every mg_* array and Zmg (re-carved each step)
    size_t mg_slab_cap = 0;
    // multi-GPU (3-D block decomposition, fluid_dist.hip)
    bool dist = false;
    struct DistState* ds = nullptr;
    long p_off = 0;              // my live particles are pa[p_off .. p_off+np)
    int* d_small = nullptr;      // device scratch ints
    int* h_small = nullptr;      // pinned mirror
    // profiling
    int prof_every = 0;
    ProfClass prof[FLUID_PROF_COUNT];
};

namespace fl {
inline bool box_empty(const Box& b) { return b.x1 < b.x0 || b.y1 < b.y0 || b.z1 < b.z0; }
Box clip_dilate(const Box& b, int d, const Grid& g);      // dilate by d cells, clipped to the window
template <typename T>
inline hipError_t dalloc(T** p, size_t n)
{
    hipError_t e = hipMalloc((void**)p, n * sizeof(T));
    if (e == hipSuccess) e = hipMemset(*p, 0, n * sizeof(T));
    return e;
}
inline size_t solver_elem(const fluid_sim* s) { return s->prm.precision == FLUID_PRECISION_FP32 ? 4 : 8; }
inline bool use_mg(const fluid_sim* s) { return s->prm.preconditioner == FLUID_PRECOND_MG && s->prm.precision == FLUID_PRECISION_FP64; }

int prof_begin(fluid_sim* s, int k, double cells);
void prof_end(fluid_sim* s, int k, int tok);
int alloc_particles(fluid_sim* s, long n);
int grow_particles(fluid_sim* s, long need);
int read_ss(fluid_sim* s);
int sort_pass(fluid_sim* s, int ax0, int ax1, int* h_tail = nullptr);   // counting sort over the x planes [ax0, ax1] of the window
int clear_dirty(fluid_sim* s);
int run_p2g(fluid_sim* s, const Box& box);
template <typename T>
Coef<T> make_coef(const fluid_sim* s);
MgCoef<double> mg_coef(const fluid_sim* s, int level);
hipError_t zero_search(fluid_sim* s, size_t lb);
int phase_flags(fluid_sim* s);
int phase_pressure_pass(fluid_sim* s, double* error);
int pic_fields(fluid_sim* s);
int fluid_create_window(const fluid_params_t* p, const Grid& g, fluid_sim_t** out);   // fluid_create on a window of the grid

// fluid_dist.hip
int dist_step(fluid_sim* s, fluid_step_stats_t* stats);
void dist_destroy(fluid_sim* s);
int dist_download_field(fluid_sim* s, int field, void* dst, size_t bytes, bool* handled);
void dist_keep_solid(fluid_sim* s, const uint8_t* solid_global);
}  // namespace fl
