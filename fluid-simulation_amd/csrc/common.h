// Internal declarations shared by the HIP translation units of libfluid_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fl {

// ---- cell flag byte ---------------------------------------------------------------------
// bit0 solid (static), bit1 fluid (per step: container>0 && !solid, fluid.cc:326,445,579,
// 1423-1425), bits 2..4 = number of non-solid 6-neighbours (the multiplicity of `scale` in
// Adiag, fluid.cc:326-407; valid on fluid cells).
constexpr uint8_t F_SOLID = 1;
constexpr uint8_t F_FLUID = 2;
constexpr int F_CNT_SHIFT = 2;

// The dense field arrays of a handle cover a WINDOW of the global grid: all of it on one GPU (nx = ny = nz = N, origin 0),
// the owned block + a halo ring on a rank of a decomposed run.  Kernels index with window-local (ix, iy, iz); a window
// edge is either an edge of the global grid (reads beyond it return the reference's background 0) or an interior cut
// whose halo is wide enough that no compute box reaches it.
struct Grid {
    int N;    // GLOBAL cells per axis
    int lo;   // global coordinate of global index 0
    int hi;   // global coordinate of global index N-1
    int nx, ny, nz;     // window dims = array dims
    int ox, oy, oz;     // global index of window index 0
    __host__ __device__ inline size_t idx(int ix, int iy, int iz) const
    {
        return ((size_t)ix * ny + (size_t)iy) * nz + (size_t)iz;
    }
    __host__ __device__ inline size_t cells() const { return (size_t)nx * ny * nz; }
    __host__ __device__ inline long sx() const { return (long)ny * nz; }   // x stride
    // coordinate of window index 0 per axis
    __host__ __device__ inline int cx0() const { return lo + ox; }
    __host__ __device__ inline int cy0() const { return lo + oy; }
    __host__ __device__ inline int cz0() const { return lo + oz; }
    // coordinate of the last window cell per axis
    __host__ __device__ inline int cx1() const { return lo + ox + nx - 1; }
    __host__ __device__ inline int cy1() const { return lo + oy + ny - 1; }
    __host__ __device__ inline int cz1() const { return lo + oz + nz - 1; }
};

// inclusive box in index space
struct Box {
    int x0, y0, z0, x1, y1, z1;
    __host__ __device__ inline int nx() const { return x1 - x0 + 1; }
    __host__ __device__ inline int ny() const { return y1 - y0 + 1; }
    __host__ __device__ inline int nz() const { return z1 - z0 + 1; }
    __host__ __device__ inline long long cells() const { return (long long)nx() * ny() * nz(); }
};

// Coefficients of the reference matrix (setA, fluid.cc:304-412): diag[k] is the float32
// value reached after k accumulations `Adiag = float(Adiag + scale)`; off = float(-scale).
template <typename T>
struct Coef {
    T diag[7];
    T inv[7];   // 1/diag, as Eigen's DiagonalPreconditioner stores it (BasicPreconditioners.h:73)
    T off;
};

// Box-local solver layout: local (i,j,k) <-> global index (x0+i-1, y0+j-1, z0+k-LBOX_K0);
// interior 1..nx, 1..ny, K0..K0+nz-1; everything else is zero padding (never an unknown).
constexpr int LBOX_K0 = 16;
struct LBox {
    int x0, y0, z0, nx, ny, nz, Lx, Ly, Lz;
    __host__ __device__ inline size_t cells() const { return (size_t)Lx * Ly * Lz; }
};

// One multigrid level: a dx x dy x dz domain inside a dense array (z stride 1, zero cells around it).
struct MLevel {
    int dx, dy, dz;
    long sx, sy;
    int ox, oy, oz;
    size_t cells;  // allocated elements
    __host__ __device__ inline size_t at(int i, int j, int k) const
    {
        return (size_t)((long)(i + ox) * sx + (long)(j + oy) * sy + (k + oz));
    }
};
template <typename T>
struct MgCoef {
    T diag[7];  // by number of non-solid neighbours
    T inv[7];
    T off;
};

// Scalars of one PCG solve, device resident.
struct PcgState {
    double bb;        // |b|^2
    double thr;       // tol^2 |b|^2
    double rr;        // last |r|^2 seen by the SQ kernel
    int done;         // 1 once |r|^2 < thr (or b == 0)
    int iters;        // Eigen's `i` (ConjugateGradient.h:70-88)
    int breakdown;    // p.Ap <= 0 or non-finite
    int pad;
};

// Per-step device scalars
struct StepState {
    int bbox_min[3];          // particle base-cell bounding box (index space)
    int bbox_max[3];
    int num_active;
    int n_out;                // particles whose base cell is outside the grid
    int max_cell;             // most particles in one cell (P2G picks its kernel by it)
    int n_drop;               // closed pockets of the pressure system found this step (kernels_droplets.hip); may exceed the buffer's capacity
    int n_tl_mg, n_tl_sq;     // active tiles of the level-0 V-cycle legs / of SQ and XR (mostly-air boxes)
    int n_rows, n_l1_old;     // z rows of 32 cells that hold an unknown (XR's list); unknowns of level 1 as the re-discretised cycle types it ...
    int n_tl_int, n_tl_bnd;   // decomposed run: active level-0 down-leg tiles that read no received cell / that do
    int n_l1_gal, n_drop_fail;   // ... and as aggregation does (any child); droplets whose own CG did not reach the tolerance this step (k_drop_solve)
    unsigned long long max_speed_bits;  // max |v_p| as non-negative double bits
    double dt;                // fluid.cc:1367 / 992-999
    double err_num;           // |b-b2|^2
    double err_den;           // |b|^2
};

constexpr int MAX_PARTIALS = 8192;

// fluid.cc:22-37
__device__ __forceinline__ double spline(double x)
{
    if (x < 0) x *= -1.0;
    if (x < 0.5) return 1.5 * (4.0 * x * x * x - 4.0 * x * x + 2.0 / 3.0);
    if (x < 1.0) return 1.5 * ((-8.0 * (x * x * x) / 6.0) + 4.0 * x * x - 4.0 * x + 4.0 / 3.0);
    return 0;
}

// spline(p - c) for the cell c = round(p) - 1 + d, d = 0..2 (the only cells a particle meets per axis).  For the outer two
// |p - c| lies in [0.5, 1.5] — round() is half away from zero, so |p - round(p)| <= 0.5, and the one subtraction p - c
// cannot round below 0.5 — and spline()'s first branch is never taken: the same values with half the arithmetic.
__device__ __forceinline__ double spline_at(double p, int c, int d)
{
    double x = p - (double)c;
    if (d == 1) return spline(x);
    if (x < 0) x *= -1.0;
    if (x < 1.0) return 1.5 * ((-8.0 * (x * x * x) / 6.0) + 4.0 * x * x - 4.0 * x + 4.0 / 3.0);
    return 0;
}

// ---- wave64 / block reductions ------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T w = __shfl_down(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}
// Sum over a block of NW waves; result valid in thread 0.  `sm` needs NW entries.
template <typename T, int NW>
__device__ __forceinline__ T block_sum(T v, T* sm)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    T r = 0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NW; ++i) r += sm[i];
    }
    return r;
}
// Sum of n doubles (n <= MAX_PARTIALS) by the whole block, in a fixed order; result
// broadcast to every thread.  `sm` needs NW+1 entries.
template <int NW>
__device__ __forceinline__ double block_sum_array(const double* __restrict__ a, int n, double* sm)
{
    double v = 0;
    for (int i = threadIdx.x; i < n; i += NW * 64) v += a[i];
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) r += sm[i];
        sm[NW] = r;
    }
    __syncthreads();
    return sm[NW];
}

// Blocks are dealt round-robin over the 8 XCDs; give each XCD (b % 8) a contiguous range of
// virtual block ids so that neighbouring tiles share one L2 (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int b, int nb)
{
    const int q = nb >> 3, r = nb & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// Sum three partial arrays in one pass; results broadcast to all threads.  sm: 12 doubles.
__device__ __forceinline__ void block_sum3(const double* __restrict__ a, int na, const double* __restrict__ b, int nb,
                                           const double* __restrict__ c, int nc, double* sm, double& ra, double& rb, double& rc)
{
    double va = 0, vb = 0, vc = 0;
    for (int i = threadIdx.x; i < na; i += 256) va += a[i];
    for (int i = threadIdx.x; i < nb; i += 256) vb += b[i];
    for (int i = threadIdx.x; i < nc; i += 256) vc += c[i];
    va = wave_sum(va); vb = wave_sum(vb); vc = wave_sum(vc);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sm[w] = va; sm[4 + w] = vb; sm[8 + w] = vc; }
    __syncthreads();
    ra = ((sm[0] + sm[1]) + sm[2]) + sm[3];
    rb = ((sm[4] + sm[5]) + sm[6]) + sm[7];
    rc = ((sm[8] + sm[9]) + sm[10]) + sm[11];
}

// ---- launchers (defined in kernels_*.hip) --------------------------------------------------
struct Particles {
    double *px, *py, *pz, *vx, *vy, *vz;
    uint32_t* pid;
    __host__ __device__ inline Particles shifted(long o) const
    {
        Particles q = *this;
        q.px += o; q.py += o; q.pz += o; q.vx += o; q.vy += o; q.vz += o; q.pid += o;
        return q;
    }
};
constexpr uint32_t PID_DEAD = 0xFFFFFFFFu;  // multi-GPU: particle handed to a neighbour rank

// particles
void launch_bin_count(hipStream_t st, Grid g, long n, Particles p, int* key, int* slot, int* cell_count, int* part, StepState* ss);
void launch_bin_scatter(hipStream_t st, long n, const int* key, const int* slot, const int* cell_start, const uint32_t* pid, int* order,
                         uint32_t* spid);
void launch_bin_rank(hipStream_t st, long n_pos, long pos0, const int* key, const int* cell_start, const int* order, const uint32_t* spid,
                     int* order2);
void launch_reorder(hipStream_t st, long n, const int* order, Particles src, Particles dst, double* w = nullptr, long wstride = 0);
void launch_weights(hipStream_t st, long n, Particles p, double* w, long stride);
void launch_axpby_box(hipStream_t st, Grid g, Box box, double a, const double* x, double b, const double* y, double* out);   // out = a x + b y on box
long p2g_max_items(Box box);
size_t p2g_part_doubles(Box box);   // size of launch_p2g's `part`
constexpr int P2G_PILED = 256;   // a cell with more particles: the particles have piled up (walls, floor), P2G takes the tile form
void launch_p2g(hipStream_t st, Grid g, Box box, Particles p, const double* pw, long wstride, const int* cell_start, const uint8_t* flags,
                double* part, int* items, float* container, double* u, double* v, double* w, double* ub, double* vb, double* wb, int2* crowd_list = nullptr,
                Particles park = Particles{});
void launch_p2g_tiles(hipStream_t st, Grid g, Box box, Particles p, const double* pw, long wstride, const int* cell_start, const uint8_t* flags,
                      float* container, double* u, double* v, double* w, double* ub, double* vb, double* wb);
void launch_g2p_tiled(hipStream_t st, Grid g, Box pb, Particles p, const int* cell_start, const double* dcx, const double* dcy, const double* dcz,
                      const double* pcx, const double* pcy, const double* pcz, double blend, StepState* ss);
void launch_g2p(hipStream_t st, Grid g, long n, Particles p, const double* dcx, const double* dcy, const double* dcz, const double* pcx,
                const double* pcy, const double* pcz, double blend, StepState* ss);
void launch_advect(hipStream_t st, Grid g, long n, Particles p, const uint8_t* flags, double max_dt, double dx, StepState* ss);
void launch_pack_particles(hipStream_t st, long n, Particles p, double* pos_aos, double* vel_aos);
void launch_unpack_records(hipStream_t st, long n, const double* rec, Particles p, long off);
void launch_pack_records(hipStream_t st, long n, Particles p, long off, double* rec);
void launch_unpack_ids(hipStream_t st, long n, const double* pos, const double* vel, const uint32_t* ids, Particles p);
void launch_pack_ids(hipStream_t st, long n, Particles p, long off, double* pos, double* vel, uint32_t* ids);
void launch_unpack_particles(hipStream_t st, long n, const double* pos_aos, const double* vel_aos, Particles p);

// grid
void launch_exclusive_scan(hipStream_t st, const int* in, int* out, long n, int* block_sums, int* total);
void launch_sort_tail(hipStream_t st, const int* cell_count, int* cell_start, long c1, long ncell);
void launch_index_scan(hipStream_t st, Grid g, const uint8_t* flags, int* indices, int* block_sums, int* total);
void launch_flags(hipStream_t st, Grid g, const uint8_t* solid, const float* container, uint8_t* flags, int x0, int x1);
void launch_index_scan_range(hipStream_t st, Grid g, const uint8_t* flags, int* indices, int* block_sums, int* total, int x0, int x1);
void launch_rhs_div(hipStream_t st, Grid g, Box box, const uint8_t* flags, const double* u, const double* v, const double* w,
                    float* rhs, float* diver, double dx, double gdt0, double gdt1, double gdt2, bool rows = false);
void launch_vel_update(hipStream_t st, Grid g, Box box, const uint8_t* flags, const double* p, double* u, double* v, double* w,
                       double k, double g0, double g1, double g2, bool rows = false);
void launch_flip_delta(hipStream_t st, Grid g, Box box, const double* u, const double* v, const double* w,
                       const double* ub, const double* vb, const double* wb, double* dcx, double* dcy, double* dcz, double* pcx, double* pcy, double* pcz);
void launch_err_norm(hipStream_t st, Grid g, Box box, const uint8_t* flags, const float* b, const float* b2, double* part, StepState* ss);
void launch_zero_step_state(hipStream_t st, StepState* ss, int N);
struct ZeroList { float* f4[4]; double* f8[7]; };
void launch_zero_fields(hipStream_t st, const ZeroList& z, Grid g, Box box, bool rows = false);
void launch_unpack_box(hipStream_t st, Grid g, Box box, const double* buf, float* container, double* u, double* v, double* w, double* ub, double* vb,
                       double* wb);

void launch_extrap_init(hipStream_t st, Grid g, const uint8_t* solid, const float* container, int* layer);
void launch_extrap_layer(hipStream_t st, Grid g, int pass, int* layer, double* u, double* v, double* w, int* n_new);
void launch_resample(hipStream_t st, Grid g, long n, Particles p, const int* cell_start, int per_cell, int xlim, double far_, int* n_parked);

void launch_spline_eval(hipStream_t st, int which, long n, const double* x, double* w);
void launch_dot(hipStream_t st, long n, const double* a, const double* b, double* part, int nb, double* out);

// pcg (kernels_pcg.hip); T = double or float
LBox make_lbox(const Box& b);
size_t lbox_max_cells(int N);
int pcg_sq_blocks(const LBox& L);
int pcg_xr_blocks(const LBox& L);
void launch_cnt_local(hipStream_t st, Grid g, LBox L, const uint8_t* flags, uint8_t* cnt);
template <typename T>
void launch_pcg_init(hipStream_t st, Grid g, LBox L, const uint8_t* cnt, const float* b, T* x, T* r, Coef<T> cf, double* part_bb,
                     double* part_rz0, PcgState* ps);
template <typename T>
void launch_pcg_sq(hipStream_t st, LBox L, const uint8_t* cnt, const T* r, const T* s_in, T* s_out, T* q, Coef<T> cf,
                   const double* part_rr, const double* part_rz_new, const double* part_rz_old, double* part_pq, PcgState* ps, int first,
                   double tol, int n_rz = -1, int zmode = 0, int sparse = 0, int n_prev = -1);   // n_prev: partials in part_rr (default: an XR launch's)
template <typename T>
void launch_pcg_xr(hipStream_t st, LBox L, const uint8_t* cnt, T* x, T* r, const T* s, const T* q, Coef<T> cf, const double* part_rz_cur,
                   const double* part_pq, double* part_rr, double* part_rz_next, PcgState* ps, int n_rz = -1, int sparse = 0);
void launch_pcg_sq_zf(hipStream_t st, LBox L, const uint8_t* cnt, const float* z, const double* s_in, double* s_out, double* q, Coef<double> cf,
                      const double* part_rr, int n_prev, const double* part_rz_new, const double* part_rz_old, double* part_pq, PcgState* ps, int first,
                      double tol, int n_rz, int sparse, const int* tlist, int nlist);   // z = the float cycle's result, kept float (tlist: listed tiles only)
// the same two over a list of active SQ tiles (mostly-air box).  n_prev = partials of the previous launch in part_rr (the init
// kernel's pcg_xr_blocks(L) for the first body, pcg_list_blocks(nlist) afterwards); both write pcg_list_blocks(nlist) partials
int pcg_list_blocks(int nlist);
template <typename T>
void launch_pcg_sq_list(hipStream_t st, LBox L, const uint8_t* cnt, const T* r, const T* s_in, T* s_out, T* q, Coef<T> cf,
                        const double* part_rr, int n_prev, const double* part_rz_new, const double* part_rz_old, double* part_pq, PcgState* ps,
                        int first, double tol, int n_rz, int zmode, const int* tlist, int nlist);
template <typename T>
void launch_pcg_xr_list(hipStream_t st, LBox L, const uint8_t* cnt, T* x, T* r, const T* s, const T* q, Coef<T> cf, const double* part_rz_cur,
                        int n_rz, const double* part_pq, double* part_rr, double* part_rz_next, PcgState* ps, const int* tlist, int nlist);
// ... and XR over the z rows of 32 cells that hold an unknown (finer than the tiles: the spray leaves most of a touched tile empty)
int pcg_row_count(const LBox& L);
int pcg_rows_blocks(int nrows);
void launch_row_list(hipStream_t st, LBox L, const uint8_t* cnt, int* flags, int* pos, int* list, int* block_sums, int* count);
// closed pockets of the pressure system (kernels_droplets.hip): found and taken out of cnt once per step, solved once per pass
constexpr int DROP_CAP = 65536;   // components the buffers hold (64 cells each); the ones beyond stay in the global solve
constexpr int DROP_NCTR = 64;     // ... in as many ranges, each with a counter of its own
void launch_drop_find(hipStream_t st, LBox L, uint8_t* cnt, int* ctr, int* pre, int* total, int* comp_n, int* comp_cells, const Box* own = nullptr);   // own (local-box coordinates): a claimed pocket lies wholly inside (decomposed run: the owned cells)
void launch_drop_solve(hipStream_t st, Grid g, LBox L, int n_comp, const int* pre, const int* comp_n, const int* comp_cells, const uint8_t* flags,
                       const float* b, Coef<double> cf, double tol, double* pressure, double* keep, int* n_fail = nullptr);   // n_fail += droplets whose CG stopped short of the tolerance
template <typename T>
void launch_pcg_xr_rows(hipStream_t st, LBox L, const uint8_t* cnt, T* x, T* r, const T* s, const T* q, Coef<T> cf, const double* part_rz_cur,
                        int n_rz, const double* part_pq, int n_pq, double* part_rr, double* part_rz_next, PcgState* ps, const int* rlist, int nrows);
template <typename T>
void launch_pcg_sq_dist(hipStream_t st, LBox L, const uint8_t* cnt, const T* r, const T* s_in, T* s_out, T* q, Coef<T> cf, const double* g_rr,
                        const double* g_rz_new, const double* g_rz_old, double* part_pq, PcgState* ps, int first, double tol, int zmode);
template <typename T>
void launch_pcg_xr_dist(hipStream_t st, LBox L, const uint8_t* cnt, T* x, T* r, const T* s, const T* q, Coef<T> cf, const double* g_rz_cur,
                        const double* g_pq, double* part_rr, double* part_rz_next, PcgState* ps);
void launch_sum2(hipStream_t st, const double* a, int na, const double* b, int nb, double* out_a, double* out_b);
void launch_pcg_az_dist_zf(hipStream_t st, LBox L, const uint8_t* cnt, const float* z, double* w, Coef<double> cf, double* part_zw, PcgState* ps, const int* tlist, int nlist);
void launch_pcg_cgear_upd_zf(hipStream_t st, LBox L, const uint8_t* cnt, double* x, double* r, double* s, double* q, const float* z, const double* w, const double* g,
                             double* cg, int cur, double* part_rr, PcgState* ps, int first, double tol);
void launch_pcg_poll_stage(hipStream_t st, const double* part_rr, int n, const PcgState* ps, double* out);   // out[2]: see k_pcg_poll_stage
void launch_pcg_poll_test(hipStream_t st, const double* g, PcgState* ps);
void launch_sum4(hipStream_t st, const double* a, int na, const double* b, int nb, const double* c, int nc, const double* e, int ne, double* out);
// Chronopoulos-Gear form of the decomposed PCG (one all-reduce per iteration): w = A z + partials of z.w; the fused update
template <typename T>
void launch_pcg_az_dist(hipStream_t st, LBox L, const uint8_t* cnt, const T* z, T* w, Coef<T> cf, double* part_zw, PcgState* ps, const int* tlist = nullptr,
                        int nlist = 0);   // tlist: over the listed SQ tiles only, pcg_list_blocks(nlist) partials
template <typename T>
void launch_pcg_cgear_upd(hipStream_t st, LBox L, const uint8_t* cnt, T* x, T* r, T* s, T* q, const T* z, const T* w, const double* g, double* cg, int cur,
                          double* part_rr, PcgState* ps, int first, double tol);
template <typename T>
void launch_stencil_apply(hipStream_t st, Grid g, Box box, const uint8_t* flags, const T* s, T* q, Coef<T> cf);
// dense sweep q = A s over the whole grid (kernels_stencil.hip): the LDS-DMA plane ring, or the lean march; false = nothing launched
template <typename T>
bool launch_stencil_march(hipStream_t st, Grid g, const uint8_t* flags, const T* s, T* q, Coef<T> cf, int variant, int cxcode);
template <typename T>
void launch_store_pressure(hipStream_t st, Grid g, LBox L, const uint8_t* cnt, const T* x, double* pressure, double* keep = nullptr,
                           const PcgState* ps = nullptr);
template <typename T>
void launch_pcg_init_guess(hipStream_t st, Grid g, LBox L, const uint8_t* cnt, const float* b, const double* guess, const double* guess2, double ca,
                           double cb, T* x, T* r, Coef<T> cf, double* part_bb, double* part_rr0, PcgState* ps);

// multigrid preconditioner (kernels_mg.hip)
MLevel mg_level0(const LBox& L);
MLevel mg_coarser(const MLevel& f);
void launch_mg_type0(hipStream_t st, Grid g, LBox L, MLevel m, const uint8_t* flags, const uint8_t* cnt, uint8_t* typ);
void launch_mg_coarsen(hipStream_t st, MLevel mf, const uint8_t* tf, MLevel mc, uint8_t* tc, uint8_t* cnt_c);
void launch_mg_coarsen_types(hipStream_t st, MLevel mf, const uint8_t* tf, MLevel mc, uint8_t* tc);   // the two halves of launch_mg_coarsen
void launch_mg_counts(hipStream_t st, MLevel m, const uint8_t* typ, uint8_t* cnt);
template <typename T>
void launch_mg_restrict(hipStream_t st, MLevel mf, const T* rf, MLevel mc, const uint8_t* cnt_c, T* fc, const PcgState* ps);
constexpr int MG_TAIL_MAX = 4;          // levels the single-block tail kernel can hold
constexpr size_t MG_TAIL_LDS = 144 * 1024;  // dynamic LDS the tail kernel may use (160 KB per CU on gfx950): levels whose u, v, f + counts fit go into the tail
size_t mg_tail_lds_bytes(int nl, const MLevel* lv, size_t elem);
int mg_up_blocks(const MLevel& m);
// T = the V-cycle's arithmetic/storage type, F / O = element types of a level's rhs / result (double at level 0)
template <typename T, typename F>
void launch_mg_down(hipStream_t st, MLevel m, const uint8_t* cnt, const F* f, T* u, T* r, MLevel mc, const uint8_t* cnt_c, T* fc, MgCoef<T> cf,
                    const PcgState* ps, const int* tlist = nullptr, int nlist = 0, bool pcr = false);   // pcr: fc = sum of the residual over each coarse cell's children (kernels_gal.hip)
template <typename T, typename F, typename O>
void launch_mg_up(hipStream_t st, MLevel m, const uint8_t* cnt, const F* f, const T* u, O* out, MLevel mc, const T* ec, MgCoef<T> cf,
                  double* part_dot, const PcgState* ps, double wc, const int* tlist = nullptr, int nlist = 0, const uint8_t* own = nullptr,
                  int pconst = 0);   // pconst: piecewise-constant prolongation (the Galerkin coarse levels of kernels_gal.hip)
// Galerkin coarse levels by 2 x 2 x 2 aggregation (kernels_gal.hip; mostly-air boxes): per-cell coefficients gd (diagonal) and gx, gy, gz (+face weights)
bool gal_fits_coarsest(const MLevel& m);
void launch_gal_level1(hipStream_t st, MLevel m0, const uint8_t* cnt0, MgCoef<float> cf0, MLevel m1, float* gd, float* gx, float* gy, float* gz, uint8_t* cnt1);
void launch_gal_coarsen(hipStream_t st, MLevel mf, const float* fd, const float* fx, const float* fy, const float* fz, const uint8_t* cntf, MLevel mc,
                        float* gd, float* gx, float* gy, float* gz, uint8_t* cntc);
int gal_tile_count(const MLevel& m);   // leg tiles (8^3) of a level
void launch_gal_tile_flags(hipStream_t st, MLevel m, const uint8_t* cnt, uint8_t* flags);   // flags[tile] = the tile holds an unknown
// (cnt of the two legs: the tile flags)
void launch_gal_down(hipStream_t st, MLevel m, const uint8_t* cnt, const float* gd, const float* gx, const float* gy, const float* gz, const float* f, float* u,
                     MLevel mc, float* fc, const PcgState* ps);
void launch_gal_up(hipStream_t st, MLevel m, const uint8_t* cnt, const float* gd, const float* gx, const float* gy, const float* gz, const float* f, const float* u,
                   float* out, MLevel mc, const float* ec, float wc, const PcgState* ps);
void launch_gal_coarsest(hipStream_t st, MLevel m, const float* gd, const float* gx, const float* gy, const float* gz, const float* f, float* u, int sweeps,
                         const PcgState* ps);
// Active-tile lists of a mostly-air box (level 0 only): flags per tile of the V-cycle legs / of the SQ kernel, and their
// compaction in ascending tile order (list[0..*count)); the legs, SQ and XR are then launched over the listed tiles only.
void launch_mg_tile_flags(hipStream_t st, MLevel m, const uint8_t* cnt, uint8_t* flags);
void launch_sq_tile_flags(hipStream_t st, LBox L, const uint8_t* cnt, uint8_t* flags);
int sq_tile_count(const LBox& L);
void launch_compact_flags(hipStream_t st, const uint8_t* flags, int n, int* list, int* count);
template <typename T>
void launch_mg_tail(hipStream_t st, int nl, const T* f0, const MLevel* lv, uint8_t* const* cnt, T* u0, const T* off, int sweeps, const PcgState* ps,
                    double wc);

}  // namespace fl
