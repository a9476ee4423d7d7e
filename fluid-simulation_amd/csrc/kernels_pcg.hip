// Matrix-free pressure solve for gfx950: the reference's sparse system (setA/setA2,
// fluid.cc:304-412,481-541) is a function of the per-cell flag byte alone, so the Eigen
// SparseMatrix + IncompleteCholesky PCG (fluid.cc:1352,1473-1474) becomes a PCG whose SpMV is
// an LDS-tiled 7-point stencil.  Loop structure and stopping rule follow
// Eigen/src/IterativeLinearSolvers/ConjugateGradient.h:28-90 with the diagonal (Jacobi)
// preconditioner (BasicPreconditioners.h:73,91: multiply by the stored reciprocal diagonal);
// two launches per iteration, all scalars stay on the device:
//
//   SQ: [beta from the previous launch's partial sums]  s' = invdiag r + beta s ;  q = A s' ;
//       partial s'.q                                           (4T+1 bytes per cell)
//   XR: [alpha = r.z / s.q]  x += alpha s ; r -= alpha q ; partial r.r, r.(invdiag r)
//                                                               (6T+1 bytes per cell)
//
// Every block of a launch re-sums the previous launch's per-block partials in a fixed order
// (<= 1024 doubles from L2) instead of a grid-wide atomic or an extra reduction launch, so all
// blocks take the same branch and results are run-to-run reproducible.
//
// The solver works on a BOX-LOCAL copy of the unknowns: the active box of the step (particle
// bounding box + 1) is laid out densely with a zero ring and padded, 128-byte aligned z rows,
// so tiles need no bounds checks, XR is a flat 16-byte-vector stream, and the working set of
// the drop scene (~1M cells) stays in L2/Infinity Cache.
//
// Bandwidth/latency-bound fp64 work: no MFMA.
#include "common.h"
#include "stencil_vec.h"

namespace fl {

// static indices only: a runtime index into a by-value kernel argument would go through scratch
template <typename T>
__device__ __forceinline__ void load_coef(T* sdiag, T* sinv, const Coef<T>& cf)
{
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            sdiag[i] = cf.diag[i];
            sinv[i] = cf.inv[i];
        }
    }
}

// ================================================================================================
// Box-local PCG
// ================================================================================================
constexpr int TX = 4, TY = 8, TZ = 32;       // tile of the SQ kernel (x,y,z), 256 threads
constexpr int PY = TY + 2, PZ = TZ + 2;      // LDS plane with y/z halo
constexpr int SQ_MAX_BLOCKS = 1024;
constexpr int XR_MAX_BLOCKS = 512;

LBox make_lbox(const Box& b)
{
    LBox L;
    L.x0 = b.x0; L.y0 = b.y0; L.z0 = b.z0;
    L.nx = b.nx(); L.ny = b.ny(); L.nz = b.nz();
    L.Lx = (L.nx + TX - 1) / TX * TX + 2;
    L.Ly = (L.ny + TY - 1) / TY * TY + 2;
    // z rows: K0 zero cells in front (128-byte aligned interior), interior, >= 1 zero cell behind, rounded to 16.
    // The last z tile may overrun the row by up to TZ cells: it then reads/writes the NEXT row's front padding
    // (zeros, never an unknown), which needs the row to be longer than the tiled extent.  Arrays carry one
    // spare row at the end for the very last wrap.
    const int tiled = (L.nz + TZ - 1) / TZ * TZ + 1;
    const int need = LBOX_K0 + L.nz + 1 > tiled ? LBOX_K0 + L.nz + 1 : tiled;
    L.Lz = (need + 15) / 16 * 16;
    return L;
}
size_t lbox_max_cells(int N)
{
    Box b{0, 0, 0, N - 1, N - 1, N - 1};
    const LBox L = make_lbox(b);
    return L.cells() + (size_t)2 * L.Lz;  // + spare rows for the wrap of the last tile
}
static inline int sq_tiles(const LBox& L) { return ((L.nx + TX - 1) / TX) * ((L.ny + TY - 1) / TY) * ((L.nz + TZ - 1) / TZ); }
int pcg_sq_blocks(const LBox& L)
{
    int n = sq_tiles(L);
    return n < SQ_MAX_BLOCKS ? (n < 1 ? 1 : n) : SQ_MAX_BLOCKS;
}
int pcg_xr_blocks(const LBox& L)
{
    long n = ((long)L.cells() / 2 + 256 * 4 - 1) / (256 * 4);
    return (int)(n < XR_MAX_BLOCKS ? (n < 1 ? 1 : n) : XR_MAX_BLOCKS);
}

// local cell -> diag count (0 = not an unknown).  One launch per step.
__global__ __launch_bounds__(256) void k_cnt_local(Grid g, LBox L, const uint8_t* __restrict__ flags, uint8_t* __restrict__ cnt)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)L.cells() + 2 * L.Lz) return;  // + the spare rows the last tile may wrap into: never unknowns
    const int k = (int)(t % L.Lz), j = (int)((t / L.Lz) % L.Ly), i = (int)(t / ((long)L.Lz * L.Ly));
    uint8_t c = 0;
    if (i >= 1 && i <= L.nx && j >= 1 && j <= L.ny && k >= LBOX_K0 && k < LBOX_K0 + L.nz) {
        const uint8_t f = flags[g.idx(L.x0 + i - 1, L.y0 + j - 1, L.z0 + k - LBOX_K0)];
        if (f & F_FLUID) c = f >> F_CNT_SHIFT;  // Adiag == 0 rows are not unknowns (fluid.cc:498)
    }
    cnt[t] = c;
}

// x = 0, r = b (ConjugateGradient.h:41), partial |b|^2 and r.(invdiag r) (:62-65)
template <typename T>
__global__ __launch_bounds__(256) void k_pcg_init_l(Grid g, LBox L, const uint8_t* __restrict__ cnt, const float* __restrict__ b,
                                                    T* __restrict__ x, T* __restrict__ r, Coef<T> cf, double* __restrict__ part_bb,
                                                    double* __restrict__ part_rz0, PcgState* ps)
{
    __shared__ double red[4];
    __shared__ T sdiag[8], sinv[8];
    load_coef(sdiag, sinv, cf);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) { ps->done = 0; ps->iters = 0; ps->breakdown = 0; ps->bb = 0; ps->thr = 0; ps->rr = 0; }
    const long n = (long)L.cells();
    double abb = 0, arz = 0;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) {
        const uint8_t c = cnt[t];
        T rv = 0;
        if (c && !(c & 0x80)) {   // (bit 7: a ring cell of a decomposed run's local box: another rank's unknown)
            const int k = (int)(t % L.Lz), j = (int)((t / L.Lz) % L.Ly), i = (int)(t / ((long)L.Lz * L.Ly));
            rv = (T)b[g.idx(L.x0 + i - 1, L.y0 + j - 1, L.z0 + k - LBOX_K0)];
            const T z = rv * sinv[c];
            abb += (double)rv * (double)rv;
            arz += (double)rv * (double)z;
        }
        x[t] = 0;
        r[t] = rv;
    }
    abb = block_sum<double, 4>(abb, red);
    arz = block_sum<double, 4>(arz, red);
    if (threadIdx.x == 0) { part_bb[blockIdx.x] = abb; part_rz0[blockIdx.x] = arz; }
}

// Start from a guess (Eigen's solveWithGuess, ConjugateGradient.h:37-43: residual = rhs - mat * x): x0 = `guess` (a global
// field, e.g. the previous solve's pressure) at the unknown cells, r0 = b - A x0, partials of b.b (threshold) and r0.r0.
// A neighbour contributes only if it is an unknown NOW (its count byte), whatever the guess field holds there.
template <typename T>
__global__ __launch_bounds__(256) void k_pcg_init_guess_l(Grid g, LBox L, const uint8_t* __restrict__ cnt, const float* __restrict__ b,
                                                          const double* __restrict__ guess, const double* __restrict__ guess2, double ca, double cb,
                                                          T* __restrict__ x, T* __restrict__ r, Coef<T> cf,
                                                          double* __restrict__ part_bb, double* __restrict__ part_rr0, PcgState* ps)
{
    // x0 = ca * guess (+ cb * guess2 when given: the extrapolation over the last two passes of the step's do..while)
    __shared__ double red[4];
    __shared__ T sdiag[8], sinv[8];
    load_coef(sdiag, sinv, cf);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) { ps->done = 0; ps->iters = 0; ps->breakdown = 0; ps->bb = 0; ps->thr = 0; ps->rr = 0; }
    const long n = (long)L.cells();
    const long sx = (long)L.Ly * L.Lz, sy = L.Lz;
    double abb = 0, arr = 0;
    // A mostly-air box is 30 rounds of one count byte and two stores per thread: four rounds' bytes are asked for together (the cells of a
    // thread, and the order it adds them in, stay what they were).
    const long stride = (long)gridDim.x * 256;
    for (long t0 = (long)blockIdx.x * 256 + threadIdx.x; t0 < n; t0 += 4 * stride) {
        uint8_t c4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const long t = t0 + q * stride; c4[q] = cnt[t < n ? t : n - 1]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long t = t0 + q * stride;
            if (t >= n) break;
            const uint8_t c = c4[q];
            T rv = 0, xv = 0;
            if (c && !(c & 0x80)) {  // unknowns are interior cells: all six neighbours exist in both layouts (bit 7: ring cell of a decomposed run, another rank's unknown — it counts as a neighbour below, not here)
                const int k = (int)(t % L.Lz), j = (int)((t / L.Lz) % L.Ly), i = (int)(t / ((long)L.Lz * L.Ly));
                const size_t gc = g.idx(L.x0 + i - 1, L.y0 + j - 1, L.z0 + k - LBOX_K0);
                const size_t gx = (size_t)g.sx(), gy = (size_t)g.nz;
                const T bv = (T)b[gc];
                auto G = [&](size_t q2) { return guess2 ? (T)(ca * guess[q2] + cb * guess2[q2]) : (T)guess[q2]; };
                // (the six neighbours' counts and guesses are all asked for, then chosen: a load behind a branch on another load is a round trip each)
                const uint8_t n0 = cnt[t - sx], n1 = cnt[t + sx], n2 = cnt[t - sy], n3 = cnt[t + sy], n4 = cnt[t - 1], n5 = cnt[t + 1];
                const T g0 = G(gc - gx), g1 = G(gc + gx), g2 = G(gc - gy), g3 = G(gc + gy), g4 = G(gc - 1), g5 = G(gc + 1);
                xv = G(gc);
                const T nb = (n0 ? g0 : (T)0) + (n1 ? g1 : (T)0) + (n2 ? g2 : (T)0) + (n3 ? g3 : (T)0) + (n4 ? g4 : (T)0) + (n5 ? g5 : (T)0);
                rv = bv - (sdiag[c] * xv + cf.off * nb);
                abb += (double)bv * (double)bv;
                arr += (double)rv * (double)rv;
            }
            x[t] = xv;
            r[t] = rv;
        }
    }
    abb = block_sum<double, 4>(abb, red);
    arr = block_sum<double, 4>(arr, red);
    if (threadIdx.x == 0) { part_bb[blockIdx.x] = abb; part_rr0[blockIdx.x] = arr; }
}

// Start of a CG body (ConjugateGradient.h:45-60 for the first, :75-85 for the others): sums the
// previous launch's partials (or reads the all-reduced scalars when n_prev == 1), decides
// convergence identically in every block, returns beta.  false = this block must exit.
template <typename T>
__device__ __forceinline__ bool pcg_head(const double* __restrict__ part_rr, const double* __restrict__ part_rz_new,
                                         const double* __restrict__ part_rz_old, int n_prev, int n_rz, PcgState* ps, int first, double tol,
                                         double* red, T& beta)
{
    const int tid = threadIdx.x;
    beta = 0;
    if (first) {
        // b == 0 -> x = 0, done.  first == 2 (started from a guess): part_rz_old holds the partials of r0.r0, and a guess
        // that already meets the threshold ends the solve with 0 iterations (ConjugateGradient.h:51-56)
        double bb, rr0, d2;
        block_sum3(part_rr, n_prev, part_rz_old, first == 2 ? n_prev : 0, part_rr, 0, red, bb, rr0, d2);
        if (first != 2) rr0 = bb;
        const bool go = bb > 0 && !(rr0 < tol * tol * bb);
        if (blockIdx.x == 0 && tid == 0) {
            ps->bb = bb;
            ps->thr = tol * tol * bb;
            ps->rr = rr0;
            if (!go) ps->done = 1;
        }
        return go;
    }
    double rr, rzn, rzo;
    block_sum3(part_rr, n_prev, part_rz_new, n_rz, part_rz_old, n_rz, red, rr, rzn, rzo);
    if (rr < ps->thr) {  // break before i++
        if (blockIdx.x == 0 && tid == 0) { ps->rr = rr; ps->done = 1; }
        return false;
    }
    beta = (T)(rzn / rzo);  // :82-83
    if (blockIdx.x == 0 && tid == 0) { ps->rr = rr; ps->iters += 1; }  // :85
    return true;
}

// SQ: s' = invdiag r + beta s ; q = A s' ; partial s'.q
// Thread (ly = tid>>5, kz = tid&31) owns the x-column lx = -1..TX of its (y,z): the x
// neighbours stay in registers, y/z neighbours go through LDS.  All global loads of the tile
// are issued BEFORE the partial-sum reduction that yields beta, so both latencies overlap.
// DIST (decomposed run): the local box is the rank's owned cells + a halo; the count byte carries bit 7 on the one ring
// of halo cells next to the owned ones (z and s are valid there, so s' is formed there too and no exchange of s' is
// needed) and is 0 beyond; q and the partial s'.q are formed on the owned cells only.  The scalars are single
// all-reduced values (n_prev = n_rz = 1).
template <typename T, bool DIST>
__global__ __launch_bounds__(256) void k_pcg_sq_l(LBox L, const uint8_t* __restrict__ cnt, const T* __restrict__ r,
                                                  const T* __restrict__ s_in, T* __restrict__ s_out, T* __restrict__ q, Coef<T> cf,
                                                  const double* __restrict__ part_rr, const double* __restrict__ part_rz_new,
                                                  const double* __restrict__ part_rz_old, double* __restrict__ part_pq, int n_prev,
                                                  PcgState* ps, int first, double tol, int n_rz, int zmode, int sparse,
                                                  const int* __restrict__ tlist, int nlist)
{
    // tlist: only the listed tiles (those holding an unknown, ascending) are swept, block b takes entries b, b + gridDim.x, ...
    // sparse (mostly-air box): a tile's count bytes are loaded and tested first and a tile without an unknown is
    // skipped (its s', q stay 0 / unread) — no prefetch of the next tile then; dense boxes keep the pipelined loads
    // zmode: `r` is already z = M^-1 r (multigrid preconditioner); otherwise z = invdiag r is formed here
    __shared__ T sT[TX * PY * PZ];
    __shared__ double red[16];
    __shared__ int s_done;
    __shared__ T sdiag[8], sinv[8];
    const int tid = threadIdx.x;
    load_coef(sdiag, sinv, cf);
    // one read per block, broadcast: block 0 of THIS launch may set done while we start
    if (tid == 0) s_done = ps->done;

    const int nty = (L.ny + TY - 1) / TY, ntz = (L.nz + TZ - 1) / TZ;
    const int ntiles = ((L.nx + TX - 1) / TX) * nty * ntz;
    const long sx = (long)L.Ly * L.Lz;
    const int ly = tid >> 5, kz = tid & 31;

    int lidx = xcd_remap(blockIdx.x, gridDim.x);  // neighbouring list entries (tiles) on the same XCD
    int tile = tlist ? (lidx < nlist ? tlist[lidx] : ntiles) : lidx;
    // the list entry of the round after this one is fetched a round early: an address that hangs on a load issued a moment ago stalls the issue of the whole tile
    auto entry = [&](int li) { return tlist ? (li < nlist ? tlist[li < nlist ? li : 0] : ntiles) : li; };
    int nxt = entry(lidx + (int)gridDim.x);
    // ---- issue the first tile's loads -------------------------------------------------------
    uint8_t fc[TX + 2], fy = 0, fz = 0;
    T rv[TX + 2], sv[TX + 2], ry = 0, sy = 0, rz = 0, sz = 0;
    long c0 = 0;
    auto issue = [&](int tl, bool counts, bool values) {
        const int tz = tl % ntz, ty = (tl / ntz) % nty, tx = tl / (ntz * nty);
        const int i0 = 1 + tx * TX, j0 = 1 + ty * TY, k0 = LBOX_K0 + tz * TZ;
        c0 = ((long)(i0 - 1) * L.Ly + (j0 + ly)) * L.Lz + k0 + kz;
#pragma unroll
        for (int m = 0; m < TX + 2; ++m) {
            if (counts) fc[m] = cnt[c0 + m * sx];
            if (values) {
                rv[m] = r[c0 + m * sx];
                sv[m] = first ? (T)0 : s_in[c0 + m * sx];
            }
        }
        {   // y halo: one cell per thread
            const int pl = tid >> 6, side = (tid >> 5) & 1;
            const long cy = ((long)(i0 + pl) * L.Ly + (side ? j0 + TY : j0 - 1)) * L.Lz + k0 + kz;
            if (counts) fy = cnt[cy];
            if (values) { ry = r[cy]; sy = first ? (T)0 : s_in[cy]; }
        }
        if (tid < 64) {  // z halo: 64 cells
            const int lx = tid >> 4, l2 = (tid >> 1) & 7, side = tid & 1;
            const long cz = ((long)(i0 + lx) * L.Ly + (j0 + l2)) * L.Lz + (side ? k0 + TZ : k0 - 1);
            if (counts) fz = cnt[cz];
            if (values) { rz = r[cz]; sz = first ? (T)0 : s_in[cz]; }
        }
    };
    if (!sparse && tile < ntiles) issue(tile, true, true);

    // ---- scalars ------------------------------------------------------------------------------
    T beta = 0;
    __syncthreads();  // s_done, coef tables
    if (s_done) return;
    if (!pcg_head<T>(part_rr, part_rz_new, part_rz_old, n_prev, n_rz, ps, first, tol, red, beta)) return;

    double acc = 0;
    while (tile < ntiles) {
        if (sparse) {
            issue(tile, true, false);
            int any = 0;
#pragma unroll
            for (int m = 1; m <= TX; ++m) any |= fc[m];
            if (!__syncthreads_or(any)) {  // (also the barrier that frees the LDS tile of the previous round)
                tile = nxt;
                lidx += gridDim.x;
                nxt = entry(lidx + (int)gridDim.x);
                continue;
            }
            issue(tile, false, true);
        }
        // ---- combine -> LDS ---------------------------------------------------------------------
        T val[TX + 2];
#pragma unroll
        for (int m = 0; m < TX + 2; ++m) val[m] = fc[m] ? (zmode ? rv[m] : rv[m] * sinv[DIST ? fc[m] & 7 : fc[m]]) + beta * sv[m] : (T)0;
#pragma unroll
        for (int lx = 0; lx < TX; ++lx) sT[(lx * PY + ly + 1) * PZ + kz + 1] = val[lx + 1];
        {
            const int pl = tid >> 6, side = (tid >> 5) & 1;
            sT[(pl * PY + (side ? TY + 1 : 0)) * PZ + kz + 1] = fy ? (zmode ? ry : ry * sinv[DIST ? fy & 7 : fy]) + beta * sy : (T)0;
        }
        if (tid < 64) {
            const int lx = tid >> 4, l2 = (tid >> 1) & 7, side = tid & 1;
            sT[(lx * PY + l2 + 1) * PZ + (side ? TZ + 1 : 0)] = fz ? (zmode ? rz : rz * sinv[DIST ? fz & 7 : fz]) + beta * sz : (T)0;
        }
        __syncthreads();
        const long cc = c0;
        uint8_t fcc[TX];
        T nbv[TX], cen[TX];
#pragma unroll
        for (int lx = 0; lx < TX; ++lx) {
            const int o = (lx * PY + ly + 1) * PZ + kz + 1;
            fcc[lx] = fc[lx + 1];
            cen[lx] = val[lx + 1];
            nbv[lx] = val[lx] + val[lx + 2] + sT[o - PZ] + sT[o + PZ] + sT[o - 1] + sT[o + 1];
        }
        const int next = nxt;
        lidx += gridDim.x;
        nxt = entry(lidx + (int)gridDim.x);
        __syncthreads();  // LDS free for the next tile
        if (!sparse && next < ntiles) issue(next, true, true);  // next tile's loads fly while this tile finishes
#pragma unroll
        for (int lx = 0; lx < TX; ++lx) {
            const long c = cc + (lx + 1) * sx;
            T qv = 0;
            if (DIST ? (fcc[lx] && !(fcc[lx] & 0x80)) : (fcc[lx] != 0)) {
                qv = sdiag[fcc[lx]] * cen[lx] + cf.off * nbv[lx];
                acc += (double)cen[lx] * (double)qv;
            }
            s_out[c] = cen[lx];
            q[c] = qv;
        }
        tile = next;
    }
    acc = block_sum<double, 4>(acc, red);
    if (tid == 0) part_pq[blockIdx.x] = acc;
}

template <typename T>
struct alignas(2 * sizeof(T)) Vec2 {
    T a, b;
};

// out_a[0] = sum a[0..na), out_b[0] = sum b[0..nb)   (one block; fixed order)
__global__ __launch_bounds__(256) void k_sum2(const double* __restrict__ a, int na, const double* __restrict__ b, int nb,
                                              double* __restrict__ out_a, double* __restrict__ out_b)
{
    __shared__ double red[16];
    double ra, rb, rc;
    block_sum3(a, na, b, nb, b, 0, red, ra, rb, rc);
    if (threadIdx.x == 0) {
        out_a[0] = ra;
        if (out_b) out_b[0] = rb;
    }
}

// XR: alpha, x += alpha s, r -= alpha q, partial |r|^2 and r.(invdiag r)   (ConjugateGradient.h:70-74,79-81)
// Flat stream over the local box, two cells (16 bytes) per lane per access, 4 accesses in flight.
template <typename T, bool SPARSE>
__global__ __launch_bounds__(256) void k_pcg_xr_l(long n2, const uint8_t* __restrict__ cnt, T* __restrict__ x, T* __restrict__ r,
                                                  const T* __restrict__ s, const T* __restrict__ q, Coef<T> cf,
                                                  const double* __restrict__ part_rz_cur, int n_xr, const double* __restrict__ part_pq,
                                                  int n_sq, double* __restrict__ part_rr, double* __restrict__ part_rz_next, PcgState* ps,
                                                  int dist)
{
    // dist (decomposed run): count bytes with bit 7 are the ring cells of the local box (another rank's unknowns): skipped
    // SPARSE (box mostly air): the four vector loads of a pair are issued only if it holds an unknown — the count words of a
    // round are loaded together, then the vectors: one dependent load on the critical path per round, a fraction of the traffic.
    // Dense boxes load everything unconditionally (index clamped into the array, the result masked where it is used): a
    // load behind a test of another load's result, or inside a branch, makes the compiler drain the queue at every access.
    __shared__ double red[16];
    __shared__ int s_done;
    __shared__ T sdiag[8], sinv[8];
    load_coef(sdiag, sinv, cf);
    if (threadIdx.x == 0) s_done = ps->done;
    typedef Vec2<T> V2;
    const V2* x2 = (const V2*)x;
    const V2* r2 = (const V2*)r;
    const V2* s2 = (const V2*)s;
    const V2* q2 = (const V2*)q;
    const uint16_t* c2 = (const uint16_t*)cnt;
    // Block b streams ONE contiguous chunk, and chunks are handed out XCD-contiguously like the SQ tiles
    // (x-major), so the slab an XCD wrote in SQ is the slab it reads here: its L2 still holds part of it.
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long e0 = (long)vb * per, e1 = e0 + per < n2 ? e0 + per : n2;
    constexpr int U = 4;
    V2 xv[U], rv[U], sv[U], qv[U];
    uint16_t cv[U];
    long i = e0 + threadIdx.x;
    auto issue = [&](long base) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long e = base + u * 256, ec = e < n2 ? e : n2 - 1;
            cv[u] = c2[ec];
            if (!SPARSE) { xv[u] = x2[ec]; rv[u] = r2[ec]; sv[u] = s2[ec]; qv[u] = q2[ec]; }
        }
        if (SPARSE) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long e = base + u * 256;
                if (e >= e1) cv[u] = 0;
                if (cv[u]) { xv[u] = x2[e]; rv[u] = r2[e]; sv[u] = s2[e]; qv[u] = q2[e]; }
            }
        }
    };
    issue(i);
    __syncthreads();
    if (s_done) return;
    double rz, pq, d3;
    block_sum3(part_rz_cur, n_xr, part_pq, n_sq, part_pq, 0, red, rz, pq, d3);
    if (!(pq > 0) || !(rz == rz)) {  // not SPD / NaN: stop instead of spreading NaNs
        if (blockIdx.x == 0 && threadIdx.x == 0) { ps->breakdown = 1; ps->done = 1; }
        return;
    }
    const T alpha = (T)(rz / pq);
    double arr = 0, arz = 0;
    for (; i < e1; i += U * 256) {
        V2 xo[U], ro[U];
        uint16_t cw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cw[u] = i + u * 256 < e1 ? cv[u] : (uint16_t)0;
            xo[u] = xv[u]; ro[u] = rv[u];
            if (cw[u]) {
                int ca = cw[u] & 0xff, cb = cw[u] >> 8;
                if (dist) { ca = (ca & 0x80) ? 0 : ca; cb = (cb & 0x80) ? 0 : cb; }
                if (ca) {
                    xo[u].a = xo[u].a + alpha * sv[u].a;
                    ro[u].a = ro[u].a - alpha * qv[u].a;
                    const T z = ro[u].a * sinv[ca];
                    arr += (double)ro[u].a * (double)ro[u].a;
                    arz += (double)ro[u].a * (double)z;
                }
                if (cb) {
                    xo[u].b = xo[u].b + alpha * sv[u].b;
                    ro[u].b = ro[u].b - alpha * qv[u].b;
                    const T z = ro[u].b * sinv[cb];
                    arr += (double)ro[u].b * (double)ro[u].b;
                    arz += (double)ro[u].b * (double)z;
                }
            }
        }
        const long ib = i;
        if (!SPARSE || i + U * 256 < e1) issue(i + U * 256);   // the next round's loads fly while this one is written
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (cw[u]) {
                ((V2*)x)[ib + u * 256] = xo[u];
                ((V2*)r)[ib + u * 256] = ro[u];
            }
    }
    arr = block_sum<double, 4>(arr, red);
    arz = block_sum<double, 4>(arz, red);
    if (threadIdx.x == 0) { part_rr[blockIdx.x] = arr; part_rz_next[blockIdx.x] = arz; }
}


// XR over the listed SQ tiles (mostly-air box): the same update and partials as k_pcg_xr_l, thread = one (y, z) column of the
// tile's TX planes, block b takes list entries b, b + gridDim.x, ...
template <typename T>
__global__ __launch_bounds__(256) void k_pcg_xr_t(LBox L, const uint8_t* __restrict__ cnt, T* __restrict__ x, T* __restrict__ r,
                                                  const T* __restrict__ s, const T* __restrict__ q, Coef<T> cf,
                                                  const double* __restrict__ part_rz_cur, int n_xr, const double* __restrict__ part_pq,
                                                  int n_sq, double* __restrict__ part_rr, double* __restrict__ part_rz_next, PcgState* ps,
                                                  const int* __restrict__ tlist, int nlist)
{
    __shared__ double red[16];
    __shared__ int s_done;
    __shared__ T sdiag[8], sinv[8];
    load_coef(sdiag, sinv, cf);
    if (threadIdx.x == 0) s_done = ps->done;
    const int nty = (L.ny + TY - 1) / TY, ntz = (L.nz + TZ - 1) / TZ;
    const long sx = (long)L.Ly * L.Lz;
    const int ly = threadIdx.x >> 5, kz = threadIdx.x & 31;
    uint8_t cv[TX];
    T xv[TX], rv[TX], sv[TX], qv[TX];
    long c0 = 0;
    auto issue = [&](int tl) {
        const int tz = tl % ntz, ty = (tl / ntz) % nty, tx = tl / (ntz * nty);
        c0 = ((long)(1 + tx * TX) * L.Ly + (1 + ty * TY + ly)) * L.Lz + LBOX_K0 + tz * TZ + kz;
#pragma unroll
        for (int m = 0; m < TX; ++m) {
            const long c = c0 + m * sx;
            cv[m] = cnt[c];
            xv[m] = x[c]; rv[m] = r[c]; sv[m] = s[c]; qv[m] = q[c];
        }
    };
    int lidx = xcd_remap(blockIdx.x, gridDim.x);
    if (lidx < nlist) issue(tlist[lidx]);
    __syncthreads();
    if (s_done) return;
    double rz, pq, d3;
    block_sum3(part_rz_cur, n_xr, part_pq, n_sq, part_pq, 0, red, rz, pq, d3);
    if (!(pq > 0) || !(rz == rz)) {  // not SPD / NaN: stop instead of spreading NaNs
        if (blockIdx.x == 0 && threadIdx.x == 0) { ps->breakdown = 1; ps->done = 1; }
        return;
    }
    const T alpha = (T)(rz / pq);
    double arr = 0, arz = 0;
    while (lidx < nlist) {
        const long cc = c0;
        uint8_t cw[TX];
        T xo[TX], ro[TX];
#pragma unroll
        for (int m = 0; m < TX; ++m) {
            cw[m] = cv[m];
            xo[m] = xv[m] + alpha * sv[m];
            ro[m] = rv[m] - alpha * qv[m];
        }
        lidx += gridDim.x;
        if (lidx < nlist) issue(tlist[lidx]);   // the next tile's loads fly while this one is written
#pragma unroll
        for (int m = 0; m < TX; ++m) {
            if (cw[m]) {
                const T z = ro[m] * sinv[cw[m]];
                arr += (double)ro[m] * (double)ro[m];
                arz += (double)ro[m] * (double)z;
                x[cc + m * sx] = xo[m];
                r[cc + m * sx] = ro[m];
            }
        }
    }
    arr = block_sum<double, 4>(arr, red);
    arz = block_sum<double, 4>(arz, red);
    if (threadIdx.x == 0) { part_rr[blockIdx.x] = arr; part_rz_next[blockIdx.x] = arz; }
}

// XR over a list of z ROWS of 32 cells (mostly-air box).  The tile lists above skip the tiles without an unknown, but the spray
// touches two thirds of the tiles with a droplet or two each: 4 700 tiles x 1 024 cells swept for 1.44 M unknowns.  A row =
// 32 consecutive z cells of one (x, y) line (one quarter-wave, 256 bytes per vector); rows without an unknown hold zeros in
// every solver vector and need neither reading nor writing.  Row r of the list -> (ix, iy, tz) by division; a wave takes two
// rows, a block 8, block b strides the list.  Same update, same partial-sum scheme as k_pcg_xr_t.
template <typename T>
__global__ __launch_bounds__(256) void k_pcg_xr_rows(LBox L, const uint8_t* __restrict__ cnt, T* __restrict__ x, T* __restrict__ r,
                                                     const T* __restrict__ s, const T* __restrict__ q, Coef<T> cf,
                                                     const double* __restrict__ part_rz_cur, int n_xr, const double* __restrict__ part_pq,
                                                     int n_sq, double* __restrict__ part_rr, double* __restrict__ part_rz_next, PcgState* ps,
                                                     const int* __restrict__ rlist, int nrows)
{
    __shared__ double red[16];
    __shared__ int s_done;
    __shared__ T sdiag[8], sinv[8];
    load_coef(sdiag, sinv, cf);
    if (threadIdx.x == 0) s_done = ps->done;
    const int ntz = (L.nz + 31) / 32;
    const int sub = threadIdx.x >> 5, kz = threadIdx.x & 31;
    // U rows in flight per thread.  Every load is unconditional (a list index past the end is clamped to the last row and the
    // count byte masked where it is used) and a row's list entry is fetched one round before its vectors: a load inside a
    // branch, or an address that hangs on a load issued a moment ago, drains the whole queue at every row.
    constexpr int U = 2;
    long c[U];
    uint8_t cv[U];
    T xv[U], rv[U], sv[U], qv[U];
    int rown[U];
    const int stride = gridDim.x * 8;
    int ri = xcd_remap(blockIdx.x, gridDim.x) * 8 + sub;
    auto row_at = [&](int rr) { return rlist[rr < nrows ? rr : nrows - 1]; };
    auto cell_row = [&](int row) {
        const int tz = row % ntz, iy = (row / ntz) % L.ny, ix = row / (ntz * L.ny);
        return ((long)(1 + ix) * L.Ly + (1 + iy)) * L.Lz + LBOX_K0 + tz * 32 + kz;
    };
    auto issue = [&](int u, int row) {
        c[u] = cell_row(row);
        cv[u] = cnt[c[u]];
        xv[u] = x[c[u]]; rv[u] = r[c[u]]; sv[u] = s[c[u]]; qv[u] = q[c[u]];
    };
#pragma unroll
    for (int u = 0; u < U; ++u) rown[u] = row_at(ri + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        issue(u, rown[u]);
        rown[u] = row_at(ri + (U + u) * stride);
    }
    __syncthreads();
    if (s_done) return;
    double rz, pq, d3;
    block_sum3(part_rz_cur, n_xr, part_pq, n_sq, part_pq, 0, red, rz, pq, d3);
    if (!(pq > 0) || !(rz == rz)) {  // not SPD / NaN: stop instead of spreading NaNs
        if (blockIdx.x == 0 && threadIdx.x == 0) { ps->breakdown = 1; ps->done = 1; }
        return;
    }
    const T alpha = (T)(rz / pq);
    double arr = 0, arz = 0;
    for (; ri < nrows; ri += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long cc = c[u];
            const uint8_t cw = ri + u * stride < nrows ? cv[u] : (uint8_t)0;
            const T xo = xv[u] + alpha * sv[u], ro = rv[u] - alpha * qv[u];
            issue(u, rown[u]);                              // the row after next: its loads fly while this one is written
            rown[u] = row_at(ri + (2 * U + u) * stride);    // ... and the list entry of the one after that
            if (cw) {
                const T z = ro * sinv[cw];
                arr += (double)ro * (double)ro;
                arz += (double)ro * (double)z;
                x[cc] = xo;
                r[cc] = ro;
            }
        }
    }
    arr = block_sum<double, 4>(arr, red);
    arz = block_sum<double, 4>(arz, red);
    if (threadIdx.x == 0) { part_rr[blockIdx.x] = arr; part_rz_next[blockIdx.x] = arz; }
}
// flags[row] = 1 if the 32-cell z row holds an unknown (rows numbered (ix * ny + iy) * ntz + tz over the box interior)
__global__ __launch_bounds__(256) void k_row_flags(LBox L, const uint8_t* __restrict__ cnt, int nrows_all, int* __restrict__ flags)
{
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5), kz = threadIdx.x & 31;
    const int ntz = (L.nz + 31) / 32;
    int any = 0;
    if (row < nrows_all) {
        const int tz = row % ntz, iy = (row / ntz) % L.ny, ix = row / (ntz * L.ny);
        any = cnt[((long)(1 + ix) * L.Ly + (1 + iy)) * L.Lz + LBOX_K0 + tz * 32 + kz];   // (beyond nz the padding holds 0)
    }
    const unsigned long long m = __ballot(any != 0);
    if (kz == 0 && row < nrows_all) flags[row] = ((m >> (threadIdx.x & 32)) & 0xFFFFFFFFull) != 0;
}
__global__ __launch_bounds__(256) void k_row_scatter(int n, const int* __restrict__ flags, const int* __restrict__ pos, int* __restrict__ list)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && flags[i]) list[pos[i]] = i;
}

// flags[t] = SQ tile t (numbered as k_pcg_sq_l decodes it) holds an unknown
__global__ __launch_bounds__(256) void k_sq_tile_flags(LBox L, const uint8_t* __restrict__ cnt, uint8_t* __restrict__ flags)
{
    const int nty = (L.ny + TY - 1) / TY, ntz = (L.nz + TZ - 1) / TZ;
    const int tl = blockIdx.x;
    const int tz = tl % ntz, ty = (tl / ntz) % nty, tx = tl / (ntz * nty);
    const int ly = threadIdx.x >> 5, kz = threadIdx.x & 31;
    const long c0 = ((long)(1 + tx * TX) * L.Ly + (1 + ty * TY + ly)) * L.Lz + LBOX_K0 + tz * TZ + kz;
    int any = 0;
#pragma unroll
    for (int m = 0; m < TX; ++m) any |= cnt[c0 + m * (long)L.Ly * L.Lz];
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) flags[tl] = any != 0;
}

// list[0..count) = the indices i < n with flags[i] != 0, ascending (one block: a few thousand tiles)
__global__ __launch_bounds__(1024) void k_compact_flags(const uint8_t* __restrict__ flags, int n, int* __restrict__ list, int* __restrict__ count)
{
    __shared__ int wsum[16];
    __shared__ int base;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int b0 = 0; b0 < n; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        const bool f = i < n && flags[i];
        const unsigned long long m = __ballot(f);
        if (lane == 0) wsum[w] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int k = 0; k < w; ++k) off += wsum[k];
        if (f) list[off + __popcll(m & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int k = 0; k < 16; ++k) t += wsum[k];
            base += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base;
}

// local x -> global pressure field (VectorXd p scattered back to cells, fluid.cc:637)
template <typename T>
__global__ __launch_bounds__(256) void k_store_pressure_l(Grid g, LBox L, const uint8_t* __restrict__ cnt, const T* __restrict__ x,
                                                          double* __restrict__ pressure, double* __restrict__ keep, const PcgState* ps)
{
    // keep != nullptr: a second copy that no per-step clearing touches (the next solve's starting guess); the solve
    // started from a guess, and for b == 0 the answer is x = 0 whatever the guess was (ConjugateGradient.h:45-50)
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)L.cells()) return;
    const int k = (int)(t % L.Lz), j = (int)((t / L.Lz) % L.Ly), i = (int)(t / ((long)L.Lz * L.Ly));
    if (i >= 1 && i <= L.nx && j >= 1 && j <= L.ny && k >= LBOX_K0 && k < LBOX_K0 + L.nz) {
        const bool zero = keep && !(ps->bb > 0);
        const double pv = cnt[t] && !(cnt[t] & 0x80) && !zero ? (double)x[t] : 0.0;
        const size_t c = g.idx(L.x0 + i - 1, L.y0 + j - 1, L.z0 + k - LBOX_K0);
        pressure[c] = pv;
        if (keep) keep[c] = pv;
    }
}

void launch_cnt_local(hipStream_t st, Grid g, LBox L, const uint8_t* flags, uint8_t* cnt)
{
    hipLaunchKernelGGL(k_cnt_local, dim3((unsigned)((L.cells() + 2 * L.Lz + 255) / 256)), dim3(256), 0, st, g, L, flags, cnt);
}
template <typename T>
void launch_pcg_init(hipStream_t st, Grid g, LBox L, const uint8_t* cnt, const float* b, T* x, T* r, Coef<T> cf, double* part_bb,
                     double* part_rz0, PcgState* ps)
{
    hipLaunchKernelGGL((k_pcg_init_l<T>), dim3(pcg_xr_blocks(L)), dim3(256), 0, st, g, L, cnt, b, x, r, cf, part_bb, part_rz0, ps);
}
template <typename T>
void launch_pcg_init_guess(hipStream_t st, Grid g, LBox L, const uint8_t* cnt, const float* b, const double* guess, const double* guess2, double ca,
                           double cb, T* x, T* r, Coef<T> cf, double* part_bb, double* part_rr0, PcgState* ps)
{
    hipLaunchKernelGGL((k_pcg_init_guess_l<T>), dim3(pcg_xr_blocks(L)), dim3(256), 0, st, g, L, cnt, b, guess, guess2, ca, cb, x, r, cf, part_bb,
                       part_rr0, ps);
}
template <typename T>
void launch_pcg_sq(hipStream_t st, LBox L, const uint8_t* cnt, const T* r, const T* s_in, T* s_out, T* q, Coef<T> cf,
                   const double* part_rr, const double* part_rz_new, const double* part_rz_old, double* part_pq, PcgState* ps, int first,
                   double tol, int n_rz, int zmode, int sparse, int n_prev)
{
    const int nx = pcg_xr_blocks(L);
    hipLaunchKernelGGL((k_pcg_sq_l<T, false>), dim3(pcg_sq_blocks(L)), dim3(256), 0, st, L, cnt, r, s_in, s_out, q, cf, part_rr,
                       part_rz_new, part_rz_old, part_pq, n_prev < 0 ? nx : n_prev, ps, first, tol, n_rz < 0 ? nx : n_rz, zmode, sparse, (const int*)nullptr, 0);
}
// decomposed run: g_* = single all-reduced scalars; cnt carries the ring bit (k_cnt_pcg); writes pcg_sq_blocks(L) partials of s'.q
template <typename T>
void launch_pcg_sq_dist(hipStream_t st, LBox L, const uint8_t* cnt, const T* r, const T* s_in, T* s_out, T* q, Coef<T> cf, const double* g_rr,
                        const double* g_rz_new, const double* g_rz_old, double* part_pq, PcgState* ps, int first, double tol, int zmode)
{
    hipLaunchKernelGGL((k_pcg_sq_l<T, true>), dim3(pcg_sq_blocks(L)), dim3(256), 0, st, L, cnt, r, s_in, s_out, q, cf, g_rr, g_rz_new, g_rz_old,
                       part_pq, 1, ps, first, tol, 1, zmode, 0, (const int*)nullptr, 0);
}
// ... and x += alpha s, r -= alpha q on the owned cells with alpha = g_rz / g_pq; writes pcg_xr_blocks(L) partials of |r|^2 (and r.invdiag r)
template <typename T>
void launch_pcg_xr_dist(hipStream_t st, LBox L, const uint8_t* cnt, T* x, T* r, const T* s, const T* q, Coef<T> cf, const double* g_rz_cur,
                        const double* g_pq, double* part_rr, double* part_rz_next, PcgState* ps)
{
    hipLaunchKernelGGL((k_pcg_xr_l<T, false>), dim3(pcg_xr_blocks(L)), dim3(256), 0, st, (long)(L.cells() / 2), cnt, x, r, s, q, cf, g_rz_cur, 1,
                       g_pq, 1, part_rr, part_rz_next, ps, 1);
}
void launch_sum2(hipStream_t st, const double* a, int na, const double* b, int nb, double* out_a, double* out_b)
{
    hipLaunchKernelGGL(k_sum2, dim3(1), dim3(256), 0, st, a, na, b, nb, out_a, out_b);
}
template <typename T>
void launch_pcg_xr(hipStream_t st, LBox L, const uint8_t* cnt, T* x, T* r, const T* s, const T* q, Coef<T> cf, const double* part_rz_cur,
                   const double* part_pq, double* part_rr, double* part_rz_next, PcgState* ps, int n_rz, int sparse)
{
    if (sparse)
        hipLaunchKernelGGL((k_pcg_xr_l<T, true>), dim3(pcg_xr_blocks(L)), dim3(256), 0, st, (long)(L.cells() / 2), cnt, x, r, s, q, cf, part_rz_cur,
                           n_rz < 0 ? pcg_xr_blocks(L) : n_rz, part_pq, pcg_sq_blocks(L), part_rr, part_rz_next, ps, 0);
    else
        hipLaunchKernelGGL((k_pcg_xr_l<T, false>), dim3(pcg_xr_blocks(L)), dim3(256), 0, st, (long)(L.cells() / 2), cnt, x, r, s, q, cf, part_rz_cur,
                           n_rz < 0 ? pcg_xr_blocks(L) : n_rz, part_pq, pcg_sq_blocks(L), part_rr, part_rz_next, ps, 0);
}

int sq_tile_count(const LBox& L) { return sq_tiles(L); }
// (every block re-sums the previous launch's partials: 2048 blocks and more measured slower than 1024 with ~5 tiles each)
int pcg_list_blocks(int nlist) { return nlist < 1 ? 1 : (nlist < SQ_MAX_BLOCKS ? nlist : SQ_MAX_BLOCKS); }
void launch_sq_tile_flags(hipStream_t st, LBox L, const uint8_t* cnt, uint8_t* flags)
{
    hipLaunchKernelGGL(k_sq_tile_flags, dim3(sq_tiles(L)), dim3(256), 0, st, L, cnt, flags);
}
void launch_compact_flags(hipStream_t st, const uint8_t* flags, int n, int* list, int* count)
{
    hipLaunchKernelGGL(k_compact_flags, dim3(1), dim3(1024), 0, st, flags, n, list, count);
}
template <typename T>
void launch_pcg_sq_list(hipStream_t st, LBox L, const uint8_t* cnt, const T* r, const T* s_in, T* s_out, T* q, Coef<T> cf,
                        const double* part_rr, int n_prev, const double* part_rz_new, const double* part_rz_old, double* part_pq, PcgState* ps,
                        int first, double tol, int n_rz, int zmode, const int* tlist, int nlist)
{
    hipLaunchKernelGGL((k_pcg_sq_l<T, false>), dim3(pcg_list_blocks(nlist)), dim3(256), 0, st, L, cnt, r, s_in, s_out, q, cf, part_rr,
                       part_rz_new, part_rz_old, part_pq, n_prev, ps, first, tol, n_rz, zmode, 0, tlist, nlist);
}
template <typename T>
void launch_pcg_xr_list(hipStream_t st, LBox L, const uint8_t* cnt, T* x, T* r, const T* s, const T* q, Coef<T> cf, const double* part_rz_cur,
                        int n_rz, const double* part_pq, double* part_rr, double* part_rz_next, PcgState* ps, const int* tlist, int nlist)
{
    const int nb = pcg_list_blocks(nlist);
    hipLaunchKernelGGL((k_pcg_xr_t<T>), dim3(nb), dim3(256), 0, st, L, cnt, x, r, s, q, cf, part_rz_cur, n_rz, part_pq, nb, part_rr,
                       part_rz_next, ps, tlist, nlist);
}
int pcg_row_count(const LBox& L) { return L.nx * L.ny * ((L.nz + 31) / 32); }
int pcg_rows_blocks(int nrows) { const int b = (nrows + 7) / 8; return b < 1 ? 1 : (b < SQ_MAX_BLOCKS ? b : SQ_MAX_BLOCKS); }
// list[0 .. *count) = the z rows of 32 cells that hold an unknown, ascending; flags / pos: pcg_row_count(L) ints each
void launch_row_list(hipStream_t st, LBox L, const uint8_t* cnt, int* flags, int* pos, int* list, int* block_sums, int* count)
{
    const int n = pcg_row_count(L);
    hipLaunchKernelGGL(k_row_flags, dim3((unsigned)((n + 7) / 8)), dim3(256), 0, st, L, cnt, n, flags);
    launch_exclusive_scan(st, flags, pos, (long)n, block_sums, count);
    hipLaunchKernelGGL(k_row_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, (const int*)flags, (const int*)pos, list);
}
template <typename T>
void launch_pcg_xr_rows(hipStream_t st, LBox L, const uint8_t* cnt, T* x, T* r, const T* s, const T* q, Coef<T> cf, const double* part_rz_cur,
                        int n_rz, const double* part_pq, int n_pq, double* part_rr, double* part_rz_next, PcgState* ps, const int* rlist, int nrows)
{
    hipLaunchKernelGGL((k_pcg_xr_rows<T>), dim3(pcg_rows_blocks(nrows)), dim3(256), 0, st, L, cnt, x, r, s, q, cf, part_rz_cur, n_rz, part_pq, n_pq,
                       part_rr, part_rz_next, ps, rlist, nrows);
}
template <typename T>
void launch_store_pressure(hipStream_t st, Grid g, LBox L, const uint8_t* cnt, const T* x, double* pressure, double* keep, const PcgState* ps)
{
    hipLaunchKernelGGL((k_store_pressure_l<T>), dim3((unsigned)((L.cells() + 255) / 256)), dim3(256), 0, st, g, L, cnt, x, pressure, keep, ps);
}

// ================================================================================================
// q = A s on the GLOBAL dense layout (fluid_stencil_apply: operator parity + dense micro-benchmark)
// Tile = 4 x 4 x 64 cells, one 64-lane wave per 512-byte row, halo staged in LDS.
// ================================================================================================
constexpr int GX = 4, GY = 4, GZ = 64;
constexpr int GLY = GY + 2, GLZ = GZ + 2;
constexpr int G_MAX_BLOCKS = 2048;

__device__ __forceinline__ bool active(uint8_t f) { return (f & F_FLUID) && (f >> F_CNT_SHIFT); }

struct Tiles {
    int ntx, nty, ntz;
    __host__ __device__ int count() const { return ntx * nty * ntz; }
};
static inline Tiles make_tiles(const Box& b)
{
    Tiles t;
    t.ntx = (b.nx() + GX - 1) / GX;
    t.nty = (b.ny() + GY - 1) / GY;
    t.ntz = (b.nz() + GZ - 1) / GZ;
    return t;
}

template <typename T>
__global__ __launch_bounds__(256) void k_stencil_g(Grid g, Box box, Tiles tl, const uint8_t* __restrict__ flags, const T* __restrict__ s_in,
                                                   T* __restrict__ q, Coef<T> cf)
{
    __shared__ T sT[(GX + 2) * GLY * GLZ];
    __shared__ T sdiag[8], sinv[8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    load_coef(sdiag, sinv, cf);
    const int N = g.N;
    const long sxl = (long)N * N;
    const int ntiles = tl.count();
    for (int tile = xcd_remap(blockIdx.x, gridDim.x); tile < ntiles; tile += gridDim.x) {
        const int tz = tile % tl.ntz, ty = (tile / tl.ntz) % tl.nty, tx = tile / (tl.ntz * tl.nty);
        const int x0 = box.x0 + tx * GX, y0 = box.y0 + ty * GY, z0 = box.z0 + tz * GZ;
        __syncthreads();  // LDS reuse across tiles
        T sc[4];
        uint8_t fc[4];
        unsigned inb = 0;
        // interior rows: wave wv owns rows wv, wv+4, wv+8, wv+12 (row = lx*GY + ly)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = wv + 4 * k, lx = row / GY, ly = row % GY;
            const int gx = x0 + lx, gy = y0 + ly, gz = z0 + lane;
            const bool in = gx <= box.x1 && gy <= box.y1 && gz <= box.z1;
            T val = 0;
            uint8_t f = 0;
            if (in) {
                const size_t c = (size_t)gx * sxl + (size_t)gy * N + gz;
                f = flags[c];
                if (active(f)) val = s_in[c];
                inb |= 1u << k;
            }
            sc[k] = val;
            fc[k] = f;
            sT[((lx + 1) * GLY + (ly + 1)) * GLZ + lane + 1] = val;
        }
        // x/y face halo rows: 16 rows, 4 per wave
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int h = wv * 4 + k, face = h >> 2, j = h & 3;
            const int lx = face == 0 ? -1 : (face == 1 ? GX : j);
            const int ly = face == 2 ? -1 : (face == 3 ? GY : j);
            const int gx = x0 + lx, gy = y0 + ly, gz = z0 + lane;
            T val = 0;
            if (gx >= 0 && gx < N && gy >= 0 && gy < N && gz < N) {
                const size_t c = (size_t)gx * sxl + (size_t)gy * N + gz;
                if (active(flags[c])) val = s_in[c];
            }
            sT[((lx + 1) * GLY + (ly + 1)) * GLZ + lane + 1] = val;
        }
        // z halo of the 16 interior rows: 32 cells
        if (tid < 32) {
            const int row = tid >> 1, side = tid & 1, lx = row / GY, ly = row % GY;
            const int gx = x0 + lx, gy = y0 + ly, gz = side ? z0 + GZ : z0 - 1;
            T val = 0;
            if (gx < N && gy < N && gz >= 0 && gz < N) {
                const size_t c = (size_t)gx * sxl + (size_t)gy * N + gz;
                if (active(flags[c])) val = s_in[c];
            }
            sT[((lx + 1) * GLY + (ly + 1)) * GLZ + (side ? GZ + 1 : 0)] = val;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (inb & (1u << k)) {
                const int row = wv + 4 * k, lx = row / GY, ly = row % GY;
                const size_t c = (size_t)(x0 + lx) * sxl + (size_t)(y0 + ly) * N + (z0 + lane);
                T qv = 0;
                if (active(fc[k])) {
                    const int o = ((lx + 1) * GLY + (ly + 1)) * GLZ + lane + 1;
                    const T nb = sT[o - GLY * GLZ] + sT[o + GLY * GLZ] + sT[o - GLZ] + sT[o + GLZ] + sT[o - 1] + sT[o + 1];
                    qv = sdiag[fc[k] >> F_CNT_SHIFT] * sc[k] + cf.off * nb;
                }
                q[c] = qv;
            }
        }
    }
}

// ---- dense sweep, x-marching ("2.5D") ------------------------------------------------------------
// One block = footprint MY x MZ in (y,z), marched through a chunk of x planes.  The x
// neighbours of a cell stay in the thread's registers (s at x-1, x, x+1), the y/z neighbours
// come from ONE double-buffered LDS plane, so each s value is fetched once per block (+ the
// 1-cell y/z rim, served by L2 when neighbouring footprints share an XCD) and there is one
// barrier per plane.  Loads run D planes ahead of use to keep ~64 KB per CU in flight.
constexpr int MZ = 64;

template <typename T, int MY, int MD>
__global__ __launch_bounds__(MY * 64) void k_stencil_march(Grid g, int cxlen, int nty, int ntz, const uint8_t* __restrict__ flags,
                                                       const T* __restrict__ s, T* __restrict__ q, Coef<T> cf)
{
    __shared__ T pl[2][MY + 2][MZ + 2];
    __shared__ T sdiag[8], sinv[8];
    const int tid = threadIdx.x, wy = tid >> 6, lz = tid & 63;
    load_coef(sdiag, sinv, cf);
    const int N = g.N;
    const long sx = (long)N * N;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int tz = vb % ntz, ty = (vb / ntz) % nty, cx = vb / (ntz * nty);
    const int xa = cx * cxlen, xe = xa + cxlen < N ? xa + cxlen : N;
    const int y = ty * MY + wy, z = tz * MZ + lz;
    const bool cv = y < N && z < N;
    const long col = (long)y * N + z;
    // rim duty: threads 0..127 the two y rows, 128..143 the 2 x MY z cells
    int hy = -1, hz = -1, hly = 0, hlz = 0;
    if (tid < 128) {
        const int r = tid >> 6;
        hy = r ? ty * MY + MY : ty * MY - 1;
        hz = tz * MZ + (tid & 63);
        hly = r ? MY + 1 : 0;
        hlz = (tid & 63) + 1;
    } else if (tid < 128 + 2 * MY) {
        const int k = tid - 128, r = k >> 1, side = k & 1;
        hy = ty * MY + r;
        hz = side ? tz * MZ + MZ : tz * MZ - 1;
        hly = r + 1;
        hlz = side ? MZ + 1 : 0;
    }
    const bool hduty = tid < 128 + 2 * MY;
    const bool hv = hduty && hy >= 0 && hy < N && hz >= 0 && hz < N;
    const long hcol = (long)hy * N + hz;
    auto ld = [&](int x, long colx, bool ok, T& val, uint8_t& f) {
        if (ok && x >= 0 && x < N && x <= xe) {   // plane xe is the last one this chunk reads (x+ neighbour of its last plane)
            const long c = (long)x * sx + colx;
            f = flags[c];
            val = s[c];
        } else {
            f = 0;
            val = 0;
        }
    };
    auto mk = [](T v, uint8_t f) { return ((f & F_FLUID) && (f >> F_CNT_SHIFT)) ? v : (T)0; };
    T vm1, v0, hv0;
    uint8_t fm1, f0, hf0;
    ld(xa - 1, col, cv, vm1, fm1);
    ld(xa, col, cv, v0, f0);
    ld(xa, hcol, hv, hv0, hf0);
    T qv[MD], hq[MD];
    uint8_t qf[MD], hqf[MD];
#pragma unroll
    for (int d = 0; d < MD; ++d) {
        ld(xa + 1 + d, col, cv, qv[d], qf[d]);
        ld(xa + 1 + d, hcol, hv, hq[d], hqf[d]);
    }
    T sm1 = mk(vm1, fm1), s0 = mk(v0, f0), h0 = mk(hv0, hf0);
    __syncthreads();  // coef tables
    for (int xb = xa; xb < xe; xb += MD) {
#pragma unroll
        for (int d = 0; d < MD; ++d) {
            const int x = xb + d;
            if (x < xe) {  // block-uniform
                const int buf = x & 1;
                pl[buf][wy + 1][lz + 1] = s0;
                if (hduty) pl[buf][hly][hlz] = h0;
                __syncthreads();
                const T sp1 = mk(qv[d], qf[d]);
                if (cv) {
                    T out = 0;
                    if ((f0 & F_FLUID) && (f0 >> F_CNT_SHIFT)) {
                        const T nb = sm1 + sp1 + pl[buf][wy][lz + 1] + pl[buf][wy + 2][lz + 1] + pl[buf][wy + 1][lz] + pl[buf][wy + 1][lz + 2];
                        out = sdiag[f0 >> F_CNT_SHIFT] * s0 + cf.off * nb;
                    }
                    __builtin_nontemporal_store(out, &q[(long)x * sx + col]);  // streamed once: keep s, not q, in cache
                }
                sm1 = s0;
                s0 = sp1;
                f0 = qf[d];
                h0 = mk(hq[d], hqf[d]);
                ld(x + 1 + MD, col, cv, qv[d], qf[d]);
                ld(x + 1 + MD, hcol, hv, hq[d], hqf[d]);
            }
        }
    }
}

template <typename T, int MY, int MD>
static void march_launch(hipStream_t st, Grid g, int cxlen, const uint8_t* flags, const T* s, T* q, Coef<T> cf)
{
    const int nty = (g.N + MY - 1) / MY, ntz = (g.N + MZ - 1) / MZ, ncx = (g.N + cxlen - 1) / cxlen;
    hipLaunchKernelGGL((k_stencil_march<T, MY, MD>), dim3(nty * ntz * ncx), dim3(MY * 64), 0, st, g, cxlen, nty, ntz, flags, s, q, cf);
}

// ---- dense sweep, x-marching, 16 bytes per lane ----------------------------------------------------------------------
// The same march with V = 16 / sizeof(T) cells per lane along z (float: 4, double: 2): a wave covers 64 V cells of a z
// row with ONE 16-byte load per lane and plane (1 KB per wave instead of 256 B for float), the LDS plane is read and
// written as 16-byte vectors (y neighbours), the z neighbours inside a lane are registers and across lanes two wave
// shuffles; lanes 0 / 63 take the cell before / after the wave's z range from a scalar rim load.  Per cell: 1/V LDS
// write, 2/V LDS reads, 2/V shuffles instead of 1 write + 4 reads — the scalar float kernel was bound by exactly that
// (54 % of the HBM peak against 75-80 % for double).  Needs N % V == 0 (rows 16-byte aligned); same term order as the scalar
// kernel (x-, x+, y-, y+, z-, z+), so the results are bit-identical.

template <typename T, int MY, int MD>
__global__ __launch_bounds__(MY * 64) void k_stencil_vec(Grid g, int cxlen, int nty, int ntz, const uint8_t* __restrict__ flags,
                                                          const T* __restrict__ s, T* __restrict__ q, Coef<T> cf)
{
    constexpr int V = 16 / (int)sizeof(T), MZV = 64 * V;
    typedef typename VecT<T, V>::type vec;
    typedef typename FlagT<V>::type fvec;
    __shared__ __attribute__((aligned(16))) T pl[2][MY + 2][MZV];
    __shared__ T sdiag[8], sinv[8];
    const int tid = threadIdx.x, wy = tid >> 6, lane = tid & 63;
    load_coef(sdiag, sinv, cf);
    const int N = g.N;
    const long sx = (long)N * N;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int tz = vb % ntz, ty = (vb / ntz) % nty, cx = vb / (ntz * nty);
    const int xa = cx * cxlen, xe = xa + cxlen < N ? xa + cxlen : N;
    const int y = ty * MY + wy, z0 = tz * MZV + lane * V;
    const bool cv = y < N && z0 < N;                      // N % V == 0: a lane's cells are all inside or all outside
    const long col = (long)y * N + z0;
    // rim rows: waves 0 and 1 also carry the row below / above the footprint
    const int hy = wy == 0 ? ty * MY - 1 : ty * MY + MY;
    const bool hduty = wy < 2;
    const bool hv = hduty && hy >= 0 && hy < N && z0 < N;
    const long hcol = (long)hy * N + z0;
    // z rim: lane 0 the cell before the wave's range, lane 63 the cell after it
    const int rz = lane == 0 ? tz * MZV - 1 : tz * MZV + MZV;
    const bool rduty = lane == 0 || lane == 63;
    const bool rv = rduty && y < N && rz >= 0 && rz < N;
    const long rcol = (long)y * N + rz;
    auto act = [](uint8_t f) { return (f & F_FLUID) && (f >> F_CNT_SHIFT); };
    auto ldv = [&](int x, long colx, bool ok, vec& val, fvec& f) {
        if (ok && x >= 0 && x < N && x <= xe) {   // plane xe is the last one this chunk reads
            const long c = (long)x * sx + colx;
            f = *reinterpret_cast<const fvec*>(flags + c);
            val = *reinterpret_cast<const vec*>(s + c);
        } else {
            f = 0;
            val = (vec)(T)0;
        }
    };
    auto lds1 = [&](int x, long colx, bool ok, T& val, uint8_t& f) {
        if (ok && x >= 0 && x < N && x <= xe) {
            const long c = (long)x * sx + colx;
            f = flags[c];
            val = s[c];
        } else {
            f = 0;
            val = 0;
        }
    };
    auto mkv = [&](vec v, fvec f) {
        vec o;
#pragma unroll
        for (int c = 0; c < V; ++c) o[c] = act((uint8_t)(f >> (8 * c))) ? v[c] : (T)0;
        return o;
    };
    vec vm1, v0, hv0;
    fvec fm1, f0, hf0;
    T r0;
    uint8_t rf0;
    ldv(xa - 1, col, cv, vm1, fm1);
    ldv(xa, col, cv, v0, f0);
    ldv(xa, hcol, hv, hv0, hf0);
    lds1(xa, rcol, rv, r0, rf0);
    vec qv[MD], hq[MD];
    fvec qf[MD], hqf[MD];
    T rq[MD];
    uint8_t rqf[MD];
#pragma unroll
    for (int d = 0; d < MD; ++d) {
        ldv(xa + 1 + d, col, cv, qv[d], qf[d]);
        ldv(xa + 1 + d, hcol, hv, hq[d], hqf[d]);
        lds1(xa + 1 + d, rcol, rv, rq[d], rqf[d]);
    }
    vec sm1 = mkv(vm1, fm1), s0 = mkv(v0, f0), h0 = mkv(hv0, hf0);
    T rim = act(rf0) ? r0 : (T)0;
    __syncthreads();  // coef tables
    for (int xb = xa; xb < xe; xb += MD) {
#pragma unroll
        for (int d = 0; d < MD; ++d) {
            const int x = xb + d;
            if (x < xe) {  // block-uniform
                const int buf = x & 1;
                *reinterpret_cast<vec*>(&pl[buf][wy + 1][lane * V]) = s0;
                if (hduty) *reinterpret_cast<vec*>(&pl[buf][wy == 0 ? 0 : MY + 1][lane * V]) = h0;
                __syncthreads();
                const vec sp1 = mkv(qv[d], qf[d]);
                if (cv) {
                    const vec up = *reinterpret_cast<const vec*>(&pl[buf][wy][lane * V]);
                    const vec dn = *reinterpret_cast<const vec*>(&pl[buf][wy + 2][lane * V]);
                    T left = __shfl_up(s0[V - 1], 1, 64), right = __shfl_down(s0[0], 1, 64);
                    if (lane == 0) left = rim;
                    if (lane == 63) right = rim;
                    vec out;
#pragma unroll
                    for (int c = 0; c < V; ++c) {
                        const uint8_t fc = (uint8_t)(f0 >> (8 * c));
                        const T zl = c ? s0[c > 0 ? c - 1 : 0] : left, zr = c < V - 1 ? s0[c < V - 1 ? c + 1 : 0] : right;
                        const T nb = sm1[c] + sp1[c] + up[c] + dn[c] + zl + zr;
                        out[c] = act(fc) ? sdiag[fc >> F_CNT_SHIFT] * s0[c] + cf.off * nb : (T)0;
                    }
                    __builtin_nontemporal_store(out, reinterpret_cast<vec*>(&q[(long)x * sx + col]));  // streamed once: keep s, not q, in cache
                }
                sm1 = s0;
                s0 = sp1;
                f0 = qf[d];
                h0 = mkv(hq[d], hqf[d]);
                rim = act(rqf[d]) ? rq[d] : (T)0;
                ldv(x + 1 + MD, col, cv, qv[d], qf[d]);
                ldv(x + 1 + MD, hcol, hv, hq[d], hqf[d]);
                lds1(x + 1 + MD, rcol, rv, rq[d], rqf[d]);
            }
        }
    }
}

template <typename T, int MY, int MD>
static void vec_launch(hipStream_t st, Grid g, int cxlen, const uint8_t* flags, const T* s, T* q, Coef<T> cf)
{
    constexpr int V = 16 / (int)sizeof(T), MZV = 64 * V;
    const int nty = (g.N + MY - 1) / MY, ntz = (g.N + MZV - 1) / MZV, ncx = (g.N + cxlen - 1) / cxlen;
    hipLaunchKernelGGL((k_stencil_vec<T, MY, MD>), dim3(nty * ntz * ncx), dim3(MY * 64), 0, st, g, cxlen, nty, ntz, flags, s, q, cf);
}

// ---- dense sweep, x-marching, independent waves ("rows") ---------------------------------------------------------------
// Measured from HBM (launches rotating over > 1 GiB of operands, fluid_stencil_apply_hbm) the LDS-plane marches above reach
// 53-59 % of the peak where a plain copy of the same arrays with the flag stream reaches 72-75 %: a block's waves are
// coupled by one barrier per plane (the slowest load of 16 waves sets the pace, and HBM latency varies far more than the
// Infinity Cache's), and every block re-reads a rim that its neighbours own.  Here a WAVE is the unit: it owns R
// consecutive z rows (V = 16 / sizeof(T) cells per lane, 64 V cells per row segment) and marches them through a chunk
// of x planes; per plane it loads rows y0-1 .. y0+R itself, 16 bytes per lane each — the two outer rows are some other
// wave's own rows, fetched at about the same time, so they come from L2 — keeps the x neighbours in registers and takes
// the z neighbours inside a lane from registers, across lanes from two shuffles.  No LDS plane, no barrier in the loop:
// every wave streams like a copy loop with MD planes in flight.  Same term order as the other forms (bit-identical).
template <typename T, int R, int MD>
__global__ __launch_bounds__(256) void k_stencil_rows(Grid g, int cxlen, int nyg, int ntz, const uint8_t* __restrict__ flags,
                                                      const T* __restrict__ s, T* __restrict__ q, Coef<T> cf)
{
    constexpr int V = 16 / (int)sizeof(T), MZV = 64 * V;
    typedef typename VecT<T, V>::type vec;
    typedef typename FlagT<V>::type fvec;
    __shared__ T sdiag[8], sinv[8];
    load_coef(sdiag, sinv, cf);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int N = g.N;
    const long sx = (long)N * N;
    // wave id: blocks dealt XCD-contiguously, the 4 waves of a block are neighbours in (z, y)
    const int w = xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    const int tz = w % ntz, yg = (w / ntz) % nyg, cx = w / (ntz * nyg);
    const int xa = cx * cxlen, xe = xa + cxlen < N ? xa + cxlen : N;
    if (xa >= N) return;
    const int y0 = yg * R, z0 = tz * MZV + lane * V;
    const bool zin = z0 < N;   // N % V == 0: a lane's cells are all inside or all outside
    // z rim of the wave's segment: lane 0 the cell before it, lane 63 the cell after it
    const int rz = lane == 0 ? tz * MZV - 1 : tz * MZV + MZV;
    const bool rduty = (lane == 0 || lane == 63) && rz >= 0 && rz < N;
    auto act = [](uint8_t f) { return (f & F_FLUID) && (f >> F_CNT_SHIFT); };
    auto mkv = [&](vec v, fvec f) {
        vec o;
#pragma unroll
        for (int c = 0; c < V; ++c) o[c] = act((uint8_t)(f >> (8 * c))) ? v[c] : (T)0;
        return o;
    };
    struct Plane {
        vec v[R + 2];     // rows y0-1 .. y0+R
        fvec f[R + 2];
        T rv[R];          // rim cells of the own rows
        uint8_t rf[R];
    };
    auto load = [&](int x, Plane& P) {
        const bool xok = x >= 0 && x < N && x <= xe;
#pragma unroll
        for (int j = 0; j < R + 2; ++j) {
            const int y = y0 - 1 + j;
            if (xok && zin && y >= 0 && y < N) {
                const long c = (long)x * sx + (long)y * N + z0;
                P.f[j] = *reinterpret_cast<const fvec*>(flags + c);
                P.v[j] = *reinterpret_cast<const vec*>(s + c);
            } else {
                P.f[j] = 0;
                P.v[j] = (vec)(T)0;
            }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int y = y0 + j;
            if (xok && rduty && y < N) {
                const long c = (long)x * sx + (long)y * N + rz;
                P.rf[j] = flags[c];
                P.rv[j] = s[c];
            } else {
                P.rf[j] = 0;
                P.rv[j] = 0;
            }
        }
    };
    Plane A[MD];
    vec sm1[R], cen[R + 2];
    fvec f0[R];
    T rim[R];
    {
        Plane P;
        load(xa - 1, P);
#pragma unroll
        for (int j = 0; j < R; ++j) sm1[j] = mkv(P.v[j + 1], P.f[j + 1]);
        load(xa, P);
#pragma unroll
        for (int j = 0; j < R + 2; ++j) cen[j] = mkv(P.v[j], P.f[j]);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            f0[j] = P.f[j + 1];
            rim[j] = act(P.rf[j]) ? P.rv[j] : (T)0;
        }
    }
#pragma unroll
    for (int d = 0; d < MD; ++d) load(xa + 1 + d, A[d]);
    for (int xb = xa; xb < xe; xb += MD) {
#pragma unroll
        for (int d = 0; d < MD; ++d) {
            const int x = xb + d;
            if (x < xe) {  // wave-uniform
                vec nxt[R + 2];
#pragma unroll
                for (int j = 0; j < R + 2; ++j) nxt[j] = mkv(A[d].v[j], A[d].f[j]);
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const vec s0 = cen[j + 1], up = cen[j], dn = cen[j + 2], sp1 = nxt[j + 1];
                    T left = __shfl_up(s0[V - 1], 1, 64), right = __shfl_down(s0[0], 1, 64);
                    if (lane == 0) left = rim[j];
                    if (lane == 63) right = rim[j];
                    vec out;
#pragma unroll
                    for (int c = 0; c < V; ++c) {
                        const uint8_t fc = (uint8_t)(f0[j] >> (8 * c));
                        const T zl = c ? s0[c > 0 ? c - 1 : 0] : left, zr = c < V - 1 ? s0[c < V - 1 ? c + 1 : 0] : right;
                        const T nb = sm1[j][c] + sp1[c] + up[c] + dn[c] + zl + zr;
                        out[c] = act(fc) ? sdiag[fc >> F_CNT_SHIFT] * s0[c] + cf.off * nb : (T)0;
                    }
                    if (zin && y0 + j < N) __builtin_nontemporal_store(out, reinterpret_cast<vec*>(&q[(long)x * sx + (long)(y0 + j) * N + z0]));
                    sm1[j] = s0;
                    f0[j] = A[d].f[j + 1];
                    rim[j] = act(A[d].rf[j]) ? A[d].rv[j] : (T)0;
                }
#pragma unroll
                for (int j = 0; j < R + 2; ++j) cen[j] = nxt[j];
                load(x + 1 + MD, A[d]);
            }
        }
    }
}

template <typename T, int R, int MD>
static void rows_launch(hipStream_t st, Grid g, int cxlen, const uint8_t* flags, const T* s, T* q, Coef<T> cf)
{
    constexpr int V = 16 / (int)sizeof(T), MZV = 64 * V;
    const int nyg = (g.N + R - 1) / R, ntz = (g.N + MZV - 1) / MZV, ncx = (g.N + cxlen - 1) / cxlen;
    const long waves = (long)nyg * ntz * ncx;
    hipLaunchKernelGGL((k_stencil_rows<T, R, MD>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, g, cxlen, nyg, ntz, flags, s, q, cf);
}

// ---- dense sweep, x-marching, lean ("lean") -----------------------------------------------------------------------------
// The marches above spend ~70 instructions per cell, half of them scalar branches (every guarded load and every
// `active ? .. : 0` became a branch): from HBM they stop at 53-59 % of the peak where a copy of the same bytes reaches
// 72-75 % (tools/sweep.py 256 hbm).  Same data flow as k_stencil_vec — one row of 64 V cells per wave and plane, x
// neighbours in registers, y neighbours through one double-buffered LDS plane, one barrier per plane — rebuilt for a
// short instruction stream:
//   * loads are unconditional, from clamped addresses; what lies outside the grid is masked, planes outside the
//     chunk's reach are skipped by ONE wave-uniform branch;
//   * the "unknown" predicate of the V flag bytes of a lane is formed on the packed word (SWAR) and expanded to one
//     all-ones / zero word per cell (v_bfe_i32); values are masked by AND, results too: no per-cell branch;
//   * the rim rows (y0 - 1, y0 + MY) belong to two EXTRA waves that only load, mask and publish them, so no wave of the
//     block carries two rows to the barrier;
//   * z neighbours across lanes by DPP wave shifts (zero fill at the wave's ends = the grid's edge when a row is one
//     wave's width; else lanes 0 / 63 load the two rim cells).
// Same term order as the other forms: bit-identical results.
template <typename T>
__device__ __forceinline__ T dpp_wave_shr1(T v)   // lane i <- lane i-1, lane 0 <- 0
{
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
    } else {
        const long long b = __builtin_bit_cast(long long, v);
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x138, 0xf, 0xf, true);
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x138, 0xf, 0xf, true);
        return __builtin_bit_cast(T, ((long long)hi << 32) | lo);
    }
}
template <typename T>
__device__ __forceinline__ T dpp_wave_shl1(T v)   // lane i <- lane i+1, lane 63 <- 0
{
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
    } else {
        const long long b = __builtin_bit_cast(long long, v);
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x130, 0xf, 0xf, true);
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x130, 0xf, 0xf, true);
        return __builtin_bit_cast(T, ((long long)hi << 32) | lo);
    }
}

// AXIS = the axis of the march: 0 = x (a wave steps from plane to plane: successive loads of a wave lie N^2 elements apart),
// 1 = y (a wave owns one z row segment of ONE x plane and walks it row by row: its loads are one contiguous stream of N-element
// rows; the waves of a block are MY neighbouring x planes and exchange the x neighbours through the LDS plane, the y neighbours
// stay in registers).  The sum keeps the order x-, x+, y-, y+, z-, z+ either way.
// MODE 1 (probe): q = masked s with the same loads, stores and march, no LDS, barrier or arithmetic.
template <typename T, int MY, int MD, int MODE = 0, int AXIS = 0, bool RIMS = false>
__global__ __launch_bounds__((MY + 2) * 64) void k_stencil_lean(Grid g, int cxlen, int nty, int ntz, const uint8_t* __restrict__ flags,
                                                                const T* __restrict__ s, T* __restrict__ q, Coef<T> cf)
{
    constexpr int V = 16 / (int)sizeof(T), MZV = 64 * V;
    typedef typename VecT<T, V>::type vec;
    typedef typename FlagT<V>::type fvec;
    __shared__ __attribute__((aligned(16))) T pl[2][MY + 2][MZV];
    __shared__ T sdiag[8], sinv[8];
    const int tid = threadIdx.x, wy = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    load_coef(sdiag, sinv, cf);
    const int N = g.N;
    const long sm = AXIS == 0 ? (long)N * N : (long)N;   // stride of the march axis (index m below)
    const long sr = AXIS == 0 ? (long)N : (long)N * N;   // stride of the row axis (index r: the waves of a block)
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int tz = vb % ntz, ty = (vb / ntz) % nty, cx = vb / (ntz * nty);
    const int xa = cx * cxlen, xe = xa + cxlen < N ? xa + cxlen : N;
    const int xlast = xe < N ? xe : N - 1;          // last march index anybody of this chunk reads
    const bool own = wy < MY;                       // waves MY, MY+1: the rim rows
    const int y = own ? ty * MY + wy : (wy == MY ? ty * MY - 1 : ty * MY + MY);
    const int lrow = own ? wy + 1 : (wy == MY ? 0 : MY + 1);
    const int z0 = tz * MZV + lane * V;
    const bool cv = y >= 0 && y < N && z0 < N;      // N % V == 0: a lane's cells are all inside or all outside
    const unsigned cvm = cv ? 0xFFFFFFFFu : 0u;
    const long col = (long)min(max(y, 0), N - 1) * sr + min(z0, N - V);
    // z rim (only when a row is wider than one wave): lane 0 the cell before the wave's range, lane 63 the cell after it
    const int rz = lane < 32 ? tz * MZV - 1 : tz * MZV + MZV;
    const bool rv = own && (lane == 0 || lane == 63) && y < N && rz >= 0 && rz < N;
    const long rcol = (long)min(max(y, 0), N - 1) * sr + min(max(rz, 0), N - 1);
    // per-cell masks (all ones / zero) of a packed flag word: unknown = fluid and at least one non-solid neighbour
    auto active_bits = [](unsigned w) { return (w >> 1) & ((((w >> 2) & 0x07070707u) + 0x7F7F7F7Fu) >> 7) & 0x01010101u; };
    auto mkv = [&](vec v, unsigned a) {
        vec o;
#pragma unroll
        for (int c = 0; c < V; ++c) o[c] = and_mask<T>(v[c], __builtin_amdgcn_sbfe((int)a, 8 * c, 1));
        return o;
    };
    // Loads are unconditional and their results are masked only where they are consumed: a select right behind a load (or a load
    // inside a branch) makes the compiler wait for it at once, and the MD planes in flight collapse to one.
    auto ldp = [&](int x, vec& v, unsigned& w) {   // one step of this wave's row: x is wave-uniform
        const long c = (long)min(max(x, 0), xlast) * sm + col;
        w = (unsigned)*reinterpret_cast<const fvec*>(flags + c);
        v = *reinterpret_cast<const vec*>(s + c);
    };
    auto okw = [&](int x, unsigned w) { return x >= 0 && x <= xlast ? w & cvm : 0u; };
    auto ldr = [&](int x, T& v, unsigned& f) {     // the rim cell of lanes 0 / 63 (every lane loads: no branch)
        v = 0;
        f = 0;
        if constexpr (RIMS) {
            const long c = (long)min(max(x, 0), xlast) * sm + rcol;
            f = flags[c];
            v = s[c];
        }
    };
    auto okr = [&](int x, unsigned f) { return rv && x >= 0 && x <= xlast ? f : 0u; };
    vec vm1, v0;
    unsigned wm1, w0, rf0;
    T r0;
    ldp(xa - 1, vm1, wm1);
    ldp(xa, v0, w0);
    ldr(xa, r0, rf0);
    vec qv[MD];
    unsigned qw[MD], rqf[MD];
    T rq[MD];
#pragma unroll
    for (int d = 0; d < MD; ++d) {
        ldp(xa + 1 + d, qv[d], qw[d]);
        ldr(xa + 1 + d, rq[d], rqf[d]);
    }
    w0 = okw(xa, w0);
    vec sm1 = mkv(vm1, active_bits(okw(xa - 1, wm1))), s0 = mkv(v0, active_bits(w0));
    T rim = and_mask<T>(r0, __builtin_amdgcn_sbfe((int)active_bits(okr(xa, rf0)), 0, 1));
    const T off = cf.off;
    __syncthreads();  // coef table
    // one step of the march; d = the slot of the ring that holds step x+1 and is refilled with step x+1+MD
    auto step = [&](int x, vec& nv, unsigned& nw, T& nr, unsigned& nrf) {
        const int buf = x & 1;
        const unsigned wn = okw(x + 1, nw);
        if (MODE == 1) {
            if (own && cv) __builtin_nontemporal_store(s0, reinterpret_cast<vec*>(&q[(long)x * sm + col]));
            s0 = mkv(nv, active_bits(wn));
            ldp(x + 1 + MD, nv, nw);
            return;
        }
        *reinterpret_cast<vec*>(&pl[buf][lrow][lane * V]) = s0;
        __syncthreads();
        const unsigned an = active_bits(wn);
        const vec sp1 = mkv(nv, an);
        if (own) {   // wave-uniform
            const vec up = *reinterpret_cast<const vec*>(&pl[buf][lrow - 1][lane * V]);
            const vec dn = *reinterpret_cast<const vec*>(&pl[buf][lrow + 1][lane * V]);
            T left = dpp_wave_shr1<T>(s0[V - 1]), right = dpp_wave_shl1<T>(s0[0]);
            if (RIMS) {
                left = lane == 0 ? rim : left;
                right = lane == 63 ? rim : right;
            }
            const unsigned a0 = active_bits(w0);
            vec out;
#pragma unroll
            for (int c = 0; c < V; ++c) {
                const T zl = c ? s0[c > 0 ? c - 1 : 0] : left, zr = c < V - 1 ? s0[c < V - 1 ? c + 1 : 0] : right;
                const T nb = AXIS == 0 ? sm1[c] + sp1[c] + up[c] + dn[c] + zl + zr : up[c] + dn[c] + sm1[c] + sp1[c] + zl + zr;
                const T r = sdiag[__builtin_amdgcn_ubfe(w0, 8 * c + F_CNT_SHIFT, 3)] * s0[c] + off * nb;
                out[c] = and_mask<T>(r, __builtin_amdgcn_sbfe((int)a0, 8 * c, 1));
            }
            if (cv) __builtin_nontemporal_store(out, reinterpret_cast<vec*>(&q[(long)x * sm + col]));  // streamed once: keep s, not q, in cache
        }
        sm1 = s0;
        s0 = sp1;
        w0 = wn;
        rim = and_mask<T>(nr, __builtin_amdgcn_sbfe((int)active_bits(okr(x + 1, nrf)), 0, 1));
        ldp(x + 1 + MD, nv, nw);
        ldr(x + 1 + MD, nr, nrf);
    };
    // whole rounds of the ring run unguarded: a guard inside the round is a path on which the newest load is the next one
    // consumed, and the compiler then drains the queue at every step
    int xb = xa;
    for (; xb + MD <= xe; xb += MD) {
#pragma unroll
        for (int d = 0; d < MD; ++d) step(xb + d, qv[d], qw[d], rq[d], rqf[d]);
    }
#pragma unroll
    for (int d = 0; d < MD; ++d)
        if (xb + d < xe) step(xb + d, qv[d], qw[d], rq[d], rqf[d]);   // block-uniform
}

template <typename T, int MY, int MD, int MODE = 0, int AXIS = 0>
static void lean_launch(hipStream_t st, Grid g, int cxlen, const uint8_t* flags, const T* s, T* q, Coef<T> cf)
{
    constexpr int V = 16 / (int)sizeof(T), MZV = 64 * V;
    const int nty = (g.N + MY - 1) / MY, ntz = (g.N + MZV - 1) / MZV, ncx = (g.N + cxlen - 1) / cxlen;
    if (ntz > 1)
        hipLaunchKernelGGL((k_stencil_lean<T, MY, MD, MODE, AXIS, true>), dim3(nty * ntz * ncx), dim3((MY + 2) * 64), 0, st, g, cxlen, nty, ntz, flags, s, q, cf);
    else
        hipLaunchKernelGGL((k_stencil_lean<T, MY, MD, MODE, AXIS, false>), dim3(nty * ntz * ncx), dim3((MY + 2) * 64), 0, st, g, cxlen, nty, ntz, flags, s, q, cf);
}

// ---- dense sweep with a linear front ("front") -------------------------------------------------------------------------
// No march: a wave computes ONE z row of ONE plane and loads the five rows it needs itself — centre, y-1, y+1 (its sibling
// waves' centres: L1), x-1, x+1 (the rows the same position of the neighbouring planes: L2 / Infinity Cache, fetched a moment ago
// or a moment later by the blocks of those planes).  Blocks are numbered in MEMORY order and NOT regrouped per XCD, so the chip
// works on a front of a few consecutive planes and HBM sees one linear read stream and one linear write stream — what the
// copy has and every march lacks (thousands of 4-16 KB pieces advancing at once).  Price: ~3x the L2 read traffic.
template <typename T, int MY>
__global__ __launch_bounds__(MY * 64) void k_stencil_front(Grid g, int nty, int ntz, const uint8_t* __restrict__ flags, const T* __restrict__ s,
                                                            T* __restrict__ q, Coef<T> cf)
{
    constexpr int V = 16 / (int)sizeof(T), MZV = 64 * V;
    typedef typename VecT<T, V>::type vec;
    typedef typename FlagT<V>::type fvec;
    __shared__ T sdiag[8], sinv[8];
    load_coef(sdiag, sinv, cf);
    const int N = g.N;
    const long sx = (long)N * N;
    const int lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const int tz = b % ntz, ty = (b / ntz) % nty, x = b / (ntz * nty);
    const int y = ty * MY + wy, z0 = tz * MZV + lane * V;
    const bool cv = y < N && z0 < N;
    const long c0 = (long)x * sx + (long)min(y, N - 1) * N + min(z0, N - V);
    auto active_bits = [](unsigned w) { return (w >> 1) & ((((w >> 2) & 0x07070707u) + 0x7F7F7F7Fu) >> 7) & 0x01010101u; };
    auto row = [&](long c, bool ok, vec& v, unsigned& a) {   // masked values of one row (ok is wave-uniform)
        if (ok) {
            const unsigned w = (unsigned)*reinterpret_cast<const fvec*>(flags + c);
            const vec t = *reinterpret_cast<const vec*>(s + c);
            a = cv ? active_bits(w) : 0u;
#pragma unroll
            for (int k = 0; k < V; ++k) v[k] = and_mask<T>(t[k], __builtin_amdgcn_sbfe((int)a, 8 * k, 1));
        } else {
            a = 0;
            v = (vec)(T)0;
        }
    };
    vec s0, xm, xp, ym, yp;
    unsigned a0, au;
    unsigned w0 = cv ? (unsigned)*reinterpret_cast<const fvec*>(flags + c0) : 0u;
    row(c0, true, s0, a0);
    row(c0 - sx, x > 0, xm, au);
    row(c0 + sx, x < N - 1, xp, au);
    row(c0 - N, y > 0 && y < N, ym, au);
    row(c0 + N, y < N - 1, yp, au);
    T left = dpp_wave_shr1<T>(s0[V - 1]), right = dpp_wave_shl1<T>(s0[0]);
    if (ntz > 1) {   // a row wider than one wave: lanes 0 / 63 fetch the cell before / after the wave's range
        const int rz = lane == 0 ? tz * MZV - 1 : tz * MZV + MZV;
        T rim = 0;
        if ((lane == 0 || lane == 63) && cv && rz >= 0 && rz < N) {
            const long c = (long)x * sx + (long)y * N + rz;
            const uint8_t f = flags[c];
            rim = ((f & F_FLUID) && (f >> F_CNT_SHIFT)) ? s[c] : (T)0;
        }
        left = lane == 0 ? rim : left;
        right = lane == 63 ? rim : right;
    }
    __syncthreads();   // coef table
    vec out;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const T zl = k ? s0[k > 0 ? k - 1 : 0] : left, zr = k < V - 1 ? s0[k < V - 1 ? k + 1 : 0] : right;
        const T nb = xm[k] + xp[k] + ym[k] + yp[k] + zl + zr;
        const T r = sdiag[__builtin_amdgcn_ubfe(w0, 8 * k + F_CNT_SHIFT, 3)] * s0[k] + cf.off * nb;
        out[k] = and_mask<T>(r, __builtin_amdgcn_sbfe((int)a0, 8 * k, 1));
    }
    if (cv) __builtin_nontemporal_store(out, reinterpret_cast<vec*>(&q[c0]));
}
template <typename T, int MY>
static void front_launch(hipStream_t st, Grid g, const uint8_t* flags, const T* s, T* q, Coef<T> cf)
{
    constexpr int V = 16 / (int)sizeof(T), MZV = 64 * V;
    const int nty = (g.N + MY - 1) / MY, ntz = (g.N + MZV - 1) / MZV;
    hipLaunchKernelGGL((k_stencil_front<T, MY>), dim3((unsigned)((long)nty * ntz * g.N)), dim3(MY * 64), 0, st, g, nty, ntz, flags, s, q, cf);
}

// ---- streaming probes (developer: FLUID_MARCH_VARIANT=20001 / 20002): what a plain copy of the same arrays reaches ----
// 20001: q = s, 16 bytes per lane, non-temporal stores; 20002: q = active(flag) ? s : 0 (adds the 1-byte-per-cell flag stream).
// They bound what the stencil sweep can reach from HBM on this part (tools/sweep.py ... hbm).
template <typename T, bool FLAGS>
__global__ __launch_bounds__(256) void k_stream_probe(long n16, const uint8_t* __restrict__ flags, const T* __restrict__ s, T* __restrict__ q)
{
    constexpr int V = 16 / (int)sizeof(T);
    typedef typename VecT<T, V>::type vec;
    typedef typename FlagT<V>::type fvec;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        vec v = reinterpret_cast<const vec*>(s)[i];
        if (FLAGS) {
            const fvec f = reinterpret_cast<const fvec*>(flags)[i];
#pragma unroll
            for (int c = 0; c < V; ++c) {
                const uint8_t fc = (uint8_t)(f >> (8 * c));
                v[c] = ((fc & F_FLUID) && (fc >> F_CNT_SHIFT)) ? v[c] : (T)0;
            }
        }
        __builtin_nontemporal_store(v, reinterpret_cast<vec*>(q) + i);
    }
}

// variant = MY*100 + MD of the one-cell-per-lane march, 10000 + ... of the 16-bytes-per-lane march, 30000 + R*100 + MD of the
// barrier-free rows form, 40000 / 60000 + MY*100 + MD of the lean march along x / y, 20001 / 20002 / 5xxxx / 7xxxx: copy probes;
// cxlen = planes per chunk.  0 = the measured best at 256^3 FROM HBM (tools/sweep.py 256 hbm ..., profiles/r03/stencil_sweep_hbm.txt):
// the lean march along x, 4 rows + 2 rim waves per block, 2 planes in flight, chunks of 16 (float) / 32 (double) planes —
// 4.4 TB/s (float) and 4.9 TB/s (double) where a copy of the same arrays with the flag stream reaches 5.8 / 5.9 TB/s.  Every
// form of the march measured lands within 5 % of that — one cell or 16 bytes per lane, with or without the LDS plane and its
// barrier, 70 or 15 instructions per cell, along x or along y (each wave one contiguous stream), and the march run as a plain
// copy (no LDS, no arithmetic) as well: the gap to the linear copy is not in the kernel's arithmetic or synchronisation.
template <typename T>
void launch_stencil_march(hipStream_t st, Grid g, const uint8_t* flags, const T* s, T* q, Coef<T> cf, int variant, int cxlen)
{
    constexpr int V = 16 / (int)sizeof(T);
    const bool can_vec = g.N % V == 0 && ((uintptr_t)s & 15) == 0 && ((uintptr_t)q & 15) == 0;
    if (variant >= 900000 && launch_stencil_dma<T>(st, g, flags, s, q, cf, variant - 900000, cxlen)) return;   // 900000 + D G NP RY: the LDS-DMA plane ring
    if (variant >= 900000) { variant = 0; cxlen = 0; }
    if (variant >= 80000 && can_vec) {   // 80000 + MY: the linear-front sweep, MY rows (waves) per block
        switch (variant - 80000) {
        case 2: front_launch<T, 2>(st, g, flags, s, q, cf); break;
        case 4: front_launch<T, 4>(st, g, flags, s, q, cf); break;
        case 8: front_launch<T, 8>(st, g, flags, s, q, cf); break;
        default: front_launch<T, 16>(st, g, flags, s, q, cf); break;
        }
        return;
    }
    if (variant >= 60000 && can_vec) {   // 60000 + MY*100 + MD: the lean march along y (every wave one contiguous stream); 70000 + ...: as a copy (probe)
        if (cxlen <= 0) cxlen = 32;
        switch (variant - 60000) {
        case 402: lean_launch<T, 4, 2, 0, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 403: lean_launch<T, 4, 3, 0, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 404: lean_launch<T, 4, 4, 0, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 602: lean_launch<T, 6, 2, 0, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 604: lean_launch<T, 6, 4, 0, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 802: lean_launch<T, 8, 2, 0, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 804: lean_launch<T, 8, 4, 0, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 1402: lean_launch<T, 14, 2, 0, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 1404: lean_launch<T, 14, 4, 0, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 408: lean_launch<T, 4, 8, 0, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 416: lean_launch<T, 4, 16, 0, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 616: lean_launch<T, 6, 16, 0, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 808: lean_launch<T, 8, 8, 0, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 816: lean_launch<T, 8, 16, 0, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 1416: lean_launch<T, 14, 16, 0, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 10402: lean_launch<T, 4, 2, 1, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 10804: lean_launch<T, 8, 4, 1, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 10416: lean_launch<T, 4, 16, 1, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 10816: lean_launch<T, 8, 16, 1, 1>(st, g, cxlen, flags, s, q, cf); break;
        default: lean_launch<T, 8, 2, 0, 1>(st, g, cxlen, flags, s, q, cf); break;
        }
        return;
    }
    if (variant >= 40000 && can_vec) {   // 40000 + MY*100 + MD: the lean LDS-plane march, MY rows (+ 2 rim waves), MD planes in flight
        if (cxlen <= 0) cxlen = 32;
        switch (variant - 40000) {
        case 402: lean_launch<T, 4, 2>(st, g, cxlen, flags, s, q, cf); break;
        case 403: lean_launch<T, 4, 3>(st, g, cxlen, flags, s, q, cf); break;
        case 404: lean_launch<T, 4, 4>(st, g, cxlen, flags, s, q, cf); break;
        case 408: lean_launch<T, 4, 8>(st, g, cxlen, flags, s, q, cf); break;
        case 602: lean_launch<T, 6, 2>(st, g, cxlen, flags, s, q, cf); break;
        case 604: lean_launch<T, 6, 4>(st, g, cxlen, flags, s, q, cf); break;
        case 802: lean_launch<T, 8, 2>(st, g, cxlen, flags, s, q, cf); break;
        case 808: lean_launch<T, 8, 8>(st, g, cxlen, flags, s, q, cf); break;
        case 1408: lean_launch<T, 14, 8>(st, g, cxlen, flags, s, q, cf); break;
        case 803: lean_launch<T, 8, 3>(st, g, cxlen, flags, s, q, cf); break;
        case 804: lean_launch<T, 8, 4>(st, g, cxlen, flags, s, q, cf); break;
        case 806: lean_launch<T, 8, 6>(st, g, cxlen, flags, s, q, cf); break;
        case 1402: lean_launch<T, 14, 2>(st, g, cxlen, flags, s, q, cf); break;
        case 1404: lean_launch<T, 14, 4>(st, g, cxlen, flags, s, q, cf); break;
        case 10402: lean_launch<T, 4, 2, 1>(st, g, cxlen, flags, s, q, cf); break;   // probe: the march as a copy
        case 10802: lean_launch<T, 8, 2, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 10804: lean_launch<T, 8, 4, 1>(st, g, cxlen, flags, s, q, cf); break;
        default: lean_launch<T, 8, 4>(st, g, cxlen, flags, s, q, cf); break;
        }
        return;
    }
    if (variant >= 30000 && can_vec) {   // 30000 + R*100 + MD: independent waves, R rows each, MD planes in flight
        if (cxlen <= 0) cxlen = 32;
        switch (variant - 30000) {
        case 101: rows_launch<T, 1, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 102: rows_launch<T, 1, 2>(st, g, cxlen, flags, s, q, cf); break;
        case 103: rows_launch<T, 1, 3>(st, g, cxlen, flags, s, q, cf); break;
        case 104: rows_launch<T, 1, 4>(st, g, cxlen, flags, s, q, cf); break;
        case 201: rows_launch<T, 2, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 202: rows_launch<T, 2, 2>(st, g, cxlen, flags, s, q, cf); break;
        case 203: rows_launch<T, 2, 3>(st, g, cxlen, flags, s, q, cf); break;
        case 401: rows_launch<T, 4, 1>(st, g, cxlen, flags, s, q, cf); break;
        case 402: rows_launch<T, 4, 2>(st, g, cxlen, flags, s, q, cf); break;
        default: rows_launch<T, 1, 2>(st, g, cxlen, flags, s, q, cf); break;
        }
        return;
    }
    if (variant >= 20000 && can_vec) {
        const long n16 = (long)g.cells() / V;
        const int nb = cxlen > 0 ? cxlen * 256 : 2048;
        if (variant == 20001) hipLaunchKernelGGL((k_stream_probe<T, false>), dim3(nb), dim3(256), 0, st, n16, flags, s, q);
        else hipLaunchKernelGGL((k_stream_probe<T, true>), dim3(nb), dim3(256), 0, st, n16, flags, s, q);
        return;
    }
    if (variant == 0 && can_vec) {
        // measured from HBM at 256^3 (tools/stencil_hbm.py): 4 planes in flight, chunks of 32 (fp32) / 64 (fp64) planes
        const int cx = (sizeof(T) == 4 ? 32 : 64) >> (g.N < 256 ? 1 : 0);
        lean_launch<T, 4, 4>(st, g, cx, flags, s, q, cf);
        return;
    }
    if (can_vec && variant >= 10000) {
        const int v = variant >= 10000 ? variant - 10000 : 402;
        if (cxlen <= 0) cxlen = 16;
        switch (v) {
        case 404: vec_launch<T, 4, 4>(st, g, cxlen, flags, s, q, cf); break;
        case 408: vec_launch<T, 4, 8>(st, g, cxlen, flags, s, q, cf); break;
        case 804: vec_launch<T, 8, 4>(st, g, cxlen, flags, s, q, cf); break;
        case 808: vec_launch<T, 8, 8>(st, g, cxlen, flags, s, q, cf); break;
        case 1604: vec_launch<T, 16, 4>(st, g, cxlen, flags, s, q, cf); break;
        case 202: vec_launch<T, 2, 2>(st, g, cxlen, flags, s, q, cf); break;
        case 204: vec_launch<T, 2, 4>(st, g, cxlen, flags, s, q, cf); break;
        case 402: vec_launch<T, 4, 2>(st, g, cxlen, flags, s, q, cf); break;
        default: vec_launch<T, 4, 2>(st, g, cxlen, flags, s, q, cf); break;
        }
        return;
    }
    if (cxlen <= 0) cxlen = 32;
    switch (variant) {
    case 804: march_launch<T, 8, 4>(st, g, cxlen, flags, s, q, cf); break;
    case 808: march_launch<T, 8, 8>(st, g, cxlen, flags, s, q, cf); break;
    case 1604: march_launch<T, 16, 4>(st, g, cxlen, flags, s, q, cf); break;
    case 1608: march_launch<T, 16, 8>(st, g, cxlen, flags, s, q, cf); break;
    case 404: march_launch<T, 4, 4>(st, g, cxlen, flags, s, q, cf); break;
    case 408: march_launch<T, 4, 8>(st, g, cxlen, flags, s, q, cf); break;
    default: march_launch<T, 16, 4>(st, g, cxlen, flags, s, q, cf); break;  // best of the sweep at 256^3 (profiles/r01/stencil_sweep.txt)
    }
}

template <typename T>
void launch_stencil_apply(hipStream_t st, Grid g, Box box, const uint8_t* flags, const T* s, T* q, Coef<T> cf)
{
    const Tiles tl = make_tiles(box);
    int nb = tl.count();
    if (nb > G_MAX_BLOCKS) nb = G_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL((k_stencil_g<T>), dim3(nb), dim3(256), 0, st, g, box, tl, flags, s, q, cf);
}

#define INST(T)                                                                                                                        \
    template void launch_pcg_init<T>(hipStream_t, Grid, LBox, const uint8_t*, const float*, T*, T*, Coef<T>, double*, double*,          \
                                     PcgState*);                                                                                       \
    template void launch_pcg_sq<T>(hipStream_t, LBox, const uint8_t*, const T*, const T*, T*, T*, Coef<T>, const double*, const double*, \
                                   const double*, double*, PcgState*, int, double, int, int, int, int);                                \
    template void launch_pcg_xr<T>(hipStream_t, LBox, const uint8_t*, T*, T*, const T*, const T*, Coef<T>, const double*, const double*, \
                                   double*, double*, PcgState*, int, int);                                                                \
    template void launch_pcg_sq_list<T>(hipStream_t, LBox, const uint8_t*, const T*, const T*, T*, T*, Coef<T>, const double*, int,      \
                                        const double*, const double*, double*, PcgState*, int, double, int, int, const int*, int);      \
    template void launch_pcg_xr_list<T>(hipStream_t, LBox, const uint8_t*, T*, T*, const T*, const T*, Coef<T>, const double*, int,      \
                                        const double*, double*, double*, PcgState*, const int*, int);                                    \
    template void launch_pcg_xr_rows<T>(hipStream_t, LBox, const uint8_t*, T*, T*, const T*, const T*, Coef<T>, const double*, int,      \
                                        const double*, int, double*, double*, PcgState*, const int*, int);                               \
    template void launch_stencil_apply<T>(hipStream_t, Grid, Box, const uint8_t*, const T*, T*, Coef<T>);                              \
    template void launch_stencil_march<T>(hipStream_t, Grid, const uint8_t*, const T*, T*, Coef<T>, int, int);                                   \
    template void launch_store_pressure<T>(hipStream_t, Grid, LBox, const uint8_t*, const T*, double*, double*, const PcgState*);      \
    template void launch_pcg_init_guess<T>(hipStream_t, Grid, LBox, const uint8_t*, const float*, const double*, const double*, double, double, T*, T*, Coef<T>, \
                                           double*, double*, PcgState*);                                                               \
    template void launch_pcg_sq_dist<T>(hipStream_t, LBox, const uint8_t*, const T*, const T*, T*, T*, Coef<T>, const double*, const double*, \
                                        const double*, double*, PcgState*, int, double, int);                                            \
    template void launch_pcg_xr_dist<T>(hipStream_t, LBox, const uint8_t*, T*, T*, const T*, const T*, Coef<T>, const double*,          \
                                        const double*, double*, double*, PcgState*);
INST(double)
INST(float)

}  // namespace fl
