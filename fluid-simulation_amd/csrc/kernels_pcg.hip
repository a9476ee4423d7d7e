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
// ZT: element type of `r` — the float cycle's z stays float in memory (4 bytes less written by the up leg and read here per cell; the
// conversion is exact, so nothing else changes)
template <typename T, bool DIST, bool AZ = false, typename ZT = T>
__global__ __launch_bounds__(256) void k_pcg_sq_l(LBox L, const uint8_t* __restrict__ cnt, const ZT* __restrict__ r,
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
    ZT rv[TX + 2], ry = 0, rz = 0;
    T sv[TX + 2], sy = 0, sz = 0;
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
    // AZ (Chronopoulos-Gear form of the decomposed solve): q = A r and the partials of r.q only — no scalars to derive, s_in unread
    // (the caller passes first = 1), s_out unwritten
    if (!AZ && !pcg_head<T>(part_rr, part_rz_new, part_rz_old, n_prev, n_rz, ps, first, tol, red, beta)) return;

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
        for (int m = 0; m < TX + 2; ++m) val[m] = fc[m] ? (zmode ? (T)rv[m] : (T)rv[m] * sinv[DIST ? fc[m] & 7 : fc[m]]) + beta * sv[m] : (T)0;
#pragma unroll
        for (int lx = 0; lx < TX; ++lx) sT[(lx * PY + ly + 1) * PZ + kz + 1] = val[lx + 1];
        {
            const int pl = tid >> 6, side = (tid >> 5) & 1;
            sT[(pl * PY + (side ? TY + 1 : 0)) * PZ + kz + 1] = fy ? (zmode ? (T)ry : (T)ry * sinv[DIST ? fy & 7 : fy]) + beta * sy : (T)0;
        }
        if (tid < 64) {
            const int lx = tid >> 4, l2 = (tid >> 1) & 7, side = tid & 1;
            sT[(lx * PY + l2 + 1) * PZ + (side ? TZ + 1 : 0)] = fz ? (zmode ? (T)rz : (T)rz * sinv[DIST ? fz & 7 : fz]) + beta * sz : (T)0;
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
            if (!AZ) s_out[c] = cen[lx];
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

// out[0..3] = the sums of four partial arrays (one block; fixed order; an array with n = 0 gives 0)
__global__ __launch_bounds__(256) void k_sum4(const double* __restrict__ a, int na, const double* __restrict__ b, int nb, const double* __restrict__ c, int nc,
                                              const double* __restrict__ e, int ne, double* __restrict__ out)
{
    __shared__ double red[16];
    double ra, rb, rc, re, d1, d2;
    block_sum3(a, na, b, nb, c, nc, red, ra, rb, rc);
    __syncthreads();
    block_sum3(e, ne, e, 0, e, 0, red, re, d1, d2);
    if (threadIdx.x == 0) { out[0] = ra; out[1] = rb; out[2] = rc; out[3] = re; }
}

// Chronopoulos-Gear form of the preconditioned CG loop (decomposed solve, FLUID_DIST_CG=cgear): with z = M^-1 r and w = A z
// already formed, ONE all-reduce carries {|r|^2, gamma = r.z, delta = w.z} of the iteration (g[0..2]; g[3] = |r0|^2 of a solve
// started from a guess), and this kernel does the rest of the body in one stream over the local box:
//   beta = gamma / gamma_old ; alpha = gamma / (delta - beta gamma / alpha_old)
//   s = z + beta s ; q = w + beta q  (= A s by recurrence) ; x += alpha s ; r -= alpha q ; partial |r|^2
// on the owned unknowns (count byte non-zero, bit 7 clear).  Same start, stopping rule and iteration count as pcg_head
// (ConjugateGradient.h:28-90); in exact arithmetic the same iterates.  cg = {gamma, alpha} of the last two bodies, by parity.
template <typename T, typename ZT = T>
__global__ __launch_bounds__(256) void k_pcg_cgear_upd(long n2, const uint8_t* __restrict__ cnt, T* __restrict__ x, T* __restrict__ r, T* __restrict__ s,
                                                       T* __restrict__ q, const ZT* __restrict__ z, const T* __restrict__ w, const double* __restrict__ g,
                                                       double* __restrict__ cg, int cur, double* __restrict__ part_rr, PcgState* ps, int first, double tol)
{
    __shared__ double red[16];
    __shared__ int s_done;
    if (threadIdx.x == 0) s_done = ps->done;
    typedef Vec2<T> V2;
    const uint16_t* c2 = (const uint16_t*)cnt;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long e0 = (long)vb * per, e1 = e0 + per < n2 ? e0 + per : n2;
    __syncthreads();
    if (s_done) return;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    const double gamma = g[1], delta = g[2];
    double beta = 0, alpha;
    if (first) {
        const double bb = g[0], rr0 = first == 2 ? g[3] : bb;
        const bool go = bb > 0 && !(rr0 < tol * tol * bb);
        if (lead) {
            ps->bb = bb;
            ps->thr = tol * tol * bb;
            ps->rr = rr0;
            if (!go) ps->done = 1;
        }
        if (!go) return;
        alpha = gamma / delta;
        if (!(delta > 0) || !(gamma == gamma)) {
            if (lead) { ps->breakdown = 1; ps->done = 1; }
            return;
        }
    } else {
        const double rr = g[0];
        if (rr < ps->thr) {
            if (lead) { ps->rr = rr; ps->done = 1; }
            return;
        }
        const double go = cg[2 * (cur ^ 1)], ao = cg[2 * (cur ^ 1) + 1];
        beta = gamma / go;
        const double den = delta - beta * gamma / ao;      // = s.As of the standard recurrence
        if (!(den > 0) || !(gamma == gamma)) {
            if (lead) { ps->breakdown = 1; ps->done = 1; }
            return;
        }
        alpha = gamma / den;
        if (lead) { ps->rr = rr; ps->iters += 1; }
    }
    if (lead) { cg[2 * cur] = gamma; cg[2 * cur + 1] = alpha; }
    const T be = (T)beta, al = (T)alpha;
    double arr = 0;
    for (long i = e0 + threadIdx.x; i < e1; i += 256) {
        const uint16_t cw = c2[i];
        int ca = cw & 0xff, cb = cw >> 8;
        ca = (ca & 0x80) ? 0 : ca;
        cb = (cb & 0x80) ? 0 : cb;
        if (!(ca | cb)) continue;
        const Vec2<ZT> zf = ((const Vec2<ZT>*)z)[i];   // (ZT = float: the float cycle's z as it left it, converted exactly)
        const V2 zv = {(T)zf.a, (T)zf.b}, wv = ((const V2*)w)[i];
        V2 sv = ((V2*)s)[i], qv = ((V2*)q)[i], xv = ((V2*)x)[i], rv = ((V2*)r)[i];
        if (ca) {
            sv.a = zv.a + be * sv.a;
            qv.a = wv.a + be * qv.a;
            xv.a = xv.a + al * sv.a;
            rv.a = rv.a - al * qv.a;
            arr += (double)rv.a * (double)rv.a;
        }
        if (cb) {
            sv.b = zv.b + be * sv.b;
            qv.b = wv.b + be * qv.b;
            xv.b = xv.b + al * sv.b;
            rv.b = rv.b - al * qv.b;
            arr += (double)rv.b * (double)rv.b;
        }
        ((V2*)s)[i] = sv; ((V2*)q)[i] = qv; ((V2*)x)[i] = xv; ((V2*)r)[i] = rv;
    }
    arr = block_sum<double, 4>(arr, red);
    if (threadIdx.x == 0) part_rr[blockIdx.x] = arr;
}

// The convergence poll of the Chronopoulos-Gear loop (decomposed run).  A body learns the global |r|^2 of the body before it from its one
// all-reduce, so a solve of n bodies would need an (n + 1)-th — a whole V-cycle — only to find out that it is over.  Instead, where the host
// polls anyway: out = {this rank's sum of the last body's |r|^2 partials, done so far}, one all-reduce (SUM) of the two, then the test of
// ConjugateGradient.h:76-79 on the global value — the same on every rank, and the same value the next body's head would have seen.
__global__ __launch_bounds__(256) void k_pcg_poll_stage(const double* __restrict__ part_rr, int n, const PcgState* ps, double* __restrict__ out)
{
    __shared__ double red[16];
    double ra, rb, rc;
    block_sum3(part_rr, n, part_rr, 0, part_rr, 0, red, ra, rb, rc);
    if (threadIdx.x == 0) {
        out[0] = ra;
        out[1] = ps->done ? 1.0 : 0.0;
    }
}
__global__ void k_pcg_poll_test(const double* __restrict__ g, PcgState* ps)
{
    if (g[1] > 0) ps->done = 1;   // some rank's body has ended the solve already (start test, breakdown, its own head)
    else if (g[0] < ps->thr) { ps->rr = g[0]; ps->done = 1; }   // (iters: Eigen leaves the loop before counting the body)
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
// the same with z = M^-1 r as the float cycle left it (zmode; tlist = nullptr: every tile)
void launch_pcg_sq_zf(hipStream_t st, LBox L, const uint8_t* cnt, const float* z, const double* s_in, double* s_out, double* q, Coef<double> cf,
                      const double* part_rr, int n_prev, const double* part_rz_new, const double* part_rz_old, double* part_pq, PcgState* ps, int first,
                      double tol, int n_rz, int sparse, const int* tlist, int nlist)
{
    hipLaunchKernelGGL((k_pcg_sq_l<double, false, false, float>), dim3(tlist ? pcg_list_blocks(nlist) : pcg_sq_blocks(L)), dim3(256), 0, st, L, cnt, z, s_in, s_out, q,
                       cf, part_rr, part_rz_new, part_rz_old, part_pq, n_prev, ps, first, tol, n_rz, 1, tlist ? 0 : sparse, tlist, nlist);
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
void launch_sum4(hipStream_t st, const double* a, int na, const double* b, int nb, const double* c, int nc, const double* e, int ne, double* out)
{
    hipLaunchKernelGGL(k_sum4, dim3(1), dim3(256), 0, st, a, na, b, nb, c, nc, e, ne, out);
}
// Chronopoulos-Gear form (decomposed solve): w = A z on the owned cells + pcg_sq_blocks(L) partials of z.w ...
template <typename T>
void launch_pcg_az_dist(hipStream_t st, LBox L, const uint8_t* cnt, const T* z, T* w, Coef<T> cf, double* part_zw, PcgState* ps, const int* tlist, int nlist)
{
    hipLaunchKernelGGL((k_pcg_sq_l<T, true, true>), dim3(tlist ? pcg_list_blocks(nlist) : pcg_sq_blocks(L)), dim3(256), 0, st, L, cnt, z, (const T*)nullptr,
                       (T*)nullptr, w, cf, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, part_zw, 1, ps, 1, 0.0, 1, 1, 0, tlist, nlist);
}
// ... and the rest of the body from the all-reduced scalars g; writes pcg_xr_blocks(L) partials of |r|^2
template <typename T>
void launch_pcg_cgear_upd(hipStream_t st, LBox L, const uint8_t* cnt, T* x, T* r, T* s, T* q, const T* z, const T* w, const double* g, double* cg, int cur,
                          double* part_rr, PcgState* ps, int first, double tol)
{
    hipLaunchKernelGGL((k_pcg_cgear_upd<T>), dim3(pcg_xr_blocks(L)), dim3(256), 0, st, (long)(L.cells() / 2), cnt, x, r, s, q, z, w, g, cg, cur, part_rr, ps, first, tol);
}
// the same two with z = M^-1 r kept float (the float cycle's result: 8 bytes per cell and body less)
void launch_pcg_az_dist_zf(hipStream_t st, LBox L, const uint8_t* cnt, const float* z, double* w, Coef<double> cf, double* part_zw, PcgState* ps, const int* tlist, int nlist)
{
    hipLaunchKernelGGL((k_pcg_sq_l<double, true, true, float>), dim3(tlist ? pcg_list_blocks(nlist) : pcg_sq_blocks(L)), dim3(256), 0, st, L, cnt, z, (const double*)nullptr,
                       (double*)nullptr, w, cf, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, part_zw, 1, ps, 1, 0.0, 1, 1, 0, tlist, nlist);
}
void launch_pcg_cgear_upd_zf(hipStream_t st, LBox L, const uint8_t* cnt, double* x, double* r, double* s, double* q, const float* z, const double* w, const double* g,
                             double* cg, int cur, double* part_rr, PcgState* ps, int first, double tol)
{
    hipLaunchKernelGGL((k_pcg_cgear_upd<double, float>), dim3(pcg_xr_blocks(L)), dim3(256), 0, st, (long)(L.cells() / 2), cnt, x, r, s, q, z, w, g, cg, cur, part_rr, ps, first, tol);
}
void launch_pcg_poll_stage(hipStream_t st, const double* part_rr, int n, const PcgState* ps, double* out)
{
    hipLaunchKernelGGL(k_pcg_poll_stage, dim3(1), dim3(256), 0, st, part_rr, n, ps, out);
}
void launch_pcg_poll_test(hipStream_t st, const double* g, PcgState* ps) { hipLaunchKernelGGL(k_pcg_poll_test, dim3(1), dim3(1), 0, st, g, ps); }
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
    template void launch_store_pressure<T>(hipStream_t, Grid, LBox, const uint8_t*, const T*, double*, double*, const PcgState*);      \
    template void launch_pcg_init_guess<T>(hipStream_t, Grid, LBox, const uint8_t*, const float*, const double*, const double*, double, double, T*, T*, Coef<T>, \
                                           double*, double*, PcgState*);                                                               \
    template void launch_pcg_sq_dist<T>(hipStream_t, LBox, const uint8_t*, const T*, const T*, T*, T*, Coef<T>, const double*, const double*, \
                                        const double*, double*, PcgState*, int, double, int);                                            \
    template void launch_pcg_xr_dist<T>(hipStream_t, LBox, const uint8_t*, T*, T*, const T*, const T*, Coef<T>, const double*,          \
                                        const double*, double*, double*, PcgState*);                                                    \
    template void launch_pcg_az_dist<T>(hipStream_t, LBox, const uint8_t*, const T*, T*, Coef<T>, double*, PcgState*, const int*, int);  \
    template void launch_pcg_cgear_upd<T>(hipStream_t, LBox, const uint8_t*, T*, T*, T*, T*, const T*, const T*, const double*, double*, int, double*, PcgState*, int, double);
INST(double)
INST(float)

}  // namespace fl
