// Geometric multigrid V-cycle used as the PCG preconditioner (z = M^-1 r) of the pressure solve.
//
// The reference preconditions with Eigen's IncompleteCholesky (fluid.cc:1352) — serial triangular
// solves.  Any SPD preconditioner leaves the converged solution of A p = b unchanged; this one makes
// the iteration count nearly independent of the grid size (20 / 21 / 31 at 128^3 / 256^3 / 512^3 against
// 308 / 562 at the first two for Jacobi, tests/experiments/mg_prototype.py).  Structure (after McAdams, Sifakis, Teran 2010,
// restated):
//   * levels: cells coarsened 2x2x2; a coarse cell is AIR (Dirichlet, p=0) if any child is air,
//     SOLID (Neumann) if all children are solid, else FLUID; operator = the same 7-point form with
//     the off-diagonal divided by 4 per level, diagonal = (#non-solid neighbours) x |off|;
//   * smoother: two damped-Jacobi sweeps before and two after the coarse correction, weights in
//     reversed order (symmetric -> M is SPD);
//   * transfer: cell-centred trilinear prolongation P (weights 3/4,1/4 per axis), restriction P^T/8,
//     the correction into the two finest levels over-weighted;
//   * coarsest level (<= 8^3): red-black Gauss-Seidel in LDS by one block, forward then reversed.
// Level 0 lives in the solver's box-local layout (LBox); coarser levels use the same indexing scheme
// (MLevel).  Only unknown cells are ever written (every array is zeroed once per step).  The cycle is
// latency-bound, so each level costs ONE kernel per leg (k_mg_down, k_mg_up: LDS tiles that recompute their
// halo) and all levels of <= 4 k cells run inside one block (k_mg_tail); it computes in float inside the
// double PCG (fluid_api.hip, mg_vcycle_t).
#include "common.h"
#include <cstdlib>
#include <cstring>

namespace fl {

// Damped-Jacobi weights of the two sweeps on each side of the coarse correction: (W1, W2) before, (W2, W1) after
// (reversed, so M stays symmetric).  The two sweeps apply the polynomial (1 - W1 x)(1 - W2 x) of x = D^-1 A to the
// error; 1/W1, 1/W2 are the roots of the degree-2 Chebyshev polynomial of the interval [0.5, 2] (the upper three
// quarters of the spectrum (0,2)): |p| <= 0.22 there, < 1 on all of (0,2).  25 PCG iterations per solve at 256^3
// against 26 for (2/3, 1.2) and 30 for plain (2/3, 2/3).
constexpr double MG_W1 = 0.5617, MG_W2 = 1.3895;
// The coarse-grid correction of the two finest levels is over-weighted (k_mg_up's `wc`; fluid_api.hip: 1.25 into level 0,
// 1.1 into level 1, 1 deeper and inside the tail): the coarse operators are re-discretisations (off-diagonal / 4 per level),
// not Galerkin products P^T A P, and with cell-centred trilinear P they under-estimate the correction.  A scalar keeps M
// symmetric.  PCG iterations per solve (tools/wc_sweep.sh): 128^3 23 -> 20, 256^3 25 -> 21, 512^3 31 -> 31 (there any
// weight on the deeper levels costs iterations: uniform 1.15 gave 20.6 / 22.7 / 33).

// static indices only (a runtime index into a by-value kernel argument goes through scratch)
template <typename T>
__device__ __forceinline__ void mg_load_coef(T* sd, T* si, const MgCoef<T>& cf)
{
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            sd[i] = cf.diag[i];
            si[i] = cf.inv[i];
        }
    }
    __syncthreads();
}

// thread index -> domain cell; 32-bit arithmetic (a level never has 2^31 cells): the 64-bit div/mod sequence was
// the dominant cost of these kernels on large, mostly-air boxes
__device__ __forceinline__ bool mg_cell(const MLevel& m, long t, int& i, int& j, int& k)
{
    const unsigned n = (unsigned)m.dx * (unsigned)m.dy * (unsigned)m.dz;
    if (t >= (long)n) return false;
    const unsigned u = (unsigned)t, dz = (unsigned)m.dz, dy = (unsigned)m.dy;
    const unsigned q = u / dz;
    k = (int)(u - q * dz);
    const unsigned p = q / dy;
    j = (int)(q - p * dy);
    i = (int)p;
    return true;
}

// level-0 cell type from the global flags: 0 solid (or off the grid), 1 air, 2 unknown
__global__ __launch_bounds__(256) void k_mg_type0(Grid g, LBox L, MLevel m, const uint8_t* __restrict__ flags, const uint8_t* __restrict__ cnt,
                                                  uint8_t* __restrict__ typ)
{
    int i, j, k;
    if (!mg_cell(m, (long)blockIdx.x * 256 + threadIdx.x, i, j, k)) return;
    const int gx = L.x0 + i - 1, gy = L.y0 + j - 1, gz = L.z0 + k - 1;
    uint8_t t = 0;
    const size_t c = m.at(i, j, k);
    if (gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny && gz >= 0 && gz < g.nz) {
        const uint8_t f = flags[g.idx(gx, gy, gz)];
        t = (f & F_SOLID) ? 0 : (cnt[c] ? 2 : 1);
    }
    typ[c] = t;
}

__global__ __launch_bounds__(256) void k_mg_coarsen(MLevel mf, const uint8_t* __restrict__ tf, MLevel mc, uint8_t* __restrict__ tc)
{
    int I, J, K;
    if (!mg_cell(mc, (long)blockIdx.x * 256 + threadIdx.x, I, J, K)) return;
    bool any_air = false, all_solid = true;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int i = 2 * I + (a & 1), j = 2 * J + ((a >> 1) & 1), k = 2 * K + (a >> 2);
        uint8_t t = 0;
        if (i < mf.dx && j < mf.dy && k < mf.dz) t = tf[mf.at(i, j, k)];
        any_air |= (t == 1);
        all_solid &= (t == 0);
    }
    tc[mc.at(I, J, K)] = any_air ? 1 : (all_solid ? 0 : 2);
}

// cnt = number of non-solid 6-neighbours of an unknown cell (0 = not an unknown); array cells outside the
// domain hold type 0
__global__ __launch_bounds__(256) void k_mg_cnt(MLevel m, const uint8_t* __restrict__ typ, uint8_t* __restrict__ cnt)
{
    int i, j, k;
    if (!mg_cell(m, (long)blockIdx.x * 256 + threadIdx.x, i, j, k)) return;
    const size_t c = m.at(i, j, k);
    uint8_t n = 0;
    if (typ[c] == 2) {
        n = (typ[c - m.sx] != 0) + (typ[c + m.sx] != 0) + (typ[c - m.sy] != 0) + (typ[c + m.sy] != 0) + (typ[c - 1] != 0) + (typ[c + 1] != 0);
    }
    cnt[c] = n;
}

// Coefficients by neighbour count n: diag = dg[n], 1/diag = iv[n] (tables in LDS), off-diagonal = off.

// How a kernel touches the arrays that other workgroups of the SAME launch produce or consume.  Between launches plain
// accesses are enough (a kernel boundary publishes everything); inside the persistent coarse-level kernel (k_mg_coarse)
// every such access is an `sc1` one — stores write through, loads bypass the CU's L1 — paired with one agent-scope
// counter add per workgroup after its stores have drained and an `sc1` poll on the consumer side (MI355X guide,
// "Workgroup dispatch, XCD placement & inter-workgroup visibility": placement-independent, no L2 write-back fence).
struct IoPlain {
    template <typename X> static __device__ __forceinline__ X ld(const X* p) { return *p; }
    template <typename X> static __device__ __forceinline__ void st(X* p, X v) { *p = v; }
};
struct IoSc1 {
    template <typename X> static __device__ __forceinline__ X ld(const X* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    template <typename X> static __device__ __forceinline__ void st(X* p, X v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};


// f_c = (1/8) P^T r_f : a coarse cell gathers its 4x4x4 fine neighbourhood, weights (1/4,3/4,3/4,1/4) per axis
template <typename T, typename IO = IoPlain>
__device__ __forceinline__ void d_restrict(const MLevel& mf, const T* __restrict__ rf, const MLevel& mc, const uint8_t* __restrict__ cnt_c,
                                           T* __restrict__ fc, long t)
{
    int I, J, K;
    if (!mg_cell(mc, t, I, J, K)) return;
    const size_t C = mc.at(I, J, K);
    T out = 0;
    if (!cnt_c[C]) return;
    {
        auto w = [](int a) { return (a == 0 || a == 3) ? (T)0.25 : (T)0.75; };  // no private array: no scratch
        const int i0 = 2 * I - 1, j0 = 2 * J - 1, k0 = 2 * K - 1;
        T acc = 0;
        if (i0 >= 0 && i0 + 3 < mf.dx && j0 >= 0 && j0 + 3 < mf.dy && k0 >= 0 && k0 + 3 < mf.dz) {
            const T* p = rf + mf.at(i0, j0, k0);  // fast path: all 64 inside, 16 rows of 4 contiguous values
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const T* q = p + a * mf.sx + b * mf.sy;
                    acc += w(a) * w(b) * ((T)0.25 * (q[0] + q[3]) + (T)0.75 * (q[1] + q[2]));
                }
            }
        } else {
            for (int a = 0; a < 4; ++a) {
                const int i = i0 + a;
                if (i < 0 || i >= mf.dx) continue;
                for (int b = 0; b < 4; ++b) {
                    const int j = j0 + b;
                    if (j < 0 || j >= mf.dy) continue;
                    T row = 0;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int k = k0 + d;
                        if (k >= 0 && k < mf.dz) row += w(d) * rf[mf.at(i, j, k)];
                    }
                    acc += w(a) * w(b) * row;
                }
            }
        }
        out = acc * (T)0.125;
    }
    IO::st(fc + C, out);
}

// ---- LDS-tiled legs of the V-cycle -----------------------------------------------------------------------
// The solve is launch-bound (a level-0 sweep over ~700k cells is ~9 us, ~3 of them launch latency, and a level needs
// 5-6 of them per cycle), so each leg of a level is ONE kernel: a block owns a TX x TY x TZ tile, stages the tile plus
// a halo in LDS and recomputes the halo's share of the earlier sweeps itself instead of waiting for a grid-wide sync.
//   down: u1 = W1 D^-1 f (tile + halo H), u2 = second pre-sweep (halo H-1), r = f - A u2 (halo H-2); with RESTRICT
//         (H = 3) the tile's 2x2x2-coarsened cells gather their 4x4x4 residuals straight from LDS, else (H = 2) r goes
//         to HBM for the restriction kernel / the tail;
//   up:   v0 = u + P e (tile + halo 2, the coarse correction staged in LDS first), two post-sweeps, z = result, plus
//         the block's partial of f.z (level 0: the r.z of PCG).
// (Skipping tiles without an unknown in mostly-air boxes was tried twice — a block-wide test of the count bytes, and
// per-tile flags built once per step — and lost both times: in the splash 35-65 % of the tiles hold a droplet, and
// the extra dependent load in front of every block cost more than the empty tiles saved.)
// Index arithmetic is per-tile with compile-time region sizes (no 64-bit div/mod: the thread-per-cell kernels spent
// a third of their VALU time there); neighbour sums keep the order of d_smooth / d_resid.
template <int RY, int RZ>
__device__ __forceinline__ void region_cell(int t, int& x, int& y, int& z)
{
    x = t / (RY * RZ);
    const int r = t - x * (RY * RZ);
    y = r / RZ;
    z = r - y * RZ;
}
__device__ __forceinline__ bool in_level(const MLevel& m, int i, int j, int k)
{
    return (unsigned)i < (unsigned)m.dx && (unsigned)j < (unsigned)m.dy && (unsigned)k < (unsigned)m.dz;
}

// Thread mapping: a thread owns one (y,z) COLUMN of a region and walks along x, so the div/mod that finds the column
// runs once per stage, addresses advance by one multiply-add, the x-neighbours of the sweeps slide through registers
// (5 LDS reads per point instead of 7) and the in-plane part of the trilinear interpolation is shared by the fine
// planes above one coarse plane.  (A first version that spread region points over threads by a linear index was
// VALU-bound on exactly that index arithmetic: 26 us for the level-0 up leg.)  All global loads of a block are issued
// up front into registers from clamped (always readable) addresses, so the in-level tests never delay a load.
__device__ __forceinline__ int clampi(int v, int hi) { return min(max(v, 0), hi); }

// LDS of one tile of a leg (bytes, 16-byte aligned pieces): the legs are device functions over caller-provided LDS so that
// the stand-alone kernels (one tile per 256-thread block) and the persistent coarse-level kernel (four tiles per
// 1024-thread block) run the same code
template <typename T, int TX, int TY, int TZ, bool RESTRICT>
struct DownTile {
    static constexpr int H = RESTRICT ? 3 : 2;
    static constexpr int AX = TX + 2 * H, AY = TY + 2 * H, AZ = TZ + 2 * H;  // u1 and the count bytes
    static constexpr int BX = AX - 2, BY = AY - 2, BZ = AZ - 2;              // u2
    static constexpr int CX = BX - 2, CY = BY - 2, CZ = BZ - 2;              // r (the tile itself unless RESTRICT)
    static constexpr int nA = AX * AY * AZ, nB = BX * BY * BZ, nR = RESTRICT ? CX * CY * CZ : 4;
    static constexpr int oA = 0, oB = (oA + nA * (int)sizeof(T) + 15) / 16 * 16, oR = (oB + nB * (int)sizeof(T) + 15) / 16 * 16,
                         oC = (oR + nR * (int)sizeof(T) + 15) / 16 * 16, bytes = (oC + nA + 15) / 16 * 16;
};
template <typename T, int TX, int TY, int TZ>
struct UpTile {
    static constexpr int AX = TX + 4, AY = TY + 4, AZ = TZ + 4;              // v0 = u + P e and the count bytes
    static constexpr int BX = TX + 2, BY = TY + 2, BZ = TZ + 2;              // v1
    static constexpr int EX = TX / 2 + 4, EY = TY / 2 + 4, EZ = TZ / 2 + 4;  // coarse correction under the v0 region
    static constexpr int nA = AX * AY * AZ, nB = BX * BY * BZ, nE = EX * EY * EZ;
    static constexpr int oA = 0, oB = (oA + nA * (int)sizeof(T) + 15) / 16 * 16, oE = (oB + nB * (int)sizeof(T) + 15) / 16 * 16,
                         oC = (oE + nE * (int)sizeof(T) + 15) / 16 * 16, bytes = (oC + nA + 15) / 16 * 16;
};

// One tile of the down leg by NG groups of 256 threads (`col` = thread within its group, `grp` = the group).  The x planes of
// every stage are split among the groups — a coarse level of a few hundred tiles leaves most CUs with ONE tile, whose 256 threads
// walk 14 planes per stage one after the other: four groups walk 3-4 each (NG = 4, levels >= 1 with few tiles) and the leg's
// dependent chain shrinks accordingly; a level with many tiles per CU is bound by LDS throughput, not by the chain: NG = 1.
// sd / si: coefficient tables in LDS; with `cf` the body fills them itself (after issuing its loads), else the caller has, behind
// a barrier.  `live` = false: a group without a tile of its own walks a valid one for the barriers' sake and stores nothing.
// rhs(q): the right-hand side at array index q (the plain form loads f[q]; the fused XR + down leg of level 0 forms r - alpha q there);
// core(q): called once for every unknown of the tile itself, by the thread that forms its residual (the fused leg updates x and r there).
template <typename T, typename F, int TX, int TY, int TZ, bool RESTRICT, typename IO, int NG, typename RhsFn, typename CoreFn>
__device__ __forceinline__ void mg_down_body_fn(const MLevel& m, const uint8_t* __restrict__ cnt, T* __restrict__ u,
                                                T* __restrict__ r, const MLevel& mc, const uint8_t* __restrict__ cnt_c, T* __restrict__ fc,
                                                const MgCoef<T>* cf, T off, int tile, int gx, int gy, bool live, int col, char* lds, T* sd, T* si,
                                                int grp, RhsFn rhs, CoreFn core)
{
    typedef DownTile<T, TX, TY, TZ, RESTRICT> D;
    constexpr int H = D::H, AX = D::AX, AY = D::AY, AZ = D::AZ, BX = D::BX, BY = D::BY, BZ = D::BZ, CX = D::CX, CY = D::CY, CZ = D::CZ;
    constexpr int PA = (AX + NG - 1) / NG, PB = (BX + NG - 1) / NG;   // planes of regions A / B per group
    constexpr int NHC = NG * (256 / (CY * CZ)), XC = (CX + NHC - 1) / NHC;   // the r columns are few: NHC threads (over all groups) share one, XC planes each
    static_assert(AY * AZ <= 256 && NHC >= 1 && TX % 2 == 0 && TY % 2 == 0 && TZ % 2 == 0, "tile shape");
    T* sA = (T*)(lds + D::oA);
    T* sB = (T*)(lds + D::oB);
    T* sR = (T*)(lds + D::oR);
    uint8_t* sC = (uint8_t*)(lds + D::oC);
    const int tbx = tile % gx, tby = (tile / gx) % gy, tbz = tile / (gx * gy);
    const int i0 = tbz * TX, j0 = tby * TY, k0 = tbx * TZ;
    const T w1 = (T)MG_W1, w2 = (T)MG_W2;
    const long sx = m.sx;
    // ---- every global load of the block ----
    const int ya = col / AZ, za = col - ya * AZ;
    const bool actA = col < AY * AZ;
    const bool okA = actA && (unsigned)(j0 - H + ya) < (unsigned)m.dy && (unsigned)(k0 - H + za) < (unsigned)m.dz;
    const size_t qa = m.at(0, clampi(j0 - H + ya, m.dy - 1), clampi(k0 - H + za, m.dz - 1));
    const int xa0 = grp * PA;
    int ca[PA];
    T fa[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int x = xa0 + p, i = i0 - H + x;
        const size_t q = qa + (size_t)((long)clampi(i, m.dx - 1) * sx);
        const int c = cnt[q];
        fa[p] = (T)rhs(q);  // (same loop as the count load: split into two loops, the selects below stall the f loads behind the counts)
        ca[p] = (okA && x < AX && (unsigned)i < (unsigned)m.dx) ? c : 0;
    }
    const int yb = col / BZ, zb = col - yb * BZ;
    const bool actB = col < BY * BZ;
    const size_t qb = m.at(0, clampi(j0 - H + 1 + yb, m.dy - 1), clampi(k0 - H + 1 + zb, m.dz - 1));
    const int xb0 = grp * PB;
    T fb[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) fb[p] = (T)rhs(qb + (size_t)((long)clampi(i0 - H + 1 + xb0 + p, m.dx - 1) * sx));
    const int hc = grp * (256 / (CY * CZ)) + col / (CY * CZ), cc = col % (CY * CZ);
    const int yc = cc / CZ, zc = cc - yc * CZ;
    const bool actC = col / (CY * CZ) < 256 / (CY * CZ);
    const int xc0 = hc * XC;
    const size_t qc = m.at(0, clampi(j0 - H + 2 + yc, m.dy - 1), clampi(k0 - H + 2 + zc, m.dz - 1));
    T fr[XC];
#pragma unroll
    for (int x = 0; x < XC; ++x) fr[x] = (T)rhs(qc + (size_t)((long)clampi(i0 - H + 2 + xc0 + x, m.dx - 1) * sx));
    if (cf) mg_load_coef(sd, si, *cf);
    // ---- u1 = W1 D^-1 f on region A ----
    if (actA) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int x = xa0 + p;
            if (x < AX) {
                sC[x * AY * AZ + col] = (uint8_t)ca[p];
                sA[x * AY * AZ + col] = w1 * si[ca[p]] * fa[p];  // si[0] = 0
            }
        }
    }
    __syncthreads();
    // ---- u2 on region B (branch-free: a non-unknown has n = 0, si[0] = sd[0] = 0 and u1 = 0, so 0 falls out) ----
    if (actB && xb0 < BX) {
        const int a0 = (yb + 1) * AZ + zb + 1 + xb0 * AY * AZ;
        T cm = sA[a0], c0 = sA[a0 + AY * AZ];
        const bool in_yz = live && (unsigned)(yb - (H - 1)) < (unsigned)TY && (unsigned)(zb - (H - 1)) < (unsigned)TZ;
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int x = xb0 + p;
            if (x < BX) {
                const int a = a0 + (p + 1) * AY * AZ;
                const T cp = sA[a + AY * AZ];
                const int n = sC[a];
                const T nb = cm + cp + sA[a - AZ] + sA[a + AZ] + sA[a - 1] + sA[a + 1];
                const T v = c0 + w2 * si[n] * (fb[p] - (sd[n] * c0 + off * nb));
                sB[(x * BY + yb) * BZ + zb] = v;
                if (n && in_yz && x >= H - 1 && x < H - 1 + TX) IO::st(u + qb + (size_t)((long)(i0 - H + 1 + x) * sx), v);  // n != 0: in the level, no clamp
                cm = c0;
                c0 = cp;
            }
        }
    }
    __syncthreads();
    // ---- r = f - A u2 on region C ----
    T rk[XC];   // (kept for the piecewise-constant restriction below)
#pragma unroll
    for (int x = 0; x < XC; ++x) rk[x] = 0;
    if (actC && xc0 < CX) {
        const int b0 = ((xc0 + 1) * BY + yc + 1) * BZ + zc + 1;
        T cm = sB[b0 - BY * BZ], c0 = sB[b0];
#pragma unroll
        for (int x = 0; x < XC; ++x) {
            if (xc0 + x < CX) {
                const int b = b0 + x * BY * BZ;
                const T cp = sB[b + BY * BZ];
                const int n = sC[((xc0 + x + 2) * AY + yc + 2) * AZ + zc + 2];
                const T nb = cm + cp + sB[b - BZ] + sB[b + BZ] + sB[b - 1] + sB[b + 1];
                const T v = n ? fr[x] - (sd[n] * c0 + off * nb) : (T)0;
                rk[x] = v;
                if (RESTRICT) sR[((xc0 + x) * CY + yc) * CZ + zc] = v;
                else if (n && live) {
                    IO::st(r + qc + (size_t)((long)(i0 + xc0 + x) * sx), v);
                    core(qc + (size_t)((long)(i0 + xc0 + x) * sx));
                }
                cm = c0;
                c0 = cp;
            }
        }
    }
    if constexpr (!RESTRICT) {
        // fc given without the trilinear restriction: the Galerkin levels' restriction (kernels_gal.hip) — the coarse right-hand side is the SUM
        // of the residual over a coarse cell's 8 children, all of them cells of this tile: through the LDS region u2 no longer needs
        if (fc) {   // (block-uniform)
            __syncthreads();
            if (actC && xc0 < CX) {
#pragma unroll
                for (int x = 0; x < XC; ++x)
                    if (xc0 + x < CX) sB[((xc0 + x) * CY + yc) * CZ + zc] = rk[x];
            }
            __syncthreads();
            constexpr int QX = TX / 2, QY = TY / 2, QZ = TZ / 2;
            for (int t = grp * 256 + col; t < QX * QY * QZ; t += 256 * NG) {
                int X, Y, Z;
                region_cell<QY, QZ>(t, X, Y, Z);
                const int I = i0 / 2 + X, J = j0 / 2 + Y, K = k0 / 2 + Z;
                if (!live || !in_level(mc, I, J, K)) continue;
                const T* p = sB + ((2 * X) * CY + 2 * Y) * CZ + 2 * Z;
                // (x pairs first, then y, then z)
                const T acc = ((p[0] + p[CY * CZ]) + (p[CZ] + p[CY * CZ + CZ])) + ((p[1] + p[CY * CZ + 1]) + (p[CZ + 1] + p[CY * CZ + CZ + 1]));
                IO::st(fc + mc.at(I, J, K), acc);
            }
        }
    }
    if (RESTRICT) {
        __syncthreads();
        constexpr int QX = TX / 2, QY = TY / 2, QZ = TZ / 2;
        for (int t = grp * 256 + col; t < QX * QY * QZ; t += 256 * NG) {
            int X, Y, Z;
            region_cell<QY, QZ>(t, X, Y, Z);
            const int I = i0 / 2 + X, J = j0 / 2 + Y, K = k0 / 2 + Z;
            if (!live || !in_level(mc, I, J, K)) continue;
            const size_t C = mc.at(I, J, K);
            if (!cnt_c[C]) continue;
            auto w = [](int a) { return (a == 0 || a == 3) ? (T)0.25 : (T)0.75; };
            const T* p = sR + ((2 * X) * CY + 2 * Y) * CZ + 2 * Z;  // region C starts one fine cell before the tile
            T acc = 0;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const T* q = p + (a * CY + b) * CZ;
                    acc += w(a) * w(b) * ((T)0.25 * (q[0] + q[3]) + (T)0.75 * (q[1] + q[2]));
                }
            }
            IO::st(fc + C, acc * (T)0.125);
        }
    }
}

template <typename T, typename F, int TX, int TY, int TZ, bool RESTRICT, typename IO, int NG = 1>
__device__ __forceinline__ void mg_down_body(const MLevel& m, const uint8_t* __restrict__ cnt, const F* __restrict__ f, T* __restrict__ u,
                                             T* __restrict__ r, const MLevel& mc, const uint8_t* __restrict__ cnt_c, T* __restrict__ fc,
                                             const MgCoef<T>* cf, T off, int tile, int gx, int gy, bool live, int col, char* lds, T* sd, T* si,
                                             int grp = 0)
{
    mg_down_body_fn<T, F, TX, TY, TZ, RESTRICT, IO, NG>(m, cnt, u, r, mc, cnt_c, fc, cf, off, tile, gx, gy, live, col, lds, sd, si, grp,
                                                        [&](size_t q) { return IO::ld(f + q); }, [](size_t) {});
}

template <typename T, typename F, int TX, int TY, int TZ, bool RESTRICT, int NG = 1>
__global__ __launch_bounds__(256 * NG) void k_mg_down(MLevel m, const uint8_t* __restrict__ cnt, const F* __restrict__ f, T* __restrict__ u,
                                                      T* __restrict__ r, MLevel mc, const uint8_t* __restrict__ cnt_c, T* __restrict__ fc,
                                                      MgCoef<T> cf, const PcgState* ps, int gx, int gy, const int* __restrict__ tlist)
{
    __shared__ __attribute__((aligned(16))) char lds[DownTile<T, TX, TY, TZ, RESTRICT>::bytes];
    __shared__ T sd[8], si[8];
    if (ps && ps->done) return;
    // 1-D launch: virtual tile ids are dealt so that each XCD (own L2) gets a contiguous run of tiles, z fastest;
    // tlist (mostly-air box): only the tiles that hold an unknown are launched, in ascending order (k_mg_tile_flags)
    const int tile = tlist ? tlist[xcd_remap(blockIdx.x, gridDim.x)] : xcd_remap(blockIdx.x, gridDim.x);
    mg_down_body<T, F, TX, TY, TZ, RESTRICT, IoPlain, NG>(m, cnt, f, u, r, mc, cnt_c, fc, &cf, cf.off, tile, gx, gy, true, NG > 1 ? (int)(threadIdx.x & 255) : (int)threadIdx.x,
                                                          lds, sd, si, NG > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0);
}

// One tile of the up leg; see mg_down_body for col / grp / lds / sd / si / cf / live.  red: 4 doubles (part_dot only, NG = 1).
template <typename T, typename F, typename O, int TX, int TY, int TZ, typename IO, int NG = 1>
__device__ __forceinline__ void mg_up_body(const MLevel& m, const uint8_t* __restrict__ cnt, const F* __restrict__ f, const T* __restrict__ u,
                                           O* __restrict__ out, const MLevel& mc, const T* __restrict__ ec, const MgCoef<T>* cf, T off,
                                           double* __restrict__ part_dot, int tile, int gx, int gy, T wc, const uint8_t* __restrict__ own, bool live,
                                           int col, char* lds, T* sd, T* si, double* red, int grp = 0, bool pconst = false)
{
    // pconst: the correction of a fine cell is its PARENT's value (piecewise-constant prolongation: the Galerkin coarse levels of kernels_gal.hip)
    // own (decomposed run, level 0): the PCG's count bytes — the partial f.out counts the rank's owned unknowns only
    // (non-zero byte without bit 7); the result itself is written on every unknown of the local box
    typedef UpTile<T, TX, TY, TZ> D;
    constexpr int AX = D::AX, AY = D::AY, AZ = D::AZ, BX = D::BX, BY = D::BY, BZ = D::BZ, EX = D::EX, EY = D::EY, EZ = D::EZ;
    constexpr int NE = (EX * EY * EZ + 256 * NG - 1) / (256 * NG);
    constexpr int PA = (AX + NG - 1) / NG, PB = (BX + NG - 1) / NG;
    constexpr int NHT = NG * (256 / (TY * TZ)), XT = (TX + NHT - 1) / NHT;   // NHT threads (over all groups) share a tile column, XT planes each
    static_assert(AY * AZ <= 256 && 256 % (TY * TZ) == 0 && TX % 2 == 0 && TY % 2 == 0 && TZ % 2 == 0, "tile shape");
    T* sA = (T*)(lds + D::oA);
    T* sB = (T*)(lds + D::oB);
    T* sE = (T*)(lds + D::oE);
    uint8_t* sC = (uint8_t*)(lds + D::oC);
    const int tbx = tile % gx, tby = (tile / gx) % gy, tbz = tile / (gx * gy);
    const int i0 = tbz * TX, j0 = tby * TY, k0 = tbx * TZ;
    const int I0 = i0 / 2 - 2, J0 = j0 / 2 - 2, K0 = k0 / 2 - 2;
    const long sx = m.sx;
    const int tid = grp * 256 + col;
    // ---- every global load of the block ----
    T ee[NE];
#pragma unroll
    for (int it = 0; it < NE; ++it) {
        int x, y, z;
        region_cell<EY, EZ>(tid + 256 * NG * it, x, y, z);
        const int I = I0 + x, J = J0 + y, K = K0 + z;
        // the coarse arrays carry a ring of zeros (indices -1 and d*), nothing beyond it
        const bool ok = I >= -1 && I <= mc.dx && J >= -1 && J <= mc.dy && K >= -1 && K <= mc.dz;
        const T v = IO::ld(ec + mc.at(min(max(I, -1), mc.dx), min(max(J, -1), mc.dy), min(max(K, -1), mc.dz)));
        ee[it] = ok ? v : (T)0;
    }
    const int ya = col / AZ, za = col - ya * AZ;
    const bool actA = col < AY * AZ;
    const bool okA = actA && (unsigned)(j0 - 2 + ya) < (unsigned)m.dy && (unsigned)(k0 - 2 + za) < (unsigned)m.dz;
    const size_t qa = m.at(0, clampi(j0 - 2 + ya, m.dy - 1), clampi(k0 - 2 + za, m.dz - 1));
    const int xa0 = grp * PA;
    int ca[PA];
    T ua[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int x = xa0 + p, i = i0 - 2 + x;
        const size_t q = qa + (size_t)((long)clampi(i, m.dx - 1) * sx);
        const int c = cnt[q];
        ua[p] = IO::ld(u + q);
        ca[p] = (okA && (NG == 1 || x < AX) && (unsigned)i < (unsigned)m.dx) ? c : 0;
    }
    const int yb = col / BZ, zb = col - yb * BZ;
    const bool actB = col < BY * BZ;
    const size_t qb = m.at(0, clampi(j0 - 1 + yb, m.dy - 1), clampi(k0 - 1 + zb, m.dz - 1));
    const int xb0 = grp * PB;
    T fb[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) fb[p] = (T)IO::ld(f + qb + (size_t)((long)clampi(i0 - 1 + xb0 + p, m.dx - 1) * sx));
    const int hl = col / (TY * TZ), ct = col - hl * (TY * TZ);
    const int ht = NG == 1 ? hl : grp * (256 / (TY * TZ)) + hl;
    const int yt = ct / TZ, zt = ct - yt * TZ;
    const int xt0 = ht * XT;
    const size_t qt = m.at(0, clampi(j0 + yt, m.dy - 1), clampi(k0 + zt, m.dz - 1));
    F ft[XT];  // the rhs in its own precision: level 0 dots it with the result (r.z of PCG)
#pragma unroll
    for (int x = 0; x < XT; ++x) ft[x] = IO::ld(f + qt + (size_t)((long)clampi(i0 + xt0 + x, m.dx - 1) * sx));
    if (cf) mg_load_coef(sd, si, *cf);
#pragma unroll
    for (int it = 0; it < NE; ++it) {
        const int t = tid + 256 * NG * it;
        if (t < EX * EY * EZ) sE[t] = ee[it];
    }
    __syncthreads();
    // ---- v0 = u + P e on region A: bilinear in (y,z) once per coarse plane, then linear in x (region A starts at
    //      the even cell i0 - 2 and sE two coarse cells before i0 / 2: fine plane x sits over coarse plane (x>>1)+1) ----
    if (actA) {
        const int eb = ((ya >> 1) + 1) * EZ + (za >> 1) + 1;
        const int sy = (ya & 1) ? EZ : -EZ, sz = (za & 1) ? 1 : -1;
        const T a = (T)0.75, b = (T)0.25;
        T pl[EX];
#pragma unroll
        for (int I = 0; I < EX; ++I) {
            const T* p = sE + I * EY * EZ + eb;
            pl[I] = pconst ? p[0] : a * a * p[0] + a * b * (p[sy] + p[sz]) + b * b * p[sy + sz];
        }
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int x = xa0 + p;
            if (NG == 1 || x < AX) {
                const int I = (x >> 1) + 1, In = (x & 1) ? I + 1 : I - 1;
                // (NG > 1: x is not a compile-time constant; the planes are picked by a short select chain over the EX registers)
                T pI = pl[0], pN = pl[0];
                if constexpr (NG == 1) {
                    pI = pl[(p >> 1) + 1], pN = pl[(p & 1) ? (p >> 1) + 2 : (p >> 1)];   // x = p: compile-time planes
                } else {
#pragma unroll
                    for (int e = 1; e < EX; ++e) { pI = e == I ? pl[e] : pI; pN = e == In ? pl[e] : pN; }
                }
                const T pe = pconst ? pI : a * pI + b * pN;
                sC[x * AY * AZ + col] = (uint8_t)ca[p];
                sA[x * AY * AZ + col] = ca[p] ? ua[p] + wc * pe : (T)0;
            }
        }
    }
    __syncthreads();
    // ---- first post-sweep on region B (branch-free: n = 0 has v0 = 0 and si[0] = 0) ----
    if (actB && (NG == 1 || xb0 < BX)) {
        const int a0 = (yb + 1) * AZ + zb + 1 + xb0 * AY * AZ;
        T cm = sA[a0], c0 = sA[a0 + AY * AZ];
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int x = xb0 + p;
            if (NG == 1 || x < BX) {
                const int a = a0 + (p + 1) * AY * AZ;
                const T cp = sA[a + AY * AZ];
                const int n = sC[a];
                const T nb = cm + cp + sA[a - AZ] + sA[a + AZ] + sA[a - 1] + sA[a + 1];
                sB[(x * BY + yb) * BZ + zb] = c0 + (T)MG_W2 * si[n] * (fb[p] - (sd[n] * c0 + off * nb));
                cm = c0;
                c0 = cp;
            }
        }
    }
    __syncthreads();
    // ---- second post-sweep on the tile, result + partial of f.out ----
    double acc = 0;
    if (NG == 1 || xt0 < TX) {
        const int b0 = ((xt0 + 1) * BY + yt + 1) * BZ + zt + 1;
        T cm = sB[b0 - BY * BZ], c0 = sB[b0];
#pragma unroll
        for (int x = 0; x < XT; ++x) {
            if (NG == 1 || xt0 + x < TX) {
                const int b = b0 + x * BY * BZ;
                const T cp = sB[b + BY * BZ];
                const int n = sC[((xt0 + x + 2) * AY + yt + 2) * AZ + zt + 2];
                const T nb = cm + cp + sB[b - BZ] + sB[b + BZ] + sB[b - 1] + sB[b + 1];
                const T o = c0 + (T)MG_W1 * si[n] * ((T)ft[x] - (sd[n] * c0 + off * nb));
                if (n && live) {
                    const size_t qo = qt + (size_t)((long)(i0 + xt0 + x) * sx);
                    IO::st(out + qo, (O)o);  // n != 0: in the level, no clamp
                    if (!own || (own[qo] && !(own[qo] & 0x80))) acc += (double)ft[x] * (double)o;
                }
                cm = c0;
                c0 = cp;
            }
        }
    }
    if (NG == 1 && part_dot) {
        acc = block_sum<double, 4>(acc, red);
        if (threadIdx.x == 0) part_dot[blockIdx.x] = acc;
    }
}

template <typename T, typename F, typename O, int TX, int TY, int TZ>
__global__ __launch_bounds__(256, 5) void k_mg_up(MLevel m, const uint8_t* __restrict__ cnt, const F* __restrict__ f, const T* __restrict__ u,
                                                                      O* __restrict__ out, MLevel mc, const T* __restrict__ ec, MgCoef<T> cf,
                                                                      double* __restrict__ part_dot, const PcgState* ps, int gx, int gy, T wc,
                                                                      const int* __restrict__ tlist, const uint8_t* __restrict__ own, int pconst)
{
    __shared__ __attribute__((aligned(16))) char lds[UpTile<T, TX, TY, TZ>::bytes];
    __shared__ T sd[8], si[8];
    __shared__ double red[4];
    if (ps && ps->done) return;
    const int tile = tlist ? tlist[xcd_remap(blockIdx.x, gridDim.x)] : xcd_remap(blockIdx.x, gridDim.x);  // see k_mg_down
    mg_up_body<T, F, O, TX, TY, TZ, IoPlain, 1>(m, cnt, f, u, out, mc, ec, &cf, cf.off, part_dot, tile, gx, gy, wc, own, true, (int)threadIdx.x, lds, sd, si, red, 0,
                                                pconst != 0);
}

// ---- restriction of the level-0 residual (its down kernel has no room for a halo of 3) -----------------
template <typename T>
__global__ __launch_bounds__(256) void k_mg_restrict(MLevel mf, const T* __restrict__ rf, MLevel mc, const uint8_t* __restrict__ cnt_c,
                                                     T* __restrict__ fc, const PcgState* ps)
{
    if (ps && ps->done) return;
    d_restrict<T>(mf, rf, mc, cnt_c, fc, (long)blockIdx.x * 256 + threadIdx.x);
}

// ---- tail: the whole sub-V-cycle of the small levels in ONE block, entirely in LDS ------------------------
// Levels of a few thousand cells are pure latency as separate kernels.  One block of 1024 threads loads the rhs of the
// first tail level (restricted by the down kernel of the level above), walks down and up through the tail levels with
// __syncthreads() between the stages — u, v(=r), f and the count bytes of every tail level stay in LDS (compact
// layout, ring of zeros) — and writes only the correction of its first level back.  The coarsest level (<= 8^3) is
// solved by symmetric red-black Gauss-Seidel, forward then reversed.
// One CU runs all of it, so the kernel is bound by its instruction count: every thread works out its cells ONCE (the
// first tail level has up to MG_TAIL_Q0 cells per thread, the others one: LDS index, count, parity, the coarse cell
// under it, the fine block above it) and the ~20 stages are then a handful of LDS reads and flops each; the stage
// bodies are branch-free (a non-unknown has n = 0, si[0] = sd[0] = 0 and holds 0 everywhere).
constexpr int MG_TAIL_Q0 = 4;
template <typename T>
struct MgTail {
    int nl;                          // levels in the tail; the last is the coarsest
    MLevel mg[MG_TAIL_MAX];          // global layouts: counts in, f0 in, u0 out
    const uint8_t* cnt[MG_TAIL_MAX];
    const T* f0;
    T* u0;
    MLevel ml[MG_TAIL_MAX];          // LDS layouts (ring: one cell below, two above)
    int o_u[MG_TAIL_MAX], o_v[MG_TAIL_MAX], o_f[MG_TAIL_MAX], o_c[MG_TAIL_MAX];  // LDS byte offsets
    int lds_bytes;
    T off[MG_TAIL_MAX];              // off-diagonal of each level; diag = -off * n
    int sweeps;
    double wc;                       // weight of the coarse corrections inside the tail
};

// what a thread knows about one of its cells
struct TailCell {
    int c;      // LDS index in its level (-1: no cell)
    int n;      // count byte
    int up;     // LDS index (coarser level) of the coarse cell under it
    int down;   // LDS index (finer level) of the corner (2I-1, 2J-1, 2K-1) of its 4x4x4 restriction block
    int bits;   // bit 0-2: i,j,k odd; bit 3: red
};
__device__ __forceinline__ TailCell tail_cell(const MLevel& m, bool has_coarser, const MLevel& coarser, bool has_finer, const MLevel& finer, int t)
{
    TailCell d;
    d.c = -1; d.n = 0; d.up = 0; d.down = 0; d.bits = 0;
    int i, j, k;
    if (mg_cell(m, t, i, j, k)) {
        d.c = (int)m.at(i, j, k);
        d.bits = (i & 1) | ((j & 1) << 1) | ((k & 1) << 2) | ((((i + j + k) & 1) == 0) << 3);
        if (has_coarser) d.up = (int)coarser.at(i >> 1, j >> 1, k >> 1);
        if (has_finer) d.down = (int)finer.at(2 * i - 1, 2 * j - 1, 2 * k - 1);
    }
    return d;
}
template <typename T>
__device__ __forceinline__ T tail_nb(const T* p, int c, int sx, int sy)
{
    return p[c - sx] + p[c + sx] + p[c - sy] + p[c + sy] + p[c - 1] + p[c + 1];
}

// the whole tail by the 1024 threads of one block; tail_lds: a.lds_bytes of LDS, sd / si: [MG_TAIL_MAX][8] in LDS
template <typename T, typename IO>
__device__ __forceinline__ void mg_tail_body(const MgTail<T>& a, double* tail_lds, T (*sd)[8], T (*si)[8], int tid)
{
    char* base = (char*)tail_lds;
#define TU(lv) ((T*)(base + a.o_u[lv]))
#define TV(lv) ((T*)(base + a.o_v[lv]))
#define TF(lv) ((T*)(base + a.o_f[lv]))
#define TC(lv) ((uint8_t*)(base + a.o_c[lv]))
    // ---- cells of this thread, counts and the rhs of the first level (all global loads up front) ----
    TailCell z[MG_TAIL_Q0];   // first tail level
    T zf[MG_TAIL_Q0];
    int zg[MG_TAIL_Q0];
#pragma unroll
    for (int q = 0; q < MG_TAIL_Q0; ++q) {
        z[q] = tail_cell(a.ml[0], a.nl > 1, a.ml[1], false, a.ml[0], tid + 1024 * q);
        zg[q] = 0;
        zf[q] = 0;
        if (z[q].c >= 0) {
            int i, j, k;
            mg_cell(a.mg[0], tid + 1024 * q, i, j, k);
            zg[q] = (int)a.mg[0].at(i, j, k);
            z[q].n = a.cnt[0][zg[q]];
            zf[q] = IO::ld(a.f0 + zg[q]);
        }
    }
    TailCell w[MG_TAIL_MAX];  // deeper levels: one cell per thread (w[0] unused)
#pragma unroll
    for (int l = 1; l < MG_TAIL_MAX; ++l) {
        w[l].c = -1; w[l].n = 0; w[l].up = 0; w[l].down = 0; w[l].bits = 0;
        if (l < a.nl) {
            w[l] = tail_cell(a.ml[l], l + 1 < a.nl, a.ml[l + 1 < MG_TAIL_MAX ? l + 1 : l], true, a.ml[l - 1], tid);
            if (w[l].c >= 0) {
                int i, j, k;
                mg_cell(a.mg[l], tid, i, j, k);
                w[l].n = a.cnt[l][a.mg[l].at(i, j, k)];
            }
        }
    }
    if (tid < a.nl * 8) {
        const int l = tid >> 3, n = tid & 7;
        T off = a.off[0];
#pragma unroll
        for (int q = 1; q < MG_TAIL_MAX; ++q) off = (l == q) ? a.off[q] : off;  // static indices into the kernarg
        const T d = -off * (T)n;
        sd[l][n] = d;
        si[l][n] = n ? (T)1 / d : (T)0;
    }
    for (int q = tid; q < a.lds_bytes / 8; q += 1024) tail_lds[q] = 0;  // rings and non-unknowns read 0 / "not an unknown"
    __syncthreads();
#pragma unroll
    for (int q = 0; q < MG_TAIL_Q0; ++q) {
        if (z[q].c >= 0) {
            TC(0)[z[q].c] = (uint8_t)z[q].n;
            TF(0)[z[q].c] = z[q].n ? zf[q] : (T)0;
        }
    }
#pragma unroll
    for (int l = 1; l < MG_TAIL_MAX; ++l)
        if (l < a.nl && w[l].c >= 0) TC(l)[w[l].c] = (uint8_t)w[l].n;
    __syncthreads();
    const T w1 = (T)MG_W1, w2 = (T)MG_W2;
    // OWN(l, body): run body(cell) for each cell this thread owns at level l
#define OWN(lv, ...)                                                                   \
    do {                                                                               \
        if ((lv) == 0) {                                                               \
            _Pragma("unroll") for (int q_ = 0; q_ < MG_TAIL_Q0; ++q_) {                \
                const TailCell& cell = z[q_];                                          \
                if (cell.c >= 0) { __VA_ARGS__ }                                       \
            }                                                                          \
        } else {                                                                       \
            const TailCell& cell = w[(lv) ? (lv) : 1];                                 \
            if (cell.c >= 0) { __VA_ARGS__ }                                           \
        }                                                                              \
    } while (0)
    // ---- down: two pre-sweeps, residual, restriction ----
#pragma unroll
    for (int l = 0; l < MG_TAIL_MAX - 1; ++l) {
        if (l < a.nl - 1) {
            const int sx = (int)a.ml[l].sx, sy = (int)a.ml[l].sy;
            T *U = TU(l), *V = TV(l), *F = TF(l);
            const T off = a.off[l];
            OWN(l, V[cell.c] = w1 * si[l][cell.n] * F[cell.c];);
            __syncthreads();
            OWN(l, const T v = V[cell.c]; U[cell.c] = v + w2 * si[l][cell.n] * (F[cell.c] - (sd[l][cell.n] * v + off * tail_nb(V, cell.c, sx, sy))););
            __syncthreads();
            OWN(l, V[cell.c] = cell.n ? F[cell.c] - (sd[l][cell.n] * U[cell.c] + off * tail_nb(U, cell.c, sx, sy)) : (T)0;);
            __syncthreads();
            {
                const TailCell& cc = w[l + 1];  // f_c = (1/8) P^T r: 4x4x4 gather, weights (1/4,3/4,3/4,1/4) per axis
                if (cc.c >= 0 && cc.n) {
                    auto wt = [](int q) { return (q == 0 || q == 3) ? (T)0.25 : (T)0.75; };
                    const T* p = V + cc.down;
                    T acc = 0;
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
#pragma unroll
                        for (int y = 0; y < 4; ++y) {
                            const T* q = p + x * sx + y * sy;
                            acc += wt(x) * wt(y) * ((T)0.25 * (q[0] + q[3]) + (T)0.75 * (q[1] + q[2]));
                        }
                    }
                    TF(l + 1)[cc.c] = acc * (T)0.125;
                }
            }
            __syncthreads();
        }
    }
    // ---- coarsest: red-black Gauss-Seidel in place (u starts at 0), forward sweeps then reversed (symmetric) ----
#pragma unroll
    for (int lc = 0; lc < MG_TAIL_MAX; ++lc) {
        if (lc == a.nl - 1) {
            const TailCell& cell = lc == 0 ? z[0] : w[lc ? lc : 1];  // <= 8^3 cells: one per thread
            T* su = TU(lc);
            const int sx = (int)a.ml[lc].sx, sy = (int)a.ml[lc].sy;
            const bool mine = cell.c >= 0 && cell.n;
            const bool isred = (cell.bits & 8) != 0;
            const T fv = mine ? TF(lc)[cell.c] : (T)0, inv = si[lc][cell.n], off = a.off[lc];
            for (int s = 0; s < 2 * a.sweeps; ++s) {
                const bool fwd = s < a.sweeps;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bool col = fwd ? (h == 0) : (h == 1);
                    if (mine && isred == col) su[cell.c] = (fv - off * tail_nb(su, cell.c, sx, sy)) * inv;
                    __syncthreads();
                }
            }
        }
    }
    // ---- up: prolongation, two post-sweeps ----
#pragma unroll
    for (int l = MG_TAIL_MAX - 2; l >= 0; --l) {
        if (l < a.nl - 1) {
            const int sx = (int)a.ml[l].sx, sy = (int)a.ml[l].sy;
            const int cx = (int)a.ml[l + 1].sx, cy = (int)a.ml[l + 1].sy;
            T *U = TU(l), *V = TV(l), *F = TF(l);
            const T* E = TU(l + 1);
            const T off = a.off[l];
            OWN(l, {
                const int ex = (cell.bits & 1) ? cx : -cx, ey = (cell.bits & 2) ? cy : -cy, ez = (cell.bits & 4) ? 1 : -1;
                const T* e = E + cell.up;
                const T p = (T)0.75, m = (T)0.25;
                const T pe = p * p * p * e[0] + p * p * m * (e[ex] + e[ey] + e[ez]) + p * m * m * (e[ex + ey] + e[ex + ez] + e[ey + ez]) +
                             m * m * m * e[ex + ey + ez];
                if (cell.n) U[cell.c] += (T)a.wc * pe;
            });
            __syncthreads();
            OWN(l, const T v = U[cell.c]; V[cell.c] = v + w2 * si[l][cell.n] * (F[cell.c] - (sd[l][cell.n] * v + off * tail_nb(U, cell.c, sx, sy))););
            __syncthreads();
            OWN(l, const T v = V[cell.c]; U[cell.c] = v + w1 * si[l][cell.n] * (F[cell.c] - (sd[l][cell.n] * v + off * tail_nb(V, cell.c, sx, sy))););
            __syncthreads();
        }
    }
#pragma unroll
    for (int q = 0; q < MG_TAIL_Q0; ++q)
        if (z[q].c >= 0 && z[q].n) IO::st(a.u0 + zg[q], TU(0)[z[q].c]);
#undef OWN
#undef TU
#undef TV
#undef TF
#undef TC
}

template <typename T>
__global__ __launch_bounds__(1024) void k_mg_tail(MgTail<T> a, const PcgState* ps)
{
    extern __shared__ double tail_lds[];
    __shared__ T sd[MG_TAIL_MAX][8], si[MG_TAIL_MAX][8];
    if (ps && ps->done) return;
    mg_tail_body<T, IoPlain>(a, tail_lds, sd, si, threadIdx.x);
}

// ---- launchers ----------------------------------------------------------------------------------
static inline unsigned mg_blocks(const MLevel& m) { return (unsigned)(((long)m.dx * m.dy * m.dz + 255) / 256); }

MLevel mg_level0(const LBox& L)
{
    MLevel m;
    m.dx = L.nx + 2; m.dy = L.ny + 2; m.dz = L.nz + 2;
    m.sx = (long)L.Ly * L.Lz; m.sy = L.Lz;
    m.ox = 0; m.oy = 0; m.oz = LBOX_K0 - 1;
    m.cells = L.cells() + 2 * (size_t)L.Lz;
    return m;
}
MLevel mg_coarser(const MLevel& f)
{
    MLevel m;
    m.dx = (f.dx + 1) / 2; m.dy = (f.dy + 1) / 2; m.dz = (f.dz + 1) / 2;
    const int Lz = (16 + m.dz + 1 + 15) / 16 * 16;
    m.sy = Lz; m.sx = (long)(m.dy + 2) * Lz;
    m.ox = 1; m.oy = 1; m.oz = 16;
    m.cells = (size_t)(m.dx + 2) * m.sx + Lz;
    return m;
}

void launch_mg_type0(hipStream_t st, Grid g, LBox L, MLevel m, const uint8_t* flags, const uint8_t* cnt, uint8_t* typ)
{
    hipLaunchKernelGGL(k_mg_type0, dim3(mg_blocks(m)), dim3(256), 0, st, g, L, m, flags, cnt, typ);
}
void launch_mg_coarsen_types(hipStream_t st, MLevel mf, const uint8_t* tf, MLevel mc, uint8_t* tc)
{
    if (mg_blocks(mc)) hipLaunchKernelGGL(k_mg_coarsen, dim3(mg_blocks(mc)), dim3(256), 0, st, mf, tf, mc, tc);
}
void launch_mg_counts(hipStream_t st, MLevel m, const uint8_t* typ, uint8_t* cnt)
{
    if (mg_blocks(m)) hipLaunchKernelGGL(k_mg_cnt, dim3(mg_blocks(m)), dim3(256), 0, st, m, typ, cnt);
}
void launch_mg_coarsen(hipStream_t st, MLevel mf, const uint8_t* tf, MLevel mc, uint8_t* tc, uint8_t* cnt_c)
{
    hipLaunchKernelGGL(k_mg_coarsen, dim3(mg_blocks(mc)), dim3(256), 0, st, mf, tf, mc, tc);
    hipLaunchKernelGGL(k_mg_cnt, dim3(mg_blocks(mc)), dim3(256), 0, st, mc, (const uint8_t*)tc, cnt_c);
}
template <typename T>
void launch_mg_restrict(hipStream_t st, MLevel mf, const T* rf, MLevel mc, const uint8_t* cnt_c, T* fc, const PcgState* ps)
{
    if (mg_blocks(mc)) hipLaunchKernelGGL((k_mg_restrict<T>), dim3(mg_blocks(mc)), dim3(256), 0, st, mf, rf, mc, cnt_c, fc, ps);
}
// tile shapes of the LDS-tiled legs
constexpr int MG_TX = 8, MG_TY = 8, MG_TZ = 16;   // down (no restriction) and up
constexpr int MG_RX = 8, MG_RY = 8, MG_RZ = 8;    // down with the restriction folded in (halo 3)
constexpr unsigned MG_FEW_TILES = 384;            // up to this many tiles a leg takes four thread groups per tile (FLUID_MG_NG4=0: never)
// flags[t] = tile t of the level-0 legs (MG_TX x MG_TY x MG_TZ cells, numbered as k_mg_down / k_mg_up decode them) holds an unknown
template <int TX, int TY, int TZ>
__global__ __launch_bounds__(256) void k_mg_tile_flags(MLevel m, const uint8_t* __restrict__ cnt, int gx, int gy, uint8_t* __restrict__ flags)
{
    const int tile = blockIdx.x;
    const int tbx = tile % gx, tby = (tile / gx) % gy, tbz = tile / (gx * gy);
    const int i0 = tbz * TX, j0 = tby * TY, k0 = tbx * TZ;
    int any = 0;
    for (int t = threadIdx.x; t < TX * TY * TZ; t += 256) {
        const int z = t % TZ, y = (t / TZ) % TY, x = t / (TZ * TY);
        const int i = i0 + x, j = j0 + y, k = k0 + z;
        if (i < m.dx && j < m.dy && k < m.dz) any |= cnt[m.at(i, j, k)];
    }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) flags[tile] = any != 0;
}

static inline bool mg_ng4() { return true; }   // a tile of a small level is worked by four thread groups
static inline dim3 mg_tiles(const MLevel& m, int tx, int ty, int tz)
{
    return dim3((unsigned)((m.dz + tz - 1) / tz), (unsigned)((m.dy + ty - 1) / ty), (unsigned)((m.dx + tx - 1) / tx));
}
int mg_up_blocks(const MLevel& m)
{
    const dim3 g = mg_tiles(m, MG_TX, MG_TY, MG_TZ);
    return (int)(g.x * g.y * g.z);
}
// T = the V-cycle's own arithmetic and storage type (u, r, coarse rhs, LDS); F = element type of this level's rhs and
// O of its result: at level 0 both are the PCG's double vectors whatever T is, on the other levels they are T.
// both pre-sweeps + residual; with a coarse level (fc != nullptr) the restricted residual goes straight to fc and r is not written
template <typename T, typename F>
void launch_mg_down(hipStream_t st, MLevel m, const uint8_t* cnt, const F* f, T* u, T* r, MLevel mc, const uint8_t* cnt_c, T* fc, MgCoef<T> cf,
                    const PcgState* ps, const int* tlist, int nlist, bool pcr)
{
    // pcr: fc is the coarse right-hand side of the Galerkin levels (sum of the children): the leg WITHOUT the trilinear restriction forms it
    if (m.dx <= 0 || m.dy <= 0 || m.dz <= 0) return;   // an empty local level (decomposed run)
    if (fc && !pcr) {
        const dim3 g = mg_tiles(m, MG_RX, MG_RY, MG_RZ);
        // few tiles (about one per CU or fewer): the leg is bound by one tile's chain of dependent stages — four groups of 256 threads share it
        if (sizeof(F) == sizeof(T) && g.x * g.y * g.z <= MG_FEW_TILES && mg_ng4())
            hipLaunchKernelGGL((k_mg_down<T, F, MG_RX, MG_RY, MG_RZ, true, 4>), dim3(g.x * g.y * g.z), dim3(1024), 0, st, m, cnt, f, u, r, mc, cnt_c, fc, cf,
                               ps, (int)g.x, (int)g.y, (const int*)nullptr);
        else
            hipLaunchKernelGGL((k_mg_down<T, F, MG_RX, MG_RY, MG_RZ, true>), dim3(g.x * g.y * g.z), dim3(256), 0, st, m, cnt, f, u, r, mc, cnt_c, fc, cf,
                               ps, (int)g.x, (int)g.y, (const int*)nullptr);
    } else {
        const dim3 g = mg_tiles(m, MG_TX, MG_TY, MG_TZ);
        hipLaunchKernelGGL((k_mg_down<T, F, MG_TX, MG_TY, MG_TZ, false>), dim3(tlist ? (unsigned)nlist : g.x * g.y * g.z), dim3(256), 0, st, m, cnt, f, u,
                           r, mc, cnt_c, fc, cf, ps, (int)g.x, (int)g.y, tlist);
    }
}
// XR + the level-0 down leg (k_mg_down_xr): mg_up_blocks(m) blocks, as many |r|^2 partials
template <typename T, typename F, typename O>
void launch_mg_up(hipStream_t st, MLevel m, const uint8_t* cnt, const F* f, const T* u, O* out, MLevel mc, const T* ec, MgCoef<T> cf,
                  double* part_dot, const PcgState* ps, double wc, const int* tlist, int nlist, const uint8_t* own, int pconst)
{
    const dim3 g = mg_tiles(m, MG_TX, MG_TY, MG_TZ);
    if (!tlist && g.x * g.y * g.z == 0) return;   // an empty local level (decomposed run: a block outside the active box)
    // (a four-group form like the down leg's was measured at 256^3: up legs of levels 1-2 5.93 -> 6.13 us — a leg is ~3 us of launch and drain +
    // ~2 us of load latency, the arithmetic was never the long part — so only the down leg has one)
    hipLaunchKernelGGL((k_mg_up<T, F, O, MG_TX, MG_TY, MG_TZ>), dim3(tlist ? (unsigned)nlist : g.x * g.y * g.z), dim3(256), 0, st, m, cnt, f, u, out, mc,
                       ec, cf, part_dot, ps, (int)g.x, (int)g.y, (T)wc, tlist, own, pconst);
}
// flags of the level-0 leg tiles (mg_up_blocks(m) of them), see k_mg_tile_flags
void launch_mg_tile_flags(hipStream_t st, MLevel m, const uint8_t* cnt, uint8_t* flags)
{
    const dim3 g = mg_tiles(m, MG_TX, MG_TY, MG_TZ);
    hipLaunchKernelGGL((k_mg_tile_flags<MG_TX, MG_TY, MG_TZ>), dim3(g.x * g.y * g.z), dim3(256), 0, st, m, cnt, (int)g.x, (int)g.y, flags);
}

// LDS footprint of a tail that starts at lv[0] (compact arrays, ring of one cell below and two above — the 4x4x4
// restriction block of the last coarse cell of an odd-sized level reaches index d+1: u, v, f as T and one count byte)
static MLevel tail_lds_level(const MLevel& g)
{
    MLevel m;
    m.dx = g.dx; m.dy = g.dy; m.dz = g.dz;
    m.sy = g.dz + 3; m.sx = (long)(g.dy + 3) * (g.dz + 3);
    m.ox = m.oy = m.oz = 1;
    m.cells = (size_t)(g.dx + 3) * m.sx;
    return m;
}
// 0 if the levels cannot run in the tail kernel (too many cells per thread / level), else the LDS bytes they need
size_t mg_tail_lds_bytes(int nl, const MLevel* lv, size_t elem)
{
    size_t total = 0;
    for (int l = 0; l < nl; ++l) {
        const long cells = (long)lv[l].dx * lv[l].dy * lv[l].dz;
        if (cells > (l == 0 ? 1024L * MG_TAIL_Q0 : 1024L)) return 0;
        const size_t c = tail_lds_level(lv[l]).cells;
        total += 3 * ((c * elem + 15) / 16 * 16) + (c + 15) / 16 * 16;
    }
    return total;
}
// levels[0..nl) of the tail, f0 = rhs of lv[0] (global layout); result (the correction of lv[0]) in u0
template <typename T>
static MgTail<T> make_mg_tail(int nl, const T* f0, const MLevel* lv, uint8_t* const* cnt, T* u0, const T* off, int sweeps, double wc)
{
    MgTail<T> a;
    a.wc = wc;
    a.nl = nl;
    a.f0 = f0; a.u0 = u0;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 15) / 16 * 16; return (int)at; };
    for (int l = 0; l < MG_TAIL_MAX; ++l) {
        const int q = l < nl ? l : nl - 1;
        a.mg[l] = lv[q]; a.cnt[l] = cnt[q]; a.off[l] = off[q];
        a.ml[l] = tail_lds_level(lv[q]);
        if (l < nl) {
            const size_t c = a.ml[l].cells;
            a.o_u[l] = take(c * sizeof(T)); a.o_v[l] = take(c * sizeof(T)); a.o_f[l] = take(c * sizeof(T)); a.o_c[l] = take(c);
        } else {
            a.o_u[l] = a.o_u[q]; a.o_v[l] = a.o_v[q]; a.o_f[l] = a.o_f[q]; a.o_c[l] = a.o_c[q];
        }
    }
    a.lds_bytes = (int)o;
    a.sweeps = sweeps;
    return a;
}
template <typename T>
void launch_mg_tail(hipStream_t st, int nl, const T* f0, const MLevel* lv, uint8_t* const* cnt, T* u0, const T* off, int sweeps, const PcgState* ps,
                    double wc)
{
    const MgTail<T> a = make_mg_tail<T>(nl, f0, lv, cnt, u0, off, sweeps, wc);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)k_mg_tail<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MG_TAIL_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_mg_tail<T>), dim3(1), dim3(1024), (size_t)a.lds_bytes, st, a, ps);
}

#define INSTMG(T)                                                                                                                        \
    template void launch_mg_down<T, T>(hipStream_t, MLevel, const uint8_t*, const T*, T*, T*, MLevel, const uint8_t*, T*, MgCoef<T>, const PcgState*, \
                                       const int*, int, bool); \
    template void launch_mg_up<T, T, T>(hipStream_t, MLevel, const uint8_t*, const T*, const T*, T*, MLevel, const T*, MgCoef<T>, double*,  \
                                        const PcgState*, double, const int*, int, const uint8_t*, int);                                                                          \
    template void launch_mg_restrict<T>(hipStream_t, MLevel, const T*, MLevel, const uint8_t*, T*, const PcgState*);                     \
    template void launch_mg_tail<T>(hipStream_t, int, const T*, const MLevel*, uint8_t* const*, T*, const T*, int, const PcgState*, double);
INSTMG(double)
INSTMG(float)
// level 0 of a single-precision V-cycle inside the double-precision PCG
template void launch_mg_down<float, double>(hipStream_t, MLevel, const uint8_t*, const double*, float*, float*, MLevel, const uint8_t*, float*,
                                            MgCoef<float>, const PcgState*, const int*, int, bool);
template void launch_mg_up<float, double, double>(hipStream_t, MLevel, const uint8_t*, const double*, const float*, double*, MLevel, const float*,
                                                  MgCoef<float>, double*, const PcgState*, double, const int*, int, const uint8_t*, int);
template void launch_mg_up<float, double, float>(hipStream_t, MLevel, const uint8_t*, const double*, const float*, float*, MLevel, const float*,
                                                  MgCoef<float>, double*, const PcgState*, double, const int*, int, const uint8_t*, int);

}  // namespace fl
