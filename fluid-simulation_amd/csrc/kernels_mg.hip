// Geometric multigrid V-cycle used as the PCG preconditioner (z = M^-1 r) of the pressure solve.
//
// The reference preconditions with Eigen's IncompleteCholesky (fluid.cc:1352) — serial triangular
// solves.  Any SPD preconditioner leaves the converged solution of A p = b unchanged; this one makes
// the iteration count independent of the grid size (27-28 at 128^3 and 256^3 against 308 / 562 for
// Jacobi, tools/mg_prototype.py).  Structure (after McAdams, Sifakis, Teran 2010, restated):
//   * levels: cells coarsened 2x2x2; a coarse cell is AIR (Dirichlet, p=0) if any child is air,
//     SOLID (Neumann) if all children are solid, else FLUID; operator = the same 7-point form with
//     the off-diagonal divided by 4 per level, diagonal = (#non-solid neighbours) x |off|;
//   * smoother: damped Jacobi (omega 2/3), 2 sweeps before and 2 after (symmetric -> M is SPD);
//   * transfer: cell-centred trilinear prolongation P (weights 3/4,1/4 per axis), restriction P^T/8;
//   * coarsest level (<= 8^3): red-black Gauss-Seidel in LDS by one block, forward then reversed.
// Level 0 lives in the solver's box-local layout (LBox); coarser levels use the same indexing
// scheme (MLevel).  Only unknown cells are ever written (every array is zeroed once per step), so a mostly-air
// active box costs one count byte per air cell.  All kernels are one thread per cell with neighbours read through L1/L2: levels
// >= 1 are tiny and launch-bound; level 0 costs ~4 stencil sweeps per cycle.
#include "common.h"

namespace fl {

// Damped-Jacobi weights of the two sweeps on each side of the coarse correction: (W1, W2) before, (W2, W1) after
// (reversed, so M stays symmetric).  (2/3, 1.2) instead of (2/3, 2/3) takes ~15 % fewer PCG iterations in the
// prototype; |(1 - W1 x)(1 - W2 x)| < 1 on the spectrum (0,2) of D^-1 A, so the smoother still converges.
constexpr double MG_W1 = 2.0 / 3.0, MG_W2 = 1.2;

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
    if (gx >= 0 && gx < g.N && gy >= 0 && gy < g.N && gz >= 0 && gz < g.N) {
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

// ---- per-cell bodies (shared by the per-level kernels and the single-block tail kernel) ---------
// Coefficients by neighbour count n: diag = dg[n], 1/diag = iv[n] (tables in LDS), off-diagonal = off.

// damped-Jacobi sweep: u_out = u_in + omega D^-1 (f - A u_in); returns f*u_out
template <typename T>
__device__ __forceinline__ double d_smooth(const MLevel& m, const uint8_t* __restrict__ cnt, const T* __restrict__ f,
                                           const T* __restrict__ u_in, T* __restrict__ u_out, const T* dg, const T* iv, T off, T omega, long t)
{
    int i, j, k;
    if (!mg_cell(m, t, i, j, k)) return 0;
    const size_t c = m.at(i, j, k);
    const int n = cnt[c];
    if (!n) return 0;  // not an unknown: its entries stay 0 (arrays are zeroed once per step), no 8-byte store per air cell
    const T fv = f[c], uc = u_in[c];
    const T nb = u_in[c - m.sx] + u_in[c + m.sx] + u_in[c - m.sy] + u_in[c + m.sy] + u_in[c - 1] + u_in[c + 1];
    const T out = uc + omega * iv[n] * (fv - (dg[n] * uc + off * nb));
    u_out[c] = out;
    return (double)fv * (double)out;
}

// two sweeps starting from u = 0 in one pass: u1 = omega D^-1 f is formed on the fly at the 7 points
template <typename T>
__device__ __forceinline__ void d_smooth0(const MLevel& m, const uint8_t* __restrict__ cnt, const T* __restrict__ f, T* __restrict__ u_out,
                                          const T* dg, const T* iv, T off, long t)
{
    int i, j, k;
    if (!mg_cell(m, t, i, j, k)) return;
    const size_t c = m.at(i, j, k);
    const int n = cnt[c];
    if (!n) return;
    const T w1 = (T)MG_W1, w2 = (T)MG_W2;
    auto u1 = [&](size_t q) { return w1 * iv[cnt[q]] * f[q]; };  // iv[0] = 0: non-unknowns give 0
    const T fv = f[c], uc = w1 * iv[n] * fv;
    const T nb = u1(c - m.sx) + u1(c + m.sx) + u1(c - m.sy) + u1(c + m.sy) + u1(c - 1) + u1(c + 1);
    u_out[c] = uc + w2 * iv[n] * (fv - (dg[n] * uc + off * nb));
}

template <typename T>
__device__ __forceinline__ void d_resid(const MLevel& m, const uint8_t* __restrict__ cnt, const T* __restrict__ f, const T* __restrict__ u,
                                        T* __restrict__ r, const T* dg, T off, long t)
{
    int i, j, k;
    if (!mg_cell(m, t, i, j, k)) return;
    const size_t c = m.at(i, j, k);
    const int n = cnt[c];
    if (!n) return;
    const T nb = u[c - m.sx] + u[c + m.sx] + u[c - m.sy] + u[c + m.sy] + u[c - 1] + u[c + 1];
    r[c] = f[c] - (dg[n] * u[c] + off * nb);
}

// f_c = (1/8) P^T r_f : a coarse cell gathers its 4x4x4 fine neighbourhood, weights (1/4,3/4,3/4,1/4) per axis
template <typename T>
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
    fc[C] = out;
}

// u += P e : a fine cell interpolates from its 8 nearest coarse cells (non-unknown coarse cells hold 0)
template <typename T>
__device__ __forceinline__ void d_prolong(const MLevel& mf, const uint8_t* __restrict__ cnt_f, T* __restrict__ u, const MLevel& mc,
                                          const T* __restrict__ ec, long t)
{
    int i, j, k;
    if (!mg_cell(mf, t, i, j, k)) return;
    const size_t c = mf.at(i, j, k);
    if (!cnt_f[c]) return;
    const int I = i >> 1, J = j >> 1, K = k >> 1;
    const int di = (i & 1) ? 1 : -1, dj = (j & 1) ? 1 : -1, dk = (k & 1) ? 1 : -1;
    const size_t C = mc.at(I, J, K);  // the coarse arrays carry a ring of zeros: I+di etc. are always addressable
    const long sx = (long)di * mc.sx, sy = (long)dj * mc.sy, sz = dk;
    const T a = (T)0.75, b = (T)0.25;
    const T v = a * a * a * ec[C] + a * a * b * (ec[C + sx] + ec[C + sy] + ec[C + sz]) +
                a * b * b * (ec[C + sx + sy] + ec[C + sx + sz] + ec[C + sy + sz]) + b * b * b * ec[C + sx + sy + sz];
    u[c] += v;
}

// Fused forms for the launch-bound intermediate levels (a 100k-cell level costs ~4.5 us per launch whatever it does):
// (a) both pre-sweeps AND the residual in one pass: r = f - A u2 needs u2 at the 7 points, each of which needs
//     u1 = W1 D^-1 f at its own 7 points (footprint radius 2 on f); (b) prolongation folded into the first post-sweep.
template <typename T>
__device__ __forceinline__ T d_u2_at(const MLevel& m, const uint8_t* __restrict__ cnt, const T* __restrict__ f, const T* dg, const T* iv,
                                     T off, size_t q)
{
    const int n = cnt[q];
    if (!n) return (T)0;  // not an unknown (possibly a ring cell): its neighbours are never touched
    const T w1 = (T)MG_W1, w2 = (T)MG_W2;
    auto u1 = [&](size_t p) { return w1 * iv[cnt[p]] * f[p]; };
    const T fv = f[q], uc = w1 * iv[n] * fv;
    const T nb = u1(q - m.sx) + u1(q + m.sx) + u1(q - m.sy) + u1(q + m.sy) + u1(q - 1) + u1(q + 1);
    return uc + w2 * iv[n] * (fv - (dg[n] * uc + off * nb));
}
template <typename T>
__device__ __forceinline__ void d_smooth0_resid(const MLevel& m, const uint8_t* __restrict__ cnt, const T* __restrict__ f, T* __restrict__ u,
                                                T* __restrict__ r, const T* dg, const T* iv, T off, long t)
{
    int i, j, k;
    if (!mg_cell(m, t, i, j, k)) return;
    const size_t c = m.at(i, j, k);
    const int n = cnt[c];
    T uo = 0, ro = 0;
    if (!n) return;
    {
        uo = d_u2_at<T>(m, cnt, f, dg, iv, off, c);
        const T nb = d_u2_at<T>(m, cnt, f, dg, iv, off, c - m.sx) + d_u2_at<T>(m, cnt, f, dg, iv, off, c + m.sx) +
                     d_u2_at<T>(m, cnt, f, dg, iv, off, c - m.sy) + d_u2_at<T>(m, cnt, f, dg, iv, off, c + m.sy) +
                     d_u2_at<T>(m, cnt, f, dg, iv, off, c - 1) + d_u2_at<T>(m, cnt, f, dg, iv, off, c + 1);
        ro = f[c] - (dg[n] * uo + off * nb);
    }
    u[c] = uo;
    r[c] = ro;
}
// value of (u + P e) at fine array index q = at(i,j,k) (0 for a non-unknown)
template <typename T>
__device__ __forceinline__ T d_upe_at(const MLevel& mf, const uint8_t* __restrict__ cnt_f, const T* __restrict__ u, const MLevel& mc,
                                      const T* __restrict__ ec, int i, int j, int k)
{
    const size_t q = mf.at(i, j, k);
    if (!cnt_f[q]) return (T)0;
    const int I = i >> 1, J = j >> 1, K = k >> 1;
    const int di = (i & 1) ? 1 : -1, dj = (j & 1) ? 1 : -1, dk = (k & 1) ? 1 : -1;
    const size_t C = mc.at(I, J, K);
    const long sx = (long)di * mc.sx, sy = (long)dj * mc.sy, sz = dk;
    const T a = (T)0.75, b = (T)0.25;
    return u[q] + a * a * a * ec[C] + a * a * b * (ec[C + sx] + ec[C + sy] + ec[C + sz]) +
           a * b * b * (ec[C + sx + sy] + ec[C + sx + sz] + ec[C + sy + sz]) + b * b * b * ec[C + sx + sy + sz];
}
template <typename T>
__device__ __forceinline__ void d_prolong_smooth(const MLevel& mf, const uint8_t* __restrict__ cnt_f, const T* __restrict__ f,
                                                 const T* __restrict__ u, T* __restrict__ u_out, const MLevel& mc, const T* __restrict__ ec,
                                                 const T* dg, const T* iv, T off, long t)
{
    int i, j, k;
    if (!mg_cell(mf, t, i, j, k)) return;
    const size_t c = mf.at(i, j, k);
    const int n = cnt_f[c];
    T out = 0;
    if (!n) return;
    {
        const T vc = d_upe_at<T>(mf, cnt_f, u, mc, ec, i, j, k);
        const T nb = d_upe_at<T>(mf, cnt_f, u, mc, ec, i - 1, j, k) + d_upe_at<T>(mf, cnt_f, u, mc, ec, i + 1, j, k) +
                     d_upe_at<T>(mf, cnt_f, u, mc, ec, i, j - 1, k) + d_upe_at<T>(mf, cnt_f, u, mc, ec, i, j + 1, k) +
                     d_upe_at<T>(mf, cnt_f, u, mc, ec, i, j, k - 1) + d_upe_at<T>(mf, cnt_f, u, mc, ec, i, j, k + 1);
        out = vc + (T)MG_W2 * iv[n] * (f[c] - (dg[n] * vc + off * nb));
    }
    u_out[c] = out;
}

template <typename T>
__global__ __launch_bounds__(256) void k_mg_smooth0_resid(MLevel m, const uint8_t* __restrict__ cnt, const T* __restrict__ f, T* __restrict__ u,
                                                          T* __restrict__ r, MgCoef<T> cf, const PcgState* ps)
{
    __shared__ T sd[8], si[8];
    if (ps && ps->done) return;
    mg_load_coef(sd, si, cf);
    d_smooth0_resid<T>(m, cnt, f, u, r, sd, si, cf.off, (long)blockIdx.x * 256 + threadIdx.x);
}
template <typename T>
__global__ __launch_bounds__(256) void k_mg_prolong_smooth(MLevel mf, const uint8_t* __restrict__ cnt_f, const T* __restrict__ f,
                                                           const T* __restrict__ u, T* __restrict__ u_out, MLevel mc,
                                                           const T* __restrict__ ec, MgCoef<T> cf, const PcgState* ps)
{
    __shared__ T sd[8], si[8];
    if (ps && ps->done) return;
    mg_load_coef(sd, si, cf);
    d_prolong_smooth<T>(mf, cnt_f, f, u, u_out, mc, ec, sd, si, cf.off, (long)blockIdx.x * 256 + threadIdx.x);
}

// ---- per-level kernels (levels too large for one block) ---------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_mg_smooth(MLevel m, const uint8_t* __restrict__ cnt, const T* __restrict__ f, const T* __restrict__ u_in,
                                                   T* __restrict__ u_out, MgCoef<T> cf, T omega, double* __restrict__ part_dot, const PcgState* ps)
{
    __shared__ double red[4];
    __shared__ T sd[8], si[8];
    if (ps && ps->done) return;  // uniform: written by an earlier launch
    mg_load_coef(sd, si, cf);
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    double acc = 0;
    if (u_in) acc = d_smooth<T>(m, cnt, f, u_in, u_out, sd, si, cf.off, omega, t);
    else d_smooth0<T>(m, cnt, f, u_out, sd, si, cf.off, t);
    if (part_dot) {
        acc = block_sum<double, 4>(acc, red);
        if (threadIdx.x == 0) part_dot[blockIdx.x] = acc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_mg_resid(MLevel m, const uint8_t* __restrict__ cnt, const T* __restrict__ f, const T* __restrict__ u,
                                                  T* __restrict__ r, MgCoef<T> cf, const PcgState* ps)
{
    __shared__ T sd[8], si[8];
    if (ps && ps->done) return;
    mg_load_coef(sd, si, cf);
    d_resid<T>(m, cnt, f, u, r, sd, cf.off, (long)blockIdx.x * 256 + threadIdx.x);
}

template <typename T>
__global__ __launch_bounds__(256) void k_mg_restrict(MLevel mf, const T* __restrict__ rf, MLevel mc, const uint8_t* __restrict__ cnt_c,
                                                     T* __restrict__ fc, const PcgState* ps)
{
    if (ps && ps->done) return;
    d_restrict<T>(mf, rf, mc, cnt_c, fc, (long)blockIdx.x * 256 + threadIdx.x);
}

template <typename T>
__global__ __launch_bounds__(256) void k_mg_prolong(MLevel mf, const uint8_t* __restrict__ cnt_f, T* __restrict__ u, MLevel mc,
                                                    const T* __restrict__ ec, const PcgState* ps)
{
    if (ps && ps->done) return;
    d_prolong<T>(mf, cnt_f, u, mc, ec, (long)blockIdx.x * 256 + threadIdx.x);
}

// ---- tail: the whole sub-V-cycle of the small levels in ONE block ---------------------------------------
// Levels of <= ~16k cells are pure launch latency as separate kernels (7 launches x ~3 us each).  One
// block of 1024 threads walks down and up through them with __syncthreads() between the stages; the
// coarsest level (<= 8^3) is solved by symmetric red-black Gauss-Seidel in LDS.
template <typename T>
struct MgTail {
    int nl;                 // levels in the tail; the last is the coarsest
    MLevel m[MG_TAIL_MAX];
    const uint8_t* cnt[MG_TAIL_MAX];
    T* u[MG_TAIL_MAX];
    T* v[MG_TAIL_MAX];
    T* f[MG_TAIL_MAX];      // f[0] is the rhs handed down by the caller
    T* r[MG_TAIL_MAX];
    T off[MG_TAIL_MAX];     // off-diagonal of each level; diag = -off * n
    int sweeps;
};

template <typename T>
__global__ __launch_bounds__(1024) void k_mg_tail(MgTail<T> a, const PcgState* ps)
{
    __shared__ T sd[MG_TAIL_MAX][8], si[MG_TAIL_MAX][8];
    __shared__ T su[10 * 10 * 10];
    if (ps && ps->done) return;
    const int tid = threadIdx.x;
    if (tid < a.nl * 8) {
        const int l = tid >> 3, n = tid & 7;
        T off = a.off[0];
#pragma unroll
        for (int q = 1; q < MG_TAIL_MAX; ++q) off = (l == q) ? a.off[q] : off;  // static indices into the kernarg
        const T d = -off * (T)n;
        sd[l][n] = d;
        si[l][n] = n ? (T)1 / d : (T)0;
    }
    __syncthreads();
#define TAIL_FOR(lv) for (long t = tid; t < (long)a.m[lv].dx * a.m[lv].dy * a.m[lv].dz; t += 1024)
#pragma unroll
    for (int l = 0; l < MG_TAIL_MAX - 1; ++l) {
        if (l < a.nl - 1) {
            TAIL_FOR(l) d_smooth0<T>(a.m[l], a.cnt[l], a.f[l], a.u[l], sd[l], si[l], a.off[l], t);
            __syncthreads();
            TAIL_FOR(l) d_resid<T>(a.m[l], a.cnt[l], a.f[l], a.u[l], a.r[l], sd[l], a.off[l], t);
            __syncthreads();
            TAIL_FOR(l + 1) d_restrict<T>(a.m[l], a.r[l], a.m[l + 1], a.cnt[l + 1], a.f[l + 1], t);
            __syncthreads();
        }
    }
    // coarsest: red-black Gauss-Seidel in LDS, forward sweeps then reversed (symmetric)
#pragma unroll
    for (int lc = 0; lc < MG_TAIL_MAX; ++lc) {
        if (lc == a.nl - 1) {
            const MLevel& m = a.m[lc];
            for (int q = tid; q < 1000; q += 1024) su[q] = 0;
            int i = 0, j = 0, k = 0;
            const bool ok = mg_cell(m, tid, i, j, k);
            size_t c = 0;
            int n = 0, lidx = 0;
            T fv = 0, inv = 0;
            bool isred = false;
            if (ok) {
                c = m.at(i, j, k);
                n = a.cnt[lc][c];
                fv = a.f[lc][c];
                inv = si[lc][n];
                lidx = ((i + 1) * 10 + (j + 1)) * 10 + (k + 1);
                isred = ((i + j + k) & 1) == 0;
            }
            __syncthreads();
            for (int s = 0; s < 2 * a.sweeps; ++s) {
                const bool fwd = s < a.sweeps;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bool col = fwd ? (h == 0) : (h == 1);
                    if (ok && n && isred == col) {
                        const T nb = su[lidx - 100] + su[lidx + 100] + su[lidx - 10] + su[lidx + 10] + su[lidx - 1] + su[lidx + 1];
                        su[lidx] = (fv - a.off[lc] * nb) * inv;
                    }
                    __syncthreads();
                }
            }
            if (ok) a.u[lc][c] = n ? su[lidx] : (T)0;
            __syncthreads();
        }
    }
#pragma unroll
    for (int l = MG_TAIL_MAX - 2; l >= 0; --l) {
        if (l < a.nl - 1) {
            TAIL_FOR(l) d_prolong<T>(a.m[l], a.cnt[l], a.u[l], a.m[l + 1], a.u[l + 1], t);
            __syncthreads();
            TAIL_FOR(l) d_smooth<T>(a.m[l], a.cnt[l], a.f[l], a.u[l], a.v[l], sd[l], si[l], a.off[l], (T)MG_W2, t);
            __syncthreads();
            TAIL_FOR(l) d_smooth<T>(a.m[l], a.cnt[l], a.f[l], a.v[l], a.u[l], sd[l], si[l], a.off[l], (T)MG_W1, t);
            __syncthreads();
        }
    }
#undef TAIL_FOR
}

// ---- launchers ----------------------------------------------------------------------------------
static inline unsigned mg_blocks(const MLevel& m) { return (unsigned)(((long)m.dx * m.dy * m.dz + 255) / 256); }
int mg_smooth_blocks(const MLevel& m) { return (int)mg_blocks(m); }

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
void launch_mg_coarsen(hipStream_t st, MLevel mf, const uint8_t* tf, MLevel mc, uint8_t* tc, uint8_t* cnt_c)
{
    hipLaunchKernelGGL(k_mg_coarsen, dim3(mg_blocks(mc)), dim3(256), 0, st, mf, tf, mc, tc);
    hipLaunchKernelGGL(k_mg_cnt, dim3(mg_blocks(mc)), dim3(256), 0, st, mc, (const uint8_t*)tc, cnt_c);
}
template <typename T>
void launch_mg_smooth(hipStream_t st, MLevel m, const uint8_t* cnt, const T* f, const T* u_in, T* u_out, MgCoef<T> cf, int sweep,
                      double* part_dot, const PcgState* ps)
{
    // sweep: 0 = both pre-sweeps from u = 0 (u_in == nullptr), 1 = first post-sweep (W2), 2 = second post-sweep (W1)
    const T omega = (T)(sweep == 1 ? MG_W2 : MG_W1);
    hipLaunchKernelGGL((k_mg_smooth<T>), dim3(mg_blocks(m)), dim3(256), 0, st, m, cnt, f, u_in, u_out, cf, omega, part_dot, ps);
}
template <typename T>
void launch_mg_resid(hipStream_t st, MLevel m, const uint8_t* cnt, const T* f, const T* u, T* r, MgCoef<T> cf, const PcgState* ps)
{
    hipLaunchKernelGGL((k_mg_resid<T>), dim3(mg_blocks(m)), dim3(256), 0, st, m, cnt, f, u, r, cf, ps);
}
template <typename T>
void launch_mg_restrict(hipStream_t st, MLevel mf, const T* rf, MLevel mc, const uint8_t* cnt_c, T* fc, const PcgState* ps)
{
    hipLaunchKernelGGL((k_mg_restrict<T>), dim3(mg_blocks(mc)), dim3(256), 0, st, mf, rf, mc, cnt_c, fc, ps);
}
template <typename T>
void launch_mg_prolong(hipStream_t st, MLevel mf, const uint8_t* cnt_f, T* u, MLevel mc, const T* ec, const PcgState* ps)
{
    hipLaunchKernelGGL((k_mg_prolong<T>), dim3(mg_blocks(mf)), dim3(256), 0, st, mf, cnt_f, u, mc, ec, ps);
}
template <typename T>
void launch_mg_smooth0_resid(hipStream_t st, MLevel m, const uint8_t* cnt, const T* f, T* u, T* r, MgCoef<T> cf, const PcgState* ps)
{
    hipLaunchKernelGGL((k_mg_smooth0_resid<T>), dim3(mg_blocks(m)), dim3(256), 0, st, m, cnt, f, u, r, cf, ps);
}
template <typename T>
void launch_mg_prolong_smooth(hipStream_t st, MLevel mf, const uint8_t* cnt_f, const T* f, const T* u, T* u_out, MLevel mc, const T* ec,
                              MgCoef<T> cf, const PcgState* ps)
{
    hipLaunchKernelGGL((k_mg_prolong_smooth<T>), dim3(mg_blocks(mf)), dim3(256), 0, st, mf, cnt_f, f, u, u_out, mc, ec, cf, ps);
}
// levels[0..nl) of the tail; f[0] = rhs of the first tail level; result in u[0]
template <typename T>
void launch_mg_tail(hipStream_t st, int nl, const MLevel* lv, uint8_t* const* cnt, T* const* u, T* const* v, T* const* f, T* const* r,
                    const T* off, int sweeps, const PcgState* ps)
{
    MgTail<T> a;
    a.nl = nl;
    for (int l = 0; l < MG_TAIL_MAX; ++l) {
        const int q = l < nl ? l : nl - 1;
        a.m[l] = lv[q]; a.cnt[l] = cnt[q]; a.u[l] = u[q]; a.v[l] = v[q]; a.f[l] = f[q]; a.r[l] = r[q]; a.off[l] = off[q];
    }
    a.sweeps = sweeps;
    hipLaunchKernelGGL((k_mg_tail<T>), dim3(1), dim3(1024), 0, st, a, ps);
}

#define INSTMG(T)                                                                                                                   \
    template void launch_mg_smooth<T>(hipStream_t, MLevel, const uint8_t*, const T*, const T*, T*, MgCoef<T>, int, double*, const PcgState*); \
    template void launch_mg_resid<T>(hipStream_t, MLevel, const uint8_t*, const T*, const T*, T*, MgCoef<T>, const PcgState*);        \
    template void launch_mg_restrict<T>(hipStream_t, MLevel, const T*, MLevel, const uint8_t*, T*, const PcgState*);                  \
    template void launch_mg_prolong<T>(hipStream_t, MLevel, const uint8_t*, T*, MLevel, const T*, const PcgState*);                   \
    template void launch_mg_smooth0_resid<T>(hipStream_t, MLevel, const uint8_t*, const T*, T*, T*, MgCoef<T>, const PcgState*);          \
    template void launch_mg_prolong_smooth<T>(hipStream_t, MLevel, const uint8_t*, const T*, const T*, T*, MLevel, const T*, MgCoef<T>,  \
                                              const PcgState*);                                                                       \
    template void launch_mg_tail<T>(hipStream_t, int, const MLevel*, uint8_t* const*, T* const*, T* const*, T* const*, T* const*, const T*, int, \
                                    const PcgState*);
INSTMG(double)

}  // namespace fl
