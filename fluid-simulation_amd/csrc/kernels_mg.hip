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
// scheme (MLevel).  All kernels are one thread per cell with neighbours read through L1/L2: levels
// >= 1 are tiny and launch-bound; level 0 costs ~4 stencil sweeps per cycle.
#include "common.h"

namespace fl {

constexpr double MG_OMEGA = 2.0 / 3.0;

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

__device__ __forceinline__ bool mg_cell(const MLevel& m, long t, int& i, int& j, int& k)
{
    if (t >= (long)m.dx * m.dy * m.dz) return false;
    k = (int)(t % m.dz);
    j = (int)((t / m.dz) % m.dy);
    i = (int)(t / ((long)m.dz * m.dy));
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

// One damped-Jacobi sweep: u_out = u_in + omega D^-1 (f - A u_in); u_in == nullptr means u_in = 0.
// Optionally the partial of f.u_out (the r.z of the PCG when this is the last sweep of level 0).
template <typename T>
__global__ __launch_bounds__(256) void k_mg_smooth(MLevel m, const uint8_t* __restrict__ cnt, const T* __restrict__ f, const T* __restrict__ u_in,
                                                   T* __restrict__ u_out, MgCoef<T> cf, double* __restrict__ part_dot, const PcgState* ps)
{
    __shared__ double red[4];
    __shared__ T sd[8], si[8];
    if (ps && ps->done) return;  // uniform: written by an earlier launch
    mg_load_coef(sd, si, cf);
    int i, j, k;
    double acc = 0;
    if (mg_cell(m, (long)blockIdx.x * 256 + threadIdx.x, i, j, k)) {
        const size_t c = m.at(i, j, k);
        const int n = cnt[c];
        T out = 0;
        if (n) {
            const T fv = f[c], inv = si[n];
            if (u_in) {
                const T uc = u_in[c];
                const T nb = u_in[c - m.sx] + u_in[c + m.sx] + u_in[c - m.sy] + u_in[c + m.sy] + u_in[c - 1] + u_in[c + 1];
                out = uc + (T)MG_OMEGA * inv * (fv - (sd[n] * uc + cf.off * nb));
            } else {
                out = (T)MG_OMEGA * inv * fv;
            }
            acc = (double)fv * (double)out;
        }
        u_out[c] = out;
    }
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
    int i, j, k;
    if (!mg_cell(m, (long)blockIdx.x * 256 + threadIdx.x, i, j, k)) return;
    const size_t c = m.at(i, j, k);
    const int n = cnt[c];
    T out = 0;
    if (n) {
        const T nb = u[c - m.sx] + u[c + m.sx] + u[c - m.sy] + u[c + m.sy] + u[c - 1] + u[c + 1];
        out = f[c] - (sd[n] * u[c] + cf.off * nb);
    }
    r[c] = out;
}

// f_c = (1/8) P^T r_f : a coarse cell gathers its 4x4x4 fine neighbourhood, weights (1/4,3/4,3/4,1/4) per axis
template <typename T>
__global__ __launch_bounds__(256) void k_mg_restrict(MLevel mf, const T* __restrict__ rf, MLevel mc, const uint8_t* __restrict__ cnt_c,
                                                     T* __restrict__ fc, const PcgState* ps)
{
    if (ps && ps->done) return;
    int I, J, K;
    if (!mg_cell(mc, (long)blockIdx.x * 256 + threadIdx.x, I, J, K)) return;
    const size_t C = mc.at(I, J, K);
    T out = 0;
    if (cnt_c[C]) {
        auto w = [](int a) { return (a == 0 || a == 3) ? (T)0.25 : (T)0.75; };  // no private array: no scratch
        T acc = 0;
        for (int a = 0; a < 4; ++a) {
            const int i = 2 * I - 1 + a;
            if (i < 0 || i >= mf.dx) continue;
            for (int b = 0; b < 4; ++b) {
                const int j = 2 * J - 1 + b;
                if (j < 0 || j >= mf.dy) continue;
                T row = 0;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int k = 2 * K - 1 + d;
                    if (k >= 0 && k < mf.dz) row += w(d) * rf[mf.at(i, j, k)];
                }
                acc += w(a) * w(b) * row;
            }
        }
        out = acc * (T)0.125;
    }
    fc[C] = out;
}

// u += P e : a fine cell interpolates from its 8 nearest coarse cells (non-unknown coarse cells hold 0)
template <typename T>
__global__ __launch_bounds__(256) void k_mg_prolong(MLevel mf, const uint8_t* __restrict__ cnt_f, T* __restrict__ u, MLevel mc,
                                                    const T* __restrict__ ec, const PcgState* ps)
{
    if (ps && ps->done) return;
    int i, j, k;
    if (!mg_cell(mf, (long)blockIdx.x * 256 + threadIdx.x, i, j, k)) return;
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

// Coarsest level (domain <= 8^3): symmetric red-black Gauss-Seidel in LDS, `sweeps` forward (R,B) then
// `sweeps` reversed (B,R).  One block of 512 threads, one thread per cell.
template <typename T>
__global__ __launch_bounds__(512) void k_mg_coarsest(MLevel m, const uint8_t* __restrict__ cnt, const T* __restrict__ f, T* __restrict__ u,
                                                     MgCoef<T> cf, int sweeps, const PcgState* ps)
{
    __shared__ T su[10 * 10 * 10];
    __shared__ T sd[8], si[8];
    if (ps && ps->done) return;
    mg_load_coef(sd, si, cf);
    const int t = threadIdx.x;
    for (int q = t; q < 1000; q += 512) su[q] = 0;
    int i, j, k;
    const bool ok = mg_cell(m, t, i, j, k);
    size_t c = 0;
    int n = 0, l = 0;
    T fv = 0, inv = 0;
    bool red = false;
    if (ok) {
        c = m.at(i, j, k);
        n = cnt[c];
        fv = f[c];
        inv = n ? si[n] : (T)0;
        l = ((i + 1) * 10 + (j + 1)) * 10 + (k + 1);
        red = ((i + j + k) & 1) == 0;
    }
    __syncthreads();
    for (int s = 0; s < 2 * sweeps; ++s) {
        const bool fwd = s < sweeps;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool col = fwd ? (h == 0) : (h == 1);  // forward: red then black; reversed: black then red
            if (ok && n && red == col) {
                const T nb = su[l - 100] + su[l + 100] + su[l - 10] + su[l + 10] + su[l - 1] + su[l + 1];
                su[l] = (fv - cf.off * nb) * inv;
            }
            __syncthreads();
        }
    }
    if (ok) u[c] = n ? su[l] : (T)0;
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
void launch_mg_smooth(hipStream_t st, MLevel m, const uint8_t* cnt, const T* f, const T* u_in, T* u_out, MgCoef<T> cf, double* part_dot,
                      const PcgState* ps)
{
    hipLaunchKernelGGL((k_mg_smooth<T>), dim3(mg_blocks(m)), dim3(256), 0, st, m, cnt, f, u_in, u_out, cf, part_dot, ps);
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
void launch_mg_coarsest(hipStream_t st, MLevel m, const uint8_t* cnt, const T* f, T* u, MgCoef<T> cf, int sweeps, const PcgState* ps)
{
    hipLaunchKernelGGL((k_mg_coarsest<T>), dim3(1), dim3(512), 0, st, m, cnt, f, u, cf, sweeps, ps);
}

#define INSTMG(T)                                                                                                                   \
    template void launch_mg_smooth<T>(hipStream_t, MLevel, const uint8_t*, const T*, const T*, T*, MgCoef<T>, double*, const PcgState*); \
    template void launch_mg_resid<T>(hipStream_t, MLevel, const uint8_t*, const T*, const T*, T*, MgCoef<T>, const PcgState*);        \
    template void launch_mg_restrict<T>(hipStream_t, MLevel, const T*, MLevel, const uint8_t*, T*, const PcgState*);                  \
    template void launch_mg_prolong<T>(hipStream_t, MLevel, const uint8_t*, T*, MLevel, const T*, const PcgState*);                   \
    template void launch_mg_coarsest<T>(hipStream_t, MLevel, const uint8_t*, const T*, T*, MgCoef<T>, int, const PcgState*);
INSTMG(double)

}  // namespace fl
