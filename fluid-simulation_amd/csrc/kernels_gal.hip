// The V-cycle's coarse levels with Galerkin operators by 2 x 2 x 2 aggregation — for mostly-air boxes (FLUID_MG_GALERKIN).
//
// kernels_mg.hip re-discretises the 7-point operator on every level and calls a coarse cell air as soon as one child is: a
// shallow pool loses a layer of coarse cells per level (1.41 M unknowns -> 141 k -> 14.5 k -> 1.1 k where 8 x fewer per level
// would leave 176 k, 22 k, 2.7 k) and the long horizontal modes get no coarse correction: 30+ iterations where the dense
// free-fall system needs 21 (tests/experiments/mg_late_phase.py).  Here a coarse cell is an unknown if ANY child is, and its row is
// the sum of its children's rows (P = piecewise constant over the unknown children, R = P^T, A_c = P^T A P): still a 7-point
// operator — aggregates are face neighbours or not at all — but with per-cell coefficients, so the free surface stays where it
// is on every level.  The coefficients are small integers x scale: the diagonal `gd` and the weights of the +x / +y / +z faces
// `gx, gy, gz` (0 where either side is no unknown) in arrays laid out like the level's other arrays.  Transfers: the restriction
// is the sum of the 8 children, the prolongation the parent's value times an over-correction (piecewise-constant P under-
// estimates smooth corrections; 1.8 measured best).  Same damped-Jacobi sweeps (MG_W1, MG_W2 before, reversed after) — M stays SPD.
//
// Level 0 keeps the reference's own coefficients and the kernels of kernels_mg.hip (its down leg also forms the coarse right-hand side: the sum
// of its residual over each coarse cell's children, x pairs first, then y, then z; its up leg takes the parent's value: `pconst`); this file
// holds the coefficient set-up, the legs of the levels >= 1 (8^3 tiles + halo 2 staged in LDS as plain 3-D arrays: these levels are small
// and latency-bound, the code is kept simple) and the coarsest level's symmetric red-black Gauss-Seidel in one block.
#include "common.h"

namespace fl {

constexpr float GAL_W1 = 0.5617f, GAL_W2 = 1.6f;   // MG_W1 of kernels_mg.hip; the second sweep heavier than its 1.3895 (step 445: 1.39 -> 22 iterations, 1.6 ... 2.0 -> 21; step 210: 59 -> 56)

__device__ __forceinline__ bool gal_in(const MLevel& m, int i, int j, int k)
{
    return (unsigned)i < (unsigned)m.dx && (unsigned)j < (unsigned)m.dy && (unsigned)k < (unsigned)m.dz;
}
__device__ __forceinline__ bool gal_cell(const MLevel& m, long t, int& i, int& j, int& k)
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

// level 1 from the fine count bytes (0 = no unknown, else the index into the reference's diagonal table)
__global__ __launch_bounds__(256) void k_gal_level1(MLevel m0, const uint8_t* __restrict__ cnt0, MgCoef<float> cf0, MLevel m1, float* __restrict__ gd,
                                                    float* __restrict__ gx, float* __restrict__ gy, float* __restrict__ gz, uint8_t* __restrict__ cnt1)
{
    int I, J, K;
    if (!gal_cell(m1, (long)blockIdx.x * 256 + threadIdx.x, I, J, K)) return;
    const float scale = -cf0.off;
    auto n_at = [&](int i, int j, int k) -> int { return gal_in(m0, i, j, k) ? (int)(cnt0[m0.at(i, j, k)] & 7) : 0; };
    int n[8];
    float dsum = 0;
    int any = 0;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        n[a] = n_at(2 * I + (a & 1), 2 * J + ((a >> 1) & 1), 2 * K + (a >> 2));
        if (n[a]) { dsum += cf0.diag[n[a]]; any = 1; }
    }
    int inner = 0, fx = 0, fy = 0, fz = 0;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        if (!n[a]) continue;
        const int i = 2 * I + (a & 1), j = 2 * J + ((a >> 1) & 1), k = 2 * K + (a >> 2);
        if (!(a & 1)) inner += n[a | 1] != 0; else fx += n_at(i + 1, j, k) != 0;
        if (!(a & 2)) inner += n[a | 2] != 0; else fy += n_at(i, j + 1, k) != 0;
        if (!(a & 4)) inner += n[a | 4] != 0; else fz += n_at(i, j, k + 1) != 0;
    }
    const size_t C = m1.at(I, J, K);
    gd[C] = any ? dsum - 2.0f * scale * (float)inner : 0.0f;
    gx[C] = scale * (float)fx; gy[C] = scale * (float)fy; gz[C] = scale * (float)fz;
    cnt1[C] = (uint8_t)any;
}
// level l + 1 from level l
__global__ __launch_bounds__(256) void k_gal_coarsen(MLevel mf, const float* __restrict__ fd, const float* __restrict__ fx, const float* __restrict__ fy,
                                                     const float* __restrict__ fz, const uint8_t* __restrict__ cntf, MLevel mc, float* __restrict__ gd,
                                                     float* __restrict__ gx, float* __restrict__ gy, float* __restrict__ gz, uint8_t* __restrict__ cntc)
{
    int I, J, K;
    if (!gal_cell(mc, (long)blockIdx.x * 256 + threadIdx.x, I, J, K)) return;
    float dsum = 0, inner = 0, ox = 0, oy = 0, oz = 0;
    int any = 0;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int i = 2 * I + (a & 1), j = 2 * J + ((a >> 1) & 1), k = 2 * K + (a >> 2);
        if (!gal_in(mf, i, j, k)) continue;
        const size_t q = mf.at(i, j, k);
        if (!cntf[q]) continue;
        any = 1;
        dsum += fd[q];
        if (!(a & 1)) inner += fx[q]; else ox += fx[q];
        if (!(a & 2)) inner += fy[q]; else oy += fy[q];
        if (!(a & 4)) inner += fz[q]; else oz += fz[q];
    }
    const size_t C = mc.at(I, J, K);
    gd[C] = any ? dsum - 2.0f * inner : 0.0f;
    gx[C] = ox; gy[C] = oy; gz[C] = oz;
    cntc[C] = (uint8_t)any;
}

constexpr int GT = 8, GA = GT + 4, GB = GT + 2;              // tile, region A (halo 2), region B (halo 1)
constexpr int GNA = GA * GA * GA, GNB = GB * GB * GB, GNT = GT * GT * GT;
constexpr int GNTH = 512;   // threads of a leg block (a tile's three stages are 4 + 2 + 1 rounds of them)
struct GalLds {
    float f[GNA], d[GNA], wx[GNA], wy[GNA], wz[GNA], a[GNA], b[GNB], r[GNT];
};
__device__ __forceinline__ float gal_inv(float d) { return d > 0 ? 1.0f / d : 0.0f; }
// A u at region-A index q (its six neighbours are inside A), u over region A
__device__ __forceinline__ float gal_apply_a(const GalLds& s, const float* u, int q)
{
    return s.d[q] * u[q] - (s.wx[q] * u[q + GA * GA] + s.wx[q - GA * GA] * u[q - GA * GA] + s.wy[q] * u[q + GA] + s.wy[q - GA] * u[q - GA] +
                            s.wz[q] * u[q + 1] + s.wz[q - 1] * u[q - 1]);
}
// ... with u over region B (index p) and the coefficients at the matching region-A index q
__device__ __forceinline__ float gal_apply_b(const GalLds& s, const float* u, int p, int q)
{
    return s.d[q] * u[p] - (s.wx[q] * u[p + GB * GB] + s.wx[q - GA * GA] * u[p - GB * GB] + s.wy[q] * u[p + GB] + s.wy[q - GA] * u[p - GB] +
                            s.wz[q] * u[p + 1] + s.wz[q - 1] * u[p - 1]);
}
// flags[tile] = the tile's own cells hold an unknown (once per step; a leg tile without one has nothing to do: its outputs stay the zeros
// the step's clearing left)
__global__ __launch_bounds__(256) void k_gal_tile_flags(MLevel m, const uint8_t* __restrict__ cnt, int ntx, int nty, uint8_t* __restrict__ flags)
{
    const int tile = blockIdx.x;
    const int tz = tile % ntx, ty = (tile / ntx) % nty, tx = tile / (ntx * nty);
    const int i0 = tx * GT, j0 = ty * GT, k0 = tz * GT;
    int any = 0;
    for (int t = threadIdx.x; t < GNT; t += 256) {
        const int i = i0 + t / (GT * GT), j = j0 + (t / GT) % GT, k = k0 + t % GT;
        if (gal_in(m, i, j, k)) any |= cnt[m.at(i, j, k)];
    }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) flags[tile] = any != 0;
}
// region A of a tile: coefficients, rhs and (UP) u + wc * the parent's correction, every load issued before the first is used
// (addresses clamped into the level, values masked afterwards)
template <bool UP>
__device__ __forceinline__ void gal_load(GalLds& s, const MLevel& m, const float* __restrict__ gd, const float* __restrict__ gx, const float* __restrict__ gy,
                                         const float* __restrict__ gz, const float* __restrict__ f, int i0, int j0, int k0, const float* __restrict__ u,
                                         const MLevel& mc, const float* __restrict__ ec, float wc)
{
    constexpr int NIT = (GNA + GNTH - 1) / GNTH;
    float d[NIT], wx[NIT], wy[NIT], wz[NIT], ff[NIT], uu[NIT], ee[NIT];
    bool in[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int t = min((int)threadIdx.x + GNTH * it, GNA - 1);
        const int x = t / (GA * GA), y = (t / GA) % GA, z = t % GA;
        const int i = i0 - 2 + x, j = j0 - 2 + y, k = k0 - 2 + z;
        in[it] = gal_in(m, i, j, k);
        const int ic = min(max(i, 0), m.dx - 1), jc = min(max(j, 0), m.dy - 1), kc = min(max(k, 0), m.dz - 1);
        const size_t q = m.at(ic, jc, kc);
        d[it] = gd[q]; wx[it] = gx[q]; wy[it] = gy[q]; wz[it] = gz[q]; ff[it] = f[q];
        uu[it] = 0; ee[it] = 0;
        if (UP) { uu[it] = u[q]; ee[it] = ec[mc.at(ic >> 1, jc >> 1, kc >> 1)]; }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int t = (int)threadIdx.x + GNTH * it;
        if (t < GNA) {
            const float dd = in[it] ? d[it] : 0.0f;
            s.d[t] = dd;
            s.wx[t] = in[it] ? wx[it] : 0.0f; s.wy[t] = in[it] ? wy[it] : 0.0f; s.wz[t] = in[it] ? wz[it] : 0.0f;
            s.f[t] = in[it] ? ff[it] : 0.0f;
            s.a[t] = dd > 0 ? (UP ? uu[it] + wc * ee[it] : GAL_W1 * gal_inv(dd) * ff[it]) : 0.0f;
        }
    }
}

// down leg of a level >= 1: both pre-sweeps from u = 0, residual, restriction (sum of the children) into fc
__global__ __launch_bounds__(GNTH) void k_gal_down(MLevel m, const uint8_t* __restrict__ tflags, const float* __restrict__ gd, const float* __restrict__ gx,
                                                  const float* __restrict__ gy, const float* __restrict__ gz, const float* __restrict__ f, float* __restrict__ u,
                                                  MLevel mc, float* __restrict__ fc, const PcgState* ps, int ntx, int nty)
{
    __shared__ GalLds s;
    if (ps && ps->done) return;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tz = tile % ntx, ty = (tile / ntx) % nty, tx = tile / (ntx * nty);
    const int i0 = tx * GT, j0 = ty * GT, k0 = tz * GT;
    if (!tflags[tile]) return;   // (block-uniform)
    gal_load<false>(s, m, gd, gx, gy, gz, f, i0, j0, k0, nullptr, mc, nullptr, 0.0f);
    __syncthreads();
    for (int t = threadIdx.x; t < GNB; t += GNTH) {
        const int x = t / (GB * GB), y = (t / GB) % GB, z = t % GB;
        const int q = ((x + 1) * GA + y + 1) * GA + z + 1;
        const float dd = s.d[q];
        const float v = s.a[q] + GAL_W2 * gal_inv(dd) * (s.f[q] - gal_apply_a(s, s.a, q));
        s.b[t] = v;
        if (x >= 1 && x <= GT && y >= 1 && y <= GT && z >= 1 && z <= GT && dd > 0) u[m.at(i0 + x - 1, j0 + y - 1, k0 + z - 1)] = v;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < GNT; t += GNTH) {
        const int x = t / (GT * GT), y = (t / GT) % GT, z = t % GT;
        const int q = ((x + 2) * GA + y + 2) * GA + z + 2, p = ((x + 1) * GB + y + 1) * GB + z + 1;
        s.r[t] = s.d[q] > 0 ? s.f[q] - gal_apply_b(s, s.b, p, q) : 0.0f;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int X = threadIdx.x >> 4, Y = (threadIdx.x >> 2) & 3, Z = threadIdx.x & 3;
        const int I = i0 / 2 + X, J = j0 / 2 + Y, K = k0 / 2 + Z;
        if (gal_in(mc, I, J, K)) {
            float acc = 0;
#pragma unroll
            for (int a = 0; a < 8; ++a) acc += s.r[((2 * X + (a & 1)) * GT + 2 * Y + ((a >> 1) & 1)) * GT + 2 * Z + (a >> 2)];
            fc[mc.at(I, J, K)] = acc;
        }
    }
}
// up leg of a level >= 1: u + wc * (the parent's correction), both post-sweeps
__global__ __launch_bounds__(GNTH) void k_gal_up(MLevel m, const uint8_t* __restrict__ tflags, const float* __restrict__ gd, const float* __restrict__ gx,
                                                const float* __restrict__ gy, const float* __restrict__ gz, const float* __restrict__ f,
                                                const float* __restrict__ u, float* __restrict__ out, MLevel mc, const float* __restrict__ ec, float wc,
                                                const PcgState* ps, int ntx, int nty)
{
    __shared__ GalLds s;
    if (ps && ps->done) return;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tz = tile % ntx, ty = (tile / ntx) % nty, tx = tile / (ntx * nty);
    const int i0 = tx * GT, j0 = ty * GT, k0 = tz * GT;
    if (!tflags[tile]) return;   // (block-uniform)
    gal_load<true>(s, m, gd, gx, gy, gz, f, i0, j0, k0, u, mc, ec, wc);
    __syncthreads();
    for (int t = threadIdx.x; t < GNB; t += GNTH) {
        const int x = t / (GB * GB), y = (t / GB) % GB, z = t % GB;
        const int q = ((x + 1) * GA + y + 1) * GA + z + 1;
        s.b[t] = s.a[q] + GAL_W2 * gal_inv(s.d[q]) * (s.f[q] - gal_apply_a(s, s.a, q));
    }
    __syncthreads();
    for (int t = threadIdx.x; t < GNT; t += GNTH) {
        const int x = t / (GT * GT), y = (t / GT) % GT, z = t % GT;
        const int q = ((x + 2) * GA + y + 2) * GA + z + 2, p = ((x + 1) * GB + y + 1) * GB + z + 1;
        if (s.d[q] > 0) out[m.at(i0 + x, j0 + y, k0 + z)] = s.b[p] + GAL_W1 * gal_inv(s.d[q]) * (s.f[q] - gal_apply_b(s, s.b, p, q));
    }
}

// the coarsest level in one block: symmetric red-black Gauss-Seidel from u = 0 (`sweeps` forward, `sweeps` backward)
constexpr int GAL_CMAX = 3072;   // cells of the coarsest level including its ring of zeros
__global__ __launch_bounds__(1024) void k_gal_coarsest(MLevel m, const float* __restrict__ gd, const float* __restrict__ gx, const float* __restrict__ gy,
                                                       const float* __restrict__ gz, const float* __restrict__ f, float* __restrict__ u, int sweeps,
                                                       const PcgState* ps)
{
    __shared__ float su[GAL_CMAX];
    if (ps && ps->done) return;
    const int px = m.dx + 2, py = m.dy + 2, pz = m.dz + 2, n = m.dx * m.dy * m.dz;
    for (int t = threadIdx.x; t < px * py * pz; t += 1024) su[t] = 0;
    constexpr int PER = (GAL_CMAX + 1023) / 1024;
    float inv[PER], ff[PER], w[PER][6];
    int at[PER], col[PER];
    size_t gq[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int t = threadIdx.x + 1024 * e;
        inv[e] = 0; at[e] = 0; col[e] = -1; gq[e] = 0; ff[e] = 0;
        for (int q = 0; q < 6; ++q) w[e][q] = 0;
        if (t < n) {
            const int k = t % m.dz, j = (t / m.dz) % m.dy, i = t / (m.dz * m.dy);
            const size_t q = m.at(i, j, k);
            const float d = gd[q];
            if (d > 0) {
                inv[e] = 1.0f / d; ff[e] = f[q]; gq[e] = q;
                at[e] = ((i + 1) * py + j + 1) * pz + k + 1;
                col[e] = (i + j + k) & 1;
                w[e][0] = gx[q]; w[e][1] = gx[q - m.sx]; w[e][2] = gy[q]; w[e][3] = gy[q - m.sy]; w[e][4] = gz[q]; w[e][5] = gz[q - 1];   // (the ring holds zeros)
            }
        }
    }
    __syncthreads();
    auto half = [&](int c) {
#pragma unroll
        for (int e = 0; e < PER; ++e)
            if (col[e] == c) {
                const int a = at[e];
                su[a] = inv[e] * (ff[e] + w[e][0] * su[a + py * pz] + w[e][1] * su[a - py * pz] + w[e][2] * su[a + pz] + w[e][3] * su[a - pz] +
                                  w[e][4] * su[a + 1] + w[e][5] * su[a - 1]);
            }
        __syncthreads();
    };
    for (int it = 0; it < sweeps; ++it) { half(0); half(1); }
    for (int it = 0; it < sweeps; ++it) { half(1); half(0); }
#pragma unroll
    for (int e = 0; e < PER; ++e)
        if (col[e] >= 0) u[gq[e]] = su[at[e]];
}

static inline unsigned gal_blocks(const MLevel& m) { return (unsigned)(((long)m.dx * m.dy * m.dz + 255) / 256); }

bool gal_fits_coarsest(const MLevel& m) { return (long)(m.dx + 2) * (m.dy + 2) * (m.dz + 2) <= GAL_CMAX; }

void launch_gal_level1(hipStream_t st, MLevel m0, const uint8_t* cnt0, MgCoef<float> cf0, MLevel m1, float* gd, float* gx, float* gy, float* gz, uint8_t* cnt1)
{
    hipLaunchKernelGGL(k_gal_level1, dim3(gal_blocks(m1)), dim3(256), 0, st, m0, cnt0, cf0, m1, gd, gx, gy, gz, cnt1);
}
void launch_gal_coarsen(hipStream_t st, MLevel mf, const float* fd, const float* fx, const float* fy, const float* fz, const uint8_t* cntf, MLevel mc,
                        float* gd, float* gx, float* gy, float* gz, uint8_t* cntc)
{
    hipLaunchKernelGGL(k_gal_coarsen, dim3(gal_blocks(mc)), dim3(256), 0, st, mf, fd, fx, fy, fz, cntf, mc, gd, gx, gy, gz, cntc);
}
static inline void gal_tiles(const MLevel& m, int& ntx, int& nty, unsigned& n)
{
    ntx = (m.dz + GT - 1) / GT; nty = (m.dy + GT - 1) / GT;
    n = (unsigned)(ntx * nty * ((m.dx + GT - 1) / GT));
}
int gal_tile_count(const MLevel& m)
{
    int ntx, nty; unsigned n;
    gal_tiles(m, ntx, nty, n);
    return (int)n;
}
void launch_gal_tile_flags(hipStream_t st, MLevel m, const uint8_t* cnt, uint8_t* flags)
{
    int ntx, nty; unsigned n;
    gal_tiles(m, ntx, nty, n);
    hipLaunchKernelGGL(k_gal_tile_flags, dim3(n), dim3(256), 0, st, m, cnt, ntx, nty, flags);
}
void launch_gal_down(hipStream_t st, MLevel m, const uint8_t* cnt, const float* gd, const float* gx, const float* gy, const float* gz, const float* f, float* u,
                     MLevel mc, float* fc, const PcgState* ps)
{
    int ntx, nty; unsigned n;
    gal_tiles(m, ntx, nty, n);
    hipLaunchKernelGGL(k_gal_down, dim3(n), dim3(GNTH), 0, st, m, cnt, gd, gx, gy, gz, f, u, mc, fc, ps, ntx, nty);
}
void launch_gal_up(hipStream_t st, MLevel m, const uint8_t* cnt, const float* gd, const float* gx, const float* gy, const float* gz, const float* f, const float* u,
                   float* out, MLevel mc, const float* ec, float wc, const PcgState* ps)
{
    int ntx, nty; unsigned n;
    gal_tiles(m, ntx, nty, n);
    hipLaunchKernelGGL(k_gal_up, dim3(n), dim3(GNTH), 0, st, m, cnt, gd, gx, gy, gz, f, u, out, mc, ec, wc, ps, ntx, nty);
}
void launch_gal_coarsest(hipStream_t st, MLevel m, const float* gd, const float* gx, const float* gy, const float* gz, const float* f, float* u, int sweeps,
                         const PcgState* ps)
{
    hipLaunchKernelGGL(k_gal_coarsest, dim3(1), dim3(1024), 0, st, m, gd, gx, gy, gz, f, u, sweeps, ps);
}

}  // namespace fl
