// Grid-side kernels of the PIC/FLIP step for gfx950: cell flags + unknown numbering
// (integer, bit-exact), RHS/divergence, velocity update (gather form), FLIP delta field.
#include "common.h"

namespace fl {

// ---- exclusive prefix sum over ints ---------------------------------------------------------
// Three launches: chunk sums, one-block scan of the chunk sums, per-chunk scan + offset.
// MODE 0: in = int array, out = exclusive sums.
// MODE 1: in = flag bytes (1 where F_FLUID), out = the reference's `indices` grid: running
//         count on fluid cells in x-major / z-fastest order, -1 elsewhere (fluid.cc:1388,1416-1433).
constexpr int SCAN_T = 256;
constexpr int SCAN_E = 8;
constexpr int SCAN_CHUNK = SCAN_T * SCAN_E;

template <int MODE>
__device__ __forceinline__ int scan_load(const void* in, long i)
{
    if (MODE == 0) return ((const int*)in)[i];
    return (((const uint8_t*)in)[i] & F_FLUID) ? 1 : 0;
}

template <int MODE>
__global__ __launch_bounds__(SCAN_T) void k_scan_sums(const void* __restrict__ in, long n, int* __restrict__ block_sums)
{
    __shared__ int sm[4];
    long base = (long)blockIdx.x * SCAN_CHUNK + (long)threadIdx.x * SCAN_E;
    int s = 0;
#pragma unroll
    for (int e = 0; e < SCAN_E; ++e)
        if (base + e < n) s += scan_load<MODE>(in, base + e);
    s = block_sum<int, 4>(s, sm);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s;
}

// exclusive scan of block_sums[0..nb) in place by ONE block of 1024 threads; total -> *total
__global__ __launch_bounds__(1024) void k_scan_block_sums(int* __restrict__ block_sums, int nb, int* __restrict__ total)
{
    __shared__ int wsum[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int base = 0; base < nb; base += 1024) {
        int i = base + threadIdx.x;
        int v = i < nb ? block_sums[i] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        int carry = carry_s;
        if (i < nb) block_sums[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

template <int MODE>
__global__ __launch_bounds__(SCAN_T) void k_scan_final(const void* __restrict__ in, long n, const int* __restrict__ block_sums,
                                                       int* __restrict__ out)
{
    __shared__ int wsum[4];
    long base = (long)blockIdx.x * SCAN_CHUNK + (long)threadIdx.x * SCAN_E;
    int v[SCAN_E];
    int s = 0;
#pragma unroll
    for (int e = 0; e < SCAN_E; ++e) {
        v[e] = (base + e < n) ? scan_load<MODE>(in, base + e) : 0;
        s += v[e];
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int off = block_sums[blockIdx.x] + inc - s;
    for (int k = 0; k < w; ++k) off += wsum[k];
#pragma unroll
    for (int e = 0; e < SCAN_E; ++e) {
        if (base + e < n) {
            if (MODE == 0) out[base + e] = off;
            else out[base + e] = v[e] ? off : -1;
        }
        off += v[e];
    }
}

template <int MODE>
static void scan_impl(hipStream_t st, const void* in, int* out, long n, int* block_sums, int* total)
{
    int nb = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
    hipLaunchKernelGGL(k_scan_sums<MODE>, dim3(nb), dim3(SCAN_T), 0, st, in, n, block_sums);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, st, block_sums, nb, total);
    hipLaunchKernelGGL(k_scan_final<MODE>, dim3(nb), dim3(SCAN_T), 0, st, in, n, (const int*)block_sums, out);
}

void launch_exclusive_scan(hipStream_t st, const int* in, int* out, long n, int* block_sums, int* total)
{
    scan_impl<0>(st, in, out, n, block_sums, total);
}
void launch_index_scan(hipStream_t st, Grid g, const uint8_t* flags, int* indices, int* block_sums, int* total)
{
    scan_impl<1>(st, flags, indices, (long)g.cells(), block_sums, total);
}

// ---- flags -----------------------------------------------------------------------------------
// fluid(c) = container(c) > 0 && !solid(c)            (fluid.cc:326,445,579,1423-1425)
// cnt(c)   = #{6-neighbours that are not solid}       (multiplicity of `scale` in Adiag(c),
//            fluid.cc:328-372 for the + side, 378-405 and 334/349/365 for the - side; an
//            off-grid neighbour reads background 0 = not solid)
__global__ __launch_bounds__(256) void k_flags(Grid g, const uint8_t* __restrict__ solid, const float* __restrict__ container,
                                               uint8_t* __restrict__ flags, long c_begin, long c_end)
{
    long c = c_begin + (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= c_end) return;
    const int iz = (int)(c % g.nz), iy = (int)((c / g.nz) % g.ny), ix = (int)(c / g.sx());
    const uint8_t sol = solid[c] ? F_SOLID : 0;
    uint8_t f = sol;
    if (!sol && container[c] > 0) {
        int cnt = 0;
        cnt += (ix > 0) ? !solid[c - g.sx()] : 1;
        cnt += (ix < g.nx - 1) ? !solid[c + g.sx()] : 1;
        cnt += (iy > 0) ? !solid[c - g.nz] : 1;
        cnt += (iy < g.ny - 1) ? !solid[c + g.nz] : 1;
        cnt += (iz > 0) ? !solid[c - 1] : 1;
        cnt += (iz < g.nz - 1) ? !solid[c + 1] : 1;
        f = (uint8_t)(F_FLUID | (cnt << F_CNT_SHIFT));
    }
    flags[c] = f;
}

// the same, four cells of a z row per thread (nz a multiple of 4): one word of solid bytes, one float4, one word of flags stored —
// a byte per thread moves 1.3 TB/s over a mostly-air box
__global__ __launch_bounds__(256) void k_flags4(Grid g, const uint8_t* __restrict__ solid, const float* __restrict__ container,
                                                uint8_t* __restrict__ flags, long c_begin, long c_end)
{
    const long c = c_begin + ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= c_end) return;
    const int iz = (int)(c % g.nz), iy = (int)((c / g.nz) % g.ny), ix = (int)(c / g.sx());
    const uint32_t so = *(const uint32_t*)(solid + c);
    const float4 co = *(const float4*)(container + c);
    const float cv[4] = {co.x, co.y, co.z, co.w};
    bool any = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) any |= !((so >> (8 * q)) & 0xff) && cv[q] > 0;
    uint32_t out = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) out |= (((so >> (8 * q)) & 0xff) ? (uint32_t)F_SOLID : 0u) << (8 * q);
    if (any) {
        const uint32_t xm = ix > 0 ? *(const uint32_t*)(solid + c - g.sx()) : 0u, xp = ix < g.nx - 1 ? *(const uint32_t*)(solid + c + g.sx()) : 0u;
        const uint32_t ym = iy > 0 ? *(const uint32_t*)(solid + c - g.nz) : 0u, yp = iy < g.ny - 1 ? *(const uint32_t*)(solid + c + g.nz) : 0u;
        const uint32_t zm = iz > 0 ? solid[c - 1] : 0u, zp = iz + 4 < g.nz ? solid[c + 4] : 0u;
        out = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t sol = (so >> (8 * q)) & 0xff;
            uint32_t f = sol ? (uint32_t)F_SOLID : 0u;
            if (!sol && cv[q] > 0) {
                int cnt = 0;
                cnt += !((xm >> (8 * q)) & 0xff);
                cnt += !((xp >> (8 * q)) & 0xff);
                cnt += !((ym >> (8 * q)) & 0xff);
                cnt += !((yp >> (8 * q)) & 0xff);
                cnt += !(q == 0 ? zm : (so >> (8 * (q - 1))) & 0xff);
                cnt += !(q == 3 ? zp : (so >> (8 * (q + 1))) & 0xff);
                f = (uint32_t)(F_FLUID | (cnt << F_CNT_SHIFT));
            }
            out |= f << (8 * q);
        }
    }
    *(uint32_t*)(flags + c) = out;
}

// x planes [x0, x1] (inclusive); the whole grid for x0=0, x1=N-1
void launch_flags(hipStream_t st, Grid g, const uint8_t* solid, const float* container, uint8_t* flags, int x0, int x1)
{
    const long n2 = g.sx(), c0 = (long)x0 * n2, c1 = (long)(x1 + 1) * n2;
    if (c1 <= c0) return;
    if (g.nz % 4 == 0)   // (planes then start on a multiple of 4 cells: hipMalloc'ed fields, whole words)
        hipLaunchKernelGGL(k_flags4, dim3((unsigned)(((c1 - c0) / 4 + 255) / 256)), dim3(256), 0, st, g, solid, container, flags, c0, c1);
    else
        hipLaunchKernelGGL(k_flags, dim3((unsigned)((c1 - c0 + 255) / 256)), dim3(256), 0, st, g, solid, container, flags, c0, c1);
}

void launch_index_scan_range(hipStream_t st, Grid g, const uint8_t* flags, int* indices, int* block_sums, int* total, int x0, int x1)
{
    const long n2 = g.sx();
    if (x1 < x0) return;
    scan_impl<1>(st, flags + (long)x0 * n2, indices + (long)x0 * n2, (long)(x1 - x0 + 1) * n2, block_sums, total);
}
// ---- box helpers -----------------------------------------------------------------------------
struct CellIt {
    int ix, iy, iz;
    size_t c;
    bool ok;
};
__device__ __forceinline__ CellIt box_cell(const Grid& g, const Box& box)
{
    CellIt r;
    const int nz = box.nz(), ny = box.ny();
    long t = (long)blockIdx.x * 256 + threadIdx.x;
    r.ok = t < box.cells();
    if (!r.ok) { r.ix = r.iy = r.iz = 0; r.c = 0; return r; }
    r.iz = (int)(t % nz) + box.z0;
    r.iy = (int)((t / nz) % ny) + box.y0;
    r.ix = (int)(t / ((long)nz * ny)) + box.x0;
    r.c = g.idx(r.ix, r.iy, r.iz);
    return r;
}

// ---- setRHS + setDiver -------------------------------------------------------------------------
// fluid.cc:414-479 then 566-610, per fluid cell, with the reference's float32 grid
// narrowing after every accumulation and its term order (-x,+x,-y,+y,-z,+z).
// one fluid cell of setRHS + setDiver from its loaded values (both kernels below): s?? = that neighbour exists and is solid
__device__ __forceinline__ void rhs_div_vals(double uc, double vc, double wc, double ui, double vj, double wk, bool sxm, bool sxp, bool sym,
                                             bool syp, bool szm, bool szp, double dx, double g0, double g1, double g2, float& r, float& d)
{
    const double scale = 1.0 / dx;
    r = 0.0f;
    if (sxm) r = (float)((double)r - (scale * (uc + g0)));
    if (sxp) r = (float)((double)r + (scale * (ui + g0)));
    if (sym) r = (float)((double)r - (scale * (vc + g1)));
    if (syp) r = (float)((double)r + (scale * (vj + g1)));
    if (szm) r = (float)((double)r - (scale * (wc + g2)));
    if (szp) r = (float)((double)r + (scale * (wk + g2)));
    double du = 0, dv = 0, dw = 0;
    if (!sxp) du = (ui - uc) / dx;
    if (!syp) dv = (vj - vc) / dx;
    if (!szp) dw = (wk - wc) / dx;
    d = (float)(((double)r) - du - dv - dw);
}
__device__ __forceinline__ void rhs_div_cell(const Grid& g, const uint8_t* __restrict__ flags, const double* __restrict__ u,
                                             const double* __restrict__ v, const double* __restrict__ w, size_t c, int ix, int iy, int iz,
                                             double dx, double g0, double g1, double g2, float& r, float& d)
{
    const long sx = g.sx(), sy = g.nz;
    const bool hxm = ix > 0, hxp = ix < g.nx - 1, hym = iy > 0, hyp = iy < g.ny - 1, hzm = iz > 0, hzp = iz < g.nz - 1;
    const double uc = u[c], vc = v[c], wc = w[c];
    const double ui = hxp ? u[c + sx] : 0.0, vj = hyp ? v[c + sy] : 0.0, wk = hzp ? w[c + 1] : 0.0;
    const bool sxm = hxm && (flags[c - sx] & F_SOLID), sxp = hxp && (flags[c + sx] & F_SOLID);
    const bool sym = hym && (flags[c - sy] & F_SOLID), syp = hyp && (flags[c + sy] & F_SOLID);
    const bool szm = hzm && (flags[c - 1] & F_SOLID), szp = hzp && (flags[c + 1] & F_SOLID);
    rhs_div_vals(uc, vc, wc, ui, vj, wk, sxm, sxp, sym, syp, szm, szp, dx, g0, g1, g2, r, d);
}

__global__ __launch_bounds__(256) void k_rhs_div(Grid g, Box box, const uint8_t* __restrict__ flags, const double* __restrict__ u,
                                                 const double* __restrict__ v, const double* __restrict__ w, float* __restrict__ rhs,
                                                 float* __restrict__ diver, double dx, double g0, double g1, double g2)
{
    CellIt it = box_cell(g, box);
    if (!it.ok) return;
    const size_t c = it.c;
    float r = 0.0f, d = 0.0f;
    if (flags[c] & F_FLUID) rhs_div_cell(g, flags, u, v, w, c, it.ix, it.iy, it.iz, dx, g0, g1, g2, r, d);
    rhs[c] = r;
    diver[c] = d;
}

// The same over whole z rows of the box's x-y range, four cells per thread (nz a multiple of 4; the box spans most of z: the late
// phases, where 4/5 of the box is air): one word of flags decides, a row of air is two float4 stores.  The cells of those rows outside the
// box get the zeros they hold already (one GPU only: `rows`; a rank of a decomposed run does not own them).
__global__ __launch_bounds__(256) void k_rhs_div4(Grid g, Box box, const uint8_t* __restrict__ flags, const double* __restrict__ u,
                                                  const double* __restrict__ v, const double* __restrict__ w, float* __restrict__ rhs,
                                                  float* __restrict__ diver, double dx, double g0, double g1, double g2)
{
    const int nzq = g.nz >> 2;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)box.nx() * box.ny() * nzq) return;
    const int iz = (int)(t % nzq) * 4, iy = (int)((t / nzq) % box.ny()) + box.y0, ix = (int)(t / ((long)nzq * box.ny())) + box.x0;
    const size_t c = g.idx(ix, iy, iz);
    const uint32_t f4 = *(const uint32_t*)(flags + c);
    float r[4] = {0.0f, 0.0f, 0.0f, 0.0f}, d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (f4 & (0x01010101u * F_FLUID)) {
        // everything the four cells can want, asked for at once (one body per cell behind its own flag is four round trips in a row)
        const long sx = g.sx(), sy = g.nz;
        const bool hxm = ix > 0, hxp = ix < g.nx - 1, hym = iy > 0, hyp = iy < g.ny - 1;
        const size_t cxp = hxp ? c + sx : c, cyp = hyp ? c + sy : c, cxm = hxm ? c - sx : c, cym = hym ? c - sy : c;
        const double2 ua = *(const double2*)(u + c), ub = *(const double2*)(u + c + 2), va = *(const double2*)(v + c), vb = *(const double2*)(v + c + 2);
        const double2 wa = *(const double2*)(w + c), wb = *(const double2*)(w + c + 2);
        const double2 xa = *(const double2*)(u + cxp), xb = *(const double2*)(u + cxp + 2), ya = *(const double2*)(v + cyp), yb = *(const double2*)(v + cyp + 2);
        const double w4 = w[iz + 4 < g.nz ? c + 4 : c];
        const uint32_t fxm = *(const uint32_t*)(flags + cxm), fxp = *(const uint32_t*)(flags + cxp), fym = *(const uint32_t*)(flags + cym),
                       fyp = *(const uint32_t*)(flags + cyp);
        const uint32_t fzm = iz > 0 ? flags[c - 1] : 0u, fzp = iz + 4 < g.nz ? flags[c + 4] : 0u;
        const double uc[4] = {ua.x, ua.y, ub.x, ub.y}, vc[4] = {va.x, va.y, vb.x, vb.y}, wc[5] = {wa.x, wa.y, wb.x, wb.y, w4};
        const double ui[4] = {xa.x, xa.y, xb.x, xb.y}, vj[4] = {ya.x, ya.y, yb.x, yb.y};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!(((f4 >> (8 * q)) & F_FLUID) && iz + q >= box.z0 && iz + q <= box.z1)) continue;
            const bool hzp = iz + q < g.nz - 1;
            const uint32_t zm = q == 0 ? fzm : (f4 >> (8 * (q - 1))) & 0xffu, zp = q == 3 ? fzp : (f4 >> (8 * (q + 1))) & 0xffu;
            rhs_div_vals(uc[q], vc[q], wc[q], hxp ? ui[q] : 0.0, hyp ? vj[q] : 0.0, hzp ? wc[q + 1] : 0.0,
                         hxm && ((fxm >> (8 * q)) & F_SOLID), hxp && ((fxp >> (8 * q)) & F_SOLID), hym && ((fym >> (8 * q)) & F_SOLID),
                         hyp && ((fyp >> (8 * q)) & F_SOLID), (zm & F_SOLID) != 0, hzp && (zp & F_SOLID), dx, g0, g1, g2, r[q], d[q]);
        }
    }
    *(float4*)(rhs + c) = make_float4(r[0], r[1], r[2], r[3]);
    *(float4*)(diver + c) = make_float4(d[0], d[1], d[2], d[3]);
}

void launch_rhs_div(hipStream_t st, Grid g, Box box, const uint8_t* flags, const double* u, const double* v, const double* w, float* rhs,
                    float* diver, double dx, double gdt0, double gdt1, double gdt2, bool rows)
{
    if (rows && g.nz % 4 == 0 && 4 * box.nz() >= 3 * g.nz) {
        const long n = (long)box.nx() * box.ny() * (g.nz >> 2);
        hipLaunchKernelGGL(k_rhs_div4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, box, flags, u, v, w, rhs, diver, dx, gdt0, gdt1, gdt2);
        return;
    }
    hipLaunchKernelGGL(k_rhs_div, dim3((unsigned)((box.cells() + 255) / 256)), dim3(256), 0, st, g, box, flags, u, v, w, rhs, diver, dx,
                       gdt0, gdt1, gdt2);
}

// ---- velUpdate, gather form ------------------------------------------------------------------
// fluid.cc:612-703.  The reference sweeps cells in x-major order and, at fluid cell c, adds
// -k p(c)+g' to all three stored components of c and +k p(c) to component a of c+e_a.  Every
// cell therefore sees, in this order: the +k p(c-e_a) of its three lower neighbours (they are
// visited first), then its own term.  The second sweep zeroes v(c) and v(c+e_a).a for every
// solid c.  Gathered per cell the result is bit-identical and needs no atomics.
__global__ __launch_bounds__(256) void k_vel_update(Grid g, Box box, const uint8_t* __restrict__ flags, const double* __restrict__ p,
                                                    double* __restrict__ u, double* __restrict__ v, double* __restrict__ w, double k,
                                                    double g0, double g1, double g2)
{
    CellIt it = box_cell(g, box);
    if (!it.ok) return;
    const size_t c = it.c;
    const long sx = g.sx(), sy = g.nz;
    const uint8_t f = flags[c];
    const uint8_t fxm = it.ix > 0 ? flags[c - sx] : 0, fym = it.iy > 0 ? flags[c - sy] : 0, fzm = it.iz > 0 ? flags[c - 1] : 0;
    double uc = u[c], vc = v[c], wc = w[c];
    if (fxm & F_FLUID) uc = uc + k * p[c - sx];   // :646
    if (fym & F_FLUID) vc = vc + k * p[c - sy];   // :653
    if (fzm & F_FLUID) wc = wc + k * p[c - 1];    // :660
    if (f & F_FLUID) {
        const double pre = p[c];
        uc = uc - k * pre + g0;                   // :639
        vc = vc - k * pre + g1;                   // :640
        wc = wc - k * pre + g2;                   // :641
    }
    if (f & F_SOLID) { uc = 0; vc = 0; wc = 0; }  // :682
    if (fxm & F_SOLID) uc = 0;                    // :686
    if (fym & F_SOLID) vc = 0;                    // :691
    if (fzm & F_SOLID) wc = 0;                    // :696
    u[c] = uc; v[c] = vc; w[c] = wc;
}

// The same over whole z rows of the box's x-y range, four cells per thread (see k_rhs_div4; one GPU only).  A cell changes only if it or
// one of its three lower neighbours is fluid or solid: the flag words decide before any of the 48 bytes per cell are touched, and 4/5 of a
// late-phase box is air next to air.
__global__ __launch_bounds__(256) void k_vel_update4(Grid g, Box box, const uint8_t* __restrict__ flags, const double* __restrict__ p,
                                                     double* __restrict__ u, double* __restrict__ v, double* __restrict__ w, double k,
                                                     double g0, double g1, double g2)
{
    const int nzq = g.nz >> 2;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)box.nx() * box.ny() * nzq) return;
    const int iz = (int)(t % nzq) * 4, iy = (int)((t / nzq) % box.ny()) + box.y0, ix = (int)(t / ((long)nzq * box.ny())) + box.x0;
    const size_t c = g.idx(ix, iy, iz);
    const long sx = g.sx(), sy = g.nz;
    const size_t cxm = ix > 0 ? c - sx : c, cym = iy > 0 ? c - sy : c;
    const uint32_t f4 = *(const uint32_t*)(flags + c);
    const uint32_t fx4 = ix > 0 ? *(const uint32_t*)(flags + cxm) : 0u, fy4 = iy > 0 ? *(const uint32_t*)(flags + cym) : 0u;
    const uint32_t fz0 = iz > 0 ? flags[c - 1] : 0u;
    const uint32_t fz4 = (f4 << 8) | fz0;   // byte q = flags of the cell below cell q in z
    if (!((f4 | fx4 | fy4 | fz4) & (0x01010101u * (F_FLUID | F_SOLID)))) return;
    const double2 ua = *(const double2*)(u + c), ub = *(const double2*)(u + c + 2), va = *(const double2*)(v + c), vb = *(const double2*)(v + c + 2);
    const double2 wa = *(const double2*)(w + c), wb = *(const double2*)(w + c + 2);
    const double2 pa = *(const double2*)(p + c), pb = *(const double2*)(p + c + 2);
    const double2 xa = *(const double2*)(p + cxm), xb = *(const double2*)(p + cxm + 2), ya = *(const double2*)(p + cym), yb = *(const double2*)(p + cym + 2);
    const double pz0 = p[iz > 0 ? c - 1 : c];
    double uc[4] = {ua.x, ua.y, ub.x, ub.y}, vc[4] = {va.x, va.y, vb.x, vb.y}, wc[4] = {wa.x, wa.y, wb.x, wb.y};
    const double pc[4] = {pa.x, pa.y, pb.x, pb.y}, px[4] = {xa.x, xa.y, xb.x, xb.y}, py[4] = {ya.x, ya.y, yb.x, yb.y};
    const double pz[4] = {pz0, pa.x, pa.y, pb.x};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (iz + q < box.z0 || iz + q > box.z1) continue;
        const uint32_t f = (f4 >> (8 * q)) & 0xffu, fxm = (fx4 >> (8 * q)) & 0xffu, fym = (fy4 >> (8 * q)) & 0xffu, fzm = (fz4 >> (8 * q)) & 0xffu;
        if (fxm & F_FLUID) uc[q] = uc[q] + k * px[q];   // :646
        if (fym & F_FLUID) vc[q] = vc[q] + k * py[q];   // :653
        if (fzm & F_FLUID) wc[q] = wc[q] + k * pz[q];   // :660
        if (f & F_FLUID) {
            const double pre = pc[q];
            uc[q] = uc[q] - k * pre + g0;               // :639
            vc[q] = vc[q] - k * pre + g1;               // :640
            wc[q] = wc[q] - k * pre + g2;               // :641
        }
        if (f & F_SOLID) { uc[q] = 0; vc[q] = 0; wc[q] = 0; }  // :682
        if (fxm & F_SOLID) uc[q] = 0;                   // :686
        if (fym & F_SOLID) vc[q] = 0;                   // :691
        if (fzm & F_SOLID) wc[q] = 0;                   // :696
    }
    *(double2*)(u + c) = make_double2(uc[0], uc[1]); *(double2*)(u + c + 2) = make_double2(uc[2], uc[3]);
    *(double2*)(v + c) = make_double2(vc[0], vc[1]); *(double2*)(v + c + 2) = make_double2(vc[2], vc[3]);
    *(double2*)(w + c) = make_double2(wc[0], wc[1]); *(double2*)(w + c + 2) = make_double2(wc[2], wc[3]);
}

void launch_vel_update(hipStream_t st, Grid g, Box box, const uint8_t* flags, const double* p, double* u, double* v, double* w, double k,
                       double g0, double g1, double g2, bool rows)
{
    if (rows && g.nz % 4 == 0 && 4 * box.nz() >= 3 * g.nz) {
        const long n = (long)box.nx() * box.ny() * (g.nz >> 2);
        hipLaunchKernelGGL(k_vel_update4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, box, flags, p, u, v, w, k, g0, g1, g2);
        return;
    }
    hipLaunchKernelGGL(k_vel_update, dim3((unsigned)((box.cells() + 255) / 256)), dim3(256), 0, st, g, box, flags, p, u, v, w, k, g0, g1,
                       g2);
}

// ---- FLIP delta field ------------------------------------------------------------------------
// dc(c) = getVelocity(c, vels) - getVelocity(c, velBeforeUpdate)  (fluid.cc:59-70, 239-240, 252)
__global__ __launch_bounds__(256) void k_flip_delta(Grid g, Box box, const double* __restrict__ u, const double* __restrict__ v,
                                                    const double* __restrict__ w, const double* __restrict__ ub,
                                                    const double* __restrict__ vb, const double* __restrict__ wb,
                                                    double* __restrict__ dcx, double* __restrict__ dcy, double* __restrict__ dcz,
                                                    double* __restrict__ pcx, double* __restrict__ pcy, double* __restrict__ pcz)
{
    CellIt it = box_cell(g, box);
    if (!it.ok) return;
    const size_t c = it.c;
    const long sx = g.sx(), sy = g.nz;
    const bool hxp = it.ix < g.nx - 1, hyp = it.iy < g.ny - 1, hzp = it.iz < g.nz - 1;
    const double cu = (u[c] + (hxp ? u[c + sx] : 0.0)) / 2.0, pu = (ub[c] + (hxp ? ub[c + sx] : 0.0)) / 2.0;
    const double cv = (v[c] + (hyp ? v[c + sy] : 0.0)) / 2.0, pv = (vb[c] + (hyp ? vb[c + sy] : 0.0)) / 2.0;
    const double cw = (w[c] + (hzp ? w[c + 1] : 0.0)) / 2.0, pw = (wb[c] + (hzp ? wb[c + 1] : 0.0)) / 2.0;
    dcx[c] = cu - pu;
    dcy[c] = cv - pv;
    dcz[c] = cw - pw;
    if (pcx) {  // PIC blend only: getVelocity(c, vels) itself (clampedCatmullRom, fluid.cc:163)
        pcx[c] = cu;
        pcy[c] = cv;
        pcz[c] = cw;
    }
}

void launch_flip_delta(hipStream_t st, Grid g, Box box, const double* u, const double* v, const double* w, const double* ub,
                       const double* vb, const double* wb, double* dcx, double* dcy, double* dcz, double* pcx, double* pcy, double* pcz)
{
    hipLaunchKernelGGL(k_flip_delta, dim3((unsigned)((box.cells() + 255) / 256)), dim3(256), 0, st, g, box, u, v, w, ub, vb, wb, dcx, dcy,
                       dcz, pcx, pcy, pcz);
}

// ---- error = |b-b2| / |b| --------------------------------------------------------------------
// fluid.cc:1483 over the unknowns that setA2/setOnlyB fill (Adiag != 0, :498,556).
__global__ __launch_bounds__(256) void k_err_partial(Grid g, Box box, const uint8_t* __restrict__ flags, const float* __restrict__ b,
                                                     const float* __restrict__ b2, double* __restrict__ part)
{
    __shared__ double sm[4];
    const long ncells = box.cells();
    const int nz = box.nz(), ny = box.ny();
    double num = 0, den = 0;
    // four rounds' flag bytes and right-hand sides are asked for together (a thread's cells and the order it adds them in are unchanged)
    const long stride = (long)gridDim.x * 256;
    for (long t0 = (long)blockIdx.x * 256 + threadIdx.x; t0 < ncells; t0 += 4 * stride) {
        uint8_t f4[4];
        float b4[4], c4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            long t = t0 + q * stride;
            t = t < ncells ? t : ncells - 1;
            const int iz = (int)(t % nz) + box.z0, iy = (int)((t / nz) % ny) + box.y0, ix = (int)(t / ((long)nz * ny)) + box.x0;
            const size_t c = g.idx(ix, iy, iz);
            f4[q] = flags[c];
            b4[q] = b[c];
            c4[q] = b2[c];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (t0 + q * stride >= ncells) break;
            if ((f4[q] & F_FLUID) && (f4[q] >> F_CNT_SHIFT)) {
                const double bb = (double)b4[q], d = bb - (double)c4[q];
                num += d * d;
                den += bb * bb;
            }
        }
    }
    num = block_sum<double, 4>(num, sm);
    den = block_sum<double, 4>(den, sm);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = num; part[2 * blockIdx.x + 1] = den; }
}
__global__ __launch_bounds__(256) void k_err_final(const double* __restrict__ part, int nb, StepState* ss)
{
    __shared__ double sm[4];
    double num = 0, den = 0;
    for (int i = threadIdx.x; i < nb; i += 256) { num += part[2 * i]; den += part[2 * i + 1]; }
    num = block_sum<double, 4>(num, sm);
    den = block_sum<double, 4>(den, sm);
    if (threadIdx.x == 0) { ss->err_num = num; ss->err_den = den; }
}

void launch_err_norm(hipStream_t st, Grid g, Box box, const uint8_t* flags, const float* b, const float* b2, double* part, StepState* ss)
{
    long nb = (box.cells() + 255) / 256;
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_err_partial, dim3((unsigned)nb), dim3(256), 0, st, g, box, flags, b, b2, part);
    hipLaunchKernelGGL(k_err_final, dim3(1), dim3(256), 0, st, (const double*)part, (int)nb, ss);
}

__global__ void k_zero_step_state(StepState* ss, int N)
{
    ss->bbox_min[0] = ss->bbox_min[1] = ss->bbox_min[2] = N;
    ss->bbox_max[0] = ss->bbox_max[1] = ss->bbox_max[2] = -1;
    ss->num_active = 0;
    ss->n_out = 0;
    ss->n_drop_fail = 0;
    ss->max_speed_bits = 0ull;
    ss->err_num = 0;
    ss->err_den = 0;
}
void launch_zero_step_state(hipStream_t st, StepState* ss, int N)
{
    hipLaunchKernelGGL(k_zero_step_state, dim3(1), dim3(1), 0, st, ss, N);
}

// starts of the buckets behind the scanned cell range [.., c1): everything between c1 and the grid's end is empty
__global__ void k_sort_tail(const int* __restrict__ cell_count, int* __restrict__ cell_start, long c1, long ncell)
{
    const int t = cell_start[c1];              // particles in the cells (the scan's total)
    cell_start[ncell] = t;
    cell_start[ncell + 1] = t + cell_count[ncell];
    cell_start[ncell + 2] = t + cell_count[ncell] + cell_count[ncell + 1];
}
void launch_sort_tail(hipStream_t st, const int* cell_count, int* cell_start, long c1, long ncell)
{
    hipLaunchKernelGGL(k_sort_tail, dim3(1), dim3(1), 0, st, cell_count, cell_start, c1, ncell);
}

// Zero the cells of `box` in the 4 float and 7 double step fields in ONE launch (11 separate fills cost ~5 us each;
// whole x planes instead of the box: 430 MB instead of 50 at 256^3).  Static indices into the by-value argument only.
__global__ __launch_bounds__(256) void k_zero_fields(ZeroList z, Grid g, Box box)
{
    const int a = blockIdx.y;
    float* pf = z.f4[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) pf = (a == q) ? z.f4[q] : pf;
    double* pd = z.f8[0];
#pragma unroll
    for (int q = 1; q < 7; ++q) pd = (a - 4 == q) ? z.f8[q] : pd;
    CellIt it = box_cell(g, box);
    if (!it.ok) return;
    if (a < 4) pf[it.c] = 0.0f;
    else pd[it.c] = 0.0;
}
// whole z rows of the box's x-y range, 16-byte stores (a box that spans most of z; one GPU: the cells of those rows outside the box hold
// zeros already)
__global__ __launch_bounds__(256) void k_zero_fields4(ZeroList z, Grid g, Box box)
{
    const int a = blockIdx.y;
    float* pf = z.f4[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) pf = (a == q) ? z.f4[q] : pf;
    double* pd = z.f8[0];
#pragma unroll
    for (int q = 1; q < 7; ++q) pd = (a - 4 == q) ? z.f8[q] : pd;
    const int nzq = g.nz >> 2;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)box.nx() * box.ny() * nzq) return;
    const int iz = (int)(t % nzq) * 4, iy = (int)((t / nzq) % box.ny()) + box.y0, ix = (int)(t / ((long)nzq * box.ny())) + box.x0;
    const size_t c = g.idx(ix, iy, iz);
    if (a < 4) *(float4*)(pf + c) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    else { *(double2*)(pd + c) = make_double2(0.0, 0.0); *(double2*)(pd + c + 2) = make_double2(0.0, 0.0); }
}
void launch_zero_fields(hipStream_t st, const ZeroList& z, Grid g, Box box, bool rows)
{
    if (box.cells() <= 0) return;
    if (rows && g.nz % 4 == 0 && 4 * box.nz() >= 3 * g.nz) {
        const long n = (long)box.nx() * box.ny() * (g.nz >> 2);
        hipLaunchKernelGGL(k_zero_fields4, dim3((unsigned)((n + 255) / 256), 11), dim3(256), 0, st, z, g, box);
        return;
    }
    hipLaunchKernelGGL(k_zero_fields, dim3((unsigned)((box.cells() + 255) / 256), 11), dim3(256), 0, st, z, g, box);
}

// ---- multi-GPU, replicated solve: the P2G result of every slab gathered onto every rank -----------------------
// buf = 4 planes of box.cells() doubles [container | u | v | w], packed by k_pack_box_own (kernels_dist.hip) and SUM all-reduced
__global__ __launch_bounds__(256) void k_unpack_box(Grid g, Box box, const double* __restrict__ buf, float* __restrict__ container,
                                                    double* __restrict__ u, double* __restrict__ v, double* __restrict__ w,
                                                    double* __restrict__ ub, double* __restrict__ vb, double* __restrict__ wb)
{
    CellIt it = box_cell(g, box);
    if (!it.ok) return;
    const size_t n = (size_t)box.cells(), t = (size_t)blockIdx.x * 256 + threadIdx.x;
    container[it.c] = (float)buf[t];   // exact: a float widened, added to zeros, narrowed
    const double a = buf[n + t], b = buf[2 * n + t], c = buf[3 * n + t];
    u[it.c] = a; v[it.c] = b; w[it.c] = c;
    ub[it.c] = a; vb[it.c] = b; wb[it.c] = c;   // velBeforeUpdate (fluid.cc:1455)
}
void launch_unpack_box(hipStream_t st, Grid g, Box box, const double* buf, float* container, double* u, double* v, double* w, double* ub, double* vb,
                       double* wb)
{
    if (box.cells() > 0)
        hipLaunchKernelGGL(k_unpack_box, dim3((unsigned)((box.cells() + 255) / 256)), dim3(256), 0, st, g, box, buf, container, u, v, w, ub, vb, wb);
}

}  // namespace fl

// ---- known-answer hooks (include/fluid_hip.h) ---------------------------------------------------------------
namespace fl {
__global__ __launch_bounds__(256) void k_spline_eval(int which, long n, const double* __restrict__ x, double* __restrict__ w)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double p = x[i];
    if (which == 0) w[i] = spline(p);
    else w[i] = spline_at(p, (int)round(p) - 1 + (which - 1), which - 1);
}
void launch_spline_eval(hipStream_t st, int which, long n, const double* x, double* w)
{
    if (n > 0) hipLaunchKernelGGL(k_spline_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, which, n, x, w);
}
// per-block partials of a.b (grid-stride, wave shuffles, one value per block), then the fixed-order re-summation of the
// partials by one block: the reduction scheme of the PCG kernels (k_pcg_sq_l / k_pcg_xr_l partials + block_sum_array)
__global__ __launch_bounds__(256) void k_dot_partial(long n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ part)
{
    __shared__ double sm[4];
    double acc = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += a[i] * b[i];
    acc = block_sum<double, 4>(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_dot_final(const double* __restrict__ part, int nb, double* __restrict__ out)
{
    __shared__ double sm[5];
    const double r = block_sum_array<4>(part, nb, sm);
    if (threadIdx.x == 0) out[0] = r;
}
void launch_dot(hipStream_t st, long n, const double* a, const double* b, double* part, int nb, double* out)
{
    hipLaunchKernelGGL(k_dot_partial, dim3(nb), dim3(256), 0, st, n, a, b, part);
    hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(256), 0, st, (const double*)part, nb, out);
}
// ---- extrapolate (fluid.cc:705-802; SURVEY 8(f) row f3: dead code in the reference, an optional entry point here) --------------
// Breadth-first layers in gather form.  layer[c]: -2 = defined from the start but not a source (outside W, solid), 0 = the sources
// (weights > 0: P2G only ever writes non-solid cells inside W), k > 0 = defined by pass k, -1 = not yet.  Pass k: every cell still at
// -1 adds, on top of what it holds, the velocities of its neighbours of layer k-1 (26-neighbourhood clamped to the grid, scanned
// x, y, z ascending: the order the reference's first sweep accumulates in) and divides by their number.  A pass writes only cells
// nobody reads in that pass (readers look at layer k-1), so it works in place.
__global__ __launch_bounds__(256) void k_extrap_init(Grid g, const uint8_t* __restrict__ solid, const float* __restrict__ container, int* __restrict__ layer)
{
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= (long)g.cells()) return;
    const int iz = (int)(c % g.nz), iy = (int)((c / g.nz) % g.ny), ix = (int)(c / ((long)g.nz * g.ny));
    const int N = g.N;
    const bool outsideW = ix < 2 || ix > N - 3 || iy < 2 || iy > N - 3 || iz < 2 || iz > N - 3;
    layer[c] = (outsideW || solid[c]) ? -2 : (container[c] > 0 ? 0 : -1);
}
__global__ __launch_bounds__(256) void k_extrap_layer(Grid g, int pass, int* __restrict__ layer, double* __restrict__ u, double* __restrict__ v,
                                                      double* __restrict__ w, int* __restrict__ n_new)
{
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    int made = 0;
    if (c < (long)g.cells() && layer[c] == -1) {
        const int iz = (int)(c % g.nz), iy = (int)((c / g.nz) % g.ny), ix = (int)(c / ((long)g.nz * g.ny));
        double su = u[c], sv = v[c], sw = w[c];
        int cnt = 0;
        for (int i = ix > 0 ? ix - 1 : ix; i <= (ix < g.nx - 1 ? ix + 1 : ix); ++i)
            for (int j = iy > 0 ? iy - 1 : iy; j <= (iy < g.ny - 1 ? iy + 1 : iy); ++j)
                for (int k = iz > 0 ? iz - 1 : iz; k <= (iz < g.nz - 1 ? iz + 1 : iz); ++k) {
                    const size_t q = g.idx(i, j, k);
                    if (layer[q] == pass - 1) {
                        su = u[q] + su; sv = v[q] + sv; sw = w[q] + sw;   // fluid.cc:746
                        ++cnt;
                    }
                }
        if (cnt) {
            u[c] = su / cnt; v[c] = sv / cnt; w[c] = sw / cnt;          // fluid.cc:757 (Vec3d / int)
            layer[c] = pass;
            made = 1;
        }
    }
    if (__syncthreads_or(made) && threadIdx.x == 0) atomicAdd(n_new, 1);
}
void launch_extrap_init(hipStream_t st, Grid g, const uint8_t* solid, const float* container, int* layer)
{
    hipLaunchKernelGGL(k_extrap_init, dim3((unsigned)((g.cells() + 255) / 256)), dim3(256), 0, st, g, solid, container, layer);
}
void launch_extrap_layer(hipStream_t st, Grid g, int pass, int* layer, double* u, double* v, double* w, int* n_new)
{
    hipLaunchKernelGGL(k_extrap_layer, dim3((unsigned)((g.cells() + 255) / 256)), dim3(256), 0, st, g, pass, layer, u, v, w, n_new);
}

// ---- resample (fluid.cc:1053-1080; row f3) ---------------------------------------------------------------------------------
// After the counting sort a cell's particles are contiguous and in ascending original-index order (k_bin_rank): the particle at
// sorted position j of cell c is the (j - cell_start[c])-th of its cell in the reference's index order.  Those beyond `per_cell`
// are parked at (far, far, far) (the reference's (100, 100, 100) = boundary + 40); only cells with x coordinate < xlim are looked at.
__global__ __launch_bounds__(256) void k_resample(Grid g, long n, Particles p, const int* __restrict__ cell_start, int per_cell, int xlim, double far_,
                                                  int* __restrict__ n_parked)
{
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    int parked = 0;
    if (j < n) {
        const int rx = (int)round(p.px[j]), ry = (int)round(p.py[j]), rz = (int)round(p.pz[j]);
        const int bx = rx - g.cx0(), by = ry - g.cy0(), bz = rz - g.cz0();
        if (rx < xlim && bx >= 0 && bx < g.nx && by >= 0 && by < g.ny && bz >= 0 && bz < g.nz) {
            const long c = (long)g.idx(bx, by, bz);
            if (j - cell_start[c] >= per_cell) {
                p.px[j] = far_; p.py[j] = far_; p.pz[j] = far_;
                parked = 1;
            }
        }
    }
    const unsigned long long m = __ballot(parked);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_parked, __popcll(m));
}
void launch_resample(hipStream_t st, Grid g, long n, Particles p, const int* cell_start, int per_cell, int xlim, double far_, int* n_parked)
{
    if (n > 0) hipLaunchKernelGGL(k_resample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, n, p, cell_start, per_cell, xlim, far_, n_parked);
}

// out = a x + b y over the cells of box (dense layout)
__global__ __launch_bounds__(256) void k_axpby_box(Grid g, Box box, double a, const double* __restrict__ x, double b, const double* __restrict__ y,
                                                   double* __restrict__ out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= box.cells()) return;
    const int nz = box.nz(), ny = box.ny();
    const size_t c = g.idx(box.x0 + (int)(i / ((long)nz * ny)), box.y0 + (int)((i / nz) % ny), box.z0 + (int)(i % nz));
    out[c] = a * x[c] + b * y[c];
}
void launch_axpby_box(hipStream_t st, Grid g, Box box, double a, const double* x, double b, const double* y, double* out)
{
    if (box.cells() <= 0) return;
    hipLaunchKernelGGL(k_axpby_box, dim3((unsigned)((box.cells() + 255) / 256)), dim3(256), 0, st, g, box, a, x, b, y, out);
}

}  // namespace fl
