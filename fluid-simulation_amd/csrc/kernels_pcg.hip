// Matrix-free pressure solve for gfx950: the reference's sparse system (setA/setA2,
// fluid.cc:304-412,481-541) is a function of the per-cell flag byte alone, so the Eigen
// SparseMatrix + IncompleteCholesky PCG (fluid.cc:1352,1473-1474) becomes a PCG whose SpMV is
// an LDS-tiled 7-point stencil.  Loop structure and stopping rule follow
// Eigen/src/IterativeLinearSolvers/ConjugateGradient.h:28-90 with the diagonal (Jacobi)
// preconditioner; two launches per iteration, all scalars stay on the device:
//
//   SQ: [beta from the previous launch's partial sums]  s' = r/diag + beta s ;  q = A s' ;
//       partial s'.q                                           (4T+1 bytes per cell)
//   XR: [alpha = r.z / s.q]  x += alpha s ; r -= alpha q ; partial r.r, r.(r/diag)
//                                                               (6T+1 bytes per cell)
//
// Every block of a launch re-sums the previous launch's per-block partials in a fixed order
// (<= 2048 doubles from L2) instead of a grid-wide atomic or an extra reduction launch, so all
// blocks take the same branch and results are run-to-run reproducible.
//
// Bandwidth-bound integer/fp work: no MFMA.  Tile = 4 x 4 x 64 cells (x,y,z; z fastest ->
// one 64-lane wave per 512-byte row), halo staged in LDS, x/y/z neighbours read from LDS.
#include "common.h"

namespace fl {

constexpr int TX = 4, TY = 4, TZ = 64;
constexpr int LY = TY + 2, LZ = TZ + 2;
constexpr int LDS_CELLS = (TX + 2) * LY * LZ;
constexpr int SQ_MAX_BLOCKS = 2048;
constexpr int XR_MAX_BLOCKS = 1024;

__device__ __forceinline__ bool active(uint8_t f) { return (f & F_FLUID) && (f >> F_CNT_SHIFT); }

// Blocks are dealt round-robin over the 8 XCDs; give each XCD (b % 8) a contiguous range of
// virtual block ids so that neighbouring tiles share one L2 (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int b, int nb)
{
    const int q = nb >> 3, r = nb & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// static indices only: a runtime index into a by-value kernel argument would go through scratch
template <typename T>
__device__ __forceinline__ void load_diag(T* sdiag, const Coef<T>& cf)
{
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) sdiag[i] = cf.diag[i];
    }
}

struct Tiles {
    int ntx, nty, ntz;
    __host__ __device__ int count() const { return ntx * nty * ntz; }
};
static inline Tiles make_tiles(const Box& b)
{
    Tiles t;
    t.ntx = (b.nx() + TX - 1) / TX;
    t.nty = (b.ny() + TY - 1) / TY;
    t.ntz = (b.nz() + TZ - 1) / TZ;
    return t;
}

// FUSED = true : the SQ kernel of the PCG.   FUSED = false : q = A s only (s read from s_in).
template <typename T, bool FUSED>
__global__ __launch_bounds__(256) void k_stencil(Grid g, Box box, Tiles tl, const uint8_t* __restrict__ flags, const T* __restrict__ r,
                                                 const T* __restrict__ s_in, T* __restrict__ s_out, T* __restrict__ q, Coef<T> cf,
                                                 const double* __restrict__ part_rr, const double* __restrict__ part_rz_new,
                                                 const double* __restrict__ part_rz_old, double* __restrict__ part_pq, int n_prev,
                                                 PcgState* ps, int first, double tol)
{
    __shared__ T sT[LDS_CELLS];
    __shared__ double red[8];
    __shared__ int s_done;
    __shared__ T sdiag[8];  // dynamic index by diag count: LDS, not a kernarg select chain
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    load_diag(sdiag, cf);
    T beta = 0;
    if (FUSED) {
        // one read per block, broadcast: block 0 of THIS launch may set done while we start
        if (tid == 0) s_done = ps->done;
        __syncthreads();
        if (s_done) return;
        if (first) {
            // ConjugateGradient.h:45-60: b == 0 -> x = 0, done.
            const double bb = block_sum_array<4>(part_rr, n_prev, red);
            if (blockIdx.x == 0 && tid == 0) {
                ps->bb = bb;
                ps->thr = tol * tol * bb;
                ps->rr = bb;
                if (!(bb > 0)) ps->done = 1;
            }
            if (!(bb > 0)) return;
        } else {
            const double rr = block_sum_array<4>(part_rr, n_prev, red);
            if (rr < ps->thr) {  // ConjugateGradient.h:75-77: break before i++
                if (blockIdx.x == 0 && tid == 0) { ps->rr = rr; ps->done = 1; }
                return;
            }
            const double rzn = block_sum_array<4>(part_rz_new, n_prev, red);
            const double rzo = block_sum_array<4>(part_rz_old, n_prev, red);
            beta = (T)(rzn / rzo);  // :82-83
            if (blockIdx.x == 0 && tid == 0) { ps->rr = rr; ps->iters += 1; }  // :85
        }
    }
    const int N = g.N;
    const long sxl = (long)N * N;
    double acc = 0;
    const int ntiles = tl.count();
    for (int tile = xcd_remap(blockIdx.x, gridDim.x); tile < ntiles; tile += gridDim.x) {
        const int tz = tile % tl.ntz, ty = (tile / tl.ntz) % tl.nty, tx = tile / (tl.ntz * tl.nty);
        const int x0 = box.x0 + tx * TX, y0 = box.y0 + ty * TY, z0 = box.z0 + tz * TZ;
        __syncthreads();  // LDS reuse across tiles
        T sc[4];
        uint8_t fc[4];
        unsigned inb = 0;
        // interior rows: wave wv owns rows wv, wv+4, wv+8, wv+12 (row = lx*TY + ly)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = wv + 4 * k, lx = row / TY, ly = row % TY;
            const int gx = x0 + lx, gy = y0 + ly, gz = z0 + lane;
            const bool in = gx <= box.x1 && gy <= box.y1 && gz <= box.z1;
            T val = 0;
            uint8_t f = 0;
            if (in) {
                const size_t c = (size_t)gx * sxl + (size_t)gy * N + gz;
                f = flags[c];
                if (active(f)) {
                    if (FUSED) {
                        val = r[c] / sdiag[f >> F_CNT_SHIFT];
                        if (!first) val = val + beta * s_in[c];
                    } else {
                        val = s_in[c];
                    }
                }
                if (FUSED) s_out[c] = val;
                inb |= 1u << k;
            }
            sc[k] = val;
            fc[k] = f;
            sT[((lx + 1) * LY + (ly + 1)) * LZ + lane + 1] = val;
        }
        // x/y face halo rows: 16 rows, 4 per wave
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int h = wv * 4 + k, face = h >> 2, j = h & 3;
            const int lx = face == 0 ? -1 : (face == 1 ? TX : j);
            const int ly = face == 2 ? -1 : (face == 3 ? TY : j);
            const int gx = x0 + lx, gy = y0 + ly, gz = z0 + lane;
            T val = 0;
            if (gx >= 0 && gx < N && gy >= 0 && gy < N && gz < N) {
                const size_t c = (size_t)gx * sxl + (size_t)gy * N + gz;
                const uint8_t f = flags[c];
                if (active(f)) {
                    if (FUSED) {
                        val = r[c] / sdiag[f >> F_CNT_SHIFT];
                        if (!first) val = val + beta * s_in[c];
                    } else {
                        val = s_in[c];
                    }
                }
            }
            sT[((lx + 1) * LY + (ly + 1)) * LZ + lane + 1] = val;
        }
        // z halo of the 16 interior rows: 32 cells
        if (tid < 32) {
            const int row = tid >> 1, side = tid & 1, lx = row / TY, ly = row % TY;
            const int gx = x0 + lx, gy = y0 + ly, gz = side ? z0 + TZ : z0 - 1;
            T val = 0;
            if (gx < N && gy < N && gz >= 0 && gz < N) {
                const size_t c = (size_t)gx * sxl + (size_t)gy * N + gz;
                const uint8_t f = flags[c];
                if (active(f)) {
                    if (FUSED) {
                        val = r[c] / sdiag[f >> F_CNT_SHIFT];
                        if (!first) val = val + beta * s_in[c];
                    } else {
                        val = s_in[c];
                    }
                }
            }
            sT[((lx + 1) * LY + (ly + 1)) * LZ + (side ? TZ + 1 : 0)] = val;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (inb & (1u << k)) {
                const int row = wv + 4 * k, lx = row / TY, ly = row % TY;
                const size_t c = (size_t)(x0 + lx) * sxl + (size_t)(y0 + ly) * N + (z0 + lane);
                T qv = 0;
                if (active(fc[k])) {
                    const int o = ((lx + 1) * LY + (ly + 1)) * LZ + lane + 1;
                    const T nb = sT[o - LY * LZ] + sT[o + LY * LZ] + sT[o - LZ] + sT[o + LZ] + sT[o - 1] + sT[o + 1];
                    qv = sdiag[fc[k] >> F_CNT_SHIFT] * sc[k] + cf.off * nb;
                    acc += (double)sc[k] * (double)qv;
                }
                q[c] = qv;
            }
        }
    }
    if (FUSED) {
        acc = block_sum<double, 4>(acc, red);
        if (tid == 0) part_pq[blockIdx.x] = acc;
    }
}

// XR kernel: alpha, x += alpha s, r -= alpha q, partial |r|^2 and r.(r/diag)
// ConjugateGradient.h:70-74,79-81.
template <typename T>
__global__ __launch_bounds__(256) void k_pcg_xr(Grid g, Box box, const uint8_t* __restrict__ flags, T* __restrict__ x, T* __restrict__ r,
                                                const T* __restrict__ s, const T* __restrict__ q, Coef<T> cf,
                                                const double* __restrict__ part_rz_cur, int n_xr, const double* __restrict__ part_pq,
                                                int n_sq, double* __restrict__ part_rr, double* __restrict__ part_rz_next, PcgState* ps)
{
    __shared__ double red[8];
    __shared__ int s_done;
    __shared__ T sdiag[8];
    load_diag(sdiag, cf);
    if (threadIdx.x == 0) s_done = ps->done;
    __syncthreads();
    if (s_done) return;
    const double rz = block_sum_array<4>(part_rz_cur, n_xr, red);
    const double pq = block_sum_array<4>(part_pq, n_sq, red);
    if (!(pq > 0) || !(rz == rz)) {  // not SPD / NaN: stop instead of spreading NaNs
        if (blockIdx.x == 0 && threadIdx.x == 0) { ps->breakdown = 1; ps->done = 1; }
        return;
    }
    const T alpha = (T)(rz / pq);
    const long ncells = box.cells();
    const int nz = box.nz(), ny = box.ny();
    double arr = 0, arz = 0;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < ncells; t += (long)gridDim.x * 256) {
        const int iz = (int)(t % nz) + box.z0, iy = (int)((t / nz) % ny) + box.y0, ix = (int)(t / ((long)nz * ny)) + box.x0;
        const size_t c = g.idx(ix, iy, iz);
        const uint8_t f = flags[c];
        if (active(f)) {
            const T xn = x[c] + alpha * s[c];
            const T rn = r[c] - alpha * q[c];
            x[c] = xn;
            r[c] = rn;
            const T z = rn / sdiag[f >> F_CNT_SHIFT];
            arr += (double)rn * (double)rn;
            arz += (double)rn * (double)z;
        }
    }
    arr = block_sum<double, 4>(arr, red);
    arz = block_sum<double, 4>(arz, red);
    if (threadIdx.x == 0) { part_rr[blockIdx.x] = arr; part_rz_next[blockIdx.x] = arz; }
}

// x = 0, r = b (ConjugateGradient.h:41: residual = rhs - A*0), partial |b|^2 and r.(r/diag) (:62-65)
template <typename T>
__global__ __launch_bounds__(256) void k_pcg_init(Grid g, Box box, const uint8_t* __restrict__ flags, const float* __restrict__ b,
                                                  T* __restrict__ x, T* __restrict__ r, Coef<T> cf, double* __restrict__ part_bb,
                                                  double* __restrict__ part_rz0, PcgState* ps)
{
    __shared__ double red[8];
    __shared__ T sdiag[8];
    load_diag(sdiag, cf);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) { ps->done = 0; ps->iters = 0; ps->breakdown = 0; ps->bb = 0; ps->thr = 0; ps->rr = 0; }
    const long ncells = box.cells();
    const int nz = box.nz(), ny = box.ny();
    double abb = 0, arz = 0;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < ncells; t += (long)gridDim.x * 256) {
        const int iz = (int)(t % nz) + box.z0, iy = (int)((t / nz) % ny) + box.y0, ix = (int)(t / ((long)nz * ny)) + box.x0;
        const size_t c = g.idx(ix, iy, iz);
        const uint8_t f = flags[c];
        T rv = 0;
        if (active(f)) {
            rv = (T)b[c];
            const T z = rv / sdiag[f >> F_CNT_SHIFT];
            abb += (double)rv * (double)rv;
            arz += (double)rv * (double)z;
        }
        x[c] = 0;
        r[c] = rv;
    }
    abb = block_sum<double, 4>(abb, red);
    arz = block_sum<double, 4>(arz, red);
    if (threadIdx.x == 0) { part_bb[blockIdx.x] = abb; part_rz0[blockIdx.x] = arz; }
}

template <typename T>
__global__ __launch_bounds__(256) void k_store_pressure(Grid g, Box box, const uint8_t* __restrict__ flags, const T* __restrict__ x,
                                                        double* __restrict__ pressure)
{
    const long ncells = box.cells();
    const int nz = box.nz(), ny = box.ny();
    long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= ncells) return;
    const int iz = (int)(t % nz) + box.z0, iy = (int)((t / nz) % ny) + box.y0, ix = (int)(t / ((long)nz * ny)) + box.x0;
    const size_t c = g.idx(ix, iy, iz);
    pressure[c] = active(flags[c]) ? (double)x[c] : 0.0;
}

int pcg_sq_blocks(Box box)
{
    int n = make_tiles(box).count();
    return n < SQ_MAX_BLOCKS ? (n < 1 ? 1 : n) : SQ_MAX_BLOCKS;
}
int pcg_xr_blocks(Box box)
{
    long n = (box.cells() + 255) / 256;
    return (int)(n < XR_MAX_BLOCKS ? (n < 1 ? 1 : n) : XR_MAX_BLOCKS);
}

template <typename T>
void launch_pcg_init(hipStream_t st, Grid g, Box box, const uint8_t* flags, const float* b, T* x, T* r, Coef<T> cf, double* part_bb,
                     double* part_rz0, PcgState* ps)
{
    hipLaunchKernelGGL((k_pcg_init<T>), dim3(pcg_xr_blocks(box)), dim3(256), 0, st, g, box, flags, b, x, r, cf, part_bb, part_rz0, ps);
}
template <typename T>
void launch_pcg_sq(hipStream_t st, Grid g, Box box, const uint8_t* flags, const T* r, const T* s_in, T* s_out, T* q, Coef<T> cf,
                   const double* part_rr, const double* part_rz_new, const double* part_rz_old, double* part_pq, PcgState* ps, int first,
                   double tol)
{
    hipLaunchKernelGGL((k_stencil<T, true>), dim3(pcg_sq_blocks(box)), dim3(256), 0, st, g, box, make_tiles(box), flags, r, s_in, s_out, q,
                       cf, part_rr, part_rz_new, part_rz_old, part_pq, pcg_xr_blocks(box), ps, first, tol);
}
template <typename T>
void launch_pcg_xr(hipStream_t st, Grid g, Box box, const uint8_t* flags, T* x, T* r, const T* s, const T* q, Coef<T> cf,
                   const double* part_rz_cur, const double* part_pq, double* part_rr, double* part_rz_next, PcgState* ps)
{
    hipLaunchKernelGGL((k_pcg_xr<T>), dim3(pcg_xr_blocks(box)), dim3(256), 0, st, g, box, flags, x, r, s, q, cf, part_rz_cur,
                       pcg_xr_blocks(box), part_pq, pcg_sq_blocks(box), part_rr, part_rz_next, ps);
}
template <typename T>
void launch_stencil_apply(hipStream_t st, Grid g, Box box, const uint8_t* flags, const T* s, T* q, Coef<T> cf)
{
    hipLaunchKernelGGL((k_stencil<T, false>), dim3(pcg_sq_blocks(box)), dim3(256), 0, st, g, box, make_tiles(box), flags, (const T*)nullptr,
                       s, (T*)nullptr, q, cf, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (double*)nullptr, 0,
                       (PcgState*)nullptr, 0, 0.0);
}
template <typename T>
void launch_store_pressure(hipStream_t st, Grid g, Box box, const uint8_t* flags, const T* x, double* pressure)
{
    hipLaunchKernelGGL((k_store_pressure<T>), dim3((unsigned)((box.cells() + 255) / 256)), dim3(256), 0, st, g, box, flags, x, pressure);
}

#define INST(T)                                                                                                                          \
    template void launch_pcg_init<T>(hipStream_t, Grid, Box, const uint8_t*, const float*, T*, T*, Coef<T>, double*, double*, PcgState*); \
    template void launch_pcg_sq<T>(hipStream_t, Grid, Box, const uint8_t*, const T*, const T*, T*, T*, Coef<T>, const double*,           \
                                   const double*, const double*, double*, PcgState*, int, double);                                       \
    template void launch_pcg_xr<T>(hipStream_t, Grid, Box, const uint8_t*, T*, T*, const T*, const T*, Coef<T>, const double*,           \
                                   const double*, double*, double*, PcgState*);                                                          \
    template void launch_stencil_apply<T>(hipStream_t, Grid, Box, const uint8_t*, const T*, T*, Coef<T>);                                \
    template void launch_store_pressure<T>(hipStream_t, Grid, Box, const uint8_t*, const T*, double*);
INST(double)
INST(float)

}  // namespace fl
