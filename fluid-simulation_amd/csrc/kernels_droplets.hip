// Closed pockets of the pressure system ("droplets"), solved on their own.
//
// fluid.cc:304-412 builds one matrix over every fluid cell with a non-zero diagonal and hands it to one CG solve
// (fluid.cc:624-637).  After the splash that system is block diagonal: the pool is one connected component, and every
// airborne droplet — a handful of fluid cells with no fluid neighbour outside the droplet — is a block of its own (at 256^3,
// step 445: 11 575 components, 11 415 of them with at most 64 cells, together 10 % of the unknowns but 32 % of the solver's
// tiles and 26 % of its rows; tools/spray_stats.py).  A block that shares no entry with the rest has the same solution
// whether it is solved with the rest or alone, so the small ones are taken out of the global solve:
//   k_drop_find   once per step (the flags are fixed for all passes of the step): every 8^3 block of the solver box whose core
//                 holds a possible first cell of a component (an unknown with no unknown before it on any axis) floods the
//                 unknowns of its 16^3 window from those cells (min-label propagation in LDS) and claims the components that
//                 (a) lie wholly inside the window with a free layer around them (closed: every neighbour is seen),
//                 (b) have their first cell (lowest index) in the block's own 8^3 core (one owner), (c) have <= 64 cells;
//   k_drop_clear  their count bytes are zeroed, so tile lists, row lists, the multigrid hierarchy and every PCG kernel of
//                 the step work on the system without them;
//   k_drop_solve  once per pass: one wave per droplet runs Jacobi-preconditioned CG on its <= 64 unknowns in registers (the
//                 same matrix entries, right-hand side and tolerance as the global solve) and writes the pressure cells.
// Everything is a fixed function of the flags: the cells of a droplet are sorted by index before the solve, so the sums do
// not depend on the order in which blocks or atomics happened to run.
#include "common.h"

namespace fl {

constexpr int DW = 16, DA = 4, DC = 8;            // window, apron, core
constexpr int DCELLS = DW * DW * DW;
constexpr int DROP_MAXSWEEPS = 10;   // cells a pocket may reach from its first cell (path length)

__device__ __forceinline__ double wsum_d(double v)   // every lane gets the total; fixed order
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Slots: the buffers hold DROP_CAP components in DROP_NCTR ranges of DROP_CAP / DROP_NCTR, each filled through a counter of its own
// (a block adds its claims to counter blockIdx % DROP_NCTR with ONE atomic: ten thousand returning atomics on a single
// address cost more than the whole search).  The counters lie 256 bytes apart.
// own: the cells (local-box coordinates, inclusive) a claimed component must lie in — the whole box on one GPU; the rank's OWNED cells in a
// decomposed run (a pocket that straddles a cut stays in the global solve; the pressure of a claimed one is written by its one owner)
__global__ __launch_bounds__(512) void k_drop_find(LBox L, Box own, const uint8_t* __restrict__ cnt, int ncx, int ncy, int ncz,
                                                   int* __restrict__ ctr, int* __restrict__ comp_n, int* __restrict__ comp_cells)
{
    __shared__ int s_claims, s_base, s_nu, s_wclaims[8];
    __shared__ uint16_t ulist[DCELLS];   // the unknowns of the window: the sweeps visit these only
    constexpr unsigned AIR = 0xFFFFu, NEW = 0xFFFEu;   // not an unknown / an unknown the search has not reached
    __shared__ uint16_t lab[DCELLS], lab2[DCELLS];
    // per label = per seed = per cell of the core (index core_of(label))
    __shared__ uint8_t bad[DC * DC * DC], dead[DC * DC * DC], seedrow[DC * DC];
    __shared__ int meta[DC * DC * DC];    // number of cells, then (claimed) ~slot
    __shared__ int fill[DC * DC * DC];    // cells emitted so far
    auto core_of = [](unsigned w) { return (int)((((w >> 8) - DA) * DC + (((w >> 4) & 15) - DA)) * DC + ((w & 15) - DA)); };
    const int tid = threadIdx.x;
    const int bz = blockIdx.x % ncz, by = (blockIdx.x / ncz) % ncy, bx = blockIdx.x / (ncz * ncy);
    const int i0 = 1 + bx * DC - DA, j0 = 1 + by * DC - DA, k0 = LBOX_K0 + bz * DC - DA;   // window origin (local box coordinates)
    const long sx = (long)L.Ly * L.Lz, sy = L.Lz;
    // The first cell of a component (lowest index) has no unknown before it on any axis.  A block whose core holds no such cell
    // owns nothing — that is every block inside the pool and nearly every block on its surface (the cell below is water).
    // (one wave reads the core and the cells before it as 8-byte words: a row of the core is one aligned word of the count array,
    // whose rows are multiples of 16 bytes and start 16 bytes in)
    const int ci = tid >> 6, cj = (tid >> 3) & 7, ck = tid & 7;
    if (tid < 64) {
        const int i = i0 + DA + (tid >> 3), j = j0 + DA + (tid & 7), kb = k0 + DA;
        unsigned row = 0;
        if (i < L.Lx && j < L.Ly && kb + 7 < L.Lz) {
            const uint8_t* p = cnt + i * sx + j * sy + kb;
            const uint64_t c = *reinterpret_cast<const uint64_t*>(p), cx = *reinterpret_cast<const uint64_t*>(p - sx),
                           cy = *reinterpret_cast<const uint64_t*>(p - sy);
            unsigned before = p[-1];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned me = (unsigned)(c >> (8 * q)) & 0xff;
                if (me && !((cx >> (8 * q)) & 0xff) && !((cy >> (8 * q)) & 0xff) && !before) row |= 1u << q;
                before = me;
            }
        }
        seedrow[tid] = (uint8_t)row;
    }
    if (tid == 0) s_nu = 0;
    if (!__syncthreads_or(tid < 64 && seedrow[tid] != 0)) return;
    const bool seed = (seedrow[ci * DC + cj] >> ck) & 1;
    for (int id = tid; id < DCELLS / 4; id += 512) {   // the window, four cells (one aligned word) at a time
        const int r = id >> 2, q = id & 3, wi = r >> 4, wj = r & 15;
        const int i = i0 + wi, j = j0 + wj, k = k0 + 4 * q;
        unsigned word = 0;
        if (i >= 0 && i < L.Lx && j >= 0 && j < L.Ly && k + 3 < L.Lz) word = *reinterpret_cast<const unsigned*>(cnt + i * sx + j * sy + k);
        int mine = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) mine += ((word >> (8 * e)) & 0xff) != 0;
        int at = mine ? atomicAdd(&s_nu, mine) : 0;   // (the order of the list does not reach the result: the sweeps are synchronous)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool un = ((word >> (8 * e)) & 0xff) != 0;
            const int w = r * DW + 4 * q + e;
            lab[w] = lab2[w] = un ? (uint16_t)NEW : (uint16_t)AIR;
            if (un) ulist[at++] = (uint16_t)w;
        }
    }
    bad[tid] = 0;
    meta[tid] = 0;
    fill[tid] = 0;
    __syncthreads();
    const int nu = s_nu;
    if (seed) {
        const int w = ((DA + ci) * DW + DA + cj) * DW + DA + ck;
        lab[w] = lab2[w] = (uint16_t)w;
    }
    __syncthreads();
    // The search spreads from the seeds one cell per sweep (every sweep reads the labels of the sweep before: which cells a
    // sweep reaches does not depend on the order the threads run in): a reached cell takes the lowest label among itself and
    // its reached neighbours.  A droplet is done in a few sweeps and the loop ends; a seed that hangs on the pool, or a pocket
    // that winds further than DROP_MAXSWEEPS cells from its first cell, is found out below and stays in the global solve.
    // A label that has reached more than 64 cells cannot be claimed any more (dead: its flood is the pool's); the sweeps end
    // when no live label moved.  The counts are read as they stood at the end of the sweep before, like the labels.
    uint16_t* cur = lab;
    uint16_t* nxt = lab2;
    if (seed) meta[(ci * DC + cj) * DC + ck] = 1;
    for (int sweep = 0; sweep < DROP_MAXSWEEPS; ++sweep) {
        __syncthreads();
        dead[tid] = meta[tid] > 64;
        __syncthreads();
        int changed = 0;
        for (int q = tid; q < nu; q += 512) {
            const int w = ulist[q];
            const unsigned l = cur[w];
            const int wi = w >> 8, wj = (w >> 4) & 15, wk = w & 15;
            unsigned m = l;   // NEW is larger than every label: min() ignores unreached neighbours and reaches an unreached cell
            if (wk > 0) m = min(m, (unsigned)cur[w - 1]);
            if (wk < DW - 1) m = min(m, (unsigned)cur[w + 1]);
            if (wj > 0) m = min(m, (unsigned)cur[w - DW]);
            if (wj < DW - 1) m = min(m, (unsigned)cur[w + DW]);
            if (wi > 0) m = min(m, (unsigned)cur[w - DW * DW]);
            if (wi < DW - 1) m = min(m, (unsigned)cur[w + DW * DW]);
            if (m < l) {
                if (l == NEW) atomicAdd(&meta[core_of(m)], 1);
                changed |= !dead[core_of(m)];
            }
            nxt[w] = (uint16_t)m;
        }
        uint16_t* t = cur; cur = nxt; nxt = t;
        if (!__syncthreads_or(changed)) break;
    }
    __syncthreads();
    meta[tid] = 0;   // (counted again below, by final label)
    __syncthreads();
    if (cur != lab) {   // (block-uniform) the labels of the last sweep go back into lab
        for (int q = tid; q < nu; q += 512) lab[ulist[q]] = cur[ulist[q]];
        __syncthreads();
    }
    // A label is bad if one of its cells (a) lies on the window's outer layer (the component may go on outside), (b) touches an
    // unknown with another label or none (the search has not finished there: every label region of an unfinished component
    // has such a border), or (c) comes before the label's own cell (the component's first cell is not a seed of this core:
    // another block owns it).  A label without a bad cell is a whole component whose first cell lies in this core.
    for (int q = tid; q < nu; q += 512) {
        const int w = ulist[q];
        const unsigned l = lab[w];
        if (l >= NEW) continue;
        const int wi = w >> 8, wj = (w >> 4) & 15, wk = w & 15;
        bool b = wi == 0 || wi == DW - 1 || wj == 0 || wj == DW - 1 || wk == 0 || wk == DW - 1 || (unsigned)w < l;
        b = b || i0 + wi < own.x0 || i0 + wi > own.x1 || j0 + wj < own.y0 || j0 + wj > own.y1 || k0 + wk < own.z0 || k0 + wk > own.z1;
        if (!b) {
            const unsigned n0 = lab[w - 1], n1 = lab[w + 1], n2 = lab[w - DW], n3 = lab[w + DW], n4 = lab[w - DW * DW], n5 = lab[w + DW * DW];
            b = (n0 != AIR && n0 != l) || (n1 != AIR && n1 != l) || (n2 != AIR && n2 != l) || (n3 != AIR && n3 != l) ||
                (n4 != AIR && n4 != l) || (n5 != AIR && n5 != l);
        }
        if (b) bad[core_of(l)] = 1;
        atomicAdd(&meta[core_of(l)], 1);
    }
    __syncthreads();
    int my_root = -1, my_n = 0, my_local = 0;   // thread = core cell: a root if it kept its own label
    {
        const int w = ((DA + ci) * DW + DA + cj) * DW + DA + ck;
        if (lab[w] == (unsigned)w) {
            const int n = meta[tid];
            meta[tid] = 0;
            if (!bad[tid] && n <= 64) { my_root = tid; my_n = n; }
        }
    }
    {   // the claims of a block are numbered in the order of their core cells (not in the order atomics arrive): when a range of the
        // buffers overflows, the same roots of the block are dropped in every run
        const unsigned long long bal = __ballot(my_root >= 0);
        const int lane = tid & 63, wv = tid >> 6;
        if (lane == 0) s_wclaims[wv] = __popcll(bal);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) { before += q < wv ? s_wclaims[q] : 0; all += s_wclaims[q]; }
        my_local = before + __popcll(bal & ((1ull << lane) - 1ull));
        if (tid == 0) s_claims = all;
    }
    __syncthreads();
    constexpr int RANGE = DROP_CAP / DROP_NCTR;
    const int c = blockIdx.x % DROP_NCTR;
    if (tid == 0) s_base = s_claims > 0 ? atomicAdd(&ctr[c * 64], s_claims) : 0;
    __syncthreads();
    if (s_claims == 0) return;
    if (my_root >= 0 && s_base + my_local < RANGE) {
        const int slot = c * RANGE + s_base + my_local;
        comp_n[slot] = my_n;
        meta[my_root] = ~slot;
    }
    __syncthreads();
    for (int q = tid; q < nu; q += 512) {
        const int w = ulist[q];
        const unsigned l = lab[w];
        if (l >= NEW || meta[core_of(l)] >= 0) continue;
        const int slot = ~meta[core_of(l)];
        const int pos = atomicAdd(&fill[core_of(l)], 1);
        const int wi = w >> 8, wj = (w >> 4) & 15, wk = w & 15;
        comp_cells[(size_t)slot * 64 + pos] = (int)((i0 + wi) * sx + (j0 + wj) * sy + (k0 + wk));
    }
}

__global__ __launch_bounds__(256) void k_drop_clear(const int* __restrict__ ctr, const int* __restrict__ comp_n, const int* __restrict__ comp_cells,
                                                    uint8_t* __restrict__ cnt, int* __restrict__ pre, int* __restrict__ total)
{
    constexpr int RANGE = DROP_CAP / DROP_NCTR;
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // where each range starts in the dense numbering of the solve
        int acc = 0;
        for (int c = 0; c < DROP_NCTR; ++c) {
            pre[c] = acc;
            acc += min(ctr[c * 64], RANGE);
        }
        pre[DROP_NCTR] = acc;
        *total = acc;
    }
    const int t = blockIdx.x * 256 + threadIdx.x, slot = t >> 6, pos = t & 63;
    if ((slot % RANGE) >= min(ctr[(slot / RANGE) * 64], RANGE) || pos >= comp_n[slot]) return;
    cnt[comp_cells[t]] = 0;
}

// one wave per droplet
__global__ __launch_bounds__(256) void k_drop_solve(Grid g, LBox L, int n_comp, const int* __restrict__ pre, const int* __restrict__ comp_n, const int* __restrict__ comp_cells,
                                                    const uint8_t* __restrict__ flags, const float* __restrict__ b, Coef<double> cf, double tol,
                                                    double* __restrict__ pressure, double* __restrict__ keep, int* __restrict__ n_fail)
{
    __shared__ int skey[4][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int dense = blockIdx.x * 4 + wv;
    if (dense >= n_comp) return;
    int slot;
    {   // the range that holds the dense index (pre[] ascends)
        const int below = __popcll(__ballot(lane < DROP_NCTR && pre[lane < DROP_NCTR ? lane : 0] <= dense));   // ranges starting at or before it
        const int c = below - 1;
        slot = c * (DROP_CAP / DROP_NCTR) + dense - pre[c];
    }
    const int n = comp_n[slot];
    int key = lane < n ? comp_cells[(size_t)slot * 64 + lane] : 0x7fffffff;
    // bitonic sort of the 64 keys across the wave (ascending): the order of the unknowns is the order of their cells
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int other = __shfl_xor(key, j, 64);
            const bool up = (lane & k) == 0, lower = (lane & j) == 0;
            key = (lower == up) ? min(key, other) : max(key, other);
        }
    skey[wv][lane] = key;
    __builtin_amdgcn_wave_barrier();
    const bool live = lane < n;
    const long sx = (long)L.Ly * L.Lz, sy = L.Lz;
    auto find = [&](long t) {   // lane that holds cell t, or -1
        int lo = 0, hi = n - 1, r = -1;
#pragma unroll 1
        while (lo <= hi) {
            const int mid = (lo + hi) >> 1, v = skey[wv][mid];
            if (v == t) { r = mid; break; }
            if (v < t) lo = mid + 1; else hi = mid - 1;
        }
        return r;
    };
    int nb[6] = {-1, -1, -1, -1, -1, -1};
    double diag = 1, inv = 1, bv = 0;
    size_t c = 0;
    if (live) {
        const long t = key;
        nb[0] = find(t - sx); nb[1] = find(t + sx); nb[2] = find(t - sy); nb[3] = find(t + sy); nb[4] = find(t - 1); nb[5] = find(t + 1);
        const int k = (int)(t % L.Lz), j = (int)((t / L.Lz) % L.Ly), i = (int)(t / sx);
        c = g.idx(L.x0 + i - 1, L.y0 + j - 1, L.z0 + k - LBOX_K0);
        const int cntv = flags[c] >> F_CNT_SHIFT;
        diag = cf.diag[cntv];
        inv = cf.inv[cntv];
        bv = (double)b[c];
    }
    const double off = cf.off;
    // ConjugateGradient.h:28-90 with the diagonal preconditioner, x0 = 0
    double x = 0, r = bv, z = r * inv, s = z;
    const double bb = wsum_d(bv * bv);
    double rz = wsum_d(r * z), rr = bb;
    const double thr = tol * tol * bb;
    if (bb > 0) {
#pragma unroll 1
        for (int it = 0; it < n + 8 && rr > thr; ++it) {
            double q = diag * s;
            double acc = 0;
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                const double sn = __shfl(s, nb[d] < 0 ? lane : nb[d], 64);
                acc += nb[d] < 0 ? 0.0 : sn;
            }
            q = live ? q + off * acc : 0.0;
            const double pq = wsum_d(s * q);
            if (!(pq > 0)) break;
            const double alpha = rz / pq;
            x += alpha * s;
            r -= alpha * q;
            z = r * inv;
            rr = wsum_d(r * r);
            const double rzn = wsum_d(r * z);
            if (!(rzn > 0)) break;
            s = z + (rzn / rz) * s;
            rz = rzn;
        }
    }
    if (live) {
        pressure[c] = x;
        if (keep) keep[c] = x;
    }
    // a droplet that left the loop by its iteration cap or a breakdown, not by the stopping rule: counted, fluid_step reports it
    if (lane == 0 && bb > 0 && !(rr <= thr) && n_fail) atomicAdd(n_fail, 1);
}

// cells: 64 ints per component; blocks of 8^3 over the interior of the local box.  ctr: 64 * DROP_NCTR ints; pre: DROP_NCTR + 1 ints
void launch_drop_find(hipStream_t st, LBox L, uint8_t* cnt, int* ctr, int* pre, int* total, int* comp_n, int* comp_cells, const Box* own)
{
    const Box all{0, 0, 0, L.Lx - 1, L.Ly - 1, L.Lz - 1};
    const int ncx = (L.nx + DC - 1) / DC, ncy = (L.ny + DC - 1) / DC, ncz = (L.nz + DC - 1) / DC;
    hipMemsetAsync(ctr, 0, (size_t)64 * DROP_NCTR * sizeof(int), st);
    hipLaunchKernelGGL(k_drop_find, dim3((unsigned)(ncx * ncy * ncz)), dim3(512), 0, st, L, own ? *own : all, cnt, ncx, ncy, ncz, ctr, comp_n, comp_cells);
    hipLaunchKernelGGL(k_drop_clear, dim3((unsigned)((size_t)DROP_CAP * 64 / 256)), dim3(256), 0, st, ctr, comp_n, comp_cells, cnt, pre, total);
}
void launch_drop_solve(hipStream_t st, Grid g, LBox L, int n_comp, const int* pre, const int* comp_n, const int* comp_cells, const uint8_t* flags,
                       const float* b, Coef<double> cf, double tol, double* pressure, double* keep, int* n_fail)
{
    if (n_comp <= 0) return;
    hipLaunchKernelGGL(k_drop_solve, dim3((unsigned)((n_comp + 3) / 4)), dim3(256), 0, st, g, L, n_comp, pre, comp_n, comp_cells, flags, b, cf, tol, pressure, keep, n_fail);
}

}  // namespace fl
