// Particle-side kernels of the PIC/FLIP step for gfx950 (wave64).
//
// The reference scatters from particles to cells under one std::mutex per cell
// (fluid.cc:265-299, 843-882).  Here particles are counting-sorted by base cell every step and
// the transfer is a GATHER per cell over the 27 neighbouring cell lists: no atomics on the
// fields, and the summation order is a pure function of the input (bitwise reproducible).
#include "common.h"

namespace fl {

// ---- counting sort by base cell ----------------------------------------------------------
// key = linear index of round(pos) (C round(): half away from zero, fluid.cc:267-269), or
// NCELL for a particle whose base cell is off the grid (it can reach no cell:
// its clamped support lies in the solid shell, fluid.cc:271-276,288).
__global__ __launch_bounds__(256) void k_bin_count(Grid g, long n, Particles p, int* __restrict__ key, int* __restrict__ slot,
                                                   int* __restrict__ cell_count, int* __restrict__ part)
{
    // per-block bbox partials (no same-address atomics: 80k waves hammering 6 words cost 5.6 ms)
    __shared__ int sm[4][8];
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {-1, -1, -1};
    int nout = 0, cmax = 0;
    const long ncell = (long)g.cells();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        int bx = (int)round(p.px[i]) - g.cx0(), by = (int)round(p.py[i]) - g.cy0(), bz = (int)round(p.pz[i]) - g.cz0();
        bool in = bx >= 0 && bx < g.nx && by >= 0 && by < g.ny && bz >= 0 && bz < g.nz;
        int k = in ? (int)g.idx(bx, by, bz) : (int)ncell;
        const bool dead = p.pid[i] == PID_DEAD;  // migrated to a neighbour rank: last bucket, dropped after the sort
        if (dead) { k = (int)ncell + 1; in = false; }
        key[i] = k;
        // Particles are still in last step's cell order, so neighbouring lanes mostly share a key: one returning
        // atomic per RUN of equal keys in the wave (its leader adds the run length, the others take base + offset)
        // instead of one per particle.  Any slot order inside a cell will do: k_bin_rank re-ranks a cell by id.
        {
            const int lane = threadIdx.x & 63;
            const int kp = __shfl_up(k, 1, 64);
            const bool head = lane == 0 || kp != k;
            const unsigned long long hm = __ballot(head);  // lanes past n are not here: their bits are 0
            const unsigned long long below = hm & ((2ull << lane) - 1ull);          // heads at or below me
            const int start = 63 - __clzll((long long)below);
            const unsigned long long above = (lane == 63) ? 0ull : (hm >> (lane + 1)) << (lane + 1);
            const unsigned long long act = __ballot(true);
            const int stop = above ? __ffsll((long long)above) - 1 : 64 - __clzll((long long)act) + 0;  // first head after me, or one past the last active lane
            int base = 0;
            if (head) {
                base = atomicAdd(&cell_count[k], stop - start);
                if (in && base + stop - start > cmax) cmax = base + stop - start;  // the last add of a cell sees its full count
            }
            base = __shfl(base, start, 64);
            slot[i] = base + (lane - start);
        }
        if (in) {
            mn[0] = bx < mn[0] ? bx : mn[0]; mx[0] = bx > mx[0] ? bx : mx[0];
            mn[1] = by < mn[1] ? by : mn[1]; mx[1] = by > mx[1] ? by : mx[1];
            mn[2] = bz < mn[2] ? bz : mn[2]; mx[2] = bz > mx[2] ? bz : mx[2];
        } else if (!dead) {
            nout++;
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int lo = mn[a], hi = mx[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            int l2 = __shfl_down(lo, o, 64), h2 = __shfl_down(hi, o, 64);
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        if (lane == 0) { sm[w][a] = lo; sm[w][3 + a] = hi; }
    }
    nout = wave_sum(nout);
    cmax = wave_max(cmax);
    if (lane == 0) { sm[w][6] = nout; sm[w][7] = cmax; }
    __syncthreads();
    if (threadIdx.x < 8) {
        const int a = threadIdx.x;
        int v = sm[0][a];
        for (int k = 1; k < 4; ++k) {
            int t = sm[k][a];
            v = a < 3 ? (t < v ? t : v) : (a == 6 ? v + t : (t > v ? t : v));
        }
        part[blockIdx.x * 8 + a] = v;
    }
}

__global__ __launch_bounds__(256) void k_bin_bbox(const int* __restrict__ part, int nb, StepState* ss)
{
    __shared__ int sm[4][8];
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {-1, -1, -1}, nout = 0, cmax = 0;
    for (int b = threadIdx.x; b < nb; b += 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int l = part[b * 8 + a], h = part[b * 8 + 3 + a];
            mn[a] = l < mn[a] ? l : mn[a];
            mx[a] = h > mx[a] ? h : mx[a];
        }
        nout += part[b * 8 + 6];
        cmax = part[b * 8 + 7] > cmax ? part[b * 8 + 7] : cmax;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int lo = mn[a], hi = mx[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            int l2 = __shfl_down(lo, o, 64), h2 = __shfl_down(hi, o, 64);
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        if (lane == 0) { sm[w][a] = lo; sm[w][3 + a] = hi; }
    }
    nout = wave_sum(nout);
    cmax = wave_max(cmax);
    if (lane == 0) { sm[w][6] = nout; sm[w][7] = cmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int a = 0; a < 3; ++a) {
            int lo = sm[0][a], hi = sm[0][3 + a];
            for (int k = 1; k < 4; ++k) {
                lo = sm[k][a] < lo ? sm[k][a] : lo;
                hi = sm[k][3 + a] > hi ? sm[k][3 + a] : hi;
            }
            ss->bbox_min[a] = lo;
            ss->bbox_max[a] = hi;
        }
        ss->n_out = sm[0][6] + sm[1][6] + sm[2][6] + sm[3][6];
        int cm = sm[0][7];
        for (int k = 1; k < 4; ++k) cm = sm[k][7] > cm ? sm[k][7] : cm;
        ss->max_cell = cm;
    }
}

__global__ __launch_bounds__(256) void k_bin_scatter(long n, const int* __restrict__ key, const int* __restrict__ slot,
                                                     const int* __restrict__ cell_start, const uint32_t* __restrict__ pid,
                                                     int* __restrict__ order, uint32_t* __restrict__ spid)
{
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int d = cell_start[key[i]] + slot[i];
    order[d] = (int)i;
    spid[d] = pid[i];  // ids in (unordered) cell order, contiguous per cell: the rank pass below streams them
}

// Slots were handed out by atomics in arrival order; put every cell's list into ascending original-id order so
// that all later sums have a fixed order.  One thread per particle counts the smaller ids of its cell (ids are
// unique): O(k) per thread over a contiguous range, parallel over the k particles of the cell.  (A serial
// insertion sort per cell is O(k^2) on ONE thread: 47 ms per step once settled water piles 900 particles into a cell.)
constexpr int RANK_CH = 2048;
__global__ __launch_bounds__(256) void k_bin_rank(long n_pos, long pos0, const int* __restrict__ key, const int* __restrict__ cell_start,
                                                  const int* __restrict__ order, const uint32_t* __restrict__ spid, int* __restrict__ order2)
{
    // rank of my id among the ids of my cell.  The block's 256 sorted positions belong to consecutive cells, so the ids
    // they must look at are ONE contiguous range [A, B): staged through LDS in chunks (a cell of 10^4 particles read
    // its ids 10^4 times from L2 otherwise: 2 GB of L2 traffic, 0.3 ms, in the settled pool)
    __shared__ uint32_t sid[RANK_CH];
    __shared__ int sAB[2];
    const long j = pos0 + (long)blockIdx.x * 256 + threadIdx.x;
    const long end = pos0 + n_pos;
    const bool act = j < end;
    int src = 0, a = 0, b = 0;
    uint32_t mine = 0;
    if (act) {
        src = order[j];
        const int k = key[src];
        a = cell_start[k];
        b = cell_start[k + 1];
        mine = spid[j];
    }
    if (threadIdx.x == 0) sAB[0] = a;
    if (act && (j + 1 == end || threadIdx.x == 255)) sAB[1] = b;
    __syncthreads();
    const int A = sAB[0], B = sAB[1];
    int rank = 0;
    for (int c0 = A; c0 < B; c0 += RANK_CH) {
        const int c1 = c0 + RANK_CH < B ? c0 + RANK_CH : B;
        if (c0 != A) __syncthreads();  // the previous chunk has been consumed
        for (int t = c0 + threadIdx.x; t < c1; t += 256) sid[t - c0] = spid[t];
        __syncthreads();
        const int lo = a > c0 ? a : c0, hi = b < c1 ? b : c1;
        for (int t = lo; t < hi; ++t) rank += sid[t - c0] < mine;
    }
    if (act) order2[a + rank] = src;
}

// fluid.cc:22-37 at the three cells base-1, base, base+1 of each axis (see k_p2g_rows): w[(axis * 3 + d) * stride + j]
__device__ __forceinline__ void axis_weights(double px, double py, double pz, double* __restrict__ w, long stride, long j)
{
    const double q[3] = {px, py, pz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int b = (int)round(q[a]);
#pragma unroll
        for (int d = 0; d < 3; ++d) w[(a * 3 + d) * stride + j] = spline_at(q[a], b - 1 + d, d);
    }
}

// sorted order <- unsorted; w != nullptr: the 9 P2G axis weights of every particle are written on the way (the positions
// are in registers here anyway: saves k_weights' pass over them)
__global__ __launch_bounds__(256) void k_reorder(long n, const int* __restrict__ order, Particles s, Particles d, double* __restrict__ w,
                                                 long wstride)
{
    long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    int i = order[j];
    const double px = s.px[i], py = s.py[i], pz = s.pz[i];
    d.px[j] = px; d.py[j] = py; d.pz[j] = pz;
    d.vx[j] = s.vx[i]; d.vy[j] = s.vy[i]; d.vz[j] = s.vz[i];
    d.pid[j] = s.pid[i];
    if (w) axis_weights(px, py, pz, w, wstride, j);
}

// the 9 axis weights of every (sorted) particle (multi-GPU path: the ghost particles arrive after the sort): w[a*3+d][j] = spline(pos_a - (base_a - 1 + d))
__global__ __launch_bounds__(256) void k_weights(long n, Particles p, double* __restrict__ w, long stride)
{
    long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    axis_weights(p.px[j], p.py[j], p.pz[j], w, stride, j);
}

// ---- particle -> grid (gather form, marching over source rows) -------------------------------
// fluid.cc:1106-1148 (P2Gtransfer) + 265-299 (p2gCatmullRom) + 843-882 (interpolate).
// Cell c receives w*v from every particle whose base cell b satisfies |b-c|<=1 per axis
// (support base-1..base+1 clamped to the grid, fluid.cc:271-276), unless c is solid
// (:288; "within W" is implied by non-solid, see fluid_set_solid).  weights is a float32
// accumulator updated as float(weights + w) (:292); the velocity sum is fp64 (:293) and is
// divided by double(weights) where weights>0 (:1138-1142).  container (:873) accumulates the
// same w under "w>0", i.e. the same float sequence, so one array serves both.
// The post-P2G velocity is also stored as velBeforeUpdate (fluid.cc:1455).
//
// Separable weights: w(p,c) = sx*sy*sz with s_a = spline(pos_a - c_a) (fluid.cc:291), and a particle only ever
// meets the cells base-1, base, base+1 per axis, so 9 spline values per particle (axis_weights, bit-identical to
// evaluating spline at (pos - cell)) replace 81 evaluations; the product keeps the reference's association
// (sx*sy)*sz.
//
// Work decomposition.  The particles of a grid row (fixed x,y) are CONTIGUOUS in the sorted arrays.  A block owns the
// source x-plane rx, a segment [Y0, Y1] of target columns and a z piece of <= 62 cells; it walks the source rows
// ry = Y0-1 .. Y1+1 and stages each row's weights and velocities into LDS once with coalesced loads.  Wave e (0..2) stands for the
// target columns X = rx - 1 + e; lane L for the z cell zc = tz0 - 1 + L twice over: as a SOURCE cell it walks its own
// particles of the staged row once and forms all 3 (target y) x 3 (target z) products (10 LDS reads for 9
// particle-cell pairs); as a TARGET cell (lanes 1..zt) it then collects the sums of the sources zc-1, zc, zc+1 with
// lane shifts.  The wave keeps running sums for the target columns ry-1, ry, ry+1: after row ry the column ry-1 has met
// its three source rows of this x-plane and its partial is written.  Each target cell so receives three partials
// (from rx = X-1, X, X+1), added in that order by k_p2g_combine.  Sum order per target: rows ascending (x, then y),
// sources ascending z, particles ascending (sorted order, ties by id) — a pure function of the input.  Every block
// writes all its partials (zeros for empty rows), so nothing is cleared.
// History (256^3 bench scene, 5.3 M particles): a lane-per-cell loop straight from global memory 7.4 ms; 2 x 2-column
// tiles staging their 4 x 4 window of rows (every row staged by 4 tiles, 2.2 GB fetched) 0.38 ms with, per tile row,
// at most 2.25 of 4 waves busy; 4 x 4 tiles 1.29 ms (9 of 16 waves busy); this form stages (ys + 2) / ys x for segments of ys
// columns (1.1-1.2 x on the bench scene): 0.21 ms + 0.02 for k_p2g_combine.
// Tried without gain: computing the weights while staging (positions instead of 9 weights: +0.15 ms of fp64 work),
// reading the next row's cell ranges a row ahead, a register-prefetch pipeline over the chunks, one 96 B record per particle
// (array of structures: one stream per block instead of 12) for the staged data and for the partials (0.32 ms against 0.26, and
// the reorder pass that writes them +60 us), a kernel of its own pre-reducing the cells above 256 particles (no change where
// particles pile up: there the per-row latency of whole wall planes is the critical path, see k_p2g_tiles).
constexpr int P2G_SLOTS = 256 * 4;  // blocks resident at once: 256 CUs x 4 (37.6 KB of LDS, 168 VGPRs)
constexpr int P2G_THREADS = 192;
constexpr int P2G_CH = 384;       // particles staged per chunk: 12 x 8 B x 392 = 37.6 KB of LDS, 4 blocks per CU (168 VGPRs: 3 waves per
                                  // SIMD).  A row piece of this scene holds ~360 particles; 256 and 320 (two chunks per row) measured 2.3 x slower
// LDS slot of staged particle k: lane z reads particle a_z + t with a_z growing by ~8 (particles per cell) from
// lane to lane; splitting by k mod 8 keeps neighbouring lanes on neighbouring slots.
constexpr int P2G_SEG = P2G_CH / 8 + 1;
__device__ __forceinline__ int p2g_slot(int k) { return (k & 7) * P2G_SEG + (k >> 3); }
constexpr int P2G_LDS = 8 * P2G_SEG;
constexpr int P2G_HEAVY = 48;  // a longer per-lane window is swept by the whole wave
constexpr int P2G_CROWD = 18;  // CROWD form: cells of this many particles or more are summed by k_p2g_crowd_sum (18 slots per piece x 6 arrays = 108 sums)
constexpr int P2G_PIECE = 512; // ... in pieces of this many particles, one wave each
constexpr int P2G_BUDGET = 8192;  // particles per work item and z piece before a y segment is cut further (a regular 256^3 segment: ~6500)
constexpr int P2G_ZT = 62;     // most target cells a wave takes (lanes 0 and 63 are sources only); the launcher splits nz evenly

// part: [3 source x-planes][4: weight, u, v, w][cells of box].  VEC: the weight and velocity arrays are 16 B aligned with an
// even stride (the launcher checks), so two particles are staged per lane and load
// CROWD (piled particles, mostly-air boxes): the sums of every cell of P2G_CROWD or more particles have been formed by k_p2g_crowd_sum and
// are parked in `crowd` (six arrays, see there); such a cell is not walked here — its lane adds the 36 parked values of its target plane —
// and a chunk that holds particles of such cells only is not staged.
template <bool VEC, bool CROWD>
__global__ __launch_bounds__(P2G_THREADS, 3) void k_p2g_rows(Grid g, Box box, Particles p, const double* __restrict__ pw, long wstride,
                                                          const int* __restrict__ cell_start, double* __restrict__ part, long cells, int zt, const int* __restrict__ items,
                                                          Particles crowd, uint8_t* __restrict__ colflag, int ntz)
{
    __shared__ double sr[12][P2G_LDS];   // wx0..2, wy0..2, wz0..2, vx, vy, vz
    const int tid = threadIdx.x, e = tid >> 6, lane = tid & 63;
    for (int item = blockIdx.x; item < items[0]; item += gridDim.x) {   // items[0]: block-uniform count
    const int4 wi = reinterpret_cast<const int4*>(items)[1 + item];      // x-plane, first and last target column, z piece
    const int rx = box.x0 - 1 + wi.x, Y0 = wi.y, Y1 = wi.z, tz0 = box.z0 + wi.w * zt;
    const int X = rx - 1 + e, zc = tz0 - 1 + lane;
    const bool colx = X >= box.x0 && X <= box.x1;                  // my target plane is in the box
    const bool src = colx && zc >= 0 && zc < g.nz;                 // my cell exists: it may hold particles
    const bool tgt = colx && lane >= 1 && lane <= zt && zc <= box.z1;
    const bool rowx = rx >= 0 && rx < g.nx;
    const int zlo = tz0 > 0 ? tz0 - 1 : 0, zhi = tz0 + zt < g.nz - 1 ? tz0 + zt : g.nz - 1;
    const double* swx = sr[e];   // a particle of plane rx meets plane X = rx - 1 + e with its x weight e
    // partial slot = rx - X + 1 = 2 - e
    double* const out = part + (long)(2 - e) * 4 * cells + (long)(X - box.x0) * box.ny() * box.nz() + (zc - box.z0);
    double C[3][4];  // running sums of the target columns ry-1, ry, ry+1: weight, u, v, w
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) C[a][q] = 0;
    // the particle ranges of the item's rows, 64 rows at a time (lane = row): after the splash more than half of the row pieces are
    // empty, and an empty row must not cost a trip to memory
    int pjb = 0, pje = 0, nca = 0, ncz = 0;
    bool have = false;   // nca / ncz hold the row's cell range already
    uint8_t* const cflag = colflag + ((long)(2 - e) * box.nx() + (X - box.x0)) * box.ny() * ntz + wi.w;
    for (int ry = Y0 - 1; ry <= Y1 + 1; ++ry) {
        const int ri = ry - (Y0 - 1);
        if ((ri & 63) == 0) {
            const int y = ry + lane;
            pjb = pje = 0;
            if (rowx && y >= 0 && y < g.ny && y <= Y1 + 1) {
                pjb = cell_start[g.idx(rx, y, zlo)];
                pje = cell_start[g.idx(rx, y, zhi) + 1];
            }
        }
        double T[3][3][4];  // this row's sums by target y (ry-1..ry+1), target z (zc-1..zc+1), value
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int q = 0; q < 4; ++q) T[a][d][q] = 0;
        const int jb = __shfl(pjb, ri & 63, 64), je = __shfl(pje, ri & 63, 64);
        // my source cell's particles in this row — fetched while the row before was worked on (a row piece after the splash holds a handful of
        // particles: its visit is trips to memory, one fewer this way)
        int ca = nca, cz = ncz;
        if (je > jb && !have && src) {
            const size_t r0 = g.idx(rx, ry, zc);
            ca = cell_start[r0];
            cz = cell_start[r0 + 1];
        }
        have = false;
        if (((ri + 1) & 63) != 0 && ry < Y1 + 1 && __shfl(pje, (ri + 1) & 63, 64) > __shfl(pjb, (ri + 1) & 63, 64)) {
            have = true;
            nca = ncz = 0;
            if (src) {
                const size_t r1 = g.idx(rx, ry + 1, zc);
                nca = cell_start[r1];
                ncz = cell_start[r1 + 1];
            }
        }
        if (je > jb) {  // block-uniform
            const int ca_all = ca, n_all = cz - ca;
            if (CROWD && n_all >= P2G_CROWD) cz = ca;   // not walked: its sums are parked
            for (int cb = VEC ? (jb & ~1) : jb; cb < je; cb += P2G_CH) {
                const int ce = cb + P2G_CH < je ? cb + P2G_CH : je;
                if (CROWD) {
                    // (also the barrier that frees the staged chunk before) a chunk without a particle that anyone walks is not staged
                    if (!__syncthreads_or((ca > cb ? ca : cb) < (cz < ce ? cz : ce))) continue;
                } else
                    __syncthreads();  // the previous chunk has been consumed
                if (VEC) {
                    // 16 B accesses from an even index; the odd particle before jb / after je - 1 is staged but never read
                    for (int j = cb + 2 * tid; j < ce; j += 2 * P2G_THREADS) {
                        const int k0 = p2g_slot(j - cb), k1 = p2g_slot(j + 1 - cb);
                        double2 t[6];   // (two batches of six arrays: the twelve at once are 48 registers the walk's 36 sums need)
#pragma unroll
                        for (int q = 0; q < 6; ++q) t[q] = *reinterpret_cast<const double2*>(pw + q * wstride + j);
#pragma unroll
                        for (int q = 0; q < 6; ++q) { sr[q][k0] = t[q].x; sr[q][k1] = t[q].y; }
#pragma unroll
                        for (int q = 0; q < 3; ++q) t[q] = *reinterpret_cast<const double2*>(pw + (6 + q) * wstride + j);
                        t[3] = *reinterpret_cast<const double2*>(p.vx + j);
                        t[4] = *reinterpret_cast<const double2*>(p.vy + j);
                        t[5] = *reinterpret_cast<const double2*>(p.vz + j);
#pragma unroll
                        for (int q = 0; q < 6; ++q) { sr[6 + q][k0] = t[q].x; sr[6 + q][k1] = t[q].y; }
                    }
                } else {
                    for (int j = cb + tid; j < ce; j += P2G_THREADS) {
                        const int k = p2g_slot(j - cb);
#pragma unroll
                        for (int q = 0; q < 9; ++q) sr[q][k] = pw[q * wstride + j];
                        sr[9][k] = p.vx[j]; sr[10][k] = p.vy[j]; sr[11][k] = p.vz[j];
                    }
                }
                __syncthreads();
                // Every lane of the wave sits on the same target plane, so the x weight table is wave-uniform.  A lane
                // whose cell is crowded (settled water piles up to ~10^4 particles into one cell) would serialise the
                // whole wave: such cells are swept by all 64 lanes together and wave-reduced (fixed order).
                int lo = ca > cb ? ca : cb, hi = cz < ce ? cz : ce;
                if (hi < lo) hi = lo;
                const bool heavy = !CROWD && hi - lo > P2G_HEAVY;   // (CROWD: a walked cell has at most 17 particles)
                if (!heavy) {
                    for (int j = lo; j < hi; ++j) {
                        const int k = p2g_slot(j - cb);
                        const double wx = swx[k];
                        const double vx = sr[9][k], vy = sr[10][k], vz = sr[11][k];
                        const double z0 = sr[6][k], z1 = sr[7][k], z2 = sr[8][k];
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            const double xy = wx * sr[3 + a][k];   // (sx*sy)*sz: the reference's association (fluid.cc:291)
                            const double c0 = xy * z0, c1 = xy * z1, c2 = xy * z2;
                            T[a][0][0] += c0; T[a][0][1] = T[a][0][1] + c0 * vx; T[a][0][2] = T[a][0][2] + c0 * vy; T[a][0][3] = T[a][0][3] + c0 * vz;
                            T[a][1][0] += c1; T[a][1][1] = T[a][1][1] + c1 * vx; T[a][1][2] = T[a][1][2] + c1 * vy; T[a][1][3] = T[a][1][3] + c1 * vz;
                            T[a][2][0] += c2; T[a][2][1] = T[a][2][1] + c2 * vx; T[a][2][2] = T[a][2][2] + c2 * vy; T[a][2][3] = T[a][2][3] + c2 * vz;
                        }
                    }
                }
                unsigned long long hm = CROWD ? 0ull : __ballot(heavy);
                while (hm) {
                    const int L = __ffsll((long long)hm) - 1;
                    hm &= hm - 1;
                    const int lo_L = __shfl(lo, L, 64), hi_L = __shfl(hi, L, 64);
#pragma unroll
                    for (int a = 0; a < 3; ++a) {  // one target y per sweep: 12 live sums instead of 36 (registers)
                        double H[3][4];
#pragma unroll
                        for (int d = 0; d < 3; ++d)
#pragma unroll
                            for (int q = 0; q < 4; ++q) H[d][q] = 0;
                        for (int j = lo_L + lane; j < hi_L; j += 64) {
                            const int k = p2g_slot(j - cb);
                            const double xy = swx[k] * sr[3 + a][k];
                            const double vx = sr[9][k], vy = sr[10][k], vz = sr[11][k];
                            const double zw[3] = {sr[6][k], sr[7][k], sr[8][k]};
#pragma unroll
                            for (int d = 0; d < 3; ++d) {
                                const double cw = xy * zw[d];
                                H[d][0] += cw; H[d][1] += cw * vx; H[d][2] += cw * vy; H[d][3] += cw * vz;
                            }
                        }
#pragma unroll
                        for (int d = 0; d < 3; ++d)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const double t = __shfl(wave_sum(H[d][q]), 0, 64);
                                if (lane == L) T[a][d][q] += t;
                            }
                    }
                }
            }
            if (CROWD) {
                // parked value (x offset e, target y a, target z d, value q) of a cell whose first particle is ca: linear index
                // ((3 e + a) * 12 + 4 d + q) over six arrays of 18 values per 512-particle piece, pieces added in order
                const double* c0 = e == 0 ? crowd.px : (e == 1 ? crowd.pz : crowd.vy);
                const double* c1 = e == 0 ? crowd.py : (e == 1 ? crowd.vx : crowd.vz);
                if (n_all >= P2G_CROWD) {
                    for (int at = ca_all; at < ca_all + n_all; at += P2G_PIECE) {
                        const long o18 = ca_all + 18L * ((at - ca_all) / P2G_PIECE);
#pragma unroll
                        for (int a = 0; a < 3; ++a)
#pragma unroll
                            for (int d = 0; d < 3; ++d)
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const int Lr = a * 12 + d * 4 + q;
                                    T[a][d][q] += (Lr < 18 ? c0 : c1)[o18 + Lr % 18];
                                }
                    }
                }
            }
        }
        // target zc collects: source zc-1 reaches it with its z index 2, zc with 1, zc+1 with 0 (d = target - source + 1)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double below = __shfl_up(T[a][2][q], 1, 64), above = __shfl_down(T[a][0][q], 1, 64);
                C[a][q] = ((C[a][q] + below) + T[a][1][q]) + above;
            }
        // column ry-1 has met rows ry-2, ry-1, ry of this x-plane.  A column piece that received nothing (weight sums all zero) is only
        // flagged: neither written here nor read by k_p2g_combine
        if (ry - 1 >= Y0 && colx) {
            const bool any = __any(tgt && C[0][0] != 0);
            if (lane == 0) cflag[(long)(ry - 1 - box.y0) * ntz] = any;
            if (any && tgt) {
                double* o = out + (long)(ry - 1 - box.y0) * box.nz();
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q * cells] = C[0][q];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { C[0][q] = C[1][q]; C[1][q] = C[2][q]; C[2][q] = 0; }
    }
    }  // items
}

// ---- crowded cells on the matrix cores (piled particles, mostly-air boxes) --------------------------------------------------
// After the splash a third of the particles sit in cells of 18 .. 10^4 (step 445 of the 256^3 drop: 46 k cells above 17 hold 41 % of
// the particles); walking them lane by lane, or wave by wave, is what made P2G 1.1 - 1.5 ms there.  What any target can want from a
// source cell is 27 x 4 sums over its particles — a contraction with the particles on k: an fp64 MFMA (v_mfma_f64_16x16x4_f64) with
// A = 9 rows (x offset, y offset: sx * sy) x 4 particles, B = 4 particles x 12 columns (z offset, value: sz, sz * v) gives all 108 in
// its 16 x 16 result; a wave takes a piece of <= 512 particles of one cell (its particles staged 64 at a time, lane = particle, then
// 16 steps of four).  The 108 sums of piece j of a cell are parked in the slots ca + 18 j .. + 17 of the six arrays of the OTHER
// particle buffer (free between the reorder pass and the next sort; a cell of >= 18 particles owns at least that many slots): linear
// index (3 dx + dy) * 12 + 4 dz + value, 18 to an array, so the row kernel's wave for x offset e finds its 36 values in arrays 2e, 2e + 1.
// Every sum is a fixed function of the cell's sorted particle list (steps of four from the cell's first particle, pieces in order).
// (The weight keeps the reference's association (sx*sy)*sz; a velocity sum is (sx*sy)*(sz*v) where the walk forms ((sx*sy)*sz)*v.)
constexpr int PC_WAVES = 4;
typedef double pc_d4 __attribute__((ext_vector_type(4)));
// list: one entry per piece of every cell of >= P2G_CROWD particles in box — (first particle of the piece, piece << 9 | particles in it - 1);
// count[0] = entries (zero on entry).  A thread looks at four cells along z; one returning atomic per block of 1024 cells.
__global__ __launch_bounds__(256) void k_p2g_crowd_list(Grid g, Box box, const int* __restrict__ cell_start, int2* __restrict__ list, int* __restrict__ count)
{
    __shared__ int s_w[4], s_base;
    const int nz = box.nz(), ny = box.ny(), nz4 = (nz + 3) / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    int first[5] = {0, 0, 0, 0, 0};
    int pieces = 0;
    if (i < (long)box.nx() * ny * nz4) {
        const int z = 4 * (int)(i % nz4), y = (int)((i / nz4) % ny), x = (int)(i / ((long)nz4 * ny));
        const int* cs = cell_start + g.idx(box.x0 + x, box.y0 + y, box.z0 + z);
        const int m = nz - z < 4 ? nz - z : 4;
#pragma unroll
        for (int k = 0; k < 5; ++k) first[k] = cs[k <= m ? k : m];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int n = first[k + 1] - first[k];
            pieces += n >= P2G_CROWD ? (n + P2G_PIECE - 1) / P2G_PIECE : 0;
        }
    }
    // exclusive prefix over the block (the order of the list does not reach the sums)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = pieces;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
    }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        s_base = tot ? atomicAdd(count, tot) : 0;
    }
    __syncthreads();
    if (!pieces) return;
    int at = s_base + inc - pieces;
    for (int w = 0; w < wv; ++w) at += s_w[w];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int n = first[k + 1] - first[k];
        if (n < P2G_CROWD) continue;
        for (int j = 0; j * P2G_PIECE < n; ++j) {
            const int len = n - j * P2G_PIECE < P2G_PIECE ? n - j * P2G_PIECE : P2G_PIECE;
            list[at++] = make_int2(first[k] + j * P2G_PIECE, (j << 9) | (len - 1));
        }
    }
}
__global__ __launch_bounds__(64 * PC_WAVES) void k_p2g_crowd_sum(Particles p, const double* __restrict__ pw, long wstride,
                                                                const int2* __restrict__ list, int* __restrict__ count, Particles park)
{
    __shared__ double sw[PC_WAVES][12][64];   // wx0..2, wy0..2, wz0..2, vx, vy, vz of the 64 staged particles of each wave
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 15, gq = lane >> 4;
    const int total = count[0];
    double(*my)[64] = sw[wv];
    // as A: row n = (x offset, y offset); as B: column n = (z offset, value)
    const double* ax = my[n < 9 ? n / 3 : 0];
    const double* ay = my[3 + (n < 9 ? n % 3 : 0)];
    const double* bz = my[6 + (n < 12 ? n >> 2 : 0)];
    const double* bv = my[8 + ((n & 3) ? (n & 3) : 1)];
    for (int ent = blockIdx.x * PC_WAVES + wv; ent < total; ent += gridDim.x * PC_WAVES) {
        const int2 en = list[ent];
        const int piece = en.y >> 9, lo = en.x, hi = lo + (en.y & 511) + 1, ca = lo - P2G_PIECE * piece;
        pc_d4 D = {0, 0, 0, 0};
        for (int b = lo; b < hi; b += 64) {
            const int j = b + lane;
            const bool ok = j < hi;
            double v[12];
#pragma unroll
            for (int q = 0; q < 9; ++q) v[q] = ok ? pw[q * wstride + j] : 0.0;
            v[9] = ok ? p.vx[j] : 0.0, v[10] = ok ? p.vy[j] : 0.0, v[11] = ok ? p.vz[j] : 0.0;
            __builtin_amdgcn_wave_barrier();   // (the wave's own LDS rows: every lane has read the batch before)
#pragma unroll
            for (int q = 0; q < 12; ++q) my[q][lane] = v[q];
            __builtin_amdgcn_wave_barrier();
            const int nb = hi - b < 64 ? hi - b : 64;
#pragma unroll 4
            for (int s4 = 0; s4 < 16; ++s4) {
                if (4 * s4 >= nb) break;   // wave-uniform; a slot past the piece holds zeros
                const int k = 4 * s4 + gq;
                const double A = n < 9 ? ax[k] * ay[k] : 0.0;
                const double zw = bz[k], zv = zw * bv[k];
                const double B = n < 12 ? ((n & 3) ? zv : zw) : 0.0;
                D = __builtin_amdgcn_mfma_f64_16x16x4f64(A, B, D, 0, 0, 0);
            }
        }
        // D: column n, rows gq + 4 i in register i: linear index row * 12 + n
        if (n < 12) {
            const long o18 = ca + 18L * piece;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int row = gq + 4 * i;
                if (row < 9) {
                    const int L = row * 12 + n, arr = L / 18;
                    double* dst = arr == 0 ? park.px : (arr == 1 ? park.py : (arr == 2 ? park.pz : (arr == 3 ? park.vx : (arr == 4 ? park.vy : park.vz))));
                    dst[o18 + L % 18] = D[i];
                }
            }
        }
    }
}

// The work list of k_p2g_rows.  A regular cut (x-plane) x (nseg equal y segments) x (z piece) would hand every row of a
// plane where particles have piled up (against a wall, after the splash) to the same few blocks: 15 ms instead of 0.25 at
// 256^3.  So a segment is cut further by its particle count, down to single columns: the particles of rows Y0-1..Y1+1 of a
// plane are one contiguous range of the sorted arrays (two cell_start reads), `budget` of them per item and z piece.
// items: [0] count (zero on entry; k_p2g_combine puts it back), then int4 (plane, Y0, Y1, z piece) from int 4 on.
__global__ __launch_bounds__(256) void k_p2g_items(Grid g, Box box, const int* __restrict__ cell_start, int nseg, int ntz, int budget,
                                                   int* __restrict__ items)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= (box.nx() + 2) * nseg) return;
    const int sy = t % nseg, bx = t / nseg;
    const int rx = box.x0 - 1 + bx;
    const int Y0 = box.y0 + sy * box.ny() / nseg, Y1 = box.y0 + (sy + 1) * box.ny() / nseg - 1, len = Y1 - Y0 + 1;
    long c = 0;
    if (rx >= 0 && rx < g.nx) {
        const int ya = Y0 > 0 ? Y0 - 1 : 0, yb = Y1 < g.ny - 1 ? Y1 + 1 : g.ny - 1;
        c = (long)cell_start[g.idx(rx, yb, g.nz - 1) + 1] - cell_start[g.idx(rx, ya, 0)];
    }
    long nsub = (c + (long)budget * ntz - 1) / ((long)budget * ntz);
    nsub = nsub < 1 ? 1 : (nsub > len ? len : nsub);
    const int base = atomicAdd(items, (int)nsub * ntz);  // the order of the items does not reach the sums
    int4* o = reinterpret_cast<int4*>(items) + 1 + base;
    for (int i = 0; i < (int)nsub; ++i)
        for (int tz = 0; tz < ntz; ++tz) o[i * ntz + tz] = make_int4(bx, Y0 + (int)(i * len / nsub), Y0 + (int)((i + 1) * len / nsub) - 1, tz);
}

// the three x-plane partials of a target cell, ascending source x; weights is the reference's float32 accumulator (rounded
// once per partial here: 1e-7 from the reference's per-particle rounding, inside the reference's own TBB-order noise), the
// velocity sums are fp64 and divided by double(weights) where weights > 0 (fluid.cc:1138-1142)
__global__ __launch_bounds__(256) void k_p2g_combine(Grid g, Box box, const double* __restrict__ part, long cells,
                                                     const uint8_t* __restrict__ flags, float* __restrict__ container,
                                                     double* __restrict__ u, double* __restrict__ v, double* __restrict__ w,
                                                     double* __restrict__ ub, double* __restrict__ vb, double* __restrict__ wb,
                                                     int* __restrict__ items, const uint8_t* __restrict__ colflag, int zt, int ntz)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) items[0] = items[1] = 0;  // k_p2g_rows is done with the work list (and with the crowded cells' list, counted in items[1]): empty for the next launch
    if (i >= cells) return;
    const int nz = box.nz(), ny = box.ny();
    const int z = (int)(i % nz), y = (int)((i / nz) % ny), x = (int)(i / ((long)nz * ny));
    const size_t c = g.idx(box.x0 + x, box.y0 + y, box.z0 + z);
    if (flags[c] & F_SOLID) return;  // solid cells receive nothing (:288,870): fields stay 0
    float wf = 0.0f;
    double s[3] = {0, 0, 0};
    const uint8_t* cf = colflag + ((long)x * ny + y) * ntz + z / zt;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (!cf[(long)k * box.nx() * ny * ntz]) continue;   // nothing reached this column piece from that x-plane: its partial was not written
        const double* q = part + (long)k * 4 * cells + i;
        wf = (float)((double)wf + q[0]);
#pragma unroll
        for (int a = 0; a < 3; ++a) s[a] += q[(a + 1) * cells];
    }
    if (wf > 0) {
        const double wd = (double)wf;
        s[0] /= wd; s[1] /= wd; s[2] /= wd;
    }
    container[c] = wf;
    u[c] = s[0]; v[c] = s[1]; w[c] = s[2];
    ub[c] = s[0]; vb[c] = s[1]; wb[c] = s[2];
}

// ---- particle -> grid, tile form (piled particles) -------------------------------------------------
// The same sums per 2 x 2 target columns: a block stages the 4 x 4 window of source rows around its columns (every row is
// staged by 4 tiles) and wave = column.  Slower than k_p2g_rows on evenly filled water (0.38 ms against 0.25 at 256^3) but
// its 16 rows come from four x-planes, so particles piled against an x wall (600 k in one plane, 10^5 in one row after the
// 256^3 splash has settled) are spread over many more blocks: 1.8 ms where the row-marching form needs 4-15 ms.  The host
// takes this one once a cell holds more than P2G_PILED particles.
constexpr int P2GT_T = 2;
constexpr int P2GT_THREADS = P2GT_T * P2GT_T * 64;
constexpr int P2GT_CH = 512;  // particles staged per chunk (two per thread): 12 arrays x 8 B x 520 = 50 KB of LDS, 3 blocks per CU.
                              // Measured at 256^3: 192: 0.91 ms, 256: 0.82, 320: 0.49, 384: 0.48, 448: 0.55, 512: 0.41, 576: 0.51
constexpr int P2GT_SEG = P2GT_CH / 8 + 1;
__device__ __forceinline__ int p2gt_slot(int k) { return (k & 7) * P2GT_SEG + (k >> 3); }
constexpr int P2GT_LDS = 8 * P2GT_SEG;
// Lane L of a wave stands for the z-cell zc = tz0 - 1 + L of its column twice over: as a SOURCE cell it walks its own
// particles of the staged row once and forms, for each of the three target cells zc-1, zc, zc+1 it can reach, the
// partial sums of w and w*v (12 accumulators; 8 LDS reads per particle for 3 particle-cell pairs); as a TARGET cell
// (lanes 1..62) it then collects the partials of the sources zc-1, zc, zc+1 with two lane shifts per value.  (The first
// version let every target lane walk the particles of its three source cells itself: 18 LDS reads per particle and
// three times the trips.)  Sum order per target: rows ascending (x,y), sources ascending z, particles ascending.

__global__ __launch_bounds__(P2GT_THREADS) void k_p2g_tiles(Grid g, Box box, Particles p, const double* __restrict__ pw, long wstride,
                                             const int* __restrict__ cell_start, const uint8_t* __restrict__ flags,
                                             float* __restrict__ container, double* __restrict__ u, double* __restrict__ v,
                                             double* __restrict__ w, double* __restrict__ ub, double* __restrict__ vb, double* __restrict__ wb,
                                             int zt)
{
    __shared__ double sw[9][P2GT_LDS];   // wx0..2, wy0..2, wz0..2
    __shared__ double sv[3][P2GT_LDS];   // vx, vy, vz
    __shared__ int srow[(P2GT_T + 2) * (P2GT_T + 2)][2];   // particle range of every source row of the window
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int ntz = (box.nz() + zt - 1) / zt, nty = (box.ny() + P2GT_T - 1) / P2GT_T;  // zt <= P2G_ZT target cells per wave
    const int tile = blockIdx.x;
    // faces first: piles form against the walls, and a block that starts late with a crowded wall plane is the kernel's tail.
    // x tiles are taken alternately from both ends (the slowest index, so the first blocks dispatched are the two x faces)
    const int tz = tile % ntz, ty = (tile / ntz) % nty, txs = tile / (ntz * nty);
    const int ntx = (box.nx() + P2GT_T - 1) / P2GT_T;
    const int tx = (txs & 1) ? ntx - 1 - (txs >> 1) : (txs >> 1);
    const int tx0 = box.x0 + tx * P2GT_T, ty0 = box.y0 + ty * P2GT_T, tz0 = box.z0 + tz * zt;
    const int ix = tx0 + wv / P2GT_T, iy = ty0 + wv % P2GT_T, zc = tz0 - 1 + lane;
    const bool col = ix <= box.x1 && iy <= box.y1;                 // my column is in the box
    const bool src = col && zc >= 0 && zc < g.nz;                  // my cell exists: it may hold particles
    const bool tgt = col && lane >= 1 && lane <= zt && zc <= box.z1;
    const size_t c = tgt ? g.idx(ix, iy, zc) : 0;
    const bool live = tgt && !(flags[c] & F_SOLID);  // solid cells receive nothing (:288,870)
    const int zlo = tz0 > 0 ? tz0 - 1 : 0, zhi = tz0 + zt < g.nz - 1 ? tz0 + zt : g.nz - 1;
    float wf = 0.0f;
    double su = 0, sv_ = 0, sw_ = 0;
    // The ranges of all 16 source rows at once (one thread per row): in a mostly empty box (the splash: 80 % of the tiles
    // hold no particle) a block otherwise spends its life in 16 dependent pairs of loads that each find nothing.
    if (tid < (P2GT_T + 2) * (P2GT_T + 2)) {
        const int rx = tx0 - 1 + tid / (P2GT_T + 2), ry = ty0 - 1 + tid % (P2GT_T + 2);
        int jb = 0, je = 0;
        if (rx >= 0 && rx < g.nx && ry >= 0 && ry < g.ny) {
            jb = cell_start[g.idx(rx, ry, zlo)];
            je = cell_start[g.idx(rx, ry, zhi) + 1];
        }
        srow[tid][0] = jb;
        srow[tid][1] = je;
    }
    __syncthreads();
    for (int rx = tx0 - 1; rx <= tx0 + P2GT_T; ++rx) {
        if (rx < 0 || rx >= g.nx) continue;
        for (int ry = ty0 - 1; ry <= ty0 + P2GT_T; ++ry) {
            if (ry < 0 || ry >= g.ny) continue;
            const int ri = (rx - tx0 + 1) * (P2GT_T + 2) + (ry - ty0 + 1);
            const int jb = srow[ri][0], je = srow[ri][1];
            if (je == jb) continue;  // block-uniform
            const int dxi = ix - rx + 1, dyi = iy - ry + 1;  // which axis weight of a particle of this row meets my column
            const bool mine = src && dxi >= 0 && dxi <= 2 && dyi >= 0 && dyi <= 2;
            int ca = 0, cz = 0;  // my source cell's particles in this row
            if (mine) {
                const size_t r0 = g.idx(rx, ry, zc);
                ca = cell_start[r0];
                cz = cell_start[r0 + 1];
            }
            const double* swx = sw[dxi < 0 ? 0 : (dxi > 2 ? 2 : dxi)];
            const double* swy = sw[3 + (dyi < 0 ? 0 : (dyi > 2 ? 2 : dyi))];
            double S[3] = {0, 0, 0}, U[3] = {0, 0, 0}, V[3] = {0, 0, 0}, W[3] = {0, 0, 0};  // by target: zc-1, zc, zc+1
            for (int cb = jb; cb < je; cb += P2GT_CH) {
                const int ce = cb + P2GT_CH < je ? cb + P2GT_CH : je;
                __syncthreads();  // the previous chunk has been consumed
                for (int j = cb + tid; j < ce; j += P2GT_THREADS) {
                    const int k = p2gt_slot(j - cb);
#pragma unroll
                    for (int q = 0; q < 9; ++q) sw[q][k] = pw[q * wstride + j];
                    sv[0][k] = p.vx[j]; sv[1][k] = p.vy[j]; sv[2][k] = p.vz[j];
                }
                __syncthreads();
                // Every lane of the wave sits on the same (x,y) column, so the x/y weight tables are wave-uniform.  A lane
                // whose cell is crowded (settled water piles up to ~10^4 particles into one cell) would serialise the
                // whole wave: such cells are swept by all 64 lanes together and wave-reduced (fixed order).
                int lo = ca > cb ? ca : cb, hi = cz < ce ? cz : ce;
                if (hi < lo) hi = lo;
                const bool heavy = hi - lo > P2G_HEAVY;
                if (!heavy) {
                    for (int j = lo; j < hi; ++j) {
                        const int k = p2gt_slot(j - cb);
                        const double a = swx[k] * swy[k];   // (sx*sy)*sz: the reference's association (fluid.cc:291)
                        const double vx = sv[0][k], vy = sv[1][k], vz = sv[2][k];
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            const double cw = a * sw[6 + d][k];
                            S[d] += cw;
                            U[d] = U[d] + cw * vx;
                            V[d] = V[d] + cw * vy;
                            W[d] = W[d] + cw * vz;
                        }
                    }
                }
                unsigned long long hm = __ballot(heavy);
                while (hm) {
                    const int L = __ffsll((long long)hm) - 1;
                    hm &= hm - 1;
                    const int lo_L = __shfl(lo, L, 64), hi_L = __shfl(hi, L, 64);
                    double ps[3] = {0, 0, 0}, pu[3] = {0, 0, 0}, pv[3] = {0, 0, 0}, pq[3] = {0, 0, 0};
                    for (int j = lo_L + lane; j < hi_L; j += 64) {
                        const int k = p2gt_slot(j - cb);
                        const double a = swx[k] * swy[k];
                        const double vx = sv[0][k], vy = sv[1][k], vz = sv[2][k];
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            const double cw = a * sw[6 + d][k];
                            ps[d] += cw;
                            pu[d] += cw * vx;
                            pv[d] += cw * vy;
                            pq[d] += cw * vz;
                        }
                    }
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        double t0 = wave_sum(ps[d]), t1 = wave_sum(pu[d]), t2 = wave_sum(pv[d]), t3 = wave_sum(pq[d]);
                        t0 = __shfl(t0, 0, 64); t1 = __shfl(t1, 0, 64); t2 = __shfl(t2, 0, 64); t3 = __shfl(t3, 0, 64);
                        if (lane == L) { S[d] += t0; U[d] += t1; V[d] += t2; W[d] += t3; }
                    }
                }
            }
            // target zc collects: source zc-1 reaches it with its weight index 2, zc with 1, zc+1 with 0 (d = target - source + 1);
            // container keeps the reference's float accumulator, one rounding per source cell instead of one per particle
            const double s0 = __shfl_up(S[2], 1, 64), s2 = __shfl_down(S[0], 1, 64);
            const double u0 = __shfl_up(U[2], 1, 64), u2 = __shfl_down(U[0], 1, 64);
            const double v0 = __shfl_up(V[2], 1, 64), v2 = __shfl_down(V[0], 1, 64);
            const double w0 = __shfl_up(W[2], 1, 64), w2 = __shfl_down(W[0], 1, 64);
            if (live) {
                wf = (float)((double)wf + s0);
                wf = (float)((double)wf + S[1]);
                wf = (float)((double)wf + s2);
                su = ((su + u0) + U[1]) + u2;
                sv_ = ((sv_ + v0) + V[1]) + v2;
                sw_ = ((sw_ + w0) + W[1]) + w2;
            }
        }
    }
    if (!live) return;  // fields stay 0
    if (wf > 0) {
        const double wd = (double)wf;
        su /= wd; sv_ /= wd; sw_ /= wd;
    }
    container[c] = wf;
    u[c] = su; v[c] = sv_; w[c] = sw_;
    ub[c] = su; vb[c] = sv_; wb[c] = sw_;
}

// ---- grid -> particle, FLIP -----------------------------------------------------------------
// fluid.cc:978-991 + CatmullRomFLIP 210-263.  dc* holds getVelocity(new) - getVelocity(old)
// per cell (k_flip_delta), i.e. the (velc - velp) term of :252.
// blend < 1 (build extension): v' = blend (v + delta) + (1 - blend) v_pic with v_pic the same weighted gather of
// getVelocity(c, vels) (pc*), i.e. the reference's unused clampedCatmullRom (fluid.cc:125-207) without its clamp.
__global__ __launch_bounds__(256) void k_g2p(Grid g, long n, Particles p, const double* __restrict__ dcx, const double* __restrict__ dcy,
                                             const double* __restrict__ dcz, const double* __restrict__ pcx, const double* __restrict__ pcy,
                                             const double* __restrict__ pcz, double blend, StepState* ss)
{
    __shared__ double sm[4];
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    double len = 0;
    if (i < n && p.pid[i] != PID_DEAD) {   // (dead: a ghost of a decomposed run, already served by P2G)
        const double cx = p.px[i], cy = p.py[i], cz = p.pz[i];
        const int wlo = g.lo + 2, whi = g.hi - 2;
        const int fcx = (int)round(cx), fcy = (int)round(cy), fcz = (int)round(cz);
        // The 27 weights are products of 3 x 3 axis values (same values and association as spline()*spline()*spline()).
        // A cell takes part iff it is on the grid (clamp, :216-221) and within W (:237) — per axis: its coordinate lies in
        // [wlo, whi].  An axis value that fails gets weight 0 and a clamped (readable) index, so the 27-cell loop has no
        // branches: the skipped cells add exactly +0 to every sum.
        double wx[3], wy[3], wz[3];
        int ox[3], oy[3], oz[3];
        const int sxs = (int)g.sx();   // N <= 1024: N^3 fits an int
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int ax = fcx - 1 + d, ay = fcy - 1 + d, az = fcz - 1 + d;
            wx[d] = (ax >= wlo && ax <= whi) ? spline_at(cx, ax, d) : 0.0;
            wy[d] = (ay >= wlo && ay <= whi) ? spline_at(cy, ay, d) : 0.0;
            wz[d] = (az >= wlo && az <= whi) ? spline_at(cz, az, d) : 0.0;
            // clamped to the window (one GPU: the grid, :216-221): a readable index for the cells whose weight is 0
            ox[d] = (min(max(ax, g.cx0()), g.cx1()) - g.cx0()) * sxs;
            oy[d] = (min(max(ay, g.cy0()), g.cy1()) - g.cy0()) * g.nz;
            oz[d] = min(max(az, g.cz0()), g.cz1()) - g.cz0();
        }
        double weight = 0, d0 = 0, d1 = 0, d2 = 0, q0 = 0, q1 = 0, q2 = 0;
#pragma unroll
        for (int xi = 0; xi < 3; ++xi)
#pragma unroll
            for (int yi = 0; yi < 3; ++yi)
#pragma unroll
                for (int zi = 0; zi < 3; ++zi) {
                    const int c = ox[xi] + oy[yi] + oz[zi];
                    const double cw = wx[xi] * wy[yi] * wz[zi];
                    weight += cw;
                    d0 += dcx[c] * cw;
                    d1 += dcy[c] * cw;
                    d2 += dcz[c] * cw;
                    if (pcx) {  // block-uniform
                        q0 += pcx[c] * cw;
                        q1 += pcy[c] * cw;
                        q2 += pcz[c] * cw;
                    }
                }
        double vx = p.vx[i], vy = p.vy[i], vz = p.vz[i];
        if (weight != 0) {  // :258-262
            vx += d0 / weight;
            vy += d1 / weight;
            vz += d2 / weight;
            if (pcx) {
                vx = blend * vx + (1.0 - blend) * (q0 / weight);
                vy = blend * vy + (1.0 - blend) * (q1 / weight);
                vz = blend * vz + (1.0 - blend) * (q2 / weight);
            }
            p.vx[i] = vx; p.vy[i] = vy; p.vz[i] = vz;
        }
        len = sqrt(vx * vx + vy * vy + vz * vz);  // Vec3::length, math/Vec3.h:224-230
        if (!(len == len)) len = 0;               // NaN never raises the max (maxSpeed < NaN is false, :982)
    }
    len = wave_max(len);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = len;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = sm[0];
        for (int k = 1; k < 4; ++k) m = sm[k] > m ? sm[k] : m;
        if (m > 0) atomicMax(&ss->max_speed_bits, (unsigned long long)__double_as_longlong(m));
    }
}

// ---- grid -> particle through LDS tiles (single GPU) --------------------------------------------------------
// Same arithmetic as k_g2p, other data path.  After the sort the particles of a cell row (fixed x,y; ascending z) are
// contiguous and their base cell IS their sort key, so a block takes a tile of G2P_TX x G2P_TY x G2P_TZ cells, stages
// the FLIP delta (and PIC) fields of the tile + 1 halo cell in LDS once, and its threads then gather the 27 x 3 values
// of each of the tile's ~3800 particles from LDS instead of sending 81 divergent loads per particle through the
// texture addresser (0.29 ms at 256^3 for k_g2p, the gathers of neighbouring particles hitting the same lines).
// Particles whose base cell is off the grid (last bucket) change nothing but their speed counts: k_g2p on that range.
constexpr int G2P_TX = 4, G2P_TY = 4, G2P_TZ = 30;
template <bool PIC>
__global__ __launch_bounds__(256) void k_g2p_tiled(Grid g, Box pb, Particles p, const int* __restrict__ cell_start,
                                                   const double* __restrict__ dcx, const double* __restrict__ dcy,
                                                   const double* __restrict__ dcz, const double* __restrict__ pcx,
                                                   const double* __restrict__ pcy, const double* __restrict__ pcz, double blend, StepState* ss)
{
    constexpr int LX = G2P_TX + 2, LY = G2P_TY + 2, LZ = G2P_TZ + 2, LN = LX * LY * LZ, NROW = G2P_TX * G2P_TY;
    __shared__ double sf[PIC ? 6 : 3][LN];
    __shared__ int srow[NROW][2];
    __shared__ double sm[4];
    const int tid = threadIdx.x;
    const int nty = (pb.ny() + G2P_TY - 1) / G2P_TY, ntz = (pb.nz() + G2P_TZ - 1) / G2P_TZ;
    const int tile = blockIdx.x;
    const int tz = tile % ntz, ty = (tile / ntz) % nty, tx = tile / (ntz * nty);
    const int x0 = pb.x0 + tx * G2P_TX, y0 = pb.y0 + ty * G2P_TY, z0 = pb.z0 + tz * G2P_TZ;
    const int z1 = z0 + G2P_TZ - 1 < pb.z1 ? z0 + G2P_TZ - 1 : pb.z1;
    if (tid < NROW) {  // particle range of every cell row of the tile
        const int cx = x0 + tid / G2P_TY, cy = y0 + tid % G2P_TY;
        int a = 0, b = 0;
        if (cx <= pb.x1 && cy <= pb.y1) {
            a = cell_start[g.idx(cx, cy, z0)];
            b = cell_start[g.idx(cx, cy, z1) + 1];
        }
        srow[tid][0] = a;
        srow[tid][1] = b;
    }
    for (int t = tid; t < LN; t += 256) {
        const int lx = t / (LY * LZ), r = t - lx * (LY * LZ), ly = r / LZ, lz = r - ly * LZ;
        const int gx = x0 - 1 + lx, gy = y0 - 1 + ly, gz = z0 - 1 + lz;
        const bool in = gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny && gz >= 0 && gz < g.nz;
        const size_t c = in ? g.idx(gx, gy, gz) : 0;
        sf[0][t] = in ? dcx[c] : 0.0;
        sf[1][t] = in ? dcy[c] : 0.0;
        sf[2][t] = in ? dcz[c] : 0.0;
        if (PIC) {
            sf[PIC ? 3 : 0][t] = in ? pcx[c] : 0.0;
            sf[PIC ? 4 : 0][t] = in ? pcy[c] : 0.0;
            sf[PIC ? 5 : 0][t] = in ? pcz[c] : 0.0;
        }
    }
    __syncthreads();
    const int wlo = g.lo + 2, whi = g.hi - 2;
    double len = 0;
    for (int row = 0; row < NROW; ++row) {
        const int a = srow[row][0], b = srow[row][1];
        for (int i = a + tid; i < b; i += 256) {
            const double cx = p.px[i], cy = p.py[i], cz = p.pz[i];
            const int fcx = (int)round(cx), fcy = (int)round(cy), fcz = (int)round(cz);
            // axis weights masked by "on the grid and within W" (k_g2p); local index of the base cell in the staged tile
            double wx[3], wy[3], wz[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int ax = fcx - 1 + d, ay = fcy - 1 + d, az = fcz - 1 + d;
                wx[d] = (ax >= wlo && ax <= whi) ? spline_at(cx, ax, d) : 0.0;
                wy[d] = (ay >= wlo && ay <= whi) ? spline_at(cy, ay, d) : 0.0;
                wz[d] = (az >= wlo && az <= whi) ? spline_at(cz, az, d) : 0.0;
            }
            const int lb = ((fcx - g.cx0() - x0) * LY + (fcy - g.cy0() - y0)) * LZ + (fcz - g.cz0() - z0);  // cell (base-1) in tile coordinates
            double weight = 0, d0 = 0, d1 = 0, d2 = 0, q0 = 0, q1 = 0, q2 = 0;
#pragma unroll
            for (int xi = 0; xi < 3; ++xi)
#pragma unroll
                for (int yi = 0; yi < 3; ++yi)
#pragma unroll
                    for (int zi = 0; zi < 3; ++zi) {
                        const int c = lb + (xi * LY + yi) * LZ + zi;
                        const double cw = wx[xi] * wy[yi] * wz[zi];
                        weight += cw;
                        d0 += sf[0][c] * cw;
                        d1 += sf[1][c] * cw;
                        d2 += sf[2][c] * cw;
                        if (PIC) {
                            q0 += sf[PIC ? 3 : 0][c] * cw;
                            q1 += sf[PIC ? 4 : 0][c] * cw;
                            q2 += sf[PIC ? 5 : 0][c] * cw;
                        }
                    }
            double vx = p.vx[i], vy = p.vy[i], vz = p.vz[i];
            if (weight != 0) {  // :258-262
                vx += d0 / weight;
                vy += d1 / weight;
                vz += d2 / weight;
                if (PIC) {
                    vx = blend * vx + (1.0 - blend) * (q0 / weight);
                    vy = blend * vy + (1.0 - blend) * (q1 / weight);
                    vz = blend * vz + (1.0 - blend) * (q2 / weight);
                }
                p.vx[i] = vx; p.vy[i] = vy; p.vz[i] = vz;
            }
            double l = sqrt(vx * vx + vy * vy + vz * vz);  // Vec3::length, math/Vec3.h:224-230
            if (!(l == l)) l = 0;                          // NaN never raises the max (maxSpeed < NaN is false, :982)
            len = l > len ? l : len;
        }
    }
    len = wave_max(len);
    if ((tid & 63) == 0) sm[tid >> 6] = len;
    __syncthreads();
    if (tid == 0) {
        double m = sm[0];
        for (int k = 1; k < 4; ++k) m = sm[k] > m ? sm[k] : m;
        if (m > 0) atomicMax(&ss->max_speed_bits, (unsigned long long)__double_as_longlong(m));
    }
}

__device__ __forceinline__ bool is_solid(const Grid& g, const uint8_t* flags, int x, int y, int z)
{
    // saccessor.getValue outside the filled box -> background 0 -> not solid (fluid.cc:46-57)
    // (a decomposed run: the window's halo ring holds every cell an owned particle can test)
    if (x < g.cx0() || x > g.cx1() || y < g.cy0() || y > g.cy1() || z < g.cz0() || z > g.cz1()) return false;
    return flags[g.idx(x - g.cx0(), y - g.cy0(), z - g.cz0())] & F_SOLID;
}

// fluid.cc:992-1036: new dt from maxSpeed, move, stuck-particle handling with e = 0.
__global__ __launch_bounds__(256) void k_advect(Grid g, long n, Particles p, const uint8_t* __restrict__ flags, double max_dt, double dx,
                                                StepState* ss)
{
    const double maxSpeed = __longlong_as_double((long long)ss->max_speed_bits);
    double timestep;
    if (maxSpeed != 0) timestep = max_dt < dx / maxSpeed ? max_dt : dx / maxSpeed;
    else timestep = max_dt;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && p.pid[i] != PID_DEAD) {
        const double e = 0;
        double P0 = p.px[i], P1 = p.py[i], P2 = p.pz[i];
        double V0 = p.vx[i], V1 = p.vy[i], V2 = p.vz[i];
        const double q0 = P0 + timestep * V0, q1 = P1 + timestep * V1, q2 = P2 + timestep * V2;
        if (is_solid(g, flags, (int)round(q0), (int)round(q1), (int)round(q2))) {
            const double vx = V0 * timestep, vy = V1 * timestep, vz = V2 * timestep;
            // Coord(double,double,double): the two untouched axes truncate toward zero (:1017-1025)
            if (is_solid(g, flags, (int)round(P0 + vx), (int)P1, (int)P2)) V0 *= -1.0 * e;
            if (is_solid(g, flags, (int)P0, (int)round(P1 + vy), (int)P2)) V1 *= -1.0 * e;
            if (is_solid(g, flags, (int)P0, (int)P1, (int)round(P2 + vz))) V2 *= -1.0 * e;
            P0 += V0 * timestep; P1 += V1 * timestep; P2 += V2 * timestep;
            p.vx[i] = V0; p.vy[i] = V1; p.vz[i] = V2;
        } else {
            P0 = q0; P1 = q1; P2 = q2;
        }
        p.px[i] = P0; p.py[i] = P1; p.pz[i] = P2;
    }
}

// dt is published by its own 1-thread launch AFTER k_advect so that no block of k_advect can
// read a max_speed/dt pair from two different steps.
__global__ void k_publish_dt(double max_dt, double dx, StepState* ss)
{
    const double maxSpeed = __longlong_as_double((long long)ss->max_speed_bits);
    ss->dt = (maxSpeed != 0) ? (max_dt < dx / maxSpeed ? max_dt : dx / maxSpeed) : max_dt;
}

// ---- multi-GPU: particle records (7 doubles: pos, vel, id) ----------------------------------------------
// records (7 doubles) -> SoA at p[off + j]
__global__ __launch_bounds__(256) void k_unpack_records(long n, const double* __restrict__ rec, Particles p, long off)
{
    long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const double* d = rec + (size_t)j * 7;
    p.px[off + j] = d[0]; p.py[off + j] = d[1]; p.pz[off + j] = d[2];
    p.vx[off + j] = d[3]; p.vy[off + j] = d[4]; p.vz[off + j] = d[5];
    p.pid[off + j] = (uint32_t)d[6];
}
// SoA p[off + j] -> records (7 doubles); ghosts of a boundary plane are one contiguous sorted range
__global__ __launch_bounds__(256) void k_pack_records(long n, Particles p, long off, double* __restrict__ rec)
{
    long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    double* d = rec + (size_t)j * 7;
    d[0] = p.px[off + j]; d[1] = p.py[off + j]; d[2] = p.pz[off + j];
    d[3] = p.vx[off + j]; d[4] = p.vy[off + j]; d[5] = p.vz[off + j];
    d[6] = (double)p.pid[off + j];
}
__global__ __launch_bounds__(256) void k_unpack_ids(long n, const double* __restrict__ pos, const double* __restrict__ vel,
                                                    const uint32_t* __restrict__ ids, Particles p)
{
    long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    p.px[j] = pos[3 * j]; p.py[j] = pos[3 * j + 1]; p.pz[j] = pos[3 * j + 2];
    if (vel) { p.vx[j] = vel[3 * j]; p.vy[j] = vel[3 * j + 1]; p.vz[j] = vel[3 * j + 2]; }
    else { p.vx[j] = 0; p.vy[j] = 0; p.vz[j] = 0; }
    p.pid[j] = ids[j];
}
// device order (no unsort): pos/vel AoS + ids
__global__ __launch_bounds__(256) void k_pack_ids(long n, Particles p, long off, double* __restrict__ pos, double* __restrict__ vel,
                                                  uint32_t* __restrict__ ids)
{
    long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    pos[3 * j] = p.px[off + j]; pos[3 * j + 1] = p.py[off + j]; pos[3 * j + 2] = p.pz[off + j];
    vel[3 * j] = p.vx[off + j]; vel[3 * j + 1] = p.vy[off + j]; vel[3 * j + 2] = p.vz[off + j];
    ids[j] = p.pid[off + j];
}

// ---- host <-> device particle layout -------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack(long n, Particles p, double* __restrict__ pos, double* __restrict__ vel)
{
    long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const size_t o = 3 * (size_t)p.pid[j];
    pos[o] = p.px[j]; pos[o + 1] = p.py[j]; pos[o + 2] = p.pz[j];
    vel[o] = p.vx[j]; vel[o + 1] = p.vy[j]; vel[o + 2] = p.vz[j];
}
__global__ __launch_bounds__(256) void k_unpack(long n, const double* __restrict__ pos, const double* __restrict__ vel, Particles p)
{
    long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    p.px[j] = pos[3 * j]; p.py[j] = pos[3 * j + 1]; p.pz[j] = pos[3 * j + 2];
    if (vel) { p.vx[j] = vel[3 * j]; p.vy[j] = vel[3 * j + 1]; p.vz[j] = vel[3 * j + 2]; }
    else { p.vx[j] = 0; p.vy[j] = 0; p.vz[j] = 0; }
    p.pid[j] = (uint32_t)j;
}

static inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

void launch_bin_count(hipStream_t st, Grid g, long n, Particles p, int* key, int* slot, int* cell_count, int* part, StepState* ss)
{
    if (n <= 0) return;
    unsigned nb = nblk(n);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_bin_count, dim3(nb), dim3(256), 0, st, g, n, p, key, slot, cell_count, part);
    hipLaunchKernelGGL(k_bin_bbox, dim3(1), dim3(256), 0, st, (const int*)part, (int)nb, ss);
}
void launch_bin_scatter(hipStream_t st, long n, const int* key, const int* slot, const int* cell_start, const uint32_t* pid, int* order,
                         uint32_t* spid)
{
    if (n > 0) hipLaunchKernelGGL(k_bin_scatter, dim3(nblk(n)), dim3(256), 0, st, n, key, slot, cell_start, pid, order, spid);
}
// positions [pos0, pos0+n_pos) of the sorted array (all of them on one GPU; the owned slab without ghosts on a rank)
void launch_bin_rank(hipStream_t st, long n_pos, long pos0, const int* key, const int* cell_start, const int* order, const uint32_t* spid,
                     int* order2)
{
    if (n_pos > 0) hipLaunchKernelGGL(k_bin_rank, dim3(nblk(n_pos)), dim3(256), 0, st, n_pos, pos0, key, cell_start, order, spid, order2);
}
void launch_reorder(hipStream_t st, long n, const int* order, Particles src, Particles dst, double* w, long wstride)
{
    if (n > 0) hipLaunchKernelGGL(k_reorder, dim3(nblk(n)), dim3(256), 0, st, n, order, src, dst, w, wstride);
}
void launch_weights(hipStream_t st, long n, Particles p, double* w, long stride)
{
    if (n > 0) hipLaunchKernelGGL(k_weights, dim3(nblk(n)), dim3(256), 0, st, n, p, w, stride);
}
// z pieces and regular y segments of a P2G launch over box; items needed: 4 + 4 * p2g_max_items(box) ints
static void p2g_cut(const Box& box, int& ntz, int& zt, int& nseg)
{
    // z is cut into equal pieces of at most P2G_ZT cells (89 cells: 45 + 44, not 62 + 27): balanced blocks, and at 8 particles
    // per cell a row piece of <= 46 cells fits one staged chunk
    ntz = (box.nz() + P2G_ZT - 1) / P2G_ZT;
    zt = (box.nz() + ntz - 1) / ntz;
    // y segments: a segment of ys columns costs ys + 2 staged rows, and the blocks run in rounds of P2G_SLOTS at a time;
    // take the count with the shortest estimated makespan (256^3 bench box, 92 columns: 5 segments, 930 blocks, one round)
    nseg = 1;
    long best = -1;
    const long per = (long)(box.nx() + 2) * ntz;
    for (int k = 1; k <= box.ny(); ++k) {
        const long ys = (box.ny() + k - 1) / k, rounds = (per * k + P2G_SLOTS - 1) / P2G_SLOTS, cost = rounds * (ys + 2);
        if (best < 0 || cost < best) { best = cost; nseg = k; }
    }
}
// doubles of the row form's partial buffer over box: 12 per cell + the column-piece flags
size_t p2g_part_doubles(Box box)
{
    int ntz, zt, nseg;
    p2g_cut(box, ntz, zt, nseg);
    return (size_t)12 * box.cells() + ((size_t)3 * box.nx() * box.ny() * ntz + 7) / 8 + 8;
}
long p2g_max_items(Box box)
{
    int ntz, zt, nseg;
    p2g_cut(box, ntz, zt, nseg);
    return (long)(box.nx() + 2) * box.ny() * ntz;  // every segment cut down to single columns
}
// pw: the particles' axis weights (launch_reorder / launch_weights); part: 12 doubles per cell of box; items: work list,
// 4 + 4 * p2g_max_items(box) ints, items[0] == 0 on entry and on exit
// crowd_list != nullptr (piled particles, mostly-air box): the cells of >= P2G_CROWD particles are summed on the matrix cores first
// (list of at least n_particles / 16 + 64 int2, `park` = the other particle buffer), the rows walk the rest
void launch_p2g(hipStream_t st, Grid g, Box box, Particles p, const double* pw, long wstride, const int* cell_start, const uint8_t* flags,
                double* part, int* items, float* container, double* u, double* v, double* w, double* ub, double* vb, double* wb, int2* crowd_list,
                Particles park)
{
    int ntz, zt, nseg;
    p2g_cut(box, ntz, zt, nseg);
    const long cells = box.cells();
    const int budget = P2G_BUDGET;
    const unsigned nt = (unsigned)((box.nx() + 2) * nseg * ntz);  // the regular cut fills the chip; further items are taken in a grid-stride loop
    hipLaunchKernelGGL(k_p2g_items, dim3(nblk((long)(box.nx() + 2) * nseg)), dim3(256), 0, st, g, box, cell_start, nseg, ntz, budget, items);
    const bool vec = ((((uintptr_t)pw | (uintptr_t)p.vx | (uintptr_t)p.vy | (uintptr_t)p.vz) & 15) == 0) && (wstride & 1) == 0;
    uint8_t* colflag = reinterpret_cast<uint8_t*>(part + 12 * cells);   // 3 x nx x ny x ntz bytes behind the partials (p2g_part_doubles)
    if (crowd_list) {
        // the source cells of box: one cell further on every side (a decomposed run holds particles there)
        Box sb = box;
        sb.x0 = sb.x0 > 0 ? sb.x0 - 1 : 0, sb.y0 = sb.y0 > 0 ? sb.y0 - 1 : 0, sb.z0 = sb.z0 > 0 ? sb.z0 - 1 : 0;
        sb.x1 = sb.x1 < g.nx - 1 ? sb.x1 + 1 : g.nx - 1, sb.y1 = sb.y1 < g.ny - 1 ? sb.y1 + 1 : g.ny - 1, sb.z1 = sb.z1 < g.nz - 1 ? sb.z1 + 1 : g.nz - 1;
        hipLaunchKernelGGL(k_p2g_crowd_list, dim3(nblk((long)sb.nx() * sb.ny() * ((sb.nz() + 3) / 4))), dim3(256), 0, st, g, sb, cell_start, crowd_list, items + 1);
        hipLaunchKernelGGL(k_p2g_crowd_sum, dim3(2048), dim3(64 * PC_WAVES), 0, st, p, pw, wstride, crowd_list, items + 1, park);
        if (vec) hipLaunchKernelGGL((k_p2g_rows<true, true>), dim3(nt), dim3(P2G_THREADS), 0, st, g, box, p, pw, wstride, cell_start, part, cells, zt, items, park, colflag, ntz);
        else hipLaunchKernelGGL((k_p2g_rows<false, true>), dim3(nt), dim3(P2G_THREADS), 0, st, g, box, p, pw, wstride, cell_start, part, cells, zt, items, park, colflag, ntz);
    } else if (vec) hipLaunchKernelGGL((k_p2g_rows<true, false>), dim3(nt), dim3(P2G_THREADS), 0, st, g, box, p, pw, wstride, cell_start, part, cells, zt, items, park, colflag, ntz);
    else hipLaunchKernelGGL((k_p2g_rows<false, false>), dim3(nt), dim3(P2G_THREADS), 0, st, g, box, p, pw, wstride, cell_start, part, cells, zt, items, park, colflag, ntz);
    hipLaunchKernelGGL(k_p2g_combine, dim3(nblk(cells)), dim3(256), 0, st, g, box, part, cells, flags, container, u, v, w, ub, vb, wb, items, colflag, zt, ntz);
}
void launch_p2g_tiles(hipStream_t st, Grid g, Box box, Particles p, const double* pw, long wstride, const int* cell_start, const uint8_t* flags,
                      float* container, double* u, double* v, double* w, double* ub, double* vb, double* wb)
{
    const int ntz = (box.nz() + P2G_ZT - 1) / P2G_ZT, zt = (box.nz() + ntz - 1) / ntz;
    const unsigned nt = (unsigned)(((box.nx() + P2GT_T - 1) / P2GT_T) * ((box.ny() + P2GT_T - 1) / P2GT_T) * ntz);
    hipLaunchKernelGGL(k_p2g_tiles, dim3(nt), dim3(P2GT_THREADS), 0, st, g, box, p, pw, wstride, cell_start, flags, container, u, v, w, ub, vb, wb, zt);
}
// pb = bounding box of the particles' base cells (after the sort, positions untouched since)
void launch_g2p_tiled(hipStream_t st, Grid g, Box pb, Particles p, const int* cell_start, const double* dcx, const double* dcy, const double* dcz,
                      const double* pcx, const double* pcy, const double* pcz, double blend, StepState* ss)
{
    if (pb.cells() <= 0) return;
    const unsigned nt = (unsigned)(((pb.nx() + G2P_TX - 1) / G2P_TX) * ((pb.ny() + G2P_TY - 1) / G2P_TY) * ((pb.nz() + G2P_TZ - 1) / G2P_TZ));
    if (pcx) hipLaunchKernelGGL(k_g2p_tiled<true>, dim3(nt), dim3(256), 0, st, g, pb, p, cell_start, dcx, dcy, dcz, pcx, pcy, pcz, blend, ss);
    else hipLaunchKernelGGL(k_g2p_tiled<false>, dim3(nt), dim3(256), 0, st, g, pb, p, cell_start, dcx, dcy, dcz, pcx, pcy, pcz, blend, ss);
}
void launch_g2p(hipStream_t st, Grid g, long n, Particles p, const double* dcx, const double* dcy, const double* dcz, const double* pcx,
                const double* pcy, const double* pcz, double blend, StepState* ss)
{
    if (n > 0) hipLaunchKernelGGL(k_g2p, dim3(nblk(n)), dim3(256), 0, st, g, n, p, dcx, dcy, dcz, pcx, pcy, pcz, blend, ss);
}
void launch_advect(hipStream_t st, Grid g, long n, Particles p, const uint8_t* flags, double max_dt, double dx, StepState* ss)
{
    if (n > 0) hipLaunchKernelGGL(k_advect, dim3(nblk(n)), dim3(256), 0, st, g, n, p, flags, max_dt, dx, ss);
    hipLaunchKernelGGL(k_publish_dt, dim3(1), dim3(1), 0, st, max_dt, dx, ss);
}
void launch_unpack_records(hipStream_t st, long n, const double* rec, Particles p, long off)
{
    if (n > 0) hipLaunchKernelGGL(k_unpack_records, dim3(nblk(n)), dim3(256), 0, st, n, rec, p, off);
}
void launch_pack_records(hipStream_t st, long n, Particles p, long off, double* rec)
{
    if (n > 0) hipLaunchKernelGGL(k_pack_records, dim3(nblk(n)), dim3(256), 0, st, n, p, off, rec);
}
void launch_unpack_ids(hipStream_t st, long n, const double* pos, const double* vel, const uint32_t* ids, Particles p)
{
    if (n > 0) hipLaunchKernelGGL(k_unpack_ids, dim3(nblk(n)), dim3(256), 0, st, n, pos, vel, ids, p);
}
void launch_pack_ids(hipStream_t st, long n, Particles p, long off, double* pos, double* vel, uint32_t* ids)
{
    if (n > 0) hipLaunchKernelGGL(k_pack_ids, dim3(nblk(n)), dim3(256), 0, st, n, p, off, pos, vel, ids);
}
void launch_pack_particles(hipStream_t st, long n, Particles p, double* pos_aos, double* vel_aos)
{
    if (n > 0) hipLaunchKernelGGL(k_pack, dim3(nblk(n)), dim3(256), 0, st, n, p, pos_aos, vel_aos);
}
void launch_unpack_particles(hipStream_t st, long n, const double* pos_aos, const double* vel_aos, Particles p)
{
    if (n > 0) hipLaunchKernelGGL(k_unpack, dim3(nblk(n)), dim3(256), 0, st, n, pos_aos, vel_aos, p);
}

}  // namespace fl
