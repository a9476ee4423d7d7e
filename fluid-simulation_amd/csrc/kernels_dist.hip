// Kernels of the decomposed (multi-GPU) step only: halo pack / unpack for any dense layout, flags of a box, the global
// unknown numbering from row counts, particle migration / ghost classification, ownership masks of the solver.
#include "common.h"
#include "dist_kernels.h"

namespace fl {

// ---- halo pack / unpack ---------------------------------------------------------------------------------------
// blockIdx.y = box * narr + array; a block strides over the elements of its box.  Staging layout: per box (= per peer)
// contiguous, the arrays one after the other — what one ncclSend / ncclRecv moves.
template <typename E, bool PACK>
__global__ __launch_bounds__(256) void k_halo_copy(HaloArgs a, E* __restrict__ stage)
{
    const int b = blockIdx.y / a.narr, q = blockIdx.y - b * a.narr;
    const int n0 = a.n[b][0], n1 = a.n[b][1], n2 = a.n[b][2];
    const long cnt = (long)n0 * n1 * n2;
    E* arr = (E*)a.arr[0];
#pragma unroll
    for (int k = 1; k < HALO_MAX_ARR; ++k) arr = (q == k) ? (E*)a.arr[k] : arr;   // static indices into the kernarg
    E* st = stage + (size_t)a.off[b] * a.narr + (size_t)q * cnt;
    const long base = a.base + (long)a.lo[b][0] * a.sx + (long)a.lo[b][1] * a.sy + a.lo[b][2];
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < cnt; e += (long)gridDim.x * 256) {
        const int k = (int)(e % n2), j = (int)((e / n2) % n1), i = (int)(e / ((long)n2 * n1));
        const long c = base + (long)i * a.sx + (long)j * a.sy + k;
        if (PACK) st[e] = arr[c];
        else arr[c] = st[e];
    }
}

template <typename E>
static void halo_launch(hipStream_t st, const HaloArgs& a, void* stage, bool pack)
{
    if (a.nbox <= 0 || a.narr <= 0) return;
    long mx = 1;
    for (int b = 0; b < a.nbox; ++b) mx = std::max(mx, (long)a.n[b][0] * a.n[b][1] * a.n[b][2]);
    unsigned gx = (unsigned)std::min<long>((mx + 255) / 256, 256);
    if (pack) hipLaunchKernelGGL((k_halo_copy<E, true>), dim3(gx, a.nbox * a.narr), dim3(256), 0, st, a, (E*)stage);
    else hipLaunchKernelGGL((k_halo_copy<E, false>), dim3(gx, a.nbox * a.narr), dim3(256), 0, st, a, (E*)stage);
}
void launch_halo_copy(hipStream_t st, const HaloArgs& a, int elem, void* stage, bool pack)
{
    if (elem == 1) halo_launch<uint8_t>(st, a, stage, pack);
    else if (elem == 4) halo_launch<uint32_t>(st, a, stage, pack);
    else halo_launch<unsigned long long>(st, a, stage, pack);
}

// ---- flags of a box -------------------------------------------------------------------------------------------
// k_flags (kernels_grid.hip) over a box of the window instead of whole x planes: the owned block of a rank
__global__ __launch_bounds__(256) void k_flags_box(Grid g, Box box, const uint8_t* __restrict__ solid, const float* __restrict__ container,
                                                   uint8_t* __restrict__ flags)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= box.cells()) return;
    const int nz = box.nz(), ny = box.ny();
    const int iz = (int)(t % nz) + box.z0, iy = (int)((t / nz) % ny) + box.y0, ix = (int)(t / ((long)nz * ny)) + box.x0;
    const size_t c = g.idx(ix, iy, iz);
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
void launch_flags_box(hipStream_t st, Grid g, Box box, const uint8_t* solid, const float* container, uint8_t* flags)
{
    if (box.cells() > 0) hipLaunchKernelGGL(k_flags_box, dim3((unsigned)((box.cells() + 255) / 256)), dim3(256), 0, st, g, box, solid, container, flags);
}

// ---- unknown numbering across blocks (fluid.cc:1416-1433) -------------------------------------------------------------
// The reference numbers the fluid cells in x-major / z-fastest order over the whole grid.  With 3-D blocks a grid row
// (x, y) is cut into dims[2] segments owned by different ranks: every rank counts the fluid cells of its row segments
// into a table [x][y][segment] over the rows of the active box (zeros elsewhere), one SUM all-reduce fills the table on
// every rank, an exclusive scan over it (the table order IS the numbering order) gives each segment's first number, and
// the owner numbers its segment from there.
// rows: table of (RX x RY x nseg) ints over the box rows; this rank fills column `seg` for the rows inside own.
__global__ __launch_bounds__(256) void k_row_counts(Grid g, Box own, int rx0, int ry0, int RY, int nseg, int seg, const uint8_t* __restrict__ flags,
                                                    int* __restrict__ rows)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int ny = own.ny();
    if (t >= own.nx() * ny) return;
    const int ix = own.x0 + t / ny, iy = own.y0 + t % ny;
    const uint8_t* f = flags + g.idx(ix, iy, own.z0);
    int c = 0;
    for (int k = 0; k < own.nz(); ++k) c += (f[k] & F_FLUID) ? 1 : 0;
    rows[((size_t)(ix + g.ox - rx0) * RY + (iy + g.oy - ry0)) * nseg + seg] = c;
}
__global__ __launch_bounds__(256) void k_row_number(Grid g, Box own, int rx0, int ry0, int RY, int nseg, int seg, const uint8_t* __restrict__ flags,
                                                    const int* __restrict__ starts, int* __restrict__ indices)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int ny = own.ny();
    if (t >= own.nx() * ny) return;
    const int ix = own.x0 + t / ny, iy = own.y0 + t % ny;
    const size_t c0 = g.idx(ix, iy, own.z0);
    int n = starts[((size_t)(ix + g.ox - rx0) * RY + (iy + g.oy - ry0)) * nseg + seg];
    for (int k = 0; k < own.nz(); ++k) {
        const bool fl = flags[c0 + k] & F_FLUID;
        indices[c0 + k] = fl ? n : -1;
        n += fl ? 1 : 0;
    }
}
__global__ __launch_bounds__(256) void k_fill_box_int(Grid g, Box box, int* __restrict__ a, int v)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= box.cells()) return;
    const int nz = box.nz(), ny = box.ny();
    a[g.idx((int)(t / ((long)nz * ny)) + box.x0, (int)((t / nz) % ny) + box.y0, (int)(t % nz) + box.z0)] = v;
}
void launch_row_counts(hipStream_t st, Grid g, Box own, int rx0, int ry0, int RY, int nseg, int seg, const uint8_t* flags, int* rows)
{
    const long n = (long)own.nx() * own.ny();
    if (n > 0 && own.nz() > 0) hipLaunchKernelGGL(k_row_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, own, rx0, ry0, RY, nseg, seg, flags, rows);
}
void launch_row_number(hipStream_t st, Grid g, Box own, int rx0, int ry0, int RY, int nseg, int seg, const uint8_t* flags, const int* starts,
                       int* indices)
{
    const long n = (long)own.nx() * own.ny();
    if (n > 0 && own.nz() > 0) hipLaunchKernelGGL(k_row_number, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, own, rx0, ry0, RY, nseg, seg, flags, starts, indices);
}
void launch_fill_box_int(hipStream_t st, Grid g, Box box, int* a, int v)
{
    if (box.cells() > 0) hipLaunchKernelGGL(k_fill_box_int, dim3((unsigned)((box.cells() + 255) / 256)), dim3(256), 0, st, g, box, a, v);
}

// ---- particles: migration and ghosts in ONE routing round ---------------------------------------------------------------
// Direction code d = (dx+1)*9 + (dy+1)*3 + (dz+1), 13 = me.  Every block within one cell of a particle's base cell needs
// the particle: the block that holds the base cell as its OWNER, the others as a GHOST for their P2G sums over the cells
// next to the face (support base-1..base+1, fluid.cc:271-276).  At the start of a step every live particle sits in the
// arrays of the block that owned it last step, and CFL (at most one cell per step, fluid.cc:992-999) keeps its base cell
// within one cell of that block — so the previous owner sends ONE copy to every adjacent block within one cell of the
// base cell (up to 7: faces, edges, corner) and keeps the particle itself: as owner if the base cell is still its own, as a
// ghost if it has just left (the receiver tells owner from ghost by the base cell alone).  A side without a neighbour
// block is the edge of the grid: nothing is sent there (off the grid a particle is inert and stays with the edge block).
__device__ __forceinline__ void base_of(const Grid& g, const Particles& p, long i, int b[3])
{
    b[0] = (int)round(p.px[i]) - g.lo; b[1] = (int)round(p.py[i]) - g.lo; b[2] = (int)round(p.pz[i]) - g.lo;   // GLOBAL index
}
__device__ __forceinline__ void put_record(double* d, const Particles& p, long i)
{
    d[0] = p.px[i]; d[1] = p.py[i]; d[2] = p.pz[i];
    d[3] = p.vx[i]; d[4] = p.vy[i]; d[5] = p.vz[i];
    d[6] = (double)p.pid[i];
}
// PASS 0: count per direction into cnt[27]; PASS 1: write the records at cursor[d]++ (cursors preset to the directions' offsets)
template <int PASS>
__global__ __launch_bounds__(256) void k_route(Grid g, OwnBox ob, long n, Particles p, int* __restrict__ cnt, double* __restrict__ rec)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (p.pid[i] == PID_DEAD) return;
    int b[3];
    base_of(g, p, i, b);
    int s[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)   // the neighbour slab of axis a the base cell is within one cell of (blocks are >= 8 wide: at most one)
        s[a] = (ob.has_lo[a] && b[a] <= ob.lo[a]) ? -1 : ((ob.has_hi[a] && b[a] >= ob.hi[a] - 1) ? 1 : 0);
    if (!(s[0] | s[1] | s[2])) return;
    for (int m = 1; m < 8; ++m) {   // every non-empty sub-selection of the touched sides
        if (((m & 1) && !s[0]) || ((m & 2) && !s[1]) || ((m & 4) && !s[2])) continue;
        const int e0 = (m & 1) ? s[0] : 0, e1 = (m & 2) ? s[1] : 0, e2 = (m & 4) ? s[2] : 0;
        const int d = (e0 + 1) * 9 + (e1 + 1) * 3 + (e2 + 1);
        const int k = atomicAdd(&cnt[d], 1);
        if (PASS == 1) put_record(rec + (size_t)k * 7, p, i);
    }
}
void launch_route(hipStream_t st, Grid g, OwnBox ob, long n, Particles p, int* cnt, double* rec, int pass)
{
    if (n <= 0) return;
    const dim3 gr((unsigned)((n + 255) / 256)), bl(256);
    if (pass == 0) hipLaunchKernelGGL((k_route<0>), gr, bl, 0, st, g, ob, n, p, cnt, rec);
    else hipLaunchKernelGGL((k_route<1>), gr, bl, 0, st, g, ob, n, p, cnt, rec);
}

// After P2G the ghosts (copies from the neighbours and own particles that have just left the block) have served: mark
// every particle whose base cell is not in the owned block dead (G2P / advect skip it, the next sort drops the dead bucket).  Sides without a neighbour extend to infinity: off-grid particles belong to the edge block.
__global__ __launch_bounds__(256) void k_kill_ghosts(Grid g, OwnBox ob, long n, Particles p)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int b[3];
    base_of(g, p, i, b);
    bool own = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) own = own && !(b[a] < ob.lo[a] && ob.has_lo[a]) && !(b[a] >= ob.hi[a] && ob.has_hi[a]);
    if (!own) p.pid[i] = PID_DEAD;
}
void launch_kill_ghosts(hipStream_t st, Grid g, OwnBox ob, long n, Particles p)
{
    if (n > 0) hipLaunchKernelGGL(k_kill_ghosts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, ob, n, p);
}

// live particles -> AoS pos / vel / ids at an atomic cursor (device order is not kept: the caller sorts by id)
__global__ __launch_bounds__(256) void k_pack_live(long n, Particles p, double* __restrict__ pos, double* __restrict__ vel, uint32_t* __restrict__ ids,
                                                   int* __restrict__ cursor)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || p.pid[i] == PID_DEAD) return;
    const int j = atomicAdd(cursor, 1);
    if (pos) {
        pos[3 * (size_t)j] = p.px[i]; pos[3 * (size_t)j + 1] = p.py[i]; pos[3 * (size_t)j + 2] = p.pz[i];
        vel[3 * (size_t)j] = p.vx[i]; vel[3 * (size_t)j + 1] = p.vy[i]; vel[3 * (size_t)j + 2] = p.vz[i];
        ids[j] = p.pid[i];
    }
}
void launch_pack_live(hipStream_t st, long n, Particles p, double* pos, double* vel, uint32_t* ids, int* cursor)
{
    if (n > 0) hipLaunchKernelGGL(k_pack_live, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, p, pos, vel, ids, cursor);
}

// ---- solver ownership ----------------------------------------------------------------------------------------------
// cnt_pcg = the count byte of the local solver layout restricted to the owned cells + one ring (bit 7 set on the ring:
// the search vector is formed there, q and the dot products are not).  own = owned cells in LBox coordinates
// (interior index, 1-based x/y, K0-based z as in LBox), inclusive.
__global__ __launch_bounds__(256) void k_cnt_pcg(LBox L, Box own, const uint8_t* __restrict__ cnt, uint8_t* __restrict__ out)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)L.cells() + 2 * L.Lz) return;
    const int k = (int)(t % L.Lz), j = (int)((t / L.Lz) % L.Ly), i = (int)(t / ((long)L.Lz * L.Ly));
    uint8_t c = 0;
    if (t < (long)L.cells()) {
        c = cnt[t];
        const bool in = i >= own.x0 && i <= own.x1 && j >= own.y0 && j <= own.y1 && k >= own.z0 && k <= own.z1;
        const bool ring = i >= own.x0 - 1 && i <= own.x1 + 1 && j >= own.y0 - 1 && j <= own.y1 + 1 && k >= own.z0 - 1 && k <= own.z1 + 1;
        if (!ring) c = 0;
        else if (!in && c) c |= 0x80;
    }
    out[t] = c;
}
void launch_cnt_pcg(hipStream_t st, LBox L, Box own, const uint8_t* cnt, uint8_t* out)
{
    hipLaunchKernelGGL(k_cnt_pcg, dim3((unsigned)((L.cells() + 2 * L.Lz + 255) / 256)), dim3(256), 0, st, L, own, cnt, out);
}

// Gather of a replicated level: every rank has written its owned cells of the level's (global-layout) array; zero the rest
// of the domain, and the SUM all-reduce of the array over the ranks assembles the level (every cell has one owner, the
// others add exact zeros).  own: in the level's cell coordinates, inclusive (empty: x1 < x0).
template <typename T>
__global__ __launch_bounds__(256) void k_mask_outside(MLevel m, Box own, T* __restrict__ a)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)m.dx * m.dy * m.dz) return;
    const int k = (int)(t % m.dz), j = (int)((t / m.dz) % m.dy), i = (int)(t / ((long)m.dz * m.dy));
    const bool in = i >= own.x0 && i <= own.x1 && j >= own.y0 && j <= own.y1 && k >= own.z0 && k <= own.z1;
    if (!in) a[m.at(i, j, k)] = 0;
}
template <typename T>
void launch_mask_outside(hipStream_t st, MLevel m, Box own, T* a)
{
    const long n = (long)m.dx * m.dy * m.dz;
    if (n > 0) hipLaunchKernelGGL((k_mask_outside<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m, own, a);
}
template void launch_mask_outside<float>(hipStream_t, MLevel, Box, float*);
template void launch_mask_outside<double>(hipStream_t, MLevel, Box, double*);
template void launch_mask_outside<uint8_t>(hipStream_t, MLevel, Box, uint8_t*);

// level-0 type of the cells of a LOCAL level-0 domain (k_mg_type0 with an explicit cell origin): 0 solid / off the window, 1 air, 2 unknown.
// Domain cell (i, j, k) = window cell (w0 + i, w1 + j, w2 + k); cnt = count bytes in the level's own layout.
__global__ __launch_bounds__(256) void k_mg_type_local(Grid g, MLevel m, int w0, int w1, int w2, const uint8_t* __restrict__ flags,
                                                       const uint8_t* __restrict__ cnt, uint8_t* __restrict__ typ)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)m.dx * m.dy * m.dz) return;
    const int k = (int)(t % m.dz), j = (int)((t / m.dz) % m.dy), i = (int)(t / ((long)m.dz * m.dy));
    const int gx = w0 + i, gy = w1 + j, gz = w2 + k;
    const size_t c = m.at(i, j, k);
    uint8_t ty = 0;
    if (gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny && gz >= 0 && gz < g.nz) {
        const uint8_t f = flags[g.idx(gx, gy, gz)];
        ty = (f & F_SOLID) ? 0 : (cnt[c] ? 2 : 1);
    }
    typ[c] = ty;
}
void launch_mg_type_local(hipStream_t st, Grid g, MLevel m, int w0, int w1, int w2, const uint8_t* flags, const uint8_t* cnt, uint8_t* typ)
{
    const long n = (long)m.dx * m.dy * m.dz;
    if (n > 0) hipLaunchKernelGGL(k_mg_type_local, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, m, w0, w1, w2, flags, cnt, typ);
}

// velBeforeUpdate = vels.deepCopy() (fluid.cc:1455) over a box: the cells gathered from the other blocks
__global__ __launch_bounds__(256) void k_copy_vel_before(Grid g, Box box, const double* __restrict__ u, const double* __restrict__ v,
                                                         const double* __restrict__ w, double* __restrict__ ub, double* __restrict__ vb,
                                                         double* __restrict__ wb)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= box.cells()) return;
    const int nz = box.nz(), ny = box.ny();
    const size_t c = g.idx((int)(t / ((long)nz * ny)) + box.x0, (int)((t / nz) % ny) + box.y0, (int)(t % nz) + box.z0);
    ub[c] = u[c]; vb[c] = v[c]; wb[c] = w[c];
}
void launch_copy_vel_before(hipStream_t st, Grid g, Box box, const double* u, const double* v, const double* w, double* ub, double* vb, double* wb)
{
    if (box.cells() > 0) hipLaunchKernelGGL(k_copy_vel_before, dim3((unsigned)((box.cells() + 255) / 256)), dim3(256), 0, st, g, box, u, v, w, ub, vb, wb);
}

// P2G result of the owned cells of the active box, packed for the SUM all-reduce of the replicated mode
// (k_pack_box with a 3-D ownership box; own in window coordinates, inclusive)
__global__ __launch_bounds__(256) void k_pack_box_own(Grid g, Box box, Box own, const float* __restrict__ container, const double* __restrict__ u,
                                                      const double* __restrict__ v, const double* __restrict__ w, double* __restrict__ buf)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = box.cells();
    if (t >= n) return;
    const int nz = box.nz(), ny = box.ny();
    const int iz = (int)(t % nz) + box.z0, iy = (int)((t / nz) % ny) + box.y0, ix = (int)(t / ((long)nz * ny)) + box.x0;
    const size_t c = g.idx(ix, iy, iz);
    const bool o = ix >= own.x0 && ix <= own.x1 && iy >= own.y0 && iy <= own.y1 && iz >= own.z0 && iz <= own.z1;
    buf[t] = o ? (double)container[c] : 0.0;
    buf[n + t] = o ? u[c] : 0.0;
    buf[2 * n + t] = o ? v[c] : 0.0;
    buf[3 * n + t] = o ? w[c] : 0.0;
}
void launch_pack_box_own(hipStream_t st, Grid g, Box box, Box own, const float* container, const double* u, const double* v, const double* w, double* buf)
{
    if (box.cells() > 0) hipLaunchKernelGGL(k_pack_box_own, dim3((unsigned)((box.cells() + 255) / 256)), dim3(256), 0, st, g, box, own, container, u, v, w, buf);
}

// Active-tile lists of a decomposed rank's level-0 down leg: act[t] = tile t holds an unknown, cls[t] = 1 where the tile reads a cell the
// residual's halo exchange writes: fi = active tiles that can be swept while the exchange is in flight, fb = the active ones that cannot
__global__ __launch_bounds__(256) void k_split_flags(const uint8_t* __restrict__ act, const uint8_t* __restrict__ cls, int n, uint8_t* __restrict__ fi,
                                                     uint8_t* __restrict__ fb)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const uint8_t a = act[t], c = cls[t];
    fi[t] = a && !c;
    fb[t] = a && c;
}
void launch_split_flags(hipStream_t st, const uint8_t* act, const uint8_t* cls, int n, uint8_t* fi, uint8_t* fb)
{
    if (n > 0) hipLaunchKernelGGL(k_split_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, act, cls, n, fi, fb);
}

}  // namespace fl
