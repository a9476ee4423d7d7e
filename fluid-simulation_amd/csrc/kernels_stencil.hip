// Dense sweep q = A s from HBM through an LDS-DMA plane ring (gfx950 only).
//
// A = the reference's pressure matrix (setA, fluid.cc:304-412): a function of the per-cell flag byte, applied matrix-free.
// This is the bandwidth-bound kernel of the path ((2T+1) bytes per cell, x N^3) — the "pressure-stencil kernel
// at 256^3" of the north star.  Every register-staged march of rounds 1-3 stopped at ~4.9 TB/s; this form moves the operands
// with the CDNA4 transport instead (`global_load_lds_dwordx4`: HBM -> LDS with no VGPR in between):
//
//   * the grid is cut into x chunks, one per XCD (blocks b and b + 8 share an XCD), and every plane of a chunk into slabs of
//     RY whole z rows, one per workgroup: the 32 CUs of an XCD together fetch WHOLE planes, so each XCD advances one linear
//     front through s, one through the flags and one through q; chunks are one plane longer than N / 8, so the eight fronts
//     do not sit at equal offsets of their power-of-two-sized regions (aligned fronts cost 8-10 % of the HBM rate);
//   * NL loader waves fill a ring of D + 2 LDS slots — a slot = the slab's RY + 2 rows of s and of flag bytes of one plane,
//     lane-linear (16 bytes per lane, 1 KiB per instruction; the source address is per lane, so ragged row lengths cost
//     nothing) — and keep D planes in flight behind a counted `s_waitcnt vmcnt`; they do nothing else;
//   * NC consumer waves own 16-byte pieces of the slab's rows (NP per lane), keep x-1 and x in registers and read x+1, the
//     y neighbours, the z neighbours and the flag bytes from the ring; results leave by non-temporal 16-byte stores;
//   * one raw `s_barrier` per plane orders loaders and consumers (no `__syncthreads()`: its fence would drain the DMA queue).
//
// Same term order as the other forms (x-, x+, y-, y+, z-, z+; a + b is commutative, so the marching direction does not show):
// bit-identical results, `test_marching_stencil_forms_agree`.
#include "common.h"
#include "stencil_vec.h"

namespace fl {

struct DmaGeo {
    int N, ppr, fpr;          // cells per axis; 16-byte pieces per row of s / of flag bytes
    int RY;                   // rows a block owns; it stages RY + 2
    int GS, GTOT;             // LDS-DMA instructions per plane and slab: s, s + flags
    int NL, NC, D;            // loader waves, consumer waves, planes in flight
    int slot_bytes;           // GTOT KiB; the flag bytes start at GS KiB
    int nslab, nchunk, cxlen;
    int own_pieces;           // RY * ppr
};

template <int G>
__device__ __forceinline__ void dma_wait_groups(int c)   // wait until at most c groups of G LDS-DMA loads are outstanding
{
    switch (c) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * G) : "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * G) : "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * G) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * G) : "memory"); break;
    }
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <typename T_, int G, int NP>
__global__ __launch_bounds__(1024) void k_stencil_dma(DmaGeo geo, const uint8_t* __restrict__ flags, const T_* __restrict__ s, T_* __restrict__ q, Coef<T_> cf)
{
    constexpr int V = 16 / (int)sizeof(T_);
    typedef typename VecT<T_, V>::type vec;
    typedef typename FlagT<V>::type fvec;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // ring slots, then the coefficient table: ONE LDS object
    const int NS = geo.D + 2;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int N = geo.N;
    const int xc = blockIdx.x & 7, bj = blockIdx.x >> 3;
    const int slab = bj % geo.nslab, chunk = xc + 8 * (bj / geo.nslab);
    if (chunk >= geo.nchunk) return;
    const int xa = chunk * geo.cxlen, xe = min(xa + geo.cxlen, N), len = xe - xa;
    if (len <= 0) return;
    // The march of a chunk is ONE sequence of steps: its len planes with one neighbour plane in front and one behind (no output there).
    const int T = len + 2;                            // steps
    auto plane_of = [&](int t) { return xa - 1 + t; };
    auto is_out = [&](int t) { return t >= 1 && t <= len; };
    const int y0 = slab * geo.RY;
    T_* sdiag = reinterpret_cast<T_*>(lds + (size_t)NS * geo.slot_bytes);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) sdiag[i] = cf.diag[i];
    }
    __syncthreads();   // nothing is in flight yet

    if (wave < geo.NL) {
        // ---------------- loader ----------------
        const char* src[G];
        long pstride[G];
        int ldsoff[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int i = min(wave + j * geo.NL, geo.GTOT - 1);   // surplus instructions repeat the last one (same bytes, same place)
            ldsoff[j] = i * 1024;
            if (i < geo.GS) {
                const int p = min(i * 64 + lane, (geo.RY + 2) * geo.ppr - 1);
                const int row = p / geo.ppr, col = p - row * geo.ppr;
                const int y = min(max(y0 - 1 + row, 0), N - 1);
                src[j] = reinterpret_cast<const char*>(s) + ((long)y * N + col * V) * (long)sizeof(T_);
                pstride[j] = (long)N * N * (long)sizeof(T_);
            } else {
                const int p = min((i - geo.GS) * 64 + lane, (geo.RY + 2) * geo.fpr - 1);
                const int row = p / geo.fpr, col = p - row * geo.fpr;
                const int y = min(max(y0 - 1 + row, 0), N - 1);
                src[j] = reinterpret_cast<const char*>(flags) + ((long)y * N + col * 16);
                pstride[j] = (long)N * N;
            }
        }
        int slot_w = 0;                       // ring slot of the next step to issue (step t lives in slot t % NS)
        int issued = -1;                      // last step issued
        auto issue = [&]() {
            ++issued;
            const int x = min(max(plane_of(issued), 0), N - 1);
            unsigned char* base = lds + (size_t)slot_w * geo.slot_bytes;
#pragma unroll
            for (int j = 0; j < G; ++j)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[j] + (long)x * pstride[j]), (lds_ptr_t)(base + ldsoff[j]), 16, 0, 2);   // aux 2 = nt: every byte is streamed once
            slot_w = slot_w + 1 == NS ? 0 : slot_w + 1;
        };
        issue();   // step 0
        issue();   // step 1
        for (int m = 2; m <= min(geo.D, T - 1); ++m) issue();
        dma_wait_groups<G>(issued - 1);       // steps 0 and 1 have landed
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int t = 1; t <= T - 2; ++t) {
            if (issued < T - 1 && issued < t + geo.D) issue();
            dma_wait_groups<G>(issued - (t + 1));   // step t + 1 has landed
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ---------------- consumer ----------------
    const int cw = wave - geo.NL;
    const int s_bytes = geo.GS * 1024;
    int so[NP], fo[NP];
    unsigned mu[NP], md[NP], ml[NP], mr[NP];
    bool ok[NP];
    long qoff[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int p = (cw * NP + k) * 64 + lane;
        const int pc = min(p, geo.own_pieces - 1);
        const int r = pc / geo.ppr, c = pc - r * geo.ppr;
        const int y = y0 + r;
        ok[k] = p < geo.own_pieces && y < N;
        so[k] = ((r + 1) * geo.ppr + c) * 16;
        fo[k] = s_bytes + (r + 1) * N + c * V;
        mu[k] = y > 0 ? 0xFFFFFFFFu : 0u;
        md[k] = y + 1 < N ? 0xFFFFFFFFu : 0u;
        ml[k] = c > 0 ? 0xFFu : 0u;
        mr[k] = c < geo.ppr - 1 ? 0xFFu : 0u;
        qoff[k] = (long)min(y, N - 1) * N + c * V;
    }
    auto active_bits = [](unsigned w) { return (w >> 1) & ((((w >> 2) & 0x07070707u) + 0x7F7F7F7Fu) >> 7) & 0x01010101u; };
    auto mkv = [&](vec v, unsigned a) {
        vec o;
#pragma unroll
        for (int c = 0; c < V; ++c) o[c] = and_mask<T_>(v[c], __builtin_amdgcn_sbfe((int)a, 8 * c, 1));
        return o;
    };
    auto ldv = [&](const unsigned char* slot, int off) { return *reinterpret_cast<const vec*>(slot + off); };
    auto ldw = [&](const unsigned char* slot, int off) { return (unsigned)*reinterpret_cast<const fvec*>(slot + off); };
    auto pvalid = [&](int x) { return x >= 0 && x < N ? 0xFFFFFFFFu : 0u; };
    const T_ off = cf.off;
    const long plane = (long)N * N;

    vec sm1[NP], s0[NP];
    unsigned w0[NP];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();              // steps 0 and 1 are in slots 0 and 1
    asm volatile("" ::: "memory");
    {
        const unsigned char* sl0 = lds;
        const unsigned char* sl1 = lds + geo.slot_bytes;
        const unsigned pv0 = pvalid(plane_of(0)), pv1 = pvalid(plane_of(1));
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            sm1[k] = mkv(ldv(sl0, so[k]), active_bits(ldw(sl0, fo[k]) & pv0));
            w0[k] = ldw(sl1, fo[k]) & pv1;
            s0[k] = mkv(ldv(sl1, so[k]), active_bits(w0[k]));
        }
    }
    int slot_c = 1;                            // ring slot of step t
    for (int t = 1; t <= T - 2; ++t) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();          // step t + 1 has landed
        asm volatile("" ::: "memory");
        const int slot_n = slot_c + 1 == NS ? 0 : slot_c + 1;
        const unsigned char* sc = lds + (size_t)slot_c * geo.slot_bytes;
        const unsigned char* sn = lds + (size_t)slot_n * geo.slot_bytes;
        const int x = plane_of(t);
        const unsigned nv = pvalid(plane_of(t + 1));
        const bool out_step = is_out(t);       // block-uniform
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const unsigned wn = ldw(sn, fo[k]) & nv;
            const vec sp1 = mkv(ldv(sn, so[k]), active_bits(wn));
            if (out_step) {
                {
                    const vec up = mkv(ldv(sc, so[k] - geo.ppr * 16), active_bits(ldw(sc, fo[k] - N) & mu[k]));
                    const vec dn = mkv(ldv(sc, so[k] + geo.ppr * 16), active_bits(ldw(sc, fo[k] + N) & md[k]));
                    const unsigned fl = (unsigned)sc[fo[k] - 1] & ml[k], fr = (unsigned)sc[fo[k] + V] & mr[k];
                    const T_ left = and_mask<T_>(*reinterpret_cast<const T_*>(sc + so[k] - (int)sizeof(T_)), __builtin_amdgcn_sbfe((int)active_bits(fl), 0, 1));
                    const T_ right = and_mask<T_>(*reinterpret_cast<const T_*>(sc + so[k] + 16), __builtin_amdgcn_sbfe((int)active_bits(fr), 0, 1));
                    const unsigned a0 = active_bits(w0[k]);
                    vec out;
#pragma unroll
                    for (int c = 0; c < V; ++c) {
                        const T_ zl = c ? s0[k][c > 0 ? c - 1 : 0] : left, zr = c < V - 1 ? s0[k][c < V - 1 ? c + 1 : 0] : right;
                        const T_ nb = sm1[k][c] + sp1[c] + up[c] + dn[c] + zl + zr;
                        const T_ r = sdiag[__builtin_amdgcn_ubfe(w0[k], 8 * c + F_CNT_SHIFT, 3)] * s0[k][c] + off * nb;
                        out[c] = and_mask<T_>(r, __builtin_amdgcn_sbfe((int)a0, 8 * c, 1));
                    }
                    if (ok[k]) __builtin_nontemporal_store(out, reinterpret_cast<vec*>(q + (long)x * plane + qoff[k]));
                }
            }
            sm1[k] = s0[k];
            s0[k] = sp1;
            w0[k] = wn;
        }
        slot_c = slot_n;
    }
}

template <typename T, int G, int NP>
static bool dma_launch_t(hipStream_t st, const DmaGeo& geo, const uint8_t* flags, const T* s, T* q, Coef<T> cf)
{
    const size_t ldsb = (size_t)(geo.D + 2) * geo.slot_bytes + 64;
    auto kern = k_stencil_dma<T, G, NP>;
    static size_t have = 0;   // per instantiation
    if (ldsb > have) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return false;
        have = ldsb;
    }
    const int grid = 8 * geo.nslab * ((geo.nchunk + 7) / 8);
    hipLaunchKernelGGL(kern, dim3(grid), dim3((geo.NL + geo.NC) * 64), ldsb, st, geo, flags, s, q, cf);
    return true;
}

// Geometry for an N^3 grid; false when the form does not apply (rows of flag bytes not made of 16-byte pieces, LDS, wave count).
template <typename T>
bool stencil_dma_geo(int N, int RY, int D, int G, int NP, int cxlen, DmaGeo& geo)
{
    constexpr int V = 16 / (int)sizeof(T);
    if (N % 16 || N < 16) return false;
    geo.N = N;
    geo.ppr = N / V;
    geo.fpr = N / 16;
    geo.D = D;
    for (;; RY /= 2) {
        if (RY < 1) return false;
        geo.RY = RY;
        geo.GS = ((RY + 2) * geo.ppr + 63) / 64;
        const int gf = ((RY + 2) * geo.fpr + 63) / 64;
        geo.GTOT = geo.GS + gf;
        geo.slot_bytes = geo.GTOT * 1024;
        geo.own_pieces = RY * geo.ppr;
        geo.NL = (geo.GTOT + G - 1) / G;
        geo.NC = (geo.own_pieces + 64 * NP - 1) / (64 * NP);
        if ((size_t)(D + 2) * geo.slot_bytes + 64 <= 160 * 1024 && geo.NL + geo.NC <= 16) break;
    }
    geo.nslab = (N + geo.RY - 1) / geo.RY;
    // Eight chunks of N / 8 planes would put the eight fronts at EQUAL offsets inside their 2^k-byte regions (256^3 doubles: 16 MiB
    // apart), and aligned fronts cost 8-10 % of the HBM rate (tools/lab/stream_lab.hip, profiles/r04): one plane more per chunk
    // (the last chunk takes what is left) shifts every front by one plane against its neighbour's.
    if (cxlen <= 0) cxlen = (N + 7) / 8 + (N >= 64 ? 1 : 0);
    geo.cxlen = cxlen;
    geo.nchunk = (N + cxlen - 1) / cxlen;
    return true;
}


// ---- dense sweep, x-marching, lean ("lean") -----------------------------------------------------------------------------
// The register-staged march of rounds 2-3 (4.9 TB/s from HBM; kept for the grids the LDS-DMA form does not take: rows of flag
// bytes that are not made of 16-byte pieces).  One row of 64 V cells per wave and plane, x neighbours in registers, y neighbours
// through one double-buffered LDS plane, one barrier per plane, a short instruction stream:
//   * loads are unconditional, from clamped addresses; what lies outside the grid is masked, planes outside the
//     chunk's reach are skipped by ONE wave-uniform branch;
//   * the "unknown" predicate of the V flag bytes of a lane is formed on the packed word (SWAR) and expanded to one
//     all-ones / zero word per cell (v_bfe_i32); values are masked by AND, results too: no per-cell branch;
//   * the rim rows (y0 - 1, y0 + MY) belong to two EXTRA waves that only load, mask and publish them, so no wave of the
//     block carries two rows to the barrier;
//   * z neighbours across lanes by DPP wave shifts (zero fill at the wave's ends = the grid's edge when a row is one
//     wave's width; else lanes 0 / 63 load the two rim cells).
// Same term order as the other forms: bit-identical results.
template <typename T>
__device__ __forceinline__ T dpp_wave_shr1(T v)   // lane i <- lane i-1, lane 0 <- 0
{
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
    } else {
        const long long b = __builtin_bit_cast(long long, v);
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x138, 0xf, 0xf, true);
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x138, 0xf, 0xf, true);
        return __builtin_bit_cast(T, ((long long)hi << 32) | lo);
    }
}
template <typename T>
__device__ __forceinline__ T dpp_wave_shl1(T v)   // lane i <- lane i+1, lane 63 <- 0
{
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
    } else {
        const long long b = __builtin_bit_cast(long long, v);
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x130, 0xf, 0xf, true);
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x130, 0xf, 0xf, true);
        return __builtin_bit_cast(T, ((long long)hi << 32) | lo);
    }
}

template <typename T, int MY, int MD, bool RIMS = false>
__global__ __launch_bounds__((MY + 2) * 64) void k_stencil_lean(Grid g, int cxlen, int nty, int ntz, const uint8_t* __restrict__ flags,
                                                                const T* __restrict__ s, T* __restrict__ q, Coef<T> cf)
{
    constexpr int V = 16 / (int)sizeof(T), MZV = 64 * V;
    typedef typename VecT<T, V>::type vec;
    typedef typename FlagT<V>::type fvec;
    __shared__ __attribute__((aligned(16))) T pl[2][MY + 2][MZV];
    __shared__ T sdiag[8], sinv[8];
    const int tid = threadIdx.x, wy = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    load_coef(sdiag, sinv, cf);
    const int N = g.N;
    const long sm = (long)N * N;   // stride of the march axis
    const long sr = (long)N;       // stride of the row axis (the waves of a block)
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int tz = vb % ntz, ty = (vb / ntz) % nty, cx = vb / (ntz * nty);
    const int xa = cx * cxlen, xe = xa + cxlen < N ? xa + cxlen : N;
    const int xlast = xe < N ? xe : N - 1;          // last march index anybody of this chunk reads
    const bool own = wy < MY;                       // waves MY, MY+1: the rim rows
    const int y = own ? ty * MY + wy : (wy == MY ? ty * MY - 1 : ty * MY + MY);
    const int lrow = own ? wy + 1 : (wy == MY ? 0 : MY + 1);
    const int z0 = tz * MZV + lane * V;
    const bool cv = y >= 0 && y < N && z0 < N;      // N % V == 0: a lane's cells are all inside or all outside
    const unsigned cvm = cv ? 0xFFFFFFFFu : 0u;
    const long col = (long)min(max(y, 0), N - 1) * sr + min(z0, N - V);
    // z rim (only when a row is wider than one wave): lane 0 the cell before the wave's range, lane 63 the cell after it
    const int rz = lane < 32 ? tz * MZV - 1 : tz * MZV + MZV;
    const bool rv = own && (lane == 0 || lane == 63) && y < N && rz >= 0 && rz < N;
    const long rcol = (long)min(max(y, 0), N - 1) * sr + min(max(rz, 0), N - 1);
    // per-cell masks (all ones / zero) of a packed flag word: unknown = fluid and at least one non-solid neighbour
    auto active_bits = [](unsigned w) { return (w >> 1) & ((((w >> 2) & 0x07070707u) + 0x7F7F7F7Fu) >> 7) & 0x01010101u; };
    auto mkv = [&](vec v, unsigned a) {
        vec o;
#pragma unroll
        for (int c = 0; c < V; ++c) o[c] = and_mask<T>(v[c], __builtin_amdgcn_sbfe((int)a, 8 * c, 1));
        return o;
    };
    // Loads are unconditional and their results are masked only where they are consumed: a select right behind a load (or a load
    // inside a branch) makes the compiler wait for it at once, and the MD planes in flight collapse to one.
    auto ldp = [&](int x, vec& v, unsigned& w) {   // one step of this wave's row: x is wave-uniform
        const long c = (long)min(max(x, 0), xlast) * sm + col;
        w = (unsigned)*reinterpret_cast<const fvec*>(flags + c);
        v = *reinterpret_cast<const vec*>(s + c);
    };
    auto okw = [&](int x, unsigned w) { return x >= 0 && x <= xlast ? w & cvm : 0u; };
    auto ldr = [&](int x, T& v, unsigned& f) {     // the rim cell of lanes 0 / 63 (every lane loads: no branch)
        v = 0;
        f = 0;
        if constexpr (RIMS) {
            const long c = (long)min(max(x, 0), xlast) * sm + rcol;
            f = flags[c];
            v = s[c];
        }
    };
    auto okr = [&](int x, unsigned f) { return rv && x >= 0 && x <= xlast ? f : 0u; };
    vec vm1, v0;
    unsigned wm1, w0, rf0;
    T r0;
    ldp(xa - 1, vm1, wm1);
    ldp(xa, v0, w0);
    ldr(xa, r0, rf0);
    vec qv[MD];
    unsigned qw[MD], rqf[MD];
    T rq[MD];
#pragma unroll
    for (int d = 0; d < MD; ++d) {
        ldp(xa + 1 + d, qv[d], qw[d]);
        ldr(xa + 1 + d, rq[d], rqf[d]);
    }
    w0 = okw(xa, w0);
    vec sm1 = mkv(vm1, active_bits(okw(xa - 1, wm1))), s0 = mkv(v0, active_bits(w0));
    T rim = and_mask<T>(r0, __builtin_amdgcn_sbfe((int)active_bits(okr(xa, rf0)), 0, 1));
    const T off = cf.off;
    __syncthreads();  // coef table
    // one step of the march; d = the slot of the ring that holds step x+1 and is refilled with step x+1+MD
    auto step = [&](int x, vec& nv, unsigned& nw, T& nr, unsigned& nrf) {
        const int buf = x & 1;
        const unsigned wn = okw(x + 1, nw);
        *reinterpret_cast<vec*>(&pl[buf][lrow][lane * V]) = s0;
        __syncthreads();
        const unsigned an = active_bits(wn);
        const vec sp1 = mkv(nv, an);
        if (own) {   // wave-uniform
            const vec up = *reinterpret_cast<const vec*>(&pl[buf][lrow - 1][lane * V]);
            const vec dn = *reinterpret_cast<const vec*>(&pl[buf][lrow + 1][lane * V]);
            T left = dpp_wave_shr1<T>(s0[V - 1]), right = dpp_wave_shl1<T>(s0[0]);
            if (RIMS) {
                left = lane == 0 ? rim : left;
                right = lane == 63 ? rim : right;
            }
            const unsigned a0 = active_bits(w0);
            vec out;
#pragma unroll
            for (int c = 0; c < V; ++c) {
                const T zl = c ? s0[c > 0 ? c - 1 : 0] : left, zr = c < V - 1 ? s0[c < V - 1 ? c + 1 : 0] : right;
                const T nb = sm1[c] + sp1[c] + up[c] + dn[c] + zl + zr;
                const T r = sdiag[__builtin_amdgcn_ubfe(w0, 8 * c + F_CNT_SHIFT, 3)] * s0[c] + off * nb;
                out[c] = and_mask<T>(r, __builtin_amdgcn_sbfe((int)a0, 8 * c, 1));
            }
            if (cv) __builtin_nontemporal_store(out, reinterpret_cast<vec*>(&q[(long)x * sm + col]));  // streamed once: keep s, not q, in cache
        }
        sm1 = s0;
        s0 = sp1;
        w0 = wn;
        rim = and_mask<T>(nr, __builtin_amdgcn_sbfe((int)active_bits(okr(x + 1, nrf)), 0, 1));
        ldp(x + 1 + MD, nv, nw);
        ldr(x + 1 + MD, nr, nrf);
    };
    // whole rounds of the ring run unguarded: a guard inside the round is a path on which the newest load is the next one
    // consumed, and the compiler then drains the queue at every step
    int xb = xa;
    for (; xb + MD <= xe; xb += MD) {
#pragma unroll
        for (int d = 0; d < MD; ++d) step(xb + d, qv[d], qw[d], rq[d], rqf[d]);
    }
#pragma unroll
    for (int d = 0; d < MD; ++d)
        if (xb + d < xe) step(xb + d, qv[d], qw[d], rq[d], rqf[d]);   // block-uniform
}

template <typename T, int MY, int MD>
static void lean_launch(hipStream_t st, Grid g, int cxlen, const uint8_t* flags, const T* s, T* q, Coef<T> cf)
{
    constexpr int V = 16 / (int)sizeof(T), MZV = 64 * V;
    const int nty = (g.N + MY - 1) / MY, ntz = (g.N + MZV - 1) / MZV, ncx = (g.N + cxlen - 1) / cxlen;
    if (ntz > 1)
        hipLaunchKernelGGL((k_stencil_lean<T, MY, MD, true>), dim3(nty * ntz * ncx), dim3((MY + 2) * 64), 0, st, g, cxlen, nty, ntz, flags, s, q, cf);
    else
        hipLaunchKernelGGL((k_stencil_lean<T, MY, MD, false>), dim3(nty * ntz * ncx), dim3((MY + 2) * 64), 0, st, g, cxlen, nty, ntz, flags, s, q, cf);
}


// ---- streaming probes (developer: FLUID_MARCH_VARIANT=20001 / 20002): what a plain copy of the same arrays reaches ----
// 20001: q = s, 16 bytes per lane, non-temporal stores; 20002: q = active(flag) ? s : 0 (adds the 1-byte-per-cell flag stream).
// They bound what the stencil sweep can reach from HBM on this part (tools/sweep.py ... hbm).
template <typename T, bool FLAGS>
__global__ __launch_bounds__(256) void k_stream_probe(long n16, const uint8_t* __restrict__ flags, const T* __restrict__ s, T* __restrict__ q)
{
    constexpr int V = 16 / (int)sizeof(T);
    typedef typename VecT<T, V>::type vec;
    typedef typename FlagT<V>::type fvec;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        vec v = reinterpret_cast<const vec*>(s)[i];
        if (FLAGS) {
            const fvec f = reinterpret_cast<const fvec*>(flags)[i];
#pragma unroll
            for (int c = 0; c < V; ++c) {
                const uint8_t fc = (uint8_t)(f >> (8 * c));
                v[c] = ((fc & F_FLUID) && (fc >> F_CNT_SHIFT)) ? v[c] : (T)0;
            }
        }
        __builtin_nontemporal_store(v, reinterpret_cast<vec*>(q) + i);
    }
}


// variant (FLUID_MARCH_VARIANT, developer / tests): 0 = the default: the LDS-DMA plane ring where it applies (N a multiple of 16), else
// the lean march; 40000 + MY*100 + MD = the lean march, cxcode planes per chunk; 20001 / 20002 = the copy probes;
// 900000 + D*10000 + RY = the LDS-DMA form with D planes in flight, RY rows per block and cxcode planes per chunk (0: the defaults).
// false = nothing launched (rows not 16-byte aligned: the caller takes the tiled kernel).
template <typename T>
bool launch_stencil_march(hipStream_t st, Grid g, const uint8_t* flags, const T* s, T* q, Coef<T> cf, int variant, int cxcode)
{
    constexpr int V = 16 / (int)sizeof(T);
    if (g.N % V || ((uintptr_t)s & 15) || ((uintptr_t)q & 15)) return false;
    if (variant == 20001 || variant == 20002) {
        const long n16 = (long)g.cells() / V;
        const int nb = cxcode > 0 ? cxcode * 256 : 1024;
        if (variant == 20001) hipLaunchKernelGGL((k_stream_probe<T, false>), dim3(nb), dim3(256), 0, st, n16, flags, s, q);
        else hipLaunchKernelGGL((k_stream_probe<T, true>), dim3(nb), dim3(256), 0, st, n16, flags, s, q);
        return true;
    }
    if (variant >= 40000 && variant < 50000) {
        const int cx = cxcode > 0 ? cxcode : 32;
        switch (variant - 40000) {
        case 402: lean_launch<T, 4, 2>(st, g, cx, flags, s, q, cf); break;
        case 804: lean_launch<T, 8, 4>(st, g, cx, flags, s, q, cf); break;
        case 1402: lean_launch<T, 14, 2>(st, g, cx, flags, s, q, cf); break;
        default: lean_launch<T, 4, 4>(st, g, cx, flags, s, q, cf); break;
        }
        return true;
    }
    if (variant == 0 && g.N < 192) {
        // small grids (the metric's 128^3: 19 / 36 MB per set, a launch of ~10 us): 8 chunks of N / 8 planes of the lean march fill the
        // chip with short blocks — 10.4 us (fp64) / 8.9 us (fp32) at 128^3 against 20.8 / 12.4 for the tiled kernel and 16-17 / 11-12 for
        // the plane ring, whose blocks march too few planes there to pay for their prologue (profiles/r04/NOTES.md)
        lean_launch<T, 4, 4>(st, g, g.N / 8 > 8 ? g.N / 8 : 8, flags, s, q, cf);
        return true;
    }
    // the LDS-DMA plane ring: measured defaults at 256^3 from HBM (tools/dma_lab.py, profiles/r04/stencil_sweep_hbm.txt):
    // double: 8 loader waves (3 instructions each per plane) + 8 consumer waves with two pieces per lane; float: 4 + 8 with one
    // (planes in flight: 2 for double, 3 for float — 0.699 / 0.697 of the 8 TB/s peak; 3 / 2: 0.693 / 0.679)
    int D = sizeof(T) == 4 ? 3 : 2, RY = 8, cxlen = 0;
    if (variant > 900000) { D = (variant - 900000) / 10000; RY = (variant - 900000) % 100; cxlen = cxcode; }
    if (D < 2 || D > 6 || RY < 1) { D = sizeof(T) == 4 ? 3 : 2; RY = 8; }
    DmaGeo geo;
    constexpr int G = sizeof(T) == 4 ? 4 : 3, NP = sizeof(T) == 4 ? 1 : 2;
    if (((uintptr_t)flags & 15) == 0 && g.nx == g.N && g.ny == g.N && g.nz == g.N && stencil_dma_geo<T>(g.N, RY, D, G, NP, cxlen, geo) &&
        dma_launch_t<T, G, NP>(st, geo, flags, s, q, cf))
        return true;
    // measured from HBM at 256^3 (round 3): 4 planes in flight, chunks of 32 (fp32) / 64 (fp64) planes
    lean_launch<T, 4, 4>(st, g, (sizeof(T) == 4 ? 32 : 64) >> (g.N < 256 ? 1 : 0), flags, s, q, cf);
    return true;
}

template bool launch_stencil_march<double>(hipStream_t, Grid, const uint8_t*, const double*, double*, Coef<double>, int, int);
template bool launch_stencil_march<float>(hipStream_t, Grid, const uint8_t*, const float*, float*, Coef<float>, int, int);

}  // namespace fl
