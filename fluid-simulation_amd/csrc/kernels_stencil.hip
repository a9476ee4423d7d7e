// Dense sweep q = A s from HBM through an LDS-DMA plane ring (gfx950 only).
//
// A = the reference's pressure matrix (setA, fluid.cc:304-412): a function of the per-cell flag byte, applied matrix-free.
// This is the bandwidth-bound kernel of the path ((2T+1) bytes per cell, x N^3) — the "pressure-stencil kernel
// at 256^3" of the north star.  Every register-staged march of rounds 1-3 stopped at ~4.9 TB/s; this form moves the operands
// with the CDNA4 transport instead (`global_load_lds_dwordx4`: HBM -> LDS with no VGPR in between):
//
//   * the grid is cut into x chunks, one per XCD (blocks b and b + 8 share an XCD), and every plane of a chunk into slabs of
//     RY whole z rows, one per workgroup: the 32 CUs of an XCD together fetch WHOLE planes, so each XCD advances one linear
//     front through s, one through the flags and one through q; neighbouring chunks march in opposite directions, so the
//     two planes they share are fetched by both at about the same time (the second fetch finds them in the Infinity Cache);
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
    int rot;                  // planes of rotation per chunk (0: every chunk starts at its first plane)
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

// MODE 0: the stencil; 1 (probe): q = masked s through the same transport (no neighbour reads, no arithmetic)
template <typename T_, int G, int NP, bool NT, int MODE>
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
    // The chunk's planes are marched in ROTATED order — xa + rot .. xe - 1, then xa .. xa + rot - 1, rot = geo.rot * chunk mod len —
    // so that the fronts of the eight XCDs do not sit at equal offsets inside their 2^k-byte regions (measured: aligned fronts
    // cost 8-10 % of the HBM rate, tools/lab/stream_lab.hip).  The march is ONE sequence of steps: [xa+rot-1, xa+rot .. xe-1, xe]
    // and, if rot > 0, [xa-1, xa .. xa+rot-1, xa+rot]; the first and last plane of each segment are only neighbours (no output).
    const int rot = (int)(((long)geo.rot * chunk) % len);
    const int lenA = len - rot;                       // output planes of the first segment
    const int T = rot ? len + 4 : len + 2;            // steps
    auto plane_of = [&](int t) { return t <= lenA + 1 ? xa + rot - 1 + t : xa - 1 + (t - lenA - 2); };
    auto is_out = [&](int t) { return t <= lenA + 1 ? (t >= 1 && t <= lenA) : (t >= lenA + 3 && t <= T - 2); };
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
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[j] + (long)x * pstride[j]), (lds_ptr_t)(base + ldsoff[j]), 16, 0, NT ? 2 : 0);
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
                if constexpr (MODE == 1) {
                    if (ok[k]) __builtin_nontemporal_store(s0[k], reinterpret_cast<vec*>(q + (long)x * plane + qoff[k]));
                } else {
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

template <typename T, int G, int NP, bool NT, int MODE>
static bool dma_launch_t(hipStream_t st, const DmaGeo& geo, const uint8_t* flags, const T* s, T* q, Coef<T> cf)
{
    const size_t ldsb = (size_t)(geo.D + 2) * geo.slot_bytes + 64;
    auto kern = k_stencil_dma<T, G, NP, NT, MODE>;
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
bool stencil_dma_geo(int N, int RY, int D, int G, int NP, int cxlen, int rot, DmaGeo& geo)
{
    constexpr int V = 16 / (int)sizeof(T);
    if (N % 16 || N < 16) return false;
    geo.N = N;
    geo.ppr = N / V;
    geo.fpr = N / 16;
    geo.D = D;
    geo.rot = rot;
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
    if (cxlen <= 0) cxlen = (N + 7) / 8;
    geo.cxlen = cxlen;
    geo.nchunk = (N + cxlen - 1) / cxlen;
    return true;
}

// variant = D * 10000 + G * 1000 + NP * 100 + RY (0 = the default); cxcode = cxlen + 1000 * nt + 10000 * rot + 100000 * mode
template <typename T>
bool launch_stencil_dma(hipStream_t st, Grid g, const uint8_t* flags, const T* s, T* q, Coef<T> cf, int variant, int cxcode)
{
    if (g.nx != g.N || g.ny != g.N || g.nz != g.N) return false;
    if (((uintptr_t)s & 15) || ((uintptr_t)q & 15) || ((uintptr_t)flags & 15)) return false;
    int D = 3, G = 6, NP = 2, RY = 8, cxlen = 0, nt = 1, rot = 3, mode = 0;
    if (variant > 0) {
        D = variant / 10000; G = (variant / 1000) % 10; NP = (variant / 100) % 10; RY = variant % 100;
        cxlen = cxcode % 1000; nt = (cxcode / 1000) % 10; rot = (cxcode / 10000) % 10; mode = (cxcode / 100000) % 10;
    }
    if (D < 2 || D > 6) return false;
    DmaGeo geo;
    if (!stencil_dma_geo<T>(g.N, RY, D, G, NP, cxlen, rot, geo)) return false;
#define DMA_CASE(GG, PP)                                                                                                     \
    if (G == GG && NP == PP) {                                                                                               \
        if (mode == 1) return nt ? dma_launch_t<T, GG, PP, true, 1>(st, geo, flags, s, q, cf) : dma_launch_t<T, GG, PP, false, 1>(st, geo, flags, s, q, cf); \
        return nt ? dma_launch_t<T, GG, PP, true, 0>(st, geo, flags, s, q, cf) : dma_launch_t<T, GG, PP, false, 0>(st, geo, flags, s, q, cf);               \
    }
    DMA_CASE(6, 2)
    DMA_CASE(3, 2)
    DMA_CASE(4, 2)
    DMA_CASE(2, 2)
    DMA_CASE(3, 1)
    DMA_CASE(2, 1)
    DMA_CASE(4, 1)
#undef DMA_CASE
    return false;
}

template bool launch_stencil_dma<double>(hipStream_t, Grid, const uint8_t*, const double*, double*, Coef<double>, int, int);
template bool launch_stencil_dma<float>(hipStream_t, Grid, const uint8_t*, const float*, float*, Coef<float>, int, int);

}  // namespace fl
