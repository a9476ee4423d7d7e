// Developer lab (not part of the library): which ACCESS PATTERN costs the marching stencil its last 10 %?
// Copies an N^3 double array + its flag bytes (q = flag ? s : 0, 17 bytes per cell like the stencil) from HBM with different
// mappings of (block, step, lane) -> address, launches rotating over > 1 GiB of separate sets.
//   hipcc --offload-arch=gfx950 -O3 -o stream_lab stream_lab.hip && ./stream_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void cp(const d2* s, const unsigned short* f, d2* q, long i)
{
    d2 v = s[i];
    const unsigned short w = f[i];
    v[0] = (w & 0xff) ? v[0] : 0.0;
    v[1] = (w >> 8) ? v[1] : 0.0;
    __builtin_nontemporal_store(v, q + i);
}
// P0: linear grid-stride
__global__ __launch_bounds__(256) void p0(long n16, const d2* s, const unsigned short* f, d2* q)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) cp(s, f, q, i);
}
// P1: each XCD (block % 8) copies one contiguous eighth, grid-stride inside it; skew = extra pieces between the starting points (rotation inside the region)
__global__ __launch_bounds__(256) void p1(long n16, const d2* s, const unsigned short* f, d2* q, long skew)
{
    const long R = n16 / 8, xcd = blockIdx.x & 7, j = blockIdx.x >> 3, nb = gridDim.x >> 3;
    const long rot = (xcd * skew) % R;
    for (long i = j * 256 + threadIdx.x; i < R; i += nb * 256) {
        long k = i + rot;
        if (k >= R) k -= R;
        cp(s, f, q, xcd * R + k);
    }
}
// P2: the march: block = (xcd = chunk, slab j); per plane a slab of `rows` rows of N cells; planes of the chunk one after another
// threads: 256; a slab has rows * N / 2 pieces.  rotp = planes of rotation per chunk (fronts out of phase)
__global__ __launch_bounds__(1024) void p2(int N, int rows, int cxlen, const d2* s, const unsigned short* f, d2* q, int rotp, int nchunk)
{
    const int xcd = blockIdx.x & 7, bj = blockIdx.x >> 3, nslab = N / rows;
    const int slab = bj % nslab, chunk = xcd + 8 * (bj / nslab);
    if (chunk >= nchunk) return;
    const long ppp = (long)N * N / 2;            // pieces per plane
    const int sp = rows * N / 2;                 // pieces per slab
    for (int m = 0; m < cxlen; ++m) {
        int mm = m + rotp * chunk;
        mm %= cxlen;
        const int x = chunk * cxlen + mm;
        if (x >= N) continue;
        const long base = (long)x * ppp + (long)slab * sp;
        for (int t = threadIdx.x; t < sp; t += blockDim.x) cp(s, f, q, base + t);
    }
}
// P3: one front: every block takes `rows` rows of EVERY plane (blocks = N / rows), planes in order
__global__ __launch_bounds__(1024) void p3(int N, int rows, const d2* s, const unsigned short* f, d2* q)
{
    const long ppp = (long)N * N / 2;
    const int sp = rows * N / 2;
    for (int x = 0; x < N; ++x) {
        const long base = (long)x * ppp + (long)blockIdx.x * sp;
        for (int t = threadIdx.x; t < sp; t += blockDim.x) cp(s, f, q, base + t);
    }
}
// P4: the march with chunks interleaved at a coarser granularity: chunk c owns planes {c*g .. c*g+g-1} + k*8*g (g planes every 8g)
__global__ __launch_bounds__(1024) void p4(int N, int rows, int g, const d2* s, const unsigned short* f, d2* q)
{
    const int xcd = blockIdx.x & 7, slab = blockIdx.x >> 3;
    const long ppp = (long)N * N / 2;
    const int sp = rows * N / 2;
    for (int x0 = xcd * g; x0 < N; x0 += 8 * g)
        for (int x = x0; x < x0 + g && x < N; ++x) {
            const long base = (long)x * ppp + (long)slab * sp;
            for (int t = threadIdx.x; t < sp; t += blockDim.x) cp(s, f, q, base + t);
        }
}

int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 256;
    const long n = (long)N * N * N, n16 = n / 2;
    const size_t setb = n * 17;
    const int nsets = (int)(((size_t)3 << 29) / setb) + 2;   // > 1.5 GiB
    std::vector<d2*> S(nsets), Q(nsets);
    std::vector<unsigned short*> F(nsets);
    for (int i = 0; i < nsets; ++i) {
        CK(hipMalloc(&S[i], n * 8)); CK(hipMalloc(&Q[i], n * 8)); CK(hipMalloc(&F[i], n));
        CK(hipMemset(S[i], 1, n * 8)); CK(hipMemset(Q[i], 0, n * 8)); CK(hipMemset(F[i], 1, n));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < nsets; ++i) launch(i);
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 2 * nsets; ++i) launch(i % nsets);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 2 * nsets;
            if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        printf("%-44s %7.1f us  %5.0f GB/s  %.3f\n", name, best * 1e3, setb / best / 1e6, setb / best / 1e6 / 8000);
        fflush(stdout);
    };
    char nm[128];
    for (int nb : {512, 1024, 2048, 4096, 8192}) {
        snprintf(nm, sizeof nm, "P0 linear, %d blocks", nb);
        timeit(nm, [&](int i) { hipLaunchKernelGGL(p0, dim3(nb), dim3(256), 0, 0, n16, S[i], F[i], Q[i]); });
    }
    for (int nb : {1024, 2048, 4096})
        for (long skew : {0L, 4096L + 64, 65536L * 3 + 192}) {
            snprintf(nm, sizeof nm, "P1 XCD eighths, %d blocks, skew %ld", nb, skew);
            timeit(nm, [&](int i) { hipLaunchKernelGGL(p1, dim3(nb), dim3(256), 0, 0, n16, S[i], F[i], Q[i], skew); });
        }
    for (int rows : {4, 8})
        for (int thr : {512, 1024})
            for (int rot : {0, 1, 2, 3, 5, 7}) {
                snprintf(nm, sizeof nm, "P2 march 8 chunks, %d rows, %d thr, rot %d", rows, thr, rot);
                timeit(nm, [&](int i) { hipLaunchKernelGGL(p2, dim3(8 * (N / rows)), dim3(thr), 0, 0, N, rows, N / 8, S[i], F[i], Q[i], rot, 8); });
            }
    for (int g : {1, 2, 4, 8})
        for (int rows : {4, 8}) {
            snprintf(nm, sizeof nm, "P4 cyclic groups of %d planes, %d rows, 512 thr", g, rows);
            timeit(nm, [&](int i) { hipLaunchKernelGGL(p4, dim3(8 * (N / rows)), dim3(512), 0, 0, N, rows, g, S[i], F[i], Q[i]); });
        }
    return 0;
}
