// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// Plain C++ CPU restatement of ONE PIC/FLIP step of the reference program
// (/root/reference/fluid.cc:1368-1507 and the functions it calls).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
// only as the checker / the timed CPU baseline — never as the thing shipped.
//
// PINNING STATUS
//   * pressure solve (fluid.cc:1473-1474, Eigen ConjugateGradient<..,IncompleteCholesky>):
//     PINNED — oracle/_ref/libeigen_ref.so is built from the reference's own vendored
//     Eigen 3.3.4 headers where they lie (oracle/Makefile); tests compare this file's
//     CG against it and against the known-answer vectors restated from
//     openvdb/unittest/TestConjGradient.cc:58-105 (tests/golden/).
//   * every other function (spline, P2G, flags/index, RHS, divergence, matrix
//     coefficients, velocity update, FLIP gather/advect): PARITY UNPINNED.  fluid.cc needs
//     libopenvdb 4.0.2 + TBB + Boost + IlmBase Half, none of which exist in this image
//     (.MISSING_LARGE_BLOBS:5,7), so the reference cannot be built and it has no tests
//     or golden files of its own for this path.  These functions are line-by-line
//     restatements; each cites the reference lines it follows.
//
// Semantics kept from the reference's storage layer (openvdb::Grid accessors):
//   * FloatGrid stores float32: every setValue(c, <double expr>) narrows to float.
//   * Vec3dGrid stores 3 doubles; Vec3d ops are component-wise, true division
//     (openvdb/math/Vec3.h:224-230,265-291,499-534).
//   * reads outside the filled box [lo,hi]^3 return the grid's background (0).
//   * Coord(int,int,int) built from doubles truncates toward zero (math/Coord.h:59-61).
// Generalisation: the reference's literals B=60 (cells -60..60) and
// W=58 become lo..hi and lo+2..hi-2, with lo=-(N/2), hi=lo+N-1 (N=121 -> -60..60).
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off; no -ffast-math).

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <atomic>
#include <thread>

namespace {

// fluid.cc:22-37
inline double spline(double x)
{
    if (x < 0) x *= -1.0;
    if (x < 0.5) return 1.5 * (4.0 * x * x * x - 4.0 * x * x + 2.0 / 3.0);
    if (x < 1.0) return 1.5 * ((-8.0 * (x * x * x) / 6.0) + 4.0 * x * x - 4.0 * x + 4.0 / 3.0);
    return 0;
}

typedef void (*ref_solver_fn)(int n, int nnz, const int* rows, const int* cols, const double* vals,
                              const double* b, double* x, int* iters, double* err);

// nthreads > 1: the particle loops run on host threads the way the reference runs them under tbb::parallel_for
// (fluid.cc:845,978,1126), the scatters under a lock per cell like its std::mutex cube (fluid.cc:290-294,872-874; striped here).
// The float32 accumulations then depend on the thread schedule exactly as they do in the reference (SURVEY 8c "determinism
// caveat"); nthreads = 1 is the serial, bitwise reproducible order every parity test uses.
template <typename F>
static void parallel_for(int nthreads, size_t n, F fn)
{
    if (nthreads <= 1 || n < 4096) { fn((size_t)0, n, 0); return; }
    std::vector<std::thread> th;
    const size_t per = (n + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; ++t) {
        const size_t a = std::min(n, per * t), b = std::min(n, per * (t + 1));
        if (a < b) th.emplace_back([=] { fn(a, b, t); });
    }
    for (auto& x : th) x.join();
}
struct CellLocks {
    static constexpr size_t N = (size_t)1 << 16;
    std::atomic<unsigned char> l[N];
    CellLocks() { for (auto& x : l) x.store(0); }
    void lock(size_t k) { auto& a = l[k & (N - 1)]; while (a.exchange(1, std::memory_order_acquire)) {} }
    void unlock(size_t k) { l[k & (N - 1)].store(0, std::memory_order_release); }
};
static CellLocks g_locks;

struct Oracle {
    int N, lo, hi, wlo, whi;
    double dx, rho, g[3], max_dt, outer_tol, update_frac;
    double dt;              // fluid.cc:1367 — carried from step to step
    double cg_tol;          // Eigen default: NumTraits<double>::epsilon() (IterativeSolverBase.h:283)
    double flip_blend = 1.0;  // 1 = pure FLIP (fluid.cc:981); < 1 blends in the PIC gather of the unused clampedCatmullRom (:125-207)
    int nthreads;           // 1 = serial (deterministic order, like a serial tbb::parallel_for)
    ref_solver_fn ref_solver;

    size_t ncell;
    std::vector<float> solid, container, output, weights, Adiag, Aplusi, Aplusj, Aplusk, diver, rhs;
    std::vector<int32_t> indices;
    std::vector<double> vel, velBefore;  // AoS xyz per cell, like Vec3dGrid
    std::vector<double> pressure;        // dense field p[index(c)] scattered back to cells
    std::vector<double> ppos, pvel;      // AoS xyz per particle, like vector<Vec3d>
    std::vector<double> b, b2, p;
    int numActive;
    // stats of the last step
    int outer_passes, cg_iters_total, cg_iters_last;
    double error, maxSpeed, relres_last;

    inline bool inRange(int x, int y, int z) const
    {
        return x >= lo && x <= hi && y >= lo && y <= hi && z >= lo && z <= hi;
    }
    inline size_t idx(int x, int y, int z) const
    {
        return ((size_t)(x - lo) * N + (size_t)(y - lo)) * N + (size_t)(z - lo);
    }
    // accessor.getValue with background 0 outside the box
    inline float getF(const std::vector<float>& g_, int x, int y, int z) const
    {
        return inRange(x, y, z) ? g_[idx(x, y, z)] : 0.0f;
    }
    inline double getV(const std::vector<double>& v_, int x, int y, int z, int a) const
    {
        return inRange(x, y, z) ? v_[3 * idx(x, y, z) + a] : 0.0;
    }
    // fluid.cc:38-45 with bound=60 -> [lo,hi]; bound=58 -> [wlo,whi]
    inline bool withinB(int x, int y, int z) const { return inRange(x, y, z); }
    inline bool withinW(int x, int y, int z) const
    {
        return x >= wlo && x <= whi && y >= wlo && y <= whi && z >= wlo && z <= whi;
    }
    // fluid.cc:46-57
    inline bool isSolid(int x, int y, int z) const { return getF(solid, x, y, z) == 1; }
};

// fluid.cc:59-70
inline void getVelocity(const Oracle& o, const std::vector<double>& v, int x, int y, int z, double out[3])
{
    out[0] = (o.getV(v, x, y, z, 0) + o.getV(v, x + 1, y, z, 0)) / 2.0;
    out[1] = (o.getV(v, x, y, z, 1) + o.getV(v, x, y + 1, z, 1)) / 2.0;
    out[2] = (o.getV(v, x, y, z, 2) + o.getV(v, x, y, z + 1, 2)) / 2.0;
}

inline void support(const Oracle& o, double c, int& mn, int& mx)
{
    // fluid.cc:267-276 (same in 212-221, 848-857)
    int fc = (int)round(c);
    mn = fc - 1 > o.lo ? fc - 1 : o.lo;
    mx = fc + 1 < o.hi ? fc + 1 : o.hi;
}

// fluid.cc:1106-1148 (P2Gtransfer) + 265-299 (p2gCatmullRom); serial particle order.
void p2g(Oracle& o)
{
    std::fill(o.vel.begin(), o.vel.end(), 0.0);          // fluid.cc:1378
    std::fill(o.weights.begin(), o.weights.end(), 0.0f);  // fluid.cc:1108
    const size_t np = o.ppos.size() / 3;
    const bool mt = o.nthreads > 1;
    parallel_for(o.nthreads, np, [&](size_t i0, size_t i1, int) {
    for (size_t i = i0; i < i1; ++i) {
        const double cx = o.ppos[3 * i], cy = o.ppos[3 * i + 1], cz = o.ppos[3 * i + 2];
        const double* vc = &o.pvel[3 * i];
        int minx, maxx, miny, maxy, minz, maxz;
        support(o, cx, minx, maxx);
        support(o, cy, miny, maxy);
        support(o, cz, minz, maxz);
        for (int x = minx; x <= maxx; ++x)
            for (int y = miny; y <= maxy; ++y)
                for (int z = minz; z <= maxz; ++z) {
                    if (!o.isSolid(x, y, z) && o.withinW(x, y, z)) {  // :288 (bound-2)
                        double cw = spline(cx - x) * spline(cy - y) * spline(cz - z);  // :291
                        size_t k = o.idx(x, y, z);
                        if (mt) g_locks.lock(k);                     // :290 lockGrid[...]->lock()
                        o.weights[k] = (float)(o.weights[k] + cw);   // :292 FloatGrid narrowing
                        o.vel[3 * k + 0] = o.vel[3 * k + 0] + cw * vc[0];  // :293
                        o.vel[3 * k + 1] = o.vel[3 * k + 1] + cw * vc[1];
                        o.vel[3 * k + 2] = o.vel[3 * k + 2] + cw * vc[2];
                        if (mt) g_locks.unlock(k);
                    }
                }
    }
    });
    // fluid.cc:1130-1146
    parallel_for(o.nthreads, o.ncell, [&](size_t k0, size_t k1, int) {
    for (size_t k = k0; k < k1; ++k) {
        double w = o.weights[k];
        if (w > 0) {
            o.vel[3 * k + 0] /= w;
            o.vel[3 * k + 1] /= w;
            o.vel[3 * k + 2] /= w;
        }
    }
    });
}

// fluid.cc:1388-1455: indices=-1, container=0, PointList::interpolate (843-882),
// index sweep (1416-1433), output copy (1434-1448), velBeforeUpdate (1455).
void flags_index(Oracle& o)
{
    std::fill(o.indices.begin(), o.indices.end(), -1);
    std::fill(o.container.begin(), o.container.end(), 0.0f);
    const size_t np = o.ppos.size() / 3;
    const bool mt = o.nthreads > 1;
    parallel_for(o.nthreads, np, [&](size_t i0, size_t i1, int) {
    for (size_t i = i0; i < i1; ++i) {
        const double cx = o.ppos[3 * i], cy = o.ppos[3 * i + 1], cz = o.ppos[3 * i + 2];
        int minx, maxx, miny, maxy, minz, maxz;
        support(o, cx, minx, maxx);
        support(o, cy, miny, maxy);
        support(o, cz, minz, maxz);
        for (int x = minx; x <= maxx; ++x)
            for (int y = miny; y <= maxy; ++y)
                for (int z = minz; z <= maxz; ++z) {
                    double cw = spline(cx - x) * spline(cy - y) * spline(cz - z);  // :869
                    if (!o.isSolid(x, y, z) && cw > 0) {                            // :870
                        size_t k = o.idx(x, y, z);
                        if (mt) g_locks.lock(k);                                    // :872
                        o.container[k] = (float)(o.container[k] + cw);              // :873
                        if (mt) g_locks.unlock(k);
                    }
                }
    }
    });
    int numActive = 0;
    for (int x = o.lo; x <= o.hi; x++)
        for (int y = o.lo; y <= o.hi; y++)
            for (int z = o.lo; z <= o.hi; z++)
                if (!o.isSolid(x, y, z) && o.withinW(x, y, z))       // :1423
                    if (o.container[o.idx(x, y, z)] > 0) {
                        o.indices[o.idx(x, y, z)] = numActive;
                        numActive++;
                    }
    o.numActive = numActive;
    for (int x = o.lo; x <= o.hi; x++)
        for (int y = o.lo; y <= o.hi; y++)
            for (int z = o.lo; z <= o.hi; z++)
                if (!o.isSolid(x, y, z)) o.output[o.idx(x, y, z)] = o.container[o.idx(x, y, z)];  // :1444
    o.velBefore = o.vel;  // :1455
}

// fluid.cc:414-479
void setRHS(Oracle& o, double dt)
{
    std::fill(o.rhs.begin(), o.rhs.end(), 0.0f);  // :1466 / :1478
    double scale = 1.0 / o.dx;
    double g[3] = {o.g[0] * dt, o.g[1] * dt, o.g[2] * dt};  // :420
    for (int x = o.lo; x <= o.hi; x++)
        for (int y = o.lo; y <= o.hi; y++)
            for (int z = o.lo; z <= o.hi; z++) {
                size_t k = o.idx(x, y, z);
                double val = o.container[k];
                if (val > 0 && !o.isSolid(x, y, z)) {
                    float& r = o.rhs[k];
                    if (o.withinB(x - 1, y, z) && o.isSolid(x - 1, y, z)) r = (float)(r - (scale * (o.getV(o.vel, x, y, z, 0) + g[0])));
                    if (o.withinB(x + 1, y, z) && o.isSolid(x + 1, y, z)) r = (float)(r + (scale * (o.getV(o.vel, x + 1, y, z, 0) + g[0])));
                    if (o.withinB(x, y - 1, z) && o.isSolid(x, y - 1, z)) r = (float)(r - (scale * (o.getV(o.vel, x, y, z, 1) + g[1])));
                    if (o.withinB(x, y + 1, z) && o.isSolid(x, y + 1, z)) r = (float)(r + (scale * (o.getV(o.vel, x, y + 1, z, 1) + g[1])));
                    if (o.withinB(x, y, z - 1) && o.isSolid(x, y, z - 1)) r = (float)(r - (scale * (o.getV(o.vel, x, y, z, 2) + g[2])));
                    if (o.withinB(x, y, z + 1) && o.isSolid(x, y, z + 1)) r = (float)(r + (scale * (o.getV(o.vel, x, y, z + 1, 2) + g[2])));
                }
            }
}

// fluid.cc:566-610
void setDiver(Oracle& o)
{
    std::fill(o.diver.begin(), o.diver.end(), 0.0f);  // :1465 / :1477
    for (int x = o.lo; x <= o.hi; x++)
        for (int y = o.lo; y <= o.hi; y++)
            for (int z = o.lo; z <= o.hi; z++) {
                size_t k = o.idx(x, y, z);
                if (o.container[k] > 0 && !o.isSolid(x, y, z)) {
                    double u = 0, vd = 0, w = 0;
                    if (!o.isSolid(x + 1, y, z)) u = (o.getV(o.vel, x + 1, y, z, 0) - o.getV(o.vel, x, y, z, 0)) / o.dx;
                    if (!o.isSolid(x, y + 1, z)) vd = (o.getV(o.vel, x, y + 1, z, 1) - o.getV(o.vel, x, y, z, 1)) / o.dx;
                    if (!o.isSolid(x, y, z + 1)) w = (o.getV(o.vel, x, y, z + 1, 2) - o.getV(o.vel, x, y, z, 2)) / o.dx;
                    o.diver[k] = (float)((o.rhs[k]) - u - vd - w);  // :605
                }
            }
}

// fluid.cc:304-412
void setA(Oracle& o, double dt)
{
    std::fill(o.Adiag.begin(), o.Adiag.end(), 0.0f);
    std::fill(o.Aplusi.begin(), o.Aplusi.end(), 0.0f);
    std::fill(o.Aplusj.begin(), o.Aplusj.end(), 0.0f);
    std::fill(o.Aplusk.begin(), o.Aplusk.end(), 0.0f);
    double scale = dt / (o.rho * o.dx * o.dx);
    static const int E[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    std::vector<float>* Ap[3] = {&o.Aplusi, &o.Aplusj, &o.Aplusk};
    for (int x = o.lo; x <= o.hi; x++)
        for (int y = o.lo; y <= o.hi; y++)
            for (int z = o.lo; z <= o.hi; z++) {
                size_t k = o.idx(x, y, z);
                double val = o.container[k];
                if (!o.isSolid(x, y, z) && val > 0) {  // :326
                    for (int a = 0; a < 3; ++a) {
                        int nx = x + E[a][0], ny = y + E[a][1], nz = z + E[a][2];
                        if (!o.isSolid(nx, ny, nz)) {
                            double val2 = o.getF(o.container, nx, ny, nz);
                            if (val2 > 0) {
                                size_t kn = o.idx(nx, ny, nz);  // val2>0 implies in range
                                o.Adiag[k] = (float)(o.Adiag[k] + scale);
                                o.Adiag[kn] = (float)(o.Adiag[kn] + scale);
                                (*Ap[a])[k] = (float)(-1 * scale);
                            } else {
                                o.Adiag[k] = (float)(o.Adiag[k] + scale);
                            }
                        }
                    }
                } else if (!o.isSolid(x, y, z) && o.withinW(x, y, z)) {  // :376
                    for (int a = 0; a < 3; ++a) {
                        int nx = x + E[a][0], ny = y + E[a][1], nz = z + E[a][2];
                        if (!o.isSolid(nx, ny, nz)) {
                            double val2 = o.getF(o.container, nx, ny, nz);
                            if (val2 > 0) {
                                size_t kn = o.idx(nx, ny, nz);
                                o.Adiag[kn] = (float)(o.Adiag[kn] + scale);
                            }
                        }
                    }
                }
            }
}

// fluid.cc:481-541 (setA2): triplets of both triangles + diagonal, b from diver.
void setA2(Oracle& o, std::vector<int>& tr, std::vector<int>& tc, std::vector<double>& tv, std::vector<double>& b)
{
    tr.clear(); tc.clear(); tv.clear();
    b.assign((size_t)o.numActive, 0.0);
    auto push = [&](int r, int c, double v) { tr.push_back(r); tc.push_back(c); tv.push_back(v); };
    for (int x = o.lo; x <= o.hi; x++)
        for (int y = o.lo; y <= o.hi; y++)
            for (int z = o.lo; z <= o.hi; z++) {
                size_t k = o.idx(x, y, z);
                if (o.Adiag[k] != 0) {
                    int index = o.indices[k];
                    if (o.Aplusi[k] != 0) push(index, o.indices[o.idx(x + 1, y, z)], o.Aplusi[k]);
                    if (o.Aplusj[k] != 0) push(index, o.indices[o.idx(x, y + 1, z)], o.Aplusj[k]);
                    if (o.Aplusk[k] != 0) push(index, o.indices[o.idx(x, y, z + 1)], o.Aplusk[k]);
                    if (o.getF(o.Aplusi, x - 1, y, z) != 0) push(index, o.indices[o.idx(x - 1, y, z)], o.Aplusi[o.idx(x - 1, y, z)]);
                    if (o.getF(o.Aplusj, x, y - 1, z) != 0) push(index, o.indices[o.idx(x, y - 1, z)], o.Aplusj[o.idx(x, y - 1, z)]);
                    if (o.getF(o.Aplusk, x, y, z - 1) != 0) push(index, o.indices[o.idx(x, y, z - 1)], o.Aplusk[o.idx(x, y, z - 1)]);
                    push(index, index, o.Adiag[k]);
                    b[index] = o.diver[k];  // :535
                }
            }
}

// fluid.cc:543-564
void setOnlyB(Oracle& o, std::vector<double>& b)
{
    b.assign((size_t)o.numActive, 0.0);
    for (size_t k = 0; k < o.ncell; ++k)
        if (o.Adiag[k] != 0) b[o.indices[k]] = o.diver[k];
}

// Matrix-free y = A x over the index space (A as setA2 assembles it).
struct StencilA {
    // per unknown: diag and up to 6 neighbour (index, value)
    std::vector<double> diag;
    std::vector<int> nb;      // 6 per unknown, -1 if none
    std::vector<double> nv;   // 6 per unknown
    int n;
};

void build_stencil(const Oracle& o, StencilA& A)
{
    A.n = o.numActive;
    A.diag.assign(A.n, 0.0);
    A.nb.assign((size_t)6 * A.n, -1);
    A.nv.assign((size_t)6 * A.n, 0.0);
    for (int x = o.lo; x <= o.hi; x++)
        for (int y = o.lo; y <= o.hi; y++)
            for (int z = o.lo; z <= o.hi; z++) {
                size_t k = o.idx(x, y, z);
                if (o.Adiag[k] != 0) {
                    int i = o.indices[k];
                    A.diag[i] = o.Adiag[k];
                    if (o.Aplusi[k] != 0) { A.nb[6 * i + 0] = o.indices[o.idx(x + 1, y, z)]; A.nv[6 * i + 0] = o.Aplusi[k]; }
                    if (o.Aplusj[k] != 0) { A.nb[6 * i + 1] = o.indices[o.idx(x, y + 1, z)]; A.nv[6 * i + 1] = o.Aplusj[k]; }
                    if (o.Aplusk[k] != 0) { A.nb[6 * i + 2] = o.indices[o.idx(x, y, z + 1)]; A.nv[6 * i + 2] = o.Aplusk[k]; }
                    if (o.getF(o.Aplusi, x - 1, y, z) != 0) { A.nb[6 * i + 3] = o.indices[o.idx(x - 1, y, z)]; A.nv[6 * i + 3] = o.Aplusi[o.idx(x - 1, y, z)]; }
                    if (o.getF(o.Aplusj, x, y - 1, z) != 0) { A.nb[6 * i + 4] = o.indices[o.idx(x, y - 1, z)]; A.nv[6 * i + 4] = o.Aplusj[o.idx(x, y - 1, z)]; }
                    if (o.getF(o.Aplusk, x, y, z - 1) != 0) { A.nb[6 * i + 5] = o.indices[o.idx(x, y, z - 1)]; A.nv[6 * i + 5] = o.Aplusk[o.idx(x, y, z - 1)]; }
                }
            }
}

inline void spmv(const StencilA& A, const std::vector<double>& x, std::vector<double>& y)
{
    for (int i = 0; i < A.n; ++i) {
        double s = A.diag[i] * x[i];
        for (int j = 0; j < 6; ++j) {
            int c = A.nb[6 * i + j];
            if (c >= 0) s += A.nv[6 * i + j] * x[c];
        }
        y[i] = s;
    }
}

// Eigen/src/IterativeLinearSolvers/ConjugateGradient.h:28-90 restated with Eigen's
// DiagonalPreconditioner (Jacobi: multiply by the stored 1/diag, BasicPreconditioners.h:73,91)
// in place of IncompleteCholesky; same start (x=0),
// same stopping rule (|r|^2 < tol^2 |b|^2 on the recursively updated residual),
// same iteration cap 2n (IterativeSolverBase.h:362-363), same b=0 early-out.
void cg_jacobi(const StencilA& A, const std::vector<double>& rhs, std::vector<double>& x, double tol, int& iters, double& relres)
{
    const int n = A.n;
    x.assign(n, 0.0);
    std::vector<double> residual(rhs), p(n), z(n), tmp(n), invdiag(n);
    for (int i = 0; i < n; ++i) invdiag[i] = 1.0 / A.diag[i];
    double rhsNorm2 = 0;
    for (int i = 0; i < n; ++i) rhsNorm2 += rhs[i] * rhs[i];
    if (rhsNorm2 == 0) { iters = 0; relres = 0; return; }
    double threshold = tol * tol * rhsNorm2;
    double residualNorm2 = rhsNorm2;
    if (residualNorm2 < threshold) { iters = 0; relres = sqrt(residualNorm2 / rhsNorm2); return; }
    for (int i = 0; i < n; ++i) p[i] = invdiag[i] * residual[i];
    double absNew = 0;
    for (int i = 0; i < n; ++i) absNew += residual[i] * p[i];
    int maxIters = 2 * n, i = 0;
    while (i < maxIters) {
        spmv(A, p, tmp);
        double pq = 0;
        for (int k = 0; k < n; ++k) pq += p[k] * tmp[k];
        double alpha = absNew / pq;
        residualNorm2 = 0;
        for (int k = 0; k < n; ++k) {
            x[k] += alpha * p[k];
            residual[k] -= alpha * tmp[k];
            residualNorm2 += residual[k] * residual[k];
        }
        if (residualNorm2 < threshold) break;
        double absOld = absNew;
        absNew = 0;
        for (int k = 0; k < n; ++k) { z[k] = invdiag[k] * residual[k]; absNew += residual[k] * z[k]; }
        double beta = absNew / absOld;
        for (int k = 0; k < n; ++k) p[k] = z[k] + beta * p[k];
        i++;
    }
    relres = sqrt(residualNorm2 / rhsNorm2);
    iters = i;
}

// fluid.cc:1472-1474: assemble + solve.  Uses the vendored-Eigen build when registered.
void solve(Oracle& o)
{
    std::vector<int> tr, tc;
    std::vector<double> tv;
    setA2(o, tr, tc, tv, o.b);
    o.p.assign((size_t)o.numActive, 0.0);
    int iters = 0;
    double relres = 0;
    if (o.numActive > 0) {
        if (o.ref_solver) {
            o.ref_solver(o.numActive, (int)tr.size(), tr.data(), tc.data(), tv.data(), o.b.data(), o.p.data(), &iters, &relres);
        } else {
            StencilA A;
            build_stencil(o, A);
            cg_jacobi(A, o.b, o.p, o.cg_tol, iters, relres);
        }
    }
    o.cg_iters_last = iters;
    o.cg_iters_total += iters;
    o.relres_last = relres;
    std::fill(o.pressure.begin(), o.pressure.end(), 0.0);
    for (size_t k = 0; k < o.ncell; ++k)
        if (o.indices[k] >= 0 && o.Adiag[k] != 0) o.pressure[k] = o.p[o.indices[k]];
}

// fluid.cc:612-703, dt argument = the caller's dt/10 (:1475)
void velUpdate(Oracle& o, double dt)
{
    double scale = dt / (o.rho * o.dx);
    for (int x = o.lo; x <= o.hi; x++)
        for (int y = o.lo; y <= o.hi; y++)
            for (int z = o.lo; z <= o.hi; z++) {
                size_t k = o.idx(x, y, z);
                double val = o.container[k];
                if (!o.isSolid(x, y, z) && val > 0) {
                    double pre = o.p[o.indices[k]];
                    double g[3] = {o.g[0] * dt, o.g[1] * dt, o.g[2] * dt};
                    o.vel[3 * k + 0] = o.vel[3 * k + 0] - scale * pre + g[0];
                    o.vel[3 * k + 1] = o.vel[3 * k + 1] - scale * pre + g[1];
                    o.vel[3 * k + 2] = o.vel[3 * k + 2] - scale * pre + g[2];
                    if (o.withinB(x + 1, y, z)) o.vel[3 * o.idx(x + 1, y, z) + 0] += scale * pre;
                    if (o.withinB(x, y + 1, z)) o.vel[3 * o.idx(x, y + 1, z) + 1] += scale * pre;
                    if (o.withinB(x, y, z + 1)) o.vel[3 * o.idx(x, y, z + 1) + 2] += scale * pre;
                }
            }
    for (int x = o.lo; x <= o.hi; x++)
        for (int y = o.lo; y <= o.hi; y++)
            for (int z = o.lo; z <= o.hi; z++)
                if (o.isSolid(x, y, z)) {
                    size_t k = o.idx(x, y, z);
                    o.vel[3 * k] = o.vel[3 * k + 1] = o.vel[3 * k + 2] = 0;
                    if (o.withinB(x + 1, y, z)) o.vel[3 * o.idx(x + 1, y, z) + 0] = 0;
                    if (o.withinB(x, y + 1, z)) o.vel[3 * o.idx(x, y + 1, z) + 1] = 0;
                    if (o.withinB(x, y, z + 1)) o.vel[3 * o.idx(x, y, z + 1) + 2] = 0;
                }
}

// One pass of the do{...}while body, fluid.cc:1457-1484.  Returns error.
double pressure_pass(Oracle& o)
{
    setRHS(o, o.dt);          // :1469
    setDiver(o);              // :1470
    setA(o, o.dt);            // :1471
    solve(o);                 // :1472-1474
    velUpdate(o, o.dt * o.update_frac);  // :1475  (dt/10)
    setRHS(o, o.dt);          // :1479
    setDiver(o);              // :1480
    setOnlyB(o, o.b2);        // :1481
    double num = 0, den = 0;
    for (int i = 0; i < o.numActive; ++i) {
        double d = o.b[i] - o.b2[i];
        num += d * d;
        den += o.b[i] * o.b[i];
    }
    return sqrt(num) / sqrt(den);  // :1483 (0/0 -> NaN, x/0 -> inf like Eigen norms)
}

// fluid.cc:972-1038 + CatmullRomFLIP 210-263
void flip_advect(Oracle& o)
{
    const double e = 0;
    const size_t np = o.ppos.size() / 3;
    double maxSpeed = 0.0;
    std::vector<double> tmax((size_t)std::max(o.nthreads, 1), 0.0);   // the reference: one mutex-guarded maximum (:977-990)
    parallel_for(o.nthreads, np, [&](size_t i0, size_t i1, int tid) {
    double maxSpeed = 0.0;
    for (size_t i = i0; i < i1; ++i) {
        const double cx = o.ppos[3 * i], cy = o.ppos[3 * i + 1], cz = o.ppos[3 * i + 2];
        int minx, maxx, miny, maxy, minz, maxz;
        support(o, cx, minx, maxx);
        support(o, cy, miny, maxy);
        support(o, cz, minz, maxz);
        double weight = 0, delta[3] = {0, 0, 0}, pic[3] = {0, 0, 0};
        for (int x = minx; x <= maxx; ++x)
            for (int y = miny; y <= maxy; ++y)
                for (int z = minz; z <= maxz; ++z)
                    if (o.withinW(x, y, z)) {  // :237
                        double velc[3], velp[3];
                        getVelocity(o, o.vel, x, y, z, velc);
                        getVelocity(o, o.velBefore, x, y, z, velp);
                        double cw = spline(cx - x) * spline(cy - y) * spline(cz - z);
                        weight += cw;
                        delta[0] += (velc[0] - velp[0]) * cw;  // :252
                        delta[1] += (velc[1] - velp[1]) * cw;
                        delta[2] += (velc[2] - velp[2]) * cw;
                        pic[0] += velc[0] * cw;  // clampedCatmullRom :172-174 (same cells, same weights; its clamp is commented out)
                        pic[1] += velc[1] * cw;
                        pic[2] += velc[2] * cw;
                    }
        if (weight != 0) {  // :258-262
            o.pvel[3 * i + 0] += delta[0] / weight;
            o.pvel[3 * i + 1] += delta[1] / weight;
            o.pvel[3 * i + 2] += delta[2] / weight;
            if (o.flip_blend < 1.0) {  // build extension: v' = b (v + delta) + (1 - b) v_pic
                for (int a = 0; a < 3; ++a)
                    o.pvel[3 * i + a] = o.flip_blend * o.pvel[3 * i + a] + (1.0 - o.flip_blend) * (pic[a] / weight);
            }
        }
        double len = sqrt(o.pvel[3 * i] * o.pvel[3 * i] + o.pvel[3 * i + 1] * o.pvel[3 * i + 1] + o.pvel[3 * i + 2] * o.pvel[3 * i + 2]);
        if (maxSpeed < len) maxSpeed = len;
    }
    tmax[tid] = maxSpeed;
    });
    for (double m : tmax) if (maxSpeed < m) maxSpeed = m;
    o.maxSpeed = maxSpeed;
    double timestep;
    if (maxSpeed != 0) timestep = o.max_dt < o.dx / maxSpeed ? o.max_dt : o.dx / maxSpeed;  // :992-999
    else timestep = o.max_dt;
    o.dt = timestep;  // written back to the caller's dt (double& timestep)
    parallel_for(o.nthreads, np, [&](size_t i0, size_t i1, int) {
    for (size_t i = i0; i < i1; ++i) {
        double* P = &o.ppos[3 * i];
        double* V = &o.pvel[3 * i];
        double position[3] = {P[0] + timestep * V[0], P[1] + timestep * V[1], P[2] + timestep * V[2]};
        int rx = (int)round(position[0]), ry = (int)round(position[1]), rz = (int)round(position[2]);
        if (o.isSolid(rx, ry, rz)) {
            double vx = V[0] * timestep, vy = V[1] * timestep, vz = V[2] * timestep;
            // Coord(double,double,double) -> Int32 truncation for the two untouched axes (:1017-1025)
            if (o.isSolid((int)round(P[0] + vx), (int)P[1], (int)P[2])) V[0] *= -1.0 * e;
            if (o.isSolid((int)P[0], (int)round(P[1] + vy), (int)P[2])) V[1] *= -1.0 * e;
            if (o.isSolid((int)P[0], (int)P[1], (int)round(P[2] + vz))) V[2] *= -1.0 * e;
            P[0] += V[0] * timestep;
            P[1] += V[1] * timestep;
            P[2] += V[2] * timestep;
        } else {
            P[0] = position[0]; P[1] = position[1]; P[2] = position[2];
        }
    }
    });
}

// fluid.cc:1378-1490
// fluid.cc:705-802 `extrapolate` — dead code in the reference (its only call, at the end of P2Gtransfer, is commented out:
// fluid.cc:1147), restated for SURVEY 8(f) row f3 as it would run THERE: `defined` is P2Gtransfer's grid (fluid.cc:1109-1146: true
// outside W, on solid cells and wherever weights > 0), bound = boundary (60 -> [lo,hi]), the literal scan range -60..60 -> [lo,hi],
// isWithinBounds(c, 58) -> withinW.  Breadth-first: every undefined cell next to the current layer (26-neighbourhood clamped to
// +-bound) receives the sum of that layer's adjacent velocities on top of what it holds, then the sum is divided by the count.
void extrapolate(Oracle& o)
{
    const int bound_lo = o.lo, bound_hi = o.hi;
    std::vector<int> nd(o.ncell, 0);                       // numTimesDefined, :707-710
    std::vector<uint8_t> defined(o.ncell, 0);
    for (int x = o.lo; x <= o.hi; ++x)                    // P2Gtransfer's defined grid, :1109-1125 and :1138-1143
        for (int y = o.lo; y <= o.hi; ++y)
            for (int z = o.lo; z <= o.hi; ++z) {
                const size_t k = o.idx(x, y, z);
                defined[k] = (!o.withinW(x, y, z) || o.isSolid(x, y, z) || (double)o.weights[k] > 0) ? 1 : 0;
            }
    struct C3 { int x, y, z; };
    std::vector<C3> definedCoord;
    auto spread = [&](const C3& c, std::vector<C3>& found) {   // :723-748 / :766-791
        const int minx = c.x == bound_lo ? c.x : c.x - 1, miny = c.y == bound_lo ? c.y : c.y - 1, minz = c.z == bound_lo ? c.z : c.z - 1;
        const int maxx = c.x == bound_hi ? c.x : c.x + 1, maxy = c.y == bound_hi ? c.y : c.y + 1, maxz = c.z == bound_hi ? c.z : c.z + 1;
        const size_t kc = o.idx(c.x, c.y, c.z);
        for (int i = minx; i <= maxx; ++i)
            for (int j = miny; j <= maxy; ++j)
                for (int k = minz; k <= maxz; ++k) {
                    const size_t kn = o.idx(i, j, k);
                    if (!defined[kn]) {
                        if (nd[kn] == 0) found.push_back(C3{i, j, k});
                        nd[kn] = nd[kn] + 1;
                        for (int a = 0; a < 3; ++a) o.vel[3 * kn + a] = o.vel[3 * kc + a] + o.vel[3 * kn + a];   // :746
                    }
                }
    };
    for (int x = o.lo; x <= o.hi; ++x)                    // :714-753
        for (int y = o.lo; y <= o.hi; ++y)
            for (int z = o.lo; z <= o.hi; ++z)
                if (defined[o.idx(x, y, z)] && !o.isSolid(x, y, z) && o.withinW(x, y, z)) spread(C3{x, y, z}, definedCoord);
    auto settle = [&](const std::vector<C3>& v) {        // :755-759 / :794-798
        for (const C3& c : v) {
            const size_t k = o.idx(c.x, c.y, c.z);
            for (int a = 0; a < 3; ++a) o.vel[3 * k + a] = o.vel[3 * k + a] / nd[k];   // Vec3d / int
            defined[k] = 1;
        }
    };
    settle(definedCoord);
    while (definedCoord.size()) {                          // :760-801
        std::vector<C3> tempCoord;
        for (const C3& c : definedCoord) spread(c, tempCoord);
        settle(tempCoord);
        definedCoord = tempCoord;
    }
}

// fluid.cc:1053-1080 `PointList::resample(numParticlesPerCell)` — never called by the reference's main(); restated serially (the
// tbb::parallel_for in index order: under real TBB WHICH particles of an over-full cell go is schedule-dependent).  A particle whose
// base cell already holds numParticlesPerCell earlier particles is parked at (100, 100, 100) — generalised: boundary + 40 — and only
// cells with rx < 50 (boundary - 10) are looked at.
void resample(Oracle& o, int per_cell)
{
    std::vector<int> number(o.ncell, 0);
    const size_t np = o.ppos.size() / 3;
    const int xlim = o.hi - 10;
    const double far_ = (double)(o.hi + 40);
    for (size_t i = 0; i < np; ++i) {
        const int rx = (int)round(o.ppos[3 * i]), ry = (int)round(o.ppos[3 * i + 1]), rz = (int)round(o.ppos[3 * i + 2]);
        if (rx < xlim && o.inRange(rx, ry, rz)) {        // (off the grid the reference indexes its lock cube out of bounds: left alone here)
            const size_t k = o.idx(rx, ry, rz);
            if (number[k] + 1 > per_cell) o.ppos[3 * i] = o.ppos[3 * i + 1] = o.ppos[3 * i + 2] = far_;
            else number[k] = number[k] + 1;
        }
    }
}

void step(Oracle& o, int max_passes)
{
    o.outer_passes = 0;
    o.cg_iters_total = 0;
    p2g(o);
    flags_index(o);
    double error;
    do {
        error = pressure_pass(o);
        o.outer_passes++;
        if (max_passes > 0 && o.outer_passes >= max_passes) break;
    } while (error > o.outer_tol);
    o.error = error;
    flip_advect(o);
}

}  // namespace

extern "C" {

void* oracle_create(int N, double dx, double rho, const double* g, double max_dt, double outer_tol, double update_frac)
{
    Oracle* o = new Oracle();
    o->N = N;
    o->lo = -(N / 2);
    o->hi = o->lo + N - 1;
    o->wlo = o->lo + 2;
    o->whi = o->hi - 2;
    o->dx = dx; o->rho = rho;
    o->g[0] = g[0]; o->g[1] = g[1]; o->g[2] = g[2];
    o->max_dt = max_dt; o->outer_tol = outer_tol; o->update_frac = update_frac;
    o->dt = max_dt;
    o->cg_tol = 2.220446049250313e-16;
    o->nthreads = 1;
    o->ref_solver = nullptr;
    o->ncell = (size_t)N * N * N;
    for (auto* v : {&o->solid, &o->container, &o->output, &o->weights, &o->Adiag, &o->Aplusi, &o->Aplusj, &o->Aplusk, &o->diver, &o->rhs})
        v->assign(o->ncell, 0.0f);
    o->indices.assign(o->ncell, 0);
    o->vel.assign(3 * o->ncell, 0.0);
    o->velBefore.assign(3 * o->ncell, 0.0);
    o->pressure.assign(o->ncell, 0.0);
    o->numActive = 0;
    o->outer_passes = o->cg_iters_total = o->cg_iters_last = 0;
    o->error = o->maxSpeed = o->relres_last = 0;
    // fluid.cc:1256-1266: solid = 1 outside [wlo,whi]^3
    for (int x = o->lo; x <= o->hi; x++)
        for (int y = o->lo; y <= o->hi; y++)
            for (int z = o->lo; z <= o->hi; z++)
                if (!o->withinW(x, y, z)) o->solid[o->idx(x, y, z)] = 1;
    return o;
}
void oracle_destroy(void* h) { delete (Oracle*)h; }
void oracle_set_ref_solver(void* h, void* fn) { ((Oracle*)h)->ref_solver = (ref_solver_fn)fn; }
void oracle_set_cg_tol(void* h, double tol) { ((Oracle*)h)->cg_tol = tol; }
void oracle_set_threads(void* h, int n) { ((Oracle*)h)->nthreads = n < 1 ? 1 : n; }
void oracle_set_flip_blend(void* h, double b) { ((Oracle*)h)->flip_blend = b; }
void oracle_set_dt(void* h, double dt) { ((Oracle*)h)->dt = dt; }
double oracle_get_dt(void* h) { return ((Oracle*)h)->dt; }
// Every cell outside [wlo,whi]^3 must stay solid: the reference indexes pressure(-1) for a
// fluid cell outside W (fluid.cc:1423 vs :637), which is undefined behaviour there.
int oracle_set_solid(void* h, const uint8_t* s)
{
    Oracle* o = (Oracle*)h;
    for (int x = o->lo; x <= o->hi; x++)
        for (int y = o->lo; y <= o->hi; y++)
            for (int z = o->lo; z <= o->hi; z++)
                if (!o->withinW(x, y, z) && !s[o->idx(x, y, z)]) return -1;
    for (size_t k = 0; k < o->ncell; ++k) o->solid[k] = s[k] ? 1.0f : 0.0f;
    return 0;
}
void oracle_set_particles(void* h, long n, const double* pos, const double* vel)
{
    Oracle* o = (Oracle*)h;
    o->ppos.assign(pos, pos + 3 * n);
    o->pvel.assign(vel, vel + 3 * n);
}
long oracle_num_particles(void* h) { return (long)(((Oracle*)h)->ppos.size() / 3); }
void oracle_get_particles(void* h, double* pos, double* vel)
{
    Oracle* o = (Oracle*)h;
    memcpy(pos, o->ppos.data(), o->ppos.size() * sizeof(double));
    memcpy(vel, o->pvel.data(), o->pvel.size() * sizeof(double));
}
void oracle_p2g(void* h) { p2g(*(Oracle*)h); }
void oracle_flags_index(void* h) { flags_index(*(Oracle*)h); }
void oracle_rhs_div(void* h) { Oracle* o = (Oracle*)h; setRHS(*o, o->dt); setDiver(*o); }
void oracle_build_matrix(void* h) { Oracle* o = (Oracle*)h; setA(*o, o->dt); }
void oracle_solve(void* h) { solve(*(Oracle*)h); }
void oracle_vel_update(void* h) { Oracle* o = (Oracle*)h; velUpdate(*o, o->dt * o->update_frac); }
double oracle_pressure_pass(void* h) { return pressure_pass(*(Oracle*)h); }
void oracle_flip_advect(void* h) { flip_advect(*(Oracle*)h); }
void oracle_step(void* h, int max_passes) { step(*(Oracle*)h, max_passes); }
void oracle_extrapolate(void* h) { extrapolate(*(Oracle*)h); }
void oracle_resample(void* h, int per_cell) { resample(*(Oracle*)h, per_cell); }

// stats: [dt, numActive, outer_passes, cg_iters_total, cg_iters_last, relres_last, error, maxSpeed]
void oracle_stats(void* h, double* out)
{
    Oracle* o = (Oracle*)h;
    out[0] = o->dt; out[1] = o->numActive; out[2] = o->outer_passes; out[3] = o->cg_iters_total;
    out[4] = o->cg_iters_last; out[5] = o->relres_last; out[6] = o->error; out[7] = o->maxSpeed;
}

// field ids match include/fluid_hip.h FLUID_FIELD_*
int oracle_get_field(void* h, int id, void* dst)
{
    Oracle* o = (Oracle*)h;
    const size_t n = o->ncell;
    auto cpf = [&](const std::vector<float>& v) { memcpy(dst, v.data(), n * sizeof(float)); return 0; };
    switch (id) {
    case 0: return cpf(o->container);
    case 1: return cpf(o->weights);
    case 2: {  // vel as 3 SoA planes [3][N^3] double
        double* d = (double*)dst;
        for (size_t k = 0; k < n; ++k) { d[k] = o->vel[3 * k]; d[n + k] = o->vel[3 * k + 1]; d[2 * n + k] = o->vel[3 * k + 2]; }
        return 0; }
    case 3: {
        double* d = (double*)dst;
        for (size_t k = 0; k < n; ++k) { d[k] = o->velBefore[3 * k]; d[n + k] = o->velBefore[3 * k + 1]; d[2 * n + k] = o->velBefore[3 * k + 2]; }
        return 0; }
    case 4: memcpy(dst, o->indices.data(), n * sizeof(int32_t)); return 0;
    case 5: return cpf(o->rhs);
    case 6: return cpf(o->diver);
    case 7: memcpy(dst, o->pressure.data(), n * sizeof(double)); return 0;
    case 8: return cpf(o->output);
    case 9: {  // solid as u8
        uint8_t* d = (uint8_t*)dst;
        for (size_t k = 0; k < n; ++k) d[k] = o->solid[k] == 1 ? 1 : 0;
        return 0; }
    case 10: return cpf(o->Adiag);
    case 11: return cpf(o->Aplusi);
    case 12: return cpf(o->Aplusj);
    case 13: return cpf(o->Aplusk);
    }
    return -1;
}

// raw vectors of the last pass (index space)
int oracle_num_active(void* h) { return ((Oracle*)h)->numActive; }
void oracle_get_b(void* h, double* b, double* b2, double* p)
{
    Oracle* o = (Oracle*)h;
    if (b) memcpy(b, o->b.data(), o->b.size() * sizeof(double));
    if (b2 && !o->b2.empty()) memcpy(b2, o->b2.data(), o->b2.size() * sizeof(double));
    if (p) memcpy(p, o->p.data(), o->p.size() * sizeof(double));
}

// assembled matrix in triplet form (as setA2 pushes it); call with NULLs to get nnz
int oracle_get_triplets(void* h, int* rows, int* cols, double* vals)
{
    Oracle* o = (Oracle*)h;
    std::vector<int> tr, tc;
    std::vector<double> tv;
    setA2(*o, tr, tc, tv, o->b);  // also (re)fills b, as setA2 does (fluid.cc:535)
    if (rows) {
        memcpy(rows, tr.data(), tr.size() * sizeof(int));
        memcpy(cols, tc.data(), tc.size() * sizeof(int));
        memcpy(vals, tv.data(), tv.size() * sizeof(double));
    }
    return (int)tr.size();
}

// stand-alone pieces for known-answer tests
double oracle_spline(double x) { return spline(x); }

// Jacobi-CG on a general small SPD system given as triplets (KAT: TestConjGradient.cc:58-105)
void oracle_cg_triplets(int n, int nnz, const int* rows, const int* cols, const double* vals, const double* b,
                        double* x, double tol, int* iters, double* relres)
{
    std::vector<double> diag(n, 0.0), r(b, b + n), p(n), z(n), q(n);
    for (int k = 0; k < nnz; ++k) if (rows[k] == cols[k]) diag[rows[k]] += vals[k];
    auto mv = [&](const std::vector<double>& v, std::vector<double>& y) {
        std::fill(y.begin(), y.end(), 0.0);
        for (int k = 0; k < nnz; ++k) y[rows[k]] += vals[k] * v[cols[k]];
    };
    std::fill(x, x + n, 0.0);
    double bb = 0;
    for (int i = 0; i < n; ++i) bb += b[i] * b[i];
    *iters = 0; *relres = 0;
    if (bb == 0) return;
    double thr = tol * tol * bb, rr = bb, rz = 0;
    for (int i = 0; i < n; ++i) { p[i] = r[i] / diag[i]; rz += r[i] * p[i]; }
    int it = 0;
    while (it < 2 * n) {
        mv(p, q);
        double pq = 0;
        for (int i = 0; i < n; ++i) pq += p[i] * q[i];
        double a = rz / pq;
        rr = 0;
        for (int i = 0; i < n; ++i) { x[i] += a * p[i]; r[i] -= a * q[i]; rr += r[i] * r[i]; }
        if (rr < thr) break;
        double rzo = rz; rz = 0;
        for (int i = 0; i < n; ++i) { z[i] = r[i] / diag[i]; rz += r[i] * z[i]; }
        double be = rz / rzo;
        for (int i = 0; i < n; ++i) p[i] = z[i] + be * p[i];
        it++;
    }
    *iters = it; *relres = sqrt(rr / bb);
}

}  // extern "C"
