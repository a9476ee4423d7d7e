// The snow-MPM step of the reference (`./run.sh mpm`) for gfx950: kernels and host code behind
// include/mpm_hip.h.  Not a port of mpm.cc: the reference assembles a 3n x 3n sparse matrix through a std::map of 3x3 blocks
// (729 node pairs per particle, two SVDs and three QR solves per pair, mpm.cc:646-701) and hands it to Eigen's CG; here the
// matrix never exists.  A v is applied per particle:  dF = (sum_j v_j (x) grad w_jp) F_p,  A_p = d2Psi/dF2 : dF,
// y_i = v_i + beta dt^2 / m_i * sum_p vol_p A_p F_p^T grad w_ip  — the same linear operator (deformHeader.h:241-272 is
// linear in dF), one particle pass per CG iteration, with the polar decomposition of a particle computed once per step.
// M = D^-1 K (D = node masses, K the symmetric energy Hessian) is not symmetric, and the reference's solver object multiplies
// by the TRANSPOSE of the matrix it is given (Eigen's ConjugateGradient with Lower|Upper on a column-major matrix,
// ConjugateGradient.h:202-212 — harmless for the symmetric matrices it is meant for): the program's velocities solve
// (I + beta dt^2 K D^-1) x = b.  That is what this file solves by default (transpose_system = 1); either operator is
// self-adjoint in a weighted inner product (u^T D^-1 v, or u^T D v for the matrix as assembled), so the CG below uses
// weighted dot products; the stopping rule stays Eigen's plain |r|^2 <= tol^2 |b|^2.
//
// Data: dense node arrays over -B..B (z fastest) and SoA particle arrays in HBM, all fp64 except the float32 node mass
// (FloatGrid) which is summed in fp64 and rounded once.  The three particle -> node sums (transfer, forces, operator) are GATHERS
// per node over the cell lists of the sorted particles (k_mpm_gather): no atomics, the order of every sum is a function of the
// input alone — two runs give the same bits.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mpm_hip.h"

int fluid_fail(int code, const std::string& msg);
namespace fl {
// kernels_particles.hip: order2 = every cell's list re-ranked by ascending id, 256 sorted positions per block with the ids of
// their cells staged through LDS (cell_start needs its one-past-the-end entry)
void launch_bin_rank(hipStream_t st, long n_pos, long pos0, const int* key, const int* cell_start, const int* order, const uint32_t* spid, int* order2);
}
#define HIPCHK(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess)                                                                                     \
            return fluid_fail(FLUID_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " @" + std::to_string(__LINE__)); \
    } while (0)

namespace {

struct MGrid {
    int B, W, N;
    __host__ __device__ inline bool in(int x, int y, int z) const { return x >= -B && x <= B && y >= -B && y <= B && z >= -B && z <= B; }
    __host__ __device__ inline long at(int x, int y, int z) const { return ((long)(x + B) * N + (y + B)) * N + (z + B); }
    __host__ __device__ inline long cells() const { return (long)N * N * N; }
};

// per-step device scalars
struct MpmState {
    int num_active;
    int cg_done;
    int max_cell;                    // first cell attaining max_force_coeff2
    int any_active;
    int cg_iters;                    // Eigen's `i` (ConjugateGradient.h:70-88)
    int num_touched;                 // nodes within one cell of a particle (the nodes the transfer and the force gather visit)
    int num_cells;                   // non-empty cells
    // "the solve ends with this iteration" as k_mpm_cg_xrp (one launch for the residual update AND the convergence test) records it: slot =
    // parity of the NEXT iteration, value = ((launch + 1) << 2) | status, so that a block of the launch that writes it can tell its own mark
    // from an earlier launch's (a block that started late would otherwise skip its share of the last update of x); every later kernel
    // treats a set slot like cg_done, the host folds the status into cg_done
    int pend[2];
    int pad_;
    unsigned long long max_speed_bits, max_grad_bits, max_fp_bits, max_fe_bits, max_coeff_bits;
    double dt;
    double bb, pq, rr;               // |b|^2, <p,Ap>_w, |r|^2
    double rho2[2];                  // <r,r>_w of the current / next iteration (slot = iteration parity)
    double max_force[3], max_mi;
};

__device__ __forceinline__ int solve_over(const MpmState* st) { return st->cg_done | st->pend[0] | st->pend[1]; }

struct Part {   // SoA, stride = capacity
    double *pos, *vel, *FE, *FP, *gradV, *volume;
    // per-step cache for the operator: R, inverse of getDelR's 3x3 matrix (the cofactor matrix is formed again from F where it is used)
    double *R, *Minv, *coef;         // coef: mu_p, lambda_p, J
    double* node9;                   // 10 doubles per particle for the node gathers: the stress (forces) / A_p F_p^T (operator) + its scale
    double* wfac;                    // 12 doubles per particle, once per step: s2 and grad factors (deformHeader.h:99-101) of the nodes base-1, base per axis
    long cap;
};

__device__ __forceinline__ double ld(const double* a, long cap, int k, long i) { return a[(long)k * cap + i]; }
__device__ __forceinline__ void stv(double* a, long cap, int k, long i, double v) { a[(long)k * cap + i] = v; }

// mpm.cc:25-41 (half-cell shift, coefficient 1.0)
__device__ __forceinline__ double mspline(double x)
{
    x -= 0.5;
    if (x < 0) x *= -1.0;
    if (x < 0.5) return 1.0 * (4.0 * x * x * x - 4.0 * x * x + 2.0 / 3.0);
    if (x <= 1.0) return 1.0 * ((-8.0 * (x * x * x) / 6.0) + 4.0 * x * x - 4.0 * x + 4.0 / 3.0);
    return 0;
}
// deformHeader.h:38-53
__device__ __forceinline__ double mspline2(double x)
{
    if (x < 0) x *= -1.0;
    if (x < 0.5) return 1.0 * (4.0 * x * x * x - 4.0 * x * x + 2.0 / 3.0);
    if (x < 1.0) return 1.0 * ((-8.0 * (x * x * x) / 6.0) + 4.0 * x * x - 4.0 * x + 4.0 / 3.0);
    return 0;
}
// deformHeader.h:54-88
__device__ __forceinline__ double mspline_grad(double x)
{
    if (x >= 0) {
        if (x < 0.5) return 1.0 * (12.0 * x * x - 8.0 * x);
        if (x <= 1.0) return 1.0 * ((-8.0 * (x * x) / 2.0) + 8.0 * x - 4.0);
        return 0;
    }
    if (x > -0.5) return 1.0 * (-12.0 * x * x - 8.0 * x);
    if (x >= -1.0) return 1.0 * ((8.0 * (x * x) / 2.0) + 8.0 * x + 4.0);
    return 0;
}

// The 27-node neighbourhood of a particle (base = round(pos), base +- 1 clamped to +-B: mpm.cc:503-513 and every other
// particle loop) with the separable factors of the weights: w (mpm.cc spline of pos - node), s2 / g (deformHeader.h:99-101).
struct Nbh {
    int lo[3], hi[3];
    double w[3][3], s2[3][3], g[3][3];   // [axis][node - lo]
};
__device__ __forceinline__ void neighbourhood(const MGrid& G, const double p[3], Nbh& nb, bool grads)
{
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int f = (int)round(p[a]);
        nb.lo[a] = f - 1 > -G.B ? f - 1 : -G.B;
        nb.hi[a] = f + 1 < G.B ? f + 1 : G.B;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int c = nb.lo[a] + k;
            nb.w[a][k] = mspline(p[a] - c);
            if (grads) {
                nb.s2[a][k] = mspline2(0.5 + c - p[a]);
                nb.g[a][k] = mspline_grad(p[a] - c - 0.5);
            }
        }
    }
}
// getGradW, deformHeader.h:99-103
__device__ __forceinline__ void grad_w(const Nbh& nb, int i, int j, int k, double g[3])
{
    g[0] = -1 * nb.g[0][i] * nb.s2[1][j] * nb.s2[2][k];
    g[1] = -1 * nb.s2[0][i] * nb.g[1][j] * nb.s2[2][k];
    g[2] = -1 * nb.s2[0][i] * nb.s2[1][j] * nb.g[2][k];
}

__device__ __forceinline__ void atomic_max_pos(unsigned long long* slot, double v)
{
    if (v > 0) atomicMax(slot, (unsigned long long)__double_as_longlong(v));   // non-negative doubles order like their bits
}

// ---- 3x3 helpers (row-major double[9]) ----
__device__ __forceinline__ void mat_mul(const double* a, const double* b, double* r)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
__device__ __forceinline__ void mat_mul_bt(const double* a, const double* b, double* r)   // a * b^T
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r[3 * i + j] = a[3 * i] * b[3 * j] + a[3 * i + 1] * b[3 * j + 1] + a[3 * i + 2] * b[3 * j + 2];
}
__device__ __forceinline__ double mat_det(const double* a)
{
    return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}
// getJFmt, deformHeader.h:227-239: the cofactor matrix J F^-T
__device__ __forceinline__ void cofactor(const double* F, double* r)
{
    r[0] = F[4] * F[8] - F[5] * F[7], r[1] = F[5] * F[6] - F[3] * F[8], r[2] = F[3] * F[7] - F[4] * F[6];
    r[3] = F[2] * F[7] - F[1] * F[8], r[4] = F[0] * F[8] - F[2] * F[6], r[5] = F[1] * F[6] - F[0] * F[7];
    r[6] = F[1] * F[5] - F[2] * F[4], r[7] = F[2] * F[3] - F[0] * F[5], r[8] = F[0] * F[4] - F[1] * F[3];
}
// F = U diag(s) V^T by one-sided Jacobi (in place of Eigen::JacobiSVD, deformHeader.h:24, mpm.cc:545): the step only ever
// uses products that do not depend on the order or signs of the factors.
__device__ void svd3(const double* F, double* U, double* s, double* V)
{
    double A[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) A[k] = F[k], V[k] = (k % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            double al = 0, be = 0, ga = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                al += A[3 * k + p] * A[3 * k + p];
                be += A[3 * k + q] * A[3 * k + q];
                ga += A[3 * k + p] * A[3 * k + q];
            }
            if (ga == 0.0 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
            rotated = true;
            double zeta = (be - al) / (2.0 * ga);
            double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double ap = A[3 * k + p], aq = A[3 * k + q];
                A[3 * k + p] = c * ap - sn * aq;
                A[3 * k + q] = sn * ap + c * aq;
                double vp = V[3 * k + p], vq = V[3 * k + q];
                V[3 * k + p] = c * vp - sn * vq;
                V[3 * k + q] = sn * vp + c * vq;
            }
        }
        if (!rotated) break;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double n = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
        s[j] = n;
#pragma unroll
        for (int k = 0; k < 3; ++k) U[3 * k + j] = n > 0 ? A[3 * k + j] / n : 0.0;
    }
}


// ---- constitutive pieces shared by the step kernels and by mpm_eval (the known-answer hook) ----
struct Setup {          // per particle, once per step
    double R[9], S[9], Minv[9], sigma[9], mu, lambda, J;
};
// getR / getS (deformHeader.h:22-36), getSigma (:273-307), and what getdPsydx2 recomputes for every node pair (:253-263):
// the hardened coefficients, J, the cofactor matrix (getJFmt, :227-239) and the inverse of getDelR's 3x3 matrix (:139-142)
__device__ __forceinline__ void particle_setup(const double* F, const double* FP, double mu0, double lambda0, double eps, Setup& o)
{
    double U[9], V[9], sv[3];
    svd3(F, U, sv, V);
    mat_mul_bt(U, V, o.R);
    double VD[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) VD[3 * r + c] = V[3 * r + c] * sv[c];
    mat_mul_bt(VD, V, o.S);
    const double Jp = mat_det(FP);
    const double hard = exp(eps * (1 - Jp));
    o.mu = mu0 * hard, o.lambda = lambda0 * hard;
    o.J = mat_det(F);
    double FmR[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) FmR[k] = F[k] - o.R[k];
    mat_mul_bt(FmR, F, o.sigma);
#pragma unroll
    for (int k = 0; k < 9; ++k) o.sigma[k] = 2 * o.mu * o.sigma[k] + ((k % 4 == 0) ? o.lambda * (o.J - 1) * o.J : 0.0);
    const double* S = o.S;
    const double m[9] = {S[0] + S[4], S[5], -1 * S[2], S[5], S[0] + S[8], S[1], -1 * S[2], S[1], S[4] + S[8]};
    double mi[9];
    cofactor(m, mi);                           // cof(m) = det(m) m^-T; m is symmetric
    const double dm = mat_det(m);
#pragma unroll
    for (int k = 0; k < 9; ++k) o.Minv[k] = dm != 0 ? mi[k] / dm : 0.0;
}
// Ap = d2Psi/dF2 : dF (dPsydFdF, deformHeader.h:241-249, for a general dF): 2 mu dF - 2 mu dR + lambda cof (cof : dF) + lambda (J - 1) dcof,
// dR from getDelR (:133-147), dcof = the derivative of the cofactor matrix along dF (what getdJF / doubleDot42 tabulate, :148-212)
// The four terms are formed and added one after the other (same left-to-right order as :248), each from the inputs it alone needs, so
// that R / Minv and cof need not be live together; cof is cofactor(F) again, the expression of the reference's setup.
__device__ __forceinline__ void hessian_apply(const double* F, const double* R, const double* Mi, double mu, double lambda, double J, const double* dF,
                                              double* Ap)
{
    {
        double RtdF[9];
        {
            const double Rt[9] = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]};
            mat_mul(Rt, dF, RtdF);
        }
        const double rhs01 = RtdF[1] - RtdF[3], rhs02 = RtdF[2] - RtdF[6], rhs12 = RtdF[5] - RtdF[7];   // R^T dF - dF^T R
        const double x0 = Mi[0] * rhs01 + Mi[1] * rhs02 + Mi[2] * rhs12;
        const double x1 = Mi[3] * rhs01 + Mi[4] * rhs02 + Mi[5] * rhs12;
        const double x2 = Mi[6] * rhs01 + Mi[7] * rhs02 + Mi[8] * rhs12;
        const double rdr[9] = {0, x0, x1, -1 * x0, 0, x2, -1 * x1, -1 * x2, 0};
        double dR[9];
        mat_mul(R, rdr, dR);
#pragma unroll
        for (int k = 0; k < 9; ++k) Ap[k] = 2 * mu * dF[k] - 2 * mu * dR[k];
    }
    {
        double cf[9];
        cofactor(F, cf);
        double dd = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) dd += cf[k] * dF[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) Ap[k] += lambda * cf[k] * dd;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int r1 = (r + 1) % 3, r2 = (r + 2) % 3, c1 = (c + 1) % 3, c2 = (c + 2) % 3;
            const double dcf = dF[3 * r1 + c1] * F[3 * r2 + c2] + F[3 * r1 + c1] * dF[3 * r2 + c2] - dF[3 * r1 + c2] * F[3 * r2 + c1] -
                               F[3 * r1 + c2] * dF[3 * r2 + c1];
            Ap[3 * r + c] += lambda * (J - 1) * dcf;   // deformHeader.h:248
        }
}
// the singular-value clamp of updateDeformationGradient, mpm.cc:543-555
__device__ __forceinline__ void clamp_update(const double* tFE, const double* FP, double minv, double maxv, double* nFE, double* nFP)
{
    double F[9], U[9], V[9], sv[3];
    mat_mul(tFE, FP, F);
    svd3(tFE, U, sv, V);
    double UD[9], VDi[9];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double s = sv[c] > minv ? sv[c] : minv;
        s = s < maxv ? s : maxv;
#pragma unroll
        for (int r = 0; r < 3; ++r) UD[3 * r + c] = U[3 * r + c] * s, VDi[3 * r + c] = V[3 * r + c] * (1.0 / s);
    }
    double T[9];
    mat_mul_bt(UD, V, nFE);
    mat_mul_bt(VDi, U, T);
    mat_mul(T, F, nFP);
}
// mpm_eval: one thread per item, plain AoS arrays
__global__ void k_mpm_eval(int what, long n, const double* __restrict__ a, const double* __restrict__ b, double p0, double p1, double p2,
                           double* __restrict__ out0, double* __restrict__ out1)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double A[9], B[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) A[k] = a[9 * i + k], B[k] = b ? b[9 * i + k] : ((k % 4 == 0) ? 1.0 : 0.0);
    if (what == MPM_EVAL_POLAR || what == MPM_EVAL_SIGMA) {
        Setup st;
        particle_setup(A, B, p0, p1, p2, st);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (what == MPM_EVAL_POLAR) out0[9 * i + k] = st.R[k], out1[9 * i + k] = st.S[k];
            else out0[9 * i + k] = st.sigma[k];
        }
    } else if (what == MPM_EVAL_HESSIAN) {   // a = F, b = dF, p0 = lambda, p1 = mu (taken as they are: no hardening)
        double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        Setup st;
        particle_setup(A, I, 0.0, 0.0, 0.0, st);
        double Ap[9];
        hessian_apply(A, st.R, st.Minv, p1, p0, st.J, B, Ap);
#pragma unroll
        for (int k = 0; k < 9; ++k) out0[9 * i + k] = Ap[k];
    } else if (what == MPM_EVAL_CLAMP) {     // a = tFE, b = FP, p0 = minv, p1 = maxv
        double nFE[9], nFP[9];
        clamp_update(A, B, p0, p1, nFE, nFP);
#pragma unroll
        for (int k = 0; k < 9; ++k) out0[9 * i + k] = nFE[k], out1[9 * i + k] = nFP[k];
    }
}

// ---- particle -> node sums as gathers over cell lists ----
// The reference scatters from particles to their 27 nodes (mpm.cc:218-253, 596-644, 646-701).  The first version here did the
// same with fp64 atomics (aggregated along runs of equal base cell inside a wave): fast enough, but the order of an atomic sum
// is whatever the hardware makes it, so two runs differed in the last bits, and the aggregation cost the operator kernel 972
// ds_bpermute per particle and 186 VGPRs.  Now every step sorts the particles by base cell AND, inside a cell, by their upload
// index (k_mpm_sort_rank), the per-particle part of a sum (stress, A_p F_p^T) is written to `node9`, and one workgroup per NODE
// walks the particle lists of the 27 cells around it: wave w takes the cells w, w+4, ..., lanes stride the particles of a cell,
// every lane adds in list order, lanes and waves are folded in a fixed order.  The weights are evaluated at the node itself:
// nb.w[a][node - lo] of the scatter form IS mspline(p[a] - node[a]).
template <int NQ>
__device__ __forceinline__ void gather_fold(double (&acc)[NQ], double (*sh)[4])
{
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double v = acc[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        acc[q] = v;
    }
    __syncthreads();   // the LDS words may still be read by the previous node's epilogue
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) sh[q][threadIdx.x >> 6] = acc[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = ((sh[q][0] + sh[q][1]) + sh[q][2]) + sh[q][3];
}
// does particle position p reach node c along one axis (mpm.cc:503-513: nodes base-1 .. base+1, clamped to the grid), and the slot rule of
// the 3-wide loops (node - lo <= 2)
__device__ __forceinline__ bool reaches(const MGrid& G, double p, int c)
{
    const int f = (int)round(p);
    const int lo = f - 1 > -G.B ? f - 1 : -G.B, hi = f + 1 < G.B ? f + 1 : G.B;
    return c >= lo && c <= hi && c - lo <= 2;
}

// MODE 0: interpolate + P2Gtransfer (mpm.cc:773-811, 218-253, 996-999) over the touched nodes: massd, vel (3 x cells)
// MODE 1: populateGridForces' node sums (mpm.cc:596-644) over the touched nodes: forces (3 x cells); node9 = sigma, scale = volume
// MODE 2: the operator's node sums over the unknowns: y = v + sum; node9 = A_p F_p^T, scale = beta dt^2 vol_p
template <int MODE>
__global__ void __launch_bounds__(256) k_mpm_gather(MGrid G, Part P, const int* __restrict__ cell_start, const int* __restrict__ cell_count,
                                                    const int* __restrict__ list, const int* __restrict__ n_list, const MpmState* st, int in_solve,
                                                    int transposed, const double* __restrict__ invm, const double* __restrict__ v,
                                                    double* __restrict__ out0, double* __restrict__ out1)
{
    constexpr int NQ = MODE == 0 ? 4 : 3;
    __shared__ double sh[NQ][4];
    if (MODE == 2 && in_solve && solve_over(st)) return;   // speculative launches past convergence do nothing
    const int nn = *n_list;
    const long C = G.cells();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int q = blockIdx.x; q < nn; q += gridDim.x) {
        const long k = list[q];
        const int nz = (int)(k % G.N) - G.B, ny = (int)((k / G.N) % G.N) - G.B, nx = (int)(k / ((long)G.N * G.N)) - G.B;
        const bool mom = abs(nx) <= G.B - 2 && abs(ny) <= G.B - 2 && abs(nz) <= G.B - 2;
        double acc[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) acc[i] = 0;
        // Which cells can reach this node: base cells n-1 .. n+1 per axis.  For the gradient weights (MODE 1, 2) the node one ABOVE the
        // base cell never carries weight — mspline2(0.5 + c - p) needs c - p < 0.5 and mspline_grad(p - c - 0.5) vanishes at and below
        // -1, while p < base + 0.5 — so only the cells n and n+1 are walked (8 lists instead of 27); the transfer keeps all 27 (its
        // spline evaluates to 2.2e-16, not 0, at the closed end of its support: mpm.cc:25-41)
        constexpr int NC = MODE == 0 ? 27 : 8;
        for (int ci = wv; ci < NC; ci += 4) {
            const int fx = MODE == 0 ? nx + ci / 9 - 1 : nx + (ci >> 2), fy = MODE == 0 ? ny + (ci / 3) % 3 - 1 : ny + ((ci >> 1) & 1),
                      fz = MODE == 0 ? nz + ci % 3 - 1 : nz + (ci & 1);
            if (!G.in(fx, fy, fz)) continue;
            const long c = G.at(fx, fy, fz);
            const int cnt = cell_count[c];
            if (!cnt) continue;
            const long j0 = cell_start[c];
            for (long j = j0 + lane; j < j0 + cnt; j += 64) {
                const double p[3] = {ld(P.pos, P.cap, 0, j), ld(P.pos, P.cap, 1, j), ld(P.pos, P.cap, 2, j)};
                if (!(reaches(G, p[0], nx) && reaches(G, p[1], ny) && reaches(G, p[2], nz))) continue;
                if (MODE == 0) {
                    const double cw = mspline(p[0] - nx) * mspline(p[1] - ny) * mspline(p[2] - nz);
                    if (cw == 0) continue;   // (adds nothing: the velocity loads can stay away)
                    acc[0] += cw > 0 ? cw : 0.0;
                    if (mom) {
                        acc[1] += ld(P.vel, P.cap, 0, j) * cw;
                        acc[2] += ld(P.vel, P.cap, 1, j) * cw;
                        acc[3] += ld(P.vel, P.cap, 2, j) * cw;
                    }
                } else {
                    // getGradW (deformHeader.h:99-103) at this node
                    const double s2x = mspline2(0.5 + nx - p[0]), s2y = mspline2(0.5 + ny - p[1]), s2z = mspline2(0.5 + nz - p[2]);
                    const double gx = mspline_grad(p[0] - nx - 0.5), gy = mspline_grad(p[1] - ny - 0.5), gz = mspline_grad(p[2] - nz - 0.5);
                    if ((s2x == 0 && gx == 0) || (s2y == 0 && gy == 0) || (s2z == 0 && gz == 0)) continue;   // grad w = 0: adds nothing
                    const double g[3] = {-1 * gx * s2y * s2z, -1 * s2x * gy * s2z, -1 * s2x * s2y * gz};
                    double m[9];
#pragma unroll
                    for (int i = 0; i < 9; ++i) m[i] = ld(P.node9, P.cap, i, j);
                    const double sc = ld(P.node9, P.cap, 9, j);
                    const double f = MODE == 1 ? -1 * sc : (transposed ? sc : sc * invm[q]);
#pragma unroll
                    for (int r = 0; r < 3; ++r) acc[r] += f * (m[3 * r] * g[0] + m[3 * r + 1] * g[1] + m[3 * r + 2] * g[2]);
                }
            }
        }
        gather_fold<NQ>(acc, sh);
        if (threadIdx.x == 0) {
            if (MODE == 0) {
                if (acc[0] > 0) out0[k] = acc[0];
                if (mom) out1[k] = acc[1], out1[C + k] = acc[2], out1[2 * C + k] = acc[3];
            } else if (MODE == 1) {
                out0[k] = acc[0], out0[C + k] = acc[1], out0[2 * C + k] = acc[2];
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) out0[3 * (long)q + r] = v[3 * (long)q + r] + acc[r];
            }
        }
    }
}
// The operator's node sums run once per CG iteration.  A particle of base cell f carries gradient weight on the nodes f-1 and f of
// every axis only (k_mpm_gather), so a CELL's particles feed exactly its 8 corner nodes f - 1 + (a, b, c).  One WAVE per non-empty
// cell, lane = particle: a lane reads its particle once (position factors, A_p F_p^T, scale: 22 doubles, coalesced, one round trip
// for the whole cell), forms the 8 x 3 contributions and the wave folds them by a halving butterfly (lane bits 5, 4, 3 against the
// corner bits: 12 + 6 + 3 exchanges, then 9 over the low bits) — fixed order, no LDS, no barrier.  A cell above CELL_HEAVY
// particles is shared by the four waves of its block (their sums folded through LDS in wave order).  The 8 x 3 partial sums of a
// cell go to part[compact cell index]; a node's value is the sum over its 8 cells, taken in a fixed order by the kernel that
// consumes it (k_mpm_cg_pq, k_mpm_apply_combine).
// (Earlier forms: one wave per (node, cell), every particle read 8 times from L2 — 59 us per application on the 219 k particle cone;
// one 512-thread block per cell staging its list through LDS, wave = corner — 21.8 us, three blocks per CU; one wave per cell with
// lane = (slot, corner) striding the list — 23.7 us: a memory round trip per eight particles.)
constexpr int CELL_HEAVY = 128;
constexpr int CELL_WAVES = 4;
__global__ void __launch_bounds__(64 * CELL_WAVES) k_mpm_apply_cells(MGrid G, Part P, const int* __restrict__ cell_start, const int* __restrict__ cell_count,
                                                                     const int* __restrict__ clist, const MpmState* st, int in_solve, int coop, double* __restrict__ part)
{
    __shared__ double sh[CELL_WAVES][8][3];
    if (in_solve && solve_over(st)) return;
    const int nc = st->num_cells;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // the sums of one cell over its particles first + lane, first + lane + stride, ...: afterwards every lane holds the three
    // components of corner (lane >> 3) (bit 5 = a, bit 4 = b, bit 3 = c of node = cell - 1 + (a, b, c))
    auto cell_sum = [&](long c, int first, int stride, double* out) {
        const int fz = (int)(c % G.N) - G.B, fy = (int)((c / G.N) % G.N) - G.B, fx = (int)(c / ((long)G.N * G.N)) - G.B;
        const int cnt = cell_count[c];
        const long j0 = cell_start[c];
        double acc[24];
#pragma unroll
        for (int k = 0; k < 24; ++k) acc[k] = 0;
        for (int i = first + lane; i < cnt; i += stride) {
            const long j = j0 + i;
            double w[12], m[9];
#pragma unroll
            for (int k = 0; k < 12; ++k) w[k] = ld(P.wfac, P.cap, k, j);   // s2 (x0 x1 y0 y1 z0 z1), grad (same order)
#pragma unroll
            for (int k = 0; k < 9; ++k) m[k] = ld(P.node9, P.cap, k, j);
            const double sc = ld(P.node9, P.cap, 9, j);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int ca = s >> 2, cb = (s >> 1) & 1, cc = s & 1;
                const double s2x = w[ca], s2y = w[2 + cb], s2z = w[4 + cc], gx = w[6 + ca], gy = w[8 + cb], gz = w[10 + cc];
                const double g[3] = {-1 * gx * s2y * s2z, -1 * s2x * gy * s2z, -1 * s2x * s2y * gz};   // getGradW, deformHeader.h:99-103
#pragma unroll
                for (int r = 0; r < 3; ++r) acc[3 * s + r] += sc * (m[3 * r] * g[0] + m[3 * r + 1] * g[1] + m[3 * r + 2] * g[2]);
            }
        }
        // halve: a lane keeps the corners whose bit matches its lane bit and takes the partner's sums for them
        double h12[12], h6[6];
        const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const double keep = b5 ? acc[12 + k] : acc[k], send = b5 ? acc[k] : acc[12 + k];
            h12[k] = keep + __shfl_xor(send, 32);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double keep = b4 ? h12[6 + k] : h12[k], send = b4 ? h12[k] : h12[6 + k];
            h6[k] = keep + __shfl_xor(send, 16);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double keep = b3 ? h6[3 + k] : h6[k], send = b3 ? h6[k] : h6[3 + k];
            double v = keep + __shfl_xor(send, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            out[k] = v;
        }
        // a corner node outside the grid takes nothing (the reference's loops stop at the grid's edge)
        const int s = lane >> 3;
        if (!G.in(fx - 1 + (s >> 2), fy - 1 + ((s >> 1) & 1), fz - 1 + (s & 1))) out[0] = out[1] = out[2] = 0;
    };
    // coop (the host's choice when the average cell is heavy: few cells, e.g. the reference's scene of 16 cells x 390): a block per cell
    const int cpb = coop ? 1 : CELL_WAVES;
    for (int base = blockIdx.x * cpb; base < nc; base += gridDim.x * cpb) {
        if (!coop) {
            const int b = base + wv;
            const long c = b < nc ? clist[b] : 0;
            const bool heavy = b < nc && cell_count[c] > CELL_HEAVY;
            if (b < nc && !heavy) {
                double v[3];
                cell_sum(c, 0, 64, v);
                if ((lane & 7) == 0) {
                    double* o = part + (long)b * 24 + 3 * (lane >> 3);
                    o[0] = v[0], o[1] = v[1], o[2] = v[2];
                }
            }
            if (!__syncthreads_or(heavy)) continue;
        }
        for (int w = 0; w < cpb && base + w < nc; ++w) {   // block-uniform
            const long cw = clist[base + w];
            if (!coop && cell_count[cw] <= CELL_HEAVY) continue;
            double v[3];
            cell_sum(cw, wv * 64, 64 * CELL_WAVES, v);
            if ((lane & 7) == 0) sh[wv][lane >> 3][0] = v[0], sh[wv][lane >> 3][1] = v[1], sh[wv][lane >> 3][2] = v[2];
            __syncthreads();
            if (threadIdx.x < 24) {
                const int cr = threadIdx.x / 3, r = threadIdx.x % 3;
                double t = sh[0][cr][r];
#pragma unroll
                for (int k = 1; k < CELL_WAVES; ++k) t += sh[k][cr][r];
                part[(long)(base + w) * 24 + threadIdx.x] = t;
            }
            __syncthreads();
        }
    }
}
// value of unknown q, component r: v + (mass factor) x the partial sums of the 8 cells around the node, cells in a fixed order
__device__ __forceinline__ double apply_combine_one(const MGrid& G, const int* __restrict__ active_cell, const int* __restrict__ cidx,
                                                    const double* __restrict__ part, const double* __restrict__ invm, int transposed,
                                                    const double* __restrict__ v, long t)
{
    const long q = t / 3;
    const int r = (int)(t - 3 * q);
    const long k = active_cell[q];
    const int nz = (int)(k % G.N) - G.B, ny = (int)((k / G.N) % G.N) - G.B, nx = (int)(k / ((long)G.N * G.N)) - G.B;
    double sum = 0;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const int fx = nx + (d >> 2), fy = ny + ((d >> 1) & 1), fz = nz + (d & 1);
        if (!G.in(fx, fy, fz)) continue;
        const int b = cidx[G.at(fx, fy, fz)];
        if (b < 0) continue;
        sum += part[(long)b * 24 + 3 * (7 - d) + r];   // the node is corner (1,1,1) - d of that cell
    }
    return v[t] + (transposed ? sum : sum * invm[q]);
}
__global__ void k_mpm_apply_combine(MGrid G, const MpmState* st, const int* __restrict__ active_cell, const int* __restrict__ cidx,
                                    const double* __restrict__ part, const double* __restrict__ invm, int transposed, const double* __restrict__ v,
                                    double* __restrict__ y)
{
    const long n3 = 3L * st->num_active;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n3; t += (long)gridDim.x * blockDim.x)
        y[t] = apply_combine_one(G, active_cell, cidx, part, invm, transposed, v, t);
}
// non-empty cells (their compact numbering indexes the operator's partial sums) and the fullest cell
__global__ void k_mpm_cell_flags(long C, const int* __restrict__ cell_count, int* __restrict__ flag)
{
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < C) flag[k] = cell_count[k] > 0;
}

// nodes the transfer and the force sums must visit: not solid (mpm.cc:233,620) and within one cell of a particle
__global__ void k_mpm_touched(MGrid G, const uint8_t* __restrict__ solid, const int* __restrict__ cell_count, int* __restrict__ flag)
{
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= G.cells()) return;
    const int z = (int)(k % G.N) - G.B, y = (int)((k / G.N) % G.N) - G.B, x = (int)(k / ((long)G.N * G.N)) - G.B;
    int any = 0;
    if (!solid[k]) {
        for (int a = -1; a <= 1; ++a)
            for (int b = -1; b <= 1; ++b)
                for (int c = -1; c <= 1; ++c)
                    if (G.in(x + a, y + b, z + c)) any |= cell_count[G.at(x + a, y + b, z + c)];
    }
    flag[k] = any ? 1 : 0;
}

// per cell: container = float(mass); vels /= w or 0 (mpm.cc:1000-1015); active flag (mpm.cc:1346-1364); output grid
// (mpm.cc:1366-1380); velBeforeUpdate (mpm.cc:1390)
__global__ void k_mpm_cells(MGrid G, const uint8_t* __restrict__ solid, const double* __restrict__ massd, float* __restrict__ container,
                            float* __restrict__ output, double* __restrict__ vel, double* __restrict__ velb, int* __restrict__ flag)
{
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long C = G.cells();
    if (k >= C) return;
    const float cf = (float)massd[k];
    container[k] = cf;
    const double w = cf;
    const int z = (int)(k % G.N) - G.B, y = (int)((k / G.N) % G.N) - G.B, x = (int)(k / ((long)G.N * G.N)) - G.B;
    const bool act = !solid[k] && abs(x) <= G.W && abs(y) <= G.W && abs(z) <= G.W && w > 0.1;
    flag[k] = act ? 1 : 0;
    if (!solid[k] && w > 0.1) output[k] = (float)w;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double v = w > 0.1 ? vel[a * C + k] / w : 0.0;
        vel[a * C + k] = v;
        velb[a * C + k] = v;
    }
}

// ---- exclusive scan of the active flags in cell order -> indices (mpm.cc:1346-1364) ----
constexpr int SCAN_T = 256, SCAN_PER = 4, SCAN_BLK = SCAN_T * SCAN_PER;
__global__ void __launch_bounds__(SCAN_T) k_mpm_scan_sums(long C, const int* __restrict__ flag, int* __restrict__ sums)
{
    __shared__ int sh[SCAN_T / 64];
    long base = (long)blockIdx.x * SCAN_BLK + (long)threadIdx.x * SCAN_PER;
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k)
        if (base + k < C) s += flag[base + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int k = 0; k < SCAN_T / 64; ++k) t += sh[k];
        sums[blockIdx.x] = t;
    }
}
__global__ void __launch_bounds__(1024) k_mpm_scan_blocks(int nb, int* __restrict__ sums, int* __restrict__ total)
{
    // one block: exclusive scan of the block sums in chunks of 1024
    __shared__ int sh[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        int i = base + threadIdx.x;
        int v = i < nb ? sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            int t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        int incl = sh[threadIdx.x];
        if (i < nb) sums[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = carry;
}
__global__ void __launch_bounds__(SCAN_T) k_mpm_scan_final(long C, const int* __restrict__ flag, const int* __restrict__ sums,
                                                           int* __restrict__ indices, int* __restrict__ active_cell)
{
    __shared__ int sh[SCAN_T];
    long base = (long)blockIdx.x * SCAN_BLK + (long)threadIdx.x * SCAN_PER;
    int f[SCAN_PER], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) {
        f[k] = base + k < C ? flag[base + k] : 0;
        s += f[k];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < SCAN_T; o <<= 1) {
        int t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    int run = sums[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k)
        if (base + k < C) {
            if (f[k]) {
                indices[base + k] = run;
                active_cell[run] = (int)(base + k);
                ++run;
            } else
                indices[base + k] = -1;
        }
}

// ---- particles in base-cell order (counting sort, every step) ----
// The scatter kernels run over `order`: lanes of a wave then share their base cell, so the wave-level aggregation above
// turns thousands of same-address atomics into one per wave and node.  (That was the first version; the sums are node
// gathers over the cell lists now, see k_mpm_gather.)
__global__ void k_mpm_sort_count(MGrid G, long n, Part P, int* __restrict__ cell_count, int* __restrict__ key, int* __restrict__ rank)
{
    // The arrays are nearly sorted already (last step's order, particles move a fraction of a cell per step): the lanes of a
    // wave mostly share their cell, and one atomic per lane on the same counter would serialise 64-fold.  Runs of equal cells
    // take ONE atomic (by their last lane) and number themselves from its result.
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    const int lane = threadIdx.x & 63;
    int k = -1 - lane;
    if (valid) {
        int c[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int f = (int)round(ld(P.pos, P.cap, a, i));
            c[a] = f < -G.B ? -G.B : (f > G.B ? G.B : f);
        }
        k = (int)G.at(c[0], c[1], c[2]);
    }
    const int prev = __shfl_up(k, 1);
    const unsigned long long heads = __ballot(lane == 0 || prev != k);
    const unsigned long long below = heads & (lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1));
    const int start = 63 - __clzll((long long)below);
    const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
    const int tail = above ? lane + __ffsll((long long)above) - 1 : 63;   // last lane of my run
    int base = 0;
    if (valid && lane == tail) base = atomicAdd(&cell_count[k], tail - start + 1);
    base = __shfl(base, tail);
    if (valid) {
        key[i] = k;
        rank[i] = base + (lane - start);
    }
}
__global__ void __launch_bounds__(SCAN_T) k_mpm_scan_excl(long C, const int* __restrict__ vals, const int* __restrict__ sums, int* __restrict__ out)
{
    __shared__ int sh[SCAN_T];
    long base = (long)blockIdx.x * SCAN_BLK + (long)threadIdx.x * SCAN_PER;
    int f[SCAN_PER], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) {
        f[k] = base + k < C ? vals[base + k] : 0;
        s += f[k];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < SCAN_T; o <<= 1) {
        int t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    int run = sums[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k)
        if (base + k < C) {
            out[base + k] = run;
            run += f[k];
        }
}
__global__ void k_mpm_sort_place(long n, const int* __restrict__ key, const int* __restrict__ rank, const int* __restrict__ cell_start,
                                 const int* __restrict__ pid, int* __restrict__ order, int* __restrict__ spid)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d = cell_start[key[i]] + rank[i];
    order[d] = (int)i;
    spid[d] = pid[i];   // the upload indices in (unordered) cell order, contiguous per cell: the rank pass streams them
}
// (inside a cell the lists are then re-ranked by ascending upload index, fl::launch_bin_rank, so that the node gathers add in an order
// that is a function of the input alone)

// The particle arrays themselves are put into that order (double-buffered), so that every per-particle kernel of the step —
// the operator application runs once per CG iteration — reads them coalesced; `pid` keeps the upload index of each slot.
__global__ void k_mpm_permute(long n, long cap, const int* __restrict__ order, Part src, Part dst, const int* __restrict__ pid_src,
                              int* __restrict__ pid_dst)
{
    long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long i = order[j];
#pragma unroll
    for (int k = 0; k < 3; ++k) dst.pos[k * cap + j] = src.pos[k * cap + i], dst.vel[k * cap + j] = src.vel[k * cap + i];
#pragma unroll
    for (int k = 0; k < 9; ++k) dst.FE[k * cap + j] = src.FE[k * cap + i], dst.FP[k * cap + j] = src.FP[k * cap + i];
    dst.volume[j] = src.volume[i];
    pid_dst[j] = pid_src[i];
}
__global__ void k_mpm_iota(long n, int* __restrict__ a)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (int)i;
}

// ---- findVolume, mpm.cc:739-772 ----
__global__ void __launch_bounds__(128) k_mpm_volume(MGrid G, long n, Part P, const uint8_t* __restrict__ solid, const float* __restrict__ container)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double p[3] = {ld(P.pos, P.cap, 0, i), ld(P.pos, P.cap, 1, i), ld(P.pos, P.cap, 2, i)};
    Nbh nb;
    neighbourhood(G, p, nb, false);
    double vol = P.volume[i];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int x = nb.lo[0] + a, y = nb.lo[1] + b, z = nb.lo[2] + c;
                const bool in = x <= nb.hi[0] && y <= nb.hi[1] && z <= nb.hi[2];
                const long k = in ? G.at(x, y, z) : 0;
                if (in && !solid[k]) vol += (double)container[k] * nb.w[0][a] * nb.w[1][b] * nb.w[2][c];
            }
    P.volume[i] = 1.0 / vol;
}

// ---- populateGridForces, first loop (mpm.cc:596-644) + the per-particle part of getdPsydx2 (deformHeader.h:253-263) ----
// per particle: polar factors, stress, the operator's cache; the node sums follow in k_mpm_gather<1>
__global__ void __launch_bounds__(128) k_mpm_forces(MGrid G, long n, Part P, double mu0, double lambda0, double eps)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double F[9], FP[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) F[k] = ld(P.FE, P.cap, k, i), FP[k] = ld(P.FP, P.cap, k, i);
    Setup su;
    particle_setup(F, FP, mu0, lambda0, eps, su);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        stv(P.R, P.cap, k, i, su.R[k]);
        stv(P.Minv, P.cap, k, i, su.Minv[k]);
        stv(P.node9, P.cap, k, i, su.sigma[k]);
    }
    stv(P.coef, P.cap, 0, i, su.mu), stv(P.coef, P.cap, 1, i, su.lambda), stv(P.coef, P.cap, 2, i, su.J);
    stv(P.node9, P.cap, 9, i, P.volume[i]);
    // the gradient-weight factors of the two nodes per axis that carry any (base - 1, base): position only, so once per step — the
    // operator's cell sums read them instead of evaluating six splines per (particle, node) pair
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double pa = ld(P.pos, P.cap, a, i);
        const int f = (int)round(pa);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = f - 1 + k;
            stv(P.wfac, P.cap, 2 * a + k, i, mspline2(0.5 + c - pa));
            stv(P.wfac, P.cap, 6 + 2 * a + k, i, mspline_grad(pa - c - 0.5));
        }
    }
}

// ---- right-hand side and the "Max Force" line, mpm.cc:383-417 ----
__global__ void k_mpm_rhs(MGrid G, const int* __restrict__ active_cell, const MpmState* st, const float* __restrict__ container,
                          const double* __restrict__ vel, const double* __restrict__ forces, double g0, double g1, double g2,
                          double* __restrict__ b, double* __restrict__ invm, double* __restrict__ bb_part, unsigned long long* coeff_bits)
{
    __shared__ double sh[4];
    const int na = st->num_active;
    const double dt = st->dt;
    const long C = G.cells();
    double acc = 0, best = 0;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < na; k += gridDim.x * blockDim.x) {
        const long c = active_cell[k];
        const double mi = container[c];
        invm[k] = 1.0 / mi;
        const double f[3] = {forces[c], forces[C + c], forces[2 * C + c]};
        const double gr[3] = {g0, g1, g2};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double v = vel[a * C + c] + dt * ((1.0 / mi) * f[a] + gr[a]);
            b[3 * (long)k + a] = v;
            acc += v * v;
        }
        double maxf = fmax(fabs(f[0]), fmax(fabs(f[1]), fabs(f[2])));
        best = fmax(best, maxf / mi);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o), best = fmax(best, __shfl_down(best, o));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc, atomic_max_pos(coeff_bits, best);
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int k = 0; k < (int)blockDim.x / 64; ++k) t += sh[k];
        bb_part[blockIdx.x] = t;
    }
}
// first active cell (x-major order) attaining the maximum of maxf / mi: the reference's strict `>` keeps the first
__global__ void k_mpm_maxforce_cell(MGrid G, const int* __restrict__ active_cell, MpmState* st, const float* __restrict__ container,
                                    const double* __restrict__ forces)
{
    const int na = st->num_active;
    const long C = G.cells();
    const double target = __longlong_as_double((long long)st->max_coeff_bits);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < na; k += gridDim.x * blockDim.x) {
        const long c = active_cell[k];
        const double mi = container[c];
        double maxf = fmax(fabs(forces[c]), fmax(fabs(forces[C + c]), fabs(forces[2 * C + c])));
        if (target > 0 && maxf / mi == target) atomicMin(&st->max_cell, (int)c);
    }
}
__global__ void k_mpm_maxforce_final(MGrid G, MpmState* st, const float* __restrict__ container, const double* __restrict__ forces)
{
    const long C = G.cells();
    st->any_active = st->num_active > 0;   // every cell with mass > 0.1 lies inside the walls: same set as the unknowns
    const int c = st->max_cell;
    if (c >= 0 && c < C && st->max_coeff_bits) {
        const double mi = container[c];
        for (int a = 0; a < 3; ++a) st->max_force[a] = st->dt * forces[a * C + c] / mi;
        st->max_mi = mi;
    } else {
        st->max_force[0] = st->max_force[1] = st->max_force[2] = 0, st->max_mi = 0;
    }
}

// ---- the operator: y = v + beta dt^2 D^-1 K v (mpm.cc:646-701 + 418-441, matrix-free) ----
// Two launches per application: per PARTICLE the gather of G = sum_j v_j (x) grad w_j from the unknown vector, one application of
// the energy Hessian and A_p F_p^T (k_mpm_apply_particles, below); per unknown NODE the sum over the particles around it
// (k_mpm_gather<2>).  v, y: 3 * num_active doubles in unknown order.
__global__ void __launch_bounds__(128) k_mpm_apply_particles(MGrid G, long n, Part P, const MpmState* st, double beta, int in_solve, const double* __restrict__ vd)
{
    if (in_solve && solve_over(st)) return;   // speculative launches past convergence do nothing
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // G = sum_j v_j (x) grad w_j over the unknown nodes: of a particle's 27 nodes only the eight base - 1, base per axis carry gradient
    // weight (the third node's factors are exactly 0, see k_mpm_forces), and their factors are in wfac since the forces pass — no
    // spline is evaluated here.  The operand comes as a NODE-indexed array (vd: 3 x cells, the mass factor of the transposed system
    // already applied, 0 on every node that is not an unknown — written by the kernel that made the vector, k_mpm_spread /
    // k_mpm_cg_init / k_mpm_cg_p / k_mpm_cg_xrp): 24 independent loads at fixed offsets from the particle's corner node instead of
    // node -> unknown number -> value.  Nodes in the order of the reference's loops (x outermost).
    int f[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) f[a] = (int)round(ld(P.pos, P.cap, a, i));
    double w[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) w[k] = ld(P.wfac, P.cap, k, i);   // s2 (x0 x1 y0 y1 z0 z1), grad (same order)
    const long C = G.cells();
    double vv[24];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int cx = f[0] - 1 + (s >> 2), cy = f[1] - 1 + ((s >> 1) & 1), cz = f[2] - 1 + (s & 1);
        // a node beyond the grid's edge reads node 0 and is multiplied away (a select would let the compiler put the loads behind branches)
        const bool in = G.in(cx, cy, cz);
        const long k = in ? G.at(cx, cy, cz) : 0;
        const double keep = in ? 1.0 : 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d) vv[3 * s + d] = keep * vd[d * C + k];
    }
    double Gm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int ca = s >> 2, cb = (s >> 1) & 1, cc = s & 1;
        const double g[3] = {-1 * w[6 + ca] * w[2 + cb] * w[4 + cc], -1 * w[ca] * w[8 + cb] * w[4 + cc], -1 * w[ca] * w[2 + cb] * w[10 + cc]};
        const double vx = vv[3 * s], vy = vv[3 * s + 1], vz = vv[3 * s + 2];
#pragma unroll
        for (int d = 0; d < 3; ++d) Gm[d] += vx * g[d], Gm[3 + d] += vy * g[d], Gm[6 + d] += vz * g[d];
    }
    double F[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) F[k] = ld(P.FE, P.cap, k, i);
    double dF[9];
    mat_mul(Gm, F, dF);                         // rows of getDelFE (deformHeader.h:107-132), summed over nodes and directions
    double R[9], Mi[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = ld(P.R, P.cap, k, i);
    // Minv = cof(m) / det(m) of a symmetric m is symmetric to the bit (the mirrored cofactor is the same two products): six loads
    Mi[0] = ld(P.Minv, P.cap, 0, i), Mi[1] = ld(P.Minv, P.cap, 1, i), Mi[2] = ld(P.Minv, P.cap, 2, i), Mi[4] = ld(P.Minv, P.cap, 4, i);
    Mi[5] = ld(P.Minv, P.cap, 5, i), Mi[8] = ld(P.Minv, P.cap, 8, i), Mi[3] = Mi[1], Mi[6] = Mi[2], Mi[7] = Mi[5];
    const double mu = ld(P.coef, P.cap, 0, i), lambda = ld(P.coef, P.cap, 1, i), J = ld(P.coef, P.cap, 2, i);
    double Ap[9];
    hessian_apply(F, R, Mi, mu, lambda, J, dF, Ap);
    double ApFt[9];
    mat_mul_bt(Ap, F, ApFt);                    // A_p F^T
#pragma unroll
    for (int k = 0; k < 9; ++k) stv(P.node9, P.cap, k, i, ApFt[k]);
    const double dt = st->dt;
    stv(P.node9, P.cap, 9, i, beta * dt * dt * P.volume[i]);
}

// ---- CG vector kernels (3 * num_active doubles, weighted dots) ----
// A = I + c D^-1 K is self-adjoint in <u, v> = u^T D v, its transpose I + c K D^-1 in u^T D^-1 v
// the operator's operand by node (k_mpm_apply_particles): vd[d * C + node] = (mass factor) * v[3 q + d] for unknown q at that node
struct Dense {
    double* vd;
    const double* invm;
    long C;
    int transposed;
    // transposed system (what the reference's Eigen solves, see mpm_hip.h): K D^-1 v — the mass divides the input
    __device__ __forceinline__ void put(const int* __restrict__ active_cell, long k, double v) const
    {
        const long q = k / 3;
        vd[(k - 3 * q) * C + active_cell[q]] = (transposed ? invm[q] : 1.0) * v;
    }
};
__device__ __forceinline__ double dot_weight(float mass, int transposed) { return transposed ? 1.0 / (double)mass : (double)mass; }
constexpr int RED_BLOCKS = 256;
__device__ __forceinline__ void block_sum2(double a, double b, double* pa, double* pb)
{
    __shared__ double sh[2][4];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o), b += __shfl_down(b, o);
    if ((threadIdx.x & 63) == 0) sh[0][threadIdx.x >> 6] = a, sh[1][threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0, tb = 0;
        for (int k = 0; k < (int)blockDim.x / 64; ++k) ta += sh[0][k], tb += sh[1][k];
        pa[blockIdx.x] = ta, pb[blockIdx.x] = tb;
    }
}
// every thread of the block gets the sum of n partials, added in the same order in every block
__device__ __forceinline__ double block_total(const double* __restrict__ part, int n)
{
    __shared__ double sh[4];
    double a = 0;
    for (int k = threadIdx.x; k < n; k += blockDim.x) a += part[k];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    __syncthreads();   // sh may still be read from a previous call
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    double t = 0;
    for (int k = 0; k < (int)blockDim.x / 64; ++k) t += sh[k];
    return t;
}
// start of a solve: |b|^2 and <b,b>_w from the partials of k_mpm_cg_init
__global__ void k_mpm_cg_start(int nparts, const double* __restrict__ part1, const double* __restrict__ part2, MpmState* st)
{
    const double a = block_total(part1, nparts), b = block_total(part2, nparts);
    if (threadIdx.x) return;
    st->bb = a, st->rho2[0] = b, st->rr = a;
    st->cg_done = (a == 0) ? 1 : 0;   // b = 0 => x = 0 (IterativeSolverBase / ConjugateGradient.h:44-50)
    st->pend[0] = st->pend[1] = 0;
}
// r = p = b, x = 0, y(=Ap) = p; partials of |b|^2 and <b,b>_w
__global__ void k_mpm_cg_init(const int* __restrict__ active_cell, const MpmState* st, const float* __restrict__ container, int transposed,
                              const double* __restrict__ b, double* __restrict__ x, double* __restrict__ r, double* __restrict__ p,
                              double* __restrict__ q, double* pa, double* pb, Dense dn)
{
    const long n3 = 3L * st->num_active;
    double a = 0, d = 0;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n3; k += (long)gridDim.x * blockDim.x) {
        const double v = b[k], m = dot_weight(container[active_cell[k / 3]], transposed);
        x[k] = 0, r[k] = v, p[k] = v, q[k] = v;
        dn.put(active_cell, k, v);
        a += v * v, d += m * v * v;
    }
    block_sum2(a, d, pa, pb);
}
// One CG iteration = k_mpm_apply + the three kernels below.  Scalars are not folded by 1-thread kernels in between: every block
// of the consumer re-sums the producer's 256 partials from L2 (same order everywhere, so every block takes the same
// decisions), block 0 records them in the state.  `par` = iteration parity: <r,r>_w lives in two slots so that the kernel that
// writes the next one never races with readers of the current one.
// (inside a solve it also forms q = A p from the operator's partial sums per cell: k_mpm_apply_combine folded in, one launch less per iteration)
__global__ void k_mpm_cg_pq(MGrid G, const int* __restrict__ active_cell, const MpmState* st, const float* __restrict__ container, int transposed,
                            const double* __restrict__ p, double* __restrict__ q, double* pa, double* pb, const int* __restrict__ cidx,
                            const double* __restrict__ apart, const double* __restrict__ invm)
{
    const long n3 = solve_over(st) ? 0 : 3L * st->num_active;
    double d = 0;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n3; k += (long)gridDim.x * blockDim.x) {
        const double qk = apply_combine_one(G, active_cell, cidx, apart, invm, transposed, p, k);
        q[k] = qk;
        d += dot_weight(container[active_cell[k / 3]], transposed) * p[k] * qk;
    }
    block_sum2(0.0, d, pa, pb);
}
// alpha = <r,r>_w / <p,Ap>_w; x += alpha p; r -= alpha q; partials |r|^2, <r,r>_w
__global__ void k_mpm_cg_xr(const int* __restrict__ active_cell, MpmState* st, const float* __restrict__ container, int transposed, int par,
                            const double* __restrict__ part_pq, double* __restrict__ x, double* __restrict__ r, const double* __restrict__ p,
                            const double* __restrict__ q, double* pa, double* pb)
{
    __shared__ int s_done;   // one read per block: block 0 of this very launch may set the flag while other blocks start
    if (threadIdx.x == 0) s_done = st->cg_done;
    __syncthreads();
    if (s_done) return;
    const double pq = block_total(part_pq, RED_BLOCKS);
    if (!(pq > 0) || !isfinite(pq)) {   // breakdown: leave x as it is (every block sees the same value)
        if (blockIdx.x == 0 && threadIdx.x == 0) st->pq = pq, st->cg_done = 2;
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) st->pq = pq;
    const long n3 = 3L * st->num_active;
    const double alpha = st->rho2[par] / pq;
    double a = 0, d = 0;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n3; k += (long)gridDim.x * blockDim.x) {
        x[k] += alpha * p[k];
        const double rv = r[k] - alpha * q[k];
        r[k] = rv;
        a += rv * rv, d += dot_weight(container[active_cell[k / 3]], transposed) * rv * rv;
    }
    block_sum2(a, d, pa, pb);
}
// convergence test (ConjugateGradient.h:76-79), then p = r + beta p and q = p (the identity part of the next product)
__global__ void k_mpm_cg_p(MpmState* st, int par, double tol, long max_iters, const double* __restrict__ part_rr, const double* __restrict__ part_rho,
                           const double* __restrict__ r, double* __restrict__ p, double* __restrict__ q, const int* __restrict__ active_cell, Dense dn)
{
    __shared__ int s_done;
    if (threadIdx.x == 0) s_done = st->cg_done;
    __syncthreads();
    if (s_done) return;
    const double rr = block_total(part_rr, RED_BLOCKS), rho_new = block_total(part_rho, RED_BLOCKS);
    const bool conv = rr < tol * tol * st->bb;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->rr = rr, st->rho2[par ^ 1] = rho_new;
        if (conv) st->cg_done = 1;                       // Eigen leaves the loop before counting the iteration
        else if (++st->cg_iters >= max_iters) st->cg_done = 3;
    }
    if (conv) return;
    const long n3 = 3L * st->num_active;
    const double beta = rho_new / st->rho2[par];
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n3; k += (long)gridDim.x * blockDim.x) {
        const double v = r[k] + beta * p[k];
        p[k] = v, q[k] = v;
        dn.put(active_cell, k, v);
    }
}
// k_mpm_cg_xr and k_mpm_cg_p as ONE launch for small systems (the reference's scene has 39 values, the scaled cone 9 906): the vector
// is short enough for EVERY block to form the whole new residual's two norms itself — the same order in every block, so all of them
// take the same decisions — instead of handing partials to a third launch; a block then writes only its own share of x, r, p.  The new
// residual goes into a second buffer (other blocks still read the old one in their reduction).
constexpr long MPM_XRP_MAX = 2048;   // (9 906 values, the scaled cone: 64 blocks each reducing the whole vector cost more than the launch they save: solve 2.0 -> 2.7 ms)
__global__ void k_mpm_cg_xrp(const int* __restrict__ active_cell, MpmState* st, const float* __restrict__ container, int transposed, int par, double tol,
                             long max_iters, long launch, const double* __restrict__ part_pq, double* __restrict__ x, const double* __restrict__ r, double* __restrict__ r_new,
                             double* __restrict__ p, const double* __restrict__ q, Dense dn)
{
    __shared__ int s_done;
    __shared__ double sh[2][4];
    if (threadIdx.x == 0) {
        const int mine = (int)((launch + 1) << 2), a0 = st->pend[0], a1 = st->pend[1];   // a mark of THIS launch (block 0 may be ahead of us) does not count
        s_done = st->cg_done | ((a0 & ~3) != mine ? a0 : 0) | ((a1 & ~3) != mine ? a1 : 0);
    }
    __syncthreads();
    if (s_done) return;
    const double pq = block_total(part_pq, RED_BLOCKS);
    if (!(pq > 0) || !isfinite(pq)) {   // breakdown: leave x as it is (every block sees the same value)
        if (blockIdx.x == 0 && threadIdx.x == 0) st->pq = pq, st->cg_done = 2;
        return;
    }
    const long n3 = 3L * st->num_active;
    const double alpha = st->rho2[par] / pq;
    double a = 0, d = 0;
    for (long k = threadIdx.x; k < n3; k += blockDim.x) {
        const double rv = r[k] - alpha * q[k];
        a += rv * rv, d += dot_weight(container[active_cell[k / 3]], transposed) * rv * rv;
    }
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o), d += __shfl_xor(d, o);
    if ((threadIdx.x & 63) == 0) sh[0][threadIdx.x >> 6] = a, sh[1][threadIdx.x >> 6] = d;
    __syncthreads();
    double rr = 0, rho_new = 0;
    for (int k = 0; k < (int)blockDim.x / 64; ++k) rr += sh[0][k], rho_new += sh[1][k];
    const bool conv = rr < tol * tol * st->bb;
    const double beta = rho_new / st->rho2[par];
    __syncthreads();   // every thread has read the state before block 0 changes it
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->pq = pq, st->rr = rr, st->rho2[par ^ 1] = rho_new;
        if (conv) st->pend[par ^ 1] = (int)((launch + 1) << 2) | 1;                 // Eigen leaves the loop before counting the iteration
        else if (++st->cg_iters >= max_iters) st->pend[par ^ 1] = (int)((launch + 1) << 2) | 3;
    }
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n3; k += (long)gridDim.x * blockDim.x) {
        const double pk = p[k], rv = r[k] - alpha * q[k];
        x[k] += alpha * pk;
        r_new[k] = rv;
        if (!conv) {
            const double v = rv + beta * pk;
            p[k] = v;
            dn.put(active_cell, k, v);
        }
    }
}
// the operand of one application outside a solve (mpm_apply_matrix) / the operand's nodes back to 0 once the unknowns are done with
__global__ void k_mpm_spread(const MpmState* st, const int* __restrict__ active_cell, const double* __restrict__ v, Dense dn, int clear)
{
    const long n3 = 3L * st->num_active;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n3; k += (long)gridDim.x * blockDim.x) {
        if (clear) dn.vd[(k % 3) * dn.C + active_cell[k / 3]] = 0;
        else dn.put(active_cell, k, v[k]);
    }
}
__global__ void k_mpm_copy(long n, const double* __restrict__ a, double* __restrict__ b)
{
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) b[k] = a[k];
}

// ---- updateVelocity, mpm.cc:705-737 ----
__global__ void k_mpm_update_velocity(MGrid G, const uint8_t* __restrict__ solid, const float* __restrict__ container,
                                      const int* __restrict__ indices, const double* __restrict__ x, double* __restrict__ vel)
{
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long C = G.cells();
    if (k >= C || solid[k]) return;
    const int idx = indices[k];
    const bool on = (double)container[k] > 0.1 && idx >= 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) vel[a * C + k] = on ? x[3 * (long)idx + a] : 0.0;
}

// ---- updateDeformationGradient, mpm.cc:493-586 ----
__global__ void __launch_bounds__(128) k_mpm_deform(MGrid G, long n, Part P, const uint8_t* __restrict__ solid, const double* __restrict__ vel, MpmState* st,
                             double minv, double maxv)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double t = 0, t2 = 0, t3 = 0;
    if (i < n) {
    const long C = G.cells();
    double p[3] = {ld(P.pos, P.cap, 0, i), ld(P.pos, P.cap, 1, i), ld(P.pos, P.cap, 2, i)};
    Nbh nb;
    neighbourhood(G, p, nb, true);
    double gv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int x = nb.lo[0] + a, y = nb.lo[1] + b, z = nb.lo[2] + c;
                const bool in = x <= nb.hi[0] && y <= nb.hi[1] && z <= nb.hi[2];
                const long k = in ? G.at(x, y, z) : 0;
                const bool ok = in && !solid[k];
                double g[3];
                grad_w(nb, a, b, c, g);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double vr = ok ? vel[r * C + k] : 0.0;
#pragma unroll
                    for (int d = 0; d < 3; ++d) gv[3 * r + d] += vr * g[d];
                }
            }
    const double dt = st->dt;
    double FE[9], FP[9], A[9], tFE[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        stv(P.gradV, P.cap, k, i, gv[k]);
        FE[k] = ld(P.FE, P.cap, k, i), FP[k] = ld(P.FP, P.cap, k, i);
        A[k] = ((k % 4 == 0) ? 1.0 : 0.0) + dt * gv[k];
    }
    mat_mul(A, FE, tFE);
    double nFE[9], nFP[9];
    clamp_update(tFE, FP, minv, maxv, nFE, nFP);
    double mx = gv[0], mn = gv[0];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        stv(P.FE, P.cap, k, i, nFE[k]), stv(P.FP, P.cap, k, i, nFP[k]);
        mx = fmax(mx, gv[k]), mn = fmin(mn, gv[k]);
    }
    t = fmax(mx, -1 * mn), t2 = mat_det(nFP), t3 = mat_det(nFE);
    }
    // wave-level maxima (the reference's maxima start at 0: negative values never show), one atomic per wave
    for (int o = 32; o > 0; o >>= 1) t = fmax(t, __shfl_xor(t, o)), t2 = fmax(t2, __shfl_xor(t2, o)), t3 = fmax(t3, __shfl_xor(t3, o));
    if ((threadIdx.x & 63) == 0) atomic_max_pos(&st->max_grad_bits, t), atomic_max_pos(&st->max_fp_bits, t2), atomic_max_pos(&st->max_fe_bits, t3);
}


// ---- FLIPadvect, mpm.cc:906-969 with CatmullRomFLIP mpm.cc:163-216 and getVelocity mpm.cc:64-76 ----
__device__ __forceinline__ double vel_at(const MGrid& G, const double* __restrict__ g, long C, int a, int x, int y, int z)
{
    return G.in(x, y, z) ? g[a * C + G.at(x, y, z)] : 0.0;   // reads beyond the grid return the background 0
}
__global__ void __launch_bounds__(128) k_mpm_flip(MGrid G, long n, Part P, const double* __restrict__ vel, const double* __restrict__ velb, MpmState* st)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double len = 0;
    if (i < n) {
        const long C = G.cells();
        double p[3] = {ld(P.pos, P.cap, 0, i), ld(P.pos, P.cap, 1, i), ld(P.pos, P.cap, 2, i)};
        Nbh nb;
        neighbourhood(G, p, nb, false);
        double weight = 0, delta[3] = {0, 0, 0};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int x = nb.lo[0] + a, y = nb.lo[1] + b, z = nb.lo[2] + c;
                    if (x > nb.hi[0] || y > nb.hi[1] || z > nb.hi[2]) continue;
                    if (abs(x) > G.W || abs(y) > G.W || abs(z) > G.W) continue;
                    const double cw = nb.w[0][a] * nb.w[1][b] * nb.w[2][c];
                    const double vc[3] = {(vel_at(G, vel, C, 0, x, y, z) + vel_at(G, vel, C, 0, x + 1, y, z)) / 2.0,
                                          (vel_at(G, vel, C, 1, x, y, z) + vel_at(G, vel, C, 1, x, y + 1, z)) / 2.0,
                                          (vel_at(G, vel, C, 2, x, y, z) + vel_at(G, vel, C, 2, x, y, z + 1)) / 2.0};
                    const double vp[3] = {(vel_at(G, velb, C, 0, x, y, z) + vel_at(G, velb, C, 0, x + 1, y, z)) / 2.0,
                                          (vel_at(G, velb, C, 1, x, y, z) + vel_at(G, velb, C, 1, x, y + 1, z)) / 2.0,
                                          (vel_at(G, velb, C, 2, x, y, z) + vel_at(G, velb, C, 2, x, y, z + 1)) / 2.0};
                    weight += cw;
#pragma unroll
                    for (int d = 0; d < 3; ++d) delta[d] += (vc[d] - vp[d]) * cw;
                }
        double v[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            v[d] = ld(P.vel, P.cap, d, i) + (weight == 0 ? 0.0 : delta[d] / weight);
            stv(P.vel, P.cap, d, i, v[d]);
        }
        len = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    }
    for (int o = 32; o > 0; o >>= 1) len = fmax(len, __shfl_xor(len, o));
    if ((threadIdx.x & 63) == 0) atomic_max_pos(&st->max_speed_bits, len);
}
__global__ void k_mpm_timestep(MpmState* st, double max_dt, double dx)
{
    const double ms = __longlong_as_double((long long)st->max_speed_bits);
    st->dt = ms != 0 ? (max_dt < dx / ms ? max_dt : dx / ms) : max_dt;   // mpm.cc:929-936
}
__global__ void __launch_bounds__(128) k_mpm_advect(MGrid G, long n, Part P, const uint8_t* __restrict__ solid, const MpmState* st)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ts = st->dt, e = 0.0;
    double p[3], v[3], q[3];
    int r[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        p[a] = ld(P.pos, P.cap, a, i), v[a] = ld(P.vel, P.cap, a, i);
        q[a] = p[a] + ts * v[a];
        r[a] = (int)(q[a] > 0 ? ceil(q[a]) : floor(q[a]));
    }
    auto is_solid = [&](int x, int y, int z) { return G.in(x, y, z) && solid[G.at(x, y, z)] != 0; };
    if (is_solid(r[0], r[1], r[2])) {
        // Coord(int, double, double): the doubles become Int32 by truncation (math/Coord.h:61)
        if (is_solid(r[0], (int)p[1], (int)p[2])) v[0] *= -1.0 * e;
        if (is_solid((int)p[0], r[1], (int)p[2])) v[1] *= -1.0 * e;
        if (is_solid((int)p[0], (int)p[1], r[2])) v[2] *= -1.0 * e;
#pragma unroll
        for (int a = 0; a < 3; ++a) stv(P.vel, P.cap, a, i, v[a]), stv(P.pos, P.cap, a, i, p[a] + v[a] * ts);
    } else {
#pragma unroll
        for (int a = 0; a < 3; ++a) stv(P.pos, P.cap, a, i, q[a]);
    }
}

// AoS <-> SoA of `w` doubles per particle
// host arrays are in upload order: slot j holds the particle uploaded as number pid[j]
__global__ void k_mpm_to_soa(long n, long cap, int w, const int* __restrict__ pid, const double* __restrict__ aos, double* __restrict__ soa)
{
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n * w) soa[(k % w) * cap + k / w] = aos[(long)pid[k / w] * w + k % w];
}
__global__ void k_mpm_to_aos(long n, long cap, int w, const int* __restrict__ pid, const double* __restrict__ soa, double* __restrict__ aos)
{
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n * w) aos[(long)pid[k / w] * w + k % w] = soa[(k % w) * cap + k / w];
}
__global__ void k_mpm_identity(long n, long cap, double* __restrict__ FE, double* __restrict__ FP, double* __restrict__ volume)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < 9; ++k) FE[k * cap + i] = FP[k * cap + i] = (k % 4 == 0) ? 1.0 : 0.0;
    volume[i] = 0.0;
}
__global__ void k_mpm_interleave(long C, const double* __restrict__ soa, double* __restrict__ aos)
{
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < C) aos[3 * k] = soa[k], aos[3 * k + 1] = soa[C + k], aos[3 * k + 2] = soa[2 * C + k];
}

inline unsigned blocks_for(long n, int t) { return (unsigned)((n + t - 1) / t); }
constexpr int GATHER_BLOCKS = 2048;   // workgroups of a node gather (each strides the node list: its length stays on the device)
inline double bits_to_double(unsigned long long b)
{
    double d;
    memcpy(&d, &b, 8);
    return d;
}

}  // namespace

struct mpm_sim {
    mpm_params_t prm;
    MGrid G;
    hipStream_t st = nullptr;
    long C = 0, n = 0;
    Part P{}, P2{};               // P2: the other buffer of pos, vel, FE, FP, volume (the step starts by sorting P into it)
    int *pid = nullptr, *pid2 = nullptr;   // upload index of each particle slot
    uint8_t* solid = nullptr;
    float *container = nullptr, *output = nullptr;
    double *massd = nullptr, *vel = nullptr, *velb = nullptr, *forces = nullptr;
    int *flag = nullptr, *indices = nullptr, *active_cell = nullptr, *sums = nullptr;
    int *cell_count = nullptr, *cell_start = nullptr, *key = nullptr, *rank = nullptr, *order = nullptr, *order2 = nullptr, *rank2 = nullptr;   // counting sort by base cell
    int *tflag = nullptr, *tidx = nullptr, *tlist = nullptr;   // nodes within one cell of a particle: flags, scratch numbering, list
    double *b = nullptr, *x = nullptr, *r = nullptr, *p = nullptr, *q = nullptr, *part = nullptr;
    double* r2 = nullptr;   // the residual's second buffer (k_mpm_cg_xrp writes the new residual beside the one other blocks still read)
    double* invm = nullptr;       // 1 / node mass per unknown
    double* vdense = nullptr;     // the operator's operand by node (3 x cells, 0 off the unknowns), see Dense
    double* apart = nullptr;      // the operator's partial node sums (32 x 3 per unknown), grown with the unknown count
    size_t apart_cap = 0;
    int *cflag = nullptr, *cidx = nullptr, *clist = nullptr;   // non-empty cells: flags, compact index per cell (-1: empty), list
    int num_cells = 0;
    double* stage = nullptr;
    size_t stage_bytes = 0;
    MpmState* state = nullptr;
    MpmState* h_state = nullptr;   // pinned
    int step_no = 0, num_active = 0;
    long last_iters = 0;          // CG iterations of the previous step's solve (sizes the first unpolled batch)
    bool mid_step = false;
    mpm_step_stats_t stats{};
    double dt = 0.001;
    hipEvent_t ev[8] = {};
};

namespace {

template <typename T>
int dalloc(T** p, size_t n)
{
    HIPCHK(hipMalloc((void**)p, (n ? n : 1) * sizeof(T)));
    HIPCHK(hipMemset(*p, 0, (n ? n : 1) * sizeof(T)));
    return 0;
}
int ensure_stage(mpm_sim* s, size_t bytes)
{
    if (bytes <= s->stage_bytes) return 0;
    if (s->stage) HIPCHK(hipFree(s->stage));
    s->stage = nullptr, s->stage_bytes = 0;
    HIPCHK(hipMalloc((void**)&s->stage, bytes));
    s->stage_bytes = bytes;
    return 0;
}
void free_particles(mpm_sim* s)
{
    int** ia[] = {&s->key, &s->rank, &s->rank2, &s->order, &s->order2, &s->pid, &s->pid2};
    for (auto p : ia) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    double** a[] = {&s->P.pos, &s->P.vel, &s->P.FE, &s->P.FP, &s->P.gradV, &s->P.volume, &s->P.R, &s->P.Minv, &s->P.coef, &s->P.node9, &s->P.wfac,
                    &s->P2.pos, &s->P2.vel, &s->P2.FE, &s->P2.FP, &s->P2.volume};
    for (auto p : a) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    s->P.cap = s->P2.cap = 0;
}
int alloc_particles(mpm_sim* s, long cap)
{
    free_particles(s);
    s->P.cap = s->P2.cap = cap;
    if (dalloc(&s->P.pos, 3 * cap) || dalloc(&s->P.vel, 3 * cap) || dalloc(&s->P.FE, 9 * cap) || dalloc(&s->P.FP, 9 * cap) ||
        dalloc(&s->P.gradV, 9 * cap) || dalloc(&s->P.volume, cap) || dalloc(&s->P.R, 9 * cap) || dalloc(&s->P.Minv, 9 * cap) ||
        dalloc(&s->P.coef, 3 * cap) || dalloc(&s->P.node9, 10 * cap) || dalloc(&s->P.wfac, 12 * cap) || dalloc(&s->key, cap) || dalloc(&s->rank, cap) ||
        dalloc(&s->order, cap) || dalloc(&s->order2, cap) || dalloc(&s->rank2, cap) ||
        dalloc(&s->pid, cap) || dalloc(&s->pid2, cap) || dalloc(&s->P2.pos, 3 * cap) || dalloc(&s->P2.vel, 3 * cap) || dalloc(&s->P2.FE, 9 * cap) ||
        dalloc(&s->P2.FP, 9 * cap) || dalloc(&s->P2.volume, cap))
        return FLUID_ERR_HIP;
    HIPCHK(hipDeviceSynchronize());   // dalloc's hipMemset runs on the null stream, which the handle's non-blocking stream does not wait for
    return 0;
}

// y = A v on the device (v in s->p, result in s->q)
int apply_operator(mpm_sim* s, int in_solve)
{
    if (s->n) {
        const Dense dn{s->vdense, s->invm, s->C, s->prm.transpose_system};
        const unsigned ub = blocks_for(3L * std::max(s->num_active, 1), 256);
        if (!in_solve) k_mpm_spread<<<ub, 256, 0, s->st>>>(s->state, s->active_cell, s->p, dn, 0);
        k_mpm_apply_particles<<<blocks_for(s->n, 128), 128, 0, s->st>>>(s->G, s->n, s->P, s->state, s->prm.beta, in_solve, s->vdense);
        if (!in_solve) k_mpm_spread<<<ub, 256, 0, s->st>>>(s->state, s->active_cell, s->p, dn, 1);
        const int coop = s->n > (long)std::max(s->num_cells, 1) * CELL_HEAVY;
        const long cblocks = coop ? s->num_cells : (s->num_cells + CELL_WAVES - 1) / CELL_WAVES;
        k_mpm_apply_cells<<<(unsigned)std::min<long>(std::max<long>(cblocks, 1), 8192), 64 * CELL_WAVES, 0, s->st>>>(s->G, s->P, s->cell_start, s->cell_count, s->clist,
                                                                                                                   s->state, in_solve, coop, s->apart);
        // inside a solve the next kernel (k_mpm_cg_pq) forms q from the partial sums itself
        if (!in_solve)
            k_mpm_apply_combine<<<blocks_for(3L * std::max(s->num_active, 1), 256), 256, 0, s->st>>>(s->G, s->state, s->active_cell, s->cidx, s->apart, s->invm,
                                                                                                 s->prm.transpose_system, s->p, s->q);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int read_state(mpm_sim* s)
{
    HIPCHK(hipMemcpyAsync(s->h_state, s->state, sizeof(MpmState), hipMemcpyDeviceToHost, s->st));
    HIPCHK(hipStreamSynchronize(s->st));
    if (!s->h_state->cg_done) s->h_state->cg_done = std::max(s->h_state->pend[0] & 3, s->h_state->pend[1] & 3);   // (k_mpm_cg_xrp's end-of-solve slots)
    return 0;
}

}  // namespace

extern "C" {

int mpm_default_params(mpm_params_t* p)
{
    if (!p) return fluid_fail(FLUID_ERR_ARG, "mpm_default_params: null");
    memset(p, 0, sizeof *p);
    p->B = 15, p->W = 13, p->device = 0, p->cg_max_iters = 0;
    p->dx = 1.0, p->gravity[0] = 0, p->gravity[1] = -10, p->gravity[2] = 0;
    p->youngs_modulus = 48000, p->poisson_ratio = 0.47, p->beta = 0.5, p->hardening = 10;
    p->theta_c = 0.025, p->theta_s = 0.0075, p->max_dt = 0.001, p->dt0 = 0.001;
    p->cg_tol = 2.220446049250313e-16;
    p->transpose_system = 1;
    return 0;
}

int mpm_create(const mpm_params_t* prm, mpm_sim_t** out)
{
    if (!prm || !out) return fluid_fail(FLUID_ERR_ARG, "mpm_create: null argument");
    if (prm->B < 3 || prm->B > 400 || prm->W < 1 || prm->W > prm->B) return fluid_fail(FLUID_ERR_ARG, "mpm_create: need 3 <= B <= 400 and 1 <= W <= B");
    if (!(prm->dx > 0) || !(prm->max_dt > 0) || !(prm->dt0 > 0)) return fluid_fail(FLUID_ERR_ARG, "mpm_create: dx, max_dt, dt0 must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fluid_fail(FLUID_ERR_HIP, "mpm_create: no HIP device (this library has no CPU path)");
    if (prm->device < 0 || prm->device >= ndev) return fluid_fail(FLUID_ERR_ARG, "mpm_create: bad device ordinal");
    HIPCHK(hipSetDevice(prm->device));
    mpm_sim* s = new mpm_sim;
    s->prm = *prm;
    s->G = MGrid{prm->B, prm->W, 2 * prm->B + 1};
    s->C = s->G.cells();
    s->dt = prm->dt0;
    const long C = s->C;
    const int nb = (int)blocks_for(C, SCAN_BLK);
    int rc = 0;
    if (hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking) != hipSuccess) rc = FLUID_ERR_HIP;
    rc = rc || dalloc(&s->solid, C) || dalloc(&s->container, C) || dalloc(&s->output, C) || dalloc(&s->massd, C) || dalloc(&s->vel, 3 * C) ||
         dalloc(&s->velb, 3 * C) || dalloc(&s->forces, 3 * C) || dalloc(&s->flag, C) || dalloc(&s->indices, C) || dalloc(&s->active_cell, C) ||
         dalloc(&s->sums, nb) || dalloc(&s->cell_count, C) || dalloc(&s->cell_start, C + 1) || dalloc(&s->tflag, C) || dalloc(&s->tidx, C) || dalloc(&s->tlist, C) || dalloc(&s->cflag, C) || dalloc(&s->cidx, C) || dalloc(&s->clist, C) || dalloc(&s->part, 4 * RED_BLOCKS) || dalloc(&s->state, 1);
    // unknowns live inside the walls only: (2W+1)^3 at most
    const long maxu = (long)(2 * prm->W + 1) * (2 * prm->W + 1) * (2 * prm->W + 1);
    rc = rc || dalloc(&s->b, 3 * maxu) || dalloc(&s->x, 3 * maxu) || dalloc(&s->r, 3 * maxu) || dalloc(&s->r2, 3 * maxu) || dalloc(&s->p, 3 * maxu) || dalloc(&s->q, 3 * maxu) || dalloc(&s->invm, maxu) || dalloc(&s->vdense, 3 * C);
    if (!rc && hipHostMalloc((void**)&s->h_state, sizeof(MpmState)) != hipSuccess) rc = FLUID_ERR_HIP;
    for (int k = 0; k < 8 && !rc; ++k)
        if (hipEventCreate(&s->ev[k]) != hipSuccess) rc = FLUID_ERR_HIP;
    if (rc) {
        mpm_destroy(s);
        return fluid_fail(FLUID_ERR_HIP, "mpm_create: device allocation failed");
    }
    std::vector<uint8_t> h(C);
    const MGrid& G = s->G;
    for (int x = -G.B; x <= G.B; ++x)
        for (int y = -G.B; y <= G.B; ++y)
            for (int z = -G.B; z <= G.B; ++z) h[G.at(x, y, z)] = (std::abs(x) > G.W || std::abs(y) > G.W || std::abs(z) > G.W) ? 1 : 0;
    HIPCHK(hipMemcpy(s->solid, h.data(), C, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(s->indices, 0xFF, C * sizeof(int)));
    HIPCHK(hipDeviceSynchronize());   // the fills above ran on the null stream; s->st is non-blocking
    *out = s;
    return 0;
}

int mpm_destroy(mpm_sim_t* s)
{
    if (!s) return 0;
    free_particles(s);
    void* a[] = {s->solid, s->container, s->output, s->massd, s->vel, s->velb, s->forces, s->flag, s->indices, s->active_cell, s->sums,
                 s->part, s->state, s->b, s->x, s->r, s->r2, s->p, s->q, s->stage, s->cell_count, s->cell_start, s->invm, s->tflag, s->tidx, s->tlist, s->apart, s->cflag, s->cidx, s->clist, s->vdense};
    for (void* p : a)
        if (p) (void)hipFree(p);
    if (s->h_state) (void)hipHostFree(s->h_state);
    for (int k = 0; k < 8; ++k)
        if (s->ev[k]) (void)hipEventDestroy(s->ev[k]);
    if (s->st) (void)hipStreamDestroy(s->st);
    delete s;
    return 0;
}

int mpm_upload_particles(mpm_sim_t* s, int64_t n, const double* pos, const double* vel, int64_t* kept)
{
    if (!s || n < 0 || (n && !pos)) return fluid_fail(FLUID_ERR_ARG, "mpm_upload_particles: bad argument");
    // PointList::add, mpm.cc:471-491
    std::vector<double> hp, hv;
    hp.reserve(3 * (size_t)n), hv.reserve(3 * (size_t)n);
    const int bd = s->G.B - 2;
    for (int64_t i = 0; i < n; ++i) {
        const double* p = pos + 3 * i;
        if (std::fabs(p[0]) < bd && std::fabs(p[1]) < bd && std::fabs(p[2]) < bd) {
            hp.insert(hp.end(), p, p + 3);
            if (vel)
                hv.insert(hv.end(), vel + 3 * i, vel + 3 * i + 3);
            else
                hv.push_back(0.0), hv.push_back(-50.0), hv.push_back(0.0);
        }
    }
    const long m = (long)hp.size() / 3;
    if (m > s->P.cap || !s->P.pos)
        if (alloc_particles(s, m > 64 ? m : 64)) return FLUID_ERR_HIP;
    s->n = m;
    if (kept) *kept = m;
    s->step_no = 0;
    s->mid_step = false;
    s->dt = s->prm.dt0;
    if (!m) return 0;
    if (ensure_stage(s, sizeof(double) * 3 * m)) return FLUID_ERR_HIP;
    k_mpm_iota<<<blocks_for(m, 256), 256, 0, s->st>>>(m, s->pid);
    HIPCHK(hipMemcpyAsync(s->stage, hp.data(), sizeof(double) * 3 * m, hipMemcpyHostToDevice, s->st));
    k_mpm_to_soa<<<blocks_for(3 * m, 256), 256, 0, s->st>>>(m, s->P.cap, 3, s->pid, s->stage, s->P.pos);
    HIPCHK(hipStreamSynchronize(s->st));
    HIPCHK(hipMemcpyAsync(s->stage, hv.data(), sizeof(double) * 3 * m, hipMemcpyHostToDevice, s->st));
    k_mpm_to_soa<<<blocks_for(3 * m, 256), 256, 0, s->st>>>(m, s->P.cap, 3, s->pid, s->stage, s->P.vel);
    k_mpm_identity<<<blocks_for(m, 256), 256, 0, s->st>>>(m, s->P.cap, s->P.FE, s->P.FP, s->P.volume);
    HIPCHK(hipMemsetAsync(s->P.gradV, 0, sizeof(double) * 9 * s->P.cap, s->st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->st));
    return 0;
}

int64_t mpm_num_particles(const mpm_sim_t* s) { return s ? s->n : -1; }
int32_t mpm_num_active(const mpm_sim_t* s) { return s ? s->num_active : -1; }

int mpm_set_state(mpm_sim_t* s, const double* FE, const double* FP, const double* volume, int32_t step_no)
{
    if (!s || step_no < 0) return fluid_fail(FLUID_ERR_ARG, "mpm_set_state: bad argument");
    if (s->mid_step) return fluid_fail(FLUID_ERR_STATE, "mpm_set_state: between mpm_step_solve and mpm_step_advance");
    const long m = s->n;
    const double* src[3] = {FE, FP, volume};
    double* dst[3] = {s->P.FE, s->P.FP, s->P.volume};
    const int w[3] = {9, 9, 1};
    for (int k = 0; k < 3; ++k) {
        if (!src[k] || !m) continue;
        if (ensure_stage(s, sizeof(double) * w[k] * m)) return FLUID_ERR_HIP;
        HIPCHK(hipMemcpyAsync(s->stage, src[k], sizeof(double) * w[k] * m, hipMemcpyHostToDevice, s->st));
        k_mpm_to_soa<<<blocks_for((long)w[k] * m, 256), 256, 0, s->st>>>(m, s->P.cap, w[k], s->pid, s->stage, dst[k]);
        HIPCHK(hipStreamSynchronize(s->st));
    }
    s->step_no = step_no;
    return 0;
}

int mpm_set_dt(mpm_sim_t* s, double dt)
{
    if (!s || !(dt > 0)) return fluid_fail(FLUID_ERR_ARG, "mpm_set_dt: bad argument");
    s->dt = dt;
    return 0;
}
double mpm_get_dt(const mpm_sim_t* s) { return s ? s->dt : 0.0; }

// mpm.cc:1313-1404: transfer, numbering, forces, right-hand side, solve, updateVelocity
int mpm_step_solve(mpm_sim_t* s, mpm_step_stats_t* out)
{
    if (!s) return fluid_fail(FLUID_ERR_ARG, "mpm_step_solve: null handle");
    if (s->mid_step) return fluid_fail(FLUID_ERR_STATE, "mpm_step_solve: the previous step was not advanced");
    const MGrid G = s->G;
    const long C = s->C, n = s->n;
    const mpm_params_t& pr = s->prm;
    hipStream_t st = s->st;
    const unsigned pb = blocks_for(n, 128), cb = blocks_for(C, 256);
    const int nb = (int)blocks_for(C, SCAN_BLK);

    // step scalars
    memset(s->h_state, 0, sizeof(MpmState));
    s->h_state->dt = s->dt;
    s->h_state->max_cell = 0x7fffffff;
    HIPCHK(hipMemcpyAsync(s->state, s->h_state, sizeof(MpmState), hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(s->ev[0], st));
    // mpm.cc:1315-1331,1431: indices = -1, gridForces = 0, container = 0, vels = 0
    HIPCHK(hipMemsetAsync(s->massd, 0, sizeof(double) * C, st));
    HIPCHK(hipMemsetAsync(s->vel, 0, sizeof(double) * 3 * C, st));
    HIPCHK(hipMemsetAsync(s->forces, 0, sizeof(double) * 3 * C, st));
    if (n) {
        HIPCHK(hipMemsetAsync(s->cell_count, 0, sizeof(int) * C, st));
        k_mpm_sort_count<<<blocks_for(n, 256), 256, 0, st>>>(G, n, s->P, s->cell_count, s->key, s->rank);
        k_mpm_scan_sums<<<nb, SCAN_T, 0, st>>>(C, s->cell_count, s->sums);
        k_mpm_scan_blocks<<<1, 1024, 0, st>>>(nb, s->sums, nullptr);
        k_mpm_scan_excl<<<nb, SCAN_T, 0, st>>>(C, s->cell_count, s->sums, s->cell_start);
        k_mpm_sort_place<<<blocks_for(n, 256), 256, 0, st>>>(n, s->key, s->rank, s->cell_start, s->pid, s->order, s->rank2);
        // inside a cell: ascending upload index (the fluid path's rank pass: ids staged through LDS — a thread per particle walking its
        // cell's 400 ids took 85-134 us on the reference's scene)
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(s->cell_start + C), (int)n, 1, st));
        fl::launch_bin_rank(st, n, 0, s->key, s->cell_start, s->order, (const uint32_t*)s->rank2, s->order2);
        k_mpm_permute<<<blocks_for(n, 256), 256, 0, st>>>(n, s->P.cap, s->order2, s->P, s->P2, s->pid, s->pid2);
        std::swap(s->P.pos, s->P2.pos), std::swap(s->P.vel, s->P2.vel), std::swap(s->P.FE, s->P2.FE), std::swap(s->P.FP, s->P2.FP);
        std::swap(s->P.volume, s->P2.volume), std::swap(s->pid, s->pid2);
    }
    if (n) {
        k_mpm_touched<<<cb, 256, 0, st>>>(G, s->solid, s->cell_count, s->tflag);
        k_mpm_scan_sums<<<nb, SCAN_T, 0, st>>>(C, s->tflag, s->sums);
        k_mpm_scan_blocks<<<1, 1024, 0, st>>>(nb, s->sums, &s->state->num_touched);
        k_mpm_scan_final<<<nb, SCAN_T, 0, st>>>(C, s->tflag, s->sums, s->tidx, s->tlist);
        k_mpm_cell_flags<<<cb, 256, 0, st>>>(C, s->cell_count, s->cflag);
        k_mpm_scan_sums<<<nb, SCAN_T, 0, st>>>(C, s->cflag, s->sums);
        k_mpm_scan_blocks<<<1, 1024, 0, st>>>(nb, s->sums, &s->state->num_cells);
        k_mpm_scan_final<<<nb, SCAN_T, 0, st>>>(C, s->cflag, s->sums, s->cidx, s->clist);
        k_mpm_gather<0><<<GATHER_BLOCKS, 256, 0, st>>>(G, s->P, s->cell_start, s->cell_count, s->tlist, &s->state->num_touched, s->state, 0, 0, nullptr, nullptr, s->massd, s->vel);
    }
    k_mpm_cells<<<cb, 256, 0, st>>>(G, s->solid, s->massd, s->container, s->output, s->vel, s->velb, s->flag);
    k_mpm_scan_sums<<<nb, SCAN_T, 0, st>>>(C, s->flag, s->sums);
    k_mpm_scan_blocks<<<1, 1024, 0, st>>>(nb, s->sums, &s->state->num_active);
    k_mpm_scan_final<<<nb, SCAN_T, 0, st>>>(C, s->flag, s->sums, s->indices, s->active_cell);
    if (s->step_no == 0 && n) k_mpm_volume<<<pb, 128, 0, st>>>(G, n, s->P, s->solid, s->container);   // mpm.cc:1343-1346
    HIPCHK(hipEventRecord(s->ev[1], st));
    // populateGridForces (mpm.cc:1395): mu = E / (2 (1 + nu)), lambda = E nu / ((1 + nu) (1 - 2 nu))
    const double mu0 = pr.youngs_modulus / (2 * (1 + pr.poisson_ratio));
    const double lambda0 = pr.youngs_modulus * pr.poisson_ratio / ((1 + pr.poisson_ratio) * (1 - 2 * pr.poisson_ratio));
    if (n) {
        k_mpm_forces<<<pb, 128, 0, st>>>(G, n, s->P, mu0, lambda0, pr.hardening);
        k_mpm_gather<1><<<GATHER_BLOCKS, 256, 0, st>>>(G, s->P, s->cell_start, s->cell_count, s->tlist, &s->state->num_touched, s->state, 0, 0, nullptr, nullptr, s->forces, nullptr);
    }
    HIPCHK(hipEventRecord(s->ev[2], st));
    // populateMatrices' right-hand side (mpm.cc:383-416) and cg.compute / cg.solve (mpm.cc:1404-1405)
    k_mpm_rhs<<<RED_BLOCKS, 256, 0, st>>>(G, s->active_cell, s->state, s->container, s->vel, s->forces, pr.gravity[0], pr.gravity[1], pr.gravity[2],
                                          s->b, s->invm, s->part, &s->state->max_coeff_bits);
    k_mpm_maxforce_cell<<<RED_BLOCKS, 256, 0, st>>>(G, s->active_cell, s->state, s->container, s->forces);
    k_mpm_maxforce_final<<<1, 1, 0, st>>>(G, s->state, s->container, s->forces);
    k_mpm_cg_init<<<RED_BLOCKS, 256, 0, st>>>(s->active_cell, s->state, s->container, pr.transpose_system, s->b, s->x, s->r, s->p, s->q, s->part, s->part + RED_BLOCKS,
                                          Dense{s->vdense, s->invm, s->C, pr.transpose_system});
    k_mpm_cg_start<<<1, 256, 0, st>>>(RED_BLOCKS, s->part, s->part + RED_BLOCKS, s->state);
    HIPCHK(hipGetLastError());
    if (read_state(s)) return FLUID_ERR_HIP;
    s->num_active = s->h_state->num_active;
    s->num_cells = s->h_state->num_cells;
    {
        const size_t need = (size_t)24 * (size_t)std::max(s->num_cells, 1);
        if (need > s->apart_cap) {
            if (s->apart) { HIPCHK(hipStreamSynchronize(s->st)); HIPCHK(hipFree(s->apart)); s->apart = nullptr; s->apart_cap = 0; }
            HIPCHK(hipMalloc((void**)&s->apart, (need + need / 4) * sizeof(double)));
            s->apart_cap = need + need / 4;
        }
    }
    const long max_iters = pr.cg_max_iters > 0 ? pr.cg_max_iters : (2L * 3 * s->num_active > 0 ? 2L * 3 * s->num_active : 1);
    // Iterations are launched in batches without waiting for the convergence test: every kernel of an iteration returns
    // at once when the device-side flag is set, and the host reads the 100-byte state once per batch.
    float ms_apply = 0;
    int n_apply = 0;
    // first batch: what the previous step's solve needed (counts change slowly from step to step), then two at a time
    long batch = s->last_iters > 0 ? s->last_iters + 1 : 4;
    long launched = 0;
    const Dense dn{s->vdense, s->invm, s->C, pr.transpose_system};
    while (!s->h_state->cg_done) {
        for (long it = 0; it < batch; ++it) {
            const bool timed = it == 0;
            if (timed) HIPCHK(hipEventRecord(s->ev[6], st));
            if (apply_operator(s, 1)) return FLUID_ERR_HIP;
            if (timed) HIPCHK(hipEventRecord(s->ev[7], st));
            double *pa = s->part, *pb = s->part + RED_BLOCKS, *pc = s->part + 2 * RED_BLOCKS, *pd = s->part + 3 * RED_BLOCKS;
            k_mpm_cg_pq<<<RED_BLOCKS, 256, 0, st>>>(G, s->active_cell, s->state, s->container, pr.transpose_system, s->p, s->q, pa, pb, s->cidx, s->apart, s->invm);
            if (3L * s->num_active <= MPM_XRP_MAX) {
                double *rc_ = (launched & 1) ? s->r2 : s->r, *rn_ = (launched & 1) ? s->r : s->r2;
                k_mpm_cg_xrp<<<64, 256, 0, st>>>(s->active_cell, s->state, s->container, pr.transpose_system, (int)(launched & 1), pr.cg_tol, max_iters, launched, pb, s->x, rc_, rn_,
                                                 s->p, s->q, dn);
            } else {
                k_mpm_cg_xr<<<RED_BLOCKS, 256, 0, st>>>(s->active_cell, s->state, s->container, pr.transpose_system, (int)(launched & 1), pb, s->x, s->r, s->p,
                                                        s->q, pc, pd);
                k_mpm_cg_p<<<RED_BLOCKS, 256, 0, st>>>(s->state, (int)(launched & 1), pr.cg_tol, max_iters, pc, pd, s->r, s->p, s->q, s->active_cell, dn);
            }
            ++launched;
        }
        HIPCHK(hipGetLastError());
        if (read_state(s)) return FLUID_ERR_HIP;
        float ms = 0;
        if (hipEventElapsedTime(&ms, s->ev[6], s->ev[7]) == hipSuccess) ms_apply += ms, ++n_apply;
        batch = 2;
    }
    // the operand's nodes back to 0: the next step's unknowns are other nodes
    k_mpm_spread<<<blocks_for(3L * std::max(s->num_active, 1), 256), 256, 0, st>>>(s->state, s->active_cell, s->p, dn, 1);
    const int iters = s->h_state->cg_iters;
    s->last_iters = iters;
    const double cg_error = s->h_state->bb > 0 ? std::sqrt(s->h_state->rr / s->h_state->bb) : 0.0;
    HIPCHK(hipEventRecord(s->ev[3], st));
    k_mpm_update_velocity<<<cb, 256, 0, st>>>(G, s->solid, s->container, s->indices, s->x, s->vel);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    s->mid_step = true;
    mpm_step_stats_t& o = s->stats;
    memset(&o, 0, sizeof o);
    const MpmState& h = *s->h_state;
    o.dt_in = s->dt, o.dt_out = s->dt, o.cg_error = cg_error;
    for (int a = 0; a < 3; ++a) o.max_force[a] = h.max_force[a];
    o.max_mi = h.max_mi, o.max_force_coeff2 = bits_to_double(h.max_coeff_bits);
    o.num_active = h.num_active, o.cg_iters = iters, o.any_active = h.any_active;
    o.cg_status = h.cg_done;
    float ms;
    double* dst[3] = {&o.ms_transfer, &o.ms_forces, &o.ms_solve};
    for (int k = 0; k < 3; ++k)
        if (hipEventElapsedTime(&ms, s->ev[k], s->ev[k + 1]) == hipSuccess) *dst[k] = ms;
    o.ms_apply_avg = n_apply ? ms_apply / n_apply : 0.0;
    if (out) *out = o;
    if (h.cg_done == 2) return fluid_fail(FLUID_ERR_SOLVER, "mpm_step_solve: CG breakdown (<p, A p> <= 0 or not finite): the operator is not positive for these particles");
    return 0;
}

// mpm.cc:1410-1431: updateDeformationGradient, FLIPadvect (the grid velocity stays readable until the next step's transfer)
int mpm_step_advance(mpm_sim_t* s, mpm_step_stats_t* out)
{
    if (!s) return fluid_fail(FLUID_ERR_ARG, "mpm_step_advance: null handle");
    if (!s->mid_step) return fluid_fail(FLUID_ERR_STATE, "mpm_step_advance: call mpm_step_solve first");
    const MGrid G = s->G;
    const long n = s->n;
    const mpm_params_t& pr = s->prm;
    hipStream_t st = s->st;
    const unsigned pb = blocks_for(n, 128);
    HIPCHK(hipEventRecord(s->ev[3], st));
    if (n) k_mpm_deform<<<pb, 128, 0, st>>>(G, n, s->P, s->solid, s->vel, s->state, 1 - pr.theta_c, 1 + pr.theta_s);
    HIPCHK(hipEventRecord(s->ev[4], st));
    if (n) k_mpm_flip<<<pb, 128, 0, st>>>(G, n, s->P, s->vel, s->velb, s->state);
    k_mpm_timestep<<<1, 1, 0, st>>>(s->state, pr.max_dt, pr.dx);
    if (n) k_mpm_advect<<<pb, 128, 0, st>>>(G, n, s->P, s->solid, s->state);
    HIPCHK(hipEventRecord(s->ev[5], st));
    HIPCHK(hipGetLastError());
    if (read_state(s)) return FLUID_ERR_HIP;
    const MpmState& h = *s->h_state;
    s->dt = h.dt;
    s->step_no++;
    s->mid_step = false;
    mpm_step_stats_t& o = s->stats;
    o.dt_out = h.dt;
    o.max_speed = bits_to_double(h.max_speed_bits), o.max_grad = bits_to_double(h.max_grad_bits);
    o.max_fp = bits_to_double(h.max_fp_bits), o.max_fe = bits_to_double(h.max_fe_bits);
    float ms;
    if (hipEventElapsedTime(&ms, s->ev[3], s->ev[4]) == hipSuccess) o.ms_deform = ms;
    if (hipEventElapsedTime(&ms, s->ev[4], s->ev[5]) == hipSuccess) o.ms_advect = ms;
    if (out) *out = o;
    return 0;
}

int mpm_step(mpm_sim_t* s, mpm_step_stats_t* out)
{
    int rc = mpm_step_solve(s, nullptr);
    if (rc) return rc;
    return mpm_step_advance(s, out);
}

int mpm_download_particles(mpm_sim_t* s, int32_t what, double* out)
{
    if (!s || !out || what < 0 || what > 5) return fluid_fail(FLUID_ERR_ARG, "mpm_download_particles: bad argument");
    const double* src[] = {s->P.pos, s->P.vel, s->P.FE, s->P.FP, s->P.gradV, s->P.volume};
    const int w[] = {3, 3, 9, 9, 9, 1};
    const long m = s->n;
    if (!m) return 0;
    if (ensure_stage(s, sizeof(double) * w[what] * m)) return FLUID_ERR_HIP;
    k_mpm_to_aos<<<blocks_for((long)w[what] * m, 256), 256, 0, s->st>>>(m, s->P.cap, w[what], s->pid, src[what], s->stage);
    HIPCHK(hipMemcpyAsync(out, s->stage, sizeof(double) * w[what] * m, hipMemcpyDeviceToHost, s->st));
    HIPCHK(hipStreamSynchronize(s->st));
    return 0;
}

int mpm_download_field(mpm_sim_t* s, int32_t field, void* out)
{
    if (!s || !out) return fluid_fail(FLUID_ERR_ARG, "mpm_download_field: null argument");
    const long C = s->C;
    switch (field) {
        case MPM_F_CONTAINER: HIPCHK(hipMemcpyAsync(out, s->container, 4 * C, hipMemcpyDeviceToHost, s->st)); break;
        case MPM_F_OUTPUT: HIPCHK(hipMemcpyAsync(out, s->output, 4 * C, hipMemcpyDeviceToHost, s->st)); break;
        case MPM_F_INDICES: HIPCHK(hipMemcpyAsync(out, s->indices, 4 * C, hipMemcpyDeviceToHost, s->st)); break;
        case MPM_F_SOLID: {
            std::vector<uint8_t> h(C);
            HIPCHK(hipMemcpyAsync(h.data(), s->solid, C, hipMemcpyDeviceToHost, s->st));
            HIPCHK(hipStreamSynchronize(s->st));
            for (long k = 0; k < C; ++k) ((float*)out)[k] = h[k] ? 1.f : 0.f;
            return 0;
        }
        case MPM_F_VEL_BEFORE:
        case MPM_F_FORCES:
        case MPM_F_VEL: {
            const double* src = field == MPM_F_VEL_BEFORE ? s->velb : (field == MPM_F_FORCES ? s->forces : s->vel);
            if (ensure_stage(s, sizeof(double) * 3 * C)) return FLUID_ERR_HIP;
            k_mpm_interleave<<<blocks_for(C, 256), 256, 0, s->st>>>(C, src, s->stage);
            HIPCHK(hipMemcpyAsync(out, s->stage, sizeof(double) * 3 * C, hipMemcpyDeviceToHost, s->st));
            break;
        }
        default: return fluid_fail(FLUID_ERR_ARG, "mpm_download_field: unknown field");
    }
    HIPCHK(hipStreamSynchronize(s->st));
    return 0;
}

int mpm_download_system(mpm_sim_t* s, double* b, double* x, int64_t count)
{
    if (!s) return fluid_fail(FLUID_ERR_ARG, "mpm_download_system: null handle");
    if (count != 3 * (int64_t)s->num_active)
        return fluid_fail(FLUID_ERR_ARG, "mpm_download_system: the buffers hold " + std::to_string(count) + " doubles, the last solve's system has " +
                                             std::to_string(3 * (int64_t)s->num_active));
    const size_t bytes = sizeof(double) * 3 * (size_t)s->num_active;
    if (b && bytes) HIPCHK(hipMemcpyAsync(b, s->b, bytes, hipMemcpyDeviceToHost, s->st));
    if (x && bytes) HIPCHK(hipMemcpyAsync(x, s->x, bytes, hipMemcpyDeviceToHost, s->st));
    HIPCHK(hipStreamSynchronize(s->st));
    return 0;
}

int mpm_apply_matrix(mpm_sim_t* s, const double* v, double* y)
{
    if (!s || !v || !y) return fluid_fail(FLUID_ERR_ARG, "mpm_apply_matrix: null argument");
    if (!s->mid_step) return fluid_fail(FLUID_ERR_STATE, "mpm_apply_matrix: only between mpm_step_solve and mpm_step_advance");
    const size_t n3 = 3 * (size_t)s->num_active;
    if (!n3) return 0;
    HIPCHK(hipMemcpyAsync(s->p, v, sizeof(double) * n3, hipMemcpyHostToDevice, s->st));
    k_mpm_copy<<<blocks_for((long)n3, 256), 256, 0, s->st>>>((long)n3, s->p, s->q);
    if (apply_operator(s, 0)) return FLUID_ERR_HIP;
    HIPCHK(hipMemcpyAsync(y, s->q, sizeof(double) * n3, hipMemcpyDeviceToHost, s->st));
    HIPCHK(hipStreamSynchronize(s->st));
    return 0;
}

int mpm_eval(int32_t what, int64_t n, const double* a, const double* b, double p0, double p1, double p2, double* out0, double* out1)
{
    if (n < 0 || !a || !out0 || what < MPM_EVAL_POLAR || what > MPM_EVAL_CLAMP) return fluid_fail(FLUID_ERR_ARG, "mpm_eval: bad argument");
    if ((what == MPM_EVAL_POLAR || what == MPM_EVAL_CLAMP) && !out1) return fluid_fail(FLUID_ERR_ARG, "mpm_eval: this function has two results");
    if ((what == MPM_EVAL_HESSIAN || what == MPM_EVAL_CLAMP) && !b) return fluid_fail(FLUID_ERR_ARG, "mpm_eval: this function takes two matrices");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fluid_fail(FLUID_ERR_HIP, "mpm_eval: no HIP device (this library has no CPU path)");
    if (!n) return 0;
    const size_t bytes = sizeof(double) * 9 * (size_t)n;
    double *da = nullptr, *db = nullptr, *d0 = nullptr, *d1 = nullptr;
    auto cleanup = [&]() { for (double* p : {da, db, d0, d1}) if (p) (void)hipFree(p); };
    if (hipMalloc((void**)&da, bytes) != hipSuccess || (b && hipMalloc((void**)&db, bytes) != hipSuccess) || hipMalloc((void**)&d0, bytes) != hipSuccess ||
        hipMalloc((void**)&d1, bytes) != hipSuccess) {
        cleanup();
        return fluid_fail(FLUID_ERR_HIP, "mpm_eval: device allocation failed");
    }
    hipError_t e = hipMemcpy(da, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && b) e = hipMemcpy(db, b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_mpm_eval<<<blocks_for(n, 128), 128>>>(what, n, da, db, p0, p1, p2, d0, d1);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out0, d0, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess && out1) e = hipMemcpy(out1, d1, bytes, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fluid_fail(FLUID_ERR_HIP, std::string("mpm_eval: ") + hipGetErrorString(e));
    return 0;
}

}  // extern "C"
