// ORACLE — TEST INFRASTRUCTURE ONLY (builds into oracle/_ref/, git-ignored).
//
// The reference's actual pressure solver, compiled from the reference's own vendored
// Eigen 3.3.4 headers where they lie (/root/reference/Eigen, header-only): exactly the
// object declared at fluid.cc:1352 and driven at fluid.cc:1460,1472-1474
//   Eigen::ConjugateGradient<SparseMatrix<double>, Lower|Upper, IncompleteCholesky<double>>
//   A.setFromTriplets(...) (fluid.cc:540); cg.compute(A); p = cg.solve(b);
// No reference source is copied: this file only instantiates the vendored templates.
#include <Eigen/Eigen>
#include <Eigen/Sparse>
#include <Eigen/IterativeLinearSolvers>
#include <vector>

typedef Eigen::Triplet<double> Triplet;  // fluid.cc:21

extern "C" void eigen_ref_icpcg(int n, int nnz, const int* rows, const int* cols, const double* vals,
                                const double* b, double* x, int* iters, double* err)
{
    std::vector<Triplet> tripletList;
    tripletList.reserve(nnz);
    for (int k = 0; k < nnz; ++k) tripletList.push_back(Triplet(rows[k], cols[k], vals[k]));
    Eigen::SparseMatrix<double> A(n, n);
    A.setFromTriplets(tripletList.begin(), tripletList.end());
    Eigen::ConjugateGradient<Eigen::SparseMatrix<double>, Eigen::Lower | Eigen::Upper, Eigen::IncompleteCholesky<double>> cg;
    Eigen::VectorXd bv = Eigen::Map<const Eigen::VectorXd>(b, n);
    cg.compute(A);
    Eigen::VectorXd p = cg.solve(bv);
    Eigen::Map<Eigen::VectorXd>(x, n) = p;
    if (iters) *iters = (int)cg.iterations();
    if (err) *err = cg.error();
}

// Same system through Eigen's own DiagonalPreconditioner (Jacobi) — used to pin the
// restated CG loop iteration-for-iteration.
extern "C" void eigen_ref_jacobi_cg(int n, int nnz, const int* rows, const int* cols, const double* vals,
                                    const double* b, double* x, int* iters, double* err)
{
    std::vector<Triplet> tripletList;
    for (int k = 0; k < nnz; ++k) tripletList.push_back(Triplet(rows[k], cols[k], vals[k]));
    Eigen::SparseMatrix<double> A(n, n);
    A.setFromTriplets(tripletList.begin(), tripletList.end());
    Eigen::ConjugateGradient<Eigen::SparseMatrix<double>, Eigen::Lower | Eigen::Upper> cg;
    Eigen::VectorXd bv = Eigen::Map<const Eigen::VectorXd>(b, n);
    cg.compute(A);
    Eigen::VectorXd p = cg.solve(bv);
    Eigen::Map<Eigen::VectorXd>(x, n) = p;
    if (iters) *iters = (int)cg.iterations();
    if (err) *err = cg.error();
}

extern "C" const char* eigen_ref_version()
{
#define STR2(x) #x
#define STR(x) STR2(x)
    return STR(EIGEN_WORLD_VERSION) "." STR(EIGEN_MAJOR_VERSION) "." STR(EIGEN_MINOR_VERSION);
}
