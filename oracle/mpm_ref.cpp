// ORACLE — TEST INFRASTRUCTURE ONLY (builds into oracle/_ref/, git-ignored).
//
// The reference's own constitutive functions of the snow-MPM step, compiled from the lines the Makefile takes out of
// /root/reference at build time (nothing of the reference is committed, the extracts are deleted after the compile):
//   mpm.cc:24-41            `factor`, spline()
//   deformHeader.h:22-88    getR, getS (Eigen::JacobiSVD), spline2, getSplineGradient
//   deformHeader.h:107-249  getDelFE, getDelR (colPivHouseholderQr), getdJF, doubleDot42, doubleDot22, getJFmt, dPsydFdF
//   deformHeader.h:273-313  getSigma
// These depend on the vendored Eigen 3.3.4 only (header-only, /root/reference/Eigen).  getGradW / getdPsydx2
// (deformHeader.h:90-105, 251-272) take OpenVDB types and are NOT built: no stand-in headers.
// The wrappers below only marshal plain arrays (row-major 3x3) in and out.
#include <Eigen/Eigen>
#include <Eigen/SVD>
#include <cmath>
#include <iostream>

#include "_ref/mpm_cc_extract.inc"
#include "_ref/deform_extract.inc"

namespace {
Eigen::Matrix3d in3(const double* a)
{
    Eigen::Matrix3d m;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m(i, j) = a[3 * i + j];
    return m;
}
void out3(const Eigen::Matrix3d& m, double* a)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) a[3 * i + j] = m(i, j);
}
}  // namespace

extern "C" {
double mpm_ref_spline(double x) { return spline(x); }
double mpm_ref_spline2(double x) { return spline2(x); }
double mpm_ref_spline_gradient(double x) { return getSplineGradient(x); }
void mpm_ref_getR(const double* F, double* out) { out3(getR(in3(F)), out); }
void mpm_ref_getS(const double* F, double* out) { out3(getS(in3(F)), out); }
void mpm_ref_getSigma(double mu0, double lambda0, double epsilon, const double* FE, const double* FP, double* out)
{
    out3(getSigma(mu0, lambda0, epsilon, in3(FE), in3(FP)), out);
}
void mpm_ref_dPsydFdF(const double* gradW, const double* F, double lambda, double mu, int i, double* out)
{
    Eigen::Matrix3d f = in3(F);
    out3(dPsydFdF(Eigen::Vector3d(gradW[0], gradW[1], gradW[2]), f, getR(f), getS(f), lambda, mu, f.determinant(), i), out);
}
// The singular-value clamp of updateDeformationGradient, mpm.cc:543-555, written against the same Eigen::JacobiSVD
// object the reference uses (that function body reads PointList members, so it cannot be taken as it is)
void mpm_ref_clamp(const double* tFEa, const double* FPa, double minv, double maxv, double* FEout, double* FPout)
{
    Eigen::Matrix3d tFE = in3(tFEa), F = tFE * in3(FPa);
    Eigen::JacobiSVD<Eigen::Matrix3d> svd(tFE, Eigen::ComputeThinU | Eigen::ComputeThinV);
    Eigen::Vector3d singular = svd.singularValues();
    for (int k = 0; k < 3; ++k) {
        singular(k) = singular(k) > minv ? singular(k) : minv;
        singular(k) = singular(k) < maxv ? singular(k) : maxv;
    }
    out3(svd.matrixU() * singular.asDiagonal() * svd.matrixV().transpose(), FEout);
    out3(svd.matrixV() * singular.asDiagonal().inverse() * svd.matrixU().transpose() * F, FPout);
}
}
