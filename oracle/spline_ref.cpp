// ORACLE — TEST INFRASTRUCTURE ONLY.  The reference's own spline() (fluid.cc:22-37), compiled from the lines the Makefile
// extracts from /root/reference/fluid.cc into the git-ignored oracle/_ref/ at build time (nothing of the reference is
// committed).  It has no dependency, so this is the reference's code itself: the known-answer source for row a1.
#include "_ref/spline_extract.inc"

extern "C" double ref_spline(double x) { return spline(x); }
extern "C" void ref_spline_n(long n, const double* x, double* w)
{
    for (long i = 0; i < n; ++i) w[i] = spline(x[i]);
}
