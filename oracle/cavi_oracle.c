/*
 * oracle/cavi_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the reference's CAVI hot path
 * (/root/reference/schpf/hpf_numba.py, schpf/loss.py, and the update order of
 * schpf/scHPF_.py:642-715).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (schpf_amd/) never does.
 *
 * Pinned (tests/test_oracle_golden.py) against the golden vectors in
 * tests/golden/, which were produced by running the reference itself
 * (tests/golden/make_golden.py).
 *
 * Third-party arithmetic: the reference calls SciPy's C `psi` and `gammaln`
 * through ctypes (hpf_numba.py:16-22; SciPy version unpinned, setup.py:13;
 * 1.15.3 in the build container).  SciPy is not part of /root/reference, so
 * orc_psi below restates the published Cephes algorithm for psi(x), x > 0
 * (upward recurrence to x >= 10, then the Bernoulli asymptotic series) and
 * gammaln is libm's lgamma; both are pinned to SciPy's values by
 * tests/golden/psi_gammaln.npz.
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

/* psi(x) for x > 0.  Cephes psi.c: w = sum 1/(x+j) until x+j >= 10, then
 * ln s - 1/(2s) - sum_n B_2n / (2n s^2n). */
double orc_psi(double x)
{
    static const double A[7] = {
        8.33333333333333333333E-2, -2.10927960927960927961E-2, 7.57575757575757575758E-3,
        -4.16666666666666666667E-3, 3.96825396825396825397E-3, -8.33333333333333333333E-3,
        8.33333333333333333333E-2};
    double s = x, w = 0.0;
    while (s < 10.0) {
        w += 1.0 / s;
        s += 1.0;
    }
    double y = 0.0;
    if (s < 1.0e17) {
        double z = 1.0 / (s * s);
        double p = A[0];
        for (int i = 1; i < 7; ++i) p = p * z + A[i];
        y = z * p;
    }
    return log(s) - 0.5 / s - y - w;
}

double orc_gammaln(double x) { return lgamma(x); }

void orc_psi_array(long n, const double *x, double *out)
{
    for (long i = 0; i < n; ++i) out[i] = orc_psi(x[i]);
}

void orc_gammaln_array(long n, const double *x, double *out)
{
    for (long i = 0; i < n; ++i) out[i] = lgamma(x[i]);
}

#define REAL double
#define SUFFIX _f64
#define RLOG log
#define REXP exp
#include "cavi_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef RLOG
#undef REXP

#define REAL float
#define SUFFIX _f32
#define RLOG logf
#define REXP expf
#include "cavi_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef RLOG
#undef REXP
