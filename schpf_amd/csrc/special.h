// Special functions of the fused Gamma update (kernels.hip gamma_update_kernel), written for its instruction count:
// the kernel evaluates psi(shape) - log(rate), shape / rate and an exponential for every (row, factor) of both
// sides each iteration and is bound by how many VALU instructions that takes (profiles/r03/update_kernel_variant.txt,
// profiles/r06).  Replaces SciPy's psi (bound by the reference at hpf_numba.py:16-18) and libm's log / exp on the
// path hpf_numba.py:83-94.
//
// Plain C++ over fma / frexp / ldexp, so that the same text compiles for the host: tests/test_special_host.py builds it
// with g++ (the hardware reciprocal seed replaced by a 24-bit one) and checks it against SciPy's values on the
// psi_gammaln.npz grid.  Accuracy: psi within 4e-15 (relative or absolute) on [1e-4, 1e6]; log and exp within 2 ulp.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define SCHPF_SF __host__ __device__ __forceinline__
#else
#define SCHPF_SF inline
#endif

namespace schpf {

// a * b + C for a compile-time constant C: on the device one v_fma_f64 that reads C from a scalar register pair.  Left to
// itself hipcc (ROCm 7.2, gfx950: no literal operands in VOP3) keeps the ~45 double constants of the series below in
// VECTOR registers or rebuilds them with two v_mov_b32 per use and copies them into the accumulator of a v_fmac: three to
// four VALU instructions per Horner step, a third of the update kernel's instructions; a scalar operand costs SALU moves.
SCHPF_SF double fma_c(double a, double b, double C)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(C));
    return r;
#else
    return std::fma(a, b, C);
#endif
}

// 1 / x for a normal x whose reciprocal is normal: hardware seed (~2^-24 on gfx950) + two Newton steps (error squared
// twice: below the rounding of the last fma) -- five instructions where the IEEE division sequence takes ~15
SCHPF_SF double fast_rcp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
#else
    double r = 1.0 / x;                      // host build: a seed with the device seed's precision
    uint64_t b;
    std::memcpy(&b, &r, 8);
    b &= ~((1ull << 29) - 1);
    std::memcpy(&r, &b, 8);
#endif
    r = std::fma(std::fma(-x, r, 1.0), r, r);
    r = std::fma(std::fma(-x, r, 1.0), r, r);
    return r;
}

// log(z), z >= 0.  z = m 2^e with m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716:
// ten terms of the odd series (the eleventh is below 1e-18).  ~25 instructions, no table; libm's is ~70.
SCHPF_SF double fast_log(double z)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int e = __builtin_amdgcn_frexp_exp(z);
    double m = __builtin_amdgcn_frexp_mant(z);                  // [0.5, 1)
#else
    int e;
    double m = std::frexp(z, &e);
#endif
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double s = (m - 1.0) * fast_rcp(m + 1.0);
    const double s2 = s * s;
    double p = 1.0 / 21.0;
    p = fma_c(p, s2, 1.0 / 19.0);
    p = fma_c(p, s2, 1.0 / 17.0);
    p = fma_c(p, s2, 1.0 / 15.0);
    p = fma_c(p, s2, 1.0 / 13.0);
    p = fma_c(p, s2, 1.0 / 11.0);
    p = fma_c(p, s2, 1.0 / 9.0);
    p = fma_c(p, s2, 1.0 / 7.0);
    p = fma_c(p, s2, 1.0 / 5.0);
    p = fma_c(p, s2, 1.0 / 3.0);
    const double two_s = s + s;
    const double lm = std::fma(two_s * s2, p, two_s);           // log m
    const double ed = (double)e;
    double lg = std::fma(ed, 0.693147180559945286, lm) + ed * 2.319046813846299558e-17;   // ln 2 = hi + lo
    // the ends of the range by selects, not by a branch into libm (its inlined body costs the kernel ~25 registers):
    // frexp takes denormals as they are; log 0 = -inf, log inf = inf, and a NaN came through the arithmetic as NaN
    lg = z == 0.0 ? -HUGE_VAL : lg;
    lg = z == HUGE_VAL ? HUGE_VAL : lg;
    return lg;
}

// exp(d) for d <= ~1 (the kernel shifts by the row's maximum; below -745 the result is 0 like libm's).
// d = n ln 2 + r, |r| <= 0.347; exp r by its series to r^13 (the next term is 2e-17 relative); ~22 instructions.
SCHPF_SF double fast_exp(double d)
{
    d = d < -1000.0 ? -1000.0 : d;
    const double n = std::rint(d * 1.44269504088896340736);
    double r = std::fma(-n, 6.93147180369123816490e-01, d);     // fdlibm's ln2HI / ln2LO: n * ln2HI is exact
    r = std::fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma_c(p, r, 1.0 / 479001600.0);
    p = fma_c(p, r, 1.0 / 39916800.0);
    p = fma_c(p, r, 1.0 / 3628800.0);
    p = fma_c(p, r, 1.0 / 362880.0);
    p = fma_c(p, r, 1.0 / 40320.0);
    p = fma_c(p, r, 1.0 / 5040.0);
    p = fma_c(p, r, 1.0 / 720.0);
    p = fma_c(p, r, 1.0 / 120.0);
    p = fma_c(p, r, 1.0 / 24.0);
    p = fma_c(p, r, 1.0 / 6.0);
    p = std::fma(p, r, 0.5);
    p = std::fma(p, r, 1.0);
    p = std::fma(p, r, 1.0);
    return std::ldexp(p, (int)n);
}

// psi(x) - log(rate) for x > 0, given inv_rate = 1 / rate:  psi(x) = psi(x + 10) - sum_{j<10} 1 / (x + j), and
// psi(x + 10) by the asymptotic series (Cephes' psi, which SciPy's follows for x >= 10: seven Bernoulli terms).
//
// The recurrence without a loop: the ten factors pair up as (x + j)(x + 9 - j) = u + c_j with u = x (x + 9),
// c = {0, 8, 14, 18, 20}, so  sum_j 1 / (x + j) = (2 x + 9) Q(u) / P(u),  P = prod (u + c),  Q = dP/du -- two quartics
// in u with positive coefficients (no cancellation for u > 0), ~13 instructions where the data-dependent loop took ~90
// per wave (every lane pays the longest trip of its wave, and some lane almost always starts below 1).  ONE reciprocal
// serves 1 / (x + 10) and 1 / P, and log(x + 10) - log(rate) is one logarithm of (x + 10) * inv_rate.
// Above 1e8 the shift is skipped (P would overflow near 1e30; there psi(x) = log x - 1 / 2x to the last bit).
SCHPF_SF double digamma_less_log(double x, double inv_rate)
{
    const bool big = x >= 1e8;
    const double xs = big ? 1.0 : x;
    const double u = std::fma(xs, xs, 9.0 * xs);
    double P = u + 60.0;
    P = fma_c(P, u, 1308.0);
    P = fma_c(P, u, 12176.0);
    P = fma_c(P, u, 40320.0);
    P *= u;
    double Q = std::fma(5.0, u, 240.0);
    Q = fma_c(Q, u, 3924.0);
    Q = fma_c(Q, u, 24352.0);
    Q = fma_c(Q, u, 40320.0);
    const double num = std::fma(2.0, xs, 9.0) * Q;
    const double xp = big ? x : x + 10.0;
    const double rD = fast_rcp(big ? x : xp * P);
    const double r = big ? rD : rD * P;                 // 1 / xp
    const double S = big ? 0.0 : num * rD * xp;         // num / P
    const double z = r * r;
    double p = 8.33333333333333333333E-2;
    p = fma_c(p, z, -2.10927960927960927961E-2);
    p = fma_c(p, z, 7.57575757575757575758E-3);
    p = fma_c(p, z, -4.16666666666666666667E-3);
    p = fma_c(p, z, 3.96825396825396825397E-3);
    p = fma_c(p, z, -8.33333333333333333333E-3);
    p = fma_c(p, z, 8.33333333333333333333E-2);
    return fast_log(xp * inv_rate) - 0.5 * r - z * p - S;
}

// psi(x), x > 0 (the stateless mirror schpf_digamma and the E[log] operator)
SCHPF_SF double digamma(double x) { return digamma_less_log(x, 1.0); }

}  // namespace schpf
