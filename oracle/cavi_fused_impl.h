/*
 * oracle/cavi_fused_impl.h -- TEST INFRASTRUCTURE / CPU BASELINE, NOT PRODUCT CODE.
 *
 * "Fused OpenMP" CPU comparator of SURVEY.md 8(d), variant (ii): the best CPU form of one
 * default-order CAVI iteration (/root/reference/schpf/scHPF_.py:657-714) this build knows --
 * so that the GPU figure is not flattered by the reference's execution shape (Xphi
 * materialised, two SERIAL scatter-adds; hpf_numba.py:54-156), which is variant (i),
 * cavi_oracle_impl.h.  Same algebra as the device sweep: exp() hoisted out of
 * the per-nonzero loop,
 *     phi_k = Et[i,k] Eb[g,k] / s_ig,   Et = exp(Elog - rowmax),
 * X*phi never materialised, one thread-parallel pass over the rows (CSR) for the theta side
 * and one over the columns (CSC) for the beta side, no atomics.  Included twice by
 * cavi_fused.c (REAL = double / float).  Checked against variant (i) in
 * tests/test_oracle_golden.py::test_fused_cpu_variant_matches_reference_structure.
 */
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* tab[i,:] = exp(Elog[i,:] - max_k Elog[i,:]),  Elog = psi(shape) - log(rate)  (hpf_numba.py:83-94) */
static void FN(fused_tables)(long n, int K, const REAL *shape, const REAL *rate, REAL *tab, int nthreads)
{
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (long i = 0; i < n; ++i) {
        REAL mx = -INFINITY;
        REAL *t = tab + (size_t)i * K;
        for (int k = 0; k < K; ++k) {
            t[k] = (REAL)(orc_psi((double)shape[i * K + k]) - (double)RLOG(rate[i * K + k]));
            if (t[k] > mx) mx = t[k];
        }
        for (int k = 0; k < K; ++k) t[k] = REXP(t[k] - mx);
    }
}

/* out[k] = sum_i shape[i,k]/rate[i,k], per-thread partials added in thread order */
static void FN(fused_colsum)(long n, int K, const REAL *shape, const REAL *rate, double *out, int nthreads)
{
    double *part = (double *)calloc((size_t)nthreads * K, sizeof(double));
#pragma omp parallel num_threads(nthreads)
    {
        double *p = part + (size_t)omp_get_thread_num() * K;
#pragma omp for schedule(static)
        for (long i = 0; i < n; ++i)
            for (int k = 0; k < K; ++k) p[k] += (double)(shape[i * K + k] / rate[i * K + k]);
    }
    for (int k = 0; k < K; ++k) {
        double s = 0.0;
        for (int t = 0; t < nthreads; ++t) s += part[(size_t)t * K + k];
        out[k] = s;
    }
    free(part);
}

/* One side of the iteration: for every major row m (cell with its CSR row, or gene with its CSC
 * column): shape[m,k] = prior + Tmaj[m,k] * sum_nz (x / s) * Tmin[minor,k]   (scHPF_.py:699-700 / :709-710)
 *          rate[m,k]  = cap_shape[m]/cap_rate[m] + S_other[k]                 (:701-703 / :711-713)
 *          cap_rate[m] = cap_prior + sum_k shape/rate                         (:704 / :714) */
static void FN(fused_side)(long n_major, int K, const long *ptr, const int *minor, const REAL *val,
                           const REAL *tmaj, const REAL *tmin, double prior, const REAL *cap_shape,
                           REAL *cap_rate, double cap_prior, const double *s_other, REAL *shape, REAL *rate,
                           int nthreads)
{
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 32)
    for (long m = 0; m < n_major; ++m) {
        REAL acc[256];
        const REAL *tm = tmaj + (size_t)m * K;
        for (int k = 0; k < K; ++k) acc[k] = 0;
        for (long j = ptr[m]; j < ptr[m + 1]; ++j) {
            const REAL *b = tmin + (size_t)minor[j] * K;
            REAL s = 0;
#pragma omp simd reduction(+ : s)
            for (int k = 0; k < K; ++k) s += tm[k] * b[k];
            const REAL w = val[j] / s;
#pragma omp simd
            for (int k = 0; k < K; ++k) acc[k] += w * b[k];
        }
        const double cap = (double)cap_shape[m] / (double)cap_rate[m];
        double esum = 0.0;
        for (int k = 0; k < K; ++k) {
            const REAL sh = (REAL)(prior + (double)(tm[k] * acc[k]));
            const REAL rt = (REAL)(cap + s_other[k]);
            shape[m * K + k] = sh;
            rate[m * K + k] = rt;
            esum += (double)(sh / rt);
        }
        cap_rate[m] = (REAL)(cap_prior + esum);
    }
}

/* default (non-batched, non-simultaneous) order, scHPF_.py:697-714.  K <= 256. */
void FN(orc_fused_iteration)(int N, int G, int K, const long *rptr, const int *rcol, const REAL *rval,
                             const long *cptr, const int *crow, const REAL *cval, double a, double c,
                             double bp, double dp, REAL *xis, REAL *xir, REAL *ths, REAL *thr, REAL *ets,
                             REAL *etr, REAL *bes, REAL *ber, int nthreads)
{
    REAL *et = (REAL *)malloc(sizeof(REAL) * (size_t)N * K);
    REAL *eb = (REAL *)malloc(sizeof(REAL) * (size_t)G * K);
    double s_theta[256], s_beta[256];
    FN(fused_tables)(N, K, ths, thr, et, nthreads);
    FN(fused_tables)(G, K, bes, ber, eb, nthreads);
    FN(fused_colsum)(N, K, ths, thr, s_theta, nthreads);              /* OLD theta, :701-703 */
    FN(fused_side)(G, K, cptr, crow, cval, eb, et, c, ets, etr, dp, s_theta, bes, ber, nthreads);
    FN(fused_colsum)(G, K, bes, ber, s_beta, nthreads);               /* NEW beta, :711-713 */
    FN(fused_side)(N, K, rptr, rcol, rval, et, eb, a, xis, xir, bp, s_beta, ths, thr, nthreads);
    free(et);
    free(eb);
}

#undef FN
#undef CAT
#undef CAT_
