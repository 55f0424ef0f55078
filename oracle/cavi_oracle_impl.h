/*
 * oracle/cavi_oracle_impl.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Type-generic body of the CPU oracle; included twice by cavi_oracle.c with
 * REAL = double (suffix _f64) and REAL = float (suffix _f32).
 *
 * Every function restates, loop for loop, what one reference function
 * computes.  Citations are /root/reference paths (schpf/hpf_numba.py unless
 * stated otherwise).  The execution shape of the reference is kept on purpose
 * (Xphi is materialised, the two scatter-adds are serial) because this file is
 * also the timed "numba-structure" CPU baseline of bench.py.
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* E[log x] of a Gamma(shape, rate): psi(shape) - log(rate).
 * hpf_numba.py:83-87 (theta), :90-94 (beta).  psi is always evaluated in
 * double (the reference binds SciPy's double psi, hpf_numba.py:16-18) and the
 * difference is stored in the model dtype. */
static void FN(orc_elogx)(long n, const REAL *shape, const REAL *rate, REAL *out, int nthreads)
{
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (long i = 0; i < n; ++i)
        out[i] = (REAL)(orc_psi((double)shape[i]) - (double)RLOG(rate[i]));
}

/* compute_Xphi_data, hpf_numba.py:54-114.
 * x: nonzero values (already converted to REAL), row/col: int32 indices.
 * out: (nnz, K) row-major = x[i] * softmax_k(Elog theta[row,k] + Elog beta[col,k]),
 * softmax evaluated with the max-shift of :101-109. */
void FN(orc_xphi)(long nnz, int N, int G, int K, const REAL *x, const int *row, const int *col,
                  const REAL *ths, const REAL *thr, const REAL *bes, const REAL *ber,
                  REAL *out, int nthreads)
{
    REAL *elt = (REAL *)malloc(sizeof(REAL) * (size_t)N * K);
    REAL *elb = (REAL *)malloc(sizeof(REAL) * (size_t)G * K);
    FN(orc_elogx)((long)N * K, ths, thr, elt, nthreads);
    FN(orc_elogx)((long)G * K, bes, ber, elb, nthreads);
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (long i = 0; i < nnz; ++i) {
        const REAL *t = elt + (size_t)row[i] * K;
        const REAL *b = elb + (size_t)col[i] * K;
        REAL *o = out + (size_t)i * K;
        REAL largest = t[0] + b[0];
        for (int k = 0; k < K; ++k) {          /* logrho, :100-101 */
            o[k] = t[k] + b[k];
            if (o[k] > largest) largest = o[k];
        }
        REAL normalizer = 0;                    /* :104-109 */
        for (int k = 0; k < K; ++k) {
            o[k] = REXP(o[k] - largest);
            normalizer += o[k];
        }
        /* :111-112.  X_data is an integer array in the reference, and
         * int32 * float32 promotes to float64 (numba and NumPy alike), so the
         * product/quotient is formed in double and rounded on the store. */
        for (int k = 0; k < K; ++k)
            o[k] = (REAL)((double)x[i] * (double)o[k] / (double)normalizer);
    }
    free(elt);
    free(elb);
}

/* compute_loading_shape_update, hpf_numba.py:128-156: serial scatter-add of
 * the rows of Xphi by index, on top of the shape prior.
 * scatter_threads <= 1: the reference's own serial loop (what bench.py's CPU
 * baseline times).  scatter_threads > 1 (the parity tests at BASELINE sizes):
 * thread t owns the destination rows [t*nkeep/T, (t+1)*nkeep/T) and walks ALL
 * nonzeros in the same order i = 0 .. nnz-1, adding only those that land in
 * its rows -- every out[j,k] receives the same terms in the same order as in
 * the serial loop, so the result is the same bits, T times sooner. */
void FN(orc_shape_update)(long nnz, int K, const REAL *xphi, const int *keep, int nkeep,
                          double prior, REAL *out, int scatter_threads)
{
    for (long i = 0; i < (long)nkeep * K; ++i) out[i] = (REAL)prior;   /* :151 */
    if (scatter_threads <= 1) {
        for (long i = 0; i < nnz; ++i) {                                 /* :152-155 */
            REAL *o = out + (size_t)keep[i] * K;
            const REAL *p = xphi + (size_t)i * K;
            for (int k = 0; k < K; ++k) o[k] += p[k];
        }
        return;
    }
#pragma omp parallel num_threads(scatter_threads)
    {
        const int T = omp_get_num_threads(), t = omp_get_thread_num();
        const int lo = (int)((long)nkeep * t / T), hi = (int)((long)nkeep * (t + 1) / T);
        for (long i = 0; i < nnz; ++i) {
            const int j = keep[i];
            if (j < lo || j >= hi) continue;
            REAL *o = out + (size_t)j * K;
            const REAL *p = xphi + (size_t)i * K;
            for (int k = 0; k < K; ++k) o[k] += p[k];
        }
    }
}

/* compute_loading_rate_update, hpf_numba.py:159-177:
 * out[i,k] = prior_shape[i]/prior_rate[i] + sum_j other_shape[j,k]/other_rate[j,k] */
void FN(orc_rate_update)(int n, int m, int K, const REAL *pvs, const REAL *pvr,
                         const REAL *olvs, const REAL *olvr, REAL *out)
{
    REAL *sum = (REAL *)calloc((size_t)K, sizeof(REAL));
    for (long j = 0; j < m; ++j)                                         /* :167-170 */
        for (int k = 0; k < K; ++k) sum[k] += olvs[j * K + k] / olvr[j * K + k];
    for (long i = 0; i < n; ++i) {                                       /* :172-176 */
        REAL prior_e_x = pvs[i] / pvr[i];
        for (int k = 0; k < K; ++k) out[i * K + k] = prior_e_x + sum[k];
    }
    free(sum);
}

/* compute_capacity_rate_update, hpf_numba.py:180-188 (and the inline numpy
 * forms scHPF_.py:680,685,695,704,714): out[i] = prior_rate + sum_k shape/rate.
 * The reference loops k outer, i inner (:185-187); kept. */
void FN(orc_capacity_rate)(int n, int K, const REAL *shape, const REAL *rate, double prior_rate,
                           REAL *out)
{
    for (long i = 0; i < n; ++i) out[i] = (REAL)prior_rate;
    for (int k = 0; k < K; ++k)
        for (long i = 0; i < n; ++i) out[i] += shape[i * K + k] / rate[i * K + k];
}

/* compute_pois_llh, hpf_numba.py:24-51: per nonzero
 * x*log(sum_k E[theta]E[beta]) - sum_k(...) - gammaln(x+1). */
void FN(orc_pois_llh)(long nnz, int N, int G, int K, const REAL *x, const int *row, const int *col,
                      const REAL *ths, const REAL *thr, const REAL *bes, const REAL *ber,
                      REAL *out, int nthreads)
{
    REAL *et = (REAL *)malloc(sizeof(REAL) * (size_t)N * K);
    REAL *eb = (REAL *)malloc(sizeof(REAL) * (size_t)G * K);
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (long i = 0; i < (long)N * K; ++i) et[i] = ths[i] / thr[i];     /* :33-36 */
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (long i = 0; i < (long)G * K; ++i) eb[i] = bes[i] / ber[i];     /* :38-41 */
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (long i = 0; i < nnz; ++i) {                                     /* :44-50 */
        const REAL *t = et + (size_t)row[i] * K;
        const REAL *b = eb + (size_t)col[i] * K;
        REAL e_rate = 0;
        for (int k = 0; k < K; ++k) e_rate += t[k] * b[k];
        /* integer x promotes the expression to double (see orc_xphi) */
        out[i] = (REAL)((double)x[i] * (double)RLOG(e_rate) - (double)e_rate
                        - lgamma((double)x[i] + 1.0));
    }
    free(et);
    free(eb);
}

/* One CAVI iteration in the reference's default (non-batched) order,
 * scHPF_.py:657-714.  All arrays are updated in place.
 *   xphi_ws : (nnz, K) workspace.  If use_given_xphi != 0 it already holds
 *             X*phi (the t==0 random responsibilities, scHPF_.py:652-655) and
 *             compute_Xphi_data is skipped.
 *   freeze_genes (scHPF_.py:668,682,697) skips the gene block.
 *   simultaneous (scHPF_.py:666-685): gene updates computed from the OLD theta
 *             but assigned after the cell updates.
 *   scatter_threads: see orc_shape_update (1 = the reference's serial scatter-adds).
 * eta_shape / xi_shape are the constants of scHPF_.py:616-618. */
void FN(orc_cavi_iteration)(long nnz, int N, int G, int K, const REAL *x, const int *row,
                            const int *col, double a, double c, double bp, double dp,
                            REAL *xis, REAL *xir, REAL *ths, REAL *thr, REAL *ets, REAL *etr,
                            REAL *bes, REAL *ber, REAL *xphi_ws, int use_given_xphi,
                            int freeze_genes, int simultaneous, int nthreads, int scatter_threads)
{
    if (!use_given_xphi)
        FN(orc_xphi)(nnz, N, G, K, x, row, col, ths, thr, bes, ber, xphi_ws, nthreads);

    if (simultaneous) {
        REAL *bvs = NULL, *bvr = NULL;
        if (!freeze_genes) {                                             /* :668-673 */
            bvs = (REAL *)malloc(sizeof(REAL) * (size_t)G * K);
            bvr = (REAL *)malloc(sizeof(REAL) * (size_t)G * K);
            FN(orc_shape_update)(nnz, K, xphi_ws, col, G, c, bvs, scatter_threads);
            FN(orc_rate_update)(G, N, K, ets, etr, ths, thr, bvr);
        }
        REAL *tvs = (REAL *)malloc(sizeof(REAL) * (size_t)N * K);         /* :675-680 */
        REAL *tvr = (REAL *)malloc(sizeof(REAL) * (size_t)N * K);
        FN(orc_shape_update)(nnz, K, xphi_ws, row, N, a, tvs, scatter_threads);
        FN(orc_rate_update)(N, G, K, xis, xir, bes, ber, tvr);
        memcpy(ths, tvs, sizeof(REAL) * (size_t)N * K);
        memcpy(thr, tvr, sizeof(REAL) * (size_t)N * K);
        free(tvs);
        free(tvr);
        FN(orc_capacity_rate)(N, K, ths, thr, bp, xir);
        if (!freeze_genes) {                                             /* :682-685 */
            memcpy(bes, bvs, sizeof(REAL) * (size_t)G * K);
            memcpy(ber, bvr, sizeof(REAL) * (size_t)G * K);
            FN(orc_capacity_rate)(G, K, bes, ber, dp, etr);
            free(bvs);
            free(bvr);
        }
        return;
    }

    if (!freeze_genes) {                                                 /* :697-704 */
        REAL *bvr = (REAL *)malloc(sizeof(REAL) * (size_t)G * K);
        FN(orc_shape_update)(nnz, K, xphi_ws, col, G, c, bes, scatter_threads);
        FN(orc_rate_update)(G, N, K, ets, etr, ths, thr, bvr);          /* OLD theta */
        memcpy(ber, bvr, sizeof(REAL) * (size_t)G * K);
        free(bvr);
        FN(orc_capacity_rate)(G, K, bes, ber, dp, etr);
    }
    {                                                                    /* :706-714 */
        REAL *tvr = (REAL *)malloc(sizeof(REAL) * (size_t)N * K);
        FN(orc_shape_update)(nnz, K, xphi_ws, row, N, a, ths, scatter_threads);
        FN(orc_rate_update)(N, G, K, xis, xir, bes, ber, tvr);          /* NEW beta */
        memcpy(thr, tvr, sizeof(REAL) * (size_t)N * K);
        free(tvr);
        FN(orc_capacity_rate)(N, K, ths, thr, bp, xir);
    }
}

#undef FN
#undef CAT
#undef CAT_
