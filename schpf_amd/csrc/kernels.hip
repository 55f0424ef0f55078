// HIP kernels for the scHPF CAVI hot path on gfx950 (MI355X, CDNA4).
//
// Replaces, on the device, the five numba functions of the reference
// (schpf/hpf_numba.py:24-188) and the update sequence that drives them
// (schpf/scHPF_.py:657-714).  Wave = 64 lanes everywhere; no MFMA (the path has
// no dense contraction); the roofline that bounds it is HBM / L2 gather.
//
// This file: the fused Gamma update, the reductions, the t=0 host-responsibilities path
// and the stateless operator mirrors.  The sweep kernels are in sweep_impl.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <climits>

#include "kernels.h"
#include "special.h"

namespace schpf {

// order-preserving integer image of a float (and back): integer max instead of canonicalising v_max_f64 pairs
__device__ __forceinline__ int float_order_key(float f)
{
    const int b = __float_as_int(f);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float float_from_order_key(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

// ------------------------------------------------------- fused Gamma update + tables
// One thread per (row, factor).  Replaces, for one side (theta or beta) and in one
// launch: compute_loading_shape_update (hpf_numba.py:128-156; here only the fixed-order
// reduction of the sweep's chunk partials), compute_loading_rate_update (:159-177), the
// capacity-rate line (scHPF_.py:704 / :714), and the E[log x] precompute of the NEXT
// iteration's compute_Xphi_data (:83-94) with its digamma.
//   shape = prior + acc;  rate = E[cap_old] + S_other[k];  cap_rate = cap_prior + sum_k E
//   E = shape/rate;  L = psi(shape) - log(rate);  Et = exp(L - max_k L)
// plus per-block column sums of E (the "sum over the other loading" of the other side).
//   * the special functions are special.h's: the digamma recurrence without its data-dependent loop, a series
//     logarithm and exponential, Newton reciprocals (1 / rate shared by E = shape / rate and by the logarithm's argument)
//     -- SQ_INSTS_VALU per launch -6 % (-30 % for the table refresh), wave cycles -14 % against the round-3 kernel with
//     libm's functions and the looped recurrence, at the SAME 26.6 us per launch: the kernel waits on its chain of
//     dependent memory round trips 60 % of its wave time (profiles/r06/update_kernel_counters_c3_f64.txt);
//   * WAVE_ROWS: a row's K threads never straddle a wavefront (64 / K rows per wave, rb = 4 * (64 / K) rows per
//     block: the same 12 at K = 20), so the row maximum / row sum walk through LDS needs no __syncthreads -- LDS
//     operations of one wave complete in order -- and the four waves of a block run independently inside the group
//     loop.  Only where at least 56 of a wave's 64 lanes stay busy (K = 20: 60; K = 50 keeps the block-wide form).
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// MINW: waves per SIMD the register allocation aims at.  8 (61 VGPRs; the series constants are scalar operands, special.h
// fma_c: before that 114): all 2048 blocks of a launch resident at once -- C3 f64 updates 0.0604 -> 0.0545 ms, the C5
// share 0.152 -> 0.130.  4 (69 VGPRs = seven waves) for the small problems whose blocks sum the other side's per-block
// column sums themselves (s_other_nb > 0: sixteen loads in flight in the prologue): C2 0.0178 vs 0.0189 at 8
// (profiles/r06/ab_update_waves.txt).
template <typename T, int SRC, bool WAVE_ROWS, int MINW>
__global__ __launch_bounds__(256, MINW) void gamma_update_kernel(UpdateArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int K = a.K, KP = a.KP, rb = a.rows_per_block;
    const int KS2 = (K + 1) & ~1, KS4 = (K + 3) & ~3;
    double *sE = lds;
    int *sKey = reinterpret_cast<int *>(lds + (size_t)rb * KS2);
    double *sS = lds + (size_t)rb * KS2 + (size_t)rb * KS4 / 2;
    const int t = threadIdx.x;
    int r, k;
    bool lane_on;
    if (WAVE_ROWS) {
        const int rpw = 64 / K, l = t & 63, rw = l / K;
        k = l - rw * K;
        r = (t >> 6) * rpw + rw;
        lane_on = rw < rpw;
    } else {
        r = t / K;
        k = t - r * K;
        lane_on = r < rb;
    }
    if (SRC != SRC_NONE && a.s_other_nb > 0) {
        // the other side's column sums from its per-block partials (small problems, capi.hip fuse_sums): thread (r, k)
        // takes blocks r, r + rb, ...; then factor k's rb values in turn -- the same fixed order in every block
        double p = 0.0;
        if (lane_on) {
            const double *__restrict__ src = a.s_other_part + k;
            const size_t st = (size_t)rb * K;
            double q[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            int b = r;
            for (; b + 7 * rb < a.s_other_nb; b += 8 * rb) {
                double v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = src[(size_t)b * K + j * st];
#pragma unroll
                for (int j = 0; j < 8; ++j) q[j] += v[j];
            }
            for (; b < a.s_other_nb; b += rb) q[0] += src[(size_t)b * K];
            p = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
        }
        if (lane_on) sE[r * K + k] = p;
        __syncthreads();
        if (t < K) {
            double tot = 0.0;
            for (int q = 0; q < rb; ++q) tot += sE[q * K + t];
            sS[t] = tot;
        }
        __syncthreads();
    }
    const double s_other_k = (SRC != SRC_NONE && lane_on)
                                 ? (a.s_other_nb > 0 ? sS[k] : (a.s_other_t ? (double)a.s_other_t[k] : a.s_other[k]))
                                 : 0.0;
    const int groups = (a.n + rb - 1) / rb;
    if (lane_on && k == 0) {   // the rows' padding: neutral for the sum and for the maximum
        for (int q = K; q < KS2; ++q) sE[r * KS2 + q] = 0.0;
        for (int q = K; q < KS4; ++q) sKey[r * KS4 + q] = INT_MIN;
    }
    if (WAVE_ROWS) wave_lds_fence(); else __syncthreads();
    double csum = 0.0;   // this thread's share of the block's column sum of E
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int row = grp * rb + r;
        const bool on = lane_on && row < a.n;
        double E = 0.0, L = 0.0;
        if (on) {
            double shape, rate;
            if (SRC == SRC_NONE) {
                shape = (double)a.shape[(size_t)row * K + k];
                rate = (double)a.rate[(size_t)row * K + k];
            } else {
                double acc = 0.0;
                if (SRC == SRC_PARTIALS) {
                    acc = sum_strided(a.partials + (size_t)a.cptr[row] * KP + k, a.cptr[row + 1] - a.cptr[row],
                                      (size_t)KP);
                } else if (SRC == SRC_STRIDED) {
                    acc = sum_strided(a.partials + (size_t)a.pfirst[row] * KP + k, a.pcount[row],
                                      (size_t)a.pstride * KP);
                } else {
                    acc = (double)a.dense[(size_t)row * K + k];
                }
                shape = a.prior_shape + acc;
                rate = (double)a.cap_shape[row] * fast_rcp((double)a.cap_rate[row]) + s_other_k;
                a.shape[(size_t)row * K + k] = (T)shape;
                a.rate[(size_t)row * K + k] = (T)rate;
                shape = (double)(T)shape;  // tables follow the stored (rounded) parameters
                rate = (double)(T)rate;
            }
            const double inv_rate = fast_rcp(rate);
            E = shape * inv_rate;
            L = digamma_less_log(shape, inv_rate);   // psi in double whatever T is (hpf_numba.py:16-18)
            a.tab_e[(size_t)row * KP + k] = (T)E;
            a.tab_log[(size_t)row * KP + k] = (T)L;
            E = (double)(T)E;
            L = (double)(T)L;
            sE[r * KS2 + k] = E;
            sKey[r * KS4 + k] = float_order_key((float)L);
        }
        if (WAVE_ROWS) wave_lds_fence(); else __syncthreads();
        if (on) {
            // every thread of a row walks the row once: the shift of the exponentials (the row's largest L, rounded to
            // float -- any shift within a few units of the maximum serves, it cancels in phi) and the row's sum of E
            const double2 *__restrict__ e2 = reinterpret_cast<const double2 *>(sE + r * KS2);
            const int4 *__restrict__ k4 = reinterpret_cast<const int4 *>(sKey + r * KS4);
            int mk = INT_MIN;
            double sum = 0.0;
#pragma unroll 4
            for (int q = 0; q < KS2 / 2; ++q) {
                const double2 v = e2[q];
                sum += v.x;
                sum += v.y;
            }
#pragma unroll 2
            for (int q = 0; q < KS4 / 4; ++q) {
                const int4 v = k4[q];
                mk = max(mk, max(max(v.x, v.y), max(v.z, v.w)));
            }
            a.tab_exp[(size_t)row * KP + k] = (T)fast_exp(L - (double)float_from_order_key(mk));
            if (k == 0 && SRC != SRC_NONE) a.cap_rate_out[row] = (T)(a.cap_prior_rate + sum);
        }
        csum += E;
        if (WAVE_ROWS) wave_lds_fence(); else __syncthreads();
    }
    // column sums of E over the block's rows: one pass over the rb shares at the end instead of one per group
    __syncthreads();
    if (lane_on) sE[r * K + k] = csum;
    __syncthreads();
    if (t < K) {
        double c = 0.0;
        for (int q = 0; q < rb; ++q) c += sE[q * K + t];
        a.colsum_part[(size_t)blockIdx.x * K + t] = c;
    }
}

// colsum_part [nblocks, K] -> out[K] (double), fixed order.  One workgroup per factor: every
// thread has its (few) loads in flight at once, then a fixed-shape tree in LDS -- one memory
// round trip instead of a serial walk over the blocks.
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const double *__restrict__ part, int nblocks,
                                                             int K, double *__restrict__ out,
                                                             void *mirror, int mirror_is_f32)
{
    __shared__ double red[256];
    const int t = threadIdx.x, k = blockIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = t;
    for (; b + 768 < nblocks; b += 1024) {
        const double v0 = part[(size_t)b * K + k], v1 = part[(size_t)(b + 256) * K + k];
        const double v2 = part[(size_t)(b + 512) * K + k], v3 = part[(size_t)(b + 768) * K + k];
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; b < nblocks; b += 256) s0 += part[(size_t)b * K + k];
    red[t] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (t < m) red[t] += red[t + m];
        __syncthreads();
    }
    if (t == 0) {
        const double tot = red[0];
        out[k] = tot;
        if (mirror) {
            if (mirror_is_f32) ((float *)mirror)[k] = (float)tot;
            else ((double *)mirror)[k] = tot;
        }
    }
}

// Reduce the chunk partials of every row in fixed order into a dense [n, K] matrix
// (the gene-side accumulator that is all-reduced across GPUs when cells are sharded).
template <typename T>
__global__ __launch_bounds__(256) void combine_partials_kernel(const T *__restrict__ partials,
                                                               const int *__restrict__ cptr, int n, int K,
                                                               int KP, T *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * K) return;
    const int row = (int)(i / K), k = (int)(i - (size_t)row * K);
    out[i] = (T)sum_strided(partials + (size_t)cptr[row] * KP + k, cptr[row + 1] - cptr[row], (size_t)KP);
}

template <typename T>
__global__ __launch_bounds__(256) void combine_strided_kernel(const T *__restrict__ partials,
                                                              const int *__restrict__ pfirst,
                                                              const int *__restrict__ pcount, int64_t pstride,
                                                              int n, int K, int KP, T *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * K) return;
    const int row = (int)(i / K), k = (int)(i - (size_t)row * K);
    out[i] = (T)sum_strided(partials + (size_t)pfirst[row] * KP + k, pcount[row], (size_t)pstride * KP);
}

// sum of n doubles -> out[0]; single block, fixed order.
__global__ __launch_bounds__(256) void sum_doubles_kernel(const double *__restrict__ v, int64_t n,
                                                          double *__restrict__ out)
{
    __shared__ double red[256];
    double s = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t i = threadIdx.x;
    for (; i + 768 < n; i += 1024) {
        const double v0 = v[i], v1 = v[i + 256], v2 = v[i + 512], v3 = v[i + 768];
        s += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; i < n; i += 256) s += v[i];
    red[threadIdx.x] = (s + s1) + (s2 + s3);
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// sum_i lgamma(x_i + 1) over the stored counts (the constant term of the loss,
// hpf_numba.py:49-50): per-block partials, reduced by sum_doubles_kernel.
__global__ __launch_bounds__(256) void gammaln_sum_kernel(const float *__restrict__ x, int64_t n,
                                                          double *__restrict__ block_out)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        s += lgamma((double)x[i] + 1.0);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_out[blockIdx.x] = red[0];
}

// sum over the explicitly stored zeros of r = sum_k E[theta_ik] E[beta_gk]: their whole term of the
// loss (hpf_numba.py:43-50 with x = 0).  One block, fixed order; real matrices have none or few.
template <typename T>
__global__ __launch_bounds__(256) void zero_rate_sum_kernel(const int *__restrict__ row, const int *__restrict__ col,
                                                            int64_t n, const T *__restrict__ et,
                                                            const T *__restrict__ eb, int K, int KP,
                                                            double *__restrict__ out)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const T *t = et + (size_t)row[i] * KP;
        const T *b = eb + (size_t)col[i] * KP;
        T r = T(0);
        for (int k = 0; k < K; ++k) r += t[k] * b[k];
        s += (double)r;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// ------------------------------------------------------ t = 0 random responsibilities
// scHPF_.py:652-655: X*phi with phi ~ Dirichlet(1_K), drawn by the caller (NumPy global
// RNG, for seed parity) and uploaded as (nnz, K) float64 in the caller's COO order.
// acc[row, k] = sum over the row's nonzeros, in sorted order (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void segment_sum_kernel(const double *__restrict__ xphi,
                                                          const int *__restrict__ order,
                                                          const int64_t *__restrict__ mptr, int n, int K,
                                                          T *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * K) return;
    const int row = (int)(i / K), k = (int)(i - (size_t)row * K);
    double acc = 0.0;
    for (int64_t j = mptr[row]; j < mptr[row + 1]; ++j) acc += xphi[(size_t)order[j] * K + k];
    out[i] = (T)acc;
}

// ------------------------------------------------------ stateless operator mirrors
// Array-in / array-out counterparts of the reference's numba callables, in the
// caller's COO order.  Used by schpf_amd.hpf_hip and by the parity tests.

// E[log x] table: psi(shape) - log(rate)  (hpf_numba.py:83-94)
template <typename T>
__global__ __launch_bounds__(256) void elog_kernel(const T *__restrict__ shape, const T *__restrict__ rate,
                                                   int64_t n, T *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (T)(digamma((double)shape[i]) - log((double)rate[i]));
}
template <typename T>
__global__ __launch_bounds__(256) void ratio_kernel(const T *__restrict__ shape, const T *__restrict__ rate,
                                                    int64_t n, T *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = shape[i] / rate[i];
}

// compute_Xphi_data (hpf_numba.py:97-112), max-shifted softmax exactly as the reference.
template <typename T>
__global__ __launch_bounds__(256) void xphi_coo_kernel(const T *__restrict__ x, const int *__restrict__ row,
                                                       const int *__restrict__ col, const T *__restrict__ elt,
                                                       const T *__restrict__ elb, int64_t nnz, int K,
                                                       T *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nnz) return;
    const T *t = elt + (size_t)row[i] * K;
    const T *b = elb + (size_t)col[i] * K;
    T *o = out + (size_t)i * K;
    T mx = t[0] + b[0];
    for (int k = 1; k < K; ++k) {
        T v = t[k] + b[k];
        mx = v > mx ? v : mx;
    }
    T norm = T(0);
    for (int k = 0; k < K; ++k) {
        T v = (T)exp((double)(t[k] + b[k] - mx));
        o[k] = v;
        norm += v;
    }
    const double xv = (double)x[i];
    for (int k = 0; k < K; ++k) o[k] = (T)(xv * (double)o[k] / (double)norm);
}

// compute_pois_llh (hpf_numba.py:43-50)
template <typename T>
__global__ __launch_bounds__(256) void llh_coo_kernel(const T *__restrict__ x, const int *__restrict__ row,
                                                      const int *__restrict__ col, const T *__restrict__ et,
                                                      const T *__restrict__ eb, int64_t nnz, int K,
                                                      T *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nnz) return;
    const T *t = et + (size_t)row[i] * K;
    const T *b = eb + (size_t)col[i] * K;
    T r = T(0);
    for (int k = 0; k < K; ++k) r += t[k] * b[k];
    const double xv = (double)x[i];
    out[i] = (T)(xv * log((double)r) - (double)r - lgamma(xv + 1.0));
}

// compute_loading_shape_update (hpf_numba.py:128-156): rows of Xphi summed per index in
// sorted (deterministic) order on top of the prior.
template <typename T>
__global__ __launch_bounds__(256) void shape_update_kernel(const T *__restrict__ xphi,
                                                           const int *__restrict__ order,
                                                           const int64_t *__restrict__ ptr, int n, int K,
                                                           double prior, T *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * K) return;
    const int row = (int)(i / K), k = (int)(i - (size_t)row * K);
    T acc = (T)prior;
    for (int64_t j = ptr[row]; j < ptr[row + 1]; ++j) acc += xphi[(size_t)order[j] * K + k];
    out[i] = acc;
}

// column sums of shape/rate over m rows: per-block partials (double)
template <typename T>
__global__ __launch_bounds__(256) void ratio_colsum_kernel(const T *__restrict__ shape,
                                                           const T *__restrict__ rate, int m, int K,
                                                           double *__restrict__ part)
{
    extern __shared__ double lds[];
    const int rb = 256 / K;
    const int t = threadIdx.x, r = t / K, k = t - r * K;
    double s = 0.0;
    if (r < rb)
        for (int row = blockIdx.x * rb + r; row < m; row += gridDim.x * rb)
            s += (double)(shape[(size_t)row * K + k] / rate[(size_t)row * K + k]);
    lds[t] = (r < rb) ? s : 0.0;
    __syncthreads();
    if (t < K) {
        double c = 0.0;
        for (int q = 0; q < rb; ++q) c += lds[q * K + t];
        part[(size_t)blockIdx.x * K + t] = c;
    }
}
// compute_loading_rate_update (hpf_numba.py:172-176): out[i,k] = ps[i]/pr[i] + S[k]
template <typename T>
__global__ __launch_bounds__(256) void rate_update_kernel(const T *__restrict__ ps, const T *__restrict__ pr,
                                                          const double *__restrict__ S, int n, int K,
                                                          T *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * K) return;
    const int row = (int)(i / K), k = (int)(i - (size_t)row * K);
    out[i] = (T)((double)(ps[row] / pr[row]) + S[k]);
}
// compute_capacity_rate_update (hpf_numba.py:180-188)
template <typename T>
__global__ __launch_bounds__(256) void capacity_rate_kernel(const T *__restrict__ shape,
                                                            const T *__restrict__ rate, int n, int K,
                                                            double prior, T *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    T acc = (T)prior;
    for (int k = 0; k < K; ++k) acc += shape[(size_t)i * K + k] / rate[(size_t)i * K + k];
    out[i] = acc;
}

__global__ __launch_bounds__(256) void digamma_array_kernel(const double *__restrict__ x, int64_t n,
                                                            double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = digamma(x[i]);
}
__global__ __launch_bounds__(256) void gammaln_array_kernel(const double *__restrict__ x, int64_t n,
                                                            double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = lgamma(x[i]);
}

// ------------------------------------------------------------------------ launchers
// never a zero-sized grid: every kernel bounds-checks, an empty problem launches one idle block
static inline unsigned blocks_for(int64_t n) { return n > 0 ? (unsigned)((n + 255) / 256) : 1u; }

static bool update_wave_rows(int K) { return K <= 64 && (64 / K) * K >= 56; }
int update_rows_per_block(int K) { return update_wave_rows(K) ? 4 * (64 / K) : 256 / K; }

template <typename T, bool WR, int MINW> static void launch_update_w(const UpdateArgs<T> &a, int src, dim3 grid, size_t lds, hipStream_t st)
{
    dim3 block(256);
    if (src == SRC_NONE) hipLaunchKernelGGL((gamma_update_kernel<T, SRC_NONE, WR, MINW>), grid, block, lds, st, a);
    else if (src == SRC_PARTIALS) hipLaunchKernelGGL((gamma_update_kernel<T, SRC_PARTIALS, WR, MINW>), grid, block, lds, st, a);
    else if (src == SRC_STRIDED) hipLaunchKernelGGL((gamma_update_kernel<T, SRC_STRIDED, WR, MINW>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((gamma_update_kernel<T, SRC_DENSE, WR, MINW>), grid, block, lds, st, a);
}
template <typename T, bool WR> static void launch_update_t(const UpdateArgs<T> &a, int src, dim3 grid, size_t lds, hipStream_t st)
{
    if (a.s_other_nb > 0) launch_update_w<T, WR, 4>(a, src, grid, lds, st);
    else launch_update_w<T, WR, 8>(a, src, grid, lds, st);
}
template <typename T> hipError_t launch_gamma_update(const UpdateArgs<T> &a, int src, int nblocks, hipStream_t st)
{
    const size_t lds = ((size_t)2 * a.rows_per_block * (a.K + 3) + 2 * a.K) * sizeof(double);   // E rows, key rows, sums (padded strides)
    if (a.rows_per_block != update_rows_per_block(a.K)) return hipErrorInvalidValue;
    dim3 grid((unsigned)nblocks);
    if (update_wave_rows(a.K)) launch_update_t<T, true>(a, src, grid, lds, st);
    else launch_update_t<T, false>(a, src, grid, lds, st);
    return hipGetLastError();
}

hipError_t launch_colsum_reduce(const double *part, int nblocks, int K, double *out, void *mirror,
                                int mirror_is_f32, hipStream_t st)
{
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)K), dim3(256), 0, st, part, nblocks, K, out, mirror,
                       mirror_is_f32);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_combine_partials(const T *partials, const int *cptr, int n, int K, int KP, T *out,
                                   hipStream_t st)
{
    hipLaunchKernelGGL((combine_partials_kernel<T>), dim3(blocks_for((int64_t)n * K)), dim3(256), 0, st,
                       partials, cptr, n, K, KP, out);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_combine_strided(const T *partials, const int *pfirst, const int *pcount, int64_t pstride, int n,
                                  int K, int KP, T *out, hipStream_t st)
{
    hipLaunchKernelGGL((combine_strided_kernel<T>), dim3(blocks_for((int64_t)n * K)), dim3(256), 0, st,
                       partials, pfirst, pcount, pstride, n, K, KP, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void gather_by_order_kernel(const int *__restrict__ order, const int *__restrict__ col,
                                                              const float *__restrict__ val, int64_t nnz,
                                                              int *__restrict__ out_col, float *__restrict__ out_val)
{
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * 256) {
        const int64_t src = order ? (int64_t)order[j] : j;
        out_col[j] = col[src];
        out_val[j] = val[src];
    }
}
hipError_t launch_gather_by_order(const int *order, const int *col, const float *val, int64_t nnz, int *out_col,
                                  float *out_val, hipStream_t st)
{
    if (nnz <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((nnz + 255) / 256 < 65536 ? (nnz + 255) / 256 : 65536);
    hipLaunchKernelGGL(gather_by_order_kernel, dim3(grid), dim3(256), 0, st, order, col, val, nnz, out_col, out_val);
    return hipGetLastError();
}
// one wavefront per batch row: coalesced copy of the row's run
__global__ __launch_bounds__(256) void gather_rows_kernel(const int *__restrict__ rows, int n_rows,
                                                          const int64_t *__restrict__ src_ptr, const int *__restrict__ src_col,
                                                          const float *__restrict__ src_val, const int64_t *__restrict__ dst_ptr,
                                                          int *__restrict__ out_row, int *__restrict__ out_col,
                                                          float *__restrict__ out_val)
{
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n_rows; i += gridDim.x * 4) {
        const int64_t s0 = src_ptr[rows[i]], len = src_ptr[rows[i] + 1] - s0, d0 = dst_ptr[i];
        for (int64_t j = lane; j < len; j += 64) {
            out_row[d0 + j] = i;
            out_col[d0 + j] = src_col[s0 + j];
            out_val[d0 + j] = src_val[s0 + j];
        }
    }
}
hipError_t launch_gather_rows(const int *rows, int n_rows, const int64_t *src_ptr, const int *src_col,
                              const float *src_val, const int64_t *dst_ptr, int *out_row, int *out_col,
                              float *out_val, hipStream_t st)
{
    if (n_rows <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((n_rows + 3) / 4 < 65536 ? (n_rows + 3) / 4 : 65536);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, st, rows, n_rows, src_ptr, src_col, src_val,
                       dst_ptr, out_row, out_col, out_val);
    return hipGetLastError();
}

hipError_t launch_sum_doubles(const double *v, int64_t n, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(sum_doubles_kernel, dim3(1), dim3(256), 0, st, v, n, out);
    return hipGetLastError();
}
hipError_t launch_gammaln_sum(const float *x, int64_t n, double *block_out, int nblocks, hipStream_t st)
{
    hipLaunchKernelGGL(gammaln_sum_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, x, n, block_out);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_zero_rate_sum(const int *row, const int *col, int64_t n, const T *et, const T *eb, int K, int KP,
                                double *out, hipStream_t st)
{
    hipLaunchKernelGGL((zero_rate_sum_kernel<T>), dim3(1), dim3(256), 0, st, row, col, n, et, eb, K, KP, out);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_segment_sum(const double *xphi, const int *order, const int64_t *mptr, int n, int K, T *out,
                              hipStream_t st)
{
    hipLaunchKernelGGL((segment_sum_kernel<T>), dim3(blocks_for((int64_t)n * K)), dim3(256), 0, st, xphi,
                       order, mptr, n, K, out);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_elog(const T *shape, const T *rate, int64_t n, T *out, hipStream_t st)
{
    hipLaunchKernelGGL((elog_kernel<T>), dim3(blocks_for(n)), dim3(256), 0, st, shape, rate, n, out);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_ratio(const T *shape, const T *rate, int64_t n, T *out, hipStream_t st)
{
    hipLaunchKernelGGL((ratio_kernel<T>), dim3(blocks_for(n)), dim3(256), 0, st, shape, rate, n, out);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_xphi_coo(const T *x, const int *row, const int *col, const T *elt, const T *elb, int64_t nnz,
                           int K, T *out, hipStream_t st)
{
    hipLaunchKernelGGL((xphi_coo_kernel<T>), dim3(blocks_for(nnz)), dim3(256), 0, st, x, row, col, elt, elb,
                       nnz, K, out);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_llh_coo(const T *x, const int *row, const int *col, const T *et, const T *eb, int64_t nnz,
                          int K, T *out, hipStream_t st)
{
    hipLaunchKernelGGL((llh_coo_kernel<T>), dim3(blocks_for(nnz)), dim3(256), 0, st, x, row, col, et, eb, nnz,
                       K, out);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_shape_update(const T *xphi, const int *order, const int64_t *ptr, int n, int K, double prior,
                               T *out, hipStream_t st)
{
    hipLaunchKernelGGL((shape_update_kernel<T>), dim3(blocks_for((int64_t)n * K)), dim3(256), 0, st, xphi,
                       order, ptr, n, K, prior, out);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_ratio_colsum(const T *shape, const T *rate, int m, int K, double *part, int nblocks,
                               hipStream_t st)
{
    hipLaunchKernelGGL((ratio_colsum_kernel<T>), dim3((unsigned)nblocks), dim3(256), 256 * sizeof(double), st,
                       shape, rate, m, K, part);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_rate_update(const T *ps, const T *pr, const double *S, int n, int K, T *out, hipStream_t st)
{
    hipLaunchKernelGGL((rate_update_kernel<T>), dim3(blocks_for((int64_t)n * K)), dim3(256), 0, st, ps, pr, S,
                       n, K, out);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_capacity_rate(const T *shape, const T *rate, int n, int K, double prior, T *out,
                                hipStream_t st)
{
    hipLaunchKernelGGL((capacity_rate_kernel<T>), dim3(blocks_for(n)), dim3(256), 0, st, shape, rate, n, K,
                       prior, out);
    return hipGetLastError();
}
hipError_t launch_digamma_array(const double *x, int64_t n, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(digamma_array_kernel, dim3(blocks_for(n)), dim3(256), 0, st, x, n, out);
    return hipGetLastError();
}
hipError_t launch_gammaln_array(const double *x, int64_t n, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(gammaln_array_kernel, dim3(blocks_for(n)), dim3(256), 0, st, x, n, out);
    return hipGetLastError();
}

// explicit instantiations for the two model dtypes
#define SCHPF_INSTANTIATE(T)                                                                                   \
    template hipError_t launch_gamma_update<T>(const UpdateArgs<T> &, int, int, hipStream_t);                  \
    template hipError_t launch_combine_partials<T>(const T *, const int *, int, int, int, T *, hipStream_t);   \
    template hipError_t launch_combine_strided<T>(const T *, const int *, const int *, int64_t, int, int, int,  \
                                                  T *, hipStream_t);                                           \
    template hipError_t launch_zero_rate_sum<T>(const int *, const int *, int64_t, const T *, const T *, int,  \
                                                int, double *, hipStream_t);                                  \
    template hipError_t launch_segment_sum<T>(const double *, const int *, const int64_t *, int, int, T *,     \
                                              hipStream_t);                                                    \
    template hipError_t launch_elog<T>(const T *, const T *, int64_t, T *, hipStream_t);                       \
    template hipError_t launch_ratio<T>(const T *, const T *, int64_t, T *, hipStream_t);                      \
    template hipError_t launch_xphi_coo<T>(const T *, const int *, const int *, const T *, const T *,          \
                                           int64_t, int, T *, hipStream_t);                                    \
    template hipError_t launch_llh_coo<T>(const T *, const int *, const int *, const T *, const T *, int64_t,  \
                                          int, T *, hipStream_t);                                              \
    template hipError_t launch_shape_update<T>(const T *, const int *, const int64_t *, int, int, double,      \
                                               T *, hipStream_t);                                              \
    template hipError_t launch_ratio_colsum<T>(const T *, const T *, int, int, double *, int, hipStream_t);    \
    template hipError_t launch_rate_update<T>(const T *, const T *, const double *, int, int, T *,             \
                                              hipStream_t);                                                    \
    template hipError_t launch_capacity_rate<T>(const T *, const T *, int, int, double, T *, hipStream_t);
SCHPF_INSTANTIATE(float)
SCHPF_INSTANTIATE(double)

}  // namespace schpf
