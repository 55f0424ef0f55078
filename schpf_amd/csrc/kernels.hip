// HIP kernels for the scHPF CAVI hot path on gfx950 (MI355X, CDNA4).
//
// Replaces, on the device, the five numba functions of the reference
// (schpf/hpf_numba.py:24-188) and the update sequence that drives them
// (schpf/scHPF_.py:657-714).  Wave = 64 lanes everywhere; no MFMA (the path has
// no dense contraction); the roofline that bounds it is HBM / L2 gather.
//
// Algebra used by the fused sweeps (DESIGN.md "restatement"):
//   phi_k = exp(Elt[i,k] + Elb[g,k]) / sum_k(...)  (hpf_numba.py:97-112)
//         = Et[i,k] * Eb[g,k] / sum_k Et[i,k] Eb[g,k],
//   Et[i,k] = exp(Elt[i,k] - max_k Elt[i,:]),  Eb likewise per gene,
// so exp() is evaluated (N + G) * K times per iteration instead of nnz * K, and
// X*phi (nnz x K, hpf_numba.py:97) is never materialised:
//   sum_g x phi_k = Et[i,k] * sum_g (x / s_ig) Eb[g,k],   s_ig = sum_k Et Eb.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace schpf {

// ------------------------------------------------------------------ special functions
// psi(x), x > 0: upward recurrence to x >= 10 then the Bernoulli asymptotic series
// (Cephes psi).  Replaces the SciPy C psi the reference binds (hpf_numba.py:16-18).
// Absolute error < 4e-16 on [1e-4, 1e6] against SciPy (tests/test_special_gpu.py).
__device__ __forceinline__ double dev_digamma(double x)
{
    double w = 0.0;
    while (x < 10.0) {
        w += 1.0 / x;
        x += 1.0;
    }
    const double z = 1.0 / (x * x);
    double p = 8.33333333333333333333E-2;
    p = p * z - 2.10927960927960927961E-2;
    p = p * z + 7.57575757575757575758E-3;
    p = p * z - 4.16666666666666666667E-3;
    p = p * z + 3.96825396825396825397E-3;
    p = p * z - 8.33333333333333333333E-3;
    p = p * z + 8.33333333333333333333E-2;
    return log(x) - 0.5 / x - z * p - w;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// streamed once per sweep: keep it out of the way of the gathered tables in L2
__device__ __forceinline__ uint4 stream_load(const uint4 *p)
{
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <typename T> struct Tiny;
template <> struct Tiny<double> { static __device__ __forceinline__ double v() { return 1e-280; } };
template <> struct Tiny<float> { static __device__ __forceinline__ float v() { return 1e-30f; } };

template <typename T, int LPC> __device__ __forceinline__ T group_sum(T v)
{
#pragma unroll
    for (int m = 1; m < LPC; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}
template <typename T, int LPC> __device__ __forceinline__ T group_max(T v)
{
#pragma unroll
    for (int m = 1; m < LPC; m <<= 1) {
        T o = __shfl_xor(v, m, 64);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// KL contiguous values, 16-byte vector loads (rows are KP*sizeof(T) = multiple of 16 B).
template <int KL> __device__ __forceinline__ void load_row(const float *__restrict__ p, float (&v)[KL])
{
#pragma unroll
    for (int q = 0; q < KL / 4; ++q) {
        float4 t = reinterpret_cast<const float4 *>(p)[q];
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
}
template <int KL> __device__ __forceinline__ void load_row(const double *__restrict__ p, double (&v)[KL])
{
#pragma unroll
    for (int q = 0; q < KL / 2; ++q) {
        double2 t = reinterpret_cast<const double2 *>(p)[q];
        v[2 * q] = t.x; v[2 * q + 1] = t.y;
    }
}
template <int KL> __device__ __forceinline__ void store_row(float *__restrict__ p, const float (&v)[KL])
{
#pragma unroll
    for (int q = 0; q < KL / 4; ++q)
        reinterpret_cast<float4 *>(p)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
template <int KL> __device__ __forceinline__ void store_row(double *__restrict__ p, const double (&v)[KL])
{
#pragma unroll
    for (int q = 0; q < KL / 2; ++q)
        reinterpret_cast<double2 *>(p)[q] = make_double2(v[2 * q], v[2 * q + 1]);
}

// ------------------------------------------------------------------------ the sweep
// One wavefront streams one slice of the plan (plan.h).  A group of LPC adjacent lanes
// owns one chunk (<= chunk_len nonzeros of one major row) and KL = KP / LPC factors
// each; the major's K-vector and the K accumulators live in registers, the minor's
// K-vector is gathered per nonzero (L2-resident window of the table).
//
// MODE_PHI : acc_k += (x / s) * Eb[minor,k];   out row = acc_k * Et[major,k]
//            = this chunk's share of sum x*phi_k (hpf_numba.py:97-112 fused with
//            :152-155).  Numerically degenerate nonzeros (s underflows) take the
//            reference's max-shifted log-domain form from the Elog tables and go to
//            `extra` with atomics (rare; flagged).
// MODE_LLH : sum over the chunk of x*log(r) - r, r = sum_k E[theta]E[beta]
//            (hpf_numba.py:43-50 minus the constant gammaln term); one double per wave.
template <typename T, int KL, int LPC, int MODE>
__global__ __launch_bounds__(256) void sweep_kernel(SweepArgs<T> a)
{
    constexpr int CPW = 64 / LPC;
    constexpr int KP = KL * LPC;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int slice = a.wave_slice[wave];
    if (slice < 0) {
        if (MODE == MODE_LLH && (threadIdx.x & 63) == 0) a.wave_out[wave] = 0.0;
        return;
    }
    const int lane = threadIdx.x & 63;
    const int slot = lane / LPC;
    const int sub = lane % LPC;
    const int major = a.chunk_major[(size_t)slice * CPW + slot];
    const bool live = major >= 0;

    T tm[KL];
    T acc[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) { tm[k] = T(0); acc[k] = T(0); }
    if (live) load_row<KL>(a.tab_major + (size_t)major * KP + sub * KL, tm);
    double llh = 0.0;

    const uint4 *__restrict__ ep = a.entries + a.slice_off[slice] + slot;
    const int steps = a.slice_steps[slice];
    const T *__restrict__ tabm = a.tab_minor + sub * KL;

    for (int p = 0; p < steps; ++p) {
        const uint4 e = stream_load(ep + (size_t)p * CPW);
        T b0[KL], b1[KL];
        load_row<KL>(tabm + (size_t)e.x * KP, b0);
        load_row<KL>(tabm + (size_t)e.z * KP, b1);
        const T x0 = (T)__uint_as_float(e.y);
        const T x1 = (T)__uint_as_float(e.w);
        T s0 = T(0), s1 = T(0);
#pragma unroll
        for (int k = 0; k < KL; ++k) { s0 += tm[k] * b0[k]; s1 += tm[k] * b1[k]; }
        s0 = group_sum<T, LPC>(s0);
        s1 = group_sum<T, LPC>(s1);
        if (MODE == MODE_PHI) {
            const bool bad0 = x0 > T(0) && !(s0 >= Tiny<T>::v());
            const bool bad1 = x1 > T(0) && !(s1 >= Tiny<T>::v());
            const T w0 = (x0 > T(0) && !bad0) ? x0 / s0 : T(0);
            const T w1 = (x1 > T(0) && !bad1) ? x1 / s1 : T(0);
#pragma unroll
            for (int k = 0; k < KL; ++k) acc[k] += w0 * b0[k] + w1 * b1[k];
            if (__builtin_expect(bad0 || bad1, 0)) {
                // log-domain fallback, the reference's own form (hpf_numba.py:98-112)
#pragma unroll 1
                for (int u = 0; u < 2; ++u) {
                    const bool bad = u ? bad1 : bad0;
                    if (!bad) continue;
                    const unsigned idx = u ? e.z : e.x;
                    const T x = u ? x1 : x0;
                    T lr[KL];
                    T lm[KL];
                    load_row<KL>(a.log_major + (size_t)major * KP + sub * KL, lr);
                    load_row<KL>(a.log_minor + (size_t)idx * KP + sub * KL, lm);
                    T mx = -INFINITY;
#pragma unroll
                    for (int k = 0; k < KL; ++k) {
                        lr[k] += lm[k];
                        if (sub * KL + k < a.K) mx = lr[k] > mx ? lr[k] : mx;
                    }
                    mx = group_max<T, LPC>(mx);
                    T ss = T(0);
#pragma unroll
                    for (int k = 0; k < KL; ++k) {
                        lr[k] = (sub * KL + k < a.K) ? (T)exp((double)(lr[k] - mx)) : T(0);
                        ss += lr[k];
                    }
                    ss = group_sum<T, LPC>(ss);
#pragma unroll
                    for (int k = 0; k < KL; ++k)
                        if (sub * KL + k < a.K)
                            atomicAdd(a.extra + (size_t)major * KP + sub * KL + k, x * lr[k] / ss);
                    *a.extra_flag = 1;
                }
            }
        } else {
            if (x0 > T(0)) llh += (double)x0 * log((double)s0) - (double)s0;
            if (x1 > T(0)) llh += (double)x1 * log((double)s1) - (double)s1;
        }
    }

    if (MODE == MODE_PHI) {
        if (live) {
#pragma unroll
            for (int k = 0; k < KL; ++k) acc[k] *= tm[k];
            const int nat = a.chunk_natid[(size_t)slice * CPW + slot];
            store_row<KL>(a.partials + (size_t)nat * KP + sub * KL, acc);
        }
    } else {
        if (sub != 0) llh = 0.0;
        llh = wave_sum(llh);
        if (lane == 0) a.wave_out[wave] = llh;
    }
}

// ------------------------------------------------------- fused Gamma update + tables
// One thread per (row, factor).  Replaces, for one side (theta or beta) and in one
// launch: compute_loading_shape_update (hpf_numba.py:128-156; here only the fixed-order
// reduction of the sweep's chunk partials), compute_loading_rate_update (:159-177), the
// capacity-rate line (scHPF_.py:704 / :714), and the E[log x] precompute of the NEXT
// iteration's compute_Xphi_data (:83-94) with its digamma.
//   shape = prior + acc;  rate = E[cap_old] + S_other[k];  cap_rate = cap_prior + sum_k E
//   E = shape/rate;  L = psi(shape) - log(rate);  Et = exp(L - max_k L)
// plus per-block column sums of E (the "sum over the other loading" of the other side).
template <typename T, int SRC>
__global__ __launch_bounds__(256) void gamma_update_kernel(UpdateArgs<T> a)
{
    extern __shared__ double lds[];  // [rb*K] E, [rb*K] L, [K] column sums
    const int K = a.K, KP = a.KP, rb = a.rows_per_block;
    double *sE = lds;
    double *sL = lds + (size_t)rb * K;
    double *sC = sL + (size_t)rb * K;
    const int t = threadIdx.x;
    const int r = t / K, k = t - r * K;
    const bool lane_on = r < rb;
    if (t < K) sC[t] = 0.0;
    const int groups = (a.n + rb - 1) / rb;
    const bool use_extra = (SRC == SRC_PARTIALS || SRC == SRC_DENSE) && a.extra_flag && *a.extra_flag;
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int row = grp * rb + r;
        const bool on = lane_on && row < a.n;
        double E = 0.0, L = -INFINITY;
        if (on) {
            double shape, rate;
            if (SRC == SRC_NONE) {
                shape = (double)a.shape[(size_t)row * K + k];
                rate = (double)a.rate[(size_t)row * K + k];
            } else {
                double acc = 0.0;
                if (SRC == SRC_PARTIALS) {
                    const int c0 = a.cptr[row], c1 = a.cptr[row + 1];
                    for (int c = c0; c < c1; ++c) acc += (double)a.partials[(size_t)c * KP + k];
                } else {
                    acc = (double)a.dense[(size_t)row * K + k];
                }
                if (use_extra) {
                    acc += (double)a.extra[(size_t)row * KP + k];
                    a.extra[(size_t)row * KP + k] = T(0);  // self-cleaning
                }
                shape = a.prior_shape + acc;
                rate = (double)a.cap_shape[row] / (double)a.cap_rate[row] + a.s_other[k];
                a.shape[(size_t)row * K + k] = (T)shape;
                a.rate[(size_t)row * K + k] = (T)rate;
                shape = (double)(T)shape;  // tables follow the stored (rounded) parameters
                rate = (double)(T)rate;
            }
            E = shape / rate;
            L = dev_digamma(shape) - log(rate);
            a.tab_e[(size_t)row * KP + k] = (T)E;
            a.tab_log[(size_t)row * KP + k] = (T)L;
            E = (double)(T)E;
            L = (double)(T)L;
            sE[r * K + k] = E;
            sL[r * K + k] = L;
        }
        __syncthreads();
        if (on) {
            double mx = sL[r * K];
            for (int q = 1; q < K; ++q) mx = fmax(mx, sL[r * K + q]);
            a.tab_exp[(size_t)row * KP + k] = (T)exp(L - mx);
            if (k == 0 && SRC != SRC_NONE) {
                double sum = 0.0;
                for (int q = 0; q < K; ++q) sum += sE[r * K + q];
                a.cap_rate_out[row] = (T)(a.cap_prior_rate + sum);
            }
        }
        if (t < K) {
            const int nr = min(rb, a.n - grp * rb);
            double c = 0.0;
            for (int q = 0; q < nr; ++q) c += sE[q * K + t];
            sC[t] += c;
        }
        __syncthreads();
    }
    if (t < K) a.colsum_part[(size_t)blockIdx.x * K + t] = sC[t];
}

// colsum_part [nblocks, K] -> out[K] (double), fixed order.
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const double *__restrict__ part, int nblocks,
                                                             int K, double *__restrict__ out,
                                                             void *mirror, int mirror_is_f32)
{
    __shared__ double red[256];
    const int t = threadIdx.x;
    const int lanes = 256 / K;          // partial accumulators per factor
    const int k = t % K, j = t / K;
    double s = 0.0;
    if (j < lanes)
        for (int b = j; b < nblocks; b += lanes) s += part[(size_t)b * K + k];
    red[t] = s;
    __syncthreads();
    if (t < K) {
        double tot = 0.0;
        for (int q = 0; q < lanes; ++q) tot += red[q * K + t];
        out[t] = tot;
        if (mirror) {
            if (mirror_is_f32) ((float *)mirror)[t] = (float)tot;
            else ((double *)mirror)[t] = tot;
        }
    }
}

// Reduce the chunk partials of every row in fixed order into a dense [n, K] matrix
// (the gene-side accumulator that is all-reduced across GPUs when cells are sharded).
template <typename T>
__global__ __launch_bounds__(256) void combine_partials_kernel(const T *__restrict__ partials,
                                                               const int *__restrict__ cptr, int n, int K,
                                                               int KP, T *__restrict__ extra,
                                                               const int *__restrict__ extra_flag,
                                                               T *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * K) return;
    const int row = (int)(i / K), k = (int)(i - (size_t)row * K);
    double acc = 0.0;
    for (int c = cptr[row]; c < cptr[row + 1]; ++c) acc += (double)partials[(size_t)c * KP + k];
    if (extra_flag && *extra_flag) {
        acc += (double)extra[(size_t)row * KP + k];
        extra[(size_t)row * KP + k] = T(0);
    }
    out[i] = (T)acc;
}

// sum of n doubles -> out[0]; single block, fixed order.
__global__ __launch_bounds__(256) void sum_doubles_kernel(const double *__restrict__ v, int64_t n,
                                                          double *__restrict__ out)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += v[i];
    red[threadIdx.x] = s;
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

// Device-side variant for matrices too large for a host draw: phi_k = e_k / sum e,
// e_k ~ Exp(1) from a counter-based hash of (seed, cell, gene, k), so the cell sweep
// and the gene sweep regenerate identical responsibilities.  One lane group per chunk,
// same plan as the sweeps.  Not seed-compatible with NumPy (documented).
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double exp1_draw(uint64_t seed, uint64_t cell, uint64_t gene, unsigned k)
{
    uint64_t h = mix64(seed ^ mix64(cell * 0x100000001B3ull + gene) ^ ((uint64_t)k << 48));
    double u = ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);  // (0,1)
    return -log(u);
}
template <typename T, int KL, int LPC>
__global__ __launch_bounds__(256) void random_phi_sweep_kernel(SweepArgs<T> a, uint64_t seed,
                                                               int major_is_cell)
{
    constexpr int CPW = 64 / LPC;
    constexpr int KP = KL * LPC;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int slice = a.wave_slice[wave];
    if (slice < 0) return;
    const int lane = threadIdx.x & 63;
    const int slot = lane / LPC, sub = lane % LPC;
    const int major = a.chunk_major[(size_t)slice * CPW + slot];
    if (major < 0) return;  // whole lane group leaves together
    double acc[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) acc[k] = 0.0;
    const uint4 *__restrict__ ep = a.entries + a.slice_off[slice] + slot;
    const int steps = a.slice_steps[slice];
    for (int p = 0; p < steps; ++p) {
        const uint4 e = ep[(size_t)p * CPW];
#pragma unroll 1
        for (int u = 0; u < 2; ++u) {
            const unsigned minor = u ? e.z : e.x;
            const double x = (double)__uint_as_float(u ? e.w : e.y);
            if (!(x > 0.0)) continue;
            const uint64_t cell = major_is_cell ? (uint64_t)major : (uint64_t)minor;
            const uint64_t gene = major_is_cell ? (uint64_t)minor : (uint64_t)major;
            double d[KL];
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < KL; ++k) {
                const int kk = sub * KL + k;
                d[k] = kk < a.K ? exp1_draw(seed, cell, gene, (unsigned)kk) : 0.0;
                s += d[k];
            }
            s = group_sum<double, LPC>(s);
#pragma unroll
            for (int k = 0; k < KL; ++k) acc[k] += x * d[k] / s;
        }
    }
    const int nat = a.chunk_natid[(size_t)slice * CPW + slot];
    T out[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) out[k] = (T)acc[k];
    store_row<KL>(a.partials + (size_t)nat * KP + sub * KL, out);
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
    if (i < n) out[i] = (T)(dev_digamma((double)shape[i]) - log((double)rate[i]));
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
    if (i < n) out[i] = dev_digamma(x[i]);
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

template <typename T, int KL, int LPC>
static hipError_t launch_sweep_t(const SweepArgs<T> &a, int mode, int64_t n_waves, hipStream_t st)
{
    if (sizeof(T) == 4 && (KL % 4) != 0) return hipErrorInvalidValue;  // float rows are float4-granular
    if (n_waves == 0) return hipSuccess;
    dim3 grid((unsigned)(n_waves / 4)), block(256);
    if (mode == MODE_PHI)
        hipLaunchKernelGGL((sweep_kernel<T, KL, LPC, MODE_PHI>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((sweep_kernel<T, KL, LPC, MODE_LLH>), grid, block, 0, st, a);
    return hipGetLastError();
}
template <typename T, int KL, int LPC>
static hipError_t launch_random_t(const SweepArgs<T> &a, uint64_t seed, int major_is_cell, int64_t n_waves,
                                  hipStream_t st)
{
    if (sizeof(T) == 4 && (KL % 4) != 0) return hipErrorInvalidValue;
    if (n_waves == 0) return hipSuccess;
    hipLaunchKernelGGL((random_phi_sweep_kernel<T, KL, LPC>), dim3((unsigned)(n_waves / 4)), dim3(256), 0, st,
                       a, seed, major_is_cell);
    return hipGetLastError();
}

#define SCHPF_FOR_LPC(T, KL, LPC_VAR, CALL)                                   \
    switch (LPC_VAR) {                                                        \
    case 1: { constexpr int LPC = 1; return CALL; }                           \
    case 2: { constexpr int LPC = 2; return CALL; }                           \
    case 4: { constexpr int LPC = 4; return CALL; }                           \
    case 8: { constexpr int LPC = 8; return CALL; }                           \
    default: return hipErrorInvalidValue;                                     \
    }
#define SCHPF_DISPATCH(T, kl, lpc, CALLEXPR)                                                   \
    switch (kl) {                                                                              \
    case 2: { constexpr int KL = 2; SCHPF_FOR_LPC(T, KL, lpc, CALLEXPR) }                      \
    case 6: { constexpr int KL = 6; SCHPF_FOR_LPC(T, KL, lpc, CALLEXPR) }                      \
    case 10: { constexpr int KL = 10; SCHPF_FOR_LPC(T, KL, lpc, CALLEXPR) }                    \
    case 4: { constexpr int KL = 4; SCHPF_FOR_LPC(T, KL, lpc, CALLEXPR) }                      \
    case 8: { constexpr int KL = 8; SCHPF_FOR_LPC(T, KL, lpc, CALLEXPR) }                      \
    case 12: { constexpr int KL = 12; SCHPF_FOR_LPC(T, KL, lpc, CALLEXPR) }                    \
    case 16: { constexpr int KL = 16; SCHPF_FOR_LPC(T, KL, lpc, CALLEXPR) }                    \
    case 20: { constexpr int KL = 20; SCHPF_FOR_LPC(T, KL, lpc, CALLEXPR) }                    \
    case 24: { constexpr int KL = 24; SCHPF_FOR_LPC(T, KL, lpc, CALLEXPR) }                    \
    case 32: { constexpr int KL = 32; SCHPF_FOR_LPC(T, KL, lpc, CALLEXPR) }                    \
    default: return hipErrorInvalidValue;                                                      \
    }

template <typename T>
hipError_t launch_sweep(const SweepArgs<T> &a, int kl, int lpc, int mode, int64_t n_waves, hipStream_t st)
{
    SCHPF_DISPATCH(T, kl, lpc, (launch_sweep_t<T, KL, LPC>(a, mode, n_waves, st)))
}
template <typename T>
hipError_t launch_random_phi(const SweepArgs<T> &a, int kl, int lpc, uint64_t seed, int major_is_cell,
                             int64_t n_waves, hipStream_t st)
{
    SCHPF_DISPATCH(T, kl, lpc, (launch_random_t<T, KL, LPC>(a, seed, major_is_cell, n_waves, st)))
}

template <typename T> hipError_t launch_gamma_update(const UpdateArgs<T> &a, int src, int nblocks, hipStream_t st)
{
    const size_t lds = ((size_t)2 * a.rows_per_block * a.K + a.K) * sizeof(double);
    dim3 grid((unsigned)nblocks), block(256);
    if (src == SRC_NONE) hipLaunchKernelGGL((gamma_update_kernel<T, SRC_NONE>), grid, block, lds, st, a);
    else if (src == SRC_PARTIALS) hipLaunchKernelGGL((gamma_update_kernel<T, SRC_PARTIALS>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((gamma_update_kernel<T, SRC_DENSE>), grid, block, lds, st, a);
    return hipGetLastError();
}

hipError_t launch_colsum_reduce(const double *part, int nblocks, int K, double *out, void *mirror,
                                int mirror_is_f32, hipStream_t st)
{
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3(1), dim3(256), 0, st, part, nblocks, K, out, mirror,
                       mirror_is_f32);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_combine_partials(const T *partials, const int *cptr, int n, int K, int KP, T *extra,
                                   const int *extra_flag, T *out, hipStream_t st)
{
    hipLaunchKernelGGL((combine_partials_kernel<T>), dim3(blocks_for((int64_t)n * K)), dim3(256), 0, st,
                       partials, cptr, n, K, KP, extra, extra_flag, out);
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
    template hipError_t launch_sweep<T>(const SweepArgs<T> &, int, int, int, int64_t, hipStream_t);            \
    template hipError_t launch_random_phi<T>(const SweepArgs<T> &, int, int, uint64_t, int, int64_t,           \
                                             hipStream_t);                                                     \
    template hipError_t launch_gamma_update<T>(const UpdateArgs<T> &, int, int, hipStream_t);                  \
    template hipError_t launch_combine_partials<T>(const T *, const int *, int, int, int, T *, const int *,    \
                                                   T *, hipStream_t);                                          \
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
