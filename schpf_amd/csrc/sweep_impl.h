// The sweep kernels (included by sweep_f32.hip / sweep_f64.hip, one model dtype each).
//
// Algebra (DESIGN.md "restatement of the responsibilities"):
//   phi_k = exp(Elt[i,k] + Elb[g,k]) / sum_k(...)             (reference hpf_numba.py:97-112)
//         = Et[i,k] * Eb[g,k] / s_ig,   s_ig = sum_k Et[i,k] Eb[g,k],
//   Et[i,k] = exp(Elt[i,k] - max_k Elt[i,:]),  Eb likewise per gene,
// so exp() is evaluated (N + G) * K times per iteration instead of nnz * K and X*phi
// (nnz x K, hpf_numba.py:97) is never materialised:
//   sum_g x_ig phi_igk = Et[i,k] * sum_g (x_ig / s_ig) Eb[g,k].
//
// Mapping.  One wavefront streams one slice of the plan (plan.h).  A group of LPC adjacent
// lanes owns one chunk (<= chunk_len nonzeros of one major row).  A table row is KP = NV *
// LPC * VEC values (VEC = values per 16 bytes); lane `sub` of the group holds the 16-byte
// vectors q * LPC + sub, q = 0..NV-1 -- so ONE load instruction makes the LPC lanes of a
// group read LPC * 16 contiguous bytes: with LPC = 4 that is exactly one 64-byte L1/L2
// sector per group per instruction, which is what the texture-addresser (the measured
// limiter of this gather, profiles/r01) charges for.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace schpf {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// streamed once per sweep: non-temporal so it does not evict the gathered table from L2
__device__ __forceinline__ uint4 stream_load(const uint4 *p)
{
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <typename T> struct Vec16;
template <> struct Vec16<float> {
    typedef float4 type;
    static constexpr int N = 4;
    static __device__ __forceinline__ void unpack(const float4 &t, float *v) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ float4 pack(const float *v) { return make_float4(v[0], v[1], v[2], v[3]); }
    static __device__ __forceinline__ float tiny() { return 1e-30f; }
};
template <> struct Vec16<double> {
    typedef double2 type;
    static constexpr int N = 2;
    static __device__ __forceinline__ void unpack(const double2 &t, double *v) { v[0] = t.x; v[1] = t.y; }
    static __device__ __forceinline__ double2 pack(const double *v) { return make_double2(v[0], v[1]); }
    static __device__ __forceinline__ double tiny() { return 1e-280; }
};

// this lane's NV vectors of a row: vector q lives at 16-byte slot q * LPC + sub
template <typename T, int NV, int LPC>
__device__ __forceinline__ void load_lane(const T *__restrict__ row, int sub, T (&v)[NV * Vec16<T>::N])
{
    typedef typename Vec16<T>::type V;
    const V *__restrict__ p = reinterpret_cast<const V *>(row) + sub;
#pragma unroll
    for (int q = 0; q < NV; ++q) Vec16<T>::unpack(p[q * LPC], &v[q * Vec16<T>::N]);
}
template <typename T, int NV, int LPC>
__device__ __forceinline__ void store_lane(T *__restrict__ row, int sub, const T (&v)[NV * Vec16<T>::N])
{
    typedef typename Vec16<T>::type V;
    V *__restrict__ p = reinterpret_cast<V *>(row) + sub;
#pragma unroll
    for (int q = 0; q < NV; ++q) p[q * LPC] = Vec16<T>::pack(&v[q * Vec16<T>::N]);
}
// factor index of register slot r of lane `sub`
template <typename T, int LPC> __device__ __forceinline__ int factor_of(int r, int sub)
{
    return ((r / Vec16<T>::N) * LPC + sub) * Vec16<T>::N + (r % Vec16<T>::N);
}

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
        const T o = __shfl_xor(v, m, 64);
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

// s = sum_k a_k b_k over the whole group (every lane of the group gets the same value)
template <typename T, int KL, int LPC>
__device__ __forceinline__ T group_dot(const T (&x)[KL], const T (&y)[KL])
{
    T s = T(0);
#pragma unroll
    for (int k = 0; k < KL; ++k) s += x[k] * y[k];
    return group_sum<T, LPC>(s);
}

// MODE_PHI : acc_k += (x / s) * Eb[minor,k];  partial row = acc_k * Et[major,k]  -- this
//            chunk's share of sum x*phi_k (hpf_numba.py:97-112 fused with :152-155).
//            Nonzeros whose s underflows are skipped here and redone after the loop in the
//            reference's own max-shifted log-domain form from the E[log] tables (cold path,
//            atomics into `extra`, flagged for the update kernel).
// MODE_LLH : sum over the chunk of x*log(r) - r, r = sum_k E[theta]E[beta]
//            (hpf_numba.py:43-50 minus the constant gammaln term); one double per wave.
template <typename T, int NV, int LPC, int MODE>
__global__ __launch_bounds__(256) void sweep_kernel(SweepArgs<T> a)
{
    constexpr int VEC = Vec16<T>::N;
    constexpr int KL = NV * VEC;
    constexpr int CPW = 64 / LPC;
    constexpr int KP = KL * LPC;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int slice = a.wave_slice[wave];
    const int lane = threadIdx.x & 63;
    if (slice < 0) {
        if (MODE == MODE_LLH && lane == 0) a.wave_out[wave] = 0.0;
        return;
    }
    const int slot = lane / LPC;
    const int sub = lane % LPC;
    const int major = a.chunk_major[(size_t)slice * CPW + slot];
    const bool live = major >= 0;

    T tm[KL], acc[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) { tm[k] = T(0); acc[k] = T(0); }
    if (live) load_lane<T, NV, LPC>(a.tab_major + (size_t)major * KP, sub, tm);
    double llh = 0.0;
    bool any_bad = false;

    const uint4 *__restrict__ ep = a.entries + a.slice_off[slice] + slot;
    const int steps = a.slice_steps[slice];
    const T *__restrict__ tabm = a.tab_minor;
    const T tiny = Vec16<T>::tiny();

    uint4 e = steps > 0 ? stream_load(ep) : make_uint4(0, 0, 0, 0);
    for (int p = 0; p < steps; ++p) {
        // two nonzeros per step; the next step's entries are requested before this step's math
        T b0[KL], b1[KL];
        load_lane<T, NV, LPC>(tabm + (size_t)e.x * KP, sub, b0);
        load_lane<T, NV, LPC>(tabm + (size_t)e.z * KP, sub, b1);
        const T x0 = (T)__uint_as_float(e.y);
        const T x1 = (T)__uint_as_float(e.w);
        if (p + 1 < steps) e = stream_load(ep + (size_t)(p + 1) * CPW);
        const T s0 = group_dot<T, KL, LPC>(tm, b0);
        const T s1 = group_dot<T, KL, LPC>(tm, b1);
        if (MODE == MODE_PHI) {
            const bool ok0 = s0 >= tiny, ok1 = s1 >= tiny;   // false for NaN too
            const T w0 = (x0 > T(0) && ok0) ? x0 / s0 : T(0);
            const T w1 = (x1 > T(0) && ok1) ? x1 / s1 : T(0);
            any_bad |= (x0 > T(0) && !ok0) || (x1 > T(0) && !ok1);
#pragma unroll
            for (int k = 0; k < KL; ++k) acc[k] += w0 * b0[k] + w1 * b1[k];
        } else {
            if (x0 > T(0)) llh += (double)x0 * log((double)s0) - (double)s0;
            if (x1 > T(0)) llh += (double)x1 * log((double)s1) - (double)s1;
        }
    }

    if (MODE == MODE_LLH) {
        if (sub != 0) llh = 0.0;
        llh = wave_sum(llh);
        if (lane == 0) a.wave_out[wave] = llh;
        return;
    }

    if (live) {
#pragma unroll
        for (int k = 0; k < KL; ++k) acc[k] *= tm[k];
        const int nat = a.chunk_natid[(size_t)slice * CPW + slot];
        store_lane<T, NV, LPC>(a.partials + (size_t)nat * KP, sub, acc);
    }

    // ---- cold path: nonzeros whose product-form normaliser underflowed -------------------
    if (__builtin_expect(__any(any_bad), 0)) {
        if (!any_bad) return;      // whole groups leave together (any_bad is group-uniform)
        T lt[KL];
        load_lane<T, NV, LPC>(a.log_major + (size_t)major * KP, sub, lt);
#pragma unroll 1
        for (int p = 0; p < steps; ++p) {
            const uint4 ee = ep[(size_t)p * CPW];
#pragma unroll 1
            for (int u = 0; u < 2; ++u) {
                const unsigned idx = u ? ee.z : ee.x;
                const T x = (T)__uint_as_float(u ? ee.w : ee.y);
                T b[KL];
                load_lane<T, NV, LPC>(tabm + (size_t)idx * KP, sub, b);
                const T s = group_dot<T, KL, LPC>(tm, b);     // same arithmetic as the hot loop
                if (!(x > T(0)) || s >= tiny) continue;
                // the reference's form: softmax_k of Elt + Elb with a max shift (hpf_numba.py:98-112)
                T lr[KL];
                load_lane<T, NV, LPC>(a.log_minor + (size_t)idx * KP, sub, lr);
                T mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < KL; ++k) {
                    lr[k] += lt[k];
                    if (factor_of<T, LPC>(k, sub) < a.K) mx = lr[k] > mx ? lr[k] : mx;
                }
                mx = group_max<T, LPC>(mx);
                T ss = T(0);
#pragma unroll
                for (int k = 0; k < KL; ++k) {
                    lr[k] = factor_of<T, LPC>(k, sub) < a.K ? (T)exp((double)(lr[k] - mx)) : T(0);
                    ss += lr[k];
                }
                ss = group_sum<T, LPC>(ss);
#pragma unroll
                for (int k = 0; k < KL; ++k) {
                    const int f = factor_of<T, LPC>(k, sub);
                    if (f < a.K) atomicAdd(a.extra + (size_t)major * KP + f, (T)((double)x * (double)lr[k] / (double)ss));
                }
            }
        }
        *a.extra_flag = 1;
    }
}

// ------------------------------------------------------ t = 0 random responsibilities
// Device-side variant of scHPF_.py:652-655 for matrices too large for a host draw:
// phi_k = e_k / sum e, e_k ~ Exp(1) from a counter-based hash of (seed, cell, gene, k), so
// the cell sweep and the gene sweep regenerate identical responsibilities.  Same plan and
// lane mapping as the sweeps.  Not seed-compatible with NumPy (documented).
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double exp1_draw(uint64_t seed, uint64_t cell, uint64_t gene, unsigned k)
{
    const uint64_t h = mix64(seed ^ mix64(cell * 0x100000001B3ull + gene) ^ ((uint64_t)k << 48));
    const double u = ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);  // (0,1)
    return -log(u);
}
template <typename T, int NV, int LPC>
__global__ __launch_bounds__(256) void random_phi_sweep_kernel(SweepArgs<T> a, uint64_t seed, int major_is_cell)
{
    constexpr int VEC = Vec16<T>::N;
    constexpr int KL = NV * VEC;
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
                const int f = factor_of<T, LPC>(k, sub);
                d[k] = f < a.K ? exp1_draw(seed, cell, gene, (unsigned)f) : 0.0;
                s += d[k];
            }
            s = group_sum<double, LPC>(s);
            const double w = x / s;
#pragma unroll
            for (int k = 0; k < KL; ++k) acc[k] += w * d[k];
        }
    }
    const int nat = a.chunk_natid[(size_t)slice * CPW + slot];
    T out[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) out[k] = (T)acc[k];
    store_lane<T, NV, LPC>(a.partials + (size_t)nat * KP, sub, out);
}

// ------------------------------------------------------------------------ launchers
template <typename T, int NV, int LPC>
static hipError_t launch_sweep_t(const SweepArgs<T> &a, int mode, int64_t n_waves, hipStream_t st)
{
    if (n_waves == 0) return hipSuccess;
    dim3 grid((unsigned)(n_waves / 4)), block(256);
    if (mode == MODE_PHI)
        hipLaunchKernelGGL((sweep_kernel<T, NV, LPC, MODE_PHI>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((sweep_kernel<T, NV, LPC, MODE_LLH>), grid, block, 0, st, a);
    return hipGetLastError();
}
template <typename T, int NV, int LPC>
static hipError_t launch_random_t(const SweepArgs<T> &a, uint64_t seed, int major_is_cell, int64_t n_waves,
                                  hipStream_t st)
{
    if (n_waves == 0) return hipSuccess;
    hipLaunchKernelGGL((random_phi_sweep_kernel<T, NV, LPC>), dim3((unsigned)(n_waves / 4)), dim3(256), 0, st,
                       a, seed, major_is_cell);
    return hipGetLastError();
}

#define SCHPF_FOR_LPC(LPC_VAR, CALL)                                          \
    switch (LPC_VAR) {                                                        \
    case 1: { constexpr int LPC = 1; return CALL; }                           \
    case 2: { constexpr int LPC = 2; return CALL; }                           \
    case 4: { constexpr int LPC = 4; return CALL; }                           \
    case 8: { constexpr int LPC = 8; return CALL; }                           \
    case 16: { constexpr int LPC = 16; return CALL; }                         \
    default: return hipErrorInvalidValue;                                     \
    }
#define SCHPF_DISPATCH(nv, lpc, CALLEXPR)                                     \
    switch (nv) {                                                             \
    case 1: { constexpr int NV = 1; SCHPF_FOR_LPC(lpc, CALLEXPR) }            \
    case 2: { constexpr int NV = 2; SCHPF_FOR_LPC(lpc, CALLEXPR) }            \
    case 3: { constexpr int NV = 3; SCHPF_FOR_LPC(lpc, CALLEXPR) }            \
    case 4: { constexpr int NV = 4; SCHPF_FOR_LPC(lpc, CALLEXPR) }            \
    case 5: { constexpr int NV = 5; SCHPF_FOR_LPC(lpc, CALLEXPR) }            \
    case 6: { constexpr int NV = 6; SCHPF_FOR_LPC(lpc, CALLEXPR) }            \
    case 8: { constexpr int NV = 8; SCHPF_FOR_LPC(lpc, CALLEXPR) }            \
    default: return hipErrorInvalidValue;                                     \
    }

template <typename T>
hipError_t launch_sweep(const SweepArgs<T> &a, int nv, int lpc, int mode, int64_t n_waves, hipStream_t st)
{
    SCHPF_DISPATCH(nv, lpc, (launch_sweep_t<T, NV, LPC>(a, mode, n_waves, st)))
}
template <typename T>
hipError_t launch_random_phi(const SweepArgs<T> &a, int nv, int lpc, uint64_t seed, int major_is_cell,
                             int64_t n_waves, hipStream_t st)
{
    SCHPF_DISPATCH(nv, lpc, (launch_random_t<T, NV, LPC>(a, seed, major_is_cell, n_waves, st)))
}

}  // namespace schpf
