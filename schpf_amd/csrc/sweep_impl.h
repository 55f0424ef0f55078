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
#include <atomic>
#include <stdint.h>
#include <type_traits>

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

// Cross-lane exchange inside a lane group with DPP (no LDS traffic): after each step every lane
// of the 2^step-lane sub-group holds the same value, so quad_perm (xor 1, xor 2), then
// row_half_mirror (i <-> 7-i) and row_mirror (i <-> 15-i) complete an all-reduce over 16 lanes.
template <int STEP> __device__ __forceinline__ int dpp_partner(int v)
{
    // every lane has a valid source under these patterns; mov_dpp (no `old` operand) spares the
    // copies / zero-initialisation that update_dpp needs for its tied destination
    if (STEP == 0) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    if (STEP == 1) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    if (STEP == 2) return __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    return __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true);                 // row_mirror
}
template <int STEP> __device__ __forceinline__ float partner(float v)
{
    return __int_as_float(dpp_partner<STEP>(__float_as_int(v)));
}
template <int STEP> __device__ __forceinline__ double partner(double v)
{
    const int lo = dpp_partner<STEP>(__double2loint(v));
    const int hi = dpp_partner<STEP>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <typename T, int LPC> __device__ __forceinline__ T group_sum(T v)
{
    if (LPC >= 2) v += partner<0>(v);
    if (LPC >= 4) v += partner<1>(v);
    if (LPC >= 8) v += partner<2>(v);
    if (LPC >= 16) v += partner<3>(v);
    return v;
}
template <typename T, int LPC> __device__ __forceinline__ T group_max(T v)
{
    if (LPC >= 2) { const T o = partner<0>(v); v = o > v ? o : v; }
    if (LPC >= 4) { const T o = partner<1>(v); v = o > v ? o : v; }
    if (LPC >= 8) { const T o = partner<2>(v); v = o > v ? o : v; }
    if (LPC >= 16) { const T o = partner<3>(v); v = o > v ? o : v; }
    return v;
}

// x / s for s known to be a normal positive number: hardware reciprocal (+ one Newton step in
// f64) instead of the IEEE division sequence -- no scaling / fix-up is needed here.
//   f64: v_rcp_f64 is good to ~2^-24, the Newton step squares that: relative error <= ~2^-48
//        (4e-15) per weight, three orders of magnitude below the tightest parity tolerance
//        (1e-12) and of random sign, so it averages out in the sums over a row's nonzeros;
//   f32: v_rcp_f32 is good to 1 ulp: x * rcp(s) is within 2 ulp.
// fast_div_exact keeps the residual correction (<= 1 ulp) for the paths that are not VALU-bound.
__device__ __forceinline__ double fast_div(double x, double s)
{
    double r = __builtin_amdgcn_rcp(s);
    r = fma(fma(-s, r, 1.0), r, r);
    return x * r;
}
__device__ __forceinline__ float fast_div(float x, float s) { return x * __builtin_amdgcn_rcpf(s); }
__device__ __forceinline__ double fast_div_exact(double x, double s)
{
    double r = __builtin_amdgcn_rcp(s);
    r = fma(fma(-s, r, 1.0), r, r);
    const double q = x * r;
    return fma(fma(-s, q, x), r, q);
}
__device__ __forceinline__ float fast_div_exact(float x, float s)
{
    const float r = __builtin_amdgcn_rcpf(s);
    const float q = x * r;
    return fmaf(fmaf(-s, q, x), r, q);
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__device__ __forceinline__ double fma_t(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return fmaf(a, b, c); }

// x / s without a branch: padding entries carry x = 0 (weight 0); with an underflowed normaliser
// (ok == false) the quotient is inf/NaN garbage and is masked -- the cold path redoes that group.
template <typename T> __device__ __forceinline__ T safe_weight(T x, T s, bool ok)
{
    const T q = fast_div_exact(x, s);
    return ok ? q : T(0);
}

// s = sum_k a_k b_k over the whole group (every lane of the group gets the same value)
template <typename T, int KL, int LPC>
__device__ __forceinline__ T group_dot(const T (&x)[KL], const T (&y)[KL])
{
    // independent partial sums: a single chain of KL dependent FMAs leaves the SIMD idle whenever
    // fewer than ~4 waves have VALU work ready.  f64: two chains.  f32: four, because the compiler
    // packs pairs of chains into v_pk_fma_f32 and two chains would again be ONE dependent chain.
    T t;
    if constexpr (sizeof(T) == 4 && KL % 4 == 0) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KL; k += 4) {
            const f32x2 x0 = {x[k], x[k + 1]}, y0 = {y[k], y[k + 1]};
            const f32x2 x1 = {x[k + 2], x[k + 3]}, y1 = {y[k + 2], y[k + 3]};
            a0 = __builtin_elementwise_fma(x0, y0, a0);
            a1 = __builtin_elementwise_fma(x1, y1, a1);
        }
        const f32x2 h = a0 + a1;     // one v_pk_add_f32, then one add (pairwise sums first cost three moves more)
        t = h.x + h.y;
    } else {
        T s0 = T(0), s1 = T(0);
#pragma unroll
        for (int k = 0; k + 1 < KL; k += 2) {
            s0 = fma_t(x[k], y[k], s0);
            s1 = fma_t(x[k + 1], y[k + 1], s1);
        }
        if (KL & 1) s0 = fma_t(x[KL - 1], y[KL - 1], s0);
        t = s0 + s1;
    }
    return group_sum<T, LPC>(t);
}


// Cold path shared by both sweeps.  When the product-form normaliser of ANY nonzero of a lane
// group underflowed, the group's whole accumulator is recomputed in the reference's own
// max-shifted log-domain form (hpf_numba.py:98-112) from the E[log] tables and replaces the
// fast result -- no atomics, no double counting, still deterministic.  Deliberately register-
// lean (tables are re-read from L2 three times per nonzero): it must not cost the hot loop
// occupancy.  acc[] receives sum x * phi_k directly (no multiplication by Et afterwards).
template <typename T, int NV, int LPC>
__device__ __forceinline__ void slow_nonzero(const T *__restrict__ lt_row, const T *__restrict__ lm_row, int sub,
                                          int K, T x, T (&acc)[NV * Vec16<T>::N])
{
    typedef typename Vec16<T>::type V;
    constexpr int VEC = Vec16<T>::N;
    const V *__restrict__ pt = reinterpret_cast<const V *>(lt_row) + sub;
    const V *__restrict__ pm = reinterpret_cast<const V *>(lm_row) + sub;
    T mx = -INFINITY;
#pragma unroll 1
    for (int q = 0; q < NV; ++q) {
        T a[VEC], b[VEC];
        Vec16<T>::unpack(pt[q * LPC], a);
        Vec16<T>::unpack(pm[q * LPC], b);
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            if ((q * LPC + sub) * VEC + v < K) { const T l = a[v] + b[v]; mx = l > mx ? l : mx; }
    }
    mx = group_max<T, LPC>(mx);
    double ss = 0.0;
#pragma unroll 1
    for (int q = 0; q < NV; ++q) {
        T a[VEC], b[VEC];
        Vec16<T>::unpack(pt[q * LPC], a);
        Vec16<T>::unpack(pm[q * LPC], b);
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            if ((q * LPC + sub) * VEC + v < K) ss += exp((double)(a[v] + b[v] - mx));
    }
    ss = group_sum<double, LPC>(ss);
    const double scale = (double)x / ss;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        T a[VEC], b[VEC];
        Vec16<T>::unpack(pt[q * LPC], a);
        Vec16<T>::unpack(pm[q * LPC], b);
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            if ((q * LPC + sub) * VEC + v < K) acc[q * VEC + v] += (T)(scale * exp((double)(a[v] + b[v] - mx)));
    }
}

// whole-group recompute for the gather plan: one chunk, `steps` uint4 steps `stride` apart
template <typename T, int NV, int LPC>
__device__ __noinline__ void slow_chunk(const uint4 *__restrict__ ep, int steps, int stride,
                                        const T *__restrict__ lt_row, const T *__restrict__ log_minor, int sub,
                                        int K, T *__restrict__ out_row)
{
    constexpr int KL = NV * Vec16<T>::N;
    constexpr int KP = KL * LPC;
    T acc[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) acc[k] = T(0);
#pragma unroll 1
    for (int p = 0; p < steps; ++p) {
        const uint4 ee = ep[(size_t)p * stride];
        if (__uint_as_float(ee.y) > 0.f)
            slow_nonzero<T, NV, LPC>(lt_row, log_minor + (size_t)ee.x * KP, sub, K, (T)__uint_as_float(ee.y), acc);
        if (__uint_as_float(ee.w) > 0.f)
            slow_nonzero<T, NV, LPC>(lt_row, log_minor + (size_t)ee.z * KP, sub, K, (T)__uint_as_float(ee.w), acc);
    }
    store_lane<T, NV, LPC>(out_row, sub, acc);
}
// ... and for the tile plan: a row's nonzeros over the windows [w0, w1) of its task
template <bool PACK> struct TileEntry;
// Minor row of an entry: its index field is the LDS position of the row in 16-byte units
// (plan.h); w = window (window mode) or epoch (ring mode: the row's sub-window is the one of
// [w, w + ring - 2] that lives in the entry's slot).
__device__ __forceinline__ int entry_minor(unsigned off16, int w, int win_rows, int row_slots, int ring, int slot16)
{
    if (ring <= 1) return w * win_rows + (int)(off16 / (unsigned)row_slots);
    const int slot = (int)(off16 / (unsigned)slot16);
    const int r = (int)((off16 - (unsigned)slot * (unsigned)slot16) / (unsigned)row_slots);
    const int ahead = (slot - w % ring + ring) % ring;
    return (w + ahead) * win_rows + r;
}
template <typename T, int NV, int LPC, bool PACK>
__device__ __noinline__ void slow_task_row(const void *__restrict__ entries, size_t pos, const uint16_t *__restrict__ st,
                                           int w0, int w1, int win_rows, int ring, int slot16, int single, int stride,
                                           const T *__restrict__ lt_row, const T *__restrict__ log_minor, int sub,
                                           int K, T *__restrict__ out_row, const int *__restrict__ minor_of_block)
{
    typedef TileEntry<PACK> EF;
    constexpr int KL = NV * Vec16<T>::N;
    constexpr int KP = KL * LPC;
    constexpr int ROW_SLOTS = KP * (int)sizeof(T) / 16;
    T acc[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) acc[k] = T(0);
#pragma unroll 1
    for (int w = w0; w < w1; ++w) {
        const int steps = single ? ((int)st[w] + 1) >> 1 : (int)st[w];   // stored slots; an unused half slot has count 0
#pragma unroll 1
        for (int p = 0; p < steps; ++p) {
            const typename EF::type ee = EF::load(entries, pos + (size_t)p * stride);
#pragma unroll 1
            for (int u = 0; u < 2; ++u) {
                const float x = EF::val(ee, u);
                if (x > 0.f) {
                    int minor = entry_minor(EF::idx(ee, u), w, win_rows, ROW_SLOTS, ring, slot16);
                    if (minor_of_block) minor = minor_of_block[minor];   // balanced windows: virtual -> table row
                    slow_nonzero<T, NV, LPC>(lt_row, log_minor + (size_t)minor * KP, sub, K, (T)x, acc);
                }
            }
        }
        pos += (size_t)steps * stride;
    }
    store_lane<T, NV, LPC>(out_row, sub, acc);
}

// MODE_PHI : acc_k += (x / s) * Eb[minor,k];  partial row = acc_k * Et[major,k]  -- this
//            chunk's share of sum x*phi_k (hpf_numba.py:97-112 fused with :152-155).
//            If the normaliser s of any nonzero of the group underflows, the group's result is
//            recomputed in the reference's own max-shifted log-domain form (slow_nonzero).
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
    const int steps = __builtin_amdgcn_readfirstlane(a.slice_steps[slice]);
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
            const T w0 = safe_weight(x0, s0, ok0), w1 = safe_weight(x1, s1, ok1);
            any_bad |= (x0 > T(0) && !ok0) || (x1 > T(0) && !ok1);
#pragma unroll
            for (int k = 0; k < KL; ++k) acc[k] = fma_t(w1, b1[k], fma_t(w0, b0[k], acc[k]));
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
        const int nat = a.chunk_natid[(size_t)slice * CPW + slot];
        T *out_row = a.partials + (size_t)nat * KP;
        if (__builtin_expect(any_bad, 0)) {   // group-uniform; rare: see slow_nonzero
            slow_chunk<T, NV, LPC>(ep, steps, CPW, a.log_major + (size_t)major * KP, a.log_minor, sub, a.K, out_row);
        } else {
#pragma unroll
            for (int k = 0; k < KL; ++k) acc[k] *= tm[k];
            store_lane<T, NV, LPC>(out_row, sub, acc);
        }
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
// one 64-bit mix per nonzero (base), then a 32-bit finaliser per factor and a hardware log: the
// start only has to be a valid random point of the simplex, not a high-grade variate
__device__ __forceinline__ uint64_t draw_base(uint64_t seed, uint64_t cell, uint64_t gene)
{
    return mix64(seed ^ mix64(cell * 0x100000001B3ull + gene));
}
__device__ __forceinline__ double exp1_draw(uint64_t base, unsigned k)
{
    uint32_t z = (uint32_t)base ^ ((uint32_t)(base >> 32) + k * 0x9E3779B9u);
    z ^= z >> 16; z *= 0x85EBCA6Bu; z ^= z >> 13; z *= 0xC2B2AE35u; z ^= z >> 16;   // murmur3 fmix32
    const float u = ((float)(z >> 8) + 0.5f) * (1.0f / 16777216.0f);                 // (0,1)
    return (double)(-__logf(u));
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
            const uint64_t base = draw_base(seed, cell, gene);
            double d[KL];
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < KL; ++k) {
                const int f = factor_of<T, LPC>(k, sub);
                d[k] = f < a.K ? exp1_draw(base, (unsigned)f) : 0.0;
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


// ------------------------------------------------------------- the LDS-staged sweep
// Entry formats of the tile plan (plan.h): two nonzeros per step slot.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <bool PACK> struct TileEntry;
template <> struct TileEntry<true> {      // {idx0 | idx1 << 16, cnt0 | cnt1 << 16}
    typedef uint2 type;
    static __device__ __forceinline__ uint2 load(const void *base, size_t i)
    {
        const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(base) + i);
        return make_uint2(v.x, v.y);
    }
    static __device__ __forceinline__ unsigned idx(const uint2 &e, int u) { return u ? e.x >> 16 : e.x & 0xFFFFu; }
    static __device__ __forceinline__ float val(const uint2 &e, int u) { return (float)(u ? e.y >> 16 : e.y & 0xFFFFu); }
};
template <> struct TileEntry<false> {     // {idx0, val0, idx1, val1}
    typedef uint4 type;
    static __device__ __forceinline__ uint4 load(const void *base, size_t i)
    {
        return stream_load(reinterpret_cast<const uint4 *>(base) + i);
    }
    static __device__ __forceinline__ unsigned idx(const uint4 &e, int u) { return u ? e.z : e.x; }
    static __device__ __forceinline__ float val(const uint4 &e, int u) { return __uint_as_float(u ? e.w : e.y); }
};

// One workgroup = one task of a tile plan (plan.h): a block of major rows (one per lane
// group) x a range of minor windows.  Per window: the window's slice of the minor table is
// copied HBM/L2 -> LDS once (coalesced 16-byte lanes), then every group streams its row's
// nonzeros of that window (sliced-ELL, coalesced, non-temporal) and gathers the minor
// K-vectors from LDS with ds_read_b128 -- no per-nonzero L2->L1 line fills.
// The entry stream runs through a 4-slot register ring: slot i is refilled with step p+4 as
// soon as step p has been taken out of it, and the ring for the NEXT window is primed before
// that window's staging barrier, so HBM latency hides behind four steps of math or a staging.
// sum over nonzeros of x * log(r) - r with a TABLE-DRIVEN logarithm: r = m 2^e (v_frexp), the top six mantissa bits
// pick c_i = (64.5 + i) / 128 from a 64-entry table {log c_i, 1 / c_i} in LDS, t = m / c_i - 1 is within 2^-7 and
// log(1 + t) is seven terms of its series (the eighth is below 2e-18): ~20 VALU instructions and one ds_read_b128 per
// nonzero, good to ~2e-16 absolute, for any count x.  History: a libm log per nonzero was ~80 instructions (round 1);
// log(prod r^x) with the product kept as mantissa x 2^exponent ~45 (round 2, counts up to 15 only).  The loss pass
// went 0.46 -> see profiles/r04 at C3.
struct LlhAccumulator {
    double sum = 0.0, rsum = 0.0;
    static constexpr int TABLE_BYTES = 64 * 16;
    // 64 threads of the workgroup fill the table (every task: ~100 instructions); a barrier follows before its first use
    static __device__ __forceinline__ void fill_table(double2 *tab, int tid)
    {
        if (tid < 64) {
            const double c = (64.5 + (double)tid) * (1.0 / 128.0);
            tab[tid] = make_double2(log(c), 1.0 / c);
        }
    }
    __device__ __forceinline__ void add(double x, double r, const double2 *tab)
    {
        rsum += r;
        if (!(r >= 2.3e-308 && r <= 1.7e308)) { sum += x * log(r); return; }   // zero, denormal, inf, NaN: divergent, rare
        const int e = __builtin_amdgcn_frexp_exp(r);
        const double m = __builtin_amdgcn_frexp_mant(r);                          // [0.5, 1)
        const int i = (__double2hiint(m) >> 14) & 63;                             // top six mantissa bits
        const double2 c = tab[i];
        const double t = fma(m, c.y, -1.0);
        double p = 1.0 / 7.0;
        p = fma(p, t, -1.0 / 6.0);
        p = fma(p, t, 1.0 / 5.0);
        p = fma(p, t, -1.0 / 4.0);
        p = fma(p, t, 1.0 / 3.0);
        p = fma(p, t, -1.0 / 2.0);
        p = fma(p * t, t, t);                                                     // log(1 + t)
        const double ed = (double)e;
        double lg = fma(ed, 0.693147180559945286, c.x);                           // ln 2 = hi + lo
        lg = fma(ed, 2.319046813846299558e-17, lg) + p;
        sum = fma(x, lg, sum);
    }
    __device__ __forceinline__ double total() const { return sum - rsum; }
};

// the minor row an entry points at: LDS position in 16-byte units
template <typename T> __device__ __forceinline__ const T *lds_row(const unsigned char *lds, unsigned off16)
{
    return reinterpret_cast<const T *>(lds + (off16 << 4));
}

// BAL: balanced windows (plan.h) -- an instantiation of its own (1024-thread workgroups only), so that the row-list
// staging costs the kernels of index-cut windows neither registers nor instructions
template <typename T, int NV, int LPC, int MODE, int MAXT, bool PACK, bool BAL = false>
__device__ __forceinline__ void tile_sweep_task_window(const TileArgs<T> &a, const int task)
{
    typedef TileEntry<PACK> EF;
    typedef typename EF::type E;
    constexpr int VEC = Vec16<T>::N;
    constexpr int KL = NV * VEC;
    constexpr int GPW = 64 / LPC;
    constexpr int KP = KL * LPC;
    constexpr int RING = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

    const int blk = a.task_block[task], w0 = a.task_w0[task], w1 = a.task_w1[task];
    const int stage_end = a.task_stage_end ? a.task_stage_end[task] : w1;   // readable horizon of the half-window schedule
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int grp = lane / LPC, sub = lane % LPC;
    const int gpb = GPW * a.wpb;
    const int g = wv * GPW + grp;
    const int major = a.block_rows[(size_t)blk * gpb + g];
    const bool live = major >= 0;

    T tm[KL], acc[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) { tm[k] = T(0); acc[k] = T(0); }
    if (MODE != MODE_RANDOM && live) load_lane<T, NV, LPC>(a.tab_major + (size_t)major * KP, sub, tm);
    double llh = 0.0;
    LlhAccumulator lacc;   // MODE_LLH
    // ... and its logarithm table, behind the window in LDS (a.llh_tab_off; the window loop's first barrier publishes it)
    const double2 *llh_tab = reinterpret_cast<const double2 *>(lds_raw + (MODE == MODE_LLH ? a.llh_tab_off : 0));
    if (MODE == MODE_LLH) LlhAccumulator::fill_table(reinterpret_cast<double2 *>(lds_raw + a.llh_tab_off), (int)threadIdx.x);
    bool any_bad = false;
    // narrow rows: two minor rows in registers (a 512-thread workgroup has twice the registers per lane)
    // (the loss pass keeps no accumulators, but pairing its wide rows -- K = 50: 112 bytes per lane -- spills 25-34
    // registers into the step loop: loss evaluation at the C5 share 1.30 -> 1.64 ms in f64, 0.49 -> 0.55 in f32; round 4)
    constexpr bool PAIR = KL * (int)sizeof(T) <= (MAXT <= 512 ? 192 : 96);
    constexpr bool PIPE = PAIR && MODE != MODE_RANDOM;
    // wide rows: ONE row in registers, refilled vector by vector.  float64 only: the float32 kernel with wide rows
    // (K = 50: 28 floats per lane) is 18 % slower with it (profiles/r03/ab_rolling_wide_rows.txt) and keeps the plain loop
    constexpr bool ROLL = !PAIR && MODE != MODE_RANDOM && sizeof(T) == 8;
    T bA[KL], bB[KL];                                  // PIPE: the rows of the step being / about to be computed
    float xc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};         //       counts of that step [step parity][nonzero]

    // position (in step slots) of this group's entries; advances window by window
    size_t pos = (size_t)a.task_wave_off[(size_t)task * a.wpb + wv] + grp;
    const uint16_t *__restrict__ st = a.steps + ((size_t)blk * a.wpb + wv) * a.n_windows;

    E ring[RING];
    int steps = __builtin_amdgcn_readfirstlane((int)st[w0]);
    // every ring load is unconditional (the entry buffer is padded, plan.cpp), so the number of
    // loads in flight is a compile-time fact and the waits can be s_waitcnt vmcnt(RING - 1)
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = EF::load(a.entries, pos + (size_t)i * GPW);

    // asynchronous global -> LDS copy (global_load_lds_dwordx4): a wave instruction moves 1 KiB (LDS
    // address = wave-uniform base + 16 * lane), no staging registers and every piece in flight at
    // once; the __syncthreads that follows drains them (vmcnt(0)) first
    // balanced windows (plan.h, BAL): the window is a LIST of table rows.  Wave wv copies the window's rows
    // [wv * RPW, (wv + 1) * RPW); their numbers are fetched one window AHEAD, a lane each (one or two registers that
    // live through the step loop), and handed to the copying lanes by ds_bpermute -- no load in front of the copies
    constexpr int ROW_SLOTS = KP * (int)sizeof(T) / 16;
    constexpr int RPI = 64 / ROW_SLOTS;     // whole rows per copy instruction
    const int rpw = BAL ? (a.win_rows + a.wpb - 1) / a.wpb : 0;
    const bool rows_ahead = BAL && rpw <= 128;
    int rows_lo = -1, rows_hi = -1;
    auto fetch_rows = [&](int sw) {
        if (!rows_ahead) return;
        const int r0 = sw * a.win_rows;
        const int nr = min(a.win_rows, a.n_minor - r0);
        const int *__restrict__ list = a.minor_of + (size_t)blk * a.n_virtual + r0;
        const int l = wv * rpw + lane;
        rows_lo = (lane < rpw && l < nr) ? list[l] : -1;
        rows_hi = (lane + 64 < rpw && l + 64 < nr) ? list[l + 64] : -1;
    };
    if (BAL && MODE != MODE_RANDOM) fetch_rows(w0);
    auto stage = [&](int sw, int slot) {
        const int r0 = sw * a.win_rows;
        const int nr = min(a.win_rows, a.n_minor - r0);
        if (BAL && rows_ahead) {
            // a wave instruction copies RPI whole rows (lane -> row lane / ROW_SLOTS, 16-byte piece lane % ROW_SLOTS; the LDS
            // side of the DMA is base + 16 * lane, so rows land back to back)
            const int rr = lane / ROW_SLOTS, q = lane - rr * ROW_SLOTS;
            unsigned char *dst = lds_raw + (size_t)slot * a.slot_bytes + (size_t)wv * rpw * ROW_SLOTS * 16;
            const unsigned char *__restrict__ tab = reinterpret_cast<const unsigned char *>(a.tab_minor);
            const int n_u = (rpw + RPI - 1) / RPI;
            for (int u = 0; u < n_u; ++u) {   // scalar loop
                const int src = u * RPI + rr;                      // the lane's row among the wave's
                int row = __builtin_amdgcn_ds_bpermute((src & 63) << 2, rows_lo);
                if (rpw > 64) {
                    const int hi = __builtin_amdgcn_ds_bpermute((src & 63) << 2, rows_hi);
                    row = src < 64 ? row : hi;
                }
                if (rr < RPI && src < rpw && row >= 0)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(tab + ((size_t)row * ROW_SLOTS + q) * 16),
                        (__attribute__((address_space(3))) void *)(dst + (size_t)u * RPI * ROW_SLOTS * 16), 16, 0, 0);
            }
            return;
        }
        if constexpr (BAL) {
            // (very narrow rows: more than 128 rows per wave) the row numbers of a batch of copy instructions are
            // fetched in the staging itself, then the copies go out.  A wave instruction copies RPI whole rows
            // (lane -> row lane / ROW_SLOTS, 16-byte piece lane % ROW_SLOTS; the LDS side of the DMA is base + 16 * lane,
            // so rows land back to back); the row numbers of a batch of instructions are fetched first, then the copies
            // go out.  Measured alternatives (profiles/r04/ab_balanced_windows.txt): row numbers fetched before the
            // barrier (registers the step loop does not have: 177 spilled, 3 x slower) and row lists staged through LDS
            // (the compiler drains the copies in front of every later LDS read; slower than this)
            constexpr int BATCH = 6;
            const int rr = lane / ROW_SLOTS, q = lane - rr * ROW_SLOTS;
            const int *__restrict__ list = a.minor_of + (size_t)blk * a.n_virtual + r0;
            unsigned char *dst = lds_raw + (size_t)slot * a.slot_bytes;
            const unsigned char *__restrict__ tab = reinterpret_cast<const unsigned char *>(a.tab_minor);
            const int n_inst = (nr + RPI - 1) / RPI;
            for (int i0 = wv; i0 < n_inst; i0 += BATCH * a.wpb) {   // scalar loop
                int row[BATCH];
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const int j = (i0 + u * a.wpb) * RPI + rr;
                    row[u] = (rr < RPI && j < nr) ? list[j] : -1;
                }
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const int i = i0 + u * a.wpb;
                    if (i < n_inst && row[u] >= 0)
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void *)(tab + ((size_t)row[u] * ROW_SLOTS + q) * 16),
                            (__attribute__((address_space(3))) void *)(dst + (size_t)i * RPI * ROW_SLOTS * 16), 16, 0, 0);
                }
            }
            return;
        }
        const unsigned char *__restrict__ src = reinterpret_cast<const unsigned char *>(a.tab_minor + (size_t)r0 * KP);
        unsigned char *dst = lds_raw + (size_t)slot * a.slot_bytes;
        const int nbytes = nr * KP * (int)sizeof(T);                  // a multiple of 16
        for (int off = wv * 1024; off < nbytes; off += a.wpb * 1024) {   // scalar loop
            const int o = off + lane * 16;
            if (o < nbytes)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + o),
                                                 (__attribute__((address_space(3))) void *)(dst + off), 16, 0, 0);
        }
    };
    // Window schedule: the whole LDS is window w, refilled between two barriers.  Half-window
    // schedule (a.ring slots, plan.h): the first epoch fills every slot, a later one only the slot
    // that the last epoch's own sub-window had.
    const int L = a.ring > 1 ? a.ring : 1;
    for (int w = w0; w < w1; ++w) {
        if (MODE != MODE_RANDOM) {
            __syncthreads();                       // previous window fully consumed
            const int sw0 = (L == 1 || w == w0) ? w : w + L - 1;
            const int sw1 = min(w + L, stage_end);
            for (int sw = sw0; sw < sw1; ++sw) stage(sw, L > 1 ? sw % L : 0);
            __syncthreads();
            if (BAL && w + 1 < w1) fetch_rows(w + 1);   // the next window's rows, under this window's steps
        }
        // stored step slots of this (wave, window); `single`: steps counts nonzeros, an odd count leaves the second
        // half of its last slot unexecuted in the one-nonzero-at-a-time loop (elsewhere that half has count 0)
        const int nsl = a.single ? (steps + 1) >> 1 : steps;
        if (PIPE) {
            // Rolling LDS pipeline, one nonzero deep: the minor rows of step p+1 are fetched from the
            // window while step p is still being computed -- row A' right after nonzero A has been
            // consumed, into the same registers, then the same for B.  A wave never starts a step
            // by waiting a full LDS latency (measured: the unpipelined loop overlapped the LDS read
            // phase and the FMA phase of the 4 waves of a SIMD poorly).
            if (nsl > 0) {   // prologue: the first step's rows (the ring was primed before the barrier)
                const E c = ring[0];
                unsigned i0 = EF::idx(c, 0), i1 = EF::idx(c, 1);
                xc[0][0] = EF::val(c, 0); xc[0][1] = EF::val(c, 1);
                asm volatile("" : "+v"(i0), "+v"(i1), "+v"(xc[0][0]), "+v"(xc[0][1]));
                load_lane<T, NV, LPC>(lds_row<T>(lds_raw, i0), sub, bA);
                load_lane<T, NV, LPC>(lds_row<T>(lds_raw, i1), sub, bB);
            }
            // one step of the pipeline; I = position in the entry ring (a compile-time constant: the register
            // arrays are indexed with it)
            auto pipe_step = [&](auto I_) {
                constexpr int I = decltype(I_)::value;
                    const E cn = ring[(I + 1) % RING];
                    unsigned n0 = EF::idx(cn, 0), n1 = EF::idx(cn, 1);
                    // counts alternate between two register pairs (RING is even), no copies
                    xc[(I + 1) & 1][0] = EF::val(cn, 0); xc[(I + 1) & 1][1] = EF::val(cn, 1);
                    asm volatile("" : "+v"(n0), "+v"(n1), "+v"(xc[(I + 1) & 1][0]), "+v"(xc[(I + 1) & 1][1]));
                    const T x0 = (T)xc[I & 1][0], x1 = (T)xc[I & 1][1];
                    // both normalisers first: two independent dot / reciprocal chains in flight
                    // (measured -1 % f64, -3 % f32 against finishing nonzero A before starting B:
                    // profiles/r02/ab_step_variants.log).  No test of the normaliser here: a
                    // product-form s that underflowed (zero / denormal) makes the reciprocal inf,
                    // and inf * b or 0 * inf poisons EVERY accumulator of every lane of the group
                    // (inf or NaN) -- detected once, after the task, and the group is then redone by
                    // the cold path.  A small but normal s is exact enough: its largest term is a
                    // normal number.
                    const T s0 = group_dot<T, KL, LPC>(tm, bA);
                    const T s1 = group_dot<T, KL, LPC>(tm, bB);
                    if (MODE == MODE_PHI) {
                        const T q0 = fast_div(x0, s0);
                        const T q1 = fast_div(x1, s1);
#pragma unroll
                        for (int k = 0; k < KL; ++k) acc[k] = fma_t(q0, bA[k], acc[k]);
                        load_lane<T, NV, LPC>(lds_row<T>(lds_raw, n0), sub, bA);
                        // nothing moves across: row A' must be requested BEFORE nonzero B is accumulated.  The
                        // scheduling barrier holds the machine scheduler; in the straight-line float32 turn the
                        // optimiser had already hoisted B's accumulation above the loads (hipcc -S: the next dot
                        // product then waited for rows requested a few instructions earlier), so there the order is
                        // also pinned in the IR: the loads stay before a memory clobber, B's FMAs behind it
                        if constexpr (sizeof(T) == 4 && KL % 2 == 0) {
                            asm volatile("" ::: "memory");
                            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                            for (int k = 0; k < KL; k += 2) {   // in register pairs: the accumulation stays v_pk_fma_f32
                                f32x2 t = {acc[k], acc[k + 1]};
                                asm volatile("" : "+v"(t));
                                acc[k] = t.x; acc[k + 1] = t.y;
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 0; k < KL; ++k) acc[k] = fma_t(q1, bB[k], acc[k]);
                        load_lane<T, NV, LPC>(lds_row<T>(lds_raw, n1), sub, bB);
                    } else {
                        load_lane<T, NV, LPC>(lds_row<T>(lds_raw, n0), sub, bA);
                        load_lane<T, NV, LPC>(lds_row<T>(lds_raw, n1), sub, bB);
                    }
                    if (MODE == MODE_LLH) {
                        if (LPC == 1) {
                            if (x0 > T(0)) lacc.add((double)x0, (double)s0, llh_tab);
                            if (x1 > T(0)) lacc.add((double)x1, (double)s1, llh_tab);
                        } else {
                            // every lane of the group knows s0 and s1: lane 0 takes the first
                            // nonzero, lane 1 the second
                            const T sm = (sub & 1) ? s1 : s0;
                            const T xm = (sub & 1) ? x1 : x0;
                            if (sub < 2 && xm > T(0)) lacc.add((double)xm, (double)sm, llh_tab);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
            };
            // whole turns of the ring run without a branch per step: with the four guarded steps in one loop
            // body the compiler gave the row registers different homes on different paths and paid for it with
            // 10 v_mov_b32 per step in the float32 kernel (hipcc -S; 16 % of its VALU instructions)
            int p = 0;
            for (; p + RING <= nsl; p += RING) {
#define SCHPF_PIPE_STEP(I)                                                                              \
    ring[I] = EF::load(a.entries, pos + (size_t)(p + I + RING) * GPW);                                  \
    pipe_step(std::integral_constant<int, I>{});
                SCHPF_PIPE_STEP(0) SCHPF_PIPE_STEP(1) SCHPF_PIPE_STEP(2) SCHPF_PIPE_STEP(3)
#undef SCHPF_PIPE_STEP
            }
            if (p < nsl) {
                // the ragged turn: slot i was decoded one step ago: refill it; decode the NEXT step's slot.  Past
                // the window's last step that is the next window's entry or padding: its indices are in range, the
                // rows read with them are never used
#define SCHPF_PIPE_STEP(I)                                                                              \
    ring[I] = EF::load(a.entries, pos + (size_t)(p + I + RING) * GPW);                                  \
    if (p + I < nsl) pipe_step(std::integral_constant<int, I>{});
                SCHPF_PIPE_STEP(0) SCHPF_PIPE_STEP(1) SCHPF_PIPE_STEP(2) SCHPF_PIPE_STEP(3)
#undef SCHPF_PIPE_STEP
            }
        } else if (ROLL) {
            // Wide rows (two of them do not fit the registers beside the accumulators; K = 50 in f64: 14 doubles
            // per lane): the pipeline of the paired loop with ONE row buffer.  The row of the next nonzero -- the
            // step's second, or the next step's first -- is fetched into the buffer vector by vector, each 16-byte
            // piece right behind the accumulation that read it last, so that its LDS latency runs under the rest of
            // the accumulation instead of in front of the next dot product (the loop this replaces fetched a row,
            // waited, and only then started).  As in the paired loop there is no test of the normaliser: an
            // underflowed s poisons the group's accumulators (inf / NaN), which is detected once after the task.
            typedef typename Vec16<T>::type V16;
            unsigned second = 0;   // LDS position of the current step's second row
            const int nz = a.single ? steps : 2 * steps;   // nonzeros (slot halves) this wave executes in the window
            if (nsl > 0) {         // prologue: the first step's first row (the ring was primed before the barrier)
                const E c = ring[0];
                unsigned i0 = EF::idx(c, 0);
                second = EF::idx(c, 1);
                xc[0][0] = EF::val(c, 0); xc[0][1] = EF::val(c, 1);
                asm volatile("" : "+v"(i0), "+v"(second), "+v"(xc[0][0]), "+v"(xc[0][1]));
                load_lane<T, NV, LPC>(lds_row<T>(lds_raw, i0), sub, bA);
            }
            // one nonzero: its weight from the row in the buffer, then the buffer becomes row `next`
            auto roll_nonzero = [&](const T x, const unsigned next) {
                const T s = group_dot<T, KL, LPC>(tm, bA);
                const V16 *__restrict__ np = reinterpret_cast<const V16 *>(lds_row<T>(lds_raw, next)) + sub;
                if (MODE == MODE_PHI) {
                    const T q = fast_div(x, s);
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[v * VEC + e] = fma_t(q, bA[v * VEC + e], acc[v * VEC + e]);
                        Vec16<T>::unpack(np[v * LPC], &bA[v * VEC]);
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < NV; ++v) Vec16<T>::unpack(np[v * LPC], &bA[v * VEC]);
                    // lane 0 of the group keeps the group's share
                    if ((LPC == 1 || sub == 0) && x > T(0)) lacc.add((double)x, (double)s, llh_tab);
                }
            };
            auto roll_step = [&](auto I_) {
                constexpr int I = decltype(I_)::value;
                const E cn = ring[(I + 1) % RING];
                unsigned n0 = EF::idx(cn, 0), n1 = EF::idx(cn, 1);
                xc[(I + 1) & 1][0] = EF::val(cn, 0); xc[(I + 1) & 1][1] = EF::val(cn, 1);
                asm volatile("" : "+v"(n0), "+v"(n1), "+v"(xc[(I + 1) & 1][0]), "+v"(xc[(I + 1) & 1][1]));
                roll_nonzero((T)xc[I & 1][0], second);
                __builtin_amdgcn_sched_barrier(0);
                roll_nonzero((T)xc[I & 1][1], n0);
                second = n1;
                __builtin_amdgcn_sched_barrier(0);
            };
            // a turn with guards: slot p + I holds the nonzeros 2 (p + I) and 2 (p + I) + 1 of the wave's nz; whole slots only
#define SCHPF_ROLL_GUARDED(I)                                                        \
    ring[I] = EF::load(a.entries, pos + (size_t)(p + I + RING) * GPW);               \
    if (2 * (p + I) + 1 < nz) roll_step(std::integral_constant<int, I>{});
            int p = 0;
            for (; 2 * (p + RING) <= nz; p += RING) {
#define SCHPF_ROLL_STEP(I)                                                           \
    ring[I] = EF::load(a.entries, pos + (size_t)(p + I + RING) * GPW);               \
    roll_step(std::integral_constant<int, I>{});
                SCHPF_ROLL_STEP(0) SCHPF_ROLL_STEP(1) SCHPF_ROLL_STEP(2) SCHPF_ROLL_STEP(3)
#undef SCHPF_ROLL_STEP
            }
            if (2 * p + 1 < nz) {
                SCHPF_ROLL_GUARDED(0) SCHPF_ROLL_GUARDED(1) SCHPF_ROLL_GUARDED(2) SCHPF_ROLL_GUARDED(3)
            }
#undef SCHPF_ROLL_GUARDED
            // an odd count (`single`): the first nonzero of slot nz / 2 alone.  Its row is in the buffer (the last whole
            // slot's second nonzero, or the prologue, fetched it), its count was decoded into the register pair of the
            // slot's parity; the row fetched behind it is never used
            if (nz & 1) {
                const float xl = ((nz >> 1) & 1) ? xc[1][0] : xc[0][0];
                roll_nonzero((T)xl, second);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
        for (int p = 0; p < nsl; p += RING) {
#pragma unroll
            for (int i = 0; i < RING; ++i) {
                const E c = ring[i];
                unsigned i0 = EF::idx(c, 0), i1 = EF::idx(c, 1);
                float xf0 = EF::val(c, 0), xf1 = EF::val(c, 1);
                // decode before the refill so that the slot's registers are free for it (otherwise
                // the refill lands in other registers and the loop needs copies behind a vmcnt(0))
                asm volatile("" : "+v"(i0), "+v"(i1), "+v"(xf0), "+v"(xf1));
                ring[i] = EF::load(a.entries, pos + (size_t)(p + i + RING) * GPW);   // may run past: padded
                if (p + i < nsl) {                                     // scalar branch
                    if (MODE == MODE_RANDOM) {
                        // t = 0 responsibilities (reference scHPF_.py:652-655), counter-based draws
#pragma unroll 1
                        for (int u = 0; u < 2; ++u) {
                            unsigned minor = (unsigned)entry_minor(u ? i1 : i0, w, a.win_rows, KP * (int)sizeof(T) / 16, a.ring, a.slot_bytes / 16);
                            if (a.minor_of) minor = (unsigned)a.minor_of[(size_t)blk * a.n_virtual + minor];
                            const double x = (double)(u ? xf1 : xf0);
                            if (!(x > 0.0)) continue;
                            const uint64_t cell = a.major_is_cell ? (uint64_t)major : (uint64_t)minor;
                            const uint64_t gene = a.major_is_cell ? (uint64_t)minor : (uint64_t)major;
                            const uint64_t base = draw_base(a.seed, cell, gene);
                            double d[KL];
                            double s = 0.0;
#pragma unroll
                            for (int k = 0; k < KL; ++k) {
                                const int f = factor_of<T, LPC>(k, sub);
                                d[k] = f < a.K ? exp1_draw(base, (unsigned)f) : 0.0;
                                s += d[k];
                            }
                            s = group_sum<double, LPC>(s);
                            const double wgt = x / s;
#pragma unroll
                            for (int k = 0; k < KL; ++k) acc[k] += (T)(wgt * d[k]);
                        }
                    } else {
                        // wide rows (two of them do not fit the registers): one nonzero at a time.  As in the
                        // pipelined loop there is no test of the normaliser: an underflowed s poisons the
                        // group's accumulators (inf / NaN), which is detected once after the task.  With `single`
                        // step counts an odd count does not execute the second half of its last slot
                        const int halves = a.single ? min(2, steps - 2 * (p + i)) : 2;
#pragma unroll 1
                        for (int u = 0; u < halves; ++u) {
                            T b[KL];
                            load_lane<T, NV, LPC>(lds_row<T>(lds_raw, u ? i1 : i0), sub, b);
                            const T x = (T)(u ? xf1 : xf0);
                            const T s = group_dot<T, KL, LPC>(tm, b);
                            if (MODE == MODE_PHI) {
                                const T q = fast_div(x, s);
#pragma unroll
                                for (int k = 0; k < KL; ++k) acc[k] = fma_t(q, b[k], acc[k]);
                            } else {
                                // lane 0 of the group keeps the group's share
                                if ((LPC == 1 || sub == 0) && x > T(0)) lacc.add((double)x, (double)s, llh_tab);
                            }
                        }
                    }
                    // keep the steps apart: without this fence the compiler hoists every LDS row
                    // load of the unrolled ring to the top and spills
                    asm volatile("" ::: "memory");
                }
            }
        }
        pos += (size_t)nsl * GPW;
        if (w + 1 < w1) {   // prime the ring for the next window before its staging barrier
            steps = __builtin_amdgcn_readfirstlane((int)st[w + 1]);
#pragma unroll
            for (int i = 0; i < RING; ++i) ring[i] = EF::load(a.entries, pos + (size_t)i * GPW);
        }
    }

    if (MODE == MODE_LLH) {
        llh += lacc.total();   // zero where nothing was added
        // which lanes hold a share: all (LPC 1), lanes 0-1 of a group (paired steps), lane 0 (else)
        if (LPC > 1 && !(PAIR ? sub < 2 : sub == 0)) llh = 0.0;
        llh = wave_sum(llh);
        if (lane == 0) a.wave_out[(size_t)task * a.wpb + wv] = llh;
        return;
    }
    T *out_row = a.partials + ((size_t)task * gpb + g) * KP;
    if (MODE == MODE_PHI) {   // non-finite accumulators <=> some normaliser underflowed (see the loops)
        T probe = T(0);
#pragma unroll
        for (int k = 0; k < KL; ++k) probe = fma_t(acc[k], T(0), probe);   // 0, or NaN
        any_bad = !(probe == T(0));
    }
    if (MODE == MODE_PHI && __builtin_expect(any_bad && live, 0)) {   // group-uniform; rare: see slow_nonzero
        slow_task_row<T, NV, LPC, PACK>(a.entries, (size_t)a.task_wave_off[(size_t)task * a.wpb + wv] + grp, st, w0, w1,
                                        a.win_rows, a.ring, a.slot_bytes / 16, a.single, GPW, a.log_major + (size_t)major * KP,
                                        a.log_minor, sub, a.K, out_row,
                                        BAL ? a.minor_of + (size_t)blk * a.n_virtual : nullptr);
        return;
    }
    if (MODE == MODE_PHI) {
#pragma unroll
        for (int k = 0; k < KL; ++k) acc[k] = live ? acc[k] * tm[k] : T(0);
    }
    // dead groups write zeros too: every partial row of the task is defined after a sweep
    store_lane<T, NV, LPC>(out_row, sub, acc);
}


template <typename T, int NV, int LPC, int MODE, int MAXT, bool PACK, bool BAL = false>
__device__ __forceinline__ void tile_sweep_task(const TileArgs<T> &a, const int task)
{
    tile_sweep_task_window<T, NV, LPC, MODE, MAXT, PACK, BAL>(a, task);
}

// Shader clock DURING a sweep launch: workgroup 0 (persistent launches: resident from the first slot until the list is
// empty, i.e. for the whole launch) stamps the shader-cycle counter (s_memtime; one tick = one shader cycle,
// MI355X_MICROARCH.md) and the constant-rate counter (s_memrealtime) on entry and adds the differences on exit: the
// quotient is the clock the chip sustained under THIS kernel's load -- DVFS moves it by +-10 % between boxes and bodies
// (same guide, "DVFS give-back").  The stamps live in memory, not in registers, across the task loop.
__device__ __forceinline__ void clock_probe_begin(unsigned long long *p)
{
    if (p && blockIdx.x == 0 && threadIdx.x == 0) {
        p[2] = __builtin_readcyclecounter();
        p[3] = __builtin_amdgcn_s_memrealtime();
    }
}
__device__ __forceinline__ void clock_probe_end(unsigned long long *p)
{
    if (p && blockIdx.x == 0 && threadIdx.x == 0) {
        p[0] += __builtin_readcyclecounter() - p[2];
        p[1] += __builtin_amdgcn_s_memrealtime() - p[3];
        p[4] += 1;
    }
}

// launch slot -> task: longest tasks first (plan.h task_order), so the launch has a short tail
// a.queue: persistent workgroups, as in tile_sweep_dual_kernel below
template <typename T, int NV, int LPC, int MODE, int MAXT, bool PACK, bool BAL = false>
__global__ __launch_bounds__(MAXT) void tile_sweep_kernel(TileArgs<T> a)
{
    __shared__ int next_slot;
    int slot = blockIdx.x;
    if (MODE != MODE_RANDOM) clock_probe_begin(a.clock_probe);
    for (;;) {
        const int task = a.task_order ? a.task_order[slot] : slot;
        tile_sweep_task<T, NV, LPC, MODE, MAXT, PACK, BAL>(a, task);
        if (MODE == MODE_RANDOM) return;
        if (!a.queue) { clock_probe_end(a.clock_probe); return; }
        __syncthreads();
        if (threadIdx.x == 0) next_slot = (int)gridDim.x + atomicAdd(&a.queue[0], 1);
        __syncthreads();
        slot = next_slot;
        if (slot >= a.n_tasks) break;
    }
    clock_probe_end(a.clock_probe);
    if (threadIdx.x == 0 && atomicAdd(&a.queue[1], 1) == (int)gridDim.x - 1) {
        a.queue[0] = 0;
        a.queue[1] = 0;
    }
}
// Both sweeps of an iteration in ONE launch (they read the same old tables and write disjoint
// partials): order[slot] = task of the cell-side plan, or ~task of the gene-side plan, merged
// longest-first.  One launch has one tail instead of two and the two task pools fill each
// other's idle compute units.
//
// queue == nullptr: one workgroup per slot.  Otherwise PERSISTENT workgroups (as many as the device
// holds at once): a workgroup starts on slot blockIdx.x and, when its task is done, draws the next
// slot from queue[0] -- no workgroup launch / teardown between the tasks of a compute unit, and the
// longest-first list is balanced by who is free, not by the dispatcher's round-robin.  queue[1]
// counts the workgroups that have found the list empty; the last one zeroes both words for the
// next launch.
template <typename T, int NV, int LPC, int MAXT, bool PACK, bool BAL = false>
__global__ __launch_bounds__(MAXT) void tile_sweep_dual_kernel(TileArgs<T> a0, TileArgs<T> a1,
                                                              const int *__restrict__ order, int n_slots,
                                                              int *__restrict__ queue)
{
    __shared__ int next_slot;
    int slot = blockIdx.x;
    clock_probe_begin(a0.clock_probe);
    for (;;) {
        const int code = order[slot];
        if (code >= 0) tile_sweep_task<T, NV, LPC, MODE_PHI, MAXT, PACK, BAL>(a0, code);
        else tile_sweep_task<T, NV, LPC, MODE_PHI, MAXT, PACK, BAL>(a1, ~code);
        if (!queue) { clock_probe_end(a0.clock_probe); return; }
        __syncthreads();                                   // the window and next_slot are free again
        if (threadIdx.x == 0) next_slot = (int)gridDim.x + atomicAdd(&queue[0], 1);
        __syncthreads();
        slot = next_slot;
        if (slot >= n_slots) break;
    }
    clock_probe_end(a0.clock_probe);
    if (threadIdx.x == 0 && atomicAdd(&queue[1], 1) == (int)gridDim.x - 1) {
        queue[0] = 0;
        queue[1] = 0;
    }
}

// The opt-in to more than 64 KiB of dynamic LDS is a per-DEVICE property of a kernel: one bit per device
// and instantiation (a process may drive several GPUs: scHPF.fit(devices=...), run_trials_pool).  Returns
// whether the calling thread's current device still has to raise it, and marks it raised.
static inline bool lds_opt_in_pending(std::atomic<uint64_t> &raised)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;   // unknown: raise every time
    const uint64_t bit = (uint64_t)1 << dev;
    return (raised.fetch_or(bit) & bit) == 0;
}

template <typename T, int NV, int LPC, int MAXT, bool PACK, bool BAL = false>
static hipError_t launch_tile_b(const TileArgs<T> &a_in, int mode, int64_t n_tasks, int threads, size_t lds_bytes,
                                hipStream_t st)
{
    TileArgs<T> a = a_in;
    if (mode == MODE_RANDOM || a.resident <= 0 || a.resident >= n_tasks) a.queue = nullptr;   // one round
    a.n_tasks = (int)n_tasks;
    dim3 grid((unsigned)(a.queue ? a.resident : n_tasks)), block((unsigned)threads);
    if (lds_bytes > 64 * 1024) {   // opt in to the full 160 KiB of a CU, once per instantiation
        static std::atomic<uint64_t> raised{0};
        if (lds_opt_in_pending(raised)) {
            // not the full 160 KiB: the kernels have a static word of LDS of their own (next_slot)
            hipError_t e = hipFuncSetAttribute((const void *)tile_sweep_kernel<T, NV, LPC, MODE_PHI, MAXT, PACK, BAL>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void *)tile_sweep_kernel<T, NV, LPC, MODE_LLH, MAXT, PACK, BAL>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
            if (e != hipSuccess) { raised = 0; return e; }
        }
    }
    if (mode == MODE_PHI)
        hipLaunchKernelGGL((tile_sweep_kernel<T, NV, LPC, MODE_PHI, MAXT, PACK, BAL>), grid, block, lds_bytes, st, a);
    else if (mode == MODE_LLH)
        hipLaunchKernelGGL((tile_sweep_kernel<T, NV, LPC, MODE_LLH, MAXT, PACK, BAL>), grid, block, lds_bytes, st, a);
    else   // one-off: the 1024-thread bound serves every workgroup size (one instantiation instead of two)
        hipLaunchKernelGGL((tile_sweep_kernel<T, NV, LPC, MODE_RANDOM, 1024, PACK>), grid, block, 0, st, a);
    return hipGetLastError();
}
// the launch bound caps the register budget: 1024 threads -> 128 VGPRs, 512 -> 256
template <typename T, int NV, int LPC>
static hipError_t launch_tile_t(const TileArgs<T> &a, int mode, int packed, int64_t n_tasks, int threads,
                                size_t lds_bytes, hipStream_t st)
{
    if (n_tasks == 0) return hipSuccess;
    if (threads <= 512)
        return packed ? launch_tile_b<T, NV, LPC, 512, true>(a, mode, n_tasks, threads, lds_bytes, st)
                      : launch_tile_b<T, NV, LPC, 512, false>(a, mode, n_tasks, threads, lds_bytes, st);
    if (a.minor_of && mode != MODE_RANDOM)   // balanced windows: capi.hip builds them for 1024-thread workgroups only
        return packed ? launch_tile_b<T, NV, LPC, 1024, true, true>(a, mode, n_tasks, threads, lds_bytes, st)
                      : launch_tile_b<T, NV, LPC, 1024, false, true>(a, mode, n_tasks, threads, lds_bytes, st);
    return packed ? launch_tile_b<T, NV, LPC, 1024, true>(a, mode, n_tasks, threads, lds_bytes, st)
                  : launch_tile_b<T, NV, LPC, 1024, false>(a, mode, n_tasks, threads, lds_bytes, st);
}

template <typename T, int NV, int LPC, int MAXT, bool PACK, bool BAL = false>
static hipError_t launch_dual_b(const TileArgs<T> &a0, const TileArgs<T> &a1, const int *order, int64_t n_slots,
                                int threads, size_t lds_bytes, int *queue, int resident, hipStream_t st)
{
    if (lds_bytes > 64 * 1024) {
        static std::atomic<uint64_t> raised{0};
        if (lds_opt_in_pending(raised)) {
            // not the full 160 KiB: the kernel has a static word of LDS of its own (next_slot)
            hipError_t e = hipFuncSetAttribute((const void *)tile_sweep_dual_kernel<T, NV, LPC, MAXT, PACK, BAL>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
            if (e != hipSuccess) { raised = 0; return e; }
        }
    }
    if (queue && resident >= n_slots) queue = nullptr;   // one round: nothing to draw
    const unsigned grid = queue ? (unsigned)resident : (unsigned)n_slots;
    hipLaunchKernelGGL((tile_sweep_dual_kernel<T, NV, LPC, MAXT, PACK, BAL>), dim3(grid), dim3((unsigned)threads), lds_bytes,
                       st, a0, a1, order, (int)n_slots, queue);
    return hipGetLastError();
}
template <typename T, int NV, int LPC>
static hipError_t launch_dual_t(const TileArgs<T> &a0, const TileArgs<T> &a1, const int *order, int packed,
                                int64_t n_slots, int threads, size_t lds_bytes, int *queue, int resident,
                                hipStream_t st)
{
    if (n_slots == 0) return hipSuccess;
    if (threads <= 512)
        return packed ? launch_dual_b<T, NV, LPC, 512, true>(a0, a1, order, n_slots, threads, lds_bytes, queue, resident, st)
                      : launch_dual_b<T, NV, LPC, 512, false>(a0, a1, order, n_slots, threads, lds_bytes, queue, resident, st);
    if ((a0.minor_of != nullptr) != (a1.minor_of != nullptr)) return hipErrorInvalidValue;   // both plans balanced, or neither
    if (a0.minor_of)
        return packed ? launch_dual_b<T, NV, LPC, 1024, true, true>(a0, a1, order, n_slots, threads, lds_bytes, queue, resident, st)
                      : launch_dual_b<T, NV, LPC, 1024, false, true>(a0, a1, order, n_slots, threads, lds_bytes, queue, resident, st);
    return packed ? launch_dual_b<T, NV, LPC, 1024, true>(a0, a1, order, n_slots, threads, lds_bytes, queue, resident, st)
                  : launch_dual_b<T, NV, LPC, 1024, false>(a0, a1, order, n_slots, threads, lds_bytes, queue, resident, st);
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

// Only the (vectors per lane, lanes per row) pairs that capi.hip choose_config can pick are instantiated
// (kernels.h tile_combo_ok / gather_combo_ok list them): every pair costs 16 tile kernels or 3 gather
// kernels per dtype, and the full 9 x 5 grid made a 21 MB library.
#define SCHPF_COMBO(nv_, lpc_, CALLEXPR)                                      \
    if (nv == nv_ && lpc == lpc_) { constexpr int NV = nv_; constexpr int LPC = lpc_; return CALLEXPR; }
#ifdef SCHPF_DEV_FAST   /* development builds: only the K = 20 and K = 50 instantiations (tools/devbuild.sh) */
#define SCHPF_DISPATCH_TILE(nv, lpc, CALLEXPR)                                \
    SCHPF_COMBO(5, 1, CALLEXPR) SCHPF_COMBO(5, 2, CALLEXPR) SCHPF_COMBO(7, 4, CALLEXPR) SCHPF_COMBO(7, 2, CALLEXPR) \
    return hipErrorInvalidValue;
#define SCHPF_DISPATCH(nv, lpc, CALLEXPR)                                     \
    SCHPF_COMBO(5, 4, CALLEXPR) SCHPF_COMBO(3, 4, CALLEXPR) SCHPF_COMBO(7, 4, CALLEXPR) SCHPF_COMBO(4, 4, CALLEXPR) \
    return hipErrorInvalidValue;
#else
// tile sweeps: LPC 1 with 1..7 vectors; LPC 2, 4, 8 with 4..7; LPC 16 with 4 (rows of at most 1 KiB)
#define SCHPF_DISPATCH_TILE(nv, lpc, CALLEXPR)                                \
    SCHPF_COMBO(1, 1, CALLEXPR) SCHPF_COMBO(2, 1, CALLEXPR) SCHPF_COMBO(3, 1, CALLEXPR) SCHPF_COMBO(4, 1, CALLEXPR) \
    SCHPF_COMBO(5, 1, CALLEXPR) SCHPF_COMBO(6, 1, CALLEXPR) SCHPF_COMBO(7, 1, CALLEXPR)                             \
    SCHPF_COMBO(4, 2, CALLEXPR) SCHPF_COMBO(5, 2, CALLEXPR) SCHPF_COMBO(6, 2, CALLEXPR) SCHPF_COMBO(7, 2, CALLEXPR) \
    SCHPF_COMBO(4, 4, CALLEXPR) SCHPF_COMBO(5, 4, CALLEXPR) SCHPF_COMBO(6, 4, CALLEXPR) SCHPF_COMBO(7, 4, CALLEXPR) \
    SCHPF_COMBO(4, 8, CALLEXPR) SCHPF_COMBO(5, 8, CALLEXPR) SCHPF_COMBO(6, 8, CALLEXPR) SCHPF_COMBO(7, 8, CALLEXPR) \
    SCHPF_COMBO(4, 16, CALLEXPR)                                              \
    return hipErrorInvalidValue;
// gather sweeps: LPC 4 with 1..8, 10 vectors; LPC 8 with 6, 7, 8, 10; LPC 16 with 6, 7, 8
#define SCHPF_DISPATCH(nv, lpc, CALLEXPR)                                     \
    SCHPF_COMBO(1, 4, CALLEXPR) SCHPF_COMBO(2, 4, CALLEXPR) SCHPF_COMBO(3, 4, CALLEXPR) SCHPF_COMBO(4, 4, CALLEXPR) \
    SCHPF_COMBO(5, 4, CALLEXPR) SCHPF_COMBO(6, 4, CALLEXPR) SCHPF_COMBO(7, 4, CALLEXPR) SCHPF_COMBO(8, 4, CALLEXPR) \
    SCHPF_COMBO(10, 4, CALLEXPR)                                              \
    SCHPF_COMBO(6, 8, CALLEXPR) SCHPF_COMBO(7, 8, CALLEXPR) SCHPF_COMBO(8, 8, CALLEXPR) SCHPF_COMBO(10, 8, CALLEXPR) \
    SCHPF_COMBO(6, 16, CALLEXPR) SCHPF_COMBO(7, 16, CALLEXPR) SCHPF_COMBO(8, 16, CALLEXPR)                          \
    return hipErrorInvalidValue;
#endif

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
template <typename T>
hipError_t launch_tile_sweep(const TileArgs<T> &a, int nv, int lpc, int mode, int packed, int64_t n_tasks,
                             int threads, size_t lds_bytes, hipStream_t st)
{
    SCHPF_DISPATCH_TILE(nv, lpc, (launch_tile_t<T, NV, LPC>(a, mode, packed, n_tasks, threads, lds_bytes, st)))
}

template <typename T>
hipError_t launch_tile_sweep_dual(const TileArgs<T> &a0, const TileArgs<T> &a1, const int *order, int nv, int lpc,
                                  int packed, int64_t n_slots, int threads, size_t lds_bytes, int *queue, int resident,
                                  hipStream_t st)
{
    SCHPF_DISPATCH_TILE(nv, lpc,
                        (launch_dual_t<T, NV, LPC>(a0, a1, order, packed, n_slots, threads, lds_bytes, queue, resident, st)))
}

}  // namespace schpf
