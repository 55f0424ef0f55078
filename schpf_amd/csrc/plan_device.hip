// Tile plan built on the GPU: the O(nnz) passes of plan.cpp::build_tile_plan (sort by
// (major, minor), per-window step counts, sliced-ELL fill in LDS-bank-aware order) as device
// passes over the uploaded COO; everything that does not touch the nonzeros (blocks, tasks,
// offsets) is the shared host code of plan.cpp.  The result is bit-identical to the host
// builder's (tests/test_engine_gpu.py::test_device_plan_equals_host_plan) -- it only takes a
// tenth of the time, which matters because a whole fit at 1e8 nonzeros is ~0.3 s of iterations.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "plan.h"

namespace schpf {
namespace {

#define PD_CHECK(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ == hipErrorOutOfMemory) { (void)hipGetLastError(); throw DeviceNoMemory(std::string(#expr ": ") + hipGetErrorString(e_)); } \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr ": ") + hipGetErrorString(e_)); \
    } while (0)

struct Tmp {   // scratch device allocation, freed with the scope
    void *p = nullptr;
    explicit Tmp(size_t bytes) { PD_CHECK(hipMalloc(&p, bytes ? bytes : 16)); }
    ~Tmp() { if (p) (void)hipFree(p); }
    Tmp(const Tmp &) = delete;
    Tmp &operator=(const Tmp &) = delete;
    template <typename U> U *as() const { return static_cast<U *>(p); }
};

// key = major << minor_bits | minor: only the occupied bits are sorted
__global__ void make_keys_kernel(int64_t n, const int32_t *__restrict__ major, const int32_t *__restrict__ minor,
                                 int minor_bits, uint64_t *__restrict__ key, int32_t *__restrict__ idx)
{
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        key[j] = ((uint64_t)(uint32_t)major[j] << minor_bits) | (uint32_t)minor[j];
        idx[j] = (int32_t)j;
    }
}
__global__ void split_keys_kernel(int64_t n, const uint64_t *__restrict__ key, const int32_t *__restrict__ idx,
                                  const float *__restrict__ val, int minor_bits, int32_t *__restrict__ s_major,
                                  int32_t *__restrict__ s_minor, float *__restrict__ s_val)
{
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = key[j];
        s_major[j] = (int32_t)(k >> minor_bits);
        s_minor[j] = (int32_t)(k & (((uint64_t)1 << minor_bits) - 1));
        s_val[j] = val[idx[j]];
    }
}
// run pointers of the sorted major index (same rule as plan.cpp::sort_by_major_minor)
__global__ void run_pointers_kernel(int64_t n, const int32_t *__restrict__ s_major, int n_major, int64_t *__restrict__ mptr)
{
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        if (j == 0) {
            for (int32_t k = 0; k <= s_major[0]; ++k) mptr[k] = 0;
        } else if (s_major[j] != s_major[j - 1]) {
            for (int32_t k = s_major[j - 1] + 1; k <= s_major[j]; ++k) mptr[k] = j;
        }
    }
}
__global__ void fill_i64_kernel(int64_t n, int64_t v, int64_t *__restrict__ out)
{
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) out[j] = v;
}

struct Geometry {
    int gpb, gpw, wpb, W, win_rows, lpc, lpc_shift, n_classes, row_slots, packed;
    int ring, slot16, look;
    int single;   // steps count nonzeros (plan.h)
};

// LDS position of minor row m in 16-byte units (plan.h tile_off16)
__device__ __forceinline__ uint32_t dev_off16(const Geometry &g, int32_t m)
{
    const int32_t w = m / g.win_rows;
    const uint32_t r = (uint32_t)(m - w * g.win_rows) * (uint32_t)g.row_slots;
    return g.ring > 1 ? (uint32_t)(w % g.ring) * (uint32_t)g.slot16 + r : r;
}

// first position in [lo, hi) whose minor index is >= key (the row's minors ascend)
__device__ __forceinline__ int64_t lower_bound_minor(const int32_t *__restrict__ s_minor, int64_t lo, int64_t hi, int64_t key)
{
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)s_minor[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// One thread per (row, window) segment -- NOT per row: real count matrices have rows (genes) with
// 1e5 nonzeros next to rows with ten, and a thread per row would serialise on the longest.
// per (block, wave, window): steps = ceil(longest segment / 2)
__global__ void steps_kernel(Geometry g, int64_t n_slots, const int32_t *__restrict__ block_rows,
                             const int64_t *__restrict__ mptr, const int32_t *__restrict__ s_minor,
                             unsigned *__restrict__ steps32, int *__restrict__ err)
{
    const int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (id >= n_slots * g.W) return;
    const int64_t slot = id / g.W;
    const int w = (int)(id % g.W);
    const int32_t row = block_rows[slot];
    if (row < 0) return;
    const int64_t r0 = mptr[row], r1 = mptr[row + 1];
    if (r0 == r1) return;
    const int64_t lo = lower_bound_minor(s_minor, r0, r1, (int64_t)w * g.win_rows);
    const int64_t hi = lower_bound_minor(s_minor, lo, r1, ((int64_t)w + 1) * g.win_rows);
    if (hi == lo) return;
    const int64_t steps = g.single ? hi - lo : (hi - lo + 1) / 2;
    if (steps > 65535) { *err = 1; return; }
    const int64_t b = slot / g.gpb;
    const int v = (int)(slot % g.gpb) / g.gpw;
    atomicMax(steps32 + ((size_t)b * g.wpb + v) * g.W + w, (unsigned)steps);
}

__device__ __forceinline__ int bank_class(const Geometry &g, int32_t minor)
{
    return (int)(((dev_off16(g, minor) & 15u) >> g.lpc_shift) & (unsigned)(g.n_classes - 1));
}

// Ring mode (plan.h): the schedule of one block, plan.cpp::ring_schedule_block with the block's
// lanes as threads.  T_e goes to every wave's steps32 row, start[lane][e] = nonzeros the lane has
// consumed before epoch e.
__global__ __launch_bounds__(1024) void ring_schedule_kernel(Geometry g, const int32_t *__restrict__ block_rows,
                                                            const int64_t *__restrict__ mptr,
                                                            const int32_t *__restrict__ s_minor,
                                                            const int32_t *__restrict__ range_end,
                                                            unsigned *__restrict__ steps32, int32_t *__restrict__ start,
                                                            int *__restrict__ err)
{
    const int64_t b = blockIdx.x;
    const int gi = threadIdx.x;                       // lane group of the block; blockDim.x = gpb rounded up to waves
    const bool owner = gi < g.gpb;
    const int32_t row = owner ? block_rows[b * g.gpb + gi] : -1;
    const int64_t r0 = row >= 0 ? mptr[row] : 0, r1 = row >= 0 ? mptr[row + 1] : 0;
    int32_t *st = start + ((size_t)b * g.gpb + (owner ? gi : 0)) * ((size_t)g.W + 1);
    int64_t c_need = r0, c_hor = r0;                  // first nonzero with window >= e + 1 / >= horizon
    int32_t done = 0;
    // galloping lower bound from a cursor: the cursors only move forward
    auto advance = [&](int64_t from, int64_t bound) {
        if (from >= r1 || (int64_t)s_minor[from] >= bound) return from;
        int64_t step = 1, lo = from;                  // s_minor[lo] < bound
        while (lo + step < r1 && (int64_t)s_minor[lo + step] < bound) { lo += step; step <<= 1; }
        return lower_bound_minor(s_minor, lo + 1, lo + step < r1 ? lo + step : r1, bound);
    };
    for (int e = 0; e < g.W; ++e) {
        const int w1 = range_end[e];                  // end of the task e belongs to (plan.h range_end_of_window)
        const int hor = min(e + g.look + 1, w1);
        c_need = advance(c_need, ((int64_t)e + 1) * g.win_rows);
        if (c_hor < c_need) c_hor = c_need;
        c_hor = advance(c_hor, (int64_t)hor * g.win_rows);
        int need = (int)(c_need - r0) - done;
        // maximum over the sweep kernel's wave (gpw lane groups, a power of two <= 64) ...
        for (int m = g.gpw >> 1; m >= 1; m >>= 1) need = max(need, __shfl_xor(need, m, 64));
        const unsigned Te = (unsigned)((need + 1) / 2);   // plan.cpp::ring_schedule_block
        if (Te > 65535u) *err = 1;
        if (owner && gi % g.gpw == 0) steps32[((size_t)b * g.wpb + gi / g.gpw) * g.W + e] = Te;
        if (owner) st[e] = done;
        done += min((int32_t)(2 * Te), (int32_t)(c_hor - r0) - done);
    }
    if (owner) st[g.W] = done;
}

// One segment dealt on its own: the LDS-bank-aware order of plan.cpp::bank_order (wished class
// (rank + t) mod classes, else the fullest class; inside a class in minor order).  off = first
// step slot of the segment's (block, wave, window).
__device__ void fill_segment_rowwise(const Geometry &g, int64_t s, int64_t e_, int rank, int64_t off, int gslot,
                                     const int32_t *__restrict__ s_minor, const float *__restrict__ s_val,
                                     uint32_t *__restrict__ entries)
{
    const int n = (int)(e_ - s);
    const bool ordered = g.n_classes > 1 && n > 2;
    int cnt[16];
    int64_t cur[16];
    for (int c = 0; c < 16; ++c) { cnt[c] = 0; cur[c] = s; }
    if (ordered)
        for (int64_t j = s; j < e_; ++j) cnt[bank_class(g, s_minor[j])]++;
    for (int t = 0; t < n; ++t) {
        int64_t src;
        if (!ordered) {
            src = s + t;
        } else {
            int c = (int)(((unsigned)rank + (unsigned)t) & (unsigned)(g.n_classes - 1));
            if (cnt[c] == 0) {
                int best = 0;
                for (int k = 0; k < g.n_classes; ++k)
                    if (cnt[k] > best) { best = cnt[k]; c = k; }
            }
            int64_t q = cur[c];
            while (bank_class(g, s_minor[q]) != c) ++q;   // next unused nonzero of class c, in minor order
            src = q;
            cur[c] = q + 1;
            cnt[c]--;
        }
        const uint32_t o16 = dev_off16(g, s_minor[src]);
        const size_t step_slot = (size_t)off + (size_t)(t >> 1) * g.gpw + gslot;
        if (g.packed) {
            uint32_t *e = entries + step_slot * 2;
            const int sh = (t & 1) * 16;
            e[0] = (e[0] & ~(0xFFFFu << sh)) | (o16 << sh);
            e[1] |= (uint32_t)s_val[src] << sh;
        } else {
            uint32_t *e = entries + step_slot * 4 + (size_t)(t & 1) * 2;
            e[0] = o16;
            e[1] = __float_as_uint(s_val[src]);
        }
    }
}

// the segment of one lane group in one window: [s, e_) of the sorted arrays
__device__ __forceinline__ void segment_bounds(const Geometry &g, int64_t slot, int w, int64_t r0, int64_t r1,
                                               const int32_t *__restrict__ s_minor, const int32_t *__restrict__ start,
                                               int64_t &s, int64_t &e_)
{
    if (g.ring > 1) {   // ring mode: what the schedule gave this lane in epoch w
        const int32_t *st = start + (size_t)slot * ((size_t)g.W + 1);
        s = r0 + st[w];
        e_ = r0 + st[w + 1];
    } else {
        const int32_t base = w * g.win_rows;
        s = lower_bound_minor(s_minor, r0, r1, (int64_t)base);
        e_ = lower_bound_minor(s_minor, s, r1, (int64_t)base + g.win_rows);
    }
}

// fill, one thread per (row, window) segment.  win_off[(block, wave), window] = first step
// slot of that window's entries.
__global__ void fill_kernel(Geometry g, int64_t n_slots, const int32_t *__restrict__ block_rows,
                            const int64_t *__restrict__ mptr, const int32_t *__restrict__ s_minor,
                            const float *__restrict__ s_val, const int64_t *__restrict__ win_off,
                            const int *__restrict__ pass_rank, const int32_t *__restrict__ start,
                            uint32_t *__restrict__ entries)
{
    const int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (id >= n_slots * g.W) return;
    const int64_t slot = id / g.W;
    const int w = (int)(id % g.W);
    const int32_t row = block_rows[slot];
    if (row < 0) return;
    const int64_t r0 = mptr[row], r1 = mptr[row + 1];
    if (r0 == r1) return;
    int64_t s, e_;
    segment_bounds(g, slot, w, r0, r1, s_minor, start, s, e_);
    if (e_ == s) return;
    const int64_t b = slot / g.gpb;
    const int gi = (int)(slot % g.gpb);
    const int gslot = gi % g.gpw;
    const size_t bw = (size_t)b * g.wpb + gi / g.gpw;
    fill_segment_rowwise(g, s, e_, pass_rank[gslot], win_off[bw * g.W + w], gslot, s_minor, s_val, entries);
}

// fill with the joint bank assignment (plan.cpp::bank_order_joint).  A (block, wave, window) unit is
// worked by gpw lanes -- lane m is lane group m of the sweep's wave -- and a 64-lane workgroup here
// takes 64 / gpw units.  Every lane first sorts its segment by class into LDS (counts, per-class lists
// of positions; the minors are fetched eight at a time so that the loop pays one memory latency per
// eight nonzeros); then, position by position, the groups of a pass pick in turn -- one lane of each
// pass at a time, its choice handed to the pass by a cross-lane read -- and note the sequence in LDS;
// the entries are written afterwards, two nonzeros per 8- or 16-byte store, so that no global load
// sits inside the serial part.  Segments longer than TILE_JOINT_MAX go the per-row way.
template <int NC>
__global__ __launch_bounds__(64) void fill_joint_kernel(Geometry g, int64_t n_units, const int32_t *__restrict__ block_rows,
                                                        const int64_t *__restrict__ mptr,
                                                        const int32_t *__restrict__ s_minor,
                                                        const float *__restrict__ s_val,
                                                        const int64_t *__restrict__ win_off,
                                                        const int *__restrict__ pass_rank,
                                                        const int *__restrict__ pass_of, const int32_t *__restrict__ start,
                                                        uint32_t *__restrict__ entries)
{
    __shared__ uint8_t cnt[16 * 64], head[16 * 64];
    __shared__ uint8_t pos[TILE_JOINT_MAX * 64], seq[TILE_JOINT_MAX * 64];
    __shared__ int lane_of[64];                       // [unit of the workgroup][pass * rank] -> lane
    const int l = threadIdx.x;
    const int sub = l / g.gpw, m = l - sub * g.gpw;   // gpw divides 64
    const int upw = 64 / g.gpw;
    const int64_t unit = (int64_t)blockIdx.x * upw + sub;
    const bool member = unit < n_units;
    const int64_t bw = member ? unit / g.W : 0;
    const int w = member ? (int)(unit % g.W) : 0;
    const int rank = pass_rank[m], pass = pass_of[m];
    lane_of[sub * g.gpw + pass * NC + rank] = l;      // 4 passes of NC groups = the gpw lanes of the unit
    const int64_t slot = bw * g.gpw + m;
    const int32_t row = member ? block_rows[slot] : -1;
    int64_t s = 0, e_ = 0;
    if (row >= 0) {
        const int64_t r0 = mptr[row], r1 = mptr[row + 1];
        if (r1 > r0) segment_bounds(g, slot, w, r0, r1, s_minor, start, s, e_);
    }
    const int n = (int)(e_ - s);
    const bool joint = n > 0 && n <= TILE_JOINT_MAX;
    const int64_t off = member ? win_off[unit] : 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) cnt[c * 64 + l] = 0;
    if (joint) {
        for (int j0 = 0; j0 < n; j0 += 8) {
            int32_t mm[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) mm[u] = j0 + u < n ? s_minor[s + j0 + u] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < n) {
                    const int c = bank_class(g, mm[u]);
                    seq[(j0 + u) * 64 + l] = (uint8_t)c;
                    cnt[c * 64 + l]++;
                }
        }
        int run = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) { head[c * 64 + l] = (uint8_t)run; run += cnt[c * 64 + l]; }
        for (int j = 0; j < n; ++j) {
            const int c = seq[j * 64 + l];
            pos[(int)(head[c * 64 + l]++) * 64 + l] = (uint8_t)j;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) head[c * 64 + l] -= cnt[c * 64 + l];
    }
    __syncthreads();
    int max_n = joint ? n : 0;
    for (int d = 32; d >= 1; d >>= 1) max_n = max(max_n, __shfl_xor(max_n, d, 64));
    const int *my_lanes = lane_of + sub * g.gpw + pass * NC;
    for (int t = 0; t < max_n; ++t) {
        // tile_joint_pick with the keys of the position in registers: every lane builds its own at once, a class
        // its pass has read is struck out of everybody's, and the lane whose turn it is takes the largest left
        const bool active = joint && t < n;
        const unsigned wish = ((unsigned)rank + (unsigned)t) & (unsigned)(NC - 1);
        unsigned key[NC], best_any = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const unsigned k = active ? (unsigned)cnt[c * 64 + l] : 0u;
            const unsigned d = ((unsigned)c - wish) & (unsigned)(NC - 1);
            key[c] = k ? (k << 4) | (15u - d) : 0u;
            best_any = max(best_any, key[c]);
        }
        int my_c = 0;
        for (int a = 0; a < NC; ++a) {
            const int rk = (a + t) & (NC - 1);
            unsigned best_free = 0;
#pragma unroll
            for (int c = 0; c < NC; ++c) best_free = max(best_free, key[c]);
            const unsigned k = best_free ? best_free : best_any;
            const int mine_c = (int)((wish + (15u - (k & 15u))) & (unsigned)(NC - 1));
            const bool mine = active && rank == rk;
            if (mine) my_c = mine_c;
            const int c = __shfl(mine ? mine_c : -1, my_lanes[rk], 64);
#pragma unroll
            for (int c2 = 0; c2 < NC; ++c2) key[c2] = c2 == c ? 0u : key[c2];
        }
        if (active) {
            const int q = head[my_c * 64 + l];
            seq[t * 64 + l] = pos[q * 64 + l];
            head[my_c * 64 + l] = (uint8_t)(q + 1);
            cnt[my_c * 64 + l]--;
        }
    }
    if (joint) {
        const uint32_t pad = g.ring > 1 ? (uint32_t)(w % g.ring) * (uint32_t)g.slot16 : 0u;
        for (int t0 = 0; t0 < n; t0 += 8) {
            int32_t mm[8];
            float vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool on = t0 + u < n;
                const int64_t j = s + (on ? seq[(t0 + u) * 64 + l] : 0);
                mm[u] = on ? s_minor[j] : -1;
                vv[u] = on ? s_val[j] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                if (t0 + u >= n) break;
                const uint32_t o0 = dev_off16(g, mm[u]);
                const uint32_t o1 = mm[u + 1] >= 0 ? dev_off16(g, mm[u + 1]) : pad;
                const size_t step_slot = (size_t)off + (size_t)((t0 + u) >> 1) * g.gpw + m;
                if (g.packed) {
                    uint2 e;
                    e.x = o0 | (o1 << 16);
                    e.y = (uint32_t)vv[u] | ((uint32_t)vv[u + 1] << 16);
                    *reinterpret_cast<uint2 *>(entries + step_slot * 2) = e;
                } else {
                    uint4 e;
                    e.x = o0; e.y = __float_as_uint(vv[u]); e.z = o1; e.w = __float_as_uint(vv[u + 1]);
                    *reinterpret_cast<uint4 *>(entries + step_slot * 4) = e;
                }
            }
        }
    } else if (n > 0) {
        fill_segment_rowwise(g, s, e_, rank, off, m, s_minor, s_val, entries);
    }
}

// ring mode: unused step slots point at the first row of their epoch's own slot (plan.cpp)
__global__ void ring_pad_kernel(Geometry g, int64_t n_bw, const int64_t *__restrict__ win_off,
                                const unsigned *__restrict__ steps32, uint32_t *__restrict__ entries)
{
    const int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (id >= n_bw * g.W) return;
    const int w = (int)(id % g.W);
    const int64_t stored = g.single ? ((int64_t)steps32[id] + 1) / 2 : (int64_t)steps32[id];
    const int64_t off = win_off[id], n = stored * g.gpw;
    const uint32_t o16 = (uint32_t)(w % g.ring) * (uint32_t)g.slot16;
    for (int64_t q = off; q < off + n; ++q) {
        if (g.packed) entries[(size_t)q * 2] = o16 | (o16 << 16);
        else { entries[(size_t)q * 4] = o16; entries[(size_t)q * 4 + 2] = o16; }
    }
}

inline unsigned grid_for(int64_t n, int threads)
{
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + threads - 1) / threads, 65535 * 4));
}

}  // namespace

// d_major / d_minor / d_val: the caller's COO on the device.  presorted: already ascending in
// (major, minor).  packed_ok: every count is an integer <= 65535.  Outputs: P (host metadata; its
// entries stay empty), device buffers entries / steps (uint16) / order (int32, nullptr when
// presorted: identity) that the caller owns and frees with hipFree.
void build_tile_plan_device(void *stream, int64_t nnz, const int32_t *d_major, const int32_t *d_minor,
                            const float *d_val, bool presorted, bool packed_ok, int n_major, int n_minor,
                            const TileShape &shape, TilePlanHost &P, void **out_entries, size_t *out_entries_bytes,
                            void **out_steps, void **out_order)
{
    const int lpc = shape.lpc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    *out_entries = nullptr; *out_steps = nullptr; *out_order = nullptr; *out_entries_bytes = 0;
    const int threads = 256;
    // ---- 1. (major, minor)-sorted copies
    const int32_t *s_major = d_major, *s_minor = d_minor;
    const float *s_val = d_val;
    Tmp sm((size_t)(presorted ? 0 : nnz) * 4), sn((size_t)(presorted ? 0 : nnz) * 4), sv((size_t)(presorted ? 0 : nnz) * 4);
    int32_t *order = nullptr;
    if (!presorted && nnz > 0) {
        Tmp k_in((size_t)nnz * 8), k_out((size_t)nnz * 8), i_in((size_t)nnz * 4);
        PD_CHECK(hipMalloc((void **)&order, (size_t)nnz * 4));
        int major_bits = 1, minor_bits = 1;
        while (((int64_t)1 << major_bits) < (int64_t)std::max(n_major, 1)) ++major_bits;
        while (((int64_t)1 << minor_bits) < (int64_t)std::max(n_minor, 1)) ++minor_bits;
        const int end_bit = minor_bits + major_bits;
        hipLaunchKernelGGL(make_keys_kernel, dim3(grid_for(nnz, threads)), dim3(threads), 0, st, nnz, d_major, d_minor,
                           minor_bits, k_in.as<uint64_t>(), i_in.as<int32_t>());
        size_t temp_bytes = 0;
        PD_CHECK(rocprim::radix_sort_pairs(nullptr, temp_bytes, k_in.as<uint64_t>(), k_out.as<uint64_t>(),
                                                    i_in.as<int32_t>(), order, (int)nnz, 0, end_bit, st));
        Tmp temp(temp_bytes);
        PD_CHECK(rocprim::radix_sort_pairs(temp.p, temp_bytes, k_in.as<uint64_t>(), k_out.as<uint64_t>(),
                                                    i_in.as<int32_t>(), order, (int)nnz, 0, end_bit, st));
        hipLaunchKernelGGL(split_keys_kernel, dim3(grid_for(nnz, threads)), dim3(threads), 0, st, nnz,
                           k_out.as<uint64_t>(), order, d_val, minor_bits, sm.as<int32_t>(), sn.as<int32_t>(),
                           sv.as<float>());
        PD_CHECK(hipGetLastError());
        PD_CHECK(hipStreamSynchronize(st));   // k_in / k_out / temp are released here
        s_major = sm.as<int32_t>(); s_minor = sn.as<int32_t>(); s_val = sv.as<float>();
    }
    try {
        // ---- 2. run pointers -> host
        Tmp d_mptr((size_t)(n_major + 1) * 8);
        hipLaunchKernelGGL(fill_i64_kernel, dim3(grid_for(n_major + 1, threads)), dim3(threads), 0, st,
                           (int64_t)n_major + 1, nnz > 0 ? nnz : 0, d_mptr.as<int64_t>());
        if (nnz > 0)
            hipLaunchKernelGGL(run_pointers_kernel, dim3(grid_for(nnz, threads)), dim3(threads), 0, st, nnz, s_major,
                               n_major, d_mptr.as<int64_t>());
        std::vector<int64_t> mptr((size_t)n_major + 1);
        PD_CHECK(hipMemcpyAsync(mptr.data(), d_mptr.p, mptr.size() * 8, hipMemcpyDeviceToHost, st));
        PD_CHECK(hipStreamSynchronize(st));

        // ---- 3. host: blocks, tasks
        tile_plan_begin(P, nnz, n_major, n_minor, shape, mptr.data());
        Geometry g{};
        g.gpb = P.gpb; g.gpw = P.gpw; g.wpb = P.wpb; g.W = P.n_windows; g.win_rows = P.win_rows; g.lpc = lpc;
        g.lpc_shift = 0;
        while ((1 << g.lpc_shift) < lpc) ++g.lpc_shift;
        g.n_classes = shape.bank_order ? std::max(1, 16 / std::max(1, lpc)) : 1;
        g.row_slots = P.row_slots;
        g.ring = P.ring; g.slot16 = P.slot16;
        g.look = P.look;
        const bool ring = P.ring > 1;
        g.single = P.single ? 1 : 0;
        const int64_t n_slots = P.n_blocks * P.gpb;
        Tmp d_rows((size_t)n_slots * 4);
        PD_CHECK(hipMemcpyAsync(d_rows.p, P.block_rows.data(), (size_t)n_slots * 4, hipMemcpyHostToDevice, st));

        // ---- 4. steps
        const size_t n_steps = P.steps.size();
        Tmp d_steps32(n_steps * 4 + 4);
        PD_CHECK(hipMemsetAsync(d_steps32.p, 0, n_steps * 4 + 4, st));
        int *d_err = d_steps32.as<int>() + n_steps;
        const int64_t n_segments = n_slots * P.n_windows;
        Tmp d_start(ring ? (size_t)n_slots * ((size_t)P.n_windows + 1) * 4 : 0);
        if (n_segments > 0 && !ring)
            hipLaunchKernelGGL(steps_kernel, dim3((unsigned)((n_segments + 255) / 256)), dim3(256), 0, st, g, n_slots,
                               d_rows.as<int32_t>(), d_mptr.as<int64_t>(), s_minor, d_steps32.as<unsigned>(), d_err);
        Tmp d_range_end((size_t)P.n_windows * 4);
        PD_CHECK(hipMemcpyAsync(d_range_end.p, P.range_end_of_window.data(), (size_t)P.n_windows * 4, hipMemcpyHostToDevice, st));
        if (n_segments > 0 && ring)
            hipLaunchKernelGGL(ring_schedule_kernel, dim3((unsigned)P.n_blocks), dim3((unsigned)((P.gpb + 63) / 64 * 64)), 0, st, g,
                               d_rows.as<int32_t>(), d_mptr.as<int64_t>(), s_minor, d_range_end.as<int32_t>(), d_steps32.as<unsigned>(),
                               d_start.as<int32_t>(), d_err);
        std::vector<uint32_t> steps32(n_steps + 1);
        PD_CHECK(hipMemcpyAsync(steps32.data(), d_steps32.p, (n_steps + 1) * 4, hipMemcpyDeviceToHost, st));
        PD_CHECK(hipStreamSynchronize(st));
        if (steps32[n_steps]) throw std::invalid_argument(P.single ? "a row has more than 65535 nonzeros in one window"
                                                 : "a row has more than 131070 nonzeros in one window");
        for (size_t i = 0; i < n_steps; ++i) P.steps[i] = (uint16_t)steps32[i];

        // ---- 5. host: offsets
        std::vector<int64_t> wave_off;
        const int64_t total_padded = tile_plan_offsets(P, wave_off);
        P.packed = shape.allow_packed && packed_ok;
        g.packed = P.packed ? 1 : 0;
        const int epw = P.packed ? 2 : 4;
        const std::vector<int> pass_rank = tile_pass_rank(lpc, P.gpw);

        // ---- 6. fill
        void *d_entries = nullptr, *d_steps16 = nullptr;
        const size_t entries_bytes = (size_t)total_padded * epw * 4;
        PD_CHECK(hipMalloc(&d_entries, entries_bytes ? entries_bytes : 16));
        *out_entries = d_entries;
        *out_entries_bytes = entries_bytes;
        PD_CHECK(hipMalloc(&d_steps16, n_steps * 2 + 16));
        *out_steps = d_steps16;
        PD_CHECK(hipMemsetAsync(d_entries, 0, entries_bytes, st));
        PD_CHECK(hipMemcpyAsync(d_steps16, P.steps.data(), n_steps * 2, hipMemcpyHostToDevice, st));
        // first step slot of every ((block, wave), window)
        std::vector<int64_t> win_off(n_steps);
        for (size_t bw = 0; bw + 1 < wave_off.size(); ++bw) {
            int64_t off = wave_off[bw];
            for (int w = 0; w < P.n_windows; ++w) {
                win_off[bw * P.n_windows + w] = off;
                off += tile_stored_steps(P, P.steps[bw * P.n_windows + w]) * P.gpw;
            }
        }
        Tmp d_woff(win_off.size() * 8), d_rank(pass_rank.size() * 4);
        PD_CHECK(hipMemcpyAsync(d_woff.p, win_off.data(), win_off.size() * 8, hipMemcpyHostToDevice, st));
        PD_CHECK(hipMemcpyAsync(d_rank.p, pass_rank.data(), pass_rank.size() * 4, hipMemcpyHostToDevice, st));
        if (ring && n_steps > 0)
            hipLaunchKernelGGL(ring_pad_kernel, dim3((unsigned)((n_steps + 255) / 256)), dim3(256), 0, st, g,
                               (int64_t)P.n_blocks * P.wpb, d_woff.as<int64_t>(), d_steps32.as<unsigned>(),
                               static_cast<uint32_t *>(d_entries));
        const bool joint = shape.bank_order == 2 && lpc <= 8;
        const std::vector<int> pass_of = tile_pass_of(lpc, P.gpw);
        Tmp d_pass(pass_of.size() * 4);
        PD_CHECK(hipMemcpyAsync(d_pass.p, pass_of.data(), pass_of.size() * 4, hipMemcpyHostToDevice, st));
        if (n_segments > 0 && joint) {
            // gpw lanes per (block, wave, window)
            const size_t upw = (size_t)(64 / P.gpw);
#define SCHPF_FILL_JOINT(NC)                                                                                        \
    hipLaunchKernelGGL(fill_joint_kernel<NC>, dim3((unsigned)((n_steps + upw - 1) / upw)), dim3(64), 0, st, g, (int64_t)n_steps, \
                       d_rows.as<int32_t>(), d_mptr.as<int64_t>(), s_minor, s_val, d_woff.as<int64_t>(),            \
                       d_rank.as<int>(), d_pass.as<int>(), d_start.as<int32_t>(), static_cast<uint32_t *>(d_entries))
            if (n_steps > 0x7fffffffull) throw std::invalid_argument("too many (wave, window) units for the joint fill");
            switch (g.n_classes) {
            case 16: SCHPF_FILL_JOINT(16); break;
            case 8: SCHPF_FILL_JOINT(8); break;
            case 4: SCHPF_FILL_JOINT(4); break;
            default: SCHPF_FILL_JOINT(2); break;
            }
#undef SCHPF_FILL_JOINT
        } else if (n_segments > 0)
            hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n_segments + 255) / 256)), dim3(256), 0, st, g, n_slots,
                               d_rows.as<int32_t>(), d_mptr.as<int64_t>(), s_minor, s_val, d_woff.as<int64_t>(),
                               d_rank.as<int>(), d_start.as<int32_t>(), static_cast<uint32_t *>(d_entries));
        PD_CHECK(hipGetLastError());
        PD_CHECK(hipStreamSynchronize(st));
        P.mptr.swap(mptr);
        *out_order = order;
        if (getenv("SCHPF_VERBOSE") && atoi(getenv("SCHPF_VERBOSE"))) tile_plan_report(P);
    } catch (...) {
        if (order) (void)hipFree(order);
        if (*out_entries) { (void)hipFree(*out_entries); *out_entries = nullptr; }
        if (*out_steps) { (void)hipFree(*out_steps); *out_steps = nullptr; }
        throw;
    }
}


// ------------------------------------------------------------------ balanced windows (plan.h)
namespace {

__global__ void count_major_kernel(int64_t n, const int32_t *__restrict__ major, int32_t *__restrict__ count)
{
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(count + major[j], 1);
}
// key = block << (minor_bits + GROUP_BITS) | minor << GROUP_BITS | lane group of the major row in its block
__global__ void balance_keys_kernel(int64_t n, const int32_t *__restrict__ major, const int32_t *__restrict__ minor,
                                    const int32_t *__restrict__ slot_of_row, int gpb, int minor_bits,
                                    uint64_t *__restrict__ key)
{
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const int32_t slot = slot_of_row[major[j]];
        const uint64_t b = (uint64_t)(slot / gpb), g = (uint64_t)(slot % gpb);
        key[j] = (b << (minor_bits + BALANCE_GROUP_BITS)) | ((uint64_t)(uint32_t)minor[j] << BALANCE_GROUP_BITS) | g;
    }
}
__global__ void virtual_minor_kernel(int64_t n, const int32_t *__restrict__ major, const int32_t *__restrict__ minor,
                                     const int32_t *__restrict__ slot_of_row, int gpb, int n_minor,
                                     const int32_t *__restrict__ virt, int32_t *__restrict__ vminor)
{
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = slot_of_row[major[j]] / gpb;
        vminor[j] = virt[b * n_minor + minor[j]];
    }
}

struct BalanceDev {
    int gpb, win_rows, W, nsec, D, n_minor, n_virtual, minor_bits;
};

__device__ __forceinline__ int64_t lower_bound_key(const uint64_t *__restrict__ keys, int64_t n, uint64_t key)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint64_t o = __shfl_xor(v, m, 64);
        v = o < v ? o : v;
    }
    return v;
}

// One wave per (block, section): plan.cpp balance_section with the candidate windows as lanes.  The minor
// rows are taken in order (the greedy is sequential); the loads of a minor row's lane groups are read from the
// LDS matrix eight at a time, every lane its own window's column.
__global__ __launch_bounds__(64) void balance_kernel(BalanceDev g, const uint64_t *__restrict__ keys, int64_t n_keys,
                                                     int32_t *__restrict__ virt, int32_t *__restrict__ minor_of)
{
    extern __shared__ uint16_t bal_lds[];
    uint16_t *load = bal_lds;                       // [gpb][D]
    uint16_t *glist = bal_lds + (size_t)g.gpb * g.D;   // the lane groups of the minor row at hand
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x / g.nsec;
    const int s = blockIdx.x % g.nsec;
    const int w0 = s * g.D, w1 = min(g.W, w0 + g.D), Dn = w1 - w0;
    const int m0 = w0 * g.win_rows;
    const int m1 = (int)min((int64_t)g.n_minor, (int64_t)w1 * g.win_rows);
    for (int i = lane; i < g.gpb * g.D; i += 64) load[i] = 0;
    const int shift = g.minor_bits + BALANCE_GROUP_BITS;
    const int64_t lo = lower_bound_key(keys, n_keys, ((uint64_t)b << shift) | ((uint64_t)(uint32_t)m0 << BALANCE_GROUP_BITS));
    const int64_t hi = lower_bound_key(keys, n_keys, ((uint64_t)b << shift) | ((uint64_t)(uint32_t)m1 << BALANCE_GROUP_BITS));
    const uint32_t mmask = (uint32_t)(((uint64_t)1 << g.minor_bits) - 1);
    const uint32_t gmask = (1u << BALANCE_GROUP_BITS) - 1u;
    int my_cnt = 0;                                  // rows dealt to window w0 + lane so far
    int32_t *virt_b = virt + b * g.n_minor;
    int32_t *mo_b = minor_of + b * (int64_t)g.n_virtual;
    __syncthreads();
    int64_t pos = lo;
    while (pos < hi) {
        unsigned mx = 0, sm = 0;
        int ng = 0;
        int cur = -1;
        for (;;) {                                   // the run of one minor row, 64 keys at a time
            const bool in = pos + lane < hi;
            const uint64_t k = in ? keys[pos + lane] : ~(uint64_t)0;
            const uint32_t klo = (uint32_t)(k >> BALANCE_GROUP_BITS) & mmask;
            if (cur < 0) cur = __builtin_amdgcn_readfirstlane((int)klo);
            const uint64_t same = __ballot(in && (int)klo == cur);
            const int run = same == ~(uint64_t)0 ? 64 : __builtin_ctzll(~same);
            if (run == 0) break;
            // a COO may hold an entry twice: equal keys are neighbours, only the first counts
            const bool dup = in && pos + lane > lo && keys[pos + lane - 1] == k;
            const uint64_t fresh = __ballot(lane < run && !dup);
            const uint32_t grp = (uint32_t)k & gmask;
            const int at = ng + __popcll(fresh & (((uint64_t)1 << lane) - 1));
            if (((fresh >> lane) & 1) && at < g.gpb) glist[at] = (uint16_t)grp;
            for (int i = 0; i < run; i += 8) {
                unsigned v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int src = min(i + u, run - 1);
                    const uint32_t gi = (uint32_t)__builtin_amdgcn_readlane((int)grp, src);
                    const bool on = i + u < run && ((fresh >> src) & 1);
                    v[u] = (lane < Dn && on) ? (unsigned)load[(size_t)gi * g.D + lane] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { mx = max(mx, v[u]); sm += v[u]; }
            }
            ng += __popcll(fresh);
            pos += run;
            if (run < 64 || pos >= hi) break;
        }
        const uint64_t cost = (lane < Dn && my_cnt < g.win_rows) ? balance_cost(mx, sm, (unsigned)lane) : ~(uint64_t)0;
        const int c = (int)(wave_min_u64(cost) & 0xff);
        __syncthreads();                             // glist is written
        for (int j = lane; j < min(ng, g.gpb); j += 64) {
            uint16_t &v = load[(size_t)glist[j] * g.D + c];
            if (v < 65535) ++v;
        }
        const int p = __builtin_amdgcn_readlane(my_cnt, c);
        if (lane == c) ++my_cnt;
        if (lane == 0) {
            const int32_t v = (w0 + c) * g.win_rows + p;
            virt_b[cur] = v;
            mo_b[v] = cur;
        }
        __syncthreads();                             // the loads are up to date before the next row reads them
    }
    // the minor rows no row of the block holds fill the capacity left, window by window
    __threadfence_block();
    const int rem = lane < Dn ? g.win_rows - my_cnt : 0;
    int pre = rem;                                   // inclusive prefix over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(pre, d, 64);
        if (lane >= d) pre += o;
    }
    pre -= rem;                                      // exclusive
    int ranked = 0;
    for (int m = m0; m < m1; m += 64) {
        const bool empty = m + lane < m1 && virt_b[m + lane] < 0;
        const uint64_t bal = __ballot(empty);
        const int rank = ranked + __popcll(bal & (((uint64_t)1 << lane) - 1));
        int sel = -1, p = 0;
        for (int c = 0; c < Dn; ++c) {
            const int pc = __builtin_amdgcn_readlane(pre, c), rc = __builtin_amdgcn_readlane(rem, c);
            const int cc = __builtin_amdgcn_readlane(my_cnt, c);
            if (rank >= pc && rank < pc + rc) { sel = c; p = cc + rank - pc; }
        }
        if (empty && sel >= 0) {
            const int32_t v = (w0 + sel) * g.win_rows + p;
            virt_b[m + lane] = v;
            mo_b[v] = m + lane;
        }
        ranked += __popcll(bal);
    }
}

}  // namespace

void balance_windows_device(void *stream, int64_t nnz, const int32_t *d_major, const int32_t *d_minor, int n_major,
                            int n_minor, const TileShape &shape, int32_t *d_vminor, void **d_minor_of,
                            BalanceGeometry &geo)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    *d_minor_of = nullptr;
    if (shape.ring > 1) throw std::invalid_argument("balanced windows need whole windows (ring <= 1)");
    const int threads = 256;
    // row lengths -> the blocks the builder will cut (tile_plan_begin depends on the lengths only)
    Tmp d_count((size_t)n_major * 4);
    PD_CHECK(hipMemsetAsync(d_count.p, 0, (size_t)n_major * 4, st));
    if (nnz > 0)
        hipLaunchKernelGGL(count_major_kernel, dim3(grid_for(nnz, threads)), dim3(threads), 0, st, nnz, d_major,
                           d_count.as<int32_t>());
    std::vector<int32_t> count((size_t)n_major);
    PD_CHECK(hipMemcpyAsync(count.data(), d_count.p, (size_t)n_major * 4, hipMemcpyDeviceToHost, st));
    PD_CHECK(hipStreamSynchronize(st));
    std::vector<int64_t> mptr((size_t)n_major + 1, 0);
    for (int m = 0; m < n_major; ++m) mptr[(size_t)m + 1] = mptr[(size_t)m] + count[(size_t)m];
    TilePlanHost T;
    tile_plan_begin(T, nnz, n_major, n_minor, shape, mptr.data());
    const int gpb = T.gpb, W = T.n_windows, win_rows = T.win_rows;
    if (gpb > (1 << BALANCE_GROUP_BITS)) throw std::invalid_argument("balanced windows: more than 1024 rows per block");
    if ((int64_t)W * win_rows > 0x7fffffff) throw std::invalid_argument("balanced windows: virtual index overflow");
    geo = BalanceGeometry();
    geo.gpb = gpb; geo.win_rows = win_rows; geo.n_windows = W; geo.n_blocks = T.n_blocks;
    balance_sections(W, gpb, geo.n_sections, geo.D);
    geo.n_virtual = W * win_rows;
    std::vector<int32_t> slot_of_row((size_t)n_major, 0);
    for (int64_t i = 0; i < T.n_blocks * gpb; ++i)
        if (T.block_rows[(size_t)i] >= 0) slot_of_row[(size_t)T.block_rows[(size_t)i]] = (int32_t)i;
    Tmp d_slot((size_t)n_major * 4);
    PD_CHECK(hipMemcpyAsync(d_slot.p, slot_of_row.data(), (size_t)n_major * 4, hipMemcpyHostToDevice, st));

    int minor_bits = 1, block_bits = 1;
    while (((int64_t)1 << minor_bits) < (int64_t)std::max(n_minor, 1)) ++minor_bits;
    while (((int64_t)1 << block_bits) < std::max<int64_t>(T.n_blocks, 1)) ++block_bits;
    const int end_bit = BALANCE_GROUP_BITS + minor_bits + block_bits;
    if (end_bit > 64) throw std::invalid_argument("balanced windows: key overflow");
    Tmp virt((size_t)T.n_blocks * n_minor * 4);
    void *mo = nullptr;
    PD_CHECK(hipMalloc(&mo, (size_t)T.n_blocks * geo.n_virtual * 4 + 64));   // + slack: a list is copied in 16-byte pieces
    try {
        PD_CHECK(hipMemsetAsync(virt.p, 0xFF, (size_t)T.n_blocks * n_minor * 4, st));
        PD_CHECK(hipMemsetAsync(mo, 0xFF, (size_t)T.n_blocks * geo.n_virtual * 4 + 64, st));
        Tmp k_in((size_t)nnz * 8), k_out((size_t)nnz * 8);
        if (nnz > 0) {
            hipLaunchKernelGGL(balance_keys_kernel, dim3(grid_for(nnz, threads)), dim3(threads), 0, st, nnz, d_major, d_minor,
                               d_slot.as<int32_t>(), gpb, minor_bits, k_in.as<uint64_t>());
            size_t temp_bytes = 0;
            PD_CHECK(rocprim::radix_sort_keys(nullptr, temp_bytes, k_in.as<uint64_t>(), k_out.as<uint64_t>(), (int)nnz,
                                                       0, end_bit, st));
            Tmp temp(temp_bytes);
            PD_CHECK(rocprim::radix_sort_keys(temp.p, temp_bytes, k_in.as<uint64_t>(), k_out.as<uint64_t>(), (int)nnz,
                                                       0, end_bit, st));
            PD_CHECK(hipStreamSynchronize(st));   // temp dies here
        }
        const bool verbose = getenv("SCHPF_VERBOSE") && atoi(getenv("SCHPF_VERBOSE"));
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_sorted = now();
        BalanceDev bd{gpb, win_rows, W, geo.n_sections, geo.D, n_minor, geo.n_virtual, minor_bits};
        const size_t lds = ((size_t)gpb * geo.D + (size_t)gpb) * 2;
        const int64_t units = T.n_blocks * geo.n_sections;
        if (units > 0x7fffffff) throw std::invalid_argument("balanced windows: too many (block, section) units");
        if (units > 0)
            hipLaunchKernelGGL(balance_kernel, dim3((unsigned)units), dim3(64), lds, st, bd, k_out.as<uint64_t>(), nnz,
                               virt.as<int32_t>(), static_cast<int32_t *>(mo));
        if (verbose) {
            PD_CHECK(hipStreamSynchronize(st));
            fprintf(stderr, "[schpf_hip]     balance: %lld (block, section) units dealt in %.4f s (keys sorted before that)\n",
                    (long long)units, now() - t_sorted);
        }
        if (nnz > 0)
            hipLaunchKernelGGL(virtual_minor_kernel, dim3(grid_for(nnz, threads)), dim3(threads), 0, st, nnz, d_major, d_minor,
                               d_slot.as<int32_t>(), gpb, n_minor, virt.as<int32_t>(), d_vminor);
        PD_CHECK(hipGetLastError());
        PD_CHECK(hipStreamSynchronize(st));
    } catch (...) {
        (void)hipFree(mo);
        throw;
    }
    *d_minor_of = mo;
}

}  // namespace schpf
