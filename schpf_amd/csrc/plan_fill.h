// Sliced-ELL fill of one (block, wave, window, LDS pass) of a tile plan -- ONE implementation,
// compiled for the host builder (plan.cpp) and for the device builder (plan_device.hip), so the
// two produce the same entries bit for bit.
//
// A ds_read_b128 wave instruction is served in passes of 16 lanes; the 16 / lpc lane groups of a
// pass each read a different minor row, and two rows collide when their 16-byte slots fall on the
// same LDS banks ("class" of a row: bank_class()).  Inside a window segment the nonzeros of a row
// may be taken in any order, so for every step position t the groups of a pass pick nonzeros of
// pairwise DIFFERENT classes whenever they can: groups with the fewest classes left choose first,
// each takes its fullest class that nobody took at this position (else its fullest class: a
// conflict), inside a class in minor order.  Ordering every row on its own (a fixed rotation of
// wished classes) left a third of the LDS cycles as conflicts; with the bank order switched off
// the sweep is 9 % (f64) / 16 % (f32) slower.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SCHPF_HD __host__ __device__
#else
#define SCHPF_HD
#endif

namespace schpf {

struct FillGeometry {
    int gpw, win_rows, lpc_shift, n_classes, row_slots, packed;
};

constexpr int kMaxPassGroups = 16;   // lpc = 1: sixteen groups per pass
constexpr int kJointLimit = 192;     // longest segment of a pass up to which the groups choose jointly

SCHPF_HD inline int fill_bank_class(const FillGeometry &g, int32_t local)
{
    return (int)(((((unsigned)local * (unsigned)g.row_slots) & 15u) >> g.lpc_shift) & (unsigned)(g.n_classes - 1));
}

// first position in [lo, hi) whose minor index is >= key (a row's minors ascend)
SCHPF_HD inline int64_t fill_lower_bound(const int32_t *s_minor, int64_t lo, int64_t hi, int64_t key)
{
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)s_minor[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// nonzero number t of a row segment -> its half of step slot `step_slot`
SCHPF_HD inline void fill_store(const FillGeometry &g, uint32_t *entries, uint64_t step_slot, int t, int32_t local, float val)
{
    if (g.packed) {
        uint32_t *e = entries + step_slot * 2;
        const int sh = (t & 1) * 16;
        e[0] |= (uint32_t)local << sh;
        e[1] |= (uint32_t)val << sh;
    } else {
        uint32_t *e = entries + step_slot * 4 + (uint64_t)(t & 1) * 2;
        e[0] = (uint32_t)local;
        union { float f; uint32_t u; } cv;
        cv.f = val;
        e[1] = cv.u;
    }
}

// members: m < n_members groups of one pass; slot[m] = group index inside the wave, seg_lo[m] /
// seg_n[m] = its row's nonzeros of this window in the sorted arrays.  base = first minor row of
// the window, win_off = first step slot of the window's entries of this (block, wave).
SCHPF_HD inline void fill_pass(const FillGeometry &g, int n_members, const int *slot, const int64_t *seg_lo,
                               const int *seg_n, int32_t base, int64_t win_off, const int32_t *s_minor,
                               const float *s_val, uint32_t *entries)
{
    int cnt[kMaxPassGroups][16];
    int64_t cur[kMaxPassGroups][16];
    int max_n = 0;
    for (int m = 0; m < n_members; ++m) {
        for (int c = 0; c < 16; ++c) { cnt[m][c] = 0; cur[m][c] = seg_lo[m]; }
        for (int64_t j = seg_lo[m]; j < seg_lo[m] + seg_n[m]; ++j) cnt[m][fill_bank_class(g, s_minor[j] - base)]++;
        if (seg_n[m] > max_n) max_n = seg_n[m];
    }
    if (max_n > kJointLimit) {
        // long segments (a gene expressed in half of the cells of a window): every member on its own,
        // wished class (member + t) mod classes, else its fullest -- a fraction of the bookkeeping;
        // the joint choice below costs O(members^2 + members * classes) per position
        for (int m = 0; m < n_members; ++m)
            for (int t = 0; t < seg_n[m]; ++t) {
                int c = (int)(((unsigned)m + (unsigned)t) & (unsigned)(g.n_classes - 1));
                if (cnt[m][c] == 0) {
                    int best = 0;
                    for (int k = 0; k < g.n_classes; ++k)
                        if (cnt[m][k] > best) { best = cnt[m][k]; c = k; }
                }
                int64_t q = cur[m][c];
                while (fill_bank_class(g, s_minor[q] - base) != c) ++q;
                cur[m][c] = q + 1;
                cnt[m][c]--;
                fill_store(g, entries, (uint64_t)win_off + (uint64_t)(t >> 1) * g.gpw + slot[m], t, s_minor[q] - base, s_val[q]);
            }
        return;
    }
    for (int t = 0; t < max_n; ++t) {
        // members still active at this position, fewest remaining classes first (ties: lower slot)
        int order[kMaxPassGroups], key[kMaxPassGroups], n_act = 0;
        for (int m = 0; m < n_members; ++m) {
            if (seg_n[m] <= t) continue;
            int k = 0;
            for (int c = 0; c < g.n_classes; ++c) k += cnt[m][c] > 0;
            int i = n_act++;
            while (i > 0 && key[i - 1] > k) { key[i] = key[i - 1]; order[i] = order[i - 1]; --i; }   // stable insertion
            key[i] = k;
            order[i] = m;
        }
        unsigned taken = 0;
        for (int a = 0; a < n_act; ++a) {
            const int m = order[a];
            int c = -1, best = 0;
            for (int k = 0; k < g.n_classes; ++k)
                if (cnt[m][k] > best && !((taken >> k) & 1u)) { best = cnt[m][k]; c = k; }
            if (c < 0)
                for (int k = 0; k < g.n_classes; ++k)
                    if (cnt[m][k] > best) { best = cnt[m][k]; c = k; }
            taken |= 1u << c;
            int64_t q = cur[m][c];
            while (fill_bank_class(g, s_minor[q] - base) != c) ++q;   // next unused nonzero of the class, minor order
            cur[m][c] = q + 1;
            cnt[m][c]--;
            fill_store(g, entries, (uint64_t)win_off + (uint64_t)(t >> 1) * g.gpw + slot[m], t, s_minor[q] - base, s_val[q]);
        }
    }
}

// LDS pass (0..3) of lane group `slot` of a wave (the lanes one ds_read_b128 cycle serves)
SCHPF_HD inline int fill_pass_of_slot(int slot, int lpc)
{
    const int pass_of_quad[16] = {0, 1, 1, 0, 1, 0, 0, 1, 2, 3, 3, 2, 3, 2, 2, 3};   // lanes 4q..4q+3
    if (lpc > 16) return 0;
    return pass_of_quad[(slot * lpc) / 4];
}

}  // namespace schpf
