// Host-side construction of the sweep plans (see plan.h for the layout).
#include "plan.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <stdexcept>
#include <thread>

namespace schpf {

// ---- host threading: plan construction is O(nnz) passes with random access; at the headline
// size (1e8 nonzeros) it would otherwise dominate a whole fit ----
int host_threads()
{
    const char *s = getenv("SCHPF_HOST_THREADS");
    if (s && *s) return std::max(1, atoi(s));
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::min<unsigned>(hw ? hw : 1, 32);
}

// f(begin, end, thread_index) over [0, n) in contiguous slabs
template <typename F> static void parallel_for(int64_t n, int nth, F f)
{
    if (n <= 0) return;
    nth = (int)std::max<int64_t>(1, std::min<int64_t>(nth, n));
    if (nth == 1) { f((int64_t)0, n, 0); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)nth);
    for (int t = 0; t < nth; ++t) {
        const int64_t b = n * t / nth, e = n * (t + 1) / nth;
        th.emplace_back([=, &f] { f(b, e, t); });
    }
    for (auto &x : th) x.join();
}

// stable parallel counting sort of the sequence seq[0..n) (or 0..n-1 when seq == nullptr) by
// key[seq[j]]: out[rank] = seq[j]; ptr = run pointers per key
static void counting_sort_seq(int64_t n, const int32_t *seq, const int32_t *key, int nkeys,
                              BigVec<int32_t> &out, std::vector<int64_t> &ptr)
{
    int nth = host_threads();
    while (nth > 1 && (int64_t)nth * nkeys > (int64_t)48 << 20) nth /= 2;   // bound the counter table
    if (n < (1 << 16)) nth = 1;
    std::vector<uint32_t> counts((size_t)nth * nkeys, 0u);
    parallel_for(n, nth, [&](int64_t b, int64_t e, int t) {
        uint32_t *c = counts.data() + (size_t)t * nkeys;
        for (int64_t j = b; j < e; ++j) c[key[seq ? seq[j] : (int32_t)j]]++;
    });
    ptr.assign((size_t)nkeys + 1, 0);
    std::vector<int64_t> start((size_t)nth * nkeys);
    int64_t run = 0;
    for (int k = 0; k < nkeys; ++k) {
        ptr[(size_t)k] = run;
        for (int t = 0; t < nth; ++t) {
            start[(size_t)t * nkeys + k] = run;
            run += counts[(size_t)t * nkeys + k];
        }
    }
    ptr[(size_t)nkeys] = run;
    out.resize((size_t)n);
    parallel_for(n, nth, [&](int64_t b, int64_t e, int t) {
        int64_t *s = start.data() + (size_t)t * nkeys;
        for (int64_t j = b; j < e; ++j) {
            const int32_t pos = seq ? seq[j] : (int32_t)j;
            out[(size_t)s[key[pos]]++] = pos;
        }
    });
}

void counting_sort_positions(int64_t n, const int32_t *key, int nkeys, BigVec<int32_t> &order,
                             std::vector<int64_t> &ptr)
{
    counting_sort_seq(n, nullptr, key, nkeys, order, ptr);
}

void coo_order_flags(int64_t nnz, const int32_t *major, const int32_t *minor, bool &sorted_mm, bool &sorted_nm)
{
    const int nth = host_threads();
    std::vector<char> not_mm((size_t)nth + 1, 0), not_nm((size_t)nth + 1, 0);
    parallel_for(nnz, nth, [&](int64_t b, int64_t e, int t) {
        bool bad_mm = false, bad_nm = false;
        for (int64_t j = std::max<int64_t>(b, 1); j < e && !(bad_mm && bad_nm); ++j) {
            const int32_t M0 = major[j - 1], M1 = major[j], m0 = minor[j - 1], m1 = minor[j];
            bad_mm |= M1 < M0 || (M1 == M0 && m1 < m0);
            bad_nm |= m1 < m0 || (m1 == m0 && M1 < M0);
        }
        not_mm[(size_t)t] = bad_mm;
        not_nm[(size_t)t] = bad_nm;
    });
    sorted_mm = sorted_nm = true;
    for (int t = 0; t <= nth; ++t) { sorted_mm = sorted_mm && !not_mm[(size_t)t]; sorted_nm = sorted_nm && !not_nm[(size_t)t]; }
}

void sort_by_major_minor(int64_t nnz, const int32_t *major, const int32_t *minor, int n_major, int n_minor,
                         BigVec<int32_t> &order, std::vector<int64_t> &mptr)
{
    // What order did the caller's COO come in?  SciPy's canonical format (sum_duplicates, tocoo of
    // a CSR) is sorted by (row, col): the cell-side plan then needs no sort at all and the
    // gene-side plan one stable pass.  One parallel scan decides.
    const int nth = host_threads();
    bool sorted_mm = true, sorted_nm = true;
    coo_order_flags(nnz, major, minor, sorted_mm, sorted_nm);
    if (sorted_mm) {
        // identity order; run pointers from the positions where the major index changes
        order.resize((size_t)nnz);
        mptr.assign((size_t)n_major + 1, nnz);
        for (int32_t k = 0; nnz > 0 && k <= major[0]; ++k) mptr[(size_t)k] = 0;
        if (nnz == 0) std::fill(mptr.begin(), mptr.end(), 0);
        parallel_for(nnz, nth, [&](int64_t b, int64_t e, int) {
            for (int64_t j = b; j < e; ++j) {
                order[(size_t)j] = (int32_t)j;
                if (j > 0 && major[j] != major[j - 1])
                    for (int32_t k = major[j - 1] + 1; k <= major[j]; ++k) mptr[(size_t)k] = j;
            }
        });
        return;
    }
    if (sorted_nm) {   // already grouped by minor with ascending major inside: one stable pass by major
        counting_sort_seq(nnz, nullptr, major, n_major, order, mptr);
        return;
    }
    // general case: minor first, then stable by major
    BigVec<int32_t> by_minor;
    std::vector<int64_t> tmp_ptr;
    counting_sort_seq(nnz, nullptr, minor, n_minor, by_minor, tmp_ptr);
    counting_sort_seq(nnz, by_minor.data(), major, n_major, order, mptr);
}

namespace {

struct Chunk {
    int32_t major;
    int32_t len;
    int32_t window;
    int32_t natid;
    int64_t start;  // in (major, minor)-sorted order
};

inline uint32_t f2u(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

}  // namespace

void build_sweep_plan(int64_t nnz, const int32_t *major, const int32_t *minor, const float *val,
                      int n_major, int n_minor, int lpc, int chunk_len, int n_windows,
                      bool keep_order, SweepPlanHost &P)
{
    if (lpc < 1 || lpc > 64 || (64 % lpc) != 0) throw std::invalid_argument("lpc must divide 64");
    if (chunk_len < 2 || (chunk_len & 1)) throw std::invalid_argument("chunk_len must be even, >= 2");
    if (n_windows < 1) n_windows = 1;
    P = SweepPlanHost();
    P.n_major = n_major;
    P.n_minor = n_minor;
    P.lpc = lpc;
    P.cpw = 64 / lpc;
    P.chunk_len = chunk_len;
    P.n_windows = n_windows;
    P.nnz = nnz;

    BigVec<int32_t> order;
    std::vector<int64_t> mptr;
    sort_by_major_minor(nnz, major, minor, n_major, n_minor, order, mptr);

    // ---- windows over the minor index (equal width) ----
    const int64_t wwidth = ((int64_t)n_minor + n_windows - 1) / n_windows;

    // ---- cut every major's run into chunks ----
    std::vector<Chunk> chunks;
    chunks.reserve((size_t)(nnz / chunk_len + n_major + 16));
    P.cptr.assign((size_t)n_major + 1, 0);
    for (int m = 0; m < n_major; ++m) {
        P.cptr[m] = (int32_t)chunks.size();
        int64_t j = mptr[m], end = mptr[(size_t)m + 1];
        while (j < end) {
            int32_t w = (int32_t)(minor[order[(size_t)j]] / wwidth);
            int64_t s = j;
            while (j < end && (j - s) < chunk_len && (int32_t)(minor[order[(size_t)j]] / wwidth) == w) ++j;
            Chunk c;
            c.major = m;
            c.len = (int32_t)(j - s);
            c.window = w;
            c.natid = (int32_t)chunks.size();
            c.start = s;
            chunks.push_back(c);
        }
    }
    P.cptr[n_major] = (int32_t)chunks.size();
    P.n_chunks = (int64_t)chunks.size();

    // ---- order chunks by (window, length descending); stable so ties keep major order ----
    std::vector<int32_t> corder(chunks.size());
    std::iota(corder.begin(), corder.end(), 0);
    std::stable_sort(corder.begin(), corder.end(), [&](int32_t a, int32_t b) {
        if (chunks[a].window != chunks[b].window) return chunks[a].window < chunks[b].window;
        return chunks[a].len > chunks[b].len;
    });

    // ---- slices: cpw chunks each, never straddling a window ----
    const int cpw = P.cpw;
    std::vector<int32_t> slice_window;
    {
        size_t i = 0;
        while (i < corder.size()) {
            int32_t w = chunks[corder[i]].window;
            size_t e = i;
            while (e < corder.size() && (e - i) < (size_t)cpw && chunks[corder[e]].window == w) ++e;
            int32_t steps = (chunks[corder[i]].len + 1) / 2;  // widest chunk first in the slice
            P.slice_steps.push_back(steps);
            slice_window.push_back(w);
            for (size_t s = 0; s < (size_t)cpw; ++s) {
                if (i + s < e) {
                    P.chunk_major.push_back(chunks[corder[i + s]].major);
                    P.chunk_natid.push_back(chunks[corder[i + s]].natid);
                } else {
                    P.chunk_major.push_back(-1);
                    P.chunk_natid.push_back(-1);
                }
            }
            i = e;
        }
    }
    P.n_slices = (int64_t)P.slice_steps.size();
    P.slice_off.resize((size_t)P.n_slices);
    int64_t total = 0;
    for (int64_t s = 0; s < P.n_slices; ++s) {
        P.slice_off[(size_t)s] = total;
        total += (int64_t)P.slice_steps[(size_t)s] * cpw;
    }

    // ---- fill the sliced-ELL entries ----
    P.entries.assign((size_t)total * 4, 0u);
    {
        size_t ci = 0;  // index into corder, advancing with the slices
        for (int64_t s = 0; s < P.n_slices; ++s) {
            uint32_t *base = P.entries.data() + (size_t)P.slice_off[(size_t)s] * 4;
            for (int slot = 0; slot < cpw; ++slot) {
                if (P.chunk_major[(size_t)s * cpw + slot] < 0) continue;
                const Chunk &c = chunks[corder[ci++]];
                for (int32_t t = 0; t < c.len; ++t) {
                    int32_t pos = order[(size_t)(c.start + t)];
                    uint32_t *e = base + ((size_t)(t >> 1) * cpw + slot) * 4 + (size_t)(t & 1) * 2;
                    e[0] = (uint32_t)minor[pos];
                    e[1] = f2u(val[pos]);
                }
            }
        }
    }

    // ---- wave -> slice map, XCD-aware ----
    // Workgroup b (4 waves) is observed to run on XCD b % 8.  Window w is served by the
    // XCDs x with x % g == w % g, g = min(n_windows, 8) (n_windows is 1, 2, 4 or a
    // multiple of 8), so that each XCD's L2 only ever holds its own windows' table rows.
    {
        const int g = std::min(n_windows, 8);
        std::vector<std::vector<int32_t>> per_xcd(8);
        std::vector<int> rr(g, 0);  // round-robin cursor per residue class
        const int xcds_per_class = 8 / g;
        // deal whole workgroups (4 consecutive slices of one residue class) to an XCD
        std::vector<std::vector<int32_t>> by_class(g);
        for (int64_t s = 0; s < P.n_slices; ++s) by_class[slice_window[(size_t)s] % g].push_back((int32_t)s);
        for (int r = 0; r < g; ++r) {
            const std::vector<int32_t> &L = by_class[r];
            for (size_t i = 0; i < L.size(); i += 4) {
                int x = r + g * (rr[r] % xcds_per_class);
                rr[r]++;
                for (size_t q = 0; q < 4; ++q)
                    per_xcd[x].push_back(i + q < L.size() ? L[i + q] : -1);
            }
        }
        size_t maxwg = 0;
        for (int x = 0; x < 8; ++x) maxwg = std::max(maxwg, per_xcd[x].size() / 4);
        P.n_waves = (int64_t)maxwg * 8 * 4;
        P.wave_slice.assign((size_t)P.n_waves, -1);
        for (int x = 0; x < 8; ++x)
            for (size_t i = 0; i < per_xcd[x].size(); ++i) {
                size_t wg = (i / 4) * 8 + (size_t)x;
                P.wave_slice[wg * 4 + (i % 4)] = per_xcd[x][i];
            }
    }

    if (keep_order) {
        P.order.swap(order);
        P.mptr.swap(mptr);
    }
}


// LDS-bank-aware order of one row segment (tile plan).  A lane group reads the gathered row with
// ds_read_b128; the LDS serves 16 lanes (= 16/lpc lane groups, one "pass") per cycle and those
// reads are conflict-free iff their 16-byte slots differ modulo 16.  A row starts at LDS position
// off16 (16-byte units, tile_off16), so its class is (off16 mod 16) / lpc.  The group with rank j
// inside its pass wants class (j + t) mod n_classes at position t: if every group of a pass gets
// its wish, the pass touches each slot once.  Greedy: take the wished class if the segment still
// has such a nonzero, else from the fullest class.
// seq[t] = index (within the segment) of the nonzero placed at position t.
static void bank_order(const TilePlanHost &P, const int32_t *seg_minor, int n, bool enabled, int rank,
                       std::vector<int32_t> &seq, std::vector<int32_t> &scratch)
{
    const int lpc = P.lpc;
    const int n_classes = enabled ? std::max(1, 16 / std::max(1, lpc)) : 1;   // a power of two (lpc is)
    if (n_classes == 1 || n <= 2) {
        for (int t = 0; t < n; ++t) seq[(size_t)t] = t;
        return;
    }
    // stable counting sort of the segment's positions by class; every class is then handed out
    // front to back, i.e. in minor order
    int lpc_shift = 0;
    while ((1 << lpc_shift) < lpc) ++lpc_shift;
    const unsigned cmask = (unsigned)n_classes - 1u;
    int cnt[16] = {0}, head[16];
    scratch.resize((size_t)n * 2);
    int32_t *cls = scratch.data(), *pos = scratch.data() + n;
    for (int i = 0; i < n; ++i) {
        const unsigned c = ((tile_off16(P, seg_minor[i]) & 15u) >> lpc_shift) & cmask;
        cls[i] = (int32_t)c;
        cnt[c]++;
    }
    int run = 0;
    for (int c = 0; c < n_classes; ++c) { head[c] = run; run += cnt[c]; }
    {
        int cur[16];
        for (int c = 0; c < n_classes; ++c) cur[c] = head[c];
        for (int i = 0; i < n; ++i) pos[cur[cls[i]]++] = i;
    }
    for (int t = 0; t < n; ++t) {
        int c = (int)(((unsigned)rank + (unsigned)t) & cmask);
        if (cnt[c] == 0) {
            int best = 0;
            for (int k = 0; k < n_classes; ++k)
                if (cnt[k] > best) { best = cnt[k]; c = k; }
        }
        seq[(size_t)t] = pos[head[c]++];
        cnt[c]--;
    }
}

// Joint variant (TileShape::bank_order = 2).  The per-row rule above is blind to what the other
// groups of the pass hold: a group whose wished class is empty falls back to its fullest class,
// which another group of the pass may be reading at that very position (round-3 counters: 43 % of
// the LDS cycles of the C3 sweep are such conflicts).  Here the groups of one wave and one window
// are dealt together: at position t the groups of a pass pick in turn (the group of rank
// (a + t) mod M is a-th, so nobody is always last), each the fullest class its pass has not read
// yet at t (tile_joint_pick).  Segments longer than TILE_JOINT_MAX keep the per-row rule and
// stand outside the bookkeeping (the device builder sorts a segment in LDS).
// seg_minor[m] / seg_n[m]: the segment of lane group m of the wave; seq[m][t] as in bank_order.
struct JointScratch {
    std::vector<int32_t> pos, row_scratch;
    std::vector<uint16_t> cnt, head;
};
static void bank_order_joint(const TilePlanHost &P, int gpw, const int32_t *const *seg_minor, const int *seg_n,
                             const std::vector<int> &pass_of, const std::vector<int> &pass_rank,
                             std::vector<std::vector<int32_t>> &seq, JointScratch &S)
{
    const int lpc = P.lpc;
    const int C = std::max(1, 16 / std::max(1, lpc));
    int lpc_shift = 0;
    while ((1 << lpc_shift) < lpc) ++lpc_shift;
    const unsigned cmask = (unsigned)C - 1u;
    S.pos.resize((size_t)gpw * TILE_JOINT_MAX);
    S.cnt.assign((size_t)gpw * 16, 0);
    S.head.assign((size_t)gpw * 16, 0);
    int lane_of[4][16];
    for (int p = 0; p < 4; ++p)
        for (int r = 0; r < 16; ++r) lane_of[p][r] = -1;
    int max_n = 0;
    for (int m = 0; m < gpw; ++m) {
        const int n = seg_n[m];
        seq[(size_t)m].resize((size_t)n);
        if (C > 1) lane_of[pass_of[(size_t)m]][pass_rank[(size_t)m]] = m;
        if (n == 0) continue;
        if (C == 1 || n > TILE_JOINT_MAX) {
            bank_order(P, seg_minor[m], n, C > 1, pass_rank[(size_t)m], seq[(size_t)m], S.row_scratch);
            continue;
        }
        max_n = std::max(max_n, n);
        uint16_t *cnt = S.cnt.data() + (size_t)m * 16, *head = S.head.data() + (size_t)m * 16;
        int32_t *pos = S.pos.data() + (size_t)m * TILE_JOINT_MAX;
        for (int i = 0; i < n; ++i) cnt[((tile_off16(P, seg_minor[m][i]) & 15u) >> lpc_shift) & cmask]++;
        int run = 0, cur[16];
        for (int c = 0; c < C; ++c) { head[c] = (uint16_t)run; cur[c] = run; run += cnt[c]; }
        for (int i = 0; i < n; ++i) pos[cur[((tile_off16(P, seg_minor[m][i]) & 15u) >> lpc_shift) & cmask]++] = i;
    }
    for (int t = 0; t < max_n; ++t)
        for (int p = 0; p < 4; ++p) {
            unsigned taken = 0;
            for (int a = 0; a < C; ++a) {
                const int m = lane_of[p][(a + t) & (int)cmask];
                if (m < 0 || seg_n[m] > TILE_JOINT_MAX || t >= seg_n[m]) continue;
                uint16_t *cnt = S.cnt.data() + (size_t)m * 16, *head = S.head.data() + (size_t)m * 16;
                const int c = tile_joint_pick(cnt, 1, C, ((unsigned)pass_rank[(size_t)m] + (unsigned)t) & cmask, taken);
                taken |= 1u << c;
                seq[(size_t)m][(size_t)t] = S.pos[(size_t)m * TILE_JOINT_MAX + head[c]++];
                cnt[c]--;
            }
        }
}

// ---- pieces of the tile plan that do not touch the nonzeros; shared by the host builder below
// ---- and the device builder (plan_device.hip)

std::vector<int32_t> tile_range_starts(int W, int64_t wpt, double taper)
{
    wpt = std::max<int64_t>(1, std::min<int64_t>(wpt, std::max(W, 1)));
    const int R = std::max(1, (int)((W + wpt - 1) / wpt));
    std::vector<int32_t> start((size_t)R + 1, 0);
    // with few ranges per block the long ones would outlast the launch: equal cuts there
    if (!(taper > 0.0) || R < 6) {
        for (int r = 0; r <= R; ++r) start[(size_t)r] = (int32_t)std::min<int64_t>((int64_t)r * wpt, W);
        return start;
    }
    taper = std::min(taper, 0.9);
    const double mean = (double)W / R;
    double cum = 0.0;
    for (int r = 0; r < R; ++r) {
        cum += mean * (1.0 + taper * (1.0 - 2.0 * r / (R - 1)));
        int64_t b = (int64_t)std::floor(cum + 0.5);
        b = std::max<int64_t>(b, (int64_t)start[(size_t)r] + 1);         // no empty range
        b = std::min<int64_t>(b, (int64_t)W - (R - 1 - r));              // room for the ranges that follow
        start[(size_t)r + 1] = (int32_t)b;
    }
    start[(size_t)R] = W;
    return start;
}

// dimensions, rows by length -> blocks, tasks, partial-row bookkeeping; P.steps zeroed
void tile_plan_begin(TilePlanHost &P, int64_t nnz, int n_major, int n_minor, const TileShape &shape,
                     const int64_t *mptr)
{
    const int lpc = shape.lpc, waves_per_block = shape.waves_per_block, target_tasks = shape.target_tasks;
    if (lpc < 1 || lpc > 64 || (64 % lpc) != 0) throw std::invalid_argument("lpc must divide 64");
    if (waves_per_block < 1 || waves_per_block > 16) throw std::invalid_argument("waves_per_block in [1,16]");
    if (shape.row_slots < 1) throw std::invalid_argument("row_slots must be positive");
    P = TilePlanHost();
    P.row_slots = shape.row_slots;
    int win_rows = shape.win_rows;
    if (shape.ring > 1) {
        // the only multi-slot schedule shipped is the half-window one (slots refilled AT the epoch
        // boundary); the asynchronous ring of round 2 and the double-buffered sub-windows of round 5 live in the history
        if (shape.sync_stage != 1) throw std::invalid_argument("multi-slot plans need sync_stage = 1");
        if (shape.slot_bytes < 16 * shape.row_slots + 64 || shape.slot_bytes % 16)
            throw std::invalid_argument("slot_bytes must hold a row and be a multiple of 16");
        P.ring = shape.ring;
        P.sync_stage = 1;
        P.look = shape.ring - 1;
        P.slot16 = shape.slot_bytes / 16;
        win_rows = (P.slot16 - 4) / shape.row_slots;   // the last 64 bytes of a slot stay free
    }
    if (win_rows < 1) throw std::invalid_argument("win_rows must be positive");
    P.single = shape.single;
    if (P.single && P.look > 0) throw std::invalid_argument("single step counts do not go with the work-ahead schedule");
    if ((int64_t)std::max(P.ring, 1) * std::max<int64_t>(P.slot16, (int64_t)win_rows * shape.row_slots) > 65536)
        throw std::invalid_argument("the LDS window does not fit 16-bit positions");
    P.n_major = n_major;
    P.n_minor = n_minor;
    P.lpc = lpc;
    P.gpw = 64 / lpc;
    P.wpb = waves_per_block;
    P.gpb = P.gpw * P.wpb;
    P.win_rows = win_rows;
    P.n_windows = (n_minor + win_rows - 1) / win_rows;
    P.nnz = nnz;
    const int W = P.n_windows, gpb = P.gpb;

    // rows by length, longest first (stable): a wave's groups then carry similar loads
    std::vector<int32_t> rows((size_t)n_major);
    std::iota(rows.begin(), rows.end(), 0);
    std::stable_sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) {
        return (mptr[(size_t)a + 1] - mptr[a]) > (mptr[(size_t)b + 1] - mptr[b]);
    });
    P.n_blocks = ((int64_t)n_major + gpb - 1) / gpb;
    P.block_rows.assign((size_t)P.n_blocks * gpb, -1);
    for (int m = 0; m < n_major; ++m) P.block_rows[(size_t)m] = rows[(size_t)m];

    // tasks: window ranges of `windows_per_task` windows, window-range major / block minor
    int64_t wpt = 1;
    if (target_tasks > 0) wpt = std::max<int64_t>(1, (int64_t)W * P.n_blocks / target_tasks);
    wpt = std::min<int64_t>(wpt, W);
    if (shape.ranges > 0) wpt = std::max<int64_t>(1, ((int64_t)W + shape.ranges - 1) / shape.ranges);
    // Small problems: a launch of slightly more workgroups than the GPU runs at once takes two rounds
    // where one would do (an 1/8 shard of C3: 275 tasks of 2 windows on 256 CUs = 4 window-times;
    // 175 tasks of 3 windows = 3).  With `slots` = workgroups this orientation can have in flight,
    // between one and two rounds' worth of tasks are regrouped into one round.
    if (shape.slots > 0 && shape.ranges <= 0) {
        const int64_t tasks = ((W + wpt - 1) / wpt) * P.n_blocks;
        if (tasks > shape.slots && tasks < 2 * (int64_t)shape.slots && P.n_blocks <= shape.slots) {
            const int64_t ranges = std::max<int64_t>(1, shape.slots / P.n_blocks);
            wpt = std::min<int64_t>(W, (W + ranges - 1) / ranges);
        }
    }
    P.windows_per_task = (int)wpt;
    P.range_start = tile_range_starts(W, wpt, tile_taper_applies(((W + wpt - 1) / wpt) * P.n_blocks, shape.slots) ? shape.taper : 0.0);
    const int n_ranges = (int)P.range_start.size() - 1;
    P.range_end_of_window.assign((size_t)W, 0);
    for (int r = 0; r < n_ranges; ++r)
        for (int w = P.range_start[(size_t)r]; w < P.range_start[(size_t)r + 1]; ++w) P.range_end_of_window[(size_t)w] = P.range_start[(size_t)r + 1];
    P.n_tasks = (int64_t)n_ranges * P.n_blocks;
    P.n_partial_rows = P.n_tasks * gpb;
    P.pstride = P.n_blocks * gpb;   // consecutive ranges of one block are n_blocks tasks apart
    P.task_block.resize((size_t)P.n_tasks);
    P.task_w0.resize((size_t)P.n_tasks);
    P.task_w1.resize((size_t)P.n_tasks);
    for (int r = 0; r < n_ranges; ++r)
        for (int64_t b = 0; b < P.n_blocks; ++b) {
            const size_t t = (size_t)r * P.n_blocks + b;
            P.task_block[t] = (int32_t)b;
            P.task_w0[t] = P.range_start[(size_t)r];
            P.task_w1[t] = P.range_start[(size_t)r + 1];
        }
    P.pfirst.assign((size_t)n_major, 0);
    P.pcount.assign((size_t)n_major, n_ranges);
    for (int64_t b = 0; b < P.n_blocks; ++b)
        for (int g = 0; g < gpb; ++g) {
            const int32_t row = P.block_rows[(size_t)b * gpb + g];
            if (row >= 0) P.pfirst[(size_t)row] = (int32_t)(b * gpb + g);
        }
    P.steps.assign((size_t)P.n_blocks * P.wpb * W, 0);
}

void xcd_launch_order(const TilePlanHost *const *plans, int n_plans, int n_xcd, int bundle, std::vector<int32_t> &order)
{
    struct Item { int32_t code; int64_t work; int64_t key; };
    std::vector<Item> items;
    for (int p = 0; p < n_plans; ++p) {
        const TilePlanHost &P = *plans[p];
        for (int64_t t = 0; t < P.n_tasks; ++t)
            items.push_back({p == 0 ? (int32_t)t : ~(int32_t)t, P.task_work[(size_t)t],
                             ((int64_t)p << 40) | (int64_t)P.task_w0[(size_t)t]});
    }
    const size_t n = items.size();
    order.assign(n, 0);
    if (n == 0) return;
    n_xcd = std::max(1, n_xcd);
    bundle = std::max(1, bundle);
    // tasks of one range together, longest first
    std::stable_sort(items.begin(), items.end(), [](const Item &x, const Item &y) {
        return x.key != y.key ? x.key < y.key : x.work > y.work;
    });
    struct Bundle { size_t first, count; int64_t longest, total; };
    std::vector<Bundle> bundles;
    for (size_t i = 0; i < n;) {
        size_t j = i;
        int64_t total = 0;
        while (j < n && items[j].key == items[i].key && j - i < (size_t)bundle) total += items[j++].work;
        bundles.push_back({i, j - i, items[i].work, total});
        i = j;
    }
    std::stable_sort(bundles.begin(), bundles.end(), [](const Bundle &x, const Bundle &y) { return x.longest > y.longest; });
    std::vector<std::vector<int32_t>> queue((size_t)n_xcd);
    std::vector<int64_t> load((size_t)n_xcd, 0);
    for (const Bundle &b : bundles) {
        int x = 0;
        for (int c = 1; c < n_xcd; ++c)
            if (load[(size_t)c] < load[(size_t)x]) x = c;
        for (size_t i = b.first; i < b.first + b.count; ++i) queue[(size_t)x].push_back(items[i].code);
        load[(size_t)x] += b.total;
    }
    // XCD x owns the slots x, x + n_xcd, ...: move the (short) tail tasks of over-full queues
    auto slots_of = [&](int x) { return (n - (size_t)x + (size_t)n_xcd - 1) / (size_t)n_xcd; };
    for (int x = 0; x < n_xcd; ++x)
        while (queue[(size_t)x].size() > slots_of(x)) {
            int y = 0;
            while (queue[(size_t)y].size() >= slots_of(y)) ++y;     // exists: the sizes add up to n
            queue[(size_t)y].push_back(queue[(size_t)x].back());
            queue[(size_t)x].pop_back();
        }
    for (int x = 0; x < n_xcd; ++x)
        for (size_t q = 0; q < queue[(size_t)x].size(); ++q) order[q * (size_t)n_xcd + (size_t)x] = queue[(size_t)x][q];
}

std::vector<double> block_shares(int64_t nnz, const int32_t *major, int n_major, int rows_per_block, int64_t stride)
{
    stride = std::max<int64_t>(1, stride);
    const int64_t n_samples = (nnz + stride - 1) / stride;
    const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(host_threads(), 16), n_samples / 65536 + 1));
    std::vector<std::vector<int32_t>> local((size_t)nth, std::vector<int32_t>((size_t)n_major, 0));
    parallel_for(n_samples, nth, [&](int64_t b, int64_t e, int t) {
        int32_t *h = local[(size_t)t].data();
        for (int64_t i = b; i < e; ++i) {
            const int32_t m = major[i * stride];
            if (m >= 0 && m < n_major) ++h[m];          // indices are validated elsewhere
        }
    });
    std::vector<int32_t> &count = local[0];
    int32_t longest = 0;
    for (int m = 0; m < n_major; ++m) {
        int32_t c = 0;
        for (int t = 0; t < nth; ++t) c += local[(size_t)t][(size_t)m];
        count[(size_t)m] = c;
        longest = std::max(longest, c);
    }
    // rows by decreasing length = a histogram of lengths walked from the top
    std::vector<int64_t> rows_of_length((size_t)longest + 1, 0);
    for (int m = 0; m < n_major; ++m) ++rows_of_length[(size_t)count[(size_t)m]];
    const int64_t n_blocks = ((int64_t)n_major + rows_per_block - 1) / rows_per_block;
    std::vector<double> share((size_t)n_blocks, 0.0);
    int64_t placed = 0;
    double total = 0.0;
    for (int64_t len = longest; len >= 0; --len) {
        int64_t n = rows_of_length[(size_t)len];
        while (n > 0) {
            const int64_t b = placed / rows_per_block;
            const int64_t take = std::min<int64_t>(n, (b + 1) * rows_per_block - placed);
            share[(size_t)b] += (double)take * (double)len;
            total += (double)take * (double)len;
            placed += take;
            n -= take;
        }
    }
    if (total > 0.0)
        for (double &v : share) v /= total;
    return share;
}

RangeChoice choose_task_ranges(const int64_t blocks[2], const int64_t half_windows[2], const bool half_ok[2],
                               const std::vector<double> block_share[2],
                               double nnz, int resident, double nnz_per_second, double task_seconds,
                               const double partial_seconds[2], int min_half_per_task, double window_penalty,
                               int max_ranges, bool separate_launches, double taper)
{
    struct Pool { int64_t n_tasks; int s; int64_t r, wpt, W; double per_nnz; bool half; };
    auto pool = [&](int s, int r, Pool &p) {
        // the schedule this orientation would get with r ranges
        const bool half = half_ok[s] && half_windows[s] / r >= min_half_per_task;
        const int64_t W = half ? half_windows[s] : (half_windows[s] + 1) / 2;
        if (r > W) return false;
        const int64_t wpt = (W + r - 1) / r;
        if ((W + wpt - 1) / wpt != r) return false;            // r is not a range count the planner produces
        p = Pool{blocks[s] * r, s, r, wpt, W, (half ? 1.0 : window_penalty) / nnz_per_second, half};
        return true;
    };
    RangeChoice best{{0, 0}, {false, false}, 1e300};
    std::vector<double> durations, load;
    // list schedule, longest first, on `resident` identical workgroups (a heap of their loads)
    auto span_of = [&](const Pool *pools, int n_pools) {
        durations.clear();
        for (int i = 0; i < n_pools; ++i) {
            const Pool &p = pools[i];
            const std::vector<double> &share = block_share[p.s];
            const std::vector<int32_t> starts = tile_range_starts((int)p.W, p.wpt, tile_taper_applies(p.n_tasks, separate_launches ? resident : resident / 2) ? taper : 0.0);
            for (int64_t b = 0; b < blocks[p.s]; ++b) {
                const double block_nnz = nnz * (share.empty() ? 1.0 / (double)blocks[p.s] : share[(size_t)b]);
                const double per_window = block_nnz / (double)p.W * p.per_nnz;
                for (size_t r = 0; r + 1 < starts.size(); ++r)
                    durations.push_back((double)(starts[r + 1] - starts[r]) * per_window + task_seconds);
            }
        }
        std::sort(durations.begin(), durations.end(), std::greater<double>());
        load.assign((size_t)resident, 0.0);
        std::make_heap(load.begin(), load.end(), std::greater<double>());      // min-heap
        for (double d : durations) {
            std::pop_heap(load.begin(), load.end(), std::greater<double>());
            load.back() += d;
            std::push_heap(load.begin(), load.end(), std::greater<double>());
        }
        return *std::max_element(load.begin(), load.end());
    };
    if (separate_launches) {   // one launch per orientation: each pool has the device to itself
        best.seconds = 0.0;
        for (int s = 0; s < 2; ++s) {
            double best_s = 1e300;
            for (int r = 1; r <= max_ranges; ++r) {
                Pool p;
                if (!pool(s, r, p) || (r > 1 && p.n_tasks > 16 * (int64_t)resident)) continue;
                const double total = span_of(&p, 1) + r * partial_seconds[s];
                if (total < best_s) { best_s = total; best.ranges[s] = r; best.half[s] = p.half; }
            }
            best.seconds += best_s;
        }
        return best;
    }
    for (int rc = 1; rc <= max_ranges; ++rc) {
        Pool pools[2];
        if (!pool(0, rc, pools[0])) continue;
        // the loss pass sweeps the cell-side plan alone, at about a PHI sweep's cost per nonzero, once per
        // check interval (every 10th iteration by default): its tail counts a tenth
        const double loss_share = 0.1 * span_of(pools, 1);
        for (int rg = 1; rg <= max_ranges; ++rg) {
            if (!pool(1, rg, pools[1])) continue;
            // many blocks are many tasks already: one range per block of such an orientation is always a
            // candidate, more ranges only while the orientation stays under 16 tasks per workgroup
            if ((rc > 1 && pools[0].n_tasks > 16 * (int64_t)resident) || (rg > 1 && pools[1].n_tasks > 16 * (int64_t)resident))
                continue;
            const double total = span_of(pools, 2) + loss_share + rc * partial_seconds[0] + rg * partial_seconds[1];
            if (total < best.seconds) best = RangeChoice{{rc, rg}, {pools[0].half, pools[1].half}, total};
        }
    }
    return best;
}

// with P.steps known: task work / merged-launch order, where every (block, wave)'s entries start
// (wave_off, in step slots) and where each task's waves start; returns the number of step slots
// including the zero padding the kernel's entry prefetch may read
int64_t tile_plan_offsets(TilePlanHost &P, std::vector<int64_t> &wave_off)
{
    const int W = P.n_windows, wpb = P.wpb, gpw = P.gpw;
    const int nth = host_threads();
    // work of a task in nonzero-times: its workgroup runs, window by window, as long as its slowest wave (+ a
    // staging of the window, worth two step slots = four nonzeros; half that per half window); task_order
    // (longest first) feeds the merged cell+gene launch
    P.task_work.assign((size_t)P.n_tasks, 0);
    parallel_for(P.n_tasks, nth, [&](int64_t t0_, int64_t t1_, int) {
        for (int64_t t = t0_; t < t1_; ++t) {
            const int64_t b = P.task_block[(size_t)t];
            int64_t work = 0;
            for (int w = P.task_w0[(size_t)t]; w < P.task_w1[(size_t)t]; ++w) {
                int mx = 0;
                for (int v = 0; v < wpb; ++v) mx = std::max<int>(mx, P.steps[((size_t)b * wpb + v) * W + w]);
                // in the unit the kernel spends its time in: `single` plans run one nonzero at a time (steps count
                // nonzeros, a stored slot holds two), so a window's fixed cost weighs the same against either kind
                work += (P.single ? (int64_t)mx : 2 * (int64_t)mx) + (P.ring > 1 ? 2 : 4);
            }
            P.task_work[(size_t)t] = work;
        }
    });
    P.task_order.resize((size_t)P.n_tasks);
    std::iota(P.task_order.begin(), P.task_order.end(), 0);
    std::stable_sort(P.task_order.begin(), P.task_order.end(), [&](int32_t x, int32_t y) {
        return P.task_work[(size_t)x] > P.task_work[(size_t)y];
    });

    wave_off.assign((size_t)P.n_blocks * wpb + 1, 0);   // entries of (block, wave), window order
    for (size_t bw = 0; bw < (size_t)P.n_blocks * wpb; ++bw) {
        int64_t tot = 0;
        for (int w = 0; w < W; ++w) tot += tile_stored_steps(P, P.steps[bw * W + w]);
        wave_off[bw + 1] = wave_off[bw] + tot * gpw;
    }
    P.task_wave_off.resize((size_t)P.n_tasks * wpb);
    parallel_for(P.n_blocks, nth, [&](int64_t b0, int64_t b1, int) {
        for (int64_t b = b0; b < b1; ++b)
            for (int v = 0; v < wpb; ++v) {
                const size_t bw = (size_t)b * wpb + v;
                int64_t off = wave_off[bw];
                int r = 0;
                for (int w = 0; w < W; ++w) {
                    if (w == P.range_start[(size_t)r + 1]) ++r;
                    if (w == P.range_start[(size_t)r]) P.task_wave_off[((size_t)r * P.n_blocks + b) * wpb + v] = off;
                    off += tile_stored_steps(P, P.steps[bw * W + w]) * gpw;
                }
            }
    });
    // + zero slots: the kernel's ring prefetch (depth 4, advanced in batches of 4) reads up to
    // 2 * 4 - 1 steps past a wave's last entry; 12 steps of padding keep that inside the buffer
    return wave_off.back() + (int64_t)12 * gpw;
}

// rank of every group of a wave inside its ds_read_b128 pass (16 lanes served per LDS cycle)
std::vector<int> tile_pass_rank(int lpc, int gpw)
{
    std::vector<int> pass_rank((size_t)gpw, 0);
    static const int pass_of_quad[16] = {0, 1, 1, 0, 1, 0, 0, 1, 2, 3, 3, 2, 3, 2, 2, 3};  // lanes 4q..4q+3
    int seen[4] = {0, 0, 0, 0};
    for (int g2 = 0; g2 < gpw; ++g2) {
        const int lane0 = g2 * lpc;
        if (lpc > 16) { pass_rank[(size_t)g2] = 0; continue; }
        const int ps = pass_of_quad[lane0 / 4];
        pass_rank[(size_t)g2] = seen[ps]++;
    }
    return pass_rank;
}

std::vector<int> tile_pass_of(int lpc, int gpw)
{
    std::vector<int> pass((size_t)gpw, 0);
    static const int pass_of_quad[16] = {0, 1, 1, 0, 1, 0, 0, 1, 2, 3, 3, 2, 3, 2, 2, 3};
    for (int g2 = 0; g2 < gpw; ++g2) pass[(size_t)g2] = lpc > 16 ? 0 : pass_of_quad[g2 * lpc / 4];
    return pass;
}

void tile_plan_report(const TilePlanHost &P)
{
    // where the stored slots go: nonzeros / sliced-ELL padding inside a wave / waiting at the
    // window barrier for the slowest wave of the workgroup
    const int W = P.n_windows, wpb = P.wpb, gpw = P.gpw;
    int64_t wave_steps = 0, barrier_steps = 0;
    for (int64_t b = 0; b < P.n_blocks; ++b)
        for (int w = 0; w < W; ++w) {
            int mx = 0;
            for (int v = 0; v < wpb; ++v) {
                const int s = P.steps.empty() ? 0 : P.steps[((size_t)b * wpb + v) * W + w];
                wave_steps += s;
                mx = std::max(mx, s);
            }
            barrier_steps += (int64_t)mx * wpb;
        }
    const int per_step = P.single ? 1 : 2;   // nonzeros an executed step takes
    fprintf(stderr, "[schpf_hip]     nnz %lld, executed nonzero slots %lld (ELL fill %.3f), barrier-limited wave-steps %lld vs %lld "
            "(%.3f)\n", (long long)P.nnz, (long long)(wave_steps * gpw * per_step),
            wave_steps ? (double)P.nnz / (double)(wave_steps * gpw * per_step) : 0.0, (long long)barrier_steps,
            (long long)wave_steps, wave_steps ? (double)wave_steps / (double)barrier_steps : 0.0);
}

// Half-window schedule of one block (plan.h): for every epoch e the steps T_e each WAVE runs, and for
// every lane how many of its row's nonzeros (in minor order) it has consumed BEFORE epoch e.
//   start[g * (W + 1) + e], e = 0..W;  T[wave * W + e]
// Greedy: T_e = ceil(max over the wave's lanes of the nonzeros still owed up to sub-window e, / 2);
// every lane then takes min(2 T_e, what lies inside the readable horizon min(e + ring, task end)).
// cnt_below(g, bound) = number of the lane's nonzeros with minor < bound * win_rows.
template <typename CntBelow>
static void ring_schedule_block(const TilePlanHost &P, int64_t b, CntBelow cnt_below, int32_t *start, uint32_t *T)
{
    const int W = P.n_windows, gpb = P.gpb;
    std::vector<int32_t> done((size_t)gpb, 0);
    for (int e = 0; e < W; ++e) {
        const int w1 = P.range_end_of_window[(size_t)e];   // end of the task e belongs to
        const int hor = std::min(e + P.look + 1, w1);
        const int span = P.gpw;   // the waves meet at the barrier anyway: every wave has its own T_e
        for (int g0 = 0; g0 < gpb; g0 += span) {
            int32_t need = 0;
            for (int g = g0; g < g0 + span; ++g) need = std::max(need, cnt_below(b, g, e + 1) - done[(size_t)g]);
            const uint32_t Te = (uint32_t)((need + 1) / 2);
            T[(size_t)(g0 / P.gpw) * W + e] = Te;
            for (int g = g0; g < g0 + span; ++g) {
                start[(size_t)g * (W + 1) + e] = done[(size_t)g];
                const int32_t avail = cnt_below(b, g, hor) - done[(size_t)g];
                done[(size_t)g] += std::min<int32_t>((int32_t)(2 * Te), avail);
            }
        }
    }
    for (int g = 0; g < gpb; ++g) start[(size_t)g * (W + 1) + W] = done[(size_t)g];
}

void build_tile_plan(int64_t nnz, const int32_t *major, const int32_t *minor, const float *val,
                     int n_major, int n_minor, const TileShape &shape, bool keep_order, TilePlanHost &P)
{
    const bool verbose = getenv("SCHPF_VERBOSE") && atoi(getenv("SCHPF_VERBOSE"));
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    BigVec<int32_t> order;
    std::vector<int64_t> mptr;
    sort_by_major_minor(nnz, major, minor, n_major, n_minor, order, mptr);
    const double t1 = now();
    tile_plan_begin(P, nnz, n_major, n_minor, shape, mptr.data());
    const int W = P.n_windows, gpb = P.gpb, gpw = P.gpw, wpb = P.wpb, win_rows = P.win_rows;
    const bool ring = P.ring > 1;
    const int per_step = P.single ? 1 : 2;        // nonzeros per counted step

    // sorted copies: every later pass walks the rows' runs sequentially
    const int nth = host_threads();
    BigVec<int32_t> s_minor((size_t)nnz);   // not zero-filled: written by the parallel pass below
    BigVec<float> s_val((size_t)nnz);
    parallel_for(nnz, nth, [&](int64_t b, int64_t e, int) {
        for (int64_t j = b; j < e; ++j) {
            const int32_t pos = order[(size_t)j];
            s_minor[(size_t)j] = minor[pos];
            s_val[(size_t)j] = val[pos];
        }
    });

    const double t2 = now();
    std::vector<int> err((size_t)nth + 1, 0);
    // ring mode: consumed-before-epoch counts of every lane, [n_blocks * gpb][W + 1]
    BigVec<int32_t> starts;
    if (ring) starts.resize((size_t)P.n_blocks * gpb * ((size_t)W + 1));
    if (!ring) {
        // per (block, wave, window): steps = ceil(longest segment / 2).  Threads own whole blocks.
        parallel_for(P.n_blocks, nth, [&](int64_t b0, int64_t b1, int t) {
            for (int64_t b = b0; b < b1; ++b)
                for (int g = 0; g < gpb; ++g) {
                    const int32_t row = P.block_rows[(size_t)b * gpb + g];
                    if (row < 0) continue;
                    uint16_t *st = P.steps.data() + ((size_t)b * wpb + g / gpw) * W;
                    int64_t j = mptr[row];
                    const int64_t end = mptr[(size_t)row + 1];
                    while (j < end) {           // runs of equal window (the row is sorted by minor)
                        const int32_t w = s_minor[(size_t)j] / win_rows;
                        const int64_t bound = ((int64_t)w + 1) * win_rows;   // one division per run, not per nonzero
                        int64_t s = j;
                        while (j < end && s_minor[(size_t)j] < bound) ++j;
                        const int64_t steps = (j - s + per_step - 1) / per_step;   // two nonzeros per step (one: single)
                        if (steps > 65535) { err[(size_t)t] = 1; continue; }
                        if (steps > st[w]) st[w] = (uint16_t)steps;
                    }
                }
        });
    } else {
        parallel_for(P.n_blocks, nth, [&](int64_t b0, int64_t b1, int t) {
            std::vector<int32_t> below((size_t)gpb * ((size_t)W + 1));   // nonzeros of lane g with window < w
            std::vector<uint32_t> T((size_t)wpb * W);
            for (int64_t b = b0; b < b1; ++b) {
                for (int g = 0; g < gpb; ++g) {
                    int32_t *bl = below.data() + (size_t)g * (W + 1);
                    const int32_t row = P.block_rows[(size_t)b * gpb + g];
                    if (row < 0) { std::fill(bl, bl + W + 1, 0); continue; }
                    int64_t j = mptr[row];
                    const int64_t r0 = j, end = mptr[(size_t)row + 1];
                    for (int w = 0; w < W; ++w) {
                        bl[w] = (int32_t)(j - r0);
                        const int64_t bound = ((int64_t)w + 1) * win_rows;
                        while (j < end && s_minor[(size_t)j] < bound) ++j;
                    }
                    bl[W] = (int32_t)(j - r0);
                }
                ring_schedule_block(P, b, [&](int64_t, int g, int w) { return below[(size_t)g * (W + 1) + w]; },
                                    starts.data() + (size_t)b * gpb * ((size_t)W + 1), T.data());
                for (size_t i = 0; i < (size_t)wpb * W; ++i) {   // T[wave][epoch]
                    if (T[i] > 65535u) { err[(size_t)t] = 1; continue; }
                    P.steps[(size_t)b * wpb * W + i] = (uint16_t)T[i];
                }
            }
        });
    }
    for (int e : err)
        if (e) throw std::invalid_argument(P.single ? "a row has more than 65535 nonzeros in one window"
                                                 : "a row has more than 131070 nonzeros in one window");

    const double t3 = now();
    std::vector<int64_t> wave_off;
    const int64_t total_padded = tile_plan_offsets(P, wave_off);
    // packed entries (8 bytes per step: two 16-bit LDS positions + two 16-bit counts) when
    // every count fits 16 bits -- UMI counts do; otherwise 16 bytes per step (32-bit position, float)
    bool packed = shape.allow_packed;
    if (packed) {
        std::vector<int> big((size_t)nth + 1, 0);
        parallel_for(nnz, nth, [&](int64_t b, int64_t e, int t) {
            for (int64_t j = b; j < e; ++j) {
                const float f = s_val[(size_t)j];
                if (!(f <= 65535.0f) || f != (float)(uint32_t)f) { big[(size_t)t] = 1; break; }
            }
        });
        for (int v : big) packed = packed && !v;
    }
    P.packed = packed;
    const int epw = packed ? 2 : 4;      // 32-bit words per step slot
    P.entries.resize((size_t)total_padded * epw);
    parallel_for(total_padded * epw, nth, [&](int64_t b, int64_t e, int) {
        std::memset(P.entries.data() + b, 0, (size_t)(e - b) * sizeof(uint32_t));
    });
    // ring mode: an unused step slot must still point at a row that is valid while it is read --
    // the first row of the epoch's own slot (count 0: it contributes nothing)
    if (ring)
        parallel_for(P.n_blocks * wpb, nth, [&](int64_t bw0, int64_t bw1, int) {
            for (int64_t bw = bw0; bw < bw1; ++bw) {
                int64_t off = wave_off[(size_t)bw];
                for (int w = 0; w < W; ++w) {
                    const int64_t n = tile_stored_steps(P, P.steps[(size_t)bw * W + w]) * gpw;
                    const uint32_t o16 = (uint32_t)(w % P.ring) * (uint32_t)P.slot16;
                    for (int64_t q = off; q < off + n; ++q) {
                        if (packed) P.entries[(size_t)q * 2] = o16 | (o16 << 16);
                        else { P.entries[(size_t)q * 4] = o16; P.entries[(size_t)q * 4 + 2] = o16; }
                    }
                    off += tile_stored_steps(P, P.steps[(size_t)bw * W + w]) * gpw;
                }
            }
        });
    // fill: a segment = the nonzeros one lane consumes in one window (window mode: its row's
    // nonzeros of that window; ring mode: what the schedule gave it in that epoch); inside a
    // segment the nonzeros may be taken in any order, so they are dealt to the steps in an
    // LDS-bank-aware order (see bank_order)
    const std::vector<int> pass_rank = tile_pass_rank(P.lpc, gpw);
    auto put = [&](int64_t win_first, int slot, int t, int64_t src) {
        const uint32_t o16 = tile_off16(P, s_minor[(size_t)src]);
        const size_t step_slot = (size_t)win_first + (size_t)(t >> 1) * gpw + slot;
        if (packed) {
            uint32_t *e = P.entries.data() + step_slot * 2;
            const int sh = (t & 1) * 16;
            e[0] = (e[0] & ~(0xFFFFu << sh)) | (o16 << sh);
            e[1] |= (uint32_t)s_val[(size_t)src] << sh;
        } else {
            uint32_t *e = P.entries.data() + step_slot * 4 + (size_t)(t & 1) * 2;
            e[0] = o16;
            e[1] = f2u(s_val[(size_t)src]);
        }
    };
    const bool joint = shape.bank_order == 2 && P.lpc <= 8;
    if (joint) {
        // wave by wave, window by window: the segments of the wave's lane groups are dealt together
        const std::vector<int> pass_of = tile_pass_of(P.lpc, gpw);
        parallel_for(P.n_blocks * wpb, nth, [&](int64_t bw0, int64_t bw1, int) {
            JointScratch scratch;
            std::vector<std::vector<int32_t>> seq((size_t)gpw);
            std::vector<int64_t> cur((size_t)gpw), r0((size_t)gpw), end((size_t)gpw), seg_s((size_t)gpw);
            std::vector<const int32_t *> seg_minor((size_t)gpw);
            std::vector<int> seg_n((size_t)gpw);
            for (int64_t bw = bw0; bw < bw1; ++bw) {
                const int32_t *rows = P.block_rows.data() + (size_t)bw * gpw;   // gpb = wpb * gpw
                for (int m = 0; m < gpw; ++m) {
                    r0[(size_t)m] = cur[(size_t)m] = rows[m] < 0 ? 0 : mptr[rows[m]];
                    end[(size_t)m] = rows[m] < 0 ? 0 : mptr[(size_t)rows[m] + 1];
                }
                int64_t off = wave_off[(size_t)bw];
                for (int w = 0; w < W; ++w) {
                    for (int m = 0; m < gpw; ++m) {
                        int64_t s, e;
                        if (ring) {
                            const int32_t *st = starts.data() + ((size_t)bw * gpw + m) * ((size_t)W + 1);
                            s = r0[(size_t)m] + (rows[m] < 0 ? 0 : st[w]);
                            e = r0[(size_t)m] + (rows[m] < 0 ? 0 : st[w + 1]);
                        } else {
                            const int64_t bound = ((int64_t)w + 1) * win_rows;
                            s = e = cur[(size_t)m];
                            while (e < end[(size_t)m] && s_minor[(size_t)e] < bound) ++e;
                            cur[(size_t)m] = e;
                        }
                        seg_s[(size_t)m] = s;
                        seg_minor[(size_t)m] = s_minor.data() + s;
                        seg_n[(size_t)m] = (int)(e - s);
                    }
                    bank_order_joint(P, gpw, seg_minor.data(), seg_n.data(), pass_of, pass_rank, seq, scratch);
                    for (int m = 0; m < gpw; ++m)
                        for (int t = 0; t < seg_n[(size_t)m]; ++t) put(off, m, t, seg_s[(size_t)m] + seq[(size_t)m][(size_t)t]);
                    off += tile_stored_steps(P, P.steps[(size_t)bw * W + w]) * gpw;
                }
            }
        });
    }
    else parallel_for(P.n_blocks, nth, [&](int64_t b0, int64_t b1, int) {
        std::vector<int64_t> win_off((size_t)W);
        std::vector<int32_t> seq;
        std::vector<int32_t> buckets;   // scratch of bank_order
        for (int64_t b = b0; b < b1; ++b) {
            for (int g = 0; g < gpb; ++g) {
                const int32_t row = P.block_rows[(size_t)b * gpb + g];
                if (row < 0) continue;
                const size_t bw = (size_t)b * wpb + g / gpw;
                int64_t off = wave_off[bw];
                for (int w = 0; w < W; ++w) { win_off[(size_t)w] = off; off += tile_stored_steps(P, P.steps[bw * W + w]) * gpw; }
                const int slot = g % gpw;
                const int32_t *st = ring ? starts.data() + ((size_t)b * gpb + g) * ((size_t)W + 1) : nullptr;
                int64_t j = mptr[row];
                const int64_t r0 = j, end = mptr[(size_t)row + 1];
                int w = 0;
                while (ring ? w < W : j < end) {
                    int64_t s;
                    if (ring) {
                        s = r0 + st[w];
                        j = r0 + st[w + 1];
                    } else {
                        w = s_minor[(size_t)j] / win_rows;
                        const int64_t bound = ((int64_t)w + 1) * win_rows;
                        s = j;
                        while (j < end && s_minor[(size_t)j] < bound) ++j;
                    }
                    const int n = (int)(j - s);
                    seq.resize((size_t)n);
                    bank_order(P, s_minor.data() + s, n, shape.bank_order != 0, pass_rank[(size_t)slot], seq, buckets);
                    for (int t = 0; t < n; ++t) put(win_off[(size_t)w], slot, t, s + seq[(size_t)t]);
                    if (ring) ++w;
                }
            }
        }
    });
    if (verbose) {
        fprintf(stderr, "[schpf_hip]     sort %.3f s, sorted copies %.3f s, steps %.3f s, alloc+fill %.3f s\n", t1 - t0,
                t2 - t1, t3 - t2, now() - t3);
        tile_plan_report(P);
    }
    if (keep_order) {
        P.order.swap(order);
        P.mptr.swap(mptr);
    }
}


// ---- balanced windows (plan.h): the host reference; plan_device.hip balance_kernel follows the same rule
// One (block, section): keys = the block's nonzeros of the section's minor rows as (minor << GROUP_BITS | group),
// ascending.  virt_b / minor_of_b: this block's rows of the two tables.
static void balance_section(const uint64_t *keys, int64_t n_keys, int m0, int m1, int w0, int Dn, int D, int win_rows,
                            std::vector<uint16_t> &load, int32_t *virt_b, int32_t *minor_of_b)
{
    const uint64_t gmask = ((uint64_t)1 << BALANCE_GROUP_BITS) - 1;
    std::fill(load.begin(), load.end(), (uint16_t)0);
    int cnt[64] = {0};
    for (int64_t i = 0; i < n_keys;) {
        const int32_t m = (int32_t)(keys[i] >> BALANCE_GROUP_BITS);
        int64_t e = i;
        while (e < n_keys && (int32_t)(keys[e] >> BALANCE_GROUP_BITS) == m) ++e;
        uint64_t best = ~(uint64_t)0;
        for (int c = 0; c < Dn; ++c) {
            if (cnt[c] >= win_rows) continue;
            unsigned mx = 0, sm = 0;
            for (int64_t q = i; q < e; ++q) {
                const unsigned v = load[(size_t)(keys[q] & gmask) * D + c];
                mx = std::max(mx, v);
                sm += v;
            }
            best = std::min(best, balance_cost(mx, sm, (unsigned)c));
        }
        const int c = (int)(best & 0xff);
        for (int64_t q = i; q < e; ++q) {
            uint16_t &v = load[(size_t)(keys[q] & gmask) * D + c];
            if (v < 65535) ++v;
        }
        const int32_t v = (w0 + c) * win_rows + cnt[c]++;
        virt_b[m] = v;
        minor_of_b[v] = m;
        i = e;
    }
    // the minor rows no row of the block holds: the capacity left, window by window
    int c = 0;
    for (int32_t m = m0; m < m1; ++m) {
        if (virt_b[m] >= 0) continue;
        while (cnt[c] >= win_rows) ++c;
        const int32_t v = (w0 + c) * win_rows + cnt[c]++;
        virt_b[m] = v;
        minor_of_b[v] = m;
    }
}

void balance_windows_host(int64_t nnz, const int32_t *major, const int32_t *minor, int n_major, int n_minor,
                          const TileShape &shape, BigVec<int32_t> &vminor, std::vector<int32_t> &minor_of,
                          BalanceGeometry &geo)
{
    if (shape.ring > 1) throw std::invalid_argument("balanced windows need whole windows (ring <= 1)");
    // rows -> blocks exactly as the builder will cut them: that depends on the row lengths only
    BigVec<int32_t> order;
    std::vector<int64_t> mptr;
    counting_sort_positions(nnz, major, n_major, order, mptr);
    TilePlanHost T;
    tile_plan_begin(T, nnz, n_major, n_minor, shape, mptr.data());
    const int gpb = T.gpb, W = T.n_windows, win_rows = T.win_rows;
    if (gpb > (1 << BALANCE_GROUP_BITS)) throw std::invalid_argument("balanced windows: more than 1024 rows per block");
    geo = BalanceGeometry();
    geo.gpb = gpb; geo.win_rows = win_rows; geo.n_windows = W; geo.n_blocks = T.n_blocks;
    balance_sections(W, gpb, geo.n_sections, geo.D);
    if ((int64_t)W * win_rows > 0x7fffffff) throw std::invalid_argument("balanced windows: virtual index overflow");
    geo.n_virtual = W * win_rows;
    const int D = geo.D, nsec = geo.n_sections;
    vminor.resize((size_t)nnz);
    minor_of.assign((size_t)T.n_blocks * geo.n_virtual, -1);
    const int nth = host_threads();
    parallel_for(T.n_blocks, nth, [&](int64_t b0, int64_t b1, int) {
        std::vector<uint64_t> keys;
        std::vector<int32_t> virt_b((size_t)n_minor);
        std::vector<uint16_t> load((size_t)gpb * D);
        for (int64_t b = b0; b < b1; ++b) {
            keys.clear();
            for (int g = 0; g < gpb; ++g) {
                const int32_t row = T.block_rows[(size_t)b * gpb + g];
                if (row < 0) continue;
                for (int64_t j = mptr[row]; j < mptr[(size_t)row + 1]; ++j)
                    keys.push_back(((uint64_t)(uint32_t)minor[order[(size_t)j]] << BALANCE_GROUP_BITS) | (uint64_t)g);
            }
            std::sort(keys.begin(), keys.end());
            keys.erase(std::unique(keys.begin(), keys.end()), keys.end());   // a COO may hold an entry twice
            std::fill(virt_b.begin(), virt_b.end(), -1);
            int32_t *mo = minor_of.data() + (size_t)b * geo.n_virtual;
            size_t lo = 0;
            for (int s = 0; s < nsec; ++s) {
                const int w0 = s * D, w1 = std::min(W, w0 + D);
                const int m0 = w0 * win_rows, m1 = (int)std::min<int64_t>(n_minor, (int64_t)w1 * win_rows);
                size_t hi = lo;
                while (hi < keys.size() && (int64_t)(keys[hi] >> BALANCE_GROUP_BITS) < (int64_t)m1) ++hi;
                balance_section(keys.data() + lo, (int64_t)(hi - lo), m0, m1, w0, w1 - w0, D, win_rows, load, virt_b.data(), mo);
                lo = hi;
            }
            for (int g = 0; g < gpb; ++g) {
                const int32_t row = T.block_rows[(size_t)b * gpb + g];
                if (row < 0) continue;
                for (int64_t j = mptr[row]; j < mptr[(size_t)row + 1]; ++j) {
                    const int32_t pos = order[(size_t)j];
                    vminor[(size_t)pos] = virt_b[(size_t)minor[pos]];
                }
            }
        }
    });
}

}  // namespace schpf
