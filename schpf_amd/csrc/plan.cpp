// Host-side construction of the sweep plans (see plan.h for the layout).
#include "plan.h"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <stdexcept>

namespace schpf {

void counting_sort_positions(int64_t n, const int32_t *key, int nkeys, std::vector<int32_t> &order,
                             std::vector<int64_t> &ptr)
{
    ptr.assign((size_t)nkeys + 1, 0);
    for (int64_t i = 0; i < n; ++i) ptr[(size_t)key[i] + 1]++;
    for (int k = 0; k < nkeys; ++k) ptr[(size_t)k + 1] += ptr[k];
    std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
    order.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) order[(size_t)cur[key[i]]++] = (int32_t)i;
}

void sort_by_major_minor(int64_t nnz, const int32_t *major, const int32_t *minor, int n_major, int n_minor,
                         std::vector<int32_t> &order, std::vector<int64_t> &mptr)
{
    // minor first, then stable by major
    std::vector<int32_t> by_minor;
    std::vector<int64_t> tmp_ptr;
    counting_sort_positions(nnz, minor, n_minor, by_minor, tmp_ptr);
    mptr.assign((size_t)n_major + 1, 0);
    for (int64_t i = 0; i < nnz; ++i) mptr[(size_t)major[i] + 1]++;
    for (int m = 0; m < n_major; ++m) mptr[(size_t)m + 1] += mptr[m];
    order.resize((size_t)nnz);
    std::vector<int64_t> cur(mptr.begin(), mptr.end() - 1);
    for (int64_t j = 0; j < nnz; ++j) {
        const int32_t pos = by_minor[(size_t)j];
        order[(size_t)cur[major[pos]]++] = pos;
    }
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

    std::vector<int32_t> order;
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


void build_tile_plan(int64_t nnz, const int32_t *major, const int32_t *minor, const float *val,
                     int n_major, int n_minor, int lpc, int waves_per_block, int win_rows,
                     int target_tasks, bool keep_order, TilePlanHost &P)
{
    if (lpc < 1 || lpc > 64 || (64 % lpc) != 0) throw std::invalid_argument("lpc must divide 64");
    if (waves_per_block < 1 || waves_per_block > 16) throw std::invalid_argument("waves_per_block in [1,16]");
    if (win_rows < 1) throw std::invalid_argument("win_rows must be positive");
    P = TilePlanHost();
    P.n_major = n_major;
    P.n_minor = n_minor;
    P.lpc = lpc;
    P.gpw = 64 / lpc;
    P.wpb = waves_per_block;
    P.gpb = P.gpw * P.wpb;
    P.win_rows = win_rows;
    P.n_windows = (n_minor + win_rows - 1) / win_rows;
    P.nnz = nnz;
    const int W = P.n_windows, gpb = P.gpb, gpw = P.gpw, wpb = P.wpb;

    std::vector<int32_t> order;
    std::vector<int64_t> mptr;
    sort_by_major_minor(nnz, major, minor, n_major, n_minor, order, mptr);

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
    P.windows_per_task = (int)wpt;
    const int n_ranges = (int)((W + wpt - 1) / wpt);
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
            P.task_w0[t] = (int32_t)(r * wpt);
            P.task_w1[t] = (int32_t)std::min<int64_t>((r + 1) * wpt, W);
        }
    P.pfirst.assign((size_t)n_major, 0);
    P.pcount.assign((size_t)n_major, n_ranges);
    for (int64_t b = 0; b < P.n_blocks; ++b)
        for (int g = 0; g < gpb; ++g) {
            const int32_t row = P.block_rows[(size_t)b * gpb + g];
            if (row >= 0) P.pfirst[(size_t)row] = (int32_t)(b * gpb + g);
        }

    // per (block, wave, window): steps = ceil(longest segment / 2); then offsets and entries
    P.steps.assign((size_t)P.n_blocks * wpb * W, 0);
    std::vector<int32_t> seglen((size_t)W);
    for (int64_t b = 0; b < P.n_blocks; ++b)
        for (int g = 0; g < gpb; ++g) {
            const int32_t row = P.block_rows[(size_t)b * gpb + g];
            if (row < 0) continue;
            std::fill(seglen.begin(), seglen.end(), 0);
            for (int64_t j = mptr[row]; j < mptr[(size_t)row + 1]; ++j) seglen[(size_t)(minor[order[(size_t)j]] / win_rows)]++;
            uint16_t *st = P.steps.data() + ((size_t)b * wpb + g / gpw) * W;
            for (int w = 0; w < W; ++w) {
                const int32_t s = (seglen[(size_t)w] + 1) / 2;   // two nonzeros per step
                if (s > 65535) throw std::invalid_argument("a row has more than 131070 nonzeros in one window");
                if (s > st[w]) st[w] = (uint16_t)s;
            }
        }
    std::vector<int64_t> wave_off((size_t)P.n_blocks * wpb + 1, 0);   // entries of (block, wave), window order
    for (size_t bw = 0; bw < (size_t)P.n_blocks * wpb; ++bw) {
        int64_t tot = 0;
        for (int w = 0; w < W; ++w) tot += P.steps[bw * W + w];
        wave_off[bw + 1] = wave_off[bw] + tot * gpw;
    }
    const int64_t total = wave_off.back();
    P.entries.assign((size_t)total * 4, 0u);
    P.task_wave_off.resize((size_t)P.n_tasks * wpb);
    P.task_wave_end.resize((size_t)P.n_tasks * wpb);
    for (int64_t b = 0; b < P.n_blocks; ++b)
        for (int v = 0; v < wpb; ++v) {
            const size_t bw = (size_t)b * wpb + v;
            int64_t off = wave_off[bw];
            for (int w = 0; w < W; ++w) {
                const size_t tv = ((size_t)(w / wpt) * P.n_blocks + b) * wpb + v;
                if (w % wpt == 0) P.task_wave_off[tv] = off;
                off += (int64_t)P.steps[bw * W + w] * gpw;
                P.task_wave_end[tv] = off;
            }
        }
    // fill: walk each row's nonzeros in minor order; position inside its (window) segment = t
    std::vector<int64_t> win_off((size_t)W);
    for (int64_t b = 0; b < P.n_blocks; ++b)
        for (int g = 0; g < gpb; ++g) {
            const int32_t row = P.block_rows[(size_t)b * gpb + g];
            if (row < 0) continue;
            const size_t bw = (size_t)b * wpb + g / gpw;
            int64_t off = wave_off[bw];
            for (int w = 0; w < W; ++w) { win_off[(size_t)w] = off; off += (int64_t)P.steps[bw * W + w] * gpw; }
            const int slot = g % gpw;
            int32_t cur_w = -1, t = 0;
            for (int64_t j = mptr[row]; j < mptr[(size_t)row + 1]; ++j) {
                const int32_t pos = order[(size_t)j];
                const int32_t w = minor[pos] / win_rows;
                if (w != cur_w) { cur_w = w; t = 0; }
                uint32_t *e = P.entries.data() + ((size_t)win_off[(size_t)w] + (size_t)(t >> 1) * gpw + slot) * 4 + (size_t)(t & 1) * 2;
                e[0] = (uint32_t)(minor[pos] - w * win_rows);
                e[1] = f2u(val[pos]);
                ++t;
            }
        }
    if (keep_order) {
        P.order.swap(order);
        P.mptr.swap(mptr);
    }
}

}  // namespace schpf
