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

    // ---- sort positions by (major, minor): minor first, then stable by major ----
    std::vector<int32_t> by_minor;
    std::vector<int64_t> tmp_ptr;
    counting_sort_positions(nnz, minor, n_minor, by_minor, tmp_ptr);
    std::vector<int64_t> mptr((size_t)n_major + 1, 0);
    for (int64_t i = 0; i < nnz; ++i) mptr[(size_t)major[i] + 1]++;
    for (int m = 0; m < n_major; ++m) mptr[(size_t)m + 1] += mptr[m];
    std::vector<int32_t> order((size_t)nnz);
    {
        std::vector<int64_t> cur(mptr.begin(), mptr.end() - 1);
        for (int64_t j = 0; j < nnz; ++j) {
            int32_t pos = by_minor[(size_t)j];
            order[(size_t)cur[major[pos]]++] = pos;
        }
    }
    by_minor.clear();
    by_minor.shrink_to_fit();

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

}  // namespace schpf
