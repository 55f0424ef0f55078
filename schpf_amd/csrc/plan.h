// Sweep plan: the HBM layout of the sparse UMI matrix for one orientation.
//
// The CAVI responsibility pass touches every stored nonzero (cell i, gene g, count x)
// once per orientation:  the CELL sweep accumulates into per-cell K-vectors (major =
// cell, minor = gene; replaces the theta half of schpf/hpf_numba.py:128-156 driven by
// schpf/scHPF_.py:709-710), the GENE sweep into per-gene K-vectors (major = gene,
// minor = cell; scHPF_.py:699-700).  Both use the same layout, built here on the host:
//
//   * nonzeros are sorted by (major, minor); each major's run is cut into CHUNKS of at
//     most `chunk_len` nonzeros that do not straddle a minor WINDOW (windows keep the
//     gathered table slice of the minor side L2-resident; window w is worked on by the
//     XCDs congruent to w);
//   * chunks are ordered (window, length descending) and grouped CPW = 64 / LPC at a
//     time into SLICES -- one slice is what one wavefront streams;
//   * inside a slice nonzeros are stored step-major ("sliced ELL"): step p holds, for
//     every chunk of the slice, two nonzeros packed as uint4 {minor0, val0, minor1,
//     val1}; the 64 lanes of a wave therefore read one fully coalesced 16 B x CPW line
//     per step.  Short chunks are padded with {0, 0.0f} (a zero count contributes
//     nothing).
//
// A chunk's partial K-vector is written to row `natural id` of a partials matrix; the
// natural ids of one major are consecutive (cptr), so the fused update kernel reduces
// them in a fixed order: no atomics, run-to-run deterministic.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <sys/mman.h>
#include <memory>
#include <new>
#include <stdexcept>
#include <utility>
#include <vector>

namespace schpf {

// hipMalloc found no device memory while a plan was being built (plan_device.hip); the C ABI reports it -- like
// capi.hip's own allocations and std::bad_alloc -- with the status SCHPF_ERR_NO_MEMORY
struct DeviceNoMemory : std::runtime_error { using std::runtime_error::runtime_error; };

// std::vector whose resize() does NOT zero-fill: the big per-nonzero arrays of a plan (hundreds of
// MB) are written exactly once by parallel passes; a value-initialising resize would first-touch
// every page from ONE thread, which at 1e8 nonzeros cost more than the passes themselves.
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef NoInitAlloc<U> other; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <class U, class... A> void construct(U *p, A &&...a)
    {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
        else ::new ((void *)p) U(std::forward<A>(a)...);
    }
    // large blocks: 2 MiB-aligned and advised as huge pages (first touch of 0.5 GB is then a few
    // hundred page faults instead of 125 000)
    T *allocate(std::size_t n)
    {
        const std::size_t bytes = n * sizeof(T);
        if (bytes < ((std::size_t)4 << 20)) return static_cast<T *>(::operator new(bytes));
        void *p = nullptr;
        if (posix_memalign(&p, (std::size_t)2 << 20, bytes) != 0) throw std::bad_alloc();
        madvise(p, bytes, MADV_HUGEPAGE);
        return static_cast<T *>(p);
    }
    void deallocate(T *p, std::size_t n) noexcept
    {
        if (n * sizeof(T) < ((std::size_t)4 << 20)) ::operator delete(p);
        else free(p);
    }
};
template <class T> using BigVec = std::vector<T, NoInitAlloc<T>>;


struct SweepPlanHost {
    int n_major = 0, n_minor = 0;
    int lpc = 1;            // lanes per chunk
    int cpw = 64;           // chunks per wave (= 64 / lpc)
    int chunk_len = 0;      // max nonzeros per chunk (even)
    int n_windows = 1;
    int64_t nnz = 0;
    int64_t n_chunks = 0;   // natural chunks (rows of the partials matrix)
    int64_t n_slices = 0;
    int64_t n_waves = 0;    // launch size in wavefronts (multiple of 4)
    BigVec<uint32_t> entries;             // 4 words per (step, chunk-slot): see above
    std::vector<int64_t> slice_off;       // [n_slices] offset into entries, in uint4 units
    std::vector<int32_t> slice_steps;     // [n_slices] number of uint4 steps
    std::vector<int32_t> chunk_major;     // [n_slices * cpw], -1 for an empty slot
    std::vector<int32_t> chunk_natid;     // [n_slices * cpw]
    std::vector<int32_t> wave_slice;      // [n_waves] slice id or -1 (XCD-aware order)
    std::vector<int32_t> cptr;            // [n_major + 1] natural-chunk ranges per major
    BigVec<int32_t> order;                // [nnz] sorted position -> position in the caller's COO
    std::vector<int64_t> mptr;            // [n_major + 1] sorted-position ranges per major
};

// major/minor: int32 indices (already validated), val: float counts.
// keep_order: also fill `order`/`mptr` (needed for the t=0 responsibilities upload).
void build_sweep_plan(int64_t nnz, const int32_t *major, const int32_t *minor, const float *val,
                      int n_major, int n_minor, int lpc, int chunk_len, int n_windows,
                      bool keep_order, SweepPlanHost &out);

// ---------------------------------------------------------------------------------------
// Tile plan: the layout for the LDS-staged sweep.
//
// The gather plan above pays one or two 128-byte L2->L1 line fills per nonzero (measured: the
// texture-addresser / L1 fill path is the limiter, profiles/r01).  The tile plan removes that
// traffic: the minor table is cut into WINDOWS of `win_rows` rows that fit in LDS; a workgroup
// owns a BLOCK of `gpb` major rows (one per lane group, rows sorted by length so that a wave's
// groups are balanced) and walks a range of windows (a TASK): stage the window's table slice
// into LDS once, then every group streams its row's nonzeros of that window and gathers the
// minor K-vectors from LDS.  A row's accumulators stay in registers for the whole task, so one
// partial K-vector per (row, task) leaves the kernel.
//
//   entries   sliced-ELL per (block, wave, window): step-major, one slot per group of the wave
//             holding two nonzeros: uint4 {local0, val0, local1, val1} or, when all counts fit
//             16 bits, packed uint2 {local0 | local1 << 16, cnt0 | cnt1 << 16};
//             local = minor - window * win_rows
//   steps     [ (block * wpb + wave) * n_windows + window ]  uint16 steps of that sub-slice
//   task_*    block, first window, end window of each task; tasks are ordered window-range
//             major / block minor so that concurrently running workgroups stage the same slice
//   task_wave_off  [task * wpb + wave] offset (uint4 units) of the wave's entries at the task's
//             first window
//   partial row of (task t, group g) = t * gpb + g;  a major row's partial rows are
//             pfirst[row] + j * pstride, j < pcount[row]
//
// An entry's index field is the LDS position of its minor row in 16-byte units (off16), so the
// kernel addresses the row as lds + 16 * off16 whatever the mode.
//
// Two schedules share this layout:
//
//  * WINDOW mode (ring == 1): the whole LDS is one window of `win_rows` minor rows; per
//    (block, wave, window) the steps are the longest row segment of the wave, and all waves meet
//    at a barrier per window.  A row's nonzeros per window are Poisson-distributed, so the
//    workgroup moves at the pace of the fullest of its rows in every window (measured at
//    BASELINE C3: 0.68 of the slots carry nonzeros).  Kept for sparse x wide problems where a
//    slot would hold a nonzero or two per row.
//
//  * HALF-WINDOW schedule (ring >= 2, sync_stage = 1): the LDS is `ring` slots of slot16 * 16 bytes; a
//    "window" is now a SUB-window of win_rows = slot rows, sub-window s lives in slot s mod ring.  The
//    workgroup advances in EPOCHS, one per sub-window; all slots are readable during an epoch and are
//    refilled AT the epoch boundary by the window schedule's own exposed copy (only the slot the
//    last epoch's own sub-window had).  A row that has finished the epoch's own sub-window works
//    ahead in up to ring - 1 slots (never beyond its task); every wave has its own step count per
//    epoch.  With two slots of half a window the workgroup still stages exactly the bytes of the
//    window schedule, meets at twice as many barriers, and the kernel is the window kernel with a
//    different staging range; 0.87 of the slots carry nonzeros at C3 against 0.76.
//
//  * `single` (round 5): the steps of a (block, wave, window) count NONZEROS, not pairs -- an odd count does
//    not execute the second half of its last step slot.  Storage is unchanged (two nonzeros per slot,
//    tile_stored_steps slots per window); for the kernels that take one nonzero at a time (wide rows).
//
// (Round 2 also built an ASYNCHRONOUS ring -- five small slots, copies under the compute, no barrier --
// which was 8-13 % slower, and round 5 DOUBLE-BUFFERED sub-windows -- two slots, the next sub-window copied under the
// steps by LDS-DMA issued from inline assembly, balanced, one barrier per sub-window -- 10-15 % slower than the
// schedules above at C3 and at K = 50; neither is in the product sources any more: profiles/HISTORY.md,
// profiles/r05/ab_double_buffered_windows.txt, commit 3fb3871.)
struct TilePlanHost {
    int n_major = 0, n_minor = 0;
    int lpc = 4, gpw = 16, wpb = 8, gpb = 128;   // lanes/group, groups/wave, waves/block, groups/block
    int win_rows = 0, n_windows = 0, windows_per_task = 0;   // windows_per_task: the uniform cut the ranges derive from
    std::vector<int32_t> range_start;     // [n_ranges + 1] window ranges (tasks per block), tile_range_starts
    std::vector<int32_t> range_end_of_window;   // [n_windows] end of the range a window belongs to
    int ring = 1, slot16 = 0;             // half-window schedule: slots in the LDS, 16-byte units per slot
    int look = 0;                         // ... sub-windows beyond the epoch's own a row may work ahead in (ring - 1)
    int sync_stage = 0;                   // ... 1 whenever ring > 1 (slots refilled AT the epoch boundary)
    bool single = false;                  // steps count nonzeros (stored: (steps + 1) / 2 slots), see above
    int row_slots = 0;                    // 16-byte units per table row (KP * sizeof(T) / 16)
    int64_t nnz = 0, n_blocks = 0, n_tasks = 0, n_partial_rows = 0, pstride = 0;
    bool packed = false;                  // 8-byte entries {idx0|idx1<<16, cnt0|cnt1<<16} instead of 16-byte
    BigVec<uint32_t> entries;
    std::vector<uint16_t> steps;
    std::vector<int32_t> block_rows;      // [n_blocks * gpb] major id or -1
    std::vector<int32_t> task_block, task_w0, task_w1;
    std::vector<int64_t> task_wave_off;
    std::vector<int64_t> task_work;    // [task] wave-steps the task's workgroup sits through (barrier-limited)
    std::vector<int32_t> task_order;   // tasks by decreasing work (launch order of the merged cell+gene launch)
    std::vector<int32_t> pfirst, pcount;  // [n_major]
    BigVec<int32_t> order;                // [nnz] (major, minor)-sorted position -> caller's COO position
    std::vector<int64_t> mptr;            // [n_major + 1]
};

// Shape of a tile plan.  row_slots: 16-byte units per table row (KP * sizeof(T) / 16).
// ring <= 1: window mode with win_rows rows per window.  ring >= 2 (with sync_stage = 1): half-window
// schedule, slot_bytes per slot (win_rows is then derived: slot_bytes / row bytes).  bank_order: deal the nonzeros of a
// segment to the steps in an LDS-bank-aware order -- 0: minor order; 1: every row on its own (plan.cpp bank_order);
// 2: the rows that share an LDS pass jointly (plan.cpp bank_order_joint).
struct TileShape {
    int lpc = 1, waves_per_block = 16, win_rows = 1, target_tasks = 0, row_slots = 1;
    int ring = 1, slot_bytes = 0;
    int sync_stage = 0;        // 1 with ring >= 2: the HALF-WINDOW schedule above
    bool single = false;       // steps count nonzeros (not with the work-ahead of sync_stage 1)
    int slots = 0;          // workgroups of this orientation the GPU runs at once (0: unknown); see tile_plan_begin
    int ranges = 0;         // > 0: window ranges (tasks) per block, fixed by the caller (choose_task_ranges);
                            // target_tasks and the rounding by `slots` are then not consulted
    int bank_order = 1;
    bool allow_packed = true;
    double taper = 0.0;     // > 0: window ranges of unequal length, (1 + taper) .. (1 - taper) x the mean (tile_range_starts)
};
// How many window ranges (tasks per block) each orientation of a ONE-LAUNCH iteration should have.  The
// merged launch runs its tasks longest first on `resident` workgroups that draw from one list, so the
// launch lasts as long as a list schedule of the two task pools: with few tasks per workgroup the
// leftover of the last round costs up to a whole task (C3 f32: 294 + 260 tasks on 256 workgroups ran
// 1.37 x the balanced time), with many the per-task costs (a first window, table rows, partial rows that
// the update kernel has to sum) take over.  Model, per orientation s: blocks[s] x ranges tasks, each of
// ceil(windows / ranges) windows (the last range shorter), a window of block b = its share of nnz / windows
// at `nnz_per_second` per workgroup, + task_seconds per task; + partial_seconds[s] per range for the
// partial rows.  `windows` are counted in the unit the schedule would use (half windows where
// half_ok[s] and a task keeps >= min_half_per_task of them; whole windows cost window_penalty more work).
// Exhaustive over 1..max_ranges for both; returns the pair with the shortest modelled iteration.
// separate_launches: the two-launch iteration of a row shard -- each orientation scheduled on its own.
struct RangeChoice { int ranges[2]; bool half[2]; double seconds; };
// Shares of the nonzeros per block of `rows_per_block` rows taken in order of decreasing length, estimated
// from every `stride`-th index of the COO (threaded histogram + counting sort).
std::vector<double> block_shares(int64_t nnz, const int32_t *major, int n_major, int rows_per_block, int64_t stride);
// block_share[s][b] = share of the nonzeros that block b of orientation s holds (rows go to blocks by
// decreasing length, so a skewed matrix has a few heavy blocks whose tasks decide the tail; empty = uniform).
RangeChoice choose_task_ranges(const int64_t blocks[2], const int64_t half_windows[2], const bool half_ok[2],
                               const std::vector<double> block_share[2],
                               double nnz, int resident, double nnz_per_second, double task_seconds,
                               const double partial_seconds[2], int min_half_per_task, double window_penalty,
                               int max_ranges, bool separate_launches, double taper = 0.0);

// Launch order (slot -> task) of the tile sweep over one or two plans' tasks that keeps the tasks
// reading the SAME window range of the minor table on ONE XCD at the same time: an XCD's 32 compute
// units then stage the same table rows within a short time of each other and all but the first copy
// of a window hit the XCD's own 4 MiB L2 (the tables do not fit it: C5 share 11-56 MB; measured there
// 48 % L2 hits with the longest-first order).  Workgroup s is observed to run on XCD s % n_xcd
// (MI355X_MICROARCH.md, dispatch): tasks of one (plan, range) are cut into bundles of `bundle` (the
// workgroups an XCD holds at a time), bundles go longest first to the XCD with the least work so far
// and XCD x's queue fills the slots x, x + n_xcd, ...  A wrong placement guess is slower, never wrong.
// order[slot] = task of plans[0], or ~task of plans[1].
void xcd_launch_order(const TilePlanHost *const *plans, int n_plans, int n_xcd, int bundle,
                      std::vector<int32_t> &order);

void build_tile_plan(int64_t nnz, const int32_t *major, const int32_t *minor, const float *val,
                     int n_major, int n_minor, const TileShape &shape, bool keep_order, TilePlanHost &out);

// Step slots a (block, wave, window) occupies in the entry stream
inline int64_t tile_stored_steps(const TilePlanHost &P, int steps) { return P.single ? ((int64_t)steps + 1) / 2 : (int64_t)steps; }

// LDS position (16-byte units) of minor row m: shared by both builders and the test hook
inline uint32_t tile_off16(const TilePlanHost &P, int32_t m)
{
    const int32_t w = m / P.win_rows;
    const uint32_t r = (uint32_t)(m - w * P.win_rows) * (uint32_t)P.row_slots;
    return P.ring > 1 ? (uint32_t)(w % P.ring) * (uint32_t)P.slot16 + r : r;
}

// positions sorted by (major, minor) and the per-major run pointers
void sort_by_major_minor(int64_t nnz, const int32_t *major, const int32_t *minor, int n_major, int n_minor,
                         BigVec<int32_t> &order, std::vector<int64_t> &mptr);

// number of host threads used by the builders ($SCHPF_HOST_THREADS, default min(cores, 32))
int host_threads();

// Stable counting sort of positions by key: order[j] = original position of the j-th
// smallest key; ptr[k]..ptr[k+1] is the run of key k.
void counting_sort_positions(int64_t n, const int32_t *key, int nkeys, BigVec<int32_t> &order,
                             std::vector<int64_t> &ptr);

// pieces of build_tile_plan that do not touch the nonzeros (shared with the device builder)
void tile_plan_begin(TilePlanHost &P, int64_t nnz, int n_major, int n_minor, const TileShape &shape,
                     const int64_t *mptr);
int64_t tile_plan_offsets(TilePlanHost &P, std::vector<int64_t> &wave_off);
std::vector<int> tile_pass_rank(int lpc, int gpw);
// the LDS pass (0..3) the lane group sits in, beside its rank inside that pass (tile_pass_rank)
std::vector<int> tile_pass_of(int lpc, int gpw);
// Window ranges of one block.  W windows in ranges of `wpt` (the last one shorter): n = ceil(W / wpt) ranges.
// taper = 0, or fewer than six ranges: exactly that.  taper > 0: the same NUMBER of ranges, their lengths falling linearly from (1 + taper) to
// (1 - taper) times the mean: the merged launch runs its tasks longest first on workgroups that draw from one list, so
// the idle tail of the launch is about half the length of the LAST tasks started -- short ranges at the end of the
// list fill the gaps the long ones leave (round 3: the tail was 10 % of the CU-time of the f32 launch, 3.7 % in f64).
std::vector<int32_t> tile_range_starts(int W, int64_t wpt, double taper);
// ... and only where the orientation has tasks for more than a round and a half of the workgroups it gets (`slots`,
// TileShape::slots): in a single round unequal tasks ARE the imbalance (measured: 10k x 5k -10 %, a 1/8 shard of C3
// -7 % with tapered ranges; C3 f32 +3.4 %, half of C3 +2 %, profiles/r03/ab_tapered_ranges.txt)
inline bool tile_taper_applies(int64_t n_tasks, int slots) { return slots > 0 && 2 * n_tasks >= 3 * (int64_t)slots; }
// joint bank assignment (bank_order = 2): segments longer than this are dealt on their own (bank_order = 1 rule)
constexpr int TILE_JOINT_MAX = 128;
// The pick of one lane group at position t (both builders): counts per class cnt[0..n_classes), classes already
// read by its pass in `taken`; the fullest class that is still free, ties to the class nearest (upwards, cyclic)
// to the wish (rank + t) mod n_classes; when every class the group still has is taken, the fullest one.
template <typename Count>
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int tile_joint_pick(const Count *cnt, int stride, int n_classes, unsigned wish, unsigned taken)
{
    unsigned best_free = 0, best_any = 0;
    for (int c = 0; c < n_classes; ++c) {
        const unsigned n = (unsigned)cnt[(size_t)c * stride];
        const unsigned d = ((unsigned)c - wish) & (unsigned)(n_classes - 1);
        const unsigned key = n ? (n << 4) | (15u - d) : 0u;
        best_any = key > best_any ? key : best_any;
        const unsigned fk = ((taken >> c) & 1u) ? 0u : key;
        best_free = fk > best_free ? fk : best_free;
    }
    const unsigned key = best_free ? best_free : best_any;
    return (int)((wish + (15u - (key & 15u))) & (unsigned)(n_classes - 1));
}
void tile_plan_report(const TilePlanHost &P);

// ---------------------------------------------------------------------------------------
// BALANCED WINDOWS (round 4).  With windows cut by minor index, a row's nonzeros per window are Poisson
// distributed and every (wave, window) lasts as long as its fullest row: 0.56 of the executed step slots
// carry nonzeros at the C5 share (7 per row and window), 0.76 at C3 -- the padding executes every
// instruction of a nonzero.  But WHICH minor rows share a window is free, per block: the window is
// staged by LDS-DMA with one address per lane, so a window may be any `win_rows` rows of the table.
// For every block the minor rows are therefore dealt to the windows greedily -- in minor order, each to
// the window where the fullest of the block's rows that hold it stays lowest (ties: smallest sum of
// those rows' loads, then the lowest window) -- inside SECTIONS of at most 32 consecutive windows (the
// staging of a window then reads a bounded stretch of the table, and the sections are independent work
// for the builders).  Minor rows no row of the block holds fill the capacity left, in order.  A
// simulation of the rule gives 0.77-0.82 at the C5 share and 0.90-0.92 at C3 with WHOLE windows (the
// half-window schedule reaches 0.87 there with twice the barriers).
//
// Expressed as a renumbering: block b sees minor row m as VIRTUAL row virt_b(m) = window * win_rows +
// position; both builders run unchanged on the virtual indices (n_minor := n_virtual = n_windows *
// win_rows) and the kernel stages window w of block b from the rows minor_of[b * n_virtual + w * win_rows
// + j] (-1: no row).  Needs ring <= 1 (whole windows).
struct BalanceGeometry {
    int gpb = 0, win_rows = 0, n_windows = 0, n_sections = 0, D = 0;   // D windows per section (the last may have fewer)
    int64_t n_blocks = 0;
    int n_virtual = 0;
};
// windows per section: at most 32 (64 = the device builder's lanes was measured: slot fill 0.81 -> 0.82, the same sweep
// time, and a third longer to build -- the greedy is sequential inside a section), and the builders' load matrix
// [gpb][D] of 16-bit counts within 60 KB
inline void balance_sections(int n_windows, int gpb, int &n_sections, int &D)
{
    const int dmax = std::max(1, std::min(32, 30000 / std::max(gpb, 1)));
    n_sections = std::max(1, (n_windows + dmax - 1) / dmax);
    D = (n_windows + n_sections - 1) / n_sections;
    n_sections = (n_windows + D - 1) / D;
}
// key of a nonzero for the balancing pass: (block, minor, lane group of its major row)
constexpr int BALANCE_GROUP_BITS = 10;
// cost of putting a minor row into a window: (fullest of its rows there, sum of its rows' loads, window)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint64_t balance_cost(unsigned mx, unsigned sm, unsigned c)
{
    return ((uint64_t)mx << 40) | ((uint64_t)sm << 8) | (uint64_t)c;
}
// host builder: vminor[j] = virtual minor of nonzero j (caller's COO order), minor_of as above
void balance_windows_host(int64_t nnz, const int32_t *major, const int32_t *minor, int n_major, int n_minor,
                          const TileShape &shape, BigVec<int32_t> &vminor, std::vector<int32_t> &minor_of,
                          BalanceGeometry &geo);
// device builder (plan_device.hip): d_vminor [nnz] is written; *d_minor_of is allocated (hipMalloc, the caller frees)
void balance_windows_device(void *stream, int64_t nnz, const int32_t *d_major, const int32_t *d_minor, int n_major,
                            int n_minor, const TileShape &shape, int32_t *d_vminor, void **d_minor_of,
                            BalanceGeometry &geo);

// the (major, minor) / (minor, major) order of a host COO (one threaded scan)
void coo_order_flags(int64_t nnz, const int32_t *major, const int32_t *minor, bool &sorted_major_minor,
                     bool &sorted_minor_major);

// plan_device.hip: the same plan built by device passes over an uploaded COO (hipStream_t is a
// pointer type; declared as void * here so that host-only translation units need no HIP header)
void build_tile_plan_device(void *stream, int64_t nnz, const int32_t *d_major, const int32_t *d_minor,
                            const float *d_val, bool presorted, bool packed_ok, int n_major, int n_minor,
                            const TileShape &shape, TilePlanHost &P, void **out_entries, size_t *out_entries_bytes,
                            void **out_steps, void **out_order);

}  // namespace schpf
