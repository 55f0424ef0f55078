// Kernel argument blocks and host-callable launchers (implemented in kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace schpf {

enum { MODE_PHI = 0, MODE_LLH = 1, MODE_RANDOM = 2 };
enum { SRC_STRIDED = 3 };
enum { SRC_NONE = 0, SRC_PARTIALS = 1, SRC_DENSE = 2 };

template <typename T> struct SweepArgs {
    const uint4 *entries;       // sliced-ELL nonzeros {minor0, val0, minor1, val1}
    const int64_t *slice_off;   // [n_slices] in uint4 units
    const int *slice_steps;     // [n_slices]
    const int *chunk_major;     // [n_slices * CPW]
    const int *chunk_natid;     // [n_slices * CPW]
    const int *wave_slice;      // [n_waves]
    const T *tab_major;         // [n_major, KP] exp-shifted E[log] (PHI) or E[x] (LLH)
    const T *tab_minor;         // [n_minor, KP]
    const T *log_major;         // [n_major, KP] E[log x] (fallback only)
    const T *log_minor;         // [n_minor, KP]
    T *partials;                // [n_chunks, KP]
    double *wave_out;           // [n_waves] (LLH)
    int K;
};

// LDS-staged sweep over a tile plan (plan.h, TilePlanHost)
template <typename T> struct TileArgs {
    const void *entries;           // uint4 {off16_0, val0, off16_1, val1} or packed uint2 (plan.h)
    const uint16_t *steps;         // [(block * wpb + wave) * n_windows + window]
    const int *block_rows;         // [n_blocks * gpb]
    const int *task_block, *task_w0, *task_w1;
    // loss pass over SUB-ranges of the tasks (capi.hip loss_tasks): where the sub-task's parent task ends -- entries of
    // the half-window schedule may point one sub-window beyond the sub-task, and that one is staged too; nullptr: task_w1
    const int *task_stage_end;
    const int64_t *task_wave_off;  // [task * wpb + wave] first uint4 of the wave's entries in the task
    const int *task_order;         // [launch slot] -> task (longest first); nullptr = identity
    const T *tab_major;            // [n_major, KP]
    const T *tab_minor;            // [n_minor, KP]  (staged window by window)
    const T *log_major, *log_minor;
    T *partials;                   // [n_tasks * gpb, KP]
    double *wave_out;              // [n_tasks * wpb] (LLH)
    int K, n_minor, n_windows, win_rows, wpb;
    // balanced windows (plan.h): block b stages window w from the table rows minor_of[b * n_virtual + w * win_rows + j]
    // (-1: none); n_minor is then n_virtual.  nullptr: windows are index ranges of the table
    const int *minor_of;
    int n_virtual;
    int llh_tab_off;               // MODE_LLH: byte offset in LDS of the logarithm table (behind the window; LlhAccumulator::TABLE_BYTES)
    int ring, slot_bytes;          // ring mode (plan.h): slots in the LDS ring (<= 1: window mode), bytes per slot
    int sync_stage;                // ring mode: half-window schedule (slots refilled at the epoch boundary)
    int single;                    // steps[] count nonzeros, (steps + 1) / 2 slots are stored (plan.h)
    uint64_t seed;                 // MODE_RANDOM
    int major_is_cell;
    // persistent launch (sweep_impl.h tile_sweep_kernel): `resident` workgroups draw the n_tasks slots of
    // task_order from queue[0]; queue == nullptr: one workgroup per task
    int *queue;
    int n_tasks, resident;
    // shader-clock probe (capi.hip profile_clock): workgroup 0 adds the shader cycles (s_memtime) and the constant-rate
    // ticks (s_memrealtime) of its stay in the launch to probe[0], probe[1], and one to probe[4]; probe[2..3] hold its
    // start stamps.  nullptr: off.  The dual launch reads the cell-side argument block's pointer
    unsigned long long *clock_probe;
};

// Fixed-order sum of n values `stride` apart, four loads in flight (the partial rows of one
// major row live far apart in HBM/L2; a rolled loop would pay one memory latency per term).
template <typename T> __device__ __forceinline__ double sum_strided(const T *__restrict__ p, int n, size_t stride)
{
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int c = 0;
    for (; c + 4 <= n; c += 4) {
        const T v0 = p[0], v1 = p[stride], v2 = p[2 * stride], v3 = p[3 * stride];
        s0 += (double)v0; s1 += (double)v1; s2 += (double)v2; s3 += (double)v3;
        p += 4 * stride;
    }
    for (; c < n; ++c, p += stride) s0 += (double)*p;
    return (s0 + s1) + (s2 + s3);
}

template <typename T> struct UpdateArgs {
    int n, K, KP, rows_per_block;
    const T *partials;          // SRC_PARTIALS: [n_chunks, KP]
    const int *cptr;            //               [n + 1]
    const T *dense;             // SRC_DENSE:    [n, K]
    const int *pfirst, *pcount; // SRC_STRIDED:  partial rows pfirst[row] + j * pstride, j < pcount[row]
    int64_t pstride;
    double prior_shape;         // a or c
    const T *cap_shape;         // xi / eta shape [n]
    const T *cap_rate;          // xi / eta rate BEFORE this update [n]
    const double *s_other;      // [K] sum over the other loading of E[x] ...
    const T *s_other_t;         // ... or (non-null) the same in the model dtype: the all-reduced tail of the exchange buffer
    const double *s_other_part; // ... or (s_other_nb > 0) its per-block partials [s_other_nb, K], summed here:
    int s_other_nb;             //     small problems skip the separate reduce launch (capi.hip fuse_sums)
    double cap_prior_rate;      // bp or dp
    T *shape, *rate;            // [n, K] in/out
    T *cap_rate_out;            // [n]
    T *tab_e, *tab_log, *tab_exp;  // [n, KP]
    double *colsum_part;        // [nblocks, K]
};

// (vectors per lane, lanes per row) pairs the sweeps are instantiated for (sweep_impl.h SCHPF_DISPATCH*)
inline bool tile_combo_ok(int nv, int lpc)
{
    if (lpc == 1) return nv >= 1 && nv <= 7;
    if (lpc == 2 || lpc == 4 || lpc == 8) return nv >= 4 && nv <= 7;
    return lpc == 16 && nv == 4;
}
inline bool gather_combo_ok(int nv, int lpc)
{
    if (lpc == 4) return (nv >= 1 && nv <= 8) || nv == 10;
    if (lpc == 8) return nv == 6 || nv == 7 || nv == 8 || nv == 10;
    return lpc == 16 && nv >= 6 && nv <= 8;
}

template <typename T>
hipError_t launch_sweep(const SweepArgs<T> &a, int nv, int lpc, int mode, int64_t n_waves, hipStream_t st);
template <typename T>
hipError_t launch_random_phi(const SweepArgs<T> &a, int nv, int lpc, uint64_t seed, int major_is_cell,
                             int64_t n_waves, hipStream_t st);
template <typename T>
hipError_t launch_tile_sweep(const TileArgs<T> &a, int nv, int lpc, int mode, int packed, int64_t n_tasks,
                             int threads, size_t lds_bytes, hipStream_t st);
// cell-side (a0) and gene-side (a1) MODE_PHI sweeps in one launch; order[slot] = task | ~task.
// queue != nullptr (two zeroed ints): `resident` persistent workgroups draw the slots from it
template <typename T>
hipError_t launch_tile_sweep_dual(const TileArgs<T> &a0, const TileArgs<T> &a1, const int *order, int nv, int lpc,
                                  int packed, int64_t n_slots, int threads, size_t lds_bytes, int *queue, int resident,
                                  hipStream_t st);
template <typename T> hipError_t launch_gamma_update(const UpdateArgs<T> &a, int src, int nblocks, hipStream_t st);
// rows a 256-thread block of the update kernel takes per group (UpdateArgs::rows_per_block must be this)
int update_rows_per_block(int K);
hipError_t launch_colsum_reduce(const double *part, int nblocks, int K, double *out, void *mirror,
                                int mirror_is_f32, hipStream_t st);
template <typename T>
hipError_t launch_combine_partials(const T *partials, const int *cptr, int n, int K, int KP, T *out,
                                   hipStream_t st);
template <typename T>
hipError_t launch_combine_strided(const T *partials, const int *pfirst, const int *pcount, int64_t pstride, int n,
                                  int K, int KP, T *out, hipStream_t st);
hipError_t launch_sum_doubles(const double *v, int64_t n, double *out, hipStream_t st);
// minibatch rows from a resident row-sorted copy (capi.hip keep_rows / upload_rows):
//   out_col/val[j] = col/val[order[j]]  (order == nullptr: identity copy)
hipError_t launch_gather_by_order(const int *order, const int *col, const float *val, int64_t nnz, int *out_col,
                                  float *out_val, hipStream_t st);
//   batch row i = source row rows[i]: its (col, val) run src_ptr[rows[i]] .. src_ptr[rows[i] + 1] goes to
//   dst_ptr[i] ..., with local row index i
hipError_t launch_gather_rows(const int *rows, int n_rows, const int64_t *src_ptr, const int *src_col,
                              const float *src_val, const int64_t *dst_ptr, int *out_row, int *out_col,
                              float *out_val, hipStream_t st);
hipError_t launch_gammaln_sum(const float *x, int64_t n, double *block_out, int nblocks, hipStream_t st);
template <typename T>
hipError_t launch_zero_rate_sum(const int *row, const int *col, int64_t n, const T *et, const T *eb, int K, int KP,
                                double *out, hipStream_t st);
template <typename T>
hipError_t launch_segment_sum(const double *xphi, const int *order, const int64_t *mptr, int n, int K, T *out,
                              hipStream_t st);

template <typename T> hipError_t launch_elog(const T *shape, const T *rate, int64_t n, T *out, hipStream_t st);
template <typename T> hipError_t launch_ratio(const T *shape, const T *rate, int64_t n, T *out, hipStream_t st);
template <typename T>
hipError_t launch_xphi_coo(const T *x, const int *row, const int *col, const T *elt, const T *elb, int64_t nnz,
                           int K, T *out, hipStream_t st);
template <typename T>
hipError_t launch_llh_coo(const T *x, const int *row, const int *col, const T *et, const T *eb, int64_t nnz,
                          int K, T *out, hipStream_t st);
template <typename T>
hipError_t launch_shape_update(const T *xphi, const int *order, const int64_t *ptr, int n, int K, double prior,
                               T *out, hipStream_t st);
template <typename T>
hipError_t launch_ratio_colsum(const T *shape, const T *rate, int m, int K, double *part, int nblocks,
                               hipStream_t st);
template <typename T>
hipError_t launch_rate_update(const T *ps, const T *pr, const double *S, int n, int K, T *out, hipStream_t st);
template <typename T>
hipError_t launch_capacity_rate(const T *shape, const T *rate, int n, int K, double prior, T *out,
                                hipStream_t st);
hipError_t launch_digamma_array(const double *x, int64_t n, double *out, hipStream_t st);
hipError_t launch_gammaln_array(const double *x, int64_t n, double *out, hipStream_t st);

}  // namespace schpf
