// Pricing the two hand-over mechanisms of a SINGLE-PASS sweep (3 K FMAs per nonzero instead of 4 K: the normaliser
// s_ig computed once, hpf_numba.py:97-112 feeding both scatter-adds :152-155) on gfx950, at the access pattern of the
// C3 f64 kernel (K = 20, two lanes per row, 10 doubles per lane, 160-byte rows, 1024-thread workgroup, one per CU):
//
//  (i)  gene-side accumulators resident in the LDS window, every lane adding its 10 doubles w * Et[i,k] with
//       ds_add_f64 to the row of the nonzero's gene (the row the Eb read of the same nonzero came from);
//  (ii) the cell-major pass writing x / s (8 bytes per nonzero) into the slot the gene-major pass will read it from:
//       scattered 8-byte stores inside a tile-sized region (the most favourable case: both orientations cut into the
//       same (block x window) tiles, so that a tile's values land in one ~128 KiB stretch that can stay in L2).
//
// Reported: ns per wave-level nonzero step (32 lane groups = 32 nonzeros) per SIMD at four waves per SIMD, against
// what the mechanism would REMOVE from the two-pass kernel: the gene side's normaliser = 10 FMAs + ~6 other VALU
// instructions per nonzero-side at 2.16 ns each (profiles/r03/valu_bench.txt) = ~35 ns, and its 5 ds_read_b128.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/handover_bench.hip -o tools/micro/handover_bench && tools/micro/handover_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ROW_BYTES = 160;     // K = 20 doubles
constexpr int WIN_ROWS = 486;      // half of the 152 KiB: one half Eb rows, the other half the accumulator rows
constexpr int STEPS = 2048;

// per-lane pseudo-random row sequence, the same for the two lanes of a group
__device__ __forceinline__ unsigned next_row(unsigned &state)
{
    state = state * 1664525u + 1013904223u;
    return (state >> 8) % WIN_ROWS;
}

enum { READ_ONLY, ATOMIC_ONLY, READ_FMA_ATOMIC, READ_FMA_TWO_PASS };

template <int MODE> __global__ __launch_bounds__(1024) void lds_kernel(double *out, long long *cyc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    double *tab = reinterpret_cast<double *>(lds);                               // [WIN_ROWS][20] Eb rows
    double *accw = reinterpret_cast<double *>(lds + WIN_ROWS * ROW_BYTES);       // [WIN_ROWS][20] gene accumulators
    for (int i = threadIdx.x; i < WIN_ROWS * 20; i += blockDim.x) { tab[i] = 1.0 + 1e-6 * i; accw[i] = 0.0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, sub = lane & 1;
    unsigned state = (blockIdx.x * 1024u + (threadIdx.x >> 1)) * 2654435761u + 12345u;
    double tm[10], acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) { tm[k] = 0.5 + 0.01 * k + 1e-4 * sub; acc[k] = 0.0; }
    const long long t0 = clock64();
    for (int p = 0; p < STEPS; ++p) {
        const unsigned r = next_row(state);
        const double2 *row = reinterpret_cast<const double2 *>(tab + r * 20) + sub;
        double b[10];
        if (MODE != ATOMIC_ONLY) {
#pragma unroll
            for (int q = 0; q < 5; ++q) { const double2 v = row[q * 2]; b[2 * q] = v.x; b[2 * q + 1] = v.y; }
        } else {
#pragma unroll
            for (int k = 0; k < 10; ++k) b[k] = tm[k];
        }
        double w = 1.0;
        if (MODE == READ_FMA_ATOMIC || MODE == READ_FMA_TWO_PASS) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < 10; k += 2) { s0 = fma(tm[k], b[k], s0); s1 = fma(tm[k + 1], b[k + 1], s1); }
            double s = s0 + s1;
            const int lo = __builtin_amdgcn_mov_dpp(__double2loint(s), 0xB1, 0xF, 0xF, true);
            const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(s), 0xB1, 0xF, 0xF, true);
            s += __hiloint2double(hi, lo);
            double rc = __builtin_amdgcn_rcp(s);
            rc = fma(fma(-s, rc, 1.0), rc, rc);
            w = 3.0 * rc;
#pragma unroll
            for (int k = 0; k < 10; ++k) acc[k] = fma(w, b[k], acc[k]);
        }
        if (MODE == ATOMIC_ONLY || MODE == READ_FMA_ATOMIC) {
            // the gene side of the same nonzero: accG[g][k] += w * Et[i][k], this lane's 10 factors (16-byte vectors
            // q * 2 + sub of the row, like the reads)
            double *arow = accw + r * 20 + sub * 2;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                __hip_atomic_fetch_add(arow + q * 4, w * tm[2 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(arow + q * 4 + 1, w * tm[2 * q + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (MODE == READ_ONLY) {
#pragma unroll
            for (int k = 0; k < 10; ++k) acc[k] += b[k];
        }
    }
    const long long t1 = clock64();
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 10; ++k) s += acc[k];
    if (MODE == ATOMIC_ONLY || MODE == READ_FMA_ATOMIC) s += accw[threadIdx.x % (WIN_ROWS * 20)];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// (ii) scattered 8-byte stores: every workgroup owns a tile region of `region_bytes`; per step the first lane of
// each group stores one double at a pseudo-random 8-byte slot of the region (32 stores per wave instruction)
__global__ __launch_bounds__(1024) void scatter_kernel(double *buf, size_t region_doubles, int steps, int tiles_per_wg)
{
    const int lane = threadIdx.x & 63;
    unsigned state = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 777u;
    for (int t = 0; t < tiles_per_wg; ++t) {
        double *region = buf + ((size_t)blockIdx.x * tiles_per_wg + t) * region_doubles;
        for (int p = 0; p < steps; ++p) {
            state = state * 1664525u + 1013904223u;
            const size_t slot = (state >> 6) % region_doubles;
            if ((lane & 1) == 0) __builtin_nontemporal_store(1.0 + p, region + slot);
        }
    }
}
// reference: the same number of bytes written as coalesced 8-byte-per-lane streams
__global__ __launch_bounds__(1024) void stream_kernel(double *buf, size_t region_doubles, int steps, int tiles_per_wg)
{
    for (int t = 0; t < tiles_per_wg; ++t) {
        double *region = buf + ((size_t)blockIdx.x * tiles_per_wg + t) * region_doubles;
        for (int p = 0; p < steps / 2; ++p) {
            const size_t slot = ((size_t)p * 1024 + threadIdx.x) % region_doubles;
            __builtin_nontemporal_store(1.0 + p, region + slot);
        }
    }
}

template <int MODE> void run_lds(const char *name, int n_cu, double *out, long long *cyc)
{
    const size_t lds = 2 * WIN_ROWS * ROW_BYTES;
    CHK(hipFuncSetAttribute((const void *)lds_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(lds_kernel<MODE>, dim3(n_cu), dim3(1024), lds, 0, out, cyc);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<long long> h(n_cu);
    CHK(hipMemcpy(h.data(), cyc, n_cu * sizeof(long long), hipMemcpyDeviceToHost));
    // a SIMD runs 4 waves of the workgroup: wave steps per SIMD = 4 * STEPS
    printf("%-52s %8.3f ms   %7.1f ns per wave step and SIMD (4 waves per SIMD)\n", name, ms, ms * 1e6 / (4.0 * STEPS));
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    double *out; long long *cyc;
    CHK(hipMalloc(&out, (size_t)n_cu * 1024 * 8)); CHK(hipMalloc(&cyc, n_cu * sizeof(long long)));
    printf("(i) LDS mechanisms, one 1024-thread workgroup per CU, %d steps of 32 nonzeros per wave\n", STEPS);
    run_lds<READ_ONLY>("5 ds_read_b128 of a random row (+10 adds)", n_cu, out, cyc);
    run_lds<ATOMIC_ONLY>("10 ds_add_f64 into a random accumulator row (+10 muls)", n_cu, out, cyc);
    run_lds<READ_FMA_TWO_PASS>("one orientation of the two-pass step (reads, dot, rcp, acc)", n_cu, out, cyc);
    run_lds<READ_FMA_ATOMIC>("single-pass step (reads, dot, rcp, acc, 10 ds_add_f64)", n_cu, out, cyc);

    printf("(ii) x / s handed over through memory: 9.75e7 eight-byte values (C3), 32 stores per wave instruction\n");
    const size_t region_doubles = 128 * 1024 / 8;   // one (block, window) tile pair: ~12.4 k nonzeros at 0.76 fill
    const long long total = 97540251;
    const int tiles_per_wg = 32;
    // stores per tile: steps * 16 waves * 32 lanes
    const int steps = (int)(total / ((long long)n_cu * tiles_per_wg * 16 * 32));
    double *buf;
    CHK(hipMalloc(&buf, (size_t)n_cu * tiles_per_wg * region_doubles * 8));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(scatter_kernel, dim3(n_cu), dim3(1024), 0, 0, buf, region_doubles, steps, tiles_per_wg);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("scattered 8-byte stores inside 128 KiB tile regions (%d per tile): %.3f ms for %.2e stores\n", steps * 16 * 32, ms,
           (double)steps * 16 * 32 * tiles_per_wg * n_cu);
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream_kernel, dim3(n_cu), dim3(1024), 0, 0, buf, region_doubles, steps, tiles_per_wg);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("the same bytes as coalesced streams:                                  %.3f ms\n", ms);
    printf("what a single pass would save at C3 f64: ~0.15-0.20 ms of the 0.63 ms dual launch (the gene side's normalisers)\n");
    return 0;
}
