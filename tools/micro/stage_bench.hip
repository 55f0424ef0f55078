// Microbenchmark (GPU box): how fast can a 1024-thread workgroup per CU fill its 152 KiB LDS window
// from an L2/MALL-resident table, (a) global_load_lds DMA, (b) global_load_dwordx4 + ds_write_b128.
//   hipcc -O3 --offload-arch=gfx950 stage_bench.hip -o stage_bench && ./stage_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void stage_kernel(const unsigned char *__restrict__ tab, size_t tab_bytes, int win_bytes,
                                                     int n_win, int stride_win, float *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    float acc = 0.f;
    const size_t n_tab_win = tab_bytes / (size_t)win_bytes;
    for (int w = 0; w < n_win; ++w) {
        const size_t wi = ((size_t)blockIdx.x * stride_win + w) % n_tab_win;
        const unsigned char *src = tab + wi * (size_t)win_bytes;
        __syncthreads();
        if (MODE == 0) {
            for (int off = wv * 1024; off < win_bytes; off += wpb * 1024)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(lds + off), 16, 0, 0);
        } else {
            // registers: the wave's 1 KiB pieces as unconditional 16-byte loads (all in flight), then
            // ds_write_b128; the pieces beyond the common count under a wave-uniform branch
            const int n_piece = win_bytes / 1024;                 // 152
            const int common = n_piece / wpb;                     // 9 for 16 waves
            if (MODE == 1) {
                uint4 r[9];
#pragma unroll
                for (int i = 0; i < 9; ++i)
                    if (i < common) r[i] = *reinterpret_cast<const uint4 *>(src + (size_t)(wv + i * wpb) * 1024 + lane * 16);
                uint4 extra = make_uint4(0, 0, 0, 0);
                const bool has_extra = wv + common * wpb < n_piece;
                if (has_extra) extra = *reinterpret_cast<const uint4 *>(src + (size_t)(wv + common * wpb) * 1024 + lane * 16);
#pragma unroll
                for (int i = 0; i < 9; ++i)
                    if (i < common) *reinterpret_cast<uint4 *>(lds + (wv + i * wpb) * 1024 + lane * 16) = r[i];
                if (has_extra) *reinterpret_cast<uint4 *>(lds + (wv + common * wpb) * 1024 + lane * 16) = extra;
            } else {
                // two batches of 5 (fewer registers)
                for (int b = 0; b < 2; ++b) {
                    uint4 r[5];
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        const int piece = wv + (b * 5 + i) * wpb;
                        r[i] = piece < n_piece ? *reinterpret_cast<const uint4 *>(src + (size_t)piece * 1024 + lane * 16) : make_uint4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        const int piece = wv + (b * 5 + i) * wpb;
                        if (piece < n_piece) *reinterpret_cast<uint4 *>(lds + piece * 1024 + lane * 16) = r[i];
                    }
                }
            }
        }
        __syncthreads();
        acc += *reinterpret_cast<const float *>(lds + ((threadIdx.x * 16 + w * 4) % win_bytes));
    }
    if (acc == 123.456f) out[0] = acc;
}

int main()
{
    const int win_bytes = 152 * 1024;
    const size_t tab_bytes = (size_t)win_bytes * 73;     // 11 MB: the gene table of the C5 share
    unsigned char *tab; float *out;
    CK(hipMalloc(&tab, tab_bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(tab, 1, tab_bytes));
    CK(hipFuncSetAttribute((const void *)stage_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)stage_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)stage_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int stride : {0, 1, 7}) for (int mode = 0; mode < 3; ++mode) for (int blocks : {1024}) {
        const int n_win = 64;
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(stage_kernel<0>, dim3(blocks), dim3(1024), win_bytes, 0, tab, tab_bytes, win_bytes, n_win, stride, out);
            else if (mode == 2) hipLaunchKernelGGL(stage_kernel<2>, dim3(blocks), dim3(1024), win_bytes, 0, tab, tab_bytes, win_bytes, n_win, stride, out);
            else hipLaunchKernelGGL(stage_kernel<1>, dim3(blocks), dim3(1024), win_bytes, 0, tab, tab_bytes, win_bytes, n_win, stride, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double bytes = (double)blocks * n_win * win_bytes;
        const double per_win_us = best * 1e3 / (n_win * (blocks / 256.0));
        printf("stride %d mode %s blocks %4d: %.3f ms  %.2f TB/s  %.2f us per window per CU  %.1f B/clk/CU (2.4 GHz)\n", stride,
               mode == 0 ? "dma  " : mode == 1 ? "regs9" : "regs5", blocks, best, bytes / best / 1e9, per_win_us, win_bytes / (per_win_us * 2400.0));
    }
    return 0;
}
