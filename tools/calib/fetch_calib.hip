// FETCH_SIZE calibration for the access shapes of the tile sweep (MI355X_MICROARCH.md, HBM: the
// counter reports 1/2 of a 16-byte-per-lane streaming read; other widths are uncalibrated).
//   hipcc --offload-arch=gfx950 -O3 tools/calib/fetch_calib.hip -o gpurun_out/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE -d out -- gpurun_out/fetch_calib
// Every kernel reads the same 1 GiB buffer once (cold: a 2 GiB scrub runs in between).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void read_b64_nt(const u32x2 *p, size_t n, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const u32x2 v = __builtin_nontemporal_load(p + i);
        acc += v.x ^ v.y;
    }
    if (acc == 0x12345678u) *out = acc;
}
__global__ void read_b128_nt(const u32x4 *p, size_t n, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = __builtin_nontemporal_load(p + i);
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *out = acc;
}
__global__ void read_b128(const uint4 *p, size_t n, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *out = acc;
}
// 1 KiB per wave instruction straight into LDS, as the window staging does
__global__ __launch_bounds__(1024) void read_lds_dma(const unsigned char *p, size_t bytes, unsigned *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t per_block = 64 * 1024;   // one 64 KiB tile per iteration of a block
    unsigned acc = 0;
    for (size_t base = blockIdx.x * per_block; base < bytes; base += (size_t)gridDim.x * per_block) {
        for (int off = wv * 1024; off < (int)per_block; off += 16 * 1024)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + base + off + lane * 16),
                                             (__attribute__((address_space(3))) void *)(lds + off), 16, 0, 0);
        __syncthreads();
        acc += reinterpret_cast<unsigned *>(lds)[threadIdx.x];
        __syncthreads();
    }
    if (acc == 0x12345678u) *out = acc;
}
__global__ void scrub(uint4 *p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint4((unsigned)i, 1, 2, 3);
}
int main()
{
    const size_t bytes = (size_t)1 << 30;
    unsigned char *buf, *junk; unsigned *out;
    hipMalloc(&buf, bytes); hipMalloc(&junk, 2 * bytes); hipMalloc(&out, 4);
    hipMemset(buf, 1, bytes);
    auto cold = [&] { hipLaunchKernelGGL(scrub, dim3(2048), dim3(256), 0, 0, (uint4 *)junk, 2 * bytes / 16); };
    cold(); hipLaunchKernelGGL(read_b64_nt, dim3(4096), dim3(256), 0, 0, (const u32x2 *)buf, bytes / 8, out);
    cold(); hipLaunchKernelGGL(read_b128_nt, dim3(4096), dim3(256), 0, 0, (const u32x4 *)buf, bytes / 16, out);
    cold(); hipLaunchKernelGGL(read_b128, dim3(4096), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, out);
    cold(); hipLaunchKernelGGL(read_lds_dma, dim3(1024), dim3(1024), 64 * 1024, 0, buf, bytes, out);
    hipDeviceSynchronize();
    printf("done: every read kernel touched %zu bytes once\n", bytes);
    return 0;
}
