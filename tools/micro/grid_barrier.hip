// What a device-wide barrier costs on gfx950 against what it would replace: the boundary between two dependent
// kernel nodes of a hipGraph.  Gate for the one-launch CAVI iteration of small problems (BASELINE C2: sweep ->
// beta/eta update -> theta/xi update = three graph nodes per iteration; a persistent kernel would have three
// device-wide barriers per iteration instead).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.hip -o tools/micro/grid_barrier && tools/micro/grid_barrier
// Every phase does the same token work in both forms: each thread reads a value another workgroup wrote in the
// previous phase, adds one and writes it where another workgroup will read it (so the barrier also has to make
// global writes visible across compute units and XCDs, like the iteration's partial rows and tables).
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void phase_work(const double *__restrict__ in, double *__restrict__ out, int nwg)
{
    const int src = (blockIdx.x + 1) % nwg;   // another workgroup's slot (another XCD for consecutive ids)
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = in[(size_t)src * blockDim.x + threadIdx.x] + 1.0;
}

// counter + generation word; agent-scope release before arriving, acquire after leaving
__device__ __forceinline__ void grid_barrier(unsigned *bar, unsigned nwg)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __atomic_thread_fence(__ATOMIC_RELEASE);   // agent scope (the HIP default for the builtin): L2 write-back
        const unsigned arrived = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (arrived == nwg) {
            __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(&bar[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

template <int KIND>   // 0: own barrier, 1: cooperative_groups grid.sync()
__global__ __launch_bounds__(256) void persistent(double *a, double *b, unsigned *bar, int iters, int nwg)
{
    extern __shared__ unsigned char lds[];
    (void)lds;
    for (int it = 0; it < iters; ++it) {
        for (int ph = 0; ph < 3; ++ph) {
            const bool flip = (it * 3 + ph) & 1;
            phase_work(flip ? b : a, flip ? a : b, nwg);
            if (KIND == 0) grid_barrier(bar, (unsigned)nwg);
            else cooperative_groups::this_grid().sync();
        }
    }
}

__global__ __launch_bounds__(256) void one_phase(const double *in, double *out, int nwg)
{
    extern __shared__ unsigned char lds[];
    (void)lds;
    phase_work(in, out, nwg);
}

int main(int argc, char **argv)
{
    const int threads = 256;
    const int lds = 64 * 1024;   // the C2 sweep's window: two workgroups per compute unit
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    CHK(hipFuncSetAttribute((const void *)persistent<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHK(hipFuncSetAttribute((const void *)persistent<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHK(hipFuncSetAttribute((const void *)one_phase, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int per_cu = 0;
    CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)persistent<0>, threads, lds));
    printf("device: %s, %d CUs; %d workgroups of %d threads with %d KiB LDS fit a CU\n", prop.name, prop.multiProcessorCount,
           per_cu, threads, lds / 1024);
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 200;
    for (int nwg : {prop.multiProcessorCount / 2, prop.multiProcessorCount, prop.multiProcessorCount * per_cu}) {
        double *a, *b;
        unsigned *bar;
        CHK(hipMalloc(&a, (size_t)nwg * threads * 8)); CHK(hipMalloc(&b, (size_t)nwg * threads * 8));
        CHK(hipMalloc(&bar, 64));
        CHK(hipMemset(a, 0, (size_t)nwg * threads * 8)); CHK(hipMemset(b, 0, (size_t)nwg * threads * 8));
        CHK(hipMemset(bar, 0, 64));
        float ms[3] = {0, 0, 0};
        // (1) own barrier, plain launch of a grid that fits the device at once
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(persistent<0>, dim3(nwg), dim3(threads), lds, st, a, b, bar, iters, nwg);
            CHK(hipEventRecord(e1, st));
            CHK(hipStreamSynchronize(st));
            CHK(hipEventElapsedTime(&ms[0], e0, e1));
        }
        std::vector<double> h((size_t)nwg * threads);
        CHK(hipMemcpy(h.data(), (iters * 3) & 1 ? b : a, h.size() * 8, hipMemcpyDeviceToHost));
        bool ok = true;
        for (double v : h) ok = ok && v == 3.0 * iters * 3;   // three launches of iters * 3 phases, each + 1
        // (2) cooperative launch + grid.sync()
        {
            void *args[] = {&a, &b, &bar, (void *)&iters, &nwg};
            for (int rep = 0; rep < 3; ++rep) {
                CHK(hipEventRecord(e0, st));
                CHK(hipLaunchCooperativeKernel((const void *)persistent<1>, dim3(nwg), dim3(threads), args, lds, st));
                CHK(hipEventRecord(e1, st));
                CHK(hipStreamSynchronize(st));
                CHK(hipEventElapsedTime(&ms[1], e0, e1));
            }
        }
        // (3) the same phases as 3 * iters kernel nodes of one hipGraph
        {
            hipGraph_t g; hipGraphExec_t ge;
            CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int p = 0; p < iters * 3; ++p)
                hipLaunchKernelGGL(one_phase, dim3(nwg), dim3(threads), lds, st, (p & 1) ? b : a, (p & 1) ? a : b, nwg);
            CHK(hipStreamEndCapture(st, &g));
            CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int rep = 0; rep < 3; ++rep) {
                CHK(hipEventRecord(e0, st));
                CHK(hipGraphLaunch(ge, st));
                CHK(hipEventRecord(e1, st));
                CHK(hipStreamSynchronize(st));
                CHK(hipEventElapsedTime(&ms[2], e0, e1));
            }
            CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
        }
        printf("%4d workgroups: per phase  own barrier %.2f us (%s)   grid.sync() %.2f us   graph node %.2f us\n", nwg,
               ms[0] * 1e3 / (iters * 3), ok ? "values ok" : "VALUES WRONG", ms[1] * 1e3 / (iters * 3), ms[2] * 1e3 / (iters * 3));
        CHK(hipFree(a)); CHK(hipFree(b)); CHK(hipFree(bar));
    }
    return 0;
}
