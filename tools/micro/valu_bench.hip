// Issue rates of the instructions the sweep's step loop is made of, on gfx950 (cycles per wave64
// instruction per SIMD with 4 waves per SIMD issuing independent chains).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_bench.hip -o tools/micro/valu_bench && tools/micro/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP 4096
enum { FMA64, FMA32, PKFMA32, RCP64, RCP32, DPP32, CVT64_32, ADD64, MUL64, CVTF32U16, LDS128, LDS128_RAND, LDS64, MIX64 };

template <int OP> __global__ __launch_bounds__(1024) void k(double *out, long long *cyc, const int *perm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    double a[8], b = 1.0000001, c = 1e-9;
    float f[8], g = 1.0000001f, h = 1e-9f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3 + i; f[i] = (float)a[i]; p[i] = f2{f[i], f[i] + 1}; }
    int addr = (OP == LDS128_RAND ? perm[threadIdx.x] : (int)threadIdx.x) * 16 % 65536;
    if (OP == LDS128 || OP == LDS128_RAND || OP == LDS64)
        for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float *>(lds)[i] = (float)i;
    __syncthreads();
    long long t0 = clock64();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == FMA64) a[i] = fma(a[i], b, c);
            if (OP == ADD64) a[i] = a[i] + c;
            if (OP == MUL64) a[i] = a[i] * b;
            if (OP == FMA32) f[i] = fmaf(f[i], g, h);
            if (OP == PKFMA32) p[i] = __builtin_elementwise_fma(p[i], f2{g, g}, f2{h, h});
            if (OP == RCP64) a[i] = __builtin_amdgcn_rcp(a[i]);
            if (OP == RCP32) f[i] = __builtin_amdgcn_rcpf(f[i]);
            if (OP == DPP32) f[i] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(f[i]), 0xB1, 0xF, 0xF, true));
            if (OP == CVT64_32) { a[i] = (double)f[i]; asm volatile("" : "+v"(a[i])); }
            if (OP == CVTF32U16) { f[i] = (float)(__float_as_uint(f[i]) & 0xFFFFu); asm volatile("" : "+v"(f[i])); }
            if (OP == LDS128 || OP == LDS128_RAND) {
                float4 v = *reinterpret_cast<const float4 *>(lds + ((addr + i * 160) & 0xFFF0));
                f[i] += v.x; asm volatile("" : "+v"(f[i]));
            }
            if (OP == LDS64) {
                float2 v = *reinterpret_cast<const float2 *>(lds + ((threadIdx.x * 8 + i * 512) & 0xFFF8));
                f[i] += v.x; asm volatile("" : "+v"(f[i]));
            }
            if (OP == MIX64) {   // the step loop's mix per nonzero: 20 fma64 : 1 rcp64 : 2 dpp : ...
                a[i] = fma(a[i], b, c);
            }
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + f[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP> void run(const char *name, double *out, long long *cyc, const int *perm)
{
    hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256, 1024, 65536>>>(out, cyc, perm);
    hipEventRecord(e0);
    k<OP><<<256, 1024, 65536>>>(out, cyc, perm);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long c[256];
    hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < 256; ++i) avg += (double)c[i];
    avg /= 256;
    // per SIMD: 4 waves x REP x 8 instructions
    const double n = 4.0 * REP * 8;
    printf("%-14s %8.3f ms   clock64 ticks/instr/SIMD %6.2f   ns/instr/SIMD %6.3f\n", name, ms, avg / n, ms * 1e6 / n);
}

int main()
{
    double *out; long long *cyc; int *perm;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 8); hipMalloc(&perm, 1024 * 4);
    std::vector<int> h(1024);
    srand(1);
    for (int i = 0; i < 1024; ++i) h[i] = (rand() % 400) * 10;   // random 160-byte rows
    hipMemcpy(perm, h.data(), 4096, hipMemcpyHostToDevice);
    run<FMA64>("v_fma_f64", out, cyc, perm);
    run<ADD64>("v_add_f64", out, cyc, perm);
    run<MUL64>("v_mul_f64", out, cyc, perm);
    run<FMA32>("v_fma_f32", out, cyc, perm);
    run<PKFMA32>("v_pk_fma_f32", out, cyc, perm);
    run<RCP64>("v_rcp_f64", out, cyc, perm);
    run<RCP32>("v_rcp_f32", out, cyc, perm);
    run<DPP32>("v_mov_dpp", out, cyc, perm);
    run<CVT64_32>("v_cvt_f64_f32", out, cyc, perm);
    run<CVTF32U16>("and+cvt_f32_u32", out, cyc, perm);
    run<LDS128>("ds_read_b128", out, cyc, perm);
    run<LDS128_RAND>("ds_read_b128 r", out, cyc, perm);
    run<LDS64>("ds_read_b64", out, cyc, perm);
    return 0;
}
