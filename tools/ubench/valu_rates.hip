// Runs ON THE GPU BOX (tools/ubench/run.sh): issue cost of a few vector instructions on gfx950, wave64 -- cycles per instruction
// with 8 independent chains per wave and 1 / 4 waves per SIMD.  Plain HIP runtime, no torch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 256
template <int OP>
__global__ void k(float *out, long long *cyc, float seed)
{
    float a[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 0.001f + i; p[i] = (f2){a[i], a[i] + 0.5f}; }
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
            if (OP == 2) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (OP == 3) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
            if (OP == 6) asm volatile("v_max_f32 %0, %0, %0" : "+v"(a[i]));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP>
static void run(const char *name)
{
    float *out; long long *cyc;
    if (hipMalloc(&out, 1 << 24) != hipSuccess || hipMalloc(&cyc, 8 * 4096) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(1); }
    for (int threads : {64, 256, 1024}) {              // 1 wave on one SIMD; 1 wave per SIMD; 4 waves per SIMD (one workgroup per CU)
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.5f);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.5f);
        if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "launch failed\n"); exit(1); }
        long long h[256];
        if (hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) exit(1);
        double m = 0;
        for (int i = 0; i < 256; ++i) m += (double)h[i];
        m /= 256;
        const int waves_per_simd = threads >= 256 ? threads / 256 : 1;
        printf("%-14s %4d threads/workgroup: %7.2f wave-cycles per instruction, %6.2f SIMD cycles per instruction\n", name, threads,
               m / (REP * 8.0), m / (REP * 8.0) / waves_per_simd);
    }
    (void)hipFree(out); (void)hipFree(cyc);
}
int main()
{
    run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); run<5>("v_pk_mul_f32"); run<6>("v_max_f32");
    run<2>("v_sqrt_f32"); run<3>("v_rsq_f32"); run<4>("v_rcp_f32");
    return 0;
}
