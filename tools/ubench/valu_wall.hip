// Runs ON THE GPU BOX: vector-instruction THROUGHPUT per SIMD by wall time (HIP events), for 1 / 2 / 3 / 4 waves per SIMD on every CU --
// what a SIMD of the MI355X sustains for the instruction kinds of k_project_q, independent of any cycle counter.
// One line per (instruction, waves per SIMD): wave-instructions per SIMD per microsecond, and the SIMD cycles per instruction at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 4096
#define CHAINS 8

template <int OP>
__global__ void k(float *out, float seed)
{
    float a[CHAINS];
    f2 p[CHAINS];
    int m[CHAINS];
    for (int i = 0; i < CHAINS; ++i) { a[i] = seed + threadIdx.x * 0.001f + i; p[i] = (f2){a[i], a[i] + 0.5f}; m[i] = threadIdx.x + i; }
    int sacc = 0;
    unsigned long long smask = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) ? 0x5555555555555555ull : 0x3333333333333333ull, smask2 = 0;
    float sval = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(seed)));
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
            if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
            if (OP == 3) asm volatile("v_min_f32 %0, %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(a[i]));
            if (OP == 5) { int s; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(m[i])); sacc += s; }
            if (OP == 6) asm volatile("v_cmp_gt_f32 vcc, %0, %0" ::"v"(a[i]) : "vcc");
            if (OP == 7) asm volatile("v_pk_sub_u16 %0, %0, %0" : "+v"(m[i]));
            if (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 9) asm volatile("v_add_u32 %0, %0, %0" : "+v"(m[i]));
            if (OP == 10) asm volatile("v_fmac_f32 %0, %0, %0" : "+v"(a[i]));
            if (OP == 11) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(m[i]));
            if (OP == 12) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[(i + 1) % CHAINS]), "v"(a[(i + 2) % CHAINS]));
            if (OP == 13) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[(i + 1) % CHAINS]), "v"(a[(i + 2) % CHAINS]), "s"(smask));
            if (OP == 14) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) % CHAINS]), "v"(a[(i + 2) % CHAINS]));
            if (OP == 15) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
            if (OP == 16) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
            if (OP == 17) asm volatile("v_and_b32 %0, %0, %1" : "+v"(m[i]) : "v"(m[(i + 1) % CHAINS]));
            if (OP == 18) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(m[i]) : "v"(m[(i + 1) % CHAINS]));
            if (OP == 19) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(m[i]) : "v"(m[(i + 1) % CHAINS]));
            if (OP == 20) asm volatile("v_cmp_eq_u32 vcc, %0, %1" ::"v"(m[i]), "v"(m[(i + 1) % CHAINS]) : "vcc");
            if (OP == 21) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
            if (OP == 22) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
            if (OP == 23) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) % CHAINS]));
            if (OP == 24) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(a[i]) : "s"(sval));
            if (OP == 25) asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "s"(smask));
            if (OP == 26) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(m[i]) : "v"(m[(i + 1) % CHAINS]), "v"(m[(i + 2) % CHAINS]));
            if (OP == 27) asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(smask2) : "v"(a[i]), "v"(a[(i + 1) % CHAINS]));
        }
    }
    float s = (float)sacc + (float)(smask2 & 1);
    for (int i = 0; i < CHAINS; ++i) s += a[i] + p[i].x + p[i].y + (float)m[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static void run(const char *name, float *out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int waves_per_simd : {1, 3, 8}) {
        const int threads = 256, blocks = 256 * waves_per_simd;          // 4 waves per workgroup: one per SIMD; `waves_per_simd` workgroups per CU
        std::vector<float> ts;
        for (int it = 0; it < 5; ++it) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1.5f);
            (void)hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess) { fprintf(stderr, "launch failed\n"); exit(1); }
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (it) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        const double us = ts[ts.size() / 2] * 1e3;
        const double per_simd = (double)waves_per_simd * REP * CHAINS;     // wave-instructions each SIMD executed
        printf("%-22s %d waves/SIMD: %8.1f us  %7.1f instr/SIMD/us  = %5.2f cycles per instruction at 2.4 GHz\n", name, waves_per_simd, us, per_simd / us,
               2400.0 * us / per_simd);
        fflush(stdout);
    }
}

int main()
{
    float *out;
    if (hipMalloc(&out, 256 * 8 * 256 * 4 * 2) != hipSuccess) return 1;
    run<0>("v_fma_f32", out); run<10>("v_fmac_f32 (e32)", out); run<1>("v_pk_fma_f32", out); run<2>("v_pk_mul_f32", out); run<3>("v_min_f32", out);
    run<4>("v_cndmask_b32", out); run<5>("v_readlane_b32", out); run<6>("v_cmp_gt_f32", out); run<7>("v_pk_sub_u16", out); run<8>("v_rcp_f32", out);
    run<9>("v_add_u32", out); run<11>("v_mov_b32_dpp", out);
    run<12>("v_cndmask vcc 3 regs", out); run<13>("v_cndmask_e64 sgpr mask", out); run<14>("v_med3_f32", out); run<15>("v_floor_f32", out);
    run<16>("v_cvt_i32_f32", out); run<17>("v_and_b32", out); run<18>("v_lshl_or_b32", out); run<19>("v_pk_min_u16", out); run<20>("v_cmp_eq_u32", out);
    run<21>("v_mul_f32", out); run<22>("v_add_f32", out); run<23>("v_max_f32", out); run<24>("v_fma_f32 sgpr src", out); run<25>("v_pk_fma_f32 sgpr src", out);
    run<26>("v_bfi_b32", out); run<27>("v_cmp_gt_f32 -> sgpr", out);
    return 0;
}
