// Runs ON THE GPU BOX: what one SIMD sustains on the first pass of the long-list medoid (md_approx_tile) -- per 32x32 block of pairs three
// v_mfma_f32_32x32x2_f32 (the five-term expansion), sixteen v_sqrt_f32 and the adds of the column sums -- by how the two pipes are fed:
//   mode 0  matrix instructions only            mode 1  roots and adds only
//   mode 2  the kernel's form: six matrix instructions, then the roots and adds of THEIR results (a wave alternates between the pipes)
//   mode 3  software-pipelined: the matrix instructions of step k+1 are issued between the roots and adds of step k (two accumulator sets)
// for 1 .. 4 waves per SIMD on every CU.  Prints ns and cycles (at 2.4 GHz) per block and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define STEPS 2048          // steps per wave; a step = two blocks (the two 32-column groups of a 64-column tile)

static __device__ __forceinline__ float vsqrt_c(float x)
{
    return __builtin_amdgcn_fmed3f(__builtin_amdgcn_sqrtf(x), 0.0f, 1.0f);          // folds into v_sqrt_f32_e64 ... clamp
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float *out, float seed)
{
    __shared__ float s_rows[4][1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *rows = s_rows[wave];
    for (int i = lane; i < 1024; i += 64) rows[i] = seed * 1e-3f * (float)(i % 37);
    __builtin_amdgcn_wave_barrier();
    const bool lo = lane < 32;
    float b1[2], b2[2], b3[2];
    for (int g = 0; g < 2; ++g) { b1[g] = seed * 1e-3f * (lane + g); b2[g] = lo ? seed * 2e-3f : 1.0f; b3[g] = lo ? seed * 1e-4f * lane : 0.0f; }
    f2 s[2] = {(f2){0.f, 0.f}, (f2){0.f, 0.f}};
    const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto blocks = [&](float2 a, float a3, f16v &c0, f16v &c1) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1[0], zero, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1[1], zero, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b2[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b2[1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3[1], c1, 0, 0, 0);
    };
    auto roots = [&](const f16v &c0, const f16v &c1) {
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            s[0] += (f2){vsqrt_c(c0[q]), vsqrt_c(c0[q + 1])};
            s[1] += (f2){vsqrt_c(c1[q]), vsqrt_c(c1[q + 1])};
        }
    };
    if (MODE == 0) {
        for (int it = 0; it < STEPS; ++it) {
            const float2 a = *reinterpret_cast<const float2 *>(rows + 4 * ((it * 32 + (lane & 31)) & 255) + (lo ? 0 : 2));
            f16v c0, c1;
            blocks(a, lo ? 1.0f : 0.0f, c0, c1);
            s[0].x += c0[0] + c0[15]; s[1].x += c1[0] + c1[15];
        }
    } else if (MODE == 1) {
        f16v c0, c1;
        for (int q = 0; q < 16; ++q) { c0[q] = seed * q; c1[q] = seed * (q + 16); }
        for (int it = 0; it < STEPS; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { asm volatile("" : "+v"(c0[q])); asm volatile("" : "+v"(c1[q])); }
            roots(c0, c1);
        }
    } else if (MODE == 2) {
        for (int it = 0; it < STEPS; ++it) {
            const float2 a = *reinterpret_cast<const float2 *>(rows + 4 * ((it * 32 + (lane & 31)) & 255) + (lo ? 0 : 2));
            f16v c0, c1;
            blocks(a, lo ? 1.0f : 0.0f, c0, c1);
            roots(c0, c1);
        }
    } else if (MODE == 5 || MODE == 6) {
        // five v_mfma_f32_32x32x1_2b_f32 per step: one k per instruction, both 32-column groups at once (k = 5 exactly; the x2 shape wastes a sixth slot)
        typedef float f32v __attribute__((ext_vector_type(32)));
        f32v z32;
        for (int q = 0; q < 32; ++q) z32[q] = 0.0f;
        const float bx = b1[0], by = b2[0] * 0.5f, bz = b2[0], bn = b3[0] + 1e-5f;
        for (int it = 0; it < STEPS; ++it) {
            const float4 a = *reinterpret_cast<const float4 *>(rows + 4 * ((it * 32 + (lane & 31)) & 255));
            f32v c = __builtin_amdgcn_mfma_f32_32x32x1f32(a.x, bx, z32, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x1f32(a.y, by, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x1f32(a.z, bz, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x1f32(a.w, 1.0f, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x1f32(1.0f, bn, c, 0, 0, 0);
            if (MODE == 6) { s[0].x += c[0] + c[31]; continue; }
#pragma unroll
            for (int q = 0; q < 32; q += 4) {
                s[0] += (f2){vsqrt_c(c[q]), vsqrt_c(c[q + 1])};
                s[1] += (f2){vsqrt_c(c[q + 2]), vsqrt_c(c[q + 3])};
            }
        }
    } else if (MODE == 4) {
        // the same, unrolled twice: the two accumulator sets change roles, nothing is copied
        f16v p0, p1, c0, c1;
        {
            const float2 a = *reinterpret_cast<const float2 *>(rows + 4 * (lane & 31) + (lo ? 0 : 2));
            blocks(a, lo ? 1.0f : 0.0f, p0, p1);
        }
        for (int it = 1; it + 1 < STEPS; it += 2) {
            const float2 a = *reinterpret_cast<const float2 *>(rows + 4 * ((it * 32 + (lane & 31)) & 255) + (lo ? 0 : 2));
            const float2 a2 = *reinterpret_cast<const float2 *>(rows + 4 * ((it * 32 + 32 + (lane & 31)) & 255) + (lo ? 0 : 2));
            blocks(a, lo ? 1.0f : 0.0f, c0, c1);
            roots(p0, p1);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            blocks(a2, lo ? 1.0f : 0.0f, p0, p1);
            roots(c0, c1);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        roots(p0, p1);
    } else {
        f16v p0, p1;
        {
            const float2 a = *reinterpret_cast<const float2 *>(rows + 4 * (lane & 31) + (lo ? 0 : 2));
            blocks(a, lo ? 1.0f : 0.0f, p0, p1);
        }
        for (int it = 1; it < STEPS; ++it) {
            const float2 a = *reinterpret_cast<const float2 *>(rows + 4 * ((it * 32 + (lane & 31)) & 255) + (lo ? 0 : 2));
            f16v c0, c1;
            blocks(a, lo ? 1.0f : 0.0f, c0, c1);
            roots(p0, p1);
            // one matrix instruction, then eight of the 48 vector instructions of the step before
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            }
            p0 = c0; p1 = c1;
        }
        roots(p0, p1);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0].x + s[0].y + s[1].x + s[1].y;
}

template <int MODE>
static void run(const char *name, float *out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps : {1, 2, 3, 4}) {
        const int blocks = 256 * wps;                 // 4 waves per workgroup: one per SIMD
        std::vector<float> ts;
        for (int it = 0; it < 5; ++it) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
            (void)hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess) { fprintf(stderr, "launch failed\n"); exit(1); }
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (it) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        const double us = ts[ts.size() / 2] * 1e3;
        const double blocks_per_simd = (double)wps * STEPS * 2;
        printf("%-44s %d waves/SIMD: %8.1f us  %6.1f ns = %6.1f cycles (2.4 GHz) per 32x32 block and SIMD\n", name, wps, us, 1e3 * us / blocks_per_simd,
               2400.0 * us / blocks_per_simd);
        fflush(stdout);
    }
}

int main()
{
    float *out;
    if (hipMalloc(&out, 256 * 4 * 256 * 4 * 2) != hipSuccess) return 1;
    run<0>("matrix instructions only (3 per block)", out);
    run<1>("16 roots + 8 packed adds per block only", out);
    run<2>("6 matrix, then their roots (kernel's form)", out);
    run<3>("software-pipelined, 1 matrix : 8 vector", out);
    run<4>("software-pipelined, two accumulator sets", out);
    run<6>("5 x 32x32x1_2b matrix only", out);
    run<5>("5 x 32x32x1_2b, then the 32 roots", out);
    return 0;
}
