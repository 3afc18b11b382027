// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
// Compiled with -ffp-contract=off: every fused multiply-add in the kernels is an
// explicit fmaf(), because the float32 results must equal the reference's
// (torch's small matmuls are k-sequential fma chains).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cm3d_hip.h"

#define CM3D_CHECK_LAUNCH()                                   \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return CM3D_ERR_LAUNCH; \
    } while (0)

#define CM3D_WAVE 64

static __device__ __forceinline__ int cm3d_lane() { return threadIdx.x & 63; }

// number of set bits of `m` below this lane (v_mbcnt)
static __device__ __forceinline__ int cm3d_mbcnt(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

static __device__ __forceinline__ uint32_t cm3d_wave_or(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= (uint32_t)__shfl_xor((int)v, o, 64);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

static __device__ __forceinline__ int cm3d_wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static __device__ __forceinline__ int cm3d_wave_min(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

static __device__ __forceinline__ int cm3d_wave_max(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// inclusive scan across one wave
static __device__ __forceinline__ int cm3d_wave_incl_scan(int v)
{
    const int lane = cm3d_lane();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// Exclusive scan over a 1024-thread block; `total` receives the block sum.
// s_part must hold 16 ints.  Contains two barriers.
static __device__ __forceinline__ int cm3d_block1024_excl_scan(int v, int *s_part, int &total)
{
    const int lane = cm3d_lane(), wave = threadIdx.x >> 6;
    int inc = cm3d_wave_incl_scan(v);
    __syncthreads();
    if (lane == 63) s_part[wave] = inc;
    __syncthreads();
    int wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        int c = s_part[w];
        if (w < wave) wbase += c;
        tot += c;
    }
    total = tot;
    return wbase + inc - v;
}

// k-sequential fma chain of a row-major 3x3 times a vector: what torch.matmul(3x3, 3xN)
// produces on the reference's path (utils/pcd.py:172 of the reference).
static __device__ __forceinline__ void cm3d_rot3(const float *R, float x, float y, float z, float &ox, float &oy, float &oz)
{
    float a = R[0] * x; a = fmaf(R[1], y, a); a = fmaf(R[2], z, a);
    float b = R[3] * x; b = fmaf(R[4], y, b); b = fmaf(R[5], z, b);
    float c = R[6] * x; c = fmaf(R[7], y, c); c = fmaf(R[8], z, c);
    ox = a; oy = b; oz = c;
}
