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

// ---- wave-wide reductions and the scan, on the DATA-PARALLEL-PRIMITIVE path of the vector ALU (r04).
// __shfl_xor / __shfl_up compile to ds_bpermute_b32: a trip through the LDS crossbar and an s_waitcnt per step, six dependent steps
// per reduction -- several hundred cycles in kernels that are chains of dependent steps already (the RLE scan, the compaction's
// offsets, the frame tables).  A DPP operand costs nothing beyond its instruction: four steps inside each row of 16 lanes
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), then the four rows' values through v_readlane.
// Every lane (and the scalar unit) gets the result.
#define CM3D_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)
template <typename Op>
static __device__ __forceinline__ int cm3d_wave_reduce(int v, Op op)
{
    v = op(v, CM3D_DPP(v, 0xB1));          // quad_perm [1,0,3,2]
    v = op(v, CM3D_DPP(v, 0x4E));          // quad_perm [2,3,0,1]
    v = op(v, CM3D_DPP(v, 0x141));         // row_half_mirror
    v = op(v, CM3D_DPP(v, 0x140));         // row_mirror: every lane of a row holds the row's value
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16), r2 = __builtin_amdgcn_readlane(v, 32),
              r3 = __builtin_amdgcn_readlane(v, 48);
    return op(op(r0, r1), op(r2, r3));
}

// the same for any value made of 32-bit words (doubles, (distance, index) pairs): every word moves with the same DPP control
template <int CTRL, typename T>
static __device__ __forceinline__ T cm3d_dpp_t(T v)
{
    static_assert(sizeof(T) % 4 == 0, "whole 32-bit words");
    int w[sizeof(T) / 4];
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) w[i] = __builtin_amdgcn_update_dpp(0, w[i], CTRL, 0xF, 0xF, true);
    T o;
    __builtin_memcpy(&o, w, sizeof(T));
    return o;
}
template <typename T>
static __device__ __forceinline__ T cm3d_readlane_t(T v, int l)
{
    int w[sizeof(T) / 4];
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) w[i] = __builtin_amdgcn_readlane(w[i], l);
    T o;
    __builtin_memcpy(&o, w, sizeof(T));
    return o;
}
// op must be associative and commutative (min, max, a lexicographic minimum, ...): every lane gets the wave's value
template <typename T, typename Op>
static __device__ __forceinline__ T cm3d_wave_reduce_t(T v, Op op)
{
    v = op(v, cm3d_dpp_t<0xB1>(v));
    v = op(v, cm3d_dpp_t<0x4E>(v));
    v = op(v, cm3d_dpp_t<0x141>(v));
    v = op(v, cm3d_dpp_t<0x140>(v));
    const T r0 = cm3d_readlane_t(v, 0), r1 = cm3d_readlane_t(v, 16), r2 = cm3d_readlane_t(v, 32), r3 = cm3d_readlane_t(v, 48);
    return op(op(r0, r1), op(r2, r3));
}
struct Cm3dDistIdx { double s; int j; int pad; };          // (distance, index): lexicographic minimum = np.argmin's first minimum
static __device__ __forceinline__ Cm3dDistIdx cm3d_wave_argmin(double s, int j)
{
    Cm3dDistIdx v = {s, j, 0};
    return cm3d_wave_reduce_t(v, [](Cm3dDistIdx a, Cm3dDistIdx b) { return (b.s < a.s || (b.s == a.s && b.j < a.j)) ? b : a; });
}
struct Cm3dValIdx { float s; int j; };
static __device__ __forceinline__ Cm3dValIdx cm3d_wave_argmin_f(float s, int j)
{
    Cm3dValIdx v = {s, j};
    return cm3d_wave_reduce_t(v, [](Cm3dValIdx a, Cm3dValIdx b) { return (b.s < a.s || (b.s == a.s && b.j < a.j)) ? b : a; });
}

static __device__ __forceinline__ uint32_t cm3d_wave_or(uint32_t v)
{
    return (uint32_t)cm3d_wave_reduce((int)v, [](int a, int b) { return a | b; });
}

static __device__ __forceinline__ int cm3d_wave_sum(int v)
{
    return cm3d_wave_reduce(v, [](int a, int b) { return a + b; });
}

static __device__ __forceinline__ int cm3d_wave_min(int v)
{
    return cm3d_wave_reduce(v, [](int a, int b) { return a < b ? a : b; });
}

static __device__ __forceinline__ int cm3d_wave_max(int v)
{
    return cm3d_wave_reduce(v, [](int a, int b) { return a > b ? a : b; });
}

// inclusive scan across one wave: row_shr 1, 2, 4, 8 inside each row of 16 (zeros shift in), then the last lane of row 0 / 2 onto
// row 1 / 3 (row_bcast:15) and lane 31 onto rows 2 and 3 (row_bcast:31)
static __device__ __forceinline__ int cm3d_wave_incl_scan(int v)
{
    v += CM3D_DPP(v, 0x111);
    v += CM3D_DPP(v, 0x112);
    v += CM3D_DPP(v, 0x114);
    v += CM3D_DPP(v, 0x118);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);       // row_bcast:15, rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);       // row_bcast:31, rows 2 and 3
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
