// a9: medoid of the in-mask points -- reference src/nuscenes/2d_to_3d.py:116-119 (get_medoid),
// gather at :620, use at :645-647.
// argmin_j sum_i D[i][j], D = torch.cdist(P, P) in float32:
//   M <= 25 : direct form   agg = fma(d,d,agg) over |a_k-b_k|, sqrt
//   M  > 25 : expansion     [-2x_i,-2y_i,-2z_i,n_i,1].[x_j,y_j,z_j,1,n_j] as a k-sequential fma
//             chain, clamp at 0, sqrt   (n = (x*x + y*y) + z*z)
// (the expansion's cancellation error at global-frame magnitudes is part
// of the reference's behaviour and is reproduced, not fixed.)
// Column j lives in one thread and is summed over rows i in ascending order, so the float32
// column sums do not depend on the launch geometry.  A tile = 64 columns of one mask = one wave; rows are
// staged through LDS 256 at a time and read back as wave-wide broadcasts, 8 rows per step in packed
// float32.  VALU-bound (about 14 issue slots per pair incl. the correctly rounded sqrt); nothing M x M ever
// touches HBM.  The tiles are worked longest lists first (worklist.h).
#include "common.h"
#include "worklist.h"
#include <stdlib.h>

#ifndef MD_WAVES
#define MD_WAVES 4                       // waves per workgroup, one tile each
#endif
#define MD_THREADS (MD_WAVES * 64)
#ifndef MD_STAGE
// rows of a list staged in a wave's LDS slice at a time (4 KiB).  512 held 32 VGPRs of gathered rows and 32 KiB of LDS per workgroup:
// alone the launch runs as fast with 256 or 128 (C2 44.5 / 44.8 against 45.3 us), and with three batches in flight the smaller
// footprint (80 VGPRs instead of 96, half the LDS) is worth 2-3 % of the pass (0.154-0.156 -> 0.151 ms).  Not below MDA_STAGE:
// the matrix-pipe first pass stages its rows in the same slice.
#define MD_STAGE 256
#endif

#ifndef MD_LONG_MIN
// Lists longer than this take the two-pass route (matrix-pipe first pass, k_medoid_long) -- in a batch that holds a list of more than
// MD_BATCH_LONG points; a batch without one runs the light instantiation of the tile kernel and nothing else.  r04 (first pass a sixth
// cheaper, second pass one wave per list): 512 / 384 / 256 / 128 give C4 790 / 826 / 847 / 843 k frames/s and C1 566 / 571 / 576 / 580 k;
// the headline shape (longest lists 358-400 points) is untouched by a batch limit of 448, while a batch limit of 256 costs its
// medoid stage 10 us alone (the heavy instantiation and the second pass for a handful of lists).
#define MD_LONG_MIN 256
#endif
#ifndef MD_BATCH_LONG
#define MD_BATCH_LONG 448                // (384 sent one of the headline shape's three resident batches -- longest list 390-odd points -- down the heavy route: 35 us for that pass)
#endif
#define MD_LONG_MAX 100000               // ... and shorter than this (the error bound of the first pass is derived for M < 10^5)
static __device__ __forceinline__ bool md_two_pass(int M) { return M > MD_LONG_MIN && M < MD_LONG_MAX; }
// The work list is ordered by class, most tiles first, and classes up to MD_UNI tiles hold exactly one tile count: its FIRST descriptor
// belongs to a list of more than MD_BATCH_LONG points if any list of the batch is that long.
static_assert(MD_BATCH_LONG >= MD_LONG_MIN && MD_BATCH_LONG <= CM3D_MEDOID_TILE * MD_UNI && MD_BATCH_LONG % CM3D_MEDOID_TILE == 0,
              "the batch limit is a boundary between two of the one-count classes");
static __device__ __forceinline__ bool md_batch_long(const TileDesc *desc, int ntiles) { return ntiles > 0 && __builtin_amdgcn_readfirstlane(desc[0].M) > MD_BATCH_LONG; }

#ifndef MD_APPROX_MFMA
// First pass over long lists on the matrix pipe (md_approx_tile); 0: on the vector pipe (md_rows<false, true>).  The error bound
// of the second pass rests on v_mfma_f32_32x32x2_f32 being bit for bit the k-ordered fmaf chain, which is verified for gfx950
// (cm3d_selftest_mfma: on the device at every engine start, LiftEngine.__init__, and in tests/test_gpu_golden.py): any other
// target gets the vector-pipe form.
#if defined(__gfx950__) || !defined(__HIP_DEVICE_COMPILE__)
#define MD_APPROX_MFMA 1
#else
#define MD_APPROX_MFMA 0
#endif
#endif
typedef float f2 __attribute__((ext_vector_type(2)));      // two rows side by side: v_pk_{add,mul,fma}_f32
typedef float f16v __attribute__((ext_vector_type(16)));   // accumulator of v_mfma_f32_32x32x2_f32
typedef float f32v __attribute__((ext_vector_type(32)));   // ... of v_mfma_f32_32x32x1_2b_f32 (two 32x32 blocks)
typedef int i2 __attribute__((ext_vector_type(2)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
#define PK_FMA(a, b, c) __builtin_elementwise_fma((a), (b), (c))

// Correctly rounded float32 square root = sqrtf() for 1e-30 <= x < 1e30 (md_sqrt_ok; the caller falls back to
// sqrtf() for the whole wave when any lane is outside -- zeros on the diagonal, never anything else on real
// coordinates), all in packable arithmetic: v_rsq_f32 (1 ulp), then the coupled Newton step on (s, h) ~ (sqrt x,
// 1 / (2 sqrt x)) and a final residual correction -- the sequence LLVM emits for a correctly rounded f32 sqrt when
// no denormal can be involved.  7 packed-able operations + the rsq per value, against 8 scalar ones + v_sqrt_f32 for
// the neighbour test (md_sqrt_core2_ref, kept as the checker: cm3d_selftest_sqrt compares the two and sqrtf() over
// every float of the domain, tests/test_gpu_golden.py).
static __device__ __forceinline__ f2 md_sqrt_core2(f2 x)
{
    const f2 r = {__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)};
    f2 s = x * r;
    f2 h = r * 0.5f;
    const f2 e = PK_FMA(-h, s, (f2)(0.5f));
    h = PK_FMA(h, e, h);
    s = PK_FMA(s, e, s);
    const f2 d = PK_FMA(-s, s, x);
    return PK_FMA(d, h, s);
}

// v_sqrt_f32 is within 1 ulp; with s- / s+ its neighbours,
//   t- = fma(s-, s, -x) >= 0  <=>  s- * s >= x  -> the root is nearer to s- than to s
//   t+ = fma(s+, s, -x) <  0  <=>  s+ * s <  x  -> the root is nearer to s+ than to s
// (an exact zero residual is +0 under round-to-nearest), so bits(result) = bits(s-) + sign(t-) + sign(t+).
static __device__ __forceinline__ f2 md_sqrt_core2_ref(f2 x)
{
    const f2 s = {__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)};
    const i2 sb = __builtin_bit_cast(i2, s);
    const i2 lo = sb - 1, hi = sb + 1;
    const f2 t_lo = PK_FMA(__builtin_bit_cast(f2, lo), s, -x), t_hi = PK_FMA(__builtin_bit_cast(f2, hi), s, -x);
    const u2 c = (__builtin_bit_cast(u2, t_lo) >> 31) + (__builtin_bit_cast(u2, t_hi) >> 31);
    return __builtin_bit_cast(f2, lo + __builtin_bit_cast(i2, c));
}
static __device__ __forceinline__ bool md_sqrt_ok(float x) { return (x < 1.0e30f) & (x >= 1.0e-30f); }

// md_sqrt_core2 that takes +0 as well (-> +0).  clamp_min_(0) leaves MANY zeros, not only the diagonal: the expansion's rounding
// noise is an ulp of the squared norm (0.25 m^2 at 1.7 km from the map origin, 8 m^2 at 10 km), so every pair of points closer
// than that has an even chance of a non-positive value.  Sending a whole 8-row step of 64 columns to sqrtf() for one such pair
// cost three times the step (PMC: 21 vector instructions per row where the loop has 11).  With the reciprocal root capped,
// x = 0 gives s = 0 * r = 0, e = 1/2, d = fma(-0, 0, 0) = +0 and the result fma(0, h, 0) = +0; for x >= 1e-30 the cap
// (rsq <= 1e15 there) changes nothing, so the values are md_sqrt_core2's.  8 v_min_f32 per step: the packed pipe has no min.
static __device__ __forceinline__ f2 md_sqrt_core2z(f2 x)
{
    const f2 r = {fminf(__builtin_amdgcn_rsqf(x.x), 1.0e18f), fminf(__builtin_amdgcn_rsqf(x.y), 1.0e18f)};
    f2 s = x * r;
    f2 h = r * 0.5f;
    const f2 e = PK_FMA(-h, s, (f2)(0.5f));
    h = PK_FMA(h, e, h);
    s = PK_FMA(s, e, s);
    const f2 d = PK_FMA(-s, s, x);
    return PK_FMA(d, h, s);
}
// x - 1 in the integer order of non-negative floats: +0 becomes the LARGEST value, so an unsigned minimum is the smallest
// non-zero one
static __device__ __forceinline__ uint32_t md_nz_key(float x) { return __float_as_uint(x) - 1u; }
static __device__ __forceinline__ bool md_sqrt_okz(float x) { return x == 0.0f || md_sqrt_ok(x); }

struct TileBest { float s; int j; };

#ifdef CM3D_DIAG
// Diagnostic build only (make diag; tools/md_diag.py): s_memtime of every tile's wave at its start, after its first staged
// chunk and at its end, its placement (XCC_ID << 32 | HW_ID) and its list length.
#define MD_DIAG_WAVES 65536
__device__ int g_md_diag;
__device__ unsigned long long g_md_wave[5 * MD_DIAG_WAVES];
__device__ unsigned long long g_md_clock[4];          // s_memtime and the 100 MHz s_memrealtime at the first wave's start and the last wave's end
static __device__ __forceinline__ unsigned long long md_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
extern "C" int cm3d_md_diag_set(int flags)
{
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_md_diag), &flags, sizeof(int)) != hipSuccess) return CM3D_ERR_LAUNCH;
    unsigned long long z[4] = {~0ull, ~0ull, 0ull, 0ull};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_md_clock), z, sizeof(z)) != hipSuccess) return CM3D_ERR_LAUNCH;
    void *wv = nullptr;
    if (hipGetSymbolAddress(&wv, HIP_SYMBOL(g_md_wave)) != hipSuccess || hipMemset(wv, 0, sizeof(g_md_wave)) != hipSuccess) return CM3D_ERR_LAUNCH;
    return hipDeviceSynchronize() == hipSuccess ? CM3D_OK : CM3D_ERR_LAUNCH;
}
extern "C" int cm3d_md_diag_read_clock(unsigned long long *out_host)
{
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_md_clock), 4 * sizeof(unsigned long long)) == hipSuccess ? CM3D_OK : CM3D_ERR_LAUNCH;
}
extern "C" int cm3d_md_diag_read_waves(unsigned long long *out_host, int n_waves)
{
    if (n_waves > MD_DIAG_WAVES) return CM3D_ERR_ARG;
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_md_wave), 5 * (size_t)n_waves * sizeof(unsigned long long)) == hipSuccess ? CM3D_OK : CM3D_ERR_LAUNCH;
}
#endif
__global__ __launch_bounds__(1024) void k_medoid_desc(int n_masks, const int32_t *__restrict__ hit_off,
                                                      const int32_t *__restrict__ tile_off, int idx_cap, int tile_cap,
                                                      TileDesc *__restrict__ desc)
{
    __shared__ int s_hist[MD_CLASSES], s_cur[MD_CLASSES];
    md_build_worklist<1024>(n_masks, hit_off, tile_off, idx_cap, tile_cap, desc, s_hist, s_cur);
}

// LDS layout of a staged 64-row chunk: rows in pairs, component-major inside a pair, so that one ds_read_b128
// delivers register pairs for packed math:  float4 A[p] = {x(2p), x(2p+1), y(2p), y(2p+1)},
//                                           float4 B[p] = {z(2p), z(2p+1), w(2p), w(2p+1)}   at s4[2p], s4[2p+1].
static __device__ __forceinline__ void md_stage(float4 *s4, int row, float4 r)
{
    float *f = (float *)s4 + (row >> 1) * 8 + (row & 1);
    f[0] = r.x; f[2] = r.y; f[4] = r.z; f[6] = r.w;
}

// squared distances of the two rows of a pair to this lane's column
template <bool DIRECT>
static __device__ __forceinline__ f2 md_pair2(const float4 A, const float4 B, float qx, float qy, float qz, float qn)
{
    const f2 X = {A.x, A.y}, Y = {A.z, A.w}, Z = {B.x, B.y}, Wn = {B.z, B.w};
    if (DIRECT) {
        f2 dd = __builtin_elementwise_abs(X - qx);
        f2 agg = PK_FMA(dd, dd, (f2)(0.0f));
        dd = __builtin_elementwise_abs(Y - qy); agg = PK_FMA(dd, dd, agg);
        dd = __builtin_elementwise_abs(Z - qz); agg = PK_FMA(dd, dd, agg);
        return agg;
    }
    f2 acc = X * qx;                          // (-2 x_i) * x_j
    acc = PK_FMA(Y, (f2)(qy), acc);
    acc = PK_FMA(Z, (f2)(qz), acc);
    acc = acc + Wn;                           // fma(n_i, 1, acc)
    acc = acc + qn;                           // fma(1, n_j, acc)
    // clamp_min_(0); the in-image test only lets finite points into a mask, so acc is never NaN
    return (f2){fmaxf(acc.x, 0.0f), fmaxf(acc.y, 0.0f)};
}

// adds the distances of `cnt` staged rows to s, in ascending row order.  8 rows per step: the distance chains
// are independent (ILP, and the LDS reads of a step are issued together), only the adds into s are sequential
// -- which is what fixes the float32 sum.
// APPROX: the raw v_sqrt_f32 (within 1 ulp of the root on [1e-30, 1e30), cm3d_selftest_sqrt; 0 below) instead of the
// correctly rounded root -- the first pass over long lists (k_medoid_long settles them)
static __device__ __forceinline__ float md_asqrt(float x) { return x < 1.0e-30f ? 0.0f : __builtin_amdgcn_sqrtf(x); }

// SAFE: the caller has established (md_col_safe, md_row_safe) that every value of the expansion is either <= 0 -- clamp_min_
// makes it +0 -- or inside md_sqrt_ok's domain, so the per-step test of the extremes (two min / max trees, 12 instructions of
// a step's ~100) is not needed.  The argument: the chain ends in fl(a + n_j) with n_j this lane's squared norm.  A positive
// result with a >= -n_j / 2 is >= n_j / 2; with -n_j < a < -n_j / 2 the subtraction is exact (Sterbenz) and a multiple of
// ulp(n_j / 2) >= 2^-25 n_j.  So n_j >= 1e-22 bounds every positive value from below by 2.9e-30; and n_i, n_j < 2e29 bound
// it from above by 2 (n_i + n_j) (1 + 1e-6) < 1e30.
static __device__ __forceinline__ bool md_col_safe(float qn) { return qn >= 1.0e-22f && qn < 2.0e29f; }
static __device__ __forceinline__ bool md_row_safe(float n) { return n < 2.0e29f; }

// v_pk_add_f32 with the output clamp: both halves held to [0, 1], a NaN becomes 0 (DX10_CLAMP; tools/ubench/pk_clamp.hip shows the
// chip doing it).  On coordinates scaled by 2^-16 (md_approx_tile's comment: every value of the chain scales by exactly 2^-32, every
// distance by 2^-16, and all of them are below 1) the LAST addition of the expansion, + n_j, does clamp_min_(0) on the way.
static __device__ __forceinline__ f2 md_pk_add_clamp01(f2 a, f2 b)
{
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// SCALED (with SAFE): rows and column are scaled by 2^-16 (norms by 2^-32) and every squared distance is below 1 -- the caller has
// checked the norms (mda_point_ok) --, so clamp_min_(0) rides on the last addition and the 8 v_max_f32 of a step are gone; s is in
// scaled units.
template <bool DIRECT, bool APPROX = false, bool SAFE = false, bool SCALED = false>
static __device__ __forceinline__ float md_rows(const float4 *s4, int cnt, float qx, float qy, float qz, float qn, float s)
{
#ifndef MD_U
#define MD_U 4
#endif
    constexpr int U = MD_U;                   // pairs per step
    int ii = 0;
#ifdef CM3D_DIAG
    const int ab = __builtin_amdgcn_readfirstlane(g_md_diag);      // ablations (results wrong by construction): 2 no root, 4 no domain test, 8 no LDS reads, 16 one add per step
#endif
    for (; ii + 2 * U <= cnt; ii += 2 * U) {
        f2 d[U];
        if (DIRECT) {
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = md_pair2<true>(s4[ii + 2 * u], s4[ii + 2 * u + 1], qx, qy, qz, qn);
        } else {
            // the U chains written side by side, one operation of every chain at a time: back-to-back dependent
            // packed operations of one chain cost wait states, operations of different chains do not
            float4 A[U], B[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { A[u] = s4[ii + 2 * u]; B[u] = s4[ii + 2 * u + 1]; }
#ifdef CM3D_DIAG
            if (ab & 8) {
#pragma unroll
                for (int u = 0; u < U; ++u) { A[u] = make_float4(qx + u, qy, qz, qn + ii); B[u] = make_float4(qy, qz + u, qx, qn + 3.0f); }
            }
#endif
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = (f2){A[u].x, A[u].y} * qx;                       // (-2 x_i) * x_j
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = PK_FMA(((f2){A[u].z, A[u].w}), (f2)(qy), d[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = PK_FMA(((f2){B[u].x, B[u].y}), (f2)(qz), d[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = d[u] + (f2){B[u].z, B[u].w};                     // fma(n_i, 1, acc)
            if (SCALED) {
                const f2 qn2 = (f2)(qn);
#pragma unroll
                for (int u = 0; u < U; ++u) d[u] = md_pk_add_clamp01(d[u], qn2);                // fma(1, n_j, acc) and clamp_min_(0)
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) d[u] = d[u] + qn;                                   // fma(1, n_j, acc)
#pragma unroll
                for (int u = 0; u < U; ++u) d[u] = __builtin_elementwise_max(d[u], (f2)(0.0f)); // clamp_min_(0)
            }
        }
        if (SAFE && !DIRECT && !APPROX) {
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = md_sqrt_core2z(d[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) { s = s + d[u].x; s = s + d[u].y; }
            continue;
        }
        // md_sqrt_core's domain, tested on the extremes; packed min / max trees
        f2 lo2 = d[0], hi2 = d[0];
#pragma unroll
        for (int u = 1; u < U; ++u) { lo2 = __builtin_elementwise_min(lo2, d[u]); hi2 = __builtin_elementwise_max(hi2, d[u]); }
        const float lo = fminf(lo2.x, lo2.y), hi = fmaxf(hi2.x, hi2.y);
#ifdef CM3D_DIAG
        if (!APPROX && (ab & 6)) {
            if (!(ab & 2)) {
#pragma unroll
                for (int u = 0; u < U; ++u) d[u] = md_sqrt_core2(d[u]);
            } else if (!(ab & 4)) {
                s += (lo >= 1.0e-30f && hi < 1.0e30f) ? 0.0f : 1.0f;
            }
        } else
#endif
        if (APPROX) {
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = (f2){md_asqrt(d[u].x), md_asqrt(d[u].y)};
        } else if (!__ballot(!(lo >= 1.0e-30f && hi < 1.0e30f))) {
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = md_sqrt_core2(d[u]);
        } else {
            // some lane holds a value outside the domain: almost always a zero (md_sqrt_core2z), which is no reason to leave
            // the packed form -- only a NON-ZERO value below 1e-30, or one beyond 1e30, is
            uint32_t nz = 0xFFFFFFFFu;
#pragma unroll
            for (int u = 0; u < U; ++u) nz = min(nz, min(md_nz_key(d[u].x), md_nz_key(d[u].y)));
            if (!__ballot(!(nz >= __builtin_bit_cast(uint32_t, 1.0e-30f) - 1u && hi < 1.0e30f))) {
#pragma unroll
                for (int u = 0; u < U; ++u) d[u] = md_sqrt_core2z(d[u]);
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) d[u] = (f2){sqrtf(d[u].x), sqrtf(d[u].y)};
            }
        }
#ifdef CM3D_DIAG
        if (ab & 16) {
            f2 a2 = d[0];
#pragma unroll
            for (int u = 1; u < U; ++u) a2 = a2 + d[u];
            s = s + (a2.x + a2.y);
            continue;
        }
#endif
#pragma unroll
        for (int u = 0; u < U; ++u) { s = s + d[u].x; s = s + d[u].y; }
    }
    for (; ii < cnt; ii += 2) {               // remaining pairs; the last one may hold a single row
        const f2 d = md_pair2<DIRECT>(s4[ii], s4[ii + 1], qx, qy, qz, qn);
        const bool two = ii + 1 < cnt;
        const bool bad = !(SAFE && !DIRECT) && (!md_sqrt_okz(d.x) || (two && !md_sqrt_okz(d.y)));
        const f2 r = APPROX ? (f2){md_asqrt(d.x), md_asqrt(d.y)}
                            : (__ballot(bad) ? (f2){sqrtf(d.x), sqrtf(d.y)} : md_sqrt_core2z((f2){d.x, two ? d.y : 1.0f}));
        s = s + r.x;
        if (two) s = s + r.y;
    }
    return s;
}

// First pass over the LONG lists (md_two_pass) on the matrix pipe.  torch.cdist's expansion is a k-sequential float32 fma
// chain over k = 0..4 of [-2x_i, -2y_i, -2z_i, n_i, 1] . [x_j, y_j, z_j, 1, n_j] -- exactly what v_mfma_f32_32x32x2_f32
// computes (bit for bit a k-ordered fmaf chain, one rounding per product: cdna_hip_programming.md, FP32-input MFMA), so three
// of them (k = 0,1 | 2,3 | 4,pad) give the reference's 32 x 32 squared distances with nothing left for the vector pipe but
// max(., 0), v_sqrt_f32 and the sum.  (cm3d_selftest_mfma compares the MFMA values with the vector chain on the device.)
// The sums are APPROXIMATE (v_sqrt_f32 is within one ulp; each lane adds its own 16 rows of a tile, the halves of a column
// meet at the end): the error bound of k_medoid_long holds for any summation order, and the position comes out of its exact
// second pass.  A 64-column tile = two 32-column MFMA tiles, one wave (k_medoid_tiles); rows staged through its LDS slice as
// {-2x, -2z, -2y, n}: lanes 0-31 feed k even (x, z, the 1), lanes 32-63 k odd (y, n, 0).
#define MDA_STAGE 256
static_assert(MD_STAGE >= MDA_STAGE, "md_approx_tile stages MDA_STAGE rows in a wave's slice of MD_STAGE rows");
static __device__ __forceinline__ float md_vsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// v_sqrt_f32 with the output clamp: the result is held to [0, 1] and a NaN (the root of a negative value) becomes 0 (DX10_CLAMP, on
// in every kernel of this library) -- clamp_min_(0) and the root in ONE instruction for values known to be <= 1
static __device__ __forceinline__ float md_vsqrt_clamp01(float x)
{
    // (the compiler folds the median into the root's output modifier -- v_sqrt_f32_e64 ... clamp --; written as inline assembly the
    // instruction was invisible to its hazard recogniser, which is what places the wait states between a matrix instruction and the
    // first vector instruction that reads its result)
    return __builtin_amdgcn_fmed3f(__builtin_amdgcn_sqrtf(x), 0.0f, 1.0f);
}
// Scaled form of the first pass.  Coordinates times s = 2^-16 scale every product, every sum and every rounding of the chain by
// exactly s^2 = 2^-32 and its root by exactly s (powers of four in, powers of two out: no rounding changes as long as nothing
// underflows), so the sums come out as s times the unscaled ones, bit for bit -- and with all distances <= 2^16 m the scaled roots
// are <= 1, which is what lets md_vsqrt_clamp01 do the clamp.  MDA_N_MAX bounds the squared norms of rows and columns
// (|p_i - p_j| <= |p_i| + |p_j| <= 2 sqrt(1e9) = 63.2 km, 0.965 after scaling, expansion noise of 64 m^2 included);
// MDA_N_MIN keeps products of coordinates away from the denormals (a norm of exactly 0 is a point at the origin: all its
// products are exact zeros).  What is left of underflow inside a chain (one coordinate of 1e-20 m beside another of a metre) is
// below 4e-29 in the squared distance, 6e-15 in a root: k_medoid_long's bound allows 2e-14 per term.  Chunks of rows or tiles
// of columns outside these bounds (30 km from the map origin, say) take the unscaled form with the integer clamp.
#if defined(__gfx950__) || !defined(__HIP_DEVICE_COMPILE__)
#define MD_SCALED_ROUTES 1               // verified on gfx950 (cm3d_selftest_mfma at every engine start; tools/ubench/pk_clamp.hip)
#else
#define MD_SCALED_ROUTES 0               // any other target: the unscaled forms
#endif
#define MDA_S 1.52587890625e-05f
#define MDA_S2 2.3283064365386963e-10f
#define MDA_N_MAX 1.0e9f
#define MDA_N_MIN 1.0e-10f
// ... of a POINT: a squared norm of zero only counts when the point IS the origin.  Coordinates of 1e-23 m square to zero as well, but
// their products with other points' coordinates do not vanish, and the scaled chain would carry denormal values into the root's
// sequence outside the domain it is checked on; such points take the unscaled, tested routes.
static __device__ __forceinline__ bool mda_point_ok(float x, float y, float z, float n)
{
    return n == 0.0f ? (x == 0.0f && y == 0.0f && z == 0.0f) : (n >= MDA_N_MIN && n < MDA_N_MAX);
}

// One step of the first pass: 32 staged rows (r0 ..) against the wave's 64 columns.  v_mfma_f32_32x32x1_2b_f32 takes ONE k per
// instruction and computes two 32x32 blocks -- block lane / 32 pairs the rows (A: row lane % 32, the same in both halves) with the
// columns of that half (B: this lane's column) --, so the five terms of the expansion are five instructions for 2048 pairs: the
// 32x32x2 shape needs six (its sixth k slot multiplies 0 by 0).  Float32 matrix instructions and vector instructions do NOT overlap
// on this chip -- tools/ubench/mfma_sqrt.hip: 211 (169 in this shape) + 167 cycles per 1024 pairs alone, 377 (344) together,
// whatever the number of waves and however the two are interleaved --, so every instruction saved on either side counts.
// Accumulator registers 0-15 = block 0 (column lane % 32), 16-31 = block 1 (column 32 + lane % 32), rows 8 (q / 4) + 4 (lane / 32) + q % 4.
template <bool SCALED>
static __device__ __forceinline__ f2 md_approx_step(const float *s_rows, int r0, float a5, float bx, float by, float bz, float bn, f2 s)
{
    const int lane = cm3d_lane();
    const float4 a = *reinterpret_cast<const float4 *>(s_rows + 4 * (r0 + (lane & 31)));       // (-2x, -2y, -2z, n) of row r0 + lane % 32
    f32v c;
#pragma unroll
    for (int q = 0; q < 32; ++q) c[q] = 0.0f;
    c = __builtin_amdgcn_mfma_f32_32x32x1f32(a.x, bx, c, 0, 0, 0);            // (-2 x_i) x_j
    c = __builtin_amdgcn_mfma_f32_32x32x1f32(a.y, by, c, 0, 0, 0);            // fma(-2 y_i, y_j, .)
    c = __builtin_amdgcn_mfma_f32_32x32x1f32(a.z, bz, c, 0, 0, 0);            // fma(-2 z_i, z_j, .)
    c = __builtin_amdgcn_mfma_f32_32x32x1f32(a.w, 1.0f, c, 0, 0, 0);          // fma(n_i, 1, .)
    c = __builtin_amdgcn_mfma_f32_32x32x1f32(a5, bn, c, 0, 0, 0);             // fma(1, n_j, .); a5 = 0 for a row past the end (all its terms are 0)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        if (SCALED)     // clamp_min_(0) and the root in one (md_vsqrt_clamp01), this lane's rows of its two columns
            s += (f2){md_vsqrt_clamp01(c[q]), md_vsqrt_clamp01(c[16 + q])};
        else            // clamp_min_(0) as a signed-integer maximum on the bits (negative values and -0 -> +0), the root brought to the sums' scale
            s += (f2){md_vsqrt(__int_as_float(max(__float_as_int(c[q]), 0))), md_vsqrt(__int_as_float(max(__float_as_int(c[16 + q]), 0)))} * MDA_S;
    }
    return s;
}

template <typename Fetch>
static __device__ __forceinline__ void md_approx_tile(Fetch fetch, float *s_rows, int off, int M, int jt, float *__restrict__ approx_out)
{
    const int lane = cm3d_lane();
    // this lane's column, SCALED: (x s, y s, z s, n s^2)
    float qx = 0.f, qy = 0.f, qz = 0.f;
    {   // (the load is unconditional -- of the list's last point for a lane past its end: a load inside a divergent branch is waited for on the spot)
        const int j = jt * 64 + lane;
        const float4 q = fetch(off + min(j, M - 1));
        if (j < M) { qx = q.x; qy = q.y; qz = q.z; }
    }
    // the first stage's rows: in flight under the column's arithmetic; every later stage's are fetched under the stage before
    float4 g4[MDA_STAGE / 64];
#pragma unroll
    for (int c = 0; c < MDA_STAGE / 64; ++c) g4[c] = fetch(off + min(c * 64 + lane, M - 1));
    const float qn = (qx * qx + qy * qy) + qz * qz;
    const bool cols_ok = !__ballot(!mda_point_ok(qx, qy, qz, qn));           // (uniform)
    const float bx = qx * MDA_S, by = qy * MDA_S, bz = qz * MDA_S, bn = qn * MDA_S2;
    f2 s = {0.f, 0.f};                                        // in units of s: columns lane % 32 and 32 + lane % 32, this half's rows
    bool scaled = false;                                      // (uniform) the staged rows are in the scaled form
#ifdef CM3D_DIAG
    const int ab = __builtin_amdgcn_readfirstlane(g_md_diag);  // ablation (results wrong by construction; tools/md_long_ablate.py): 128 rows staged once per tile
#endif
    for (int i0 = 0; i0 < M; i0 += MDA_STAGE) {
#ifdef CM3D_DIAG
        if (!((ab & 128) && i0 > 0)) {
#endif
        __builtin_amdgcn_wave_barrier();                      // the previous rows' readers are done
        float4 rr[MDA_STAGE / 64];
        bool rows_ok = true;
#pragma unroll
        for (int c = 0; c < MDA_STAGE / 64; ++c) {
            rr[c] = make_float4(0.f, 0.f, 0.f, 0.f);          // rows past the end: all zeros, and their k = 4 factor is 0 too
            if (i0 + c * 64 + lane < M) {
                const float4 p = g4[c];
                rr[c] = make_float4(-2.0f * p.x, -2.0f * p.y, -2.0f * p.z, (p.x * p.x + p.y * p.y) + p.z * p.z);
                rows_ok &= mda_point_ok(p.x, p.y, p.z, rr[c].w);
            }
        }
#pragma unroll
        for (int c = 0; c < MDA_STAGE / 64; ++c) g4[c] = fetch(off + min(i0 + MDA_STAGE + c * 64 + lane, M - 1));      // the next stage's (past the end: the last row, unused)
        const bool scaled_now = MD_SCALED_ROUTES && cols_ok && !__ballot(!rows_ok);   // (uniform) this chunk of rows in the scaled form
#pragma unroll
        for (int c = 0; c < MDA_STAGE / 64; ++c) {
            float4 r = rr[c];
            if (scaled_now) r = make_float4(r.x * MDA_S, r.y * MDA_S, r.z * MDA_S, r.w * MDA_S2);
            *reinterpret_cast<float4 *>(s_rows + 4 * (c * 64 + lane)) = r;
        }
        scaled = scaled_now;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef CM3D_DIAG
        }
#endif
        const int cnt = min(MDA_STAGE, M - i0), full = cnt & ~31;
        const float a5 = full + (lane & 31) < cnt ? 1.0f : 0.0f;            // the last, partial step's rows
        if (scaled) {
            for (int r0 = 0; r0 < full; r0 += 32) s = md_approx_step<true>(s_rows, r0, 1.0f, bx, by, bz, bn, s);
            if (cnt & 31) s = md_approx_step<true>(s_rows, full, a5, bx, by, bz, bn, s);
        } else {
            // the unscaled form (rare): operands back to their own scale (exact unless a coordinate underflowed when it was scaled: such
            // a column has a norm below MDA_N_MIN)
            const float ux = bx * 65536.0f, uy = by * 65536.0f, uz = bz * 65536.0f, un = bn * 4294967296.0f;
            for (int r0 = 0; r0 < full; r0 += 32) s = md_approx_step<false>(s_rows, r0, 1.0f, ux, uy, uz, un, s);
            if (cnt & 31) s = md_approx_step<false>(s_rows, full, a5, ux, uy, uz, un, s);
        }
    }
    // the two halves of the rows of a column meet; back to metres
    const float t0 = (s.x + __shfl_xor(s.x, 32, 64)) * 65536.0f, t1 = (s.y + __shfl_xor(s.y, 32, 64)) * 65536.0f;
    if (lane < 32) {
        const int j = jt * 64 + lane;
        if (j < M) approx_out[off + j] = t0;
        if (j + 32 < M) approx_out[off + j + 32] = t1;
    }
}

// One wave = one tile = 64 columns of one mask; rows are staged 64 at a time through the wave's own
// LDS slice and read back as broadcasts.  No workgroup barrier: waves of a block are independent.
// Two instantiations, launched one behind the other over the same work list, of which exactly ONE does the work: WITH_LONG = true
// holds the matrix-pipe first pass of the long lists as well -- 16 accumulator registers and operand pairs that cost every wave
// of the launch a quarter of its occupancy (122 against 94 VGPRs) --, WITH_LONG = false only the exact loop.  The work list is
// ordered longest lists first, so its first descriptor says whether the batch can hold a long list at all (more than MD_UNI
// tiles: conservative inside the class that holds MD_LONG_MIN); the instantiation that is not needed leaves after that one load.
// On the headline shape (no list beyond 358 points) the light one runs; three batches in flight: 0.174 -> 0.166 ms per pass.
template <bool WITH_LONG, bool INDEXED>
__global__ __launch_bounds__(MD_THREADS, 4) void k_medoid_tiles(const float4 *__restrict__ points,
                                                              const int32_t *__restrict__ pt_off,
                                                              const int32_t *__restrict__ mask_frame, int n_masks,
                                                              const int32_t *__restrict__ tile_off,
                                                              const int32_t *__restrict__ hit_row,
                                                              const TileDesc *__restrict__ desc,
                                                              TileBest *__restrict__ tile_best, int tile_cap,
                                                              float *__restrict__ colsum_opt, float *__restrict__ approx_opt,
                                                              int32_t *__restrict__ long_list, int32_t *__restrict__ feedback)
{
    __shared__ float4 s_row_all[MD_WAVES][MD_STAGE];
    if (!WITH_LONG && blockIdx.x == 0 && threadIdx.x == 0) {
        long_list[0] = 0;                                                         // k_medoid_reduce counts the long masks into it
        // what the caller may use as a hint for the NEXT batch (cm3d_medoid2): does this one hold a list the two-pass route
        // would take?  The work list is ordered longest lists first: its first entry says (conservatively, inside its class).
        if (feedback) feedback[0] = md_batch_long(desc, min(tile_off[n_masks], tile_cap)) ? 1 : 0;
    }
    const int wave = threadIdx.x >> 6, lane = cm3d_lane();
#ifdef CM3D_DIAG
    const int diag = g_md_diag & 1;
    const unsigned long long t_start = diag ? md_now() : 0ull;
    unsigned long long t_staged = 0ull;
    const unsigned long long w_start = diag ? wall_clock64() : 0ull;
#endif
    float4 *s_row = s_row_all[wave];
    const int ntiles = min(tile_off[n_masks], tile_cap);
    if (ntiles <= 0) return;
    if ((approx_opt != nullptr && md_batch_long(desc, ntiles)) != WITH_LONG) return;       // the other instantiation's batch
    // (tiles in strides of the grid, longest lists first.  Handing the heavy instantiation's tiles out from a counter instead -- so that a
    // wave that walks 2000 rows does not hold its workgroup's slots -- costs more than it balances: 29 k atomics on one address, C1's stage
    // 269 -> 505 us; profiles/r04_project_experiments.txt)
    for (int t = blockIdx.x * MD_WAVES + wave; t < ntiles; t += gridDim.x * MD_WAVES) {
        const TileDesc d = desc[t];
        // the descriptor is the same in every lane: keep it in scalar registers so that the loops below are
        // uniform control flow
        const int off = __builtin_amdgcn_readfirstlane(d.off), M = __builtin_amdgcn_readfirstlane(d.M);
        const int jt = __builtin_amdgcn_readfirstlane(d.jt);
        // hit_row == NULL: `points` is the per-hit coordinate array of cm3d_compact_hits, laid out like the index lists
        // (INDEXED is a template parameter and every load below unconditional -- a lane past the list's end loads its last point --: a
        // load behind a run-time `hit_row ?` or inside a divergent branch is waited for on the spot, and the four gathers of a 256-row
        // stage then cost four memory round trips instead of one)
        const float4 *P = INDEXED ? points + pt_off[mask_frame[__builtin_amdgcn_readfirstlane(d.m)]] : points;
        auto fetch = [&](int q) { return INDEXED ? P[hit_row[q]] : P[q]; };
        const int j = jt * 64 + lane;
        const bool act = j < M;
        float qx = 0.f, qy = 0.f, qz = 0.f, qn = 0.f;
        {
            const float4 q = fetch(off + min(j, M - 1));
            if (act) {
                qx = q.x; qy = q.y; qz = q.z;
                qn = (q.x * q.x + q.y * q.y) + q.z * q.z;
            }
        }
        float s = 0.f;
        const bool col_safe = !__ballot(act && !md_col_safe(qn));
        const bool col_scal = !__ballot(act && !mda_point_ok(qx, qy, qz, qn));           // the column side of the scaled form (md_rows<.., SCALED>)
        const bool direct = M <= 25;
        const bool approx = WITH_LONG && md_two_pass(M);                  // long list (of a batch with the first pass: this instantiation): approximate sums, k_medoid_long later
#ifdef CM3D_DIAG
        if (WITH_LONG && ((g_md_diag & 512) ? !approx : ((g_md_diag & 1024) ? approx : false))) continue;      // 512: first-pass tiles only, 1024: exact tiles only
#endif
#ifdef CM3D_DIAG
        const unsigned long long t_tile = diag ? md_now() : 0ull;           // (per tile; t_start is the wave's)
        auto stamp = [&](unsigned long long staged) {
            if (diag && lane == 0 && t < MD_DIAG_WAVES) {
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
                g_md_wave[5 * t] = t_tile; g_md_wave[5 * t + 1] = staged; g_md_wave[5 * t + 2] = md_now();
                g_md_wave[5 * t + 3] = ((unsigned long long)xcc << 32) | hw; g_md_wave[5 * t + 4] = (unsigned long long)M;
                if (t == 0) { g_md_clock[0] = t_start; g_md_clock[1] = w_start; g_md_clock[2] = md_now(); g_md_clock[3] = wall_clock64(); }       // same wave, same XCD as the start stamps
            }
        };
#endif
        if (WITH_LONG && approx && MD_APPROX_MFMA) {
            md_approx_tile(fetch, reinterpret_cast<float *>(s_row), off, M, jt, approx_opt);
#ifdef CM3D_DIAG
            stamp(t_tile);
#endif
            continue;
        }
        // Rows are staged MD_STAGE (256) at a time: all their index loads, then all their point gathers are in
        // flight together (two memory latencies per 256 rows; a 64-row pipeline left the longest lists -- the
        // waves the kernel waits for -- bound by one dependent gather per chunk).
        for (int i0 = 0; i0 < M; i0 += MD_STAGE) {
            __builtin_amdgcn_wave_barrier();                  // the previous rows' readers are done
            float4 g[MD_STAGE / 64];
            bool rows_safe = true, rows_scal = true;
#pragma unroll
            for (int c = 0; c < MD_STAGE / 64; ++c)
                g[c] = fetch(off + min(i0 + c * 64 + lane, M - 1));
#pragma unroll
            for (int c = 0; c < MD_STAGE / 64; ++c) {
                if (i0 + c * 64 + lane < M) {
                    float4 r = g[c];
                    r.w = (r.x * r.x + r.y * r.y) + r.z * r.z;
                    rows_safe &= md_row_safe(r.w);
                    rows_scal &= mda_point_ok(r.x, r.y, r.z, r.w);
                    // the expansion branch only ever needs -2x, -2y, -2z of a row (exact products)
                    if (!direct) { r.x = -2.0f * r.x; r.y = -2.0f * r.y; r.z = -2.0f * r.z; }
                    g[c] = r;
                }
            }
            // this chunk in the scaled form (md_rows<.., SCALED>)?  (uniform)
            const bool scaled = MD_SCALED_ROUTES && !direct && !approx && col_scal && !__ballot(!rows_scal);
#pragma unroll
            for (int c = 0; c < MD_STAGE / 64; ++c) {
                if (i0 + c * 64 + lane < M) {
                    float4 r = g[c];
                    if (scaled) r = make_float4(r.x * MDA_S, r.y * MDA_S, r.z * MDA_S, r.w * MDA_S2);
                    md_stage(s_row, c * 64 + lane, r);
                }
            }
            // The expansion route takes 8 rows per step: the last chunk is filled up to a multiple of 8 with rows whose value is
            // -inf for every column (0 * x + ... + (-inf) + n_j), i.e. +0 after clamp_min_(0), a root of +0 and s + 0 = s: the
            // loop for the odd rows (a third of the instructions of a 100-point list) is never entered.
            int cnt = min(MD_STAGE, M - i0);
            if (!direct) {
                const int pad = (8 - (cnt & 7)) & 7;
                if (lane < pad) md_stage(s_row, cnt + lane, make_float4(0.0f, 0.0f, 0.0f, -INFINITY));
                cnt += pad;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef CM3D_DIAG
            if (diag && i0 == 0) t_staged = md_now();
#endif
            const bool safe = col_safe && !__ballot(!rows_safe);
            if (scaled)         // s goes in and comes out in its own units: the powers of two are exact both ways
                s = md_rows<false, false, true, true>(s_row, cnt, qx * MDA_S, qy * MDA_S, qz * MDA_S, qn * MDA_S2, s * MDA_S) * 65536.0f;
            else
                s = direct ? md_rows<true>(s_row, cnt, qx, qy, qz, qn, s)
                           : (approx ? md_rows<false, true>(s_row, cnt, qx, qy, qz, qn, s)
                                     : (safe ? md_rows<false, false, true>(s_row, cnt, qx, qy, qz, qn, s) : md_rows<false>(s_row, cnt, qx, qy, qz, qn, s)));
        }
        if (approx) {
            if (act) approx_opt[off + j] = s;
            continue;
        }
        if (act && colsum_opt) colsum_opt[off + j] = s;
        // first minimum over the tile's columns (torch.argmin: NaN counts as minimal, first wins)
        float bs = act ? s : INFINITY;
        int bj = act ? j : 0x7FFFFFFF;
        auto better = [](float s1, int j1, float s2, int j2) {   // is (s1,j1) ahead of (s2,j2)?
            const bool n1 = s1 != s1, n2 = s2 != s2;
            if (n1 != n2) return n1;
            if (!n1 && s1 != s2) return s1 < s2;
            return j1 < j2;
        };
        {   // (DPP moves instead of ds_bpermute round trips: common.h)
            const Cm3dValIdx red = cm3d_wave_reduce_t(Cm3dValIdx{bs, bj}, [&](Cm3dValIdx a_, Cm3dValIdx b_) { return better(b_.s, b_.j, a_.s, a_.j) ? b_ : a_; });
            bs = red.s; bj = red.j;
        }
        if (lane == 0) { tile_best[d.t].s = bs; tile_best[d.t].j = bj; }
#ifdef CM3D_DIAG
        stamp(t_staged);
#endif
    }
}

__global__ __launch_bounds__(256) void k_medoid_reduce(const float4 *__restrict__ points, const int32_t *__restrict__ pt_off,
                                                       const int32_t *__restrict__ mask_frame, int n_masks,
                                                       const int32_t *__restrict__ hit_off,
                                                       const int32_t *__restrict__ tile_off,
                                                       const int32_t *__restrict__ hit_row, int idx_cap,
                                                       const TileBest *__restrict__ tile_best, int tile_cap, int two_pass,
                                                       const TileDesc *__restrict__ desc, int32_t *__restrict__ long_list,
                                                       int32_t *__restrict__ medoid_pos, float *__restrict__ centroid)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    // k_medoid_long's masks go on its list (long_list[0] = how many, zeroed by the tile kernel; then the mask numbers, in whatever
    // order the atomics make it): its workgroups then share the long masks evenly -- walking ALL masks in strides of the grid gave
    // a workgroup as many long masks as chance would have it (C1: 1.1 on average, 5 or 6 at most, and the launch lasts as long as those)
    // (the tile kernel's own rule: lists of more than MD_LONG_MIN points, in a batch whose longest has more than MD_BATCH_LONG)
    const bool is_long = two_pass && md_batch_long(desc, min(tile_off[n_masks], tile_cap)) && m < n_masks && md_two_pass(hit_off[m + 1] - hit_off[m]);
    {
        const uint64_t lm = __ballot(is_long);
        if (lm) {
            int base = 0;
            if (cm3d_lane() == 0) base = atomicAdd(&long_list[0], (int)__popcll(lm));
            base = __builtin_amdgcn_readfirstlane(base);
            if (is_long) long_list[1 + base + cm3d_mbcnt(lm)] = m;
        }
    }
    if (m >= n_masks || is_long) return;
    const int t0 = tile_off[m], t1 = min(tile_off[m + 1], tile_cap);
    int bj = -1;
    float bs = 0.f;
    for (int tb = t0; tb < t1; tb += 8) {             // (eight tiles per round trip, unconditional loads of a clamped tile: one load per iteration before)
        TileBest bb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) bb[u] = tile_best[min(tb + u, t1 - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const TileBest b = bb[u];
            if (tb + u >= t1 || b.j == 0x7FFFFFFF) continue;
            const bool bn = b.s != b.s, cn = bs != bs;
            if (bj < 0 || (bn && !cn) || (!bn && !cn && b.s < bs)) { bs = b.s; bj = b.j; }
        }
    }
    medoid_pos[m] = bj;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (bj >= 0 && hit_off[m] + bj < idx_cap) {
        const float4 p = hit_row ? points[pt_off[mask_frame[m]] + hit_row[hit_off[m] + bj]] : points[hit_off[m] + bj];
        cx = p.x; cy = p.y; cz = p.z;
    }
    centroid[3 * m + 0] = cx; centroid[3 * m + 1] = cy; centroid[3 * m + 2] = cz;
}

// Long lists, second pass.  k_medoid_tiles left A_j, the column sums with v_sqrt_f32 roots in place of the correctly
// rounded ones, summed in some order.  With t_i the exact and t'_i the approximate terms (|t'_i - t_i| <= 2^-23 t_i, or
// <= 1e-15 below 1e-30) and float32 sums of non-negative terms (each add rounds by <= 2^-24 of a partial sum, and partial
// sums never exceed the final sum, whatever the order), |A_j - S_j| <= (2^-23 sum_i t_i + M 1e-15) + M 2^-24 (A_j + S_j) up to
// factors 1 + O(M 2^-24); solved for A_j this stays below E_j = 1.01 (M + 2) 2^-23 A_j + M 2e-15 for M < 10^5 (M 2e-14 is what the code allows: the
// scaled form of the first pass, md_approx_tile, may lose 6e-15 per term to underflow inside a chain).  A column can
// only be the (first) minimum of the exact sums if A_j - E_j <= min_k (A_k + E_k); those few columns -- the points within
// centimetres of the medoid -- get their exact float32 sums here.  A non-finite A_j makes every column a candidate; lists
// of 10^5 points and more stay on the one-pass route.
// One WAVE per long mask (r04; a workgroup of four waves per mask before, one candidate per wave at a time, each sum a chain of
// v_readlane + v_add over all rows: 52 cycles per row and candidate).  The exact float32 sum of a column is M dependent additions in
// ascending row order whatever the machine; what can be shared is everything around them.  Up to MDL_GC candidates are settled in ONE
// walk over the rows: the 64 lanes take 64 consecutive rows and compute their exact distances to every candidate (two candidates per
// packed instruction), the distances are transposed through the wave's LDS slice (row c of `t` = candidate c's 64 terms), and lane c
// adds candidate c's terms in row order -- the very additions, in the very order, of the column-per-lane loop, for all candidates at
// once.  Rows past the end contribute +0 (s + 0 = s for the non-negative or NaN sums that occur).  More than MDL_MAXC candidates
// (lists full of duplicated points): 64 columns at a time, as k_medoid_tiles does.
#define MDL_WAVES 4                      // waves per workgroup, each with long masks of its own (no workgroup barrier anywhere)
#define MDL_MAXC 64                      // candidates the row-parallel route takes
#define MDL_GC 16                        // ... and settles per walk over the rows
#define MDL_AU 16                        // first-pass sums loaded per lane and round trip
#define MDL_TS 68                        // floats per candidate in the transposition buffer: 16-byte aligned rows, lanes c and c + 8 share banks

struct MdlLds {                          // one wave's slice
    union {
        float t[MDL_GC][MDL_TS];         // row-parallel route: t[c][r] = distance of row r of the current 64 to candidate c
        float4 row[MD_STAGE];            // column loop: staged rows
    } u;
    float4 q[MDL_GC];                    // the candidates' points (x, y, z, squared norm)
    int list[MDL_MAXC];                  // the candidates' columns, ascending
    int cand[64];                        // column loop: candidates waiting
};
static_assert(sizeof(float) * MDL_GC * MDL_TS >= sizeof(float4) * MD_STAGE, "the staged rows of the column loop live in the transposition buffer");

// exact float32 distances of this lane's row (m2 = -2 p, n = |p|^2) to two candidates: md_pair2<false>'s chain + the correctly rounded root
static __device__ __forceinline__ f2 md_exact_dist2(float m2x, float m2y, float m2z, float n, float4 qa, float4 qb)
{
    f2 acc = (f2){m2x, m2x} * (f2){qa.x, qb.x};
    acc = PK_FMA(((f2){m2y, m2y}), ((f2){qa.y, qb.y}), acc);
    acc = PK_FMA(((f2){m2z, m2z}), ((f2){qa.z, qb.z}), acc);
    acc = acc + (f2){n, n};
    acc = acc + (f2){qa.w, qb.w};
    acc = (f2){fmaxf(acc.x, 0.0f), fmaxf(acc.y, 0.0f)};
    if (__ballot(!md_sqrt_okz(acc.x) || !md_sqrt_okz(acc.y))) return (f2){sqrtf(acc.x), sqrtf(acc.y)};       // both forms are the correctly rounded root
    return md_sqrt_core2z(acc);
}

template <bool INDEXED>
__global__ __launch_bounds__(64 * MDL_WAVES, 3) void k_medoid_long(const float4 *__restrict__ points, const int32_t *__restrict__ pt_off,
                                                                const int32_t *__restrict__ mask_frame, int n_masks,
                                                                const int32_t *__restrict__ hit_off, const int32_t *__restrict__ hit_row,
                                                                int idx_cap, const float *__restrict__ approx, const int32_t *__restrict__ long_list,
                                                                int32_t *__restrict__ medoid_pos, float *__restrict__ centroid)
{
    __shared__ MdlLds s_lds[MDL_WAVES];
    const int lane = cm3d_lane();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    MdlLds &L = s_lds[wave];
    // a wave takes entries blockIdx.x + wave * gridDim.x, + gridDim.x * MDL_WAVES, ... of the list of long masks (k_medoid_reduce): a short list
    // goes to the first waves of many workgroups, i.e. to many CUs
    const int n_long = min(long_list[0], n_masks);
#ifdef CM3D_DIAG
    if (g_md_diag & 256) return;                                           // ablation: no second pass
#endif
    auto wave_sync = [] {                                                  // this wave's LDS writes before its reads (and reads before the next writes)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto better = [](float s1, int j1, float s2, int j2) {   // is (s1,j1) ahead of (s2,j2)?  (torch.argmin order)
        const bool n1 = s1 != s1, n2 = s2 != s2;
        if (n1 != n2) return n1;
        if (!n1 && s1 != s2) return s1 < s2;
        return j1 < j2;
    };
    for (int li = blockIdx.x + wave * gridDim.x; li < n_long; li += gridDim.x * MDL_WAVES) {
        const int m = __builtin_amdgcn_readfirstlane(long_list[1 + li]);
        const int off = __builtin_amdgcn_readfirstlane(hit_off[m]), M = __builtin_amdgcn_readfirstlane(hit_off[m + 1]) - off;
        if (!md_two_pass(M) || off + M > idx_cap) continue;
        const float4 *P = INDEXED ? points + pt_off[mask_frame[m]] : points;
        auto fetch = [&](int q) { return INDEXED ? P[hit_row[q]] : P[q]; };                  // (loads are unconditional, of a clamped row: see k_medoid_tiles)
        const float *A = approx + off;
        const double rel = 1.01 * (double)(M + 2) * 1.1920928955078125e-07, abs_e = (double)M * 2e-14;
        // threshold = min_k (A_k + E_k); anything non-finite -> every column is a candidate
        // (eight loads in flight per round, unconditional -- a lane past the end re-reads the last sum --: one load per round trip made
        // the two walks over A longer than the walk over the rows)
        double thr = INFINITY;
        bool all = false;
        for (int j0 = 0; j0 < M; j0 += 64 * MDL_AU) {
            float av[MDL_AU];
#pragma unroll
            for (int k = 0; k < MDL_AU; ++k) av[k] = A[min(j0 + k * 64 + lane, M - 1)];
#pragma unroll
            for (int k = 0; k < MDL_AU; ++k) {
                const double a = (double)av[k];
                if (!(a >= 0.0 && a < 1e300)) all = true;
                thr = fmin(thr, a + (a * rel + abs_e));
            }
        }
        thr = cm3d_wave_reduce_t(thr, [](double a_, double b_) { return fmin(a_, b_); });
        all = __ballot(all) != 0;
        // the candidates, in ascending order (the final choice is a total order on (sum, index): any order would do)
        int C = 0;
        wave_sync();                                                       // the mask before is done with `list`
        for (int j0 = 0; j0 < M; j0 += 64 * MDL_AU) {
            float av[MDL_AU];
#pragma unroll
            for (int k = 0; k < MDL_AU; ++k) av[k] = A[min(j0 + k * 64 + lane, M - 1)];
#pragma unroll
            for (int k = 0; k < MDL_AU; ++k) {
                const int j = j0 + k * 64 + lane;
                const double a = (double)av[k];
                const bool cand = j < M && (all || (a - (a * rel + abs_e) <= thr));
                const uint64_t cm = __ballot(cand);
                if (cand) {
                    const int pos = C + cm3d_mbcnt(cm);
                    if (pos < MDL_MAXC) L.list[pos] = j;
                }
                C += (int)__popcll(cm);
            }
        }
        float best_s = INFINITY;
        int best_j = 0x7FFFFFFF;
        if (C <= MDL_MAXC) {
            for (int g0 = 0; g0 < C; g0 += MDL_GC) {
                const int Cg = min(MDL_GC, C - g0);
                wave_sync();
                if (lane < MDL_GC) {                                       // (a group's unused slots: the origin -- finite terms nobody adds up)
                    float4 q = fetch(off + L.list[g0 + min(lane, Cg - 1)]);
                    q.w = (q.x * q.x + q.y * q.y) + q.z * q.z;
                    if (lane >= Cg) q = make_float4(0.f, 0.f, 0.f, 0.f);
                    L.q[lane] = q;
                }
                wave_sync();
                float s = 0.0f;                                            // lane c < Cg: the sum of candidate g0 + c
                float4 p = fetch(off + min(lane, M - 1)), p1 = fetch(off + min(64 + lane, M - 1));
                for (int i0 = 0; i0 < M; i0 += 64) {
                    const float4 pn = fetch(off + min(i0 + 128 + lane, M - 1));      // two rounds of rows ahead of the chains (rows past the end: the last row, never used)
                    const bool valid = i0 + lane < M;
                    const float n = (p.x * p.x + p.y * p.y) + p.z * p.z;
                    const float m2x = -2.0f * p.x, m2y = -2.0f * p.y, m2z = -2.0f * p.z;
                    for (int c = 0; c < Cg; c += 2) {                      // (uniform; c + 1 <= MDL_GC - 1)
                        const f2 d = md_exact_dist2(m2x, m2y, m2z, n, L.q[c], L.q[c + 1]);
                        L.u.t[c][lane] = valid ? d.x : 0.0f;
                        L.u.t[c + 1][lane] = valid ? d.y : 0.0f;
                    }
                    wave_sync();
                    if (lane < MDL_GC) {
                        const float4 *tr = reinterpret_cast<const float4 *>(&L.u.t[lane][0]);
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            const float4 v = tr[k];
                            s = s + v.x; s = s + v.y; s = s + v.z; s = s + v.w;
                        }
                    }
                    wave_sync();
                    p = p1; p1 = pn;
                }
                const float bs = lane < Cg ? s : INFINITY;
                const int bj = lane < Cg ? L.list[g0 + lane] : 0x7FFFFFFF;
                // (DPP moves instead of ds_bpermute round trips: common.h)
                const Cm3dValIdx red = cm3d_wave_reduce_t(Cm3dValIdx{bs, bj}, [&](Cm3dValIdx a_, Cm3dValIdx b_) { return better(b_.s, b_.j, a_.s, a_.j) ? b_ : a_; });
                if (red.j != 0x7FFFFFFF && (best_j == 0x7FFFFFFF || better(red.s, red.j, best_s, best_j))) { best_s = red.s; best_j = red.j; }
            }
        } else {
            // many candidates: 64 columns at a time, column per lane (the row loop of k_medoid_tiles)
            float4 *s_row = L.u.row;
            int *s_cand = L.cand;
            int nc = 0;                                   // candidates waiting in s_cand (uniform)
            for (int j0 = 0; j0 < M + 64; j0 += 64) {     // one extra round flushes the tail
                const int j = j0 + lane;
                bool cand = false;
                if (j < M) {
                    const double a = (double)A[j];
                    cand = all || (a - (a * rel + abs_e) <= thr);
                }
                const uint64_t cm = __ballot(cand);
                const int add = (int)__popcll(cm);
                const bool last = j0 >= M;
                if (nc + add > 64 || (last && nc > 0)) {
                    // exact sums of the waiting candidates: lane l owns column s_cand[l]
                    wave_sync();
                    const bool act = lane < nc;
                    const int cj = act ? s_cand[lane] : 0;
                    float qx = 0.f, qy = 0.f, qz = 0.f, qn = 0.f;
                    {
                        const float4 q = fetch(off + cj);
                        if (act) {
                            qx = q.x; qy = q.y; qz = q.z;
                            qn = (q.x * q.x + q.y * q.y) + q.z * q.z;
                        }
                    }
                    float s = 0.f;
                    for (int i0 = 0; i0 < M; i0 += MD_STAGE) {
                        __builtin_amdgcn_wave_barrier();
                        float4 g[MD_STAGE / 64];
#pragma unroll
                        for (int c = 0; c < MD_STAGE / 64; ++c)
                            g[c] = fetch(off + min(i0 + c * 64 + lane, M - 1));
#pragma unroll
                        for (int c = 0; c < MD_STAGE / 64; ++c) {
                            if (i0 + c * 64 + lane < M) {
                                float4 r = g[c];
                                r.w = (r.x * r.x + r.y * r.y) + r.z * r.z;
                                r.x = -2.0f * r.x; r.y = -2.0f * r.y; r.z = -2.0f * r.z;
                                md_stage(s_row, c * 64 + lane, r);
                            }
                        }
                        wave_sync();
                        s = md_rows<false>(s_row, min(MD_STAGE, M - i0), qx, qy, qz, qn, s);
                    }
                    const float bs = act ? s : INFINITY;
                    const int bj = act ? cj : 0x7FFFFFFF;
                    const Cm3dValIdx red = cm3d_wave_reduce_t(Cm3dValIdx{bs, bj}, [&](Cm3dValIdx a_, Cm3dValIdx b_) { return better(b_.s, b_.j, a_.s, a_.j) ? b_ : a_; });
                    if (red.j != 0x7FFFFFFF && (best_j == 0x7FFFFFFF || better(red.s, red.j, best_s, best_j))) { best_s = red.s; best_j = red.j; }
                    nc = 0;
                    __builtin_amdgcn_wave_barrier();
                }
                if (cand) s_cand[nc + cm3d_mbcnt(cm)] = j;
                nc += add;
            }
        }
        if (lane == 0) {
            int bj = best_j == 0x7FFFFFFF ? -1 : best_j;
#ifdef CM3D_DIAG
            if (g_md_diag & 2048) bj = C;                                  // diagnostic: the number of candidates in place of the position
#endif
            medoid_pos[m] = bj;
            float cx = 0.f, cy = 0.f, cz = 0.f;
            if (bj >= 0) {
                const float4 p = fetch(off + bj);
                cx = p.x; cy = p.y; cz = p.z;
            }
            centroid[3 * m + 0] = cx; centroid[3 * m + 1] = cy; centroid[3 * m + 2] = cz;
        }
    }
}

extern "C" int64_t cm3d_tile_work_bytes(int32_t n_masks, int32_t idx_cap)
{
    if (n_masks <= 0 || idx_cap <= 0) return 0;
    return md_tile_cap(n_masks, idx_cap) * (int64_t)sizeof(TileDesc);
}

extern "C" int64_t cm3d_medoid_workspace_bytes(int32_t n_masks, int32_t idx_cap)
{
    if (n_masks <= 0 || idx_cap <= 0) return 0;
    // work list + per-tile results + the first-pass column sums of the long lists
    // ... + the list of the long masks (count, then mask numbers: k_medoid_reduce -> k_medoid_long)
    return md_tile_cap(n_masks, idx_cap) * (int64_t)(sizeof(TileBest) + sizeof(TileDesc)) + (int64_t)idx_cap * (int64_t)sizeof(float) +
           ((int64_t)n_masks + 4) * (int64_t)sizeof(int32_t);
}

extern "C" int cm3d_medoid(const float *points, const int32_t *pt_off, const int32_t *mask_frame, int32_t n_masks,
                           const int32_t *hit_off, const int32_t *tile_off, const int32_t *hit_row, int32_t idx_cap,
                           const int32_t *tile_work, int32_t *medoid_pos, float *centroid, float *colsum_opt, void *workspace,
                           int64_t workspace_bytes, cm3d_stream_t stream)
{
    return cm3d_medoid2(points, pt_off, mask_frame, n_masks, hit_off, tile_off, hit_row, idx_cap, tile_work, medoid_pos, centroid, colsum_opt,
                        workspace, workspace_bytes, 0, nullptr, stream);
}

extern "C" int cm3d_medoid2(const float *points, const int32_t *pt_off, const int32_t *mask_frame, int32_t n_masks,
                            const int32_t *hit_off, const int32_t *tile_off, const int32_t *hit_row, int32_t idx_cap,
                            const int32_t *tile_work, int32_t *medoid_pos, float *centroid, float *colsum_opt, void *workspace,
                            int64_t workspace_bytes, int32_t flags, int32_t *feedback, cm3d_stream_t stream)
{
    if (!points || !hit_off || !tile_off || !medoid_pos || !centroid || !workspace) return CM3D_ERR_ARG;
    if (hit_row && (!pt_off || !mask_frame)) return CM3D_ERR_ARG;           // hit_row == NULL: `points` is the hit_xyz array
    if (n_masks <= 0 || idx_cap <= 0) return CM3D_ERR_ARG;
    if (workspace_bytes < cm3d_medoid_workspace_bytes(n_masks, idx_cap)) return CM3D_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t tile_cap64 = md_tile_cap(n_masks, idx_cap);
    const int tile_cap = (int)(tile_cap64 > 0x7FFFFFFF ? 0x7FFFFFFF : tile_cap64);
    TileDesc *own = (TileDesc *)workspace;
    TileBest *best = (TileBest *)(own + tile_cap);
    // two passes for long lists unless the caller wants every exact column sum (colsum_opt) or CM3D_MD_TWO_PASS=0
    static int two_pass_env = -1;
    if (two_pass_env < 0) { const char *e = getenv("CM3D_MD_TWO_PASS"); two_pass_env = e ? atoi(e) : 1; }
    // (flags & 1: the caller expects no long list in this batch -- the one-pass route is exact for every length, so a wrong
    // expectation costs time, never a result -- and two launches that would find nothing to do are not made)
    float *approx = (two_pass_env && !colsum_opt && !(flags & 1)) ? (float *)(best + tile_cap) : nullptr;
    int32_t *long_list = (int32_t *)((float *)(best + tile_cap) + idx_cap);         // [0] = number of long masks, then their numbers
    const TileDesc *desc = (const TileDesc *)tile_work;
    if (!desc) {                                   // no work list from cm3d_compact_hits: build it here
        hipLaunchKernelGGL(k_medoid_desc, dim3(1), dim3(1024), 0, st, n_masks, hit_off, tile_off, idx_cap, tile_cap, own);
        CM3D_CHECK_LAUNCH();
        desc = own;
    }
    // The number of tiles is only known on the device.  A workgroup without a tile leaves after one load, but the chip still has to
    // dispatch it -- twice, once per instantiation: on the headline shape (5120 masks, 8 k tiles) a grid of 4096 against 2048 is
    // 1.6 % of the pass with three batches in flight; batches of long lists (C1, C5: 50-400 k tiles) want the large grid, whose
    // workgroups the hardware hands out as others finish (C5: 2048 workgroups +2 %, 1024 +7 % on the stage).  Half a workgroup
    // per mask lies between the two.
    int grid = (tile_cap + MD_WAVES - 1) / MD_WAVES;
    int gmax = n_masks / 2 < 1024 ? 1024 : (n_masks / 2 > 4096 ? 4096 : n_masks / 2);
    if (const char *e = getenv("CM3D_MD_GRID")) gmax = atoi(e);
    if (grid > gmax) grid = gmax;
    // (hit_row == NULL, the product's call: `points` holds the listed points themselves, cm3d_compact_hits' hit_xyz)
    auto tiles_light = hit_row ? k_medoid_tiles<false, true> : k_medoid_tiles<false, false>;
    auto tiles_heavy = hit_row ? k_medoid_tiles<true, true> : k_medoid_tiles<true, false>;
    auto second_pass = hit_row ? k_medoid_long<true> : k_medoid_long<false>;
    hipLaunchKernelGGL(tiles_light, dim3(grid), dim3(MD_THREADS), 0, st, (const float4 *)points, pt_off, mask_frame, n_masks,
                       tile_off, hit_row, desc, best, tile_cap, colsum_opt, approx, long_list, feedback);
    CM3D_CHECK_LAUNCH();
    if (approx) {             // (without a first pass -- colsum_opt, CM3D_MD_TWO_PASS=0 -- the light instantiation takes every batch)
        hipLaunchKernelGGL(tiles_heavy, dim3(grid), dim3(MD_THREADS), 0, st, (const float4 *)points, pt_off, mask_frame, n_masks,
                           tile_off, hit_row, desc, best, tile_cap, colsum_opt, approx, long_list, feedback);
        CM3D_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_medoid_reduce, dim3((n_masks + 255) / 256), dim3(256), 0, st, (const float4 *)points, pt_off, mask_frame,
                       n_masks, hit_off, tile_off, hit_row, idx_cap, best, tile_cap, approx ? 1 : 0, desc, long_list, medoid_pos, centroid);
    CM3D_CHECK_LAUNCH();
    if (approx) {
        hipLaunchKernelGGL(second_pass, dim3(n_masks < 1024 ? n_masks : 1024), dim3(64 * MDL_WAVES), 0, st, (const float4 *)points, pt_off, mask_frame, n_masks, hit_off,
                           hit_row, idx_cap, approx, long_list, medoid_pos, centroid);
        CM3D_CHECK_LAUNCH();
    }
    return CM3D_OK;
}

// ---------------------------------------------------------------------------
// Diagnostic: every float32 bit pattern in [first_bits, last_bits] (positive, inside md_sqrt_ok's domain) through
// md_sqrt_core2, md_sqrt_core2z (next to a +0, which must come out as +0), md_sqrt_core2_ref and sqrtf(); counts the values
// on which they are not all bit-identical.
__global__ __launch_bounds__(256) void k_selftest_sqrt(uint32_t first_bits, uint32_t last_bits, unsigned long long *n_bad,
                                                        uint32_t *first_bad)
{
    unsigned long long bad = 0;
    const unsigned long long span = (unsigned long long)last_bits - first_bits + 1ull;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < span;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t b = first_bits + (uint32_t)i;
        const float x = __uint_as_float(b);
        if (!md_sqrt_ok(x)) continue;
        const float want = sqrtf(x);
        const float a = md_sqrt_core2((f2){x, x}).x, c = md_sqrt_core2_ref((f2){x, x}).y, z = md_sqrt_core2z((f2){0.0f, x}).y;
        const uint32_t z0 = __float_as_uint(md_sqrt_core2z((f2){0.0f, x}).x);           // +0 next to x: +0
        if (__float_as_uint(a) != __float_as_uint(want) || __float_as_uint(c) != __float_as_uint(want) ||
            __float_as_uint(z) != __float_as_uint(want) || z0 != 0u) {
            ++bad;
            atomicMin(first_bad, b);
        }
    }
    if (bad) atomicAdd(n_bad, bad);
}

extern "C" int cm3d_selftest_sqrt(uint32_t first_bits, uint32_t last_bits, uint64_t *n_bad, uint32_t *first_bad, cm3d_stream_t stream)
{
    if (!n_bad || !first_bad || last_bits < first_bits || (last_bits >> 31)) return CM3D_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(n_bad, 0, sizeof(uint64_t), st) != hipSuccess) return CM3D_ERR_LAUNCH;
    if (hipMemsetAsync(first_bad, 0xFF, sizeof(uint32_t), st) != hipSuccess) return CM3D_ERR_LAUNCH;
    hipLaunchKernelGGL(k_selftest_sqrt, dim3(8192), dim3(256), 0, st, first_bits, last_bits, (unsigned long long *)n_bad, first_bad);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}

// ---------------------------------------------------------------------------
// Diagnostic: the squared distances of k_medoid_approx (three v_mfma_f32_32x32x2_f32) against the vector pipe's fma chain
// (md_pair2<false>, the form the exact route uses) on pseudo-random points at global-frame magnitudes: every one of the
// 32 x 32 values of every tile must be bit-identical.  n_bad (device) = number of differing values.
__global__ __launch_bounds__(64) void k_selftest_mfma(uint64_t seed, int tiles_per_wave, unsigned long long *n_bad)
{
    const int lane = cm3d_lane();
    const bool lo = lane < 32;
    uint64_t x = seed ^ (0x9E3779B97F4A7C15ull * ((uint64_t)blockIdx.x * 64 + lane + 1));
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    auto coord = [&](float centre, float spread) { return centre + spread * ((float)(next() & 0xFFFFFF) * (1.0f / 8388608.0f) - 1.0f); };
    unsigned long long bad = 0;
    for (int t = 0; t < tiles_per_wave; ++t) {
        // lane l < 32 owns row l and column l of this tile; magnitudes from a vehicle-frame cloud (x 0.01) over nuScenes' global
        // frame (x 1, x 2.5: the far corners of its maps) to 10 km and beyond (x 6, x 20), objects a few metres wide
        const float mags[6] = {1.0f, 0.01f, 2.5f, 6.0f, 1.0f, 20.0f};
        const float mag = mags[t % 6];
        const float cx = mag * (300.0f + 40.0f * (float)(t % 37)), cy = mag * (900.0f + 25.0f * (float)(t % 53)), spread = (t & 1) ? 3.0f : 40.0f;
        const float rx = coord(cx, spread), ry = coord(cy, spread), rz = coord(1.0f, 2.0f);
        const float qx = coord(cx, spread), qy = coord(cy, spread), qz = coord(1.0f, 2.0f);
        const float rn = (rx * rx + ry * ry) + rz * rz, qn = (qx * qx + qy * qy) + qz * qz;
        // operands as k_medoid_approx lays them out (values of lane l & 31 in both halves)
        const float Rx = __shfl(rx, lane & 31, 64), Ry = __shfl(ry, lane & 31, 64), Rz = __shfl(rz, lane & 31, 64), Rn = __shfl(rn, lane & 31, 64);
        const float Qx = __shfl(qx, lane & 31, 64), Qy = __shfl(qy, lane & 31, 64), Qz = __shfl(qz, lane & 31, 64), Qn = __shfl(qn, lane & 31, 64);
        f16v c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(lo ? -2.0f * Rx : -2.0f * Ry, lo ? Qx : Qy, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(lo ? -2.0f * Rz : Rn, lo ? Qz : 1.0f, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(lo ? 1.0f : 0.0f, lo ? Qn : 0.0f, c, 0, 0, 0);
        // md_approx_step's form: one k per v_mfma_f32_32x32x1_2b_f32, block lane / 32 = rows 0-31 against the columns of THAT half --
        // here lane l's own column point (all 64 lanes hold one), so block 1 pairs rows 0-31 with the points of lanes 32-63
        f32v c2;
#pragma unroll
        for (int q = 0; q < 32; ++q) c2[q] = 0.0f;
        c2 = __builtin_amdgcn_mfma_f32_32x32x1f32(-2.0f * Rx, qx, c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x1f32(-2.0f * Ry, qy, c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x1f32(-2.0f * Rz, qz, c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x1f32(Rn, 1.0f, c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x1f32(1.0f, qn, c2, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);          // C/D map of the 32x32 shapes; column = lane & 31
            const float ix = __shfl(rx, row, 64), iy = __shfl(ry, row, 64), iz = __shfl(rz, row, 64), in_ = __shfl(rn, row, 64);
            float acc = (-2.0f * ix) * Qx;                                      // the reference's chain
            acc = fmaf(-2.0f * iy, Qy, acc);
            acc = fmaf(-2.0f * iz, Qz, acc);
            acc = fmaf(in_, 1.0f, acc);
            acc = fmaf(1.0f, Qn, acc);
            // (a first product of -0 would come out of the MFMA as +0: both are clamped to 0 before the root)
            if (__float_as_uint(acc) != __float_as_uint(c[q]) && !(acc == 0.0f && c[q] == 0.0f)) ++bad;
            if (__float_as_uint(acc) != __float_as_uint(c2[q]) && !(acc == 0.0f && c2[q] == 0.0f)) ++bad;      // block 0: columns of lanes 0-31
            // block 1: the same rows against the column points of lanes 32-63
            const float Px = __shfl(qx, 32 + (lane & 31), 64), Py = __shfl(qy, 32 + (lane & 31), 64), Pz = __shfl(qz, 32 + (lane & 31), 64),
                        Pn = __shfl(qn, 32 + (lane & 31), 64);
            float acc1 = (-2.0f * ix) * Px;
            acc1 = fmaf(-2.0f * iy, Py, acc1);
            acc1 = fmaf(-2.0f * iz, Pz, acc1);
            acc1 = fmaf(in_, 1.0f, acc1);
            acc1 = fmaf(1.0f, Pn, acc1);
            if (__float_as_uint(acc1) != __float_as_uint(c2[16 + q]) && !(acc1 == 0.0f && c2[16 + q] == 0.0f)) ++bad;
        }
    }
    // The scaled routes (md_rows<.., SCALED>, md_approx_tile) rest on two more instruction-level facts, checked here as well: the
    // output clamp of v_pk_add_f32 and of v_sqrt_f32 holds results to [0, 1], turns NaN, negative values and -inf into 0, and leaves
    // everything inside the range (denormals included) as it is.
    if (blockIdx.x == 0) {
        const float in_a[8] = {0.25f, -0.5f, 1.5f, -1e-20f, __int_as_float(0x7FC00000), -INFINITY, 3e-39f, 0.999999f};
        const float in_b[8] = {0.25f, 0.25f, 0.0f, 0.0f, 0.0f, 0.1f, 0.0f, 1e-7f};
        const float a = in_a[lane & 7], b = in_b[lane & 7];
        const float plain = a + b;
        const float want = plain != plain ? 0.0f : (plain < 0.0f ? 0.0f : (plain > 1.0f ? 1.0f : plain));
        const f2 got = md_pk_add_clamp01((f2){a, b}, (f2){b, a});
        if (__float_as_uint(got.x) != __float_as_uint(want) || __float_as_uint(got.y) != __float_as_uint(want)) ++bad;
        const float x = (lane & 1) ? -(float)(lane + 1) * 0.01f : (float)(lane + 1) * 0.015f;        // negative: 0; positive (< 1): the plain root
        const float wr = x < 0.0f ? 0.0f : md_vsqrt(x);
        float got_r;                                                                                 // (the instruction itself, not what the compiler makes of the expression)
        asm volatile("v_sqrt_f32_e64 %0, %1 clamp" : "=v"(got_r) : "v"(x));
        if (__float_as_uint(got_r) != __float_as_uint(wr) || __float_as_uint(md_vsqrt_clamp01(x)) != __float_as_uint(wr)) ++bad;
    }
    if (bad) atomicAdd(n_bad, bad);
}

extern "C" int cm3d_selftest_mfma(uint64_t seed, int32_t tiles_per_wave, uint64_t *n_bad, cm3d_stream_t stream)
{
    if (!n_bad || tiles_per_wave <= 0) return CM3D_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(n_bad, 0, sizeof(uint64_t), st) != hipSuccess) return CM3D_ERR_LAUNCH;
    hipLaunchKernelGGL(k_selftest_mfma, dim3(2048), dim3(64), 0, st, seed, tiles_per_wave, (unsigned long long *)n_bad);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}
