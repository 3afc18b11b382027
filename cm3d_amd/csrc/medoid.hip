// a9: medoid of the in-mask points -- reference src/nuscenes/2d_to_3d.py:116-119 (get_medoid),
// gather at :620, use at :645-647.
// argmin_j sum_i D[i][j], D = torch.cdist(P, P) in float32:
//   M <= 25 : direct form   agg = fma(d,d,agg) over |a_k-b_k|, sqrt
//   M  > 25 : expansion     [-2x_i,-2y_i,-2z_i,n_i,1].[x_j,y_j,z_j,1,n_j] as a k-sequential fma
//             chain, clamp at 0, sqrt   (n = (x*x + y*y) + z*z)
// (the expansion's cancellation error at global-frame magnitudes is part
// of the reference's behaviour and is reproduced, not fixed.)
// Column j lives in one thread and is summed over rows i in ascending order, so the float32
// column sums do not depend on the launch geometry.  A tile = 256 columns of one mask; rows are
// staged through LDS 64 at a time and read back as wave-wide broadcasts.  VALU-bound
// (about 20 ops per pair incl. the IEEE sqrt); nothing M x M ever touches HBM.
#include "common.h"
#include <stdlib.h>

#define MD_WAVES 4                       // waves per workgroup, one tile each
#define MD_THREADS (MD_WAVES * 64)

// Correctly rounded float32 square root = sqrtf(), without the denormal pre-scaling hipcc emits around
// v_sqrt_f32: v_sqrt_f32 is within 1 ulp, the two fma residuals pick the right neighbour (0 maps to 0).
// Valid for x == 0 or 1e-30 <= x < 1e30; md_sqrt_ok() tells whether a value is in that domain, and the
// caller falls back to sqrtf() for the whole wave when any lane is not (never on real coordinates).
static __device__ __forceinline__ float md_sqrt_core(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float s_lo = __int_as_float(__float_as_int(s) - 1), s_hi = __int_as_float(__float_as_int(s) + 1);
    const float r_lo = fmaf(-s_lo, s, x), r_hi = fmaf(-s_hi, s, x);
    s = r_lo <= 0.0f ? s_lo : s;
    s = r_hi > 0.0f ? s_hi : s;
    return s;
}
static __device__ __forceinline__ bool md_sqrt_ok(float x) { return (x < 1.0e30f) & ((x >= 1.0e-30f) | (x == 0.0f)); }

struct TileBest { float s; int j; };
struct TileDesc { int m, off, M, jt, t; };  // mask, start in hit_idx, list length, tile index inside the mask, tile id

// One descriptor per 64-column tile of every index list, written in WORK order: tiles of the longest lists first
// (classes by tile count; a tile's cost is its list length), so that the waves which run longest start first and
// the short ones fill in behind them.  Results are indexed by the tile id t = tile_off[m] + jt, not by the work
// position, so the order has no influence on any output.  One workgroup.
#define MD_CLASSES 8
__global__ __launch_bounds__(1024) void k_medoid_desc(int n_masks, const int32_t *__restrict__ hit_off,
                                                      const int32_t *__restrict__ tile_off, int idx_cap, int tile_cap,
                                                      TileDesc *__restrict__ desc)
{
    __shared__ int s_hist[MD_CLASSES], s_cur[MD_CLASSES];
    if (threadIdx.x < MD_CLASSES) s_hist[threadIdx.x] = 0;
    __syncthreads();
    for (int m = threadIdx.x; m < n_masks; m += 1024) {
        const int t0 = tile_off[m], nt = max(0, min(tile_off[m + 1], tile_cap) - t0);
        if (nt > 0) atomicAdd(&s_hist[min(nt, MD_CLASSES) - 1], nt);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int c = MD_CLASSES - 1; c >= 0; --c) { s_cur[c] = run; run += s_hist[c]; }
    }
    __syncthreads();
    for (int m = threadIdx.x; m < n_masks; m += 1024) {
        const int t0 = tile_off[m], nt = max(0, min(tile_off[m + 1], tile_cap) - t0);
        if (nt <= 0) continue;
        const int off = hit_off[m];
        int M = hit_off[m + 1] - off;
        if (off + M > idx_cap) M = max(0, idx_cap - off);     // index capacity overflow: stay in bounds
        const int pos = atomicAdd(&s_cur[min(nt, MD_CLASSES) - 1], nt);
        for (int jt = 0; jt < nt; ++jt) desc[pos + jt] = TileDesc{m, off, M, jt, t0 + jt};
    }
}

// squared distance of one staged row to this lane's column
template <bool DIRECT>
static __device__ __forceinline__ float md_pair(const float4 r, float qx, float qy, float qz, float qn)
{
    if (DIRECT) {
        float dd = fabsf(r.x - qx);
        float agg = fmaf(dd, dd, 0.0f);
        dd = fabsf(r.y - qy); agg = fmaf(dd, dd, agg);
        dd = fabsf(r.z - qz); agg = fmaf(dd, dd, agg);
        return agg;
    }
    float acc = r.x * qx;                     // (-2 x_i) * x_j
    acc = fmaf(r.y, qy, acc);
    acc = fmaf(r.z, qz, acc);
    acc = fmaf(r.w, 1.0f, acc);
    acc = fmaf(1.0f, qn, acc);
    // clamp_min_(0); the in-image test only lets finite points into a mask, so acc is never NaN
    return fmaxf(acc, 0.0f);
}

// adds the distances of `cnt` staged rows to s, in ascending row order.  8 rows per step: the distance chains
// are independent (ILP, and the LDS reads of a step are issued together), only the adds into s are sequential
// -- which is what fixes the float32 sum.
template <bool DIRECT>
static __device__ __forceinline__ float md_rows(const float4 *s_row, int cnt, float qx, float qy, float qz, float qn, float s)
{
    constexpr int U = 8;
    int ii = 0;
    for (; ii + U <= cnt; ii += U) {
        float d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = md_pair<DIRECT>(s_row[ii + u], qx, qy, qz, qn);
        // md_sqrt_core's domain, tested on the extremes (a 0 on the diagonal sends its step to sqrtf)
        const float lo = fminf(fminf(fminf(d[0], d[1]), fminf(d[2], d[3])), fminf(fminf(d[4], d[5]), fminf(d[6], d[7])));
        const float hi = fmaxf(fmaxf(fmaxf(d[0], d[1]), fmaxf(d[2], d[3])), fmaxf(fmaxf(d[4], d[5]), fmaxf(d[6], d[7])));
        if (__ballot(!(lo >= 1.0e-30f && hi < 1.0e30f))) {
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = sqrtf(d[u]);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = md_sqrt_core(d[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) s = s + d[u];
    }
    for (; ii < cnt; ++ii) {
        const float v = md_pair<DIRECT>(s_row[ii], qx, qy, qz, qn);
        s = s + (__ballot(!md_sqrt_ok(v)) ? sqrtf(v) : md_sqrt_core(v));
    }
    return s;
}

// One wave = one tile = 64 columns of one mask; rows are staged 64 at a time through the wave's own
// LDS slice and read back as broadcasts.  No workgroup barrier: waves of a block are independent.
__global__ __launch_bounds__(MD_THREADS) void k_medoid_tiles(const float4 *__restrict__ points,
                                                              const int32_t *__restrict__ pt_off,
                                                              const int32_t *__restrict__ mask_frame, int n_masks,
                                                              const int32_t *__restrict__ tile_off,
                                                              const int32_t *__restrict__ hit_row,
                                                              const TileDesc *__restrict__ desc,
                                                              TileBest *__restrict__ tile_best, int tile_cap,
                                                              float *__restrict__ colsum_opt)
{
    __shared__ float4 s_row_all[MD_WAVES][64];
    const int wave = threadIdx.x >> 6, lane = cm3d_lane();
    float4 *s_row = s_row_all[wave];
    const int ntiles = min(tile_off[n_masks], tile_cap);
    for (int t = blockIdx.x * MD_WAVES + wave; t < ntiles; t += gridDim.x * MD_WAVES) {
        const TileDesc d = desc[t];
        // the descriptor is the same in every lane: keep it in scalar registers so that the loops below are
        // uniform control flow
        const int off = __builtin_amdgcn_readfirstlane(d.off), M = __builtin_amdgcn_readfirstlane(d.M);
        const int jt = __builtin_amdgcn_readfirstlane(d.jt);
        const float4 *P = points + pt_off[mask_frame[__builtin_amdgcn_readfirstlane(d.m)]];
        const int j = jt * 64 + lane;
        const bool act = j < M;
        float qx = 0.f, qy = 0.f, qz = 0.f, qn = 0.f;
        if (act) {
            const float4 q = P[hit_row[off + j]];
            qx = q.x; qy = q.y; qz = q.z;
            qn = (q.x * q.x + q.y * q.y) + q.z * q.z;
        }
        float s = 0.f;
        const bool direct = M <= 25;
        // rows are gathered one 64-row chunk ahead: the loads of chunk k+1 fly while chunk k is summed
        float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < M) nxt = P[hit_row[off + lane]];
        for (int i0 = 0; i0 < M; i0 += 64) {
            __builtin_amdgcn_wave_barrier();
            if (i0 + lane < M) {
                float4 r = nxt;
                r.w = (r.x * r.x + r.y * r.y) + r.z * r.z;
                // the expansion branch only ever needs -2x, -2y, -2z of a row (exact products)
                if (!direct) { r.x = -2.0f * r.x; r.y = -2.0f * r.y; r.z = -2.0f * r.z; }
                s_row[lane] = r;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (i0 + 64 + lane < M) nxt = P[hit_row[off + i0 + 64 + lane]];
            const int cnt = min(64, M - i0);
            s = direct ? md_rows<true>(s_row, cnt, qx, qy, qz, qn, s) : md_rows<false>(s_row, cnt, qx, qy, qz, qn, s);
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
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(bs, o, 64);
            const int oj = __shfl_xor(bj, o, 64);
            if (better(os, oj, bs, bj)) { bs = os; bj = oj; }
        }
        if (lane == 0) { tile_best[d.t].s = bs; tile_best[d.t].j = bj; }
    }
}

__global__ __launch_bounds__(256) void k_medoid_reduce(const float4 *__restrict__ points, const int32_t *__restrict__ pt_off,
                                                       const int32_t *__restrict__ mask_frame, int n_masks,
                                                       const int32_t *__restrict__ hit_off,
                                                       const int32_t *__restrict__ tile_off,
                                                       const int32_t *__restrict__ hit_row, int idx_cap,
                                                       const TileBest *__restrict__ tile_best, int tile_cap,
                                                       int32_t *__restrict__ medoid_pos, float *__restrict__ centroid)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_masks) return;
    const int t0 = tile_off[m], t1 = min(tile_off[m + 1], tile_cap);
    int bj = -1;
    float bs = 0.f;
    for (int t = t0; t < t1; ++t) {
        const TileBest b = tile_best[t];
        if (b.j == 0x7FFFFFFF) continue;
        const bool bn = b.s != b.s, cn = bs != bs;
        if (bj < 0 || (bn && !cn) || (!bn && !cn && b.s < bs)) { bs = b.s; bj = b.j; }
    }
    medoid_pos[m] = bj;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (bj >= 0 && hit_off[m] + bj < idx_cap) {
        const float4 p = points[pt_off[mask_frame[m]] + hit_row[hit_off[m] + bj]];
        cx = p.x; cy = p.y; cz = p.z;
    }
    centroid[3 * m + 0] = cx; centroid[3 * m + 1] = cy; centroid[3 * m + 2] = cz;
}

static inline int64_t md_tile_cap(int32_t n_masks, int32_t idx_cap)
{
    return (int64_t)n_masks + (int64_t)idx_cap / CM3D_MEDOID_TILE + 1;
}

extern "C" int64_t cm3d_medoid_workspace_bytes(int32_t n_masks, int32_t idx_cap)
{
    if (n_masks <= 0 || idx_cap <= 0) return 0;
    return md_tile_cap(n_masks, idx_cap) * (int64_t)(sizeof(TileBest) + sizeof(TileDesc));
}

extern "C" int cm3d_medoid(const float *points, const int32_t *pt_off, const int32_t *mask_frame, int32_t n_masks,
                           const int32_t *hit_off, const int32_t *tile_off, const int32_t *hit_row, int32_t idx_cap,
                           int32_t *medoid_pos, float *centroid, float *colsum_opt, void *workspace,
                           int64_t workspace_bytes, cm3d_stream_t stream)
{
    if (!points || !pt_off || !mask_frame || !hit_off || !tile_off || !hit_row || !medoid_pos || !centroid || !workspace)
        return CM3D_ERR_ARG;
    if (n_masks <= 0 || idx_cap <= 0) return CM3D_ERR_ARG;
    if (workspace_bytes < cm3d_medoid_workspace_bytes(n_masks, idx_cap)) return CM3D_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t tile_cap64 = md_tile_cap(n_masks, idx_cap);
    const int tile_cap = (int)(tile_cap64 > 0x7FFFFFFF ? 0x7FFFFFFF : tile_cap64);
    TileDesc *desc = (TileDesc *)workspace;
    TileBest *best = (TileBest *)(desc + tile_cap);
    hipLaunchKernelGGL(k_medoid_desc, dim3(1), dim3(1024), 0, st, n_masks, hit_off, tile_off, idx_cap, tile_cap, desc);
    CM3D_CHECK_LAUNCH();
    int grid = (tile_cap + MD_WAVES - 1) / MD_WAVES;
    int gmax = 4096;
    if (const char *e = getenv("CM3D_MD_GRID")) gmax = atoi(e);
    if (grid > gmax) grid = gmax;
    hipLaunchKernelGGL(k_medoid_tiles, dim3(grid), dim3(MD_THREADS), 0, st, (const float4 *)points, pt_off, mask_frame, n_masks,
                       tile_off, hit_row, desc, best, tile_cap, colsum_opt);
    CM3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_medoid_reduce, dim3((n_masks + 255) / 256), dim3(256), 0, st, (const float4 *)points, pt_off, mask_frame,
                       n_masks, hit_off, tile_off, hit_row, idx_cap, best, tile_cap, medoid_pos, centroid);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}
