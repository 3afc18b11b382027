// a2: sweep preparation -- reference src/nuscenes/2d_to_3d.py:437-465,
// utils/pcd.py:159-172,246-257.  HBM-bound streaming + ordered compaction:
// count per 1024-row block -> exclusive scan -> recompute + ordered scatter.
// Algorithmic bytes: 2 x 4*raw_stride (two reads of each raw row) + 16 per kept point.
#include "common.h"

#define SW_BLOCK 1024

static __device__ __forceinline__ bool sweep_keep(float x, float y, float halfw)
{
    // reference drops points with |x|<sqrt(2.3) AND |y|<sqrt(2.3) (2d_to_3d.py:442-445)
    return !(fabsf(x) < halfw && fabsf(y) < halfw);
}

__global__ __launch_bounds__(SW_BLOCK) void k_sweep_count(const float *__restrict__ raw, int raw_stride,
                                                           const int32_t *__restrict__ sweep_row_off, int nblk,
                                                           float halfw, int32_t *__restrict__ blk_cnt)
{
    const int s = blockIdx.y, b = blockIdx.x;
    const int r0 = sweep_row_off[s], n = sweep_row_off[s + 1] - r0;
    const int i = b * SW_BLOCK + threadIdx.x;
    bool keep = false;
    if (i < n) {
        const float *p = raw + (size_t)(r0 + i) * raw_stride;
        keep = sweep_keep(p[0], p[1], halfw);
    }
    __shared__ int s_w[16];
    uint64_t bal = __ballot(keep);
    if (cm3d_lane() == 0) s_w[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += s_w[w];
        blk_cnt[s * nblk + b] = t;
    }
}

// single-block exclusive scan of n ints: out[0..n-1], out[n] = total
__global__ __launch_bounds__(1024) void k_scan_i32(const int32_t *__restrict__ in, int n, int32_t *__restrict__ out)
{
    __shared__ int s_part[16];
    int carry = 0;
    for (int base = 0; base < n; base += 1024) {
        int i = base + threadIdx.x;
        int v = i < n ? in[i] : 0;
        int tot;
        int ex = cm3d_block1024_excl_scan(v, s_part, tot);
        if (i < n) out[i] = carry + ex;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry;
}

__global__ __launch_bounds__(SW_BLOCK) void k_sweep_scatter(const float *__restrict__ raw, int raw_stride,
                                                             const int32_t *__restrict__ sweep_row_off, int nblk,
                                                             const float *__restrict__ sweep_xf, float halfw,
                                                             const int32_t *__restrict__ blk_off,
                                                             float4 *__restrict__ points, int pt_cap)
{
    const int s = blockIdx.y, b = blockIdx.x;
    const int r0 = sweep_row_off[s], n = sweep_row_off[s + 1] - r0;
    if (b * SW_BLOCK >= n) return;
    __shared__ float s_xf[CM3D_SWEEP_XF_STRIDE];
    __shared__ int s_w[16];
    if (threadIdx.x < CM3D_SWEEP_XF_STRIDE) s_xf[threadIdx.x] = sweep_xf[s * CM3D_SWEEP_XF_STRIDE + threadIdx.x];
    const int i = b * SW_BLOCK + threadIdx.x;
    bool keep = false;
    float x = 0.f, y = 0.f, z = 0.f, w = 0.f;
    if (i < n) {
        const float *p = raw + (size_t)(r0 + i) * raw_stride;
        x = p[0]; y = p[1]; z = p[2]; w = p[3];
        keep = sweep_keep(x, y, halfw);
    }
    uint64_t bal = __ballot(keep);
    const int wave = threadIdx.x >> 6;
    if (cm3d_lane() == 0) s_w[wave] = __popcll(bal);
    __syncthreads();
    int wbase = 0;
    for (int k = 0; k < wave; ++k) wbase += s_w[k];
    if (keep) {
        int pos = blk_off[s * nblk + b] + wbase + cm3d_mbcnt(bal);
        // sensor -> ego (rotate then translate), ego -> global (2d_to_3d.py:450-457)
        float ax, ay, az;
        cm3d_rot3(s_xf, x, y, z, ax, ay, az);
        ax = ax + s_xf[9]; ay = ay + s_xf[10]; az = az + s_xf[11];
        float bx, by, bz;
        cm3d_rot3(s_xf + 12, ax, ay, az, bx, by, bz);
        bx = bx + s_xf[21]; by = by + s_xf[22]; bz = bz + s_xf[23];
        if (pos < pt_cap) points[pos] = make_float4(bx, by, bz, w);
    }
}

__global__ void k_sweep_pt_off(const int32_t *__restrict__ frame_sweep_off, int n_frames, int n_sweeps, int nblk,
                               const int32_t *__restrict__ blk_off, int pt_cap, int32_t *__restrict__ pt_off,
                               int32_t *__restrict__ status)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f <= n_frames) {
        int s = f < n_frames ? frame_sweep_off[f] : n_sweeps;
        pt_off[f] = blk_off[s * nblk];
    }
    if (f == 0) {
        int total = blk_off[n_sweeps * nblk];
        status[1] = total;
        if (total > pt_cap) atomicOr(&status[0], 1);
    }
}

extern "C" int64_t cm3d_sweep_prep_workspace_bytes(int32_t n_sweeps, int32_t max_rows_per_sweep)
{
    if (n_sweeps <= 0 || max_rows_per_sweep <= 0) return 0;
    int64_t nblk = (max_rows_per_sweep + SW_BLOCK - 1) / SW_BLOCK;
    return (2 * (int64_t)n_sweeps * nblk + 2) * (int64_t)sizeof(int32_t);
}

extern "C" int cm3d_sweep_prep(const float *raw, int32_t raw_stride, const int32_t *sweep_row_off, int32_t n_sweeps,
                               int32_t max_rows_per_sweep, const float *sweep_xf, const int32_t *frame_sweep_off,
                               int32_t n_frames, float halfw, float *points, int32_t pt_cap, int32_t *pt_off,
                               int32_t *status, void *workspace, int64_t workspace_bytes, cm3d_stream_t stream)
{
    if (!raw || !sweep_row_off || !sweep_xf || !frame_sweep_off || !points || !pt_off || !status || !workspace)
        return CM3D_ERR_ARG;
    if (raw_stride < 4 || n_sweeps <= 0 || n_frames <= 0 || max_rows_per_sweep <= 0 || pt_cap <= 0) return CM3D_ERR_ARG;
    if (workspace_bytes < cm3d_sweep_prep_workspace_bytes(n_sweeps, max_rows_per_sweep)) return CM3D_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (max_rows_per_sweep + SW_BLOCK - 1) / SW_BLOCK;
    int32_t *blk_cnt = (int32_t *)workspace;
    int32_t *blk_off = blk_cnt + (size_t)n_sweeps * nblk;
    dim3 grid(nblk, n_sweeps);
    hipLaunchKernelGGL(k_sweep_count, grid, dim3(SW_BLOCK), 0, st, raw, raw_stride, sweep_row_off, nblk, halfw, blk_cnt);
    CM3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_scan_i32, dim3(1), dim3(1024), 0, st, blk_cnt, n_sweeps * nblk, blk_off);
    CM3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_sweep_scatter, grid, dim3(SW_BLOCK), 0, st, raw, raw_stride, sweep_row_off, nblk, sweep_xf, halfw,
                       blk_off, (float4 *)points, pt_cap);
    CM3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_sweep_pt_off, dim3((n_frames + 256) / 256), dim3(256), 0, st, frame_sweep_off, n_frames, n_sweeps,
                       nblk, blk_off, pt_cap, pt_off, status);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}
