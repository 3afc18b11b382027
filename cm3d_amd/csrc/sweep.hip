// a2: sweep preparation -- reference src/nuscenes/2d_to_3d.py:437-465,
// utils/pcd.py:159-172,246-257.
// ONE streaming pass (HBM-bound): every raw row is transformed sensor -> ego -> global and written at
// its own row index.  Rows inside the ego box (|x| < halfw && |y| < halfw, :442-445) are not compacted
// away here -- that would need a counting pass over the whole batch first.  They are written as NaN
// points (inert in every later test) and marked in the frame's removed-row bits (one bit per row, zeroed by
// cm3d_batch_begin; sweeps of a frame may share a word, hence atomicOr -- 0.2 % of the rows); k_compact_hits subtracts
// "removed rows before me" when it emits a point index, so the index lists are exactly the indices into the
// reference's compacted cloud.
// Algorithmic bytes: 4*raw_stride per raw row read + 16 per row written.
#include "common.h"

#define SW_THREADS 256
#define SW_ROWS 2048              // rows per workgroup: 8 per thread, all 32 dword loads of a thread in flight at once
                                  // (measured: 62 us for C2 against 77 us with 4 rows per thread and 83 us with an
                                  // LDS-staged slab copy; raw rows are read once -> non-temporal loads keep the
                                  // Infinity Cache for the transformed cloud that k_project_hits reads next)

__global__ __launch_bounds__(SW_THREADS) void k_sweep_xform(const float *__restrict__ raw, int raw_stride, const float *__restrict__ intensity,
                                                             const int32_t *__restrict__ sweep_row_off,
                                                             const float *__restrict__ sweep_xf,
                                                             const int32_t *__restrict__ frame_sweep_off, int n_frames,
                                                             int n_sweeps, float halfw, float4 *__restrict__ points, int pt_cap,
                                                             int32_t *__restrict__ pt_off, uint32_t *__restrict__ removed_bits,
                                                             int32_t *__restrict__ status)
{
    __shared__ float s_xf[CM3D_SWEEP_XF_STRIDE];
    __shared__ int s_f, s_frow0;
    const int s = blockIdx.y, b = blockIdx.x, t = threadIdx.x;
    if (t == 0) s_f = -1;
    __syncthreads();
    const int r0 = sweep_row_off[s], n = sweep_row_off[s + 1] - r0;
    const int total_rows = sweep_row_off[n_sweeps];
    if (b != 0 && b * SW_ROWS >= n) return;
    const int row_base = r0 + b * SW_ROWS;
    const int cnt = max(0, min(min(SW_ROWS, n - b * SW_ROWS), pt_cap - row_base));
    if (t < CM3D_SWEEP_XF_STRIDE) s_xf[t] = sweep_xf[s * CM3D_SWEEP_XF_STRIDE + t];
    // frame of this sweep: the f with frame_sweep_off[f] <= s < frame_sweep_off[f+1] (one round of loads)
    for (int f = t; f < n_frames; f += SW_THREADS) {
        const int a = frame_sweep_off[f], e = frame_sweep_off[f + 1];
        if (a <= s && s < e) {
            s_f = f;
            s_frow0 = sweep_row_off[a];
            if (b == 0 && a == s) pt_off[f] = r0;
        }
        if (b == 0 && s == 0 && a == e) pt_off[f] = sweep_row_off[a];      // frame without sweeps
    }
    if (b == 0 && s == n_sweeps - 1 && t == 0) {
        pt_off[n_frames] = total_rows;
        status[1] = total_rows;
        if (total_rows > pt_cap) atomicOr(&status[0], 1);
    }
    __syncthreads();
    const int f = s_f, frame_row0 = s_frow0;
    if (f < 0) {                                   // sweep outside every frame: malformed offsets
        if (t == 0) atomicOr(&status[0], 4);
        return;
    }
    const float *src = raw + (size_t)row_base * raw_stride;
    const float qnan = __int_as_float(0x7FC00000);
    uint32_t *bits = removed_bits + ((size_t)(frame_row0 >> 5) + 8 * (size_t)f);      // the frame's bits (see cm3d_hip.h)
#pragma unroll
    for (int k = 0; k < SW_ROWS / SW_THREADS; ++k) {
        const int i = k * SW_THREADS + t;
        const bool live = i < cnt;
        float x = 1e30f, y = 1e30f, z = 0.f, w = 0.f;
        const int g = row_base + i;
        if (live && raw_stride == CM3D_RAW_QUADS) {      // quad layout (cm3d_hip.h): x, y, z of row g at 12 (g >> 2) + (g & 3) + 0 / 4 / 8
            const float *p = raw + (size_t)(g >> 2) * 12 + (g & 3);
            x = __builtin_nontemporal_load(p); y = __builtin_nontemporal_load(p + 4); z = __builtin_nontemporal_load(p + 8);
            w = intensity ? __builtin_nontemporal_load(intensity + g) : 0.f;
        } else if (live) {
            const float *p = src + (size_t)i * raw_stride;
            x = __builtin_nontemporal_load(p); y = __builtin_nontemporal_load(p + 1);
            z = __builtin_nontemporal_load(p + 2); w = __builtin_nontemporal_load(p + 3);
        }
        const bool drop = live && fabsf(x) < halfw && fabsf(y) < halfw;      // reference drops this row (2d_to_3d.py:442-445)
        if (drop) {
            const int r = g - frame_row0;
            atomicOr(&bits[r >> 5], 1u << (r & 31));
            points[g] = make_float4(qnan, qnan, qnan, w);
        }
        if (live && !drop) {
            // sensor -> ego (rotate then translate), ego -> global (2d_to_3d.py:450-457)
            float ax, ay, az;
            cm3d_rot3(s_xf, x, y, z, ax, ay, az);
            ax = ax + s_xf[9]; ay = ay + s_xf[10]; az = az + s_xf[11];
            float bx, by, bz;
            cm3d_rot3(s_xf + 12, ax, ay, az, bx, by, bz);
            bx = bx + s_xf[21]; by = by + s_xf[22]; bz = bz + s_xf[23];
            points[g] = make_float4(bx, by, bz, w);
        }
    }
}

extern "C" int cm3d_sweep_prep(const float *raw, int32_t raw_stride, const float *intensity, const int32_t *sweep_row_off, int32_t n_sweeps,
                               int32_t max_rows_per_sweep, const float *sweep_xf, const int32_t *frame_sweep_off,
                               int32_t n_frames, float halfw, float *points, int32_t pt_cap, int32_t *pt_off,
                               uint32_t *removed_bits, int32_t *status, cm3d_stream_t stream)
{
    if (!raw || !sweep_row_off || !sweep_xf || !frame_sweep_off || !points || !pt_off || !removed_bits || !status)
        return CM3D_ERR_ARG;
    if ((raw_stride < 4 && raw_stride != CM3D_RAW_QUADS) || n_sweeps <= 0 || n_frames <= 0 || max_rows_per_sweep <= 0 || pt_cap <= 0) return CM3D_ERR_ARG;
    if (intensity && raw_stride != CM3D_RAW_QUADS) return CM3D_ERR_ARG;
    if ((uintptr_t)points & 15) return CM3D_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sweep_xform, dim3((max_rows_per_sweep + SW_ROWS - 1) / SW_ROWS, n_sweeps), dim3(SW_THREADS), 0, st, raw,
                       raw_stride, intensity, sweep_row_off, sweep_xf, frame_sweep_off, n_frames, n_sweeps, halfw, (float4 *)points, pt_cap,
                       pt_off, removed_bits, status);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}
