// the box matching of the SAM3D fusion step.
// Reference: src/nuscenes/linear_matching.py:53-121,231-259 (and the same block of
// src/waymo/linear_matching.py): per sample, `match(pred_boxes, sam3d_boxes, 0.2, Type.TYPE_2D)` =
// waymo_open_dataset's py_metrics_ops.match with TYPE_HUNGARIAN -- bird's-eye-view IoU of rotated
// rectangles, quantised to integers, maximum-weight bipartite assignment, pairs below the IoU
// threshold dropped.  (The op's C++ is not in the reference checkout: parity unpinned)
//
// Two launches for all samples of a result file:
//   k_bev_weights   one thread per (prediction, sam3d) pair of a sample: float64 polygon clipping ->
//                   int32 weight matrix (the matrices sit in L2 for the solver)
//   k_bev_assign    one wave per sample: Hungarian method with potentials, all-integer; the column loop
//                   runs across the 64 lanes, the minimum is one wave reduction per step
// Latency / integer bound; no HBM roofline applies (a sample's matrix is a few KB).
#include "common.h"

#define BM_KMAX 1000000        // IoU quantisation: weight = (int)(iou * BM_KMAX)
#define BM_MAX_SIDE CM3D_MAX_MATCH_BOXES
#define BM_INF (1ll << 48)
#define BM_SMALL 128              // samples up to this size on both sides: k_bev_assign_small<1>, <2>

// Box record: cx, cy, length, width, cos(heading), sin(heading) (float64).
// Intersection by clipping A against the four edges of B (both counter-clockwise), in coordinates
// relative to A's centre so that global-frame magnitudes cancel before the products.
static __device__ void bev_corners(const double *__restrict__ b, double ox, double oy, double *X, double *Y)
{
    const double hl = b[2] * 0.5, hw = b[3] * 0.5, c = b[4], s = b[5];
    const double dx = b[0] - ox, dy = b[1] - oy;
    const double lc = hl * c, ls = hl * s, wc = hw * c, wsn = hw * s;
    X[0] = (dx + lc) - wsn; Y[0] = (dy + ls) + wc;
    X[1] = (dx - lc) - wsn; Y[1] = (dy - ls) + wc;
    X[2] = (dx - lc) + wsn; Y[2] = (dy - ls) - wc;
    X[3] = (dx + lc) + wsn; Y[3] = (dy + ls) - wc;
}

static __device__ double bev_iou(const double *__restrict__ a, const double *__restrict__ b)
{
    const double area_a = a[2] * a[3], area_b = b[2] * b[3];
    if (!(area_a > 0.0) || !(area_b > 0.0)) return 0.0;       // zeros(D) = "no box" (linear_matching.py:65)
    {   // circumscribed circles apart: the intersection is empty
        const double dx = b[0] - a[0], dy = b[1] - a[1];
        const double ra2 = a[2] * a[2] + a[3] * a[3], rb2 = b[2] * b[2] + b[3] * b[3];
        const double r = 0.5 * (sqrt(ra2) + sqrt(rb2));
        if (dx * dx + dy * dy > r * r) return 0.0;
    }
    double px[12], py[12], qx[12], qy[12], bx[4], by[4];
    bev_corners(a, a[0], a[1], px, py);
    bev_corners(b, a[0], a[1], bx, by);
    int n = 4;
    for (int e = 0; e < 4 && n > 0; ++e) {
        const double x1 = bx[e], y1 = by[e], ex = bx[(e + 1) & 3] - x1, ey = by[(e + 1) & 3] - y1;
        int k = 0;
        double prx = px[n - 1], pry = py[n - 1];
        double dp = ex * (pry - y1) - ey * (prx - x1);
        for (int i = 0; i < n; ++i) {
            const double cx = px[i], cy = py[i];
            const double dc = ex * (cy - y1) - ey * (cx - x1);
            if ((dc >= 0.0) != (dp >= 0.0)) {
                const double t = dp / (dp - dc);
                qx[k] = prx + t * (cx - prx);
                qy[k] = pry + t * (cy - pry);
                ++k;
            }
            if (dc >= 0.0) { qx[k] = cx; qy[k] = cy; ++k; }
            prx = cx; pry = cy; dp = dc;
        }
        n = k;
        for (int i = 0; i < n; ++i) { px[i] = qx[i]; py[i] = qy[i]; }
    }
    if (n < 3) return 0.0;
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        acc += px[i] * py[j] - px[j] * py[i];
    }
    const double inter = 0.5 * fabs(acc);
    const double uni = (area_a + area_b) - inter;
    if (!(uni > 0.0)) return 0.0;
    const double iou = inter / uni;
    return iou > 1.0 ? 1.0 : iou;
}

static __device__ __forceinline__ int bev_weight(double iou, double thr)
{
    return iou >= thr ? (int)(iou * (double)BM_KMAX) : 0;
}

__global__ __launch_bounds__(256) void k_bev_init(int32_t *__restrict__ pred_match, double *__restrict__ match_iou, int n_pred,
                                                  int32_t *__restrict__ gt_match, int n_gt)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n_pred) { pred_match[t] = -1; match_iou[t] = 0.0; }
    if (t < n_gt) gt_match[t] = -1;
}

// first sample of every 256-pair block of k_bev_weights (one binary search per block instead of one per pair)
__global__ __launch_bounds__(256) void k_bev_block_frames(const int64_t *__restrict__ pair_off, int n_frames, int64_t n_blocks,
                                                          int32_t *__restrict__ blk_frame)
{
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= n_blocks) return;
    const int64_t t = b * 256;
    int lo = 0, hi = n_frames;              // last f with pair_off[f] <= t
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (pair_off[mid] <= t) lo = mid; else hi = mid;
    }
    blk_frame[b] = lo;
}

__global__ __launch_bounds__(256) void k_bev_weights(const double *__restrict__ pred, const int32_t *__restrict__ pred_off,
                                                     const double *__restrict__ gt, const int32_t *__restrict__ gt_off,
                                                     const int64_t *__restrict__ pair_off, const int32_t *__restrict__ blk_frame,
                                                     int n_frames, int64_t total_pairs, double thr, int32_t *__restrict__ weight)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total_pairs) return;
    int f = blk_frame[blockIdx.x];
    while (f + 1 < n_frames && pair_off[f + 1] <= t) ++f;      // a block spans few samples
    const int G = gt_off[f + 1] - gt_off[f];
    const int64_t r = t - pair_off[f];
    const int p = (int)(r / G), g = (int)(r - (int64_t)p * G);
    const double iou = bev_iou(pred + (int64_t)(pred_off[f] + p) * 6, gt + (int64_t)(gt_off[f] + g) * 6);
    weight[t] = bev_weight(iou, thr);
}

static __device__ __forceinline__ long long wave_min_i64(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}

// Hungarian method (potentials u on rows, v on columns; rows <= columns, every row gets a column).
// Cost = BM_KMAX - weight >= 0, so a maximum-weight assignment is found; ties: an unassigned column wins a step
// (the search ends there -- most of a sample's costs are the same "no overlap" value), then the lowest column index (the CPU oracle runs the same steps sequentially, the total weight is checked against an
// independent solver in tests/).
__global__ __launch_bounds__(64) void k_bev_assign(const double *__restrict__ pred, const int32_t *__restrict__ pred_off,
                                                   const double *__restrict__ gt, const int32_t *__restrict__ gt_off,
                                                   const int64_t *__restrict__ pair_off, const int32_t *__restrict__ weight,
                                                   int32_t *__restrict__ pred_match, int32_t *__restrict__ gt_match,
                                                   double *__restrict__ match_iou, int32_t *__restrict__ status)
{
    __shared__ long long s_u[BM_MAX_SIDE + 1], s_v[BM_MAX_SIDE + 1], s_minv[BM_MAX_SIDE + 1];
    __shared__ int s_p[BM_MAX_SIDE + 1], s_way[BM_MAX_SIDE + 1];
    __shared__ unsigned char s_used[BM_MAX_SIDE + 1];
    const int f = blockIdx.x, lane = threadIdx.x;
    const int p0 = pred_off[f], g0 = gt_off[f];
    const int P = pred_off[f + 1] - p0, G = gt_off[f + 1] - g0;
    if (P <= 0 || G <= 0) return;
    if (P <= BM_SMALL && G <= BM_SMALL) return;                          // k_bev_assign_small's
    if (P > BM_MAX_SIDE || G > BM_MAX_SIDE) {
        if (lane == 0) atomicOr(status, 1);
        return;
    }
    const int32_t *__restrict__ Wm = weight + pair_off[f];
    const bool tr = P > G;                       // rows = the smaller side
    const int n = tr ? G : P, m = tr ? P : G;
#define BM_W(i, j) (tr ? Wm[(int64_t)((j) - 1) * G + ((i) - 1)] : Wm[(int64_t)((i) - 1) * G + ((j) - 1)])
    for (int j = lane; j <= m; j += 64) { s_u[j] = 0; s_v[j] = 0; s_p[j] = 0; s_way[j] = 0; }
    __syncthreads();
    for (int i = 1; i <= n; ++i) {
        for (int j = lane; j <= m; j += 64) { s_minv[j] = BM_INF; s_used[j] = 0; }
        if (lane == 0) s_p[0] = i;
        __syncthreads();
        int j0 = 0;
        while (true) {
            if (lane == 0) s_used[j0] = 1;
            __syncthreads();
            const int i0 = s_p[j0];
            const long long ui0 = s_u[i0];
            long long key = (BM_INF << 13);
            for (int j = 1 + lane; j <= m; j += 64) {
                if (s_used[j]) continue;
                const long long cur = (long long)(BM_KMAX - BM_W(i0, j)) - ui0 - s_v[j];
                long long mv = s_minv[j];
                if (cur < mv) { mv = cur; s_minv[j] = cur; s_way[j] = j0; }
                const long long k = mv * 8192 + (s_p[j] != 0 ? 4096 : 0) + j;      // ties: an unassigned column first, then the lowest
                key = k < key ? k : key;
            }
            key = wave_min_i64(key);
            const long long delta = key >> 13;
            const int j1 = (int)(key & 4095);
            __syncthreads();
            for (int j = lane; j <= m; j += 64) {
                if (s_used[j]) { s_u[s_p[j]] += delta; s_v[j] -= delta; }
                else s_minv[j] -= delta;
            }
            __syncthreads();
            j0 = j1;
            if (s_p[j0] == 0) break;
        }
        if (lane == 0) {
            do {
                const int j1 = s_way[j0];
                s_p[j0] = s_p[j1];
                j0 = j1;
            } while (j0);
        }
        __syncthreads();
    }
    for (int j = 1 + lane; j <= m; j += 64) {
        const int i = s_p[j];
        if (i == 0) continue;
        if (BM_W(i, j) <= 0) continue;           // below the IoU threshold: not a match
        const int pi = tr ? j - 1 : i - 1, gi = tr ? i - 1 : j - 1;
        pred_match[p0 + pi] = gi;
        gt_match[g0 + gi] = pi;
        match_iou[p0 + pi] = bev_iou(pred + (int64_t)(p0 + pi) * 6, gt + (int64_t)(g0 + gi) * 6);
    }
#undef BM_W
}

// The same method for samples whose larger side has at most 64 * CPL boxes (CPL = 1, 2: nearly all samples): column j
// lives in lane (j - 1) % 64, slot (j - 1) / 64 (potential, reduced cost, predecessor, assigned row in registers), row r's
// potential likewise, the sample's weight matrix in LDS when it fits (else read from L2) -- no barrier inside the search.
// Same steps, same ties, hence the same assignment as k_bev_assign and the oracle; int32 state is enough
// here (|values| < 2^28).
#define BM_SMALL_LDS 8192
template <int CPL>
static __device__ __forceinline__ int bm_get(const int (&a)[CPL], int idx)        // a[] of element idx (uniform idx)
{
    int r = __builtin_amdgcn_readlane(a[0], idx & 63);
#pragma unroll
    for (int k = 1; k < CPL; ++k) {
        const int t = __builtin_amdgcn_readlane(a[k], idx & 63);
        r = (idx >> 6) == k ? t : r;
    }
    return r;
}

template <int CPL>
__global__ __launch_bounds__(64) void k_bev_assign_small(const double *__restrict__ pred, const int32_t *__restrict__ pred_off,
                                                         const double *__restrict__ gt, const int32_t *__restrict__ gt_off,
                                                         const int64_t *__restrict__ pair_off, const int32_t *__restrict__ weight,
                                                         int32_t *__restrict__ pred_match, int32_t *__restrict__ gt_match,
                                                         double *__restrict__ match_iou)
{
    __shared__ int s_w[BM_SMALL_LDS];
    const int f = blockIdx.x, lane = threadIdx.x;
    const int p0 = pred_off[f], g0 = gt_off[f];
    const int P = pred_off[f + 1] - p0, G = gt_off[f + 1] - g0;
    const int big = P > G ? P : G;
    if (P <= 0 || G <= 0 || big > 64 * CPL || (CPL > 1 && big <= 64 * (CPL - 1))) return;       // another kernel's sample
    const int32_t *__restrict__ Wm = weight + pair_off[f];
    const bool tr = P > G;
    const int n = tr ? G : P, m = tr ? P : G;
    const bool in_lds = n * m <= BM_SMALL_LDS;
    if (in_lds) {                                // LDS image: row-major [n][m] in (row, column) order of the search
        for (int q = lane; q < n * m; q += 64) {
            const int i = q / m, j = q - i * m;
            s_w[q] = tr ? Wm[(int64_t)j * G + i] : Wm[(int64_t)i * G + j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    auto wgt = [&](int i, int j) -> int {        // weight of row i, column j (1-based)
        if (in_lds) return s_w[(i - 1) * m + (j - 1)];
        return tr ? Wm[(int64_t)(j - 1) * G + (i - 1)] : Wm[(int64_t)(i - 1) * G + (j - 1)];
    };
    const int INF = 1 << 30;
    int u[CPL], v[CPL], p[CPL], way[CPL];        // slot k: row / column k * 64 + lane + 1
#pragma unroll
    for (int k = 0; k < CPL; ++k) { u[k] = 0; v[k] = 0; p[k] = 0; way[k] = 0; }
    for (int i = 1; i <= n; ++i) {
        int minv[CPL];
        bool used[CPL], row_in_tree[CPL];
#pragma unroll
        for (int k = 0; k < CPL; ++k) { minv[k] = INF; used[k] = false; row_in_tree[k] = false; }
        int j0 = 0;
        while (true) {
            // column j0 joins the tree, with it the row assigned to it
            const int i0 = j0 == 0 ? i : bm_get<CPL>(p, j0 - 1);
            const int ui0 = bm_get<CPL>(u, i0 - 1);
            long long key = ((long long)INF << 9);
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int col = k * 64 + lane + 1;
                if (col == j0) used[k] = true;
                if (col == i0) row_in_tree[k] = true;
                if (col <= m && !used[k]) {
                    const int cur = (BM_KMAX - wgt(i0, col)) - ui0 - v[k];
                    if (cur < minv[k]) { minv[k] = cur; way[k] = j0; }
                    const long long kk = (long long)minv[k] * 512 + (p[k] != 0 ? 256 : 0) + col;   // unassigned column first
                    key = kk < key ? kk : key;
                }
            }
            key = wave_min_i64(key);
            const int delta = (int)(key >> 9);
            const int j1 = __builtin_amdgcn_readfirstlane((int)(key & 255));
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                if (row_in_tree[k]) u[k] += delta;
                if (used[k]) v[k] -= delta; else minv[k] -= delta;
            }
            j0 = j1;
            if (bm_get<CPL>(p, j0 - 1) == 0) break;
        }
        // augment along the predecessor chain (uniform pointer chasing through lane reads)
        do {
            const int j1 = bm_get<CPL>(way, j0 - 1);
            const int pr = j1 == 0 ? i : bm_get<CPL>(p, (j1 == 0 ? 1 : j1) - 1);
#pragma unroll
            for (int k = 0; k < CPL; ++k)
                if (k * 64 + lane + 1 == j0) p[k] = pr;
            j0 = j1;
        } while (j0);
    }
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int col = k * 64 + lane + 1;
        if (col <= m && p[k] != 0) {
            const int i = p[k];
            if (wgt(i, col) > 0) {
                const int pi = tr ? col - 1 : i - 1, gi = tr ? i - 1 : col - 1;
                pred_match[p0 + pi] = gi;
                gt_match[g0 + gi] = pi;
                match_iou[p0 + pi] = bev_iou(pred + (int64_t)(p0 + pi) * 6, gt + (int64_t)(g0 + gi) * 6);
            }
        }
    }
}

extern "C" int64_t cm3d_bev_match_workspace_bytes(int64_t total_pairs)
{
    const int64_t n = total_pairs > 0 ? total_pairs : 1;
    return (n + (n + 255) / 256) * (int64_t)sizeof(int32_t);           // weights + the first sample of every 256-pair block
}

extern "C" int cm3d_bev_match(const double *pred, const int32_t *pred_off, int32_t n_pred, const double *gt,
                              const int32_t *gt_off, int32_t n_gt, const int64_t *pair_off, int32_t n_frames,
                              int64_t total_pairs, double iou_thr, int32_t *pred_match, int32_t *gt_match, double *match_iou,
                              int32_t *status, void *workspace, int64_t workspace_bytes, cm3d_stream_t stream)
{
    if (!pred_off || !gt_off || !pair_off || !status || n_frames <= 0 || n_pred < 0 || n_gt < 0 || total_pairs < 0)
        return CM3D_ERR_ARG;
    if ((n_pred > 0 && (!pred || !pred_match || !match_iou)) || (n_gt > 0 && (!gt || !gt_match))) return CM3D_ERR_ARG;
    if (total_pairs >= ((int64_t)1 << 31) * 256) return CM3D_ERR_ARG;
    if (total_pairs > 0 && (!workspace || workspace_bytes < cm3d_bev_match_workspace_bytes(total_pairs))) return CM3D_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int n_init = n_pred > n_gt ? n_pred : n_gt;
    if (n_init > 0) {
        hipLaunchKernelGGL(k_bev_init, dim3((n_init + 255) / 256), dim3(256), 0, st, pred_match, match_iou, n_pred, gt_match, n_gt);
        CM3D_CHECK_LAUNCH();
    }
    if (total_pairs == 0) return CM3D_OK;
    int32_t *weight = (int32_t *)workspace;
    int32_t *blk_frame = weight + total_pairs;
    const int64_t n_blocks = (total_pairs + 255) / 256;
    hipLaunchKernelGGL(k_bev_block_frames, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, st, pair_off, n_frames, n_blocks,
                       blk_frame);
    CM3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_bev_weights, dim3((unsigned)n_blocks), dim3(256), 0, st, pred, pred_off, gt, gt_off, pair_off, blk_frame,
                       n_frames, total_pairs, iou_thr, weight);
    CM3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_bev_assign_small<1>, dim3(n_frames), dim3(64), 0, st, pred, pred_off, gt, gt_off, pair_off, weight, pred_match,
                       gt_match, match_iou);
    CM3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_bev_assign_small<2>, dim3(n_frames), dim3(64), 0, st, pred, pred_off, gt, gt_off, pair_off, weight, pred_match,
                       gt_match, match_iou);
    CM3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_bev_assign, dim3(n_frames), dim3(64), 0, st, pred, pred_off, gt, gt_off, pair_off, weight, pred_match,
                       gt_match, match_iou, status);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}
