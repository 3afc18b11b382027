// a10-a15: nearest lane point, box assembly, class-aware circle NMS.
//   reference: lane_yaws_distances_and_coords src/nuscenes/2d_to_3d.py:277-302,
//   stage 2 :745-817 with push_centroid :164-198, circle_nms :309-332, thresholds :850-861.
// All float64 like the reference (scipy cdist / numpy scalars).  FP64-VALU-bound (lane NN) and
// latency-bound (NMS); no HBM roofline applies.
#include "common.h"

// ---- nearest lane point (a10) ------------------------------------------------------------------
// Reference: scipy cdist (float64) between every centroid and every lane point, np.argmin (first
// minimum).  Brute force is K x L pairs; here each lane table is binned into a uniform grid
// (rebuilt on every call, it is cheap) and a centroid only looks at the cells of growing square
// rings around it.  The result is the brute-force result exactly:
//  * after ring r every unvisited point is >= r*cell away (it differs by >= r cells along one
//    axis), so the search stops once the best distance is below r*cell (with a 1e-9 margin);
//  * candidates are compared as (sqrt(d2), original index) lexicographically, so visiting order
//    does not matter and the first minimum wins on ties, like np.argmin.
#define LG_MAX_CELLS 32768          // cells per table: the whole cell index lives in LDS (128 KiB) during the build
#define LG_CELL0 4.0f               // preferred cell edge [m]
#define LG_MAX_RINGS 64             // rings (256 m at the preferred cell size) a search may cover before the wave scans its whole table
#define LG_BIG_CELL 12         // points in a cell above which the whole wave scans it together
#define LG_BUILD_THREADS 1024

struct LaneGrid { float x0, y0, h, inv_h; int gw, gh, cell_base; float margin; };
struct LanePt { float x, y; int idx, pad; };

static __device__ __forceinline__ int lg_cell_coord(float v, float v0, float inv_h, int n)
{
    // the same expression bins lane points and (clamped later) centroids
    int c = (int)floorf((v - v0) * inv_h);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

// One workgroup builds the whole index of one lane table: bounding box -> grid geometry -> cell
// counts (LDS atomics) -> exclusive scan (LDS) -> scatter into cell order.  Three passes over the
// table (it sits in L2), one launch, no global atomics, no memset.
__global__ __launch_bounds__(LG_BUILD_THREADS) void k_lane_grid_build(const float *__restrict__ lane,
                                                                      const int32_t *__restrict__ lane_off,
                                                                      LaneGrid *__restrict__ grids, int32_t *__restrict__ cell_start,
                                                                      LanePt *__restrict__ sorted)
{
    extern __shared__ __align__(16) int s_cell[];          // LG_MAX_CELLS + 1 ints
    __shared__ float s_red[4][16];
    __shared__ int s_part[16];
    __shared__ LaneGrid s_g;
    const int t = blockIdx.x;
    const int lo = lane_off[t], L = lane_off[t + 1] - lo;
    // pass 1: bounding box
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = threadIdx.x; i < L; i += LG_BUILD_THREADS) {
        const float x = lane[(size_t)(lo + i) * 3], y = lane[(size_t)(lo + i) * 3 + 1];
        if (x == x && y == y) { mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mny = fminf(mny, y); mxy = fmaxf(mxy, y); }
    }
    mnx = cm3d_wave_reduce_t(mnx, [](float a, float b) { return fminf(a, b); }); mny = cm3d_wave_reduce_t(mny, [](float a, float b) { return fminf(a, b); });
    mxx = cm3d_wave_reduce_t(mxx, [](float a, float b) { return fmaxf(a, b); }); mxy = cm3d_wave_reduce_t(mxy, [](float a, float b) { return fmaxf(a, b); });
    const int wave = threadIdx.x >> 6;
    if (cm3d_lane() == 0) { s_red[0][wave] = mnx; s_red[1][wave] = mny; s_red[2][wave] = mxx; s_red[3][wave] = mxy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            mnx = fminf(mnx, s_red[0][w]); mny = fminf(mny, s_red[1][w]);
            mxx = fmaxf(mxx, s_red[2][w]); mxy = fmaxf(mxy, s_red[3][w]);
        }
        LaneGrid g;
        if (!(mxx >= mnx)) { mnx = mny = 0.f; mxx = mxy = 0.f; }      // empty table
        float h = LG_CELL0;
        const float ex = mxx - mnx, ey = mxy - mny;
        while ((double)(floorf(ex / h) + 2.f) * (double)(floorf(ey / h) + 1.f) > (double)LG_MAX_CELLS) h *= 1.25f;
        g.x0 = mnx; g.y0 = mny; g.h = h; g.inv_h = 1.0f / h;
        g.gw = (int)floorf(ex / h) + 1; g.gh = (int)floorf(ey / h) + 1;
        g.gw |= 1;      // odd row stride (at most one empty column more): consecutive points of a lane running along y
                        // fall into cells gw apart, and a stride that is a multiple of 32 would put them all into one LDS bank
        g.cell_base = t * (LG_MAX_CELLS + 1);
        // points are binned with float32 arithmetic: a point can sit in the neighbouring cell of its
        // exact position by a few ulp of the coordinate magnitude; the ring bound gives that much away
        g.margin = 16.0f * 1.1920929e-7f * fmaxf(fmaxf(fabsf(mnx), fabsf(mxx)), fmaxf(fabsf(mny), fabsf(mxy))) + 1e-4f * h;
        s_g = g;
        grids[t] = g;
    }
    __syncthreads();
    const LaneGrid g = s_g;
    const int ncell = g.gw * g.gh;
    for (int c = threadIdx.x; c <= ncell; c += LG_BUILD_THREADS) s_cell[c] = 0;
    __syncthreads();
    // pass 2: counts
    for (int i = threadIdx.x; i < L; i += LG_BUILD_THREADS) {
        const float x = lane[(size_t)(lo + i) * 3], y = lane[(size_t)(lo + i) * 3 + 1];
        if (x == x && y == y)
            atomicAdd(&s_cell[lg_cell_coord(y, g.y0, g.inv_h, g.gh) * g.gw + lg_cell_coord(x, g.x0, g.inv_h, g.gw)], 1);
    }
    __syncthreads();
    // exclusive scan over the cells (positions are global indices into `sorted`)
    int carry = lo;
    for (int base = 0; base < ncell; base += LG_BUILD_THREADS * 4) {
        const int c0 = base + threadIdx.x * 4;
        int v[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] = c0 + q < ncell ? s_cell[c0 + q] : 0; sum += v[q]; }
        int tot;
        int ex = carry + cm3d_block1024_excl_scan(sum, s_part, tot);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (c0 + q < ncell) { s_cell[c0 + q] = ex; cell_start[g.cell_base + c0 + q] = ex; }
            ex += v[q];
        }
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) cell_start[g.cell_base + ncell] = carry;
    __syncthreads();
    // pass 3: scatter (the LDS copy of the start offsets doubles as the fill cursor)
    for (int i = threadIdx.x; i < L; i += LG_BUILD_THREADS) {
        const float x = lane[(size_t)(lo + i) * 3], y = lane[(size_t)(lo + i) * 3 + 1];
        if (x == x && y == y) {
            const int c = lg_cell_coord(y, g.y0, g.inv_h, g.gh) * g.gw + lg_cell_coord(x, g.x0, g.inv_h, g.gw);
            sorted[atomicAdd(&s_cell[c], 1)] = LanePt{x, y, i, 0};
        }
    }
}

#define LG_PF 4                                       // points of a row segment requested per round trip (k_lane_nn_grid)
// candidate (d2, j) against the running best under the reference's semantics:
// minimise sqrt(d2) (float64), ties -> smaller original index.
static __device__ __forceinline__ void lg_consider(double d2, int j, double &d2cut, double &sbest, int &jbest)
{
    if (d2 <= d2cut * (1.0 + 1e-15)) {               // otherwise sqrt(d2) > sbest for sure
        const double s = sqrt(d2);
        if (s < sbest || (s == sbest && j < jbest)) { sbest = s; jbest = j; d2cut = d2; }
    }
}

// One wave per centroid.  Ring r has 8r cells (one for r = 0); lane u takes cell u of the ring,
// scans that cell's points and the wave merges (sqrt(d2), index) lexicographically.
__global__ __launch_bounds__(256) void k_lane_nn_grid(const float *__restrict__ centroid, const int32_t *__restrict__ medoid_pos,
                                                      const int32_t *__restrict__ mask_frame, int n_masks,
                                                      const int32_t *__restrict__ lane_off, const int32_t *__restrict__ frame_lane,
                                                      const LaneGrid *__restrict__ grids, const int32_t *__restrict__ cell_start,
                                                      const LanePt *__restrict__ sorted, int max_rings,
                                                      int32_t *__restrict__ lane_idx, double *__restrict__ lane_dist)
{
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = cm3d_lane();
    if (k >= n_masks) return;
    // (everything that only needs the mask's number is requested before the first branch: behind `if (medoid_pos[k] < 0) return` the other
    // loads started a round trip later each)
    const int mp = medoid_pos[k], mf = mask_frame[k];
    const float cxf = centroid[3 * k], cyf = centroid[3 * k + 1];
    if (mp < 0) { if (lane == 0) { lane_idx[k] = -1; lane_dist[k] = INFINITY; } return; }
    const int tb = frame_lane[mf];
    const LaneGrid g = grids[tb];
    const double cx = (double)cxf, cy = (double)cyf;
    double d2cut = INFINITY, sbest = INFINITY;
    int jbest = 0x7FFFFFFF;
    bool resolved = true;
    if (lane_off[tb + 1] > lane_off[tb] && cx == cx && cy == cy) {
        resolved = false;
        // (virtual) cell of the centroid; may lie outside the grid
        const double fi = floor((cx - (double)g.x0) * (double)g.inv_h), fj = floor((cy - (double)g.y0) * (double)g.inv_h);
        const int qi = (int)fmax(-1.0e6, fmin(1.0e6, fi)), qj = (int)fmax(-1.0e6, fmin(1.0e6, fj));
        const int out_i = qi < 0 ? -qi : (qi >= g.gw ? qi - g.gw + 1 : 0);
        const int out_j = qj < 0 ? -qj : (qj >= g.gh ? qj - g.gh + 1 : 0);
        const int32_t *cs = cell_start + g.cell_base;
        // Every lane brings the point range [a, b) of one cell (empty if it has none).  Short ranges are scanned by
        // their lane; a long one (a cell where many lanes and connectors meet -- one lane would keep the whole wave
        // waiting) is handed to the whole wave, 64 points per step.  Collective: call with all lanes.
        auto visit = [&](int a, int b) {
            const bool big = b - a > LG_BIG_CELL;
            if (!big) {
                // (LG_PF points per round trip, unconditional loads of a clamped index: one load per iteration made a segment of ten points
                // ten dependent round trips -- 18 us for the launch alone, 45 with the other batches' kernels on the memory system)
                for (int q = a; q < b; q += LG_PF) {
                    LanePt p[LG_PF];
#pragma unroll
                    for (int v = 0; v < LG_PF; ++v) p[v] = sorted[min(q + v, b - 1)];
#pragma unroll
                    for (int v = 0; v < LG_PF; ++v) {
                        if (q + v < b) {
                            const double dx = cx - (double)p[v].x, dy = cy - (double)p[v].y;
                            lg_consider(dx * dx + dy * dy, p[v].idx, d2cut, sbest, jbest);
                        }
                    }
                }
            }
            for (uint64_t bm = __ballot(big); bm; bm &= bm - 1) {
                const int src = (int)__builtin_ctzll(bm);
                const int ra = __builtin_amdgcn_readlane(a, src), rb = __builtin_amdgcn_readlane(b, src);
                for (int q = ra + lane; q < rb; q += 64) {
                    const LanePt p = sorted[q];
                    const double dx = cx - (double)p.x, dy = cy - (double)p.y;
                    lg_consider(dx * dx + dy * dy, p.idx, d2cut, sbest, jbest);
                }
            }
        };
        // Rings are searched in batches that grow with the distance: [0..2] (the 5 x 5 cells around the centroid --
        // nearly every search ends there), then 3 more rings, then half as many again as are behind, ...  A batch
        // [r_lo..r_hi] is a square annulus; cut into ROW SEGMENTS it is a short list of contiguous point ranges (the
        // cells of a grid row are consecutive in the sorted copy): (r_hi - r_lo + 1) full-width rows above and below,
        // and a left and a right piece for each of the 2 r_lo - 1 rows in between.  One lane per segment, one round of
        // range loads, one stop test and one wave reduction per batch: a centroid 40 m from the nearest lane point
        // costs four batches of one step each.
        int r_done = max(out_i, out_j) - 1;                        // rings up to r_done are searched
        while (r_done < max_rings) {
            const int r_lo = r_done + 1;
            const int r_hi = r_lo == 0 ? 2 : min(max_rings, r_lo + max(2, r_lo / 2));
            const int w = r_hi - r_lo + 1;
            const int nseg = r_lo == 0 ? 5 : 2 * w + 2 * (2 * r_lo - 1);
            for (int s0 = 0; s0 < nseg; s0 += 64) {                // wave-uniform trip count: visit() is collective
                const int sg = s0 + lane;
                int row, c0, c1;
                if (r_lo == 0) { row = qj - 2 + sg; c0 = qi - 2; c1 = qi + 2; }
                else if (sg < w) { row = qj - r_hi + sg; c0 = qi - r_hi; c1 = qi + r_hi; }
                else if (sg < 2 * w) { row = qj + r_lo + (sg - w); c0 = qi - r_hi; c1 = qi + r_hi; }
                else {
                    const int m = sg - 2 * w;
                    row = qj - r_lo + 1 + (m >> 1);
                    if (m & 1) { c0 = qi + r_lo; c1 = qi + r_hi; } else { c0 = qi - r_hi; c1 = qi - r_lo; }
                }
                c0 = max(c0, 0); c1 = min(c1, g.gw - 1);
                // (unconditional loads of clamped cells, masked afterwards: inside the branch the two were waited for one after the other)
                const bool seg_ok = sg < nseg && row >= 0 && row < g.gh && c0 <= c1;
                const int rowc = min(max(row, 0), g.gh - 1), c0c = min(c0, g.gw - 1), c1c = max(c1, 0);
                int a = cs[rowc * g.gw + c0c], b = cs[rowc * g.gw + c1c + 1];
                if (!seg_ok) { a = 0; b = 0; }
                visit(a, b);
            }
            r_done = r_hi;
            // wave-wide best distance so far
            const double wb = cm3d_wave_reduce_t(sbest, [](double a, double b) { return fmin(a, b); });
            // every unvisited point is at least r_done * h - margin away
            if (wb < (double)r_done * (double)g.h - (double)g.margin) { resolved = true; break; }
            if (qi - r_done <= 0 && qi + r_done >= g.gw - 1 && qj - r_done <= 0 && qj + r_done >= g.gh - 1) { resolved = true; break; }   // whole grid seen
        }
    }
    if (!resolved) {
        // far from every lane point (the ring search gave up): exact scan of the whole table, 64 points per step.
        // The cell-sorted copy holds every non-NaN point with its original index; lg_consider keeps the
        // (distance, index) order whatever the visiting order.
        const int32_t *cs = cell_start + g.cell_base;
        const int a = cs[0], b = cs[g.gw * g.gh];
        for (int q0 = a; q0 < b; q0 += 8 * 64) {                  // 8 loads in flight per lane: a streaming scan
            LanePt p[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const int q = q0 + v * 64 + lane;
                p[v] = q < b ? sorted[q] : LanePt{0.f, 0.f, 0, 0};
            }
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                if (q0 + v * 64 + lane >= b) continue;
                const double dx = cx - (double)p[v].x, dy = cy - (double)p[v].y;
                lg_consider(dx * dx + dy * dy, p[v].idx, d2cut, sbest, jbest);
            }
        }
    }
    // lexicographic (distance, index) minimum over the lanes
    {
        const Cm3dDistIdx best = cm3d_wave_argmin(sbest, jbest);
        sbest = best.s; jbest = best.j;
    }
    if (lane == 0) {
        if (jbest == 0x7FFFFFFF) jbest = 0;      // empty table / NaN centroid: np.argmin of an all-inf/NaN row
        lane_idx[k] = jbest;
        lane_dist[k] = sbest;
    }
}

static inline size_t lg_align(size_t v) { return (v + 255) & ~(size_t)255; }

struct LgLayout { LaneGrid *grids; int32_t *cell_start; LanePt *sorted; };
static inline LgLayout lg_layout(void *grid, int n_tables, int n_lane_points)
{
    char *w = (char *)grid;
    LgLayout l;
    l.grids = (LaneGrid *)w;          w += lg_align(sizeof(LaneGrid) * (size_t)n_tables);
    l.cell_start = (int32_t *)w;      w += lg_align(sizeof(int32_t) * (size_t)n_tables * (LG_MAX_CELLS + 1));
    l.sorted = (LanePt *)w;
    (void)n_lane_points;
    return l;
}

extern "C" int64_t cm3d_lane_grid_bytes(int32_t n_tables, int32_t n_lane_points)
{
    if (n_tables <= 0 || n_lane_points <= 0) return 0;
    return (int64_t)(lg_align(sizeof(LaneGrid) * (size_t)n_tables) + lg_align(sizeof(int32_t) * (size_t)n_tables * (LG_MAX_CELLS + 1)) +
                     lg_align(sizeof(LanePt) * (size_t)n_lane_points));
}

extern "C" int cm3d_lane_grid_build(const float *lane, const int32_t *lane_off, int32_t n_tables, int32_t n_lane_points, void *grid,
                                    int64_t grid_bytes, cm3d_stream_t stream)
{
    if (!lane || !lane_off || !grid || n_tables <= 0 || n_lane_points <= 0) return CM3D_ERR_ARG;
    if (grid_bytes < cm3d_lane_grid_bytes(n_tables, n_lane_points)) return CM3D_ERR_WORKSPACE;
    const LgLayout l = lg_layout(grid, n_tables, n_lane_points);
    static bool attr_set = false;
    const size_t lds = sizeof(int) * (size_t)(LG_MAX_CELLS + 1);
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)k_lane_grid_build, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return CM3D_ERR_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_lane_grid_build, dim3(n_tables), dim3(LG_BUILD_THREADS), lds, (hipStream_t)stream, lane, lane_off, l.grids,
                       l.cell_start, l.sorted);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}

extern "C" int64_t cm3d_lane_nn_workspace_bytes(int32_t n_masks)
{
    if (n_masks <= 0) return 0;
    return 256;                 // reserved: the search needs no scratch memory in this version
}

extern "C" int cm3d_lane_nn(const float *centroid, const int32_t *medoid_pos, const int32_t *mask_frame, int32_t n_masks,
                            const float *lane, const int32_t *lane_off, const int32_t *frame_lane, int32_t n_tables,
                            int32_t n_lane_points, const void *grid, int32_t *lane_idx, double *lane_dist, void *workspace,
                            int64_t workspace_bytes, cm3d_stream_t stream)
{
    if (!centroid || !medoid_pos || !mask_frame || !lane || !lane_off || !frame_lane || !grid || !lane_idx || !lane_dist || !workspace)
        return CM3D_ERR_ARG;
    if (n_masks <= 0 || n_tables <= 0 || n_lane_points <= 0) return CM3D_ERR_ARG;
    if (workspace_bytes < cm3d_lane_nn_workspace_bytes(n_masks)) return CM3D_ERR_WORKSPACE;
    const LgLayout l = lg_layout(const_cast<void *>(grid), n_tables, n_lane_points);
    hipLaunchKernelGGL(k_lane_nn_grid, dim3((n_masks + 3) / 4), dim3(256), 0, (hipStream_t)stream, centroid, medoid_pos, mask_frame,
                       n_masks, lane_off, frame_lane, l.grids, l.cell_start, l.sorted, LG_MAX_RINGS, lane_idx, lane_dist);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}

// ---------------------------------------------------------------------------
// One wave per frame: box assembly for every mask with a centroid, then greedy circle NMS.
#define BN_MAX CM3D_MAX_MASKS_PER_FRAME

// Greedy class-aware circle NMS over the LDS arrays of one frame (one wave).
// reference circle_nms src/nuscenes/2d_to_3d.py:309-332.
static __device__ __forceinline__ void nms_phase(int nm, int lane_id, const double *__restrict__ nms_thr, const double *s_x,
                                                 const double *s_y, const double *s_s, const int *s_lab, int *s_order,
                                                 const unsigned char *s_valid, unsigned char *s_sup, unsigned char *s_keep)
{
    __syncthreads();
    // order: descending score, ties by descending index (pinned tie-break, SURVEY hard part 4)
    int nv = 0;
    for (int k = 0; k < nm; ++k) nv += s_valid[k];
    for (int k = lane_id; k < nm; k += 64) {
        if (!s_valid[k]) continue;
        int rank = 0;
        const double sk = s_s[k];
        for (int j = 0; j < nm; ++j) {
            if (!s_valid[j]) continue;
            const double sj = s_s[j];
            rank += (sj > sk || (sj == sk && j > k)) ? 1 : 0;
        }
        s_order[rank] = k;
    }
    __syncthreads();
    for (int r = 0; r < nv; ++r) {
        const int i = s_order[r];
        if (!s_sup[i]) {            // uniform: every lane reads the same LDS word
            if (lane_id == 0) s_keep[i] = 1;
            const double xi = s_x[i], yi = s_y[i];
            const int li = s_lab[i];
            for (int r2 = r + 1 + lane_id; r2 < nv; r2 += 64) {
                const int j = s_order[r2];
                if (s_sup[j]) continue;
                const double dx = xi - s_x[j], dy = yi - s_y[j];
                const double dist = dx * dx + dy * dy;
                if (dist <= nms_thr[s_lab[j]] && s_lab[j] == li) s_sup[j] = 1;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void k_box_nms(const float *__restrict__ centroid, const int32_t *__restrict__ medoid_pos,
                                                const int32_t *__restrict__ mask_off, const int32_t *__restrict__ class_id,
                                                const double *__restrict__ score, const float *__restrict__ lane,
                                                const int32_t *__restrict__ lane_off, const int32_t *__restrict__ frame_lane,
                                                const int32_t *__restrict__ lane_idx, const double *__restrict__ lane_dist,
                                                const double *__restrict__ prior_wlh, const int32_t *__restrict__ is_vehicle,
                                                const int32_t *__restrict__ nms_group, const double *__restrict__ nms_thr,
                                                int n_classes, const double *__restrict__ ego_xyz,
                                                const float *__restrict__ pose_inv, double *__restrict__ box,
                                                int32_t *__restrict__ flags)
{
    __shared__ double s_x[BN_MAX], s_y[BN_MAX], s_s[BN_MAX];
    __shared__ int s_lab[BN_MAX], s_order[BN_MAX];
    __shared__ unsigned char s_valid[BN_MAX], s_sup[BN_MAX], s_keep[BN_MAX];
    const int f = blockIdx.x;
    const int m0 = mask_off[f];
    const int nm = min(mask_off[f + 1] - m0, BN_MAX);
    const int lane_id = threadIdx.x;
    // nuScenes: boxes live in the global frame, the ego position is needed for the push-back direction.
    // Waymo (pose_inv != null): centroids arrive in the global frame, boxes leave in the vehicle frame.
    const bool waymo = pose_inv != nullptr;
    const double ex0 = waymo ? 0.0 : ego_xyz[3 * f], ey0 = waymo ? 0.0 : ego_xyz[3 * f + 1];
    double Pi[12];
    if (waymo)
        for (int q = 0; q < 12; ++q) Pi[q] = (double)pose_inv[(size_t)f * 16 + q];   // rows 0..2 of the 4x4, float32 -> float64
    const int ltab = lane_off[frame_lane[f]];

    for (int k = lane_id; k < nm; k += 64) {
        const int m = m0 + k;
        // (one wave per frame and a score of lanes at work: the kernel is as long as its chain of dependent loads.  Everything indexed by the mask is
        // requested up front, the lane point and the class's tables one round trip later; inside `if (valid)` each waited for the one before.)
        const int mp = medoid_pos[m], li = lane_idx[m];
        int cls = class_id[m];
        const float c_x = centroid[3 * m], c_y = centroid[3 * m + 1], c_z = centroid[3 * m + 2];
        const double ld_m = lane_dist[m], score_m = score[m];
        const bool valid = mp >= 0;
        if (cls < 0 || cls >= n_classes) cls = 0;
        const float yaw_m = lane[(size_t)(valid ? max(ltab + li, 0) : 0) * 3 + 2];        // (a mask without a medoid: the array's first point, unused)
        const int veh = is_vehicle[cls], grp_c = nms_group[cls];
        const double prior_l = prior_wlh[3 * cls + 0], prior_w = prior_wlh[3 * cls + 1];
        double tx = 0.0, ty = 0.0, tz = 0.0, qw = 1.0, qz = 0.0, yaw_out = 0.0, ld = 0.0;
        if (valid) {
            double cx = (double)c_x, cy = (double)c_y, cz = (double)c_z;
            if (waymo) {
                // src/waymo/2d_to_3d.py:812-816: np.dot(inv(float32 pose), [centroid, 1]) in float64
                const double gx = cx, gy = cy, gz = cz;
                cx = Pi[0] * gx + Pi[1] * gy + Pi[2] * gz + Pi[3];
                cy = Pi[4] * gx + Pi[5] * gy + Pi[6] * gz + Pi[7];
                cz = Pi[8] * gx + Pi[9] * gy + Pi[10] * gz + Pi[11];
            }
            const float yaw = yaw_m;                                           // :295, an f32 value
            yaw_out = (double)yaw; ld = ld_m;
            tx = cx; ty = cy; tz = cz;
            if (veh) {
                // :788-789 np.cos/np.sin of a float32 give float32
                const double cs = (double)cosf(yaw), sn = (double)sinf(yaw);
                // pyquaternion Quaternion(matrix=Rz): trace method on M^T
                if (cs < -cs) { const double t = 1.0 - cs - cs + 1.0; const double fct = 0.5 / sqrt(t); qw = (sn + sn) * fct; qz = t * fct; }
                else          { const double t = 1.0 + cs + cs + 1.0; const double fct = 0.5 / sqrt(t); qw = t * fct; qz = (sn + sn) * fct; }
                // push_centroid :164-198
                const double PI = 3.14159265358979323846;
                double phi = 2.0 * atan2(qw, qz);
                if (phi > PI) phi -= 2.0 * PI;
                if (phi <= -PI) phi += 2.0 * PI;
                double theta = -phi;
                if (theta != theta) theta = 0.5 * PI;
                const double ex = cx - ex0, ey = cy - ey0;
                double alpha = atan(fabs(ey) / fabs(ex));
                if (ex < 0) { if (ey < 0) alpha = -PI + alpha; else alpha = PI - alpha; }
                else        { if (ey < 0) alpha = -alpha; }
                const double l = prior_l, w = prior_w;
                const double o1 = fabs(w / (2.0 * sin(theta - alpha)));
                const double o2 = fabs(l / (2.0 * cos(theta - alpha)));
                double off = o1 < o2 ? o1 : o2;
                if (o1 != o1 || o2 != o2) off = NAN;
                tx = cx + off * cos(alpha);
                ty = cy + off * sin(alpha);
                if (waymo) {
                    // src/waymo/2d_to_3d.py:978-1001: align_mat = R_inv . Rz(global lane yaw); heading = as_euler('xyz')[2]
                    qw = atan2(Pi[4] * cs + Pi[5] * sn, Pi[0] * cs + Pi[1] * sn);
                    qz = 0.0;
                }
            } else if (waymo) {
                qw = 0.0; qz = 0.0;          // heading of identity (:1003-1010)
            }
        }
        double *b = box + (size_t)m * CM3D_BOX_STRIDE;
        b[0] = tx; b[1] = ty; b[2] = tz; b[3] = qw; b[4] = qz; b[5] = yaw_out; b[6] = ld; b[7] = score_m; b[8] = (double)cls;
        s_x[k] = tx; s_y[k] = ty; s_s[k] = score_m; s_lab[k] = grp_c;
        s_valid[k] = valid; s_sup[k] = 0; s_keep[k] = 0;
    }
    nms_phase(nm, lane_id, nms_thr, s_x, s_y, s_s, s_lab, s_order, s_valid, s_sup, s_keep);
    for (int k = lane_id; k < nm; k += 64) {
        const int fl = (s_valid[k] ? 1 : 0) | (s_keep[k] ? 2 : 0);
        flags[m0 + k] = fl;
        box[(size_t)(m0 + k) * CM3D_BOX_STRIDE + 9] = (double)fl;
    }
}

// Waymo: medoid (vehicle frame) -> global frame for the lane lookup (src/waymo/2d_to_3d.py:684-690):
// rotate with the float32 rotation, then translate, like LidarPointCloud.rotate/translate on one point.
__global__ void k_centroid_transform(const float *__restrict__ cin, const int32_t *__restrict__ medoid_pos,
                                     const int32_t *__restrict__ mask_frame, int n_masks, const float *__restrict__ pose_rt,
                                     float *__restrict__ cout)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_masks) return;
    float x = cin[3 * m], y = cin[3 * m + 1], z = cin[3 * m + 2];
    if (medoid_pos[m] >= 0) {
        const float *p = pose_rt + (size_t)mask_frame[m] * 12;     // [0..8] R row-major, [9..11] t
        float ax, ay, az;
        cm3d_rot3(p, x, y, z, ax, ay, az);
        x = ax + p[9]; y = ay + p[10]; z = az + p[11];
    }
    cout[3 * m] = x; cout[3 * m + 1] = y; cout[3 * m + 2] = z;
}

extern "C" int cm3d_centroid_transform(const float *centroid_in, const int32_t *medoid_pos, const int32_t *mask_frame,
                                       int32_t n_masks, const float *pose_rt, float *centroid_out, cm3d_stream_t stream)
{
    if (!centroid_in || !medoid_pos || !mask_frame || !pose_rt || !centroid_out || n_masks <= 0) return CM3D_ERR_ARG;
    hipLaunchKernelGGL(k_centroid_transform, dim3((n_masks + 255) / 256), dim3(256), 0, (hipStream_t)stream, centroid_in, medoid_pos,
                       mask_frame, n_masks, pose_rt, centroid_out);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}

extern "C" int cm3d_box_nms(const float *centroid, const int32_t *medoid_pos, const int32_t *mask_off, int32_t n_frames,
                            int32_t n_masks, const int32_t *class_id, const double *score, const float *lane,
                            const int32_t *lane_off, const int32_t *frame_lane, const int32_t *lane_idx,
                            const double *lane_dist, const double *prior_wlh, const int32_t *is_vehicle,
                            const int32_t *nms_group, const double *nms_thr, int32_t n_classes, const double *ego_xyz,
                            const float *pose_inv, double *box, int32_t *flags, cm3d_stream_t stream)
{
    if (!centroid || !medoid_pos || !mask_off || !class_id || !score || !lane || !lane_off || !frame_lane || !lane_idx ||
        !lane_dist || !prior_wlh || !is_vehicle || !nms_group || !nms_thr || (!ego_xyz && !pose_inv) || !box || !flags)
        return CM3D_ERR_ARG;
    if (n_frames <= 0 || n_masks <= 0 || n_classes <= 0) return CM3D_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_box_nms, dim3(n_frames), dim3(64), 0, st, centroid, medoid_pos, mask_off, class_id, score, lane,
                       lane_off, frame_lane, lane_idx, lane_dist, prior_wlh, is_vehicle, nms_group, nms_thr, n_classes, ego_xyz, pose_inv, box, flags);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}


// standalone NMS on float64 centres (frames given by frame_off), same device function as k_box_nms
__global__ __launch_bounds__(64) void k_circle_nms(const double *__restrict__ x, const double *__restrict__ y,
                                                   const double *__restrict__ score, const int32_t *__restrict__ label,
                                                   const int32_t *__restrict__ frame_off, const double *__restrict__ nms_thr,
                                                   int n_classes, int32_t *__restrict__ keep)
{
    __shared__ double s_x[BN_MAX], s_y[BN_MAX], s_s[BN_MAX];
    __shared__ int s_lab[BN_MAX], s_order[BN_MAX];
    __shared__ unsigned char s_valid[BN_MAX], s_sup[BN_MAX], s_keep[BN_MAX];
    const int f = blockIdx.x;
    const int m0 = frame_off[f];
    const int nm = min(frame_off[f + 1] - m0, BN_MAX);
    const int lane_id = threadIdx.x;
    for (int k = lane_id; k < nm; k += 64) {
        int cls = label[m0 + k];
        if (cls < 0 || cls >= n_classes) cls = 0;
        s_x[k] = x[m0 + k]; s_y[k] = y[m0 + k]; s_s[k] = score[m0 + k]; s_lab[k] = cls;
        s_valid[k] = 1; s_sup[k] = 0; s_keep[k] = 0;
    }
    nms_phase(nm, lane_id, nms_thr, s_x, s_y, s_s, s_lab, s_order, s_valid, s_sup, s_keep);
    for (int k = lane_id; k < nm; k += 64) keep[m0 + k] = s_keep[k];
}

extern "C" int cm3d_circle_nms(const double *x, const double *y, const double *score, const int32_t *label,
                               const int32_t *frame_off, int32_t n_frames, const double *nms_thr, int32_t n_classes,
                               int32_t *keep, cm3d_stream_t stream)
{
    if (!x || !y || !score || !label || !frame_off || !nms_thr || !keep) return CM3D_ERR_ARG;
    if (n_frames <= 0 || n_classes <= 0) return CM3D_ERR_ARG;
    hipLaunchKernelGGL(k_circle_nms, dim3(n_frames), dim3(64), 0, (hipStream_t)stream, x, y, score, label, frame_off, nms_thr,
                       n_classes, keep);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}

// ---------------------------------------------------------------------------
extern "C" int cm3d_abi_version(void) { return CM3D_ABI_VERSION; }

extern "C" const char *cm3d_error_string(int code)
{
    switch (code) {
    case CM3D_OK: return "ok";
    case CM3D_ERR_ARG: return "invalid argument (null pointer, non-positive size or unsupported shape)";
    case CM3D_ERR_LAUNCH: return "kernel launch failed";
    case CM3D_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown error";
    }
}
