// a4-a8: projection, in-image test, in-mask test, ordered compaction.
//   reference: src/nuscenes/2d_to_3d.py:553-620 (per mask: clone the cloud, 2x translate/rotate,
//   view_points utils/pcd.py:262-284, 5-way in-image test, floor, mask gather with the
//   floor(u)!=0 && floor(v)!=0 quirk, torch.where, two .cpu() index-tracking steps).
// Here every point is read ONCE (float4, coalesced), projected into every camera of its frame,
// and tested against the bit-packed eroded masks of that camera (bounding-box test first, so
// only points that can hit a mask touch mask memory).  Results leave as one hit word per point
// per 32 masks plus one count per (1024-point block, mask); an exclusive scan of those counts
// (k_hit_offsets) gives every block its exact output offset, and k_compact_hits writes the ascending index
// lists with wave ballot + mbcnt prefixes (no atomics on the output order, deterministic).
// HBM-bound: algorithmic bytes = 16 N + n*ceil(W*H/8) + 4*sum(M) + 4(n+1) per frame.
#include "common.h"
#include "worklist.h"
#include <cstdlib>

#define PH_THREADS 256
#define PH_PT 4                                   // points per thread
#define PH_BLOCK_PTS (PH_THREADS * PH_PT)
#define PH_MB 4                                   // masks of a camera handled together in the mask loop

// Block-local slot of a thread's j-th point: wave w owns the 256 consecutive points [256 w, 256 w + 256)
// (LiDAR points are stored ring by ring, so a wave then sees one short arc and few cameras); for a fixed j
// the 64 lanes read 64 consecutive points (1 KiB, coalesced).  Order inside a block = (wave, j, lane).
static __device__ __forceinline__ int ph_slot(int j) { return (int)(threadIdx.x >> 6) * (64 * PH_PT) + j * 64 + (int)(threadIdx.x & 63); }

typedef float f2 __attribute__((ext_vector_type(2)));     // two points side by side: v_pk_{add,mul,fma}_f32
#define PK_FMA(a, b, c) __builtin_elementwise_fma((a), (b), (c))

// k-sequential fma chain of a row-major 3x3 times two vectors (cm3d_rot3 for a pair of points)
static __device__ __forceinline__ void rot3_2(const float *R, f2 x, f2 y, f2 z, f2 &ox, f2 &oy, f2 &oz)
{
    f2 a = R[0] * x; a = PK_FMA((f2)(R[1]), y, a); a = PK_FMA((f2)(R[2]), z, a);
    f2 b = R[3] * x; b = PK_FMA((f2)(R[4]), y, b); b = PK_FMA((f2)(R[5]), z, b);
    f2 c = R[6] * x; c = PK_FMA((f2)(R[7]), y, c); c = PK_FMA((f2)(R[8]), z, c);
    ox = a; oy = b; oz = c;
}

// u = uh / zh and v = vh / zh for two points: the rcp + fma sequence hipcc emits for an IEEE float32 division,
// without v_div_scale / v_div_fixup and with the refined reciprocal shared by both quotients (see project_pair).
static __device__ __forceinline__ void ph_div_pair(f2 uh, f2 vh, f2 zh, f2 &u, f2 &v)
{
    f2 r = {__builtin_amdgcn_rcpf(zh.x), __builtin_amdgcn_rcpf(zh.y)};
    const f2 e = PK_FMA(-zh, r, (f2)(1.0f));
    r = PK_FMA(e, r, r);
    f2 q = uh * r;
    f2 t = PK_FMA(-zh, q, uh); q = PK_FMA(t, r, q);
    t = PK_FMA(-zh, q, uh);    u = PK_FMA(t, r, q);
    q = vh * r;
    t = PK_FMA(-zh, q, vh);    q = PK_FMA(t, r, q);
    t = PK_FMA(-zh, q, vh);    v = PK_FMA(t, r, q);
}

// Pinhole projection of TWO points through the reference's float32 op chain, branch-free and in packed
// float32 (the kernel is instruction-issue bound).  Returns the pixel codes (iv << 16 | iu) or -1.
// cm = camera record in LDS.
// NS / FL >= 0: compile-time stage count and translation flags (the three dataset layouts get their own
// straight-line code); NS < 0: read both from the record.
// FASTDIV: u = uh/zh and v = vh/zh through the same rcp + fma sequence hipcc emits for an IEEE division, but
// without v_div_scale / v_div_fixup and with the refined reciprocal shared by both quotients.  For zh in
// [1e-30, 1e30) the scaling steps are the identity unless the quotient is > 2^96 in magnitude or the numerator is
// below 2^-103 (quotient < 1/8) -- the range test rejects the point either way -- so accepted pixels are
// bit-identical (cm3d_selftest_div: 4e9 pairs on the device, tests/test_gpu_golden.py); `redo` is set when a
// lane that passes the depth test has zh outside that range, and the caller repeats the camera with the
// true division.
template <int NS, int FL, bool FASTDIV>
static __device__ __forceinline__ void project_pair(const float *cm, int ns_rt, int fl_rt, f2 px_, f2 py_, f2 pz_, float min_dist,
                                                    int W, int H, int &out0, int &out1, bool &redo)
{
    const int ns = NS >= 0 ? NS : ns_rt, fl = NS >= 0 ? FL : fl_rt;
    // up to three rigid stages `p += t_pre; p = R p; p += t_post` (nuScenes: global -> ego(cam time) ->
    // camera, 2d_to_3d.py:569-577; Waymo one stage; KITTI ref -> velo -> ref -> rect).  ns / fl are uniform.
    f2 ax = px_, ay = py_, az = pz_;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s < ns) {
            const float *st = cm + 15 * s;
            f2 x = ax, y = ay, z = az;
            if (fl & (1 << (2 * s))) { x = x + st[0]; y = y + st[1]; z = z + st[2]; }
            rot3_2(st + 3, x, y, z, ax, ay, az);
            if (fl & (2 << (2 * s))) { ax = ax + st[12]; ay = ay + st[13]; az = az + st[14]; }
        }
    }
    const f2 depth = az;                                      // :581
    // view_points: viewpad(4x4) @ [p;1], rows 0..2, k-sequential fma chain (pcd.py:269-282).  The chain's last
    // term, fma(0, 1, .), only turns -0 into +0, which cannot change an accepted pixel (|u| < 1 is rejected).
    const float *K = cm + 45;
    f2 uh = K[0] * ax; uh = PK_FMA((f2)(K[1]), ay, uh); uh = PK_FMA((f2)(K[2]), az, uh);
    f2 vh = K[3] * ax; vh = PK_FMA((f2)(K[4]), ay, vh); vh = PK_FMA((f2)(K[5]), az, vh);
    f2 zh = K[6] * ax; zh = PK_FMA((f2)(K[7]), ay, zh); zh = PK_FMA((f2)(K[8]), az, zh);
    f2 u, v;
    if (FASTDIV) {
        ph_div_pair(uh, vh, zh, u, v);
    } else {
        u = uh / zh; v = vh / zh;                             // IEEE division, any operand
    }
    // :597-603 in-image test, :605 floor, :608-613 truthiness quirk, folded into integer range checks:
    //   u > 0 && u < W-1 && floor(u) != 0   <=>   1 <= floor(u) <= W-2      (u = W-1 gives floor W-1;
    //   NaN converts to 0 and +-inf saturates, all outside the range).  The third row of the quirk,
    //   floor(zh/zh) != 0, holds whenever u is finite and non-zero (then zh/zh == 1 exactly).
    const int iu0 = (int)floorf(u.x), iv0 = (int)floorf(v.x), iu1 = (int)floorf(u.y), iv1 = (int)floorf(v.y);
    const bool d0 = depth.x > min_dist, d1 = depth.y > min_dist;
    const bool ok0 = d0 & ((unsigned)(iu0 - 1) <= (unsigned)(W - 3)) & ((unsigned)(iv0 - 1) <= (unsigned)(H - 3));
    const bool ok1 = d1 & ((unsigned)(iu1 - 1) <= (unsigned)(W - 3)) & ((unsigned)(iv1 - 1) <= (unsigned)(H - 3));
    out0 = ok0 ? ((iv0 << 16) | iu0) : -1;
    out1 = ok1 ? ((iv1 << 16) | iu1) : -1;
    if (FASTDIV) {
        const float hi = 9.99999e29f;
        redo = (d0 & !(zh.x >= 1.0e-30f && zh.x <= hi)) | (d1 & !(zh.y >= 1.0e-30f && zh.y <= hi));
    } else {
        redo = false;
    }
}

#define PH_NP (PH_PT / 2)                         // point pairs per thread
template <int NS, int FL>
static __device__ __forceinline__ void project_quad(const float *cm, int ns, int fl, const f2 (&X)[PH_NP], const f2 (&Y)[PH_NP],
                                                    const f2 (&Z)[PH_NP], float min_dist, int W, int H, int (&px)[PH_PT])
{
    bool redo = false;
#pragma unroll
    for (int h = 0; h < PH_NP; ++h) {
        bool r;
        project_pair<NS, FL, true>(cm, ns, fl, X[h], Y[h], Z[h], min_dist, W, H, px[2 * h], px[2 * h + 1], r);
        redo |= r;
    }
    if (__ballot(redo)) {             // never on real data: a depth-accepted point with |zh| outside [1e-30, 1e30)
#pragma unroll
        for (int h = 0; h < PH_NP; ++h) {
            bool r;
            project_pair<NS, FL, false>(cm, ns, fl, X[h], Y[h], Z[h], min_dist, W, H, px[2 * h], px[2 * h + 1], r);
        }
    }
}

// Conservative visibility cone of one camera, from its float32 record: a point can only pass the exact
// in-image test if it is in front of the camera by more than min_dist (minus a margin) and inside the
// circular cone around the optical axis that contains the whole image (plus a 2 degree margin).
// out: [0..2] camera centre (global), [3..5] optical axis (global), [6] min axial distance, [7] 1 + tan^2.
// The float32 error of this test is ~1e-4 m at nuScenes' global magnitudes -- orders below the margins --
// and a camera record the derivation does not cover (skew, non-trivial last row of K) disables the test.
static __device__ void cone_setup(const float *cm, int W, int H, float min_dist, float *out)
{
    const float *K = cm + 45;
    const int ns = (int)cm[54], fl = (int)cm[55];
    // compose the stages: p_cam = M p + c
    float M[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, c[3] = {0, 0, 0};
    for (int s = 0; s < ns && s < 3; ++s) {
        const float *st = cm + 15 * s, *Rm = st + 3;
        if (fl & (1 << (2 * s))) { c[0] += st[0]; c[1] += st[1]; c[2] += st[2]; }
        float Mn[9], cn[3];
        for (int r = 0; r < 3; ++r) {
            for (int q = 0; q < 3; ++q) Mn[3 * r + q] = Rm[3 * r] * M[q] + Rm[3 * r + 1] * M[3 + q] + Rm[3 * r + 2] * M[6 + q];
            cn[r] = Rm[3 * r] * c[0] + Rm[3 * r + 1] * c[1] + Rm[3 * r + 2] * c[2];
        }
        for (int q = 0; q < 9; ++q) M[q] = Mn[q];
        for (int q = 0; q < 3; ++q) c[q] = cn[q];
        if (fl & (2 << (2 * s))) { c[0] += st[12]; c[1] += st[13]; c[2] += st[14]; }
    }
    // camera centre: M o + c = 0  <=>  o = -M^T c  (M orthonormal)
    const float ox = -(M[0] * c[0] + M[3] * c[1] + M[6] * c[2]);
    const float oy = -(M[1] * c[0] + M[4] * c[1] + M[7] * c[2]);
    const float oz = -(M[2] * c[0] + M[5] * c[1] + M[8] * c[2]);
    out[0] = ox; out[1] = oy; out[2] = oz;
    out[3] = M[6]; out[4] = M[7]; out[5] = M[8];
    const bool plain = K[1] == 0.f && K[3] == 0.f && K[6] == 0.f && K[7] == 0.f && K[8] == 1.f && K[0] > 0.f && K[4] > 0.f;
    // orthonormality of M (a rotation up to float32 rounding)?
    float dev = 0.f;
    for (int r = 0; r < 3; ++r)
        for (int c = r; c < 3; ++c) {
            const float d = M[3 * r] * M[3 * c] + M[3 * r + 1] * M[3 * c + 1] + M[3 * r + 2] * M[3 * c + 2] - (r == c ? 1.f : 0.f);
            dev = fmaxf(dev, fabsf(d));
        }
    if (!plain || !(dev < 1e-3f)) { out[6] = -INFINITY; out[7] = INFINITY; return; }   // accept everything
    float t2 = 0.f;
    for (int cx = 0; cx < 2; ++cx)
        for (int cy = 0; cy < 2; ++cy) {
            const float a = ((cx ? (float)W : 0.f) - K[2]) / K[0], b = ((cy ? (float)H : 0.f) - K[5]) / K[4];
            t2 = fmaxf(t2, a * a + b * b);
        }
    const float t = sqrtf(t2), tm = 0.035f;            // tan(2 deg)
    const float tt = t * tm < 0.9f ? (t + tm) / (1.f - t * tm) : INFINITY;
    out[6] = min_dist - 0.05f - 1e-4f * fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz));
    out[7] = 1.f + tt * tt * 1.01f;
}

// grid (G, F).  A block walks 1024-point chunks of one frame (chunk = blockIdx.x, += gridDim.x),
// 4 points per thread (4 independent mask gathers in flight).  Per-frame tables (camera records, visibility
// cones, per-camera mask sets) are staged into LDS once per block.  Per wave and chunk:
//   for every camera that can see any of the wave's points (cone test): project the 4 points (pixel codes stay
//   in registers), then for every mask of that camera: bounding-box test, one mask word per candidate point,
//   bit test, per-mask hit count.
// ONE_PLANE (<= 32 masks per frame): the hit word of a point lives in a register; otherwise in the thread's own
// LDS slots, one per plane.
// FUSED: the kernel reads the RAW sweep rows itself (a2, reference :437-465: ego-box drop, sensor -> ego -> global, the
// very fma chains of k_sweep_xform), writes the transformed cloud for the medoid / the caller, and lists the dropped rows
// -- the cloud is then read once from HBM in the whole pass instead of raw + written + read again, and the sweep launch
// disappears.  Frames with at most PH_MAX_SWEEPS sweeps (cm3d_sweep_project_hits checks).
#define PH_MAX_SWEEPS 16
#define PH_DROP_CAP 2048                          // dropped rows a block collects in LDS; beyond: straight to the frame's list
struct PhSweepIn {
    const float *raw; int raw_stride; const int32_t *sweep_row_off; const float *sweep_xf; const int32_t *frame_sweep_off;
    int n_frames, n_sweeps; float halfw; float4 *points_out; int pt_cap; int32_t *pt_off_out, *removed_cnt, *removed_idx;
};

// raw rows of a chunk (the 4 columns the reference keeps); dead slots read as (1e30, 1e30): never dropped
template <bool FUSED>
static __device__ __forceinline__ void ph_load(float4 (&pt)[PH_PT], const float4 *__restrict__ points, const PhSweepIn &sw, int p0,
                                               int n, int chunk)
{
    const float qnan = __int_as_float(0x7FC00000);
#pragma unroll
    for (int j = 0; j < PH_PT; ++j) {
        const int i = chunk * PH_BLOCK_PTS + ph_slot(j);
        if (FUSED) {
            pt[j] = make_float4(1e30f, 1e30f, 0.f, 0.f);
            if (i < n) {
                const float *p = sw.raw + (size_t)(p0 + i) * sw.raw_stride;
                pt[j].x = __builtin_nontemporal_load(p); pt[j].y = __builtin_nontemporal_load(p + 1);
                pt[j].z = __builtin_nontemporal_load(p + 2); pt[j].w = __builtin_nontemporal_load(p + 3);
            }
        } else {
            pt[j] = i < n ? points[p0 + i] : make_float4(qnan, qnan, qnan, 0.f);
        }
    }
}

static __device__ __forceinline__ void ph_xform(const float *xf, float x, float y, float z, float &ox, float &oy, float &oz)
{
    // sensor -> ego (rotate then translate), ego -> global (2d_to_3d.py:450-457), as in k_sweep_xform
    float ax, ay, az;
    cm3d_rot3(xf, x, y, z, ax, ay, az);
    ax = ax + xf[9]; ay = ay + xf[10]; az = az + xf[11];
    cm3d_rot3(xf + 12, ax, ay, az, ox, oy, oz);
    ox = ox + xf[21]; oy = oy + xf[22]; oz = oz + xf[23];
}

// raw rows of a chunk -> global-frame points in place (dropped / dead slots become NaN points), stored to the cloud,
// dropped rows appended to the block's LDS list (or, once that is full, to the frame's list directly)
static __device__ __forceinline__ void ph_prepare(float4 (&pt)[PH_PT], const PhSweepIn &sw, int f, int sa, int ns, const int *s_srow,
                                                  int p0, int n, int chunk, int *s_drop, int *s_ndrop)
{
    const float qnan = __int_as_float(0x7FC00000);
    const int base = chunk * PH_BLOCK_PTS;
    // sweep of the wave's first and last row: equal for all but the waves that hold a sweep boundary
    int sw_lo = 0, sw_hi = 0;
    if (ns > 1) {
        const int r_lo = base + (int)(threadIdx.x >> 6) * (64 * PH_PT), r_hi = min(r_lo + 64 * PH_PT, n) - 1;
        for (int k = 1; k < ns; ++k) { sw_lo += s_srow[k] <= r_lo; sw_hi += s_srow[k] <= r_hi; }
    }
    const bool uni = sw_lo >= sw_hi;
    const float *xf_u = sw.sweep_xf + (size_t)(sa + __builtin_amdgcn_readfirstlane(sw_lo)) * CM3D_SWEEP_XF_STRIDE;   // scalar loads
#pragma unroll
    for (int j = 0; j < PH_PT; ++j) {
        const int i = base + ph_slot(j);
        const bool live = i < n;
        const float x = pt[j].x, y = pt[j].y, z = pt[j].z, w = pt[j].w;
        const bool drop = live && fabsf(x) < sw.halfw && fabsf(y) < sw.halfw;      // reference drops this row (2d_to_3d.py:442-445)
        float bx, by, bz;
        if (uni) {
            ph_xform(xf_u, x, y, z, bx, by, bz);
        } else {
            int k_sw = 0;
            for (int k = 1; k < ns; ++k) k_sw += s_srow[k] <= i;
            ph_xform(sw.sweep_xf + (size_t)(sa + k_sw) * CM3D_SWEEP_XF_STRIDE, x, y, z, bx, by, bz);
        }
        if (!live || drop) { bx = qnan; by = qnan; bz = qnan; }
        pt[j] = make_float4(bx, by, bz, w);
        if (live) sw.points_out[p0 + i] = pt[j];
        const uint64_t dm = __ballot(drop);
        if (dm) {
            const int cnt = (int)__popcll(dm);
            int pos0 = 0;
            if (cm3d_lane() == 0) pos0 = atomicAdd(s_ndrop, cnt);
            pos0 = __builtin_amdgcn_readfirstlane(pos0);
            const int n_lds = max(0, min(cnt, PH_DROP_CAP - pos0)), n_over = cnt - n_lds;
            int gbase = 0;
            if (n_over) {
                if (cm3d_lane() == 0) gbase = atomicAdd(&sw.removed_cnt[f], n_over);
                gbase = __builtin_amdgcn_readfirstlane(gbase);
            }
            if (drop) {
                const int r = cm3d_mbcnt(dm);
                if (r < n_lds) s_drop[pos0 + r] = i;
                else sw.removed_idx[p0 + gbase + (r - n_lds)] = i;
            }
        }
    }
}

template <bool ONE_PLANE, bool FUSED>
__global__ __launch_bounds__(PH_THREADS) void k_project_hits(
    const float4 *__restrict__ points, const int32_t *__restrict__ pt_off, const PhSweepIn sw, int n_points_total,
    const float *__restrict__ cams, int n_cams, const int32_t *__restrict__ mask_off,
    const int32_t *__restrict__ mask_cam, const int4 *__restrict__ bbox, const uint32_t *__restrict__ packed,
    int W, int H, int Wp, float min_dist, int nm_cap, int nblk_max, int chunks_per_block, uint32_t *__restrict__ hit_words,
    int32_t *__restrict__ hit_count, int32_t *__restrict__ blk_cnt, int32_t *__restrict__ status)
{
    const int f = blockIdx.y;
    int p0, n, sa = 0, ns = 0;
    if (FUSED) {
        sa = sw.frame_sweep_off[f];
        ns = sw.frame_sweep_off[f + 1] - sa;
        p0 = sw.sweep_row_off[sa];
        n = sw.sweep_row_off[sa + ns] - p0;
        if (blockIdx.x == 0 && threadIdx.x == 0) {              // what k_sweep_xform leaves behind for the later stages
            sw.pt_off_out[f] = p0;
            if (f == sw.n_frames - 1) {
                const int total_rows = sw.sweep_row_off[sw.n_sweeps];
                sw.pt_off_out[sw.n_frames] = total_rows;
                status[1] = total_rows;
                if (total_rows > sw.pt_cap) atomicOr(&status[0], 1);
                if (sw.frame_sweep_off[0] != 0 || sa + ns != sw.n_sweeps) atomicOr(&status[0], 4);      // sweeps outside every frame
            }
        }
        n = max(0, min(n, sw.pt_cap - p0));
    } else {
        p0 = pt_off[f];
        n = pt_off[f + 1] - p0;
    }
    const int nblk = (n + PH_BLOCK_PTS - 1) / PH_BLOCK_PTS;
    if ((int)blockIdx.x >= nblk) return;
    const int m0 = mask_off[f];
    int nm = mask_off[f + 1] - m0;
    if (nm > nm_cap) {                       // more masks than the caller's `planes` allows
        if (threadIdx.x == 0) atomicOr(&status[0], 4);
        nm = nm_cap;
    }
    const int planes = (nm + 31) >> 5;

    __shared__ float s_cam[CM3D_MAX_CAMS * CM3D_CAM_STRIDE];
    __shared__ float s_cone[CM3D_MAX_CAMS][8];           // conservative visibility cone per camera
    // dynamic LDS: masks of every camera as bit sets [n_cams][planes_cap], counts [chunks_per_block][nm_cap] (one
    // row per chunk this block walks), and (several planes only) hit words [planes][PH_BLOCK_PTS]
    extern __shared__ __align__(16) unsigned char s_dyn[];
    const int planes_cap = (nm_cap + 31) >> 5;
    uint32_t *s_cmask = reinterpret_cast<uint32_t *>(s_dyn);
    int *s_cnt = reinterpret_cast<int *>(s_cmask + CM3D_MAX_CAMS * planes_cap);
    uint32_t *s_bits = reinterpret_cast<uint32_t *>(s_cnt + chunks_per_block * nm_cap);

    // points of the first chunk are requested before the table staging so that both latencies overlap.
    // Slots past the end of the frame hold NaN points: every test below rejects them by itself.
    __shared__ int s_srow[FUSED ? PH_MAX_SWEEPS + 1 : 1];     // first row of each sweep of the frame (frame-local)
    __shared__ int s_drop[FUSED ? PH_DROP_CAP : 1];            // dropped rows of this block (frame-local row indices)
    __shared__ int s_ndrop, s_dropbase;
    float4 pt[PH_PT];
    ph_load<FUSED>(pt, points, sw, p0, n, (int)blockIdx.x);
    if (FUSED) {
        if (threadIdx.x <= ns && threadIdx.x <= PH_MAX_SWEEPS) s_srow[threadIdx.x] = sw.sweep_row_off[sa + threadIdx.x] - p0;
        if (threadIdx.x == 0) s_ndrop = 0;
    }
    for (int q = threadIdx.x; q < n_cams * CM3D_CAM_STRIDE; q += PH_THREADS)
        s_cam[q] = cams[(size_t)f * n_cams * CM3D_CAM_STRIDE + q];
    for (int q = threadIdx.x; q < CM3D_MAX_CAMS * planes_cap; q += PH_THREADS) s_cmask[q] = 0u;
    for (int k = threadIdx.x; k < chunks_per_block * nm_cap; k += PH_THREADS) s_cnt[k] = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < nm; k += PH_THREADS) {
        const int c = mask_cam[m0 + k];
        if (c < 0 || c >= n_cams) atomicOr(&status[0], 4);           // such a mask gets no points
        else atomicOr(&s_cmask[c * planes_cap + (k >> 5)], 1u << (k & 31));
    }
    // visibility cones from the staged records (LDS reads; no dependent global loads)
    if (threadIdx.x < n_cams) cone_setup(s_cam + threadIdx.x * CM3D_CAM_STRIDE, W, H, min_dist, s_cone[threadIdx.x]);
    __syncthreads();

    const size_t mask_words = (size_t)H * Wp;
    const int lane = cm3d_lane();
    // The waves of a block walk its chunks without meeting: the per-(chunk, mask) counts collect in LDS rows and are
    // flushed once, behind the only barrier after the loop (a barrier per chunk made every wave wait for the
    // slowest one of each chunk).
    int ci = 0;
    for (int chunk = blockIdx.x; chunk < nblk; chunk += gridDim.x, ++ci) {
        const int base = chunk * PH_BLOCK_PTS;
        int *s_cnt_row = s_cnt + ci * nm_cap;
        static_assert(PH_PT % 2 == 0, "points are handled in pairs");
        if (FUSED) ph_prepare(pt, sw, f, sa, ns, s_srow, p0, n, chunk, s_drop, &s_ndrop);
        // FUSED: the next chunk's raw rows (HBM, 20-byte stride) are requested now, into their own registers, and
        // arrive under the camera loop (887 k against 856 k frames/s with the request at the end of the chunk)
        float4 nxt[PH_PT];
        if (FUSED && chunk + (int)gridDim.x < nblk) ph_load<FUSED>(nxt, points, sw, p0, n, chunk + (int)gridDim.x);
        f2 X[PH_NP], Y[PH_NP], Z[PH_NP];
#pragma unroll
        for (int h = 0; h < PH_NP; ++h) {
            X[h] = (f2){pt[2 * h].x, pt[2 * h + 1].x};
            Y[h] = (f2){pt[2 * h].y, pt[2 * h + 1].y};
            Z[h] = (f2){pt[2 * h].z, pt[2 * h + 1].z};
        }
        uint32_t bits[PH_PT];
#pragma unroll
        for (int j = 0; j < PH_PT; ++j) bits[j] = 0;
        if (!ONE_PLANE) {
            for (int pl = 0; pl < planes; ++pl)
#pragma unroll
                for (int j = 0; j < PH_PT; ++j) s_bits[pl * PH_BLOCK_PTS + ph_slot(j)] = 0u;      // this thread's own slots
        }
#pragma unroll 1
        for (int c = 0; c < n_cams; ++c) {
            // conservative pre-test (a superset of the exact in-image test): is any of this wave's points
            // inside the camera's visibility cone?  A wave's 4 x 64 points are consecutive in the sweep, i.e. a
            // short arc of the scan, and most cameras are rejected here for the whole wave.
            //   inside  <=>  sdist - c6 >= 0  and  c7 * sdist^2 - r2 >= 0   (NaN compares false)
            const float *cn = s_cone[c];
            float inside;
            {
                f2 m[PH_NP];
#pragma unroll
                for (int h = 0; h < PH_NP; ++h) {
                    const f2 vx = X[h] - cn[0], vy = Y[h] - cn[1], vz = Z[h] - cn[2];
                    f2 sd = cn[3] * vx; sd = PK_FMA((f2)(cn[4]), vy, sd); sd = PK_FMA((f2)(cn[5]), vz, sd);
                    f2 r2 = vx * vx; r2 = PK_FMA(vy, vy, r2); r2 = PK_FMA(vz, vz, r2);
                    const f2 t = PK_FMA(cn[7] * sd, sd, -r2);
                    m[h] = __builtin_elementwise_min(sd - cn[6], t);
                }
                inside = fmaxf(m[0].x, m[0].y);
#pragma unroll
                for (int h = 1; h < PH_NP; ++h) inside = fmaxf(inside, fmaxf(m[h].x, m[h].y));
            }
            if (!__ballot(inside >= 0.0f)) continue;
            const int ns = __builtin_amdgcn_readfirstlane((int)s_cam[c * CM3D_CAM_STRIDE + 54]);
            const int fl = __builtin_amdgcn_readfirstlane((int)s_cam[c * CM3D_CAM_STRIDE + 55]);
            const float *cm = s_cam + c * CM3D_CAM_STRIDE;
            int px[PH_PT];
            if (ns == 2 && fl == 5) project_quad<2, 5>(cm, ns, fl, X, Y, Z, min_dist, W, H, px);          // nuScenes
            else if (ns == 1 && fl == 1) project_quad<1, 1>(cm, ns, fl, X, Y, Z, min_dist, W, H, px);     // Waymo
            else if (ns == 3 && fl == 10) project_quad<3, 10>(cm, ns, fl, X, Y, Z, min_dist, W, H, px);   // KITTI
            else project_quad<-1, 0>(cm, ns, fl, X, Y, Z, min_dist, W, H, px);
            int pxall = px[0];
#pragma unroll
            for (int j = 1; j < PH_PT; ++j) pxall &= px[j];
            if (!__ballot(pxall >= 0)) continue;                 // no point of the wave in this image
            // px = iv << 16 | iu; px = -1 gives iv = -1
#define PX_IU(j) (px[j] & 0xFFFF)
#define PX_IV(j) (px[j] >> 16)
            // the masks of this camera, PH_MB at a time: all bounding boxes, then all mask words of the batch are
            // requested before the first one is used (one memory round trip per batch instead of one per mask)
            for (int pl = 0; pl < planes; ++pl) {
                uint32_t mm = __builtin_amdgcn_readfirstlane(s_cmask[c * planes_cap + pl]);
                while (mm) {
                    int kb[PH_MB];
#pragma unroll
                    for (int b = 0; b < PH_MB; ++b) {
                        kb[b] = mm ? __builtin_ctz(mm) : -1;
                        mm = mm ? (mm & (mm - 1)) : 0u;
                    }
                    int4 bb[PH_MB];
#pragma unroll
                    for (int b = 0; b < PH_MB; ++b) bb[b] = bbox[m0 + pl * 32 + max(kb[b], 0)];      // uniform: scalar loads
                    uint32_t word[PH_MB][PH_PT];
#pragma unroll
                    for (int b = 0; b < PH_MB; ++b) {
#pragma unroll
                        for (int j = 0; j < PH_PT; ++j) word[b][j] = 0u;
                        const int x0 = __builtin_amdgcn_readfirstlane(bb[b].x), y0 = __builtin_amdgcn_readfirstlane(bb[b].y);
                        const int rx = __builtin_amdgcn_readfirstlane(bb[b].z) - x0, ry = __builtin_amdgcn_readfirstlane(bb[b].w) - y0;
                        if (kb[b] < 0 || (rx | ry) < 0) continue;          // no such mask / empty mask (wave-uniform)
                        const uint32_t *mw = packed + (size_t)(m0 + pl * 32 + kb[b]) * mask_words;
#pragma unroll
                        for (int j = 0; j < PH_PT; ++j) {
                            // iv = -1 < y0 fails the unsigned range test by itself
                            const bool cand = ((unsigned)(PX_IU(j) - x0) <= (unsigned)rx) & ((unsigned)(PX_IV(j) - y0) <= (unsigned)ry);
                            if (cand) word[b][j] = mw[(size_t)PX_IV(j) * Wp + (PX_IU(j) >> 5)];
                        }
                    }
#pragma unroll
                    for (int b = 0; b < PH_MB; ++b) {
                        if (kb[b] < 0) continue;
                        int cnt = 0;
#pragma unroll
                        for (int j = 0; j < PH_PT; ++j) {
                            const bool hit = (word[b][j] >> (px[j] & 31)) & 1u;          // word = 0 for a non-candidate
                            if (ONE_PLANE) bits[j] |= (hit ? 1u : 0u) << kb[b];
                            else if (hit) s_bits[pl * PH_BLOCK_PTS + ph_slot(j)] |= 1u << kb[b];
                            cnt += __popcll(__ballot(hit));
                        }
                        if (cnt && lane == 0) atomicAdd(&s_cnt_row[pl * 32 + kb[b]], cnt);
                    }
                }
            }
        }
        // prefetch the next chunk's points (if this block has one) under the count flush
        if (FUSED) {
#pragma unroll
            for (int j = 0; j < PH_PT; ++j) pt[j] = nxt[j];
        } else if (chunk + (int)gridDim.x < nblk) {
            ph_load<FUSED>(pt, points, sw, p0, n, chunk + (int)gridDim.x);
        }
#pragma unroll
        for (int j = 0; j < PH_PT; ++j) {
            const int idx = base + ph_slot(j);
            if (idx < n) {
                if (ONE_PLANE) hit_words[(size_t)p0 + idx] = bits[j];
                else
                    for (int pl = 0; pl < planes; ++pl)
                        hit_words[(size_t)pl * n_points_total + p0 + idx] = s_bits[pl * PH_BLOCK_PTS + ph_slot(j)];
            }
        }
    }
    __syncthreads();
    // per-(chunk, mask) counts: exact output offsets come from their exclusive scan (k_hit_offsets)
    for (int k = threadIdx.x; k < nm; k += PH_THREADS) {
        int tot = 0;
        int r = 0;
        for (int chunk = blockIdx.x; chunk < nblk; chunk += gridDim.x, ++r) {
            const int c = s_cnt[r * nm_cap + k];
            blk_cnt[((size_t)f * nblk_max + chunk) * nm_cap + k] = c;
            tot += c;
        }
        if (tot) atomicAdd(&hit_count[m0 + k], tot);
    }
    if (FUSED) {
        // the block's dropped rows join the frame's list with ONE global atomic (see k_sweep_xform)
        const int nd = min(s_ndrop, PH_DROP_CAP);
        if (nd == 0) return;
        if (threadIdx.x == 0) s_dropbase = atomicAdd(&sw.removed_cnt[f], nd);
        __syncthreads();
        const int dbase = p0 + s_dropbase;
        for (int i = threadIdx.x; i < nd; i += PH_THREADS) sw.removed_idx[dbase + i] = s_drop[i];
    }
}

// One workgroup per frame: exclusive scans of hit_count and of the medoid tile counts for the frame's masks
// (the frame's base = sum over all earlier masks, recomputed by every workgroup -- n_masks loads from L2 are
// cheaper than a launch boundary), then every mask's per-block counts turned into exclusive output offsets,
// in place.  Status bookkeeping by the last workgroup.
__global__ __launch_bounds__(1024) void k_hit_offsets(const int32_t *__restrict__ hit_count, int n_masks,
                                                      const int32_t *__restrict__ pt_off, const int32_t *__restrict__ mask_off,
                                                      int n_frames, int nm_cap, int nblk_max, int32_t *__restrict__ hit_off,
                                                      int32_t *__restrict__ tile_off, int32_t *blk_cnt, int idx_cap,
                                                      int32_t *__restrict__ status)
{
    __shared__ int s_part[16];
    __shared__ int s_red[2][16];
    const int f = blockIdx.x, t = threadIdx.x;
    const int m0 = mask_off[f];
    const int m1 = f == n_frames - 1 ? n_masks : mask_off[f + 1];
    const int nm = min(mask_off[f + 1] - m0, nm_cap);
    const int n = pt_off[f + 1] - pt_off[f];
    const int nblk = (n + PH_BLOCK_PTS - 1) / PH_BLOCK_PTS;
    // base offsets of the frame
    int a = 0, b = 0;
    for (int i = t; i < m0; i += 1024) {
        const int v = hit_count[i];
        a += v;
        b += (v + CM3D_MEDOID_TILE - 1) / CM3D_MEDOID_TILE;
    }
    a = cm3d_wave_sum(a); b = cm3d_wave_sum(b);
    if (cm3d_lane() == 0) { s_red[0][t >> 6] = a; s_red[1][t >> 6] = b; }
    __syncthreads();
    int carry = 0, tcarry = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { carry += s_red[0][w]; tcarry += s_red[1][w]; }
    for (int kb = 0; kb < m1 - m0; kb += 1024) {
        const int k = kb + t;
        const bool live = k < m1 - m0;
        const int v = live ? hit_count[m0 + k] : 0;
        const int tl = (v + CM3D_MEDOID_TILE - 1) / CM3D_MEDOID_TILE;
        int tot, ttot;
        const int ex = cm3d_block1024_excl_scan(v, s_part, tot);
        const int tex = cm3d_block1024_excl_scan(tl, s_part, ttot);
        if (live) { hit_off[m0 + k] = carry + ex; tile_off[m0 + k] = tcarry + tex; }
        if (k < nm) {
            // block by block: count -> exclusive offset (loads of 8 blocks in flight)
            int run = carry + ex;
            int32_t *p = blk_cnt + (size_t)f * nblk_max * nm_cap + k;
            int bi = 0;
            for (; bi + 8 <= nblk; bi += 8) {
                int c[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) c[u] = p[(size_t)(bi + u) * nm_cap];
#pragma unroll
                for (int u = 0; u < 8; ++u) { p[(size_t)(bi + u) * nm_cap] = run; run += c[u]; }
            }
            for (; bi < nblk; ++bi) {
                const int c = p[(size_t)bi * nm_cap];
                p[(size_t)bi * nm_cap] = run;
                run += c;
            }
        }
        carry += tot; tcarry += ttot;
        __syncthreads();
    }
    if (f == n_frames - 1 && t == 0) {
        hit_off[n_masks] = carry;
        tile_off[n_masks] = tcarry;
        status[2] = carry;
        status[3] = tcarry;
        if (carry > idx_cap) atomicOr(&status[0], 2);
    }
}

// grid (nblk_max, F): the same 1024 points and the same thread<->point map as k_project_hits.
// Order inside a block is (wave, j, lane); per present mask bit: counts per wave through LDS, then
// ballot + mbcnt positions on top of the block's exclusive offset.
__global__ __launch_bounds__(PH_THREADS) void k_compact_hits(const uint32_t *__restrict__ hit_words, int n_points_total,
                                                             const int32_t *__restrict__ pt_off,
                                                             const int32_t *__restrict__ mask_off, int nm_cap, int nblk_max,
                                                             const int32_t *__restrict__ blk_base,
                                                             const int32_t *__restrict__ removed_cnt,
                                                             const int32_t *__restrict__ removed_idx,
                                                             int32_t *__restrict__ hit_idx, int32_t *__restrict__ hit_row, int idx_cap,
                                                             int n_frames, int n_masks, const int32_t *__restrict__ hit_off,
                                                             const int32_t *__restrict__ tile_off, int tile_cap,
                                                             TileDesc *__restrict__ tile_work)
{
    const int row0 = tile_work ? 1 : 0;
    if (tile_work && blockIdx.y == 0) {
        // extra row of the grid, dispatched first: its workgroups build the medoid stage's work list from the offsets
        // the previous launch wrote (each its own share, without talking to each other), beside -- and hidden under --
        // the compaction of the hit words
        __shared__ int s_hist[MD_CLASSES], s_cur[MD_CLASSES];
        md_build_worklist<PH_THREADS, 2>(n_masks, hit_off, tile_off, idx_cap, tile_cap, tile_work, s_hist, s_cur, (int)blockIdx.x,
                                      (int)gridDim.x);
        return;
    }
    const int f = (int)blockIdx.y - row0, chunk = blockIdx.x;
    const int p0 = pt_off[f], n = pt_off[f + 1] - p0;
    const int base = chunk * PH_BLOCK_PTS;
    if (base >= n) return;
    const int m0 = mask_off[f];
    const int nm = min(mask_off[f + 1] - m0, nm_cap);
    const int planes = (nm + 31) >> 5;
    const int wave = threadIdx.x >> 6, lane = cm3d_lane();
    // the hit words and the block's output offsets of plane 0 are requested first: their latency overlaps the
    // dropped-row bookkeeping below
    const int run_first = lane < min(nm, 32) ? blk_base[((size_t)f * nblk_max + chunk) * nm_cap + lane] : 0;
    int idx[PH_PT];
    uint32_t w_first[PH_PT];
#pragma unroll
    for (int j = 0; j < PH_PT; ++j) {
        idx[j] = base + ph_slot(j);
        w_first[j] = idx[j] < n ? hit_words[(size_t)p0 + idx[j]] : 0u;
    }
    // rows the sweep preparation dropped (ego box): the emitted index of a point is its row index minus the
    // number of dropped rows before it, i.e. its index in the reference's compacted cloud
    __shared__ uint32_t s_rm[PH_BLOCK_PTS / 32];      // dropped rows of this block, one bit per row
    __shared__ int s_rm_before;                       // dropped rows of the frame before this block
    __shared__ int s_rm_pre[PH_BLOCK_PTS / 32];
    const int n_rm = removed_cnt ? removed_cnt[f] : 0;
    if (n_rm > 0) {
        if (threadIdx.x < PH_BLOCK_PTS / 32) s_rm[threadIdx.x] = 0u;
        if (threadIdx.x == 0) s_rm_before = 0;
        __syncthreads();
        int before = 0;
        for (int q = threadIdx.x; q < n_rm; q += PH_THREADS) {
            const int r = removed_idx[p0 + q];
            if (r < base) ++before;
            else if (r < base + PH_BLOCK_PTS) atomicOr(&s_rm[(r - base) >> 5], 1u << ((r - base) & 31));
        }
        before = cm3d_wave_sum(before);
        if (cm3d_lane() == 0 && before) atomicAdd(&s_rm_before, before);
        __syncthreads();
        if (threadIdx.x < PH_BLOCK_PTS / 32) {       // exclusive prefix over the bitmap words
            int pre = 0;
            for (int wq = 0; wq < (int)threadIdx.x; ++wq) pre += __popc(s_rm[wq]);
            s_rm_pre[threadIdx.x] = pre;
        }
        __syncthreads();
    }
    __shared__ int s_c[PH_THREADS / 64][32];
    for (int plane = 0; plane < planes; ++plane) {
        const uint32_t *hw = hit_words + (size_t)plane * n_points_total + p0;
        uint32_t w[PH_PT];
        uint32_t any = 0;
#pragma unroll
        for (int j = 0; j < PH_PT; ++j) { w[j] = plane == 0 ? w_first[j] : (idx[j] < n ? hw[idx[j]] : 0u); any |= w[j]; }
        const uint32_t orw = cm3d_wave_or(any);
        int mycnt = 0;                         // lane b < 32: hits of mask bit b in this wave's 256 points
        for (uint32_t r = orw; r; r &= r - 1) {
            const int b = __builtin_ctz(r);
            int c = 0;
#pragma unroll
            for (int j = 0; j < PH_PT; ++j) c += __popcll(__ballot((w[j] >> b) & 1u));
            if (lane == b) mycnt = c;
        }
        __syncthreads();               // previous plane's readers are done with s_c
        if (lane < 32) s_c[wave][lane] = mycnt;
        __syncthreads();
        if (orw) {
            // lane b < 32: running output position of mask bit b, starting at this wave's first point
            int run = 0;
            if (lane < 32) {
                const int k = plane * 32 + lane;
                run = plane == 0 ? run_first : (k < nm ? blk_base[((size_t)f * nblk_max + chunk) * nm_cap + k] : 0);
                for (int w2 = 0; w2 < wave; ++w2) run += s_c[w2][lane];
            }
            for (uint32_t r = orw; r; r &= r - 1) {
                const int b = __builtin_ctz(r);
#pragma unroll
                for (int j = 0; j < PH_PT; ++j) {
                    const bool mine = (w[j] >> b) & 1u;
                    const uint64_t mk = __ballot(mine);
                    const int basepos = __builtin_amdgcn_readlane(run, b);
                    if (mine) {
                        const int pos = basepos + cm3d_mbcnt(mk);
                        if (pos >= 0 && pos < idx_cap) {
                            int dropped = 0;
                            if (n_rm > 0) {
                                const int loc = idx[j] - base;
                                dropped = s_rm_before + s_rm_pre[loc >> 5] +
                                          __popc(s_rm[loc >> 5] & ((1u << (loc & 31)) - 1u));
                            }
                            hit_idx[pos] = idx[j] - dropped;
                            hit_row[pos] = idx[j];
                        }
                    }
                    if (lane == b) run += __popcll(mk);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Diagnostic: pseudo-random (numerator, denominator) pairs -- denominators log-uniform over the shortcut's domain
// [1e-30, 1e30), numerators log-uniform over +-[1e-38, 1e38] or small integers times the denominator +- a few ulp
// (quotients next to integers are what floor() is sensitive to) -- through ph_div_pair and through the IEEE division.
// n_bad[0] counts the pairs whose true quotient q has 1/8 <= |q| < 2^96 and differs in any bit: must stay 0.
// n_bad[1] counts differing pairs with |q| < 1/8 (a numerator below 2^-103, where v_div_scale would have rescaled):
// both quotients are then below 1 in magnitude and the pixel range test rejects the point either way.
__global__ __launch_bounds__(256) void k_selftest_div(uint64_t seed, uint64_t count, unsigned long long *n_bad)
{
    unsigned long long bad = 0, benign = 0;
    uint64_t x = seed ^ (0x9E3779B97F4A7C15ull * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1));
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    auto check = [&](float got, float want) {
        if (__float_as_uint(got) == __float_as_uint(want)) return;
        const float mag = fabsf(want);
        if (mag >= 0.125f && mag < 7.9e28f) ++bad;
        else if (mag < 0.125f && fabsf(got) < 1.0f) ++benign;
        else if (mag < 0.125f) ++bad;                        // a small quotient that the shortcut made large: not benign
    };
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t a = next(), b = next();
        // denominator: exponent 28..226 (about 1e-30..1e30), random mantissa, positive
        float den = __uint_as_float((uint32_t)((28 + (a % 199)) << 23) | (uint32_t)((a >> 20) & 0x7FFFFF));
        if (!(den >= 1.0e-30f && den <= 9.99999e29f)) den = 2.5f;
        float num;
        if (b & 1) {                 // quotient near an integer: (k * den) +- a few ulp
            const float k = (float)((b >> 8) % 4096);
            const float tweak = __uint_as_float(__float_as_uint(k * den) + (int)((b >> 40) % 9) - 4);
            num = (b & 2) ? -tweak : tweak;
        } else {
            num = __uint_as_float((uint32_t)((1 + ((b >> 8) % 253)) << 23) | (uint32_t)((b >> 20) & 0x7FFFFF) | (uint32_t)((b >> 1) & 1) << 31);
        }
        f2 u, v;
        ph_div_pair((f2){num, -num}, (f2){num * 0.5f, num}, (f2){den, den}, u, v);
        check(u.x, num / den); check(v.y, num / den); check(u.y, -num / den); check(v.x, (num * 0.5f) / den);
    }
    if (bad) atomicAdd(&n_bad[0], bad);
    if (benign) atomicAdd(&n_bad[1], benign);
}

extern "C" int cm3d_selftest_div(uint64_t seed, uint64_t count, uint64_t *n_bad, cm3d_stream_t stream)
{
    if (!n_bad || count == 0) return CM3D_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(n_bad, 0, 2 * sizeof(uint64_t), st) != hipSuccess) return CM3D_ERR_LAUNCH;
    hipLaunchKernelGGL(k_selftest_div, dim3(8192), dim3(256), 0, st, seed, count, (unsigned long long *)n_bad);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}

__global__ void k_batch_begin(int32_t *__restrict__ status, int32_t *__restrict__ hit_count, int n_masks,
                              int32_t *__restrict__ removed_cnt, int n_frames)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < CM3D_STATUS_WORDS) status[i] = 0;
    if (i < n_masks) hit_count[i] = 0;
    if (removed_cnt && i < n_frames) removed_cnt[i] = 0;
}

extern "C" int cm3d_batch_begin(int32_t *status, int32_t *hit_count, int32_t n_masks, int32_t *removed_cnt, int32_t n_frames,
                                cm3d_stream_t stream)
{
    if (!status || !hit_count || n_masks <= 0 || (removed_cnt && n_frames <= 0)) return CM3D_ERR_ARG;
    const int n = n_masks > n_frames ? n_masks : n_frames;
    hipLaunchKernelGGL(k_batch_begin, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, status, hit_count, n_masks, removed_cnt,
                       removed_cnt ? n_frames : 0);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}

static inline int ph_nm_cap(int planes)
{
    int c = planes * 32;
    return c > CM3D_MAX_MASKS_PER_FRAME ? CM3D_MAX_MASKS_PER_FRAME : c;
}

extern "C" int64_t cm3d_project_workspace_bytes(int32_t n_frames, int32_t max_pts_per_frame, int32_t planes)
{
    if (n_frames <= 0 || max_pts_per_frame <= 0 || planes <= 0) return 0;
    const int64_t nblk_max = (max_pts_per_frame + PH_BLOCK_PTS - 1) / PH_BLOCK_PTS;
    return (int64_t)n_frames * nblk_max * ph_nm_cap(planes) * (int64_t)sizeof(int32_t);
}

static int ph_launch(const PhSweepIn *fused, const float *points, const int32_t *pt_off, int32_t n_frames, int32_t max_pts_per_frame,
                     int32_t n_points_total, const float *cams, int32_t n_cams, const int32_t *mask_off, const int32_t *mask_cam,
                     const int32_t *bbox, const uint32_t *packed, int32_t n_masks, int32_t W, int32_t H, float min_dist,
                     int32_t planes, uint32_t *hit_words, int32_t *hit_count, int32_t *status, void *workspace,
                     int64_t workspace_bytes, cm3d_stream_t stream)
{
    if (!cams || !mask_off || !mask_cam || !bbox || !packed || !hit_words || !hit_count || !status || !workspace) return CM3D_ERR_ARG;
    if (n_frames <= 0 || max_pts_per_frame <= 0 || n_points_total <= 0 || n_cams <= 0 || n_cams > CM3D_MAX_CAMS ||
        n_masks <= 0 || W <= 1 || H <= 1 || W > 32767 || H > 32767 || planes <= 0)
        return CM3D_ERR_ARG;
    if (workspace_bytes < cm3d_project_workspace_bytes(n_frames, max_pts_per_frame, planes)) return CM3D_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int Wp = (W + 31) / 32;
    const int nblk_max = (max_pts_per_frame + PH_BLOCK_PTS - 1) / PH_BLOCK_PTS;
    const int nm_cap = ph_nm_cap(planes);
    // enough blocks to fill the chip (256 CUs x 6 resident), each walking a few chunks of its frame so that the
    // per-block table staging is amortised (measured on C2: 2304 -> 81 us, 4608 -> 83 us, 1536 -> 89 us)
    int gx = nblk_max;
    static long long max_blocks = 0;             // beyond that blocks walk several chunks
    if (!max_blocks) { const char *e = getenv("CM3D_PH_MAXBLK"); max_blocks = e ? atoll(e) : 2304; }
    if ((long long)gx * n_frames > max_blocks) gx = (int)((max_blocks + n_frames - 1) / n_frames);
    if (gx > nblk_max) gx = nblk_max;
    if (gx < 1) gx = 1;
    const int planes_cap = (nm_cap + 31) / 32;
    {   // the per-chunk count rows of a block live in LDS: at most 16 KiB of them (with 1024 masks per frame the
        // per-thread hit words already take 128 KiB of the 160)
        const int max_cpb = 4096 / nm_cap > 1 ? 4096 / nm_cap : 1;
        const int gx_min = (nblk_max + max_cpb - 1) / max_cpb;
        if (gx < gx_min) gx = gx_min;
    }
    const int chunks_per_block = (nblk_max + gx - 1) / gx;
    size_t lds = (size_t)CM3D_MAX_CAMS * planes_cap * sizeof(uint32_t) + (size_t)chunks_per_block * nm_cap * sizeof(int);
    PhSweepIn none = {};
    const PhSweepIn sw = fused ? *fused : none;
#define PH_LAUNCH(ONE, FUSED)                                                                                                    \
    hipLaunchKernelGGL((k_project_hits<ONE, FUSED>), dim3(gx, n_frames), dim3(PH_THREADS), lds, st, (const float4 *)points, pt_off, \
                       sw, n_points_total, cams, n_cams, mask_off, mask_cam, (const int4 *)bbox, packed, W, H, Wp, min_dist,     \
                       nm_cap, nblk_max, chunks_per_block, hit_words, hit_count, (int32_t *)workspace, status)
    if (planes_cap == 1) {
        if (fused) PH_LAUNCH(true, true); else PH_LAUNCH(true, false);
    } else {
        lds += (size_t)planes_cap * PH_BLOCK_PTS * sizeof(uint32_t);
        static size_t lds_allowed[2] = {48 * 1024, 48 * 1024};
        if (lds > lds_allowed[fused ? 1 : 0]) {
            const void *fn = fused ? (const void *)k_project_hits<false, true> : (const void *)k_project_hits<false, false>;
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return CM3D_ERR_LAUNCH;
            lds_allowed[fused ? 1 : 0] = lds;
        }
        if (fused) PH_LAUNCH(false, true); else PH_LAUNCH(false, false);
    }
#undef PH_LAUNCH
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}

extern "C" int cm3d_project_hits(const float *points, const int32_t *pt_off, int32_t n_frames, int32_t max_pts_per_frame,
                                 int32_t n_points_total, const float *cams, int32_t n_cams, const int32_t *mask_off,
                                 const int32_t *mask_cam, const int32_t *bbox, const uint32_t *packed, int32_t n_masks,
                                 int32_t W, int32_t H, float min_dist, int32_t planes, uint32_t *hit_words,
                                 int32_t *hit_count, int32_t *status, void *workspace, int64_t workspace_bytes,
                                 cm3d_stream_t stream)
{
    if (!points || !pt_off) return CM3D_ERR_ARG;
    return ph_launch(nullptr, points, pt_off, n_frames, max_pts_per_frame, n_points_total, cams, n_cams, mask_off, mask_cam, bbox,
                     packed, n_masks, W, H, min_dist, planes, hit_words, hit_count, status, workspace, workspace_bytes, stream);
}

extern "C" int cm3d_sweep_project_hits(const float *raw, int32_t raw_stride, const int32_t *sweep_row_off, int32_t n_sweeps,
                                       int32_t max_sweeps_per_frame, const float *sweep_xf, const int32_t *frame_sweep_off,
                                       float halfw, float *points, int32_t pt_cap, int32_t *pt_off, int32_t *removed_cnt,
                                       int32_t *removed_idx, int32_t n_frames, int32_t max_pts_per_frame, int32_t n_points_total,
                                       const float *cams, int32_t n_cams, const int32_t *mask_off, const int32_t *mask_cam,
                                       const int32_t *bbox, const uint32_t *packed, int32_t n_masks, int32_t W, int32_t H,
                                       float min_dist, int32_t planes, uint32_t *hit_words, int32_t *hit_count, int32_t *status,
                                       void *workspace, int64_t workspace_bytes, cm3d_stream_t stream)
{
    if (!raw || !sweep_row_off || !sweep_xf || !frame_sweep_off || !points || !pt_off || !removed_cnt || !removed_idx) return CM3D_ERR_ARG;
    if (raw_stride < 4 || n_sweeps <= 0 || pt_cap <= 0 || ((uintptr_t)points & 15)) return CM3D_ERR_ARG;
    if (max_sweeps_per_frame <= 0 || max_sweeps_per_frame > PH_MAX_SWEEPS) return CM3D_ERR_ARG;
    PhSweepIn sw;
    sw.raw = raw; sw.raw_stride = raw_stride; sw.sweep_row_off = sweep_row_off; sw.sweep_xf = sweep_xf;
    sw.frame_sweep_off = frame_sweep_off; sw.n_frames = n_frames; sw.n_sweeps = n_sweeps; sw.halfw = halfw;
    sw.points_out = (float4 *)points; sw.pt_cap = pt_cap; sw.pt_off_out = pt_off; sw.removed_cnt = removed_cnt;
    sw.removed_idx = removed_idx;
    return ph_launch(&sw, points, pt_off, n_frames, max_pts_per_frame, n_points_total, cams, n_cams, mask_off, mask_cam, bbox, packed,
                     n_masks, W, H, min_dist, planes, hit_words, hit_count, status, workspace, workspace_bytes, stream);
}

extern "C" int cm3d_compact_hits(const uint32_t *hit_words, int32_t planes, const int32_t *pt_off, int32_t n_frames,
                                 int32_t max_pts_per_frame, int32_t n_points_total, const int32_t *mask_off, int32_t n_masks,
                                 const int32_t *hit_count, const int32_t *removed_cnt, const int32_t *removed_idx,
                                 int32_t *hit_off, int32_t *tile_off, int32_t *hit_idx, int32_t *hit_row, int32_t idx_cap,
                                 int32_t *tile_work, int32_t *status, void *workspace, int64_t workspace_bytes,
                                 cm3d_stream_t stream)
{
    if (!hit_words || !pt_off || !mask_off || !hit_count || !hit_off || !tile_off || !hit_idx || !hit_row || !status || !workspace)
        return CM3D_ERR_ARG;
    if ((removed_cnt == nullptr) != (removed_idx == nullptr)) return CM3D_ERR_ARG;
    if (planes <= 0 || n_frames <= 0 || max_pts_per_frame <= 0 || n_points_total <= 0 || n_masks <= 0 || idx_cap <= 0)
        return CM3D_ERR_ARG;
    if (workspace_bytes < cm3d_project_workspace_bytes(n_frames, max_pts_per_frame, planes)) return CM3D_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nblk_max = (max_pts_per_frame + PH_BLOCK_PTS - 1) / PH_BLOCK_PTS;
    const int nm_cap = ph_nm_cap(planes);
    hipLaunchKernelGGL(k_hit_offsets, dim3(n_frames), dim3(1024), 0, st, hit_count, n_masks, pt_off, mask_off, n_frames, nm_cap,
                       nblk_max, hit_off, tile_off, (int32_t *)workspace, idx_cap, status);
    CM3D_CHECK_LAUNCH();
    const int64_t tile_cap64 = md_tile_cap(n_masks, idx_cap);
    const int tile_cap = (int)(tile_cap64 > 0x7FFFFFFF ? 0x7FFFFFFF : tile_cap64);
    hipLaunchKernelGGL(k_compact_hits, dim3(nblk_max, n_frames + (tile_work ? 1 : 0)), dim3(PH_THREADS), 0, st, hit_words,
                       n_points_total, pt_off, mask_off, nm_cap, nblk_max, (const int32_t *)workspace, removed_cnt, removed_idx, hit_idx,
                       hit_row, idx_cap, n_frames, n_masks, hit_off, tile_off, tile_cap, (TileDesc *)tile_work);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}
