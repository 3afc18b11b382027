// a2 + a4-a8: sweep preparation, projection, in-image test, in-mask test, ordered compaction.
//   reference: src/nuscenes/2d_to_3d.py:437-465 (sweep loop) and :553-620 (per mask: clone the cloud, 2x
//   translate/rotate, view_points utils/pcd.py:262-284, 5-way in-image test, floor, mask gather with the
//   floor(u)!=0 && floor(v)!=0 quirk, torch.where, two .cpu() index-tracking steps).
// Every row is read ONCE, projected into every camera of its frame that can see it, and tested against the bit-packed
// eroded masks of that camera (bounding-box test first, so only points that can hit a mask touch mask memory).
// Layout of the work: a WAVE owns "wave-chunks" of 256 consecutive rows, lane l the 4 consecutive rows 4l..4l+3 -- so a
// lane's rows are one contiguous run of bytes (4 x raw_stride dwords: `raw_stride` 16-byte loads, nothing fetched twice)
// and its four hit words one 16-byte store.  Waves never talk to each other: per-frame tables (view wedges, masks
// sorted by camera with their bounding boxes, row ranges) are built once per frame by k_frame_tables, staged into LDS
// behind the only workgroup barrier, and every wave writes its own results:
//   hit_words  one 32-bit word per row per 32 masks
//   wc_cnt     one count per (wave-chunk, mask), and the same counts summed over groups of 16 wave-chunks (integer atomics:
//              order-free) -- k_compact_hits turns them into every wave-chunk's exact output offset itself and writes the
//              ascending index lists with ballot + mbcnt ranks (no atomics on the output order: deterministic)
//   wc_info    per wave-chunk: does it hold a hit at all (the compaction never touches the hit words of the others),
//              how many of its rows the reference drops
//   removed_bits  one bit per row the reference drops (ego box, :442-445)
// The transformed cloud is optional (`points` may be NULL): the medoid stage only needs the in-mask points, and
// k_compact_hits re-derives their coordinates from the raw rows (same fma chains, same bits) into hit_xyz.
// HBM bytes per row that MUST move: 4*raw_stride read + 4 per plane written (+16 when the cloud is kept).
#include "common.h"
#include "worklist.h"
#include <algorithm>
#include <vector>
#include <cstdlib>
#include <type_traits>

#ifdef CM3D_DIAG
// Diagnostic build only (make diag -> libcm3d_hip_diag.so; tools/ph_diag.py): ablation switches and per-phase
// s_memtime sums of k_project_hits.  Nothing of this exists in the product library.
__device__ int g_ph_diag;                         // bit1 no mask loop, bit2 no camera loop, bit3 synthetic rows, bit4 stamps
__device__ unsigned long long g_ph_stamp[8];
#define PH_DIAG_WAVES 16384
__device__ unsigned long long g_ph_wave[3 * PH_DIAG_WAVES];      // bit7: s_memtime at the start and the end of every wave, XCC_ID << 32 | HW_ID
__device__ unsigned long long g_ph_count[8];     // bit6: wave-chunks, (chunk, camera) pairs behind the wedge / pre-test / projection, mask batches, masks
static __device__ __forceinline__ unsigned long long ph_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define PH_DIAG(bit) (diag & (bit))
#define PH_COUNT(k, v) do { if (diag & 64) cntk[k] += (v); } while (0)
#define PH_STAMP(k)                                                     \
    do {                                                                \
        if (diag & 16) { const unsigned long long t_ = ph_now(); acc[k] += t_ - t_prev; t_prev = t_; } \
    } while (0)
extern "C" int cm3d_diag_set(int flags)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ph_diag), &flags, sizeof(int)) != hipSuccess) return CM3D_ERR_LAUNCH;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ph_stamp), z, sizeof(z)) != hipSuccess) return CM3D_ERR_LAUNCH;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ph_count), z, sizeof(z)) != hipSuccess) return CM3D_ERR_LAUNCH;
    void *wv = nullptr;
    if (hipGetSymbolAddress(&wv, HIP_SYMBOL(g_ph_wave)) != hipSuccess || hipMemset(wv, 0, sizeof(g_ph_wave)) != hipSuccess) return CM3D_ERR_LAUNCH;
    if (hipDeviceSynchronize() != hipSuccess) return CM3D_ERR_LAUNCH;
    return CM3D_OK;
}
extern "C" int cm3d_diag_read_waves(unsigned long long *out_host, int n_waves)
{
    if (n_waves > PH_DIAG_WAVES) return CM3D_ERR_ARG;
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_ph_wave), 3 * (size_t)n_waves * sizeof(unsigned long long)) == hipSuccess ? CM3D_OK : CM3D_ERR_LAUNCH;
}
extern "C" int cm3d_diag_read_counts(unsigned long long *out_host)
{
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_ph_count), 8 * sizeof(unsigned long long)) == hipSuccess ? CM3D_OK : CM3D_ERR_LAUNCH;
}
extern "C" int cm3d_diag_read(unsigned long long *out_host)
{
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_ph_stamp), 8 * sizeof(unsigned long long)) == hipSuccess ? CM3D_OK : CM3D_ERR_LAUNCH;
}
// k_project_q: stages of the chunk loop that run (project_q.h); CM3D_PQ_STAGE sets the start value (PMC passes per stage)
int g_pq_stage = getenv("CM3D_PQ_STAGE") ? atoi(getenv("CM3D_PQ_STAGE")) : 99;
extern "C" int cm3d_diag_pq_stage(int stage) { g_pq_stage = stage; return CM3D_OK; }
#else
#define PH_DIAG(bit) 0
#define PH_STAMP(k) do { } while (0)
#define PH_COUNT(k, v) do { } while (0)
#endif

#ifndef PH_LOADMODE
#define PH_LOADMODE 1          // how a full chunk's rows are fetched: 0 lane-strided non-temporal, 1 lane-strided (plain: measured faster)
#endif
#define PH_THREADS 256                            // k_compact_hits
#ifndef PHK_THREADS
#define PHK_THREADS 256                           // k_project_hits (its waves are independent: any multiple of 64)
#endif
#define PHK_WAVES (PHK_THREADS / 64)
#define PH_WAVES (PH_THREADS / 64)
#define PH_OVERSUB_NUM 1                          // waves launched : waves resident (k_project_hits header)
#define PH_OVERSUB_DEN 1
#ifndef PH_STEAL_LISTS
#define PH_STEAL_LISTS 6                          // chunk lists of other slots a wave tries after its own (k_project_hits header)
#endif
#ifndef PH_CG
#define PH_CG 2                                   // cameras whose wedge tests run side by side (CM3D_MAX_CAMS is a multiple)
#endif
#ifndef PH_MIN_BLOCKS
#define PH_MIN_BLOCKS (1024 / PHK_THREADS)        // workgroups per CU the register budget is held to: 4 waves per SIMD
#endif
#define PH_PT 4                                   // consecutive rows per lane
#define PH_WC (64 * PH_PT)                        // rows per wave-chunk
#define PH_MB 4                                   // masks of a camera handled together in the mask loop
#define PH_MAX_SWEEPS CM3D_MAX_FUSED_SWEEPS

// Frame table: one record of FT_WORDS 32-bit words per frame in the workspace, written by k_frame_tables.
//   [0] p0 first row  [1] n rows  [2] m0 first mask  [3] nm masks (clamped to the caller's planes)
//   [4] sa first sweep [5] ns sweeps  [6] first word of the frame's removed-row bits  [7] nwc wave-chunks
//   [8..16]  cam_first[0..8]: masks of camera c = entries [cam_first[c], cam_first[c+1]) of the frame's sorted list
//   [24..40] frame-local first row of each sweep (fused path; ns <= PH_MAX_SWEEPS)
//   [17] largest pixel margin of the approximate projections  [18] bit c = camera c has an approximate projection
//   [19] smallest depth the approximate projection may accept (float bits)  [20] bit c = camera c has a non-empty mask
//   [21] 1 = fused sweep preparation (the launch counts the dropped rows of every wave-chunk)
//   [64..127] view wedges, 8 floats per camera
//   [128..255] approximate projections, 16 floats per camera (wedge_setup)
#define FT_WORDS 256
#define FT_CAMFIRST 8
#define FT_MARGIN 17
#define FT_APXOK 18
#define FT_ZMIN 19
#define FT_CAMHAS 20
#define FT_FUSED 21                              // 1: the launch prepares the sweeps itself and counts the dropped rows (wc_info, group sums)
#define FT_SROW 24
#define FT_WEDGE 64
#define FT_APX 128

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));        // 16 bytes at dword alignment
typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(4)));

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

// Conservative tests of one camera, derived from its float32 record (composed in double, so that what is left is the
// rounding of the kernels' own float32 evaluation):
//  * view wedge: a point in front of the camera can only pass the exact in-image test (0 < u < W - 1) if it lies on the inner
//    side of the two planes through the camera centre and the image's left and right edges, u >= 0 and u <= W: in camera
//    coordinates fx X + cx Z >= 0 and (W - cx) Z - fx X >= 0, linear in the global point.  out: [0..2] unit normal of the
//    left plane (global), [3] its offset + margin, [4..7] the same for the right plane; a point is kept while
//    min(n_L . p + d_L, n_R . p + d_R) >= 0.  The margin (1e-6 max|o| + 1e-3 m) covers the float32 evaluation here and in
//    the exact chain at global magnitudes ten times over.  (The circular cone of earlier versions needed 13 packed operations
//    per point pair and camera and circumscribes the image; the wedge needs 7 and is 10 % narrower.)
//    *zmin_out: smallest axial distance the depth test can accept, minus a margin (the approximate projection's depth test).
//  * approximate projection (apx, 16 floats): camera centre o [0..2], composed rotation A [3..11] (p_cam = A (p - o)),
//    fx, fy, cx, cy of K' [12..15].  Evaluated in float32 as A (p - o) it differs from the exact chain's camera
//    coordinates by less than `delta` = 1e-6 max|o| + 1e-4 metres (both forms round p +- 1e3-magnitude translations once or
//    twice: ~6e-5 m at nuScenes' global magnitudes, so the bound has a factor of several to spare), i.e. by less than
//    *margin_px = 1 + ceil(2 max(fx, fy) delta / zmin) pixels for every point the depth test can accept.
// A camera record the derivation does not cover (skew, non-trivial last row of K, stages that are not rigid) disables both
// tests for that camera (the wedge accepts everything, *apx_ok = false).
static __device__ void wedge_setup(const float *cm_global, int W, int H, float min_dist, float *out, float *apx, int *margin_px, bool *apx_ok,
                                  float *zmin_out)
{
    // the whole record in registers first (14 independent 16-byte loads in flight at once; the record is 224 bytes and
    // 16-byte aligned), so that the composition below never waits on memory stage by stage
    float cm[CM3D_CAM_STRIDE];
#pragma unroll
    for (int q = 0; q < CM3D_CAM_STRIDE / 4; ++q) {
        const float4 v = reinterpret_cast<const float4 *>(cm_global)[q];
        cm[4 * q] = v.x; cm[4 * q + 1] = v.y; cm[4 * q + 2] = v.z; cm[4 * q + 3] = v.w;
    }
    const float *K = cm + 45;
    const int ns = (int)cm[54], fl = (int)cm[55];
    // compose the stages: p_cam = M p + c
    double M[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, c[3] = {0, 0, 0};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s >= ns) break;
        const float *st = cm + 15 * s, *Rm = st + 3;
        if (fl & (1 << (2 * s))) { c[0] += (double)st[0]; c[1] += (double)st[1]; c[2] += (double)st[2]; }
        double Mn[9], cn[3];
        for (int r = 0; r < 3; ++r) {
            for (int q = 0; q < 3; ++q) Mn[3 * r + q] = (double)Rm[3 * r] * M[q] + (double)Rm[3 * r + 1] * M[3 + q] + (double)Rm[3 * r + 2] * M[6 + q];
            cn[r] = (double)Rm[3 * r] * c[0] + (double)Rm[3 * r + 1] * c[1] + (double)Rm[3 * r + 2] * c[2];
        }
        for (int q = 0; q < 9; ++q) M[q] = Mn[q];
        for (int q = 0; q < 3; ++q) c[q] = cn[q];
        if (fl & (2 << (2 * s))) { c[0] += (double)st[12]; c[1] += (double)st[13]; c[2] += (double)st[14]; }
    }
    // camera centre: M o + c = 0  <=>  o = -M^T c  (M orthonormal)
    const double ox = -(M[0] * c[0] + M[3] * c[1] + M[6] * c[2]);
    const double oy = -(M[1] * c[0] + M[4] * c[1] + M[7] * c[2]);
    const double oz = -(M[2] * c[0] + M[5] * c[1] + M[8] * c[2]);
    apx[0] = (float)ox; apx[1] = (float)oy; apx[2] = (float)oz;
    for (int q = 0; q < 9; ++q) apx[3 + q] = (float)M[q];
    apx[12] = K[0]; apx[13] = K[4]; apx[14] = K[2]; apx[15] = K[5];
    *apx_ok = false;
    *margin_px = 0;
    const bool plain = K[1] == 0.f && K[3] == 0.f && K[6] == 0.f && K[7] == 0.f && K[8] == 1.f && K[0] > 0.f && K[4] > 0.f;
    // orthonormality of M (a rotation up to float32 rounding)?
    double dev = 0.0;
    for (int r = 0; r < 3; ++r)
        for (int q = r; q < 3; ++q) {
            const double d = M[3 * r] * M[3 * q] + M[3 * r + 1] * M[3 * q + 1] + M[3 * r + 2] * M[3 * q + 2] - (r == q ? 1.0 : 0.0);
            dev = fmax(dev, fabs(d));
        }
    const float omax = (float)fmax(fmax(fabs(ox), fabs(oy)), fabs(oz));
    const float zmin = min_dist - 0.05f - 1e-4f * omax;
    *zmin_out = zmin;
    for (int q = 0; q < 8; ++q) out[q] = (q & 3) == 3 ? 1.f : 0.f;                    // accept everything
    if (!plain || !(dev < 1e-3)) return;
    {
        const double fx = (double)K[0], cxp = (double)K[2], margin = 1e-6 * (double)omax + 1e-3;
        const double nc[2][3] = {{fx, 0.0, cxp}, {-fx, 0.0, (double)W - cxp}};         // left, right plane in camera coordinates
        for (int e = 0; e < 2; ++e) {
            const double len = sqrt(nc[e][0] * nc[e][0] + nc[e][2] * nc[e][2]);
            const double a = nc[e][0] / len, b = nc[e][2] / len;                         // (the y component is 0)
            out[4 * e + 0] = (float)(M[0] * a + M[6] * b);                               // M^T n
            out[4 * e + 1] = (float)(M[1] * a + M[7] * b);
            out[4 * e + 2] = (float)(M[2] * a + M[8] * b);
            out[4 * e + 3] = (float)(a * c[0] + b * c[2] + margin);
        }
    }
    (void)H;
    if (zmin > 0.1f && dev < 1e-5) {
        const float delta = 1e-6f * omax + 1e-4f;
        const float mg = 1.f + ceilf(2.f * fmaxf(K[0], K[4]) * delta / zmin);
        if (mg < 64.f) { *margin_px = (int)mg; *apx_ok = true; }
    }
}

struct PhSweepIn {
    const float *raw; const float *intensity; int raw_stride; const int32_t *sweep_row_off; const float *sweep_xf; const int32_t *frame_sweep_off;
    int n_frames, n_sweeps; float halfw; float4 *points_out; int pt_cap; int32_t *pt_off_out; uint32_t *removed_bits;
};

// workspace of the projection / compaction pair
//   wc_cnt   [F][nwc_max][nm_cap]  hits per (wave-chunk, mask)
//   wc_info  [F][nwc_max]          per wave-chunk: bit 31 = holds a hit, low bits = rows of the chunk the reference drops
//   grp      [F][zstride]          per frame, zeroed by k_frame_tables: [ngrp_max][nm_cap] hits per (group of PH_GRP wave-chunks, mask),
//                                  then [ngrp_max] dropped rows per group -- the second level of the compaction's prefix sums
//   frame_hits [F]                 in-mask points of the frame (all masks)
#define PH_GRP 16
struct PhWs { int4 *ment; int32_t *ft; int32_t *wc_info; int32_t *wc_cnt; int32_t *queue; int32_t *grp; int32_t *frame_hits; int zstride; };
static inline int ph_nm_cap(int planes)
{
    int c = planes * 32;
    return c > CM3D_MAX_MASKS_PER_FRAME ? CM3D_MAX_MASKS_PER_FRAME : c;
}
static inline int ph_tpf_max(int64_t nwc_max) { return (int)((nwc_max + 1) / 2 > 1 ? (nwc_max + 1) / 2 : 1); }     // >= 2 wave-chunks per ticket
static inline int64_t ph_ws_layout(int n_frames, int max_pts_per_frame, int planes, void *base, PhWs *out)
{
    const int64_t nwc_max = (max_pts_per_frame + PH_WC - 1) / PH_WC, nm_cap = ph_nm_cap(planes);
    const int64_t ngrp_max = (nwc_max + PH_GRP - 1) / PH_GRP;
    const int64_t zstride = (ngrp_max * (nm_cap + 1) + 3) & ~(int64_t)3;           // whole 16-byte pieces per frame
    int64_t off = 0;
    char *b = (char *)base;
    if (out) out->ment = (int4 *)(b + off);
    off += (int64_t)n_frames * nm_cap * 32;                       // two int4 per mask entry (k_frame_tables)
    if (out) out->ft = (int32_t *)(b + off);
    off += (int64_t)n_frames * FT_WORDS * 4;
    if (out) out->wc_info = (int32_t *)(b + off);
    off += ((int64_t)n_frames * nwc_max * 4 + 15) & ~(int64_t)15;
    if (out) out->wc_cnt = (int32_t *)(b + off);
    off += (int64_t)n_frames * nwc_max * nm_cap * 4;
    off = (off + 15) & ~(int64_t)15;
    if (out) out->queue = (int32_t *)(b + off);                   // k_project_hits: entries handed out per chunk list, [n_frames][tpf], tpf <= ph_tpf_max
    off += ((int64_t)n_frames * ph_tpf_max(nwc_max) * 4 + 15) & ~(int64_t)15;
    if (out) { out->grp = (int32_t *)(b + off); out->zstride = (int)zstride; }
    off += (int64_t)n_frames * zstride * 4;
    if (out) out->frame_hits = (int32_t *)(b + off);
    off += ((int64_t)n_frames * 4 + 15) & ~(int64_t)15;
    return off;
}

// One wave per frame: row / mask ranges, view wedges, and the frame's masks sorted by camera with their bounding
// boxes (empty masks and masks of an out-of-range camera are left out: they can get no point).  Also what the sweep
// kernel leaves behind for the later stages in the fused form (pt_off, status[1]).
__global__ __launch_bounds__(64) void k_frame_tables(const PhSweepIn sw, int fused, const int32_t *__restrict__ pt_off, int n_frames,
                                                     const float *__restrict__ cams, int n_cams,
                                                     const int32_t *__restrict__ mask_off, const int32_t *__restrict__ mask_cam,
                                                     const int4 *__restrict__ bbox, int W, int H, float min_dist, int nm_cap,
                                                     int max_pts_per_frame, uint32_t mask_words, int32_t *__restrict__ ft_all,
                                                     int4 *__restrict__ ment_all, int32_t *__restrict__ queue, int tpf,
                                                     int32_t *__restrict__ grp, int zstride, int32_t *__restrict__ frame_hits,
                                                     int32_t *__restrict__ status)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    int32_t *ft = ft_all + (size_t)f * FT_WORDS;
    for (int q = lane; q < tpf; q += 64) queue[(size_t)f * tpf + q] = 0;      // the frame's chunk lists: nothing handed out yet
    {                                                                           // the frame's group sums (project adds into them)
        uint4 *z = reinterpret_cast<uint4 *>(grp + (size_t)f * zstride);
        for (int q = lane; q < (zstride >> 2); q += 64) z[q] = make_uint4(0u, 0u, 0u, 0u);
        if (lane == 0) frame_hits[f] = 0;
    }

    int p0, n, sa = 0, ns = 0;
    if (fused) {
        sa = sw.frame_sweep_off[f];
        ns = sw.frame_sweep_off[f + 1] - sa;
        p0 = sw.sweep_row_off[sa];
        n = sw.sweep_row_off[sa + ns] - p0;
        if (sw.raw_stride == CM3D_RAW_QUADS && (p0 & 3) && lane == 0) atomicOr(&status[0], 8);      // quad layout: a frame starts on a quad
        if (lane == 0) {
            sw.pt_off_out[f] = p0;
            if (f == n_frames - 1) {
                const int total_rows = sw.sweep_row_off[sw.n_sweeps];
                sw.pt_off_out[n_frames] = total_rows;
                status[1] = total_rows;
                if (total_rows > sw.pt_cap) atomicOr(&status[0], 1);
                if (sw.frame_sweep_off[0] != 0 || sa + ns != sw.n_sweeps) atomicOr(&status[0], 4);      // sweeps outside every frame
            }
        }
        n = max(0, min(n, sw.pt_cap - p0));
        if (lane <= ns && lane <= PH_MAX_SWEEPS) ft[FT_SROW + lane] = sw.sweep_row_off[sa + lane] - p0;
    } else {
        p0 = pt_off[f];
        n = pt_off[f + 1] - p0;
    }
    if (n > max_pts_per_frame) {             // more rows than the workspace was sized for: stay inside it, report
        if (lane == 0) atomicOr(&status[0], 1);
        n = max_pts_per_frame;
    }
    n = max(n, 0);
    const int m0 = mask_off[f];
    int nm = mask_off[f + 1] - m0;
    if (nm > nm_cap) {                       // more masks than the caller's `planes` allows
        if (lane == 0) atomicOr(&status[0], 4);
        nm = nm_cap;
    }
    nm = max(nm, 0);
    if (lane == 0) {
        ft[0] = p0; ft[1] = n; ft[2] = m0; ft[3] = nm; ft[4] = sa; ft[5] = ns;
        ft[6] = (p0 >> 5) + 8 * f;          // frames never overlap: sum ceil(n_g / 32) <= (p0 >> 5) + f, and a chunk owns 8 whole words
        ft[7] = (n + PH_WC - 1) / PH_WC;
        ft[FT_FUSED] = fused ? 1 : 0;
    }
    int ft_margin = 0;
    {
        int mg = 0;
        bool ok = false;
        float zmin = INFINITY;
        if (lane < CM3D_MAX_CAMS) {
            float *wedge = reinterpret_cast<float *>(ft + FT_WEDGE) + 8 * lane, *apx = reinterpret_cast<float *>(ft + FT_APX) + 16 * lane;
            if (lane < n_cams) wedge_setup(cams + ((size_t)f * n_cams + lane) * CM3D_CAM_STRIDE, W, H, min_dist, wedge, apx, &mg, &ok, &zmin);
            else                             // a slot without a camera: a wedge nothing is inside of (the projection kernel tests whole groups)
                for (int q = 0; q < 8; ++q) wedge[q] = (q & 3) == 3 ? -1.f : 0.f;
        }
        const uint64_t okm = __ballot(ok);
        mg = cm3d_wave_max(mg);
        zmin = -cm3d_wave_max(-zmin);        // the frame's smallest (the cameras' differ by 1e-4 of their distance from the origin)
        ft_margin = mg;
        if (lane == 0) { ft[FT_MARGIN] = mg; ft[FT_APXOK] = (int)(uint32_t)okm; ft[FT_ZMIN] = __float_as_int(zmin); }
    }
    // the frame's masks sorted by camera: count per camera (lane c = masks of camera c), exclusive prefix, placement by
    // ballot rank.  Every mask is loaded once per pass (twice when the frame has more than 64 masks).
    int4 *ment = ment_all + (size_t)f * nm_cap * 2;
    const int mg = ft_margin;
    auto load_mask = [&](int k, int &cam, int4 &bb, int4 &rc) {      // bb: bounds of the eroded pixels; rc: the stored rectangle (xw0, y0, wc, rows)
        cam = -1;
        bb = make_int4(0, 0, -1, -1);
        rc = make_int4(0, 0, 0, 0);
        if (k < nm) {
            cam = mask_cam[m0 + k];
            bb = bbox[2 * (m0 + k)];
            rc = bbox[2 * (m0 + k) + 1];
            if (cam < 0 || cam >= n_cams) { atomicOr(&status[0], 4); cam = -1; }         // such a mask gets no points
            else if (!(bb.z >= bb.x && bb.w >= bb.y)) cam = -1;                          // empty after the erosion
        }
    };
    int cam0;
    int4 bb0, rc0;
    load_mask(lane, cam0, bb0, rc0);                 // the first 64 masks stay in registers for both passes
    int cnt = 0;                                     // lane c: masks of camera c
    for (int k0 = 0; k0 < nm; k0 += 64) {
        int cam = cam0;
        int4 bb = bb0, rc = rc0;
        if (k0) load_mask(k0 + lane, cam, bb, rc);
        for (int c = 0; c < n_cams; ++c) {
            const int x = (int)__popcll(__ballot(cam == c));
            cnt += lane == c ? x : 0;
        }
    }
    int first = cm3d_wave_incl_scan(lane < n_cams ? cnt : 0) - (lane < n_cams ? cnt : 0);       // lane c: first entry of camera c
    const int run = __builtin_amdgcn_readlane(first, CM3D_MAX_CAMS - 1) + __builtin_amdgcn_readlane(lane < n_cams ? cnt : 0, CM3D_MAX_CAMS - 1);
    if (lane < n_cams) ft[FT_CAMFIRST + lane] = first;
    {
        const uint64_t has = __ballot(lane < n_cams && cnt > 0);
        if (lane == 0) ft[FT_CAMHAS] = (int)(uint32_t)has;
    }
    for (int k0 = 0; k0 < nm; k0 += 64) {
        int cam = cam0;
        int4 bb = bb0, rc = rc0;
        if (k0) load_mask(k0 + lane, cam, bb, rc);
        const int k = k0 + lane;
        for (int c = 0; c < n_cams; ++c) {
            const uint64_t mk = __ballot(cam == c);
            const int base = __builtin_amdgcn_readlane(first, c);
            // entry, first half: corner and extent of the bounding box as 16-bit pairs (y in the high half, like the pixel
            // codes), mask number inside the frame | words per stored row << 16, word number of the mask's (0, 0) in `packed`;
            // second half: the box grown by the approximate projection's margin, in its pixel grid shifted by +1
            // ([x0 + 1 - mg, x1 + 1 + mg], low end clamped at 0), again as corner and extent
            if (cam == c) {
                const int e = base + cm3d_mbcnt(mk);
                const int lx = max(bb.x + 1 - mg, 0), ly = max(bb.y + 1 - mg, 0);
                const int hx = min(bb.z + 1 + mg, 32766), hy = min(bb.w + 1 + mg, 32766);        // (approximate codes stop at 32001; -1 stays outside)
                // word (xw, y) of the mask = slot + (y - y0) * wc + xw - xw0 = [slot - y0 * wc - xw0] + y * wc + xw: the bracket is the
                // entry's word number (signed: it may lie before the slot; a batch holds < 2^31 mask words), wc rides on the mask number
                ment[2 * e] = make_int4(bb.x | (bb.y << 16), (bb.z - bb.x) | ((bb.w - bb.y) << 16), k | (rc.z << 16),
                                        (int)((uint32_t)(m0 + k) * mask_words) - rc.y * rc.z - rc.x);
                ment[2 * e + 1] = make_int4(lx | (ly << 16), (hx - lx) | ((hy - ly) << 16), 0, 0);
            }
            first += lane == c ? (int)__popcll(mk) : 0;
        }
    }
    if (lane == 0)
        for (int c = n_cams; c <= CM3D_MAX_CAMS; ++c) ft[FT_CAMFIRST + c] = run;
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

typedef unsigned short us2 __attribute__((ext_vector_type(2)));       // pixel codes / box corners: v_pk_{sub,min}_u16

// The 4 rows of this lane: x, y, z [, w] of row j at v[j*S .. j*S+S-1].  Only what the pass uses is fetched: 12 bytes per
// row, 16 when the transformed cloud is kept (KEEP: its fourth column is the row's).  The registers a load writes are
// held from the request to the use -- a whole camera loop for the rows requested one chunk ahead -- so columns nobody
// reads would cost registers, not only bytes: a 5-column row fetched whole held 20 VGPRs per lane, this form 12.
// The lane's rows are consecutive in memory, so the four requests of a lane touch the same cache lines as one long one.
// STRIDE > 0: compile-time row stride (floats); 0: `stride` at run time.  Slots past the end of the frame read as
// (1e30, 1e30, 0, 0): never dropped, and turned into NaN points by the caller.
typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));        // 12 bytes at dword alignment
template <bool KEEP>
struct PhRows { static constexpr int S = KEEP ? 4 : 3; float v[4 * S]; };

template <int STRIDE, bool KEEP>
static __device__ __forceinline__ void ph_load_rows(PhRows<KEEP> &r, const float *__restrict__ src, const float *__restrict__ aux, int stride,
                                                    size_t row0, int nvalid, int lane, int diag = 0)
{
    constexpr int S = PhRows<KEEP>::S;
    if (PH_DIAG(8)) {
#pragma unroll
        for (int j = 0; j < PH_PT; ++j) {
            const int i = (int)row0 + 4 * lane + j;
            r.v[j * S] = (float)(i & 1023) * 0.05f - 20.f; r.v[j * S + 1] = (float)((i >> 10) & 63) * 0.5f - 8.f;
            r.v[j * S + 2] = -1.f;
            if (KEEP) r.v[j * S + 3] = 0.f;
        }
        return;
    }
    if (STRIDE == CM3D_RAW_QUADS) {
        // quad layout (cm3d_hip.h): rows 4q..4q+3 of the batch are 12 floats x0..3 y0..3 z0..3 -- the lane's four rows are one
        // 48-byte piece, three 16-byte loads; 12 bytes per row cross HBM and nothing else.  The fourth column, when the cloud
        // is kept, comes from the intensity plane (`aux`, may be NULL: zeros).  Every frame starts on a quad (k_frame_tables
        // checks), and a frame's last quad exists whole (its spare rows belong to the frame: NaN rows, see the header).
        float q[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) q[k] = (k < 8) ? 1e30f : 0.f;
        float w4[4] = {0.f, 0.f, 0.f, 0.f};
        if (4 * lane < nvalid) {
            const float4 *p = reinterpret_cast<const float4 *>(src + (row0 + (size_t)(4 * lane)) * 3);
            const float4 a = p[0], b = p[1], c = p[2];
            q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
            q[8] = c.x; q[9] = c.y; q[10] = c.z; q[11] = c.w;
            if (KEEP && aux) {
                const float4 t = *reinterpret_cast<const float4 *>(aux + row0 + (size_t)(4 * lane));
                w4[0] = t.x; w4[1] = t.y; w4[2] = t.z; w4[3] = t.w;
            }
        }
#pragma unroll
        for (int j = 0; j < PH_PT; ++j) {
            r.v[j * S] = q[j]; r.v[j * S + 1] = q[4 + j]; r.v[j * S + 2] = q[8 + j];
            if (KEEP) r.v[j * S + 3] = w4[j];
        }
        return;
    }
    const int rs = STRIDE > 0 ? STRIDE : stride;
    if (nvalid >= PH_WC) {
        const float *p = src + (row0 + (size_t)(4 * lane)) * rs;
#pragma unroll
        for (int j = 0; j < PH_PT; ++j) {
            if (KEEP) {
                const f4u t = *reinterpret_cast<const f4u *>(p + j * rs);
                r.v[j * S] = t.x; r.v[j * S + 1] = t.y; r.v[j * S + 2] = t.z; r.v[j * S + 3] = t.w;
            } else {
                const f3u t = *reinterpret_cast<const f3u *>(p + j * rs);
                r.v[j * S] = t.x; r.v[j * S + 1] = t.y; r.v[j * S + 2] = t.z;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < PH_PT; ++j) {
            float x = 1e30f, y = 1e30f, z = 0.f, w = 0.f;
            if (4 * lane + j < nvalid) {
                const float *p = src + (row0 + (size_t)(4 * lane + j)) * rs;
                x = p[0]; y = p[1]; z = p[2];
                if (KEEP) w = p[3];
            }
            r.v[j * S] = x; r.v[j * S + 1] = y; r.v[j * S + 2] = z;
            if (KEEP) r.v[j * S + 3] = w;
        }
    }
}

// sweep of frame-local row i (ns > 1): srow = the frame table's sweep starts
static __device__ __forceinline__ int ph_sweep_of(const int32_t *srow, int ns, int i)
{
    int k_sw = 0;
    for (int k = 1; k < ns; ++k) k_sw += srow[k] <= i;
    return k_sw;
}
// The same from a register: lane k of v_srow = the frame-local first row of sweep k (k < ns <= PH_MAX_SWEEPS), lanes from ns on INT_MAX.
// One compare and a bit count for a uniform row -- the loop above is a memory round trip per sweep, and in a kernel that stores to
// global memory the compiler makes them VECTOR loads with a wait for everything in flight (the prefetched rows), twice per chunk.
static __device__ __forceinline__ int ph_sweep_of_u(int v_srow, int i_uniform)
{
    return (int)__popcll(__ballot(cm3d_lane() >= 1 && v_srow <= i_uniform));
}

// One-dimensional grid of 4-wave workgroups whose waves never synchronise.  Wave number t of the launch holds TICKET t of
// `tpf` tickets per frame: slot = t / F of frame (t % F + W slot) % F, W = waves per workgroup -- the four waves of a workgroup work on four
// different frames, and the tickets of one frame sit in tpf different workgroups on all XCDs.  Ticket (f, slot) OWNS the
// wave-chunks slot, slot + tpf, slot + 2 tpf, ... of frame f (what a chunk costs depends on where its rows point, and
// neighbouring chunks cost alike: every tpf-th one is a fair sample), and every list has its own counter in memory
// (`taken`): whoever wants the next chunk of a list draws it with one atomic add.  A wave walks its own list and, when that
// is used up, the lists of the next slots of the SAME frame (its tables are already staged), at most PH_STEAL_LISTS of
// them.  Why: workgroups do not land evenly on the CUs (one launch of 768 workgroups put between two and four on a CU),
// co-running kernels of the other batches in flight take CUs too, and with fixed shares the mean wave lived 72 % as long
// as the longest.  Why per-list counters: returning device-scope atomics on ONE address are served one at a time at ~0.6 us
// each (150 draws per frame counter doubled the launch); a list's counter sees its owner's ~9 draws and the odd thief.
// Every counter only grows and every wave tries a bounded number of lists, so every wave reaches its exit; a workgroup that
// starts late (more workgroups than fit at once) finds its lists used up and leaves.
// Per wave-chunk:
//   raw rows (requested one chunk ahead) -> ego-box drop + sensor -> ego -> global (FUSED) -> [cloud store] -> removed bits
//   for every camera that can see any of the wave's points (wedge test): project the 4 rows of every lane (pixel codes stay
//   in registers), then for every mask of that camera: bounding-box test, one mask word per candidate point, bit test,
//   hit count  -> hit words, per-(chunk, mask) counts.
// ONE_PLANE (<= 32 masks per frame): the hit word of a row and the count of a mask live in registers (count of mask k in
// lane k); otherwise in the wave's own LDS slice.
// FUSED: `src` = raw sweep rows (a2, reference :437-465); otherwise `src` = a prepared float4 cloud (STRIDE 4) whose
// dropped rows are NaN points.
template <bool ONE_PLANE, bool FUSED, int STRIDE, bool KEEP>
__global__ __launch_bounds__(PHK_THREADS, PH_MIN_BLOCKS) void k_project_hits(
    const float *__restrict__ src, const float *__restrict__ src_aux, int src_stride, const float *__restrict__ sweep_xf, float halfw,
    float4 *__restrict__ points_out,
    uint32_t *__restrict__ removed_bits, const int32_t *__restrict__ ft_all, const int4 *__restrict__ ment_all,
    const float *__restrict__ cams, int n_cams, const uint32_t *__restrict__ packed, int W, int H, int Wp, float min_dist, int nm_cap,
    int nwc_max, int n_points_total, uint32_t *__restrict__ hit_words, int32_t *__restrict__ hit_count, int32_t *__restrict__ wc_cnt,
    int n_frames, int tpf, int32_t *__restrict__ queue, int32_t *__restrict__ wc_info, int32_t *__restrict__ grp, int zstride,
    int32_t *__restrict__ frame_hits)
{
#ifdef CM3D_DIAG
    const int diag = g_ph_diag;
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
    int cntk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (diag & 16) t_prev = ph_now();
    const unsigned long long t_start = (diag & 128) ? ph_now() : 0ull;
#endif
    const int lane = cm3d_lane(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int planes_cap = (nm_cap + 31) >> 5;
    // per-wave copies of the frame's tables (nothing here is shared between waves, so nothing needs a workgroup barrier)
    __shared__ __align__(16) float s_cam_all[PHK_WAVES][CM3D_MAX_CAMS * CM3D_CAM_STRIDE];
    __shared__ __align__(16) float s_tab_all[PHK_WAVES][FT_WORDS - FT_WEDGE];       // wedges (8 floats per camera), approximate projections (16)
    __shared__ int s_first_all[PHK_WAVES][CM3D_MAX_CAMS + 1];
    float *const s_cam = s_cam_all[wave];
    float(*const s_wedge)[8] = reinterpret_cast<float(*)[8]>(s_tab_all[wave]);
    float(*const s_apx)[16] = reinterpret_cast<float(*)[16]>(s_tab_all[wave] + (FT_APX - FT_WEDGE));
    int *const s_first = s_first_all[wave];
    // dynamic LDS (several planes only), one slice per wave: hit words [planes_cap][PH_WC], counts [nm_cap]
    extern __shared__ __align__(16) unsigned char s_dyn[];
    uint32_t *s_bits = reinterpret_cast<uint32_t *>(s_dyn) + (size_t)wave * planes_cap * PH_WC;
    int *s_cnt = reinterpret_cast<int *>(reinterpret_cast<uint32_t *>(s_dyn) + (size_t)PHK_WAVES * planes_cap * PH_WC) + wave * nm_cap;
    const float qnan = __int_as_float(0x7FC00000);
    constexpr int S = PhRows<KEEP>::S;

    const int ticket = (int)blockIdx.x * PHK_WAVES + wave;
    if (ticket >= n_frames * tpf) return;                           // uniform (the last workgroup's spare waves)
    const int slot = ticket / n_frames, f = (ticket - slot * n_frames + PHK_WAVES * slot) % n_frames;
    const int32_t *ft = ft_all + (size_t)f * FT_WORDS;              // uniform: scalar loads
    // the chunk lists of the frame: taken[s] = entries of list s handed out so far (zeroed by k_frame_tables)
    int32_t *const taken = queue + (size_t)f * tpf;
    int list = slot, lists_left = PH_STEAL_LISTS;
#ifdef CM3D_DIAG
    int static_next = 0;                                            // diag bit 512: fixed shares, no draws (timing only)
#endif
    auto draw = [&](int l) {                                        // request the next entry of list l; the answer is read later
        int v = 0;
#ifdef CM3D_DIAG
        if (diag & 512) return l == slot ? static_next++ : (1 << 20);
#endif
        if (lane == 0) v = atomicAdd(&taken[l], 1);
        return v;
    };
    // Start: everything that only needs the frame's number goes out at once -- the first two entries of the own list, the
    // camera records and tables for this wave's LDS slice -- and then, behind the frame record, the rows of entry 0 (what the
    // first draw returns unless a thief was faster) without waiting for the draw: two memory round trips instead of five
    // in a row.
    int draw_v = draw(slot);
    const int draw2_v = draw(slot);
    const float4 *cg = reinterpret_cast<const float4 *>(cams + (size_t)f * n_cams * CM3D_CAM_STRIDE);
    constexpr int CAM_Q = CM3D_MAX_CAMS * (CM3D_CAM_STRIDE / 4) / 64;               // 16-byte pieces of the camera records per lane
    float4 t_cam[CAM_Q];
#pragma unroll
    for (int q = 0; q < CAM_Q; ++q) t_cam[q] = lane + 64 * q < n_cams * (CM3D_CAM_STRIDE / 4) ? cg[lane + 64 * q] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 t_tab = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < (FT_WORDS - FT_WEDGE) / 4) t_tab = reinterpret_cast<const float4 *>(ft + FT_WEDGE)[lane];
    const int t_first = lane <= CM3D_MAX_CAMS ? ft[FT_CAMFIRST + lane] : 0;
    const int v_srow = lane < min(ft[5], PH_MAX_SWEEPS + 1) ? ft[FT_SROW + lane] : 0x7FFFFFFF;       // (ph_sweep_of_u)
    const int p0 = ft[0], n = ft[1], m0 = ft[2], nm = ft[3], sa = ft[4], ns = ft[5], bits_off = ft[6], nwc = ft[7];
    if (slot >= nwc) return;                                        // more tickets than wave-chunks in this frame
    const int planes = (nm + 31) >> 5;
    // the chunk a draw stands for, or, when that list is used up, the first chunk of the next lists that still have one
    // (these draws are waited for: once per list a wave goes through); >= nwc: nothing left for this wave
    auto chunk_of = [&](int drawn_v, int from) {                    // from = the list the draw was made on
        int c = from + __builtin_amdgcn_readfirstlane(drawn_v) * tpf;
        while (c >= nwc && lists_left > 0) {
            if (from == list) {                                     // (else: an older draw on a list this wave has left since)
                --lists_left;
                list = list + 1 == tpf ? 0 : list + 1;
            }
            from = list;
            c = from + __builtin_amdgcn_readfirstlane(draw(from)) * tpf;
        }
        return c;
    };
    PhRows<KEEP> cur;
    ph_load_rows<STRIDE, KEEP>(cur, src, src_aux, src_stride, (size_t)p0 + (size_t)slot * PH_WC, min(PH_WC, n - slot * PH_WC), lane, PH_DIAG(8));
    {
#pragma unroll
        for (int q = 0; q < CAM_Q; ++q) reinterpret_cast<float4 *>(s_cam)[lane + 64 * q] = t_cam[q];
        if (lane < (FT_WORDS - FT_WEDGE) / 4) reinterpret_cast<float4 *>(s_tab_all[wave])[lane] = t_tab;
        if (lane <= CM3D_MAX_CAMS) s_first[lane] = t_first;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const int apx_okmask = ft[FT_APXOK];
    const uint32_t cam_has = (uint32_t)ft[FT_CAMHAS];                // bit c: camera c has a non-empty mask in this frame
    int chunk = chunk_of(draw_v, slot);
    if (chunk >= nwc) return;                                       // a late start: the others have been through this frame's lists
    if (chunk != slot)                                              // entry 0 was gone
        ph_load_rows<STRIDE, KEEP>(cur, src, src_aux, src_stride, (size_t)p0 + (size_t)chunk * PH_WC, min(PH_WC, n - chunk * PH_WC), lane, PH_DIAG(8));
    int c_nxt = chunk_of(draw2_v, slot);
    PH_STAMP(0);                                                    // frame setup

    const int4 *ment = ment_all + (size_t)f * nm_cap * 2;           // two int4 per entry
    int acc_cnt = 0;                                                // ONE_PLANE: lane k = hits of mask k over this wave's chunks
    int32_t *const wc_cnt_f = wc_cnt + (size_t)f * nwc_max * nm_cap;  // the frame's count rows
    int32_t *const wc_info_f = wc_info + (size_t)f * nwc_max;
    int32_t *const grp_f = grp + (size_t)f * zstride;               // [ngrp_max][nm_cap] hits, then [ngrp_max] dropped rows
    const int ngrp_max = (nwc_max + PH_GRP - 1) / PH_GRP;
    uint32_t pend_bits[PH_PT] = {0u, 0u, 0u, 0u};                   // results of the previous chunk, not stored yet
    int pend_cnt = 0, pend_chunk = -1, pend_drop = 0;
    int acc_multi = 0;                                              // !ONE_PLANE: hits this lane has reported (all masks it handled)
    // results of a chunk: hit words (16 bytes per lane and plane), per-mask counts
    auto flush_results = [&]() {
        if (pend_chunk < 0) return;                                 // uniform
        const int pcb = pend_chunk * PH_WC, pvalid = min(PH_WC, n - pcb);
        int32_t *cnt_row = wc_cnt_f + pend_chunk * nm_cap;
        bool any = false, mine = false;
        if (ONE_PLANE) {
            any = __ballot(pend_cnt != 0) != 0ull;
            // the hit words of a wave-chunk WITHOUT a hit (two thirds of them on the headline shape) are never read -- the
            // compaction looks at wc_info first -- and are not written: 16 bytes per lane of HBM traffic less
            if (any) {
                uint32_t *hw = hit_words + (size_t)p0 + pcb + 4 * lane;
                if (pvalid >= PH_WC) {
                    *reinterpret_cast<u4u *>(hw) = (u4u){pend_bits[0], pend_bits[1], pend_bits[2], pend_bits[3]};
                } else {
#pragma unroll
                    for (int j = 0; j < PH_PT; ++j)
                        if (4 * lane + j < pvalid) hw[j] = pend_bits[j];
                }
            }
            if (lane < 32) {
                cnt_row[lane] = pend_cnt;
                if (pend_cnt && !PH_DIAG(256)) atomicAdd(&grp_f[(pend_chunk / PH_GRP) * nm_cap + lane], pend_cnt);
            }
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int k = lane; k < nm; k += 64) {
                const int cv = s_cnt[k];
                cnt_row[k] = cv;
                if (cv) {
                    atomicAdd(&hit_count[m0 + k], cv);
                    atomicAdd(&grp_f[(pend_chunk / PH_GRP) * nm_cap + k], cv);
                    acc_multi += cv;
                    mine = true;
                }
            }
            any = __ballot(mine) != 0ull;
            for (int pl = 0; any && pl < planes; ++pl) {              // (a wave-chunk without a hit: see above)
                const uint4 w4 = *reinterpret_cast<const uint4 *>(&s_bits[pl * PH_WC + 4 * lane]);
                uint32_t *hw = hit_words + (size_t)pl * n_points_total + p0 + pcb + 4 * lane;
                if (pvalid >= PH_WC) {
                    *reinterpret_cast<u4u *>(hw) = (u4u){w4.x, w4.y, w4.z, w4.w};
                } else {
                    const uint32_t wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int j = 0; j < PH_PT; ++j)
                        if (4 * lane + j < pvalid) hw[j] = wv[j];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // what the compaction wants to know about the chunk before it touches anything else of it
        if (lane == 0) {
            wc_info_f[pend_chunk] = pend_drop | (any ? (int)0x80000000 : 0);
            if (pend_drop && !PH_DIAG(256)) atomicAdd(&grp_f[ngrp_max * nm_cap + pend_chunk / PH_GRP], pend_drop);
        }
    };
    int draw_from = list;
#pragma unroll 1
    do {
        const int cb = chunk * PH_WC;
        const int nvalid = min(PH_WC, n - cb);                      // uniform
        PH_COUNT(0, 1);
        f2 X[PH_NP], Y[PH_NP], Z[PH_NP];                           // rows (2h, 2h+1) of this lane side by side
        int drop_now = 0;                                           // rows of this chunk the reference drops (uniform)
        if (FUSED) {
            // sweep of the chunk's first and last row: equal for all but the chunks that hold a sweep boundary
            const int32_t *srow = ft + FT_SROW;
            int sw_lo = 0, sw_hi = 0;
            if (ns > 1) { sw_lo = ph_sweep_of_u(v_srow, cb); sw_hi = ph_sweep_of_u(v_srow, cb + nvalid - 1); }
            const float *xf_u = sweep_xf + (size_t)(sa + sw_lo) * CM3D_SWEEP_XF_STRIDE;                    // scalar loads
            uint32_t nib = 0;
            // (scalar float32 per row, the 24 coefficients as scalar operands.  Measured alternatives: the packed form -- half the
            // instructions, but the rows have to be re-paired first and every coefficient becomes a 64-bit scalar operand:
            // 62 -> 67 us; the rotations as v_mfma_f32_4x4x1_f32 -- one fmaf per element, bit-identical, the point stays in its
            // lane: 62 -> 70 us, 24 dependent matrix instructions per chunk with their issue gaps cost more than the 72 vector
            // instructions they replace)
            const bool uni = sw_lo >= sw_hi;
#pragma unroll
            for (int j = 0; j < PH_PT; ++j) {
                float bx, by, bz;
                if (uni) ph_xform(xf_u, cur.v[j * S], cur.v[j * S + 1], cur.v[j * S + 2], bx, by, bz);
                else ph_xform(sweep_xf + (size_t)(sa + ph_sweep_of(srow, ns, cb + 4 * lane + j)) * CM3D_SWEEP_XF_STRIDE, cur.v[j * S],
                              cur.v[j * S + 1], cur.v[j * S + 2], bx, by, bz);
                X[j >> 1][j & 1] = bx; Y[j >> 1][j & 1] = by; Z[j >> 1][j & 1] = bz;
            }
#pragma unroll
            for (int j = 0; j < PH_PT; ++j) {
                const bool live = 4 * lane + j < nvalid;
                const bool drop = live && fabsf(cur.v[j * S]) < halfw && fabsf(cur.v[j * S + 1]) < halfw;      // reference drops this row (2d_to_3d.py:442-445)
                if (!live || drop) { X[j >> 1][j & 1] = qnan; Y[j >> 1][j & 1] = qnan; Z[j >> 1][j & 1] = qnan; }
                nib |= (drop ? 1u : 0u) << j;
                if (KEEP && live)
                    points_out[(size_t)p0 + cb + 4 * lane + j] = make_float4(X[j >> 1][j & 1], Y[j >> 1][j & 1], Z[j >> 1][j & 1], cur.v[j * S + (KEEP ? 3 : 0)]);
            }
            drop_now = 0;
            if (__ballot(nib != 0u)) {
                // (r04: ballots and data-parallel-primitive moves instead of nine dependent ds_bpermute round trips, see project_q.h)
#pragma unroll
                for (int j = 0; j < PH_PT; ++j) drop_now += (int)__popcll(__ballot((nib >> j) & 1u));
                // this chunk's 8 words of the frame's removed-row bits (zeroed by cm3d_batch_begin): lane l holds bits 4(l&7)..+3 of word l>>3
                int vv = (int)(nib << (4 * (lane & 7)));
                vv |= __builtin_amdgcn_update_dpp(0, vv, 0xB1, 0xF, 0xF, true);
                vv |= __builtin_amdgcn_update_dpp(0, vv, 0x4E, 0xF, 0xF, true);
                vv |= __builtin_amdgcn_update_dpp(0, vv, 0x104, 0xF, 0xF, true);
                if ((lane & 7) == 0) removed_bits[(size_t)bits_off + 8 * chunk + (lane >> 3)] = (uint32_t)vv;
            }
        } else {
#pragma unroll
            for (int j = 0; j < PH_PT; ++j) {
                const bool live = 4 * lane + j < nvalid;
                X[j >> 1][j & 1] = live ? cur.v[j * S] : qnan; Y[j >> 1][j & 1] = live ? cur.v[j * S + 1] : qnan;
                Z[j >> 1][j & 1] = live ? cur.v[j * S + 2] : qnan;
            }
        }
        PH_STAMP(1);                                                // rows arrive, transform, [cloud store]
        // vmcnt counts loads and stores together and in order, and the waits the compiler places are "everything so far":
        // the one memory wait of an iteration that can be long is the one above, for rows requested a whole camera loop
        // ago.  Right behind it go the previous chunk's result stores and the request for the next chunk's rows; both
        // have the camera loop to complete before anything waits again.
        flush_results();
        PhRows<KEEP> nxt;
        if (c_nxt < nwc) {
            ph_load_rows<STRIDE, KEEP>(nxt, src, src_aux, src_stride, (size_t)p0 + (size_t)c_nxt * PH_WC, min(PH_WC, n - c_nxt * PH_WC), lane, PH_DIAG(8));
            draw_from = list;
            draw_v = draw(list);                                    // the chunk after that one: answered during the camera loop
        }
        uint32_t bits[PH_PT];
#pragma unroll
        for (int j = 0; j < PH_PT; ++j) bits[j] = 0;
        int mycnt = 0;                                              // ONE_PLANE: lane k = hits of mask k in this chunk
        if (!ONE_PLANE) {
            for (int pl = 0; pl < planes; ++pl) *reinterpret_cast<uint4 *>(&s_bits[pl * PH_WC + 4 * lane]) = make_uint4(0u, 0u, 0u, 0u);
            for (int k = lane; k < nm; k += 64) s_cnt[k] = 0;
        }
        // conservative pre-test (a superset of the exact in-image test): is any of this wave's points inside the camera's
        // view wedge (wedge_setup)?  A wave's 256 rows are consecutive in the sweep, i.e. a short arc of the scan, and most
        // cameras are rejected here for the whole wave.  NaN points compare false.
        // PH_CG cameras at a time in straight-line code: their table reads go out together and their arithmetic interleaves
        // (one camera alone is a dependent chain behind an LDS round trip).  Slots past n_cams hold a wedge nothing is inside
        // of (k_frame_tables).
        uint32_t vis = 0u;
#pragma unroll 1
        for (int cg = 0; cg < (PH_DIAG(4) ? 0 : n_cams); cg += PH_CG) {
            if (!((cam_has >> cg) & ((1u << PH_CG) - 1u))) continue;                    // no camera of the group has anything to hit
            float inside[PH_CG];
#pragma unroll
            for (int q = 0; q < PH_CG; ++q) {
                const float *cn = s_wedge[cg + q];
                f2 m[PH_NP];
#pragma unroll
                for (int h = 0; h < PH_NP; ++h) {
                    f2 l = PK_FMA((f2)(cn[2]), Z[h], (f2)(cn[3])); l = PK_FMA((f2)(cn[1]), Y[h], l); l = PK_FMA((f2)(cn[0]), X[h], l);
                    f2 r = PK_FMA((f2)(cn[6]), Z[h], (f2)(cn[7])); r = PK_FMA((f2)(cn[5]), Y[h], r); r = PK_FMA((f2)(cn[4]), X[h], r);
                    m[h] = __builtin_elementwise_min(l, r);
                }
                float in = fmaxf(m[0].x, m[0].y);
#pragma unroll
                for (int h = 1; h < PH_NP; ++h) in = fmaxf(in, fmaxf(m[h].x, m[h].y));
                inside[q] = in;
            }
#pragma unroll
            for (int q = 0; q < PH_CG; ++q)
                if (__ballot(inside[q] >= 0.0f)) vis |= 1u << (cg + q);
        }
        PH_STAMP(2);                                                // wedge tests
#pragma unroll 1
        while (vis) {
            const int c = __builtin_ctz(vis);
            vis &= vis - 1u;
            const int e0 = __builtin_amdgcn_readfirstlane(s_first[c]), e1 = __builtin_amdgcn_readfirstlane(s_first[c + 1]);
            if (e0 >= e1) continue;                                 // a camera without (non-empty) masks: nothing to hit
            PH_COUNT(1, 1);
            // Pre-test on the approximate projection (wedge_setup): which masks of this camera can any of the wave's points
            // hit?  A point in a mask lies in the mask's bounding box; its approximate pixel is within apx_margin pixels of the
            // exact one, so it lies in the box grown by that margin.  Two thirds of the (wave, camera) pairs that reach this
            // point end here, and the exact projection below then only meets the masks that have a candidate.
            const int ne = e1 - e0;
            const bool pretest = ne <= 32 && ((apx_okmask >> c) & 1) && !PH_DIAG(32);
            uint32_t cmask = ne >= 32 ? 0xFFFFFFFFu : ((1u << ne) - 1u);          // bit i = entry e0 + i is a candidate
            if (pretest) {
                const float *ap = s_apx[c];
                const float zmin = __int_as_float(ft[FT_ZMIN]);
                int pa[PH_PT];
#pragma unroll
                for (int h = 0; h < PH_NP; ++h) {
                    const f2 vx = X[h] - ap[0], vy = Y[h] - ap[1], vz = Z[h] - ap[2];
                    f2 xc = ap[3] * vx; xc = PK_FMA((f2)(ap[4]), vy, xc); xc = PK_FMA((f2)(ap[5]), vz, xc);
                    f2 yc = ap[6] * vx; yc = PK_FMA((f2)(ap[7]), vy, yc); yc = PK_FMA((f2)(ap[8]), vz, yc);
                    f2 zc = ap[9] * vx; zc = PK_FMA((f2)(ap[10]), vy, zc); zc = PK_FMA((f2)(ap[11]), vz, zc);
                    const f2 r = {__builtin_amdgcn_rcpf(zc.x), __builtin_amdgcn_rcpf(zc.y)};
                    // pixel grid shifted by +1 (folded into the principal point) and clamped to [0, 32001], so that both halves are
                    // unsigned 16-bit values; points behind zmin (NaN ones with them) become 0xFFFF, 0xFFFF: outside every box
                    const f2 ua = PK_FMA(ap[12] * xc, r, (f2)(ap[14] + 1.f)), va = PK_FMA(ap[13] * yc, r, (f2)(ap[15] + 1.f));
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int iu = (int)__builtin_amdgcn_fmed3f(ua[q], 0.f, 32001.f), iv = (int)__builtin_amdgcn_fmed3f(va[q], 0.f, 32001.f);
                        pa[2 * h + q] = zc[q] > zmin ? ((iv << 16) | iu) : -1;
                    }
                }
                cmask = 0u;
                for (int e = e0; e < e1; e += PH_MB) {
                    int2 en[PH_MB];                                 // the grown boxes (second half of the entries)
#pragma unroll
                    for (int b = 0; b < PH_MB; ++b) en[b] = *reinterpret_cast<const int2 *>(&ment[2 * min(e + b, e1 - 1) + 1]);        // uniform: scalar loads
#pragma unroll
                    for (int b = 0; b < PH_MB; ++b) {
                        if (e + b >= e1) continue;
                        const us2 org = __builtin_bit_cast(us2, __builtin_amdgcn_readfirstlane(en[b].x));
                        const us2 ext = __builtin_bit_cast(us2, __builtin_amdgcn_readfirstlane(en[b].y));
                        bool any = false;
#pragma unroll
                        for (int j = 0; j < PH_PT; ++j) {
                            const us2 d = __builtin_bit_cast(us2, pa[j]) - org;
                            const us2 m = __builtin_elementwise_min(d, ext);
                            any |= __builtin_bit_cast(uint32_t, m) == __builtin_bit_cast(uint32_t, d);
                        }
                        if (__ballot(any)) cmask |= 1u << (e + b - e0);
                    }
                }
                PH_STAMP(3);
                if (!cmask) continue;                               // no point of the wave near any mask of this camera
            }
            PH_COUNT(2, 1);
            const int cns = __builtin_amdgcn_readfirstlane((int)s_cam[c * CM3D_CAM_STRIDE + 54]);
            const int cfl = __builtin_amdgcn_readfirstlane((int)s_cam[c * CM3D_CAM_STRIDE + 55]);
            const float *cm = s_cam + c * CM3D_CAM_STRIDE;
            int px[PH_PT];
            if (cns == 2 && cfl == 5) project_quad<2, 5>(cm, cns, cfl, X, Y, Z, min_dist, W, H, px);          // nuScenes
            else if (cns == 1 && cfl == 1) project_quad<1, 1>(cm, cns, cfl, X, Y, Z, min_dist, W, H, px);     // Waymo
            else if (cns == 3 && cfl == 10) project_quad<3, 10>(cm, cns, cfl, X, Y, Z, min_dist, W, H, px);   // KITTI
            else project_quad<-1, 0>(cm, cns, cfl, X, Y, Z, min_dist, W, H, px);
            int pxall = px[0];
#pragma unroll
            for (int j = 1; j < PH_PT; ++j) pxall &= px[j];
            PH_STAMP(3);                                            // projection
            if (!__ballot(pxall >= 0) || PH_DIAG(2)) continue;      // no point of the wave in this image
            // px = iv << 16 | iu (two 16-bit halves; -1 = outside): byte offset of the point's word inside a mask and its bit,
            // once per camera
            PH_COUNT(3, 1);
            uint32_t xw4[PH_PT], iv4[PH_PT];                        // word column and row of the pixel: a mask's rows have its own stride
#pragma unroll
            for (int j = 0; j < PH_PT; ++j) {
                const uint32_t iu = (uint32_t)px[j] & 0xFFFFu;
                iv4[j] = (uint32_t)px[j] >> 16;
                xw4[j] = iu >> 5;
            }
            // the candidate masks of this camera (sorted entries: bounding box, mask number, first word), PH_MB at a time:
            // all entries, then all mask words of the batch are requested before the first one is used (one memory round
            // trip per batch).  Entries are walked through the candidate bits; a camera with more than 32 masks walks its
            // whole range (eb = first entry of a block of 32).
            for (int eb = e0; eb < e1; eb += 32) {
              uint32_t rem = ne <= 32 ? cmask : (e1 - eb >= 32 ? 0xFFFFFFFFu : ((1u << (e1 - eb)) - 1u));
              while (rem) {
                int ei[PH_MB];
#pragma unroll
                for (int b = 0; b < PH_MB; ++b) {
                    ei[b] = rem ? eb + __builtin_ctz(rem) : -1;
                    rem = rem ? (rem & (rem - 1)) : 0u;
                }
                PH_COUNT(4, 1);
                int4 en[PH_MB];
#pragma unroll
                for (int b = 0; b < PH_MB; ++b) en[b] = ment[2 * max(ei[b], e0)];                  // uniform: scalar loads
                uint32_t word[PH_MB][PH_PT];
                int kb[PH_MB];
#pragma unroll
                for (int b = 0; b < PH_MB; ++b) {
#pragma unroll
                    for (int j = 0; j < PH_PT; ++j) word[b][j] = 0u;
                    const int kw = ei[b] >= 0 ? __builtin_amdgcn_readfirstlane(en[b].z) : -1;
                    kb[b] = kw < 0 ? -1 : (kw & 0xFFFF);
                    if (kb[b] < 0) continue;                        // past the last candidate (wave-uniform)
                    const uint32_t wcm = (uint32_t)kw >> 16;        // words per stored row of this mask
                    PH_COUNT(5, 1);
                    const us2 org = __builtin_bit_cast(us2, __builtin_amdgcn_readfirstlane(en[b].x));
                    const us2 ext = __builtin_bit_cast(us2, __builtin_amdgcn_readfirstlane(en[b].y));
                    const char *mw = reinterpret_cast<const char *>(packed) + ((long long)__builtin_amdgcn_readfirstlane(en[b].w) << 2);
#pragma unroll
                    for (int j = 0; j < PH_PT; ++j) {
                        // inside the box <=> (x - x0 <= rx) and (y - y0 <= ry) as unsigned 16-bit halves; -1 (0xFFFF, 0xFFFF) is
                        // outside every box (corners are < 32767)
                        const us2 d = __builtin_bit_cast(us2, px[j]) - org;
                        const us2 m = __builtin_elementwise_min(d, ext);
                        if (__builtin_bit_cast(uint32_t, m) == __builtin_bit_cast(uint32_t, d))
                            word[b][j] = *reinterpret_cast<const uint32_t *>(mw + ((iv4[j] * wcm + xw4[j]) << 2));
                    }
                }
#pragma unroll
                for (int b = 0; b < PH_MB; ++b) {
                    if (kb[b] < 0) continue;
                    int cnt = 0;
#pragma unroll
                    for (int j = 0; j < PH_PT; ++j) {
                        const uint32_t hit = (word[b][j] >> (px[j] & 31)) & 1u;       // word = 0 for a non-candidate; bit = iu & 31
                        if (ONE_PLANE) bits[j] |= hit << kb[b];
                        else if (hit) s_bits[(kb[b] >> 5) * PH_WC + 4 * lane + j] |= 1u << (kb[b] & 31);
                        cnt += (int)__popcll(__ballot(hit != 0u));
                    }
                    if (cnt) {
                        if (ONE_PLANE) mycnt += lane == kb[b] ? cnt : 0;
                        else if (lane == 0) s_cnt[kb[b]] += cnt;
                    }
                }
              }
            }
            PH_STAMP(4);                                            // mask loop of one camera
        }
        PH_STAMP(2);
        // The chunk's results stay in registers (LDS) until the top of the next iteration (flush_results): vmcnt counts loads
        // and stores together and in order, so a store issued here would sit in front of the very next wait -- the one for
        // the prefetched rows -- and every chunk would pay a store's whole round trip to memory.
        if (ONE_PLANE) {
#pragma unroll
            for (int j = 0; j < PH_PT; ++j) pend_bits[j] = bits[j];
            pend_cnt = mycnt;
            acc_cnt += mycnt;
        }
        pend_chunk = chunk;
        pend_drop = drop_now;
        cur = nxt;
        chunk = c_nxt;
        if (c_nxt < nwc) c_nxt = chunk_of(draw_v, draw_from);
        PH_STAMP(5);                                                // wait for the next rows
    } while (chunk < nwc);
    flush_results();
    if (ONE_PLANE && lane < nm && acc_cnt) atomicAdd(&hit_count[m0 + lane], acc_cnt);
    {
        const int tot = cm3d_wave_sum(ONE_PLANE ? (lane < nm ? acc_cnt : 0) : acc_multi);
        if (lane == 0 && tot) atomicAdd(&frame_hits[f], tot);
    }
#ifdef CM3D_DIAG
    if ((diag & 16) && lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_ph_stamp[k], k == 7 ? 1ull : acc[k]);
    if ((diag & 64) && lane == 0) {
        for (int k = 0; k < 8; ++k) atomicAdd(&g_ph_count[k], (unsigned long long)cntk[k]);
    }
    if ((diag & 128) && lane == 0) {
        const unsigned long long t_end = ph_now();
        const int wid = (int)blockIdx.x * PHK_WAVES + wave;
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        if (wid < PH_DIAG_WAVES) {
            g_ph_wave[3 * wid] = t_start; g_ph_wave[3 * wid + 1] = t_end;
            g_ph_wave[3 * wid + 2] = ((unsigned long long)xcc << 32) | hw;
        }
    }
#endif
}

#include "project_q.h"

// The compaction: hit words -> ascending index lists (hit_idx) and the coordinates of every listed point (hit_xyz).
// grid (ceil(ceil(nwc_max / CP_SPAN) / 4), F + 1), one launch for everything behind the projection:
//  * row 0 of the grid (dispatched first): the per-mask hit counts -> hit_off, tile_off, their totals, the overflow flag and the
//    medoid stage's work list (md_build_from_counts; each workgroup its own share, without talking to the others);
//  * rows 1..F: one WAVE per CP_SPAN consecutive wave-chunks of a frame, the same lane <-> row map as k_project_hits (lane l
//    owns rows 4l..4l+3, so ascending row order = (lane, j) order).  What the projection left per wave-chunk (wc_info) says
//    which chunks hold a hit at all -- two thirds do not on the headline shape, and their hit words are never read.  A wave
//    that has work computes its own output offsets: in-mask points of the frames before (frame_hits), of the frame's masks
//    before (hit_count), of the groups of 16 wave-chunks before its own (the group sums) and of the chunks before it inside
//    its group (wc_cnt) -- a few dozen loads that all go out together; no scan kernel, no second launch.  Then per present
//    mask bit: four ballots, rank = hits in lower lanes + own hits in earlier rows.  Every hit also gets its coordinates
//    (gathered from the cloud when there is one, else re-derived from the raw row with the projection kernel's very fma
//    chains).  No workgroup barrier on this path; no atomics on anything that is an output.
struct PhXyzSrc { const float *raw; const float *intensity; int raw_stride; const float *sweep_xf; const float4 *points; };
// CP_SPAN: consecutive wave-chunks per wave (a template parameter; divides PH_GRP)

template <int CP_SPAN>
__global__ __launch_bounds__(PH_THREADS, 4) void k_compact_hits(const uint32_t *__restrict__ hit_words, int n_points_total,
                                                             const int32_t *__restrict__ ft_all, int nm_cap, int nwc_max,
                                                             const int32_t *__restrict__ wc_cnt, const int32_t *__restrict__ wc_info,
                                                             const int32_t *__restrict__ grp, int zstride,
                                                             const int32_t *__restrict__ frame_hits,
                                                             const int32_t *__restrict__ hit_count,
                                                             const uint32_t *__restrict__ removed_bits, const PhXyzSrc xs,
                                                             int32_t *__restrict__ hit_idx, int32_t *__restrict__ hit_row,
                                                             float4 *__restrict__ hit_xyz, int idx_cap, int n_frames, int n_masks,
                                                             int32_t *__restrict__ hit_off, int32_t *__restrict__ tile_off, int tile_cap,
                                                             TileDesc *__restrict__ tile_work, int32_t *__restrict__ status)
{
    // per wave: next output position of every mask the table allows per frame (nm_cap: 128 bytes per wave on the headline shape;
    // a fixed [CM3D_MAX_MASKS_PER_FRAME] slice was 16 KiB per workgroup, LDS the kernels of the other batches in flight could not use)
    extern __shared__ __align__(8) int s_run_dyn[];
    if (blockIdx.y == 0) {
        md_build_from_counts<PH_THREADS>(n_masks, hit_count, idx_cap, tile_cap, hit_off, tile_off, tile_work, status, s_run_dyn,
                                         (int)blockIdx.x, (int)gridDim.x);
        return;
    }
    const int f = (int)blockIdx.y - 1;
    const int lane = cm3d_lane(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c0 = ((int)blockIdx.x * PH_WAVES + wave) * CP_SPAN;
    const int32_t *ft = ft_all + (size_t)f * FT_WORDS;
    const int nwc = ft[7];
    if (c0 >= nwc) return;
    const int32_t *info_f = wc_info + (size_t)f * nwc_max;
    const int info = (lane < CP_SPAN && c0 + lane < nwc) ? info_f[c0 + lane] : 0;
    const uint32_t has = (uint32_t)__ballot(info < 0);                 // bit j: wave-chunk c0 + j holds a hit
    if (!has) return;
    const int p0 = ft[0], n = ft[1], m0 = ft[2], nm = ft[3], sa = ft[4], ns = ft[5], bits_off = ft[6], fused = ft[FT_FUSED];
    const int planes = (nm + 31) >> 5;
    int *s_run = s_run_dyn + (size_t)wave * nm_cap;
    const int g = c0 / PH_GRP, cg0 = g * PH_GRP;                       // the wave's group; its first wave-chunk
    const int32_t *grp_f = grp + (size_t)f * zstride;
    const int ngrp_max = (nwc_max + PH_GRP - 1) / PH_GRP;
    // ---- everything the offsets are made of is requested at once
    // (r04: loads in ROUNDS of unconditional, clamped loads whose values are masked afterwards.  A load inside a loop of unknown trip count or a
    // divergent branch is waited for before the next is issued: the chunks before the wave's inside its group alone were up to 15 dependent
    // round trips per wave, where the comment above said one.)
    int fsum = 0;                                                      // in-mask points of the frames before this one
    for (int q0 = 0; q0 < f; q0 += 4 * 64) {
        int t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = frame_hits[min(q0 + u * 64 + lane, f - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) fsum += q0 + u * 64 + lane < f ? t[u] : 0;
    }
    int dsum = 0;                                                      // dropped rows before wave-chunk c0 (fused: counted by the projection)
    if (fused) {
        for (int q = lane; q < g; q += 64) dsum += grp_f[ngrp_max * nm_cap + q];
        if (lane < c0 - cg0) dsum += info_f[cg0 + lane] & 0xFFFF;
    } else if (removed_bits) {
        for (int q = lane; q < 8 * c0; q += 64) dsum += __popc(removed_bits[(size_t)bits_off + q]);
    }
    const int32_t *cnt_f = wc_cnt + (size_t)f * nwc_max * nm_cap;
    int carry = 0;
    for (int k0 = 0; k0 < nm; k0 += 64) {
        const int k = k0 + lane;
        int cnt = 0, pre = 0;
        if (k < nm) {
            cnt = hit_count[m0 + k];
            for (int q = 0; q < g; q += 8) {                           // groups before the wave's (uniform trip count), eight per round
                int a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = grp_f[min(q + u, g - 1) * nm_cap + k];
#pragma unroll
                for (int u = 0; u < 8; ++u) pre += q + u < g ? a[u] : 0;
            }
            for (int c = cg0; c < c0; c += 8) {                        // wave-chunks before it inside its group (< PH_GRP: two rounds at most)
                int a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = cnt_f[(size_t)min(c + u, c0 - 1) * nm_cap + k];
#pragma unroll
                for (int u = 0; u < 8; ++u) pre += c + u < c0 ? a[u] : 0;
            }
        }
        const int inc = cm3d_wave_incl_scan(cnt);
        if (k < nm) s_run[k] = carry + inc - cnt + pre;                // + the frame's base, below
        carry += __builtin_amdgcn_readlane(inc, 63);
    }
    const int fbase = __builtin_amdgcn_readfirstlane(cm3d_wave_sum(fsum));
    int dbase = __builtin_amdgcn_readfirstlane(cm3d_wave_sum(dsum));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int32_t *srow = ft + FT_SROW;
    const int v_srow = lane < min(ns, PH_MAX_SWEEPS + 1) ? srow[lane] : 0x7FFFFFFF;                  // (ph_sweep_of_u)
    // Hit words (plane 0) of every wave-chunk of the span that holds a hit: all requested before the first is used.  The
    // array is only ever indexed at [0] and rotated, so it stays in registers without unrolling the loop.
    uint32_t wq[CP_SPAN][PH_PT];
    auto load_words = [&](int chunk, int plane, uint32_t (&w)[PH_PT]) {
        const int cb = chunk * PH_WC, nvalid = min(PH_WC, n - cb);
        const uint32_t *hw = hit_words + (size_t)plane * n_points_total + p0 + cb + 4 * lane;
        if (nvalid >= PH_WC) {
            const u4u t = *reinterpret_cast<const u4u *>(hw);
            w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
        } else {
#pragma unroll
            for (int jj = 0; jj < PH_PT; ++jj) w[jj] = 4 * lane + jj < nvalid ? hw[jj] : 0u;
        }
    };
#pragma unroll
    for (int q = 0; q < CP_SPAN; ++q) {
#pragma unroll
        for (int jj = 0; jj < PH_PT; ++jj) wq[q][jj] = 0u;
        if ((has >> q) & 1u) load_words(c0 + q, 0, wq[q]);
    }
    int2 *const tag_out = reinterpret_cast<int2 *>(hit_xyz);          // 16 bytes per position; the first 8 carry (row of the batch, sweep)
#pragma unroll 1
    for (int j = 0; j < CP_SPAN; ++j) {
        const int chunk = c0 + j;
        if (chunk >= nwc) break;                                       // uniform
        const int inf = __builtin_amdgcn_readlane(info, j);
        const int cb = chunk * PH_WC;
        const int nvalid = min(PH_WC, n - cb);
        int dropc = inf & 0xFFFF;                                      // rows of this chunk the reference drops
        uint32_t wv = 0u;                                              // lanes 0..7: the chunk's 8 words of removed-row bits
        const bool want_bits = removed_bits && (fused ? (dropc != 0 && inf < 0) : true);
        if (want_bits && lane < 8) wv = removed_bits[(size_t)bits_off + 8 * chunk + lane];
        if (!fused) dropc = __builtin_amdgcn_readfirstlane(cm3d_wave_sum(__popc(wv)));
        if (inf < 0) {
            // dropped rows before each of this lane's rows
            int dropped[PH_PT] = {dbase, dbase, dbase, dbase};
            if (__ballot(wv != 0u)) {
                int pc = __popc(wv), inc = pc;                       // lanes 0..7: inclusive prefix over the chunk's 8 words
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) { const int tt = __shfl_up(inc, o, 64); if (lane >= o) inc += tt; }
                const int pre = __shfl(inc - pc, lane >> 3, 64);
                const uint32_t myw = (uint32_t)__shfl((int)wv, lane >> 3, 64);
#pragma unroll
                for (int jj = 0; jj < PH_PT; ++jj) dropped[jj] = dbase + pre + __popc(myw & ((1u << (4 * (lane & 7) + jj)) - 1u));
            }
            // sweep of each of this lane's rows (k_hit_xyz re-derives the coordinates of the listed rows from the raw rows)
            int sweep[PH_PT] = {sa, sa, sa, sa};
            if (ns > 1) {
                const int sw_lo = ph_sweep_of_u(v_srow, cb), sw_hi = ph_sweep_of_u(v_srow, cb + nvalid - 1);
#pragma unroll
                for (int jj = 0; jj < PH_PT; ++jj) sweep[jj] = sa + (sw_lo >= sw_hi ? sw_lo : ph_sweep_of(srow, ns, cb + 4 * lane + jj));
            }
            for (int plane = 0; plane < planes; ++plane) {
                uint32_t w[PH_PT];
                if (plane == 0) {
#pragma unroll
                    for (int jj = 0; jj < PH_PT; ++jj) w[jj] = wq[0][jj];
                } else {
                    load_words(chunk, plane, w);
                }
                const uint32_t orw = cm3d_wave_or(w[0] | w[1] | w[2] | w[3]);
                for (uint32_t r = orw; r; r &= r - 1) {
                    const int b = __builtin_ctz(r);
                    uint64_t mk[PH_PT];
                    int lower = 0, total = 0;
#pragma unroll
                    for (int jj = 0; jj < PH_PT; ++jj) { mk[jj] = __ballot((w[jj] >> b) & 1u); lower += cm3d_mbcnt(mk[jj]); total += (int)__popcll(mk[jj]); }
                    const int runpos = s_run[plane * 32 + b];       // (the same address in every lane: a broadcast)
                    const int basepos = fbase + runpos;
                    int own = 0;
#pragma unroll
                    for (int jj = 0; jj < PH_PT; ++jj) {
                        if ((w[jj] >> b) & 1u) {
                            const int pos = basepos + lower + own;
                            ++own;
                            if (pos >= 0 && pos < idx_cap) {
                                const int i = cb + 4 * lane + jj;
                                hit_idx[pos] = i - dropped[jj];
                                if (hit_row) hit_row[pos] = i;
                                if (hit_xyz) tag_out[2 * (size_t)pos] = make_int2(p0 + i, sweep[jj]);
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();                  // every lane has read s_run[..] before it moves on
                    if (lane == 0) s_run[plane * 32 + b] = runpos + total;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        dbase += dropc;
        // rotate: the next wave-chunk becomes the current one
#pragma unroll
        for (int q = 0; q + 1 < CP_SPAN; ++q) {
#pragma unroll
            for (int jj = 0; jj < PH_PT; ++jj) wq[q][jj] = wq[q + 1][jj];
        }
    }
}

// Coordinates of the listed points: one thread per output position.  k_compact_hits left (row of the batch, sweep) in the
// first 8 bytes of the position's 16; here the row is fetched (from the cloud when there is one, else the raw row, which is
// put through the projection kernel's very fma chains: same bits as the cloud would hold) and the position overwritten with
// x, y, z, intensity.  Every thread is independent: tens of thousands of gathers in flight, where the compaction's waves
// would each have waited for their own.
__global__ __launch_bounds__(256) void k_hit_xyz(const PhXyzSrc xs, const int32_t *__restrict__ hit_off, int n_masks, int idx_cap,
                                                 float4 *__restrict__ hit_xyz)
{
    const int total = min(hit_off[n_masks], idx_cap);
    for (int pos = (int)(blockIdx.x * blockDim.x + threadIdx.x); pos < total; pos += (int)(gridDim.x * blockDim.x)) {
        const int2 tag = reinterpret_cast<const int2 *>(hit_xyz)[2 * (size_t)pos];
        float4 out;
        if (xs.points) out = xs.points[tag.x];
        else {
            float x, y, z, w;
            if (xs.raw_stride == CM3D_RAW_QUADS) {              // quad layout: x, y, z of row r at 12 (r >> 2) + (r & 3) + 0 / 4 / 8
                const float *p = xs.raw + (size_t)(tag.x >> 2) * 12 + (tag.x & 3);
                x = p[0]; y = p[4]; z = p[8];
                w = xs.intensity ? xs.intensity[tag.x] : 0.f;
            } else {
                const float *p = xs.raw + (size_t)tag.x * xs.raw_stride;
                x = p[0]; y = p[1]; z = p[2]; w = p[3];
            }
            float bx, by, bz;
            ph_xform(xs.sweep_xf + (size_t)tag.y * CM3D_SWEEP_XF_STRIDE, x, y, z, bx, by, bz);
            out = make_float4(bx, by, bz, w);
        }
        hit_xyz[pos] = out;
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
                              uint32_t *__restrict__ removed_bits, long long removed_words)
{
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
    if (i0 < CM3D_STATUS_WORDS) status[i0] = 0;
    for (long long i = i0; i < n_masks; i += step) hit_count[i] = 0;
    for (long long i = i0; i < removed_words; i += step) removed_bits[i] = 0u;
}

extern "C" int64_t cm3d_removed_words(int32_t n_rows, int32_t n_frames)
{
    if (n_rows < 0 || n_frames <= 0) return 0;
    return ((int64_t)n_rows >> 5) + 8 * (int64_t)n_frames + 8;
}

extern "C" int cm3d_batch_begin(int32_t *status, int32_t *hit_count, int32_t n_masks, uint32_t *removed_bits, int64_t removed_words,
                                cm3d_stream_t stream)
{
    if (!status || !hit_count || n_masks <= 0 || (removed_bits && removed_words <= 0)) return CM3D_ERR_ARG;
    const long long n = removed_bits && removed_words > n_masks ? removed_words : n_masks;
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_batch_begin, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, status, hit_count, n_masks, removed_bits,
                       removed_bits ? (long long)removed_words : 0ll);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}

extern "C" int64_t cm3d_project_workspace_bytes(int32_t n_frames, int32_t max_pts_per_frame, int32_t planes)
{
    if (n_frames <= 0 || max_pts_per_frame <= 0 || planes <= 0) return 0;
    return ph_ws_layout(n_frames, max_pts_per_frame, planes, nullptr, nullptr);
}

// Accounting aid (bench.py's byte counts; nothing on the path calls it): rows of the batch that lie in a wave-chunk with at
// least one in-mask point -- the only rows whose hit words the projection writes and the compaction reads.  Synchronous:
// copies the per-frame tables and the per-chunk flags of `workspace` to the host (after the stream's work has completed).
extern "C" int cm3d_project_hit_rows(const void *workspace, int64_t workspace_bytes, int32_t n_frames, int32_t max_pts_per_frame,
                                     int32_t planes, int64_t *rows_out, cm3d_stream_t stream)
{
    if (!workspace || !rows_out || n_frames <= 0 || max_pts_per_frame <= 0 || planes <= 0) return CM3D_ERR_ARG;
    if (workspace_bytes < cm3d_project_workspace_bytes(n_frames, max_pts_per_frame, planes) || ((uintptr_t)workspace & 15)) return CM3D_ERR_WORKSPACE;
    try {
        PhWs ws;
        ph_ws_layout(n_frames, max_pts_per_frame, planes, const_cast<void *>(workspace), &ws);
        const int nwc_max = (max_pts_per_frame + PH_WC - 1) / PH_WC;
        std::vector<int32_t> ft((size_t)n_frames * FT_WORDS), info((size_t)n_frames * nwc_max);
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return CM3D_ERR_LAUNCH;
        if (hipMemcpy(ft.data(), ws.ft, ft.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return CM3D_ERR_LAUNCH;
        if (hipMemcpy(info.data(), ws.wc_info, info.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return CM3D_ERR_LAUNCH;
        int64_t rows = 0;
        for (int f = 0; f < n_frames; ++f) {
            const int n = ft[(size_t)f * FT_WORDS + 1], nwc = ft[(size_t)f * FT_WORDS + 7];
            for (int c = 0; c < nwc && c < nwc_max; ++c)
                if (info[(size_t)f * nwc_max + c] < 0) rows += std::min(PH_WC, n - c * PH_WC);
        }
        *rows_out = rows;
    } catch (...) { return CM3D_ERR_ARG; }
    return CM3D_OK;
}

// cm3d_project_workgroups_per_cu (cm3d_hip.h): 0 = fill the chip
static int g_ph_wg_per_cu = 0;
extern "C" int cm3d_project_workgroups_per_cu(int32_t n)
{
    const int prev = g_ph_wg_per_cu;
    g_ph_wg_per_cu = n < 0 ? 0 : n;
    return prev;
}

// workgroups of the projection kernel the chip holds at once.  CM3D_PH_BLOCKS overrides (experiments).
static int ph_target_blocks(const void *kernel, size_t lds, bool share)
{
    static int forced = -1, cus = 0;
    if (forced < 0) {
        const char *e = getenv("CM3D_PH_BLOCKS");
        forced = e ? atoi(e) : 0;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    if (forced > 0) return forced;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, PHK_THREADS, lds) != hipSuccess || per_cu <= 0) per_cu = 4;
    // One workgroup per CU fewer than fit: three waves per SIMD run the launch as fast as four (measured alone: 65.7 against
    // 64.4 us), and the registers the fourth would hold are what lets the small kernels of the neighbouring batches in flight
    // (masks, compaction, medoid) run beside it instead of behind it (three batches in flight: 0.177 -> 0.174 ms per pass).
    if (per_cu >= 4) --per_cu;
    // a caller with several batches in flight leaves room beside the launch (cm3d_project_workgroups_per_cu): 768 / 512 / 384 workgroups
    // give 2.302 / 2.343 / 2.363 M frames/s on C2 with three batches in flight, 609 / 613 / 606 k on C1, 849 / 865 / 850 k on C4
    // (k_project_q only: the multi-plane kernel of frames with more than 32 masks -- hit words in LDS -- is not what the others wait for on its shapes;
    // C5, four in flight: 47.9 k frames/s with its full grid, 47.4 k at one workgroup per CU)
    if (share && g_ph_wg_per_cu > 0 && g_ph_wg_per_cu < per_cu) per_cu = g_ph_wg_per_cu;
    return cus * per_cu;
}

static int ph_launch(const PhSweepIn *fused, const float *points, const int32_t *pt_off, int32_t n_frames, int32_t max_pts_per_frame,
                     int32_t n_points_total, const float *cams, int32_t n_cams, const int32_t *mask_off, const int32_t *mask_cam,
                     const int32_t *bbox, const uint32_t *packed, int32_t n_masks, int32_t W, int32_t H, float min_dist,
                     int32_t planes, uint32_t *hit_words, int32_t *hit_count, int32_t *status, void *workspace,
                     int64_t workspace_bytes, void *ev_start, void *ev_stop, cm3d_stream_t stream)
{
    if (!cams || !mask_off || !mask_cam || !bbox || !packed || !hit_words || !hit_count || !status || !workspace) return CM3D_ERR_ARG;
    if (n_frames <= 0 || max_pts_per_frame <= 0 || n_points_total <= 0 || n_cams <= 0 || n_cams > CM3D_MAX_CAMS ||
        n_masks <= 0 || W <= 1 || H <= 1 || W > 32767 || H > 32767 || planes <= 0)
        return CM3D_ERR_ARG;
    if (workspace_bytes < cm3d_project_workspace_bytes(n_frames, max_pts_per_frame, planes) || ((uintptr_t)workspace & 15)) return CM3D_ERR_WORKSPACE;
    // a mask's first word in `packed` travels as a 32-bit word offset in its table entry (k_frame_tables): ~95 k masks of 1600x900
    if ((int64_t)n_masks * H * ((W + 31) / 32) > 0x7FFFFFFFll) return CM3D_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    PhWs ws;
    ph_ws_layout(n_frames, max_pts_per_frame, planes, workspace, &ws);
    const int Wp = (W + 31) / 32;
    const int nwc_max = (max_pts_per_frame + PH_WC - 1) / PH_WC;
    const int nm_cap = ph_nm_cap(planes);
    const int planes_cap = (nm_cap + 31) / 32;
    PhSweepIn none = {};
    const PhSweepIn sw = fused ? *fused : none;
    const float *src = fused ? sw.raw : points;
    const int stride = fused ? sw.raw_stride : 4;
    const bool keep = fused && sw.points_out != nullptr;
    // variants: prepared cloud | raw rows of 5, 4, any number of columns, each with and without the cloud store
    // ... | quad layout (12 bytes per row), with and without the cloud store
    const int which = !fused ? 0 : stride == CM3D_RAW_QUADS ? (keep ? 8 : 7) : (stride == 5 ? 1 : (stride == 4 ? 2 : 3)) + (keep ? 3 : 0);
    const bool one = planes_cap == 1;
    const void *fn;
#define PH_PICK(ONE)                                                                                                             \
    (which == 0 ? (const void *)k_project_hits<ONE, false, 4, false>                                                             \
     : which == 1 ? (const void *)k_project_hits<ONE, true, 5, false>                                                            \
     : which == 2 ? (const void *)k_project_hits<ONE, true, 4, false>                                                            \
     : which == 3 ? (const void *)k_project_hits<ONE, true, 0, false>                                                            \
     : which == 4 ? (const void *)k_project_hits<ONE, true, 5, true>                                                             \
     : which == 5 ? (const void *)k_project_hits<ONE, true, 4, true>                                                             \
     : which == 6 ? (const void *)k_project_hits<ONE, true, 0, true>                                                             \
     : which == 7 ? (const void *)k_project_hits<ONE, true, CM3D_RAW_QUADS, false> : (const void *)k_project_hits<ONE, true, CM3D_RAW_QUADS, true>)
    if (one) fn = PH_PICK(true);
    else fn = PH_PICK(false);
#undef PH_PICK
    // the quad layout has its own kernel (project_q.h) for frames of up to 96 masks; CM3D_PQ=0: k_project_hits reads the quads
    static int pq_on = -1;
    if (pq_on < 0) { const char *e = getenv("CM3D_PQ"); pq_on = e ? atoi(e) : 1; }
    // (one hit-word plane only: with 80 masks per frame -- three planes in registers, 13 entries per camera -- k_project_q<3> takes
    //  418 us per 64 frames of the 10-sweep configuration where k_project_hits with its planes in LDS takes 344: CM3D_PQ=3 to compare)
    const bool pq = pq_on && (which == 7 || which == 8) && planes_cap <= (pq_on >= 3 ? 3 : 1);
    if (pq) fn = one ? (keep ? (const void *)k_project_q<1, true> : (const void *)k_project_q<1, false>)
                     : (keep ? (const void *)k_project_q<3, true> : (const void *)k_project_q<3, false>);
    size_t lds = (one || pq) ? 0 : (size_t)PHK_WAVES * ((size_t)planes_cap * PH_WC * sizeof(uint32_t) + (size_t)nm_cap * sizeof(int));
    if (!one && !pq) {
        static size_t lds_allowed[9] = {48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024};
        if (lds > lds_allowed[which]) {
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return CM3D_ERR_LAUNCH;
            lds_allowed[which] = lds;
        }
    }
    static int blocks_one[9 + 4] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};       // cached for the register-only variants (no dynamic LDS)
    static int blocks_hint = 0;                                                   // ... under this value of cm3d_project_workgroups_per_cu
    if (blocks_hint != g_ph_wg_per_cu) {
        for (int &b : blocks_one) b = 0;
        blocks_hint = g_ph_wg_per_cu;
    }
    const int slot_id = pq ? 9 + (one ? 0 : 2) + (keep ? 1 : 0) : which;
    int target = (one || pq) ? blocks_one[slot_id] : 0;
    if (!target) {
        target = ph_target_blocks(fn, lds, pq);
        if (one || pq) blocks_one[slot_id] = target;
    }
    // tickets per frame: about PH_OVERSUB times as many waves as the chip holds at once (see the kernel's header), at least
    // two wave-chunks per ticket when the frames are that long
    static int tpf_forced = -1;
    if (tpf_forced < 0) { const char *e = getenv("CM3D_PH_TICKETS"); tpf_forced = e ? atoi(e) : 0; }
    int tpf = tpf_forced > 0 ? tpf_forced : (int)(((long long)PH_OVERSUB_NUM * target * PHK_WAVES / PH_OVERSUB_DEN + n_frames - 1) / n_frames);
    if (tpf > ph_tpf_max(nwc_max)) tpf = ph_tpf_max(nwc_max);
    if (tpf < 1) tpf = 1;
    const int gx = (int)(((long long)n_frames * tpf + PHK_WAVES - 1) / PHK_WAVES);
#define PH_LAUNCH(ONE, FUSED, STRIDE, KEEP)                                                                                      \
    hipLaunchKernelGGL((k_project_hits<ONE, FUSED, STRIDE, KEEP>), dim3(gx), dim3(PHK_THREADS), lds, st, src, sw.intensity, stride, sw.sweep_xf, \
                       sw.halfw, sw.points_out, sw.removed_bits, ws.ft, ws.ment, cams, n_cams, packed, W, H, Wp, min_dist, nm_cap, \
                       nwc_max, n_points_total, hit_words, hit_count, ws.wc_cnt, n_frames, tpf, ws.queue, ws.wc_info, ws.grp, ws.zstride,     \
                       ws.frame_hits)
#define PH_LAUNCH_S(ONE)                                                                                                         \
    do {                                                                                                                         \
        if (which == 0) PH_LAUNCH(ONE, false, 4, false);                                                                         \
        else if (which == 1) PH_LAUNCH(ONE, true, 5, false);                                                                     \
        else if (which == 2) PH_LAUNCH(ONE, true, 4, false);                                                                     \
        else if (which == 3) PH_LAUNCH(ONE, true, 0, false);                                                                     \
        else if (which == 4) PH_LAUNCH(ONE, true, 5, true);                                                                      \
        else if (which == 5) PH_LAUNCH(ONE, true, 4, true);                                                                      \
        else if (which == 6) PH_LAUNCH(ONE, true, 0, true);                                                                      \
        else if (which == 7) PH_LAUNCH(ONE, true, CM3D_RAW_QUADS, false);                                                        \
        else PH_LAUNCH(ONE, true, CM3D_RAW_QUADS, true);                                                                         \
    } while (0)
    hipLaunchKernelGGL(k_frame_tables, dim3(n_frames), dim3(64), 0, st, sw, fused ? 1 : 0, pt_off, n_frames, cams, n_cams, mask_off, mask_cam,
                       (const int4 *)bbox, W, H, min_dist, nm_cap, max_pts_per_frame, (uint32_t)H * (uint32_t)Wp, ws.ft, ws.ment, ws.queue, tpf,
                       ws.grp, ws.zstride, ws.frame_hits, status);
    CM3D_CHECK_LAUNCH();
    // optional timing events around the projection kernel itself (the table kernel above is not part of it)
    if (ev_start && hipEventRecord((hipEvent_t)ev_start, st) != hipSuccess) return CM3D_ERR_LAUNCH;
    if (pq) {
        PqArgs qa;
        qa.raw = src; qa.inten = sw.intensity; qa.sweep_xf = sw.sweep_xf; qa.points_out = sw.points_out; qa.removed_bits = sw.removed_bits;
        qa.ft_all = ws.ft; qa.ment_all = ws.ment; qa.cams = cams; qa.packed = packed; qa.hit_words = hit_words; qa.hit_count = hit_count;
        qa.wc_cnt = ws.wc_cnt; qa.queue = ws.queue; qa.wc_info = ws.wc_info; qa.grp = ws.grp; qa.frame_hits = ws.frame_hits;
        qa.halfw = sw.halfw; qa.min_dist = min_dist; qa.n_cams = n_cams; qa.W = W; qa.H = H; qa.nm_cap = nm_cap; qa.nwc_max = nwc_max;
        qa.n_points_total = n_points_total; qa.n_frames = n_frames; qa.tpf = tpf; qa.zstride = ws.zstride;
        qa.stage = 99;
#ifdef CM3D_DIAG
        qa.stage = g_pq_stage;
#endif
        if (one) { if (keep) hipLaunchKernelGGL((k_project_q<1, true>), dim3(gx), dim3(PHK_THREADS), 0, st, qa);
                   else hipLaunchKernelGGL((k_project_q<1, false>), dim3(gx), dim3(PHK_THREADS), 0, st, qa); }
        else { if (keep) hipLaunchKernelGGL((k_project_q<3, true>), dim3(gx), dim3(PHK_THREADS), 0, st, qa);
               else hipLaunchKernelGGL((k_project_q<3, false>), dim3(gx), dim3(PHK_THREADS), 0, st, qa); }
    } else if (one) PH_LAUNCH_S(true);
    else PH_LAUNCH_S(false);
    if (ev_stop && hipEventRecord((hipEvent_t)ev_stop, st) != hipSuccess) return CM3D_ERR_LAUNCH;
#undef PH_LAUNCH_S
#undef PH_LAUNCH
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}

extern "C" int cm3d_project_hits(const float *points, const int32_t *pt_off, int32_t n_frames, int32_t max_pts_per_frame,
                                 int32_t n_points_total, const float *cams, int32_t n_cams, const int32_t *mask_off,
                                 const int32_t *mask_cam, const int32_t *bbox, const uint32_t *packed, int32_t n_masks,
                                 int32_t W, int32_t H, float min_dist, int32_t planes, uint32_t *hit_words,
                                 int32_t *hit_count, int32_t *status, void *workspace, int64_t workspace_bytes,
                                 void *ev_start, void *ev_stop, cm3d_stream_t stream)
{
    if (!points || !pt_off) return CM3D_ERR_ARG;
    return ph_launch(nullptr, points, pt_off, n_frames, max_pts_per_frame, n_points_total, cams, n_cams, mask_off, mask_cam, bbox,
                     packed, n_masks, W, H, min_dist, planes, hit_words, hit_count, status, workspace, workspace_bytes, ev_start, ev_stop,
                     stream);
}

extern "C" int cm3d_sweep_project_hits(const float *raw, int32_t raw_stride, const float *intensity, const int32_t *sweep_row_off, int32_t n_sweeps,
                                       int32_t max_sweeps_per_frame, const float *sweep_xf, const int32_t *frame_sweep_off,
                                       float halfw, float *points, int32_t pt_cap, int32_t *pt_off, uint32_t *removed_bits,
                                       int32_t n_frames, int32_t max_pts_per_frame, int32_t n_points_total,
                                       const float *cams, int32_t n_cams, const int32_t *mask_off, const int32_t *mask_cam,
                                       const int32_t *bbox, const uint32_t *packed, int32_t n_masks, int32_t W, int32_t H,
                                       float min_dist, int32_t planes, uint32_t *hit_words, int32_t *hit_count, int32_t *status,
                                       void *workspace, int64_t workspace_bytes, void *ev_start, void *ev_stop, cm3d_stream_t stream)
{
    if (!raw || !sweep_row_off || !sweep_xf || !frame_sweep_off || !pt_off || !removed_bits) return CM3D_ERR_ARG;
    if ((raw_stride < 4 && raw_stride != CM3D_RAW_QUADS) || n_sweeps <= 0 || pt_cap <= 0 || ((uintptr_t)points & 15)) return CM3D_ERR_ARG;
    if (raw_stride == CM3D_RAW_QUADS && (((uintptr_t)raw & 15) || ((uintptr_t)intensity & 15))) return CM3D_ERR_ARG;
    if (raw_stride != CM3D_RAW_QUADS && intensity) return CM3D_ERR_ARG;                       // rows carry their own fourth column
    if (max_sweeps_per_frame <= 0 || max_sweeps_per_frame > PH_MAX_SWEEPS) return CM3D_ERR_ARG;
    PhSweepIn sw;
    sw.raw = raw; sw.intensity = intensity; sw.raw_stride = raw_stride; sw.sweep_row_off = sweep_row_off; sw.sweep_xf = sweep_xf;
    sw.frame_sweep_off = frame_sweep_off; sw.n_frames = n_frames; sw.n_sweeps = n_sweeps; sw.halfw = halfw;
    sw.points_out = (float4 *)points; sw.pt_cap = pt_cap; sw.pt_off_out = pt_off; sw.removed_bits = removed_bits;
    return ph_launch(&sw, nullptr, nullptr, n_frames, max_pts_per_frame, n_points_total, cams, n_cams, mask_off, mask_cam, bbox, packed,
                     n_masks, W, H, min_dist, planes, hit_words, hit_count, status, workspace, workspace_bytes, ev_start, ev_stop, stream);
}

extern "C" int cm3d_compact_hits(const uint32_t *hit_words, int32_t planes, int32_t n_frames, int32_t max_pts_per_frame,
                                 int32_t n_points_total, const int32_t *mask_off, int32_t n_masks, const int32_t *hit_count,
                                 const uint32_t *removed_bits, const float *raw, int32_t raw_stride, const float *intensity, const float *sweep_xf,
                                 const float *points, int32_t *hit_off, int32_t *tile_off, int32_t *hit_idx, int32_t *hit_row,
                                 float *hit_xyz, int32_t idx_cap, int32_t *tile_work, int32_t *status, void *workspace,
                                 int64_t workspace_bytes, cm3d_stream_t stream)
{
    if (!hit_words || !mask_off || !hit_count || !hit_off || !tile_off || !hit_idx || !status || !workspace) return CM3D_ERR_ARG;
    if (planes <= 0 || n_frames <= 0 || max_pts_per_frame <= 0 || n_points_total <= 0 || n_masks <= 0 || idx_cap <= 0)
        return CM3D_ERR_ARG;
    if (hit_xyz && !points && !(raw && sweep_xf && (raw_stride >= 4 || raw_stride == CM3D_RAW_QUADS))) return CM3D_ERR_ARG;        // nothing to take the coordinates from
    if (intensity && raw_stride != CM3D_RAW_QUADS) return CM3D_ERR_ARG;
    if (((uintptr_t)hit_xyz & 15) || ((uintptr_t)points & 15)) return CM3D_ERR_ARG;
    if (workspace_bytes < cm3d_project_workspace_bytes(n_frames, max_pts_per_frame, planes) || ((uintptr_t)workspace & 15)) return CM3D_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    PhWs ws;
    ph_ws_layout(n_frames, max_pts_per_frame, planes, workspace, &ws);
    const int nwc_max = (max_pts_per_frame + PH_WC - 1) / PH_WC;
    const int nm_cap = ph_nm_cap(planes);
    const int64_t tile_cap64 = md_tile_cap(n_masks, idx_cap);
    const int tile_cap = (int)(tile_cap64 > 0x7FFFFFFF ? 0x7FFFFFFF : tile_cap64);
    PhXyzSrc xs;
    xs.raw = raw; xs.intensity = intensity; xs.raw_stride = raw_stride; xs.sweep_xf = sweep_xf; xs.points = (const float4 *)points;
    // wave-chunks per wave: 4; 8 for frames of 150 k points and more (C4: 906 -> 920 k frames/s; C2 2.59 -> 2.55 M with 8, C1 and C5 no difference);
    // CM3D_CP_SPAN = 2 / 4 / 8 forces one (experiments, tests/test_gpu_golden.py)
    static int span_forced = -1;
    if (span_forced < 0) { const char *e = getenv("CM3D_CP_SPAN"); span_forced = e ? atoi(e) : 0; if (span_forced != 2 && span_forced != 4 && span_forced != 8) span_forced = 0; }
    const int span = span_forced ? span_forced : (nwc_max >= 600 ? 8 : 4);
    // dynamic LDS: a slice of nm_cap ints per wave; the builder row needs MD_FROM_COUNTS_LDS ints
    const size_t cp_lds = sizeof(int) * (size_t)std::max(PH_WAVES * nm_cap, (int)MD_FROM_COUNTS_LDS(PH_THREADS));
#define CP_LAUNCH(SPAN)                                                                                                          \
    hipLaunchKernelGGL(k_compact_hits<SPAN>, dim3(((nwc_max + SPAN - 1) / SPAN + PH_WAVES - 1) / PH_WAVES, n_frames + 1),         \
                       dim3(PH_THREADS), cp_lds, st, hit_words, n_points_total, ws.ft, nm_cap, nwc_max, ws.wc_cnt, ws.wc_info, ws.grp,   \
                       ws.zstride, ws.frame_hits, hit_count, removed_bits, xs, hit_idx, hit_row, (float4 *)hit_xyz, idx_cap,     \
                       n_frames, n_masks, hit_off, tile_off, tile_cap, (TileDesc *)tile_work, status)
    if (span == 8) CP_LAUNCH(8);
    else if (span == 2) CP_LAUNCH(2);
    else CP_LAUNCH(4);
#undef CP_LAUNCH
    CM3D_CHECK_LAUNCH();
    if (hit_xyz) {
        // (the number of listed points is only known on the device: a fixed grid strides over them)
        int64_t blocks = ((int64_t)idx_cap + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_hit_xyz, dim3((unsigned)blocks), dim3(256), 0, st, xs, hit_off, n_masks, idx_cap, (float4 *)hit_xyz);
    }
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}
