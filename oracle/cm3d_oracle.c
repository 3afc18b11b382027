/*
 * cm3d_oracle.c -- CPU restatement of CM3D's 2D->3D pseudo-label lifting path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the CPU
 * baseline.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.  The product path (cm3d_amd/) never calls into it.
 *
 * It restates, in plain C, what the reference executes (file:line cites are
 * relative to the reference checkout, src/nuscenes/ unless noted), in the
 * order the reference executes it (per mask: clone the whole cloud, re-project
 * it, full-frame 3x3 erode, O(M^2) medoid ...).  float32 arithmetic follows
 * the op order that torch-CPU produces for the reference's tensor ops
 * (SURVEY.md appendix B): small matmuls are k-sequential fmaf chains, cdist
 * has a direct (<=25 rows) and an expansion (>25 rows) branch.
 *
 * Parity pinning: the reference ships no tests/golden vectors.  This oracle
 * is pinned against (a) the reference's own importable helpers (get_medoid,
 * view_points, LidarPointCloud.translate/rotate, push_centroid, circle_nms,
 * lane_yaws_distances_and_coords) executed in the build container and frozen
 * under tests/golden/ by tests/golden/gen_golden.py, and (b) torch-CPU ops.
 * Third-party semantics that are absent from the reference checkout
 * (cv2.erode 4.8.1, pycocotools 2.0.7 RLE, pyquaternion 0.9.9) are restated
 * from their published behaviour: those pieces are "parity unpinned".
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 * -ffp-contract=off matters: every fused multiply-add below is explicit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* camera record layout shared with include/cm3d_hip.h (CM3D_CAM_STRIDE floats): up to three rigid
 * stages, each `p += t_pre; p = R p; p += t_post` (either translation optional), then K:
 *  stage s at [15s..15s+14]: t_pre(3), R(9 row-major), t_post(3)
 *    nuScenes: stage 0 = (f32(-ego_pose.translation), f32(R_ego^T), -) 2d_to_3d.py:570-571,
 *              stage 1 = (f32(-calibrated_sensor.translation), f32(R_cs^T), -) :576-577
 *    Waymo:    one stage (src/waymo/2d_to_3d.py:575-576)
 *    KITTI:    (-, C2V_R, C2V_t), (-, V2C_R, V2C_t), (-, R0, -)  src/kitti/2d_to_3d.py:1238-1240
 *  [45..53] K    row-major 3x3 scaled intrinsics, K[2][2]=1 (:585-587)
 *  [54]     number of stages, [55] flags: bit 2s = t_pre present, bit 2s+1 = t_post present
 */
#define CAM_STRIDE 64
#define CAM_K 45

/* ------------------------------------------------------------------ */
/* a2: sweep preparation  (2d_to_3d.py:437-465, utils/pcd.py:159-172,246-257) */
/* raw: (n_raw, stride) float32 rows of a .bin sweep, first 4 columns used.
 * Drops |x|<halfw && |y|<halfw (halfw = f32(sqrt(2.3)), :442-445), then
 * p = R_cs p; p += t_cs; p = R_ego p; p += t_ego, each matmul row a
 * k-sequential fma chain.  Output rows (x,y,z,intensity), order preserved. */
ORC_API int64_t orc_sweep_prep(const float *raw, int64_t n_raw, int stride,
                               const float *R_cs, const float *t_cs,
                               const float *R_ego, const float *t_ego,
                               float halfw, float *out)
{
    int64_t n = 0;
    for (int64_t i = 0; i < n_raw; ++i) {
        const float *p = raw + i * stride;
        float x = p[0], y = p[1], z = p[2];
        if (fabsf(x) < halfw && fabsf(y) < halfw) continue;
        float a[3];
        for (int r = 0; r < 3; ++r) {
            float acc = R_cs[3 * r] * x;
            acc = fmaf(R_cs[3 * r + 1], y, acc);
            acc = fmaf(R_cs[3 * r + 2], z, acc);
            a[r] = acc;
        }
        for (int r = 0; r < 3; ++r) a[r] = a[r] + t_cs[r];
        float b[3];
        for (int r = 0; r < 3; ++r) {
            float acc = R_ego[3 * r] * a[0];
            acc = fmaf(R_ego[3 * r + 1], a[1], acc);
            acc = fmaf(R_ego[3 * r + 2], a[2], acc);
            b[r] = acc;
        }
        for (int r = 0; r < 3; ++r) b[r] = b[r] + t_ego[r];
        float *o = out + 4 * n;
        o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = p[3];
        ++n;
    }
    return n;
}

/* ------------------------------------------------------------------ */
/* a1: COCO RLE (pycocotools 2.0.7 maskApi.c rleFrString / rleDecode; the
 * reference calls pycocotools.mask.decode at 2d_to_3d.py:425). parity unpinned:
 * restated from the published format. */
ORC_API int64_t orc_rle_string_to_counts(const char *s, int64_t len, uint32_t *cnts, int64_t cap)
{
    int64_t m = 0, p = 0;
    while (p < len) {
        long x = 0; int k = 0, more = 1;
        while (more) {
            if (p >= len) return -1;
            char c = s[p] - 48;
            x |= (long)(c & 0x1f) << (5 * k);
            more = c & 0x20; p++; k++;
            if (!more && (c & 0x10)) x |= (long)(~0UL << (5 * k));
        }
        if (m > 2) x += (long)cnts[m - 2];
        if (m >= cap) return -2;
        cnts[m++] = (uint32_t)x;
    }
    return m;
}

ORC_API int64_t orc_rle_counts_to_string(const uint32_t *cnts, int64_t m, char *s, int64_t cap)
{
    int64_t p = 0;
    for (int64_t i = 0; i < m; ++i) {
        long x = (long)cnts[i];
        if (i > 2) x -= (long)cnts[i - 2];
        int more = 1;
        while (more) {
            char c = x & 0x1f; x >>= 5;
            more = (c & 0x10) ? x != -1 : x != 0;
            if (more) c |= 0x20;
            c += 48;
            if (p >= cap) return -2;
            s[p++] = c;
        }
    }
    return p;
}

/* runs alternate 0,1,0,... over the column-major (h=W_img, w=H_img) array,
 * i.e. over the row-major (H_img, W_img) image.  out: total bytes of 0/1. */
ORC_API int orc_rle_to_dense(const uint32_t *cnts, int64_t m, int64_t total, uint8_t *out)
{
    int64_t p = 0; uint8_t v = 0;
    for (int64_t i = 0; i < m; ++i) {
        int64_t c = cnts[i];
        if (p + c > total) return -1;
        memset(out + p, v, (size_t)c);
        p += c; v = !v;
    }
    if (p != total) return -1;
    return 0;
}

ORC_API int64_t orc_dense_to_rle(const uint8_t *img, int64_t total, uint32_t *cnts, int64_t cap)
{
    int64_t m = 0; uint8_t v = 0; uint32_t c = 0;
    for (int64_t i = 0; i < total; ++i) {
        uint8_t b = img[i] != 0;
        if (b != v) { if (m >= cap) return -2; cnts[m++] = c; c = 0; v = b; }
        c++;
    }
    if (m >= cap) return -2;
    cnts[m++] = c;
    return m;
}

/* ------------------------------------------------------------------ */
/* a3: cv2.erode(mask, ones(3,3)) (2d_to_3d.py:526-527): anchor centre,
 * BORDER_CONSTANT with the morphology default border value, so out-of-image
 * neighbours never lower the minimum.  parity unpinned (opencv not in the
 * reference checkout).  in/out: (H, W) uint8 image layout. */
ORC_API void orc_erode3x3(const uint8_t *in, int H, int W, uint8_t *out)
{
    for (int y = 0; y < H; ++y) {
        int y0 = y > 0 ? y - 1 : 0, y1 = y < H - 1 ? y + 1 : H - 1;
        for (int x = 0; x < W; ++x) {
            int x0 = x > 0 ? x - 1 : 0, x1 = x < W - 1 ? x + 1 : W - 1;
            uint8_t m = 255;
            for (int yy = y0; yy <= y1; ++yy)
                for (int xx = x0; xx <= x1; ++xx) {
                    uint8_t v = in[(int64_t)yy * W + xx];
                    if (v < m) m = v;
                }
            out[(int64_t)y * W + x] = m;
        }
    }
}

/* ------------------------------------------------------------------ */
/* a4-a8: per-mask projection + in-mask test, executed the way the reference
 * does: the whole cloud, once per mask (2d_to_3d.py:553-620).
 * pts: (N,4) rows x,y,z,i (global frame).  eroded: (H,W) uint8 image, the
 * reference indexes its (W,H) transpose as [floor(u), floor(v)] (:544,610).
 * out_idx: ascending point indices (track_points, :606,617).  Returns M. */
ORC_API int64_t orc_points_in_mask(const float *pts, int64_t N, const float *cam,
                                   const uint8_t *eroded, int W, int H,
                                   float min_dist, int32_t *out_idx, float *scratch)
{
    /* scratch: 3*N floats, plays the role of torch.clone (:553) */
    float *X = scratch, *Y = scratch + N, *Z = scratch + 2 * N;
    for (int64_t i = 0; i < N; ++i) { X[i] = pts[4 * i]; Y[i] = pts[4 * i + 1]; Z[i] = pts[4 * i + 2]; }
    int stages = (int)cam[54], flags = (int)cam[55];
    for (int s = 0; s < stages; ++s) {
        const float *t = cam + 15 * s, *R = cam + 15 * s + 3, *tp = cam + 15 * s + 12;
        /* translate (pcd.py:159-165): row-wise add */
        if (flags & (1 << (2 * s))) {
            for (int64_t i = 0; i < N; ++i) X[i] = X[i] + t[0];
            for (int64_t i = 0; i < N; ++i) Y[i] = Y[i] + t[1];
            for (int64_t i = 0; i < N; ++i) Z[i] = Z[i] + t[2];
        }
        /* rotate (pcd.py:167-172): torch.matmul(3x3, 3xN) */
        for (int64_t i = 0; i < N; ++i) {
            float x = X[i], y = Y[i], z = Z[i], o[3];
            for (int r = 0; r < 3; ++r) {
                float acc = R[3 * r] * x;
                acc = fmaf(R[3 * r + 1], y, acc);
                acc = fmaf(R[3 * r + 2], z, acc);
                o[r] = acc;
            }
            X[i] = o[0]; Y[i] = o[1]; Z[i] = o[2];
        }
        /* KITTI: the translation is the 4th term of the (n x 4) @ (4 x 3) product (kitti_utils.py:224-230):
         * fma(1, t, acc) == acc + t */
        if (flags & (2 << (2 * s))) {
            for (int64_t i = 0; i < N; ++i) X[i] = X[i] + tp[0];
            for (int64_t i = 0; i < N; ++i) Y[i] = Y[i] + tp[1];
            for (int64_t i = 0; i < N; ++i) Z[i] = Z[i] + tp[2];
        }
    }
    const float *K = cam + CAM_K;
    int64_t M = 0;
    const float wlim = (float)(W - 1), hlim = (float)(H - 1);
    for (int64_t i = 0; i < N; ++i) {
        float x = X[i], y = Y[i], z = Z[i];
        /* view_points (pcd.py:262-284): viewpad(4x4) @ [p;1], rows 0..2 */
        float h[3];
        for (int r = 0; r < 3; ++r) {
            float acc = K[3 * r] * x;
            acc = fmaf(K[3 * r + 1], y, acc);
            acc = fmaf(K[3 * r + 2], z, acc);
            acc = fmaf(0.0f, 1.0f, acc);
            h[r] = acc;
        }
        float u = h[0] / h[2], v = h[1] / h[2];
        /* :597-603 */
        if (!(z > min_dist && u > 0.0f && u < wlim && v > 0.0f && v < hlim)) continue;
        /* :605-613 incl. the truthiness quirk (floor(u)!=0 && floor(v)!=0) */
        int64_t iu = (int64_t)floorf(u), iv = (int64_t)floorf(v);
        float w = h[2] / h[2];
        int64_t iw = (int64_t)floorf(w);
        if (iu != 0 && iv != 0 && iw != 0 && eroded[iv * W + iu]) out_idx[M++] = (int32_t)i;
    }
    return M;
}


/* a4-a5 only: camera-frame depth and pixel coordinates of every point (used to pin the
 * transform chain against the reference's translate/rotate/view_points, golden G1).
 * out: (N,3) rows u, v, depth. */
ORC_API void orc_project_points(const float *pts, int64_t N, const float *cam, float *out)
{
    int stages = (int)cam[54], flags = (int)cam[55];
    const float *K = cam + CAM_K;
    for (int64_t i = 0; i < N; ++i) {
        float x = pts[4 * i], y = pts[4 * i + 1], z = pts[4 * i + 2];
        for (int s = 0; s < stages; ++s) {
            const float *t = cam + 15 * s, *R = cam + 15 * s + 3, *tp = cam + 15 * s + 12;
            if (flags & (1 << (2 * s))) { x = x + t[0]; y = y + t[1]; z = z + t[2]; }
            float o[3];
            for (int r = 0; r < 3; ++r) {
                float acc = R[3 * r] * x;
                acc = fmaf(R[3 * r + 1], y, acc);
                acc = fmaf(R[3 * r + 2], z, acc);
                o[r] = acc;
            }
            x = o[0]; y = o[1]; z = o[2];
            if (flags & (2 << (2 * s))) { x = x + tp[0]; y = y + tp[1]; z = z + tp[2]; }
        }
        float h[3];
        for (int r = 0; r < 3; ++r) {
            float acc = K[3 * r] * x;
            acc = fmaf(K[3 * r + 1], y, acc);
            acc = fmaf(K[3 * r + 2], z, acc);
            acc = fmaf(0.0f, 1.0f, acc);
            h[r] = acc;
        }
        out[3 * i] = h[0] / h[2]; out[3 * i + 1] = h[1] / h[2]; out[3 * i + 2] = z;
    }
}

/* ------------------------------------------------------------------ */
/* a9: get_medoid (2d_to_3d.py:116-119): argmin(cdist(P,P).sum(0)).
 * pts (N,4); idx: M indices into pts.  colsum (optional, M floats) receives
 * the per-column sums in this oracle's fixed order (sequential over rows i).
 * torch.cdist: both sides <=25 rows -> direct; otherwise matmul expansion. */
ORC_API int64_t orc_medoid(const float *pts, const int32_t *idx, int64_t M, float *colsum)
{
    if (M <= 0) return -1;
    float *s = (float *)calloc((size_t)M, sizeof(float));
    float *q = (float *)malloc((size_t)M * 4 * sizeof(float));
    for (int64_t j = 0; j < M; ++j) {
        const float *p = pts + 4 * (int64_t)idx[j];
        q[4 * j] = p[0]; q[4 * j + 1] = p[1]; q[4 * j + 2] = p[2];
        q[4 * j + 3] = (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2];
    }
    if (M <= 25) {
        for (int64_t i = 0; i < M; ++i)
            for (int64_t j = 0; j < M; ++j) {
                float agg = 0.0f;
                for (int k = 0; k < 3; ++k) {
                    float d = fabsf(q[4 * i + k] - q[4 * j + k]);
                    agg = fmaf(d, d, agg);
                }
                s[j] = s[j] + sqrtf(agg);
            }
    } else {
        for (int64_t i = 0; i < M; ++i) {
            float ax = -2.0f * q[4 * i], ay = -2.0f * q[4 * i + 1], az = -2.0f * q[4 * i + 2], an = q[4 * i + 3];
            for (int64_t j = 0; j < M; ++j) {
                float acc = ax * q[4 * j];
                acc = fmaf(ay, q[4 * j + 1], acc);
                acc = fmaf(az, q[4 * j + 2], acc);
                acc = fmaf(an, 1.0f, acc);
                acc = fmaf(1.0f, q[4 * j + 3], acc);
                acc = acc > 0.0f ? acc : (acc != acc ? acc : 0.0f);
                s[j] = s[j] + sqrtf(acc);
            }
        }
    }
    int64_t best = 0;
    for (int64_t j = 1; j < M; ++j) {
        /* torch.argmin: first minimum; NaN counts as minimal */
        if (s[best] != s[best]) break;
        if (s[j] < s[best] || s[j] != s[j]) best = j;
    }
    if (colsum) memcpy(colsum, s, (size_t)M * sizeof(float));
    free(s); free(q);
    return best;
}

/* ------------------------------------------------------------------ */
/* a10: lane_yaws_distances_and_coords (2d_to_3d.py:277-302): inputs rounded
 * to float32 (torch.Tensor), scipy cdist in float64 on (x,y), argmin (first
 * minimum) and min per centroid.  cent: (K,3) f32, lane: (L,3) f32 x,y,yaw. */
ORC_API void orc_lane_nn(const float *cent, int64_t K, const float *lane, int64_t L,
                         int32_t *out_j, double *out_dist)
{
    for (int64_t k = 0; k < K; ++k) {
        double cx = (double)cent[3 * k], cy = (double)cent[3 * k + 1];
        double best = INFINITY; int64_t bj = 0;
        for (int64_t j = 0; j < L; ++j) {
            double dx = cx - (double)lane[3 * j], dy = cy - (double)lane[3 * j + 1];
            double d = sqrt(dx * dx + dy * dy);
            if (d < best) { best = d; bj = j; }
        }
        out_j[k] = (int32_t)bj; out_dist[k] = best;
    }
}

/* ------------------------------------------------------------------ */
/* a13: orientation + push_centroid (2d_to_3d.py:164-198, 788-806).
 * centroid: medoid xyz (f32); prior: [w,l,h] from the table (:760);
 * yaw: lane yaw (an f32 value, :295); ego: LIDAR_TOP ego_pose translation
 * (:793-795); is_vehicle: name in {car,truck,bus,construction_vehicle,
 * trailer,barrier} (:763).  out_t[3] translation, out_q[4] wxyz rotation. */
ORC_API void orc_box_assemble(const float *centroid, const double *prior, float yaw,
                              const double *ego, int is_vehicle, double *out_t, double *out_q)
{
    double c[3] = { (double)centroid[0], (double)centroid[1], (double)centroid[2] };
    if (!is_vehicle) {            /* :802-806: identity rotation, raw medoid */
        out_t[0] = c[0]; out_t[1] = c[1]; out_t[2] = c[2];
        out_q[0] = 1.0; out_q[1] = 0.0; out_q[2] = 0.0; out_q[3] = 0.0;
        return;
    }
    /* :788-789: np.cos/np.sin of an np.float32 -> float32 results */
    double cs = (double)cosf(yaw), sn = (double)sinf(yaw);
    /* pyquaternion 0.9.9 Quaternion(matrix=Rz): trace method on M^T */
    double qw, qz;
    if (cs < -cs) { double t = 1.0 - cs - cs + 1.0; double f = 0.5 / sqrt(t); qw = (sn + sn) * f; qz = t * f; }
    else          { double t = 1.0 + cs + cs + 1.0; double f = 0.5 / sqrt(t); qw = t * f; qz = (sn + sn) * f; }
    out_q[0] = qw; out_q[1] = 0.0; out_q[2] = 0.0; out_q[3] = qz;
    /* :173-175: scipy from_quat([w,x,y,z]) reads it as [x,y,z,w] => a rotation
     * about x by phi = 2*atan2(qw, qz); theta = -phi (wrapped to (-pi,pi]). */
    double phi = 2.0 * atan2(qw, qz);
    if (phi > M_PI) phi -= 2.0 * M_PI;
    if (phi <= -M_PI) phi += 2.0 * M_PI;
    double theta = -phi;
    if (theta != theta) theta = 0.5 * M_PI;
    double ex = c[0] - ego[0], ey = c[1] - ego[1];
    double alpha = atan(fabs(ey) / fabs(ex));
    if (ex < 0) { if (ey < 0) alpha = -M_PI + alpha; else alpha = M_PI - alpha; }
    else        { if (ey < 0) alpha = -alpha; }
    double l = prior[0], w = prior[1];   /* names as in :169-170 */
    double o1 = fabs(w / (2.0 * sin(theta - alpha)));
    double o2 = fabs(l / (2.0 * cos(theta - alpha)));
    double off = o1 < o2 ? o1 : o2;
    if (o1 != o1 || o2 != o2) off = NAN;  /* np.min propagates NaN */
    out_t[0] = c[0] + off * cos(alpha);
    out_t[1] = c[1] + off * sin(alpha);
    out_t[2] = c[2];
}

/* ------------------------------------------------------------------ */
/* a15: circle_nms (2d_to_3d.py:309-332) with the driver's per-label squared
 * thresholds (:850-861).  Order = descending score; ties broken by DESCENDING
 * original index (a reversed stable ascending sort) -- a deliberate pin, the
 * reference's reversed unstable argsort leaves tie order undefined.
 * keep[n] gets 0/1; returns number kept. */
ORC_API int orc_circle_nms(const double *x, const double *y, const double *score,
                           const int32_t *label, int n, const double *thr_by_label, uint8_t *keep)
{
    int *order = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    uint8_t *sup = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int i = 0; i < n; ++i) order[i] = i;
    /* insertion sort: descending score, then descending index */
    for (int a = 1; a < n; ++a) {
        int v = order[a], b = a - 1;
        while (b >= 0 && (score[order[b]] < score[v] || (score[order[b]] == score[v] && order[b] < v))) {
            order[b + 1] = order[b]; --b;
        }
        order[b + 1] = v;
    }
    int kept = 0;
    memset(keep, 0, (size_t)n);
    for (int _i = 0; _i < n; ++_i) {
        int i = order[_i];
        if (sup[i]) continue;
        keep[i] = 1; kept++;
        for (int _j = _i + 1; _j < n; ++_j) {
            int j = order[_j];
            if (sup[j]) continue;
            double dx = x[i] - x[j], dy = y[i] - y[j];
            double dist = dx * dx + dy * dy;
            if (dist <= thr_by_label[label[j]] && label[j] == label[i]) sup[j] = 1;
        }
    }
    free(order); free(sup);
    return kept;
}

/* ------------------------------------------------------------------ */
/* a17 Waymo deltas (reference src/waymo/2d_to_3d.py).
 * orc_centroid_transform: medoid (vehicle frame) -> global, float32 rotate then translate (:684-690).
 * orc_box_assemble_waymo: global float32 centroid -> vehicle frame with inv(float32 pose) in float64
 * (:812-816), push_centroid(ego_frame=True) with the GLOBAL lane yaw (:995, :175-205), heading =
 * as_euler('xyz')[2] of R_inv . Rz(yaw) (:978-1001); other classes: raw centroid, heading 0. */
ORC_API void orc_centroid_transform(const float *c, const float *pose_rt, float *out)
{
    float o[3];
    for (int r = 0; r < 3; ++r) {
        float acc = pose_rt[3 * r] * c[0];
        acc = fmaf(pose_rt[3 * r + 1], c[1], acc);
        acc = fmaf(pose_rt[3 * r + 2], c[2], acc);
        o[r] = acc;
    }
    for (int r = 0; r < 3; ++r) out[r] = o[r] + pose_rt[9 + r];
}

ORC_API void orc_box_assemble_waymo(const float *centroid_global, const float *pose_inv, const double *prior, float yaw,
                                    int is_vehicle, double *out_t, double *out_heading)
{
    double g[3] = { (double)centroid_global[0], (double)centroid_global[1], (double)centroid_global[2] };
    double c[3];
    for (int r = 0; r < 3; ++r)
        c[r] = (double)pose_inv[4 * r] * g[0] + (double)pose_inv[4 * r + 1] * g[1] + (double)pose_inv[4 * r + 2] * g[2] + (double)pose_inv[4 * r + 3];
    if (!is_vehicle) { out_t[0] = c[0]; out_t[1] = c[1]; out_t[2] = c[2]; *out_heading = 0.0; return; }
    double cs = (double)cosf(yaw), sn = (double)sinf(yaw);
    double qw, qz;
    if (cs < -cs) { double t = 1.0 - cs - cs + 1.0; double f = 0.5 / sqrt(t); qw = (sn + sn) * f; qz = t * f; }
    else          { double t = 1.0 + cs + cs + 1.0; double f = 0.5 / sqrt(t); qw = t * f; qz = (sn + sn) * f; }
    double phi = 2.0 * atan2(qw, qz);
    if (phi > M_PI) phi -= 2.0 * M_PI;
    if (phi <= -M_PI) phi += 2.0 * M_PI;
    double theta = -phi;
    if (theta != theta) theta = 0.5 * M_PI;
    double ex = c[0], ey = c[1];
    double alpha = atan(fabs(ey) / fabs(ex));
    if (ex < 0) { if (ey < 0) alpha = -M_PI + alpha; else alpha = M_PI - alpha; }
    else        { if (ey < 0) alpha = -alpha; }
    double l = prior[0], w = prior[1];
    double o1 = fabs(w / (2.0 * sin(theta - alpha)));
    double o2 = fabs(l / (2.0 * cos(theta - alpha)));
    double off = o1 < o2 ? o1 : o2;
    if (o1 != o1 || o2 != o2) off = NAN;
    out_t[0] = c[0] + off * cos(alpha);
    out_t[1] = c[1] + off * sin(alpha);
    out_t[2] = c[2];
    *out_heading = atan2((double)pose_inv[4] * cs + (double)pose_inv[5] * sn, (double)pose_inv[0] * cs + (double)pose_inv[1] * sn);
}

/* ------------------------------------------------------------------ */
/* f4: box matching of the SAM3D fusion step (src/nuscenes/linear_matching.py:53-121,231-259;
 * src/waymo/linear_matching.py alike).  `match(pred, sam3d, 0.2, TYPE_2D)` is waymo_open_dataset's
 * py_metrics_ops.match with TYPE_HUNGARIAN; its C++ is NOT in the reference checkout (pip dependency
 * waymo-open-dataset-tf, src/nuscenes/requirements.txt) -> PARITY UNPINNED.  Restated from its published
 * behaviour: IoU of the two rotated rectangles in the ground plane (convex polygon intersection), weight =
 * (int)(iou * 1e6) when iou >= the type's threshold else 0, maximum-weight bipartite assignment (Hungarian
 * method on cost = 1e6 - weight, padded to square with zero-weight edges), pairs with zero weight dropped.
 * Where several assignments reach the maximum the third-party solver's choice is not known; this
 * restatement resolves a step's tie by "unassigned column first, then lowest column index".
 * box = cx, cy, length, width, cos(heading), sin(heading). */
static void orc_bev_corners(const double *b, double ox, double oy, double *X, double *Y)
{
    const double hl = b[2] * 0.5, hw = b[3] * 0.5, c = b[4], s = b[5];
    const double dx = b[0] - ox, dy = b[1] - oy;
    const double lc = hl * c, ls = hl * s, wc = hw * c, wsn = hw * s;
    X[0] = (dx + lc) - wsn; Y[0] = (dy + ls) + wc;
    X[1] = (dx - lc) - wsn; Y[1] = (dy - ls) + wc;
    X[2] = (dx - lc) + wsn; Y[2] = (dy - ls) - wc;
    X[3] = (dx + lc) + wsn; Y[3] = (dy + ls) - wc;
}

ORC_API double orc_bev_iou(const double *a, const double *b)
{
    const double area_a = a[2] * a[3], area_b = b[2] * b[3];
    if (!(area_a > 0.0) || !(area_b > 0.0)) return 0.0;
    {
        const double dx = b[0] - a[0], dy = b[1] - a[1];
        const double ra2 = a[2] * a[2] + a[3] * a[3], rb2 = b[2] * b[2] + b[3] * b[3];
        const double r = 0.5 * (sqrt(ra2) + sqrt(rb2));
        if (dx * dx + dy * dy > r * r) return 0.0;
    }
    double px[12], py[12], qx[12], qy[12], bx[4], by[4];
    orc_bev_corners(a, a[0], a[1], px, py);
    orc_bev_corners(b, a[0], a[1], bx, by);
    int n = 4;
    for (int e = 0; e < 4 && n > 0; ++e) {      /* Sutherland-Hodgman: keep what lies left of edge e of b */
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

#define ORC_KMAX 1000000

/* One sample: pred (P,6), gt (G,6) -> pred_match[P] (gt index or -1), gt_match[G], match_iou[P]; weight_out
 * (P*G int32, optional) receives the quantised weights.  Returns the total weight of the assignment. */
ORC_API int64_t orc_bev_match(const double *pred, int P, const double *gt, int G, double iou_thr, int32_t *pred_match,
                              int32_t *gt_match, double *match_iou, int32_t *weight_out)
{
    for (int i = 0; i < P; ++i) { pred_match[i] = -1; match_iou[i] = 0.0; }
    for (int j = 0; j < G; ++j) gt_match[j] = -1;
    if (P <= 0 || G <= 0) return 0;
    int32_t *W = (int32_t *)malloc(sizeof(int32_t) * (size_t)P * (size_t)G);
    for (int p = 0; p < P; ++p)
        for (int g = 0; g < G; ++g) {
            const double iou = orc_bev_iou(pred + 6 * p, gt + 6 * g);
            W[(size_t)p * G + g] = iou >= iou_thr ? (int32_t)(iou * (double)ORC_KMAX) : 0;
        }
    if (weight_out) memcpy(weight_out, W, sizeof(int32_t) * (size_t)P * (size_t)G);
    const int tr = P > G;
    const int n = tr ? G : P, m = tr ? P : G;
#define ORC_W(i, j) (tr ? W[(size_t)((j) - 1) * G + ((i) - 1)] : W[(size_t)((i) - 1) * G + ((j) - 1)])
    int64_t *u = calloc((size_t)m + 1, sizeof(int64_t)), *v = calloc((size_t)m + 1, sizeof(int64_t));
    int64_t *minv = malloc(sizeof(int64_t) * ((size_t)m + 1));
    int *p = calloc((size_t)m + 1, sizeof(int)), *way = calloc((size_t)m + 1, sizeof(int));
    unsigned char *used = malloc((size_t)m + 1);
    const int64_t INF = (int64_t)1 << 50;
    for (int i = 1; i <= n; ++i) {
        p[0] = i;
        int j0 = 0;
        for (int j = 0; j <= m; ++j) { minv[j] = INF; used[j] = 0; }
        do {
            used[j0] = 1;
            const int i0 = p[j0];
            int64_t delta = INF;
            int j1 = 0;
            for (int j = 1; j <= m; ++j) {
                if (used[j]) continue;
                const int64_t cur = (int64_t)(ORC_KMAX - ORC_W(i0, j)) - u[i0] - v[j];
                if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
                /* ties: an unassigned column first (the search ends there), then the lowest index */
                if (minv[j] < delta || (minv[j] == delta && p[j] == 0 && p[j1] != 0)) { delta = minv[j]; j1 = j; }
            }
            for (int j = 0; j <= m; ++j) {
                if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
                else minv[j] -= delta;
            }
            j0 = j1;
        } while (p[j0] != 0);
        do {
            const int j1 = way[j0];
            p[j0] = p[j1];
            j0 = j1;
        } while (j0);
    }
    int64_t total = 0;
    for (int j = 1; j <= m; ++j) {
        const int i = p[j];
        if (i == 0) continue;
        const int w = ORC_W(i, j);
        if (w <= 0) continue;
        const int pi = tr ? j - 1 : i - 1, gi = tr ? i - 1 : j - 1;
        pred_match[pi] = gi;
        gt_match[gi] = pi;
        match_iou[pi] = orc_bev_iou(pred + 6 * pi, gt + 6 * gi);
        total += w;
    }
#undef ORC_W
    free(W); free(u); free(v); free(minv); free(p); free(way); free(used);
    return total;
}
