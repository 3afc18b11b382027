/*
 * cm3d_hip.h -- C-ABI of libcm3d_hip.so: the MI355X (gfx950) kernels of CM3D's
 * 2D->3D pseudo-label lifting path.
 *
 * The reference has no FFI: its hot path is inline torch/numpy/OpenCV code in
 * src/{nuscenes,waymo,kitti}/2d_to_3d.py.  Each entry point below replaces the
 * block of reference code it cites (paths relative to the reference checkout,
 * src/nuscenes/ unless noted); INTEGRATION.md shows the ctypes stub a maintainer
 * of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in _host;
 *  - the caller owns every buffer; nothing is allocated or freed inside;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*),
 *    never synchronises, and is safe to capture into a hipGraph;
 *  - return value: CM3D_OK or a negative CM3D_ERR_*; nothing throws;
 *  - thread-safe for distinct streams and distinct workspaces; no global state;
 *  - index arithmetic is int32: a batch holds < 2^31 points / mask words.
 *
 * Batch data model ("lift batch"): F frames; frame f owns sweeps
 * [frame_sweep_off[f], frame_sweep_off[f+1]), points [pt_off[f], pt_off[f+1])
 * and masks [mask_off[f], mask_off[f+1]).
 */
#ifndef CM3D_HIP_H
#define CM3D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CM3D_ABI_VERSION 4

#define CM3D_OK 0
#define CM3D_ERR_ARG (-1)      /* null pointer / non-positive size / unsupported shape */
#define CM3D_ERR_LAUNCH (-2)   /* hipGetLastError() != hipSuccess after a launch      */
#define CM3D_ERR_WORKSPACE (-3)/* workspace smaller than cm3d_*_workspace_bytes says  */

#define CM3D_CAM_STRIDE 64       /* floats per camera record, see cm3d_project_hits   */
#define CM3D_SWEEP_XF_STRIDE 24  /* floats per sweep transform, see cm3d_sweep_prep   */
#define CM3D_RAW_QUADS 3         /* raw_stride value that selects the quad layout of the raw rows, see cm3d_sweep_prep (ABI v4) */
#define CM3D_MAX_CAMS 8          /* cameras per frame                                  */
#define CM3D_MAX_MASKS_PER_FRAME 1024
#define CM3D_BBOX_STRIDE 8                 /* int32 per mask in `bbox` (ABI v3): eroded bounds [0..3], stored rectangle [4..7] */
#define CM3D_BOX_STRIDE 10       /* doubles per box record, see cm3d_box_nms           */
#define CM3D_MEDOID_TILE 64      /* columns per medoid tile (one wave)                 */
#define CM3D_MAX_MATCH_BOXES 1024 /* boxes per sample and side, see cm3d_bev_match    */
#define CM3D_MATCH_BOX_STRIDE 6  /* doubles per box of cm3d_bev_match                  */

/* status word written by kernels (int32[4] in device memory, zero it per batch):
 *  [0] bit0: point capacity overflow (cm3d_sweep_prep), bit1: hit-index capacity
 *      overflow (cm3d_compact_hits), bit2: too many masks in a frame / cams out of range,
 *      bit3: quad layout with a frame that does not start on a quad (cm3d_sweep_project_hits)
 *  [1] total points produced by cm3d_sweep_prep
 *  [2] total hit indices required by cm3d_compact_hits
 *  [3] total medoid tiles */
#define CM3D_STATUS_WORDS 4

typedef void *cm3d_stream_t;

int cm3d_abi_version(void);
const char *cm3d_error_string(int code);

/* Resets the per-pass device state: the status word, hit_count[n_masks] and (when not NULL) the removed-row bits
 * (removed_words = cm3d_removed_words(rows of the batch, n_frames) 32-bit words).  First call of every pass over a batch. */
int64_t cm3d_removed_words(int32_t n_rows, int32_t n_frames);
int cm3d_batch_begin(int32_t *status, int32_t *hit_count, int32_t n_masks, uint32_t *removed_bits, int64_t removed_words,
                     cm3d_stream_t stream);

/* ---- a2: sweep preparation -------------------------------------------------
 * Replaces 2d_to_3d.py:437-465 + utils/pcd.py:159-172,246-257: strip a raw sweep to 4 columns,
 * sensor->ego->global (rotate then translate, twice), sweeps of a frame back to back; one streaming pass.
 * The ego-box rows (|x|<halfw && |y|<halfw, halfw = f32(sqrt(2.3)), :442-445) are NOT compacted away:
 * they are written as NaN points (inert downstream) and marked in a bit set, and cm3d_compact_hits turns row
 * indices into indices of the reference's compacted cloud.  So point p of frame f sits at row pt_off[f]+p
 * of its raw rows.
 *  raw          float[rows][raw_stride]   all sweeps of the batch, back to back (raw_stride >= 4: the .bin files' rows), or
 *                                         raw_stride == CM3D_RAW_QUADS (ABI v4): the QUAD layout, float[ceil(rows/4)][3][4] --
 *                                         rows 4q..4q+3 of the batch as x0 x1 x2 x3 y0 y1 y2 y3 z0 z1 z2 z3, 16-byte aligned.
 *                                         12 bytes per row cross HBM (the path never reads the .bin files' ring index, and the
 *                                         intensity only for the points it lists), a lane's four rows are three 16-byte loads.
 *                                         In this layout every frame's first row is a multiple of 4: the loader pads a frame
 *                                         with up to 3 rows of NaN coordinates, which belong to the frame's last sweep and are
 *                                         rows like any other (never dropped, in no mask, at the end of the frame, so no index
 *                                         of a real row moves); status bit 8 reports a frame that starts inside a quad
 *  intensity    float[rows] or NULL       quad layout only: the rows' fourth column (goes into points / hit_xyz; NULL: zeros)
 *  sweep_row_off int32[S+1]               row offsets of the sweeps into raw
 *  sweep_xf     float[S][24]              [0..8] R_cs, [9..11] t_cs, [12..20] R_ego, [21..23] t_ego (float32,
 *                                         exactly the tensors the reference passes to rotate/translate)
 *  frame_sweep_off int32[F+1]
 *  points       float[pt_cap][4]  OUT     x,y,z,intensity in the global frame at the raw row index (pt_cap >= rows)
 *  pt_off       int32[F+1]        OUT     first row of each frame (= sweep_row_off[frame_sweep_off[f]])
 *  removed_bits uint32[cm3d_removed_words(rows, F)] IN/OUT  one bit per dropped row; must be zero on entry
 *                                         (cm3d_batch_begin).  Frame f's bits start at word (pt_off[f] >> 5) + 8 f,
 *                                         bit (r & 31) of word r >> 5 for its frame-local row r */
int cm3d_sweep_prep(const float *raw, int32_t raw_stride, const float *intensity, const int32_t *sweep_row_off, int32_t n_sweeps,
                    int32_t max_rows_per_sweep, const float *sweep_xf, const int32_t *frame_sweep_off,
                    int32_t n_frames, float halfw, float *points, int32_t pt_cap, int32_t *pt_off,
                    uint32_t *removed_bits, int32_t *status, cm3d_stream_t stream);

/* ---- a1: COCO-RLE expansion ------------------------------------------------
 * Replaces pycocotools.mask.decode at 2d_to_3d.py:425 (+ the transpose at :428): run
 * lengths -> dense uint8 [n][H][W] image-layout masks of 0/1.
 *  rle_counts uint32[]  run lengths of all masks back to back (alternating 0-run,1-run,...)
 *  rle_off    int32[n+1]
 *  workspace: cm3d_rle_workspace_bytes(total_runs) */
int64_t cm3d_rle_workspace_bytes(int32_t total_runs);
int cm3d_rle_to_dense(const uint32_t *rle_counts, const int32_t *rle_off, int32_t n_masks, int32_t total_runs,
                      int32_t W, int32_t H, uint8_t *dense, void *workspace, int64_t workspace_bytes,
                      cm3d_stream_t stream);

/* ---- a3: 3x3 erosion + bit-packing ------------------------------------------
 * Replaces cv2.erode(mask, ones((3,3))) + astype(bool) + transpose + H2D at
 * 2d_to_3d.py:526-527,542-544.  Out-of-image neighbours are ignored.
 *  dense   uint8[n][H][W]   non-zero = set
 *  packed  uint32[n][H*Wp]  OUT, Wp = (W+31)/32: one slot of H*Wp words per mask.  A mask's eroded bits are stored as the rows of a
 *          RECTANGLE of whole words (ABI v3): word columns xw0 .. xw0+wc-1, image rows y0 .. y0+rows-1, row after row from the start
 *          of the slot, wc words each: bit (x&31) of slot word (y - y0) * wc + (x>>5) - xw0 = eroded pixel (x,y).  cm3d_erode_pack
 *          stores the whole image (xw0 = y0 = 0, wc = Wp, rows = H: the layout of ABI v2); cm3d_rle_erode_pack the rectangle of the
 *          mask's set pixels, so that a mask's words are contiguous in memory.  Words outside the rectangle are unspecified, and
 *          n*H*Wp must stay below 2^31 (mask offsets are signed 32-bit word numbers).
 *  bbox    int32[n][CM3D_BBOX_STRIDE] OUT: [0..3] x0,y0,x1,y1 inclusive bounds of the eroded mask (x0>x1 when empty),
 *          [4..7] xw0, y0, wc, rows of the stored rectangle (all 0 for a mask without a set pixel) */
int cm3d_erode_pack(const uint8_t *dense, int32_t n_masks, int32_t W, int32_t H, uint32_t *packed,
                    int32_t *bbox, cm3d_stream_t stream);

/* f1: same result straight from run lengths, no dense intermediate
 * (fuses 2d_to_3d.py:425 with :526-527,542-544).  Stores the rectangle of the mask's set pixels (see `packed` above): only words
 * that can hold an eroded pixel are written. */
int cm3d_rle_erode_pack(const uint32_t *rle_counts, const int32_t *rle_off, int32_t n_masks, int32_t total_runs,
                        int32_t W, int32_t H, uint32_t *packed, int32_t *bbox, void *workspace,
                        int64_t workspace_bytes, cm3d_stream_t stream);

/* The same launch carrying the per-pass reset of cm3d_batch_begin (status word, hit_count[n_count_masks], removed-row bits): the mask stage is the
 * first stage of a pass over resident run lengths, and a launch of its own for the reset cost the pass a launch boundary (r04).  Only for passes
 * whose FIRST call is this one (cm3d_sweep_prep, which writes status and removed bits, must not have run before it in the pass). */
int cm3d_rle_erode_pack_begin(const uint32_t *rle_counts, const int32_t *rle_off, int32_t n_masks, int32_t total_runs,
                              int32_t W, int32_t H, uint32_t *packed, int32_t *bbox, void *workspace, int64_t workspace_bytes,
                              int32_t *status, int32_t *hit_count, int32_t n_count_masks, uint32_t *removed_bits, int64_t removed_words,
                              cm3d_stream_t stream);

/* ---- a4-a7: projection + in-image + in-mask test ----------------------------
 * Replaces the per-mask block 2d_to_3d.py:553-613 (clone, 2x translate/rotate,
 * view_points, in-image test, floor, mask gather incl. the floor(u)!=0 && floor(v)!=0
 * quirk) for ALL masks of ALL frames in one pass over the points.
 *  cams  float[F][n_cams][CM3D_CAM_STRIDE]: up to three rigid stages, each `p += t_pre; p = R p; p += t_post`
 *        (either translation optional), then K':
 *        stage s at [15s .. 15s+14] = t_pre(3), R(9, row-major), t_post(3);  [45..53] K' (3x3);
 *        [54] number of stages (2 nuScenes :569-577, 1 Waymo src/waymo/2d_to_3d.py:575-576,
 *        3 KITTI src/kitti/2d_to_3d.py:1238-1240 = ref->velo, velo->ref, ref->rect);
 *        [55] flags: bit 2s = stage s has t_pre, bit 2s+1 = stage s has t_post.
 *        All float32 exactly as the reference hands them to translate/rotate/matmul/view_points.
 *  hit_words uint32[planes][n_points_total] OUT, planes = (max masks per frame + 31)/32;
 *        bit (k&31) of hit_words[k>>5][p] = point p lies in mask mask_off[f]+k.  An intermediate for cm3d_compact_hits:
 *        written per block of 256 consecutive rows of a frame, and only for blocks that hold at least one in-mask point
 *        (the per-block flags travel in the workspace); the words of the other blocks keep what the buffer held before
 *  hit_count int32[n_masks] IN/OUT accumulated with atomics; zeroed by cm3d_batch_begin
 *  workspace: cm3d_project_workspace_bytes(F, max_pts_per_frame, planes), 16-byte aligned; it receives the per-frame
 *        tables and the per-(256-row chunk, mask) hit counts and must be handed unchanged to cm3d_compact_hits
 *  ev_start, ev_stop: optional hipEvent_t (NULL = none), recorded on `stream` right before and after the projection
 *        kernel itself -- behind the small per-frame table kernel the call launches first -- for callers that time it */
int64_t cm3d_project_workspace_bytes(int32_t n_frames, int32_t max_pts_per_frame, int32_t planes);
/* Accounting aid for bench.py's byte counts (nothing on the path calls it; synchronous): rows of the batch that lie in a block of
 * 256 rows with at least one in-mask point, read back from the workspace of the last cm3d_(sweep_)project_hits on `stream`. */
int cm3d_project_hit_rows(const void *workspace, int64_t workspace_bytes, int32_t n_frames, int32_t max_pts_per_frame,
                          int32_t planes, int64_t *rows_out, cm3d_stream_t stream);
int cm3d_project_hits(const float *points, const int32_t *pt_off, int32_t n_frames, int32_t max_pts_per_frame,
                      int32_t n_points_total, const float *cams, int32_t n_cams, const int32_t *mask_off,
                      const int32_t *mask_cam, const int32_t *bbox, const uint32_t *packed, int32_t n_masks,
                      int32_t W, int32_t H, float min_dist, int32_t planes, uint32_t *hit_words,
                      int32_t *hit_count, int32_t *status, void *workspace, int64_t workspace_bytes,
                      void *ev_start, void *ev_stop, cm3d_stream_t stream);

/* a2 + a4-a7 in one launch: cm3d_sweep_prep folded into cm3d_project_hits.  The kernel reads the raw sweep rows,
 * applies the ego-box drop and the sensor -> ego -> global chains (2d_to_3d.py:437-465) on the fly, writes pt_off, the
 * removed-row bits and status[1] exactly as cm3d_sweep_prep does, and produces hit_words / hit_count exactly as
 * cm3d_project_hits does.  `points` is OPTIONAL: NULL = the transformed cloud is not materialised at all (the later stages
 * only need the in-mask points, which cm3d_compact_hits re-derives from the raw rows into hit_xyz); non-NULL = the cloud
 * is also written, bit for bit what cm3d_sweep_prep writes.  max_sweeps_per_frame (host-side knowledge of
 * frame_sweep_off) must be <= 16; use the two separate calls otherwise.  Arguments as in cm3d_sweep_prep and
 * cm3d_project_hits.  Rows of 4 or 5 floats (the reference's .bin layouts) and the quad layout have their own instantiations. */
#define CM3D_MAX_FUSED_SWEEPS 16
/* How much of the chip one projection launch takes (process-wide; r04).  0 (the default): as many workgroups as fit minus one per
 * CU -- the fastest launch when nothing else runs.  n >= 1: n workgroups per CU.  A caller that keeps several batches in flight on
 * streams of their own (cm3d_amd.lifting.LiftPipeline) asks for 2: the launch alone takes 62 instead of 52 us on the headline shape, but
 * the kernels of the other batches run beside it instead of behind it (three batches in flight: +1-2 % frames/s on C2, C1 and C4).
 * Returns the previous value.  Results do not depend on it.  (Honoured by the launch for frames of up to 32 masks on the quad layout, the one the
 * other batches' kernels queue behind; the multi-plane launch keeps its grid.) */
int cm3d_project_workgroups_per_cu(int32_t n);

int cm3d_sweep_project_hits(const float *raw, int32_t raw_stride, const float *intensity, const int32_t *sweep_row_off, int32_t n_sweeps,
                            int32_t max_sweeps_per_frame, const float *sweep_xf, const int32_t *frame_sweep_off,
                            float halfw, float *points, int32_t pt_cap, int32_t *pt_off, uint32_t *removed_bits,
                            int32_t n_frames, int32_t max_pts_per_frame, int32_t n_points_total,
                            const float *cams, int32_t n_cams, const int32_t *mask_off, const int32_t *mask_cam,
                            const int32_t *bbox, const uint32_t *packed, int32_t n_masks, int32_t W, int32_t H,
                            float min_dist, int32_t planes, uint32_t *hit_words, int32_t *hit_count, int32_t *status,
                            void *workspace, int64_t workspace_bytes, void *ev_start, void *ev_stop, cm3d_stream_t stream);

/* ---- a7-a8: ordered compaction of the hits ----------------------------------
 * Replaces torch.where + the two .cpu() index-tracking steps at 2d_to_3d.py:606,613-617.
 *  removed_bits  from cm3d_sweep_prep / cm3d_sweep_project_hits (NULL when the points were not produced by them)
 *  raw, raw_stride, intensity, sweep_xf  the raw rows the fused launch read (NULL when `points` is given)
 *  points   float[pt_cap][4] the transformed cloud, or NULL when it was not materialised
 *  hit_off  int32[n_masks+1] OUT exclusive scan of hit_count
 *  tile_off int32[n_masks+1] OUT exclusive scan of ceil(hit_count/CM3D_MEDOID_TILE)
 *  hit_idx  int32[idx_cap]   OUT ascending point indices of mask m at [hit_off[m], hit_off[m+1]): indices into the
 *                                frame's cloud WITHOUT the dropped rows, i.e. the reference's track_points
 *  hit_row  int32[idx_cap]   OUT, optional: the same points as frame-local row indices
 *  hit_xyz  float[idx_cap][4] OUT, optional: global-frame x,y,z,intensity of every listed point (the gather of :620),
 *                                laid out like hit_idx; from `points` when given, else from the raw rows through the
 *                                very fma chains of the sweep preparation (bit-identical)
 *  tile_work OUT, optional (may be NULL): cm3d_tile_work_bytes(n_masks, idx_cap) bytes; the work list of
 *            cm3d_medoid (one record per medoid tile, longest lists first), built beside the compaction
 *  workspace: the buffer cm3d_project_hits / cm3d_sweep_project_hits filled (per wave-chunk: hit counts per mask, whether it
 *            holds a hit at all, dropped rows; their sums over groups of 16 wave-chunks; in-mask points per frame)
 * Two launches, no scan kernel: the compaction computes its own output offsets from those counts, never reads the hit words of a
 * wave-chunk without a hit, builds hit_off / tile_off / tile_work in the same launch, and leaves (row, sweep) tags in hit_xyz;
 * a one-thread-per-point launch then fetches and transforms the listed rows (all gathers in flight at once). */
int cm3d_compact_hits(const uint32_t *hit_words, int32_t planes, int32_t n_frames, int32_t max_pts_per_frame,
                      int32_t n_points_total, const int32_t *mask_off, int32_t n_masks, const int32_t *hit_count,
                      const uint32_t *removed_bits, const float *raw, int32_t raw_stride, const float *intensity, const float *sweep_xf,
                      const float *points, int32_t *hit_off, int32_t *tile_off, int32_t *hit_idx, int32_t *hit_row,
                      float *hit_xyz, int32_t idx_cap, int32_t *tile_work, int32_t *status, void *workspace,
                      int64_t workspace_bytes, cm3d_stream_t stream);
int64_t cm3d_tile_work_bytes(int32_t n_masks, int32_t idx_cap);

/* ---- a9: medoid --------------------------------------------------------------
 * Replaces get_medoid (2d_to_3d.py:116-119) + the gather at :620,645-647:
 * argmin_j sum_i cdist(P,P)[i][j] with torch.cdist's float32 arithmetic (direct form
 * for <=25 points, matmul expansion otherwise), rows summed in ascending i, first minimum.
 *  points / pt_off / mask_frame / hit_row: either the cloud, the frames' first rows, the frame of every mask and the
 *             row-index list of cm3d_compact_hits (gathers go through it) -- or hit_row = NULL and `points` = the hit_xyz
 *             array of cm3d_compact_hits (float[idx_cap][4], laid out like hit_idx; pt_off and mask_frame are then unused
 *             and may be NULL): the in-mask points are then read as contiguous runs
 *  tile_work  the work list cm3d_compact_hits wrote for the same hit_off / tile_off, or NULL (then it is built here,
 *             one more launch)
 *  medoid_pos int32[n_masks]    OUT position in the mask's index list (-1 if the list is empty)
 *  centroid   float[n_masks][3] OUT global-frame xyz of the medoid point
 *  colsum_opt float[idx_cap] OUT, optional (may be NULL): every column sum, laid out like hit_idx.  When NULL, lists of
 *             more than 256 points (in a batch whose longest list has more than 448) are settled in two passes (approximate sums for all columns, exact float32 sums only
 *             for the columns a proven error bound cannot exclude): the same position, less work
 *  workspace: cm3d_medoid_workspace_bytes(n_masks, idx_cap) */
int64_t cm3d_medoid_workspace_bytes(int32_t n_masks, int32_t idx_cap);
int cm3d_medoid(const float *points, const int32_t *pt_off, const int32_t *mask_frame, int32_t n_masks,
                const int32_t *hit_off, const int32_t *tile_off, const int32_t *hit_row, int32_t idx_cap,
                const int32_t *tile_work, int32_t *medoid_pos, float *centroid, float *colsum_opt, void *workspace,
                int64_t workspace_bytes, cm3d_stream_t stream);
/* The same with a hint and a feedback word (ABI v3), for callers that run batch after batch:
 *  flags     bit 0: expect no list of more than 448 points in this batch -- every list then takes the exact one-pass route and the two
 *            launches of the two-pass route that would find nothing to do are not made.  Only ever a matter of time: the one-pass
 *            route is exact for every length.
 *  feedback  optional int32[1] the DEVICE can write (device memory, or page-locked host memory that is mapped into the device's address
 *            space: then the caller reads it without a copy): 1 if this batch holds such a list, else 0, written by the first launch.
 *            cm3d_amd.lifting.LiftEngine feeds it back as the next pass's flag. */
int cm3d_medoid2(const float *points, const int32_t *pt_off, const int32_t *mask_frame, int32_t n_masks,
                 const int32_t *hit_off, const int32_t *tile_off, const int32_t *hit_row, int32_t idx_cap,
                 const int32_t *tile_work, int32_t *medoid_pos, float *centroid, float *colsum_opt, void *workspace,
                 int64_t workspace_bytes, int32_t flags, int32_t *feedback, cm3d_stream_t stream);

/* Diagnostic for the tests: `count` pseudo-random (numerator, denominator) pairs, denominators over the projection
 * kernel's shortcut domain [1e-30, 1e30), quotients next to integers over-represented, through the kernel's division
 * sequence and through the IEEE division; n_bad (device, 2 words): [0] = differing quotients q with 1/8 <= |q| < 2^96
 * (must be 0), [1] = differing quotients with |q| < 1/8 in both forms (numerators below 2^-103; the pixel range test
 * rejects those points whatever the low bits are). */
int cm3d_selftest_div(uint64_t seed, uint64_t count, uint64_t *n_bad, cm3d_stream_t stream);

/* Diagnostic for the tests: the first pass over long medoid lists takes its squared distances from the matrix pipe
 * (three v_mfma_f32_32x32x2_f32 = the reference's five-term fma chain).  2048 waves x tiles_per_wave tiles of 32 x 32 pairs of
 * pseudo-random points at global-frame magnitudes through the MFMAs and through the vector fma chain; *n_bad (device) = values
 * that differ in any bit (must be 0). */
int cm3d_selftest_mfma(uint64_t seed, int32_t tiles_per_wave, uint64_t *n_bad, cm3d_stream_t stream);

/* Diagnostic for the tests: runs every float32 bit pattern in [first_bits, last_bits] (positive values) that lies in
 * the medoid kernel's fast-path domain [1e-30, 1e30) through the kernel's square root, its reference form and sqrtf();
 * *n_bad (device) = number of values on which the three are not bit-identical, *first_bad = smallest such pattern. */
int cm3d_selftest_sqrt(uint32_t first_bits, uint32_t last_bits, uint64_t *n_bad, uint32_t *first_bad, cm3d_stream_t stream);

/* ---- a10: nearest lane point --------------------------------------------------
 * Replaces lane_yaws_distances_and_coords (2d_to_3d.py:277-302): float64 Euclidean
 * distance on (x,y) between float32-rounded centroids and lane points, first minimum.
 *  lane      float[L_total][3]  x,y,yaw (float32-rounded, as torch.Tensor(...) does at :278)
 *  lane_off  int32[T+1]         lane tables back to back
 *  frame_lane int32[F]          table of each frame
 *  lane_idx  int32[n_masks] OUT index into the frame's table (-1 for masks without centroid)
 *  lane_dist double[n_masks] OUT
 *  Exactly the brute-force result (first minimum of the float64 distance); internally the lane points are
 *  binned into a uniform grid per table (cm3d_lane_grid_build) and searched in growing rings, one wave per
 *  centroid; centroids farther than ~10 cells from every lane fall back to a brute-force kernel.
 *  grid: the buffer cm3d_lane_grid_build filled; workspace: cm3d_lane_nn_workspace_bytes(n_masks) bytes, reserved
 *  (centroids the grid search cannot settle are scanned against their whole table by the same wave) */
/* Spatial index of the lane tables (uniform grid per table, points in cell order): built by one
 * workgroup per table; depends only on the lane tables, so a driver may build it on a side stream. */
int64_t cm3d_lane_grid_bytes(int32_t n_tables, int32_t n_lane_points);
int cm3d_lane_grid_build(const float *lane, const int32_t *lane_off, int32_t n_tables, int32_t n_lane_points, void *grid,
                         int64_t grid_bytes, cm3d_stream_t stream);
int64_t cm3d_lane_nn_workspace_bytes(int32_t n_masks);
int cm3d_lane_nn(const float *centroid, const int32_t *medoid_pos, const int32_t *mask_frame, int32_t n_masks,
                 const float *lane, const int32_t *lane_off, const int32_t *frame_lane, int32_t n_tables,
                 int32_t n_lane_points, const void *grid, int32_t *lane_idx, double *lane_dist, void *workspace,
                 int64_t workspace_bytes, cm3d_stream_t stream);

/* ---- a11-a15: box assembly + class-aware circle NMS ----------------------------
 * Replaces stage 2 (2d_to_3d.py:745-817: shape prior, lane-yaw rotation, push_centroid
 * :164-198) and circle_nms (:309-332 with the thresholds of :850-861).
 *  class_id int32[n_masks] index into the class tables; score double[n_masks]
 *  prior_wlh double[n_classes][3]; is_vehicle int32[n_classes] (pushed classes, :763);
 *  nms_group int32[n_classes] NMS label of each class (the class itself for nuScenes, the Waymo type
 *  for Waymo, src/waymo/cfg/prompt_cfg.py:286-297); nms_thr double[] squared-distance threshold per NMS label
 *  ego_xyz double[F][3]   LIDAR_TOP ego_pose translation of each frame (:793-795)            (nuScenes mode)
 *  pose_inv float[F][16]  NULL for nuScenes.  Waymo mode (a17, src/waymo/2d_to_3d.py:812-816,978-1001):
 *        inverse of the float32 frame pose; centroids are global, the kernel maps them back to the vehicle
 *        frame in float64, pushes with ego_frame=True and emits the heading instead of a quaternion.
 *  box   double[n_masks][CM3D_BOX_STRIDE] OUT: tx,ty,tz, qw,qz (rotation = [qw,0,0,qz]; Waymo: heading,0),
 *        lane yaw, lane dist, score, class id, flags -- the fixed-size record that the multi-GPU gather ships
 *  flags int32[n_masks] OUT: bit0 box exists (mask had points), bit1 box survives NMS */
int cm3d_box_nms(const float *centroid, const int32_t *medoid_pos, const int32_t *mask_off, int32_t n_frames,
                 int32_t n_masks, const int32_t *class_id, const double *score, const float *lane,
                 const int32_t *lane_off, const int32_t *frame_lane, const int32_t *lane_idx,
                 const double *lane_dist, const double *prior_wlh, const int32_t *is_vehicle,
                 const int32_t *nms_group, const double *nms_thr, int32_t n_classes, const double *ego_xyz,
                 const float *pose_inv, double *box, int32_t *flags, cm3d_stream_t stream);

/* a17 (Waymo): medoid in the vehicle frame -> global frame, float32 rotate then translate
 * (src/waymo/2d_to_3d.py:684-690).  pose_rt float[F][12]: [0..8] rotation row-major, [9..11] translation. */
int cm3d_centroid_transform(const float *centroid_in, const int32_t *medoid_pos, const int32_t *mask_frame,
                            int32_t n_masks, const float *pose_rt, float *centroid_out, cm3d_stream_t stream);

/* circle_nms alone (2d_to_3d.py:309-332) on float64 centres: boxes [frame_off[f], frame_off[f+1]) form
 * one sample; keep[i] = 1 for the survivors.  Order: descending score, ties by descending index. */
int cm3d_circle_nms(const double *x, const double *y, const double *score, const int32_t *label,
                    const int32_t *frame_off, int32_t n_frames, const double *nms_thr, int32_t n_classes,
                    int32_t *keep, cm3d_stream_t stream);

/* ---- f4: box matching of the SAM3D fusion step ------------------------------------
 * Replaces `match(pred_boxes, sam3d_boxes, 0.2, Type.TYPE_2D)` of src/nuscenes/linear_matching.py:53-121,
 * called per sample at :231-259 (src/waymo/linear_matching.py:251-283 alike): waymo_open_dataset's
 * py_metrics_ops.match with TYPE_HUNGARIAN -- bird's-eye-view IoU of rotated rectangles, quantised to
 * integers (x 1e6), maximum-weight assignment, pairs with IoU < iou_thr dropped.  All samples in one call.
 *  pred  double[n_pred][6]  cx, cy, length, width, cos(heading), sin(heading) -- the float32-rounded box
 *        values of `tf.convert_to_tensor(..., dtype=float)` (:248-249) widened to double; a box with
 *        length*width == 0 is "no box" (:65) and never matches
 *  pred_off int32[F+1], gt_off int32[F+1]   sample f owns pred [pred_off[f], pred_off[f+1]) and
 *        gt (SAM3D) boxes [gt_off[f], gt_off[f+1]); at most CM3D_MAX_MATCH_BOXES per sample and side
 *        (status bit0 is set and the sample left unmatched otherwise)
 *  pair_off int64[F+1]      prefix sums of P_f * G_f; total_pairs = pair_off[F]
 *  pred_match int32[n_pred] OUT: index of the matched gt box inside its sample, or -1
 *  gt_match   int32[n_gt]   OUT: index of the matched prediction inside its sample, or -1
 *  match_iou  double[n_pred] OUT: IoU of the match (0 when unmatched)
 *  status int32[1] (zero it before the call); workspace: cm3d_bev_match_workspace_bytes(total_pairs) */
int64_t cm3d_bev_match_workspace_bytes(int64_t total_pairs);
int cm3d_bev_match(const double *pred, const int32_t *pred_off, int32_t n_pred, const double *gt,
                   const int32_t *gt_off, int32_t n_gt, const int64_t *pair_off, int32_t n_frames,
                   int64_t total_pairs, double iou_thr, int32_t *pred_match, int32_t *gt_match, double *match_iou,
                   int32_t *status, void *workspace, int64_t workspace_bytes, cm3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CM3D_HIP_H */
