/*
 * cm3d_reader.h -- C-ABI of libcm3d_reader.so: the host-side loader that feeds the lifting kernels.
 *
 * Replaces the I/O block of the reference's frame loop (paths relative to the reference checkout, src/nuscenes/):
 *   2d_to_3d.py:422-428   open(<f>_masks.pkl), pickle.load, pycocotools decode of every RLE string
 *   2d_to_3d.py:437-441 + utils/pcd.py:246-257   np.fromfile of every LIDAR sweep (.bin, rows of 5 float32)
 * for a whole batch of frames at once, on a pool of threads, straight into caller-owned (page-locked) buffers laid out
 * like the kernels' inputs (include/cm3d_hip.h: `raw` + `sweep_row_off`, `rle_counts` + `rle_off`).  Pure host code:
 * no HIP call, nothing allocated for the caller, no global state besides the pool behind the handle.
 *
 * Mask files are Python pickles of `[{'size': [W, H], 'counts': bytes}, ...]` (gen_2d_masks_detic.py:468-472,506).  The
 * parser understands the pickle opcodes such a list is written with (protocols 2-5); anything else makes the call return
 * CM3D_RD_ERR_FORMAT with the index of the offending file, and the caller can fall back to its own unpickler for it.
 */
#ifndef CM3D_READER_H
#define CM3D_READER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CM3D_RD_OK 0
#define CM3D_RD_ERR_ARG (-1)
#define CM3D_RD_ERR_IO (-2)       /* open / stat / read failed; *bad_index = the file               */
#define CM3D_RD_ERR_FORMAT (-3)   /* not a list of RLE dicts / malformed RLE string / odd file size */
#define CM3D_RD_ERR_CAPACITY (-4) /* an output buffer is too small; the *needed* sizes are reported  */

typedef struct cm3d_reader cm3d_reader;

/* n_threads <= 0: one per online core (at most 64). */
cm3d_reader *cm3d_reader_open(int32_t n_threads);
void cm3d_reader_close(cm3d_reader *r);
int32_t cm3d_reader_threads(const cm3d_reader *r);

/* All sweeps of a batch: file i holds rows of `stride` float32 (utils/pcd.py:250: reshape((-1, 5))).
 *  raw_out        float[cap_rows][stride]  OUT rows of all files back to back, in file order
 *  sweep_row_off  int32[n_files+1]         OUT row offsets (sweep_row_off[n_files] = total rows)
 * Returns CM3D_RD_ERR_CAPACITY (with sweep_row_off filled in, nothing read) when total rows > cap_rows. */
int cm3d_reader_load_sweeps(cm3d_reader *r, const char *const *paths, int32_t n_files, int32_t stride, float *raw_out,
                            int64_t cap_rows, int32_t *sweep_row_off, int32_t *bad_index);

/* The same files into the QUAD layout of include/cm3d_hip.h (cm3d_sweep_prep; replaces the column strip of utils/pcd.py:250-257 on the way):
 * x, y, z of batch rows 4q..4q+3 side by side, every frame padded to a multiple of 4 rows with NaN rows that belong to its last sweep.
 *  file_stride     floats per row in the files (5 for nuScenes)
 *  frame_sweep_off int32[n_frames+1]        which files make up which frame (frame_sweep_off[n_frames] = n_files)
 *  quads_out       float[cap_rows/4][3][4]  OUT, 16-byte aligned
 *  intensity_out   float[cap_rows] or NULL  OUT the rows' fourth column (NULL: not wanted)
 *  sweep_row_off   int32[n_files+1]         OUT row offsets in the padded numbering
 *  frame_rows      int32[n_frames]          OUT rows of each frame without its padding
 * Returns CM3D_RD_ERR_CAPACITY (offsets filled in, nothing read) when the padded total exceeds cap_rows. */
int cm3d_reader_load_sweeps_quads(cm3d_reader *r, const char *const *paths, int32_t n_files, int32_t file_stride,
                                  const int32_t *frame_sweep_off, int32_t n_frames, float *quads_out, float *intensity_out,
                                  int64_t cap_rows, int32_t *sweep_row_off, int32_t *frame_rows, int32_t *bad_index);

/* All mask files of a batch (one per frame; a NULL or empty path = a frame without masks).
 *  counts_out     uint32[cap_counts]       OUT run lengths of all masks back to back (alternating 0-run, 1-run, ...)
 *  rle_off        int32[cap_masks+1]       OUT run-length offsets per mask
 *  frame_mask_off int32[n_files+1]         OUT mask offsets per frame
 *  mask_wh        int32[cap_masks][2]      OUT the 'size' entry of every mask (W, H)
 *  needed         int64[2]                 OUT total run lengths, total masks (also on CM3D_RD_ERR_CAPACITY)
 * Every mask's run lengths must add up to W*H (checked). */
int cm3d_reader_load_masks(cm3d_reader *r, const char *const *paths, int32_t n_files, uint32_t *counts_out, int64_t cap_counts,
                           int32_t *rle_off, int32_t *frame_mask_off, int32_t *mask_wh, int32_t cap_masks, int64_t *needed,
                           int32_t *bad_index);

/* One COCO compressed RLE string -> run lengths (pycocotools rleFrString).  Returns the number of run lengths, or a
 * negative CM3D_RD_ERR_*; counts_out may be NULL to only count. */
int64_t cm3d_rle_string_to_counts(const uint8_t *s, int64_t len, uint32_t *counts_out, int64_t cap);

/* ---- the rest of the reference's host-side frame loop: tables, per-frame table walk, <f>_data.json, result writer --------
 * Replaces (src/nuscenes/2d_to_3d.py of the reference):
 *   :382       NuScenes(VER_NAME, INPUT_PATH): the scene / sample / sample_data / ego_pose / calibrated_sensor / sensor / log
 *              tables, parsed once (on the reader's pool) into a flat index
 *   :415-441, :489-503  the per-frame walk: key-frame LIDAR_TOP sample_data and its `next` chain with every sweep's
 *              calibrated_sensor + ego_pose (-> float32[24] sweep transform: R_cs, t_cs, R_ego, t_ego as the reference hands
 *              them to rotate / translate), the six cameras' ego_pose + calibrated_sensor (-> float32[64] camera records,
 *              include/cm3d_hip.h CM3D_CAM_STRIDE), the LIDAR_TOP ego translation (push_centroid's origin, :793-795)
 *   :422-423   the frame's file names and json.load of <f>_data.json (labels -> class index after the renames of :122-132,
 *              detection_scores, cam_nums)
 *   :929-930   json.dump of {"meta": ..., "results": {token: [box, ...]}}
 * Quaternion -> matrix and the float64 -> float32 casts follow cm3d_amd/geometry.py operation for operation, so the records
 * are bit-identical to the Python path's (tests/test_reader.py). */
typedef struct cm3d_tables cm3d_tables;
typedef struct cm3d_manifest cm3d_manifest;

cm3d_tables *cm3d_tables_open(cm3d_reader *r, const char *dataroot, const char *version, int32_t *err);
void cm3d_tables_close(cm3d_tables *t);
int32_t cm3d_tables_scene_samples(const cm3d_tables *t, const char *scene_name);                 /* < 0: unknown scene */
int32_t cm3d_tables_scene_location(const cm3d_tables *t, const char *scene_name, char *out, int32_t cap);
int64_t cm3d_tables_scene_names(const cm3d_tables *t, char *out, int64_t cap);                   /* NUL-separated; returns bytes needed */
int64_t cm3d_tables_job_tokens(const cm3d_tables *t, const char *const *scene_names, int32_t n_scenes, char *out, int64_t cap,
                               int32_t *rows_out, int64_t cap_rows);

/* The walk over the frames of `scene_names` (a batch).  class_names: detection names in class-index order.  On an unreadable
 * frame (missing files without missing_ok, malformed json, a label outside class_names) the manifest is still returned, with
 * *err set and cm3d_manifest_sizes()[5] = the frame: the caller falls back to its own reader for that batch. */
cm3d_manifest *cm3d_tables_manifest(const cm3d_tables *t, cm3d_reader *r, const char *const *scene_names, int32_t n_scenes,
                                    const char *mask_dir, int32_t n_sweeps, double ratio, const char *const *class_names,
                                    int32_t n_classes, int32_t missing_ok, int32_t *err);
void cm3d_manifest_close(cm3d_manifest *m);
/* out[6]: frames, sweeps, masks, bytes of the sweep-path blob, bytes of the mask-path blob, first unreadable frame (-1: none) */
void cm3d_manifest_sizes(const cm3d_manifest *m, int64_t *out);
const char *cm3d_manifest_bad_label(const cm3d_manifest *m);
/* arrays sized by cm3d_manifest_sizes; any pointer may be NULL:
 *  sample_index int32[F] (row in sample.json), frame_sweep_off int32[F+1], sweep_xf float[S][24], cams float[F][6][64],
 *  ego_xyz double[F][3], frame_mask_off int32[F+1], mask_cam int32[M], class_id int32[M], score double[M],
 *  sweep_paths / mask_paths: NUL-separated with their offset arrays int32[S+1] / int32[F+1] */
int cm3d_manifest_copy(const cm3d_manifest *m, int32_t *sample_index, int32_t *frame_sweep_off, float *sweep_xf, float *cams,
                       double *ego_xyz, int32_t *frame_mask_off, int32_t *mask_cam, int32_t *class_id, double *score,
                       char *sweep_paths, int32_t *sweep_path_off, char *mask_paths, int32_t *mask_path_off);
/* cm3d_reader_load_sweeps / cm3d_reader_load_masks on the manifest's own file lists */
int cm3d_manifest_load_sweeps(cm3d_reader *r, const cm3d_manifest *m, int32_t stride, float *raw_out, int64_t cap_rows,
                              int32_t *sweep_row_off, int32_t *bad_index);
int cm3d_manifest_load_sweeps_quads(cm3d_reader *r, const cm3d_manifest *m, int32_t file_stride, float *quads_out, float *intensity_out,
                                    int64_t cap_rows, int32_t *sweep_row_off, int32_t *frame_rows, int32_t *bad_index);
int cm3d_manifest_load_masks(cm3d_reader *r, const cm3d_manifest *m, uint32_t *counts_out, int64_t cap_counts, int32_t *rle_off,
                             int32_t *frame_mask_off, int32_t *mask_wh, int32_t cap_masks, int64_t *needed, int32_t *bad_index);

/* Result writer (:929-930): records double[n][10] (0-2 translation, 3 qw, 4 qz, 5 index into tokens, 7 score, 8 class), tokens =
 * n_tokens JSON-quoted sample tokens, NUL-separated, in output order; per class the constant text pieces around the numbers
 * (rendered once by the caller); floats are written as Python's repr() writes them.  prefix = everything up to and including
 * '"results": {'; NULL: only the comma-separated `"token": [...]` entries are written (a caller that streams the file batch by
 * batch joins them itself).  Returns bytes written, or -(bytes needed). */
int64_t cm3d_write_results_json(const double *records, int64_t n, const char *tokens, int32_t n_tokens, const char *const *cls_mid,
                                const char *const *cls_score, const char *const *cls_tail, int32_t n_classes, const char *prefix,
                                char *out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* CM3D_READER_H */
