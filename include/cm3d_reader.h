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

#ifdef __cplusplus
}
#endif
#endif /* CM3D_READER_H */
