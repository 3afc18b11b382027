// a1/a3/f1: mask expansion, 3x3 erosion and bit-packing.
//   reference: pycocotools decode at src/nuscenes/2d_to_3d.py:425, cv2.erode at :526-527,
//   bool/transpose/H2D at :542-544.
// HBM-bound streaming kernels (no MFMA):
//   k_erode_pack      reads n*W*H dense bytes once (+ 2 halo rows per band), writes n*H*Wp*4
//   k_rle_ends + k_rle_to_dense   run lengths -> run ends (KBs) -> n*W*H dense bytes
//   k_rle_erode_pack  streams the run lengths (KBs), writes only the rectangle of packed words that can
//                     hold an eroded pixel (rows/words outside a mask's bbox are never read downstream);
//                     latency-bound, one workgroup per mask
// A tile of packed rows (plus a one-word halo column on each side) lives in LDS; the erosion is
// 9 word reads + shifts per output word; out-of-image neighbours count as set (cv2's border rule).
#include "common.h"
#include <cstdlib>

#define EP_THREADS 256
#define EP_MAX_WP 128          // supports W <= 4096
#define EP_LDS_WORDS 8192      // dense kernel: at most 32 KiB of packed rows per workgroup (dynamic LDS)
#define RLE_LDS_WORDS 4096     // RLE kernel: 16 KiB tiles (as narrow as the mask's rectangle).  The workgroups are latency-bound,
                               // so residency counts: measured on C2 8 KiB 50 us, 16 KiB 45 us, 24 KiB 48 us, 32 KiB 59 us

// one bit per non-zero byte of a 16-byte chunk -> 16 bits
static __device__ __forceinline__ uint32_t pack16(uint4 v)
{
    auto nz4 = [](uint32_t w) -> uint32_t {
        // 0x01 in every non-zero byte, then gather the four flags into a nibble
        uint32_t t = ((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w;
        t = (t >> 7) & 0x01010101u;
        return (t * 0x01020408u) >> 24;   // bit k = byte k != 0
    };
    return (nz4(v.x) & 0xF) | ((nz4(v.y) & 0xF) << 4) | ((nz4(v.z) & 0xF) << 8) | ((nz4(v.w) & 0xF) << 12);
}

// LDS tile: rows+2 rows of lw = wc + 2 words.  LDS row r holds image row ya + r, column c holds packed
// word xw0 - 1 + c of that row.  Erodes LDS rows 1..rows / columns 1..wc, stores them, reduces the bbox.
static __device__ __forceinline__ void erode_tile_store(const uint32_t *s_rows, int lw, int wc, int xw0, int ya, int rows, int W,
                                                         int Wp, uint32_t *__restrict__ out_mask, int32_t *__restrict__ bbox4)
{
    const uint32_t tail_mask = (W & 31) ? ((1u << (W & 31)) - 1u) : 0xFFFFFFFFu;
    int minx = 0x7FFFFFFF, miny = 0x7FFFFFFF, maxx = -1, maxy = -1;
    const int nwords = rows * wc;
    // (r, c) = (q / wc, q % wc) kept incrementally: one division per thread, none in the loop
    int r = (int)threadIdx.x / wc, c = (int)threadIdx.x - r * wc;
    const int dr_step = EP_THREADS / wc, dc_step = EP_THREADS - dr_step * wc;
    for (int q = threadIdx.x; q < nwords; q += EP_THREADS, r += dr_step, c += dc_step) {
        if (c >= wc) { c -= wc; ++r; }               // output row ya + 1 + r, word xw0 + c
        uint32_t e = 0xFFFFFFFFu;
#pragma unroll
        for (int dr = 0; dr < 3; ++dr) {
            const uint32_t *row = s_rows + (r + dr) * lw + c;      // row[0] left, row[1] centre, row[2] right
            const uint32_t ce = row[1];
            const uint32_t left = (ce << 1) | (row[0] >> 31);      // pixel x-1
            const uint32_t right = (ce >> 1) | (row[2] << 31);     // pixel x+1
            e &= ce & left & right;
        }
        const int xw = xw0 + c, y = ya + 1 + r;
        if (xw == Wp - 1) e &= tail_mask;
        out_mask[(size_t)y * Wp + xw] = e;
        if (e) {
            minx = min(minx, xw * 32 + __builtin_ctz(e));
            maxx = max(maxx, xw * 32 + 31 - __builtin_clz(e));
            miny = min(miny, y);
            maxy = max(maxy, y);
        }
    }
    minx = cm3d_wave_min(minx); miny = cm3d_wave_min(miny);
    maxx = cm3d_wave_max(maxx); maxy = cm3d_wave_max(maxy);
    if (cm3d_lane() == 0 && maxx >= 0) {
        atomicMin(&bbox4[0], minx); atomicMin(&bbox4[1], miny);
        atomicMax(&bbox4[2], maxx); atomicMax(&bbox4[3], maxy);
    }
}

__global__ void k_bbox_init(int32_t *__restrict__ bbox, int n, int Wp, int H)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        bbox[CM3D_BBOX_STRIDE * i + 0] = 0x7FFFFFFF; bbox[CM3D_BBOX_STRIDE * i + 1] = 0x7FFFFFFF;
        bbox[CM3D_BBOX_STRIDE * i + 2] = -1;         bbox[CM3D_BBOX_STRIDE * i + 3] = -1;
        // stored rectangle: the whole image, rows of Wp words (k_erode_pack)
        bbox[CM3D_BBOX_STRIDE * i + 4] = 0; bbox[CM3D_BBOX_STRIDE * i + 5] = 0; bbox[CM3D_BBOX_STRIDE * i + 6] = Wp; bbox[CM3D_BBOX_STRIDE * i + 7] = H;
    }
}

// grid (bands, n_masks).  Fast path W % 32 == 0: 16-byte coalesced loads, 16 bits per lane,
// lane pairs merged with one shuffle.  Other widths: byte loads (parity sizes only).
__global__ __launch_bounds__(EP_THREADS) void k_erode_pack(const uint8_t *__restrict__ dense, int W, int H, int Wp,
                                                            int band_rows, uint32_t *__restrict__ packed,
                                                            int32_t *__restrict__ bbox)
{
    extern __shared__ __align__(16) uint32_t s_rows[];
    const int m = blockIdx.y;
    const int y0 = blockIdx.x * band_rows;
    const int rows = min(band_rows, H - y0);
    const uint8_t *img = dense + (size_t)m * W * H;
    const int lrows = rows + 2, lw = Wp + 2;
    for (int r = threadIdx.x; r < lrows; r += EP_THREADS) {        // halo columns lie outside the image
        s_rows[r * lw] = 0xFFFFFFFFu;
        s_rows[r * lw + lw - 1] = 0xFFFFFFFFu;
    }
    if ((W & 31) == 0) {
        const int cpr = W >> 4;                 // 16-byte chunks per row
        const int nchunks = lrows * cpr;        // even, since cpr is even
        // the band (with halo rows) is one contiguous byte range of the image: stream it with
        // 4 independent 16-byte loads in flight per lane
        const int y_lo = y0 - 1;
        for (int q0 = threadIdx.x; q0 < nchunks; q0 += 4 * EP_THREADS) {
            uint4 v[4];
            bool inimg[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = q0 + k * EP_THREADS;
                const int r = q / cpr;
                const int y = y_lo + r;
                inimg[k] = q < nchunks && y >= 0 && y < H;
                v[k] = make_uint4(0, 0, 0, 0);
                if (inimg[k]) v[k] = *reinterpret_cast<const uint4 *>(img + (size_t)y * W + (q - r * cpr) * 16);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = q0 + k * EP_THREADS;
                if (q < nchunks) {          // q and its pair partner q^1 are both < nchunks or both not
                    const int r = q / cpr, cx = q - r * cpr;
                    const uint32_t bits = inimg[k] ? pack16(v[k]) : 0xFFFFu;
                    const uint32_t hi = (uint32_t)__shfl_xor((int)bits, 1, 64);
                    if ((q & 1) == 0) s_rows[r * lw + 1 + (cx >> 1)] = bits | (hi << 16);
                }
            }
        }
    } else {
        const int nwords = lrows * Wp;
        for (int q = threadIdx.x; q < nwords; q += EP_THREADS) {
            const int r = q / Wp, xw = q - r * Wp;
            const int y = y0 - 1 + r;
            uint32_t bits = 0xFFFFFFFFu;
            if (y >= 0 && y < H) {
                bits = 0;
                const uint8_t *p = img + (size_t)y * W + xw * 32;
                const int cnt = min(32, W - xw * 32);
                for (int k = 0; k < cnt; ++k) bits |= (p[k] != 0 ? 1u : 0u) << k;
                if (cnt < 32) bits |= ~((1u << cnt) - 1u);   // beyond the right edge: counts as set
            }
            s_rows[r * lw + 1 + xw] = bits;
        }
    }
    __syncthreads();
    erode_tile_store(s_rows, lw, Wp, 0, y0 - 1, rows, W, Wp, packed + (size_t)m * H * Wp, bbox + CM3D_BBOX_STRIDE * m);
}

static inline int ep_band_rows(int Wp)
{
    int b = EP_LDS_WORDS / (Wp + 2) - 2;
    return b > 62 ? 62 : b;
}

extern "C" int cm3d_erode_pack(const uint8_t *dense, int32_t n_masks, int32_t W, int32_t H, uint32_t *packed,
                               int32_t *bbox, cm3d_stream_t stream)
{
    if (!dense || !packed || !bbox) return CM3D_ERR_ARG;
    if (n_masks <= 0 || W <= 0 || H <= 0 || W > 32 * EP_MAX_WP || W > 32767 || H > 32767) return CM3D_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int Wp = (W + 31) / 32;
    const int band_rows = ep_band_rows(Wp);
    if (band_rows < 1) return CM3D_ERR_ARG;
    const int bands = (H + band_rows - 1) / band_rows;
    hipLaunchKernelGGL(k_bbox_init, dim3((n_masks + 255) / 256), dim3(256), 0, st, bbox, n_masks, (W + 31) / 32, H);
    CM3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_erode_pack, dim3(bands, n_masks), dim3(EP_THREADS), (size_t)(band_rows + 2) * (Wp + 2) * 4, st, dense, W, H,
                       Wp, band_rows, packed, bbox);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}

// ---------------------------------------------------------------------------
// RLE: run lengths -> inclusive run ends (per mask) in the workspace, plus the rectangle of the
// set pixels: mrect[m] = {y_first, y_last, first word, last word} (y_first > y_last when empty).
// 256 threads, 8 consecutive runs per thread; also resets the mask's bbox.
#define RE_THREADS 256
#define RE_PER 8
__global__ __launch_bounds__(RE_THREADS) void k_rle_ends(const uint32_t *__restrict__ cnts, const int32_t *__restrict__ rle_off,
                                                         int W, uint32_t *__restrict__ ends, int4 *__restrict__ mrect,
                                                         int32_t *__restrict__ bbox)
{
    __shared__ int s_w[RE_THREADS / 64];
    __shared__ int s_lo[2], s_hi[2];
    const int m = blockIdx.x;
    const int o = rle_off[m], n = rle_off[m + 1] - o;
    if (threadIdx.x == 0) {
        s_lo[0] = 0x7FFFFFFF; s_lo[1] = 0x7FFFFFFF; s_hi[0] = -1; s_hi[1] = -1;
        if (bbox) { bbox[CM3D_BBOX_STRIDE * m + 0] = 0x7FFFFFFF; bbox[CM3D_BBOX_STRIDE * m + 1] = 0x7FFFFFFF; bbox[CM3D_BBOX_STRIDE * m + 2] = -1; bbox[CM3D_BBOX_STRIDE * m + 3] = -1; }
    }
    int carry = 0;
    int ylo = 0x7FFFFFFF, yhi = -1, xlo = 0x7FFFFFFF, xhi = -1;
    const int wave = threadIdx.x >> 6, lane = cm3d_lane();
    for (int base = 0; base < n; base += RE_THREADS * RE_PER) {
        const int i0 = base + threadIdx.x * RE_PER;
        int v[RE_PER], sum = 0;
#pragma unroll
        for (int q = 0; q < RE_PER; ++q) { v[q] = i0 + q < n ? (int)cnts[o + i0 + q] : 0; sum += v[q]; }
        const int inc = cm3d_wave_incl_scan(sum);
        __syncthreads();
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < RE_THREADS / 64; ++w) { const int c = s_w[w]; if (w < wave) wbase += c; tot += c; }
        int run = carry + wbase + inc - sum;          // start pixel of this thread's first run
#pragma unroll
        for (int q = 0; q < RE_PER; ++q) {
            if (i0 + q < n) {
                const int s = run, e = run + v[q];
                ends[o + i0 + q] = (uint32_t)e;
                if (((i0 + q) & 1) && v[q] > 0) {                     // a 1-run [s, e)
                    const int ys = s / W, ye = (e - 1) / W;
                    ylo = min(ylo, ys); yhi = max(yhi, ye);
                    if (ys == ye) { xlo = min(xlo, s - ys * W); xhi = max(xhi, e - 1 - ys * W); }
                    else { xlo = 0; xhi = W - 1; }
                }
            }
            run += v[q];
        }
        carry += tot;
    }
    ylo = cm3d_wave_min(ylo); xlo = cm3d_wave_min(xlo); yhi = cm3d_wave_max(yhi); xhi = cm3d_wave_max(xhi);
    __syncthreads();
    if (lane == 0 && yhi >= 0) { atomicMin(&s_lo[0], ylo); atomicMin(&s_lo[1], xlo); atomicMax(&s_hi[0], yhi); atomicMax(&s_hi[1], xhi); }
    __syncthreads();
    if (threadIdx.x == 0)
        mrect[m] = s_hi[0] >= 0 ? make_int4(s_lo[0], s_hi[0], s_lo[1] >> 5, s_hi[1] >> 5) : make_int4(1, 0, 1, 0);
}

// first run r in [0,n) with ends[r] > p  (n if none)
static __device__ __forceinline__ int rle_find(const uint32_t *__restrict__ ends, int n, uint32_t p)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (ends[mid] > p) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// a1: each thread produces 16 output bytes.  grid (chunks, n_masks)
__global__ __launch_bounds__(256) void k_rle_to_dense(const uint32_t *__restrict__ ends_all,
                                                      const int32_t *__restrict__ rle_off, int total_px,
                                                      uint8_t *__restrict__ dense)
{
    const int m = blockIdx.y;
    const int o = rle_off[m], n = rle_off[m + 1] - o;
    const uint32_t *ends = ends_all + o;
    uint8_t *out = dense + (size_t)m * total_px;
    for (int q = blockIdx.x * 256 + threadIdx.x; q * 16 < total_px; q += gridDim.x * 256) {
        const uint32_t base = (uint32_t)q * 16u;
        int r = rle_find(ends, n, base);
        uint32_t e = r < n ? ends[r] : 0xFFFFFFFFu;
        uint32_t w[4] = {0, 0, 0, 0};
        const int cnt = min(16, total_px - (int)base);
        for (int k = 0; k < cnt; ++k) {
            uint32_t p = base + k;
            while (r < n && p >= e) { ++r; e = r < n ? ends[r] : 0xFFFFFFFFu; }
            uint32_t v = (r < n) ? (uint32_t)(r & 1) : 0u;
            w[k >> 2] |= v << (8 * (k & 3));
        }
        if (cnt == 16 && (((size_t)m * total_px) & 15) == 0) {
            *reinterpret_cast<uint4 *>(out + base) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (int k = 0; k < cnt; ++k) out[base + k] = (uint8_t)((w[k >> 2] >> (8 * (k & 3))) & 0xFF);
        }
    }
}

// sets the bits of pixels [s, e) (row-major pixel indices, clipped by the caller to the rows held in LDS)
static __device__ __forceinline__ void rle_paint(uint32_t *s_rows, int lw, int xw0, int ya, int W, uint32_t s, uint32_t e)
{
    while (s < e) {
        const uint32_t y = s / W, x = s - y * W;
        const uint32_t xe = min((uint32_t)W, x + (e - s));   // exclusive end within this row
        uint32_t *row = s_rows + ((int)y - ya) * lw + 1 - xw0;    // row[xw] = packed word xw
        const uint32_t w0 = x >> 5, w1 = (xe - 1) >> 5;
        const uint32_t m0 = 0xFFFFFFFFu << (x & 31);
        const uint32_t m1 = 0xFFFFFFFFu >> (31 - ((xe - 1) & 31));
        if (w0 == w1) atomicOr(&row[w0], m0 & m1);
        else {
            atomicOr(&row[w0], m0);
            for (uint32_t w = w0 + 1; w < w1; ++w) atomicOr(&row[w], 0xFFFFFFFFu);
            atomicOr(&row[w1], m1);
        }
        s += xe - x;
    }
}

// One chunk of a mask's run lengths: thread t takes runs [base + 8 t, base + 8 t + 8); returns the start pixel
// of its first run (block-wide exclusive scan on top of `carry`, which is advanced).  Two barriers.
#define RS_PER 8
#define RS_CHUNK (EP_THREADS * RS_PER)
static __device__ __forceinline__ int rle_chunk_scan(const uint32_t *__restrict__ cnts, int n, int base, int (&v)[RS_PER], int *s_w,
                                                     int &carry)
{
    const int i0 = base + (int)threadIdx.x * RS_PER;
    int sum = 0;
#pragma unroll
    for (int q = 0; q < RS_PER; ++q) { v[q] = i0 + q < n ? (int)cnts[i0 + q] : 0; sum += v[q]; }
    const int inc = cm3d_wave_incl_scan(sum);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if (cm3d_lane() == 63) s_w[wave] = inc;
    __syncthreads();
    int wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < EP_THREADS / 64; ++w) { const int c = s_w[w]; if (w < wave) wbase += c; tot += c; }
    const int start = carry + wbase + inc - sum;
    carry += tot;
    return start;
}

// f1: run lengths -> packed tile in LDS -> erode -> store.  ONE workgroup per mask, one launch:
//   pass 1 streams the run lengths (block scan, 2048 runs per step) and finds the rectangle of the set pixels;
//   then, tile by tile over that rectangle (the tile is as wide as the rectangle + halo and as tall as LDS
//   allows, so nearly every mask is a single tile): clear, paint the 1-runs, erode, store, reduce the bbox.
// A mask of up to 2048 runs keeps its runs in registers between the passes; longer lists are streamed again
// per tile.  Nothing outside the rectangle is written; the workgroup owns the mask's bbox (no global atomics).
__global__ __launch_bounds__(EP_THREADS) void k_rle_erode_pack(const uint32_t *__restrict__ cnts_all,
                                                                const int32_t *__restrict__ rle_off, int W, int H, int Wp,
                                                                int lds_words, uint32_t *__restrict__ packed,
                                                                int32_t *__restrict__ bbox)
{
    extern __shared__ __align__(16) uint32_t s_rows[];
    __shared__ int s_w[EP_THREADS / 64];
    __shared__ int s_rect[4];                   // ylo, xlo, yhi, xhi of the set pixels
    __shared__ int s_bb[4];                     // bbox of the eroded pixels
    const int m = blockIdx.x;
    const int o = rle_off[m], n = rle_off[m + 1] - o;
    const uint32_t *cnts = cnts_all + o;
    if (threadIdx.x == 0) {
        s_rect[0] = 0x7FFFFFFF; s_rect[1] = 0x7FFFFFFF; s_rect[2] = -1; s_rect[3] = -1;
        s_bb[0] = 0x7FFFFFFF; s_bb[1] = 0x7FFFFFFF; s_bb[2] = -1; s_bb[3] = -1;
    }
    // ---- pass 1: rectangle
    int v0[RS_PER], start0 = 0;                 // the first chunk stays in registers
    {
        int carry = 0;
        int ylo = 0x7FFFFFFF, yhi = -1, xlo = 0x7FFFFFFF, xhi = -1;
        for (int base = 0; base < n; base += RS_CHUNK) {
            int v[RS_PER];
            int run = rle_chunk_scan(cnts, n, base, v, s_w, carry);
            if (base == 0) {
                start0 = run;
#pragma unroll
                for (int q = 0; q < RS_PER; ++q) v0[q] = v[q];
            }
            const int i0 = base + (int)threadIdx.x * RS_PER;
#pragma unroll
            for (int q = 0; q < RS_PER; ++q) {
                if (((i0 + q) & 1) && v[q] > 0) {                     // a 1-run [s, e)
                    const int s = run, e = run + v[q];
                    const int ys = s / W, ye = (e - 1) / W;
                    ylo = min(ylo, ys); yhi = max(yhi, ye);
                    if (ys == ye) { xlo = min(xlo, s - ys * W); xhi = max(xhi, e - 1 - ys * W); }
                    else { xlo = 0; xhi = W - 1; }
                }
                run += v[q];
            }
        }
        ylo = cm3d_wave_min(ylo); xlo = cm3d_wave_min(xlo); yhi = cm3d_wave_max(yhi); xhi = cm3d_wave_max(xhi);
        __syncthreads();                        // s_rect initialised
        if (cm3d_lane() == 0 && yhi >= 0) {
            atomicMin(&s_rect[0], ylo); atomicMin(&s_rect[1], xlo); atomicMax(&s_rect[2], yhi); atomicMax(&s_rect[3], xhi);
        }
        __syncthreads();
    }
    const int ry0 = s_rect[0], ry1 = min(s_rect[2], H - 1);
    if (s_rect[2] < 0) {                        // empty mask
        if (threadIdx.x == 0) { bbox[CM3D_BBOX_STRIDE * m + 0] = 0x7FFFFFFF; bbox[CM3D_BBOX_STRIDE * m + 1] = 0x7FFFFFFF; bbox[CM3D_BBOX_STRIDE * m + 2] = -1; bbox[CM3D_BBOX_STRIDE * m + 3] = -1; }
        if (threadIdx.x >= 4 && threadIdx.x < 8) bbox[CM3D_BBOX_STRIDE * m + threadIdx.x] = 0;
        return;
    }
    const int xw0 = s_rect[1] >> 5, wc = (min(s_rect[3], W - 1) >> 5) - xw0 + 1, lw = wc + 2;
    int br = lds_words / lw - 2;                // output rows per tile
    br = min(br, ry1 - ry0 + 1);
    const uint32_t pad = (W & 31) ? ~((1u << (W & 31)) - 1u) : 0u;
    // ---- tiles
    for (int y0 = ry0; y0 <= ry1; y0 += br) {
        __syncthreads();                        // the previous tile's readers are done
        const int rows = min(br, ry1 - y0 + 1);
        const int lrows = rows + 2;
        const int ya = y0 - 1;                  // image row of LDS row 0
        // initial tile: ones outside the image, zeros inside; pad bits of a row's last word are ones
        {
            int r = (int)threadIdx.x / lw, c = (int)threadIdx.x - r * lw;
            const int dr_step = EP_THREADS / lw, dc_step = EP_THREADS - dr_step * lw;
            for (int q = threadIdx.x; q < lrows * lw; q += EP_THREADS, r += dr_step, c += dc_step) {
                if (c >= lw) { c -= lw; ++r; }
                const int y = ya + r, xw = xw0 - 1 + c;
                uint32_t v = 0u;
                if (y < 0 || y >= H || xw < 0 || xw >= Wp) v = 0xFFFFFFFFu;
                else if (xw == Wp - 1) v = pad;
                s_rows[q] = v;
            }
        }
        __syncthreads();
        const int yc0 = max(ya, 0), yc1 = min(ya + lrows - 1, H - 1);            // image rows held in LDS
        const uint32_t px0 = (uint32_t)yc0 * W, px1 = (uint32_t)(yc1 + 1) * W;   // pixel range [px0, px1)
        // every 1-run (odd index) overlapping the tile sets its bits (all of them lie inside the word range)
        if (n <= RS_CHUNK) {
            int run = start0;
            const int i0 = (int)threadIdx.x * RS_PER;
#pragma unroll
            for (int q = 0; q < RS_PER; ++q) {
                if (((i0 + q) & 1) && v0[q] > 0) {
                    const uint32_t s = max((uint32_t)run, px0), e = min((uint32_t)(run + v0[q]), px1);
                    rle_paint(s_rows, lw, xw0, ya, W, s, e);
                }
                run += v0[q];
            }
        } else {
            int carry = 0;
            for (int base = 0; base < n; base += RS_CHUNK) {
                int v[RS_PER];
                int run = rle_chunk_scan(cnts, n, base, v, s_w, carry);
                const int i0 = base + (int)threadIdx.x * RS_PER;
#pragma unroll
                for (int q = 0; q < RS_PER; ++q) {
                    if (((i0 + q) & 1) && v[q] > 0) {
                        const uint32_t s = max((uint32_t)run, px0), e = min((uint32_t)(run + v[q]), px1);
                        rle_paint(s_rows, lw, xw0, ya, W, s, e);
                    }
                    run += v[q];
                }
                if ((uint32_t)carry >= px1) break;          // uniform: the rest lies below the tile
            }
        }
        __syncthreads();
        erode_tile_store(s_rows, lw, wc, xw0, ya, rows, W, Wp, packed + (size_t)m * H * Wp, s_bb);
    }
    __syncthreads();
    if (threadIdx.x < 4) bbox[CM3D_BBOX_STRIDE * m + threadIdx.x] = s_bb[threadIdx.x];
    // this form stores image rows as they are: the stored rectangle is the whole image
    if (threadIdx.x >= 4 && threadIdx.x < 8) bbox[CM3D_BBOX_STRIDE * m + threadIdx.x] = threadIdx.x == 6 ? Wp : (threadIdx.x == 7 ? H : 0);
}

// ---------------------------------------------------------------------------
// f1, the form for ordinary masks (a few hundred runs): one workgroup per mask whose WAVES NEVER SYNCHRONISE until the mask's
// bounding box is put together at the very end.  k_rle_erode_pack above gives a mask 256 threads and pays for it with eight
// barriers while three lanes in four have no run to scan (an instance mask of 1600x900 has 20..600 runs).  Here the mask's
// rows are cut into up to four BANDS, one wave each (a mask of up to 128 runs is one band: the other waves leave at once);
// every wave of a mask scans all its runs itself (8 per lane, one wave scan per 512 runs -- a few hundred instructions,
// cheaper than a barrier and a hand-over), paints the runs that touch its band into its own LDS tile, and erodes it with a
// sliding window: a lane owns one word column of a stretch of rows and walks down it, keeping the horizontal AND of the
// previous two rows in registers -- 3 LDS reads per output word instead of 9.  The launch is bound by instruction issue and by
// its longest wave (measured: one wave for a whole 60 000-pixel mask runs 20 us by itself), hence the bands.
// Same outputs bit for bit (tests/test_gpu_golden.py runs both forms against the oracle).
#ifndef RW_WAVES
#define RW_WAVES 4
#endif
#define RW_THREADS (64 * RW_WAVES)
#define RW_PER 8
#define RW_CHUNK (64 * RW_PER)
#define RW_LDS_WORDS 1024      // per wave (measured on C2, one batch at a time: 512 -> 52 us, 1024 -> 39.5, 2048 -> 42, 4096 -> 50)
#define RW_BAND_RUNS 128       // runs per band a mask is cut into (two runs per row: 64 rows)

static __device__ __forceinline__ void rw_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// runs [base + 8 lane, +8) of the mask; returns the start pixel of the lane's first run, advances carry (uniform)
static __device__ __forceinline__ int rw_chunk_scan(const uint32_t *__restrict__ cnts, int n, int base, int lane, int (&v)[RW_PER], int &carry)
{
    const int i0 = base + lane * RW_PER;
    int sum = 0;
    // (r04: the 16-byte loads took only lanes whose runs happen to start on a 16-byte boundary -- a quarter of the masks --, the others eight
    // loads inside eight divergent branches, which the compiler waits for one after the other.  A whole chunk inside the mask -- uniform --
    // is two 16-byte loads per lane from a 4-byte-aligned address, the chip's unaligned mode; the mask's last chunk eight loads of a clamped
    // index, unconditional, so that they are in flight together.)
    typedef uint32_t rw_u4 __attribute__((ext_vector_type(4), aligned(4)));
    if (base + RW_CHUNK <= n) {
        const rw_u4 a = *reinterpret_cast<const rw_u4 *>(cnts + i0), b = *reinterpret_cast<const rw_u4 *>(cnts + i0 + 4);
        v[0] = (int)a.x; v[1] = (int)a.y; v[2] = (int)a.z; v[3] = (int)a.w; v[4] = (int)b.x; v[5] = (int)b.y; v[6] = (int)b.z; v[7] = (int)b.w;
    } else {
#pragma unroll
        for (int q = 0; q < RW_PER; ++q) v[q] = (int)cnts[min(i0 + q, n - 1)];
#pragma unroll
        for (int q = 0; q < RW_PER; ++q) v[q] = i0 + q < n ? v[q] : 0;
    }
#pragma unroll
    for (int q = 0; q < RW_PER; ++q) sum += v[q];
    const int inc = cm3d_wave_incl_scan(sum);
    const int start = carry + inc - sum;
    carry += __builtin_amdgcn_readlane(inc, 63);
    return start;
}

// pixel index -> row: s / W through a float reciprocal and one correction each way (exact: the estimate is off by less
// than one for W * H < 2^31, H < 2^23); the compiler's 32-bit division costs three times the instructions and registers
static __device__ __forceinline__ int rw_row_of(uint32_t s, int W, float rcpW)
{
    int y = (int)((float)s * rcpW);
    y -= (uint32_t)y * (uint32_t)W > s ? 1 : 0;
    y += (uint32_t)(y + 1) * (uint32_t)W <= s ? 1 : 0;
    return y;
}

// the (at most four) 1-runs among a lane's eight runs as (start pixel, length), from the start pixel of its first run
static __device__ __forceinline__ void rw_one_runs(int run, const int (&v)[RW_PER], int (&s1)[RW_PER / 2], int (&l1)[RW_PER / 2])
{
#pragma unroll
    for (int q = 0; q < RW_PER / 2; ++q) {
        s1[q] = run + v[2 * q];
        l1[q] = v[2 * q + 1];
        run += v[2 * q] + v[2 * q + 1];
    }
}

#ifdef CM3D_DIAG
// Diagnostic build only (tools/rw_diag.py): s_memtime at the start and the end of every mask's wave, its placement and its run count
#define RW_DIAG_WAVES 32768
__device__ int g_rw_diag;
__device__ unsigned long long g_rw_wave[4 * RW_DIAG_WAVES];
static __device__ __forceinline__ unsigned long long rw_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
extern "C" int cm3d_rw_diag_set(int flags)
{
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_rw_diag), &flags, sizeof(int)) != hipSuccess) return CM3D_ERR_LAUNCH;
    void *wv = nullptr;
    if (hipGetSymbolAddress(&wv, HIP_SYMBOL(g_rw_wave)) != hipSuccess || hipMemset(wv, 0, sizeof(g_rw_wave)) != hipSuccess) return CM3D_ERR_LAUNCH;
    return hipDeviceSynchronize() == hipSuccess ? CM3D_OK : CM3D_ERR_LAUNCH;
}
extern "C" int cm3d_rw_diag_read_waves(unsigned long long *out_host, int n_waves)
{
    if (n_waves > RW_DIAG_WAVES) return CM3D_ERR_ARG;
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_rw_wave), 4 * (size_t)n_waves * sizeof(unsigned long long)) == hipSuccess ? CM3D_OK : CM3D_ERR_LAUNCH;
}
#endif

struct RwBegin { int32_t *status; int32_t *hit_count; uint32_t *removed_bits; long long removed_words; int n_masks; };

__global__ __launch_bounds__(RW_THREADS, 5) void k_rle_erode_pack_wave(const uint32_t *__restrict__ cnts_all, const int32_t *__restrict__ rle_off,
                                                                        int n_masks, int W, int H, int Wp, int lds_words,
                                                                        uint32_t *__restrict__ packed, int32_t *__restrict__ bbox, int max_bands, int diag,
                                                                        const RwBegin begin)
{
#ifndef CM3D_DIAG
    diag = 0;                                                   // (the ablation switches exist in the diagnostic build only: each was a loop-invariant mask in scalar registers)
#endif
    // cm3d_rle_erode_pack_begin: the per-pass reset (cm3d_batch_begin's: status word, hit counts, removed-row bits) rides on this launch -- the
    // first of a pass -- instead of a launch of its own.  Nothing in this kernel reads or writes those arrays; whatever does runs behind it.
    if (begin.status) {
        const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
        if (i0 < CM3D_STATUS_WORDS) begin.status[i0] = 0;
        for (long long i = i0; i < begin.n_masks; i += step) begin.hit_count[i] = 0;
        for (long long i = i0; i < begin.removed_words; i += step) begin.removed_bits[i] = 0u;
    }
    extern __shared__ __align__(16) uint32_t s_all[];
    __shared__ int s_part[RW_WAVES][4];                         // the bands' shares of the bounding box
    const int lane = cm3d_lane(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // max_bands > 1: one workgroup per mask, wave w its band w; 1: one wave per mask, and a wave goes on to mask + (waves of the
    // launch) when the grid is smaller than the number of masks (the launcher holds it to a few waves per SIMD: see there)
    const int m_first = max_bands > 1 ? (int)blockIdx.x : (int)blockIdx.x * RW_WAVES + wave;
    const int m_stride = max_bands > 1 ? (int)gridDim.x : (int)gridDim.x * RW_WAVES;
    for (int m = m_first; m < n_masks; m += m_stride) {         // (max_bands > 1: the grid covers the masks, one round, uniform over the workgroup)
#ifdef CM3D_DIAG
    const int wdiag = g_rw_diag;
    const unsigned long long t_start = wdiag ? rw_now() : 0ull;
#endif
    const int band = max_bands > 1 ? wave : 0;
    uint32_t *s_rows = s_all + (size_t)wave * lds_words;
    const int o = rle_off[m], n = rle_off[m + 1] - o;
    const uint32_t *cnts = cnts_all + o;
    const int nb = max(1, min(max_bands, (n + RW_BAND_RUNS - 1) / RW_BAND_RUNS));     // bands = waves at work on this mask
    int bminx = 0x7FFFFFFF, bminy = 0x7FFFFFFF, bmaxx = -1, bmaxy = -1;               // bounding box of the band's eroded pixels
    int rect_x = 0, rect_y = 0, rect_w = 0, rect_h = 0;                               // the stored rectangle (uniform; every band finds the same)
    if (band < nb) {
    const float rcpW = 1.0f / (float)W;
    // ---- pass 1: rectangle of the set pixels (the 1-runs of the first chunk stay in registers)
    int s0[RW_PER / 2], l0[RW_PER / 2];
    int ylo = 0x7FFFFFFF, yhi = -1, xlo = 0x7FFFFFFF, xhi = -1;
    int carry0 = 0;                             // pixels covered by the first chunk of runs
    {
        int carry = 0;
#pragma unroll 1
        for (int base = 0; base < n; base += RW_CHUNK) {
            int v[RW_PER], s1[RW_PER / 2], l1[RW_PER / 2];
            const int run = rw_chunk_scan(cnts, n, base, lane, v, carry);
            rw_one_runs(run, v, s1, l1);
            if (base == 0) {
                carry0 = carry;
#pragma unroll
                for (int q = 0; q < RW_PER / 2; ++q) { s0[q] = s1[q]; l0[q] = l1[q]; }
            }
#pragma unroll 1
            for (int q = 0; q < RW_PER / 2; ++q) {
                const int s = s1[0], len = l1[0];
#pragma unroll
                for (int r = 0; r + 1 < RW_PER / 2; ++r) { s1[r] = s1[r + 1]; l1[r] = l1[r + 1]; }
                if (len > 0) {                                          // a 1-run [s, s + len)
                    const int ys = rw_row_of((uint32_t)s, W, rcpW), ye = rw_row_of((uint32_t)(s + len - 1), W, rcpW);
                    ylo = min(ylo, ys); yhi = max(yhi, ye);
                    if (ys == ye) { xlo = min(xlo, s - ys * W); xhi = max(xhi, s + len - 1 - ys * W); }
                    else { xlo = 0; xhi = W - 1; }
                }
            }
        }
        ylo = __builtin_amdgcn_readfirstlane(cm3d_wave_min(ylo)); xlo = __builtin_amdgcn_readfirstlane(cm3d_wave_min(xlo));
        yhi = __builtin_amdgcn_readfirstlane(cm3d_wave_max(yhi)); xhi = __builtin_amdgcn_readfirstlane(cm3d_wave_max(xhi));
    }
    if (!(yhi < 0 || (diag & 8))) {             // (an empty mask: nothing to paint, the box stays empty)
    const int my0 = ylo, my1 = min(yhi, H - 1);                        // rows of the mask's set pixels; this wave's band of them:
    const int bandr = (my1 - my0 + nb) / nb;
    const int ry0 = my0 + band * bandr, ry1 = min(my1, ry0 + bandr - 1);
    const int xw0 = xlo >> 5, wc = (min(xhi, W - 1) >> 5) - xw0 + 1, lw = wc + 2;
    rect_x = xw0; rect_y = my0; rect_w = max(wc, 0); rect_h = max(my1 - my0 + 1, 0);     // (never negative: run lengths that overshoot W*H must not poison the table entries)
    int br = lds_words / lw - 2;                // output rows per tile
    br = min(br, ry1 - ry0 + 1);
    const uint32_t pad = (W & 31) ? ~((1u << (W & 31)) - 1u) : 0u;
    const uint32_t tail_mask = ~pad;
    uint32_t *out_mask = packed + (size_t)m * H * Wp;
    // (bounding box of the eroded pixels: per lane the OR of its column's words and its first / last non-empty row)
    const int nseg = max(1, 64 / wc);           // stretches of rows a tile is cut into (one lane per stretch and word column)
    for (int y0 = ry0; y0 <= ry1; y0 += br) {
        const int rows = min(br, ry1 - y0 + 1);
        const int lrows = rows + 2;
        const int ya = y0 - 1;                  // image row of LDS row 0
        rw_lds_sync();                          // the previous tile's readers are done
        // initial tile: zeros inside the image (16 bytes per lane and step), then the few places that are ones
        {
            const int nq = (lrows * lw + 3) >> 2;
            for (int q = lane; q < nq; q += 64) reinterpret_cast<uint4 *>(s_rows)[q] = make_uint4(0u, 0u, 0u, 0u);
            if (xw0 == 0 || xw0 + wc == Wp || ya < 0 || ya + lrows - 1 >= H) {          // the rectangle touches the image's border (uniform)
                rw_lds_sync();
                if (xw0 == 0) for (int r = lane; r < lrows; r += 64) s_rows[r * lw] = 0xFFFFFFFFu;                       // left of the image
                if (xw0 + wc == Wp) {
                    for (int r = lane; r < lrows; r += 64) { s_rows[r * lw + lw - 1] = 0xFFFFFFFFu; if (pad) s_rows[r * lw + lw - 2] = pad; }
                }
                rw_lds_sync();
                if (ya < 0) for (int c = lane; c < lw; c += 64) s_rows[c] = 0xFFFFFFFFu;                                  // above the image
                if (ya + lrows - 1 >= H) for (int c = lane; c < lw; c += 64) s_rows[(lrows - 1) * lw + c] = 0xFFFFFFFFu;  // below it
            }
        }
        rw_lds_sync();
        const int yc0 = max(ya, 0), yc1 = min(ya + lrows - 1, H - 1);            // image rows held in LDS
        const uint32_t px0 = (uint32_t)yc0 * W, px1 = (uint32_t)(yc1 + 1) * W;   // pixel range [px0, px1)
        {
            int carry = 0;
#pragma unroll 1
            for (int base = 0; base < ((diag & 4) ? 0 : n); base += RW_CHUNK) {
                int s1[RW_PER / 2], l1[RW_PER / 2];
                if (base == 0) {                                    // uniform
#pragma unroll
                    for (int q = 0; q < RW_PER / 2; ++q) { s1[q] = s0[q]; l1[q] = l0[q]; }
                } else {
                    int v[RW_PER];
                    const int run = rw_chunk_scan(cnts, n, base, lane, v, carry);
                    rw_one_runs(run, v, s1, l1);
                }
#pragma unroll 1
                for (int q = 0; q < RW_PER / 2; ++q) {
                    const uint32_t rs = (uint32_t)s1[0], re = rs + (uint32_t)l1[0];
#pragma unroll
                    for (int r = 0; r + 1 < RW_PER / 2; ++r) { s1[r] = s1[r + 1]; l1[r] = l1[r + 1]; }
                    uint32_t ps = max(rs, px0);
                    const uint32_t pe = min(re, px1);
                    while (ps < pe) {                               // (one round per image row the run touches)
                        const int y = rw_row_of(ps, W, rcpW);
                        const uint32_t x = ps - (uint32_t)y * W;
                        const uint32_t xe = min((uint32_t)W, x + (pe - ps));   // exclusive end within this row
                        uint32_t *row = s_rows + (y - ya) * lw + 1 - xw0;      // row[xw] = packed word xw
                        const uint32_t w0 = x >> 5, w1 = (xe - 1) >> 5;
                        const uint32_t m0 = 0xFFFFFFFFu << (x & 31);
                        const uint32_t m1 = 0xFFFFFFFFu >> (31 - ((xe - 1) & 31));
                        if (w0 == w1) atomicOr(&row[w0], m0 & m1);
                        else {
                            atomicOr(&row[w0], m0);
                            for (uint32_t w = w0 + 1; w < w1; ++w) atomicOr(&row[w], 0xFFFFFFFFu);
                            atomicOr(&row[w1], m1);
                        }
                        ps += xe - x;
                    }
                }
                if (base == 0) carry = carry0;                      // pixels covered by the first chunk (pass 1)
                if ((uint32_t)carry >= px1) break;                  // uniform: the rest lies below the tile
            }
        }
        rw_lds_sync();
        // erosion, sliding window down a word column: h(r) = centre & left & right of LDS row r; out(r) = h(r-1) & h(r) & h(r+1)
        const int rps = (rows + nseg - 1) / nseg;           // output rows per band
        for (int cb = 0; cb < ((diag & 2) ? 0 : wc); cb += 64) {
            const int seg = wc >= 64 ? 0 : lane / wc, c = wc >= 64 ? cb + lane : lane - seg * wc;
            const int r0 = seg * rps, r1 = min(rows, r0 + rps);                  // output rows [r0, r1) of the tile (LDS rows r0+1 .. r1)
            if (c < wc && seg < nseg && r0 < r1) {
                const uint32_t *row = s_rows + r0 * lw + c;
                auto hrow = [&](const uint32_t *rw) {
                    const uint32_t ce = rw[1];
                    return ce & ((ce << 1) | (rw[0] >> 31)) & ((ce >> 1) | (rw[2] << 31));
                };
                uint32_t h0 = hrow(row), h1 = hrow(row + lw);
                row += 2 * lw;
                const int xw = xw0 + c;
                const uint32_t keep = xw == Wp - 1 ? tail_mask : 0xFFFFFFFFu;
                // the eroded words go out as PACKED ROWS OF THE RECTANGLE of the mask's set pixels (word columns xw0 .. xw0 + wc - 1, rows
                // my0 .. my1), one row behind the other: consecutive stores fill whole cache lines.  At the image's row stride a mask's
                // 40-byte row pieces were 1.5 M partial-line writes per batch, and with three batches in flight they cost every kernel
                // that streams from HBM beside them: 20 of 142 us per pass (tools/stage_ablate.py), 4 in this form.
                uint32_t *dst = out_mask + (size_t)(ya + 1 + r0 - my0) * wc + (xw - xw0);
                uint32_t colany = 0u;
                int first = 0x7FFFFFFF, last = -1;
                for (int r = r0; r < r1; ++r, row += lw, dst += wc) {
                    const uint32_t h2 = hrow(row);
                    const uint32_t e = h0 & h1 & h2 & keep;
                    h0 = h1; h1 = h2;
                    if (!(diag & 1)) *dst = e;
                    colany |= e;
                    if (e) { first = min(first, r); last = r; }
                }
                if (colany) {
                    bminx = min(bminx, xw * 32 + __builtin_ctz(colany));
                    bmaxx = max(bmaxx, xw * 32 + 31 - __builtin_clz(colany));
                    bminy = min(bminy, ya + 1 + first);
                    bmaxy = max(bmaxy, ya + 1 + last);
                }
            }
        }
    }
    }
    }
    bminx = cm3d_wave_min(bminx); bminy = cm3d_wave_min(bminy); bmaxx = cm3d_wave_max(bmaxx); bmaxy = cm3d_wave_max(bmaxy);
#ifdef CM3D_DIAG
    if (wdiag && lane == 0 && band == 0 && m < RW_DIAG_WAVES) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        g_rw_wave[4 * m] = t_start; g_rw_wave[4 * m + 1] = rw_now();
        g_rw_wave[4 * m + 2] = ((unsigned long long)xcc << 32) | hw; g_rw_wave[4 * m + 3] = (unsigned long long)n;
    }
#endif
    if (nb == 1) {                              // uniform over the workgroup (or max_bands == 1): one band, no hand-over
        if (band == 0 && lane < 8)
            bbox[CM3D_BBOX_STRIDE * m + lane] = lane == 0 ? bminx : lane == 1 ? bminy : lane == 2 ? bmaxx : lane == 3 ? bmaxy
                                                : lane == 4 ? rect_x : lane == 5 ? rect_y : lane == 6 ? rect_w : rect_h;
        continue;
    }
    if (lane < 4) s_part[wave][lane] = lane == 0 ? bminx : lane == 1 ? bminy : lane == 2 ? bmaxx : bmaxy;
    __syncthreads();
    if (wave == 0 && lane < 4) {
        int v = s_part[0][lane];
        for (int w = 1; w < RW_WAVES; ++w) v = lane < 2 ? min(v, s_part[w][lane]) : max(v, s_part[w][lane]);
        bbox[CM3D_BBOX_STRIDE * m + lane] = v;
    }
    if (wave == 0 && lane >= 4 && lane < 8) bbox[CM3D_BBOX_STRIDE * m + lane] = lane == 4 ? rect_x : lane == 5 ? rect_y : lane == 6 ? rect_w : rect_h;
    }
}

static inline size_t rle_align16(size_t v) { return (v + 15) & ~(size_t)15; }

extern "C" int64_t cm3d_rle_workspace_bytes(int32_t total_runs)
{
    // run ends (4 B per run) followed by one int4 rectangle per mask; a mask has at least one run
    return total_runs > 0 ? (int64_t)(rle_align16((size_t)total_runs * 4) + (size_t)total_runs * 16) : 0;
}

extern "C" int cm3d_rle_to_dense(const uint32_t *rle_counts, const int32_t *rle_off, int32_t n_masks, int32_t total_runs,
                                 int32_t W, int32_t H, uint8_t *dense, void *workspace, int64_t workspace_bytes,
                                 cm3d_stream_t stream)
{
    if (!rle_counts || !rle_off || !dense || !workspace) return CM3D_ERR_ARG;
    if (n_masks <= 0 || total_runs <= 0 || n_masks > total_runs || W <= 0 || H <= 0 || (int64_t)W * H >= (1ll << 31)) return CM3D_ERR_ARG;
    if (workspace_bytes < cm3d_rle_workspace_bytes(total_runs)) return CM3D_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    uint32_t *ends = (uint32_t *)workspace;
    int4 *mrect = (int4 *)((char *)workspace + rle_align16((size_t)total_runs * 4));
    hipLaunchKernelGGL(k_rle_ends, dim3(n_masks), dim3(RE_THREADS), 0, st, rle_counts, rle_off, W, ends, mrect, (int32_t *)nullptr);
    CM3D_CHECK_LAUNCH();
    const int total_px = W * H;
    int chunks = (total_px / 16 + 255) / 256;
    if (chunks > 64) chunks = 64;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(k_rle_to_dense, dim3(chunks, n_masks), dim3(256), 0, st, ends, rle_off, total_px, dense);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}

extern "C" int cm3d_batch_begin(int32_t *status, int32_t *hit_count, int32_t n_masks, uint32_t *removed_bits, int64_t removed_words,
                                cm3d_stream_t stream);

static int rle_erode_pack_impl(const uint32_t *rle_counts, const int32_t *rle_off, int32_t n_masks, int32_t total_runs,
                               int32_t W, int32_t H, uint32_t *packed, int32_t *bbox, void *workspace,
                               int64_t workspace_bytes, const RwBegin begin, cm3d_stream_t stream)
{
    if (!rle_counts || !rle_off || !packed || !bbox || !workspace) return CM3D_ERR_ARG;
    if (n_masks <= 0 || total_runs <= 0 || n_masks > total_runs || W <= 0 || H <= 0 || W > 32 * EP_MAX_WP || W > 32767 || H > 32767)
        return CM3D_ERR_ARG;
    if (workspace_bytes < cm3d_rle_workspace_bytes(total_runs)) return CM3D_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int Wp = (W + 31) / 32;
    static int lds_words = 0;
    if (!lds_words) { const char *e = getenv("CM3D_RLE_LDS_WORDS"); lds_words = e ? atoi(e) : RLE_LDS_WORDS; }
    if (lds_words / (Wp + 2) - 2 < 1) return CM3D_ERR_ARG;      // tile height of a full-width mask
    (void)workspace;                                            // reserved (the run ends are not materialised)
    // ordinary masks (on average at most 1024 runs): one wave per mask; lists of thousands of runs: one workgroup per mask
    static int form = -1, lds_wave = 0, bands = 1;
    if (form < 0) {
        const char *bd = getenv("CM3D_RLE_BANDS");                // 1: a wave per mask; 2..4: a workgroup per mask, its rows in up to that many bands
        bands = bd ? atoi(bd) : 1;
        if (bands < 1 || bands > RW_WAVES) bands = 1;
        const char *e = getenv("CM3D_RLE_FORM");                  // "wave" / "block" force one form (experiments, tests)
        form = e ? (e[0] == 'w' ? 1 : 2) : 0;
        const char *w = getenv("CM3D_RLEW_LDS_WORDS");
        lds_wave = w ? atoi(w) : RW_LDS_WORDS;
        if (lds_wave < 3 * (EP_MAX_WP + 2)) lds_wave = 3 * (EP_MAX_WP + 2);
        lds_wave = (lds_wave + 3) & ~3;
    }
    const bool wave_form = form == 1 || (form == 0 && (int64_t)total_runs <= (int64_t)n_masks * 1024);
    if (wave_form) {
        const size_t lds = (size_t)RW_WAVES * lds_wave * 4;
        static size_t lds_allowed = 48 * 1024;
        if (lds > lds_allowed) {
            if (hipFuncSetAttribute((const void *)k_rle_erode_pack_wave, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return CM3D_ERR_LAUNCH;
            lds_allowed = lds;
        }
        const char *dg = getenv("CM3D_RLE_DIAG");                     // experiments only (read per call: tools/stage_ablate.py switches it
        const int rle_diag = dg ? atoi(dg) : 0;                       // between passes): 1 no stores, 2 no erosion, 4 no paint, 8 stop behind pass 1
        int grid = bands > 1 ? n_masks : (n_masks + RW_WAVES - 1) / RW_WAVES;
        static int gcap = -1;
        if (gcap < 0) { const char *e = getenv("CM3D_RLE_GRID"); gcap = e ? atoi(e) : 0; }
        if (bands == 1 && gcap > 0 && grid > gcap) grid = gcap;
        hipLaunchKernelGGL(k_rle_erode_pack_wave, dim3(grid), dim3(RW_THREADS), lds, st,
                           rle_counts, rle_off, n_masks, W, H, Wp, lds_wave, packed, bbox, bands, rle_diag, begin);
    } else {
        if (begin.status) {                 // (the workgroup-per-mask form does not carry the reset: a launch of its own, as cm3d_batch_begin makes it)
            const int rc = cm3d_batch_begin(begin.status, begin.hit_count, begin.n_masks, begin.removed_bits, begin.removed_words, stream);
            if (rc != CM3D_OK) return rc;
        }
        hipLaunchKernelGGL(k_rle_erode_pack, dim3(n_masks), dim3(EP_THREADS), (size_t)lds_words * 4, st, rle_counts, rle_off, W, H, Wp,
                           lds_words, packed, bbox);
    }
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}

extern "C" int cm3d_rle_erode_pack(const uint32_t *rle_counts, const int32_t *rle_off, int32_t n_masks, int32_t total_runs,
                                   int32_t W, int32_t H, uint32_t *packed, int32_t *bbox, void *workspace,
                                   int64_t workspace_bytes, cm3d_stream_t stream)
{
    const RwBegin none = {nullptr, nullptr, nullptr, 0, 0};
    return rle_erode_pack_impl(rle_counts, rle_off, n_masks, total_runs, W, H, packed, bbox, workspace, workspace_bytes, none, stream);
}

extern "C" int cm3d_rle_erode_pack_begin(const uint32_t *rle_counts, const int32_t *rle_off, int32_t n_masks, int32_t total_runs,
                                         int32_t W, int32_t H, uint32_t *packed, int32_t *bbox, void *workspace,
                                         int64_t workspace_bytes, int32_t *status, int32_t *hit_count, int32_t n_count_masks,
                                         uint32_t *removed_bits, int64_t removed_words, cm3d_stream_t stream)
{
    if (!status || !hit_count || n_count_masks < 0 || removed_words < 0 || (removed_words > 0 && !removed_bits)) return CM3D_ERR_ARG;
    const RwBegin begin = {status, hit_count, removed_bits, (long long)removed_words, n_count_masks};
    return rle_erode_pack_impl(rle_counts, rle_off, n_masks, total_runs, W, H, packed, bbox, workspace, workspace_bytes, begin, stream);
}
