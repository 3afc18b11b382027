// a1/a3/f1: mask expansion, 3x3 erosion and bit-packing.
//   reference: pycocotools decode at src/nuscenes/2d_to_3d.py:425, cv2.erode at :526-527,
//   bool/transpose/H2D at :542-544.
// All three kernels are HBM-bound streaming kernels (no MFMA):
//   k_erode_pack      reads n*W*H dense bytes once (+ 2 halo rows per band), writes n*H*Wp*4
//   k_rle_to_dense    reads the run ends (KBs), writes n*W*H
//   k_rle_erode_pack  reads the run ends (KBs), writes n*H*Wp*4
// A band of packed rows lives in LDS; the erosion is 9 word reads + shifts per output word.
#include "common.h"

#define EP_THREADS 256
#define EP_MAX_WP 128          // supports W <= 4096
#define EP_LDS_WORDS 8192      // at most 32 KiB of packed rows per workgroup (dynamic LDS, sized per launch)

// one bit per non-zero byte of a 16-byte chunk -> 16 bits
static __device__ __forceinline__ uint32_t pack16(uint4 v)
{
    auto nz4 = [](uint32_t w) -> uint32_t {
        // 0x01 in every non-zero byte, then gather the four flags into a nibble
        uint32_t t = ((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w;
        t = (t >> 7) & 0x01010101u;
        return (t * 0x01020408u) >> 24;   // bit k = byte k != 0  (erode_pack)
    };
    return (nz4(v.x) & 0xF) | ((nz4(v.y) & 0xF) << 4) | ((nz4(v.z) & 0xF) << 8) | ((nz4(v.w) & 0xF) << 12);
}

// Erode the LDS band and write rows [y0, y0+rows) of one mask; also reduces the bbox.
// s_rows holds packed rows y0-1 .. y0+rows (row index r = y - (y0-1)), rows outside the
// image are all-ones (the reference's erosion ignores out-of-image neighbours).
static __device__ __forceinline__ void erode_band_store(const uint32_t *s_rows, int Wp, int W, int H, int y0, int rows,
                                                         uint32_t *__restrict__ out_mask, int32_t *__restrict__ bbox4)
{
    const uint32_t tail_mask = (W & 31) ? ((1u << (W & 31)) - 1u) : 0xFFFFFFFFu;
    int minx = 0x7FFFFFFF, miny = 0x7FFFFFFF, maxx = -1, maxy = -1;
    const int nwords = rows * Wp;
    for (int q = threadIdx.x; q < nwords; q += EP_THREADS) {
        const int r = q / Wp, xw = q - r * Wp;
        uint32_t e = 0xFFFFFFFFu;
#pragma unroll
        for (int dr = 0; dr < 3; ++dr) {
            const uint32_t *row = s_rows + (r + dr) * Wp;
            uint32_t c = row[xw];
            // neighbours beyond the image edge count as set
            uint32_t l = xw > 0 ? row[xw - 1] : 0xFFFFFFFFu;
            uint32_t rr = xw < Wp - 1 ? row[xw + 1] : 0xFFFFFFFFu;
            uint32_t left = (c << 1) | (l >> 31);      // pixel x-1
            uint32_t right = (c >> 1) | (rr << 31);    // pixel x+1
            e &= c & left & right;
        }
        if (xw == Wp - 1) e &= tail_mask;
        out_mask[(size_t)(y0 + r) * Wp + xw] = e;
        if (e) {
            int y = y0 + r;
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

__global__ void k_bbox_init(int32_t *__restrict__ bbox, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        bbox[4 * i + 0] = 0x7FFFFFFF; bbox[4 * i + 1] = 0x7FFFFFFF;
        bbox[4 * i + 2] = -1;         bbox[4 * i + 3] = -1;
    }
}

// grid (bands, n_masks).  Fast path W % 32 == 0: 16-byte coalesced loads, 16 bits per lane,
// lane pairs merged with one DPP-class shuffle.  Other widths: byte loads (parity sizes only).
__global__ __launch_bounds__(EP_THREADS) void k_erode_pack(const uint8_t *__restrict__ dense, int W, int H, int Wp,
                                                            int band_rows, uint32_t *__restrict__ packed,
                                                            int32_t *__restrict__ bbox)
{
    extern __shared__ __align__(16) uint32_t s_rows[];
    const int m = blockIdx.y;
    const int y0 = blockIdx.x * band_rows;
    const int rows = min(band_rows, H - y0);
    const uint8_t *img = dense + (size_t)m * W * H;
    const int lrows = rows + 2;                 // with halo
    if ((W & 31) == 0) {
        const int cpr = W >> 4;                 // 16-byte chunks per row
        const int nchunks = lrows * cpr;        // even, since cpr is even
        // the band (with halo) is one contiguous byte range of the image: stream it with
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
                    if ((q & 1) == 0) s_rows[r * Wp + (cx >> 1)] = bits | (hi << 16);
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
            s_rows[q] = bits;
        }
    }
    __syncthreads();
    erode_band_store(s_rows, Wp, W, H, y0, rows, packed + (size_t)m * H * Wp, bbox + 4 * m);
}

extern "C" int cm3d_erode_pack(const uint8_t *dense, int32_t n_masks, int32_t W, int32_t H, uint32_t *packed,
                               int32_t *bbox, cm3d_stream_t stream)
{
    if (!dense || !packed || !bbox) return CM3D_ERR_ARG;
    if (n_masks <= 0 || W <= 0 || H <= 0 || W > 32 * EP_MAX_WP || W > 32767 || H > 32767) return CM3D_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int Wp = (W + 31) / 32;
    int band_rows = EP_LDS_WORDS / Wp - 2;
    if (band_rows > 62) band_rows = 62;
    if (band_rows < 1) return CM3D_ERR_ARG;
    const int bands = (H + band_rows - 1) / band_rows;
    hipLaunchKernelGGL(k_bbox_init, dim3((n_masks + 255) / 256), dim3(256), 0, st, bbox, n_masks);
    CM3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_erode_pack, dim3(bands, n_masks), dim3(EP_THREADS), (size_t)(band_rows + 2) * Wp * 4, st, dense, W, H, Wp, band_rows, packed, bbox);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}

// ---------------------------------------------------------------------------
// RLE: run lengths -> inclusive run ends (per mask) in the workspace, plus the first / last
// image row that holds a set pixel (yrange[m] = {y_first, y_last}, y_first > y_last when empty).
__global__ __launch_bounds__(1024) void k_rle_ends(const uint32_t *__restrict__ cnts, const int32_t *__restrict__ rle_off,
                                                   int W, uint32_t *__restrict__ ends, int2 *__restrict__ yrange)
{
    __shared__ int s_part[16];
    __shared__ unsigned int s_lo, s_hi;
    const int m = blockIdx.x;
    const int o = rle_off[m], n = rle_off[m + 1] - o;
    if (threadIdx.x == 0) { s_lo = 0xFFFFFFFFu; s_hi = 0u; }
    int carry = 0;
    unsigned int lo = 0xFFFFFFFFu, hi = 0u;
    for (int base = 0; base < n; base += 1024) {
        int i = base + threadIdx.x;
        int v = i < n ? (int)cnts[o + i] : 0;
        int tot;
        int ex = cm3d_block1024_excl_scan(v, s_part, tot);
        if (i < n) {
            ends[o + i] = (uint32_t)(carry + ex + v);
            if ((i & 1) && v > 0) {                       // a 1-run [start, end)
                lo = min(lo, (unsigned int)(carry + ex));
                hi = max(hi, (unsigned int)(carry + ex + v));
            }
        }
        carry += tot;
        __syncthreads();
    }
    if (lo != 0xFFFFFFFFu) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    __syncthreads();
    if (threadIdx.x == 0)
        yrange[m] = s_hi > 0 ? make_int2((int)(s_lo / (unsigned int)W), (int)((s_hi - 1) / (unsigned int)W)) : make_int2(1, 0);
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

// f1: RLE -> packed band in LDS -> erode -> store.  grid (bands, n_masks).
// Bands that cannot hold an eroded pixel (outside the row range of the mask's set pixels) exit
// at once and write nothing: rows outside [bbox.y0, bbox.y1] are never read downstream.
__global__ __launch_bounds__(EP_THREADS) void k_rle_erode_pack(const uint32_t *__restrict__ ends_all,
                                                                const int32_t *__restrict__ rle_off,
                                                                const int2 *__restrict__ yrange, int W, int H, int Wp,
                                                                int band_rows, uint32_t *__restrict__ packed,
                                                                int32_t *__restrict__ bbox)
{
    extern __shared__ __align__(16) uint32_t s_rows[];
    __shared__ int s_range[2];
    const int m = blockIdx.y;
    const int y0 = blockIdx.x * band_rows;
    const int rows = min(band_rows, H - y0);
    const int2 yr = yrange[m];
    if (yr.x > yr.y || y0 > yr.y || y0 + rows - 1 < yr.x) return;   // no set pixel in rows y0..y0+rows-1
    const int lrows = rows + 2;
    const int o = rle_off[m], n = rle_off[m + 1] - o;
    const uint32_t *ends = ends_all + o;
    const int ya = y0 - 1, yb = y0 + rows;          // first / last LDS row (may lie outside the image)
    // rows outside the image are all ones, rows inside start at zero; bits beyond W in the
    // last word of a row are ones (they only ever act as "neighbour beyond the edge")
    const uint32_t pad = (W & 31) ? ~((1u << (W & 31)) - 1u) : 0u;
    for (int q = threadIdx.x; q < lrows * Wp; q += EP_THREADS) {
        const int r = q / Wp, xw = q - r * Wp;
        const int y = ya + r;
        s_rows[q] = (y < 0 || y >= H) ? 0xFFFFFFFFu : (xw == Wp - 1 ? pad : 0u);
    }
    if (threadIdx.x < 2) s_range[threadIdx.x] = 0;
    __syncthreads();
    const int yc0 = max(ya, 0), yc1 = min(yb, H - 1);      // image rows held in LDS
    const uint32_t px0 = (uint32_t)yc0 * W, px1 = (uint32_t)(yc1 + 1) * W;   // pixel range [px0, px1)
    // cooperative lower bounds (ends is ascending): #runs with end <= px0, #runs with end <= px1-1
    int c0 = 0, c1 = 0;
    for (int i = threadIdx.x; i < n; i += EP_THREADS) {
        const uint32_t e = ends[i];
        c0 += e <= px0 ? 1 : 0;
        c1 += e <= px1 - 1 ? 1 : 0;
    }
    c0 = cm3d_wave_sum(c0); c1 = cm3d_wave_sum(c1);
    if (cm3d_lane() == 0) { atomicAdd(&s_range[0], c0); atomicAdd(&s_range[1], c1); }
    __syncthreads();
    const int r_first = s_range[0], r_last = min(s_range[1], n - 1);
    // every 1-run (odd index) overlapping the band sets its bits, row by row
    for (int r = r_first + threadIdx.x; r <= r_last; r += EP_THREADS) {
        if (!(r & 1)) continue;
        uint32_t s = r > 0 ? ends[r - 1] : 0u, e = ends[r];
        s = max(s, px0); e = min(e, px1);
        while (s < e) {
            const uint32_t y = s / W, x = s - y * W;
            const uint32_t xe = min((uint32_t)W, x + (e - s));   // exclusive end within this row
            uint32_t *row = s_rows + (y - ya) * Wp;
            uint32_t w0 = x >> 5, w1 = (xe - 1) >> 5;
            uint32_t m0 = 0xFFFFFFFFu << (x & 31);
            uint32_t m1 = 0xFFFFFFFFu >> (31 - ((xe - 1) & 31));
            if (w0 == w1) atomicOr(&row[w0], m0 & m1);
            else {
                atomicOr(&row[w0], m0);
                for (uint32_t w = w0 + 1; w < w1; ++w) atomicOr(&row[w], 0xFFFFFFFFu);
                atomicOr(&row[w1], m1);
            }
            s += xe - x;
        }
    }
    __syncthreads();
    erode_band_store(s_rows, Wp, W, H, y0, rows, packed + (size_t)m * H * Wp, bbox + 4 * m);
}

extern "C" int64_t cm3d_rle_workspace_bytes(int32_t total_runs)
{
    // run ends (4 B per run, padded to 16) followed by the per-mask row range is sized by the caller's
    // n_masks; to keep the signature simple the row ranges live behind the ends: 8 B per run bounds it
    return total_runs > 0 ? (((int64_t)total_runs * 4 + 15) / 16) * 16 + (int64_t)total_runs * 8 : 0;
}

extern "C" int cm3d_rle_to_dense(const uint32_t *rle_counts, const int32_t *rle_off, int32_t n_masks, int32_t total_runs,
                                 int32_t W, int32_t H, uint8_t *dense, void *workspace, int64_t workspace_bytes,
                                 cm3d_stream_t stream)
{
    if (!rle_counts || !rle_off || !dense || !workspace) return CM3D_ERR_ARG;
    if (n_masks <= 0 || total_runs <= 0 || W <= 0 || H <= 0 || (int64_t)W * H >= (1ll << 31)) return CM3D_ERR_ARG;
    if (workspace_bytes < cm3d_rle_workspace_bytes(total_runs)) return CM3D_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    uint32_t *ends = (uint32_t *)workspace;
    int2 *yrange = (int2 *)((char *)workspace + (((size_t)total_runs * 4 + 15) / 16) * 16);
    if (n_masks > total_runs) return CM3D_ERR_ARG;
    hipLaunchKernelGGL(k_rle_ends, dim3(n_masks), dim3(1024), 0, st, rle_counts, rle_off, W, ends, yrange);
    CM3D_CHECK_LAUNCH();
    const int total_px = W * H;
    int chunks = (total_px / 16 + 255) / 256;
    if (chunks > 64) chunks = 64;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(k_rle_to_dense, dim3(chunks, n_masks), dim3(256), 0, st, ends, rle_off, total_px, dense);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}

extern "C" int cm3d_rle_erode_pack(const uint32_t *rle_counts, const int32_t *rle_off, int32_t n_masks, int32_t total_runs,
                                   int32_t W, int32_t H, uint32_t *packed, int32_t *bbox, void *workspace,
                                   int64_t workspace_bytes, cm3d_stream_t stream)
{
    if (!rle_counts || !rle_off || !packed || !bbox || !workspace) return CM3D_ERR_ARG;
    if (n_masks <= 0 || total_runs <= 0 || W <= 0 || H <= 0 || W > 32 * EP_MAX_WP || W > 32767 || H > 32767) return CM3D_ERR_ARG;
    if (workspace_bytes < cm3d_rle_workspace_bytes(total_runs)) return CM3D_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    uint32_t *ends = (uint32_t *)workspace;
    const int Wp = (W + 31) / 32;
    int band_rows = EP_LDS_WORDS / Wp - 2;
    if (band_rows > 62) band_rows = 62;
    if (band_rows < 1) return CM3D_ERR_ARG;
    const int bands = (H + band_rows - 1) / band_rows;
    int2 *yrange = (int2 *)((char *)workspace + (((size_t)total_runs * 4 + 15) / 16) * 16);
    if (n_masks > total_runs) return CM3D_ERR_ARG;
    hipLaunchKernelGGL(k_rle_ends, dim3(n_masks), dim3(1024), 0, st, rle_counts, rle_off, W, ends, yrange);
    CM3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_bbox_init, dim3((n_masks + 255) / 256), dim3(256), 0, st, bbox, n_masks);
    CM3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_rle_erode_pack, dim3(bands, n_masks), dim3(EP_THREADS), (size_t)(band_rows + 2) * Wp * 4, st, ends, rle_off, yrange, W, H, Wp, band_rows,
                       packed, bbox);
    CM3D_CHECK_LAUNCH();
    return CM3D_OK;
}
