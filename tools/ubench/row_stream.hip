// Runs ON THE GPU BOX (tools/ubench/run.sh): what a persistent wave that owns 256-row chunks (lane l = rows 4l..4l+3) can stream
// from HBM, by row layout, rows in flight per wave and waves per CU -- the skeleton of k_project_hits without its arithmetic.
//   layout 0: rows of 5 floats, 4 x dwordx3 per lane (the .bin files as they are)          20 B/row cross HBM
//   layout 1: rows of 4 floats, 4 x dwordx3 per lane                                        16 B/row
//   layout 2: quads  x0..3 y0..3 z0..3 (48 B per 4 rows), 3 x dwordx4 per lane              12 B/row
//   layout 3: chunk-planar x[256] y[256] z[256], 3 fully coalesced dwordx4 per lane         12 B/row
//   layout 4: layout 2 through LDS-DMA (global_load_lds_dwordx4 x3, then ds_read_b128 x3)   12 B/row
// Plain HIP runtime, no torch.  Prints one line per configuration: time, TB/s of the bytes that must move.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define WC 256

template <int LAYOUT>
struct Rows { float v[12]; };

template <int LAYOUT>
static __device__ __forceinline__ void load_rows(Rows<LAYOUT> &r, const float *__restrict__ src, size_t chunk, int lane)
{
    if (LAYOUT == 0 || LAYOUT == 1) {
        constexpr int RS = LAYOUT == 0 ? 5 : 4;
        const float *p = src + (chunk * WC + 4 * (size_t)lane) * RS;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f3u t = *reinterpret_cast<const f3u *>(p + j * RS);
            r.v[3 * j] = t.x; r.v[3 * j + 1] = t.y; r.v[3 * j + 2] = t.z;
        }
    } else if (LAYOUT == 2) {
        const f4 *p = reinterpret_cast<const f4 *>(src + chunk * (WC * 3) + 12 * (size_t)lane);
#pragma unroll
        for (int j = 0; j < 3; ++j) { const f4 t = p[j]; r.v[4 * j] = t.x; r.v[4 * j + 1] = t.y; r.v[4 * j + 2] = t.z; r.v[4 * j + 3] = t.w; }
    } else {
        const f4 *p = reinterpret_cast<const f4 *>(src + chunk * (WC * 3)) + lane;
#pragma unroll
        for (int j = 0; j < 3; ++j) { const f4 t = p[64 * j]; r.v[4 * j] = t.x; r.v[4 * j + 1] = t.y; r.v[4 * j + 2] = t.z; r.v[4 * j + 3] = t.w; }
    }
}

// DEPTH chunks in flight per wave (registers); static interleaved chunk lists (wave t takes t, t + T, ...)
template <int LAYOUT, int DEPTH, int WORK>
__global__ __launch_bounds__(256) void k_stream(const float *__restrict__ src, long long n_chunks, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long T = (long long)gridDim.x * 4, t = (long long)blockIdx.x * 4 + wave;
    Rows<LAYOUT> buf[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (t + d * T < n_chunks) load_rows<LAYOUT>(buf[d], src, (size_t)(t + d * T), lane);
    float acc = 0.f;
    for (long long c = t; c < n_chunks; c += DEPTH * T) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (c + d * T >= n_chunks) break;
            Rows<LAYOUT> cur = buf[d];
            if (c + (d + DEPTH) * T < n_chunks) load_rows<LAYOUT>(buf[d], src, (size_t)(c + (d + DEPTH) * T), lane);
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 12; ++q) s += cur.v[q];
            // WORK dependent vector instructions per chunk stand for the camera loop
#pragma unroll 1
            for (int w = 0; w < WORK; ++w) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(s));
            acc += s;
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

// layout 4: rows of a chunk land in the wave's LDS ring through LDS-DMA, SLOTS chunks in flight, no registers held
template <int SLOTS, int WORK>
__global__ __launch_bounds__(256) void k_stream_dma(const float *__restrict__ src, long long n_chunks, float *__restrict__ out)
{
    extern __shared__ __align__(16) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long T = (long long)gridDim.x * 4, t = (long long)blockIdx.x * 4 + wave;
    const unsigned ring = (unsigned)(size_t)lds + (unsigned)wave * SLOTS * 3072u;       // LDS byte address of the wave's ring
    auto dma = [&](long long chunk, int slot) {
        const float *g = src + (size_t)chunk * (WC * 3) + 4 * lane;                      // lane's 16 bytes of each 1-KiB piece
        const unsigned dst = ring + (unsigned)slot * 3072u;
        unsigned keep;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g + 256 * j), "s"(dst + 1024u * j) : "memory");
    };
#pragma unroll
    for (int d = 0; d < SLOTS; ++d) dma(std::min(t + d * T, n_chunks - 1), d);           // (a dummy load keeps the count of operations fixed)
    float acc = 0.f;
    int slot = 0;
    for (long long c = t; c < n_chunks; c += T) {
        // everything but the SLOTS - 1 youngest chunks (3 operations each) has landed
        if (SLOTS == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (SLOTS == 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        if (SLOTS == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        if (SLOTS == 4) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        const f4 *p = reinterpret_cast<const f4 *>(lds + (size_t)wave * SLOTS * 3072 + (size_t)slot * 3072 + 48 * lane);
        const f4 a = p[0], b = p[1], d4 = p[2];
        float s = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((d4.x + d4.y) + (d4.z + d4.w));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               // the slot is free again
        dma(std::min(c + (long long)SLOTS * T, n_chunks - 1), slot);
#pragma unroll 1
        for (int w = 0; w < WORK; ++w) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(s));
        acc += s;
        slot = slot + 1 == SLOTS ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

static float *g_src, *g_out;
static const long long ROWS = 8960000;                 // the headline batch: 256 frames x 35 000 rows

template <typename F>
static double time_us(F launch)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    std::vector<float> ts;
    for (int i = 0; i < 9; ++i) {
        (void)hipEventRecord(a, 0);
        launch();
        (void)hipEventRecord(b, 0);
        if (hipEventSynchronize(b) != hipSuccess) { fprintf(stderr, "launch failed\n"); exit(1); }
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (i >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2] * 1e3;
}

template <int LAYOUT, int DEPTH, int WORK>
static void run(int wg_per_cu)
{
    const long long n_chunks = ROWS / WC;
    const int bpr = LAYOUT == 0 ? 20 : (LAYOUT == 1 ? 16 : 12);
    const double us = time_us([&] { hipLaunchKernelGGL((k_stream<LAYOUT, DEPTH, WORK>), dim3(256 * wg_per_cu), dim3(256), 0, 0, g_src, n_chunks, g_out); });
    printf("layout %d  depth %d  work %4d  %d workgroups/CU: %7.1f us  %5.2f TB/s (%d B/row)\n", LAYOUT, DEPTH, WORK, wg_per_cu, us, ROWS * (double)bpr / us / 1e6, bpr);
    fflush(stdout);
}
template <int SLOTS, int WORK>
static void run_dma(int wg_per_cu)
{
    const long long n_chunks = ROWS / WC;
    const size_t ldsb = 4 * SLOTS * 3072;
    const double us = time_us([&] { hipLaunchKernelGGL((k_stream_dma<SLOTS, WORK>), dim3(256 * wg_per_cu), dim3(256), ldsb, 0, g_src, n_chunks, g_out); });
    printf("layout 4  slots %d  work %4d  %d workgroups/CU: %7.1f us  %5.2f TB/s (12 B/row, LDS-DMA)\n", SLOTS, WORK, wg_per_cu, us, ROWS * 12.0 / us / 1e6);
    fflush(stdout);
}

int main()
{
    const size_t bytes = (size_t)ROWS * 20 + 4096;
    if (hipMalloc(&g_src, bytes) != hipSuccess || hipMalloc(&g_out, 256 * 8 * 256 * 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    (void)hipMemset(g_src, 0, bytes);
    // a 1-GiB buffer written between configurations would flush the Infinity Cache; the batch (108-179 MB) is streamed once per
    // launch and launches follow each other, so part of it may be served on-die: the same holds for the real kernel's passes.
    for (int wg : {3, 4}) {
        run<0, 1, 0>(wg); run<0, 2, 0>(wg); run<1, 1, 0>(wg); run<1, 2, 0>(wg);
        run<2, 1, 0>(wg); run<2, 2, 0>(wg); run<2, 3, 0>(wg); run<3, 1, 0>(wg); run<3, 2, 0>(wg);
        run_dma<1, 0>(wg); run_dma<2, 0>(wg); run_dma<3, 0>(wg); run_dma<4, 0>(wg);
    }
    // with a camera loop's worth of dependent issue per chunk (1000 instructions ~ 4000 cycles for a lone wave)
    for (int wg : {3, 4}) {
        run<0, 1, 1000>(wg); run<0, 2, 1000>(wg); run<2, 1, 1000>(wg); run<2, 2, 1000>(wg); run<3, 2, 1000>(wg);
        run_dma<2, 1000>(wg); run_dma<3, 1000>(wg);
    }
    run<2, 2, 1000>(6); run<2, 2, 1000>(8); run_dma<3, 1000>(6); run_dma<3, 1000>(8);
    run<2, 2, 500>(3); run<2, 2, 500>(4); run<2, 2, 500>(6); run<2, 2, 500>(8);
    return 0;
}
