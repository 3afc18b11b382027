// Work list of the medoid stage (a9): one descriptor per 64-column tile of every index list.
// Shared by medoid.hip (stand-alone builder) and project.hip (built beside the compaction, see k_compact_hits).
#pragma once
#include "common.h"

struct TileDesc { int m, off, M, jt, t; };  // mask, start in hit_idx, list length, tile index inside the mask, tile id

// One descriptor per 64-column tile of every index list, written in WORK order: tiles of the longest lists first
// (classes by tile count; a tile's cost is its list length), so that the waves which run longest start first and
// the short ones fill in behind them.  Results are indexed by the tile id t = tile_off[m] + jt, not by the work
// position, so the order has no influence on any output.  One workgroup of NT threads.
#define MD_UNI 7                         // classes 0..6: exactly 1..7 tiles (lists of up to 448 points)
#define MD_CLASSES 16                    // classes 7..15: 8-9, 10-12, 13-16, 17-21, 22-28, 29-37, 38-49, 50-65, 66+ tiles
#define MD_DESC_PER 8                    // masks per thread per round: all their loads are in flight together

// class of a list with nt >= 1 tiles: geometric steps of about 1.3 above 7 tiles, so that "longest first" holds to
// within a class width for the long lists that carry most of the work of a multi-sweep batch
static __device__ __forceinline__ int md_class(int nt)
{
    if (nt <= MD_UNI) return nt - 1;
    return MD_UNI + (nt >= 10) + (nt >= 13) + (nt >= 17) + (nt >= 22) + (nt >= 29) + (nt >= 38) + (nt >= 50) + (nt >= 66);
}

// Adds the tiles of the wave's lanes, class by class, to a LANE-DISTRIBUTED accumulator: lane c of `acc` holds the sum of
// class c (one VGPR instead of MD_CLASSES scalar registers -- the builder shares a kernel with the compaction, whose
// workgroups pay for every register the builder needs).  Lanes of a class below MD_UNI all carry the same tile
// count, so a ballot and a popcount give the class sum without touching memory.
static __device__ __forceinline__ void md_class_sums(int nt, int &acc)
{
    const int lane = cm3d_lane();
#pragma unroll
    for (int c = 0; c < MD_UNI; ++c) {
        const int x = (int)__popcll(__ballot(nt == c + 1)) * (c + 1);
        acc += lane == c ? x : 0;
    }
    if (__ballot(nt > MD_UNI)) {
        const int cls = nt > MD_UNI ? md_class(nt) : -1;
#pragma unroll
        for (int c = MD_UNI; c < MD_CLASSES; ++c) {
            if (!__ballot(cls == c)) continue;
            const int x = __builtin_amdgcn_readfirstlane(cm3d_wave_sum(cls == c ? nt : 0));
            acc += lane == c ? x : 0;
        }
    }
}

// tile capacity of a batch: every mask may end with a partial tile
static inline int64_t md_tile_cap(int32_t n_masks, int32_t idx_cap)
{
    return (int64_t)n_masks + (int64_t)idx_cap / CM3D_MEDOID_TILE + 1;
}

// s_hist / s_cur: MD_CLASSES ints of LDS each.  Contains workgroup barriers: call with all NT threads.
// The list can be built by `nparts` workgroups that never talk to each other: masks are cut into slots of NT
// consecutive masks, workgroup `part` places the masks of a contiguous run of slots.  Every workgroup counts ALL
// masks (class totals -> where each class starts; the counts of the slots before its own -> where its share of a
// class starts), which costs n_masks loads from L2, and writes only its own share.
template <int NT, int PER = MD_DESC_PER>
static __device__ __forceinline__ void md_build_worklist(int n_masks, const int32_t *__restrict__ hit_off,
                                                         const int32_t *__restrict__ tile_off, int idx_cap, int tile_cap,
                                                         TileDesc *__restrict__ desc, int *s_hist, int *s_cur, int part = 0,
                                                         int nparts = 1)
{
    const int lane = cm3d_lane();
    if (threadIdx.x < MD_CLASSES) { s_hist[threadIdx.x] = 0; s_cur[threadIdx.x] = 0; }
    __syncthreads();
    const int nslots = (n_masks + NT - 1) / NT;
    const int per_part = (nslots + nparts - 1) / nparts;
    const int slot0 = part * per_part, slot1 = min(nslots, slot0 + per_part);      // this workgroup's slots
    auto tiles_of = [&](int m) {                        // tile count of mask m (0 past the end)
        if (m >= n_masks) return 0;
        const int a = tile_off[m];
        return max(0, min(tile_off[m + 1], tile_cap) - a);
    };
    // pass 1: tiles per class -- of all masks (s_hist) and of the masks in the slots before this workgroup's (s_cur)
    int wrest = 0, wbefore = 0, wmine = 0;              // lane c: tiles of class c (slots after / before / of this workgroup)
    for (int g0 = 0; g0 < nslots; g0 += PER) {
        int nt[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) nt[q] = tiles_of((g0 + q) * NT + (int)threadIdx.x);      // loads in flight together
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int g = g0 + q;
            if (g >= nslots) continue;                   // uniform
            if (g < slot0) md_class_sums(nt[q], wbefore);
            else if (g < slot1) md_class_sums(nt[q], wmine);
            else md_class_sums(nt[q], wrest);
        }
    }
    if (lane < MD_CLASSES) {
        const int all = wrest + wbefore + wmine;
        if (all) atomicAdd(&s_hist[lane], all);
        if (wbefore) atomicAdd(&s_cur[lane], wbefore);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int c = MD_CLASSES - 1; c >= 0; --c) { const int n = s_hist[c]; s_cur[c] += run; run += n; }     // longest lists first
    }
    __syncthreads();
    // pass 2 (own slots only): the wave reserves its share of every class with one atomic, positions inside it are
    // ballot ranks
    int vbase = 0;                                      // lane c: next free position of class c in this wave's share
    if (lane < MD_CLASSES && wmine) vbase = atomicAdd(&s_cur[lane], wmine);
    for (int g0 = slot0; g0 < slot1; g0 += PER) {
        int t0[PER], nt[PER], off[PER], M[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int m = (g0 + q) * NT + (int)threadIdx.x;
            t0[q] = 0; nt[q] = 0; off[q] = 0; M[q] = 0;
            if (g0 + q < slot1 && m < n_masks) {
                t0[q] = tile_off[m];
                nt[q] = tile_off[m + 1];
                off[q] = hit_off[m];
                M[q] = hit_off[m + 1];
            }
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            nt[q] = max(0, min(nt[q], tile_cap) - t0[q]);
            M[q] -= off[q];
            if (off[q] + M[q] > idx_cap) M[q] = max(0, idx_cap - off[q]);     // index capacity overflow: stay in bounds
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            if (g0 + q >= slot1) continue;               // uniform
            int pos = 0;
#pragma unroll
            for (int c = 0; c < MD_UNI; ++c) {
                const uint64_t mk = __ballot(nt[q] == c + 1);
                if (!mk) continue;
                if (nt[q] == c + 1) pos = __builtin_amdgcn_readlane(vbase, c) + cm3d_mbcnt(mk) * (c + 1);
                vbase += lane == c ? (int)__popcll(mk) * (c + 1) : 0;
            }
            if (__ballot(nt[q] > MD_UNI)) {
                const int cls = nt[q] > MD_UNI ? md_class(nt[q]) : -1;
#pragma unroll
                for (int c = MD_UNI; c < MD_CLASSES; ++c) {
                    if (!__ballot(cls == c)) continue;
                    const int v = cls == c ? nt[q] : 0;
                    const int inc = cm3d_wave_incl_scan(v);
                    if (v) pos = __builtin_amdgcn_readlane(vbase, c) + inc - v;
                    vbase += lane == c ? __builtin_amdgcn_readlane(inc, 63) : 0;
                }
            }
            if (nt[q] <= 0) continue;
            const int m = (g0 + q) * NT + (int)threadIdx.x;
            for (int jt = 0; jt < nt[q]; ++jt) desc[pos + jt] = TileDesc{m, off[q], M[q], jt, t0[q] + jt};
        }
    }
}


// The same work list straight from the per-mask hit COUNTS (cm3d_compact_hits: the offsets do not exist yet when its launch
// starts), together with what k_hit_offsets used to produce in a launch of its own: hit_off[m] = exclusive prefix of the
// counts, tile_off[m] = exclusive prefix of the tile counts, their totals and the overflow flag.  `nparts` workgroups that
// never talk to each other, each placing the masks of its own slots; every one of them reads ALL counts once (class totals,
// and the sums of the slots before its own).  On an index-capacity overflow (status bit 1) no tile is listed at all
// (tile_off = 0 everywhere): the host raises on that bit and nothing downstream may run off the end of a buffer.
// s_i: MD_FROM_COUNTS_LDS(NT) ints of LDS, 8-byte aligned.  Contains workgroup barriers: call with all NT threads.
#define MD_FROM_COUNTS_LDS(NT) (2 * MD_CLASSES + 2 * ((NT) / 64) + 8)
template <int NT>
static __device__ __forceinline__ void md_build_from_counts(int n_masks, const int32_t *__restrict__ hit_count, int idx_cap, int tile_cap,
                                                            int32_t *__restrict__ hit_off, int32_t *__restrict__ tile_off,
                                                            TileDesc *__restrict__ desc, int32_t *__restrict__ status, int *s_i,
                                                            int part, int nparts)
{
    constexpr int NW = NT / 64;
    int *s_hist = s_i, *s_cur = s_i + MD_CLASSES, *s_wc = s_i + 2 * MD_CLASSES, *s_wt = s_wc + NW, *s_tot = s_wt + NW;
    unsigned long long *s_tot64 = reinterpret_cast<unsigned long long *>(s_tot + 4);      // (s_i is 8-byte aligned and the offset even)
    const int lane = cm3d_lane(), wave = (int)threadIdx.x >> 6;
    if (threadIdx.x < MD_CLASSES) { s_hist[threadIdx.x] = 0; s_cur[threadIdx.x] = 0; }
    if (threadIdx.x < 8) s_tot[threadIdx.x] = 0;                     // 4 ints + 2 x 64 bits
    __syncthreads();
    const int nslots = (n_masks + NT - 1) / NT;
    const int per_part = (nslots + nparts - 1) / nparts;
    const int slot0 = min(nslots, part * per_part), slot1 = min(nslots, slot0 + per_part);      // this workgroup's slots
    // pass 1: all counts once -- tiles per class (all masks / the masks before this workgroup's slots), hits and tiles before
    // its slots, and the totals
    int wrest = 0, wbefore = 0, wmine = 0;              // lane c: tiles of class c
    long long cnt_before = 0, cnt_all = 0;
    int nt_before = 0, nt_all = 0;
    constexpr int PER = 4;
    for (int g0 = 0; g0 < nslots; g0 += PER) {
        int cv[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int m = (g0 + q) * NT + (int)threadIdx.x;
            cv[q] = (g0 + q < nslots && m < n_masks) ? hit_count[m] : 0;                        // loads in flight together
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int g = g0 + q;
            if (g >= nslots) continue;                   // uniform
            const int nt = (cv[q] + CM3D_MEDOID_TILE - 1) / CM3D_MEDOID_TILE;
            cnt_all += cv[q]; nt_all += nt;
            if (g < slot0) { md_class_sums(nt, wbefore); cnt_before += cv[q]; nt_before += nt; }
            else if (g < slot1) md_class_sums(nt, wmine);
            else md_class_sums(nt, wrest);
        }
    }
    if (lane < MD_CLASSES) {
        const int all = wrest + wbefore + wmine;
        if (all) atomicAdd(&s_hist[lane], all);
        if (wbefore) atomicAdd(&s_cur[lane], wbefore);
    }
    {
        // hit totals in 64 bits (a batch can hold more hits than its index buffer: that is the overflow this reports)
        long long cb = cnt_before, ca = cnt_all;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { cb += __shfl_xor(cb, o, 64); ca += __shfl_xor(ca, o, 64); }
        const int tb = cm3d_wave_sum(nt_before), ta = cm3d_wave_sum(nt_all);
        if (lane == 0) {
            atomicAdd(&s_tot64[0], (unsigned long long)cb); atomicAdd(&s_tot64[1], (unsigned long long)ca);
            atomicAdd(&s_tot[2], tb); atomicAdd(&s_tot[3], ta);
        }
    }
    __syncthreads();
    const long long total64 = (long long)s_tot64[1];
    const int base_cnt = (int)(s_tot64[0] > 0x7FFFFFFFull ? 0x7FFFFFFFull : s_tot64[0]);
    const int total_cnt = (int)(total64 > 0x7FFFFFFFll ? 0x7FFFFFFFll : total64), base_nt = s_tot[2], total_nt = s_tot[3];
    const bool overflow = total64 > (long long)idx_cap || total_nt > tile_cap || total_nt < 0;
    if (threadIdx.x == 0) {
        int run = 0;
        for (int c = MD_CLASSES - 1; c >= 0; --c) { const int n = s_hist[c]; s_cur[c] += run; run += n; }     // longest lists first
        if (part == 0) {
            hit_off[n_masks] = total_cnt;
            tile_off[n_masks] = overflow ? 0 : total_nt;
            status[2] = total_cnt;
            status[3] = total_nt;
            if (overflow) atomicOr(&status[0], 2);
        }
    }
    __syncthreads();
    // pass 2 (own slots only): exact offsets by a block scan per slot; the wave reserves its share of every class with one
    // atomic, positions inside it are ballot ranks
    int vbase = 0;                                      // lane c: next free position of class c in this wave's share
    if (lane < MD_CLASSES && wmine) vbase = atomicAdd(&s_cur[lane], wmine);
    int carry_cnt = base_cnt, carry_nt = base_nt;
    for (int g = slot0; g < slot1; ++g) {
        const int m = g * NT + (int)threadIdx.x;
        const int M = m < n_masks ? hit_count[m] : 0;
        const int nt = (M + CM3D_MEDOID_TILE - 1) / CM3D_MEDOID_TILE;
        const int ic = cm3d_wave_incl_scan(M), it = cm3d_wave_incl_scan(nt);
        __syncthreads();                                 // the previous slot's readers are done
        if (lane == 63) { s_wc[wave] = ic; s_wt[wave] = it; }
        __syncthreads();
        int wb_c = 0, wb_t = 0, tot_c = 0, tot_t = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const int a = s_wc[w], b = s_wt[w]; if (w < wave) { wb_c += a; wb_t += b; } tot_c += a; tot_t += b; }
        const int off = carry_cnt + wb_c + ic - M, t0 = carry_nt + wb_t + it - nt;
        carry_cnt += tot_c; carry_nt += tot_t;
        if (m < n_masks) { hit_off[m] = off; tile_off[m] = overflow ? 0 : t0; }
        const int ntl = (overflow || !desc) ? 0 : nt;   // tiles this lane lists
        int pos = 0;
#pragma unroll
        for (int c = 0; c < MD_UNI; ++c) {
            const uint64_t mk = __ballot(nt == c + 1);
            if (!mk) continue;
            if (nt == c + 1) pos = __builtin_amdgcn_readlane(vbase, c) + cm3d_mbcnt(mk) * (c + 1);
            vbase += lane == c ? (int)__popcll(mk) * (c + 1) : 0;
        }
        if (__ballot(nt > MD_UNI)) {
            const int cls = nt > MD_UNI ? md_class(nt) : -1;
#pragma unroll
            for (int c = MD_UNI; c < MD_CLASSES; ++c) {
                if (!__ballot(cls == c)) continue;
                const int v = cls == c ? nt : 0;
                const int inc = cm3d_wave_incl_scan(v);
                if (v) pos = __builtin_amdgcn_readlane(vbase, c) + inc - v;
                vbase += lane == c ? __builtin_amdgcn_readlane(inc, 63) : 0;
            }
        }
        for (int jt = 0; jt < ntl; ++jt) desc[pos + jt] = TileDesc{m, off, M, jt, t0 + jt};
    }
}
