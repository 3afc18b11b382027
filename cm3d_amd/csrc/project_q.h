// k_project_q: the projection launch for the QUAD layout of the raw rows (cm3d_hip.h, cm3d_sweep_prep) -- the same work per
// wave-chunk as k_project_hits (project.hip: sweep preparation, view wedges, approximate pre-test, exact chain, mask gather; the
// arithmetic helpers are shared, so every accepted pixel and every transformed coordinate has the same bits), rebuilt around what
// the instruction stream of k_project_hits showed (r04): a wave there spent 45 % of its time parked at s_waitcnt and issued 350
// scalar instructions per chunk beside 540 vector ones, most of both for things that do not change between chunks --
//   * the sweep's 24 coefficients were fetched again for every ROW (four scalar-load round trips per chunk, each behind a full
//     s_waitcnt), in four copies of the code (uniform / boundary chunk x first / later rows);
//   * every candidate mask cost a scalar load of its table entry (two per mask: pre-test and gather) with a wait behind it;
//   * 43 scalar registers lived in spill lanes.
// Here everything a wave needs for its whole life sits in REGISTER LANES or its LDS slice, loaded once:
//   * the frame's mask entries -- lane e of six registers holds entry e (two sets of 64 for up to 96 masks) -- and a
//     v_readlane with the entry number brings a field into a scalar register: one instruction, no memory, no wait;
//   * the current sweep's coefficients in one register (lane k = coefficient k), re-read only when the chunk's sweep changes;
//     a chunk that holds a sweep boundary runs the chain once per sweep and selects per row (the rare case costs more, the
//     common one nothing);
//   * rows arrive as quads: three 16-byte loads per lane, x / y / z of the lane's four rows already pairwise in consecutive
//     registers, which is the operand shape of the packed float32 instructions;
//   * hit words of up to 96 masks stay in registers (NPL = 1 or 3 planes): no LDS traffic for the 80-mask configuration.
// Work distribution (tickets, per-list counters, stealing inside the frame), the results and their layout are k_project_hits':
// k_frame_tables before it and k_compact_hits behind it do not know which of the two ran.
#pragma once

struct PqArgs {
    const float *raw; const float *inten; const float *sweep_xf; float4 *points_out; uint32_t *removed_bits;
    const int32_t *ft_all; const int4 *ment_all; const float *cams; const uint32_t *packed;
    uint32_t *hit_words; int32_t *hit_count; int32_t *wc_cnt; int32_t *queue; int32_t *wc_info; int32_t *grp; int32_t *frame_hits;
    float halfw, min_dist;
    int n_cams, W, H, nm_cap, nwc_max, n_points_total, n_frames, tpf, zstride;
    int stage;                                    // 99 = everything; the diagnostic build stops the chunk loop's stages earlier (timing only)
};

// the chunk loop's stage cuts and ablation bits exist in the diagnostic build only: in the product every test of `stage` was a loop-invariant
// condition the compiler kept as a 64-bit mask in two scalar registers (seven of them, in spill lanes)
#ifdef CM3D_DIAG
#define PQ_STAGE(a) ((a).stage)
#else
#define PQ_STAGE(a) 99
#endif
#ifndef PQ_OPT_LDSV
#define PQ_OPT_LDSV 1       // LDS slices addressed from a vector register
#endif
#ifndef PQ_OPT_SCALV
#define PQ_OPT_SCALV 1      // uniform operands of vector instructions in vector registers
#endif
#ifndef PQ_OPT_PTRV
#define PQ_OPT_PTRV 0       // per-lane base addresses in vector registers (no fewer spills on top of the other two, 1 % slower in flight)
#endif
#ifndef PQ_MIN_BLOCKS
#define PQ_MIN_BLOCKS 3                           // waves per SIMD the register budget is held to (168 registers): what the launch fills anyway
#endif

// sensor -> ego -> global for the lane's four rows, two rows per packed instruction: the k-sequential fma chains of k_sweep_xform /
// ph_xform element by element (same bits).  xf = the sweep's 24 coefficients in the wave's LDS slice: they come as VECTOR registers
// (six broadcast ds_read_b128, no vector-ALU time) -- measured on this chip (tools/ubench/valu_wall.hip, profiles/r04_valu_wall.txt): a
// packed fma with register operands occupies the SIMD 4.6 cycles for two rows, an fma with a SCALAR operand 4.5 for one, and every
// v_readlane that brings a coefficient into a scalar register 8.8 more.
static __device__ __forceinline__ void pq_xform4(const float *xf, const float (&q)[12], float (&o)[12])
{
    float c[CM3D_SWEEP_XF_STRIDE];
#pragma unroll
    for (int k = 0; k < CM3D_SWEEP_XF_STRIDE / 4; ++k) {
        const float4 t = reinterpret_cast<const float4 *>(xf)[k];
        c[4 * k] = t.x; c[4 * k + 1] = t.y; c[4 * k + 2] = t.z; c[4 * k + 3] = t.w;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f2 x = {q[2 * h], q[2 * h + 1]}, y = {q[4 + 2 * h], q[5 + 2 * h]}, z = {q[8 + 2 * h], q[9 + 2 * h]};
        f2 ax, ay, az, bx, by, bz;
        rot3_2(c, x, y, z, ax, ay, az);                             // sensor -> ego (rotate then translate), ego -> global (2d_to_3d.py:450-457)
        ax = ax + c[9]; ay = ay + c[10]; az = az + c[11];
        rot3_2(c + 12, ax, ay, az, bx, by, bz);
        bx = bx + c[21]; by = by + c[22]; bz = bz + c[23];
        o[2 * h] = bx.x; o[2 * h + 1] = bx.y; o[4 + 2 * h] = by.x; o[5 + 2 * h] = by.y; o[8 + 2 * h] = bz.x; o[9 + 2 * h] = bz.y;
    }
}

#ifdef CM3D_DIAG
#define PQ_STAMP(k) do { if ((a.stage & 255) == 100) { const unsigned long long t_ = ph_now(); acc[k] += t_ - t_prev; t_prev = t_; } } while (0)
#else
#define PQ_STAMP(k) do { } while (0)
#endif
#ifdef CM3D_DIAG
// one interval only (two s_memtime per chunk: little disturbance): stage & 255 == WHICH
#define PQ_IV_BEGIN(which) unsigned long long iv_t0_##which = 0; if ((a.stage & 255) == (which)) iv_t0_##which = ph_now()
#define PQ_IV_END(which) do { if ((a.stage & 255) == (which)) { acc[1] += ph_now() - iv_t0_##which; acc[2] += 1; } } while (0)
#else
#define PQ_IV_BEGIN(which) do { } while (0)
#define PQ_IV_END(which) do { } while (0)
#endif

#ifdef PQ_NUM_VGPR
#define PQ_VGPR_ATTR __attribute__((amdgpu_num_vgpr(PQ_NUM_VGPR)))
#else
#define PQ_VGPR_ATTR
#endif
template <int NPL, bool KEEP>
__global__ __launch_bounds__(PHK_THREADS, PQ_MIN_BLOCKS) PQ_VGPR_ATTR void k_project_q(const PqArgs a)
{
#ifdef CM3D_DIAG
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = (PQ_STAGE(a) & 255) >= 100 ? ph_now() : 0ull;
    const unsigned long long t_prev0 = t_prev;
#endif
    constexpr int NE = NPL == 1 ? 1 : 2;          // sets of 64 mask entries held in register lanes
    constexpr int NC = NPL == 1 ? 1 : 2;          // count registers: lane k of set s = hits of mask 64 s + k
    const int lane = cm3d_lane(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    __shared__ __align__(16) float s_cam_all[PHK_WAVES][CM3D_MAX_CAMS * CM3D_CAM_STRIDE];
    __shared__ __align__(16) float s_tab_all[PHK_WAVES][FT_WORDS - FT_WEDGE];
    __shared__ __align__(16) float s_xf_all[PHK_WAVES][2][CM3D_SWEEP_XF_STRIDE];     // the current sweep's coefficients, and a second set for a chunk with a sweep boundary
    // (the wave's LDS slices are addressed from a VECTOR register: an LDS instruction takes its address from one anyway, and five base
    // addresses in scalar registers were five of the 46 this kernel kept in spill lanes, each use a v_mov or a v_readlane)
#if PQ_OPT_LDSV
    const int wave_v = (int)(threadIdx.x >> 6);
#else
    const int wave_v = wave;
#endif
    float *const s_xf = s_xf_all[wave_v][0], *const s_xf2 = s_xf_all[wave_v][1];
    float *const s_cam = s_cam_all[wave_v];
    float(*const s_wedge)[8] = reinterpret_cast<float(*)[8]>(s_tab_all[wave_v]);
    float(*const s_apx)[16] = reinterpret_cast<float(*)[16]>(s_tab_all[wave_v] + (FT_APX - FT_WEDGE));
    const float qnan = __int_as_float(0x7FC00000);
    const int n_frames = a.n_frames, tpf = a.tpf, n_cams = a.n_cams;

    const int ticket = (int)blockIdx.x * PHK_WAVES + wave;
    if (ticket >= n_frames * tpf) return;
    const int slot = ticket / n_frames, f = (ticket - slot * n_frames + PHK_WAVES * slot) % n_frames;
    const int32_t *ft = a.ft_all + (size_t)f * FT_WORDS;
    int32_t *const taken = a.queue + (size_t)f * tpf;
    int list = slot, lists_left = PH_STEAL_LISTS;
#ifdef CM3D_DIAG
    int static_next = 0;
#endif
    auto draw = [&](int l) {
        int v = 0;
#ifdef CM3D_DIAG
        if (PQ_STAGE(a) & 512) return l == slot ? static_next++ : (1 << 20);     // timing only: fixed shares, no draws
#endif
        if (lane == 0) v = atomicAdd(&taken[l], 1);
        return v;
    };
    // everything that only needs the frame's number goes out at once (see k_project_hits)
    int draw_v = draw(slot);
    const int draw2_v = draw(slot);
    const float4 *cg = reinterpret_cast<const float4 *>(a.cams + (size_t)f * n_cams * CM3D_CAM_STRIDE);
    constexpr int CAM_Q = CM3D_MAX_CAMS * (CM3D_CAM_STRIDE / 4) / 64;
    float4 t_cam[CAM_Q];
#pragma unroll
    for (int q = 0; q < CAM_Q; ++q) t_cam[q] = lane + 64 * q < n_cams * (CM3D_CAM_STRIDE / 4) ? cg[lane + 64 * q] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 t_tab = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < (FT_WORDS - FT_WEDGE) / 4) t_tab = reinterpret_cast<const float4 *>(ft + FT_WEDGE)[lane];
    // lane c (<= CM3D_MAX_CAMS): first entry of camera c; read with v_readlane
    const int v_first = lane <= CM3D_MAX_CAMS ? ft[FT_CAMFIRST + lane] : 0;
    // lane k (< the frame's sweeps): frame-local first row of sweep k; INT_MAX beyond (ph_sweep_of_u)
    const int v_srow = lane < min(ft[5], PH_MAX_SWEEPS + 1) ? ft[FT_SROW + lane] : 0x7FFFFFFF;
    // the frame's mask entries in register lanes: lane e of set s = entry 64 s + e (k_frame_tables: sorted by camera)
    const int4 *ment = a.ment_all + (size_t)f * a.nm_cap * 2;
    int e_box[NE], e_ext[NE], e_kw[NE], e_off[NE], e_gbox[NE], e_gext[NE];
#pragma unroll
    for (int s = 0; s < NE; ++s) {
        const int e = min(64 * s + lane, a.nm_cap - 1);             // (entries past the frame's last are never looked at)
        const int4 m = ment[2 * e];
        const int2 g = *reinterpret_cast<const int2 *>(&ment[2 * e + 1]);
        e_box[s] = m.x; e_ext[s] = m.y; e_kw[s] = m.z; e_off[s] = m.w; e_gbox[s] = g.x; e_gext[s] = g.y;
    }
    auto ent = [&](const int (&v)[NE], int e) {                     // field of entry e (uniform) -> scalar register
        const int x0 = __builtin_amdgcn_readlane(v[0], e & 63);
        if (NE == 1) return x0;
        const int x1 = __builtin_amdgcn_readlane(v[NE - 1], e & 63);
        return e < 64 ? x0 : x1;
    };
    const int p0 = ft[0], n = ft[1], m0 = ft[2], nm = ft[3], sa = ft[4], ns = ft[5], bits_off = ft[6], nwc = ft[7];
    if (slot >= nwc) return;
    auto chunk_of = [&](int drawn_v, int from) {
        int c = from + __builtin_amdgcn_readfirstlane(drawn_v) * tpf;
        while (c >= nwc && lists_left > 0) {
            if (from == list) {
                --lists_left;
                list = list + 1 == tpf ? 0 : list + 1;
            }
            from = list;
            c = from + __builtin_amdgcn_readfirstlane(draw(from)) * tpf;
        }
        return c;
    };
    // Per-lane base addresses in VECTOR registers (pinned by the empty asm: the compiler would put the uniform part back into scalar
    // registers) for everything the chunk loop reads or writes with a per-lane address: a 64-bit base in scalar registers costs two of
    // the 102 the loop does not have (46 lived in spill lanes), here it is one v_lshl_add_u64 with the chunk's offset per use.
    const float *raw_lane = a.raw + ((size_t)p0 + 4 * lane) * 3;
    uint32_t *hw_lane = a.hit_words + (size_t)p0 + 4 * lane;
    uint32_t *rb_lane = a.removed_bits + (size_t)bits_off + (lane >> 3);
    int32_t *cnt_lane = a.wc_cnt + (size_t)f * a.nwc_max * a.nm_cap + lane;
    int32_t *grp_lane = a.grp + (size_t)f * a.zstride + lane;
#if PQ_OPT_PTRV
    asm volatile("" : "+v"(raw_lane), "+v"(hw_lane), "+v"(rb_lane), "+v"(cnt_lane), "+v"(grp_lane));
#endif
    // the lane's four rows of a chunk: x0..3 y0..3 z0..3 [+ the four intensities when the cloud is kept]
    struct Rows { float q[12]; float w[KEEP ? 4 : 1]; };
    auto load_rows = [&](Rows &r, int chunk) {
        const int cb = chunk * PH_WC, nvalid = min(PH_WC, n - cb);
#pragma unroll
        for (int k = 0; k < 12; ++k) r.q[k] = (k < 8) ? 1e30f : 0.f;     // (rows past the frame's end: never dropped, never live)
        if (KEEP) { r.w[0] = 0.f; r.w[KEEP ? 1 : 0] = 0.f; r.w[KEEP ? 2 : 0] = 0.f; r.w[KEEP ? 3 : 0] = 0.f; }
        if (4 * lane < nvalid) {
            const size_t row = (size_t)p0 + cb + 4 * lane;
            const float4 *p = reinterpret_cast<const float4 *>(raw_lane + (size_t)cb * 3);
            const float4 x = p[0], y = p[1], z = p[2];
            r.q[0] = x.x; r.q[1] = x.y; r.q[2] = x.z; r.q[3] = x.w; r.q[4] = y.x; r.q[5] = y.y; r.q[6] = y.z; r.q[7] = y.w;
            r.q[8] = z.x; r.q[9] = z.y; r.q[10] = z.z; r.q[11] = z.w;
            if (KEEP && a.inten) {
                const float4 t = *reinterpret_cast<const float4 *>(a.inten + row);
                r.w[0] = t.x; r.w[KEEP ? 1 : 0] = t.y; r.w[KEEP ? 2 : 0] = t.z; r.w[KEEP ? 3 : 0] = t.w;
            }
        }
    };
    Rows cur;
    load_rows(cur, slot);
    {
#pragma unroll
        for (int q = 0; q < CAM_Q; ++q) reinterpret_cast<float4 *>(s_cam)[lane + 64 * q] = t_cam[q];
        if (lane < (FT_WORDS - FT_WEDGE) / 4) reinterpret_cast<float4 *>(s_tab_all[wave_v])[lane] = t_tab;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const int apx_okmask = ft[FT_APXOK];
    const uint32_t cam_has = (uint32_t)ft[FT_CAMHAS];
    const float zmin = __int_as_float(ft[FT_ZMIN]);
    int chunk = chunk_of(draw_v, slot);
    if (chunk >= nwc) return;
    if (chunk != slot) load_rows(cur, chunk);
    int c_nxt = chunk_of(draw2_v, slot);
    PQ_STAMP(0);                                                    // start-up

    // the sweep whose coefficients the wave's LDS slice holds; -1: none yet
    int vxf_sweep = -1;
    auto load_xf = [&](int sw, float *dst) {                        // sw: sweep inside the frame; lane k fetches coefficient k
        if (lane < CM3D_SWEEP_XF_STRIDE) dst[lane] = a.sweep_xf[(size_t)(sa + sw) * CM3D_SWEEP_XF_STRIDE + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    int acc_cnt[NC];
#pragma unroll
    for (int s = 0; s < NC; ++s) acc_cnt[s] = 0;
    int32_t *const wc_info_f = a.wc_info + (size_t)f * a.nwc_max;
    int32_t *const grp_f = a.grp + (size_t)f * a.zstride;
    const int ngrp_max = (a.nwc_max + PH_GRP - 1) / PH_GRP;
    uint32_t pend_bits[NPL][PH_PT];
    int pend_cnt[NC];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int j = 0; j < PH_PT; ++j) pend_bits[pl][j] = 0u;
#pragma unroll
    for (int s = 0; s < NC; ++s) pend_cnt[s] = 0;
    int pend_chunk = -1, pend_drop = 0;
    auto flush_results = [&]() {
        if (pend_chunk < 0) return;
#ifdef CM3D_DIAG
        if (PQ_STAGE(a) & 1024) return;                                 // timing only: no result stores at all
#endif
        const int pcb = pend_chunk * PH_WC, pvalid = min(PH_WC, n - pcb);
        int32_t *cnt_row = cnt_lane + pend_chunk * a.nm_cap;              // (this lane's entry)
        bool mine = pend_cnt[0] != 0;
        if (NC == 2) mine |= pend_cnt[NC - 1] != 0;
        const bool any = __ballot(mine) != 0ull;
        if (any) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                if (pl * 32 >= nm) break;                           // uniform
                uint32_t *hw = hw_lane + (size_t)pl * a.n_points_total + pcb;
                if (pvalid >= PH_WC) {
                    *reinterpret_cast<u4u *>(hw) = (u4u){pend_bits[pl][0], pend_bits[pl][1], pend_bits[pl][2], pend_bits[pl][3]};
                } else {
#pragma unroll
                    for (int j = 0; j < PH_PT; ++j)
                        if (4 * lane + j < pvalid) hw[j] = pend_bits[pl][j];
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NC; ++s) {
            const int k = 64 * s + lane;
            if (k < nm) {
                cnt_row[64 * s] = pend_cnt[s];
                if (pend_cnt[s] && !(PQ_STAGE(a) & 256)) atomicAdd(&grp_lane[(pend_chunk / PH_GRP) * a.nm_cap + 64 * s], pend_cnt[s]);
            }
        }
        if (lane == 0) {
            wc_info_f[pend_chunk] = pend_drop | (any ? (int)0x80000000 : 0);
            if (pend_drop && !(PQ_STAGE(a) & 256)) atomicAdd(&grp_f[ngrp_max * a.nm_cap + pend_chunk / PH_GRP], pend_drop);
        }
    };
    int draw_from = list;
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): every start-up load has landed (tables, entries, first rows) before the loop
    // Uniform values that are only ever operands of vector instructions live in VECTOR registers (the empty asm hides that they are
    // uniform): each was a scalar register -- the kernel kept 46 of those in spill lanes, 8.8 cycles per reload --, and a vector
    // instruction with a scalar operand occupies the SIMD 4.5 cycles instead of 3 (tools/ubench/valu_wall.hip).
    int W = a.W, H = a.H;
    float min_dist = a.min_dist, halfw = a.halfw, zmin_v = zmin;
#if PQ_OPT_SCALV
    asm volatile("" : "+v"(W), "+v"(H), "+v"(min_dist), "+v"(halfw), "+v"(zmin_v));
#endif
#pragma unroll 1
    do {
        const int cb = chunk * PH_WC;
        const int nvalid = min(PH_WC, n - cb);
        PQ_IV_BEGIN(105);
        // ---- sweep preparation (reference :437-465): ego-box drop on the raw coordinates, sensor -> ego -> global
        int sw_lo = 0, sw_hi = 0;
        if (ns > 1) { sw_lo = ph_sweep_of_u(v_srow, cb); sw_hi = ph_sweep_of_u(v_srow, cb + nvalid - 1); }
        if (sw_lo != vxf_sweep) { load_xf(sw_lo, s_xf); vxf_sweep = sw_lo; }
        float g[12];
        if ((PQ_STAGE(a) & 255) >= 1) pq_xform4(s_xf, cur.q, g);
        else {
#pragma unroll
            for (int k = 0; k < 12; ++k) g[k] = cur.q[k];
        }
        if (sw_hi > sw_lo) {                                        // a sweep boundary inside the chunk: once more per further sweep, rows selected
            for (int sw = sw_lo + 1; sw <= sw_hi; ++sw) {
                load_xf(sw, s_xf2);
                float g2[12];
                pq_xform4(s_xf2, cur.q, g2);
                const int first = __builtin_amdgcn_readlane(v_srow, sw);          // frame-local first row of sweep sw
#pragma unroll
                for (int j = 0; j < PH_PT; ++j)
                    if (cb + 4 * lane + j >= first) { g[j] = g2[j]; g[4 + j] = g2[4 + j]; g[8 + j] = g2[8 + j]; }
            }
        }
        uint32_t nib = 0;
        // Nearly every chunk is full and holds no row inside the ego box: eight compares say so, and nothing is selected at all.
        // (A v_cndmask on VCC occupies the SIMD for 23 cycles on this chip -- tools/ubench/valu_wall.hip -- and the general path below
        // needs three per row.)
        bool boxed = false;
#pragma unroll
        for (int j = 0; j < PH_PT; ++j) boxed |= fabsf(cur.q[j]) < halfw && fabsf(cur.q[4 + j]) < halfw;
        if (nvalid < PH_WC || __ballot(boxed)) {
#pragma unroll
            for (int j = 0; j < PH_PT; ++j) {
                const bool live = 4 * lane + j < nvalid;
                const bool drop = live && fabsf(cur.q[j]) < halfw && fabsf(cur.q[4 + j]) < halfw;       // :442-445
                if (!live || drop) { g[j] = qnan; g[4 + j] = qnan; g[8 + j] = qnan; }
                nib |= (drop ? 1u : 0u) << j;
            }
        }
        if (KEEP) {
#pragma unroll
            for (int j = 0; j < PH_PT; ++j)
                if (4 * lane + j < nvalid) a.points_out[(size_t)p0 + cb + 4 * lane + j] = make_float4(g[j], g[4 + j], g[8 + j], cur.w[KEEP ? j : 0]);
        }
        f2 X[PH_NP], Y[PH_NP], Z[PH_NP];
#pragma unroll
        for (int h = 0; h < PH_NP; ++h) { X[h] = (f2){g[2 * h], g[2 * h + 1]}; Y[h] = (f2){g[4 + 2 * h], g[5 + 2 * h]}; Z[h] = (f2){g[8 + 2 * h], g[9 + 2 * h]}; }
        int drop_now = 0;
        if (__ballot(nib != 0u)) {
            // dropped rows of the chunk: four ballots (no cross-lane traffic: the wave sum through ds_bpermute was six dependent LDS
            // round trips, the merge of the bit nibbles three more -- a thousand cycles in every chunk with a dropped row)
#pragma unroll
            for (int j = 0; j < PH_PT; ++j) drop_now += (int)__popcll(__ballot((nib >> j) & 1u));
            // this chunk's 8 words of the removed-row bits: lane l holds bits 4 (l & 7) .. + 3 of word l >> 3; OR over each group of
            // 8 lanes with data-parallel-primitive moves (quad_perm [1,0,3,2], [2,3,0,1], then row_shl:4: lane i reads lane i + 4)
            int vv = (int)(nib << (4 * (lane & 7)));
            vv |= __builtin_amdgcn_update_dpp(0, vv, 0xB1, 0xF, 0xF, true);
            vv |= __builtin_amdgcn_update_dpp(0, vv, 0x4E, 0xF, 0xF, true);
            vv |= __builtin_amdgcn_update_dpp(0, vv, 0x104, 0xF, 0xF, true);
            int l7 = lane & 7;
            asm volatile("" : "+v"(l7));                            // (not a loop invariant: its mask would sit in two scalar registers for a path few chunks take)
            if (l7 == 0) rb_lane[8 * chunk] = (uint32_t)vv;
        }
        PQ_STAMP(1);                                                // rows arrive, transform, dropped rows
        PQ_IV_END(105);
        // ---- the previous chunk's results out, the next chunk's rows and the draw after it in
        {
            PQ_IV_BEGIN(106);
            flush_results();
            PQ_IV_END(106);
        }
        PQ_IV_BEGIN(107);
        Rows nxt;
        if (c_nxt < nwc) {
            load_rows(nxt, c_nxt);
            draw_from = list;
            draw_v = draw(list);
        }
        uint32_t bits[NPL][PH_PT];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int j = 0; j < PH_PT; ++j) bits[pl][j] = 0u;
        int mycnt[NC];
#pragma unroll
        for (int s = 0; s < NC; ++s) mycnt[s] = 0;
        PQ_STAMP(2);                                                // results out, next rows + draw issued
        PQ_IV_END(107);
        // ---- view wedges (wedge_setup): which cameras can see any of the wave's points
        PQ_IV_BEGIN(104);
        uint32_t vis = 0u;
#pragma unroll 1
        for (int cgp = 0; cgp < ((PQ_STAGE(a) & 255) >= 2 ? n_cams : 0); cgp += PH_CG) {
            if (!((cam_has >> cgp) & ((1u << PH_CG) - 1u))) continue;
            float inside[PH_CG];
#pragma unroll
            for (int q = 0; q < PH_CG; ++q) {
                const float *cn = s_wedge[cgp + q];
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
                if (__ballot(inside[q] >= 0.0f)) vis |= 1u << (cgp + q);
        }
        PQ_STAMP(3);                                                // view wedges
#pragma unroll 1
        while (vis) {
            const int c = __builtin_ctz(vis);
            vis &= vis - 1u;
            const int e0 = __builtin_amdgcn_readlane(v_first, c), e1 = __builtin_amdgcn_readlane(v_first, c + 1);
            if (e0 >= e1 || (PQ_STAGE(a) & 255) < 3) continue;
            // ---- approximate projection (wedge_setup): pixel codes for the grown boxes
            const bool pretest = (apx_okmask >> c) & 1;
            int pa[PH_PT];
            if (pretest) {
                const float *ap = s_apx[c];
#pragma unroll
                for (int h = 0; h < PH_NP; ++h) {
                    const f2 vx = X[h] - ap[0], vy = Y[h] - ap[1], vz = Z[h] - ap[2];
                    f2 xc = ap[3] * vx; xc = PK_FMA((f2)(ap[4]), vy, xc); xc = PK_FMA((f2)(ap[5]), vz, xc);
                    f2 yc = ap[6] * vx; yc = PK_FMA((f2)(ap[7]), vy, yc); yc = PK_FMA((f2)(ap[8]), vz, yc);
                    f2 zc = ap[9] * vx; zc = PK_FMA((f2)(ap[10]), vy, zc); zc = PK_FMA((f2)(ap[11]), vz, zc);
                    const f2 r = {__builtin_amdgcn_rcpf(zc.x), __builtin_amdgcn_rcpf(zc.y)};
                    const f2 ua = PK_FMA(ap[12] * xc, r, (f2)(ap[14] + 1.f)), va = PK_FMA(ap[13] * yc, r, (f2)(ap[15] + 1.f));
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int iu = (int)__builtin_amdgcn_fmed3f(ua[q], 0.f, 32001.f), iv = (int)__builtin_amdgcn_fmed3f(va[q], 0.f, 32001.f);
                        // code, or -1 unless zc > zmin: the sign of zmin - zc spread over the word and inverted (integer operations at
                        // full rate instead of a compare and a select on VCC; a NaN depth keeps a positive sign here -- it comes from the
                        // NaN of a dead row -- and is killed like a small one; if it were not, the exact chain would reject the point)
                        const int keep = __float_as_int(zmin_v - zc[q]) >> 31;        // all ones: zc > zmin
                        pa[2 * h + q] = ((iv << 16) | iu) | ~keep;
                    }
                }
            }
            PQ_STAMP(4);                                            // approximate projection
            bool projected = false;
            int px[PH_PT];
            uint32_t xw4[PH_PT], iv4[PH_PT];
            // the camera's entries in blocks of 32: candidate bits from the grown boxes, then the exact chain (once, when the first
            // block with a candidate is met), then the gather
#pragma unroll 1
            for (int eb = e0; eb < e1; eb += 32) {
                const int en = min(32, e1 - eb);
                uint32_t cmask = en >= 32 ? 0xFFFFFFFFu : ((1u << en) - 1u);
                if (pretest) {
                    cmask = 0u;
#pragma unroll 1
                    for (int i = 0; i < en; ++i) {
                        const us2 org = __builtin_bit_cast(us2, ent(e_gbox, eb + i));
                        const us2 ext = __builtin_bit_cast(us2, ent(e_gext, eb + i));
                        bool any = false;
#pragma unroll
                        for (int j = 0; j < PH_PT; ++j) {
                            const us2 d = __builtin_bit_cast(us2, pa[j]) - org;
                            const us2 m = __builtin_elementwise_min(d, ext);
                            any |= __builtin_bit_cast(uint32_t, m) == __builtin_bit_cast(uint32_t, d);
                        }
                        if (__ballot(any)) cmask |= 1u << i;
                    }
                    PQ_STAMP(5);                                    // grown-box tests
                    if (!cmask) continue;
                }
                if ((PQ_STAGE(a) & 255) < 4) break;
                if (!projected) {
                    projected = true;
                    const int cns = __builtin_amdgcn_readfirstlane((int)s_cam[c * CM3D_CAM_STRIDE + 54]);
                    const int cfl = __builtin_amdgcn_readfirstlane((int)s_cam[c * CM3D_CAM_STRIDE + 55]);
                    const float *cm = s_cam + c * CM3D_CAM_STRIDE;
                    if (cns == 2 && cfl == 5) project_quad<2, 5>(cm, cns, cfl, X, Y, Z, min_dist, W, H, px);
                    else if (cns == 1 && cfl == 1) project_quad<1, 1>(cm, cns, cfl, X, Y, Z, min_dist, W, H, px);
                    else if (cns == 3 && cfl == 10) project_quad<3, 10>(cm, cns, cfl, X, Y, Z, min_dist, W, H, px);
                    else project_quad<-1, 0>(cm, cns, cfl, X, Y, Z, min_dist, W, H, px);
                    int pxall = px[0];
#pragma unroll
                    for (int j = 1; j < PH_PT; ++j) pxall &= px[j];
                    if (!__ballot(pxall >= 0)) break;                // no point of the wave in this image: done with the camera
#pragma unroll
                    for (int j = 0; j < PH_PT; ++j) {
                        iv4[j] = (uint32_t)px[j] >> 16;
                        xw4[j] = ((uint32_t)px[j] & 0xFFFFu) >> 5;
                    }
                }
                PQ_STAMP(6);                                        // exact chain
                uint32_t rem = (PQ_STAGE(a) & 255) >= 5 ? cmask : 0u;
                // candidate masks of the block, up to PH_MB at a time: all their words are requested before the first is used (one
                // memory round trip per batch).  A batch is straight-line code for its number of masks (mask_batch<NB>): every load
                // has its use on the same path, so the compiler's s_waitcnt bookkeeping never carries a "pending" word register
                // around the chunk loop -- a conditional skip between a load and its use did, and every later write to such a
                // register (the pre-test of the NEXT camera) then waited for vmcnt(0): the prefetched rows, a whole HBM round trip,
                // once per chunk (r04).
                auto mask_batch = [&](auto nb_tag) {
                    constexpr int NB = decltype(nb_tag)::value;
                    int ei[NB];
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        ei[b] = eb + __builtin_ctz(rem);
                        rem &= rem - 1;
                    }
                    uint32_t word[NB][PH_PT];
                    int kb[NB];
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const int kw = ent(e_kw, ei[b]);
                        kb[b] = kw & 0xFFFF;
                        const uint32_t wcm = (uint32_t)kw >> 16;
                        const us2 org = __builtin_bit_cast(us2, ent(e_box, ei[b]));
                        const us2 ext = __builtin_bit_cast(us2, ent(e_ext, ei[b]));
                        const char *mw = reinterpret_cast<const char *>(a.packed) + ((long long)ent(e_off, ei[b]) << 2);
#pragma unroll
                        for (int j = 0; j < PH_PT; ++j) {
                            const us2 d = __builtin_bit_cast(us2, px[j]) - org;
                            const us2 m = __builtin_elementwise_min(d, ext);
                            word[b][j] = 0u;
                            if (__builtin_bit_cast(uint32_t, m) == __builtin_bit_cast(uint32_t, d))
                                word[b][j] = *reinterpret_cast<const uint32_t *>(mw + ((iv4[j] * wcm + xw4[j]) << 2));
                        }
                    }
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        int cnt = 0;
                        uint32_t hb[PH_PT];
#pragma unroll
                        for (int j = 0; j < PH_PT; ++j) {
                            hb[j] = (word[b][j] >> (px[j] & 31)) & 1u;
                            cnt += (int)__popcll(__ballot(hb[j] != 0u));
                        }
                        const int sh = kb[b] & 31;
                        if (NPL == 1 || kb[b] < 32) {
#pragma unroll
                            for (int j = 0; j < PH_PT; ++j) bits[0][j] |= hb[j] << sh;
                        } else if (kb[b] < 64) {
#pragma unroll
                            for (int j = 0; j < PH_PT; ++j) bits[NPL > 1 ? 1 : 0][j] |= hb[j] << sh;
                        } else {
#pragma unroll
                            for (int j = 0; j < PH_PT; ++j) bits[NPL > 2 ? 2 : 0][j] |= hb[j] << sh;
                        }
                        if (NC == 1 || kb[b] < 64) mycnt[0] += lane == kb[b] ? cnt : 0;
                        else mycnt[NC - 1] += lane == kb[b] - 64 ? cnt : 0;
                    }
                };
                PQ_IV_BEGIN(103);
                while (rem) {
                    const int nb = __builtin_popcount(rem);
                    if (nb >= 4) mask_batch(std::integral_constant<int, 4>());
                    else if (nb == 3) mask_batch(std::integral_constant<int, 3>());
                    else if (nb == 2) mask_batch(std::integral_constant<int, 2>());
                    else mask_batch(std::integral_constant<int, 1>());
                }
                PQ_IV_END(103);
            }
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int j = 0; j < PH_PT; ++j) pend_bits[pl][j] = bits[pl][j];
#pragma unroll
        for (int s = 0; s < NC; ++s) { pend_cnt[s] = mycnt[s]; acc_cnt[s] += mycnt[s]; }
        PQ_STAMP(7);                                                // gather (the rest of the camera loop)
        PQ_IV_END(104);
        pend_chunk = chunk;
        pend_drop = drop_now;
        {
            PQ_IV_BEGIN(101);
            cur = nxt;
            chunk = c_nxt;
            if (c_nxt < nwc) c_nxt = chunk_of(draw_v, draw_from);
#ifdef CM3D_DIAG
            if ((PQ_STAGE(a) & 255) == 101) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            PQ_IV_END(101);
        }
        PQ_STAMP(0);                                                // wait for the draw (and the rows requested before it)
    } while (chunk < nwc);
    flush_results();
    int tot = 0;
#pragma unroll
    for (int s = 0; s < NC; ++s) {
        const int k = 64 * s + lane;
        if (k < nm && acc_cnt[s]) atomicAdd(&a.hit_count[m0 + k], acc_cnt[s]);
        tot += k < nm ? acc_cnt[s] : 0;
    }
    tot = cm3d_wave_sum(tot);
    if (lane == 0 && tot) atomicAdd(&a.frame_hits[f], tot);
#ifdef CM3D_DIAG
    if ((PQ_STAGE(a) & 255) >= 100 && lane == 0) {
        acc[0] = ph_now() - t_prev0;
        // (per-wave slots, plain stores: thousands of atomics on one address at the end of the early waves held up the draws of the late ones)
        const int wid = (int)blockIdx.x * PHK_WAVES + wave;
        if ((PQ_STAGE(a) & 255) == 100) {
            for (int k = 0; k < 8; ++k) atomicAdd(&g_ph_stamp[k], acc[k]);
            atomicAdd(&g_ph_count[0], 1ull);
        } else if (wid < PH_DIAG_WAVES) {
            g_ph_wave[3 * wid] = acc[0]; g_ph_wave[3 * wid + 1] = acc[1]; g_ph_wave[3 * wid + 2] = acc[2];
        }
    }
#endif
}
