// gsr_blend_fwd.hip -- forward alpha compositing.
//
// Same per-pixel arithmetic and control flow as the reference's renderCUDA
// (DGR/cuda_rasterizer/forward.cu:261-374): per pixel, front to back over the tile's
// depth-sorted list: power > 0 -> skip; alpha = min(0.99, o exp(power)); alpha < 1/255 -> skip; T (1 - alpha) < 1e-4 ->
// stop (that instance is NOT blended); n_contrib = 1-based list position of the last blended instance; C + T bg.
//
// What differs is who looks at what.  The reference (and round 1 of this library) walks the list in LOCK STEP: all
// pixels of a block evaluate the same instance.  For GauSTAR's ~3.6 px surface splats that leaves 10 of 64 lanes with
// anything to do.  Here each pixel walks ITS OWN candidates: gsr_mask.h reduces "which instances of this 64-entry unit
// can reach alpha >= 1/255 at this pixel" to one 64-bit word per (unit, pixel), and a lane just iterates the set bits
// of its words (v_ffbl_b32, clear lowest bit) -- different lanes of a wave are at different list positions at the same
// time; what they share is the tile's instance records, staged in LDS CH entries at a time and gathered per lane
// (ds_read_b128 with per-lane addresses).  A wave's trip count is the LARGEST per-pixel candidate count of its 8x8 block
// (config C: ~36 per block and chunk, against ~160 lock-step instance visits), and ~70 % of a trip's lanes are live.
//
//  * workgroup = tile (16x16 pixels, launch order = `order`, longest lists first), wave = 8x8 block, lane = pixel;
//  * the tile's list is depth-sorted right here first (gsr_sort.h; lists above 2 048 entries by their own kernels);
//  * per chunk of CH list positions: 256 threads park the chunk's records (make_rec: exp2-domain conic, opacity,
//    colour) in LDS -- and, when a backward pass may follow, in global memory in list order (rec_a/b/c) --; every wave
//    turns the 64 instances it just parked per fetch round into the candidate words of all four blocks (row interval
//    solve + 64x64 bit transpose, gsr_mask.h), to LDS for the walk and to global memory for the backward; a bit
//    summary of a lane's non-empty words lets an exhausted word be replaced in one step, never by a scan;
//  * the walk is a single branch-free loop: [replace an exhausted word] -> lowest set bit -> gather -> the reference's
//    tests -> blend;
//  * whenever a pixel moves on to a word of a new 64-entry segment its running (T, C) is stored as that segment's
//    snapshot: the state the backward's units resume from (gsr_blend_bwd.hip); the tile's first snapshot slot keeps
//    the final (T, C).
//
// Template over the number of colour channels C: 3 is the reference's NUM_CHANNELS (cuda_rasterizer/config.h:15);
// 6 renders TWO targets that share geometry (GauSTAR's RGB + depth-as-colour passes, refine.py:552 and :607) in one
// walk, 4 = RGB + one scalar target -- alpha, T, termination and n_contrib do not depend on colour, so channels 0-2 /
// 3-5 are bit-identical to two separate 3-channel renders.
#include "gsr_internal.h"
#include "gsr_mask.h"
#include "gsr_sort.h"
#include <cstdlib>

namespace gsr {

constexpr uint32_t LONG_LIST = 2048;   // lists above this get their candidate words from tile_mask_kernel (below)

template <int C, int CH>
__global__ void __launch_bounds__(256)
blend_fwd_kernel(int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ order,
                 uint32_t* point_list, const uint64_t* __restrict__ sort_keys, const float4* __restrict__ g0,
                 const float4* __restrict__ g1,
                 const float* __restrict__ feats, const float* __restrict__ bg, float* __restrict__ out_color,
                 float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ seg_off,
                 uint2* __restrict__ masks, float4* __restrict__ snap, float4* __restrict__ rec_a, float4* __restrict__ rec_b,
                 RecTail<C>* __restrict__ rec_c, float4* __restrict__ zero_ptr, uint32_t zero_n,
                 uint32_t* __restrict__ counters, uint32_t counters_tp, uint64_t* __restrict__ trace)
{
    const uint64_t t_start = trace ? wall_clock64() : 0;
    // Side job: the backward's accumulation table (48 B per Gaussian) has to be zero before blend_bwd runs.  When the
    // caller hands it over at forward time every workgroup clears its slice here instead of a separate fill (a 5 us blit
    // plus its dispatch) in front of the backward.
    if (zero_ptr != nullptr) {
        const uint32_t per = (zero_n + gridDim.x - 1u) / gridDim.x;
        const uint32_t i0 = blockIdx.x * per, i1 = min(zero_n, i0 + per);
        for (uint32_t i = i0 + threadIdx.x; i < i1; i += 256u) zero_ptr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    constexpr int NH = CH / 32;                                       // 32-bit mask words per lane and chunk
    constexpr int PT = CH / 256;                                      // list positions per thread and chunk
    static_assert(CH % 256 == 0 && NH <= 32, "a chunk is a whole number of 256-thread fetch rounds; nz is one dword");
    // LDS: ga[CH] | gb[CH] | gc[CH] (the staged instances, see make_rec) | mk[block][word][lane]; the sort in front of
    // the walk (below) uses the same bytes for its cross-wave stages.  C = 3 at CH = 512: 34 KB, four workgroups per CU.
    constexpr size_t GC_BYTES = (sizeof(RecTail<C>) * CH + 15) / 16 * 16;
    constexpr size_t REC_BYTES = 32 * CH + GC_BYTES, MK_BYTES = sizeof(uint32_t) * 4 * NH * 64;
    constexpr size_t SORT_BYTES = 2 * SORT_SMALL_CAP * 8;
    constexpr size_t LDS_BYTES = REC_BYTES + MK_BYTES > SORT_BYTES ? REC_BYTES + MK_BYTES : SORT_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    float4* const ga = reinterpret_cast<float4*>(smem);
    float4* const gb = ga + CH;
    RecTail<C>* const gc = reinterpret_cast<RecTail<C>*>(smem + 32 * CH);
    uint32_t (*mk)[NH][64] = reinterpret_cast<uint32_t(*)[NH][64]>(smem + REC_BYTES);
    const int tile = (int)order[blockIdx.x];
    // Second side job (fused forward): this tile's eight shard counters and eight scatter cursors live in a library-owned
    // block that has to be all zero again for the next view's preprocess; scatter, their last reader, is done.
    if (counters != nullptr && threadIdx.x < 2 * NSHARD)
        counters[(size_t)threadIdx.x * counters_tp + tile] = 0u;      // rows 0-7: counts, rows 8-15: cursors ([16][Tp])
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = tile % gx, ty = tile / gx;
    const int px = tx * TILE + (wave & 1) * SUB + (lane & 7), py = ty * TILE + (wave >> 1) * SUB + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    constexpr int SV = snap_vecs(C);
    const int pix_in_tile = 16 * (py - ty * TILE) + (px - tx * TILE);
    const bool keep = snap != nullptr;   // a backward pass may follow

    // (wave-uniform values are pinned to scalar registers: the per-chunk bookkeeping below then runs on the scalar unit)
    const uint2 rg = ranges[tile];
    const uint32_t list0 = __builtin_amdgcn_readfirstlane(rg.x);
    const uint32_t n = __builtin_amdgcn_readfirstlane(rg.y - rg.x);
    const uint32_t unit0 = __builtin_amdgcn_readfirstlane(seg_off[tile]);
    // Depth sort of this tile's list, right here (lists up to 2 048 entries; longer ones were sorted by tile_sort_big_kernel
    // before this launch).  As a kernel of its own the sort is latency-bound (key loads, cross-lane exchanges, barriers:
    // 25 us at a fraction of the vector ALU) and the blend then starts from a cold chip; inside the blend kernel one
    // tile's sort overlaps the other resident tiles' blending, and the sorted ids are read back while still in L2.
    // The kernel's span is its longest tile (a pixel's walk is serial), and co-resident waves share a SIMD's issue slots:
    // long lists get issue priority -- from their sort on -- so that they do not also run at 1/7 speed.
    if (n > 1024u) __builtin_amdgcn_s_setprio(3);
    else if (n > 704u) __builtin_amdgcn_s_setprio(2);
    else if (n > 448u) __builtin_amdgcn_s_setprio(1);
    if (sort_keys != nullptr) {
        if (n >= 1u && n <= 2048u) sort_small_tile(reinterpret_cast<uint64_t*>(smem), sort_keys + list0, point_list + list0, n);
        __syncthreads();   // ids visible to the four waves; the sort's LDS is free
    }
    const uint32_t* list = point_list + list0;
    const TransposeConsts tc(lane);
    const bool words_ready = n > LONG_LIST;   // (wave-uniform) candidate words already in global memory

    float T = 1.0f;
    float Cc[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) Cc[ch] = 0.f;
    uint32_t last = 0;
    bool done = !inside;
    uint32_t seg_cur = 0;   // segment (SNAP_SEG list positions) of the word this pixel consumed last

    // Two-stage software pipeline over the dependent gather (list -> id -> records): ids are fetched two chunks ahead,
    // records and mask words one chunk ahead, so no global-memory latency sits between a chunk's barrier and its walk.
    uint32_t gid_nxt[PT];
    float4 a_nxt[PT], b_nxt[PT];
    float col_nxt[PT][C];
    const auto fetch_ids = [&](uint32_t c0) {
#pragma unroll
        for (int q = 0; q < PT; q++) {
            const uint32_t k = c0 + q * 256 + threadIdx.x;
            gid_nxt[q] = k < n ? list[k] : 0xffffffffu;
        }
    };
    const auto fetch_records = [&]() {
#pragma unroll
        for (int q = 0; q < PT; q++) {
            const uint32_t gid = gid_nxt[q];
            a_nxt[q] = make_float4(0.f, 0.f, 1.f, 0.f);
            b_nxt[q] = make_float4(1.f, 0.f, -1.f, 0.f);   // tau = -1: no instance
#pragma unroll
            for (int ch = 0; ch < C; ch++) col_nxt[q][ch] = 0.f;
            if (gid != 0xffffffffu) {
                a_nxt[q] = g0[gid];
                b_nxt[q] = g1[gid];
                if constexpr (C % 2 == 0) {   // rows of an even channel count are 8-byte aligned
                    const float2* pf = reinterpret_cast<const float2*>(feats + (size_t)C * gid);
#pragma unroll
                    for (int ch = 0; ch < C; ch += 2) { const float2 v = pf[ch / 2]; col_nxt[q][ch] = v.x; col_nxt[q][ch + 1] = v.y; }
                } else {
#pragma unroll
                    for (int ch = 0; ch < C; ch++) col_nxt[q][ch] = feats[(size_t)C * gid + ch];
                }
            }
        }
    };
    fetch_ids(0);
    fetch_records();
    fetch_ids(CH);

    for (uint32_t c0 = 0; c0 < n; c0 += CH) {
        // every pixel of the tile saturated: stop (also the barrier that frees the LDS of the previous chunk)
        if (__syncthreads_or(!done) == 0) break;
        // ---- park the chunk's records and reduce them to per-pixel candidate words (gsr_mask.h): this wave holds, per
        // fetch round q, the 64 instances of unit 4 q + wave of the chunk, one per lane.  The words go to LDS for the
        // walk and, when a backward pass may follow, to global memory: the backward's units find their snapshots by them.
        const uint32_t u_lo = c0 >> 6;
#pragma unroll
        for (int q = 0; q < PT; q++) {
            // (positions past the end of the list park zeros: a lane without a candidate reads some slot of its current
            // word and multiplies it by a zero weight -- the slot has to hold finite numbers)
            const InstRec<C> rec = make_rec<C>(a_nxt[q], b_nxt[q], col_nxt[q]);
            ga[q * 256 + threadIdx.x] = rec.a;
            gb[q * 256 + threadIdx.x] = rec.b;
            gc[q * 256 + threadIdx.x] = rec.t;
            if (keep) {
                // the backward's units read their 64 records as three contiguous rows instead of gathering them again
                // through list -> id -> geometry state (a chain of three dependent trips to memory at the head of a unit)
                const uint32_t k = c0 + q * 256 + threadIdx.x;
                if (k < n) {
                    rec_a[list0 + k] = rec.a;
                    rec_b[list0 + k] = rec.b;
                    rec_c[list0 + k] = rec.t;
                }
            }
            if (c0 + q * 256 + wave * 64 < n) {   // (wave-uniform) the unit exists
                const int hw = 2 * (4 * q + wave);
                uint2* const gm = masks + ((size_t)(unit0 + u_lo + 4 * q + wave) * 4) * 64 + lane;
                if (words_ready) {
                    // long list: tile_mask_kernel has produced the words, one wave per unit, before this launch -- a
                    // 12 000-entry tile would otherwise spend its time on 184 units' worth of interval solves, serially
#pragma unroll
                    for (int blk = 0; blk < 4; blk++) {
                        const uint2 m = gm[blk * 64];
                        mk[blk][hw][lane] = m.x;
                        mk[blk][hw + 1][lane] = m.y;
                    }
                } else {
                    unit_masks(a_nxt[q], b_nxt[q], tx * TILE, ty * TILE, tc, [&](int blk, uint32_t lo, uint32_t hi) {
                        mk[blk][hw][lane] = lo;
                        mk[blk][hw + 1][lane] = hi;
                        if (keep) gm[blk * 64] = make_uint2(lo, hi);
                    });
                }
            }
        }
        fetch_records();
        fetch_ids(c0 + 2 * CH);
        __syncthreads();
        // which of this pixel's words are non-empty (a pixel that is done consumes none)
        uint32_t nz = 0;
#pragma unroll
        for (int hh = 0; hh < NH; hh++)
            if (c0 + 32u * hh < n) nz |= (mk[wave][hh][lane] != 0u ? 1u : 0u) << hh;
        nz = done ? 0u : nz;
        // ---- the walk: every lane through its own candidates.  (h, cur) = the word being consumed and its remaining
        // bits, (nh, nw) = the next non-empty word, read one step ahead, nz = the non-empty words behind it.  Everything
        // but the snapshot store is branch-free: lanes need a new word in different trips, and a conditional block that
        // almost every trip enters for a few lanes costs more than selects for all.
        uint32_t cur = 0, nw = 0;
        int h = 0, nh = 0;
        if (nz != 0u) { nh = __builtin_ctz(nz); nw = mk[wave][nh][lane]; nz &= nz - 1u; }
        while (true) {
            const bool need = cur == 0u;
            cur = need ? nw : cur;
            h = need ? nh : h;
            if (keep) {
                // first word of a new segment: the running (T, C) is the pixel's state at the segment's boundary (and at
                // every boundary it skipped) -- what the backward blend's units resume from (gsr_blend_bwd.hip)
                const uint32_t seg_new = (c0 + (uint32_t)h * 32u) / (uint32_t)SNAP_SEG;
                if (need && cur != 0u && seg_new != seg_cur) {
                    seg_cur = seg_new;
                    store_snapshot<C>(snap + ((size_t)(unit0 + seg_new * (SNAP_SEG / 64)) * 256 + pix_in_tile) * SV, T, Cc);
                }
            }
            {   // refill the look-ahead slot (reads a word every trip; consumed only by lanes that needed one)
                const int t = __builtin_ctz(nz | 0x80000000u) & (NH - 1);
                const uint32_t wnext = mk[wave][t][lane];
                nw = need ? (nz != 0u ? wnext : 0u) : nw;
                nh = need ? t : nh;
                nz = need ? (nz & (nz - 1u)) : nz;
            }
            if (__ballot(cur != 0u) == 0ull) break;
            // a lane without a candidate evaluates some record of its current word and drops it
            const bool act = cur != 0u;
            const int j = __builtin_ctz(cur | 0x80000000u);
            const uint32_t slot = (uint32_t)(h * 32 + j);
            cur &= cur - 1u;
            const float4 A = ga[slot], B = gb[slot];
            const RecTail<C> K = gc[slot];
            float col[C];
            col[0] = B.z; col[1] = B.w;
#pragma unroll
            for (int ch = 2; ch < C; ch++) col[ch] = K.c[ch - 2];
            const float dx = A.x - pxf, dy = A.y - pyf;
            const float power = pair_exp2_arg(A.z, A.w, B.x, dx, dy);   // exp2 domain, see conic_to_exp2
            const float alpha = fminf(ALPHA_MAX, B.y * __builtin_amdgcn_exp2f(power));
            const bool ok = act && power <= 0.0f && alpha >= ALPHA_MIN;
            const float test_T = T * (1.0f - alpha);
            const bool stop = ok && test_T < T_EPS;
            const bool upd = ok != stop;   // stop implies ok
            const float w = upd ? alpha * T : 0.0f;
#pragma unroll
            for (int ch = 0; ch < C; ch++) Cc[ch] += col[ch] * w;
            T = upd ? test_T : T;
            last = upd ? c0 + slot + 1u : last;
            // a pixel that terminates drops the rest of its candidates
            done = done || stop;
            cur = stop ? 0u : cur;
            nw = stop ? 0u : nw;
            nz = stop ? 0u : nz;
        }
    }
    if (inside) {
        const size_t pix = (size_t)W * py + px;
        const size_t HW = (size_t)H * W;
        final_T[pix] = T;
        n_contrib[pix] = last;
#pragma unroll
        for (int ch = 0; ch < C; ch++) out_color[ch * HW + pix] = Cc[ch] + T * bg[ch];
        // a tile with more than one segment: the first unit's snapshot slot (never used as a boundary) keeps the
        // final (T, C), from which the backward derives "colour behind a boundary" = C_final - C_snap
        if (keep && n > (uint32_t)SNAP_SEG) store_snapshot<C>(snap + ((size_t)unit0 * 256 + pix_in_tile) * SV, T, Cc);
    }
    if (trace && lane == 0) {   // last wave to finish wins the end stamp
        if (wave == 0) trace[2 * blockIdx.x] = t_start;
        atomicMax((unsigned long long*)&trace[2 * blockIdx.x + 1], (unsigned long long)wall_clock64());
    }
}

// Candidate words of the tiles above LONG_LIST entries (close-up views), one wave per unit: a 12 000-entry tile is 188
// independent waves here instead of 47 serial mask phases of its forward workgroup.
__global__ void __launch_bounds__(64)
tile_mask_kernel(int gx, const uint4* __restrict__ unit_info, const uint32_t* __restrict__ point_list,
                 const float4* __restrict__ g0, const float4* __restrict__ g1, uint2* __restrict__ masks)
{
    const uint32_t unit = blockIdx.x;
    const uint4 info = unit_info[unit];   // {tile, first entry, entries, first unit}
    const int tile = (int)info.x;
    const uint32_t n = info.z;
    if (n <= LONG_LIST) return;
    const int lane = threadIdx.x;
    const TransposeConsts tc(lane);
    const uint32_t k = (unit - info.w) * 64u + (uint32_t)lane;
    float4 a = make_float4(0.f, 0.f, 1.f, 0.f), b = make_float4(1.f, 0.f, -1.f, 0.f);   // tau = -1: no instance
    if (k < n) { const uint32_t gid = point_list[info.y + k]; a = g0[gid]; b = g1[gid]; }
    uint2* const gm = masks + (size_t)unit * 256 + lane;
    unit_masks(a, b, (tile % gx) * TILE, (tile / gx) * TILE, tc, [&](int blk, uint32_t lo, uint32_t hi) { gm[blk * 64] = make_uint2(lo, hi); });
}

template <int C>
static void launch_fwd_c(int W, int H, int U, uint32_t max_count, const float* bg, const float* feats, GeomState g, ImageState im,
                         BinState b, float* out_color, bool keep_masks, void* zero_ptr, size_t zero_bytes, uint32_t* counters,
                         bool sort_small, hipStream_t st)
{
    const Tiles t = tiles_of(W, H);
    if (max_count > LONG_LIST && U > 0)   // (such lists were sorted by the big-sort kernels before: launch_tile_sort)
        tile_mask_kernel<<<U, 64, 0, st>>>(t.gx, b.unit_info, b.point_list, g.g0, g.g1, b.masks);
    blend_fwd_kernel<C, FWD_CHUNK><<<t.T, 256, 0, st>>>(W, H, t.gx, im.ranges, im.order, b.point_list,
                                                        sort_small ? b.keys : nullptr, g.g0, g.g1, feats, bg,
                                                        out_color, im.final_T, im.n_contrib, im.seg_off, b.masks,
                                                        keep_masks ? b.snap : nullptr, b.rec_a, b.rec_b,
                                                        static_cast<RecTail<C>*>(b.rec_c), static_cast<float4*>(zero_ptr),
                                                        (uint32_t)(zero_bytes / 16), counters,
                                                        (uint32_t)shard_stride(t.T), g_trace);
}

void launch_blend_fwd(int C, int W, int H, int U, uint32_t max_count, const float* bg, const float* feats, GeomState g, ImageState im,
                      BinState b, float* out_color, bool keep_masks, void* zero_ptr, size_t zero_bytes, uint32_t* counters,
                      bool sort_small, hipStream_t st)
{
    if (C == 6) launch_fwd_c<6>(W, H, U, max_count, bg, feats, g, im, b, out_color, keep_masks, zero_ptr, zero_bytes, counters, sort_small, st);
    else if (C == 4) launch_fwd_c<4>(W, H, U, max_count, bg, feats, g, im, b, out_color, keep_masks, zero_ptr, zero_bytes, counters, sort_small, st);
    else launch_fwd_c<3>(W, H, U, max_count, bg, feats, g, im, b, out_color, keep_masks, zero_ptr, zero_bytes, counters, sort_small, st);
}

}  // namespace gsr
