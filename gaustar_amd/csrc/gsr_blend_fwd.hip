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
#include "gsr_plan.h"
#include <cstdlib>

namespace gsr {


// MODE 0: one workgroup per tile (launch order = `order`): the list is sorted and walked here, front to back.  That is every
// view whose longest list stays below SPLIT_FROM entries.  A longer list -- a surface seen edge-on stacks thousands of thin
// splats onto a tile without saturating it, and a pixel's walk is serial: 11 787 entries took one workgroup 514 us; config B's
// two 2 100-entry pole tiles ran alone for 60 us of a 114 us launch -- makes the view SPLIT: MODE 2, ONE launch of
// part_capacity + T workgroups.  The first part_capacity are PART workers: every list above PART_FROM entries (sorted before
// the launch) is walked in parts of one chunk, one workgroup per part, every part starting from T = 1, C = 0 and independent of
// all others.  A part that has stored its result draws a ticket at its tile (release fence, agent-scope atomic); the one that
// draws the LAST ticket COMBINES the tile's parts in order (acquire fence first) -- T_in C_in -> C_in + T_in C_part,
// T_in T_part per pixel, re-basing the parts' per-unit snapshots the same way -- and writes the tile's pixels.  Nobody waits for
// anybody, so no assumption about dispatch order is made; the tile's own workgroup (second part of the grid) does nothing.
// (MODE 1: the part workers as a launch of their own, without tickets -- kept for the two-launch comparison build.)
// A part is folded in like that only for pixels that cannot have terminated inside it -- the product of ALL its (1 - alpha)
// keeps T above 1e-4 (then every prefix does) and the part itself did not stop -- the others walk that one chunk again
// with their true state, exactly as a short list would (a pixel terminates once, so at most one chunk per pixel).
// PLANNED (MODE 0 only; gsr_internal.h "planned binning"): `ranges` / `order` / `seg_off` are the PLAN's -- {first entry,
// capacity} of the tile's bucket, the plan's launch order, the bucket's first unit --, the list's length is what preprocess
// claimed on the tile's cursor, and the list is always sorted here (a planned bucket holds at most 2 048 entries).  The tile's
// workgroup also writes the backward's unit table for ALL units of the bucket (a unit past the list's end is an empty work
// item there), leaves {first, first + count} in the image state like the exact path's scan, and hands the cursor back zeroed.
// A view that did not fit its plan (flag word == this view's token) is left alone: the host renders it the exact way.
template <int C, int CH, int MODE, bool PLANNED = false>
__global__ void __launch_bounds__(256)
blend_fwd_kernel(int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ order,
                 uint32_t* point_list, const uint64_t* __restrict__ sort_keys, const float4* __restrict__ g0,
                 const float4* __restrict__ g1,
                 const float* __restrict__ feats, const float* __restrict__ bg, float* __restrict__ out_color,
                 float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ seg_off,
                 uint2* __restrict__ masks, float4* __restrict__ snap, float4* __restrict__ rec_a, float4* __restrict__ rec_b,
                 RecTail<C>* __restrict__ rec_c, const uint2* __restrict__ part_list, const uint32_t* __restrict__ totals,
                 float4* __restrict__ part_fin, uint32_t* __restrict__ part_last, uint32_t* __restrict__ part_ticket,
                 uint32_t part_grid, bool parts_sort, uint32_t split_n, bool keep,
                 float4* __restrict__ zero_ptr,
                 uint32_t zero_n, uint32_t* __restrict__ counters, uint32_t counters_tp, uint64_t* __restrict__ trace,
                 PlanRun plan, PlanJob job)
{
    static_assert(!PLANNED || MODE == 0, "planned views are never split");
    const uint64_t t_start = trace ? wall_clock64() : 0;
    // (MODE 2) the first part_grid workgroups are part workers, the others the tiles in launch order
    const bool is_part = MODE == 1 || (MODE == 2 && blockIdx.x < part_grid);
    // (MODE 0, exact path, the caller keeps a plan for this camera: workgroup 0 -- dispatched first -- builds the plan of the
    // camera's next view instead of blending a tile, gsr_plan.h; the tiles follow from workgroup 1 on)
    // (PLANNED, round 6: the same workgroup in a planned view, from the view's cursors -- it also reports the view's verdict)
    const bool has_job = MODE == 0 && job.enabled != 0u;
    const uint32_t tile_block = MODE == 2 ? blockIdx.x - part_grid : blockIdx.x - (has_job ? 1u : 0u);
    // Side job: the backward's accumulation table (48 B per Gaussian) has to be zero before blend_bwd runs.  When the
    // caller hands it over at forward time every workgroup clears its slice here instead of a separate fill (a 5 us blit
    // plus its dispatch) in front of the backward.
    if (MODE != 1 && zero_ptr != nullptr) {
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
    if constexpr (MODE == 0) {
        static_assert(LDS_BYTES >= 4 * PLAN_LDS_T, "the plan job stages the tile counts in the kernel's LDS");
        if (has_job && blockIdx.x == 0) {
            if constexpr (PLANNED) {
                // the verdict on the view, for the host (which waits for nothing else): first thing the first workgroup does.
                // A view that outgrew its plan leaves no plan behind: the exact view the host renders instead does.
                const bool misfit = plan.sync[9 * PLAN_SYNC_STRIDE] == plan.token;
                if (threadIdx.x == 0) {
                    plan.host_pad[0] = misfit ? 2u : 1u;
                    __hip_atomic_store(&plan.host_pad[1], plan.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                if (misfit) return;
            }
            plan_build_block<256>(job, ranges, order, reinterpret_cast<uint32_t*>(smem));
            return;
        }
    }
    if (is_part && blockIdx.x >= totals[6]) return;   // parts of this view (scatter_kernel lists them)
    uint32_t part_c0 = 0;
    int tile;
    if (!is_part) {
        tile = (int)order[tile_block];
    } else {
        const uint2 part = part_list[blockIdx.x];
        tile = (int)part.x;
        part_c0 = __builtin_amdgcn_readfirstlane(part.y);
    }
    // Second side job (fused forward): this tile's eight shard counters and eight scatter cursors live in a library-owned
    // block that has to be all zero again for the next view's preprocess; scatter, their last reader, is done.
    if (!is_part && counters != nullptr && threadIdx.x < 2 * NSHARD)
        counters[(size_t)threadIdx.x * counters_tp + tile] = 0u;      // rows 0-7: counts, rows 8-15: cursors ([16][Tp])
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = tile % gx, ty = tile / gx;
    const int px = tx * TILE + (wave & 1) * SUB + (lane & 7), py = ty * TILE + (wave >> 1) * SUB + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    constexpr int SV = snap_vecs(C);
    const int pix_in_tile = 16 * (py - ty * TILE) + (px - tx * TILE);
    const bool snaps = keep && snap != nullptr;   // a backward pass may follow

    // (wave-uniform values are pinned to scalar registers: the per-chunk bookkeeping below then runs on the scalar unit)
    const uint2 rg = ranges[tile];
    const uint32_t list0 = __builtin_amdgcn_readfirstlane(rg.x);
    uint32_t n_ = rg.y - rg.x;
    if constexpr (PLANNED) {
        const bool misfit = plan.sync[9 * PLAN_SYNC_STRIDE] == plan.token;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            // the verdict on the view, for the host (which waits for nothing else): first thing the first workgroup does
            // (a launch that carries the plan job: that workgroup, above)
            plan.host_pad[0] = misfit ? 2u : 1u;
            __hip_atomic_store(&plan.host_pad[1], plan.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // The count stays where it is -- nobody writes this view's cursors during the launch, so the plan job reads all of them
        // whenever it gets to them --; what is handed back zeroed is the tile's cursor in the OTHER block, on which the view before
        // this one claimed (its launches have retired) and the next one will.
        n_ = plan.cursor[(size_t)tile * PLAN_CURSOR_STRIDE];
        if (threadIdx.x == 0) plan.cursor_other[(size_t)tile * PLAN_CURSOR_STRIDE] = 0u;
        if (misfit) return;
    }
    const uint32_t n = __builtin_amdgcn_readfirstlane(n_);
    const uint32_t unit0 = __builtin_amdgcn_readfirstlane(seg_off[tile]);
    if constexpr (PLANNED) {
        if (threadIdx.x == 0) { plan.im_ranges[tile] = make_uint2(list0, list0 + n); plan.im_seg_off[tile] = unit0; }
        if (keep)
            for (uint32_t u = threadIdx.x; u < (rg.y >> 6); u += 256u) plan.unit_info[unit0 + u] = make_uint4((uint32_t)tile, list0, n, unit0);
    }
    const bool long_list = n > split_n;   // (wave-uniform) blended in parts: split_n = PART_FROM in views that split at all
    if (MODE == 2 && !is_part && long_list) return;   // its parts do everything, the last of them the combine
    // Depth sort of this tile's list, right here (lists up to 2 048 entries; longer ones were sorted by tile_sort_big_kernel
    // before this launch).  As a kernel of its own the sort is latency-bound (key loads, cross-lane exchanges, barriers:
    // 25 us at a fraction of the vector ALU) and the blend then starts from a cold chip; inside the blend kernel one
    // tile's sort overlaps the other resident tiles' blending, and the sorted ids are read back while still in L2.
    // The kernel's span is its longest tile (a pixel's walk is serial), and co-resident waves share a SIMD's issue slots:
    // long lists get issue priority -- from their sort on -- so that they do not also run at 1/7 speed.
#ifdef GSR_FWD_PHASES   // devtool build: wave 0's stamps {start, sorted, then per chunk: parked + words done, walked} behind the traces
    uint64_t* const ph = (trace && !is_part && threadIdx.x == 0) ? trace + 2 * (size_t)gridDim.x + 2 * 65536 + (size_t)tile_block * 16 : nullptr;
    int ph_i = 0;
#define FWD_PHASE() do { if (ph && ph_i < 16) ph[ph_i++] = wall_clock64(); } while (0)
    if (ph) { ph[ph_i++] = t_start; }
    FWD_PHASE();
#else
#define FWD_PHASE() do { } while (0)
#endif
#ifndef GSR_FWD_SORT_STAGGER
#define GSR_FWD_SORT_STAGGER 1
#endif
#ifndef GSR_FWD_TAIL_PRIO
#define GSR_FWD_TAIL_PRIO 704
#endif
    const auto prio_by_length = [&]() {
        // (round 6: the launch does not end with its longest tiles -- those finish at 70-73 us of 77-81 -- but with tiles of its
        // SECOND generation: 20-30 us tiles that start when a first-generation slot frees up, ~50 us in, tools/fwd_phases.py
        // "ends by rank".  From launch position 704 on a non-empty tile runs at the highest priority from start to end:
        // - 1 us per view, in-process A/B at 512 / 640 / 768 / 896 / 1 024: - 0.6 / - 1.4 / - 1.3 / - 0.5 / - 0.2 us; 0 = off)
        if (GSR_FWD_TAIL_PRIO && tile_block >= (uint32_t)GSR_FWD_TAIL_PRIO && n > 0u) { __builtin_amdgcn_s_setprio(3); return; }
        if (n > 1024u) __builtin_amdgcn_s_setprio(3);
        else if (n > 704u) __builtin_amdgcn_s_setprio(2);
        else if (n > 448u) __builtin_amdgcn_s_setprio(1);
    };
#if GSR_FWD_SORT_STAGGER
    // (the launch order's first thousand tiles are its longest and all sort at once, four per CU, on one LDS pipe; distinct
    // priorities for the sort by the tile's quarter of the launch order let a CU's sorters finish one after the other and move
    // on to phases with another resource mix: blend_fwd 86.5 -> 85.0 us)
    if (GSR_FWD_TAIL_PRIO && tile_block >= (uint32_t)GSR_FWD_TAIL_PRIO && n > 0u) __builtin_amdgcn_s_setprio(3);
    else switch ((tile_block >> 8) & 3u) {
        case 0: __builtin_amdgcn_s_setprio(3); break;
        case 1: __builtin_amdgcn_s_setprio(2); break;
        case 2: __builtin_amdgcn_s_setprio(1); break;
        default: __builtin_amdgcn_s_setprio(0); break;
    }
#else
    prio_by_length();
#endif
    // (the sorted keys are still in LDS when the tile asks for the ids of its first two chunks: read there, they spare the walk's
    // first dependent trip to memory -- the ids were written to the list a moment ago, for the later chunks and the backward)
    const uint64_t* ids_lds = nullptr;
    if (!is_part) {
        if (sort_keys != nullptr) {
#ifdef GSR_FWD_PHASES
            if (n >= 1u && n <= 2048u) ids_lds = sort_small_tile(reinterpret_cast<uint64_t*>(smem), sort_keys + list0, point_list + list0, n, [&]() { FWD_PHASE(); });
#else
            if (n >= 1u && n <= 2048u) ids_lds = sort_small_tile(reinterpret_cast<uint64_t*>(smem), sort_keys + list0, point_list + list0, n);
#endif
            __syncthreads();   // ids visible to the four waves; the sort's LDS is free
            FWD_PHASE();
        }
#if GSR_FWD_SORT_STAGGER
        prio_by_length();
#endif
    } else if (parts_sort && sort_keys != nullptr && n <= SORT_SMALL_CAP) {
        // A part sorts the WHOLE list of its tile for itself (every part of the tile writes the same ids to the same places): no
        // workgroup waits for another before its walk, and no sort launch stands in front of the blend.  Only in views whose
        // longest list fits the in-kernel sort (parts_sort): a view with a list above 2 048 entries needs launch_tile_sort's merge
        // passes anyway, and then they sort every split list (config B: blend 85 us behind a 27 us sort launch, against 102 us
        // with its shorter split lists sorted here five times over; an in-place LDS network for 4 096 keys was tried as well:
        // 78 stages x 64 KB of LDS traffic, ~45 us per part).
        sort_small_tile(reinterpret_cast<uint64_t*>(smem), sort_keys + list0, point_list + list0, n);
        __syncthreads();
    }
    const uint32_t* list = point_list + list0;
    const TransposeConsts tc(lane);

    float T = 1.0f;
    float Cc[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) Cc[ch] = 0.f;
    uint32_t last = 0;
    bool done = !inside;
    // segment (SNAP_SEG list positions) of the word this pixel consumed last.  A part that is not the tile's first starts
    // "nowhere": the first word of ANY of its units, the first one included, opens a segment and leaves a snapshot
    uint32_t seg_cur = (is_part && part_c0 != 0u) ? 0xffffffffu : 0u;

    // Two-stage software pipeline over the dependent gather (list -> id -> records): ids are fetched two chunks ahead,
    // records and mask words one chunk ahead, so no global-memory latency sits between a chunk's barrier and its walk.
    uint32_t gid_nxt[PT];
    float4 a_nxt[PT], b_nxt[PT];
    float col_nxt[PT][C];
    const auto fetch_ids = [&](uint32_t c0) {
#pragma unroll
        for (int q = 0; q < PT; q++) {
            const uint32_t k = c0 + q * 256 + threadIdx.x;
            gid_nxt[q] = k < n ? (ids_lds != nullptr ? (uint32_t)ids_lds[k] : list[k]) : 0xffffffffu;
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
    // ---- park the chunk's records (held in a_nxt / b_nxt / col_nxt) and reduce them to per-pixel candidate words
    // (gsr_mask.h): this wave holds, per fetch round q, the 64 instances of unit 4 q + wave of the chunk, one per lane.
    // The words go to LDS for the walk and, when a backward pass may follow (or the chunk is a part: the combining
    // workgroup reads them), to global memory.  words_known: the part has left them there already.
    const auto park = [&](uint32_t c0, bool words_known, bool words_out, bool recs_out) {
        const uint32_t u_lo = c0 >> 6;
#pragma unroll
        for (int q = 0; q < PT; q++) {
            // (positions past the end of the list park zeros: a lane without a candidate reads some slot of its current
            // word and multiplies it by a zero weight -- the slot has to hold finite numbers)
            const InstRec<C> rec = make_rec<C>(a_nxt[q], b_nxt[q], col_nxt[q]);
            ga[q * 256 + threadIdx.x] = rec.a;
            gb[q * 256 + threadIdx.x] = rec.b;
            gc[q * 256 + threadIdx.x] = rec.t;
            if (recs_out) {
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
                if (words_known) {
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
                        if (words_out) gm[blk * 64] = make_uint2(lo, hi);
                    });
                }
            }
        }
    };
    // which of this pixel's words of the parked chunk are non-empty
    const auto nonempty_words = [&](uint32_t c0) {
        uint32_t nz = 0;
#pragma unroll
        for (int hh = 0; hh < NH; hh++)
            if (c0 + 32u * hh < n) nz |= (mk[wave][hh][lane] != 0u ? 1u : 0u) << hh;
        return nz;
    };
    // ---- the walk: every lane through its own candidates of the parked chunk.  (h, cur) = the word being consumed and its
    // remaining bits, (nh, nw) = the next non-empty word, read one step ahead, nz = the non-empty words behind it.
    // Everything but the snapshot store is branch-free: lanes need a new word in different trips, and a conditional block
    // that almost every trip enters for a few lanes costs more than selects for all.  Returns whether the pixel stopped.
#ifndef GSR_FWD_PIPE
#define GSR_FWD_PIPE 1
#endif
#if GSR_FWD_PIPE
    // Round 5: the gather of the NEXT candidate's record is requested before the arithmetic of the current one (explicit
    // ds_reads: where they are issued is the point).  A lane's state is kept EAGER for that: (h, cur) = the word being consumed --
    // non-empty at every trip's start unless the lane is out of candidates --, (nh, nw) = the next non-empty word, already in a
    // register, nz = the non-empty words behind it; a trip that takes a word's last bit moves on to the next word at its END,
    // so the next candidate's position is known at the start of every trip.  The walk of a tile's slowest block is a chain of
    // trips (tools/fwd_phases.py); this takes the LDS round trip out of every link.
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    using TailRegs = std::conditional_t<sizeof(RecTail<C>) == 4, float, std::conditional_t<sizeof(RecTail<C>) == 8, f32x2_, f32x4_>>;
    struct RecRegs { f32x4_ a, b; TailRegs k; };
    const uint32_t ga_addr = (uint32_t)(size_t)ga, gc_addr = (uint32_t)(size_t)gc;
    const auto request = [&](RecRegs& r, uint32_t slot) {
        const uint32_t ad = ga_addr + slot * 16u;
        asm volatile("ds_read_b128 %0, %1" : "=&v"(r.a) : "v"(ad) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(r.b) : "v"(ad), "n"(16 * CH) : "memory");
        const uint32_t ak = gc_addr + slot * (uint32_t)sizeof(RecTail<C>);
        if constexpr (sizeof(RecTail<C>) == 4) asm volatile("ds_read_b32 %0, %1" : "=&v"(r.k) : "v"(ak) : "memory");
        else if constexpr (sizeof(RecTail<C>) == 8) asm volatile("ds_read_b64 %0, %1" : "=&v"(r.k) : "v"(ak) : "memory");
        else asm volatile("ds_read_b128 %0, %1" : "=&v"(r.k) : "v"(ak) : "memory");
    };
    const auto walk = [&](uint32_t c0, uint32_t nz) {
        const uint32_t mk_lane = (uint32_t)(size_t)&mk[wave][0][lane];   // LDS byte address of this lane's word 0 (+ 256 per word)
        uint32_t cur = 0, nw = 0;
        int h = 0, nh = 0;
        bool stopped = false;
        if (nz != 0u) { h = __builtin_ctz(nz); cur = mk[wave][h][lane]; nz &= nz - 1u; }
        if (nz != 0u) { nh = __builtin_ctz(nz); nw = mk[wave][nh][lane]; nz &= nz - 1u; }
        uint32_t slot = (uint32_t)(h * 32 + __builtin_ctz(cur | 0x80000000u));
        RecRegs rec0, rec1;
        request(rec0, slot);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rec0.a), "+v"(rec0.b), "+v"(rec0.k) : : "memory");
        // one trip: the candidate whose record is in `rec`; the next candidate's record is requested into `nxt`
        const auto trip = [&](RecRegs& rec, RecRegs& nxt) {
            const bool act = cur != 0u;   // a lane without a candidate evaluates some record and drops it
            if (snaps || is_part) {   // (a part always leaves them: the combining workgroup finds its way by them)
                // first word of a new segment: the running (T, C) is the pixel's state at the segment's boundary (and at
                // every boundary it skipped) -- what the backward blend's units resume from (gsr_blend_bwd.hip)
                const uint32_t seg_new = (c0 + (uint32_t)h * 32u) / (uint32_t)SNAP_SEG;
                if (act && seg_new != seg_cur) {
                    seg_cur = seg_new;
                    store_snapshot<C>(snap + ((size_t)(unit0 + seg_new * (SNAP_SEG / 64)) * 256 + pix_in_tile) * SV, T, Cc);
                }
            }
            // where the next candidate lies: behind the current one in this word, or the first of the next word
            const uint32_t rest = cur & (cur - 1u);
            const bool exhausted = rest == 0u;
            const uint32_t ncur = exhausted ? nw : rest;
            const int nh2 = exhausted ? nh : h;
            const uint32_t nslot = (uint32_t)(nh2 * 32 + __builtin_ctz(ncur | 0x80000000u));
            // the word behind the look-ahead word and the next candidate's record: requested here, used after the arithmetic
            const int t = __builtin_ctz(nz | 0x80000000u) & (NH - 1);
            uint32_t wnext;
            asm volatile("ds_read_b32 %0, %1" : "=v"(wnext) : "v"(mk_lane + (uint32_t)t * 256u) : "memory");
            request(nxt, nslot);
            asm volatile("" : "+v"(rec.a), "+v"(rec.b), "+v"(rec.k));   // (the arithmetic below stays behind the requests)
            float col[C];
            col[0] = rec.b[2]; col[1] = rec.b[3];
            if constexpr (C == 3) col[2] = rec.k;
            else {
#pragma unroll
                for (int ch = 2; ch < C; ch++) col[ch] = rec.k[ch - 2];
            }
            const float dx = rec.a[0] - pxf, dy = rec.a[1] - pyf;
            const float power = pair_exp2_arg(rec.a[2], rec.a[3], rec.b[0], dx, dy);   // exp2 domain, see conic_to_exp2
            const float alpha = fminf(ALPHA_MAX, rec.b[1] * __builtin_amdgcn_exp2f(power));
            const bool ok = act && power <= 0.0f && alpha >= ALPHA_MIN;
            const float test_T = T * (1.0f - alpha);
            const bool stop = ok && test_T < T_EPS;
            const bool upd = ok != stop;   // stop implies ok
            const float w = upd ? alpha * T : 0.0f;
#pragma unroll
            for (int ch = 0; ch < C; ch++) Cc[ch] += col[ch] * w;
            T = upd ? test_T : T;
            last = upd ? c0 + slot + 1u : last;
            // (the trip's results are operands of the wait: the compiler otherwise sinks most of the arithmetic BEHIND it)
            if constexpr (C == 3)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wnext), "+v"(nxt.a), "+v"(nxt.b), "+v"(nxt.k), "+v"(T), "+v"(last), "+v"(Cc[0]), "+v"(Cc[1]), "+v"(Cc[2]) : : "memory");
            else if constexpr (C == 4)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wnext), "+v"(nxt.a), "+v"(nxt.b), "+v"(nxt.k), "+v"(T), "+v"(last), "+v"(Cc[0]), "+v"(Cc[1]), "+v"(Cc[2]), "+v"(Cc[3]) : : "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wnext), "+v"(nxt.a), "+v"(nxt.b), "+v"(nxt.k), "+v"(T), "+v"(last), "+v"(Cc[0]), "+v"(Cc[1]), "+v"(Cc[2]), "+v"(Cc[3]), "+v"(Cc[4]), "+v"(Cc[5]) : : "memory");
            cur = ncur;
            h = nh2;
            nw = exhausted ? (nz != 0u ? wnext : 0u) : nw;
            nh = exhausted ? t : nh;
            nz = exhausted ? (nz & (nz - 1u)) : nz;
            // a pixel that terminates drops the rest of its candidates
            stopped = stopped || stop;
            cur = stop ? 0u : cur;
            nw = stop ? 0u : nw;
            nz = stop ? 0u : nz;
            slot = nslot;
        };
        // (two trips per loop iteration on two register sets: no copies of the prefetched record; the second trip of an
        // iteration may find every lane out of candidates -- it then blends nothing)
        while (__ballot(cur != 0u) != 0ull) {
            trip(rec0, rec1);
            trip(rec1, rec0);
        }
        return stopped;
    };
#else
    const auto walk = [&](uint32_t c0, uint32_t nz) {
        const uint32_t mk_lane = (uint32_t)(size_t)&mk[wave][0][lane];   // LDS byte address of this lane's word 0 (+ 256 per word)
        uint32_t cur = 0, nw = 0;
        int h = 0, nh = 0;
        bool stopped = false;
        if (nz != 0u) { nh = __builtin_ctz(nz); nw = mk[wave][nh][lane]; nz &= nz - 1u; }
        while (true) {
            const bool need = cur == 0u;
            cur = need ? nw : cur;
            h = need ? nh : h;
            if (snaps || is_part) {   // (a part always leaves them: the combining workgroup finds its way by them)
                // first word of a new segment: the running (T, C) is the pixel's state at the segment's boundary (and at
                // every boundary it skipped) -- what the backward blend's units resume from (gsr_blend_bwd.hip)
                const uint32_t seg_new = (c0 + (uint32_t)h * 32u) / (uint32_t)SNAP_SEG;
                if (need && cur != 0u && seg_new != seg_cur) {
                    seg_cur = seg_new;
                    store_snapshot<C>(snap + ((size_t)(unit0 + seg_new * (SNAP_SEG / 64)) * 256 + pix_in_tile) * SV, T, Cc);
                }
            }
            // refill of the look-ahead slot: the word is REQUESTED here, by every lane and in every trip (an explicit
            // ds_read: left to the compiler the read sits in a block only the lanes that need a word enter, followed by
            // a wait for it -- a full LDS round trip in front of most trips), and consumed after the pair's arithmetic
            const int t = __builtin_ctz(nz | 0x80000000u) & (NH - 1);
            uint32_t wnext;
            asm volatile("ds_read_b32 %0, %1" : "=v"(wnext) : "v"(mk_lane + (uint32_t)t * 256u) : "memory");
            if (__ballot(cur != 0u) == 0ull) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wnext) : : "memory");   // (nothing may stay in flight into a dead register)
                break;
            }
            // a lane without a candidate evaluates some record of its current word and drops it
            const bool act = cur != 0u;
            const int j = __builtin_ctz(cur | 0x80000000u);
            const uint32_t slot = (uint32_t)(h * 32 + j);
            cur &= cur - 1u;
            const float4 A = ga[slot], B = gb[slot];
            // (the colour tail's index goes through an opaque copy: from `slot` itself the compiler derives slot * 4 as
            // slot * 16 - slot * 12 with a 64-bit multiply-add, a quarter-rate instruction in every trip)
            uint32_t slot_c = slot;
            asm("" : "+v"(slot_c));
            const RecTail<C> K = gc[slot_c];
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
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wnext) : : "memory");   // (long landed: the gathers above were issued after it)
            nw = need ? (nz != 0u ? wnext : 0u) : nw;
            nh = need ? t : nh;
            nz = need ? (nz & (nz - 1u)) : nz;
            // a pixel that terminates drops the rest of its candidates
            stopped = stopped || stop;
            cur = stop ? 0u : cur;
            nw = stop ? 0u : nw;
            nz = stop ? 0u : nz;
        }
        return stopped;
    };
#endif
    // where a part of this tile keeps its per-pixel result: parts are numbered (first unit / 8) + (first list entry of
    // the tile / PART_FROM) -- increasing along a tile, and strictly increasing from one split tile to the next, whose
    // list starts more than PART_FROM entries later; at most U / 8 + R / PART_FROM + 1
    const auto part_slot = [&](uint32_t c0) {
        return ((size_t)((unit0 + (c0 >> 6)) / 8u + list0 / PART_FROM) * 256 + (size_t)pix_in_tile);
    };

    // ---- combine the parts of this tile, in list order (MODE 0 has none; MODE 2: the part that drew the last ticket)
    const auto combine_parts = [&]() {
        // ---- combine the parts, in list order.  Phase A, all pixels in step: a part the pixel certainly passes through is
        // folded in; at the first one it may not, the pixel PARKS with its state at that part's start.
        const float thr = T_EPS * 1.001f;   // (0.1 % of slack for the rounding of the products: a borderline pixel parks)
        // re-base the snapshot a part left for this pixel in `unit` (relative to the part's start) on the state (T, Cc)
        const auto rebase = [&](uint32_t unit) {
            float4* const sp = snap + ((size_t)unit * 256 + pix_in_tile) * SV;
            float Ts, cs[C];
            load_snapshot<C>(sp, Ts, cs);
#pragma unroll
            for (int ch = 0; ch < C; ch++) cs[ch] = Cc[ch] + T * cs[ch];
            store_snapshot<C>(sp, T * Ts, cs);
        };
        const auto words_of = [&](uint32_t unit) { return masks[((size_t)unit * 4 + wave) * 64 + lane]; };
        bool parked = false;
        uint32_t park_c0 = 0, park_lp = 0;
        float park_Tp = 1.f;
        for (uint32_t c0 = 0; c0 < n; c0 += CH) {
            if (__syncthreads_or(!done && !parked) == 0) break;
            if (done || parked) continue;
            const size_t ps = part_slot(c0);
            float Tp, Cp[C];
            load_snapshot<C>(part_fin + ps * SV, Tp, Cp);
            const uint32_t lp = part_last[ps];
            // (the candidate words of the part's units are requested with its result: one trip to memory instead of two)
            constexpr int NU = CH / 64;
            uint2 wd[NU];
            const uint32_t u_first = unit0 + (c0 >> 6);
#pragma unroll
            for (int uu = 0; uu < NU; uu++)
                wd[uu] = (snaps && c0 + 64u * uu < n && (c0 != 0u || uu != 0)) ? words_of(u_first + uu) : make_uint2(0u, 0u);
            if ((lp & 0x80000000u) != 0u || T * Tp < thr) {   // the pixel may terminate inside this part
                parked = true; park_c0 = c0; park_lp = lp; park_Tp = Tp;
                continue;
            }
            if (snaps) {
                // re-base the part's snapshots (the pixel left one in every unit in which it has a candidate; the tile's very
                // first slot is not a boundary, see below).  All snapshots are requested together, then the stores: as eight
                // dependent word -> snapshot -> store chains this loop cost 6 us per part.
                float Ts[NU], cs[NU][C];
#pragma unroll
                for (int uu = 0; uu < NU; uu++) {
                    Ts[uu] = 0.f;
#pragma unroll
                    for (int ch = 0; ch < C; ch++) cs[uu][ch] = 0.f;
                    if ((wd[uu].x | wd[uu].y) != 0u) load_snapshot<C>(snap + ((size_t)(u_first + uu) * 256 + pix_in_tile) * SV, Ts[uu], cs[uu]);
                }
#pragma unroll
                for (int uu = 0; uu < NU; uu++) {
                    if ((wd[uu].x | wd[uu].y) != 0u) {
#pragma unroll
                        for (int ch = 0; ch < C; ch++) cs[uu][ch] = Cc[ch] + T * cs[uu][ch];
                        store_snapshot<C>(snap + ((size_t)(u_first + uu) * 256 + pix_in_tile) * SV, T * Ts[uu], cs[uu]);
                    }
                }
            }
#pragma unroll
            for (int ch = 0; ch < C; ch++) Cc[ch] += T * Cp[ch];
            T *= Tp;
            last = (lp & 0x7fffffffu) != 0u ? (lp & 0x7fffffffu) : last;
        }
        // Phase B, every parked pixel on its own (they sit in different parts; no barrier from here on): the units of its
        // part that it certainly passes through are folded in one by one -- the state a unit ends in is the snapshot of the
        // next unit with a candidate, or the part's result -- and from the first one it may not pass it walks its candidates
        // exactly, records and words gathered from memory, until it terminates (a borderline pixel that does not after all
        // simply walks on, through the rest of the list).
        if (parked) {
            const uint32_t u_part = unit0 + (park_c0 >> 6);
            const uint32_t n_units = min((uint32_t)(CH / 64), (n - park_c0 + 63u) / 64u);
            // a part that stopped has valid snapshots only up to the unit of its last contributor
            const uint32_t trust = (park_lp & 0x80000000u) ? (((park_lp & 0x7fffffffu) != 0u ? (park_lp & 0x7fffffffu) - 1u - park_c0 : 0u) >> 6)
                                                          : 0xffffffffu;
            const float T_in = T;
            float C_in[C];
#pragma unroll
            for (int ch = 0; ch < C; ch++) C_in[ch] = Cc[ch];
            uint32_t uw = n_units;   // first unit (of the part) the pixel walks exactly
            int pass_hi = -1;        // deepest unit of the part it passes through
            for (uint32_t uu = 0; uu < n_units; uu++) {
                const uint2 wd = words_of(u_part + uu);
                if ((wd.x | wd.y) == 0u) continue;
                // local state at the end of this unit
                float Te = park_Tp;
                bool have_end = (park_lp & 0x80000000u) == 0u;
                for (uint32_t u2 = uu + 1; u2 < n_units; u2++) {
                    const uint2 w2 = words_of(u_part + u2);
                    if ((w2.x | w2.y) != 0u) {
                        float ce[C];
                        load_snapshot<C>(snap + ((size_t)(u_part + u2) * 256 + pix_in_tile) * SV, Te, ce);
                        have_end = u2 <= trust;
                        break;
                    }
                }
                if (uu < trust && have_end && T_in * Te >= thr) {
                    // passes through: the unit's own snapshot becomes absolute
                    if (snaps && (park_c0 != 0u || uu != 0u)) rebase(u_part + uu);   // (T, Cc still hold the part's start state)
                    pass_hi = (int)uu;
                    continue;
                }
                uw = uu;
                break;
            }
            // the pixel blended the units it passed through: its last contributor so far is the deepest candidate of those
            // units that passes the alpha test (no running state needed for that; the words are tight, so the first look
            // almost always hits)
            for (int uu = pass_hi; uu >= 0; uu--) {
                const uint2 wd = words_of(u_part + (uint32_t)uu);
                bool found = false;
                for (int half = 1; half >= 0 && !found; half--) {
                    uint32_t bits = half ? wd.y : wd.x;
                    while (bits != 0u) {
                        const int bpos = 31 - __builtin_clz(bits);
                        bits &= ~(1u << bpos);
                        const uint32_t k = park_c0 + 64u * (uint32_t)uu + 32u * (uint32_t)half + (uint32_t)bpos;
                        const float4 A = rec_a[list0 + k], B = rec_b[list0 + k];
                        const float dx = A.x - pxf, dy = A.y - pyf;
                        const float power = pair_exp2_arg(A.z, A.w, B.x, dx, dy);
                        const float alpha = fminf(ALPHA_MAX, B.y * __builtin_amdgcn_exp2f(power));
                        if (power <= 0.0f && alpha >= ALPHA_MIN) { last = k + 1u; found = true; break; }
                    }
                }
                if (found) break;
            }
            // state at the start of unit uw: the part's start state carried through the unit's own (relative) snapshot
            if (uw < n_units) {
                if (park_c0 != 0u || uw != 0u) {
                    float Ts, cs[C];
                    load_snapshot<C>(snap + ((size_t)(u_part + uw) * 256 + pix_in_tile) * SV, Ts, cs);
#pragma unroll
                    for (int ch = 0; ch < C; ch++) Cc[ch] = C_in[ch] + T_in * cs[ch];
                    T = T_in * Ts;
                }
                // (the tile's first unit has no snapshot: the pixel's state there is the start state)
            }
            // exact walk from unit uw of the part to termination
            const uint32_t u_end = unit0 + (n + 63u) / 64u;
            for (uint32_t unit = u_part + uw; unit < u_end && !done; unit++) {
                const uint2 wd = words_of(unit);
                if ((wd.x | wd.y) == 0u) continue;
                if (snaps && unit != unit0) store_snapshot<C>(snap + ((size_t)unit * 256 + pix_in_tile) * SV, T, Cc);
                const uint32_t k0 = (unit - unit0) * 64u;
                for (int half = 0; half < 2 && !done; half++) {
                    uint32_t bits = half ? wd.y : wd.x;
                    while (bits != 0u) {
                        const uint32_t k = k0 + 32u * half + (uint32_t)__builtin_ctz(bits);
                        bits &= bits - 1u;
                        const float4 A = rec_a[list0 + k], B = rec_b[list0 + k];
                        const RecTail<C> K = rec_c[list0 + k];
                        float col[C];
                        col[0] = B.z; col[1] = B.w;
#pragma unroll
                        for (int ch = 2; ch < C; ch++) col[ch] = K.c[ch - 2];
                        const float dx = A.x - pxf, dy = A.y - pyf;
                        const float power = pair_exp2_arg(A.z, A.w, B.x, dx, dy);
                        const float alpha = fminf(ALPHA_MAX, B.y * __builtin_amdgcn_exp2f(power));
                        if (power > 0.0f || alpha < ALPHA_MIN) continue;
                        const float test_T = T * (1.0f - alpha);
                        if (test_T < T_EPS) { done = true; break; }
                        const float w = alpha * T;
#pragma unroll
                        for (int ch = 0; ch < C; ch++) Cc[ch] += col[ch] * w;
                        T = test_T;
                        last = k + 1u;
                    }
                }
            }
        }
    };
    bool combine = false;
    if (is_part) {
        // ---- one part: chunk [part_c0, part_c0 + CH) from T = 1, C = 0
        fetch_ids(part_c0);
        fetch_records();
        park(part_c0, false, true, true);   // (words and records always: the combining workgroup may need them)
        __syncthreads();
        const bool stopped = walk(part_c0, done ? 0u : nonempty_words(part_c0));
        const size_t ps = part_slot(part_c0);
        store_snapshot<C>(part_fin + ps * SV, T, Cc);
        part_last[ps] = last | (stopped ? 0x80000000u : 0u);
        if constexpr (MODE != 2) return;
        // Ticket: everything this part wrote (result, snapshots, words, records) is released to the device before the ticket
        // is drawn; whoever draws the last one acquires and sees all parts.  Parts of a tile are consecutive in the part list.
        __shared__ uint32_t ticket;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0)
            ticket = __hip_atomic_fetch_add(&part_ticket[blockIdx.x - part_c0 / (uint32_t)CH], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (ticket + 1u != (n + (uint32_t)CH - 1u) / (uint32_t)CH) return;
        __threadfence();
        T = 1.0f;
#pragma unroll
        for (int ch = 0; ch < C; ch++) Cc[ch] = 0.f;
        last = 0;
        done = !inside;
        combine = true;
    } else {
        combine = MODE == 0 && long_list;
    }
    if (combine) {
        combine_parts();
    } else {
        fetch_ids(0);
        fetch_records();
        fetch_ids(CH);
        ids_lds = nullptr;   // (the first park below overwrites those bytes)
        for (uint32_t c0 = 0; c0 < n; c0 += CH) {
            // every pixel of the tile saturated: stop (also the barrier that frees the LDS of the previous chunk)
            if (__syncthreads_or(!done) == 0) break;
            park(c0, false, snaps, snaps);
            fetch_records();
            fetch_ids(c0 + 2 * CH);
            __syncthreads();
            FWD_PHASE();
            // (a pixel that is done consumes no words)
            const bool stopped = walk(c0, done ? 0u : nonempty_words(c0));
            done = done || stopped;
            FWD_PHASE();
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
        if (snaps && n > (uint32_t)SNAP_SEG) store_snapshot<C>(snap + ((size_t)unit0 * 256 + pix_in_tile) * SV, T, Cc);
    }
    if (trace && lane == 0 && !is_part) {   // last wave to finish wins the end stamp
        if (wave == 0) trace[2 * tile_block] = t_start;
        atomicMax((unsigned long long*)&trace[2 * tile_block + 1], (unsigned long long)wall_clock64());
    }
}

template <int C>
static void launch_fwd_c(int W, int H, int R, int U, uint32_t max_count, const float* bg, const float* feats, GeomState g, ImageState im,
                         BinState b, float* out_color, bool keep_masks, void* zero_ptr, size_t zero_bytes, uint32_t* counters,
                         bool sort_small, const PlanJob* job, int num_parts, hipStream_t st)
{
    const Tiles t = tiles_of(W, H);
    // (lists above 2 048 entries were sorted by the big-sort kernels before: launch_tile_sort; scatter_kernel listed the parts)
    const uint32_t split_n = split_threshold(max_count, (uint32_t)(R > 0 ? R : 0));
    // Residency knob: extra dynamic LDS lowers the number of co-resident tiles per CU (tuning only).
    static const int pad = getenv("GSR_FWD_LDS_PAD") ? atoi(getenv("GSR_FWD_LDS_PAD")) : 0;
    // A part that has stored its result releases it to the device before it draws its ticket -- a write-back of its XCD's L2 --,
    // which is nothing for a view with a handful of parts (config B: 10) and a storm for one with a thousand (config D, lists of
    // up to 11 800 entries, every list above 1 024 split: blend_fwd 524 us in one launch against 199 us with the parts as a
    // launch of their own, whose end is the release).  So: in one launch up to PARTS_IN_LAUNCH parts, two launches above.
    // (num_parts < 0: the caller did not run stage 1 right before -- decide by the longest list.)
    constexpr int PARTS_IN_LAUNCH = 64;
    static const char* e_two = getenv("GSR_FWD_TWO_LAUNCHES");   // (comparison builds: 1 = always two launches, 0 = never)
    const bool two_launches = e_two ? e_two[0] != '0' : (num_parts >= 0 ? num_parts > PARTS_IN_LAUNCH : max_count > 4096u);
    const auto go = [&](auto mode, unsigned grid, unsigned part_grid) {
        constexpr int M = decltype(mode)::value;
        blend_fwd_kernel<C, FWD_CHUNK, M><<<grid, 256, M == 1 ? 0 : pad, st>>>(
            W, H, t.gx, im.ranges, im.order, b.point_list, sort_small ? b.keys : nullptr, g.g0, g.g1, feats, bg, out_color,
            im.final_T, im.n_contrib, im.seg_off, b.masks, b.snap, b.rec_a, b.rec_b, static_cast<RecTail<C>*>(b.rec_c), b.part_list,
            im.totals, b.part_fin, b.part_last, b.part_ticket, part_grid, max_count <= SORT_SMALL_CAP, split_n, keep_masks,
            M != 1 ? static_cast<float4*>(zero_ptr) : nullptr, M != 1 ? (uint32_t)(zero_bytes / 16) : 0u, M != 1 ? counters : nullptr,
            (uint32_t)shard_stride(t.T), M != 1 ? g_trace : nullptr, PlanRun{}, M == 0 && job ? *job : PlanJob{});
    };
    // (scatter_kernel listed the parts; lists above 2 048 entries were sorted by launch_tile_sort, shorter ones sort themselves)
    if (split_n != 0xffffffffu && U > 0 && !two_launches) {
        const unsigned np = (unsigned)part_capacity(R, U);
        go(std::integral_constant<int, 2>{}, np + (unsigned)t.T, np);
    } else {
        if (split_n != 0xffffffffu && U > 0) go(std::integral_constant<int, 1>{}, (unsigned)part_capacity(R, U), 0u);
        if (split_n != 0xffffffffu && U > 0) job = nullptr;   // (the parts ran as a launch of their own: no plan for such a view)
        go(std::integral_constant<int, 0>{}, (unsigned)t.T + (job && job->enabled ? 1u : 0u), 0u);
    }
}

template <int C>
static void launch_fwd_planned_c(int W, int H, const float* bg, const float* feats, GeomState g, ImageState im, BinState b,
                                 float* out_color, bool keep_masks, void* zero_ptr, size_t zero_bytes, PlanRun plan, hipStream_t st,
                                 const PlanJob* job)
{
    const Tiles t = tiles_of(W, H);
    static const int pad = getenv("GSR_FWD_LDS_PAD") ? atoi(getenv("GSR_FWD_LDS_PAD")) : 0;
    const bool rides = job != nullptr && job->enabled != 0u;
    blend_fwd_kernel<C, FWD_CHUNK, 0, true><<<(unsigned)t.T + (rides ? 1u : 0u), 256, pad, st>>>(
        W, H, t.gx, plan.ranges, plan.order, b.point_list, b.keys, g.g0, g.g1, feats, bg, out_color, im.final_T, im.n_contrib,
        plan.seg_off, b.masks, b.snap, b.rec_a, b.rec_b, static_cast<RecTail<C>*>(b.rec_c), b.part_list, im.totals, b.part_fin,
        b.part_last, b.part_ticket, 0u, false, 0xffffffffu, keep_masks, static_cast<float4*>(zero_ptr), (uint32_t)(zero_bytes / 16),
        nullptr, 0u, g_trace, plan, rides ? *job : PlanJob{});
}

void launch_blend_fwd_planned(int C, int W, int H, const float* bg, const float* feats, GeomState g, ImageState im, BinState b,
                              float* out_color, bool keep_masks, void* zero_ptr, size_t zero_bytes, PlanRun plan, hipStream_t st,
                              const PlanJob* job)
{
    if (C == 6) launch_fwd_planned_c<6>(W, H, bg, feats, g, im, b, out_color, keep_masks, zero_ptr, zero_bytes, plan, st, job);
    else if (C == 4) launch_fwd_planned_c<4>(W, H, bg, feats, g, im, b, out_color, keep_masks, zero_ptr, zero_bytes, plan, st, job);
    else launch_fwd_planned_c<3>(W, H, bg, feats, g, im, b, out_color, keep_masks, zero_ptr, zero_bytes, plan, st, job);
}

// job (may be null): the plan of the camera's next view, built by one extra workgroup of a MODE 0 launch -- *job_rides says
// whether this view's launch carried it (a view that splits its long lists does not: such a view is not plannable anyway)
void launch_blend_fwd(int C, int W, int H, int R, int U, uint32_t max_count, const float* bg, const float* feats, GeomState g, ImageState im,
                      BinState b, float* out_color, bool keep_masks, void* zero_ptr, size_t zero_bytes, uint32_t* counters,
                      bool sort_small, hipStream_t st, const PlanJob* job, bool* job_rides, int num_parts)
{
    const bool splits = split_threshold(max_count, (uint32_t)(R > 0 ? R : 0)) != 0xffffffffu && U > 0;
    if (splits || !job || !job->enabled) job = nullptr;
    if (job_rides) *job_rides = job != nullptr;
    if (C == 6) launch_fwd_c<6>(W, H, R, U, max_count, bg, feats, g, im, b, out_color, keep_masks, zero_ptr, zero_bytes, counters, sort_small, job, num_parts, st);
    else if (C == 4) launch_fwd_c<4>(W, H, R, U, max_count, bg, feats, g, im, b, out_color, keep_masks, zero_ptr, zero_bytes, counters, sort_small, job, num_parts, st);
    else launch_fwd_c<3>(W, H, R, U, max_count, bg, feats, g, im, b, out_color, keep_masks, zero_ptr, zero_bytes, counters, sort_small, job, num_parts, st);
}

}  // namespace gsr
