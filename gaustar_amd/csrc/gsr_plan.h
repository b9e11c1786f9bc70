// gsr_plan.h -- the PLAN of a camera's next view (gsr_internal.h "planned binning"), built by ONE workgroup from the exact
// ranges and launch order of a view rendered the exact way: plan_build_block.  It rides in the exact path's forward blend as
// one extra workgroup of that launch (gsr_blend_fwd.hip, dispatched first): no launch of its own, no second stream, nothing on
// the view's critical path -- a camera's first view costs what it cost before plans existed.  Round 6: a PLANNED view carries the
// same workgroup (counts = the cursors its preprocess claimed on, which its forward blend leaves standing; launch order = the
// plan's own), writing the OTHER half of the caller's plan buffer: a camera's plan is always one visit old, however the Gaussians
// move between its visits.  Also home of the wave64 DPP scans
// the tile-offset scan (gsr_binning.hip) shares with it.
#pragma once
#include "gsr_internal.h"
#include <type_traits>
#include <utility>

namespace gsr {

// ---- wave64 inclusive scan (sum) on the DPP network: three row shifts of the input, two masked row shifts, two row
// broadcasts -- seven fused adds.  (As a ladder of six __shfl_up steps it was six dependent ds_bpermute round trips per
// scan; the kernel is ONE workgroup, so its time is the sum of such chains: tile_scan's "two wave scans + barrier" phase
// took 2.8 us of the kernel's 11.4, phase stamps of a GSR_SCAN_TRACE build.)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x, int /*lane*/)
{
    const auto dpp = [](uint32_t v, auto ctrl, auto row_mask, auto bank_mask) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl)::value, decltype(row_mask)::value,
                                                     decltype(bank_mask)::value, true);   // lanes without a source add 0
    };
    using std::integral_constant;
    uint32_t v = x + dpp(x, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{});   // row_shr:1
    v += dpp(x, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{});               // row_shr:2
    v += dpp(x, integral_constant<int, 0x113>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{});               // row_shr:3
    v += dpp(v, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xe>{});               // row_shr:4
    v += dpp(v, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xc>{});               // row_shr:8
    v += dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}, integral_constant<int, 0xf>{});               // row_bcast:15
    v += dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{}, integral_constant<int, 0xf>{});               // row_bcast:31
    return v;
}
// max over the wave, complete in lane 63 (same ladder; counts are unsigned, so a missing source contributes 0)
__device__ __forceinline__ uint32_t wave_max_to_lane63(uint32_t x)
{
    const auto dpp = [](uint32_t v, auto ctrl, auto row_mask, auto bank_mask) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl)::value, decltype(row_mask)::value,
                                                     decltype(bank_mask)::value, true);
    };
    using std::integral_constant;
    uint32_t v = max(x, dpp(x, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{}));
    v = max(v, dpp(x, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{}));
    v = max(v, dpp(x, integral_constant<int, 0x113>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{}));
    v = max(v, dpp(v, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xe>{}));
    v = max(v, dpp(v, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xc>{}));
    v = max(v, dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}, integral_constant<int, 0xf>{}));
    v = max(v, dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{}, integral_constant<int, 0xf>{}));
    return v;
}

// What the plan-building workgroup needs (by value in the forward blend's arguments); enabled == 0: no job in this launch.
struct PlanJob {
    uint32_t enabled;
    int T, gx, gy;
    uint32_t* header;        // PlanState of the caller's plan buffer
    uint2* ranges;
    uint32_t* seg_off;
    uint32_t* order;
    uint32_t split_from_word, level;
    uint32_t* host_pad;      // pinned, device-mapped: header words [0..7], then host_seq at [8]
    uint32_t host_seq;
    const uint32_t* cursor;  // (a PLANNED view re-plans, round 6) the view's tile counts = its cursors, PLAN_CURSOR_STRIDE words apart,
                             // left standing by the forward blend; nullptr: an exact view, counts from its ranges
    const uint2* prev_ranges;   // (GSR_PLAN_DIAG builds) the plan this one replaces, nullptr: none
};

// Capacity of a tile's bucket: its count plus an eighth (at least 16 entries; times 2^level), rounded UP to whole units of 64 --
// slack up to the unit boundary costs nothing but address space, a further unit costs the backward an empty work item -- at
// most the 2 048 entries the forward blend sorts itself.  A tile that was empty gets one unit if one of its eight neighbours
// was not (the surface's silhouette moves by a tile now and then), none otherwise.
// `level` (0 .. 3) doubles the slack per step: raised for a camera whose views outgrow their plans (the Gaussians move between
// its visits), see gsr_forward_planned.
// Round 6: the slack also looks at the tile's eight NEIGHBOURS.  What outgrows a bucket when the Gaussians move is not the
// interior of the surface (counts change by per cent) but the handful of tiles on its silhouette: a tile that held ten entries
// next to one that holds five hundred holds three hundred once the silhouette has moved a few pixels its way (GSR_PLAN_DIAG:
// 2-23 tiles per misfit, count / capacity up to 8).  A shift by d pixels mixes a tile's count with its neighbour's in the ratio
// d / 16, so the slack gets an eighth (times 2^level) of what the LARGEST neighbour has more -- nothing for an interior tile,
// a few units for a silhouette tile, and only there.
#ifndef GSR_PLAN_NB_SLACK
#define GSR_PLAN_NB_SLACK 1
#endif
__device__ __forceinline__ uint32_t plan_capacity(uint32_t n, bool near_occupied, uint32_t level, uint32_t nb_max = 0u)
{
    const uint32_t more = GSR_PLAN_NB_SLACK && nb_max > n ? (nb_max - n) >> 3 : 0u;
    if (n == 0u) {
        if (!near_occupied) return 0u;
        return min((64u + (more << level) + 63u) & ~63u, PLAN_MAX_LIST);
    }
    const uint32_t want = n + ((max(16u, n >> 3) + more) << level);
    return min((want + 63u) & ~63u, PLAN_MAX_LIST);
}

// One workgroup of NT threads (a multiple of 64, at most 1 024).  `im_ranges` / `order_in`: the view's exact ranges and launch
// order (written by tile_scan_kernel before this launch); a planned view: job.cursor and the current plan's order.  cnt_lds: PLAN_LDS_T words of LDS (the counts are staged there for
// images of up to 8 192 tiles -- 1920 x 1088 --: as global loads the neighbour look-ups of the empty tiles made this single
// workgroup a chain of dependent trips to memory).
// valid = every list of the source view leaves that slack below 2 048 and the view would not split its long lists
// (split_threshold_from on the CAPACITIES, with three quarters of the source view's R: a planned view is never split, so that its
// images are the exact path's bit for bit).
constexpr int PLAN_LDS_T = 8192;
template <int NT>
__device__ __forceinline__ void plan_build_block(const PlanJob& job, const uint2* __restrict__ im_ranges,
                                                 const uint32_t* __restrict__ order_in, uint32_t* __restrict__ cnt_lds)
{
    static_assert(NT % 64 == 0 && NT <= 1024, "whole waves");
    constexpr int NW = NT / 64;
    __shared__ uint32_t ws_cap[NW], ws_max[NW], ws_ne[NW], ws_n[NW];
    const int T = job.T, gx = job.gx, gy = job.gy;
    const uint32_t level = job.level;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // This workgroup shares its CU with tiles of the blend: what it costs the launch is the length of its own chain, so every
    // access to memory is lane-consecutive (a wave owns ROWS of 64 consecutive tiles; the first version gave a thread 32
    // consecutive tiles -- 64 lines per load / store instruction -- and took 105 us inside a blend of 89) and its waves
    // run at the highest priority.
    __builtin_amdgcn_s_setprio(3);
    const bool staged = T <= PLAN_LDS_T;
    uint32_t vmax = 0, ne = 0, sum_n = 0;
    const uint32_t* const cursor = job.cursor;
    const auto count_in = [&](int t) -> uint32_t {
        if (cursor != nullptr) return cursor[(size_t)t * PLAN_CURSOR_STRIDE];
        const uint2 r = im_ranges[t];
        return r.y - r.x;
    };
    // (eight counts per thread in flight at a time: a planned view's counts are its cursors, one per 128-byte line -- as one
    // load per loop trip the 32 trips of a 1080p view were 32 dependent trips to memory, most of the blend's span)
    constexpr int BATCH = 8;
    for (int t0 = tid; t0 < T; t0 += NT * BATCH) {
        uint32_t c[BATCH];
#pragma unroll
        for (int q = 0; q < BATCH; q++) {
            const int t = t0 + q * NT;
            c[q] = t < T ? count_in(t) : 0u;
        }
#pragma unroll
        for (int q = 0; q < BATCH; q++) {
            const int t = t0 + q * NT;
            if (t >= T) break;
            const uint32_t n = c[q];
            if (staged) cnt_lds[t] = min(n, 0xffffu);       // low half: the count (clipped: above 2 048 the plan is invalid anyway)
            sum_n += n;
            vmax = max(vmax, n);
            ne += n != 0u ? 1u : 0u;
        }
    }
    if (staged) __syncthreads();
    const auto count_of = [&](int t) -> uint32_t {
        if (staged) return cnt_lds[t] & 0xffffu;
        return count_in(t);
    };
    const auto cap_of = [&](int t) -> uint32_t {
        const uint32_t n = count_of(t);
        uint32_t nb_max = 0u;
        if (n == 0u || (GSR_PLAN_NB_SLACK && staged)) {   // (unstaged -- above 8 192 tiles --: empty tiles only, as global look-ups)
            const int ty = t / gx, tx = t - ty * gx;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    const int x = tx + dx, y = ty + dy;
                    if (x >= 0 && x < gx && y >= 0 && y < gy && (dx | dy) != 0) nb_max = max(nb_max, count_of(y * gx + x));
                }
        }
        return plan_capacity(n, nb_max != 0u, level, nb_max);
    };
    const int rows = (T + 63) >> 6, rpw = (rows + NW - 1) / NW;      // rows of 64 tiles; rows per wave
    const int r0 = wave * rpw, r1 = min(rows, r0 + rpw);
    // pass 1: this wave's capacities (kept in the upper half of the staged word: a neighbour's look-up reads the lower half,
    // which the store does not change) and their sum
    uint32_t sum_cap = 0;
    for (int r = r0; r < r1; r++) {
        const int t = 64 * r + lane;
        if (t < T) {
            const uint32_t c = cap_of(t);
            if (staged) cnt_lds[t] = (cnt_lds[t] & 0xffffu) | (c << 16);
            sum_cap += c;
        }
    }
    const uint32_t incl_cap = wave_incl_scan(sum_cap, lane);
    const uint32_t incl_n = wave_incl_scan(sum_n, lane), incl_ne = wave_incl_scan(ne, lane);
    vmax = wave_max_to_lane63(vmax);
    if (lane == 63) { ws_cap[wave] = incl_cap; ws_max[wave] = vmax; ws_ne[wave] = incl_ne; ws_n[wave] = incl_n; }
    __syncthreads();
    uint32_t woff = 0, total_cap = 0, gmax = 0, total_ne = 0, total_n = 0;
    for (int w = 0; w < NW; w++) {
        if (w < wave) woff += ws_cap[w];
        total_cap += ws_cap[w];
        gmax = max(gmax, ws_max[w]);
        total_ne += ws_ne[w];
        total_n += ws_n[w];
    }
    // pass 2: one wave scan per row, the running base carried in a scalar
    uint32_t run = woff;
    for (int r = r0; r < r1; r++) {
        const int t = 64 * r + lane;
        const bool in = t < T;
        const uint32_t cap = !in ? 0u : staged ? cnt_lds[t] >> 16 : cap_of(t);
        const uint32_t incl = wave_incl_scan(cap, lane);
#ifdef GSR_PLAN_DIAG   // (devtool build) how this view's counts sit in the plan being replaced: host pad words 10 .. 13 =
                       // tiles over a zero bucket, tiles over a non-zero bucket, entries over, the worst count / capacity in 1/64
        if (in && job.host_pad && job.prev_ranges) {
            const uint32_t old = job.prev_ranges[t].y, n = count_of(t);
            if (n > old) {
                atomicAdd(&job.host_pad[old == 0u ? 10 : 11], 1u);
                atomicAdd(&job.host_pad[12], n - old);
                if (old) atomicMax(&job.host_pad[13], n * 64u / old);
            }
        }
#endif
        if (in) {
            const uint32_t first = run + incl - cap;
            job.ranges[t] = make_uint2(first, cap);
            job.seg_off[t] = first >> 6;     // (capacities are whole units: the unit prefix is the entry prefix / 64)
            job.order[t] = order_in[t];
        }
        run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (tid == 0) {
        const uint32_t max_cap = plan_capacity(gmax, false, level);
        const bool fits = gmax + 16u <= PLAN_MAX_LIST && total_n < 0x7fffffffu - 64u * (uint32_t)T;
        const bool splits = split_threshold_from(max_cap, total_n - total_n / 4u, job.split_from_word) != 0xffffffffu;
        const uint4 h0 = make_uint4(fits && !splits ? 1u : 0u, (uint32_t)T, total_cap, total_cap >> 6);
        const uint4 h1 = make_uint4(max_cap, total_ne, total_n, gmax);
        job.seg_off[T] = total_cap >> 6;
        reinterpret_cast<uint4*>(job.header)[0] = h0;
        reinterpret_cast<uint4*>(job.header)[1] = h1;
        if (job.host_pad) {
            reinterpret_cast<uint4*>(job.host_pad)[0] = h0;
            reinterpret_cast<uint4*>(job.host_pad)[1] = h1;
            __hip_atomic_store(&job.host_pad[8], job.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace gsr
