// gsr_blend_fwd.hip -- forward alpha compositing.
//
// Same per-pixel arithmetic and control flow as the reference's renderCUDA
// (DGR/cuda_rasterizer/forward.cu:261-374), re-organised for CDNA4:
//
//  * One wave64 owns an 8x8 pixel block; a 16x16 tile is four independent waves (no workgroup
//    barriers, each wave stops as soon as its own 64 pixels are saturated).
//  * Inside the wave each DPP ROW (16 lanes) owns a 4x4 pixel QUADRANT and consumes its OWN queue of
//    splats: in one loop iteration the four rows blend four different Gaussians.  Surface splats
//    (~4 px radius) cover a fraction of an 8x8 block, so feeding all 64 lanes the same splat leaves most
//    lanes idle; per-quadrant queues keep them busy and shorten every wave's serial chain.
//  * Each wave walks the tile's depth-sorted list 64 instances at a time: lane i fetches instance i
//    (coalesced id load two batches ahead, two 16-byte record gathers + colour one batch ahead), parks
//    it in LDS, and tests it exactly (block_min_half_quad) against each of the four quadrants that still
//    has an unsaturated pixel; a ballot + prefix-popcount per quadrant appends the lane's slot (its LDS byte address,
//    16 bits) to that quadrant's queue -- depth order is preserved per quadrant, which is all a pixel needs.
//  * The tile's list is depth-sorted by the same workgroup right before the walk (lists of up to 2 048 entries, gsr_sort.h);
//    two side jobs of the fused forward ride along (the backward's accumulation table and the tile's counters are cleared).
//  * The blend loop is four queue slots deep and branch-free: the four exponents/alphas are independent
//    and evaluated together, only the short T-update chain is serial.  Position, conic, opacity AND
//    colour come from LDS (the reference gathers colour from global memory per pixel, forward.cu:355).
//
// n_contrib stores the 1-based list position of the last blended instance, as the reference does.
// At every SEG-th list position the running (T, C) of the pixels still alive is snapshotted for the
// segment-parallel backward pass (gsr_blend_bwd.hip) -- unless the caller announced a forward-only render
// (num_segments = 0 at stage 2: no snapshot area, no stores).
//
// LONG TILES.  A tile's walk is serial per pixel, so the kernel's span used to be its longest tile (a 1 600-entry tile
// takes 95 us while the whole image needs 68 us of machine time; close-up views with 10 000-entry tiles were entirely
// critical-path-bound).  For tiles above GSR_FWD_LONG entries (default 4 096) blend_fwd_partial_kernel first reduces
// every 64-entry segment INDEPENDENTLY (one wave per segment and 8x8 block, like the backward's units) to the
// per-pixel pair  P_s = prod (1 - alpha),  C_s = sum c alpha T_local  (T_local starts at 1) and the last contributing
// position; the main kernel then steps through a long tile's segments in O(1) each:  C += T * C_s,  T *= P_s.
// Termination stays exact: T can only fall below 1e-4 inside segment s if T * P_s < 1e-4, and then the pixel walks that
// one segment itself (see blend_fwd_long_kernel).  Pixels of the skipped segments see the same products in a different
// association (T * (a * b) instead of (T * a) * b): ulp-level, far inside the 1e-4 parity bound.
// Measured (MI355X): config D (1 M Gaussians, lists up to 11 787 entries) forward 0.56 -> 0.32 ms with the default
// threshold of 4 096.  Lower thresholds do not pay: config C's forward is throughput-bound, not bound by its longest tiles
// (skipping every tile above 1 024 entries outright leaves the main kernel at 0.095 of 0.097 ms), so the pre-reduction is
// pure extra work there (0.44 vs 0.39 ms per view at 1 024); at 2 048 config B's two 2 116-entry tiles cost more than they save.
//
// The kernel is a template over the number of colour channels C: 3 is the reference's NUM_CHANNELS
// (cuda_rasterizer/config.h:15); 6 renders TWO targets that share geometry (GauSTAR's RGB + depth-as-colour
// passes, refine.py:552 and :607) in one walk -- alpha, T, termination and n_contrib do not depend on colour,
// so channels 0-2 / 3-5 are bit-identical to two separate 3-channel renders.
#include "gsr_internal.h"
#include "gsr_sort.h"
#include <cstdlib>

namespace gsr {

template <int C>
struct __attribute__((aligned(16))) Slot {   // 48 B (C = 3) / 64 B (C = 6) per fetched instance
    float4 a;                        // x, y, conic a, b (exp2 domain, see conic_to_exp2)
    float4 b;                        // conic c (exp2 domain), opacity, list position + 1 (as uint bits), -
    float col[(C + 3) / 4 * 4];      // colour channels
};

template <int C>
struct Fetched { float4 a, b; float col[C]; };

// LDS byte address -> pointer into the LDS address space
template <int C>
__device__ __forceinline__ const Slot<C>* lds_slot(uint32_t addr)
{
    return (const Slot<C>*)(const __attribute__((address_space(3))) Slot<C>*)(uintptr_t)addr;
}

// Two-stage software pipeline over the dependent gather (list -> id -> records): ids are fetched two batches
// ahead, records one batch ahead, so neither load latency sits on the per-batch critical path.
__device__ __forceinline__ uint32_t fetch_id(uint32_t k, uint32_t n, const uint32_t* __restrict__ list)
{
    return k < n ? list[k] : 0xffffffffu;
}
template <int C>
__device__ __forceinline__ Fetched<C> fetch_record(uint32_t gid, const float4* __restrict__ g0,
                                                   const float4* __restrict__ g1, const float* __restrict__ feats)
{
    Fetched<C> f;
    f.a = make_float4(0.f, 0.f, 1.f, 0.f);
    f.b = make_float4(1.f, 0.f, -1.f, 0.f);   // tau = -1: never kept
#pragma unroll
    for (int ch = 0; ch < C; ch++) f.col[ch] = 0.f;
    if (gid != 0xffffffffu) {
        f.a = g0[gid];
        f.b = g1[gid];
        if constexpr (C % 2 == 0) {   // rows of an even channel count are 8-byte aligned
            const float2* pf = reinterpret_cast<const float2*>(feats + (size_t)C * gid);
#pragma unroll
            for (int ch = 0; ch < C; ch += 2) { const float2 v = pf[ch / 2]; f.col[ch] = v.x; f.col[ch + 1] = v.y; }
        } else {
#pragma unroll
            for (int ch = 0; ch < C; ch++) f.col[ch] = feats[(size_t)C * gid + ch];
        }
    }
    return f;
}

constexpr int QCAP = 64 + 4;   // queue capacity per quadrant (+4: the 4-deep loop reads whole words)

template <int C> struct PixState { float T; float Cc[C]; uint32_t last; bool done; };

// One 64-entry batch for one wave (8x8 pixels, DPP row = 4x4 quadrant): park the fetched instances in LDS, build the
// four quadrant queues, blend.  TERM = false drops the termination logic (segment pre-reduction).
template <int C, bool TERM>
__device__ __forceinline__ void walk_batch(Slot<C>* __restrict__ ent, uint8_t (*qi)[QCAP], const Fetched<C>& cur, uint32_t base,
                                           unsigned long long alive, int lane, int row, int sx, int sy, float pxf, float pyf,
                                           PixState<C>& ps)
{
    constexpr int CV = (C + 3) / 4;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t k = base + lane;
    {
        float a2 = cur.a.z, b2 = cur.a.w, c2 = cur.b.x;
        conic_to_exp2(a2, b2, c2);
        ent[lane].a = make_float4(cur.a.x, cur.a.y, a2, b2);
        ent[lane].b = make_float4(c2, cur.b.y, __uint_as_float(k + 1), 0.f);
    }
#pragma unroll
    for (int v = 0; v < CV; v++)
        reinterpret_cast<float4*>(ent[lane].col)[v] =
            make_float4(cur.col[4 * v], 4 * v + 1 < C ? cur.col[4 * v + 1] : 0.f, 4 * v + 2 < C ? cur.col[4 * v + 2] : 0.f,
                        4 * v + 3 < C ? cur.col[4 * v + 3] : 0.f);
    // exact reachability test against each quadrant that still has an unsaturated pixel
    int cnt[4];
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
        const float X0 = (float)(sx + (qd & 1) * 4) - cur.a.x, Y0 = (float)(sy + (qd >> 1) * 4) - cur.a.y;
        const bool keep = ((alive >> (16 * qd)) & 0xffffull) != 0ull &&
                          block_min_half_quad(cur.a.z, cur.a.w, cur.b.x, X0, X0 + 3.f, Y0, Y0 + 3.f) <= cur.b.z;
        const unsigned long long m = __ballot(keep);
        cnt[qd] = __popcll(m);
        if (keep) qi[qd][__popcll(m & lt)] = (uint8_t)lane;
        if (lane < 4) qi[qd][cnt[qd] + lane] = 64;   // pad to a multiple of 4 with the neutral instance
    }
    const int my_cnt = row == 0 ? cnt[0] : row == 1 ? cnt[1] : row == 2 ? cnt[2] : cnt[3];
    const int max_cnt = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3]));
    __builtin_amdgcn_wave_barrier();
    const uint8_t* myq = qi[row];
    for (int j = 0; j < max_cnt; j += 4) {
        // four queue positions at once; rows past their own queue end read the neutral instance
        const uint32_t packed = j < my_cnt ? *reinterpret_cast<const uint32_t*>(myq + j) : 0x40404040u;
        float4 A[4], B[4], K[4][CV];
        float alpha[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int e = (packed >> (8 * u)) & 0xff;
            A[u] = ent[e].a; B[u] = ent[e].b;
#pragma unroll
            for (int v = 0; v < CV; v++) K[u][v] = reinterpret_cast<const float4*>(ent[e].col)[v];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float dx = A[u].x - pxf, dy = A[u].y - pyf;
            const float power = pair_exp2_arg(A[u].z, A[u].w, B[u].x, dx, dy);   // exp2 domain, see conic_to_exp2
            alpha[u] = fminf(ALPHA_MAX, B[u].y * __builtin_amdgcn_exp2f(power));
            ok[u] = power <= 0.0f && alpha[u] >= ALPHA_MIN;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float test_T = ps.T * (1.0f - alpha[u]);
            const bool live = ok[u] && !ps.done;
            const bool stop = TERM && live && test_T < T_EPS;
            const bool upd = live && !stop;
            ps.done = ps.done || stop;
            const float w = upd ? alpha[u] * ps.T : 0.0f;
#pragma unroll
            for (int ch = 0; ch < C; ch++) {
                const float4 kv = K[u][ch / 4];
                ps.Cc[ch] += (ch % 4 == 0 ? kv.x : ch % 4 == 1 ? kv.y : ch % 4 == 2 ? kv.z : kv.w) * w;
            }
            ps.T = upd ? test_T : ps.T;
            ps.last = upd ? __float_as_uint(B[u].z) : ps.last;
        }
    }
    __builtin_amdgcn_wave_barrier();
}

template <int C>
__device__ __forceinline__ void init_neutral(Slot<C>* ent, int lane)
{
    if (lane == 0) {   // neutral instance: opacity 0 never passes the alpha test
        ent[64].a = make_float4(0.f, 0.f, 0.f, 0.f);
        ent[64].b = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ch = 0; ch < (C + 3) / 4 * 4; ch++) ent[64].col[ch] = 0.f;
    }
}

static_assert(SEG == 64, "a pre-reduced segment is one 64-entry batch");

// Segment pre-reduction for long tiles: unit = (tile, 64-entry segment), one wave per unit and 8x8 block.
template <int C>
__global__ void __launch_bounds__(64)
blend_fwd_partial_kernel(int W, int H, int gx, uint32_t long_thr, const uint2* __restrict__ ranges,
                         const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ unit_tile,
                         const uint32_t* __restrict__ point_list, const float4* __restrict__ g0,
                         const float4* __restrict__ g1, const float* __restrict__ feats, float4* __restrict__ part,
                         uint32_t* __restrict__ part_last)
{
    constexpr int SV = snap_vecs(C);
    __shared__ Slot<C> entries[64 + 1];
    __shared__ __attribute__((aligned(4))) uint8_t qidx[4][QCAP];
    // same XCD-aware placement as the backward: the four blocks of a unit on one XCD, runs of 8 units per XCD
    const uint32_t n_units = gridDim.x >> 2;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t grp = slot >> 2;
    uint32_t unit = (grp >> 3) * 64u + xcd * 8u + (grp & 7u);
    uint32_t wave = slot & 3u;
    const uint32_t full = (n_units >> 6) << 6;
    if (blockIdx.x >= full * 4u) { unit = blockIdx.x >> 2; wave = blockIdx.x & 3u; }
    const int tile = (int)unit_tile[unit];
    const uint2 rg = ranges[tile];
    const uint32_t n = rg.y - rg.x;
    if (n <= long_thr) return;
    const int lane = threadIdx.x, row = lane >> 4;
    const uint32_t unit0 = seg_off[tile];
    const uint32_t base = (unit - unit0) * 64u;
    const int tx = tile % gx, ty = tile / gx;
    const int sx = tx * TILE + (int)(wave & 1) * SUB, sy = ty * TILE + (int)(wave >> 1) * SUB;
    const int px = sx + (row & 1) * 4 + (lane & 3), py = sy + (row >> 1) * 4 + ((lane >> 2) & 3);
    const int pix_in_tile = 16 * (py - ty * TILE) + (px - tx * TILE);
    init_neutral<C>(entries, lane);
    PixState<C> ps;
    ps.T = 1.0f; ps.last = 0; ps.done = !(px < W && py < H);
#pragma unroll
    for (int ch = 0; ch < C; ch++) ps.Cc[ch] = 0.f;
    const Fetched<C> cur = fetch_record<C>(fetch_id(base + lane, n, point_list + rg.x), g0, g1, feats);
    walk_batch<C, false>(entries, qidx, cur, base, ~0ull, lane, row, sx, sy, (float)px, (float)py, ps);
    store_snapshot<C>(part + ((size_t)unit * 256 + pix_in_tile) * SV, ps.T, ps.Cc);
    part_last[(size_t)unit * 256 + pix_in_tile] = ps.last;
}

// The common case keeps its blend loop inline (the shared walk_batch() costs it 16 VGPRs = one wave per SIMD).
template <int C>
__global__ void __launch_bounds__(256)
// 3 channels: 97 VGPRs round up to 104 = four waves per SIMD; asking for five costs no spill (87 VGPRs) and gains 2 us.
// (4 and 6 channels spill under the same request, six waves spill for three channels: both measured slower.)
__attribute__((amdgpu_waves_per_eu(C == 3 ? 5 : 4)))
blend_fwd_kernel(int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ order,
                 uint32_t* point_list, const uint64_t* __restrict__ sort_keys, const float4* __restrict__ g0,
                 const float4* __restrict__ g1, const float* __restrict__ feats, const float* __restrict__ bg,
                 float* __restrict__ out_color, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                 const uint32_t* __restrict__ seg_off, float4* __restrict__ snap, uint32_t skip_above,
                 float4* __restrict__ zero_ptr, uint32_t zero_n, uint32_t* __restrict__ counters, uint32_t counters_tp,
                 uint64_t* __restrict__ trace)
{
    const uint64_t t_start = trace ? wall_clock64() : 0;
    // Side job: the backward's accumulation table (48 B per Gaussian) has to be zero before blend_bwd runs.  When the
    // caller hands it over at forward time every workgroup clears its slice here -- the kernel is issue-bound and leaves
    // HBM idle -- instead of a separate fill (a 5 us blit plus its dispatch) in front of the backward.
    if (zero_ptr != nullptr) {
        const uint32_t per = (zero_n + gridDim.x - 1u) / gridDim.x;
        const uint32_t i0 = blockIdx.x * per, i1 = min(zero_n, i0 + per);
        for (uint32_t i = i0 + threadIdx.x; i < i1; i += 256u) zero_ptr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    constexpr int SV = snap_vecs(C);
    constexpr int CV = (C + 3) / 4;                                   // float4s of colour per slot
    // LDS: entries[wave][batch lane] (slot 64 = neutral) | qidx[wave][quadrant][queue position] = LDS byte address of the
    // queued slot (absolute, so a queue word feeds ds_read directly: one extract per entry instead of extract +
    // multiply-add).  The sort in front of the blend (below) uses the same bytes for its cross-wave stages.
    constexpr size_t ENT_BYTES = sizeof(Slot<C>) * 4 * (64 + 1), Q_BYTES = sizeof(uint16_t) * 4 * 4 * QCAP;
    constexpr size_t LDS_BYTES = ENT_BYTES + Q_BYTES > 2048 * 8 ? ENT_BYTES + Q_BYTES : 2048 * 8;
    static_assert(ENT_BYTES % 8 == 0, "queue words are read as 8-byte pairs");
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    Slot<C> (*entries)[64 + 1] = reinterpret_cast<Slot<C>(*)[64 + 1]>(smem);
    uint16_t (*qidx)[4][QCAP] = reinterpret_cast<uint16_t(*)[4][QCAP]>(smem + ENT_BYTES);
    const int tile = (int)order[blockIdx.x];
    // Second side job (fused forward): this tile's eight shard counters and eight scatter cursors live in a library-owned
    // block that has to be all zero again for the next view's preprocess; scatter, their last reader, is done.
    if (counters != nullptr && threadIdx.x < 2 * NSHARD)
        counters[(size_t)threadIdx.x * counters_tp + tile] = 0u;      // rows 0-7: counts, rows 8-15: cursors ([16][Tp])
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = lane >> 4;                                        // DPP row = quadrant
    const int tx = tile % gx, ty = tile / gx;
    const int sx = tx * TILE + (wave & 1) * SUB, sy = ty * TILE + (wave >> 1) * SUB;
    const int px = sx + (row & 1) * 4 + (lane & 3), py = sy + (row >> 1) * 4 + ((lane >> 2) & 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const int pix_in_tile = 16 * (py - ty * TILE) + (px - tx * TILE);

    const uint2 rg = ranges[tile];
    const uint32_t n = rg.y - rg.x;
    if (n > skip_above) return;   // long tile: blend_fwd_long_kernel renders it
    // Depth sort of this tile's list, right here (lists up to 2 048 entries; longer ones were sorted by tile_sort_big_kernel
    // before this launch).  As a kernel of its own the sort is latency-bound (key loads, cross-lane exchanges, barriers:
    // 25 us at a fraction of the vector ALU) and the blend then starts from a cold chip; inside the blend kernel one
    // tile's sort overlaps the other resident tiles' blending, and the sorted ids are read back while still in L2.
    if (sort_keys != nullptr) {
        if (n >= 1u && n <= 2048u) sort_small_tile(reinterpret_cast<uint64_t*>(smem), sort_keys + rg.x, point_list + rg.x, n);
        __syncthreads();   // ids visible to the four waves; the sort's LDS is free for the queues
    }
    const uint32_t unit0 = seg_off[tile];
    const uint32_t* list = point_list + rg.x;
    Slot<C>* ent = entries[wave];
    uint16_t (*qi)[QCAP] = qidx[wave];
    const uint32_t ent_lds = (uint32_t)(uintptr_t)ent;                       // LDS byte address of this wave's slots
    const uint16_t my_slot = (uint16_t)(ent_lds + lane * (uint32_t)sizeof(Slot<C>));
    const uint32_t neutral = ent_lds + 64u * (uint32_t)sizeof(Slot<C>);
    const uint2 neutral4 = make_uint2(neutral * 0x10001u, neutral * 0x10001u);

    if (lane == 0) {   // neutral instance: opacity 0 never passes the alpha test
        ent[64].a = make_float4(0.f, 0.f, 0.f, 0.f);
        ent[64].b = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ch = 0; ch < CV * 4; ch++) ent[64].col[ch] = 0.f;
    }

    float T = 1.0f;
    float Cc[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) Cc[ch] = 0.f;
    uint32_t last = 0;
    bool done = !inside;
    const unsigned long long lt = (1ull << lane) - 1ull;

    Fetched<C> nxt = fetch_record<C>(fetch_id(lane, n, list), g0, g1, feats);
    uint32_t gid_nxt = fetch_id(64 + lane, n, list);
    for (uint32_t base = 0; base < n; base += 64) {
        const unsigned long long alive = __ballot(!done);
        if (alive == 0ull) break;
        // segment boundary: snapshot the running state of every pixel still alive (the backward blend
        // starts its segments from these instead of replaying the whole list)
        if (snap != nullptr && base != 0 && (base % SEG) == 0 && !done)
            store_snapshot<C>(snap + ((size_t)(unit0 + base / SEG) * 256 + pix_in_tile) * SV, T, Cc);
        const Fetched<C> cur = nxt;
        nxt = fetch_record<C>(gid_nxt, g0, g1, feats);      // records of batch +1 (ids arrived during the last batch)
        gid_nxt = fetch_id(base + 128 + lane, n, list);     // ids of batch +2
        const uint32_t k = base + lane;
        {
            float a2 = cur.a.z, b2 = cur.a.w, c2 = cur.b.x;
            conic_to_exp2(a2, b2, c2);
            ent[lane].a = make_float4(cur.a.x, cur.a.y, a2, b2);
            ent[lane].b = make_float4(c2, cur.b.y, __uint_as_float(k + 1), 0.f);
        }
#pragma unroll
        for (int v = 0; v < CV; v++)
            reinterpret_cast<float4*>(ent[lane].col)[v] =
                make_float4(cur.col[4 * v], 4 * v + 1 < C ? cur.col[4 * v + 1] : 0.f, 4 * v + 2 < C ? cur.col[4 * v + 2] : 0.f,
                            4 * v + 3 < C ? cur.col[4 * v + 3] : 0.f);
        // exact reachability test against each quadrant that still has an unsaturated pixel
        int cnt[4];
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            const float X0 = (float)(sx + (qd & 1) * 4) - cur.a.x, Y0 = (float)(sy + (qd >> 1) * 4) - cur.a.y;
            const bool keep = ((alive >> (16 * qd)) & 0xffffull) != 0ull &&
                              block_min_half_quad(cur.a.z, cur.a.w, cur.b.x, X0, X0 + 3.f, Y0, Y0 + 3.f) <= cur.b.z;
            const unsigned long long m = __ballot(keep);
            cnt[qd] = __popcll(m);
            if (keep) qi[qd][__popcll(m & lt)] = my_slot;
            if (lane < 4) qi[qd][cnt[qd] + lane] = (uint16_t)neutral;   // pad to a multiple of 4 with the neutral instance
        }
        const int my_cnt = row == 0 ? cnt[0] : row == 1 ? cnt[1] : row == 2 ? cnt[2] : cnt[3];
        const int max_cnt = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3]));
        __builtin_amdgcn_wave_barrier();
        const uint16_t* myq = qi[row];
        for (int j = 0; j < max_cnt; j += 4) {
            // four queue positions at once; rows past their own queue end read the neutral instance
            const uint2 packed = j < my_cnt ? *reinterpret_cast<const uint2*>(myq + j) : neutral4;
            float4 A[4], B[4], K[4][CV];
            float alpha[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t word = u < 2 ? packed.x : packed.y;
                const Slot<C>* sl = lds_slot<C>((u & 1) ? word >> 16 : word & 0xffffu);
                A[u] = sl->a; B[u] = sl->b;
#pragma unroll
                for (int v = 0; v < CV; v++) K[u][v] = reinterpret_cast<const float4*>(sl->col)[v];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float dx = A[u].x - pxf, dy = A[u].y - pyf;
                const float power = pair_exp2_arg(A[u].z, A[u].w, B[u].x, dx, dy);   // exp2 domain, see conic_to_exp2
                alpha[u] = fminf(ALPHA_MAX, B[u].y * __builtin_amdgcn_exp2f(power));
                ok[u] = power <= 0.0f && alpha[u] >= ALPHA_MIN;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float test_T = T * (1.0f - alpha[u]);
                const bool live = ok[u] && !done;
                const bool stop = live && test_T < T_EPS;
                const bool upd = live != stop;   // stop implies live: one lane-mask xor instead of a second compare
                done = done || stop;
                const float w = upd ? alpha[u] * T : 0.0f;
#pragma unroll
                for (int ch = 0; ch < C; ch++) {
                    const float4 kv = K[u][ch / 4];
                    Cc[ch] += (ch % 4 == 0 ? kv.x : ch % 4 == 1 ? kv.y : ch % 4 == 2 ? kv.z : kv.w) * w;
                }
                T = upd ? test_T : T;
                last = upd ? __float_as_uint(B[u].z) : last;
            }
        }
        __builtin_amdgcn_wave_barrier();
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
        if (snap != nullptr && n > (uint32_t)SEG) store_snapshot<C>(snap + ((size_t)unit0 * 256 + pix_in_tile) * SV, T, Cc);
    }
    if (trace && lane == 0) {   // last wave to finish wins the end stamp
        if (wave == 0) trace[2 * blockIdx.x] = t_start;
        atomicMax((unsigned long long*)&trace[2 * blockIdx.x + 1], (unsigned long long)wall_clock64());
    }
}

// Long tiles: steps through the pre-reduced segments (see the file header); one workgroup = one tile, like the main
// kernel, launched over the front of `order` where the longest lists sit.
//
// Termination.  A pixel can only fall below T = 1e-4 inside segment s if T * P_s < 1e-4.  Such a pixel is PARKED at the
// start of s (its state frozen) while the others step on; when every pixel of the wave is done or parked, each parked
// lane walks ITS OWN segment serially -- 64 entries, gathered per lane, the same per-pair arithmetic and order as
// the batch walk (no culling is needed for correctness: a culled pair fails the alpha test anyway) -- all parked
// lanes at once, each in a different segment.  A wave's 64 pixels terminate in up to 64 different segments; walking
// a whole batch for the wave whenever one of them did made this path a loss on opaque surfaces.  In the rare case
// that the exact walk does not terminate after all (T * P_s was within rounding of 1e-4), the lane steps on.
template <int C>
__global__ void __launch_bounds__(256)
blend_fwd_long_kernel(int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ order,
                      const uint32_t* __restrict__ point_list, const float4* __restrict__ g0,
                      const float4* __restrict__ g1, const float* __restrict__ feats, const float* __restrict__ bg,
                      float* __restrict__ out_color, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                      const uint32_t* __restrict__ seg_off, float4* __restrict__ snap, uint32_t long_thr,
                      const float4* __restrict__ part, const uint32_t* __restrict__ part_last)
{
    constexpr int SV = snap_vecs(C);
    const int tile = (int)order[blockIdx.x];
    const uint2 rg = ranges[tile];
    const uint32_t n = rg.y - rg.x;
    if (n <= long_thr) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = tile % gx, ty = tile / gx;
    const int sx = tx * TILE + (wave & 1) * SUB, sy = ty * TILE + (wave >> 1) * SUB;
    const int px = sx + (lane & 7), py = sy + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const int pix_in_tile = 16 * (py - ty * TILE) + (px - tx * TILE);
    const uint32_t unit0 = seg_off[tile];
    const uint32_t* list = point_list + rg.x;
    const uint32_t n_seg = (n + 63u) / 64u;
    const size_t pbase = (size_t)unit0 * 256 + pix_in_tile;

    float T = 1.0f, Cc[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) Cc[ch] = 0.f;
    uint32_t last = 0;
    bool done = !inside;
    uint32_t seg = 0;                 // next segment this pixel has to consume
    while (true) {
        // ---- step through pre-reduced segments until the pixel is done, parked or out of segments
        bool parked = false;
        {
            float Pn = 1.f, Cn[C];
            uint32_t Ln = 0;
#pragma unroll
            for (int ch = 0; ch < C; ch++) Cn[ch] = 0.f;
            // the wave advances in lock step over segment indices; lanes join at their own `seg`
            uint32_t s_lo = done ? n_seg : seg;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) s_lo = min(s_lo, (uint32_t)__shfl_xor((int)s_lo, d, 64));
            if (s_lo < n_seg) {
                load_snapshot<C>(part + (pbase + (size_t)s_lo * 256) * SV, Pn, Cn);
                Ln = part_last[pbase + (size_t)s_lo * 256];
            }
            for (uint32_t s = s_lo; s < n_seg; s++) {
                const bool active = !done && !parked && seg == s;
                if (__ballot(!done && !parked) == 0ull) break;
                if (snap != nullptr && s != 0 && active)
                    store_snapshot<C>(snap + ((size_t)(unit0 + s) * 256 + pix_in_tile) * SV, T, Cc);
                const float Ps = Pn;
                float Cs[C];
#pragma unroll
                for (int ch = 0; ch < C; ch++) Cs[ch] = Cn[ch];
                const uint32_t Ls = Ln;
                if (s + 1 < n_seg) {   // next segment's record is requested before this one is consumed
                    load_snapshot<C>(part + (pbase + (size_t)(s + 1) * 256) * SV, Pn, Cn);
                    Ln = part_last[pbase + (size_t)(s + 1) * 256];
                }
                if (active) {
                    const float Tn = T * Ps;
                    if (Tn < T_EPS) {
                        parked = true;                      // may terminate inside segment s: walk it exactly below
                    } else {
#pragma unroll
                        for (int ch = 0; ch < C; ch++) Cc[ch] += T * Cs[ch];
                        T = Tn;
                        last = Ls ? Ls : last;
                        seg = s + 1;
                    }
                }
            }
            if (!done && !parked && seg >= n_seg) done = true;   // consumed the whole list without terminating
        }
        if (__ballot(parked) == 0ull) break;
        // ---- every parked lane walks its own segment [64 seg, 64 seg + 64) serially; two-stage gather prefetch
        {
            const uint32_t base = seg * 64u;
            const uint32_t cnt = parked ? min(64u, n - base) : 0u;
            uint32_t cmax = cnt;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, d, 64));
            Fetched<C> nxt = fetch_record<C>(cnt > 0u ? list[base] : 0xffffffffu, g0, g1, feats);
            uint32_t gid_nxt = cnt > 1u ? list[base + 1] : 0xffffffffu;
            for (uint32_t k = 0; k < cmax; k++) {
                const Fetched<C> cur = nxt;
                nxt = fetch_record<C>(gid_nxt, g0, g1, feats);
                gid_nxt = k + 2 < cnt ? list[base + k + 2] : 0xffffffffu;
                if (k < cnt && !done) {
                    const float dx = cur.a.x - pxf, dy = cur.a.y - pyf;
                    float a2 = cur.a.z, b2 = cur.a.w, c2 = cur.b.x;
                    conic_to_exp2(a2, b2, c2);   // the same roundings as the batch walk's queued slots
                    const float power = pair_exp2_arg(a2, b2, c2, dx, dy);
                    const float alpha = fminf(ALPHA_MAX, cur.b.y * __builtin_amdgcn_exp2f(power));
                    if (power <= 0.0f && alpha >= ALPHA_MIN) {
                        const float test_T = T * (1.0f - alpha);
                        if (test_T < T_EPS) {
                            done = true;
                        } else {
                            const float w = alpha * T;
#pragma unroll
                            for (int ch = 0; ch < C; ch++) Cc[ch] += cur.col[ch] * w;
                            T = test_T;
                            last = base + k + 1u;
                        }
                    }
                }
            }
            if (parked) seg = seg + 1;          // if it did not terminate after all, it continues with the next segment
            if (parked && !done && seg >= n_seg) done = true;
        }
        if (__ballot(!done) == 0ull) break;
    }
    if (inside) {
        const size_t pix = (size_t)W * py + px;
        const size_t HW = (size_t)H * W;
        final_T[pix] = T;
        n_contrib[pix] = last;
#pragma unroll
        for (int ch = 0; ch < C; ch++) out_color[ch * HW + pix] = Cc[ch] + T * bg[ch];
        if (snap != nullptr) store_snapshot<C>(snap + ((size_t)unit0 * 256 + pix_in_tile) * SV, T, Cc);   // n > SEG always here
    }
}

template <int C>
static void launch_fwd_c(int W, int H, int R, int U, uint32_t max_count, const float* bg, const float* feats, GeomState g, ImageState im,
                         BinState b, float* out_color, void* zero_ptr, size_t zero_bytes, uint32_t* counters, bool sort_small,
                         hipStream_t st)
{
    const Tiles t = tiles_of(W, H);
    const uint32_t long_thr = fwd_long_threshold();
    const bool use_long = U > 0 && b.part != nullptr && max_count > long_thr && long_thr >= (uint32_t)SEG;
    // Everything stays on the caller's stream.  Running the long tiles on a library-owned helper stream (fork / join
    // events) beside the main kernel was measured and rejected: after a single use EVERY later step of the process was
    // ~0.16 ms slower (0.386 -> 0.55 ms per config-C view), far more than the overlap ever saved.
    if (use_long) {
        blend_fwd_partial_kernel<C><<<4 * U, 64, 0, st>>>(W, H, t.gx, long_thr, im.ranges, im.seg_off, b.unit_tile, b.point_list,
                                                           g.g0, g.g1, feats, b.part, b.part_last);
        // the long tiles sit at the front of `order` (front_of_order); the kernel checks each tile's length itself
        blend_fwd_long_kernel<C><<<long_thr >= 2017u ? front_of_order(R, t.T) : t.T, 256, 0, st>>>(
            W, H, t.gx, im.ranges, im.order, b.point_list, g.g0, g.g1, feats, bg, out_color, im.final_T, im.n_contrib, im.seg_off,
            b.snap, long_thr, b.part, b.part_last);
    }
    blend_fwd_kernel<C><<<t.T, 256, 0, st>>>(W, H, t.gx, im.ranges, im.order, b.point_list, sort_small ? b.keys : nullptr,
                                             g.g0, g.g1, feats, bg, out_color,
                                             im.final_T, im.n_contrib, im.seg_off, b.snap, use_long ? long_thr : 0xffffffffu,
                                             static_cast<float4*>(zero_ptr), (uint32_t)(zero_bytes / 16), counters,
                                             (uint32_t)shard_stride(t.T), g_trace);
}

void launch_blend_fwd(int C, int W, int H, int R, int U, uint32_t max_count, const float* bg, const float* feats, GeomState g,
                      ImageState im, BinState b, float* out_color, void* zero_ptr, size_t zero_bytes, uint32_t* counters,
                      bool sort_small, hipStream_t st)
{
    if (C == 6) launch_fwd_c<6>(W, H, R, U, max_count, bg, feats, g, im, b, out_color, zero_ptr, zero_bytes, counters, sort_small, st);
    else if (C == 4) launch_fwd_c<4>(W, H, R, U, max_count, bg, feats, g, im, b, out_color, zero_ptr, zero_bytes, counters, sort_small, st);
    else launch_fwd_c<3>(W, H, R, U, max_count, bg, feats, g, im, b, out_color, zero_ptr, zero_bytes, counters, sort_small, st);
}

}  // namespace gsr
