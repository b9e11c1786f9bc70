// gsr_binning.hip -- tile-offset scan, instance scatter into per-tile buckets, per-tile depth sort.
//
// Replaces the reference's global pipeline InclusiveSum -> duplicateWithKeys -> 64-bit
// DeviceRadixSort (5-6 full passes over R keys) -> identifyTileRanges
// (DGR/cuda_rasterizer/rasterizer_impl.cu:70-138, :277-317) with a counting sort on the tile id
// (counts were taken by preprocess) followed by an independent depth sort of every tile's bucket
// inside LDS.  The sorted order is the reference's: tile-major, view depth ascending (compared as
// raw float bits, all positive), ties by ascending Gaussian id -- which is what the reference's
// STABLE radix sort yields because duplicateWithKeys emits instances in id order.
#include "gsr_internal.h"
#include "gsr_sort.h"
#include "gsr_plan.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace gsr {

// Barrier of the scan kernel: LDS traffic only.  __syncthreads() also waits for the global stores in flight (ranges,
// segment offsets, totals, the pinned host pad) -- a store's full latency, 2 - 3 us, at each of the kernel's barriers, although
// nothing in the kernel ever reads those back.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One workgroup scans all T tile counts: ranges[t] = {start, start+count}, cursor[t] = start,
// totals = {R, max count, #non-empty tiles}, and builds `order`: a counting sort of the tiles by
// descending list length in 32-entry buckets (empty tiles last).
constexpr int NBUCKET = 66;   // bucket 0 = longest (>= 2048 entries) ... bucket 64 = 1..32 entries, bucket 65 = empty
__device__ __forceinline__ int length_bucket(uint32_t c)
{
    return c == 0 ? 65 : 64 - (int)min(64u, (c + 31u) >> 5);
}
// PER = tiles owned by each of the 1024 threads, held in registers (the kernel is a single workgroup, so its
// time is the sum of its dependent memory round trips: the eight shard counters of every tile are summed
// here, straight from the preprocess counters, instead of by a separate launch).
// The histogram over length buckets is privatised NCOPY ways by lane: almost all tiles of a view fall into a
// few buckets, and same-address LDS atomics serialise per lane.
constexpr int NCOPY = 16;
constexpr int NHIST = NBUCKET * NCOPY;
static_assert(NHIST <= 2048 && NHIST % 2 == 0, "two histogram entries per thread in the prefix pass");

template <int PER>
__global__ void __launch_bounds__(1024)
tile_scan_kernel(int T, const uint32_t* __restrict__ tile_count, uint2* __restrict__ ranges,
                 uint32_t* __restrict__ totals, uint32_t* __restrict__ host_totals, uint32_t host_seq,
                 uint32_t* __restrict__ order, uint32_t* __restrict__ seg_off, uint32_t view_token, uint32_t split_from_word)
{
    static_assert((NSHARD & (NSHARD - 1)) == 0, "shard = workgroup index & (NSHARD - 1)");
    __shared__ uint32_t parts_lds;   // parts (forward chunks) of the lists this view blends in parts: the host picks the launch shape by it
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t wave_seg[16];
    __shared__ uint32_t wave_max[16];
    __shared__ uint32_t wave_hist[16];
    __shared__ uint32_t hist[NHIST];   // [bucket][copy]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int copy = lane & (NCOPY - 1);
#ifdef GSR_SCAN_TRACE
    uint64_t stamp[8]; int ns = 0;
#define SCAN_STAMP() do { if (tid == 0) stamp[ns++] = wall_clock64(); } while (0)
#else
#define SCAN_STAMP() do { } while (0)
#endif
    SCAN_STAMP();
    for (int i = tid; i < NHIST; i += 1024) hist[i] = 0;
    if (tid == 0) parts_lds = 0u;
    // PER > 0: this thread's PER tiles live in registers.  PER == 0 (images above 8 192 tiles): the thread owns
    // ceil(T / 1024) consecutive tiles and re-reads their counters (L2-resident) in each of the three passes.
    constexpr bool IN_REGS = PER > 0;
    const int per = IN_REGS ? PER : (T + 1023) / 1024;
    const int t0 = tid * per;
    uint32_t c[IN_REGS ? PER : 1];
    const size_t Tp = shard_stride(T);
    auto count_of = [&](int t) -> uint32_t {
        if (t >= T) return 0u;
        uint32_t v = 0;
#pragma unroll
        for (int sh = 0; sh < NSHARD; sh++) v += tile_count[sh * Tp + t];
        return v;
    };
    if constexpr (PER == 8) {
        // shard-major rows: a thread's 8 tiles are 32 contiguous bytes in each of the 8 rows, lanes are contiguous
        // -> 16 fully coalesced 16-byte loads per thread (rows are padded, so the tail stays inside the row)
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = 0;
        if ((size_t)t0 < Tp) {
#pragma unroll
            for (int sh = 0; sh < NSHARD; sh++) {
                const uint4* pr = reinterpret_cast<const uint4*>(tile_count + sh * Tp + t0);
                const uint4 a = pr[0], b = pr[1];
                c[0] += a.x; c[1] += a.y; c[2] += a.z; c[3] += a.w; c[4] += b.x; c[5] += b.y; c[6] += b.z; c[7] += b.w;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) if (t0 + k >= T) c[k] = 0;
    } else if constexpr (IN_REGS) {
#pragma unroll
        for (int k = 0; k < PER; k++) c[k] = count_of(t0 + k);
    }
    auto cnt = [&](int k) -> uint32_t {
        if constexpr (IN_REGS) return c[k]; else return count_of(t0 + k);
    };
    uint32_t sum = 0, vmax = 0, segs = 0;
#ifdef GSR_SCAN_TRACE
    if constexpr (IN_REGS) { if (c[0] == 0xfffffff0u) hist[0] = 1; }   // (keeps the loads ahead of the stamp; never true)
#endif
    SCAN_STAMP();   // counts loaded
#pragma unroll
    for (int k = 0; k < per; k++) {
        const uint32_t ck = cnt(k);
        sum += ck;
        segs += (ck + (uint32_t)SEG - 1u) / (uint32_t)SEG;
        vmax = max(vmax, ck);
    }
    const uint32_t incl = wave_incl_scan(sum, lane);
    const uint32_t incl_seg = wave_incl_scan(segs, lane);
    vmax = wave_max_to_lane63(vmax);
    if (lane == 63) { wave_sum[wave] = incl; wave_seg[wave] = incl_seg; wave_max[wave] = vmax; }
    lds_barrier();
    SCAN_STAMP();   // wave scans + first barrier
    uint32_t woff = 0, total = 0, gmax = 0, woff_seg = 0, total_seg = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const uint32_t s_ = wave_sum[w], sg = wave_seg[w];
        if (w < wave) { woff += s_; woff_seg += sg; }
        total += s_;
        total_seg += sg;
        gmax = max(gmax, wave_max[w]);
    }
    uint32_t run = woff + incl - sum;
    uint32_t run_seg = woff_seg + incl_seg - segs;
    uint32_t n_empty = 0;   // empty tiles dominate: count them privately, one LDS atomic per thread
    const uint32_t split_n = split_threshold_from(gmax, total, split_from_word);
    uint32_t my_parts = 0;
    if constexpr (PER == 8) {
        // the 8 ranges (64 B) and 8 segment offsets (32 B) of a thread leave as 16-byte stores; the image-state carve
        // pads both arrays, and tiles >= T carry empty ranges that nobody reads
        uint32_t rs[9], sg[8];
        rs[0] = run;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            sg[k] = run_seg;
            rs[k + 1] = rs[k] + c[k];
            run_seg += (c[k] + (uint32_t)SEG - 1u) / (uint32_t)SEG;
            if (t0 + k < T && c[k] > split_n) my_parts += (c[k] + (uint32_t)FWD_CHUNK - 1u) / (uint32_t)FWD_CHUNK;
            if (t0 + k < T) {
                if (c[k] == 0) n_empty++;
                else atomicAdd(&hist[length_bucket(c[k]) * NCOPY + copy], 1u);
            }
        }
        run = rs[8];
        if (t0 + 7 < T) {
            uint4* pr = reinterpret_cast<uint4*>(ranges + t0);
#pragma unroll
            for (int k = 0; k < 8; k += 2) pr[k / 2] = make_uint4(rs[k], rs[k + 1], rs[k + 1], rs[k + 2]);
            uint4* ps = reinterpret_cast<uint4*>(seg_off + t0);
            ps[0] = make_uint4(sg[0], sg[1], sg[2], sg[3]);
            ps[1] = make_uint4(sg[4], sg[5], sg[6], sg[7]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (t0 + k < T) { ranges[t0 + k] = make_uint2(rs[k], rs[k + 1]); seg_off[t0 + k] = sg[k]; }
        }
    } else {
#pragma unroll
    for (int k = 0; k < per; k++) {
        const int t = t0 + k;
        if (t < T) {
            const uint32_t ck = cnt(k);
            ranges[t] = make_uint2(run, run + ck);
            seg_off[t] = run_seg;
            run += ck;
            run_seg += (ck + (uint32_t)SEG - 1u) / (uint32_t)SEG;
            if (ck > split_n) my_parts += (ck + (uint32_t)FWD_CHUNK - 1u) / (uint32_t)FWD_CHUNK;
            if (ck == 0) n_empty++;
            else atomicAdd(&hist[length_bucket(ck) * NCOPY + copy], 1u);
        }
    }
    }
    if (n_empty) atomicAdd(&hist[(NBUCKET - 1) * NCOPY + copy], n_empty);
    if (my_parts) atomicAdd(&parts_lds, my_parts);
    lds_barrier();
    SCAN_STAMP();   // ranges / seg_off stored, histogram counted
    uint32_t nonempty = 0, n_parts = 0;   // (thread 0)
    if (tid == 0) {
        n_parts = parts_lds;
        uint32_t empty = 0;
        for (int i = 0; i < NCOPY; i++) empty += hist[(NBUCKET - 1) * NCOPY + i];
        nonempty = (uint32_t)T - empty;
        totals[0] = total;
        totals[1] = gmax;
        totals[2] = nonempty;
        totals[3] = total_seg;
        totals[6] = 0u;           // scatter counts the parts of long lists here
        totals[5] = view_token;   // scatter compares it with totals[4] (set by a preprocess workgroup that ran out of room)
        seg_off[T] = total_seg;
    }
    // exclusive prefix over the flattened [bucket][copy] histogram (in place): two entries per thread
    uint32_t e0 = 0, e1 = 0;
    if (2 * tid < NHIST) { e0 = hist[2 * tid]; e1 = hist[2 * tid + 1]; }
    const uint32_t hs = e0 + e1;
    const uint32_t hincl = wave_incl_scan(hs, lane);
    if (lane == 63) wave_hist[wave] = hincl;
    lds_barrier();   // also orders thread 0's reads of the empty bucket before the overwrite below
    SCAN_STAMP();   // totals out, histogram prefix (wave level)
    uint32_t hoff = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) hoff += w < wave ? wave_hist[w] : 0u;
    if (2 * tid < NHIST) {
        const uint32_t excl = hoff + hincl - hs;
        hist[2 * tid] = excl;
        hist[2 * tid + 1] = excl + e0;
    }
    lds_barrier();
    SCAN_STAMP();   // histogram prefix done
    // Launch positions follow a SNAKE over bands of 256 (one workgroup per CU per band under the observed
    // round-robin dispatch): CU k gets ranks k, 511-k, 512+k, ... so per-CU sums of list lengths even out
    // instead of CU 0 collecting the longest tile of every band.  Pure scheduling heuristic.
    auto snake = [](uint32_t pos) { return (pos & 256u) ? (pos ^ 255u) : pos; };
    uint32_t empty_at = n_empty ? atomicAdd(&hist[(NBUCKET - 1) * NCOPY + copy], n_empty) : 0u;
    const auto place = [&](uint32_t pos, int t) {
        const uint32_t sp = snake(pos);
        if (sp < (uint32_t)T && (pos | 255u) < (uint32_t)T) pos = sp;   // only inside complete bands
        order[pos] = (uint32_t)t;
    };
    // (one LDS atomic per RUN of equal length classes among a thread's eight tiles instead of one per tile was tried: the
    // grouping code costs more than the atomics it saves, 11.1 -> 13.9 us)
#pragma unroll
    for (int k = 0; k < per; k++) {
        const int t = t0 + k;
        if (t < T) {
            const uint32_t ck = cnt(k);
            const uint32_t pos = (ck == 0) ? empty_at++ : atomicAdd(&hist[length_bucket(ck) * NCOPY + copy], 1u);
            place(pos, t);
        }
    }
    // The host's copy of the totals goes straight into its pinned, device-mapped landing pad -- no separate device-to-host
    // copy (a 5 us blit kernel plus its dispatch) -- followed by the call's sequence number with system-scope release; the
    // host polls that word (gsr_forward_stage1).  LAST thing thread 0 does: the release waits for the stores to the pad to
    // cross the bus (~1 us), and in the middle of the kernel the other fifteen waves stood at the next barrier for it.  The
    // host does not need the totals sooner: what it launches with them is stream-ordered behind scatter anyway.
    if (tid == 0 && host_totals) {
        *reinterpret_cast<uint4*>(host_totals) = make_uint4(total, gmax, nonempty, total_seg);
        host_totals[5] = n_parts;
        __hip_atomic_store(&host_totals[4], host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#ifdef GSR_SCAN_TRACE
    lds_barrier();
    SCAN_STAMP();   // order stored
    if (tid == 0 && (host_seq & 63u) == 0u)
        printf("scan phases (10 ns ticks): load %llu, scan+barrier %llu, ranges+hist %llu, totals+prefix1 %llu, prefix2 %llu, order %llu\n",
               stamp[1] - stamp[0], stamp[2] - stamp[1], stamp[3] - stamp[2], stamp[4] - stamp[3], stamp[5] - stamp[4], stamp[6] - stamp[5]);
#endif
}

void launch_tile_scan(ImageState im, int T, uint32_t* host_totals, uint32_t host_seq, uint32_t view_token, hipStream_t st)
{
    if (T <= 8 * 1024)          // up to 1920x1088: counts stay in registers
        tile_scan_kernel<8><<<1, 1024, 0, st>>>(T, im.tile_count, im.ranges, im.totals, host_totals, host_seq, im.order, im.seg_off, view_token, split_from());
    else                        // any larger grid (gsr_forward_stage1 caps T at 262 144 = 8k x 8k)
        tile_scan_kernel<0><<<1, 1024, 0, st>>>(T, im.tile_count, im.ranges, im.totals, host_totals, host_seq, im.order, im.seg_off, view_token, split_from());
}

// Every (Gaussian, tile) instance's sort key goes to its slot of the tile's bucket.  Workgroup b serves the 256 Gaussians
// preprocess workgroup b looked at.  Normally those left RECORDS {gaussian, depth bits, table slot, offset} and a table
// {tile, first rank of the workgroup's span within (tile, shard)} (gsr_preprocess.hip): 256 threads turn the table into
// absolute bucket positions (tile start + counts of the lower shards + first rank), then it is one load and one store per
// record -- no tile walk, no atomics.  A view in which some workgroup ran out of table or record space carries its own
// token in totals[4]: then every workgroup walks exactly the tiles preprocess counted (same stored inputs, same
// contraction-free test, same wave aggregation) and claims slots with one returning atomic per (wave, tile) on the shard's
// cursor, every lane of the group taking its own by prefix popcount.
__global__ void __launch_bounds__(256)
scatter_kernel(int P, int gx, const ushort4* __restrict__ rect, const float4* __restrict__ g0,
               const float4* __restrict__ g1, const float* __restrict__ depth, uint32_t* __restrict__ tile_cursor,
               uint64_t* __restrict__ keys, int T, const uint32_t* __restrict__ seg_off,
               uint4* __restrict__ unit_info, const uint32_t* __restrict__ tile_count,
               const uint2* __restrict__ ranges, const uint4* __restrict__ wg_recs, const uint2* __restrict__ wg_tab,
               const uint32_t* __restrict__ wg_nrec, uint32_t* __restrict__ totals, uint2* __restrict__ part_list,
               uint32_t* __restrict__ part_ticket, uint32_t split_n, void* early_base, size_t early_capacity, int early_C)
{
    if (early_base != nullptr) {
        // (uniform) launched right behind the scan, before the host has seen its totals (gsr_forward_fused): where the
        // keys, the unit table and the part list live follows from R and U exactly as on the host (carve_bin), and a
        // view that does not fit the buffer the caller guessed is left alone -- the host then runs stage 2 itself.
        // (split_n carries the view-independent threshold `split_from` here.)
        const uint32_t R = totals[0], maxc = totals[1], U = totals[3];
        const BinState eb = carve_bin(early_base, (int)R, (int)U, early_C);
        if (eb.bytes > early_capacity) return;
        keys = eb.keys; unit_info = eb.unit_info; part_list = eb.part_list; part_ticket = eb.part_ticket;
        split_n = split_threshold_from(maxc, R, split_n);
    }
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int shard = (int)(blockIdx.x & (NSHARD - 1));
    const size_t Tp = shard_stride(T);
    // side job of the first T threads: expand the per-tile segment counts into the unit table -- everything a unit of
    // the backward blend has to know about its tile in ONE 16-byte load instead of a chain of
    // three (it lives in the binning buffer, which did not exist yet when the scan kernel ran)
    if (idx < T) {
        const uint32_t u0 = seg_off[idx], u1 = seg_off[idx + 1];
        const uint2 rg = ranges[idx];
        for (uint32_t u = u0; u < u1; u++) unit_info[u] = make_uint4((uint32_t)idx, rg.x, rg.y - rg.x, u0);
        // ... and list the parts (one forward chunk each) of a list the forward blends in parts (gsr_blend_fwd.hip)
        const uint32_t n = rg.y - rg.x;
        if (n > split_n) {
            const uint32_t np = (n + (uint32_t)FWD_CHUNK - 1u) / (uint32_t)FWD_CHUNK;
            const uint32_t at = atomicAdd(&totals[6], np);
            for (uint32_t k = 0; k < np; k++) part_list[at + k] = make_uint2((uint32_t)idx, k * (uint32_t)FWD_CHUNK);
            part_ticket[at] = 0u;   // the tile's parts draw tickets here; the last one combines (gsr_blend_fwd.hip)
        }
    }
    if (totals[4] != totals[5]) {   // (uniform) every preprocess workgroup of this view recorded all of its instances
        if ((size_t)blockIdx.x * 256 >= (size_t)P) return;
        __shared__ uint32_t slot_base[WG_TAB_SLOTS];
        static_assert(WG_TAB_SLOTS % 256 == 0, "whole rounds of the workgroup's 256 threads");
        // (the record count of this wave's quarter and its first batch of records are requested HERE, together with the
        // table row: behind the barrier they were the third and fourth dependent trip to memory of a workgroup that makes
        // four -- table -> tile start and shard counts -> count -> records.  A wave's quarter always exists, so the
        // speculative read of records [0, 64) is in bounds whatever the count turns out to be)
        constexpr uint32_t WAVE_CAP_ = WG_REC_CAP / 4;
        const uint32_t wv_ = threadIdx.x >> 6;
        const uint4* const recs_ = wg_recs + (size_t)blockIdx.x * WG_REC_CAP + (size_t)wv_ * WAVE_CAP_;
        const uint32_t nr_ = wg_nrec[blockIdx.x * 4 + wv_];
        const uint4 r0_ = recs_[lane];
        uint2 e_[WG_TAB_SLOTS / 256];
#pragma unroll
        for (int q = 0; q < WG_TAB_SLOTS / 256; q++) e_[q] = wg_tab[(size_t)blockIdx.x * WG_TAB_SLOTS + q * 256 + threadIdx.x];
#pragma unroll
        for (int q = 0; q < WG_TAB_SLOTS / 256; q++) {
            const uint2 e = e_[q];
            if (e.x != 0xffffffffu) {
                // all seven lower-shard counts are requested together (a loop with `if (s < shard)` compiles to dependent trips)
                uint32_t cnt[NSHARD - 1];
                const uint32_t start = ranges[e.x].x;
#pragma unroll
                for (int s_ = 0; s_ < NSHARD - 1; s_++) cnt[s_] = tile_count[s_ * Tp + e.x];
                uint32_t base = start + e.y;
#pragma unroll
                for (int s_ = 0; s_ < NSHARD - 1; s_++) base += s_ < shard ? cnt[s_] : 0u;
                slot_base[q * 256 + threadIdx.x] = base;
            }
        }
        __syncthreads();
        // every wave takes the records of the preprocess wave at its position (its quarter of the array)
        const uint32_t nr = min(nr_, WAVE_CAP_);   // (a wave that produced more flagged the view: this path is not taken then)
        if ((uint32_t)lane < nr) keys[slot_base[r0_.z] + r0_.w] = ((uint64_t)r0_.y << 32) | r0_.x;
        for (uint32_t i = lane + 64u; i < nr; i += 64u) {
            const uint4 r = recs_[i];
            keys[slot_base[r.z] + r.w] = ((uint64_t)r.y << 32) | r.x;
        }
        return;
    }
    ushort4 r = make_ushort4(0, 0, 0, 0);
    float4 a = make_float4(0.f, 0.f, 1.f, 0.f), b = make_float4(1.f, 0.f, -1.f, 0.f);
    uint64_t key = 0;
    if (idx < P) {   // all four loads issue together; culled Gaussians carry an empty rect
        r = rect[idx];
        a = g0[idx];
        b = g1[idx];
        key = ((uint64_t)__float_as_uint(depth[idx]) << 32) | (uint32_t)idx;
        if (!(r.z > r.x && r.w > r.y)) b.z = -1.f;
    }
    // (giant splats are placed by the whole wave below, exactly as preprocess counted them: gsr_internal.h CoopSplat)
    const bool big = b.z >= 0.0f && rect_is_big(r);
    for_each_tile_aggregated(big ? make_ushort4(0, 0, 0, 0) : r, a.x, a.y, a.z, a.w, b.x, b.z, gx, lane,
                             [&](int tile, bool is_leader, int group, int rank, int leader_lane) {
                                 uint32_t base = 0;
                                 if (is_leader) {
                                     // slot = tile start + counts of the lower shards + position inside this shard
                                     // (shard cursors start at zero; the scan kernel does not expand them).  All seven
                                     // counts are loaded unconditionally so that they are in flight TOGETHER with the
                                     // atomic; `if (s < shard) below += count[s]` compiles to seven dependent round trips.
                                     uint32_t cnt[NSHARD - 1];
                                     const uint32_t start = ranges[tile].x;
#pragma unroll
                                     for (int s_ = 0; s_ < NSHARD - 1; s_++) cnt[s_] = tile_count[s_ * Tp + tile];
                                     const uint32_t old = atomicAdd(&tile_cursor[shard * Tp + tile], (uint32_t)group);
                                     base = start + old;
#pragma unroll
                                     for (int s_ = 0; s_ < NSHARD - 1; s_++) base += s_ < shard ? cnt[s_] : 0u;
                                 }
                                 base = __shfl(base, leader_lane, 64);
                                 if (tile >= 0) keys[base + (uint32_t)rank] = key;
                             });
    for (unsigned long long bigs = __ballot(big); bigs != 0ull; bigs &= bigs - 1ull) {
        const int src = __ffsll((unsigned long long)bigs) - 1;
        const CoopSplat cs(r, a.x, a.y, a.z, a.w, b.x, b.z, src);
        const uint64_t src_key = ((uint64_t)(uint32_t)__shfl((int)(key >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)key, src, 64);
        for (int base0 = 0; base0 < cs.n; base0 += 64) {
            const int tile = cs.tile(base0, lane, gx);
            if (tile >= 0) {   // (every lane has a tile of its own: one returning atomic each on the shard's cursor)
                uint32_t cnt[NSHARD - 1];
                const uint32_t start = ranges[tile].x;
#pragma unroll
                for (int s_ = 0; s_ < NSHARD - 1; s_++) cnt[s_] = tile_count[s_ * Tp + tile];
                uint32_t at = start + atomicAdd(&tile_cursor[shard * Tp + tile], 1u);
#pragma unroll
                for (int s_ = 0; s_ < NSHARD - 1; s_++) at += s_ < shard ? cnt[s_] : 0u;
                keys[at] = src_key;
            }
        }
    }
}

void launch_scatter(int P, int W, int H, int R, uint32_t max_count, GeomState g, ImageState im, BinState b, hipStream_t st)
{
    const Tiles t = tiles_of(W, H);
    const int n = P > t.T ? P : t.T;
    scatter_kernel<<<(n + 255) / 256, 256, 0, st>>>(P, t.gx, g.rect, g.g0, g.g1, g.depth, im.tile_cursor, b.keys, t.T,
                                                    im.seg_off, b.unit_info, im.tile_count, im.ranges, g.wg_recs, g.wg_tab,
                                                    g.wg_nrec, im.totals, b.part_list, b.part_ticket, split_threshold(max_count, (uint32_t)(R > 0 ? R : 0)), nullptr, 0, 3);
}

void launch_scatter_early(int P, int W, int H, int C, GeomState g, ImageState im, void* binning_base, size_t capacity, hipStream_t st)
{
    const Tiles t = tiles_of(W, H);
    const int n = P > t.T ? P : t.T;
    scatter_kernel<<<(n + 255) / 256, 256, 0, st>>>(P, t.gx, g.rect, g.g0, g.g1, g.depth, im.tile_cursor, nullptr, t.T,
                                                    im.seg_off, nullptr, im.tile_count, im.ranges, g.wg_recs, g.wg_tab,
                                                    g.wg_nrec, im.totals, nullptr, nullptr, split_from(), binning_base, capacity, C);
}

// ---- per-tile bitonic sort of 64-bit keys in LDS.
// A launch handles the tiles with LOWER < n <= CAP in LDS (CAP = power-of-two capacity of the dynamic
// LDS array); with FALLBACK it also takes the tiles above CAP, sorting them in place in global memory
// (same network, one workgroup, workgroup-scope fences) -- correct for any size, only slower.
template <int CAP, int LOWER, bool FALLBACK>
__global__ void __launch_bounds__(256)
tile_sort_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ order, uint64_t* __restrict__ keys,
                 uint32_t* __restrict__ point_list)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* s = reinterpret_cast<uint64_t*>(smem_raw);
    const uint2 rg = ranges[order[blockIdx.x]];
    const uint32_t n = rg.y - rg.x;
    if (n <= (uint32_t)LOWER) return;
    if (!FALLBACK && n > (uint32_t)CAP) return;
    const int tid = threadIdx.x;
    uint64_t* gk = keys + rg.x;
    if (n == 1) {
        if (tid == 0) point_list[rg.x] = (uint32_t)gk[0];
        return;
    }
    uint32_t np2 = 2;
    while (np2 < n) np2 <<= 1;
    if (n <= (uint32_t)CAP) {
        for (uint32_t i = tid; i < np2; i += 256) s[i] = i < n ? gk[i] : ~0ull;
        __syncthreads();
        // Compare-exchange index i touches the 2j-aligned block of elements around 2i: for j <= 64 the 64
        // consecutive indices a wave owns (in each 256-stride pass) stay inside "its" 128 consecutive
        // elements, stage after stage -- those stages need no workgroup barrier (LDS is in-order per wave).
        // Only the stages with j >= 128 exchange data between waves: a 512-entry list sorts with 6
        // barriers instead of 45.
        for (uint32_t k = 2; k <= np2; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                const bool cross = j >= 128;
                if (cross) __syncthreads();
                for (uint32_t i = tid; i < (np2 >> 1); i += 256) {
                    // i-th compare-exchange of this stage: indices (lo, lo | j), lo has bit j clear.
                    const uint32_t lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                    const uint32_t hi = lo | j;
                    const bool asc = (lo & k) == 0;
                    const uint64_t a = s[lo], b = s[hi];
                    if ((a > b) == asc) { s[lo] = b; s[hi] = a; }
                }
                if (cross) __syncthreads();
                else __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < n; i += 256) point_list[rg.x + i] = (uint32_t)s[i];
    } else {
        // Global-memory fallback on the NORMALISED bitonic network (every compare-exchange ascending;
        // the first sub-stage of a k-block pairs lo with its mirror image in the block).  Indices >= n
        // are virtual +inf keys: a pair whose upper index is virtual is already in order, so it is skipped.
        for (uint32_t k = 2; k <= np2; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = tid; i < (np2 >> 1); i += 256) {
                    const uint32_t lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                    uint32_t q = lo | j;
                    if (j == (k >> 1)) {
                        const uint32_t blk = lo & ~(k - 1);
                        q = blk + (k - 1) - (lo - blk);
                    }
                    if (q < n) {
                        const uint64_t a = gk[lo], b = gk[q];
                        if (a > b) { gk[lo] = b; gk[q] = a; }
                    }
                }
                __threadfence_block();
                __syncthreads();
            }
        }
        for (uint32_t i = tid; i < n; i += 256) point_list[rg.x + i] = (uint32_t)gk[i];
    }
}

// (the register-resident bitonic sort lives in gsr_sort.h: blend_fwd runs it in front of its walk)

__global__ void __launch_bounds__(256)
tile_sort_reg_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ order, const uint64_t* __restrict__ keys,
                     uint32_t* __restrict__ point_list)
{
    __shared__ uint64_t s[2 * SORT_SMALL_CAP];
    const uint2 rg = ranges[order[blockIdx.x]];
    const uint32_t n = rg.y - rg.x;
    if (n == 0 || n > 2048u) return;     // longer lists: tile_sort_kernel<16384, 2048, true>
    const uint64_t* gk = keys + rg.x;
    uint32_t* out = point_list + rg.x;
    sort_small_tile(s, gk, out, n);
}

// Lists of 2 049 .. 16 384 entries (close-up views: a few dozen tiles of a 1 M-Gaussian model): the same register network
// with 1 024 threads per tile, E = 4 / 8 / 16 keys per thread; the 10 cross-wave stages go through np2 * 8 bytes of LDS.
__global__ void __launch_bounds__(1024)
tile_sort_big_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ order, const uint64_t* __restrict__ keys,
                     uint32_t* __restrict__ point_list)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* s = reinterpret_cast<uint64_t*>(smem_raw);
    const uint2 rg = ranges[order[blockIdx.x]];
    const uint32_t n = rg.y - rg.x;
    if (n <= 2048u || n > 16384u) return;
    const uint64_t* gk = keys + rg.x;
    uint32_t* out = point_list + rg.x;
    if (n <= 4096u) sort_tile_in_registers<4>(s, gk, out, n, 4096u);
    else if (n <= 8192u) sort_tile_in_registers<8>(s, gk, out, n, 8192u);
    else sort_tile_in_registers<16>(s, gk, out, n, 16384u);
}

// ---- lists of a view that blends its long lists in parts (gsr_blend_fwd.hip; scatter_kernel has listed the parts): a merge
// sort over memory instead of one padded bitonic network per list -- the network of the longest list's padded size ran for
// EVERY list above 2 048 entries (config D: 155 us for 16 384 padded keys), and above 16 384 keys it fell back to global
// memory.  Runs of 2 048 keys are sorted in LDS (the forward's merge sort) by the workgroup of every fourth part; then
// ceil(log2(n / 2 048)) passes, one launch each, merge neighbouring runs: a part's workgroup produces ITS 512 output
// positions, every thread finding where its two outputs start in the two input runs by a binary search along the merge
// path (keys are unique).  The passes ping-pong between two buffers; the last one leaves the ids in point_list.
__global__ void __launch_bounds__(256)
long_sort_runs_kernel(const uint2* __restrict__ part_list, const uint32_t* __restrict__ totals, const uint2* __restrict__ ranges,
                      const uint64_t* __restrict__ keys, uint64_t* __restrict__ dst, uint32_t* __restrict__ ids_out, uint32_t self_sort_n)
{
    if (blockIdx.x >= totals[6]) return;
    const uint2 part = part_list[blockIdx.x];
    if ((part.y & (SORT_SMALL_CAP - 1u)) != 0u) return;
    const uint2 rg = ranges[part.x];
    const uint32_t n = rg.y - rg.x, m = min(SORT_SMALL_CAP, n - part.y);
    if (ids_out == nullptr && n <= self_sort_n) return;   // (its parts sort it themselves inside the forward blend's launch)
    __shared__ __attribute__((aligned(16))) uint64_t s[2 * SORT_SMALL_CAP];
    const uint64_t* gk = keys + rg.x + part.y;
    uint64_t* out = dst + rg.x + part.y;
    if (ids_out != nullptr) {   // (no merge pass follows: no list of the view exceeds one run -- the ids go straight to the list)
        sort_small_tile(s, gk, ids_out + rg.x + part.y, m);
        return;
    }
    if (m == 1u) { if (threadIdx.x == 0) out[0] = gk[0]; return; }
    if (m <= 512u) sort_tile_merge<2>(s, gk, nullptr, m, out);
    else if (m <= 1024u) sort_tile_merge<4>(s, gk, nullptr, m, out);
    else sort_tile_merge<8>(s, gk, nullptr, m, out);
}

__global__ void __launch_bounds__(256)
long_sort_merge_kernel(const uint2* __restrict__ part_list, const uint32_t* __restrict__ totals, const uint2* __restrict__ ranges,
                       const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, uint32_t* __restrict__ ids_out, uint32_t len,
                       uint32_t self_sort_n)
{
    if (blockIdx.x >= totals[6]) return;
    const uint2 part = part_list[blockIdx.x];
    const uint2 rg = ranges[part.x];
    const uint32_t n = rg.y - rg.x;
    if (n <= self_sort_n) return;
    const uint32_t o0 = part.y + 2u * threadIdx.x;   // a part is FWD_CHUNK = 512 positions: two outputs per thread
    static_assert(FWD_CHUNK == 512, "two outputs per thread of a 256-thread workgroup");
    if (o0 >= n) return;
    const uint32_t base = o0 & ~(2u * len - 1u);
    const uint32_t la = min(len, n - base);
    const uint32_t lb = base + len < n ? min(len, n - base - len) : 0u;
    const uint64_t* a = src + rg.x + base;
    const uint64_t* b = a + len;
    const uint32_t d = o0 - base;
    uint32_t lo = d > lb ? d - lb : 0u, hi = min(d, la);
    while (lo < hi) {   // merge path: how many of the first d outputs come from a
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < b[d - 1u - mid]) lo = mid + 1u; else hi = mid;
    }
    uint32_t ai = lo, bi = d - lo;
    uint64_t ka = ai < la ? a[ai] : ~0ull, kb = bi < lb ? b[bi] : ~0ull;
    uint64_t res[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const bool take_a = bi >= lb || (ai < la && ka < kb);
        res[e] = take_a ? ka : kb;
        if (take_a) { ai++; ka = ai < la ? a[ai] : ~0ull; }
        else { bi++; kb = bi < lb ? b[bi] : ~0ull; }
    }
#pragma unroll
    for (int e = 0; e < 2; e++) {
        if (o0 + (uint32_t)e < n) {
            dst[rg.x + o0 + e] = res[e];
            if (ids_out != nullptr) ids_out[rg.x + o0 + e] = (uint32_t)res[e];
        }
    }
}

bool tile_sort_launches(int R, uint32_t max_count, bool blend_sorts_small)
{
    if (R <= 0) return false;
    if (getenv("GSR_DEBUG_SORT_CAP") || getenv("GSR_SORT_LDS") || !blend_sorts_small) return true;
    // (a view that splits: the parts of lists up to 2 048 entries sort themselves, longer lists go through the merge passes)
    return split_threshold(max_count, (uint32_t)R) != 0xffffffffu ? max_count > SORT_SMALL_CAP : max_count > 2048u;
}

bool launch_tile_sort(int W, int H, int R, int U, uint32_t max_count, ImageState im, BinState b, bool blend_sorts_small, hipStream_t st)
{
    const Tiles t = tiles_of(W, H);
    bool left_small = false;
    // GSR_DEBUG_SORT_CAP=64 forces the global-memory fallback for every tile above 64 instances (tests).
    static const char* dbg = getenv("GSR_DEBUG_SORT_CAP");
    if (dbg && dbg[0] == '6') {
        tile_sort_kernel<64, 0, true><<<t.T, 256, 64 * 8, st>>>(im.ranges, im.order, b.keys, b.point_list);
        return false;
    }
    static const bool lds_sort = getenv("GSR_SORT_LDS") != nullptr;   // A/B switch: the LDS network for every tile
    if (lds_sort) tile_sort_kernel<2048, 0, false><<<t.T, 256, 2048 * 8, st>>>(im.ranges, im.order, b.keys, b.point_list);
    // the forward blend sorts each of these lists right before walking it
    else if (blend_sorts_small) left_small = true;
    else tile_sort_reg_kernel<<<t.T, 256, 0, st>>>(im.ranges, im.order, b.keys, b.point_list);
    const bool in_parts = split_threshold(max_count, (uint32_t)(R > 0 ? R : 0)) != 0xffffffffu && !lds_sort;
    // A view that splits: if no list exceeds 2 048 entries the parts sort their lists themselves inside the blend's launch
    // (blend_sorts_small; gsr_blend_fwd.hip) and nothing is launched here; otherwise every split list is sorted here, by runs of
    // 2 048 keys and merge passes over memory.
    const uint32_t self_sort_n = blend_sorts_small && max_count <= SORT_SMALL_CAP ? SORT_SMALL_CAP : 0u;
    if (in_parts && max_count > self_sort_n) {
        uint64_t* bufs[2] = {reinterpret_cast<uint64_t*>(b.rec_a), reinterpret_cast<uint64_t*>(b.rec_a) + (size_t)R};   // (free until the forward)
        const unsigned grid = (unsigned)part_capacity(R, U);
        long_sort_runs_kernel<<<grid, 256, 0, st>>>(b.part_list, im.totals, im.ranges, b.keys, bufs[0],
                                                    max_count <= SORT_SMALL_CAP ? b.point_list : nullptr, self_sort_n);
        int cur = 0;
        for (uint32_t len = SORT_SMALL_CAP; len < max_count; len <<= 1) {
            const bool last = (len << 1) >= max_count || (len << 1) == 0u;
            long_sort_merge_kernel<<<grid, 256, 0, st>>>(b.part_list, im.totals, im.ranges, bufs[cur], bufs[cur ^ 1],
                                                         last ? b.point_list : nullptr, len, self_sort_n);
            cur ^= 1;
            if (last) break;
        }
    }
    if (max_count > 2048 && !lds_sort && !in_parts) {
        // the longest lists sit at the front of `order` (32-entry length classes, snake within bands of 256): every tile
        // above 2 048 entries is within the first few bands, but the kernel checks each tile's length itself anyway
        static bool attr_big = false;
        if (!attr_big) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_sort_big_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
            attr_big = true;
        }
        const uint32_t cap = max_count > 16384u ? 16384u : max_count;
        uint32_t np2 = 4096; while (np2 < cap) np2 <<= 1;
        tile_sort_big_kernel<<<front_of_order(R, t.T), 1024, (size_t)np2 * 8, st>>>(im.ranges, im.order, b.keys, b.point_list);
    }
    if ((max_count > 16384 && !in_parts) || (lds_sort && max_count > 2048)) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_sort_kernel<16384, 2048, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
            attr_set = true;
        }
        if (lds_sort)
            tile_sort_kernel<16384, 2048, true><<<t.T, 256, 16384 * 8, st>>>(im.ranges, im.order, b.keys, b.point_list);
        else   // only the lists the register kernels do not take: the global-memory network
            tile_sort_kernel<16384, 16384, true><<<t.T, 256, 16384 * 8, st>>>(im.ranges, im.order, b.keys, b.point_list);
    }
    return left_small;
}

}  // namespace gsr
