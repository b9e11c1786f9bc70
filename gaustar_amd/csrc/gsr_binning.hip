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
#include <cstdlib>

namespace gsr {

// ---- wave64 inclusive scan via DPP-free shuffles (log-step); T is tiny, this kernel is latency-bound.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// One workgroup scans all T tile counts: ranges[t] = {start, start+count}, cursor[t] = start,
// totals = {R, max count, #non-empty tiles}, and builds `order`: a counting sort of the tiles by
// descending list length in 32-entry buckets (empty tiles last).
constexpr int NBUCKET = 66;   // bucket 0 = longest (>= 2048 entries) ... bucket 64 = 1..32 entries, bucket 65 = empty
__device__ __forceinline__ int length_bucket(uint32_t c)
{
    return c == 0 ? 65 : 64 - (int)min(64u, (c + 31u) >> 5);
}
__global__ void __launch_bounds__(1024)
tile_scan_kernel(int T, const uint32_t* __restrict__ tile_count, uint2* __restrict__ ranges,
                 uint32_t* __restrict__ tile_cursor, uint32_t* __restrict__ totals, uint32_t* __restrict__ order)
{
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t wave_max[16];
    __shared__ uint32_t carry_s;
    __shared__ uint32_t bucket_n[NBUCKET];
    if (threadIdx.x < NBUCKET) bucket_n[threadIdx.x] = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    uint32_t vmax = 0;
    __syncthreads();
    for (int base = 0; base < T; base += 1024) {
        const int t = base + tid;
        const uint32_t c = t < T ? tile_count[t] : 0u;
        vmax = max(vmax, c);
        if (t < T) atomicAdd(&bucket_n[length_bucket(c)], 1u);
        const uint32_t incl = wave_incl_scan(c, lane);
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += wave_sum[w];
        const uint32_t carry = carry_s;
        const uint32_t start = carry + woff + incl - c;
        if (t < T) {
            ranges[t] = make_uint2(start, start + c);
            tile_cursor[t] = start;
        }
        __syncthreads();
        if (tid == 1023) carry_s = start + c;
        __syncthreads();
    }
    // max reduction
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) vmax = max(vmax, (uint32_t)__shfl_xor((int)vmax, d, 64));
    if (lane == 0) wave_max[wave] = vmax;
    __syncthreads();
    if (tid == 0) {
        uint32_t m = 0;
        for (int w = 0; w < 16; w++) m = max(m, wave_max[w]);
        totals[0] = carry_s;
        totals[1] = m;
        totals[2] = (uint32_t)T - bucket_n[NBUCKET - 1];
        totals[3] = 0;
        // exclusive prefix over the buckets -> start offsets (in place)
        uint32_t run = 0;
        for (int b = 0; b < NBUCKET; b++) { const uint32_t c = bucket_n[b]; bucket_n[b] = run; run += c; }
    }
    __syncthreads();
    for (int t = tid; t < T; t += 1024) order[atomicAdd(&bucket_n[length_bucket(tile_count[t])], 1u)] = (uint32_t)t;
}

void launch_tile_scan(ImageState im, int T, hipStream_t st)
{
    tile_scan_kernel<<<1, 1024, 0, st>>>(T, im.tile_count, im.ranges, im.tile_cursor, im.totals, im.order);
}

// One thread per Gaussian: claim a slot in every reachable tile's bucket and store the sort key.  Walks exactly
// the tiles preprocess counted (same stored inputs, same contraction-free test) with the same wave
// aggregation: the wave's leader for a tile reserves popcount(mask) slots with ONE returning atomic and
// every lane of the mask takes its own slot by prefix popcount.
__global__ void __launch_bounds__(256)
scatter_kernel(int P, int gx, const ushort4* __restrict__ rect, const float4* __restrict__ g0,
               const float4* __restrict__ g1, const float* __restrict__ depth, uint32_t* __restrict__ tile_cursor,
               uint64_t* __restrict__ keys)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    ushort4 r = make_ushort4(0, 0, 0, 0);
    float4 a = make_float4(0.f, 0.f, 1.f, 0.f), b = make_float4(1.f, 0.f, -1.f, 0.f);
    uint64_t key = 0;
    if (idx < P) {
        r = rect[idx];
        if (r.z > r.x && r.w > r.y) {
            a = g0[idx];
            b = g1[idx];
            key = ((uint64_t)__float_as_uint(depth[idx]) << 32) | (uint32_t)idx;
        }
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    for_each_tile_aggregated(r, a.x, a.y, a.z, a.w, b.x, b.z, gx, [&](int tile, unsigned long long m, int leader) {
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&tile_cursor[tile], (uint32_t)__popcll(m));
        base = __shfl(base, leader, 64);
        if ((m >> lane) & 1ull) keys[base + (uint32_t)__popcll(m & lt)] = key;
    });
}

void launch_scatter(int P, int W, int H, GeomState g, ImageState im, BinState b, hipStream_t st)
{
    const Tiles t = tiles_of(W, H);
    scatter_kernel<<<(P + 255) / 256, 256, 0, st>>>(P, t.gx, g.rect, g.g0, g.g1, g.depth, im.tile_cursor, b.keys);
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
        for (uint32_t k = 2; k <= np2; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = tid; i < (np2 >> 1); i += 256) {
                    // i-th compare-exchange of this stage: indices (lo, lo | j), lo has bit j clear.
                    const uint32_t lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                    const uint32_t hi = lo | j;
                    const bool asc = (lo & k) == 0;
                    const uint64_t a = s[lo], b = s[hi];
                    if ((a > b) == asc) { s[lo] = b; s[hi] = a; }
                }
                __syncthreads();
            }
        }
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

void launch_tile_sort(int W, int H, uint32_t max_count, ImageState im, BinState b, hipStream_t st)
{
    const Tiles t = tiles_of(W, H);
    // GSR_DEBUG_SORT_CAP=64 forces the global-memory fallback for every tile above 64 instances (tests).
    static const char* dbg = getenv("GSR_DEBUG_SORT_CAP");
    if (dbg && dbg[0] == '6') {
        tile_sort_kernel<64, 0, true><<<t.T, 256, 64 * 8, st>>>(im.ranges, im.order, b.keys, b.point_list);
        return;
    }
    tile_sort_kernel<2048, 0, false><<<t.T, 256, 2048 * 8, st>>>(im.ranges, im.order, b.keys, b.point_list);
    if (max_count > 2048) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_sort_kernel<16384, 2048, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
            attr_set = true;
        }
        tile_sort_kernel<16384, 2048, true><<<t.T, 256, 16384 * 8, st>>>(im.ranges, im.order, b.keys, b.point_list);
    }
}

}  // namespace gsr
