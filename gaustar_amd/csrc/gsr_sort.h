// gsr_sort.h -- register-resident bitonic sort of one tile's 64-bit keys (depth bits << 32 | Gaussian id), shared by
// tile_sort_reg_kernel / tile_sort_big_kernel (gsr_binning.hip) and the forward blend, which can sort a tile's list
// itself right before walking it (gsr_blend_fwd.hip).  Replaces the per-tile share of the reference's device-wide
// cub::DeviceRadixSort::SortPairs (DGR/cuda_rasterizer/rasterizer_impl.cu:301-307).
#pragma once
#include "gsr_internal.h"

namespace gsr {

// ---- register-resident bitonic sort (tiles of 2 049 .. 16 384 keys: tile_sort_big_kernel; its cross-lane stage also sorts
// the 64-key runs of the merge sort below).
// The LDS network above moves 32 bytes through LDS per compare-exchange; with eight workgroups per CU that
// traffic, not the comparisons, bounded the kernel.  Here every thread keeps E consecutive keys in registers
// (blocked layout, 256 threads, E = 2 / 4 / 8 for 512 / 1 024 / 2 048 padded keys):
//   stride <  E        compare-exchange between two registers of the thread;
//   stride <  64 E     partner key comes from lane ^ (stride / E): DPP quad permutes (1, 2), ds_swizzle (4, 8, 16),
//                      ds_bpermute (32) -- crossbar only, no LDS storage, no barrier;
//   stride >= 64 E     the only stages that cross waves: one round trip through LDS (at most 3 of the 66 stages).
template <int D>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t v)
{
    if constexpr (D == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);         // quad_perm [1,0,3,2]
    else if constexpr (D == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    else if constexpr (D == 4) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);               // xor 4 (bit-mask mode)
    else if constexpr (D == 8) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x201F);
    else if constexpr (D == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);
    else return (uint32_t)__shfl_xor((int)v, 32, 64);
}
// one cross-lane stage: every key meets the key of lane ^ D held in the same register slot
template <int E, int D>
__device__ __forceinline__ void cross_lane_stage(uint64_t (&v)[E], uint32_t e0, uint32_t k, int lane)
{
    const bool lower = (lane & D) == 0;
#pragma unroll
    for (int r = 0; r < E; r++) {
        const uint64_t pv = ((uint64_t)lane_xor_u32<D>((uint32_t)(v[r] >> 32)) << 32) | lane_xor_u32<D>((uint32_t)v[r]);
        const bool asc = ((e0 + r) & k) == 0;
        const bool take_min = lower == asc;
        const bool gt = v[r] > pv;
        v[r] = (gt == take_min) ? pv : v[r];
    }
}

template <int E>
__device__ __forceinline__ void sort_tile_in_registers(uint64_t* __restrict__ s, const uint64_t* __restrict__ gk,
                                                       uint32_t* __restrict__ out, uint32_t n, uint32_t np2)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t e0 = (uint32_t)tid * E;
    // waves that own no key leave; cross-wave stages (and their barriers) exist only if np2 > 64 E, in which case
    // every wave below np2 / (64 E) stays -- and np2 = 256 E means all four
    if (e0 >= np2) {
        if (np2 > 64u * E) {   // keep the barrier count of the active waves (uniform per wave)
            for (uint32_t k = 128u * E; k <= np2; k <<= 1)
                for (uint32_t j = k >> 1; j >= 64u * E; j >>= 1) { __syncthreads(); __syncthreads(); }
        }
        return;
    }
    uint64_t v[E];
#pragma unroll
    for (int r = 0; r < E; r++) v[r] = e0 + r < n ? gk[e0 + r] : ~0ull;
    for (uint32_t k = 2; k <= np2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (j < (uint32_t)E) {
#pragma unroll
                for (int jj = 1; jj < E; jj <<= 1) {
                    if ((uint32_t)jj != j) continue;
#pragma unroll
                    for (int r = 0; r < E; r++) {
                        if (r & jj) continue;
                        const bool asc = ((e0 + r) & k) == 0;
                        const uint64_t a = v[r], b = v[r | jj];
                        const bool gt = a > b;
                        const uint64_t mn = gt ? b : a, mx = gt ? a : b;
                        v[r] = asc ? mn : mx;
                        v[r | jj] = asc ? mx : mn;
                    }
                }
            } else if (j < 64u * E) {
                switch (j / E) {   // one uniform branch per stage; the exchange pattern is an immediate
                    case 1: cross_lane_stage<E, 1>(v, e0, k, lane); break;
                    case 2: cross_lane_stage<E, 2>(v, e0, k, lane); break;
                    case 4: cross_lane_stage<E, 4>(v, e0, k, lane); break;
                    case 8: cross_lane_stage<E, 8>(v, e0, k, lane); break;
                    case 16: cross_lane_stage<E, 16>(v, e0, k, lane); break;
                    default: cross_lane_stage<E, 32>(v, e0, k, lane); break;
                }
            } else {
#pragma unroll
                for (int r = 0; r < E; r++) s[e0 + r] = v[r];
                __syncthreads();
                const bool lower = (e0 & j) == 0;
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const uint64_t pv = s[(e0 + r) ^ j];
                    const bool asc = ((e0 + r) & k) == 0;
                    const bool take_min = lower == asc;
                    const bool gt = v[r] > pv;
                    v[r] = (gt == take_min) ? pv : v[r];
                }
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int r = 0; r < E; r++)
        if (e0 + r < n) out[e0 + r] = (uint32_t)v[r];
}

// ---- merge sort for buckets of up to 2 048 keys (every tile of a typical view), 256 threads.
// The register network above costs O(log^2 n) compare-exchanges PER KEY (55 stages for 1 024 padded keys, 66 for 2 048) and
// pads n to a power of two; measured inside the forward blend it was the largest single consumer of issue slots (10.6 M vector
// + 10.5 M scalar instructions per 1080p view, 34 us on the critical path of a 1 300-entry tile).  Here:
//   1. every wave sorts runs of 64 keys in registers (one key per lane, the 21 cross-lane stages of the network above);
//   2. log2(n / 64) merge levels through LDS (two buffers, ping-pong): a thread owns E consecutive OUTPUT positions of its
//      pair of runs, finds where they start in the two inputs with one binary search along the merge path (keys are
//      unique: depth bits << 32 | Gaussian id), and merges E keys sequentially.  ~135 instructions per thread and level.
// No padding beyond the last run: runs are clipped to n.
constexpr uint32_t SORT_SMALL_CAP = 2048;                       // keys; the caller provides 2 * SORT_SMALL_CAP * 8 bytes of LDS

template <uint32_t K, uint32_t J>
__device__ __forceinline__ void run64_stages(uint64_t (&v)[1], int lane)
{
    cross_lane_stage<1, (int)J>(v, (uint32_t)lane, K, lane);
    if constexpr (J > 1) run64_stages<K, J / 2>(v, lane);
    else if constexpr (K < 64) run64_stages<K * 2, K>(v, lane);
}

// (out: the sorted Gaussian ids; out_keys, if given instead: the sorted keys themselves -- runs of a longer list)
struct NoStamp { __device__ __forceinline__ void operator()() const {} };
// -> where the sorted keys lie in LDS when the function returns (valid until the caller reuses the bytes)
template <int E, class STAMP = NoStamp>
__device__ __forceinline__ const uint64_t* sort_tile_merge(uint64_t* __restrict__ s, const uint64_t* __restrict__ gk,
                                                uint32_t* __restrict__ out, uint32_t n, uint64_t* __restrict__ out_keys = nullptr,
                                                STAMP stamp = STAMP())
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint64_t* A = s;
    uint64_t* B = s + SORT_SMALL_CAP;
    const uint32_t n_runs = (n + 63u) >> 6;
    // (a wave's up to E runs: all keys are requested before the first run is sorted -- one trip to memory, not one per run)
    uint64_t kv[E];
#pragma unroll
    for (int q = 0; q < E; q++) {
        const uint32_t i = (wave + 4u * (uint32_t)q) * 64u + lane;
        kv[q] = i < n ? gk[i] : ~0ull;
    }
#pragma unroll
    for (int q = 0; q < E; q++) {
        const uint32_t r = wave + 4u * (uint32_t)q;
        if (r < n_runs) {   // (wave-uniform)
            uint64_t v[1] = {kv[q]};
            run64_stages<2, 1>(v, (int)lane);
            A[r * 64u + lane] = v[0];
        }
    }
    __syncthreads();
    stamp();
    const uint32_t o0 = tid * (uint32_t)E;                       // this thread's output positions [o0, o0 + E) on every level
    // Round 5: the levels are built for few DEPENDENT trips to LDS -- under three other tiles' walks a dependent ds_read costs
    // ~400 cycles, and a 1 300-entry list spent 18.7 us here (a quarter of the forward's longest tile) on 5 levels x (11 binary
    // search steps + 8 sequential merge steps).  Now: a 4-ary search along the merge path (three probes = six independent reads
    // per step, log4 steps), then the next E keys of BOTH runs in one go (2 E independent reads) and the merge in registers:
    // min(a[i], b[E-1-i]) is a bitonic sequence holding the E smallest of the 2 E, three (log2 E) compare-exchange stages sort it.
    for (uint32_t len = 64u; len < n; len <<= 1) {
        uint64_t res[E];
        if (o0 < n) {
            const uint32_t base = o0 & ~(2u * len - 1u);        // E divides 2 len: the E outputs lie in one pair of runs
            const uint32_t la = min(len, n - base);
            const uint32_t lb = base + len < n ? min(len, n - base - len) : 0u;
            const uint64_t* a = A + base;
            const uint64_t* b = A + base + len;
            const uint32_t d = o0 - base;
            uint32_t lo = d > lb ? d - lb : 0u, hi = min(d, la);
#ifndef GSR_SORT_KARY
#define GSR_SORT_KARY 4
#endif
#if GSR_SORT_KARY == 4
            while (lo < hi) {                                   // merge path: how many of the first d outputs come from a
                const uint32_t span = hi - lo;
                const uint32_t m1 = lo + (span >> 2), m2 = lo + (span >> 1), m3 = lo + ((3u * span) >> 2);
                const uint64_t a1 = a[m1], b1 = b[d - 1u - m1], a2 = a[m2], b2 = b[d - 1u - m2], a3 = a[m3], b3 = b[d - 1u - m3];
                const bool p1 = a1 < b1, p2 = a2 < b2, p3 = a3 < b3;   // monotone: true ... true false ... false
                if (!p1) hi = m1;
                else if (!p2) { lo = m1 + 1u; hi = m2; }
                else if (!p3) { lo = m2 + 1u; hi = m3; }
                else lo = m3 + 1u;
            }
#else
            while (lo < hi) {                                   // merge path: how many of the first d outputs come from a
                const uint32_t mid = (lo + hi) >> 1;
                if (a[mid] < b[d - 1u - mid]) lo = mid + 1u; else hi = mid;
            }
#endif
            const uint32_t ai = lo, bi = d - lo;
            uint64_t ka[E], kb[E];
#pragma unroll
            for (int e = 0; e < E; e++) {
                ka[e] = ai + (uint32_t)e < la ? a[ai + e] : ~0ull;
                kb[e] = bi + (uint32_t)e < lb ? b[bi + e] : ~0ull;
            }
#pragma unroll
            for (int e = 0; e < E; e++) res[e] = ka[e] < kb[E - 1 - e] ? ka[e] : kb[E - 1 - e];
#pragma unroll
            for (int j = E / 2; j > 0; j >>= 1) {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    if ((e & j) == 0) {
                        const uint64_t x = res[e], y = res[e | j];
                        const bool gt = x > y;
                        res[e] = gt ? y : x;
                        res[e | j] = gt ? x : y;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < E; e++)
                if (o0 + (uint32_t)e < n) B[o0 + e] = res[e];
        }
        __syncthreads();                                        // level done: B complete, nobody reads A any more
        stamp();
        uint64_t* t = A; A = B; B = t;
    }
    if (out_keys != nullptr) { for (uint32_t i = tid; i < n; i += 256u) out_keys[i] = A[i]; }
    else { for (uint32_t i = tid; i < n; i += 256u) out[i] = (uint32_t)A[i]; }
    return A;
}

// One tile of 1 .. 2 048 keys, 256 threads, s = 2 * SORT_SMALL_CAP * 8 bytes of LDS.
// Must be called by all 256 threads of the workgroup (it contains workgroup barriers).
// -> the sorted keys in LDS (nullptr for a single key), see sort_tile_merge
template <class STAMP = NoStamp>
__device__ __forceinline__ const uint64_t* sort_small_tile(uint64_t* __restrict__ s, const uint64_t* __restrict__ gk,
                                                           uint32_t* __restrict__ out, uint32_t n, STAMP stamp = STAMP())
{
    if (n == 1) {
        if (threadIdx.x == 0) out[0] = (uint32_t)gk[0];
        return nullptr;
    }
    if (n <= 512u) return sort_tile_merge<2>(s, gk, out, n, nullptr, stamp);
    if (n <= 1024u) return sort_tile_merge<4>(s, gk, out, n, nullptr, stamp);
    return sort_tile_merge<8>(s, gk, out, n, nullptr, stamp);
}

}  // namespace gsr
