// gsr_mask.hip -- per-pixel CANDIDATE MASKS of a tile's depth-sorted list, and the per-tile depth sort in front of them.
//
// Why.  The reference's renderCUDA (DGR/cuda_rasterizer/forward.cu:261-374, backward.cu:399-557) feeds every pixel of a
// 16x16 tile every instance of the tile's list.  GauSTAR's surface splats are ~3.6 px in radius: of the (instance, pixel)
// pairs of a tile only ~4 % pass the alpha >= 1/255 test, and even a walk that culls instances per 8x8 block exactly
// keeps 10 of its 64 lanes busy.  Lock-step walks (all pixels of a wave look at the same instance) therefore spend
// their issue slots on dead lanes -- round 1's blend kernels were bound by exactly that (58 M + 42 M vector
// instructions per 1080p view for 12 M live pairs).
//
// What.  For every (tile, 64-entry segment of its list) = UNIT and every pixel of the tile this file produces a 64-bit
// word whose bit i says "instance 64*s + i of the list MAY reach alpha >= 1/255 at this pixel" -- a conservative
// superset, the blend kernels still apply the reference's exact tests to every candidate.  With the words in hand a
// pixel walks ITS OWN candidates (v_ffbl over its word) instead of the tile's list: a wave's trip count becomes the
// largest per-pixel candidate count of its 64 pixels (~36 per 8x8 block on config C) instead of the number of
// instances that touch the block (~160).
//
// How.  Lane = instance.  The alpha >= 1/255 region of a splat is the ellipse  f(X, Y) = 0.5 (a X^2 + c Y^2) + b X Y <= tau
// (tau = ln(255 opacity) + margin, stored by preprocess); per pixel ROW it is an interval in x with closed-form ends,
// so sixteen interval solves (one v_sqrt each) give the instance's 16 x 16-bit row masks for the whole tile; the four
// 8x8 blocks' 64-bit masks are byte selections of those (v_perm_b32); and a 64x64 BIT-MATRIX TRANSPOSE across the wave
// (v_permlane32_swap, byte permutes, nibble/pair/bit swaps with lane ^ s) turns "instance-major" into "pixel-major":
// 27 vector instructions per block instead of 64 ballots.  ~500 instructions per unit, all 64 lanes busy.
//
// Layout: masks[(unit * 4 + block) * 64 + lane] = uint2 {positions 0-31, positions 32-63} of the unit, block = 2*by + bx,
// lane = 8*(y - block_y0) + (x - block_x0).  The forward blend overwrites each word it consumes with the bits it
// actually blended (per half bit-reversed, see gsr_blend_fwd.hip); the backward walks only those.
#include "gsr_internal.h"
#include "gsr_sort.h"

namespace gsr {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Per-lane constants of the transpose (functions of the lane id only).
struct TransposeConsts {
    uint32_t sel16, sel8, m4, m2, m1, sh4, sh2, sh1;
    __device__ __forceinline__ explicit TransposeConsts(int lane)
    {
        sel16 = (lane & 16) ? 0x03020706u : 0x05040100u;   // v_perm selectors: bytes 0-3 = own dword, 4-7 = partner's
        sel8 = (lane & 8) ? 0x03070105u : 0x06020400u;
        m4 = (lane & 4) ? 0xf0f0f0f0u : 0x0f0f0f0fu; sh4 = (lane & 4) ? 4u : 28u;
        m2 = (lane & 2) ? 0xccccccccu : 0x33333333u; sh2 = (lane & 2) ? 2u : 30u;
        m1 = (lane & 1) ? 0xaaaaaaaau : 0x55555555u; sh1 = (lane & 1) ? 1u : 31u;
    }
};

__device__ __forceinline__ uint32_t rotr32(uint32_t v, uint32_t s) { return __builtin_amdgcn_alignbit(v, v, s); }

// 64x64 bit-matrix transpose across a wave64.  In: lane i holds row i, bit c of (hi:lo) = element (i, c).
// Out: lane c holds column c, bit i = element (i, c).  Recursive block swap, strides 32 .. 1: at stride s the lanes i and
// i ^ s exchange the s x s off-diagonal blocks -- the lane with bit s clear keeps its columns with bit s clear and takes
// the partner's same columns as its columns with bit s set, and vice versa.
__device__ __forceinline__ void transpose64(uint32_t& lo, uint32_t& hi, const TransposeConsts& k)
{
    {   // s = 32: whole dwords.  lanes 0-31: hi <- partner's lo; lanes 32-63: lo <- partner's hi.  One half exchange.
        const u32x2 r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        lo = r[0]; hi = r[1];
    }
    uint32_t x[2] = {lo, hi};
#pragma unroll
    for (int d = 0; d < 2; d++) {
        uint32_t v = x[d];
        v = __builtin_amdgcn_perm(lane_xor_u32<16>(v), v, k.sel16);              // s = 16: half-words
        v = __builtin_amdgcn_perm(lane_xor_u32<8>(v), v, k.sel8);                // s = 8: bytes
        { const uint32_t p = rotr32(lane_xor_u32<4>(v), k.sh4); v = (k.m4 & v) | (~k.m4 & p); }   // nibbles
        { const uint32_t p = rotr32(lane_xor_u32<2>(v), k.sh2); v = (k.m2 & v) | (~k.m2 & p); }   // bit pairs
        { const uint32_t p = rotr32(lane_xor_u32<1>(v), k.sh1); v = (k.m1 & v) | (~k.m1 & p); }   // bits
        x[d] = v;
    }
    lo = x[0]; hi = x[1];
}

// Candidate masks of one unit (64 consecutive list positions of one tile); called by one whole wave.
__device__ __forceinline__ void unit_masks(int lane, uint32_t k, uint32_t n, const uint32_t* __restrict__ list,
                                           const float4* __restrict__ g0, const float4* __restrict__ g1, int tile_x0,
                                           int tile_y0, const TransposeConsts& tc, uint2* __restrict__ out)
{
    // ---- lane = instance k of the list
    float4 a = make_float4(0.f, 0.f, 1.f, 0.f), b = make_float4(1.f, 0.f, -1.f, 0.f);
    if (k < n) { const uint32_t gid = list[k]; a = g0[gid]; b = g1[gid]; }
    const float ca = a.z, cb = a.w, cc = b.x, tau = b.z;
    // f <= tau along the pixel row Y (relative to the splat):  X in [(-bY - sqrt D) / a, (-bY + sqrt D) / a],
    // D = 2 a tau - (a c - b^2) Y^2.  Every rounding is pushed outwards: D is inflated by 2^-12 of its largest term (the
    // subtraction can cancel), the determinant deflated by 2^-20 of its first term, the interval widened by eps = 64 ulps
    // of the largest magnitude entering its centre; tau itself already carries the margin preprocess gave it.
    const float inva = __builtin_amdgcn_rcpf(ca);
    const float boa = cb * inva;
    float c0 = 2.0f * ca * tau * 1.000244140625f;
    float mdet = cb * cb - (ca * cc) * 0.99999904632568359375f;
    const float cxl = a.x - (float)tile_x0, Y0 = (float)tile_y0 - a.y;
    float eps = 7.62939453125e-6f * (fabsf(cxl) + fabsf(boa) * (fabsf(Y0) + 16.0f)) + 0.0009765625f;
    const bool visible = tau >= 0.0f;
    // anything this arithmetic cannot bound (non-finite or non-positive conic entries, astronomically distant centres)
    // becomes a full mask: the blend's exact test sorts it out
    const bool tame = inva > 0.0f && inva < 1e30f && c0 < 1e30f && fabsf(mdet) < 1e30f && eps < 1e6f;
    if (!visible) { c0 = -1.0f; mdet = 0.0f; eps = -1.0f; }   // D < 0 and a negative half width: every row empty
    uint32_t pk[8];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const float Y = Y0 + (float)r;
        const float D = __builtin_fmaf(Y * Y, mdet, c0);
        const float sD = __builtin_amdgcn_sqrtf(fmaxf(D, 0.0f));
        const float half = __builtin_fmaf(sD, inva, eps);
        const float mid = __builtin_fmaf(-boa, Y, cxl);
        const float lo_f = fminf(fmaxf(mid - half, 0.0f), 16.0f), hi_f = fminf(fmaxf(mid + half, -1.0f), 15.0f);
        const int li = (int)ceilf(lo_f), hi_i = (int)floorf(hi_f);
        const int wd = max(hi_i - li + 1, 0);
        uint32_t rm = ((1u << wd) - 1u) << li;                    // v_bfm_b32
        rm = (visible && !tame) ? 0xffffu : rm;
        if (r & 1) pk[r >> 1] |= rm << 16; else pk[r >> 1] = rm;
    }
    // ---- the four 8x8 blocks: byte bx of rows 8 by .. 8 by + 7, then instance-major -> pixel-major
#pragma unroll
    for (int blk = 0; blk < 4; blk++) {
        const uint32_t sel = (blk & 1) ? 0x07050301u : 0x06040200u;
        const int p0 = (blk >> 1) * 4;
        uint32_t lo = __builtin_amdgcn_perm(pk[p0 + 1], pk[p0], sel), hi = __builtin_amdgcn_perm(pk[p0 + 3], pk[p0 + 2], sel);
        transpose64(lo, hi, tc);
        out[blk * 64 + lane] = make_uint2(lo, hi);
    }
}

// One workgroup per tile (launch order = `order`, longest lists first): depth-sort the tile's bucket (lists of up to
// 2 048 entries; longer ones were sorted by tile_sort_big_kernel / tile_sort_kernel before this launch and get their masks
// from tile_mask_kernel), then the four waves share the tile's units.  Fused because the sort alone is latency-bound
// (key loads, cross-lane exchanges, barriers) while the mask arithmetic is pure vector ALU work: resident tiles overlap.
__global__ void __launch_bounds__(256)
tile_sort_mask_kernel(int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ order,
                      const uint32_t* __restrict__ seg_off, const uint64_t* __restrict__ keys,
                      uint32_t* __restrict__ point_list, const float4* __restrict__ g0, const float4* __restrict__ g1,
                      uint2* __restrict__ masks, int sort_here)
{
    __shared__ uint64_t s[2048];
    const int tile = (int)order[blockIdx.x];
    const uint2 rg = ranges[tile];
    const uint32_t n = rg.y - rg.x;
    if (n == 0 || n > 2048u) return;
    uint32_t* list = point_list + rg.x;
    if (sort_here) {
        sort_small_tile(s, keys + rg.x, list, n);
        __syncthreads();   // the sorted ids are visible to the four waves
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const TransposeConsts tc(lane);
    const uint32_t unit0 = seg_off[tile], n_units = (n + 63u) >> 6;
    const int tx = tile % gx, ty = tile / gx;
    for (uint32_t u = (uint32_t)wave; u < n_units; u += 4u)
        unit_masks(lane, u * 64u + (uint32_t)lane, n, list, g0, g1, tx * TILE, ty * TILE, tc,
                   masks + (size_t)(unit0 + u) * 256);
}

// Unit-parallel variant for the tiles above 2 048 entries (close-up views): one wave per unit, so that a 12 000-entry
// tile is 188 independent waves instead of one workgroup's serial loop.
__global__ void __launch_bounds__(64)
tile_mask_kernel(int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ seg_off,
                 const uint32_t* __restrict__ unit_tile, const uint32_t* __restrict__ point_list,
                 const float4* __restrict__ g0, const float4* __restrict__ g1, uint2* __restrict__ masks)
{
    const uint32_t unit = blockIdx.x;
    const int tile = (int)unit_tile[unit];
    const uint2 rg = ranges[tile];
    const uint32_t n = rg.y - rg.x;
    if (n <= 2048u) return;
    const int lane = threadIdx.x;
    const TransposeConsts tc(lane);
    const int tx = tile % gx, ty = tile / gx;
    unit_masks(lane, (unit - seg_off[tile]) * 64u + (uint32_t)lane, n, point_list + rg.x, g0, g1, tx * TILE, ty * TILE, tc,
               masks + (size_t)unit * 256);
}

void launch_tile_masks(int W, int H, int U, uint32_t max_count, bool sort_here, GeomState g, ImageState im, BinState b, hipStream_t st)
{
    const Tiles t = tiles_of(W, H);
    if (U <= 0) return;
    tile_sort_mask_kernel<<<t.T, 256, 0, st>>>(t.gx, im.ranges, im.order, im.seg_off, b.keys, b.point_list, g.g0, g.g1,
                                               b.masks, sort_here ? 1 : 0);
    if (max_count > 2048u)
        tile_mask_kernel<<<U, 64, 0, st>>>(t.gx, im.ranges, im.seg_off, b.unit_tile, b.point_list, g.g0, g.g1, b.masks);
}

}  // namespace gsr
