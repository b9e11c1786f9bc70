// gsr_mask.h -- per-pixel CANDIDATE MASKS of a tile's depth-sorted list (device code shared with gsr_blend_fwd.hip).
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
// The forward blend calls unit_masks() on the records of a chunk while it parks them in LDS (lane = instance, one unit
// per wave and fetch round), consumes the words from LDS and -- when a backward pass may follow -- also writes them to
// global memory, masks[(unit * 4 + block) * 64 + lane] = uint2 {positions 0-31, positions 32-63} with block = 2*by + bx and
// lane = 8*(y - block_y0) + (x - block_x0): the backward blend derives from them which instances of a unit any pixel of
// a block replays, and in which later unit a pixel's snapshot sits.
#pragma once
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

// Candidate masks of one unit (64 consecutive list positions of one tile); called by one whole wave, lane = instance
// with its geometry records a = {x, y, conic a, conic b}, b = {conic c, opacity, tau, -} (tau < 0: no instance here).
// out(block, lo, hi): lane = pixel of the block, (hi:lo) = its 64-bit word over the unit's positions.
template <class Out>
__device__ __forceinline__ void unit_masks(float4 a, float4 b, int tile_x0, int tile_y0, const TransposeConsts& tc, Out out)
{
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
        // floor + convert in one instruction each (ceil(x) = -floor(-x)), and the row mask in one v_bfm_b32: the compiler
        // spells `((1 << wd) - 1) << li` as shift, not, shift and floor / ceil / two conversions as four (-3 per row)
        int nli, hi_i;
        asm("v_cvt_flr_i32_f32_e64 %0, -%1" : "=v"(nli) : "v"(lo_f));   // = -ceil(lo_f)
        asm("v_cvt_flr_i32_f32_e32 %0, %1" : "=v"(hi_i) : "v"(hi_f));
        const int wd = max(hi_i + nli + 1, 0);                            // hi - li + 1 in [-16, 16]
        uint32_t rm;
        asm("v_bfm_b32 %0, %1, %2" : "=v"(rm) : "v"(wd), "v"(-nli));     // ((1 << wd) - 1) << li
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
        out(blk, lo, hi);
    }
}

}  // namespace gsr
