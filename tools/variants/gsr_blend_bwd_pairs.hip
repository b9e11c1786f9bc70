// gsr_blend_bwd_pairs.hip -- EXPERIMENT (round 5, VERDICT r4 task 1): the backward blend with lane = live (pixel, instance)
// PAIR instead of lane = pixel.  Built only by
//     python -m gaustar_amd.build --variant pairs --with tools/variants/gsr_blend_bwd_pairs.hip
// (defines launch_blend_bwd_variant; three channels, everything else falls back to the product's uniform pair loop).
//
// Per-pair arithmetic: the reference's renderCUDA backward, DGR/cuda_rasterizer/backward.cu:464-556, exactly as
// gsr_blend_bwd.hip restates it (same unit = (tile, 64-entry segment, 8x8 block) per wave, same head, snapshots, candidate
// words, kept set, moment contraction on the bf16 matrix pipe, re-centring, row-major flush).  What changes is who evaluates a
// pair and how the two per-pixel recurrences run:
//
//   * the unit's kept instances are worked off Q at a time (a CHUNK, deepest first).  For a chunk every pixel lane counts its
//     candidate bits among the chunk's instances (popcount), a wave scan turns the counts into offsets, and the lane writes its
//     (instance lane, pixel, rank) entries -- deepest first -- into a flat LDS list: the chunk's PAIR LIST, pixel-major;
//   * the list is evaluated 64 pairs at a time (a ROW): lane i gathers the record of its instance and the state of its pixel
//     from LDS and evaluates alpha ONCE per pair -- every lane of a row has a pair (the uniform loop has 10 of 64);
//   * T_front = T_start / prod (1 - alpha) and the projected accum_rec recurrence A' = (1 - alpha) A + alpha k (k = c . dL_dpix;
//     gsr_blend_bwd.hip "GSR_BWD_PROJ") are both products of the affine maps A -> a A + b, (a, b) = (1 - alpha, alpha k),
//     along the pixel: ONE segmented inclusive scan of (a, b) over the row, keyed by the pair's rank within its pixel (a lane
//     combines with lane i - d iff rank >= d: no key traffic), six DPP steps (row_shr 1, 2, 4, 8, row_bcast 15, 31), one carried
//     lane between rows.  A pair that fails the alpha test is the identity map (a, b) = (1, 0);
//   * w = alpha T and r = G dL_dalpha are split into exact bf16 (hi, mid, lo) AT THE WRITER -- one split per live pair
//     instead of one per (instance, pixel) cell -- and scattered into a dense [instance][pixel] table of three bf16 planes; the
//     contraction reads its A operand straight from the planes (no vector work per cell).
// Rounding: T and A come out of re-associated products (tree order instead of back-to-front), everything else is the uniform
// loop's arithmetic; the alpha decisions (power <= 0, alpha >= 1/255) are bit-identical to the forward's.
#include "gsr_bwd_util.h"

namespace gsr {

#ifndef GSR_PAIRS_Q
#define GSR_PAIRS_Q 16
#endif
#ifndef GSR_PAIRS_RMAX
#define GSR_PAIRS_RMAX 3
#endif
#ifndef GSR_PAIRS_WAVES
#define GSR_PAIRS_WAVES 4   // (120 registers at three rows per group; held at 96 = five waves it spills and loses 100 us)
#endif

namespace pairs {
constexpr int Q = GSR_PAIRS_Q;            // kept instances per chunk (a pixel has at most Q pairs per chunk: rank fits four bits)
static_assert(Q == 8 || Q == 16, "chunk: whole MFMA groups, rank within a pixel in four bits");
constexpr int ROWB = 144;                 // bytes per table row: 64 pixels of bf16 + 16 (sixteen rows 144 B apart cover all banks)
constexpr int W_ROWS = Q * ROWB + 128;    // w rows behind the r rows of the plane, shifted by 32 banks
constexpr int PLANE = W_ROWS + Q * ROWB;  // the bf16 plane: r rows [0, Q), w rows [Q, 2 Q); filled three times (hi, mid, lo passes)
constexpr int TBL = 0, TBL_BYTES = PLANE;
// While a row group is being EVALUATED the plane's bytes hold what only that phase reads: the chunk's pair list and the
// per-pixel constants.  The group's last gather has been issued before the plane is zeroed (LDS operations of a wave execute
// in order); a chunk with more rows than one group re-writes both for the next group.
constexpr int LIST = TBL;                 // pair list: u16 entries {instance lane : 6, pixel : 6, rank : 4}
constexpr int LIST_BYTES = (Q * 64 + 64 + 8) * 2;   // every pixel x every instance, + a row of slack for the look-ahead read
constexpr int SPIX = (LIST + LIST_BYTES + 15) & ~15;   // 64 x 32 B: {T_final bg.dL_dpix, x, y, -}, {dL_dpix 0..2, -}
static_assert(SPIX + 64 * 32 <= TBL + TBL_BYTES || Q == 8, "list + pixel constants must fit the plane");
constexpr int PHASE1_END = SPIX + 64 * 32;
constexpr int REC = (TBL + (TBL_BYTES > PHASE1_END ? TBL_BYTES : PHASE1_END) + 15) & ~15;
                                          // Q + 1 records of 48 B: {x, y, a', b'}, {c', opacity, c0, c1}, {c2, row offset, -, gaussian}
constexpr int REC_BYTES = (Q + 1) * 48;   //   (the last one: a null record for lanes outside the chunk)
constexpr int SLOT = REC + REC_BYTES;     // 64 bytes: instance lane -> record slot of the chunk (Q: none)
constexpr int DPIX = (SLOT + 64 + 15) & ~15;   // 64 x 8 B: a pixel's running {T, A} (updated from chunk to chunk)
constexpr int TOTAL = DPIX + 64 * 8;
static_assert(DPIX % 16 == 0 && REC % 16 == 0 && SPIX % 16 == 0, "16-byte accesses");
constexpr int SF = 12, MOM0 = 2, NM = 9;  // record floats, first moment float, moments per instance
constexpr int RMAX = GSR_PAIRS_RMAX;      // rows of a group: evaluated together, as independent instruction streams
}   // namespace pairs

// x = hi + r1's upper half + r2 exactly (three bf16 values); contraction off: `x` is a product at the call site, and fused into
// the subtraction r1 would carry the product's rounding error -- 24 significant bits, which two more bf16 values cannot hold.
__device__ __forceinline__ void bf16_rests(float x, float& r1, float& r2)
{
#pragma clang fp contract(off)
    r1 = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
    r2 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
}

__device__ __forceinline__ float bf16_rest_nc(float x)   // x minus its upper 16 bits, never fused into the producer of x
{
#pragma clang fp contract(off)
    return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
}
// colour rows of the D tile: the three split columns of a channel sit in neighbouring lanes (hi, mid, lo); t = hi + (mid + lo)
// lands in the first of them (two fused DPP adds per register; as gsr_blend_bwd.hip)
__device__ __forceinline__ void split_sum4(float v0, float v1, float v2, float v3, float& t0_, float& t1_, float& t2_, float& t3_)
{
    float s0_, s1_, s2_, s3_;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %9, %9 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %10, %10 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %11, %11 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %4, %0, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %5, %1, %9 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %6, %2, %10 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %7, %3, %11 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(s0_), "=&v"(s1_), "=&v"(s2_), "=&v"(s3_), "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_)
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3));
}

// One step of the segmented scan of affine maps in a single block (fixed order: the wait states DPP reads need behind the
// instruction that wrote their source are there by construction, see the comment at the call site).
#define GSR_PAIRS_STEP(CTRL, COND)                                                       \
    "v_cndmask_b32_e64 %[m], 0, %[a], " COND "\n\t"                                     \
    "v_fmac_f32_dpp %[b], %[b], %[m] " CTRL " bank_mask:0xf\n\t"                        \
    "v_mul_f32_dpp %[t], %[a], %[a] " CTRL " bank_mask:0xf\n\t"                         \
    "v_cndmask_b32_e64 %[a], %[a], %[t], " COND "\n\t"
// the same step for two / three rows at once, row by row inside every stage (operands a0.. b0.. m0.. t0..)
#define GSR_PAIRS_M(J, COND) "v_cndmask_b32_e64 %[m" #J "], 0, %[a" #J "], " COND "\n\t"
#define GSR_PAIRS_F(J, CTRL) "v_fmac_f32_dpp %[b" #J "], %[b" #J "], %[m" #J "] " CTRL " bank_mask:0xf\n\t"
#define GSR_PAIRS_X(J, CTRL) "v_mul_f32_dpp %[t" #J "], %[a" #J "], %[a" #J "] " CTRL " bank_mask:0xf\n\t"
#define GSR_PAIRS_A(J, COND) "v_cndmask_b32_e64 %[a" #J "], %[a" #J "], %[t" #J "], " COND "\n\t"
#define GSR_PAIRS_STEP2(CTRL, C0, C1)                                                                                    \
    GSR_PAIRS_M(0, C0) GSR_PAIRS_M(1, C1) GSR_PAIRS_F(0, CTRL) GSR_PAIRS_F(1, CTRL) GSR_PAIRS_X(0, CTRL) GSR_PAIRS_X(1, CTRL) \
    GSR_PAIRS_A(0, C0) GSR_PAIRS_A(1, C1)
#define GSR_PAIRS_STEP3(CTRL, C0, C1, C2)                                                                                \
    GSR_PAIRS_M(0, C0) GSR_PAIRS_M(1, C1) GSR_PAIRS_M(2, C2) GSR_PAIRS_F(0, CTRL) GSR_PAIRS_F(1, CTRL) GSR_PAIRS_F(2, CTRL) \
    GSR_PAIRS_X(0, CTRL) GSR_PAIRS_X(1, CTRL) GSR_PAIRS_X(2, CTRL) GSR_PAIRS_A(0, C0) GSR_PAIRS_A(1, C1) GSR_PAIRS_A(2, C2)

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GSR_PAIRS_WAVES, 8)))
blend_bwd_pairs_kernel(int W, int H, int gx, const uint4* __restrict__ unit_info, const float4* __restrict__ snap,
                       const uint2* __restrict__ masks, const uint32_t* __restrict__ point_list, const float4* __restrict__ rec_a,
                       const float4* __restrict__ rec_b, const RecTail<3>* __restrict__ rec_c, const float* __restrict__ bg,
                       const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                       const float* __restrict__ dL_dpix, float* __restrict__ grad_acc, uint64_t* __restrict__ trace)
{
    using namespace pairs;
#ifdef GSR_TRACE_DETAIL
    const uint64_t t_start = wall_clock64();
#endif
    constexpr int C = 3, SV = snap_vecs(C);
    __shared__ __attribute__((aligned(16))) unsigned char lds[TOTAL];
    float* const Rm = reinterpret_cast<float*>(lds + TBL);   // the table area doubles as staging for the B operand (head)

    // ---- unit / block of this wave: as gsr_blend_bwd.hip (XCD-aware placement: runs of 8 units per XCD, 4 blocks adjacent)
    const uint32_t n_units = gridDim.x >> 2;
    const uint32_t xcd = blockIdx.x & 7u, slot_id = blockIdx.x >> 3;
    const uint32_t grp = slot_id >> 2;
    uint32_t unit = (grp >> 3) * 64u + xcd * 8u + (grp & 7u);
    uint32_t wave_sel = slot_id & 3u;
    const uint32_t full = (n_units >> 6) << 6;
    if (blockIdx.x >= full * 4u) { unit = blockIdx.x >> 2; wave_sel = blockIdx.x & 3u; }
    const uint4 info = unit_info[unit];
    const int tile = (int)info.x;
    const uint32_t list0 = info.y;
    const int n = (int)info.z;
    const uint32_t unit0 = info.w;
    const int s0 = (int)(unit - unit0) * 64;
    const int wave = (int)wave_sel, lane = threadIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int sx = tx * TILE + (wave & 1) * SUB, sy = ty * TILE + (wave >> 1) * SUB;
    const int px = sx + (lane & 7), py = sy + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)sx, by0 = (float)sy;
    const int s1 = min(s0 + BSEG, n);
    const bool has_next = s1 < n;

    const uint32_t pix = (uint32_t)W * (uint32_t)py + (uint32_t)px;
    const uint32_t HW = (uint32_t)H * (uint32_t)W;
    const auto at32 = [](const auto* base, uint32_t byte_off) {
        return *reinterpret_cast<std::remove_reference_t<decltype(*base)>*>(reinterpret_cast<const char*>(base) + byte_off);
    };
    float T_final = 0.f;
    int my_last = 0;
    float dp[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) dp[ch] = 0.f;
    if (inside) {
        T_final = at32(final_T, pix * 4u);
        my_last = (int)at32(n_contrib, pix * 4u);
#pragma unroll
        for (int ch = 0; ch < C; ch++) dp[ch] = at32(dL_dpix, ((uint32_t)ch * HW + pix) * 4u);
    }
    const uint2* const my_words = masks + ((size_t)unit * 4 + wave) * 64 + (uint32_t)lane;
    const uint2* const words_u = masks + ((size_t)unit * 4 + wave) * 64;
    uint2 word = at32(words_u, (uint32_t)lane * 8u);
    uint2 word_next = at32(words_u + (has_next ? 256 : 0), (uint32_t)lane * 8u);
    const int pidx = 16 * (py - ty * TILE) + (px - tx * TILE);
    float Ts, Tf, cs[C], cf[C];
    const auto load_snap32 = [&](const float4* base_u, float& T_, float (&c_)[C]) {
        float v[4 * SV];
#pragma unroll
        for (int q = 0; q < SV; q++) {
            const float4 t = at32(base_u, (uint32_t)(pidx * SV + q) * 16u);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
        T_ = v[0];
#pragma unroll
        for (int ch = 0; ch < C; ch++) c_[ch] = v[ch + 1];
    };
    load_snap32(snap + (size_t)(unit + (has_next ? 1u : 0u)) * 256 * SV, Ts, cs);
    load_snap32(snap + (size_t)unit0 * 256 * SV, Tf, cf);
    // lane l holds list position s0 + 63 - l ("instance lane": ascending lanes = back to front)
    const int k = s0 + 63 - lane;
    const uint32_t kl = (uint32_t)(min(k, n - 1) - s0);
    const float4 ra = at32(rec_a + list0 + s0, kl * 16u);
    const float4 rb = at32(rec_b + list0 + s0, kl * 16u);
    const RecTail<C> rc = at32(rec_c + list0 + s0, kl * (uint32_t)sizeof(RecTail<C>));
    const uint32_t gid = at32(point_list + list0 + s0, kl * 4u);
    if (!has_next) word_next = make_uint2(0u, 0u);
    float bg_dot_dpixel = 0.f;
#pragma unroll
    for (int ch = 0; ch < C; ch++) bg_dot_dpixel += bg[ch] * dp[ch];

    float T = T_final;
    const float tf_bg = T_final * bg_dot_dpixel;
    float acc[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) acc[ch] = 0.f;
    const int my_lim = min(my_last, s1);
    if (my_last > s1) {
        if ((word_next.x | word_next.y) == 0u) {
            uint32_t useg = unit + 1u;
            const uint32_t u_end = unit0 + (uint32_t)(n + 63) / 64u;
            const auto words_of = [&](uint32_t u) { const uint2 w_ = my_words[(size_t)(u - unit) * 256]; return w_.x | w_.y; };
            do { useg++; } while (useg + 1u < u_end && words_of(useg) == 0u);
            load_snapshot<C>(snap + ((size_t)useg * 256 + pidx) * SV, Ts, cs);
        }
        const float inv = __builtin_amdgcn_rcpf(Ts);
        T = Ts;
#pragma unroll
        for (int ch = 0; ch < C; ch++) acc[ch] = (cf[ch] - cs[ch]) * inv;
    }
    float accd = 0.f;
#pragma unroll
    for (int ch = 0; ch < C; ch++) accd = __builtin_fmaf(acc[ch], dp[ch], accd);
    {
        const int lim = my_lim - s0;
        word.x &= lim >= 32 ? 0xffffffffu : lim > 0 ? (1u << lim) - 1u : 0u;
        word.y &= lim >= 64 ? 0xffffffffu : lim > 32 ? (1u << (lim - 32)) - 1u : 0u;
    }
    const unsigned long long kany = wave_or_u64_lds(lds_byte_address(Rm), word.x, word.y);
#ifdef GSR_TRACE_DETAIL   // per-wave stamps {start, head done, end, hardware id} (tests/devtools/trace_bwd_waves.py)
    const uint64_t t_head = wall_clock64();
    const auto stamp = [&](uint64_t t_end) {
        if (trace && lane == 0) {
            uint64_t* tw = trace + ((size_t)unit * 4 + wave) * 4;
            tw[0] = t_start; tw[1] = t_head; tw[2] = t_end;
            tw[3] = ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
    };
    if (kany == 0ull) stamp(t_head);
#endif
    if (kany == 0ull) return;

    // ---- B operand of the contraction (constant over the unit): as gsr_blend_bwd.hip, bf16 path, three channels
    const int kap = lane >> 4, col = lane & 15;
    constexpr int BROWS = 6 + 3 * C, BS = RSTRIDE;
    static_assert((BROWS + 1) * BS * 4 <= TBL_BYTES, "B-operand staging must fit the table area");
    {
        const float xr = (float)(lane & 7) - 3.5f, yr = (float)(lane >> 3) - 3.5f;
        Rm[0 * BS + lane] = 1.0f;
        Rm[1 * BS + lane] = xr;
        Rm[2 * BS + lane] = yr;
        Rm[3 * BS + lane] = xr * xr;
        Rm[4 * BS + lane] = xr * yr;
        Rm[5 * BS + lane] = yr * yr;
#pragma unroll
        for (int ch = 0; ch < C; ch++) {
            const float d1 = bf16_rest(dp[ch]), d2 = bf16_rest(d1);
            Rm[(6 + 3 * ch) * BS + lane] = dp[ch];
            Rm[(7 + 3 * ch) * BS + lane] = d1;
            Rm[(8 + 3 * ch) * BS + lane] = d2;
        }
        Rm[BROWS * BS + lane] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    u32x4 Bp[2];
    {
        const float4* pd = reinterpret_cast<const float4*>(&Rm[(col < BROWS ? col : BROWS) * BS + 16 * kap]);
        float bv[16];
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            const float4 v = pd[qd];
            bv[4 * qd] = v.x; bv[4 * qd + 1] = v.y; bv[4 * qd + 2] = v.z; bv[4 * qd + 3] = v.w;
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int q = 0; q < 4; q++) Bp[h][q] = bf16_pair(bv[8 * h + 2 * q], bv[8 * h + 2 * q + 1]);
    }
    __builtin_amdgcn_wave_barrier();

    // ---- a pixel's running {T, A} into LDS (pair lanes gather it by pixel); the null record
    {
        reinterpret_cast<float2*>(lds + DPIX)[lane] = make_float2(T, accd);
        if (lane < 3) reinterpret_cast<float4*>(lds + REC + Q * 48)[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // candidate word in "instance lane" order: bit l <-> list position s0 + 63 - l
    const uint32_t wrev_lo = __builtin_bitreverse32(word.y), wrev_hi = __builtin_bitreverse32(word.x);
    const bool keep = ((kany >> (63 - lane)) & 1ull) != 0ull;
    const unsigned long long m_keep = __ballot(keep);
    const int cnt_all = __popcll(m_keep);
    const int slot_g = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m_keep >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_keep, 0u));
    // per-lane thresholds of the two cross-row scan steps (a lane outside the receiving rows never combines)
    const uint32_t thr15 = (lane & 16) ? (uint32_t)(lane & 15) + 1u : 0xffffu;
    const uint32_t thr31 = (lane & 32) ? (uint32_t)(lane & 31) + 1u : 0xffffu;

    for (int q0 = 0; q0 < cnt_all; q0 += Q) {
        const int cnt = min(cnt_all - q0, Q);
        const bool more_chunks = q0 + Q < cnt_all;
        const bool inchunk = keep && slot_g >= q0 && slot_g < q0 + Q;
        const unsigned long long mc = __ballot(inchunk);
        // ---- records of the chunk's instances by slot, lane -> slot map
        {
            const int slot = slot_g - q0;
            lds[SLOT + lane] = (unsigned char)(inchunk ? slot : Q);
            if (inchunk) {
                float4* rs = reinterpret_cast<float4*>(lds + REC + slot * 48);
                rs[0] = ra;
                rs[1] = rb;
                rs[2] = make_float4(rc.c[0], __uint_as_float((uint32_t)(slot * ROWB)), 0.f, __uint_as_float(gid));
            }
        }
        // ---- pair list of the chunk, pixel-major, deepest first within a pixel
        const uint32_t wl0 = wrev_lo & (uint32_t)mc, wh0 = wrev_hi & (uint32_t)(mc >> 32);
        uint32_t wl = wl0, wh = wh0;
        const uint32_t c_p = (uint32_t)__builtin_popcount(wl) + (uint32_t)__builtin_popcount(wh);
        uint32_t incl = c_p;
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xc, 0xf, false);
        const int N = __builtin_amdgcn_readlane((int)incl, 63);
        // (a lambda: a chunk with more rows than one group writes list and pixel constants again after every group's passes)
        const auto write_phase1 = [&]() {
            float4* sp = reinterpret_cast<float4*>(lds + SPIX) + 2 * lane;
            sp[0] = make_float4(tf_bg, pxf, pyf, 0.f);
            sp[1] = make_float4(dp[0], dp[1], dp[2], 0.f);
            unsigned char* la = lds + LIST + 2u * (incl - c_p);
            uint32_t eb = (uint32_t)lane << 6;           // {pixel, rank 0}
            wl = wl0; wh = wh0;
            while (wl) {
                const uint32_t l = (uint32_t)__builtin_ctz(wl);
                wl &= wl - 1u;
                *reinterpret_cast<unsigned short*>(la) = (unsigned short)(eb | l);
                la += 2; eb += 0x1000u;
            }
            while (wh) {
                const uint32_t l = 32u + (uint32_t)__builtin_ctz(wh);
                wh &= wh - 1u;
                *reinterpret_cast<unsigned short*>(la) = (unsigned short)(eb | l);
                la += 2; eb += 0x1000u;
            }
            if (lane == 0) *reinterpret_cast<unsigned short*>(lds + LIST + 2 * N) = 0;   // sentinel: "a new pixel starts here"
            __builtin_amdgcn_wave_barrier();
        };
        write_phase1();

        // ---- version 2: the rows of the chunk in GROUPS of up to RMAX, all rows of a group as independent instruction streams
        // (gathers of every row requested before the first is used, the scans of the rows interleaved step by step); a group
        // keeps its (r, w) in registers and fills the ONE bf16 plane three times -- hi, mid, lo parts -- with the contraction's
        // matrix instructions behind every pass accumulating into the same registers.
        const int n_rows = (N + 63) >> 6;
        const int n_grp8 = (cnt + 7) >> 3;                // MFMA groups of eight instances in this chunk (1 or 2)
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        float a_c = 1.f, b_c = 0.f;                       // inclusive (a, b) of the last lane of the row before (uniform)
        const int arow = (col < 8 ? 0 : W_ROWS) + (col & 7) * ROWB + 32 * kap;
        const auto row_group = [&](auto NRtag, const int r0) {
            constexpr int NR = decltype(NRtag)::value;
            uint32_t e[NR], e_next[NR], slot[NR], q[NR], pixo[NR], cell[NR];
            bool live[NR], in_list[NR];
            float a[NR], b[NR], G[NR], ae[NR], kd[NR], T0[NR], A0[NR], tfbg[NR], rr[NR], ww[NR];
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int i = 64 * (r0 + j) + lane;
                e[j] = *reinterpret_cast<const unsigned short*>(lds + LIST + 2 * i);
                e_next[j] = *reinterpret_cast<const unsigned short*>(lds + LIST + 2 * i + 2);
                in_list[j] = i < N;
            }
#pragma unroll
            for (int j = 0; j < NR; j++) {
                slot[j] = lds[SLOT + (e[j] & 63u)];
                pixo[j] = (e[j] >> 1) & 0x7e0u;
                q[j] = e[j] >> 12;
            }
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const float4* rp = reinterpret_cast<const float4*>(lds + REC + slot[j] * 48);
                const float4 v0 = rp[0], v1 = rp[1];
                const float2 v2 = *reinterpret_cast<const float2*>(rp + 2);
                const float4* pp = reinterpret_cast<const float4*>(lds + SPIX + pixo[j]);
                const float4 S0 = pp[0], P1 = pp[1];                                    // {tf_bg, x, y, -}, {dL_dpix, -}
                const float2 TA = *reinterpret_cast<const float2*>(lds + DPIX + (pixo[j] >> 2));   // running {T, A}
                const float dx = v0.x - S0.y, dy = v0.y - S0.z;
                const float power = pair_exp2_arg(v0.z, v0.w, v1.x, dx, dy);
                G[j] = __builtin_amdgcn_exp2f(power);
                const float alpha = fminf(ALPHA_MAX, v1.y * G[j]);
                live[j] = in_list[j] && power <= 0.0f && alpha >= ALPHA_MIN;
                ae[j] = live[j] ? alpha : 0.f;
                a[j] = 1.f - ae[j];
                kd[j] = __builtin_fmaf(v2.x, P1.z, __builtin_fmaf(v1.w, P1.y, v1.z * P1.x));
                b[j] = ae[j] * kd[j];
                T0[j] = TA.x; A0[j] = TA.y; tfbg[j] = S0.x;
                cell[j] = __float_as_uint(v2.y) + (pixo[j] >> 4);      // row offset + 2 * pixel
            }
            // segmented inclusive scans of the affine maps, the NR rows step by step (see version 1 for the step; with the rows
            // interleaved every DPP read sits at least 2 NR instructions behind the write of its source)
            {
                unsigned long long c1[NR], c2[NR], c4[NR], c8[NR], c15[NR], c31[NR];
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    const uint32_t qq = min(q[j], (uint32_t)(lane & 15));
                    c1[j] = __ballot(qq >= 1u); c2[j] = __ballot(qq >= 2u); c4[j] = __ballot(qq >= 4u); c8[j] = __ballot(qq >= 8u);
                    c15[j] = __ballot(q[j] >= thr15); c31[j] = __ballot(q[j] >= thr31);
                }
                float m_[NR], t_[NR];
                if constexpr (NR == 1) {
                    asm volatile("s_nop 1\n\t"
                                 GSR_PAIRS_STEP("row_shr:1 row_mask:0xf", "%[c1]") GSR_PAIRS_STEP("row_shr:2 row_mask:0xf", "%[c2]")
                                 GSR_PAIRS_STEP("row_shr:4 row_mask:0xf", "%[c4]") GSR_PAIRS_STEP("row_shr:8 row_mask:0xf", "%[c8]")
                                 GSR_PAIRS_STEP("row_bcast:15 row_mask:0xa", "%[c15]") GSR_PAIRS_STEP("row_bcast:31 row_mask:0xc", "%[c31]")
                                 : [a] "+v"(a[0]), [b] "+v"(b[0]), [m] "=&v"(m_[0]), [t] "=&v"(t_[0])
                                 : [c1] "s"(c1[0]), [c2] "s"(c2[0]), [c4] "s"(c4[0]), [c8] "s"(c8[0]), [c15] "s"(c15[0]), [c31] "s"(c31[0]));
                } else if constexpr (NR == 2) {
                    asm volatile("s_nop 1\n\t"
                                 GSR_PAIRS_STEP2("row_shr:1 row_mask:0xf", "%[p0]", "%[q0]") GSR_PAIRS_STEP2("row_shr:2 row_mask:0xf", "%[p1]", "%[q1]")
                                 GSR_PAIRS_STEP2("row_shr:4 row_mask:0xf", "%[p2]", "%[q2]") GSR_PAIRS_STEP2("row_shr:8 row_mask:0xf", "%[p3]", "%[q3]")
                                 GSR_PAIRS_STEP2("row_bcast:15 row_mask:0xa", "%[p4]", "%[q4]") GSR_PAIRS_STEP2("row_bcast:31 row_mask:0xc", "%[p5]", "%[q5]")
                                 : [a0] "+v"(a[0]), [b0] "+v"(b[0]), [m0] "=&v"(m_[0]), [t0] "=&v"(t_[0]),
                                   [a1] "+v"(a[1]), [b1] "+v"(b[1]), [m1] "=&v"(m_[1]), [t1] "=&v"(t_[1])
                                 : [p0] "s"(c1[0]), [p1] "s"(c2[0]), [p2] "s"(c4[0]), [p3] "s"(c8[0]), [p4] "s"(c15[0]), [p5] "s"(c31[0]),
                                   [q0] "s"(c1[1]), [q1] "s"(c2[1]), [q2] "s"(c4[1]), [q3] "s"(c8[1]), [q4] "s"(c15[1]), [q5] "s"(c31[1]));
                } else {
                    static_assert(NR <= 3, "scan blocks are written for up to three rows");
                    asm volatile("s_nop 1\n\t"
                                 GSR_PAIRS_STEP3("row_shr:1 row_mask:0xf", "%[p0]", "%[q0]", "%[s0]") GSR_PAIRS_STEP3("row_shr:2 row_mask:0xf", "%[p1]", "%[q1]", "%[s1]")
                                 GSR_PAIRS_STEP3("row_shr:4 row_mask:0xf", "%[p2]", "%[q2]", "%[s2]")
                                 : [a0] "+v"(a[0]), [b0] "+v"(b[0]), [m0] "=&v"(m_[0]), [t0] "=&v"(t_[0]),
                                   [a1] "+v"(a[1]), [b1] "+v"(b[1]), [m1] "=&v"(m_[1]), [t1] "=&v"(t_[1]),
                                   [a2] "+v"(a[NR - 1]), [b2] "+v"(b[NR - 1]), [m2] "=&v"(m_[NR - 1]), [t2] "=&v"(t_[NR - 1])
                                 : [p0] "s"(c1[0]), [p1] "s"(c2[0]), [p2] "s"(c4[0]), [q0] "s"(c1[1]), [q1] "s"(c2[1]), [q2] "s"(c4[1]),
                                   [s0] "s"(c1[NR - 1]), [s1] "s"(c2[NR - 1]), [s2] "s"(c4[NR - 1]));
                    asm volatile("s_nop 1\n\t"
                                 GSR_PAIRS_STEP3("row_shr:8 row_mask:0xf", "%[p0]", "%[q0]", "%[s0]") GSR_PAIRS_STEP3("row_bcast:15 row_mask:0xa", "%[p1]", "%[q1]", "%[s1]")
                                 GSR_PAIRS_STEP3("row_bcast:31 row_mask:0xc", "%[p2]", "%[q2]", "%[s2]")
                                 : [a0] "+v"(a[0]), [b0] "+v"(b[0]), [m0] "=&v"(m_[0]), [t0] "=&v"(t_[0]),
                                   [a1] "+v"(a[1]), [b1] "+v"(b[1]), [m1] "=&v"(m_[1]), [t1] "=&v"(t_[1]),
                                   [a2] "+v"(a[NR - 1]), [b2] "+v"(b[NR - 1]), [m2] "=&v"(m_[NR - 1]), [t2] "=&v"(t_[NR - 1])
                                 : [p0] "s"(c8[0]), [p1] "s"(c15[0]), [p2] "s"(c31[0]), [q0] "s"(c8[1]), [q1] "s"(c15[1]), [q2] "s"(c31[1]),
                                   [s0] "s"(c8[NR - 1]), [s1] "s"(c15[NR - 1]), [s2] "s"(c31[NR - 1]));
                }
            }
            // per row, in order (the carry runs from row to row): compose with the row before, exclusive maps, T and A, r and w
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const bool cont = q[j] > (uint32_t)lane;
                const float m = cont ? a[j] : 0.f;
                b[j] = __builtin_fmaf(b_c, m, b[j]);
                a[j] = cont ? a[j] * a_c : a[j];
                float a_ex = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(a_c), __float_as_int(a[j]), 0x138, 0xf, 0xf, false));
                float b_ex = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(b_c), __float_as_int(b[j]), 0x138, 0xf, 0xf, false));
                a_ex = q[j] == 0u ? 1.f : a_ex;
                b_ex = q[j] == 0u ? 0.f : b_ex;
                a_c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[j]), 63));
                b_c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b[j]), 63));
                const float A_before = __builtin_fmaf(a_ex, A0[j], b_ex);
                const float Tn = T0[j] * __builtin_amdgcn_rcpf(a[j]);
                if (more_chunks && e_next[j] < 0x1000u && in_list[j])
                    *reinterpret_cast<float2*>(lds + DPIX + (pixo[j] >> 2)) = make_float2(Tn, __builtin_fmaf(a[j], A0[j], b[j]));
                ww[j] = ae[j] * Tn;
                const float s_ = kd[j] - A_before;
                const float rinv = __builtin_amdgcn_rcpf(1.f - ae[j]);
                rr[j] = G[j] * __builtin_fmaf(s_, Tn, -(rinv * tfbg[j]));
            }
            // the plane: every gather of the group has been issued -- zero it (it held the list and the pixel constants), then the
            // hi / mid / lo passes
            {
                __builtin_amdgcn_wave_barrier();
                float4* t4 = reinterpret_cast<float4*>(lds + TBL);
#pragma unroll
                for (int i = 0; i < (TBL_BYTES / 16 + 63) / 64; i++)
                    if (i * 64 + lane < TBL_BYTES / 16) t4[i * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int pass = 0; pass < 3; pass++) {
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    if (live[j]) {
                        unsigned char* tp = lds + TBL + cell[j];
                        *reinterpret_cast<unsigned short*>(tp) = (unsigned short)(__float_as_uint(rr[j]) >> 16);
                        *reinterpret_cast<unsigned short*>(tp + W_ROWS) = (unsigned short)(__float_as_uint(ww[j]) >> 16);
                    }
                    if (pass < 2) { rr[j] = bf16_rest_nc(rr[j]); ww[j] = bf16_rest_nc(ww[j]); }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    if (g < n_grp8) {
                        const unsigned char* ab = lds + TBL + arow + g * 8 * ROWB;
                        const u32x4 a0_ = *reinterpret_cast<const u32x4*>(ab), a1_ = *reinterpret_cast<const u32x4*>(ab + 16);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0_), __builtin_bit_cast(bf16x8, Bp[0]), acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1_), __builtin_bit_cast(bf16x8, Bp[1]), acc[g], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        };
        for (int r0 = 0; r0 < n_rows; r0 += RMAX) {
            const int left = n_rows - r0;
            if (RMAX >= 3 && left >= 3) row_group(std::integral_constant<int, (RMAX >= 3 ? 3 : 1)>{}, r0);
            else if (RMAX >= 2 && left >= 2) row_group(std::integral_constant<int, (RMAX >= 2 ? 2 : 1)>{}, r0);
            else row_group(std::integral_constant<int, 1>{}, r0);
            if (r0 + RMAX < n_rows) write_phase1();       // (rare: a chunk of more than RMAX rows) list and constants again
        }
        // moments of the chunk's instances out of the accumulators (D layout, split-column sums: as gsr_blend_bwd.hip)
        {
            const int wb_row0 = 4 * (kap & 1);
            const bool wb_take = kap < 2 ? col < 6 : (col >= 6 && col < 6 + 3 * C && (col % 3) == 0);
            float* const recf = reinterpret_cast<float*>(lds + REC);
            float* const wb_ptr = recf + wb_row0 * SF + MOM0 + (col >= 6 ? 6 + (col - 6) / 3 : col);
#pragma unroll
            for (int g = 0; g < 2; g++) {
                if (g < n_grp8) {
                    const float d0 = acc[g][0], d1 = acc[g][1], d2 = acc[g][2], d3 = acc[g][3];
                    float t0_, t1_, t2_, t3_;
                    split_sum4(d0, d1, d2, d3, t0_, t1_, t2_, t3_);
                    const bool spatial = kap < 2;
                    const float o0 = spatial ? d0 : t0_, o1 = spatial ? d1 : t1_, o2 = spatial ? d2 : t2_, o3 = spatial ? d3 : t3_;
                    if (wb_take) {
                        float* const dst = wb_ptr + g * 8 * SF;
                        const int left = cnt - g * 8 - wb_row0;
                        if (0 < left) dst[0 * SF] = o0;
                        if (1 < left) dst[1 * SF] = o1;
                        if (2 < left) dst[2 * SF] = o2;
                        if (3 < left) dst[3 * SF] = o3;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- lane = instance of the chunk: re-centre the spatial sums on the splat (as gsr_blend_bwd.hip), then row-major flush
        if (lane < cnt) {
            float* rw = reinterpret_cast<float*>(lds + REC) + lane * SF;
            const float m0 = rw[MOM0], mx = rw[MOM0 + 1], my = rw[MOM0 + 2], mxx = rw[MOM0 + 3], mxy = rw[MOM0 + 4], myy = rw[MOM0 + 5];
            const float X = rw[0] - (bx0 + 3.5f), Y = rw[1] - (by0 + 3.5f);
            rw[MOM0 + 1] = X * m0 - mx;
            rw[MOM0 + 2] = Y * m0 - my;
            rw[MOM0 + 3] = (X * X) * m0 - 2.f * X * mx + mxx;
            rw[MOM0 + 4] = (X * Y) * m0 - X * my - Y * mx + mxy;
            rw[MOM0 + 5] = (Y * Y) * m0 - 2.f * Y * my + myy;
        }
        __builtin_amdgcn_wave_barrier();
        {
            const float* recf = reinterpret_cast<const float*>(lds + REC);
            for (int idx = lane; idx < cnt * NM; idx += 64) {
                const int e = idx / NM, v = idx - e * NM;
                const size_t g = __float_as_uint(recf[e * SF + 11]);
                atomic_add_f32(grad_acc + g * GRAD_RS + v, recf[e * SF + MOM0 + v]);
            }
        }
        __builtin_amdgcn_wave_barrier();   // records, map and table are rewritten by the next chunk
    }
#ifdef GSR_TRACE_DETAIL
    stamp(wall_clock64());
#endif
}

// the hook gsr_blend_bwd.hip calls in a -DGSR_BWD_VARIANT build (GSR_BWD_PAIRS=0: the product's pair loop)
bool launch_blend_bwd_variant(int C, int W, int H, int U, const float* bg, ImageState im, BinState b, const float* dL_dpix, float* grad_acc,
                              hipStream_t st)
{
    static const int on = getenv("GSR_BWD_PAIRS") ? atoi(getenv("GSR_BWD_PAIRS")) : 1;
    if (!on || C != 3) return false;
    if (U <= 0) return true;
    const Tiles t = tiles_of(W, H);
    static const int pad = getenv("GSR_BWD_LDS_PAD") ? atoi(getenv("GSR_BWD_LDS_PAD")) : 0;
    uint64_t* tr = g_trace ? g_trace + 2 * (size_t)t.T : nullptr;
    blend_bwd_pairs_kernel<<<4 * U, 64, pad, st>>>(W, H, t.gx, b.unit_info, b.snap, b.masks, b.point_list, b.rec_a, b.rec_b,
                                                   static_cast<const RecTail<3>*>(b.rec_c), bg, im.final_T, im.n_contrib, dL_dpix,
                                                   grad_acc, tr);
    return true;
}

}  // namespace gsr
