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
#ifndef GSR_PAIRS_WAVES
#define GSR_PAIRS_WAVES 2
#endif

namespace pairs {
constexpr int Q = GSR_PAIRS_Q;            // kept instances per chunk (a pixel has at most Q pairs per chunk: rank fits four bits)
static_assert(Q == 8 || Q == 16, "chunk: whole MFMA groups, rank within a pixel in four bits");
constexpr int ROWB = 144;                 // bytes per table row: 64 pixels of bf16 + 16 (sixteen rows 144 B apart cover all banks)
constexpr int W_ROWS = Q * ROWB + 128;    // w rows behind the r rows of a plane, shifted by 32 banks
constexpr int PLANE = W_ROWS + Q * ROWB;  // one bf16 plane: r rows [0, Q), w rows [Q, 2 Q)
constexpr int TBL = 0, TBL_BYTES = 3 * PLANE;        // hi, mid, lo
constexpr int REC = TBL + TBL_BYTES;      // Q + 1 records of 48 B: {x, y, a', b'}, {c', opacity, c0, c1}, {c2, row offset, -, gaussian}
constexpr int REC_BYTES = (Q + 1) * 48;   //   (the last one: a null record for lanes outside the chunk)
constexpr int SLOT = REC + REC_BYTES;     // 64 bytes: instance lane -> record slot of the chunk (Q: none)
constexpr int PIX = (SLOT + 64 + 15) & ~15;   // 64 pixel states of 32 B: {T, A, T_final bg.dL_dpix, x}, {dL_dpix 0..2, y}
constexpr int LIST = PIX + 64 * 32;       // pair list: u16 entries {instance lane : 6, pixel : 6, rank : 4}
constexpr int LIST_BYTES = (Q * 64 + 64 + 8) * 2;   // every pixel x every instance, + a row of slack for the look-ahead read
constexpr int TOTAL = LIST + LIST_BYTES;
static_assert(PIX % 16 == 0 && REC % 16 == 0 && PLANE % 16 == 0, "16-byte accesses");
constexpr int SF = 12, MOM0 = 2, NM = 9;  // record floats, first moment float, moments per instance
}   // namespace pairs

// x = hi + r1's upper half + r2 exactly (three bf16 values); contraction off: `x` is a product at the call site, and fused into
// the subtraction r1 would carry the product's rounding error -- 24 significant bits, which two more bf16 values cannot hold.
__device__ __forceinline__ void bf16_rests(float x, float& r1, float& r2)
{
#pragma clang fp contract(off)
    r1 = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
    r2 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
}

// One step of the segmented scan of affine maps in a single block (fixed order: the wait states DPP reads need behind the
// instruction that wrote their source are there by construction, see the comment at the call site).
#define GSR_PAIRS_STEP(CTRL, COND)                                                       \
    "v_cndmask_b32_e64 %[m], 0, %[a], " COND "\n\t"                                     \
    "v_fmac_f32_dpp %[b], %[b], %[m] " CTRL " bank_mask:0xf\n\t"                        \
    "v_mul_f32_dpp %[t], %[a], %[a] " CTRL " bank_mask:0xf\n\t"                         \
    "v_cndmask_b32_e64 %[a], %[a], %[t], " COND "\n\t"

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

    // ---- per-pixel state into LDS (pair lanes gather it by pixel); the null record
    {
        float4* ps = reinterpret_cast<float4*>(lds + PIX) + 2 * lane;
        ps[0] = make_float4(T, accd, tf_bg, pxf);
        ps[1] = make_float4(dp[0], dp[1], dp[2], pyf);
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
        // ---- records of the chunk's instances by slot, lane -> slot map, table zeroed
        {
            const int slot = slot_g - q0;
            lds[SLOT + lane] = (unsigned char)(inchunk ? slot : Q);
            if (inchunk) {
                float4* rs = reinterpret_cast<float4*>(lds + REC + slot * 48);
                rs[0] = ra;
                rs[1] = rb;
                rs[2] = make_float4(rc.c[0], __uint_as_float((uint32_t)(slot * ROWB)), 0.f, __uint_as_float(gid));
            }
            float4* t4 = reinterpret_cast<float4*>(lds + TBL);
            for (int i = lane; i < TBL_BYTES / 16; i += 64) t4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // ---- pair list of the chunk, pixel-major, deepest first within a pixel
        uint32_t wl = wrev_lo & (uint32_t)mc, wh = wrev_hi & (uint32_t)(mc >> 32);
        const uint32_t c_p = (uint32_t)__builtin_popcount(wl) + (uint32_t)__builtin_popcount(wh);
        uint32_t incl = c_p;
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xc, 0xf, false);
        const int N = __builtin_amdgcn_readlane((int)incl, 63);
        {
            unsigned char* la = lds + LIST + 2u * (incl - c_p);
            uint32_t eb = (uint32_t)lane << 6;           // {pixel, rank 0}
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
        }
        __builtin_amdgcn_wave_barrier();

        // ---- rows of 64 pairs
        float a_c = 1.f, b_c = 0.f;                       // inclusive (a, b) of the previous row's last lane (uniform)
        const int n_rows = (N + 63) >> 6;
        for (int r = 0; r < n_rows; r++) {
            const int i = 64 * r + lane;
            const uint32_t e = *reinterpret_cast<const unsigned short*>(lds + LIST + 2 * i);
            const uint32_t e_next = *reinterpret_cast<const unsigned short*>(lds + LIST + 2 * i + 2);
            const uint32_t l_i = e & 63u, pixo = (e >> 1) & 0x7e0u, q = e >> 12;
            const uint32_t slot = lds[SLOT + l_i];
            const float4* rp = reinterpret_cast<const float4*>(lds + REC + slot * 48);
            const float4 v0 = rp[0], v1 = rp[1], v2 = rp[2];
            const float4* pp = reinterpret_cast<const float4*>(lds + PIX + pixo);
            const float4 P0 = pp[0], P1 = pp[1];
            const float dx = v0.x - P0.w, dy = v0.y - P1.w;
            const float power = pair_exp2_arg(v0.z, v0.w, v1.x, dx, dy);
            const float G = __builtin_amdgcn_exp2f(power);
            const float alpha = fminf(ALPHA_MAX, v1.y * G);
            const bool live = i < N && power <= 0.0f && alpha >= ALPHA_MIN;
            const float ae = live ? alpha : 0.f;
            float a = 1.f - ae;
            const float kd = __builtin_fmaf(v2.x, P1.z, __builtin_fmaf(v1.w, P1.y, v1.z * P1.x));
            float b = ae * kd;
            // segmented inclusive scan of the affine maps over the row.  A lane combines with lane i - d iff its rank within
            // its pixel is >= d (the pairs of a pixel are consecutive) and lane i - d exists in the DPP row: qq folds both.
            // Order of the block: cndmask m | fmac_dpp b | mul_dpp t | cndmask a -- b is DPP-read three instructions after it was
            // written, a two instructions after (the two wait states a DPP read needs); the leading s_nop covers the first step
            // and the two wait states a vector read of a vector-written mask register needs on gfx950.
            {
                const uint32_t qq = min(q, (uint32_t)(lane & 15));
                const unsigned long long c1 = __ballot(qq >= 1u), c2 = __ballot(qq >= 2u), c4 = __ballot(qq >= 4u), c8 = __ballot(qq >= 8u);
                const unsigned long long c15 = __ballot(q >= thr15), c31 = __ballot(q >= thr31);
                float m_, t_;
                asm volatile("s_nop 1\n\t"
                             GSR_PAIRS_STEP("row_shr:1 row_mask:0xf", "%[c1]")
                             GSR_PAIRS_STEP("row_shr:2 row_mask:0xf", "%[c2]")
                             GSR_PAIRS_STEP("row_shr:4 row_mask:0xf", "%[c4]")
                             GSR_PAIRS_STEP("row_shr:8 row_mask:0xf", "%[c8]")
                             GSR_PAIRS_STEP("row_bcast:15 row_mask:0xa", "%[c15]")
                             GSR_PAIRS_STEP("row_bcast:31 row_mask:0xc", "%[c31]")
                             : [a] "+v"(a), [b] "+v"(b), [m] "=&v"(m_), [t] "=&v"(t_)
                             : [c1] "s"(c1), [c2] "s"(c2), [c4] "s"(c4), [c8] "s"(c8), [c15] "s"(c15), [c31] "s"(c31));
            }
            // a pixel that began in the previous row: compose with that row's last lane (rank > lane)
            {
                const bool cont = q > (uint32_t)lane;
                const float m = cont ? a : 0.f;
                b = __builtin_fmaf(b_c, m, b);
                a = cont ? a * a_c : a;
            }
            // exclusive maps: the inclusive ones of the lane below (lane 0: the carry), identity at a pixel's first pair
            float a_ex = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(a_c), __float_as_int(a), 0x138, 0xf, 0xf, false));
            float b_ex = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(b_c), __float_as_int(b), 0x138, 0xf, 0xf, false));
            a_ex = q == 0u ? 1.f : a_ex;
            b_ex = q == 0u ? 0.f : b_ex;
            a_c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
            b_c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
            const float A_before = __builtin_fmaf(a_ex, P0.y, b_ex);
            const float Tn = P0.x * __builtin_amdgcn_rcpf(a);          // T in front of this pair
            if (more_chunks && e_next < 0x1000u && i < N) {            // last pair of its pixel in this chunk: state for the next
                *reinterpret_cast<float2*>(lds + PIX + pixo) = make_float2(Tn, __builtin_fmaf(a, P0.y, b));
            }
            if (live) {
                const float w = ae * Tn;
                const float s = kd - A_before;
                const float rinv = __builtin_amdgcn_rcpf(1.f - ae);
                const float rr = G * __builtin_fmaf(s, Tn, -(rinv * P0.z));
                // exact bf16 splits at the writer (hi + mid + lo == value), three planes, [instance row][pixel]
                const uint32_t cell = __float_as_uint(v2.y) + (pixo >> 4);       // row offset + 2 * pixel
                float r1, r2, w1, w2;
                bf16_rests(rr, r1, r2);
                bf16_rests(w, w1, w2);
                unsigned char* tp = lds + TBL + cell;
                *reinterpret_cast<unsigned short*>(tp) = (unsigned short)(__float_as_uint(rr) >> 16);
                *reinterpret_cast<unsigned short*>(tp + PLANE) = (unsigned short)(__float_as_uint(r1) >> 16);
                *reinterpret_cast<unsigned short*>(tp + 2 * PLANE) = (unsigned short)(__float_as_uint(r2) >> 16);
                *reinterpret_cast<unsigned short*>(tp + W_ROWS) = (unsigned short)(__float_as_uint(w) >> 16);
                *reinterpret_cast<unsigned short*>(tp + W_ROWS + PLANE) = (unsigned short)(__float_as_uint(w1) >> 16);
                *reinterpret_cast<unsigned short*>(tp + W_ROWS + 2 * PLANE) = (unsigned short)(__float_as_uint(w2) >> 16);
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- contraction per group of eight instances: A operand rows 0..7 = r, 8..15 = w of the group's instances, read
        // straight from the bf16 planes (lane: row l & 15, pixels 16 kap + 8 h .. + 7 of half h)
        {
            const int arow = (col < 8 ? 0 : W_ROWS) + (col & 7) * ROWB + 32 * kap;
            const int wb_row0 = 4 * (kap & 1);
            const bool wb_take = kap < 2 ? col < 6 : (col >= 6 && col < 6 + 3 * C && (col % 3) == 0);
            float* const recf = reinterpret_cast<float*>(lds + REC);
            float* const wb_ptr = recf + wb_row0 * SF + MOM0 + (col >= 6 ? 6 + (col - 6) / 3 : col);
            for (int g0i = 0; g0i < cnt; g0i += 8) {
                const unsigned char* ab = lds + TBL + arow + g0i * ROWB;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const u32x4 a_hi = *reinterpret_cast<const u32x4*>(ab + 16 * h);
                    const u32x4 a_mid = *reinterpret_cast<const u32x4*>(ab + PLANE + 16 * h);
                    const u32x4 a_lo = *reinterpret_cast<const u32x4*>(ab + 2 * PLANE + 16 * h);
                    const bf16x8 bb = __builtin_bit_cast(bf16x8, Bp[h]);
                    f32x4& ac = h ? acc1 : acc0;
                    ac = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_lo), bb, ac, 0, 0, 0);
                    ac = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_mid), bb, ac, 0, 0, 0);
                    ac = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_hi), bb, ac, 0, 0, 0);
                }
                const auto split_sum = [](float v0_, float v1_, float v2_, float v3_, float& t0_, float& t1_, float& t2_, float& t3_) {
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
                        : "v"(v0_), "v"(v1_), "v"(v2_), "v"(v3_));
                };
                const float d0 = acc0[0] + acc1[0], d1 = acc0[1] + acc1[1], d2 = acc0[2] + acc1[2], d3 = acc0[3] + acc1[3];
                float t0_, t1_, t2_, t3_;
                split_sum(d0, d1, d2, d3, t0_, t1_, t2_, t3_);
                const bool spatial = kap < 2;
                const float o0 = spatial ? d0 : t0_, o1 = spatial ? d1 : t1_, o2 = spatial ? d2 : t2_, o3 = spatial ? d3 : t3_;
                if (wb_take) {
                    float* const dst = wb_ptr + g0i * SF;
                    const int left = cnt - g0i - wb_row0;
                    if (0 < left) dst[0 * SF] = o0;
                    if (1 < left) dst[1 * SF] = o1;
                    if (2 < left) dst[2 * SF] = o2;
                    if (3 < left) dst[3 * SF] = o3;
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
