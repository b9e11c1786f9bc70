// gsr_blend_bwd_walk.hip -- backward alpha compositing, PER-LANE WALK variant of gsr_blend_bwd.hip.
//
// Same per-pair arithmetic (DGR/cuda_rasterizer/backward.cu:464-556), same work unit
// (tile, 64-entry segment, 8x8 block -> one wave64), same snapshots, same [instances x 64 pixels].[64 pixels x 9] moment
// contraction on the bf16 matrix pipe and the same row-major flush as gsr_blend_bwd.hip.  What differs is WHO evaluates a
// pair.  There, every lane evaluates every kept instance of the unit (one trip per kept instance: 28 trips per unit-block
// on config C at 10 live lanes of 64).  Here every lane walks the set bits of ITS OWN candidate word, deepest first -- as
// the forward blend does -- and parks its (r, w) in the r|w table row of the instance it is at; lanes of one wave are at
// different instances at the same time.  A table row exists only for ROWS kept instances at a time, so a unit is walked
// in sub-chunks: the kept instances of the upper 32 list positions (owner lanes 0..31, deepest first), ROWS at a time,
// then those of the lower 32.  A sub-chunk costs max-over-lanes(candidates in it) trips instead of one per kept instance.
//
//   records    every lane parks the record of "its" list position (lane l <-> position s0 + 63 - l) in LDS, all 64, with
//              the byte offset of the instance's table row in it; walkers gather by position (ds_read_b128, per-lane address);
//   sub-chunk  uniform: the candidate bits of a lane are cut to the sub-chunk's owners (one v_and with a ballot);
//   table      zeroed per sub-chunk (a lane only writes the rows it visits); rows 16 g + j = r, 16 g + 8 + j = w of the
//              sub-chunk's instance 8 g + j -- group g is one operand of the contraction exactly as in gsr_blend_bwd.hip.
//
// Selected at run time by GSR_BWD_WALK=<ROWS> (8, 16 or 32); three and four channels.  DESIGN.md section 6 has the counters.
#include "gsr_bwd_util.h"

#ifndef GSR_WALK_PIPE
#define GSR_WALK_PIPE 2   // 0: compiler-scheduled per-lane loop; 1: explicit batched gathers; 2: + one trip of software pipelining
#endif

namespace gsr {

template <int C, int ROWS>
__global__ void __launch_bounds__(64)
blend_bwd_walk_kernel(int W, int H, int gx, const uint4* __restrict__ unit_info, const float4* __restrict__ snap,
                      const uint2* __restrict__ masks, const uint32_t* __restrict__ point_list, const float4* __restrict__ rec_a,
                      const float4* __restrict__ rec_b, const RecTail<C>* __restrict__ rec_c, const float* __restrict__ bg,
                      const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                      const float* __restrict__ dL_dpix, float* __restrict__ grad_acc)
{
    constexpr int NM = 6 + C, SV = snap_vecs(C);
    static_assert(C == 3 || C == 4, "the walk variant is built for three and four channels (bf16 contraction)");
    static_assert(ROWS % GRP == 0 && ROWS >= GRP && ROWS <= 32 && GRP == 8, "sub-chunks are whole MFMA groups within one half");
    constexpr int NG = ROWS / GRP;                    // MFMA groups per sub-chunk
    constexpr int RF = (7 + C + 3) / 4 * 4;           // record floats: x y a b | c o col0 col1 | col2 .. col(C-1), row offset
    constexpr int RV = RF / 4;
    constexpr int ROFF = 6 + C;                       // float index of the row byte offset
    constexpr int RT = 16;                            // floats per row of rowtab: x, y, gaussian id, -, then NM moments
    static_assert(RV == 3 && 4 + NM <= RT, "");
    __shared__ __attribute__((aligned(16))) float Rm[NG * 2 * GRP * RSTRIDE];   // r|w table, see above
    __shared__ __attribute__((aligned(16))) float rec[64 * RF];
    __shared__ __attribute__((aligned(16))) float rowtab[ROWS * RT];

    // ---- placement and head: identical to gsr_blend_bwd.hip (XCD-aware unit map; every load of the head in one batch)
    const uint32_t n_units = gridDim.x >> 2;
    const uint32_t xcd = blockIdx.x & 7u, slot_id = blockIdx.x >> 3;
    const uint32_t grp_id = slot_id >> 2;
    uint32_t unit = (grp_id >> 3) * 64u + xcd * 8u + (grp_id & 7u);
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
    const int k = s0 + 63 - lane;                          // lane l owns list position s0 + 63 - l
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
    {
        const int lim = my_lim - s0;
        word.x &= lim >= 32 ? 0xffffffffu : lim > 0 ? (1u << lim) - 1u : 0u;
        word.y &= lim >= 64 ? 0xffffffffu : lim > 32 ? (1u << (lim - 32)) - 1u : 0u;
    }
    const unsigned long long kany = wave_or_u64_lds(lds_byte_address(Rm), word.x, word.y);
    if (kany == 0ull) return;

    // ---- B operand of the contraction (constant over the unit): as in gsr_blend_bwd.hip, staged through the table's rows
    const int kap = lane >> 4, col = lane & 15;
    constexpr int C1 = C < 3 ? C : 3, C2 = C - C1;
    constexpr int BROWS = 6 + 3 * C1;
    constexpr int BS = RSTRIDE;
    static_assert((BROWS + 1) * BS <= 2 * GRP * RSTRIDE, "B-operand staging must fit one group of the r|w table");
    {
        const float xr = (float)(lane & 7) - 3.5f, yr = (float)(lane >> 3) - 3.5f;
        Rm[0 * BS + lane] = 1.0f;
        Rm[1 * BS + lane] = xr;
        Rm[2 * BS + lane] = yr;
        Rm[3 * BS + lane] = xr * xr;
        Rm[4 * BS + lane] = xr * yr;
        Rm[5 * BS + lane] = yr * yr;
#pragma unroll
        for (int ch = 0; ch < C1; ch++) {
            const float d1 = bf16_rest(dp[ch]), d2 = bf16_rest(d1);
            Rm[(6 + 3 * ch) * BS + lane] = dp[ch];
            Rm[(7 + 3 * ch) * BS + lane] = d1;
            Rm[(8 + 3 * ch) * BS + lane] = d2;
        }
        Rm[BROWS * BS + lane] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    u32x4 Bp[2];
    u32x4 Bp2[C2 > 0 ? 2 : 1];
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
    if constexpr (C2 > 0) {
#pragma unroll
        for (int ch = 0; ch < C2; ch++) {
            const float d0 = dp[C1 + ch], d1 = bf16_rest(d0), d2 = bf16_rest(d1);
            Rm[(3 * ch) * BS + lane] = d0;
            Rm[(3 * ch + 1) * BS + lane] = d1;
            Rm[(3 * ch + 2) * BS + lane] = d2;
        }
        Rm[3 * C2 * BS + lane] = 0.f;
        __builtin_amdgcn_wave_barrier();
        const float4* pd = reinterpret_cast<const float4*>(&Rm[(col < 3 * C2 ? col : 3 * C2) * BS + 16 * kap]);
        float bv[16];
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            const float4 v = pd[qd];
            bv[4 * qd] = v.x; bv[4 * qd + 1] = v.y; bv[4 * qd + 2] = v.z; bv[4 * qd + 3] = v.w;
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int q = 0; q < 4; q++) Bp2[h][q] = bf16_pair(bv[8 * h + 2 * q], bv[8 * h + 2 * q + 1]);
        __builtin_amdgcn_wave_barrier();
    }

    // ---- owners: which positions are kept, their rank within their half (deepest first) and their table row
    const bool keep = ((kany >> (63 - lane)) & 1ull) != 0ull;
    const unsigned long long m = __ballot(keep);
    const uint32_t m_lo = (uint32_t)m, m_hi = (uint32_t)(m >> 32);
    const int my_half = lane >> 5;
    const int below_lo = (int)__builtin_amdgcn_mbcnt_lo(m_lo, 0u);
    const int rank_h = my_half ? (int)__builtin_amdgcn_mbcnt_hi(m_hi, 0u) : below_lo;
    {
        const int rr = rank_h % ROWS;
        const uint32_t rowoff = (uint32_t)(((rr >> 3) * 2 * GRP + (rr & 7)) * RSTRIDE * 4);
        float tail[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 2; ch < C; ch++) tail[ch - 2] = rc.c[ch - 2];
        tail[ROFF - 8] = __uint_as_float(rowoff);
        float4* rs = reinterpret_cast<float4*>(&rec[lane * RF]);
        rs[0] = ra;
        rs[1] = rb;
        rs[2] = make_float4(tail[0], tail[1], tail[2], tail[3]);
    }
    const uint32_t rw_lane = 4u * (uint32_t)lane;   // this lane's column of the r|w table (bytes)

    // where this lane's four accumulator registers go (rows 4 kap .. 4 kap + 3 of the D tile), see gsr_blend_bwd.hip
    const int wb_row0 = 4 * (kap & 1);
    const bool wb_take = kap < 2 ? col < 6 : (col >= 6 && col < 6 + 3 * C1 && (col % 3) == 0);
    const int wb_col = 4 + (col >= 6 ? 6 + (col - 6) / 3 : col);
    const bool wb_take2 = C2 > 0 && kap >= 2 && col < 3 * C2 && (col % 3) == 0;
    const int wb_col2 = 4 + 6 + C1 + (col < 3 * C2 ? col / 3 : 0);

#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        const uint32_t mh = half ? m_hi : m_lo;            // (uniform) kept owners of this half, bit = owner lane - 32 half
        const int cnt_h = __popc(mh);
        if (cnt_h == 0) continue;
        // this lane's candidates of the half in owner order: bit i <-> owner lane 32 half + i (i = 0: deepest)
        const uint32_t rev = __builtin_bitreverse32(half ? word.x : word.y);
#pragma unroll 1
        for (int sc0 = 0; sc0 < cnt_h; sc0 += ROWS) {
            const int cnt = min(ROWS, cnt_h - sc0);
            const int ng = (cnt + GRP - 1) / GRP;
            const bool mine = keep && my_half == half && (unsigned)(rank_h - sc0) < (unsigned)ROWS;
            const uint32_t cmask = (uint32_t)(__ballot(mine) >> (32 * half));
            uint32_t bits = rev & cmask;
            // zero the rows of the groups in use; owners note what the flush needs of them
            {
                float4* z = reinterpret_cast<float4*>(Rm);
                const int nz = ng * (2 * GRP * RSTRIDE / 4);
                for (int i = lane; i < nz; i += 64) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (mine) *reinterpret_cast<float4*>(&rowtab[(rank_h - sc0) * RT]) = make_float4(ra.x, ra.y, __uint_as_float(gid), 0.f);
            __builtin_amdgcn_wave_barrier();

            // ---- the walk: every lane through its own candidates of the sub-chunk, deepest first
            const auto pair = [&](const f32x4& A, const f32x4& B, const f32x4& K, bool act) {
                float cc[C];
                cc[0] = B[2]; cc[1] = B[3]; cc[2] = K[0];
                if constexpr (C > 3) cc[3] = K[1];
                const uint32_t rowoff = __float_as_uint(C == 3 ? K[1] : K[2]);
                const float dx = A[0] - pxf, dy = A[1] - pyf;
                const float power = pair_exp2_arg(A[2], A[3], B[0], dx, dy);
                const float G = __builtin_amdgcn_exp2f(power);
                const float alpha = fminf(ALPHA_MAX, B[1] * G);
                if (act && power <= 0.0f && alpha >= ALPHA_MIN) {
                    const float rinv = __builtin_amdgcn_rcpf(1.f - alpha);
                    T = T * rinv;
                    const float w = alpha * T;
                    float s = 0.f;
#pragma unroll
                    for (int ch = 0; ch < C; ch++) {
                        const float d = cc[ch] - acc[ch];
                        s = __builtin_fmaf(d, dp[ch], s);
                        acc[ch] = __builtin_fmaf(alpha, d, acc[ch]);
                    }
                    const float r = G * __builtin_fmaf(s, T, -(rinv * tf_bg));
                    float* const cell = reinterpret_cast<float*>(reinterpret_cast<char*>(Rm) + rowoff + rw_lane);
                    cell[0] = r;
                    cell[GRP * RSTRIDE] = w;
                }
            };
#if GSR_WALK_PIPE == 0
            {   // plain per-lane loop, loads left to the compiler
                const float* const rec_h = &rec[32 * half * RF];
                while (bits != 0u) {
                    const int i = __builtin_ctz(bits);
                    bits &= bits - 1u;
                    const f32x4* rp = reinterpret_cast<const f32x4*>(rec_h + i * RF);
                    pair(rp[0], rp[1], rp[2], true);
                }
            }
#else
            {
                // Uniform loop, explicit gathers: all three 16-byte reads of a record are requested together (left to the
                // compiler the colour words were read inside the live block: two LDS round trips per trip), and with
                // GSR_WALK_PIPE == 2 the record of trip k + 1 is requested before trip k is evaluated.  Two trips per
                // iteration on two register sets, so that no register with a load in flight crosses the loop's back edge
                // (see lds_request in gsr_bwd_util.h).  A lane without a candidate left gathers slot 31 and drops it.
                const uint32_t rec_addr = lds_byte_address(&rec[32 * half * RF]);
                const auto slot_of = [&](uint32_t b) { return rec_addr + (uint32_t)__builtin_ctz(b | 0x80000000u) * (uint32_t)(RF * 4); };
                SlotRegs<RV> ra_, rb_;
#if GSR_WALK_PIPE == 1
                while (__ballot(bits != 0u) != 0ull) {
                    lds_request<RV, 0>(ra_, slot_of(bits));
                    const bool act = bits != 0u;
                    bits &= bits - 1u;
                    lds_wait<0>(ra_);
                    pair(ra_.v[0], ra_.v[1], ra_.v[2], act);
                }
#else
                lds_request<RV, 0>(ra_, slot_of(bits));
                lds_wait<0>(ra_);
                while (true) {
                    {
                        const bool act = bits != 0u;
                        bits &= bits - 1u;
                        lds_request<RV, 0>(rb_, slot_of(bits));
                        pair(ra_.v[0], ra_.v[1], ra_.v[2], act);
                        lds_wait<0>(rb_);
                    }
                    if (__ballot(bits != 0u) == 0ull) break;
                    {
                        const bool act = bits != 0u;
                        bits &= bits - 1u;
                        lds_request<RV, 0>(ra_, slot_of(bits));
                        pair(rb_.v[0], rb_.v[1], rb_.v[2], act);
                        lds_wait<0>(ra_);
                    }
                    if (__ballot(bits != 0u) == 0ull) break;
                }
#endif
            }
#endif
            __builtin_amdgcn_wave_barrier();

            // ---- matrix pipe, one group of eight instances at a time (gsr_blend_bwd.hip has the layout notes)
            for (int g = 0; g < ng; g++) {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
                float av[16];
                {
                    const float4* pr = reinterpret_cast<const float4*>(&Rm[(g * 2 * GRP + col) * RSTRIDE + 16 * kap]);
#pragma unroll
                    for (int qd = 0; qd < 4; qd++) {
                        const float4 v = pr[qd];
                        av[4 * qd] = v.x; av[4 * qd + 1] = v.y; av[4 * qd + 2] = v.z; av[4 * qd + 3] = v.w;
                    }
                }
                uint32_t kMinusOneLo = 0x0000BF80u, kMinusOneHi = 0xBF800000u;   // bf16 pairs {-1, 0}, {0, -1}
                asm volatile("" : "+v"(kMinusOneLo), "+v"(kMinusOneHi));
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    u32x4 a_hi, a_mid, a_lo;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float x0 = av[8 * h + 2 * q], x1 = av[8 * h + 2 * q + 1];
                        a_hi[q] = bf16_pair(x0, x1);
                        const float y0 = bf16_rest_of(a_hi[q], x0, kMinusOneLo), y1 = bf16_rest_of(a_hi[q], x1, kMinusOneHi);
                        a_mid[q] = bf16_pair(y0, y1);
                        const float z0 = bf16_rest_of(a_mid[q], y0, kMinusOneLo), z1 = bf16_rest_of(a_mid[q], y1, kMinusOneHi);
                        a_lo[q] = bf16_pair(z0, z1);
                    }
                    const bf16x8 b = __builtin_bit_cast(bf16x8, Bp[h]);
                    f32x4& ac = h ? acc1 : acc0;
                    ac = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_lo), b, ac, 0, 0, 0);
                    ac = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_mid), b, ac, 0, 0, 0);
                    ac = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_hi), b, ac, 0, 0, 0);
                    if constexpr (C2 > 0) {
                        const bf16x8 b2 = __builtin_bit_cast(bf16x8, Bp2[h]);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_lo), b2, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_mid), b2, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_hi), b2, acc2, 0, 0, 0);
                    }
                }
                const auto split_sum = [](float v0, float v1, float v2, float v3, float& t0_, float& t1_, float& t2_, float& t3_) {
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
                };
                const float v0 = acc0[0] + acc1[0], v1 = acc0[1] + acc1[1], v2 = acc0[2] + acc1[2], v3 = acc0[3] + acc1[3];
                float t0_, t1_, t2_, t3_;
                split_sum(v0, v1, v2, v3, t0_, t1_, t2_, t3_);
                if constexpr (C2 > 0) {
                    float u0, u1, u2, u3;
                    split_sum(acc2[0], acc2[1], acc2[2], acc2[3], u0, u1, u2, u3);
                    acc2[0] = u0; acc2[1] = u1; acc2[2] = u2; acc2[3] = u3;
                }
                const bool spatial = kap < 2;
                acc0[0] = spatial ? v0 : t0_; acc0[1] = spatial ? v1 : t1_; acc0[2] = spatial ? v2 : t2_; acc0[3] = spatial ? v3 : t3_;
                const int left = cnt - g * GRP - wb_row0;     // instances of this group at or behind the lane's first row
                if (wb_take) {
                    float* const dst = &rowtab[(g * GRP + wb_row0) * RT + wb_col];
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (i < left) dst[i * RT] = acc0[i];
                }
                if constexpr (C2 > 0) {
                    if (wb_take2) {
                        float* const dst = &rowtab[(g * GRP + wb_row0) * RT + wb_col2];
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            if (i < left) dst[i * RT] = acc2[i];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();

            // ---- lane = instance of the sub-chunk: re-centre the spatial sums on the splat, then the row-major flush
            if (lane < cnt) {
                float* rw = &rowtab[lane * RT];
                const float m0 = rw[4], mx = rw[5], my = rw[6], mxx = rw[7], mxy = rw[8], myy = rw[9];
                const float X = rw[0] - (bx0 + 3.5f), Y = rw[1] - (by0 + 3.5f);
                rw[5] = X * m0 - mx;
                rw[6] = Y * m0 - my;
                rw[7] = (X * X) * m0 - 2.f * X * mx + mxx;
                rw[8] = (X * Y) * m0 - X * my - Y * mx + mxy;
                rw[9] = (Y * Y) * m0 - 2.f * Y * my + myy;
            }
            __builtin_amdgcn_wave_barrier();
            for (int idx = lane; idx < cnt * NM; idx += 64) {
                const int e = idx / NM, v = idx - e * NM;
                const size_t g = __float_as_uint(rowtab[e * RT + 2]);
                atomic_add_f32(grad_acc + g * GRAD_RS + v, rowtab[e * RT + 4 + v]);
            }
            __builtin_amdgcn_wave_barrier();   // the table and rowtab are rewritten by the next sub-chunk
        }
    }
}

template <int C, int ROWS>
static void launch_walk(int W, int H, int gx, int U, const float* bg, ImageState im, BinState b, const float* dL_dpix, float* grad_acc,
                        hipStream_t st)
{
    blend_bwd_walk_kernel<C, ROWS><<<4 * U, 64, 0, st>>>(W, H, gx, b.unit_info, b.snap, b.masks, b.point_list, b.rec_a, b.rec_b,
                                                        static_cast<const RecTail<C>*>(b.rec_c), bg, im.final_T, im.n_contrib,
                                                        dL_dpix, grad_acc);
}

// -> false if this variant does not cover the channel count (the caller falls back to the uniform pair loop)
bool launch_blend_bwd_walk(int rows, int C, int W, int H, int U, const float* bg, ImageState im, BinState b, const float* dL_dpix,
                           float* grad_acc, hipStream_t st)
{
    if (C != 3 && C != 4) return false;
    const Tiles t = tiles_of(W, H);
    if (U <= 0) return true;
    if (C == 3) {
        if (rows == 8) launch_walk<3, 8>(W, H, t.gx, U, bg, im, b, dL_dpix, grad_acc, st);
        else if (rows == 32) launch_walk<3, 32>(W, H, t.gx, U, bg, im, b, dL_dpix, grad_acc, st);
        else launch_walk<3, 16>(W, H, t.gx, U, bg, im, b, dL_dpix, grad_acc, st);
    } else {
        if (rows == 8) launch_walk<4, 8>(W, H, t.gx, U, bg, im, b, dL_dpix, grad_acc, st);
        else if (rows == 32) launch_walk<4, 32>(W, H, t.gx, U, bg, im, b, dL_dpix, grad_acc, st);
        else launch_walk<4, 16>(W, H, t.gx, U, bg, im, b, dL_dpix, grad_acc, st);
    }
    return true;
}

// the hook gsr_blend_bwd.hip calls in a -DGSR_BWD_VARIANT build; rows from GSR_BWD_WALK (8 / 16 / 32), 0 = the product's pair loop
bool launch_blend_bwd_variant(int C, int W, int H, int U, const float* bg, ImageState im, BinState b, const float* dL_dpix, float* grad_acc,
                              hipStream_t st)
{
    static const int rows = getenv("GSR_BWD_WALK") ? atoi(getenv("GSR_BWD_WALK")) : 16;
    return rows > 0 && launch_blend_bwd_walk(rows, C, W, H, U, bg, im, b, dL_dpix, grad_acc, st);
}

}  // namespace gsr
