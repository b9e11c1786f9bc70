// gsr_blend_bwd_pk.hip -- backward alpha compositing, PACKED-MATH variant of gsr_blend_bwd.hip (three channels).
//
// Same unit, head, keep-set, contraction, re-centring and flush as gsr_blend_bwd.hip, and the same per-pair arithmetic
// (DGR/cuda_rasterizer/backward.cu:464-556).  What differs is how the uniform pair loop is issued: kept instances are worked
// off TWO per trip, their queue slots interleaved field by field ({xA, xB}, {yA, yB}, ...) so that a ds_read_b128 lands
// them in even-aligned register pairs, and every operation of the pair test that is independent between the two instances
// -- two subtractions, three multiplies, two fused multiply-adds -- is one v_pk_{add,mul,fma}_f32 for both (4.8 cycles for two
// results against 2 x 2.7 .. 4.4; rounding identical to the scalar instructions, so the forward's and the backward's
// alpha >= 1/255 decisions still agree bit for bit).  In the live part, which is serial from instance to instance, the
// first two colour channels of an instance go through packed subtract / fma as well.
// Selected at run time by GSR_BWD_PK=1 (default off).  Measured in round 4 (NOTEBOOK.md "Round 4", profiles/r04_bwd_pk_counters.txt):
// 59.5 -> 55.9 M vector instructions per view, parity unchanged, but the pair of slots in flight costs 16 more registers than
// the scalar loop's: held at 80 registers (six waves per SIMD) it spills and loses 17 .. 61 us, at five waves it loses 11 us
// -- what five waves cost the scalar loop too.  The scalar loop stays the product.
//   GSR_PK_EARLY  1: next pair requested before this pair's test (40 slot registers), with the uniform j < cnt skip
//                 0: requested between test and live parts (26 slot registers), no skip (a request must not cross a branch)
//   GSR_PK_WAVES  waves per SIMD the register allocator is held to
#include "gsr_bwd_util.h"

namespace gsr {

#ifndef GSR_PK_EARLY
#define GSR_PK_EARLY 1
#endif
#ifndef GSR_PK_WAVES
#define GSR_PK_WAVES 5
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));

// queue slot of a PAIR of instances (A = even queue index, B = odd), 24 floats = six 16-byte vectors:
//   V0 xA xB yA yB | V1 gidA gidB - - (kept, never loaded by the loop) | V2 aA aB bA bB | V3 cA cB oA oB (exp2-domain conic,
//   opacity) | V4 posA posB kA0 kA1 (list positions as uint bits, colours) | V5 kB0 kB1 kA2 kB2
// once both instances' r and w are out, floats 6 .. 23 take their 2 x 9 moments (instance e: 6 + 9 e + m).
constexpr int PF = 24;

struct PairRegs { f32x4 v[5]; };   // V0, V2 .. V5
template <int OFF>
__device__ __forceinline__ void pair_request(PairRegs& r, uint32_t addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(r.v[0]) : "v"(addr), "n"(OFF) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(r.v[1]) : "v"(addr), "n"(OFF + 32) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(r.v[2]) : "v"(addr), "n"(OFF + 48) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(r.v[3]) : "v"(addr), "n"(OFF + 64) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(r.v[4]) : "v"(addr), "n"(OFF + 80) : "memory");
}
template <int PENDING>   // LDS operations issued after the request that may stay in flight (the previous pair's four table stores)
__device__ __forceinline__ void pair_wait(PairRegs& r)
{
    static_assert(PENDING == 0 || PENDING == 4, "");
    if constexpr (PENDING == 0)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4]) : : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4]) : : "memory");
}

template <int C>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GSR_PK_WAVES, 8)))
blend_bwd_pk_kernel(int W, int H, int gx, const uint4* __restrict__ unit_info, const float4* __restrict__ snap,
                    const uint2* __restrict__ masks, const uint32_t* __restrict__ point_list, const float4* __restrict__ rec_a,
                    const float4* __restrict__ rec_b, const RecTail<C>* __restrict__ rec_c, const float* __restrict__ bg,
                    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                    const float* __restrict__ dL_dpix, float* __restrict__ grad_acc)
{
    static_assert(C == 3 && GRP == 8, "the packed variant is written for three channels");
    constexpr int NM = 6 + C, SV = snap_vecs(C), QCAP = 16;
    __shared__ __attribute__((aligned(16))) float qf[(QCAP / 2) * PF];
    __shared__ __attribute__((aligned(16))) float Rm[2 * GRP * RSTRIDE];
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

    {
    const bool keep = ((kany >> (63 - lane)) & 1ull) != 0ull;
    const unsigned long long m = __ballot(keep);
    const int cnt_all = __popcll(m);
    for (int q0 = 0; q0 < cnt_all; q0 += QCAP) {
    const int cnt = min(cnt_all - q0, QCAP);
    const int slot = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) - q0;
    if (keep && slot >= 0 && slot < QCAP) {
        float* const ps = &qf[(slot >> 1) * PF];
        const int e = slot & 1;
        ps[0 + e] = ra.x; ps[2 + e] = ra.y; ps[4 + e] = __uint_as_float(gid);
        ps[8 + e] = ra.z; ps[10 + e] = ra.w; ps[12 + e] = rb.x; ps[14 + e] = rb.y;     // conic already in the exp2 domain
        ps[16 + e] = __uint_as_float((uint32_t)k);
        ps[e ? 20 : 18] = rb.z; ps[e ? 21 : 19] = rb.w; ps[22 + e] = rc.c[0];
        if ((cnt & 1) && slot == cnt - 1) {
            // an odd chunk: the B half of its last pair is this instance once more at a list position no pixel replays
            ps[1] = ra.x; ps[3] = ra.y; ps[5] = __uint_as_float(gid);
            ps[9] = ra.z; ps[11] = ra.w; ps[13] = rb.x; ps[15] = rb.y;
            ps[17] = __uint_as_float(0x7fffffffu);
            ps[20] = rb.z; ps[21] = rb.w; ps[23] = rc.c[0];
        }
    }
    __builtin_amdgcn_wave_barrier();

    const uint32_t q_base = lds_byte_address(qf);
    const uint32_t rw_addr = lds_byte_address(Rm) + 4u * (uint32_t)lane;
    const int wb_row0 = 4 * (kap & 1);
    const bool wb_take = kap < 2 ? col < 6 : (col >= 6 && col < 6 + 3 * C1 && (col % 3) == 0);
    const int wb_col = col >= 6 ? 6 + (col - 6) / 3 : col;
    const f32x2 pxx = {pxf, pxf}, pyy = {pyf, pyf};
    f32x2 acc01 = {acc[0], acc[1]};
    float acc2s = acc[2];
    for (int g0i = 0; g0i < cnt; g0i += GRP) {
        PairRegs nxt;
        uint32_t q_grp = q_base + (uint32_t)((g0i >> 1) * PF * 4);
        asm volatile("" : "+v"(q_grp));
        pair_request<0>(nxt, q_grp);
        static_for<GRP / 2>([&](auto PP) {
            constexpr int pp = decltype(PP)::value;
            const int j = g0i + 2 * pp;                      // (uniform) queue index of the pair's first instance
            float rA = 0.f, wA = 0.f, rB = 0.f, wB = 0.f;
            // (the next pair is requested between this pair's test and its live parts: by then 14 of the 20 slot
            // registers are dead, so current + next slot cost 26 registers, not 40 -- six waves per SIMD need <= 80 in all)
            pair_wait<(pp == 0 ? 0 : 4)>(nxt);
            const PairRegs cur = nxt;
#if GSR_PK_EARLY
            if constexpr (pp + 1 < GRP / 2) pair_request<(pp + 1) * PF * 4>(nxt, q_grp);
#endif
            bool liveA = false, liveB = false;
            f32x2 G = {0.f, 0.f};
            float alphaA = 0.f, alphaB = 0.f;
#if GSR_PK_EARLY
            if (j < cnt)
#endif
            {
#pragma clang fp contract(off)
                const f32x2 X = {cur.v[0][0], cur.v[0][1]}, Y = {cur.v[0][2], cur.v[0][3]};
                const f32x2 ca = {cur.v[1][0], cur.v[1][1]}, cb = {cur.v[1][2], cur.v[1][3]}, cc2 = {cur.v[2][0], cur.v[2][1]};
                const f32x2 op = {cur.v[2][2], cur.v[2][3]};
                const int posA = (int)__float_as_uint(cur.v[3][0]), posB = (int)__float_as_uint(cur.v[3][1]);
                // pair_exp2_arg for both instances: u = fma(a, dx, b dy); power = fma(c dy, dy, u dx) -- same operations, same
                // rounding as the scalar form the forward uses
                const f32x2 dx = X - pxx, dy = Y - pyy;
                const f32x2 u = __builtin_elementwise_fma(ca, dx, cb * dy);
                const f32x2 power = __builtin_elementwise_fma(cc2 * dy, dy, u * dx);
                G[0] = __builtin_amdgcn_exp2f(power[0]); G[1] = __builtin_amdgcn_exp2f(power[1]);
                const f32x2 og = op * G;
                alphaA = fminf(ALPHA_MAX, og[0]); alphaB = fminf(ALPHA_MAX, og[1]);
                liveA = (GSR_PK_EARLY || j < cnt) && posA < my_lim && power[0] <= 0.0f && alphaA >= ALPHA_MIN;
                liveB = (GSR_PK_EARLY || j < cnt) && posB < my_lim && power[1] <= 0.0f && alphaB >= ALPHA_MIN;
            }
            const float kA0 = cur.v[3][2], kA1 = cur.v[3][3], kA2 = cur.v[4][2], kB0 = cur.v[4][0], kB1 = cur.v[4][1], kB2 = cur.v[4][3];
#if !GSR_PK_EARLY
            if constexpr (pp + 1 < GRP / 2) pair_request<(pp + 1) * PF * 4>(nxt, q_grp);
#endif
            {
                if (liveA) {
                    const float rinv = __builtin_amdgcn_rcpf(1.f - alphaA);
                    T = T * rinv;
                    wA = alphaA * T;
                    const f32x2 k01 = {kA0, kA1};
                    const f32x2 d01 = k01 - acc01;
                    const float d2 = kA2 - acc2s;
                    const f32x2 al = {alphaA, alphaA};
                    acc01 = __builtin_elementwise_fma(al, d01, acc01);
                    acc2s = __builtin_fmaf(alphaA, d2, acc2s);
                    const float s = __builtin_fmaf(d2, dp[2], __builtin_fmaf(d01[1], dp[1], d01[0] * dp[0]));   // (order of the scalar form)
                    rA = G[0] * __builtin_fmaf(s, T, -(rinv * tf_bg));
                }
                if (liveB) {
                    const float rinv = __builtin_amdgcn_rcpf(1.f - alphaB);
                    T = T * rinv;
                    wB = alphaB * T;
                    const f32x2 k01 = {kB0, kB1};
                    const f32x2 d01 = k01 - acc01;
                    const float d2 = kB2 - acc2s;
                    const f32x2 al = {alphaB, alphaB};
                    acc01 = __builtin_elementwise_fma(al, d01, acc01);
                    acc2s = __builtin_fmaf(alphaB, d2, acc2s);
                    const float s = __builtin_fmaf(d2, dp[2], __builtin_fmaf(d01[1], dp[1], d01[0] * dp[0]));   // (order of the scalar form)
                    rB = G[1] * __builtin_fmaf(s, T, -(rinv * tf_bg));
                }
            }
            lds_store_b32<(2 * pp) * RSTRIDE * 4>(rw_addr, rA);
            lds_store_b32<(GRP + 2 * pp) * RSTRIDE * 4>(rw_addr, wA);
            lds_store_b32<(2 * pp + 1) * RSTRIDE * 4>(rw_addr, rB);
            lds_store_b32<(GRP + 2 * pp + 1) * RSTRIDE * 4>(rw_addr, wB);
        });
        __builtin_amdgcn_wave_barrier();
        // ---- matrix pipe: as in gsr_blend_bwd.hip (bf16 splits by v_dot2, six issues per group)
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        float av[16];
        {
            const float4* pr = reinterpret_cast<const float4*>(&Rm[col * RSTRIDE + 16 * kap]);
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                const float4 v = pr[qd];
                av[4 * qd] = v.x; av[4 * qd + 1] = v.y; av[4 * qd + 2] = v.z; av[4 * qd + 3] = v.w;
            }
        }
        uint32_t kMinusOneLo = 0x0000BF80u, kMinusOneHi = 0xBF800000u;
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
        }
        const float v0 = acc0[0] + acc1[0], v1 = acc0[1] + acc1[1], v2 = acc0[2] + acc1[2], v3 = acc0[3] + acc1[3];
        float t0_, t1_, t2_, t3_;
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
        const bool spatial = kap < 2;
        const float o0 = spatial ? v0 : t0_, o1 = spatial ? v1 : t1_, o2 = spatial ? v2 : t2_, o3 = spatial ? v3 : t3_;
        // D rows 4 (kap & 1) + i = instances g0i + wb_row0 + i: pair (g0i + wb_row0) / 2 + i / 2, half i & 1
        if (wb_take) {
            float* const dst = &qf[((g0i + wb_row0) >> 1) * PF + 6 + wb_col];
            const int left = cnt - g0i - wb_row0;
            if (0 < left) dst[0] = o0;
            if (1 < left) dst[9] = o1;
            if (2 < left) dst[PF] = o2;
            if (3 < left) dst[PF + 9] = o3;
        }
        __builtin_amdgcn_wave_barrier();
    }
    acc[0] = acc01[0]; acc[1] = acc01[1]; acc[2] = acc2s;

    // ---- lane = queued instance: re-centre the spatial sums on the splat, then the row-major flush
    if (lane < cnt) {
        float* const pb = &qf[(lane >> 1) * PF];
        float* const rw = pb + 6 + 9 * (lane & 1);
        const float m0 = rw[0], mx = rw[1], my = rw[2], mxx = rw[3], mxy = rw[4], myy = rw[5];
        const float X = pb[lane & 1] - (bx0 + 3.5f), Y = pb[2 + (lane & 1)] - (by0 + 3.5f);
        rw[1] = X * m0 - mx;
        rw[2] = Y * m0 - my;
        rw[3] = (X * X) * m0 - 2.f * X * mx + mxx;
        rw[4] = (X * Y) * m0 - X * my - Y * mx + mxy;
        rw[5] = (Y * Y) * m0 - 2.f * Y * my + myy;
    }
    __builtin_amdgcn_wave_barrier();
    for (int idx = lane; idx < cnt * NM; idx += 64) {
        const int e = idx / NM, v = idx - e * NM;
        const float* const pb = &qf[(e >> 1) * PF];
        const size_t g = __float_as_uint(pb[4 + (e & 1)]);
        atomic_add_f32(grad_acc + g * GRAD_RS + v, pb[6 + 9 * (e & 1) + v]);
    }
    __builtin_amdgcn_wave_barrier();
    }   // chunks of the batch
    }
}

// -> false if this variant does not cover the channel count (the caller falls back to the scalar pair loop)
bool launch_blend_bwd_pk(int C, int W, int H, int U, const float* bg, ImageState im, BinState b, const float* dL_dpix, float* grad_acc,
                         hipStream_t st)
{
    if (C != 3) return false;
    if (U <= 0) return true;
    const Tiles t = tiles_of(W, H);
    blend_bwd_pk_kernel<3><<<4 * U, 64, 0, st>>>(W, H, t.gx, b.unit_info, b.snap, b.masks, b.point_list, b.rec_a, b.rec_b,
                                               static_cast<const RecTail<3>*>(b.rec_c), bg, im.final_T, im.n_contrib, dL_dpix, grad_acc);
    return true;
}

// the hook gsr_blend_bwd.hip calls in a -DGSR_BWD_VARIANT build (GSR_BWD_PK=0: the product's pair loop)
bool launch_blend_bwd_variant(int C, int W, int H, int U, const float* bg, ImageState im, BinState b, const float* dL_dpix, float* grad_acc,
                              hipStream_t st)
{
    static const int pk = getenv("GSR_BWD_PK") ? atoi(getenv("GSR_BWD_PK")) : 1;
    return pk > 0 && launch_blend_bwd_pk(C, W, H, U, bg, im, b, dL_dpix, grad_acc, st);
}

}  // namespace gsr
