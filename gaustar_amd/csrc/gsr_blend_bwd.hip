// gsr_blend_bwd.hip -- backward alpha compositing: per-Gaussian sums of the per-(pixel, Gaussian) terms, for three channels (the
// reference's NUM_CHANNELS), four (GauSTAR's RGB + depth-as-colour renders in one pass) and six (two targets sharing geometry).
//
// Per-pair arithmetic is the reference's renderCUDA backward (DGR/cuda_rasterizer/backward.cu:399-557; SURVEY.md section 9 item
// 10): back-to-front replay, T recovered by division, accum_rec recurrence, background term with T_final / (1 - alpha), the 0.99
// alpha clamp passing gradient as if unclamped.  What differs is the decomposition and how the per-pair terms reach memory.  The
// reference runs one block per tile over the whole list and issues 9 global float atomicAdds per contributing (pixel, Gaussian)
// pair (backward.cu:523, :545-554).  Here
//
//  * the work unit is a (tile, SEGMENT of 64 list positions, 8x8 pixel block) triple, one wave64 each.  A pixel whose last
//    contributor lies beyond the segment starts from the forward pass's snapshot at the segment's far boundary: T = T_snap,
//    accum_rec = (C_final - C_snap) / T_snap -- exactly the state the reference's back-to-front recurrence has at that list
//    position; a pixel that ends inside the segment starts from (T_final, 0) like the reference; a pixel that ended before it is
//    idle.  Units have bounded size, so the dispatcher can balance them and nothing carries a 1 600-instance serial chain;
//  * which instances of the segment a block needs comes from the forward's per-pixel candidate words (gsr_mask.h), masked with
//    "positions this pixel replays".  The unit's 64 records arrive as three contiguous rows the forward left in list order (lane i
//    takes position 63 - i, together with every other load of the unit's head); no geometric test is repeated here;
//  * (round 6; rounds 1-5: one kept instance on the block's 64 pixels per trip -- 10 of 64 lanes live, 28 dependent trips per
//    wave, w and r parked in an LDS table and contracted over the pixels on the matrix pipe: tools/variants/gsr_blend_bwd_uniform.hip)
//    the wave walks the block 4x4 SUB-BLOCK by sub-block, and a trip evaluates FOUR consecutive kept instances of that sub-block
//    on its 16 pixels:
//
//        lane = 16 rho + 4 j + kappa:   (kappa, rho) = the pixel inside the 4x4 sub-block, j = instance of the trip (0 deepest)
//
//    so a DPP row is one pixel row, a DPP bank (four lanes) one instance, and the four instances of a pixel sit four lanes apart;
//  * the per-pixel recurrences along the list -- T' = T / (1 - alpha) and the projected accum_rec A' = (1 - alpha) A + alpha k,
//    k = c . dL_dpix (dL_dalpha only ever needs accum_rec through its dot product with dL_dpix, and the recurrence is linear) --
//    are inclusive scans of affine maps over j: two fused DPP steps (row_shr:4, row_shr:8; a lane without a source keeps its
//    value, so no selects), the state carried from trip to trip through lane j = 3 by row_ror:4 into lane j = 0 -- nine vector
//    instructions for both recurrences and the exclusive value dL_dalpha needs, nothing through LDS;
//  * the kept set is the OR of the masked candidate words over the sub-block's 16 pixels (a pair touches 1.8 of a block's 4
//    sub-blocks: 51 (instance, sub-block) pairs per unit-block, 13.6 trips, where the uniform loop makes 28.2);
//  * the 6 + C moments  sum_px w dL_dpix_c,  sum_px r {1, dx, dy, dx^2, dx dy, dy^2}  (w = alpha T, r = G dL_dalpha, dx = x_splat -
//    x_pixel: dL_dcolor, dL_dopacity, dL_dmean2D and dL_dconic are fixed per-Gaussian linear maps of them, applied once per
//    Gaussian in geom_bwd) are reduced over the 16 pixels in registers: v_permlane32_swap / v_permlane16_swap fold two values per
//    swap across the rows, two quad_perm adds fold the bank, and a plain LDS read + write adds them into the instances' moment
//    records (ds_add_f32 costs the CU's LDS pipe ~190 cycles per wave instruction on this chip).  Everything is f32: no r|w table,
//    no bf16 splits, no matrix instructions -- the four-channel render's blue and depth gradients are exact again (rounds 4-5
//    packed them into two bf16 columns);
//  * the table is flushed ROW-MAJOR: one atomic instruction covers the consecutive floats of ~7 packed 48-byte records
//    grad_acc[gaussian][12] = {sum r, sum r dx, sum r dy, sum r dx^2, sum r dx dy, sum r dy^2, c_0 .. c_{C-1}}, so the memory
//    pipeline merges lanes per cache line (1.5 M atomic requests per 1080p view instead of 8 M).
// Rounding: T and A come out of re-associated products (tree order within a trip); the alpha decisions (power <= 0, alpha >=
// 1/255, position < last contributor) are the forward's bit for bit.
// Measured against the uniform loop (profiles/r06_bwd_quad_counters.txt, config C): 51.7 M vector instructions per view instead of
// 55.2 M, waiting on LDS 8.8 M wave-cycles instead of 51.2 M, kernel 116.7 -> 112.6 us (three channels), four channels - 4 us.
#include "gsr_bwd_util.h"

namespace gsr {

#ifndef GSR_QUAD_WAVES
#define GSR_QUAD_WAVES 7
#endif
#ifndef GSR_QUAD_WAVES4
#define GSR_QUAD_WAVES4 7   // four channels: 72 registers without scratch -- once the lane-derived constants of the trips are computed inside
                            // the sub-block loop (hoisted to the kernel's head they lived through its register peak: 75).  SCRATCH is poison
                            // for this launch: seven waves with one float4 of the head spilled ran 5 us SLOWER than six waves without (the
                            // residency fit says seven waves are worth 5.9 us: a kernel of 54 640 one-wave workgroups that touches scratch at
                            // all pays ~11 us; three channels at eight waves with eight registers spilled outside the trip loop: + 20 us).
                            // Parking the head's record rows in LDS, or loading them late, did not remove the spill for less
#endif
#ifndef GSR_QUAD_WAVES6
#define GSR_QUAD_WAVES6 5   // six channels: 96 registers
#endif
#ifndef GSR_QUAD_EXP
#define GSR_QUAD_EXP 0      // (instruction accounting builds: 1 = no trips, 2 = no trips and no per-chunk work)
#endif

#ifndef GSR_QUAD_Q
#define GSR_QUAD_Q 32     // (16: +4 us at three channels -- twice the per-chunk set-up and more half-empty trips)
#endif
#ifndef GSR_QUAD_Q6
#define GSR_QUAD_Q6 32    // six channels: 64-byte records and 48 bytes of pixel state = 7.5 KB, five waves per SIMD -- which its 96
                          // registers allow anyway (in-process A/B, step time against the uniform loop's 0.291 ms: six waves at 80 registers
                          // with ten spilled + 20 us, five waves with chunks of 16 - 8 us, five waves with chunks of 32 - 18 us)
#endif
// LDS layout of one wave, per channel count
template <int C> struct QuadLds {
    static constexpr int Q = C == 6 ? GSR_QUAD_Q6 : GSR_QUAD_Q;   // kept instances per chunk (a sub-block's list of a chunk: at most Q / 4 trips)
    static_assert(Q == 16 || Q == 32, "whole trips");
    static constexpr int REC_B = C <= 4 ? 48 : 64;   // record: {x, y, a', b'} {c', opacity, position, gaussian} {c0 .. c3} [{c4, c5, -, -}]
    static constexpr int MOM_B = C <= 4 ? 48 : 64;   // moment record: 6 + C sums, then pad words (the lanes without a sum add into those)
    static constexpr int STG_B = C <= 4 ? 32 : 48;   // pixel state: {T, A, last position, T_final bg.dL_dpix} {dL_dpix 0..3} [{4, 5, -, -}]
    static constexpr int REC = 0;                    // Q + 1 records (the last: a null record for the lanes past a list's end)
    static constexpr int MOM = REC + (Q + 1) * REC_B;
    static constexpr int STG = MOM + (Q + 1) * MOM_B;
    static constexpr int LST = STG + 64 * STG_B;     // 4 sub-blocks x 4 instances-of-a-trip x Q / 4 trips: record offset / 16 of the (4 t + j)-th kept instance
    static constexpr int ORW = LST + 4 * Q + 16;     // 4 x u64: the sub-blocks' kept words (16 bytes of "null record" behind the lists: look-ahead reads)
    static constexpr int TOTAL = ORW + 32;
    static_assert(STG % 16 == 0 && MOM % 16 == 0 && LST % 16 == 0 && ORW % 16 == 0, "16-byte accesses");
    static_assert((REC_B / 16) * Q < 256 && 6 + C + 2 <= MOM_B / 4, "list bytes; two pad words per moment record");
};

typedef uint32_t u32x2q __attribute__((ext_vector_type(2)));

// a + b with the upper 32 lanes of `a` and the lower 32 of `b` exchanged first:  lanes 0..31 = a.lo + a.hi, 32..63 = b.lo + b.hi
__device__ __forceinline__ float fold32(float a, float b)
{
    const u32x2q r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// the same across neighbouring rows of 16:  rows 0, 2 = a.r0 + a.r1, a.r2 + a.r3;  rows 1, 3 = b.r0 + b.r1, b.r2 + b.r3
__device__ __forceinline__ float fold16(float a, float b)
{
    const u32x2q r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum over the four lanes of a bank (quad), in every lane
__device__ __forceinline__ float fold_quad(float v)
{
    float t = v + __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    return t + __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(t), 0x4E, 0xf, 0xf, true));      // quad_perm [2,3,0,1]
}

template <int C>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(C == 6 ? GSR_QUAD_WAVES6 : C == 4 ? GSR_QUAD_WAVES4 : GSR_QUAD_WAVES, 8)))
blend_bwd_kernel(int W, int H, int gx, const uint4* __restrict__ unit_info, const float4* __restrict__ snap,
                      const uint2* __restrict__ masks, const uint32_t* __restrict__ point_list, const float4* __restrict__ rec_a,
                      const float4* __restrict__ rec_b, const RecTail<C>* __restrict__ rec_c, const float* __restrict__ bg,
                      const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                      const float* __restrict__ dL_dpix, float* __restrict__ grad_acc, uint64_t* __restrict__ trace,
                      const uint32_t* __restrict__ order)
{
    using L = QuadLds<C>;
    const uint64_t t_start = trace ? wall_clock64() : 0;
    constexpr int Q = L::Q, REC_B = L::REC_B, MOM_B = L::MOM_B, STG_B = L::STG_B, REC = L::REC, MOM = L::MOM, STG = L::STG, LST = L::LST,
                  ORW = L::ORW, TOTAL = L::TOTAL, RV = REC_B / 16;
    static_assert(C == 3 || C == 4 || C == 6, "channel counts of the C ABI");
    constexpr int SV = snap_vecs(C), NM = 6 + C;
    __shared__ __attribute__((aligned(16))) unsigned char lds[TOTAL];
    // ---- the product's head, verbatim (gsr_blend_bwd.hip): XCD-aware unit map, one scalar load for the tile, every vector load
    // of the unit in flight before the first wait
    const uint32_t n_units = gridDim.x >> 2;
    const uint32_t xcd = blockIdx.x & 7u, slot_id = blockIdx.x >> 3;
    const uint32_t grp = slot_id >> 2;
    uint32_t unit = (grp >> 3) * 64u + xcd * 8u + (grp & 7u);
    uint32_t wave_sel = slot_id & 3u;
    const uint32_t full = (n_units >> 6) << 6;
    if (blockIdx.x >= full * 4u) { unit = blockIdx.x >> 2; wave_sel = blockIdx.x & 3u; }
#ifdef GSR_EXP_LIVE
    // (experiment build, tools/dead_exit_exp.py: the debug pointer is not a launch order but a table of live bits per unit -- word
    // [n_units] = 0: every wave that finds work records its bit; 1: a wave whose bit is clear leaves right here, before any vector load --
    // what a per-(unit, block) flag written by the forward would buy, without touching the forward)
    uint32_t* const live_tab = const_cast<uint32_t*>(order);
    const bool live_use = live_tab != nullptr && live_tab[n_units] == 1u;
    if (live_use && ((live_tab[unit] >> wave_sel) & 1u) == 0u) return;
#else
    // (a launch ORDER of the units, when there is one: position in the dispatch sequence -> unit; gsr_debug_set_bwd_order)
    if (order != nullptr) unit = order[unit];
#endif
    const uint4 info = unit_info[unit];
    const int tile = (int)info.x;
    const uint32_t list0 = info.y;
    const int n = (int)info.z;
    const uint32_t unit0 = info.w;
    const int s0 = (int)(unit - unit0) * 64;
    if (s0 >= n) return;
    const int wave = (int)wave_sel, lane = threadIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int sx = tx * TILE + (wave & 1) * SUB, sy = ty * TILE + (wave >> 1) * SUB;
    const int px = sx + (lane & 7), py = sy + (lane >> 3);
    const bool inside = px < W && py < H;
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
        if constexpr (C == 4) {
            // (four channels: T + four colours are five floats of an eight-float slot -- a 16-byte and a 4-byte load instead of two
            // 16-byte ones: the three pad registers of each of the two snapshots in flight were what pushed the kernel over 72
            // registers, and a kernel that touches scratch at all loses ~20 us of this launch)
            const float4 t = at32(base_u, (uint32_t)(pidx * SV) * 16u);
            const float c3 = at32(reinterpret_cast<const float*>(base_u), (uint32_t)(pidx * SV + 1) * 16u);
            T_ = t.x; c_[0] = t.y; c_[1] = t.z; c_[2] = t.w; c_[C - 1] = c3;
            return;
        }
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
    const int k = s0 + 63 - lane;                                  // lane l holds list position s0 + 63 - l (back to front)
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
    // ---- the kept words of the four 4x4 sub-blocks: one same-address LDS OR per sub-block (pixel lane l = 8 y + x lies in
    // sub-block 2 (y >> 2) + (x >> 2)), read back by every lane
    const uint32_t lds0 = lds_byte_address(lds);
    unsigned long long ks[4];
    {
        const uint32_t sub = (uint32_t)(((lane >> 5) << 1) | ((lane >> 2) & 1));
        const uint32_t ad = lds0 + ORW + 8u * sub;
        const u32x2q zero = {0u, 0u}, mine = {word.x, word.y};
        u32x4 a01, a23;
        asm volatile("ds_write_b64 %2, %3\n\t"
                     "ds_or_b64 %2, %4\n\t"
                     "ds_read_b128 %0, %5\n\t"
                     "ds_read_b128 %1, %5 offset:16\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(a01), "=&v"(a23) : "v"(ad), "v"(zero), "v"(mine), "v"(lds0 + ORW) : "memory");
        const auto rfl = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
        ks[0] = ((unsigned long long)rfl(a01[1]) << 32) | rfl(a01[0]);
        ks[1] = ((unsigned long long)rfl(a01[3]) << 32) | rfl(a01[2]);
        ks[2] = ((unsigned long long)rfl(a23[1]) << 32) | rfl(a23[0]);
        ks[3] = ((unsigned long long)rfl(a23[3]) << 32) | rfl(a23[2]);
    }
    const unsigned long long kany = ks[0] | ks[1] | ks[2] | ks[3];
    if (kany == 0ull) return;
#ifdef GSR_EXP_LIVE
    if (live_tab != nullptr && !live_use && lane == 0) atomicOr(&live_tab[unit], 1u << wave);
#endif
    // ---- the pixels' start state, parked per pixel lane; the trips read it in their own lane order
    {
        float4* st = reinterpret_cast<float4*>(lds + STG + STG_B * lane);
        st[0] = make_float4(T, accd, __uint_as_float((uint32_t)my_lim), tf_bg);
        st[1] = make_float4(dp[0], dp[1], dp[2], C > 3 ? dp[3 < C ? 3 : 0] : 0.f);
        if constexpr (C == 6) st[2] = make_float4(dp[4 < C ? 4 : 0], dp[5 < C ? 5 : 0], 0.f, 0.f);
    }
    const bool keep = ((kany >> (63 - lane)) & 1ull) != 0ull;
    const unsigned long long m = __ballot(keep);
    const int cnt_all = __popcll(m);
    // (the unit's records arrived with the head's loads and feed the FIRST chunk; a later chunk fetches its lanes' records
    // again -- they are in L2 -- instead of keeping ten registers alive across the trips)
    float4 ra_ = ra, rb_ = rb;
    float rcc[4];   // colour channels 2 .. 5 (0 where the render has fewer)
#pragma unroll
    for (int i = 0; i < 4; i++) rcc[i] = i < C - 2 ? rc.c[i < C - 2 ? i : 0] : 0.f;
    uint32_t gid_ = gid;
    for (int q0 = 0; q0 < (GSR_QUAD_EXP == 2 ? 0 : cnt_all); q0 += Q) {
        const int cnt = min(cnt_all - q0, Q);
        const int slot = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) - q0;
        const bool mine = ((m >> lane) & 1ull) != 0ull && slot >= 0 && slot < Q;
        if (q0 > 0) {
            const uint32_t kl2 = (uint32_t)(min(s0 + 63 - lane, n - 1) - s0);
            ra_ = at32(rec_a + list0 + s0, kl2 * 16u);
            rb_ = at32(rec_b + list0 + s0, kl2 * 16u);
            const RecTail<C> rc2 = at32(rec_c + list0 + s0, kl2 * (uint32_t)sizeof(RecTail<C>));
#pragma unroll
            for (int i = 0; i < C - 2; i++) rcc[i] = rc2.c[i];
            gid_ = at32(point_list + list0 + s0, kl2 * 4u);
        }
        // records of the chunk, a null record behind them, moment records cleared, lists reset to "the null record"
        if (mine) {
            float4* r = reinterpret_cast<float4*>(lds + REC + slot * REC_B);
            r[0] = make_float4(ra_.x, ra_.y, ra_.z, ra_.w);
            r[1] = make_float4(rb_.x, rb_.y, __uint_as_float((uint32_t)(s0 + 63 - lane)), __uint_as_float(gid_));
            r[2] = make_float4(rb_.z, rb_.w, rcc[0], rcc[1]);
            if constexpr (C == 6) r[3] = make_float4(rcc[2], rcc[3], 0.f, 0.f);
        }
        if (lane == 0) {
            float4* r = reinterpret_cast<float4*>(lds + REC + Q * REC_B);
            r[0] = make_float4(0.f, 0.f, 0.f, 0.f);
            r[1] = make_float4(0.f, 0.f, __uint_as_float(0x7fffffffu), 0.f);
            r[2] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (C == 6) r[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int i = lane; i < (Q + 1) * (MOM_B / 16); i += 64) reinterpret_cast<float4*>(lds + MOM)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < Q + 4) reinterpret_cast<uint32_t*>(lds + LST)[lane] = 0x01010101u * (uint32_t)(RV * Q);   // (bytes: record offset / 16)
        __builtin_amdgcn_wave_barrier();
        int cnt_s[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const bool in_s = mine && ((ks[s] >> (63 - lane)) & 1ull) != 0ull;
            const unsigned long long ms = __ballot(in_s);
            cnt_s[s] = __popcll(ms);
            if (in_s) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ms >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms, 0u));
                lds[LST + s * Q + (rank & 3) * (Q / 4) + (rank >> 2)] = (unsigned char)(RV * slot);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int s = 0; s < (GSR_QUAD_EXP ? 0 : 4); s++) {
            const int ns = s == 0 ? cnt_s[0] : s == 1 ? cnt_s[1] : s == 2 ? cnt_s[2] : cnt_s[3];
            if (ns == 0) continue;
            // trip lane order: lane = 16 rho + 4 j + kappa.  (Derived HERE, from a copy of the lane id the compiler cannot see through:
            // computed once per wave these constants -- and what follows from them -- were hoisted to the kernel's head and lived through
            // its register peak, the loads of the unit in flight)
            // -- four channels only: they then fit the 72 registers of seven waves (75 hoisted); three and six channels have the room
            // and run 2 - 4 us faster with the constants hoisted
            int lane_o = lane;
            if constexpr (C == 4) asm volatile("" : "+v"(lane_o));
            const int rho = lane_o >> 4, jj = (lane_o >> 2) & 3, kap = lane_o & 3;
            // which of a trip's 64 lanes carry a finished sum after the fold (see below) and which moment it is
            // (the other lanes add garbage to the pad words of their instance's moment record: no branch around the store)
            const int mom_idx = kap == 0 ? ((rho & 1) * 2 + (rho >> 1))          // rows {0, 2, 1, 3} of fold A: moments 0, 2, 1, 3
                              : kap == 1 ? 4 + ((rho & 1) * 2 + (rho >> 1))      // fold B: moments 4, 6, 5, 7
                              : (kap == 2 && C == 6) ? 8 + ((rho & 1) * 2 + (rho >> 1))   // fold C of six channels: rows m8, m10, m9, m11
                              : (kap == 2 && rho == 0) ? 8
                              : (kap == 2 && rho == 2 && C == 4) ? 9       // fold C of four channels: rows m8, m8, m9, m9
                                                                 : NM + (lane_o & 1);
            // this lane's pixel of sub-block s and its state
            const int bx = s & 1, by = s >> 1;
            const int pl = 8 * (4 * by + rho) + 4 * bx + kap;
            const float4* st = reinterpret_cast<const float4*>(lds + STG + STG_B * pl);
            const float4 s_a = st[0], s_b = st[1];
            float d4 = 0.f, d5 = 0.f;
            if constexpr (C == 6) { const float4 s_c = st[2]; d4 = s_c.x; d5 = s_c.y; }
            float Tp = s_a.x, Ap = s_a.y;
            const int lim = (int)__float_as_uint(s_a.z);
            const float tfbg = s_a.w;
            const float d0 = s_b.x, d1 = s_b.y, d2 = s_b.z, d3 = s_b.w;
            const float pxf = (float)(sx + 4 * bx + kap), pyf = (float)(sy + 4 * by + rho);
            const int ntrip = (ns + 3) >> 2;
            SlotRegs<RV> rec;
            // this lane's list of the sub-block: byte t = slot of its instance in trip t, read two trips ahead
            const uint32_t lst_ad = lds0 + LST + (uint32_t)(s * Q + jj * (Q / 4));
            uint32_t idx_nxt;
            asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(idx_nxt) : "v"(lst_ad) : "memory");
            uint32_t ad = (idx_nxt << 4) + (lds0 + REC);
            lds_request<RV, 0>(rec, ad);
            asm volatile("ds_read_u8 %0, %1 offset:1" : "=v"(idx_nxt) : "v"(lst_ad) : "memory");
#pragma unroll 1
            for (int t = 0; t < ntrip; t++) {
                // one trip: four instances; their records are consumed first thing, and the next trip's are requested into the
                // same registers before the scans and folds (forty instructions that hide the round trip)
                lds_wait<0>(rec);
                asm volatile("" : "+v"(idx_nxt));   // (landed with the records)
                const uint32_t mom_ad = ad + (uint32_t)(MOM - REC) + 4u * (uint32_t)mom_idx;
                // (the moment this lane will add to, requested now: ds_add_f32 costs the CU's LDS pipe ~190 cycles per wave
                // instruction on this chip -- NOTEBOOK round 2 --; the wave owns its LDS and a trip's lanes hit distinct words, so a
                // plain read + add + write does the same)
                float mom_old;
                asm volatile("ds_read_b32 %0, %1" : "=v"(mom_old) : "v"(mom_ad) : "memory");
                const float dx = rec.v[0][0] - pxf, dy = rec.v[0][1] - pyf;
                const float power = pair_exp2_arg(rec.v[0][2], rec.v[0][3], rec.v[1][0], dx, dy);
                const float G = __builtin_amdgcn_exp2f(power);
                const float alpha = fminf(ALPHA_MAX, rec.v[1][1] * G);
                const int pos = (int)__float_as_uint(rec.v[1][2]);
                const bool live = pos < lim && power <= 0.0f && alpha >= ALPHA_MIN;
                float kd = rec.v[2][0] * d0;
                kd = __builtin_fmaf(rec.v[2][1], d1, kd);
                kd = __builtin_fmaf(rec.v[2][2], d2, kd);
                if constexpr (C > 3) kd = __builtin_fmaf(rec.v[2][3], d3, kd);
                if constexpr (C == 6) { kd = __builtin_fmaf(rec.v[RV - 1][0], d4, kd); kd = __builtin_fmaf(rec.v[RV - 1][1], d5, kd); }
                // (the next trip's records; past the list's end the bytes name the null record)
                ad = (idx_nxt << 4) + (lds0 + REC);
                lds_request<RV, 0>(rec, ad);   // (everything the trip needs of the old records has been computed above)
                asm volatile("ds_read_u8 %0, %1" : "=v"(idx_nxt) : "v"(lst_ad + (uint32_t)(t + 2)) : "memory");
                const float ae = live ? alpha : 0.0f;
                float A = 1.0f - ae;                            // the pair's map on accum_rec . dL_dpix: A' = A x + B
                const float rinv = __builtin_amdgcn_rcpf(A);
                float B = ae * kd;
                float x = rinv;                                 // becomes T in front of the pair (after the division)
                float sdl;                                      // k - accum_rec . dL_dpix behind the pair
                asm volatile("s_nop 1\n\t"
                             "v_mul_f32_dpp %0, %4, %0 row_ror:4 row_mask:0xf bank_mask:0x1\n\t"
                             "v_fmac_f32_dpp %2, %5, %1 row_ror:4 row_mask:0xf bank_mask:0x1\n\t"
                             "s_nop 0\n\t"
                             "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f32_dpp %2, %2, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 0\n\t"
                             "v_fmac_f32_dpp %2, %2, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                             "v_subrev_f32_dpp %3, %5, %6 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 0\n\t"
                             "v_subrev_f32_dpp %3, %2, %6 row_shr:4 row_mask:0xf bank_mask:0xf"
                             : "+v"(x), "+v"(A), "+v"(B), "=&v"(sdl) : "v"(Tp), "v"(Ap), "v"(kd));
                Tp = x;
                Ap = B;
                const float w = ae * x;
                const float u = __builtin_fmaf(sdl, x, -(rinv * tfbg));
                const float r = live ? G * u : 0.0f;
                // the nine per-pair terms, folded over the sub-block's 16 pixels
                // (all nine products first, each in a register of its own: the swaps then run back to back on dead values --
                // no copies, and the wait states a swap needs behind the instruction that wrote its operand are already there)
                typedef float f32x2p __attribute__((ext_vector_type(2)));
                const f32x2p dxy = {dx, dy}, d01 = {d0, d1};
                const f32x2p p12 = r * dxy;                      // v_pk_mul_f32: two products per instruction
                const f32x2p p34 = p12[0] * dxy;
                const float p5 = p12[1] * dy;
                const f32x2p p67 = w * d01;
                const float p8 = w * d2;
                float p8b = C > 3 ? w * d3 : p8;                 // (three channels: the ninth value folds with itself)
                if constexpr (C == 3) asm volatile("" : "+v"(p8b));
                const float f01 = fold32(r, p12[0]), f23 = fold32(p12[1], p34[0]), f45 = fold32(p34[1], p5), f67 = fold32(p67[0], p67[1]),
                            f88 = fold32(p8, p8b);
                float f88b = f88;
                if constexpr (C == 6) f88b = fold32(w * d4, w * d5);   // (six channels: rows m10 | m11)
                else asm volatile("" : "+v"(f88b));
                const float gA = fold16(f01, f23);               // rows: m0, m2, m1, m3
                const float gB = fold16(f45, f67);               // rows: m4, m6, m5, m7
                const float gC = fold16(f88, f88b);              // rows: m8 everywhere (four channels: m8, m8, m9, m9; six: m8, m10, m9, m11)
                const float hA = fold_quad(gA), hB = fold_quad(gB), hC = fold_quad(gC);
                const float val = kap == 0 ? hA : kap == 1 ? hB : hC;
                // (the record reads and the list byte were issued behind it)
                if constexpr (RV == 3) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(mom_old) : : "memory");
                else asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(mom_old) : : "memory");
                const float mom_new = mom_old + val;
                asm volatile("ds_write_b32 %0, %1" : : "v"(mom_ad), "v"(mom_new) : "memory");
            }
            lds_wait<0>(rec);
            asm volatile("" : "+v"(idx_nxt));
            // (every request has been waited for: nothing stays in flight into a dead register)
            // the pixels' state behind this sub-block's instances of the chunk (lane j = 3 holds it)
            if (jj == 3) {
                float2* stw = reinterpret_cast<float2*>(lds + STG + STG_B * pl);
                *stw = make_float2(Tp, Ap);
            }
            __builtin_amdgcn_wave_barrier();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
        __builtin_amdgcn_wave_barrier();
        // flush: lanes walk the (instance, moment) table row-major (one atomic instruction covers several packed records)
        for (int idx = lane; idx < cnt * NM; idx += 64) {
            const int e = idx / NM, v = idx - e * NM;
            const size_t g = __float_as_uint(reinterpret_cast<const float*>(lds + REC + e * REC_B)[7]);
            atomic_add_f32(grad_acc + g * GRAD_RS + v, reinterpret_cast<const float*>(lds + MOM + e * MOM_B)[v]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (trace && lane == 0) {   // (devtools: a unit's start and the end of its last wave -- monotone clock, max via atomic)
        if (wave == 0) trace[2 * unit] = t_start;
        atomicMax((unsigned long long*)&trace[2 * unit + 1], (unsigned long long)wall_clock64());
    }
}

// Experiment builds only (python -m gaustar_amd.build --variant NAME --with tools/variants/<file>.hip ...): the extra source
// defines launch_blend_bwd_variant and gets the first go at the launch.
#ifdef GSR_BWD_VARIANT
bool launch_blend_bwd_variant(int C, int W, int H, int U, const float* bg, ImageState im, BinState b, const float* dL_dpix, float* grad_acc,
                              hipStream_t st);
#endif
const uint32_t* g_bwd_order = nullptr;   // (experiments: gsr_debug_set_bwd_order)

void launch_blend_bwd(int C, int W, int H, int U, const float* bg, const float* feats, GeomState g, ImageState im,
                      BinState b, const float* dL_dpix, float* grad_acc, hipStream_t st)
{
    (void)feats; (void)g;
    if (U <= 0) return;
#ifdef GSR_BWD_VARIANT
    if (launch_blend_bwd_variant(C, W, H, U, bg, im, b, dL_dpix, grad_acc, st)) return;
#endif
    static const int pad = getenv("GSR_BWD_LDS_PAD") ? atoi(getenv("GSR_BWD_LDS_PAD")) : 0;   // residency knob (tuning only)
    const Tiles t = tiles_of(W, H);
    uint64_t* tr = g_trace ? g_trace + 2 * (size_t)t.T : nullptr;
    if (C == 3)
        blend_bwd_kernel<3><<<4 * U, 64, pad, st>>>(W, H, t.gx, b.unit_info, b.snap, b.masks, b.point_list, b.rec_a, b.rec_b,
                                                    static_cast<const RecTail<3>*>(b.rec_c), bg, im.final_T, im.n_contrib, dL_dpix, grad_acc, tr, g_bwd_order);
    else if (C == 4)
        blend_bwd_kernel<4><<<4 * U, 64, pad, st>>>(W, H, t.gx, b.unit_info, b.snap, b.masks, b.point_list, b.rec_a, b.rec_b,
                                                    static_cast<const RecTail<4>*>(b.rec_c), bg, im.final_T, im.n_contrib, dL_dpix, grad_acc, tr, g_bwd_order);
    else
        blend_bwd_kernel<6><<<4 * U, 64, pad, st>>>(W, H, t.gx, b.unit_info, b.snap, b.masks, b.point_list, b.rec_a, b.rec_b,
                                                    static_cast<const RecTail<6>*>(b.rec_c), bg, im.final_T, im.n_contrib, dL_dpix, grad_acc, tr, g_bwd_order);
}

}  // namespace gsr
