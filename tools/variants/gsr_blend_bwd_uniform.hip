// gsr_blend_bwd_uniform.hip -- backward alpha compositing, the UNIFORM PAIR LOOP: the product's backward blend of rounds 1-5 (one
// kept instance on the block's 64 pixels per trip, w and r parked in an LDS table, the moments contracted over the pixels on the
// matrix pipe).  Round 6 replaced it by four kept instances per trip (gaustar_amd/csrc/gsr_blend_bwd.hip; profiles/
// r06_bwd_quad_counters.txt); this file is the record and the A/B partner, built only by
//     python -m gaustar_amd.build --variant uniform --with tools/variants/gsr_blend_bwd_uniform.hip
// (defines launch_blend_bwd_variant for every channel count; GSR_BWD_UNIFORM=0 in the environment hands the launch back to the
// product's kernel inside that build: tools/ab_env.py GSR_BWD_UNIFORM 1 0).  The per-wave trace devtools
// (tests/devtools/trace_bwd_waves.py, -DGSR_TRACE_DETAIL) run on this build.
//
// Per-pair arithmetic is the reference's renderCUDA backward (DGR/cuda_rasterizer/backward.cu:399-557;
// SURVEY.md section 9 item 10): back-to-front replay, T recovered by division, accum_rec recurrence,
// background term with T_final/(1-alpha), the 0.99 alpha clamp passing gradient as if unclamped.
//
// What differs is the decomposition and how the per-pair terms reach memory.  The reference runs one block
// per tile over the whole list and issues 9 global float atomicAdds per contributing (pixel, Gaussian) pair
// (backward.cu:523, :545-554).  Here
//
//  * the work unit is a (tile, SEGMENT of 64 list positions, 8x8 pixel block) triple, one wave64 each.
//    A pixel whose last contributor lies beyond the segment starts from the forward pass's snapshot at the
//    segment's far boundary: T = T_snap, accum_rec = (C_final - C_snap) / T_snap -- exactly the state the
//    reference's back-to-front recurrence has at that list position; a pixel that ends inside the segment
//    starts from (T_final, 0) like the reference; a pixel that ended before it is idle.  Units have bounded
//    size, so the dispatcher can balance them and nothing carries a 1 600-instance serial chain;
//  * which instances of the segment the block needs comes from the forward's per-pixel candidate words (gsr_mask.h):
//    the OR over the block's pixels of (word AND "positions this pixel replays").  The unit's 64 records arrive as
//    three contiguous rows the forward left in list order (lane i takes position 63 - i, together with every other
//    load of the unit's head); only the needed ones are queued in LDS -- no geometric test is repeated here;
//  * for a queued instance every lane evaluates its pixel and produces just TWO numbers,
//    w = alpha*T and r = G*dL_dalpha.  Everything the gradients need is a sum over the block's pixels of w or r
//    times a per-pixel constant:   sum w*dL_dpix_{r,g,b}   and   sum r*{1, x, y, x^2, xy, y^2}  (x, y = pixel
//    coordinates relative to the block centre).  That is a contraction over the 64 pixels,
//        [instances x pixels] . [pixels x 9],
//    and it runs on the matrix pipe: w and r are parked in LDS one row per instance, read back transposed, and
//    reduced by matrix instructions.  This is the one place on the path that IS a contraction.  Six channels use
//    v_mfma_f32_16x16x4_f32 (exact f32 multiply-add, 16 issues cover 64 pixels): it occupies the SIMD's vector
//    multipliers for its 32 cycles (no overlap with vector work: tools/micro/mfma_valu_overlap.hip) -- what it saves is
//    instructions, 16 per eight instances instead of a 26-instruction cross-lane reduction PER instance.  Three channels
//    (the reference's case) use the bf16 pipe on exact three-way splits of the f32 values (see GSR_BWD_BF16 below): 6
//    issues of ~17 cycles instead of 16 of 32, bought with 88 vector instructions of splitting per eight instances;
//    The moments land in the fields of the instance's LDS queue slot that are dead by then;
//  * one lane per instance then re-centres the six spatial sums on the splat (dx = x_splat - x_pixel) --
//    dL_dcolor, dL_dopacity, dL_dmean2D and dL_dconic are fixed per-Gaussian linear maps of the nine moments,
//    applied once per Gaussian in geom_bwd -- and the table is flushed ROW-MAJOR: one atomic instruction covers
//    the nine consecutive floats of ~7 packed 48-byte records grad_acc[gaussian][12], so the memory pipeline
//    merges lanes per cache line (1.5 M atomic requests per 1080p view instead of 8 M).
//
// Template over the channel count C (3 = the reference's NUM_CHANNELS; 6 = two targets sharing geometry blended
// in one walk, see gsr_blend_fwd.hip): w is shared by all channels, r sums dL_dalpha over them, so a 6-channel
// unit costs ~20 % more than a 3-channel one instead of 2x.  Record of the accumulation table:
// grad_acc[gaussian][GRAD_RS] = {sum r, sum r dx, sum r dy, sum r dx^2, sum r dx dy, sum r dy^2, c_0 .. c_{C-1}}.
#include "gsr_bwd_util.h"
#ifndef GSR_BWD_PROJ
#define GSR_BWD_PROJ 1
#endif

namespace gsr {


template <int C>
__device__ __forceinline__ void
blend_bwd_unit(int W, int H, int gx, const uint4* __restrict__ unit_info, const float4* __restrict__ snap,
                 const uint2* __restrict__ masks, const uint32_t* __restrict__ point_list, const float4* __restrict__ rec_a,
                 const float4* __restrict__ rec_b, const RecTail<C>* __restrict__ rec_c, const float* __restrict__ bg,
                 const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                 const float* __restrict__ dL_dpix, float* __restrict__ grad_acc, uint64_t* __restrict__ trace,
                 const uint32_t* __restrict__ order)
{
    using L = SlotLayout<C>;
    constexpr int SF = L::FLOATS, NM = L::NM, MOM0 = L::MOM0, SV = snap_vecs(C);
    static_assert(NM <= 16 && NM <= GRAD_RS, "moment columns must fit one MFMA tile and one record");
    static_assert(L::IN_VECS == 3 || L::IN_VECS == 4, "slot reads are written for three or four float4");
    const uint64_t t_start = trace ? wall_clock64() : 0;
    // (bf16 path: 69 registers allow seven waves per SIMD, which the LDS budget only admits with 16 slots -- 5 120 bytes
    // per wave; chunks are whole MFMA groups, so a batch of 28 is grouped 8+8 | 8+4 either way)
#ifndef GSR_BWD_B2_LDS
#define GSR_BWD_B2_LDS 1   // four channels: the second B tile (channel 3's three split columns) is read from LDS per group of
                           // eight instances instead of living in eight registers (86 -> 80 registers = six waves per SIMD)
#endif
#ifndef GSR_BWD_PACK4
#define GSR_BWD_PACK4 1    // four channels in ONE B tile: channels 0, 1 keep three split columns, channels 2, 3 take two (hi + the
                           // rest rounded to bf16: 16 mantissa bits of dL_dpix, relative error <= 2^-17 on dL_dcolor[2], [3]
                           // only -- the alpha path reads dL_dpix on the vector ALU); no second tile, no second accumulator
#endif
#ifndef GSR_BWD_QCAP
#define GSR_BWD_QCAP (GSR_BWD_BF16 && (C == 3 || (C == 4 && (GSR_BWD_B2_LDS || GSR_BWD_PACK4))) ? 16 : 32)
#endif
    constexpr int QCAP = GSR_BWD_QCAP;
    static_assert(QCAP >= GRP && QCAP <= 64 && QCAP % GRP == 0, "queue capacity: whole MFMA groups, at most one batch");
    __shared__ __attribute__((aligned(16))) float qf[QCAP * SF];   // queue slots (see SlotLayout)
    __shared__ __attribute__((aligned(16))) float Rm[2 * GRP * RSTRIDE];   // rows 0..7: r, rows 8..15: w, [row][pixel lane]
    // (four channels, GSR_BWD_B2_LDS) second B tile as bf16 [column 0..2 = hi, mid, lo of dL_dpix channel 3][64 pixels] + 32 zero bytes
    constexpr bool PACK4 = GSR_BWD_PACK4 && GSR_BWD_BF16 && C == 4;
    constexpr bool B2L = GSR_BWD_B2_LDS && GSR_BWD_BF16 && GSR_BWD_BF16_TILES2 && C == 4 && !PACK4;
    __shared__ __attribute__((aligned(16))) uint32_t B2s[B2L ? 3 * 32 + 8 : 1];
    // one wave64 per workgroup: unit = (tile, segment), wave = 8x8 block of the tile.
    // XCD-aware placement: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the
    // naive map (unit = id / 4) would put the four blocks of a unit -- which read the SAME instance records -- on
    // four different L2s.  Instead the four blocks of a unit take four consecutive slots of ONE XCD.
    // Units of one tile are consecutive and also share their pixels' dL_dpix / T / n_contrib and the tile's final
    // snapshot, so an XCD takes RUNS of 8 consecutive units: of every 64 units, XCD x owns [8x, 8x + 8).
    const uint32_t n_units = gridDim.x >> 2;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t grp = slot >> 2;                       // index of this (unit) among the XCD's units
    uint32_t unit = (grp >> 3) * 64u + xcd * 8u + (grp & 7u);
    uint32_t wave_sel = slot & 3u;
    const uint32_t full = (n_units >> 6) << 6;            // units covered by complete groups of 64
    if (blockIdx.x >= full * 4u) { unit = blockIdx.x >> 2; wave_sel = blockIdx.x & 3u; }   // ragged tail: plain map
    // (a launch ORDER of the units, when there is one: position in the dispatch sequence -> unit; see launch_blend_bwd)
    if (order != nullptr) unit = order[unit];
    // Everything the unit has to know about its tile in one (scalar) load; then EVERY vector load of the unit's head is
    // requested before the first one is waited for -- pixel state, candidate words of this unit and the next, the two
    // snapshots a resuming pixel needs, and the unit's 64 instance records, which the forward left in list order
    // (rec_a/b/c).  As a chain (unit -> tile -> ranges -> n_contrib -> words -> snapshot; words -> list -> id -> geometry)
    // the head of a unit was six dependent trips to memory, a third of a unit's life, with nothing to issue meanwhile.
    const uint4 info = unit_info[unit];
    const int tile = (int)info.x;
    const uint32_t list0 = info.y;
    const int n = (int)info.z;
    const uint32_t unit0 = info.w;
    const int s0 = (int)(unit - unit0) * 64;           // this unit covers list positions [s0, s1)
    if (s0 >= n) return;   // (a planned view numbers its units by bucket CAPACITY: this one lies past the list's end)
    const int wave = (int)wave_sel, lane = threadIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int sx = tx * TILE + (wave & 1) * SUB, sy = ty * TILE + (wave >> 1) * SUB;
    const int px = sx + (lane & 7), py = sy + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)sx, by0 = (float)sy;
    const int s1 = min(s0 + BSEG, n);
    const bool has_next = s1 < n;                       // (uniform) the tile has a unit behind this one

    // (32-bit element indices on uniform base pointers: the loads take an SGPR base + a VGPR offset instead of 64-bit
    // vector address arithmetic; gsr_forward_stage1 caps the image at 8k x 8k, so channel * H * W + pixel fits)
    const uint32_t pix = (uint32_t)W * (uint32_t)py + (uint32_t)px;
    const uint32_t HW = (uint32_t)H * (uint32_t)W;
    // (explicit 32-bit BYTE offsets: `base[pix]` widens the index to 64 bits and the compiler then builds a 64-bit vector
    // address per load; one `if` around all of them: as separate conditional expressions each load got its own branch)
    const auto at32 = [](const auto* base, uint32_t byte_off) {
        return *reinterpret_cast<std::remove_reference_t<decltype(*base)>*>(reinterpret_cast<const char*>(base) + byte_off);
    };
    float T_final = 0.f;
    int my_last = 0;   // 1-based position of the last contributor
    float dp[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) dp[ch] = 0.f;
    if (inside) {
        T_final = at32(final_T, pix * 4u);
        my_last = (int)at32(n_contrib, pix * 4u);
#pragma unroll
        for (int ch = 0; ch < C; ch++) dp[ch] = at32(dL_dpix, ((uint32_t)ch * HW + pix) * 4u);
    }
    // this pixel's candidate word over the unit's 64 positions, from the forward (gsr_mask.h); the forward's lanes are
    // the block's pixels in the same row-major order as here
    const uint2* const my_words = masks + ((size_t)unit * 4 + wave) * 64 + (uint32_t)lane;   // + 256 per unit
    const uint2* const words_u = masks + ((size_t)unit * 4 + wave) * 64;                      // (uniform part)
    uint2 word = at32(words_u, (uint32_t)lane * 8u);
    // (unconditional load from a uniform, always valid address, masked once everything is in flight: as `has_next ? load : 0`
    // the compiler waited for the load inside the branch -- before the snapshot and record loads below were even issued,
    // two trips to memory in a row at the head of every unit)
    uint2 word_next = at32(words_u + (has_next ? 256 : 0), (uint32_t)lane * 8u);
    const int pidx = 16 * (py - ty * TILE) + (px - tx * TILE);
    float Ts, Tf, cs[C], cf[C];
    const auto load_snap32 = [&](const float4* base_u, float& T_, float (&c_)[C]) {   // uniform base, this pixel's slot
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
    // (both are only USED by a pixel whose last contributor lies beyond this unit -- then the tile has a next unit and more
    // than one of them -- so they are loaded unconditionally, from a slot that always exists: no defaults to set, no branches)
    load_snap32(snap + (size_t)(unit + (has_next ? 1u : 0u)) * 256 * SV, Ts, cs);
    load_snap32(snap + (size_t)unit0 * 256 * SV, Tf, cf);   // final (T, C) kept in the tile's first slot
    // lane l holds list position s0 + 63 - l (queue order == back-to-front order)
    const int k = s0 + 63 - lane;
    // (a lane whose position lies past the end of the list reads the list's last record instead of carrying zeros: no
    // candidate bit can name such a position, so the lane is never kept, and nothing has to be initialised or branched over)
    const uint32_t kl = (uint32_t)(min(k, n - 1) - s0);   // (uniform base list0 + s0, lane offset)
    const float4 ra = at32(rec_a + list0 + s0, kl * 16u);
    const float4 rb = at32(rec_b + list0 + s0, kl * 16u);
    const RecTail<C> rc = at32(rec_c + list0 + s0, kl * (uint32_t)sizeof(RecTail<C>));
    const uint32_t gid = at32(point_list + list0 + s0, kl * 4u);
    if (!has_next) word_next = make_uint2(0u, 0u);
    float bg_dot_dpixel = 0.f;
#pragma unroll
    for (int ch = 0; ch < C; ch++) bg_dot_dpixel += bg[ch] * dp[ch];

    // Per-pixel start state at the far end of the segment.
    float T = T_final;
    const float tf_bg = T_final * bg_dot_dpixel;
    // acc = colour composited BEHIND the current list position as seen from it (the reference's accum_rec).  The
    // reference folds contributor i into accum_rec lazily, when it reaches contributor i-1 (last_alpha / last_color,
    // backward.cu:505-520); folding it right after use is the same arithmetic on the same operands one step earlier
    // and needs no "last" registers.
    float acc[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) acc[ch] = 0.f;
    const int my_lim = min(my_last, s1);                 // this pixel replays positions [s0, my_lim)
    if (my_last > s1) {
        // the pixel blended instances beyond this segment: resume from the forward's snapshot taken before
        // list position s1.  accum_rec at that point = colour composited behind s1, seen from s1.
        // The forward stored a snapshot whenever the pixel moved on to a word of a new segment (gsr_blend_fwd.hip); the
        // first segment behind this unit in which the pixel has a candidate at all has one, and nothing was blended
        // between this unit's far end and that segment, so it is the state at list position s1.  Almost always that is
        // the very next segment (requested above); only a pixel whose words there are empty walks on.
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
#if GSR_BWD_PROJ
    // dL_dalpha only ever needs accum_rec through its dot product with dL_dpix, and accum_rec' = accum_rec + alpha (c -
    // accum_rec) is linear: carry A = accum_rec . dL_dpix instead of the C channels.  With k = c . dL_dpix:
    // s = sum_ch (c_ch - accum_rec_ch) dL_dpix_ch = k - A,  A' = A + alpha s  (A = accd below).  C + 2 instructions per live
    // pair instead of 3 C, and C - 1 registers less: three channels 80 -> 70 = seven waves per SIMD (59.5 -> 55.4 M vector
    // instructions per view; GSR_BWD_PROJ=0 is the per-channel form, bit-compatible with rounds 1-3).
    float accd = 0.f;
#pragma unroll
    for (int ch = 0; ch < C; ch++) accd = __builtin_fmaf(acc[ch], dp[ch], accd);
#endif

    // Which of the unit's positions ANY pixel of the block replays: the OR over the lanes of (candidate word AND
    // "positions below this pixel's limit").  No geometric test is repeated here, and only these instances are fetched.
    {
        const int lim = my_lim - s0;                         // <= 64; <= 0: nothing
        word.x &= lim >= 32 ? 0xffffffffu : lim > 0 ? (1u << lim) - 1u : 0u;
        word.y &= lim >= 64 ? 0xffffffffu : lim > 32 ? (1u << (lim - 32)) - 1u : 0u;
    }
    // (one same-address LDS atomic for the whole wave instead of two seven-step DPP ladders: three LDS instructions and
    // five vector ones where there were eighteen; the r|w table is not in use yet and lends its first eight bytes)
    const unsigned long long kany = wave_or_u64_lds(lds_byte_address(Rm), word.x, word.y);
#ifdef GSR_TRACE_DETAIL
    const uint64_t t_head = wall_clock64();
    if (trace && lane == 0 && kany == 0ull) {
        uint64_t* tw = trace + ((size_t)unit * 4 + wave) * 4;
        tw[0] = t_start; tw[1] = t_head; tw[2] = t_head;
        tw[3] = ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
#endif
    if (kany == 0ull) return;

    // B operand of the contraction, constant over the unit.  MFMA step t (0..15) consumes the four pixels
    // p = 16*kap + t, kap = 0..3; in the B operand lane l carries row kap = l >> 4, column col = l & 15.
    // Columns 0..5: {1, x, y, x^2, xy, y^2} of pixel p relative to the block centre (used by the r rows);
    // columns 6..6+C-1: dL_dpix of pixel p per channel (used by the w rows; staged through LDS once).
    const int kap = lane >> 4, col = lane & 15;
    // Every column is staged as a row of 64 floats [pixel] in the (not yet used) r|w table, then each lane reads the 16
    // pixels of ITS column with four ds_read_b128: rows 0..5 the monomials of the lane's own pixel (exact small
    // half-integers), row 6 + ch = dL_dpix channel ch, one all-zero row for the unused columns.  (Forming the monomials
    // per (lane, step) in registers took 90 vector instructions per wave.)
    // bf16 path: tile 1 = the six monomials + three split columns for each of the first three channels (15 of 16 columns);
    // channels 3 .. C-1 take a second B tile (their split columns 0 .. 3 (C - 3) - 1) fed with the SAME split A operand:
    // six more matrix issues per eight instances, no further splitting.
    // (measured, config C: four channels 0.140 -> 0.134 ms; six channels gain nothing -- 0.158 either way -- and stay on f32)
    constexpr bool BF16 = GSR_BWD_BF16 && (C == 3 || ((GSR_BWD_BF16_TILES2 || PACK4) && C == 4));
    constexpr int C1 = BF16 ? (PACK4 ? 4 : C < 3 ? C : 3) : C, C2 = BF16 ? C - C1 : 0;
    constexpr int BROWS = BF16 ? (PACK4 ? 16 : 6 + 3 * C1) : 6 + C;
    // (PACK4: columns 6-8 = channel 0, 9-11 = channel 1, 12-13 = channel 2, 14-15 = channel 3; all sixteen in use, no zero row)
    constexpr int BS = RSTRIDE;   // row stride of the staging rows: with 64 the sixteen columns a 16-lane group reads sit in the same
                              // four banks (a 16-way conflict on each of the four reads below); 68 spreads them over all 64
    static_assert((BROWS + (PACK4 ? 0 : 1)) * BS <= 2 * GRP * RSTRIDE, "B-operand staging must fit the r|w table");
    {
        const float xr = (float)(lane & 7) - 3.5f, yr = (float)(lane >> 3) - 3.5f;
        Rm[0 * BS + lane] = 1.0f;
        Rm[1 * BS + lane] = xr;
        Rm[2 * BS + lane] = yr;
        Rm[3 * BS + lane] = xr * xr;
        Rm[4 * BS + lane] = xr * yr;
        Rm[5 * BS + lane] = yr * yr;
        if constexpr (BF16) {
#pragma unroll
            for (int ch = 0; ch < C1; ch++) {
                const float d1 = bf16_rest(dp[ch]), d2 = bf16_rest(d1);
                if (PACK4 && ch >= 2) {
                    Rm[(12 + 2 * (ch - 2)) * BS + lane] = dp[ch];
                    Rm[(13 + 2 * (ch - 2)) * BS + lane] = __uint_as_float(__float_as_uint(d1) + 0x8000u);   // rounded, not cut
                    continue;
                }
                Rm[(6 + 3 * ch) * BS + lane] = dp[ch];      // (the operand takes the upper halves: hi, mid, lo)
                Rm[(7 + 3 * ch) * BS + lane] = d1;
                Rm[(8 + 3 * ch) * BS + lane] = d2;
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < C; ch++) Rm[(6 + ch) * BS + lane] = dp[ch];
        }
        if constexpr (!PACK4) Rm[BROWS * BS + lane] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    float Bf[BF16 ? 1 : 16];
    u32x4 Bp[BF16 ? 2 : 1];
    u32x4 Bp2[C2 > 0 && !B2L ? 2 : 1];
    uint32_t b2_off = 0;    // (B2L) byte offset of this lane's 16 bytes of half 0 (+ 16: half 1); unused columns read the zero block
    {
        const float4* pd = reinterpret_cast<const float4*>(&Rm[(PACK4 || col < BROWS ? col : BROWS) * BS + 16 * kap]);
        float bv[16];
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            const float4 v = pd[qd];
            bv[4 * qd] = v.x; bv[4 * qd + 1] = v.y; bv[4 * qd + 2] = v.z; bv[4 * qd + 3] = v.w;
        }
        if constexpr (BF16) {
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int q = 0; q < 4; q++) Bp[h][q] = bf16_pair(bv[8 * h + 2 * q], bv[8 * h + 2 * q + 1]);
        } else {
#pragma unroll
            for (int t = 0; t < 16; t++) Bf[t] = bv[t];
        }
    }
    __builtin_amdgcn_wave_barrier();
    if constexpr (C2 > 0 && B2L) {
        // second tile kept in LDS: every pixel lane parks the three bf16 parts of its dL_dpix channel 3, a lane of the matrix
        // operand (column col, pixels 16 kap + 8 h ..) reads its eight values back per group
        static_assert(C2 == 1, "one extra channel");
        const float d0 = dp[C1], d1 = bf16_rest(d0), d2 = bf16_rest(d1);
        unsigned short* b2h = reinterpret_cast<unsigned short*>(B2s);
        b2h[0 * 64 + lane] = (unsigned short)(__float_as_uint(d0) >> 16);
        b2h[1 * 64 + lane] = (unsigned short)(__float_as_uint(d1) >> 16);
        b2h[2 * 64 + lane] = (unsigned short)(__float_as_uint(d2) >> 16);
        if (lane < 8) B2s[3 * 32 + lane] = 0u;
        b2_off = col < 3 ? (uint32_t)(col * 128 + 32 * kap) : (uint32_t)(3 * 128);
        __builtin_amdgcn_wave_barrier();
    } else if constexpr (C2 > 0) {   // second tile: the same staging once more, rows 0 .. 3 C2 - 1 = (hi, mid, lo) of channels 3 ..
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
    // The queue holds QCAP of the batch's up to 64 kept instances at a time (LDS per workgroup decides how many units are
    // resident, and a typical batch keeps ~22): a batch that keeps more is worked off in chunks, back-to-front order intact.
    for (int q0 = 0; q0 < cnt_all; q0 += QCAP) {
    const int cnt = min(cnt_all - q0, QCAP);
    // (v_mbcnt_lo / v_mbcnt_hi: set bits of m below this lane in two instructions; `m & ((1ull << lane) - 1)` is a 64-bit
    // shift, a 64-bit subtract, two ands and two bit counts)
    const int slot = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) - q0;
    if (keep && slot >= 0 && slot < QCAP) {
        float4* qs = reinterpret_cast<float4*>(&qf[slot * SF]);
        float col[C];
        col[0] = rb.z; col[1] = rb.w;
#pragma unroll
        for (int ch = 2; ch < C; ch++) col[ch] = rc.c[ch - 2];
        qs[0] = make_float4(ra.x, ra.y, __uint_as_float(gid), ra.z);      // the conic is already in the exp2 domain
        qs[1] = make_float4(ra.w, rb.x, rb.y, __uint_as_float((uint32_t)k));
#pragma unroll
        for (int v = 2; v < L::VECS; v++) {
            const int c0 = 4 * (v - 2);
            qs[v] = make_float4(c0 < C ? col[c0 < C ? c0 : 0] : 0.f, c0 + 1 < C ? col[c0 + 1 < C ? c0 + 1 : 0] : 0.f,
                                c0 + 2 < C ? col[c0 + 2 < C ? c0 + 2 : 0] : 0.f,
                                c0 + 3 < C ? col[c0 + 3 < C ? c0 + 3 : 0] : 0.f);
        }
    }
    __builtin_amdgcn_wave_barrier();

    const uint32_t q_base = lds_byte_address(qf);
    const uint32_t rw_addr = lds_byte_address(Rm) + 4u * (uint32_t)lane;   // this lane's column of the r|w table
    // where this lane's four accumulator registers go: rows 4 kap .. 4 kap + 3 of the D tile = instances (row & 7)
    static_assert(GRP == 8, "the write-back below assumes rows 0-7 = r, 8-15 = w");
    const int wb_row0 = 4 * (kap & 1);
    // (bf16 path: the colour moment of channel ch is the sum of columns 6 + 3 ch .. + 2, gathered into the first of them)
    const bool wb_take = kap < 2 ? col < 6
                       : PACK4   ? (col == 6 || col == 9 || col == 12 || col == 14)
                       : BF16    ? (col >= 6 && col < 6 + 3 * C1 && (col % 3) == 0) : (col >= 6 && col < NM);
    float* const wb_ptr = &qf[wb_row0 * SF + MOM0 + (PACK4 && col >= 12 ? 8 + (col - 12) / 2 : BF16 && col >= 6 ? 6 + (col - 6) / 3 : col)];
    // (second tile: its w rows hold channel 3 + col / 3 in columns 0, 3, ..)
    const bool wb_take2 = C2 > 0 && kap >= 2 && col < 3 * C2 && (col % 3) == 0;
    float* const wb_ptr2 = &qf[wb_row0 * SF + MOM0 + 6 + C1 + (col < 3 * C2 ? col / 3 : 0)];
    for (int g0i = 0; g0i < cnt; g0i += GRP) {
        // ---- vector ALU: w and r of GRP instances for this lane's pixel, parked row-wise in LDS.  The slot of the
        // group's first instance is requested here, every further one while its predecessor is being evaluated.
        SlotRegs<L::IN_VECS> nxt;
        // (the group's slot address is pinned in a vector register: as a uniform value the compiler keeps it scalar and
        // copies it into a fresh vector register for every pair)
        uint32_t q_grp = q_base + (uint32_t)(g0i * SF * 4);
        asm volatile("" : "+v"(q_grp));
        lds_request<L::IN_VECS, 0>(nxt, q_grp);
        static_for<GRP>([&](auto JJ) {
            constexpr int jj = decltype(JJ)::value;
            const int j = g0i + jj;
            float r = 0.f, w = 0.f;
            // Request and wait sit in straight-line code, outside the (uniform) j < cnt branch: registers with a load in
            // flight must not cross a control-flow merge, where the compiler may copy them -- reading them before the
            // data have landed.  Likewise the wait is on the requested registers themselves, the copy comes after it.
            // (Slot j + 1 < QCAP exists in LDS; past the end of the queue it holds stale numbers nobody uses.)
            lds_wait<(jj == 0 ? 0 : 2)>(nxt);   // (jj > 0: the two table stores of pair jj - 1 were issued after this request)
            const SlotRegs<L::IN_VECS> cur = nxt;
            if constexpr (jj + 1 < GRP) lds_request<L::IN_VECS, (jj + 1) * SF * 4>(nxt, q_grp);
            if (j < cnt) {
                const float4 A = make_float4(cur.v[0][0], cur.v[0][1], cur.v[0][2], cur.v[0][3]);
                const float4 B = make_float4(cur.v[1][0], cur.v[1][1], cur.v[1][2], cur.v[1][3]);
                float cc[C];
#pragma unroll
                for (int ch = 0; ch < C; ch++) cc[ch] = cur.v[2 + ch / 4][ch % 4];
                const int pos = (int)__float_as_uint(B.w);
                const float dx = A.x - pxf, dy = A.y - pyf;
                const float power = pair_exp2_arg(A.w, B.x, B.y, dx, dy);
                const float G = __builtin_amdgcn_exp2f(power);
                const float alpha = fminf(ALPHA_MAX, B.z * G);
                const bool live = pos < my_lim && power <= 0.0f && alpha >= ALPHA_MIN;
                {
                    // (`if (live)` compiles to s_and_saveexec + s_cbranch_execz: a pair without a live pixel skips the
                    // block; an explicit ballot test around it doubled the branching and cost 13 us per view)
                    if (live) {
                        // accum_rec' = alpha c + (1 - alpha) accum_rec written as accum_rec + alpha (c - accum_rec), and
                        // projected on dL_dpix (see accd above); T_final * bg . dL_dpix is a per-pixel constant.
                        // 11 vector instructions per live pair at three channels (per-channel form: 14).
                        const float rinv = __builtin_amdgcn_rcpf(1.f - alpha);
                        T = T * rinv;
                        w = alpha * T;
#if GSR_BWD_PROJ
                        float kd = cc[0] * dp[0];
#pragma unroll
                        for (int ch = 1; ch < C; ch++) kd = __builtin_fmaf(cc[ch], dp[ch], kd);
                        const float s = kd - accd;
                        accd = __builtin_fmaf(alpha, s, accd);
#else
                        float s = 0.f;
#pragma unroll
                        for (int ch = 0; ch < C; ch++) {
                            const float d = cc[ch] - acc[ch];
                            s = __builtin_fmaf(d, dp[ch], s);
                            acc[ch] = __builtin_fmaf(alpha, d, acc[ch]);
                        }
#endif
                        r = G * __builtin_fmaf(s, T, -(rinv * tf_bg));
                    }
                }
            }
            // (explicit instructions: the wait above counts on exactly these two LDS operations behind every request --
            // the compiler must neither fuse them into one ds_write2 nor move them)
            lds_store_b32<jj * RSTRIDE * 4>(rw_addr, r);
            lds_store_b32<(GRP + jj) * RSTRIDE * 4>(rw_addr, w);
        });
        __builtin_amdgcn_wave_barrier();
        // ---- matrix pipe: [16 rows = r and w of GRP instances] x [64 pixels] . [64 pixels x 16 columns].
        // A operand: lane l carries table row (l & 15) and the 16 pixels 16*kap .. 16*kap + 15 -> 16 consecutive floats of
        // the row (four ds_read_b128).  f32 instruction: step t consumes pixel 16*kap + t; two interleaved accumulators
        // (even / odd steps) halve the dependent-accumulator chain.  bf16 instruction: K = 32 = the lanes' first (h = 0)
        // resp. second (h = 1) eight pixels; per half one issue each for the lo, mid and hi parts, smallest first.
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};   // second tile (C2 > 0)
        float ra[16];
        {
            const float4* pr = reinterpret_cast<const float4*>(&Rm[(col < 2 * GRP ? col : 0) * RSTRIDE + 16 * kap]);
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                const float4 v = (2 * GRP == 16 || col < 2 * GRP) ? pr[qd] : make_float4(0.f, 0.f, 0.f, 0.f);
                ra[4 * qd] = v.x; ra[4 * qd + 1] = v.y; ra[4 * qd + 2] = v.z; ra[4 * qd + 3] = v.w;
            }
        }
        if constexpr (BF16) {
            uint32_t kMinusOneLo = 0x0000BF80u, kMinusOneHi = 0xBF800000u;   // bf16 pairs {-1, 0}, {0, -1}
            asm volatile("" : "+v"(kMinusOneLo), "+v"(kMinusOneHi));
#pragma unroll
            for (int h = 0; h < 2; h++) {
                u32x4 a_hi, a_mid, a_lo;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float x0 = ra[8 * h + 2 * q], x1 = ra[8 * h + 2 * q + 1];
                    a_hi[q] = bf16_pair(x0, x1);
#if GSR_BWD_DOT2
                    const float y0 = bf16_rest_of(a_hi[q], x0, kMinusOneLo), y1 = bf16_rest_of(a_hi[q], x1, kMinusOneHi);
                    a_mid[q] = bf16_pair(y0, y1);
                    const float z0 = bf16_rest_of(a_mid[q], y0, kMinusOneLo), z1 = bf16_rest_of(a_mid[q], y1, kMinusOneHi);
#else
                    const float y0 = bf16_rest(x0), y1 = bf16_rest(x1);
                    a_mid[q] = bf16_pair(y0, y1);
                    const float z0 = bf16_rest(y0), z1 = bf16_rest(y1);
#endif
                    a_lo[q] = bf16_pair(z0, z1);
                }
                const bf16x8 b = __builtin_bit_cast(bf16x8, Bp[h]);
                f32x4& acc = h ? acc1 : acc0;
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_lo), b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_mid), b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_hi), b, acc, 0, 0, 0);
                if constexpr (C2 > 0) {
                    u32x4 b2r;
                    if constexpr (B2L) b2r = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(B2s) + b2_off + 16 * h);
                    else b2r = Bp2[h];
                    const bf16x8 b2 = __builtin_bit_cast(bf16x8, b2r);
                    f32x4& bcc = acc2;   // (one chain for both halves: four registers less, and the matrix pipe is far from busy)
                    bcc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_lo), b2, bcc, 0, 0, 0);
                    bcc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_mid), b2, bcc, 0, 0, 0);
                    bcc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_hi), b2, bcc, 0, 0, 0);
                }
            }
            // colour rows: the three split columns of a channel sit in neighbouring lanes of the 16-lane row
            // (lane c: hi column, c + 1: mid, c + 2: lo.  s = v + v[c + 1] holds mid + lo in lane c + 1; v + s[c + 1] is
            // hi + (mid + lo) -- two fused DPP adds per register, smallest parts first.  One block, so that the two
            // wait states a DPP read needs behind the instruction that wrote its source are there by construction: the
            // leading s_nop covers the compiler's adds, every t reads an s written four instructions earlier)
            const auto split_sum = [](float v0, float v1, float v2, float v3, float& t0_, float& t1_, float& t2_, float& t3_,
                                      float& s0_, float& s1_, float& s2_, float& s3_) {
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
            float t0_, t1_, t2_, t3_, p0_, p1_, p2_, p3_;
            split_sum(v0, v1, v2, v3, t0_, t1_, t2_, t3_, p0_, p1_, p2_, p3_);
            if constexpr (C2 > 0) {
                float u0, u1, u2, u3, q0_, q1_, q2_, q3_;
                split_sum(acc2[0], acc2[1], acc2[2], acc2[3], u0, u1, u2, u3, q0_, q1_, q2_, q3_);
                acc2[0] = u0; acc2[1] = u1; acc2[2] = u2; acc2[3] = u3;
            }
            if constexpr (PACK4) {   // the two-column channels stop at hi + rest
                const bool two = col >= 12;
                t0_ = two ? p0_ : t0_; t1_ = two ? p1_ : t1_; t2_ = two ? p2_ : t2_; t3_ = two ? p3_ : t3_;
            }
            const bool spatial = kap < 2;
            acc0[0] = spatial ? v0 : t0_; acc0[1] = spatial ? v1 : t1_; acc0[2] = spatial ? v2 : t2_; acc0[3] = spatial ? v3 : t3_;
        } else {
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t], Bf[t], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t + 1], Bf[t + 1], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) acc0[i] += acc1[i];
        }
        // D layout: lane l, register i -> operand row 4*(l >> 4) + i, column l & 15.
        // rows 0..7: r of instance row, columns 0..5 = spatial sums; rows 8..15: w of instance row-8, columns 6..6+C-1.
        // (wb_take / wb_row0 / wb_ptr are per-lane constants of the wave, see above)
        if (wb_take) {
            float* const dst = wb_ptr + g0i * SF;
            const int left = cnt - g0i - wb_row0;     // instances of this group at or behind the lane's first row
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (i < left) dst[i * SF] = acc0[i];
        }
        if constexpr (C2 > 0) {
            if (wb_take2) {
                float* const dst = wb_ptr2 + g0i * SF;
                const int left = cnt - g0i - wb_row0;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (i < left) dst[i * SF] = acc2[i];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- lane = queued instance: re-centre the spatial sums on the splat (dx = x_splat - x_pixel), in place:
    // {sum r, sum r dx, sum r dy, sum r dx^2, sum r dx dy, sum r dy^2}; the colour moments stay as they are.
    // (every queued instance is flushed: 99.4 % of them have a live pixel, the others add zeros)
    if (lane < cnt) {
        float* rw = &qf[lane * SF];
        const float m0 = rw[MOM0], mx = rw[MOM0 + 1], my = rw[MOM0 + 2], mxx = rw[MOM0 + 3], mxy = rw[MOM0 + 4],
                    myy = rw[MOM0 + 5];
        const float X = rw[0] - (bx0 + 3.5f), Y = rw[1] - (by0 + 3.5f);
        rw[MOM0 + 1] = X * m0 - mx;
        rw[MOM0 + 2] = Y * m0 - my;
        rw[MOM0 + 3] = (X * X) * m0 - 2.f * X * mx + mxx;
        rw[MOM0 + 4] = (X * Y) * m0 - X * my - Y * mx + mxy;
        rw[MOM0 + 5] = (Y * Y) * m0 - 2.f * Y * my + myy;
    }
    __builtin_amdgcn_wave_barrier();
    // flush: lanes walk the (instance, moment) table row-major, so one atomic instruction covers the consecutive
    // floats of several packed records -- the memory pipeline merges lanes that share a cache line into one
    // request instead of one per float.
    for (int idx = lane; idx < cnt * NM; idx += 64) {
        const int e = idx / NM, v = idx - e * NM;
        {
            const size_t g = __float_as_uint(qf[e * SF + 2]);
            atomic_add_f32(grad_acc + g * GRAD_RS + v, qf[e * SF + MOM0 + v]);
        }
    }
    __builtin_amdgcn_wave_barrier();   // the queue is rewritten by the next chunk / batch
    }   // chunks of the batch
    }   // batches of the unit
#ifdef GSR_TRACE_DETAIL
    if (trace && lane == 0) {
        uint64_t* tw = trace + ((size_t)unit * 4 + wave) * 4;
        tw[0] = t_start; tw[1] = t_head; tw[2] = wall_clock64();
        tw[3] = ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
    return;
#endif
    if (trace && lane == 0) {   // last wave to finish wins the end stamp (monotone clock, max via atomic)
        if (wave == 0) trace[2 * unit] = t_start;
        atomicMax((unsigned long long*)&trace[2 * unit + 1], (unsigned long long)wall_clock64());
    }
}

template <int C>
__global__ void __launch_bounds__(64)
blend_bwd_uniform_kernel(int W, int H, int gx, const uint4* __restrict__ unit_info, const float4* __restrict__ snap,
                 const uint2* __restrict__ masks, const uint32_t* __restrict__ point_list, const float4* __restrict__ rec_a,
                 const float4* __restrict__ rec_b, const RecTail<C>* __restrict__ rec_c, const float* __restrict__ bg,
                 const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                 const float* __restrict__ dL_dpix, float* __restrict__ grad_acc, uint64_t* __restrict__ trace,
                 const uint32_t* __restrict__ order)
{
    blend_bwd_unit<C>(W, H, gx, unit_info, snap, masks, point_list, rec_a, rec_b, rec_c, bg, final_T, n_contrib, dL_dpix, grad_acc, trace, order);
}
// Three channels: the register allocator is told to stay within six waves per SIMD (80 registers; left alone the per-channel
// form took 82 and the kernel ran five: 0.141 vs 0.131 ms on config C).  Six channels (104 registers) would have to spill 20 and lose.
#define GSR_BWD_SPECIALISE(CH, WAVES)                                                                                          \
    template <>                                                                                                               \
    __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, 8)))                                      \
    blend_bwd_uniform_kernel<CH>(int W, int H, int gx, const uint4* __restrict__ unit_info, const float4* __restrict__ snap,          \
                         const uint2* __restrict__ masks, const uint32_t* __restrict__ point_list,                            \
                         const float4* __restrict__ rec_a, const float4* __restrict__ rec_b,                                  \
                         const RecTail<CH>* __restrict__ rec_c, const float* __restrict__ bg,                                 \
                         const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,                           \
                         const float* __restrict__ dL_dpix, float* __restrict__ grad_acc, uint64_t* __restrict__ trace,       \
                         const uint32_t* __restrict__ order)                                                                  \
    {                                                                                                                         \
        blend_bwd_unit<CH>(W, H, gx, unit_info, snap, masks, point_list, rec_a, rec_b, rec_c, bg, final_T, n_contrib,         \
                           dL_dpix, grad_acc, trace, order);                                                                  \
    }
#ifndef GSR_BWD_WAVES3
#define GSR_BWD_WAVES3 6   // (the projected form takes 70 registers and runs seven; 8 = 64 registers spills eight and loses 10 us)
#endif
GSR_BWD_SPECIALISE(3, GSR_BWD_WAVES3)
#ifndef GSR_BWD_WAVES4
#define GSR_BWD_WAVES4 (GSR_BWD_PACK4 || GSR_BWD_B2_LDS ? 6 : 5)   // (PACK4: asked for six the compiler stops at 72 registers = seven waves, no scratch; asked for seven it spills 16 bytes)
#endif
// Four channels (two B tiles): 85 registers left alone = five waves; held at 80 for six it spilt ten and lost (0.146 vs 0.134 ms).
// Round 5: with the second B tile read from LDS per group (GSR_BWD_B2_LDS) it fits 80 registers without scratch = six waves.
GSR_BWD_SPECIALISE(4, GSR_BWD_WAVES4)


bool launch_blend_bwd_variant(int C, int W, int H, int U, const float* bg, ImageState im, BinState b, const float* dL_dpix, float* grad_acc,
                              hipStream_t st)
{
    const char* e_u = getenv("GSR_BWD_UNIFORM");   // (read per call: tools/ab_env.py flips it inside one process)
    if (e_u && e_u[0] == '0') return false;
    const Tiles t = tiles_of(W, H);
    if (U <= 0) return true;
    // Residency knob: extra dynamic LDS lowers the number of co-resident units per CU (tuning only).
    static const int pad = getenv("GSR_BWD_LDS_PAD") ? atoi(getenv("GSR_BWD_LDS_PAD")) : 0;
    uint64_t* tr = g_trace ? g_trace + 2 * (size_t)t.T : nullptr;
    const auto go = [&](auto tag) {
        constexpr int CC = decltype(tag)::value;
        blend_bwd_uniform_kernel<CC><<<4 * U, 64, pad, st>>>(W, H, t.gx, b.unit_info, b.snap, b.masks, b.point_list, b.rec_a, b.rec_b,
                                                             static_cast<const RecTail<CC>*>(b.rec_c), bg, im.final_T, im.n_contrib,
                                                             dL_dpix, grad_acc, tr, g_bwd_order);
    };
    if (C == 6) go(std::integral_constant<int, 6>{});
    else if (C == 4) go(std::integral_constant<int, 4>{});
    else go(std::integral_constant<int, 3>{});
    return true;
}

}  // namespace gsr
