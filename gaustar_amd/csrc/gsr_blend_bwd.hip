// gsr_blend_bwd.hip -- backward alpha compositing: per-Gaussian sums of the per-(pixel,Gaussian) terms.
//
// Per-pair arithmetic is the reference's renderCUDA backward (DGR/cuda_rasterizer/backward.cu:399-557;
// SURVEY.md section 9 item 10): back-to-front replay, T recovered by division, accum_rec recurrence,
// background term with T_final/(1-alpha), the 0.99 alpha clamp passing gradient as if unclamped.
//
// What differs is the decomposition and how the per-pair terms reach memory.  The reference runs one block
// per tile over the whole list and issues 9 global float atomicAdds per contributing (pixel, Gaussian) pair
// (backward.cu:523, :545-554).  Here
//
//  * the work unit is a (tile, SEGMENT of SEG list positions, 8x8 pixel block) triple, one wave64 each.
//    A pixel whose last contributor lies beyond the segment starts from the forward pass's snapshot at the
//    segment's far boundary: T = T_snap, accum_rec = (C_final - C_snap) / T_snap -- exactly the state the
//    reference's back-to-front recurrence has at that list position; a pixel that ends inside the segment
//    starts from (T_final, 0) like the reference; a pixel that ended before it is idle.  Units have bounded
//    size, so the dispatcher can balance them and nothing carries a 1 600-instance serial chain;
//  * lane i fetches instance i of the segment; instances that cannot reach alpha >= 1/255 inside the block
//    (exact test, block_min_half_quad) never enter the LDS queue;
//  * for a queued instance every lane evaluates its pixel and produces just TWO numbers,
//    w = alpha*T and r = G*dL_dalpha.  Everything the gradients need is a sum over the block's pixels of w or r
//    times a per-pixel constant:   sum w*dL_dpix_{r,g,b}   and   sum r*{1, x, y, x^2, xy, y^2}  (x, y = pixel
//    coordinates relative to the block centre).  That is a contraction over the 64 pixels,
//        [instances x pixels] . [pixels x 9],
//    and it runs on the matrix pipe: w and r are parked in LDS one row per instance, read back transposed, and
//    reduced by v_mfma_f32_16x16x4_f32 (exact f32 multiply-add, 16 issues cover 64 pixels) while the vector
//    ALU already works on the next instances.  This is the one place on the path that IS a contraction; it
//    replaces a 26-instruction cross-lane VALU reduction per instance (30 % of the kernel before);
//  * one lane per instance then re-centres the six spatial sums on the splat (dx = x_splat - x_pixel) --
//    dL_dcolor, dL_dopacity, dL_dmean2D and dL_dconic are fixed per-Gaussian linear maps of the nine moments,
//    applied once per Gaussian in geom_bwd -- and the table is flushed ROW-MAJOR: one atomic instruction covers
//    the nine consecutive floats of ~7 packed 48-byte records grad_acc[gaussian][12], so the memory pipeline
//    merges lanes per cache line (1.5 M atomic requests per 1080p view instead of 8 M).
#include "gsr_internal.h"
#include <cstdlib>

namespace gsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct __attribute__((aligned(16))) SlotB {   // 48 B per queued instance
    float4 a;   // x, y, conic_a, conic_b
    float4 b;   // conic_c, opacity, r, g
    float4 c;   // blue, list position (0-based, uint bits), gaussian id (uint bits), -
};

struct FetchedB { float4 a, b; float fr, fg, fb; uint32_t gid; };

__device__ __forceinline__ FetchedB fetch_instance_b(int k, int k_min, const uint32_t* __restrict__ list,
                                                     const float4* __restrict__ g0, const float4* __restrict__ g1,
                                                     const float* __restrict__ feats)
{
    FetchedB f;
    f.a = make_float4(0.f, 0.f, 1.f, 0.f);
    f.b = make_float4(1.f, 0.f, -1.f, 0.f);   // tau = -1: never kept
    f.fr = f.fg = f.fb = 0.f;
    f.gid = 0;
    if (k >= k_min) {
        f.gid = list[k];
        f.a = g0[f.gid];
        f.b = g1[f.gid];
        f.fr = feats[3 * (size_t)f.gid]; f.fg = feats[3 * (size_t)f.gid + 1]; f.fb = feats[3 * (size_t)f.gid + 2];
    }
    return f;
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int GRP = 8;          // instances per MFMA group: A-operand rows 0..7 carry their r, rows 8..15 their w
constexpr int RSTRIDE = 68;     // floats per row of the r|w table: 64 pixels + 4 (16-byte aligned, spreads banks)
// Once an instance's w and r are out, its queue slot only needs to keep x, y and the gaussian id: the nine
// moments overwrite the other nine floats of the 48-byte slot (no separate moment table -> more resident waves).
constexpr int SLOT_FLOATS = 12;
__device__ __forceinline__ int moment_off(int m) { return m < 8 ? 2 + m : 11; }   // skips x, y (0, 1) and the id (10)

static_assert(SEG == 64, "one fetch batch per unit");

__global__ void __launch_bounds__(64)
blend_bwd_kernel(int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ seg_off,
                 const uint32_t* __restrict__ unit_tile, const float4* __restrict__ snap,
                 const uint32_t* __restrict__ point_list, const float4* __restrict__ g0,
                 const float4* __restrict__ g1, const float* __restrict__ feats, const float* __restrict__ bg,
                 const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                 const float* __restrict__ dL_dpix, float* __restrict__ grad_acc, uint64_t* __restrict__ trace)
{
    const uint64_t t_start = trace ? wall_clock64() : 0;
    __shared__ SlotB queue[64];
    __shared__ __attribute__((aligned(16))) float Rm[2 * GRP * RSTRIDE];   // rows 0..7: r, rows 8..15: w, [row][pixel lane]
    float* const Wm = Rm + GRP * RSTRIDE;
    // one wave64 per workgroup: unit = (tile, segment), wave = 8x8 block of the tile
    const uint32_t unit = blockIdx.x >> 2;
    const int tile = (int)unit_tile[unit];
    const uint32_t unit0 = seg_off[tile];
    const int s0 = (int)(unit - unit0) * SEG;          // this unit covers list positions [s0, s1)
    const int wave = blockIdx.x & 3, lane = threadIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int sx = tx * TILE + (wave & 1) * SUB, sy = ty * TILE + (wave >> 1) * SUB;
    const int px = sx + (lane & 7), py = sy + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)sx, bx1 = (float)(sx + SUB - 1), by0 = (float)sy, by1 = (float)(sy + SUB - 1);

    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);
    const int s1 = min(s0 + SEG, n);
    const uint32_t* list = point_list + rg.x;
    SlotB* q = queue;
    float* const qf = reinterpret_cast<float*>(queue);

    const size_t pix = (size_t)W * py + px;
    const size_t HW = (size_t)H * W;
    const float T_final = inside ? final_T[pix] : 0.f;
    const int my_last = inside ? (int)n_contrib[pix] : 0;   // 1-based position of the last contributor
    float dpr = 0.f, dpg = 0.f, dpb = 0.f;
    if (inside) { dpr = dL_dpix[pix]; dpg = dL_dpix[HW + pix]; dpb = dL_dpix[2 * HW + pix]; }
    const float bg_dot_dpixel = bg[0] * dpr + bg[1] * dpg + bg[2] * dpb;

    // Per-pixel start state at the far end of the segment.
    float T = T_final;
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, last_alpha = 0.f, last_r = 0.f, last_g = 0.f, last_b = 0.f;
    const int my_lim = min(my_last, s1);                 // this pixel replays positions [s0, my_lim)
    if (my_last > s1) {
        // the pixel blended instances beyond this segment: resume from the forward's snapshot taken before
        // list position s1.  accum_rec at that point = colour composited behind s1, seen from s1.
        const int pidx = 16 * (py - ty * TILE) + (px - tx * TILE);
        const float4 sn = snap[(size_t)(unit + 1) * 256 + pidx];
        const float4 fin = snap[(size_t)unit0 * 256 + pidx];   // {C_final rgb, T_final} kept in the tile's first slot
        const float inv = __builtin_amdgcn_rcpf(sn.x);
        T = sn.x;
        acc_r = (fin.x - sn.y) * inv;
        acc_g = (fin.y - sn.z) * inv;
        acc_b = (fin.z - sn.w) * inv;
    }

    // Nothing behind the deepest position any pixel of this wave replays can matter.
    int wave_hi = my_lim;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wave_hi = max(wave_hi, __shfl_xor(wave_hi, d, 64));
    wave_hi = __builtin_amdgcn_readfirstlane(wave_hi);
    if (wave_hi <= s0) return;

    // lane l takes list position wave_hi-1-l: queue order == back-to-front order
    const int k = wave_hi - 1 - lane;
    const FetchedB cur = fetch_instance_b(k, s0, list, g0, g1, feats);

    // B operands of the contraction, constant over the unit.  MFMA step t (0..15) consumes the four pixels
    // p = 16*kap + t, kap = 0..3; in the B operand lane l carries row kap = l >> 4, column col = l & 15.
    // Columns 0..5 of the spatial operand: {1, x, y, x^2, xy, y^2} of pixel p relative to the block centre;
    // columns 0..2 of the colour operand: dL_dpix_{r,g,b} of pixel p (staged through LDS once).
    const int kap = lane >> 4, col = lane & 15;
    Rm[lane] = dpr; Rm[64 + lane] = dpg; Rm[128 + lane] = dpb; Rm[192 + lane] = 0.f;   // [channel][pixel], 4th = 0
    __builtin_amdgcn_wave_barrier();
    float Bf[16];   // columns 0..5 spatial (used by the r rows), 6..8 dL_dpix (used by the w rows), 9..15 zero
    {
        // spatial monomial of this lane's column as  base(y) + x*slope(y) + x^2*quad : x is a compile-time constant
        // per step, y takes two values per lane; all products are exact (small half-integers), one term non-zero
        const float ya = (float)(2 * kap) - 3.5f, yb = ya + 1.0f;
        const float quad = col == 3 ? 1.0f : 0.0f;
        const float base_a = col == 0 ? 1.0f : col == 2 ? ya : col == 5 ? ya * ya : 0.0f;
        const float base_b = col == 0 ? 1.0f : col == 2 ? yb : col == 5 ? yb * yb : 0.0f;
        const float slope_a = col == 1 ? 1.0f : col == 4 ? ya : 0.0f;
        const float slope_b = col == 1 ? 1.0f : col == 4 ? yb : 0.0f;
        const int ch = (col >= 6 && col < 9) ? col - 6 : 3;
        const float4* pd = reinterpret_cast<const float4*>(&Rm[ch * 64 + 16 * kap]);
        const bool spatial = col < 6;
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            const float4 v = pd[qd];
            const float dv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = 4 * qd + u;
                const float xr = (float)(t & 7) - 3.5f;
                const float sp = fmaf(xr * xr, quad, fmaf(xr, t < 8 ? slope_a : slope_b, t < 8 ? base_a : base_b));
                Bf[t] = spatial ? sp : dv[u];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();

    const bool keep = block_min_half_quad(cur.a.z, cur.a.w, cur.b.x, bx0 - cur.a.x, bx1 - cur.a.x, by0 - cur.a.y,
                                          by1 - cur.a.y) <= cur.b.z;
    const unsigned long long m = __ballot(keep);
    const int cnt = __popcll(m);
    if (keep) {
        const int slot = __popcll(m & ((1ull << lane) - 1ull));
        q[slot].a = cur.a;
        q[slot].b = make_float4(cur.b.x, cur.b.y, cur.fr, cur.fg);
        q[slot].c = make_float4(cur.fb, __uint_as_float((uint32_t)k), __uint_as_float(cur.gid), 0.f);
    }
    __builtin_amdgcn_wave_barrier();

    unsigned long long touched = 0ull;
    for (int g0i = 0; g0i < cnt; g0i += GRP) {
        // ---- vector ALU: w and r of GRP instances for this lane's pixel, parked row-wise in LDS
#pragma unroll
        for (int jj = 0; jj < GRP; jj++) {
            const int j = g0i + jj;
            float r = 0.f, w = 0.f;
            if (j < cnt) {
                const float4 A = q[j].a, B = q[j].b, Cc = q[j].c;
                const int pos = (int)__float_as_uint(Cc.y);
                const float dx = A.x - pxf, dy = A.y - pyf;
                const float power = pair_power(A.z, A.w, B.x, dx, dy);
                const float G = __expf(power);
                const float alpha = fminf(ALPHA_MAX, B.y * G);
                const bool live = pos < my_lim && power <= 0.0f && alpha >= ALPHA_MIN;
                if (__ballot(live) != 0ull) {
                    touched |= 1ull << j;
                    if (live) {
                        const float rinv = __builtin_amdgcn_rcpf(1.f - alpha);
                        T = T * rinv;
                        w = alpha * T;
                        acc_r = last_alpha * last_r + (1.f - last_alpha) * acc_r;
                        acc_g = last_alpha * last_g + (1.f - last_alpha) * acc_g;
                        acc_b = last_alpha * last_b + (1.f - last_alpha) * acc_b;
                        last_r = B.z; last_g = B.w; last_b = Cc.x;
                        float dL_dalpha = (B.z - acc_r) * dpr + (B.w - acc_g) * dpg + (Cc.x - acc_b) * dpb;
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        dL_dalpha -= (T_final * rinv) * bg_dot_dpixel;
                        r = G * dL_dalpha;
                    }
                }
            }
            Rm[jj * RSTRIDE + lane] = r;
            Wm[jj * RSTRIDE + lane] = w;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- matrix pipe: [GRP instances x 64 pixels] . [64 pixels x 16] for r (6 columns used) and w (3 used).
        // A operand: lane l carries instance row (l & 15) and, for step t, pixel 16*kap + t -> its 16 steps are
        // 16 consecutive floats of the row (four ds_read_b128).
        // two interleaved accumulators (even / odd steps) halve the dependent-accumulator chain
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        float ra[16];
        {
            const float4* pr = reinterpret_cast<const float4*>(&Rm[col * RSTRIDE + 16 * kap]);   // row col of r|w
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                const float4 v = pr[qd];
                ra[4 * qd] = v.x; ra[4 * qd + 1] = v.y; ra[4 * qd + 2] = v.z; ra[4 * qd + 3] = v.w;
            }
        }
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t], Bf[t], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t + 1], Bf[t + 1], acc1, 0, 0, 0);
        }
        // D layout: lane l, register i -> operand row 4*(l >> 4) + i, column l & 15.
        // rows 0..7: r of instance row, columns 0..5 = spatial sums; rows 8..15: w of instance row-8, columns 6..8.
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = 4 * kap + i, inst = g0i + (row & 7);
            const bool take = row < GRP ? col < 6 : (col >= 6 && col < 9);
            if (take && inst < cnt) qf[inst * SLOT_FLOATS + moment_off(col)] = acc0[i] + acc1[i];
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- lane = queued instance: re-centre the spatial sums on the splat (dx = x_splat - x_pixel) and lay the
    // nine moments out in grad_acc order {c_r, c_g, c_b, sum r, sum r dx, sum r dy, sum r dx^2, sum r dx dy, sum r dy^2}
    if (lane < cnt && ((touched >> lane) & 1ull)) {
        float* rw = &qf[lane * SLOT_FLOATS];
        const float m0 = rw[moment_off(0)], mx = rw[moment_off(1)], my = rw[moment_off(2)], mxx = rw[moment_off(3)],
                    mxy = rw[moment_off(4)], myy = rw[moment_off(5)];
        const float c0 = rw[moment_off(6)], c1 = rw[moment_off(7)], c2 = rw[moment_off(8)];
        const float X = rw[0] - (bx0 + 3.5f), Y = rw[1] - (by0 + 3.5f);
        rw[moment_off(0)] = c0; rw[moment_off(1)] = c1; rw[moment_off(2)] = c2;
        rw[moment_off(3)] = m0;
        rw[moment_off(4)] = X * m0 - mx;
        rw[moment_off(5)] = Y * m0 - my;
        rw[moment_off(6)] = (X * X) * m0 - 2.f * X * mx + mxx;
        rw[moment_off(7)] = (X * Y) * m0 - X * my - Y * mx + mxy;
        rw[moment_off(8)] = (Y * Y) * m0 - 2.f * Y * my + myy;
    }
    __builtin_amdgcn_wave_barrier();
    // flush: lanes walk the (instance, moment) table row-major, so one atomic instruction covers the nine
    // consecutive floats of ~7 packed records -- the memory pipeline merges lanes that share a cache line
    // into one request instead of nine.
    for (int idx = lane; idx < cnt * 9; idx += 64) {
        const int e = idx / 9, v = idx - e * 9;
        if ((touched >> e) & 1ull) {
            const size_t g = __float_as_uint(qf[e * SLOT_FLOATS + 10]);
            atomic_add_f32(grad_acc + g * 12 + v, qf[e * SLOT_FLOATS + moment_off(v)]);
        }
    }
    if (trace && lane == 0) {   // last wave to finish wins the end stamp (monotone clock, max via atomic)
        if (wave == 0) trace[2 * unit] = t_start;
        atomicMax((unsigned long long*)&trace[2 * unit + 1], (unsigned long long)wall_clock64());
    }
}

void launch_blend_bwd(int W, int H, int U, const float* bg, const float* feats, GeomState g, ImageState im, BinState b,
                      const float* dL_dpix, float* grad_acc, hipStream_t st)
{
    const Tiles t = tiles_of(W, H);
    if (U <= 0) return;
    // Residency knob: extra dynamic LDS lowers the number of co-resident units per CU (tuning only).
    static const int pad = getenv("GSR_BWD_LDS_PAD") ? atoi(getenv("GSR_BWD_LDS_PAD")) : 0;
    blend_bwd_kernel<<<4 * U, 64, pad, st>>>(W, H, t.gx, im.ranges, im.seg_off, b.unit_tile, b.snap, b.point_list,
                                             g.g0, g.g1, feats, bg, im.final_T, im.n_contrib, dL_dpix, grad_acc,
                                             g_trace ? g_trace + 2 * (size_t)t.T : nullptr);
}

}  // namespace gsr
