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
//  * the work unit is a (tile, SEGMENT of SEG list positions) pair.  A pixel whose last contributor lies
//    beyond the segment starts from the forward pass's snapshot at the segment's far boundary:
//    T = T_snap, accum_rec = (C_final - C_snap) / T_snap -- exactly the state the reference's back-to-front
//    recurrence has at that list position; a pixel that ends inside the segment starts from (T_final, 0)
//    like the reference; a pixel that ended before it is idle.  Units have bounded size, so the dispatcher
//    can balance them and no workgroup carries a 1 600-instance serial chain;
//  * inside a unit a wave64 owns an 8x8 pixel block and walks the segment back-to-front 64 instances at a
//    time (ids fetched two batches ahead, records one); instances that cannot reach alpha >= 1/255 inside the
//    block (exact test, block_min_half_quad) never enter the per-wave LDS queue;
//  * for a queued instance every lane evaluates its pixel; what is summed over pixels is reduced to
//    NINE linear moments  {sum w*dL_dpix_rgb, sum r, sum r*dx, sum r*dy, sum r*dx^2, sum r*dx*dy, sum r*dy^2}
//    (w = alpha*T, r = G*dL_dalpha): dL_dcolor, dL_dopacity, dL_dmean2D and dL_dconic are fixed linear
//    maps of them with per-Gaussian coefficients, applied once per Gaussian in geom_bwd;
//  * the nine values are reduced across the 64 lanes in registers by a TRANSPOSING reduction:
//    v_permlane32_swap / v_permlane16_swap + add fold eight values into one register (8 lanes per
//    value), three DPP steps finish it -- 18 VALU ops for 8 values instead of 8 x 6 shuffle-adds;
//  * totals are parked in a per-wave LDS table (one row per queued instance) and flushed once per batch
//    ROW-MAJOR: one atomic instruction covers the nine consecutive floats of ~7 packed 48-byte records
//    grad_acc[gaussian][12], so the memory pipeline merges lanes per cache line (1.5 M atomic requests per
//    1080p view instead of 8 M).
#include "gsr_internal.h"
#include <cstdlib>

namespace gsr {

struct __attribute__((aligned(16))) SlotB {   // 48 B per queued instance
    float4 a;   // x, y, conic_a, conic_b
    float4 b;   // conic_c, opacity, r, g
    float4 c;   // blue, list position (0-based, uint bits), gaussian id (uint bits), -
};

struct FetchedB { float4 a, b; float fr, fg, fb; uint32_t gid; };

// Two-stage software pipeline over the dependent gather (list -> id -> records), see gsr_blend_fwd.hip.
__device__ __forceinline__ uint32_t fetch_id_b(int k, int k_min, const uint32_t* __restrict__ list)
{
    return k >= k_min ? list[k] : 0xffffffffu;
}
__device__ __forceinline__ FetchedB fetch_record_b(uint32_t gid, const float4* __restrict__ g0,
                                                   const float4* __restrict__ g1, const float* __restrict__ feats)
{
    FetchedB f;
    f.a = make_float4(0.f, 0.f, 1.f, 0.f);
    f.b = make_float4(1.f, 0.f, -1.f, 0.f);   // tau = -1: never kept
    f.fr = f.fg = f.fb = 0.f;
    f.gid = gid;
    if (gid != 0xffffffffu) {
        f.a = g0[gid];
        f.b = g1[gid];
        f.fr = feats[3 * (size_t)gid]; f.fg = feats[3 * (size_t)gid + 1]; f.fb = feats[3 * (size_t)gid + 2];
    }
    return f;
}

// ---- cross-lane helpers -----------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v)   // full-wave lane permutation (no masked lanes)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// lanes 0-31: a[l] + a[l+32]   |   lanes 32-63: b[l-32] + b[l]
__device__ __forceinline__ float fold32(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// even rows: a[row] + a[row+1]   |   odd rows: b[row-1] + b[row]      (rows of 16 lanes)
__device__ __forceinline__ float fold16(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Sum eight per-lane values over the wave; afterwards lane l holds the total of value (l >> 3).
__device__ __forceinline__ float reduce8(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                         float v7, bool hi8)
{
    const float a0 = fold32(v0, v4), a1 = fold32(v1, v5), a2 = fold32(v2, v6), a3 = fold32(v3, v7);
    const float b0 = fold16(a0, a2);   // rows: v0 v2 v4 v6
    const float b1 = fold16(a1, a3);   // rows: v1 v3 v5 v7
    const float keep = hi8 ? b1 : b0, send = hi8 ? b0 : b1;
    float c = keep + dpp_perm<0x128>(send);   // row_ror:8  -> lanes 0-7 of a row: even value, 8-15: odd value
    c += dpp_perm<0x141>(c);                  // row_half_mirror
    c += dpp_perm<0xB1>(c);                   // quad_perm [1,0,3,2]
    c += dpp_perm<0x4E>(c);                   // quad_perm [2,3,0,1]
    return c;
}
// Sum one per-lane value over the wave; the total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_masked(float v)
{
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float reduce1_to_lane63(float v)
{
    v += dpp_perm<0x111>(v);              // row_shr:1 (bound_ctrl: lanes shifted in from outside the row read 0)
    v += dpp_perm<0x112>(v);              // row_shr:2
    v += dpp_perm<0x114>(v);              // row_shr:4
    v += dpp_perm<0x118>(v);              // row_shr:8   -> lane 15 of each row = row total
    v = dpp_add_masked<0x142, 0xa>(v);    // row_bcast:15 into rows 1,3
    v = dpp_add_masked<0x143, 0xc>(v);    // row_bcast:31 into rows 2,3
    return v;
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int ACC_STRIDE = 9;

__global__ void __launch_bounds__(64)
blend_bwd_kernel(int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ seg_off,
                 const uint32_t* __restrict__ unit_tile, const float4* __restrict__ snap,
                 const uint32_t* __restrict__ point_list, const float4* __restrict__ g0,
                 const float4* __restrict__ g1, const float* __restrict__ feats, const float* __restrict__ bg,
                 const float* __restrict__ final_T,
                 const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                 float* __restrict__ grad_acc, uint64_t* __restrict__ trace)
{
    const uint64_t t_start = trace ? wall_clock64() : 0;
    __shared__ SlotB queue[64];
    __shared__ float totals[64 * ACC_STRIDE];
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
    const bool hi8 = (lane & 8) != 0;

    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);
    const int s1 = min(s0 + SEG, n);
    const uint32_t* list = point_list + rg.x;
    SlotB* q = queue;
    float* tot = totals;

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

    // lane l takes list position hi-1-l: queue order == back-to-front order
    FetchedB nxt = fetch_record_b(fetch_id_b(wave_hi - 1 - lane, s0, list), g0, g1, feats);
    uint32_t gid_nxt = fetch_id_b(wave_hi - 65 - lane, s0, list);
    for (int hi = wave_hi; hi > s0; hi -= 64) {
        const FetchedB cur = nxt;
        const int k = hi - 1 - lane;
        nxt = fetch_record_b(gid_nxt, g0, g1, feats);   // records of the next batch (ids arrived during the last one)
        gid_nxt = fetch_id_b(k - 128, s0, list);        // ids two batches ahead
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
        for (int j = 0; j < cnt; j++) {
            const float4 A = q[j].a, B = q[j].b, Cc = q[j].c;
            const int pos = (int)__float_as_uint(Cc.y);
            const float dx = A.x - pxf, dy = A.y - pyf;
            const float power = pair_power(A.z, A.w, B.x, dx, dy);
            const float G = __expf(power);
            const float alpha = fminf(ALPHA_MAX, B.y * G);
            const bool live = pos < my_lim && power <= 0.0f && alpha >= ALPHA_MIN;
            if (__ballot(live) == 0ull) continue;

            float v_cr = 0.f, v_cg = 0.f, v_cb = 0.f, v_r = 0.f, v_rx = 0.f, v_ry = 0.f, v_rxx = 0.f, v_rxy = 0.f,
                  v_ryy = 0.f;
            if (live) {
                const float rinv = __builtin_amdgcn_rcpf(1.f - alpha);
                T = T * rinv;
                const float w = alpha * T;
                acc_r = last_alpha * last_r + (1.f - last_alpha) * acc_r;
                acc_g = last_alpha * last_g + (1.f - last_alpha) * acc_g;
                acc_b = last_alpha * last_b + (1.f - last_alpha) * acc_b;
                last_r = B.z; last_g = B.w; last_b = Cc.x;
                float dL_dalpha = (B.z - acc_r) * dpr + (B.w - acc_g) * dpg + (Cc.x - acc_b) * dpb;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha -= (T_final * rinv) * bg_dot_dpixel;
                v_cr = w * dpr; v_cg = w * dpg; v_cb = w * dpb;
                v_r = G * dL_dalpha;
                v_rx = v_r * dx; v_ry = v_r * dy;
                v_rxx = v_rx * dx; v_rxy = v_rx * dy; v_ryy = v_ry * dy;
            }
#ifdef GSR_EXP_NO_REDUCE
            const float c8 = v_cr + v_cg + v_cb + v_r + v_rx + v_ry + v_rxx + v_rxy, c1 = v_ryy;
#else
            const float c8 = reduce8(v_cr, v_cg, v_cb, v_r, v_rx, v_ry, v_rxx, v_rxy, hi8);
            const float c1 = reduce1_to_lane63(v_ryy);
#endif
            if ((lane & 7) == 0) tot[j * ACC_STRIDE + (lane >> 3)] = c8;
            if (lane == 63) tot[j * ACC_STRIDE + 8] = c1;
            touched |= 1ull << j;
        }
        __builtin_amdgcn_wave_barrier();
        // flush: lanes walk the (instance, moment) table row-major, so one atomic instruction covers the nine
        // consecutive floats of ~7 packed records -- the memory pipeline merges lanes that share a cache line
        // into one request instead of nine.
        for (int idx = lane; idx < cnt * ACC_STRIDE; idx += 64) {
            const int e = idx / ACC_STRIDE, v = idx - e * ACC_STRIDE;
            if ((touched >> e) & 1ull) {
                const size_t g = __float_as_uint(q[e].c.z);
#ifdef GSR_EXP_NO_ATOMICS
                if (tot[idx] == 123.456f) grad_acc[g * 12 + v] = tot[idx];
#else
                atomic_add_f32(grad_acc + g * 12 + v, tot[idx]);
#endif
            }
        }
        __builtin_amdgcn_wave_barrier();
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
    blend_bwd_kernel<<<4 * U, 64, pad, st>>>(W, H, t.gx, im.ranges, im.seg_off, b.unit_tile, b.snap, b.point_list, g.g0,
                                          g.g1, feats, bg, im.final_T, im.n_contrib, dL_dpix, grad_acc,
                                          g_trace ? g_trace + 2 * (size_t)t.T : nullptr);
}

}  // namespace gsr
