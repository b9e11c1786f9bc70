// gsr_blend_bwd.hip -- backward alpha compositing: dL/d{colour, mean2D, conic, opacity} per Gaussian.
//
// Per-pair arithmetic is the reference's renderCUDA backward (DGR/cuda_rasterizer/backward.cu:399-557;
// SURVEY.md section 9 item 10): back-to-front replay, T recovered by division, accum_rec recurrence,
// background term with T_final/(1-alpha), 0.5*W / 0.5*H pixel->NDC scale on the mean gradient, the
// 0.99 alpha clamp passing gradient as if unclamped.
//
// What differs is how the per-pair terms reach memory.  The reference issues 9 global float
// atomicAdds per contributing (pixel, Gaussian) pair (backward.cu:523, :545-554).  Here a wave64
// owns an 8x8 pixel block, every lane evaluates the same queued instance, the nine partial sums are
// reduced across the 64 lanes in registers with DPP row shifts / row broadcasts, and one lane issues
// the nine atomics -- at most one flush per (Gaussian, 8x8 block), skipped entirely when no lane of
// the wave contributed.  Instances whose alpha >= 1/255 box misses the block never enter the queue.
#include "gsr_internal.h"

namespace gsr {

struct __attribute__((aligned(16))) SlotB {   // 48 B per queued instance
    float4 a;   // x, y, conic_a, conic_b
    float4 b;   // conic_c, opacity, r, g
    float4 c;   // blue, list position (0-based, uint bits), gaussian id (uint bits), -
};

// Sum over the 64 lanes of a wave; the total lands in lane 63 (classic GCN DPP reduction:
// row_shr 1,2,3 + row_shr 4,8 with bank masks, then row_bcast 15 / 31).
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = dpp_add<0x111>(v);               // row_shr:1
    v = dpp_add<0x112>(v);               // row_shr:2
    v = dpp_add<0x114, 0xf, 0xe>(v);     // row_shr:4  bank_mask 0xe
    v = dpp_add<0x118, 0xf, 0xc>(v);     // row_shr:8  bank_mask 0xc
    v = dpp_add<0x142, 0xa, 0xf>(v);     // row_bcast:15 row_mask 0xa
    v = dpp_add<0x143, 0xc, 0xf>(v);     // row_bcast:31 row_mask 0xc
    return v;
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(256)
blend_bwd_kernel(int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                 const float4* __restrict__ g0, const float4* __restrict__ g1, const float* __restrict__ feats,
                 const float* __restrict__ bg, const float* __restrict__ final_T,
                 const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                 float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
                 float* __restrict__ dL_dcolor)
{
    __shared__ SlotB queue[4][64];
    const int tile = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = tile % gx, ty = tile / gx;
    const int sx = tx * TILE + (wave & 1) * SUB, sy = ty * TILE + (wave >> 1) * SUB;
    const int px = sx + (lane & 7), py = sy + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)sx, bx1 = (float)(sx + SUB - 1), by0 = (float)sy, by1 = (float)(sy + SUB - 1);

    const uint2 rg = ranges[tile];
    SlotB* q = queue[wave];

    const size_t pix = (size_t)W * py + px;
    const size_t HW = (size_t)H * W;
    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t my_last = inside ? n_contrib[pix] : 0u;   // 1-based position of the last contributor
    float dpr = 0.f, dpg = 0.f, dpb = 0.f;
    if (inside) { dpr = dL_dpix[pix]; dpg = dL_dpix[HW + pix]; dpb = dL_dpix[2 * HW + pix]; }
    const float bg_dot_dpixel = bg[0] * dpr + bg[1] * dpg + bg[2] * dpb;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    // Nothing behind the deepest contributor of this wave can matter.
    uint32_t wave_last = my_last;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, d, 64));
    wave_last = __builtin_amdgcn_readfirstlane(wave_last);

    float T = T_final;
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, last_alpha = 0.f, last_r = 0.f, last_g = 0.f, last_b = 0.f;

    for (int hi = (int)wave_last; hi > 0; hi -= 64) {
        // lane l takes list position hi-1-l: queue order == back-to-front order
        const int k = hi - 1 - lane;
        bool keep = false;
        float4 ra, rb;
        uint32_t gid = 0;
        if (k >= 0) {
            gid = point_list[rg.x + k];
            ra = g0[gid];
            rb = g1[gid];
            const float ddx = fmaxf(fmaxf(bx0 - ra.x, ra.x - bx1), 0.0f);
            const float ddy = fmaxf(fmaxf(by0 - ra.y, ra.y - by1), 0.0f);
            keep = ddx <= rb.z && ddy <= rb.w;
        }
        const unsigned long long m = __ballot(keep);
        const int cnt = __popcll(m);
        if (keep) {
            const int slot = __popcll(m & ((1ull << lane) - 1ull));
            const float fr = feats[3 * (size_t)gid], fg = feats[3 * (size_t)gid + 1], fb = feats[3 * (size_t)gid + 2];
            q[slot].a = ra;
            q[slot].b = make_float4(rb.x, rb.y, fr, fg);
            q[slot].c = make_float4(fb, __uint_as_float((uint32_t)k), __uint_as_float(gid), 0.f);
        }
        __builtin_amdgcn_wave_barrier();
        for (int j = 0; j < cnt; j++) {
            const float4 A = q[j].a, B = q[j].b, Cc = q[j].c;
            const uint32_t pos = __float_as_uint(Cc.y);
            const float dx = A.x - pxf, dy = A.y - pyf;
            const float power = pair_power(A.z, A.w, B.x, dx, dy);
            const float G = __expf(power);
            const float alpha = fminf(ALPHA_MAX, B.y * G);
            const bool live = pos < my_last && power <= 0.0f && alpha >= ALPHA_MIN;
            if (__ballot(live) == 0ull) continue;

            float v_cr = 0.f, v_cg = 0.f, v_cb = 0.f, v_mx = 0.f, v_my = 0.f, v_ca = 0.f, v_cb2 = 0.f, v_cc = 0.f,
                  v_op = 0.f;
            if (live) {
                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                acc_r = last_alpha * last_r + (1.f - last_alpha) * acc_r;
                acc_g = last_alpha * last_g + (1.f - last_alpha) * acc_g;
                acc_b = last_alpha * last_b + (1.f - last_alpha) * acc_b;
                last_r = B.z; last_g = B.w; last_b = Cc.x;
                float dL_dalpha = (B.z - acc_r) * dpr + (B.w - acc_g) * dpg + (Cc.x - acc_b) * dpb;
                v_cr = dchannel_dcolor * dpr; v_cg = dchannel_dcolor * dpg; v_cb = dchannel_dcolor * dpb;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                const float dL_dG = B.y * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * A.z - gdy * A.w;
                const float dG_ddely = -gdy * B.x - gdx * A.w;
                v_mx = dL_dG * dG_ddelx * ddelx_dx;
                v_my = dL_dG * dG_ddely * ddely_dy;
                v_ca = -0.5f * gdx * dx * dL_dG;
                v_cb2 = -0.5f * gdx * dy * dL_dG;
                v_cc = -0.5f * gdy * dy * dL_dG;
                v_op = G * dL_dalpha;
            }
            v_cr = wave_sum_to_lane63(v_cr); v_cg = wave_sum_to_lane63(v_cg); v_cb = wave_sum_to_lane63(v_cb);
            v_mx = wave_sum_to_lane63(v_mx); v_my = wave_sum_to_lane63(v_my);
            v_ca = wave_sum_to_lane63(v_ca); v_cb2 = wave_sum_to_lane63(v_cb2); v_cc = wave_sum_to_lane63(v_cc);
            v_op = wave_sum_to_lane63(v_op);
            if (lane == 63) {
                const size_t g = __float_as_uint(Cc.z);
                atomic_add_f32(&dL_dcolor[3 * g + 0], v_cr);
                atomic_add_f32(&dL_dcolor[3 * g + 1], v_cg);
                atomic_add_f32(&dL_dcolor[3 * g + 2], v_cb);
                atomic_add_f32(&dL_dmean2D[3 * g + 0], v_mx);
                atomic_add_f32(&dL_dmean2D[3 * g + 1], v_my);
                atomic_add_f32(&dL_dconic[4 * g + 0], v_ca);
                atomic_add_f32(&dL_dconic[4 * g + 1], v_cb2);
                atomic_add_f32(&dL_dconic[4 * g + 3], v_cc);
                atomic_add_f32(&dL_dopacity[g], v_op);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

void launch_blend_bwd(int W, int H, const float* bg, const float* feats, GeomState g, ImageState im, BinState b,
                      const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                      hipStream_t st)
{
    const Tiles t = tiles_of(W, H);
    blend_bwd_kernel<<<t.T, 256, 0, st>>>(W, H, t.gx, im.ranges, b.point_list, g.g0, g.g1, feats, bg, im.final_T,
                                          im.n_contrib, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor);
}

}  // namespace gsr
