// gsr_blend_fwd.hip -- forward alpha compositing.
//
// Same per-pixel arithmetic and control flow as the reference's renderCUDA
// (DGR/cuda_rasterizer/forward.cu:261-374), re-organised for CDNA4:
//
//  * One wave64 owns an 8x8 pixel block; a 16x16 tile is four independent waves (no workgroup
//    barriers, each wave stops as soon as its own 64 pixels are saturated).
//  * Each wave walks the tile's depth-sorted list 64 instances at a time: lane i fetches instance i
//    (coalesced id load + two 16-byte record gathers + colour, issued one batch AHEAD so the gather
//    latency hides behind the blend loop), tests exactly whether the splat can reach alpha >= 1/255
//    anywhere in the wave's 8x8 block (block_min_half_quad), and the survivors are compacted into a
//    per-wave LDS queue with a ballot + prefix-popcount.  With surface splats of ~4 px radius this drops
//    most of the (Gaussian, pixel) pairs the reference evaluates only to reject.
//  * The blend loop reads survivors from LDS at wave-uniform addresses (broadcast ds_read_b128) --
//    position, conic, opacity AND colour come from LDS; the reference gathers colour from global memory
//    per pixel (forward.cu:355).  It is unrolled four survivors deep and branch-free: the four Gaussian
//    exponents/alphas are independent and evaluated together, only the short T-update chain is serial.
//    A tile's list is a serial dependency per pixel, so the kernel's tail is the LATENCY of the longest
//    list on a nearly empty chip; instruction-level parallelism, not occupancy, is what shortens it.
//
// n_contrib stores the 1-based list position of the last blended instance, as the reference does.
#include "gsr_internal.h"

namespace gsr {

struct __attribute__((aligned(16))) Slot {   // 48 B per queued instance
    float4 a;   // x, y, conic_a, conic_b
    float4 b;   // conic_c, opacity, r, g
    float4 c;   // blue, list position + 1 (as uint bits), -, -
};

struct Fetched { float4 a, b; float fr, fg, fb; };

// Two-stage software pipeline over the dependent gather (list -> id -> records): ids are fetched two batches
// ahead, records one batch ahead, so neither load latency sits on the per-batch critical path.
__device__ __forceinline__ uint32_t fetch_id(uint32_t k, uint32_t n, const uint32_t* __restrict__ list)
{
    return k < n ? list[k] : 0xffffffffu;
}
__device__ __forceinline__ Fetched fetch_record(uint32_t gid, const float4* __restrict__ g0,
                                                const float4* __restrict__ g1, const float* __restrict__ feats)
{
    Fetched f;
    f.a = make_float4(0.f, 0.f, 1.f, 0.f);
    f.b = make_float4(1.f, 0.f, -1.f, 0.f);   // tau = -1: never kept
    f.fr = f.fg = f.fb = 0.f;
    if (gid != 0xffffffffu) {
        f.a = g0[gid];
        f.b = g1[gid];
        f.fr = feats[3 * (size_t)gid]; f.fg = feats[3 * (size_t)gid + 1]; f.fb = feats[3 * (size_t)gid + 2];
    }
    return f;
}

__global__ void __launch_bounds__(256)
blend_fwd_kernel(int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ order,
                 const uint32_t* __restrict__ point_list,
                 const float4* __restrict__ g0, const float4* __restrict__ g1, const float* __restrict__ feats,
                 const float* __restrict__ bg, float* __restrict__ out_color, float* __restrict__ final_T,
                 uint32_t* __restrict__ n_contrib, uint64_t* __restrict__ trace)
{
    const uint64_t t_start = trace ? wall_clock64() : 0;
    __shared__ Slot queue[4][64 + 4];   // +4 neutral slots so the 4-deep loop needs no tail handling
    const int tile = (int)order[blockIdx.x];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = tile % gx, ty = tile / gx;
    const int sx = tx * TILE + (wave & 1) * SUB, sy = ty * TILE + (wave >> 1) * SUB;
    const int px = sx + (lane & 7), py = sy + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)sx, bx1 = (float)(sx + SUB - 1), by0 = (float)sy, by1 = (float)(sy + SUB - 1);

    const uint2 rg = ranges[tile];
    const uint32_t n = rg.y - rg.x;
    const uint32_t* list = point_list + rg.x;
    Slot* q = queue[wave];

    float T = 1.0f, Cr = 0.f, Cg = 0.f, Cb = 0.f;
    uint32_t last = 0;
    bool done = !inside;

    Fetched nxt = fetch_record(fetch_id(lane, n, list), g0, g1, feats);
    uint32_t gid_nxt = fetch_id(64 + lane, n, list);
    for (uint32_t base = 0; base < n; base += 64) {
        if (__ballot(!done) == 0ull) break;
        const Fetched cur = nxt;
        nxt = fetch_record(gid_nxt, g0, g1, feats);         // records of batch +1 (ids arrived during the last batch)
        gid_nxt = fetch_id(base + 128 + lane, n, list);     // ids of batch +2
        const uint32_t k = base + lane;
        const bool keep = block_min_half_quad(cur.a.z, cur.a.w, cur.b.x, bx0 - cur.a.x, bx1 - cur.a.x, by0 - cur.a.y,
                                              by1 - cur.a.y) <= cur.b.z;
        const unsigned long long m = __ballot(keep);
        const int cnt = __popcll(m);
        if (keep) {
            const int slot = __popcll(m & ((1ull << lane) - 1ull));
            q[slot].a = cur.a;
            q[slot].b = make_float4(cur.b.x, cur.b.y, cur.fr, cur.fg);
            q[slot].c = make_float4(cur.fb, __uint_as_float(k + 1), 0.f, 0.f);
        }
        if (lane < 4) {   // neutral padding behind the survivors: opacity 0 never passes the alpha test
            q[cnt + lane].a = make_float4(0.f, 0.f, 0.f, 0.f);
            q[cnt + lane].b = make_float4(0.f, 0.f, 0.f, 0.f);
            q[cnt + lane].c = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_wave_barrier();
        for (int j = 0; j < cnt; j += 4) {
            float4 A[4], B[4], Cc[4];
            float alpha[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { A[u] = q[j + u].a; B[u] = q[j + u].b; Cc[u] = q[j + u].c; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float dx = A[u].x - pxf, dy = A[u].y - pyf;
                const float power = pair_power(A[u].z, A[u].w, B[u].x, dx, dy);
                alpha[u] = fminf(ALPHA_MAX, B[u].y * __expf(power));
                ok[u] = power <= 0.0f && alpha[u] >= ALPHA_MIN;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float test_T = T * (1.0f - alpha[u]);
                const bool live = ok[u] && !done;
                const bool stop = live && test_T < T_EPS;
                const bool upd = live && !stop;
                done = done || stop;
                const float w = upd ? alpha[u] * T : 0.0f;
                Cr += B[u].z * w; Cg += B[u].w * w; Cb += Cc[u].x * w;
                T = upd ? test_T : T;
                last = upd ? __float_as_uint(Cc[u].y) : last;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (inside) {
        const size_t pix = (size_t)W * py + px;
        const size_t HW = (size_t)H * W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = Cr + T * bg[0];
        out_color[HW + pix] = Cg + T * bg[1];
        out_color[2 * HW + pix] = Cb + T * bg[2];
    }
    if (trace && lane == 0) {   // last wave to finish wins the end stamp
        if (wave == 0) trace[2 * blockIdx.x] = t_start;
        atomicMax((unsigned long long*)&trace[2 * blockIdx.x + 1], (unsigned long long)wall_clock64());
    }
}

void launch_blend_fwd(int W, int H, const float* bg, const float* feats, GeomState g, ImageState im, BinState b,
                      float* out_color, hipStream_t st)
{
    const Tiles t = tiles_of(W, H);
    blend_fwd_kernel<<<t.T, 256, 0, st>>>(W, H, t.gx, im.ranges, im.order, b.point_list, g.g0, g.g1, feats, bg, out_color,
                                          im.final_T, im.n_contrib, g_trace);
}

}  // namespace gsr
