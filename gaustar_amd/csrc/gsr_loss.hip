// gsr_loss.hip -- the image-space losses either side of the rasterizer, fused.
//
// Reference: gaustar_utils/loss_utils.py:17-62 (l1_loss, gaussian / create_window, ssim / _ssim) combined as
// (1 - f) * l1 + f * (1 - ssim) in gaustar_trainers/refine.py:451-453, evaluated on a margin-cropped view
// (:584-594), and the masked depth / silhouette L1 terms of refine.py:634-660.  The reference evaluates SSIM as
// five depthwise 11x11 conv2d calls plus ~20 elementwise kernels and lets autograd build the backward (another
// five convolutions); at 1080p that is several milliseconds around a rasterizer that now takes 0.4.
//
// Here: two tiled passes, each a separable 11-tap Gaussian in LDS.
//   pass 1  x, y tile (+5 halo, zero outside the crop like conv2d's zero padding) -> mu_x, mu_y, E[x^2 + y^2], E[xy]
//           -> SSIM map value S and its three partial derivatives w.r.t. the x-dependent window moments
//              D1 = dS/dmu_x, D2 = dS/dE[x^2], D3 = dS/dE[xy]   (closed form, below);
//           per-workgroup partial sums of S and |x - y|.
//   pass 2  the adjoint of the window (the window is symmetric): dS_sum/dx(p) = (w*D1)(p) + 2 x(p) (w*D2)(p) + y(p) (w*D3)(p)
//           and dL/dx = (1-f)/N sign(x-y) - f/N dS_sum/dx, written in the planar [C,H,W] layout the backward blend reads.
// Partial sums are reduced in a fixed order in double: the loss value is deterministic.
#include "gsr_internal.h"

namespace gsr {

constexpr int LR = 5;                 // window radius (window_size 11, loss_utils.py:36)
#ifndef GSR_SSIM_TILE_H
#define GSR_SSIM_TILE_H 16
#endif
constexpr int LT = 32;                // output tile width
constexpr int LTY = GSR_SSIM_TILE_H;  // output tile height (16: 22 KB of LDS per workgroup in pass 1 -> 7 workgroups per CU)
constexpr int LI = LT + 2 * LR;       // input tile width (42)
constexpr int LIY = LTY + 2 * LR;     // input tile height
constexpr int LP = LI + 1;            // padded LDS row of the input tile
constexpr int RPT = LTY / 8;          // output rows per thread in the vertical pass (256 threads = 32 columns x 8 groups)
#ifndef GSR_SSIM_HS
#define GSR_SSIM_HS 4
#endif
constexpr int HS = GSR_SSIM_HS;       // consecutive output columns per thread in the horizontal pass
static_assert(LT % HS == 0, "a row is a whole number of horizontal tasks");
static_assert(LTY % 8 == 0, "eight row groups");
// gaussian(11, 1.5) exactly as loss_utils.py:23-25 builds it (float32 of exp(), divided by the float32 sum)
__device__ constexpr float GW[11] = {0x1.0d956cp-10f, 0x1.f1fe02p-8f, 0x1.26eb18p-5f, 0x1.bff0fep-4f, 0x1.b43c3ep-3f,
                                     0x1.10656p-2f,   0x1.b43c3ep-3f, 0x1.bff0fep-4f, 0x1.26eb18p-5f, 0x1.f1fe02p-8f,
                                     0x1.0d956cp-10f};
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;   // loss_utils.py:54-55
// One output of the 11-tap window.  The window is symmetric (GW[k] == GW[10 - k], bit for bit), so the taps are paired:
// five additions (the cheap instruction class) + six multiply-adds instead of eleven multiply-adds -- the two SSIM kernels
// are vector-issue-bound on exactly these loops.  (Summation order differs from a left-to-right conv2d by rounding only:
// tests/test_gpu_losses.py holds the loss to 2e-6 and the gradient to 2e-4 of its maximum against the reference's.)
__device__ __forceinline__ float win11(const float* v)
{
    float s = GW[5] * v[5];
#pragma unroll
    for (int k = 0; k < 5; k++) s = __builtin_fmaf(GW[k], v[k] + v[10 - k], s);
    return s;
}

// Pass 1 on packed f32 (v_pk_fma_f32 / v_pk_add_f32: two results per instruction): the four windowed maps travel as two PAIRS,
// (x, y) and (x^2 + y^2, x y) -- the same fused operations in the same order per component as the scalar form (the loss value
// comes out bit-identical; the D maps differ in last bits where the compiler contracts the closed-form partials differently);
// the input tile is parked as (x, y) pairs, the row sums as pairs.  The kernel is vector-issue-bound (SQ_INSTS_VALU 34.7 M per
// launch = 56 us of its 63), four fifths of it these windows: 63 -> 55 us.  (The gradient kernel was tried the same way --
// (D1, D2) as a pair -- and did not move: with the XCD-aware tile map below it is bound by its reads, 63 -> 40 us.)
#ifndef GSR_SSIM_PK
#define GSR_SSIM_PK 1
#endif
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 win11_2(const f2* v)
{
    f2 s = GW[5] * v[5];
#pragma unroll
    for (int k = 0; k < 5; k++) s = __builtin_elementwise_fma((f2)(GW[k]), v[k] + v[10 - k], s);
    return s;
}

struct View { const float* p; long long sc, sy, sx; };     // element strides of a [C,H,W]-indexed image
struct ViewW { float* p; long long sc, sy, sx; };
// Which (channel, tile) a workgroup of the SSIM kernels takes.  Workgroups go round-robin over the 8 XCDs in dispatch order
// (x fastest, then y, then z) and every XCD has its own L2: with the plain map the four neighbours of a tile -- whose 5-pixel
// halos it re-reads, 2.1x the image in all -- sit on other L2s.  Instead XCD k owns a CONTIGUOUS run of the C * tiles ids
// (row-major tiles, channel-major): neighbours in a row share an L2 and run close in time.
__device__ __forceinline__ void xcd_tile(int C_rgb, int& tx, int& ty, int& ch)
{
    const int gxn = (int)gridDim.x, n = gxn * (int)gridDim.y, N = C_rgb * n;
    const int L = ((int)blockIdx.z * (int)gridDim.y + (int)blockIdx.y) * gxn + (int)blockIdx.x;
    const int xcd = L & 7, j = L >> 3, q = N >> 3, r = N & 7;
    const int id = xcd * q + min(xcd, r) + j;
    ch = id / n;
    const int t = id - ch * n;
    ty = t / gxn; tx = t - ty * gxn;
}
// FAST instantiations of the two SSIM kernels: every image (and the D maps) spans less than 4 GB with non-negative strides, so
// an element is a uniform 64-bit base (the channel's plane) + a 32-bit byte offset in a vector register -- one add per element
// instead of a 64-bit multiply-add chain -- and halo / out-of-image elements read element 0 and are masked afterwards: loads
// without branches.  (Static count of the generic kernels: 964 / 531 vector instructions of which 426 / 218 are the windows'
// arithmetic; 64-bit address arithmetic and one exec-mask branch per conditional load were most of the rest.)
template <typename T> __device__ __forceinline__ T ld32(const T* base, uint32_t byte_off)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
// offset of a masked element: 0 when the element lies outside.  (Opaque to the optimiser on purpose: it turns
// load(in ? off : 0) into in ? load(off) : load(0), i.e. back into one exec-mask branch per load.)
__device__ __forceinline__ uint32_t masked_off(bool in, uint32_t byte_off)
{
    uint32_t o = in ? byte_off : 0u;
    asm volatile("" : "+v"(o));
    return o;
}
template <typename T> __device__ __forceinline__ void st32(T* base, uint32_t byte_off, T v)
{
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
static bool view_fits_32(const float* p, const long long* s, int C, int H, int W)
{
    if (s[0] < 0 || s[1] < 0 || s[2] < 0) return false;
    const long long last = (long long)(C - 1) * s[0] + (long long)(H - 1) * s[1] + (long long)(W - 1) * s[2];
    return p != nullptr && last < (1ll << 29);
}

__device__ __forceinline__ float block_sum_256(float v, float* red)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// The masked depth loss of the same iteration (refine.py:634-660; kernels further down) can ride along in the two SSIM
// launches as ONE MORE z-plane of workgroups: as launches of their own its two passes are ~7 us each of mostly launch and
// drain on the stream.  n_wg == 0: no depth plane.
struct DepthPlane {
    int H, W;
    const float* pred; long long psy, psx;
    const float* gt; long long gsy, gsx;
    float max_depth;
    float* partials; int n_wg;                                   // value pass: [n_wg][4] = {sum_fg, n_fg, sum_bg, n_bg}
    float depth_factor, mask_factor; const float* stats; const float* scale; float* grad; long long qsy, qsx;   // gradient pass
};

__device__ __forceinline__ void depth_stats_rows(const DepthPlane& z, int w, int n_wg, float* red)
{
    float sf = 0.f, nf = 0.f, sb = 0.f, nb = 0.f;
    // a workgroup walks whole rows (no 64-bit division per element), four elements of a thread in flight at a time
    for (int yy = w; yy < z.H; yy += n_wg) {
        const float* pr = z.pred + yy * z.psy;
        const float* gr = z.gt + yy * z.gsy;
        for (int x0 = (int)threadIdx.x; x0 < z.W; x0 += 4 * 256) {
            float p[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int xx = x0 + 256 * u;
                const bool in = xx < z.W;
                p[u] = in ? pr[xx * z.psx] : 0.f;
                g[u] = in ? gr[xx * z.gsx] : z.max_depth;   // == max_depth: neither foreground nor background
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (g[u] < z.max_depth) { sf += fabsf(p[u] - g[u]); nf += 1.f; }
                else if (g[u] > z.max_depth) { sb += fabsf(p[u] - z.max_depth); nb += 1.f; }
            }
        }
    }
    const float a = block_sum_256(sf, red), b = block_sum_256(nf, red), c = block_sum_256(sb, red), d = block_sum_256(nb, red);
    if (threadIdx.x == 0) {
        z.partials[4 * w] = a; z.partials[4 * w + 1] = b; z.partials[4 * w + 2] = c; z.partials[4 * w + 3] = d;
    }
}

__device__ __forceinline__ float depth_grad_value(float p, float g, float max_depth, float cf, float cb)
{
    float v = 0.f;
    if (g < max_depth) { const float d = p - g; v = cf * (d > 0.f ? 1.f : d < 0.f ? -1.f : 0.f); }
    else if (g > max_depth) { const float d = p - max_depth; v = cb * (d > 0.f ? 1.f : d < 0.f ? -1.f : 0.f); }
    return v;
}

// ---------------------------------------------------------------- pass 1
template <bool FAST>
__global__ void __launch_bounds__(256)
ssim_stats_kernel(int H, int W, View x, View y, float* __restrict__ D, float* __restrict__ partials, int C_rgb, DepthPlane dz)
{
#if GSR_SSIM_PK
    __shared__ f2 XY[LIY * LP];          // (x, y) per element of the input tile
    __shared__ f2 Hq2[2][LIY * LT];      // row sums: [0] = (x, y), [1] = (x^2 + y^2, x y)
#else
    __shared__ float X[LIY * LP], Y[LIY * LP];
    // (SSIM reads E[x^2] and E[y^2] only as their SUM -- sigma_x^2 + sigma_y^2 -- so x^2 + y^2 goes through the window as
    // ONE quantity: four windowed maps instead of the reference's five, loss_utils.py:47-52)
    __shared__ float Hq[4][LIY * LT];
#endif
    __shared__ float red[4];
    if ((int)blockIdx.z == C_rgb) {   // (uniform) the depth plane: the first dz.n_wg workgroups of it each take rows w, w + n_wg, ..
        const int w = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        if (w < dz.n_wg) depth_stats_rows(dz, w, dz.n_wg, red);
        return;
    }
    const int tid = threadIdx.x;
    int tbx, tby, ch;
    xcd_tile(C_rgb, tbx, tby, ch);
    const int tx0 = tbx * LT, ty0 = tby * LTY;
    const float* xp = x.p + ch * x.sc;
    const float* yp = y.p + ch * y.sc;
    // all of a thread's tile loads are issued before the first one is consumed (a rolled loop paid one full memory
    // latency per iteration: 5.8 of the workgroup's 9.6 us)
    constexpr int NL = (LIY * LI + 255) / 256;
    float xv[NL], yv[NL];
    const auto park = [&](int i, float xe, float ye) {
#if GSR_SSIM_PK
        f2 e; e.x = xe; e.y = ye;
        XY[i] = e;
#else
        X[i] = xe; Y[i] = ye;
#endif
    };
    // element i = tid + 256 k of the LIY x LI input tile: (row, column) advance by (256 / LI, 256 % LI) per k with at most
    // one carry -- one division per thread instead of one per element, and one 64-bit address per thread plus deltas
    constexpr int DR = 256 / LI, DC = 256 % LI;
    static_assert((NL - 1) * DC + LI - 1 < 2 * LI, "at most one column carry");
    const int lr0 = tid / LI, lc0 = tid - lr0 * LI;
    const long long xb = (long long)(ty0 + lr0 - LR) * x.sy + (long long)(tx0 + lc0 - LR) * x.sx;
    const long long yb = (long long)(ty0 + lr0 - LR) * y.sy + (long long)(tx0 + lc0 - LR) * y.sx;
    if constexpr (FAST) {
        const int xsy = (int)x.sy, xsx = (int)x.sx, ysy = (int)y.sy, ysx = (int)y.sx;
        const int x0 = (ty0 + lr0 - LR) * xsy + (tx0 + lc0 - LR) * xsx, y0 = (ty0 + lr0 - LR) * ysy + (tx0 + lc0 - LR) * ysx;
        const int xA = DR * xsy + DC * xsx, xB = xsy - LI * xsx, yA = DR * ysy + DC * ysx, yB = ysy - LI * ysx;   // (uniform)
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int wrap = lc0 + k * DC >= LI ? 1 : 0;
            const int r = lr0 + k * DR + wrap, c = lc0 + k * DC - wrap * LI;
            const int gy = ty0 + r - LR, gx = tx0 + c - LR;
            const bool in = ((k + 1) * 256 <= LIY * LI || tid + k * 256 < LIY * LI) && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int xo = x0 + k * xA + (wrap ? xB : 0), yo = y0 + k * yA + (wrap ? yB : 0);
            xv[k] = ld32(xp, masked_off(in, (uint32_t)xo * 4u));
            yv[k] = ld32(yp, masked_off(in, (uint32_t)yo * 4u));
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int wrap = lc0 + k * DC >= LI ? 1 : 0;
            const int r = lr0 + k * DR + wrap, c = lc0 + k * DC - wrap * LI;
            const int gy = ty0 + r - LR, gx = tx0 + c - LR;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
            if ((k + 1) * 256 <= LIY * LI || tid + k * 256 < LIY * LI) {
                park(r * LP + c, in ? xv[k] : 0.f, in ? yv[k] : 0.f);
            }
        }
    } else {
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const int wrap = lc0 + k * DC >= LI ? 1 : 0;
        const int r = lr0 + k * DR + wrap, c = lc0 + k * DC - wrap * LI;
        const int gy = ty0 + r - LR, gx = tx0 + c - LR;
        const bool in = ((k + 1) * 256 <= LIY * LI || tid + k * 256 < LIY * LI) && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const int dr = k * DR + wrap, dc = k * DC - wrap * LI;
        xv[k] = in ? xp[xb + dr * x.sy + dc * x.sx] : 0.f;
        yv[k] = in ? yp[yb + dr * y.sy + dc * y.sx] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NL; k++) {
        if ((k + 1) * 256 <= LIY * LI || tid + k * 256 < LIY * LI) {
            const int wrap = lc0 + k * DC >= LI ? 1 : 0;
            const int r = lr0 + k * DR + wrap, c = lc0 + k * DC - wrap * LI;
            park(r * LP + c, xv[k], yv[k]);
        }
    }
    }
    __syncthreads();
    // horizontal: LIY rows x 32 columns, five windowed quantities.  A thread takes HS consecutive columns of a row: their
    // HS + 10 inputs slide through registers, so an input is read from LDS once per task instead of once per tap (22 LDS
    // reads per output before, 7 now) and its three products are formed once instead of eleven times.
#if GSR_SSIM_PK
    for (int t = tid; t < LIY * (LT / HS); t += 256) {
        const int r = t / (LT / HS), c0 = (t - r * (LT / HS)) * HS;
        f2 e[HS + 10], q[HS + 10];
#pragma unroll
        for (int k = 0; k < HS + 10; k++) {
            e[k] = XY[r * LP + c0 + k];
            const f2 m = e[k].yx * e[k].yy;                        // (y y, x y)
            q[k].x = __builtin_fmaf(e[k].x, e[k].x, m.x);          // x x + y y, as the scalar form rounds it
            q[k].y = m.y;
        }
#pragma unroll
        for (int o = 0; o < HS; o++) {
            const int i = r * LT + c0 + o;
            Hq2[0][i] = win11_2(e + o); Hq2[1][i] = win11_2(q + o);
        }
    }
    __syncthreads();
    // vertical: thread = (column c, group of RPT rows); RPT + 10 rows of each pair slide through registers
    const int c = tid & 31, r0 = (tid >> 5) * RPT;
    float out[4][RPT];
#pragma unroll
    for (int p_ = 0; p_ < 2; p_++) {
        f2 col[RPT + 10];
#pragma unroll
        for (int k = 0; k < RPT + 10; k++) col[k] = Hq2[p_][(r0 + k) * LT + c];
#pragma unroll
        for (int o = 0; o < RPT; o++) {
            const f2 w = win11_2(col + o);
            out[2 * p_][o] = w.x; out[2 * p_ + 1][o] = w.y;
        }
    }
#else
    for (int t = tid; t < LIY * (LT / HS); t += 256) {
        const int r = t / (LT / HS), c0 = (t - r * (LT / HS)) * HS;
        float a[HS + 10], b[HS + 10], sq[HS + 10], ab[HS + 10];
#pragma unroll
        for (int k = 0; k < HS + 10; k++) {
            a[k] = X[r * LP + c0 + k]; b[k] = Y[r * LP + c0 + k];
            sq[k] = __builtin_fmaf(a[k], a[k], b[k] * b[k]); ab[k] = a[k] * b[k];
        }
#pragma unroll
        for (int o = 0; o < HS; o++) {
            const int i = r * LT + c0 + o;
            Hq[0][i] = win11(a + o); Hq[1][i] = win11(b + o); Hq[2][i] = win11(sq + o); Hq[3][i] = win11(ab + o);
        }
    }
    __syncthreads();
    // vertical: thread = (column c, group of 4 rows); 14 rows of each quantity slide through registers
    const int c = tid & 31, r0 = (tid >> 5) * RPT;
    float out[4][RPT];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        float col[RPT + 10];
#pragma unroll
        for (int k = 0; k < RPT + 10; k++) col[k] = Hq[q][(r0 + k) * LT + c];
#pragma unroll
        for (int o = 0; o < RPT; o++) out[q][o] = win11(col + o);
    }
#endif
    float l1 = 0.f, ssum = 0.f;
    const size_t plane = (size_t)H * W;
#pragma unroll
    for (int o = 0; o < RPT; o++) {
        const int gy = ty0 + r0 + o, gx = tx0 + c;
        if (gy < H && gx < W) {
            const float mu1 = out[0][o], mu2 = out[1][o];
            const float musq = mu1 * mu1 + mu2 * mu2;
            const float s1s2 = out[2][o] - musq, s12 = out[3][o] - mu1 * mu2;   // sigma_x^2 + sigma_y^2, sigma_xy
            // loss_utils.py:57
            const float a1 = 2.f * mu1 * mu2 + SSIM_C1, a2 = 2.f * s12 + SSIM_C2;
            const float b1 = musq + SSIM_C1, b2 = s1s2 + SSIM_C2;
            const float inv = 1.f / (b1 * b2);
            const float S = a1 * a2 * inv;
            // partials w.r.t. mu_x, E[x^2], E[xy] with s1 = E[x^2] - mu_x^2, s12 = E[xy] - mu_x mu_y
            const float d1 = 2.f * mu2 * (a2 - a1) * inv - 2.f * mu1 * S * (b2 - b1) * inv;
            const float d2 = -S / b2;
            const float d3 = 2.f * a1 * inv;
            if constexpr (FAST) {   // (three uniform plane bases, one 32-bit offset)
                const uint32_t ob = (uint32_t)(gy * W + gx) * 4u;
                st32(D + (size_t)ch * plane, ob, d1);
                st32(D + ((size_t)C_rgb + ch) * plane, ob, d2);
                st32(D + (2 * (size_t)C_rgb + ch) * plane, ob, d3);
            } else {
            const size_t o_ = (size_t)ch * plane + (size_t)gy * W + gx;
            D[o_] = d1;
            D[(size_t)C_rgb * plane + o_] = d2;
            D[2 * (size_t)C_rgb * plane + o_] = d3;
            }
            ssum += S;
#if GSR_SSIM_PK
            const f2 ctr = XY[(r0 + o + LR) * LP + c + LR];
            l1 += fabsf(ctr.x - ctr.y);
#else
            l1 += fabsf(X[(r0 + o + LR) * LP + c + LR] - Y[(r0 + o + LR) * LP + c + LR]);
#endif
        }
    }
    const float t_l1 = block_sum_256(l1, red);
    const float t_s = block_sum_256(ssum, red);
    if (tid == 0) {
        const size_t wg = ((size_t)ch * gridDim.y + tby) * gridDim.x + tbx;   // (the tile's slot: the summation order is the map's, not the dispatch's)
        partials[2 * wg] = t_l1;
        partials[2 * wg + 1] = t_s;
    }
}

// out = {loss, l1 mean, ssim mean}.  One workgroup of 1 024 threads: each sums its strided share of the per-workgroup
// partials with four independent 8-byte loads in flight (a 256-thread loop with one dependent load per trip took 13 us
// for 34 k partials), then a double-precision tree.
__global__ void __launch_bounds__(1024)
l1_ssim_finalize_kernel(int n_wg, const float* __restrict__ partials, double inv_n, float f, float* __restrict__ out)
{
    __shared__ double r1[1024], r2[1024];
    const float2* p2 = reinterpret_cast<const float2*>(partials);
    double a = 0.0, b = 0.0;
    int i = threadIdx.x;
    for (; i + 3 * 1024 < n_wg; i += 4 * 1024) {
        const float2 v0 = p2[i], v1 = p2[i + 1024], v2 = p2[i + 2048], v3 = p2[i + 3072];
        a += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
        b += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    }
    for (; i < n_wg; i += 1024) { const float2 v = p2[i]; a += (double)v.x; b += (double)v.y; }
    r1[threadIdx.x] = a; r2[threadIdx.x] = b;
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) { r1[threadIdx.x] += r1[threadIdx.x + d]; r2[threadIdx.x] += r2[threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double l1 = r1[0] * inv_n, s = r2[0] * inv_n;
        out[0] = (float)((1.0 - (double)f) * l1 + (double)f * (1.0 - s));   // refine.py:453
        out[1] = (float)l1;
        out[2] = (float)s;
    }
}

// ---------------------------------------------------------------- pass 2
template <bool FAST>
__global__ void __launch_bounds__(256)
ssim_grad_kernel(int H, int W, View x, View y, const float* __restrict__ D, float ca, float cb, const float* __restrict__ scale,
                 ViewW g, int C_rgb, DepthPlane dz)
{
    if ((int)blockIdx.z == C_rgb) {   // (uniform) the depth plane: every workgroup of it takes rows w, w + (workgroups of a plane), ..
        const int nwg = (int)(gridDim.x * gridDim.y), w = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        const float sc_ = dz.scale ? *dz.scale : 1.f;
        const float cf = dz.depth_factor == 0.f ? 0.f : sc_ * dz.depth_factor / dz.stats[2];   // a disabled term has no gradient
        const float cbk = dz.mask_factor == 0.f ? 0.f : sc_ * dz.mask_factor / dz.stats[3];
        for (int yy = w; yy < dz.H; yy += nwg)
            for (int xx = (int)threadIdx.x; xx < dz.W; xx += 256)
                dz.grad[yy * dz.qsy + xx * dz.qsx] = depth_grad_value(dz.pred[yy * dz.psy + xx * dz.psx], dz.gt[yy * dz.gsy + xx * dz.gsx],
                                                                       dz.max_depth, cf, cbk);
        return;
    }
    // (scale: the incoming d(total)/d(loss) as a DEVICE scalar -- autograd's grad_output -- so that the gradient leaves this
    // kernel final instead of being multiplied once more by an elementwise kernel over the whole image; NULL = 1)
    if (scale) { const float sc_ = *scale; ca *= sc_; cb *= sc_; }
    __shared__ float T[3][LIY * LP];
    __shared__ float Hq[3][LIY * LT];
    const int tid = threadIdx.x;
    int tbx, tby, ch;
    xcd_tile(C_rgb, tbx, tby, ch);
    const int tx0 = tbx * LT, ty0 = tby * LTY;
    const size_t plane = (size_t)H * W, vol = (size_t)C_rgb * plane;
    constexpr int NL = (LIY * LI + 255) / 256;
    float dv[NL][3];
    constexpr int DR = 256 / LI, DC = 256 % LI;   // (row, column) step per 256 elements, see ssim_stats_kernel
    const int lr0 = tid / LI, lc0 = tid - lr0 * LI;
    const float* Db = D + (size_t)ch * plane;
#pragma unroll
    for (int k = 0; k < NL; k++) {   // every load in flight before the first LDS store (see ssim_stats_kernel)
        const int wrap = lc0 + k * DC >= LI ? 1 : 0;
        const int r = lr0 + k * DR + wrap, c = lc0 + k * DC - wrap * LI;
        const int gy = ty0 + r - LR, gx = tx0 + c - LR;
        const bool in = ((k + 1) * 256 <= LIY * LI || tid + k * 256 < LIY * LI) && gy >= 0 && gy < H && gx >= 0 && gx < W;
        if constexpr (FAST) {   // unconditional loads from an always valid element, masked when they are parked
            const uint32_t ob = masked_off(in, (uint32_t)(gy * W + gx) * 4u);
            dv[k][0] = ld32(Db, ob); dv[k][1] = ld32(Db + vol, ob); dv[k][2] = ld32(Db + 2 * vol, ob);
        } else {
        const size_t o_ = (size_t)((in ? gy : 0) * W + (in ? gx : 0));   // H * W < 2^31 for any image this kernel sees
        dv[k][0] = in ? Db[o_] : 0.f;
        dv[k][1] = in ? Db[vol + o_] : 0.f;
        dv[k][2] = in ? Db[2 * vol + o_] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < NL; k++) {
        if ((k + 1) * 256 <= LIY * LI || tid + k * 256 < LIY * LI) {
            const int wrap = lc0 + k * DC >= LI ? 1 : 0;
            const int r = lr0 + k * DR + wrap, c = lc0 + k * DC - wrap * LI;
            const int gy = ty0 + r - LR, gx = tx0 + c - LR;
            const bool in = !FAST || (gy >= 0 && gy < H && gx >= 0 && gx < W);
            T[0][r * LP + c] = in ? dv[k][0] : 0.f; T[1][r * LP + c] = in ? dv[k][1] : 0.f; T[2][r * LP + c] = in ? dv[k][2] : 0.f;
        }
    }
    __syncthreads();
    // horizontal pass with the inputs sliding through registers, as in ssim_stats_kernel (33 LDS reads per output -> 10.5)
    for (int t = tid; t < LIY * (LT / HS); t += 256) {
        const int r = t / (LT / HS), c0 = (t - r * (LT / HS)) * HS;
        float d0[HS + 10], d1[HS + 10], d2[HS + 10];
#pragma unroll
        for (int k = 0; k < HS + 10; k++) { d0[k] = T[0][r * LP + c0 + k]; d1[k] = T[1][r * LP + c0 + k]; d2[k] = T[2][r * LP + c0 + k]; }
#pragma unroll
        for (int o = 0; o < HS; o++) {
            const int i = r * LT + c0 + o;
            Hq[0][i] = win11(d0 + o); Hq[1][i] = win11(d1 + o); Hq[2][i] = win11(d2 + o);
        }
    }
    __syncthreads();
    const int c = tid & 31, r0 = (tid >> 5) * RPT;
    float out[3][RPT];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        float col[RPT + 10];
#pragma unroll
        for (int k = 0; k < RPT + 10; k++) col[k] = Hq[q][(r0 + k) * LT + c];
#pragma unroll
        for (int o = 0; o < RPT; o++) out[q][o] = win11(col + o);
    }
#pragma unroll
    for (int o = 0; o < RPT; o++) {
        const int gy = ty0 + r0 + o, gx = tx0 + c;
        if (gy < H && gx < W) {
            float xv, yv;
            if constexpr (FAST) {
                xv = ld32(x.p + ch * x.sc, (uint32_t)(gy * (int)x.sy + gx * (int)x.sx) * 4u);
                yv = ld32(y.p + ch * y.sc, (uint32_t)(gy * (int)y.sy + gx * (int)y.sx) * 4u);
            } else {
                xv = x.p[ch * x.sc + gy * x.sy + gx * x.sx]; yv = y.p[ch * y.sc + gy * y.sy + gx * y.sx];
            }
            const float d = xv - yv;
            const float sgn = d > 0.f ? 1.f : d < 0.f ? -1.f : 0.f;   // d|x|/dx at 0 is 0 in torch
            const float ds = out[0][o] + 2.f * xv * out[1][o] + yv * out[2][o];
            if constexpr (FAST) st32(g.p + ch * g.sc, (uint32_t)(gy * (int)g.sy + gx * (int)g.sx) * 4u, ca * sgn - cb * ds);
            else g.p[ch * g.sc + gy * g.sy + gx * g.sx] = ca * sgn - cb * ds;
        }
    }
}

size_t l1_ssim_workspace_bytes(int C, int H, int W)
{
    const size_t n = (size_t)C * H * W;
    const size_t n_wg = (size_t)C * ((H + LTY - 1) / LTY) * ((W + LT - 1) / LT);
    return align_up(3 * n * sizeof(float)) + align_up(2 * n_wg * sizeof(float)) + 256;
}

// pass 1 alone (-> D and the per-workgroup partial sums in `workspace`); returns the number of partial pairs
constexpr int DEPTH_WGS = 1024;
static int depth_plane_wgs(int C, int H, int W, int Hd)   // workgroups of the depth plane's value pass (they walk whole rows)
{
    (void)C;
    const int plane = ((W + LT - 1) / LT) * ((H + LTY - 1) / LTY);
    const int n = Hd < DEPTH_WGS ? Hd : DEPTH_WGS;
    return n < plane ? n : plane;
}

static int launch_ssim_stats(int C, int H, int W, const float* pred, const long long* ps, const float* gt, const long long* gs_,
                             void* workspace, hipStream_t st, const DepthPlane* depth = nullptr)
{
    const size_t n = (size_t)C * H * W;
    float* D = static_cast<float*>(workspace);
    float* partials = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up(3 * n * sizeof(float)));
    const dim3 grid((W + LT - 1) / LT, (H + LTY - 1) / LTY, C + (depth ? 1 : 0));
    const View x{pred, ps[0], ps[1], ps[2]}, y{gt, gs_[0], gs_[1], gs_[2]};
    DepthPlane dz{};
    if (depth) dz = *depth;
    const bool fast = view_fits_32(pred, ps, C, H, W) && view_fits_32(gt, gs_, C, H, W) && 3 * n < (1ull << 29);
    if (fast) ssim_stats_kernel<true><<<grid, 256, 0, st>>>(H, W, x, y, D, partials, C, dz);
    else ssim_stats_kernel<false><<<grid, 256, 0, st>>>(H, W, x, y, D, partials, C, dz);
    return (int)(grid.x * grid.y * C);
}

// pass 2 alone, from the D maps pass 1 left in `workspace`
void launch_l1_ssim_grad(int C, int H, int W, const float* pred, const long long* ps, const float* gt, const long long* gs_,
                         float f, const void* workspace, const float* scale, float* grad, const long long* gstr, hipStream_t st)
{
    const size_t n = (size_t)C * H * W;
    const dim3 grid((W + LT - 1) / LT, (H + LTY - 1) / LTY, C);
    const View x{pred, ps[0], ps[1], ps[2]}, y{gt, gs_[0], gs_[1], gs_[2]};
    const ViewW g{grad, gstr[0], gstr[1], gstr[2]};
    const bool fast = view_fits_32(pred, ps, C, H, W) && view_fits_32(gt, gs_, C, H, W) && view_fits_32(grad, gstr, C, H, W) && 3 * n < (1ull << 29);
    if (fast) ssim_grad_kernel<true><<<grid, 256, 0, st>>>(H, W, x, y, static_cast<const float*>(workspace), (1.f - f) / (float)n, f / (float)n, scale, g, C, DepthPlane{});
    else ssim_grad_kernel<false><<<grid, 256, 0, st>>>(H, W, x, y, static_cast<const float*>(workspace), (1.f - f) / (float)n, f / (float)n, scale, g, C, DepthPlane{});
}

void launch_l1_ssim(int C, int H, int W, const float* pred, const long long* ps, const float* gt, const long long* gs_,
                    float f, void* workspace, float* loss_out, float* grad, const long long* gstr, hipStream_t st)
{
    const size_t n = (size_t)C * H * W;
    const float* partials = reinterpret_cast<const float*>(static_cast<char*>(workspace) + align_up(3 * n * sizeof(float)));
    const int n_wg = launch_ssim_stats(C, H, W, pred, ps, gt, gs_, workspace, st);
    l1_ssim_finalize_kernel<<<1, 1024, 0, st>>>(n_wg, partials, 1.0 / (double)n, f, loss_out);
    if (grad) launch_l1_ssim_grad(C, H, W, pred, ps, gt, gs_, f, workspace, nullptr, grad, gstr, st);
}

// ---------------------------------------------------------------- masked depth / silhouette L1 (refine.py:634-660)
//   depth term:  depth_factor * mean over {gt < max_depth} of |pred - gt|
//   mask term:   mask_factor  * mean over {gt > max_depth} of |pred - max_depth|
// acc = {sum_fg, n_fg, sum_bg, n_bg} per workgroup; out = {depth loss, mask loss, n_fg, n_bg}.
__global__ void __launch_bounds__(256)
depth_stats_kernel(int H, int W, const float* __restrict__ pred, long long psy, long long psx,
                   const float* __restrict__ gt, long long gsy, long long gsx, float max_depth,
                   float* __restrict__ partials)
{
    __shared__ float red[4];
    DepthPlane z{};
    z.H = H; z.W = W; z.pred = pred; z.psy = psy; z.psx = psx; z.gt = gt; z.gsy = gsy; z.gsx = gsx; z.max_depth = max_depth;
    z.partials = partials;
    depth_stats_rows(z, (int)blockIdx.x, (int)gridDim.x, red);
}

__global__ void __launch_bounds__(256)
depth_finalize_kernel(int n_wg, const float* __restrict__ partials, float depth_factor, float mask_factor,
                      float* __restrict__ out)
{
    __shared__ double r[4][256];   // fixed-order tree in double: deterministic
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < n_wg; i += 256) {
        const float4 q = reinterpret_cast<const float4*>(partials)[i];
        v[0] += (double)q.x; v[1] += (double)q.y; v[2] += (double)q.z; v[3] += (double)q.w;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) r[k][threadIdx.x] = v[k];
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) {
#pragma unroll
            for (int k = 0; k < 4; k++) r[k][threadIdx.x] += r[k][threadIdx.x + d];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // A factor of exactly 0 means "term absent": the reference only adds the depth term after depth_loss_from and the
        // mask term after mask_loss_from (refine.py:634-660), so a view without foreground (or background) pixels must
        // not turn a disabled term into 0 * (0/0) = nan.  An ENABLED term over an empty selection is nan, like torch's
        // mean of an empty selection.
        out[0] = depth_factor == 0.f ? 0.f : (float)((double)depth_factor * (r[0][0] / r[1][0]));
        out[1] = mask_factor == 0.f ? 0.f : (float)((double)mask_factor * (r[2][0] / r[3][0]));
        out[2] = (float)r[1][0];
        out[3] = (float)r[3][0];
    }
}

__global__ void __launch_bounds__(256)
depth_grad_kernel(int H, int W, const float* __restrict__ pred, long long psy, long long psx,
                  const float* __restrict__ gt, long long gsy, long long gsx, float max_depth, float depth_factor,
                  float mask_factor, const float* __restrict__ stats, const float* __restrict__ scale,
                  float* __restrict__ grad, long long qsy, long long qsx)
{
    const int yy = (int)blockIdx.y, xx = (int)(blockIdx.x * 256 + threadIdx.x);   // grid = (ceil(W / 256), H): no division
    if (xx >= W) return;
    const float p = pred[yy * psy + xx * psx], g = gt[yy * gsy + xx * gsx];
    const float sc_ = scale ? *scale : 1.f;                                  // (device scalar: see ssim_grad_kernel)
    const float cf = depth_factor == 0.f ? 0.f : sc_ * depth_factor / stats[2];   // a disabled term has no gradient (not 0/0)
    const float cb = mask_factor == 0.f ? 0.f : sc_ * mask_factor / stats[3];
    grad[yy * qsy + xx * qsx] = depth_grad_value(p, g, max_depth, cf, cb);
}

size_t depth_l1_workspace_bytes() { return align_up(4 * DEPTH_WGS * sizeof(float)) + 256; }

void launch_depth_l1_grad(int H, int W, const float* pred, const long long* ps, const float* gt, const long long* gs_,
                          float max_depth, float depth_factor, float mask_factor, const float* stats, const float* scale,
                          float* grad, const long long* gstr, hipStream_t st)
{
    depth_grad_kernel<<<dim3((unsigned)((W + 255) / 256), (unsigned)H), 256, 0, st>>>(H, W, pred, ps[0], ps[1], gt, gs_[0], gs_[1],
                                                                  max_depth, depth_factor, mask_factor, stats, scale,
                                                                  grad, gstr[0], gstr[1]);
}

void launch_depth_l1(int H, int W, const float* pred, const long long* ps, const float* gt, const long long* gs_,
                     float max_depth, float depth_factor, float mask_factor, void* workspace, float* loss_out,
                     float* grad, const long long* gstr, hipStream_t st)
{
    const int n_wg = H < DEPTH_WGS ? H : DEPTH_WGS;   // workgroups walk whole rows
    float* partials = static_cast<float*>(workspace);
    depth_stats_kernel<<<n_wg, 256, 0, st>>>(H, W, pred, ps[0], ps[1], gt, gs_[0], gs_[1], max_depth, partials);
    depth_finalize_kernel<<<1, 256, 0, st>>>(n_wg, partials, depth_factor, mask_factor, loss_out);
    if (grad) launch_depth_l1_grad(H, W, pred, ps, gt, gs_, max_depth, depth_factor, mask_factor, loss_out, nullptr, grad, gstr, st);
}

// ---------------------------------------------------------------- both losses of a refinement iteration, values only
// One finalize for the two reductions: out = {l1 + dssim loss, l1 mean, ssim mean, depth term, mask term, #fg, #bg, TOTAL}.
// (As separate ops the iteration paid two single-workgroup finalize kernels and two elementwise additions for the total:
// four dependent launches of ~5 us each on the stream.)  Same arithmetic and order as the two finalize kernels above.
__global__ void __launch_bounds__(1024)
rgb_depth_finalize_kernel(int n_wg, const float* __restrict__ partials, double inv_n, float f, int n_wg_d,
                          const float* __restrict__ partials_d, float depth_factor, float mask_factor, float* __restrict__ out,
                          float* __restrict__ out_keep)
{
    __shared__ double r[6][1024];
    const float2* p2 = reinterpret_cast<const float2*>(partials);
    double a = 0.0, b = 0.0;
    int i = threadIdx.x;
    for (; i + 3 * 1024 < n_wg; i += 4 * 1024) {
        const float2 v0 = p2[i], v1 = p2[i + 1024], v2 = p2[i + 2048], v3 = p2[i + 3072];
        a += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
        b += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    }
    for (; i < n_wg; i += 1024) { const float2 v = p2[i]; a += (double)v.x; b += (double)v.y; }
    // the depth partials in the summation order of depth_finalize_kernel: 256 strided accumulators (threads 0..255 here)
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (threadIdx.x < 256)
        for (int k = threadIdx.x; k < n_wg_d; k += 256) {
            const float4 q = reinterpret_cast<const float4*>(partials_d)[k];
            v[0] += (double)q.x; v[1] += (double)q.y; v[2] += (double)q.z; v[3] += (double)q.w;
        }
    r[0][threadIdx.x] = a; r[1][threadIdx.x] = b;
#pragma unroll
    for (int k = 0; k < 4; k++) r[2 + k][threadIdx.x] = v[k];
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) {
#pragma unroll
            for (int k = 0; k < 6; k++) r[k][threadIdx.x] += r[k][threadIdx.x + d];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double l1 = r[0][0] * inv_n, s = r[1][0] * inv_n;
        const float loss = (float)((1.0 - (double)f) * l1 + (double)f * (1.0 - s));   // refine.py:453
        const float dt = depth_factor == 0.f ? 0.f : (float)((double)depth_factor * (r[2][0] / r[3][0]));
        const float mt = mask_factor == 0.f ? 0.f : (float)((double)mask_factor * (r[4][0] / r[5][0]));
        out[0] = loss; out[1] = (float)l1; out[2] = (float)s;
        out[3] = dt; out[4] = mt; out[5] = (float)r[3][0]; out[6] = (float)r[5][0];
        out[7] = (loss + dt) + mt;   // (float additions in the order the separate ops formed the total)
        // the same eight numbers once more in the workspace's tail: what the gradient passes read, so that the caller can hand
        // `out` to its user outright (a 32-byte device copy in front of it was a 4.7 us kernel on the stream)
        if (out_keep)
            for (int k = 0; k < 8; k++) out_keep[k] = out[k];
    }
}

void launch_rgb_depth_loss(int C, int H, int W, const float* pred, const long long* ps, const float* gt, const long long* gs_, float f,
                           void* ws_ssim, int Hd, int Wd, const float* dpred, const long long* dps, const float* dgt,
                           const long long* dgs, float max_depth, float depth_factor, float mask_factor, void* ws_depth,
                           float* out8, hipStream_t st)
{
    const size_t n = (size_t)C * H * W;
    const float* partials = reinterpret_cast<const float*>(static_cast<char*>(ws_ssim) + align_up(3 * n * sizeof(float)));
    float* partials_d = static_cast<float*>(ws_depth);
    DepthPlane dz{};
    dz.H = Hd; dz.W = Wd; dz.pred = dpred; dz.psy = dps[0]; dz.psx = dps[1]; dz.gt = dgt; dz.gsy = dgs[0]; dz.gsx = dgs[1];
    dz.max_depth = max_depth; dz.partials = partials_d; dz.n_wg = depth_plane_wgs(C, H, W, Hd);
    const int n_wg = launch_ssim_stats(C, H, W, pred, ps, gt, gs_, ws_ssim, st, &dz);   // the depth sums ride along as a z-plane
    float* keep = reinterpret_cast<float*>(static_cast<char*>(ws_ssim) + l1_ssim_workspace_bytes(C, H, W) - 256);
    rgb_depth_finalize_kernel<<<1, 1024, 0, st>>>(n_wg, partials, 1.0 / (double)n, f, dz.n_wg, partials_d, depth_factor, mask_factor, out8, keep);
}

// both gradient passes in ONE launch (the depth image as one more z-plane of the SSIM gradient kernel)
void launch_rgb_depth_loss_grad(int C, int H, int W, const float* pred, const long long* ps, const float* gt, const long long* gs_, float f,
                                const void* ws_ssim, int Hd, int Wd, const float* dpred, const long long* dps, const float* dgt,
                                const long long* dgs, float max_depth, float depth_factor, float mask_factor, const float* stats,
                                const float* scale, float* grad, const long long* gstr, float* dgrad, const long long* dgstr,
                                hipStream_t st)
{
    const size_t n = (size_t)C * H * W;
    const dim3 grid((W + LT - 1) / LT, (H + LTY - 1) / LTY, C + 1);
    const View x{pred, ps[0], ps[1], ps[2]}, y{gt, gs_[0], gs_[1], gs_[2]};
    const ViewW g{grad, gstr[0], gstr[1], gstr[2]};
    DepthPlane dz{};
    dz.H = Hd; dz.W = Wd; dz.pred = dpred; dz.psy = dps[0]; dz.psx = dps[1]; dz.gt = dgt; dz.gsy = dgs[0]; dz.gsx = dgs[1];
    dz.max_depth = max_depth; dz.n_wg = 1; dz.depth_factor = depth_factor; dz.mask_factor = mask_factor; dz.stats = stats;
    dz.scale = scale; dz.grad = dgrad; dz.qsy = dgstr[0]; dz.qsx = dgstr[1];
    const bool fast = view_fits_32(pred, ps, C, H, W) && view_fits_32(gt, gs_, C, H, W) && view_fits_32(grad, gstr, C, H, W) && 3 * n < (1ull << 29);
    if (fast) ssim_grad_kernel<true><<<grid, 256, 0, st>>>(H, W, x, y, static_cast<const float*>(ws_ssim), (1.f - f) / (float)n, f / (float)n, scale, g, C, dz);
    else ssim_grad_kernel<false><<<grid, 256, 0, st>>>(H, W, x, y, static_cast<const float*>(ws_ssim), (1.f - f) / (float)n, f / (float)n, scale, g, C, dz);
}

}  // namespace gsr
