/*
 * gsr_oracle.c -- CPU restatement of the reference rasterizer's algorithm.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (gaustar_amd/) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / CPU baseline.
 *
 * Every function restates one stage of the reference CUDA rasterizer
 *   DGR = gaussian_splatting/submodules/diff-gaussian-rasterization
 * and cites the file:line it follows.  Matrix algebra that the reference
 * writes with glm (column-major, m[col][row]) is written out here in plain
 * row-major scalars; SURVEY.md section 9 lists the conventions.
 *
 * Parity pin: this restatement is checked (tests/test_oracle_golden.py)
 * against tests/golden/ *.npz, which hold outputs of the reference's own
 * kernels (oracle/_ref, built from the sources under /root/reference by
 * oracle/build_ref.sh and run on an MI355X by tests/golden/make_golden.py).
 *
 * Arithmetic: fp32 throughout, like the reference; built with
 * -ffp-contract=off so results do not depend on the host's FMA contraction.
 * The only deliberate deviation: the backward blend accumulates the
 * per-(pixel,Gaussian) float terms into double accumulators (the reference
 * uses float atomicAdd in a nondeterministic order, backward.cu:523-554);
 * the double sum is the order-free value every float order approximates.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK_X 16 /* DGR/cuda_rasterizer/config.h:16 */
#define BLOCK_Y 16 /* DGR/cuda_rasterizer/config.h:17 */

/* DGR/cuda_rasterizer/auxiliary.h:22-39 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* auxiliary.h:41-44 -- evaluated in double, narrowed to float. */
static inline float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * S - 1.0) * 0.5); }

/* auxiliary.h:46-56 -- C truncation toward zero, clamp to the tile grid. */
static void get_rect(float px, float py, int max_radius, int gx, int gy, int* rmin, int* rmax)
{
    rmin[0] = imin(gx, imax(0, (int)((px - max_radius) / BLOCK_X)));
    rmin[1] = imin(gy, imax(0, (int)((py - max_radius) / BLOCK_Y)));
    rmax[0] = imin(gx, imax(0, (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
    rmax[1] = imin(gy, imax(0, (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* auxiliary.h:58-77 -- matrices are column-major 4x4. */
static inline void xform4x3(const float* p, const float* m, float* o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void xform4x4(const float* p, const float* m, float* o)
{
    xform4x3(p, m, o);
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* auxiliary.h:139-164 (in_frustum): only the near-plane test is live. */
static inline int in_frustum(const float* p, const float* view, float* p_view)
{
    xform4x3(p, view, p_view);
    return !(p_view[2] <= 0.2f);
}

/* forward.cu:118-152 (computeCov3D).  R is the rotation of the RAW quaternion
 * (r,x,y,z); Sigma = R S^2 R^T; upper triangle stored. */
static void quat_to_R(const float* q, float R[3][3])
{
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}
static void compute_cov3d(const float* scale, float mod, const float* rot, float* cov3D)
{
    float R[3][3], M[3][3], s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    quat_to_R(rot, R);
    /* M = S * R^T  (M[i][j] = s_i R[j][i]);  Sigma = M^T M */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) M[i][j] = s[i] * R[j][i];
    float Sg[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Sg[i][j] = M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j];
    cov3D[0] = Sg[0][0]; cov3D[1] = Sg[0][1]; cov3D[2] = Sg[0][2];
    cov3D[3] = Sg[1][1]; cov3D[4] = Sg[1][2]; cov3D[5] = Sg[2][2];
}

/* Rows 0/1 of A = J_std * R_w2c, shared by forward.cu:74-113 and backward.cu:144-274.
 * t is clamped to +-1.3*tanfov*t.z before J is built (forward.cu:82-87). */
static void ewa_rows(const float* mean, float fx, float fy, float tanfovx, float tanfovy, const float* view,
                     float* a0, float* a1, float* t_out, float* txtz_out, float* tytz_out)
{
    float t[3];
    xform4x3(mean, view, t);
    const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    const float J00 = fx / t[2], J02 = -(fx * t[0]) / (t[2] * t[2]);
    const float J11 = fy / t[2], J12 = -(fy * t[1]) / (t[2] * t[2]);
    /* R_w2c[r][k] = view[4k + r] */
    for (int k = 0; k < 3; k++) {
        a0[k] = view[4 * k + 0] * J00 + view[4 * k + 2] * J02;
        a1[k] = view[4 * k + 1] * J11 + view[4 * k + 2] * J12;
    }
    t_out[0] = t[0]; t_out[1] = t[1]; t_out[2] = t[2];
    if (txtz_out) *txtz_out = txtz;
    if (tytz_out) *tytz_out = tytz;
}

/* forward.cu:74-113 (computeCov2D): cov = A Sigma A^T, +0.3 low-pass on the diagonal. */
static void compute_cov2d(const float* mean, float fx, float fy, float tanfovx, float tanfovy, const float* c,
                          const float* view, float* cov)
{
    float a0[3], a1[3], t[3];
    ewa_rows(mean, fx, fy, tanfovx, tanfovy, view, a0, a1, t, 0, 0);
    const float V[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
    float v0[3], v1[3];
    for (int k = 0; k < 3; k++) {
        v0[k] = V[k][0] * a0[0] + V[k][1] * a0[1] + V[k][2] * a0[2];
        v1[k] = V[k][0] * a1[0] + V[k][1] * a1[1] + V[k][2] * a1[2];
    }
    cov[0] = a0[0] * v0[0] + a0[1] * v0[1] + a0[2] * v0[2] + 0.3f;
    cov[1] = a0[0] * v1[0] + a0[1] * v1[1] + a0[2] * v1[2];
    cov[2] = a1[0] * v1[0] + a1[1] * v1[1] + a1[2] * v1[2] + 0.3f;
}

/* forward.cu:20-71 (computeColorFromSH): layout shs[P][M][3]. */
static void sh_to_rgb(int idx, int deg, int M, const float* means, const float* campos, const float* shs,
                      uint8_t* clamped, float* out)
{
    const float* pos = means + 3 * idx;
    float dir[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
    float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    dir[0] /= len; dir[1] /= len; dir[2] /= len;
    const float* sh = shs + (size_t)idx * M * 3;
    for (int ch = 0; ch < 3; ch++) {
#define S(k) sh[3 * (k) + ch]
        float r = SH_C0 * S(0);
        if (deg > 0) {
            float x = dir[0], y = dir[1], z = dir[2];
            r = r - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * S(4) + SH_C2[1] * yz * S(5) + SH_C2[2] * (2.0f * zz - xx - yy) * S(6) +
                    SH_C2[3] * xz * S(7) + SH_C2[4] * (xx - yy) * S(8);
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * S(9) + SH_C3[1] * xy * z * S(10) +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11) +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12) +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13) + SH_C3[5] * z * (xx - yy) * S(14) +
                        SH_C3[6] * x * (xx - 3.0f * yy) * S(15);
                }
            }
        }
#undef S
        r += 0.5f;
        clamped[3 * idx + ch] = (r < 0);
        out[ch] = fmaxf(r, 0.0f);
    }
}

/* forward.cu:155-256 (preprocessCUDA).  Returns sum(tiles_touched) = num_rendered
 * (rasterizer_impl.cu:277-281).  Arrays of culled Gaussians keep radii = tiles = 0. */
int gsr_oracle_preprocess(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                          const float* rotations, const float* opacities, const float* shs,
                          const float* cov3D_precomp, const float* colors_precomp, const float* view,
                          const float* proj, const float* campos, int W, int H, float tanfovx, float tanfovy,
                          int* radii, float* means2D, float* depths, float* cov3Ds, float* rgb,
                          float* conic_opacity, uint32_t* tiles_touched, uint8_t* clamped)
{
    const float focal_y = H / (2.0f * tanfovy), focal_x = W / (2.0f * tanfovx); /* rasterizer_impl.cu:222-223 */
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    long long total = 0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;
        const float* p = means3D + 3 * idx;
        float p_view[3];
        if (!in_frustum(p, view, p_view)) continue;
        float ph[4];
        xform4x4(p, proj, ph);
        float p_w = 1.0f / (ph[3] + 0.0000001f);
        float p_proj[2] = {ph[0] * p_w, ph[1] * p_w};
        const float* c3;
        if (cov3D_precomp) c3 = cov3D_precomp + 6 * idx;
        else {
            compute_cov3d(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov3Ds + 6 * idx);
            c3 = cov3Ds + 6 * idx;
        }
        float cov[3];
        compute_cov2d(p, focal_x, focal_y, tanfovx, tanfovy, c3, view, cov);
        float det = cov[0] * cov[2] - cov[1] * cov[1];
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
        float mid = 0.5f * (cov[0] + cov[2]);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float px = ndc2pix(p_proj[0], W), py = ndc2pix(p_proj[1], H);
        int rmin[2], rmax[2];
        get_rect(px, py, (int)my_radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (!colors_precomp) sh_to_rgb(idx, D, M, means3D, campos, shs, clamped, rgb + 3 * idx);
        depths[idx] = p_view[2];
        radii[idx] = (int)my_radius;
        means2D[2 * idx] = px;
        means2D[2 * idx + 1] = py;
        conic_opacity[4 * idx + 0] = conic[0];
        conic_opacity[4 * idx + 1] = conic[1];
        conic_opacity[4 * idx + 2] = conic[2];
        conic_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
        total += tiles_touched[idx];
    }
    return (int)total;
}

/* rasterizer_impl.cu:54-66 (checkFrustum) */
void gsr_oracle_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present)
{
    (void)proj;
    for (int i = 0; i < P; i++) {
        float pv[3];
        present[i] = (uint8_t)in_frustum(means3D + 3 * i, view, pv);
    }
}

typedef struct { uint64_t key; uint32_t val; uint32_t seq; } kv_t;
static int kv_cmp(const void* a, const void* b)
{
    const kv_t *x = (const kv_t*)a, *y = (const kv_t*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq);
}

/* rasterizer_impl.cu:70-111 (duplicateWithKeys), :300-308 (stable radix sort on the low
 * 32+msb bits -- tile ids never exceed those bits, so a full-key stable sort is identical),
 * :116-138 (identifyTileRanges).  ranges is [T][2], zero-filled first (:310). */
void gsr_oracle_bin(int P, int W, int H, const float* means2D, const float* depths, const int* radii, int R,
                    uint64_t* keys_sorted, uint32_t* point_list, uint32_t* ranges)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (size_t)(R > 0 ? R : 1));
    uint32_t off = 0;
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] <= 0) continue;
        int rmin[2], rmax[2];
        get_rect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
        uint32_t dbits;
        memcpy(&dbits, depths + idx, 4);
        for (int y = rmin[1]; y < rmax[1]; y++)
            for (int x = rmin[0]; x < rmax[0]; x++) {
                uint64_t key = (uint64_t)(y * gx + x);
                key <<= 32;
                key |= dbits;
                kv[off].key = key; kv[off].val = (uint32_t)idx; kv[off].seq = off;
                off++;
            }
    }
    qsort(kv, off, sizeof(kv_t), kv_cmp);
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
    for (uint32_t i = 0; i < off; i++) {
        keys_sorted[i] = kv[i].key;
        point_list[i] = kv[i].val;
        uint32_t cur = (uint32_t)(kv[i].key >> 32);
        if (i == 0) ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(kv[i - 1].key >> 32);
            if (cur != prev) { ranges[2 * prev + 1] = i; ranges[2 * cur] = i; }
        }
        if (i == off - 1) ranges[2 * cur + 1] = off;
    }
    free(kv);
}

/* forward.cu:261-374 (renderCUDA).  One pixel at a time; the batch/shared-memory
 * structure of the kernel does not change per-pixel results. */
void gsr_oracle_render_fwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D,
                           const float* features, const float* conic_opacity, const float* bg, float* final_T,
                           uint32_t* n_contrib, float* out_color)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X;
#pragma omp parallel for schedule(dynamic, 8) collapse(2)
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tile = (py / BLOCK_Y) * gx + (px / BLOCK_X);
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const float pxf = (float)px, pyf = (float)py;
            float T = 1.0f, C[3] = {0, 0, 0};
            uint32_t contributor = 0, last_contributor = 0;
            for (uint32_t k = r0; k < r1; k++) {
                contributor++;
                const uint32_t g = point_list[k];
                const float dx = means2D[2 * g] - pxf, dy = means2D[2 * g + 1] - pyf;
                const float* co = conic_opacity + 4 * g;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                const float alpha = fminf(0.99f, co[3] * expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                const float test_T = T * (1 - alpha);
                if (test_T < 0.0001f) break; /* done = true */
                for (int ch = 0; ch < 3; ch++) C[ch] += features[3 * g + ch] * alpha * T;
                T = test_T;
                last_contributor = contributor;
            }
            const size_t pix = (size_t)W * py + px;
            final_T[pix] = T;
            n_contrib[pix] = last_contributor;
            for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pix] = C[ch] + T * bg[ch];
        }
}

/* backward.cu:399-557 (renderCUDA bwd).  dL_dmean2D is [P][3] (z stays 0),
 * dL_dconic is [P][4] = (xx, xy, unused, yy).  Per-pair float terms are exactly the
 * reference's; the cross-pixel sums are held in double (see header). */
void gsr_oracle_render_bwd(int P, int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* bg,
                           const float* means2D, const float* conic_opacity, const float* colors,
                           const float* final_Ts, const uint32_t* n_contrib, const float* dL_dpixels,
                           float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X;
    double* acc = (double*)calloc((size_t)P * 9 + 1, sizeof(double));
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);
#pragma omp parallel for schedule(dynamic, 8) collapse(2)
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tile = (py / BLOCK_Y) * gx + (px / BLOCK_X);
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const size_t pix = (size_t)W * py + px;
            const float pxf = (float)px, pyf = (float)py;
            const float T_final = final_Ts[pix];
            float T = T_final;
            const uint32_t last_contributor = n_contrib[pix];
            float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0, dL_dpixel[3];
            for (int ch = 0; ch < 3; ch++) dL_dpixel[ch] = dL_dpixels[(size_t)ch * H * W + pix];
            uint32_t contributor = r1 - r0;
            for (uint32_t k = r1; k-- > r0;) {
                contributor--;
                if (contributor >= last_contributor) continue;
                const uint32_t g = point_list[k];
                const float dx = means2D[2 * g] - pxf, dy = means2D[2 * g + 1] - pyf;
                const float* co = conic_opacity + 4 * g;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                const float G = expf(power);
                const float alpha = fminf(0.99f, co[3] * G);
                if (alpha < 1.0f / 255.0f) continue;
                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.0f;
                double* a = acc + (size_t)g * 9;
                for (int ch = 0; ch < 3; ch++) {
                    const float c = colors[3 * g + ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    const float dL_dchannel = dL_dpixel[ch];
                    dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                    const float term = dchannel_dcolor * dL_dchannel;
#pragma omp atomic
                    a[ch] += term;
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                float bg_dot_dpixel = 0;
                for (int i = 0; i < 3; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                const float dL_dG = co[3] * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                const float dG_ddely = -gdy * co[2] - gdx * co[1];
                const float t3 = dL_dG * dG_ddelx * ddelx_dx, t4 = dL_dG * dG_ddely * ddely_dy;
                const float t5 = -0.5f * gdx * dx * dL_dG, t6 = -0.5f * gdx * dy * dL_dG;
                const float t7 = -0.5f * gdy * dy * dL_dG, t8 = G * dL_dalpha;
#pragma omp atomic
                a[3] += t3;
#pragma omp atomic
                a[4] += t4;
#pragma omp atomic
                a[5] += t5;
#pragma omp atomic
                a[6] += t6;
#pragma omp atomic
                a[7] += t7;
#pragma omp atomic
                a[8] += t8;
            }
        }
    for (int g = 0; g < P; g++) {
        const double* a = acc + (size_t)g * 9;
        dL_dcolors[3 * g + 0] = (float)a[0]; dL_dcolors[3 * g + 1] = (float)a[1]; dL_dcolors[3 * g + 2] = (float)a[2];
        dL_dmean2D[3 * g + 0] = (float)a[3]; dL_dmean2D[3 * g + 1] = (float)a[4]; dL_dmean2D[3 * g + 2] = 0.f;
        dL_dconic[4 * g + 0] = (float)a[5]; dL_dconic[4 * g + 1] = (float)a[6]; dL_dconic[4 * g + 2] = 0.f;
        dL_dconic[4 * g + 3] = (float)a[7];
        dL_dopacity[g] = (float)a[8];
    }
    free(acc);
}

/* backward.cu:144-274 (computeCov2DCUDA): overwrites dL_dmeans[idx], writes dL_dcov[6]. */
static void cov2d_bwd(int idx, const float* means, const float* cov3Ds, float fx, float fy, float tanfovx,
                      float tanfovy, const float* view, const float* dL_dconics, float* dL_dmeans, float* dL_dcov)
{
    const float* c = cov3Ds + 6 * idx;
    const float dL_dconic[3] = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
    float a0[3], a1[3], t[3], txtz, tytz;
    ewa_rows(means + 3 * idx, fx, fy, tanfovx, tanfovy, view, a0, a1, t, &txtz, &tytz);
    const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    const float V[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
    float v0[3], v1[3]; /* v0[k] = a0 . V[k][:],  v1[k] = a1 . V[k][:] */
    for (int k = 0; k < 3; k++) {
        v0[k] = a0[0] * V[k][0] + a0[1] * V[k][1] + a0[2] * V[k][2];
        v1[k] = a1[0] * V[k][0] + a1[1] * V[k][1] + a1[2] * V[k][2];
    }
    const float a = a0[0] * v0[0] + a0[1] * v0[1] + a0[2] * v0[2] + 0.3f;
    const float b = a0[0] * v1[0] + a0[1] * v1[1] + a0[2] * v1[2];
    const float cc = a1[0] * v1[0] + a1[1] * v1[1] + a1[2] * v1[2] + 0.3f;
    const float denom = a * cc - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float* o = dL_dcov + 6 * idx;
    if (denom2inv != 0) {
        dL_da = denom2inv * (-cc * cc * dL_dconic[0] + 2 * b * cc * dL_dconic[1] + (denom - a * cc) * dL_dconic[2]);
        dL_dc = denom2inv * (-a * a * dL_dconic[2] + 2 * a * b * dL_dconic[1] + (denom - a * cc) * dL_dconic[0]);
        dL_db = denom2inv * 2 * (b * cc * dL_dconic[0] - (denom + 2 * b * b) * dL_dconic[1] + a * b * dL_dconic[2]);
        o[0] = (a0[0] * a0[0] * dL_da + a0[0] * a1[0] * dL_db + a1[0] * a1[0] * dL_dc);
        o[3] = (a0[1] * a0[1] * dL_da + a0[1] * a1[1] * dL_db + a1[1] * a1[1] * dL_dc);
        o[5] = (a0[2] * a0[2] * dL_da + a0[2] * a1[2] * dL_db + a1[2] * a1[2] * dL_dc);
        o[1] = 2 * a0[0] * a0[1] * dL_da + (a0[0] * a1[1] + a0[1] * a1[0]) * dL_db + 2 * a1[0] * a1[1] * dL_dc;
        o[2] = 2 * a0[0] * a0[2] * dL_da + (a0[0] * a1[2] + a0[2] * a1[0]) * dL_db + 2 * a1[0] * a1[2] * dL_dc;
        o[4] = 2 * a0[2] * a0[1] * dL_da + (a0[1] * a1[2] + a0[2] * a1[1]) * dL_db + 2 * a1[1] * a1[2] * dL_dc;
    } else {
        for (int i = 0; i < 6; i++) o[i] = 0;
    }
    float dT0[3], dT1[3];
    for (int k = 0; k < 3; k++) {
        dT0[k] = 2 * v0[k] * dL_da + v1[k] * dL_db;
        dT1[k] = 2 * v1[k] * dL_dc + v0[k] * dL_db;
    }
    /* W[i][j] (glm) = view[4j + i] */
    const float dL_dJ00 = view[0] * dT0[0] + view[4] * dT0[1] + view[8] * dT0[2];
    const float dL_dJ02 = view[2] * dT0[0] + view[6] * dT0[1] + view[10] * dT0[2];
    const float dL_dJ11 = view[1] * dT1[0] + view[5] * dT1[1] + view[9] * dT1[2];
    const float dL_dJ12 = view[2] * dT1[0] + view[6] * dT1[1] + view[10] * dT1[2];
    const float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
    const float dL_dtx = x_grad_mul * -fx * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -fy * tz2 * dL_dJ12;
    const float dL_dtz = -fx * tz2 * dL_dJ00 - fy * tz2 * dL_dJ11 + (2 * fx * t[0]) * tz3 * dL_dJ02 +
                         (2 * fy * t[1]) * tz3 * dL_dJ12;
    /* auxiliary.h:89-97 transformVec4x3Transpose */
    dL_dmeans[3 * idx + 0] = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz;
    dL_dmeans[3 * idx + 1] = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz;
    dL_dmeans[3 * idx + 2] = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;
}

/* backward.cu:20-139 (computeColorFromSH bwd) */
static void sh_bwd(int idx, int deg, int M, const float* means, const float* campos, const float* shs,
                   const uint8_t* clamped, const float* dL_dcolor, float* dL_dmeans, float* dL_dshs)
{
    const float* pos = means + 3 * idx;
    const float dir_orig[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
    const float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
    const float x = dir_orig[0] / len, y = dir_orig[1] / len, z = dir_orig[2] / len;
    const float* sh = shs + (size_t)idx * M * 3;
    float* dsh = dL_dshs + (size_t)idx * M * 3;
    float dRGB[3];
    for (int ch = 0; ch < 3; ch++) dRGB[ch] = dL_dcolor[3 * idx + ch] * (clamped[3 * idx + ch] ? 0.f : 1.f);
    float dx_[3] = {0, 0, 0}, dy_[3] = {0, 0, 0}, dz_[3] = {0, 0, 0};
#define S(k) sh[3 * (k) + ch]
#define DS(k, v) dsh[3 * (k) + ch] = (v) * dRGB[ch]
    for (int ch = 0; ch < 3; ch++) {
        DS(0, SH_C0);
        if (deg > 0) {
            DS(1, -SH_C1 * y); DS(2, SH_C1 * z); DS(3, -SH_C1 * x);
            dx_[ch] = -SH_C1 * S(3); dy_[ch] = -SH_C1 * S(1); dz_[ch] = SH_C1 * S(2);
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                DS(4, SH_C2[0] * xy); DS(5, SH_C2[1] * yz); DS(6, SH_C2[2] * (2.f * zz - xx - yy));
                DS(7, SH_C2[3] * xz); DS(8, SH_C2[4] * (xx - yy));
                dx_[ch] += SH_C2[0] * y * S(4) + SH_C2[2] * 2.f * -x * S(6) + SH_C2[3] * z * S(7) + SH_C2[4] * 2.f * x * S(8);
                dy_[ch] += SH_C2[0] * x * S(4) + SH_C2[1] * z * S(5) + SH_C2[2] * 2.f * -y * S(6) + SH_C2[4] * 2.f * -y * S(8);
                dz_[ch] += SH_C2[1] * y * S(5) + SH_C2[2] * 2.f * 2.f * z * S(6) + SH_C2[3] * x * S(7);
                if (deg > 2) {
                    DS(9, SH_C3[0] * y * (3.f * xx - yy)); DS(10, SH_C3[1] * xy * z);
                    DS(11, SH_C3[2] * y * (4.f * zz - xx - yy));
                    DS(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                    DS(13, SH_C3[4] * x * (4.f * zz - xx - yy)); DS(14, SH_C3[5] * z * (xx - yy));
                    DS(15, SH_C3[6] * x * (xx - 3.f * yy));
                    dx_[ch] += (SH_C3[0] * S(9) * 3.f * 2.f * xy + SH_C3[1] * S(10) * yz + SH_C3[2] * S(11) * -2.f * xy +
                                SH_C3[3] * S(12) * -3.f * 2.f * xz + SH_C3[4] * S(13) * (-3.f * xx + 4.f * zz - yy) +
                                SH_C3[5] * S(14) * 2.f * xz + SH_C3[6] * S(15) * 3.f * (xx - yy));
                    dy_[ch] += (SH_C3[0] * S(9) * 3.f * (xx - yy) + SH_C3[1] * S(10) * xz +
                                SH_C3[2] * S(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * S(12) * -3.f * 2.f * yz +
                                SH_C3[4] * S(13) * -2.f * xy + SH_C3[5] * S(14) * -2.f * yz +
                                SH_C3[6] * S(15) * -3.f * 2.f * xy);
                    dz_[ch] += (SH_C3[1] * S(10) * xy + SH_C3[2] * S(11) * 4.f * 2.f * yz +
                                SH_C3[3] * S(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * S(13) * 4.f * 2.f * xz +
                                SH_C3[5] * S(14) * (xx - yy));
                }
            }
        }
    }
#undef S
#undef DS
    const float dd[3] = {dx_[0] * dRGB[0] + dx_[1] * dRGB[1] + dx_[2] * dRGB[2],
                         dy_[0] * dRGB[0] + dy_[1] * dRGB[1] + dy_[2] * dRGB[2],
                         dz_[0] * dRGB[0] + dz_[1] * dRGB[1] + dz_[2] * dRGB[2]};
    /* auxiliary.h:107-117 dnormvdv(float3) */
    const float* v = dir_orig;
    const float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dL_dmeans[3 * idx + 0] += ((+sum2 - v[0] * v[0]) * dd[0] - v[1] * v[0] * dd[1] - v[2] * v[0] * dd[2]) * invsum32;
    dL_dmeans[3 * idx + 1] += (-v[0] * v[1] * dd[0] + (sum2 - v[1] * v[1]) * dd[1] - v[2] * v[1] * dd[2]) * invsum32;
    dL_dmeans[3 * idx + 2] += (-v[0] * v[2] * dd[0] - v[1] * v[2] * dd[1] + (sum2 - v[2] * v[2]) * dd[2]) * invsum32;
}

/* backward.cu:278-341 (computeCov3D bwd): raw-quaternion gradient, no d(normalise). */
static void cov3d_bwd(int idx, const float* scale, float mod, const float* rot, const float* dL_dcov3Ds,
                      float* dL_dscales, float* dL_drots)
{
    float R[3][3];
    quat_to_R(rot, R);
    const float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    const float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    float M[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) M[i][j] = s[i] * R[j][i];
    const float* d = dL_dcov3Ds + 6 * idx;
    const float dS[3][3] = {{d[0], 0.5f * d[1], 0.5f * d[2]}, {0.5f * d[1], d[3], 0.5f * d[4]}, {0.5f * d[2], 0.5f * d[4], d[5]}};
    float dM[3][3]; /* dL_dM = 2 * M * dL_dSigma */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) dM[i][j] = 2.0f * (M[i][0] * dS[0][j] + M[i][1] * dS[1][j] + M[i][2] * dS[2][j]);
    /* dL_dscale_i = dot(column i of R, row i of dM) */
    for (int i = 0; i < 3; i++) dL_dscales[3 * idx + i] = R[0][i] * dM[i][0] + R[1][i] * dM[i][1] + R[2][i] * dM[i][2];
    float G[3][3]; /* G[a][b] = (dL_dMt[a][b] after the per-column scale) = s_a * dM[a][b] */
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) G[a][b] = s[a] * dM[a][b];
    float* q = dL_drots + 4 * idx;
    q[0] = 2 * z * (G[0][1] - G[1][0]) + 2 * y * (G[2][0] - G[0][2]) + 2 * x * (G[1][2] - G[2][1]);
    q[1] = 2 * y * (G[1][0] + G[0][1]) + 2 * z * (G[2][0] + G[0][2]) + 2 * r * (G[1][2] - G[2][1]) - 4 * x * (G[2][2] + G[1][1]);
    q[2] = 2 * x * (G[1][0] + G[0][1]) + 2 * r * (G[2][0] - G[0][2]) + 2 * z * (G[1][2] + G[2][1]) - 4 * y * (G[2][2] + G[0][0]);
    q[3] = 2 * r * (G[0][1] - G[1][0]) + 2 * x * (G[2][0] + G[0][2]) + 2 * y * (G[1][2] + G[2][1]) - 4 * z * (G[1][1] + G[0][0]);
}

/* backward.cu:559-633 (BACKWARD::preprocess): computeCov2DCUDA for every visible
 * Gaussian, then preprocessCUDA bwd (:346-396).  All gradient arrays must arrive
 * zero-filled (rasterize_points.cu:151-159).  cov3Ds = precomputed or forward's. */
void gsr_oracle_preprocess_bwd(int P, int D, int M, const float* means3D, const int* radii, const float* shs,
                               const uint8_t* clamped, const float* scales, const float* rotations,
                               float scale_modifier, const float* cov3Ds, const float* view, const float* proj,
                               int W, int H, float tanfovx, float tanfovy, const float* campos,
                               const float* dL_dmean2D, const float* dL_dconic, float* dL_dmeans3D,
                               float* dL_dcolor, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot)
{
    const float focal_y = H / (2.0f * tanfovy), focal_x = W / (2.0f * tanfovx); /* rasterizer_impl.cu:381-382 */
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        cov2d_bwd(idx, means3D, cov3Ds, focal_x, focal_y, tanfovx, tanfovy, view, dL_dconic, dL_dmeans3D, dL_dcov3D);
        const float* m = means3D + 3 * idx;
        float mh[4];
        xform4x4(m, proj, mh);
        const float m_w = 1.0f / (mh[3] + 0.0000001f);
        const float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
        const float gx = dL_dmean2D[3 * idx], gy = dL_dmean2D[3 * idx + 1];
        dL_dmeans3D[3 * idx + 0] += (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
        dL_dmeans3D[3 * idx + 1] += (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
        dL_dmeans3D[3 * idx + 2] += (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
        if (shs) sh_bwd(idx, D, M, means3D, campos, shs, clamped, dL_dcolor, dL_dmeans3D, dL_dsh);
        if (scales) cov3d_bwd(idx, scales + 3 * idx, scale_modifier, rotations + 4 * idx, dL_dcov3D, dL_dscale, dL_drot);
    }
}
