// gsr_producers.hip -- producers of rasterizer inputs that the reference builds as PyTorch elementwise chains
// (SURVEY.md section 8f row 2).
//
// sh_to_rgb: SuGaR.get_points_rgb (gaustar_scene/sugar_model.py:674-718) =
//     clamp_min(eval_sh(sh_levels - 1, sh[:, :sh_levels^2], normalize(positions - camera_center)) + 0.5, 0)
// with eval_sh of gaustar_utils/spherical_harmonics.py:117-172 -- ~60 elementwise kernels forward and as many
// again under autograd.  It is the same arithmetic as the rasterizer's in-kernel SH (computeColorFromSH,
// DGR/cuda_rasterizer/forward.cu:20-71 / backward.cu:20-139), so the two kernels below share sh_basis() and
// sh_colour_backward() with gsr_preprocess.hip / gsr_geom_bwd.hip.  One thread per point; sh is [P, M, 3]
// (the reference's sh_coordinates layout), only the first (D+1)^2 coefficients are used.
#include "gsr_internal.h"
#include "gsr_sort.h"   // lane_xor_u32: DPP / ds_swizzle lane exchanges

namespace gsr {

__global__ void __launch_bounds__(256)
sh_to_rgb_kernel(int P, int D, int M, const float* __restrict__ positions, const float* __restrict__ campos,
                 const float* __restrict__ shs, const float* __restrict__ shs_rest, const float* __restrict__ view,
                 float* __restrict__ rgb, int stride, const float* __restrict__ densities, float* __restrict__ opacity)
{
    // shs_rest == nullptr: shs is [P, M, 3]; otherwise shs is SuGaR's `_sh_coordinates_dc` [P, 1, 3] and shs_rest its
    // `_sh_coordinates_rest` [P, M - 1, 3] (sugar_model.py:449-450 concatenates them on every access: 53 MB written and
    // read again per render at config C) -- both land in the same LDS rows.  densities != nullptr: opacity =
    // sigmoid(density) rides along (SuGaR.strengths, sugar_model.py:442-447).
    extern __shared__ float sh_lds[];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    float* wave_rows = sh_lds + (size_t)(threadIdx.x >> 6) * 64 * sh_row_stride(M);
    if (shs_rest == nullptr) {
        sh_stage_load(wave_rows, shs, (size_t)(idx - lane), P, M, lane);   // coalesced, see gsr_internal.h
    } else {
        sh_stage_load_cols(wave_rows, shs, (size_t)(idx - lane), P, 3, sh_row_stride(M), 0, lane);
        sh_stage_load_cols(wave_rows, shs_rest, (size_t)(idx - lane), P, 3 * (M - 1), sh_row_stride(M), 3, lane);
    }
    __builtin_amdgcn_wave_barrier();
    if (idx >= P) return;
    const size_t i = (size_t)idx;
    if (densities != nullptr) opacity[i] = 1.0f / (1.0f + expf(-densities[i]));   // torch.sigmoid
    const Vec3 p = load3(positions, i);
    const float dx = p.x - campos[0], dy = p.y - campos[1], dz = p.z - campos[2];
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    float basis[SH_MAX_BASIS];
    sh_basis(D, dx * inv, dy * inv, dz * inv, basis);
    const int nb = (D + 1) * (D + 1);
    const float* sh = wave_rows + lane * sh_row_stride(M);
    float cr = 0.f, cg = 0.f, cb = 0.f;
    for (int k = 0; k < nb; k++) { cr += basis[k] * sh[3 * k]; cg += basis[k] * sh[3 * k + 1]; cb += basis[k] * sh[3 * k + 2]; }
    float* o = rgb + (size_t)stride * i;
    o[0] = fmaxf(cr + 0.5f, 0.0f);
    o[1] = fmaxf(cg + 0.5f, 0.0f);
    o[2] = fmaxf(cb + 0.5f, 0.0f);
    if (view != nullptr) {   // second target of a two-target render: view-space depth as a colour (refine.py:603-605)
        const float z = p.x * view[2] + p.y * view[6] + p.z * view[10] + view[14];   // (p, 1) . column 2 of the view matrix
        for (int c = 3; c < stride; c++) o[c] = z;                                    // stride 6: z x 3, stride 4: z once
    }
}

__global__ void __launch_bounds__(256)
sh_to_rgb_bwd_kernel(int P, int D, int M, const float* __restrict__ positions, const float* __restrict__ campos,
                     const float* __restrict__ shs, const float* __restrict__ shs_rest, const float* __restrict__ view,
                     const float* __restrict__ dL_drgb, int stride, float* __restrict__ dL_dsh, float* __restrict__ dL_dsh_rest,
                     float* __restrict__ dL_dpos, int accumulate_pos, const float* __restrict__ opacity,
                     const float* __restrict__ dL_dopacity, float* __restrict__ dL_ddensity)
{
    // (shs_rest / dL_dsh_rest: the two-array layout of the forward kernel.  accumulate_pos: dL_dpos already holds the
    // rasterizer's gradient w.r.t. the positions and takes this one on top -- the add autograd would launch.  opacity !=
    // nullptr: dL_ddensity = dL_dopacity (1 - o) o, torch's sigmoid_backward.)
    extern __shared__ float sh_lds[];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    float* wave_rows = sh_lds + (size_t)(threadIdx.x >> 6) * 64 * sh_row_stride(M);
    if (shs_rest == nullptr) {
        sh_stage_load(wave_rows, shs, (size_t)(idx - lane), P, M, lane);
    } else {
        sh_stage_load_cols(wave_rows, shs, (size_t)(idx - lane), P, 3, sh_row_stride(M), 0, lane);
        sh_stage_load_cols(wave_rows, shs_rest, (size_t)(idx - lane), P, 3 * (M - 1), sh_row_stride(M), 3, lane);
    }
    __builtin_amdgcn_wave_barrier();
    if (idx < P) {
        const size_t i = (size_t)idx;
        const Vec3 p = load3(positions, i);
        const float* gi = dL_drgb + (size_t)stride * i;
        const float dcol[3] = {gi[0], gi[1], gi[2]};
        float gx = 0.f, gy = 0.f, gz = 0.f;
        float* row = wave_rows + lane * sh_row_stride(M);
        sh_colour_backward(D, M, p, campos, row, dcol, row, gx, gy, gz);   // gradient row replaces the coefficient row
        if (view != nullptr) {   // the three depth channels all carry d z / d p = column 2 of the view matrix
            float gzv = gi[3];
            for (int c = 4; c < stride; c++) gzv += gi[c];
            gx += gzv * view[2]; gy += gzv * view[6]; gz += gzv * view[10];
        }
        if (accumulate_pos) { gx += dL_dpos[3 * i]; gy += dL_dpos[3 * i + 1]; gz += dL_dpos[3 * i + 2]; }
        dL_dpos[3 * i] = gx; dL_dpos[3 * i + 1] = gy; dL_dpos[3 * i + 2] = gz;
        if (opacity != nullptr) {
#pragma clang fp contract(off)
            const float o = opacity[i];
            dL_ddensity[i] = (dL_dopacity[i] * (1.0f - o)) * o;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (dL_dsh_rest == nullptr) {
        sh_stage_store(wave_rows, dL_dsh, (size_t)(idx - lane), P, M, lane);
    } else {
        sh_stage_store_cols(wave_rows, dL_dsh, (size_t)(idx - lane), P, 3, sh_row_stride(M), 0, lane);
        sh_stage_store_cols(wave_rows, dL_dsh_rest, (size_t)(idx - lane), P, 3 * (M - 1), sh_row_stride(M), 3, lane);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Mesh-bound Gaussians: SuGaR.points / .scaling / .quaternions (gaustar_scene/sugar_model.py:417-435, :457-476,
// :478-508) -- the per-call rebuild of means, scales and rotations of the 6 Gaussians bound to every triangle,
// ~50 PyTorch kernels forward (gathers, normalisations, cross products, a batched matrix product and
// pytorch3d's matrix_to_quaternion) and their autograd mirror, run before EACH render.
//
// One thread per FACE, looping over its G Gaussians: the face frame (normal R0, first edge bR1, bR2 = R0 x bR1) is
// built once, each Gaussian turns (bR1, bR2) by its learned unit complex number, optionally pre-multiplies the
// loose-bind rotation delta_r, and converts the matrix to a quaternion (best-conditioned candidate, as pytorch3d).
//
// Backward: the rasterizer returns dL/dq for the unit quaternion q it was given.  Every parameter upstream moves
// R = R(q) inside SO(3), so only the tangential part of dL/dq matters: for R' = exp([w]x) R, dq = 1/2 (0, w) (x) q,
//     G = dL/dw = 1/2 ( -g_w v + w g_v + v x g_v ),   q = (w, v), dL/dq = (g_w, g_v),
// and the gradient w.r.t. the matrix that reproduces G along every rotation direction is  dL/dR = 1/2 [G]x R
// (< 1/2 [G]x R, [dw]x R > = G . dw).  From there it is plain chain rule through the frame (no derivative of
// matrix_to_quaternion is ever needed); a face's three vertex gradients are summed over its G Gaussians in
// registers and leave as 9 atomics.
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float norm(V3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ V3 ld3(const float* p, size_t i) { return V3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
__device__ __forceinline__ void st3(float* p, size_t i, V3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }
// d/dx of x / max(|x|, eps) applied to g (the eps branch, reached only by degenerate input, is treated as a constant scale)
__device__ __forceinline__ V3 normalize_bwd(V3 unit, float len, float eps, V3 g)
{
    const float inv = 1.0f / fmaxf(len, eps);
    return len > eps ? inv * (g - dot(g, unit) * unit) : inv * g;
}

struct FaceFrame { V3 v0, v1, v2, e1, e2, nrm, R0, a, bR1, cr, bR2; float len, la, lc; };
__device__ __forceinline__ FaceFrame face_frame(const float* __restrict__ verts, const long long* __restrict__ faces, size_t f)
{
    FaceFrame F;
    F.v0 = ld3(verts, (size_t)faces[3 * f]); F.v1 = ld3(verts, (size_t)faces[3 * f + 1]); F.v2 = ld3(verts, (size_t)faces[3 * f + 2]);
    F.e1 = F.v1 - F.v0; F.e2 = F.v2 - F.v0;
    F.nrm = cross(F.e1, F.e2);
    F.len = norm(F.nrm);
    const V3 n0 = (1.0f / fmaxf(F.len, 1e-6f)) * F.nrm;              // pytorch3d face normal
    F.R0 = (1.0f / fmaxf(norm(n0), 1e-12f)) * n0;                    // F.normalize, sugar_model.py:483
    F.a = F.v0 - F.v1; F.la = norm(F.a);
    F.bR1 = (1.0f / fmaxf(F.la, 1e-12f)) * F.a;                      // :487
    F.cr = cross(F.R0, F.bR1); F.lc = norm(F.cr);
    F.bR2 = (1.0f / fmaxf(F.lc, 1e-12f)) * F.cr;                     // :490
    return F;
}

// R (row-major r[i][j]) of one Gaussian; also returns the unit complex number and, if present, D = R(delta_r).
struct GaussFrame { float R[3][3]; float D[3][3]; float qc, qs, lq; float dw; V3 dv; float ld; bool loose; };
__device__ __forceinline__ GaussFrame gauss_frame(const FaceFrame& F, const float* __restrict__ raw_complex,
                                                  const float* __restrict__ delta_r, size_t n)
{
    GaussFrame Gf;
    const float cx = raw_complex[2 * n], cy = raw_complex[2 * n + 1];
    Gf.lq = sqrtf(cx * cx + cy * cy);
    const float iq = 1.0f / fmaxf(Gf.lq, 1e-12f);
    Gf.qc = cx * iq; Gf.qs = cy * iq;                                                   // :493
    const V3 R1 = Gf.qc * F.bR1 + Gf.qs * F.bR2, R2 = (-Gf.qs) * F.bR1 + Gf.qc * F.bR2;   // :494-495
    const float B[3][3] = {{F.R0.x, R1.x, R2.x}, {F.R0.y, R1.y, R2.y}, {F.R0.z, R1.z, R2.z}};   // columns R0 | R1 | R2
    Gf.loose = delta_r != nullptr;
    if (Gf.loose) {
        const float r = delta_r[4 * n], i = delta_r[4 * n + 1], j = delta_r[4 * n + 2], k = delta_r[4 * n + 3];
        const float ss = r * r + i * i + j * j + k * k;
        const float two_s = 2.0f / ss;                                                  // quaternion_to_matrix
        Gf.ld = sqrtf(ss);
        Gf.dw = r / Gf.ld; Gf.dv = v3(i / Gf.ld, j / Gf.ld, k / Gf.ld);
        const float D[3][3] = {{1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r)},
                               {two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r)},
                               {two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)}};
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) {
                Gf.D[a][b] = D[a][b];
                Gf.R[a][b] = D[a][0] * B[0][b] + D[a][1] * B[1][b] + D[a][2] * B[2][b];   // bmm(delta_r_mat, R), :505
            }
    } else {
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) Gf.R[a][b] = B[a][b];
    }
    return Gf;
}

// pytorch3d matrix_to_quaternion (best-conditioned candidate) followed by F.normalize (:506-508)
__device__ __forceinline__ void matrix_to_unit_quaternion(const float (&m)[3][3], float (&q)[4])
{
    const float t0 = 1.0f + m[0][0] + m[1][1] + m[2][2], t1 = 1.0f + m[0][0] - m[1][1] - m[2][2],
                t2 = 1.0f - m[0][0] + m[1][1] - m[2][2], t3 = 1.0f - m[0][0] - m[1][1] + m[2][2];
    const float qa[4] = {t0 > 0.f ? sqrtf(t0) : 0.f, t1 > 0.f ? sqrtf(t1) : 0.f, t2 > 0.f ? sqrtf(t2) : 0.f,
                         t3 > 0.f ? sqrtf(t3) : 0.f};
    int best = 0;
#pragma unroll
    for (int c = 1; c < 4; c++) if (qa[c] > qa[best]) best = c;   // first maximum, like argmax
    float cand[4];
    if (best == 0) { cand[0] = qa[0] * qa[0]; cand[1] = m[2][1] - m[1][2]; cand[2] = m[0][2] - m[2][0]; cand[3] = m[1][0] - m[0][1]; }
    else if (best == 1) { cand[0] = m[2][1] - m[1][2]; cand[1] = qa[1] * qa[1]; cand[2] = m[1][0] + m[0][1]; cand[3] = m[0][2] + m[2][0]; }
    else if (best == 2) { cand[0] = m[0][2] - m[2][0]; cand[1] = m[1][0] + m[0][1]; cand[2] = qa[2] * qa[2]; cand[3] = m[1][2] + m[2][1]; }
    else { cand[0] = m[1][0] - m[0][1]; cand[1] = m[2][0] + m[0][2]; cand[2] = m[2][1] + m[1][2]; cand[3] = qa[3] * qa[3]; }
    const float d = 1.0f / (2.0f * fmaxf(qa[best], 0.1f));
    float n2 = 0.f;
#pragma unroll
    for (int c = 0; c < 4; c++) { cand[c] *= d; n2 += cand[c] * cand[c]; }
    const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
#pragma unroll
    for (int c = 0; c < 4; c++) q[c] = cand[c] * inv;
}

__global__ void __launch_bounds__(128)
mesh_gaussians_fwd_kernel(int F, int G, const float* __restrict__ verts, const long long* __restrict__ faces,
                          const float* __restrict__ bary, const float* __restrict__ raw_scales,
                          const float* __restrict__ raw_complex, float thickness, float min_scale, float max_scale,
                          const float* __restrict__ delta_t, const float* __restrict__ delta_r,
                          float* __restrict__ points, float* __restrict__ scaling, float* __restrict__ quats,
                          float* __restrict__ clear, long long clear_n)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = f; i < clear_n; i += (long long)gridDim.x * blockDim.x) clear[i] = 0.f;   // (side job, see gsr.h)
    if (f >= F) return;
    const FaceFrame Ff = face_frame(verts, faces, (size_t)f);
    for (int g = 0; g < G; g++) {
        const size_t n = (size_t)f * G + g;
        V3 p = (bary[3 * g] * Ff.v0 + bary[3 * g + 1] * Ff.v1) + bary[3 * g + 2] * Ff.v2;   // :428-429
        if (delta_t) p = p + ld3(delta_t, n);                                                 // :432
        st3(points, n, p);
        const float s0 = fmaxf(fminf(__expf(raw_scales[2 * n]), max_scale), min_scale);       // :461-465
        const float s1 = fmaxf(fminf(__expf(raw_scales[2 * n + 1]), max_scale), min_scale);
        st3(scaling, n, v3(thickness, s0, s1));                                               // :472-475
        const GaussFrame Gf = gauss_frame(Ff, raw_complex, delta_r, n);
        float q[4];
        matrix_to_unit_quaternion(Gf.R, q);
        reinterpret_cast<float4*>(quats)[n] = make_float4(q[0], q[1], q[2], q[3]);
    }
}

__global__ void __launch_bounds__(128)
mesh_gaussians_bwd_kernel(int F, int G, const float* __restrict__ verts, const long long* __restrict__ faces,
                          const float* __restrict__ bary, const float* __restrict__ raw_scales,
                          const float* __restrict__ raw_complex, float min_scale, float max_scale,
                          const float* __restrict__ delta_r, const float* __restrict__ dL_dpoints,
                          const float* __restrict__ dL_dscaling, const float* __restrict__ dL_dquats,
                          float* __restrict__ dL_dverts, float* __restrict__ dL_draw_scales,
                          float* __restrict__ dL_draw_complex, float* __restrict__ dL_ddelta_t,
                          float* __restrict__ dL_ddelta_r)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const FaceFrame Ff = face_frame(verts, faces, (size_t)f);
    V3 gv0 = v3(0, 0, 0), gv1 = gv0, gv2 = gv0, gR0 = gv0, gbR1 = gv0, gbR2 = gv0;
    for (int g = 0; g < G; g++) {
        const size_t n = (size_t)f * G + g;
        // means
        const V3 gm = dL_dpoints ? ld3(dL_dpoints, n) : v3(0, 0, 0);
        gv0 = gv0 + bary[3 * g] * gm; gv1 = gv1 + bary[3 * g + 1] * gm; gv2 = gv2 + bary[3 * g + 2] * gm;
        if (dL_ddelta_t) st3(dL_ddelta_t, n, gm);
        // scales: exp, then clamp_max, clamp_min masks (x <= max, y >= min)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float e = __expf(raw_scales[2 * n + j]);
            const bool pass = e <= max_scale && fminf(e, max_scale) >= min_scale;
            dL_draw_scales[2 * n + j] = (dL_dscaling && pass) ? dL_dscaling[3 * n + 1 + j] * e : 0.f;
        }
        // rotation
        const GaussFrame Gf = gauss_frame(Ff, raw_complex, delta_r, n);
        float q[4];
        matrix_to_unit_quaternion(Gf.R, q);
        const float4 gq = dL_dquats ? reinterpret_cast<const float4*>(dL_dquats)[n] : make_float4(0.f, 0.f, 0.f, 0.f);
        const V3 qv = v3(q[1], q[2], q[3]), gqv = v3(gq.y, gq.z, gq.w);
        const V3 Gw = 0.5f * (((-gq.x) * qv + q[0] * gqv) + cross(qv, gqv));           // dL/d(rotation vector)
        // dL/dR = 1/2 [G]x R, column by column
        V3 A[3];
#pragma unroll
        for (int c = 0; c < 3; c++) A[c] = 0.5f * cross(Gw, v3(Gf.R[0][c], Gf.R[1][c], Gf.R[2][c]));
        V3 gB[3];
        if (Gf.loose) {
            // delta_r: dphi = 2 vec(dd^ (x) conj(d^)), dd^ = (I - d^ d^T) dd / |d|
            const float hw = -2.0f * dot(Gw, Gf.dv);
            const V3 hv = 2.0f * (Gf.dw * Gw - cross(Gf.dv, Gw));
            const float il = 1.0f / Gf.ld;
            if (dL_ddelta_r) reinterpret_cast<float4*>(dL_ddelta_r)[n] = make_float4(hw * il, hv.x * il, hv.y * il, hv.z * il);
#pragma unroll
            for (int c = 0; c < 3; c++)   // dL/dB = D^T dL/dR
                gB[c] = v3(Gf.D[0][0] * A[c].x + Gf.D[1][0] * A[c].y + Gf.D[2][0] * A[c].z,
                           Gf.D[0][1] * A[c].x + Gf.D[1][1] * A[c].y + Gf.D[2][1] * A[c].z,
                           Gf.D[0][2] * A[c].x + Gf.D[1][2] * A[c].y + Gf.D[2][2] * A[c].z);
        } else {
#pragma unroll
            for (int c = 0; c < 3; c++) gB[c] = A[c];
        }
        gR0 = gR0 + gB[0];
        const float g_qc = dot(gB[1], Ff.bR1) + dot(gB[2], Ff.bR2), g_qs = dot(gB[1], Ff.bR2) - dot(gB[2], Ff.bR1);
        gbR1 = gbR1 + (Gf.qc * gB[1] - Gf.qs * gB[2]);
        gbR2 = gbR2 + (Gf.qs * gB[1] + Gf.qc * gB[2]);
        {   // through the normalisation of the raw complex number (:493)
            const float inv = 1.0f / fmaxf(Gf.lq, 1e-12f);
            const float d = g_qc * Gf.qc + g_qs * Gf.qs;
            const bool reg = Gf.lq > 1e-12f;
            dL_draw_complex[2 * n] = reg ? inv * (g_qc - d * Gf.qc) : inv * g_qc;
            dL_draw_complex[2 * n + 1] = reg ? inv * (g_qs - d * Gf.qs) : inv * g_qs;
        }
    }
    // face frame -> vertices
    const V3 gcr = normalize_bwd(Ff.bR2, Ff.lc, 1e-12f, gbR2);       // bR2 = normalize(R0 x bR1)
    gR0 = gR0 + cross(Ff.bR1, gcr);
    gbR1 = gbR1 + cross(gcr, Ff.R0);
    const V3 ga = normalize_bwd(Ff.bR1, Ff.la, 1e-12f, gbR1);        // bR1 = normalize(v0 - v1)
    gv0 = gv0 + ga; gv1 = gv1 - ga;
    const V3 gn = normalize_bwd(Ff.R0, Ff.len, 1e-6f, gR0);          // R0 = normalize((e1 x e2) / max(|.|, 1e-6))
    const V3 ge1 = cross(Ff.e2, gn), ge2 = cross(gn, Ff.e1);
    gv1 = gv1 + ge1; gv2 = gv2 + ge2; gv0 = gv0 - (ge1 + ge2);
    const size_t i0 = (size_t)faces[3 * (size_t)f], i1 = (size_t)faces[3 * (size_t)f + 1], i2 = (size_t)faces[3 * (size_t)f + 2];
    atomicAdd(dL_dverts + 3 * i0, gv0.x); atomicAdd(dL_dverts + 3 * i0 + 1, gv0.y); atomicAdd(dL_dverts + 3 * i0 + 2, gv0.z);
    atomicAdd(dL_dverts + 3 * i1, gv1.x); atomicAdd(dL_dverts + 3 * i1 + 1, gv1.y); atomicAdd(dL_dverts + 3 * i1 + 2, gv1.z);
    atomicAdd(dL_dverts + 3 * i2, gv2.x); atomicAdd(dL_dverts + 3 * i2 + 1, gv2.y); atomicAdd(dL_dverts + 3 * i2 + 2, gv2.z);
}

// ---- G <= 8 (GauSTAR binds 1, 3, 4 or 6 Gaussians to a triangle): one LANE per Gaussian, eight lanes per face.
// The per-face kernels above keep a single thread busy with a face's G Gaussians one after the other: 81 920 faces are
// 1 280 waves, barely one per SIMD, and the chain of matrix / quaternion arithmetic runs at the latency of a lone wave
// (24 us forward, 57 us backward for 491 520 Gaussians, against 36 MB of traffic).  Here every lane rebuilds the face frame
// (its loads are the same addresses across the face's lanes) and does ONE Gaussian; the backward sums the 18 frame / vertex
// gradient components over the face's eight lanes with three DPP exchanges each and lane 0 carries them to the vertices.
constexpr int LPF = 8;   // lanes per face

__device__ __forceinline__ float face_sum(float v)
{
    v += __uint_as_float(lane_xor_u32<1>(__float_as_uint(v)));
    v += __uint_as_float(lane_xor_u32<2>(__float_as_uint(v)));
    v += __uint_as_float(lane_xor_u32<4>(__float_as_uint(v)));
    return v;
}
__device__ __forceinline__ V3 face_sum(V3 v) { return v3(face_sum(v.x), face_sum(v.y), face_sum(v.z)); }

__global__ void __launch_bounds__(256)
mesh_gaussians_fwd8_kernel(int F, int G, const float* __restrict__ verts, const long long* __restrict__ faces,
                           const float* __restrict__ bary, const float* __restrict__ raw_scales,
                           const float* __restrict__ raw_complex, float thickness, float min_scale, float max_scale,
                           const float* __restrict__ delta_t, const float* __restrict__ delta_r,
                           float* __restrict__ points, float* __restrict__ scaling, float* __restrict__ quats,
                           float* __restrict__ clear, long long clear_n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    // side job: the vertex-gradient accumulator of the backward to come is cleared here (as a fill of its own in front of
    // the backward it was a 4.8 us launch on the stream for 0.5 MB)
    for (long long i = t; i < clear_n; i += (long long)gridDim.x * blockDim.x) clear[i] = 0.f;
    const int f = t / LPF, g = t - f * LPF;
    if (f >= F || g >= G) return;
    const size_t n = (size_t)f * G + g;
    // every input of the lane is requested before anything is stored or computed (loads behind stores or behind values
    // derived from earlier loads each cost a full trip to memory)
    const float b0 = bary[3 * g], b1 = bary[3 * g + 1], b2 = bary[3 * g + 2];
    const float2 rs = reinterpret_cast<const float2*>(raw_scales)[n];
    const V3 dt = delta_t ? ld3(delta_t, n) : v3(0.f, 0.f, 0.f);
    const FaceFrame Ff = face_frame(verts, faces, (size_t)f);
    const GaussFrame Gf = gauss_frame(Ff, raw_complex, delta_r, n);
    const V3 p = ((b0 * Ff.v0 + b1 * Ff.v1) + b2 * Ff.v2) + dt;                            // :428-432
    const float s0 = fmaxf(fminf(__expf(rs.x), max_scale), min_scale);                    // :461-465
    const float s1 = fmaxf(fminf(__expf(rs.y), max_scale), min_scale);
    float q[4];
    matrix_to_unit_quaternion(Gf.R, q);
    st3(points, n, p);
    st3(scaling, n, v3(thickness, s0, s1));                                               // :472-475
    reinterpret_cast<float4*>(quats)[n] = make_float4(q[0], q[1], q[2], q[3]);
}

#ifndef GSR_MESH_BWD_BLOCK
#define GSR_MESH_BWD_BLOCK 256
#endif
constexpr int MBB = GSR_MESH_BWD_BLOCK;   // threads per workgroup of the backward: MBB / 8 faces share one LDS vertex table
__global__ void __launch_bounds__(MBB)
mesh_gaussians_bwd8_kernel(int F, int G, const float* __restrict__ verts, const long long* __restrict__ faces,
                           const float* __restrict__ bary, const float* __restrict__ raw_scales,
                           const float* __restrict__ raw_complex, float min_scale, float max_scale,
                           const float* __restrict__ delta_r, const float* __restrict__ dL_dpoints,
                           const float* __restrict__ dL_dscaling, const float* __restrict__ dL_dquats,
                           float* __restrict__ dL_dverts, float* __restrict__ dL_draw_scales,
                           float* __restrict__ dL_draw_complex, float* __restrict__ dL_ddelta_t,
                           float* __restrict__ dL_ddelta_r)
{
    // Vertex gradients of the workgroup's 32 faces meet in a small LDS table (open addressing on the vertex index)
    // before they reach memory: neighbouring faces share vertices, and the 737 k scattered float atomics were 26 of this
    // kernel's 47 us.  (Keeping each XCD on a contiguous eighth of the faces instead changed nothing.)
    constexpr int VSLOTS = MBB / 2;   // 3 vertex references per face, 8 lanes per face: load factor <= 0.75
    __shared__ uint32_t v_key[VSLOTS];
    __shared__ float v_acc[VSLOTS][3];
    for (int i = threadIdx.x; i < VSLOTS; i += blockDim.x) { v_key[i] = 0xffffffffu; v_acc[i][0] = 0.f; v_acc[i][1] = 0.f; v_acc[i][2] = 0.f; }
    __syncthreads();
    const int vb = (int)blockIdx.x;
    const int t = vb * blockDim.x + threadIdx.x;
    const int f_raw = t / LPF, g = t - f_raw * LPF;
    const bool face_ok = f_raw < F;                 // whole groups of eight lanes share a face: the exchanges below stay in it
    const int f = face_ok ? f_raw : F - 1;
    const bool active = face_ok && g < G;
    // all of the lane's own inputs are requested first -- they do not depend on the face -- then the face frame (indices ->
    // vertices: two dependent trips that now overlap them); stores last
    const size_t n = active ? (size_t)f * G + g : 0;
    const V3 gm = dL_dpoints ? ld3(dL_dpoints, n) : v3(0, 0, 0);
    const int gb = active ? g : 0;
    const float b0 = bary[3 * gb], b1 = bary[3 * gb + 1], b2 = bary[3 * gb + 2];
    const float2 rs = reinterpret_cast<const float2*>(raw_scales)[n];
    const float gs[2] = {dL_dscaling ? dL_dscaling[3 * n + 1] : 0.f, dL_dscaling ? dL_dscaling[3 * n + 2] : 0.f};
    const float4 gq = dL_dquats ? reinterpret_cast<const float4*>(dL_dquats)[n] : make_float4(0.f, 0.f, 0.f, 0.f);
    const FaceFrame Ff = face_frame(verts, faces, (size_t)f);
    V3 gv0 = v3(0, 0, 0), gv1 = gv0, gv2 = gv0, gR0 = gv0, gbR1 = gv0, gbR2 = gv0;
    if (active) {
        const GaussFrame Gf = gauss_frame(Ff, raw_complex, delta_r, n);
        gv0 = b0 * gm; gv1 = b1 * gm; gv2 = b2 * gm;
        if (dL_ddelta_t) st3(dL_ddelta_t, n, gm);
        {   // scales: exp, then clamp_max, clamp_min masks (x <= max, y >= min)
            const float e0 = __expf(rs.x), e1 = __expf(rs.y);
            const bool p0 = e0 <= max_scale && fminf(e0, max_scale) >= min_scale, p1 = e1 <= max_scale && fminf(e1, max_scale) >= min_scale;
            reinterpret_cast<float2*>(dL_draw_scales)[n] = make_float2(p0 ? gs[0] * e0 : 0.f, p1 ? gs[1] * e1 : 0.f);
        }
        float q[4];
        matrix_to_unit_quaternion(Gf.R, q);
        const V3 qv = v3(q[1], q[2], q[3]), gqv = v3(gq.y, gq.z, gq.w);
        const V3 Gw = 0.5f * (((-gq.x) * qv + q[0] * gqv) + cross(qv, gqv));           // dL/d(rotation vector)
        V3 A[3];
#pragma unroll
        for (int c = 0; c < 3; c++) A[c] = 0.5f * cross(Gw, v3(Gf.R[0][c], Gf.R[1][c], Gf.R[2][c]));   // dL/dR = 1/2 [G]x R
        V3 gB[3];
        if (Gf.loose) {
            const float hw = -2.0f * dot(Gw, Gf.dv);
            const V3 hv = 2.0f * (Gf.dw * Gw - cross(Gf.dv, Gw));
            const float il = 1.0f / Gf.ld;
            if (dL_ddelta_r) reinterpret_cast<float4*>(dL_ddelta_r)[n] = make_float4(hw * il, hv.x * il, hv.y * il, hv.z * il);
#pragma unroll
            for (int c = 0; c < 3; c++)   // dL/dB = D^T dL/dR
                gB[c] = v3(Gf.D[0][0] * A[c].x + Gf.D[1][0] * A[c].y + Gf.D[2][0] * A[c].z,
                           Gf.D[0][1] * A[c].x + Gf.D[1][1] * A[c].y + Gf.D[2][1] * A[c].z,
                           Gf.D[0][2] * A[c].x + Gf.D[1][2] * A[c].y + Gf.D[2][2] * A[c].z);
        } else {
#pragma unroll
            for (int c = 0; c < 3; c++) gB[c] = A[c];
        }
        gR0 = gB[0];
        const float g_qc = dot(gB[1], Ff.bR1) + dot(gB[2], Ff.bR2), g_qs = dot(gB[1], Ff.bR2) - dot(gB[2], Ff.bR1);
        gbR1 = Gf.qc * gB[1] - Gf.qs * gB[2];
        gbR2 = Gf.qs * gB[1] + Gf.qc * gB[2];
        const float inv = 1.0f / fmaxf(Gf.lq, 1e-12f);   // through the normalisation of the raw complex number (:493)
        const float d = g_qc * Gf.qc + g_qs * Gf.qs;
        const bool reg = Gf.lq > 1e-12f;
        dL_draw_complex[2 * n] = reg ? inv * (g_qc - d * Gf.qc) : inv * g_qc;
        dL_draw_complex[2 * n + 1] = reg ? inv * (g_qs - d * Gf.qs) : inv * g_qs;
    }
    // sum over the face's lanes (all 64 lanes take part in the exchanges), then face frame -> vertices on lane 0
    gv0 = face_sum(gv0); gv1 = face_sum(gv1); gv2 = face_sum(gv2);
    gR0 = face_sum(gR0); gbR1 = face_sum(gbR1); gbR2 = face_sum(gbR2);
    if (face_ok && g == 0) {
        const V3 gcr = normalize_bwd(Ff.bR2, Ff.lc, 1e-12f, gbR2);       // bR2 = normalize(R0 x bR1)
        gR0 = gR0 + cross(Ff.bR1, gcr);
        gbR1 = gbR1 + cross(gcr, Ff.R0);
        const V3 ga = normalize_bwd(Ff.bR1, Ff.la, 1e-12f, gbR1);        // bR1 = normalize(v0 - v1)
        gv0 = gv0 + ga; gv1 = gv1 - ga;
        const V3 gn = normalize_bwd(Ff.R0, Ff.len, 1e-6f, gR0);          // R0 = normalize((e1 x e2) / max(|.|, 1e-6))
        const V3 ge1 = cross(Ff.e2, gn), ge2 = cross(gn, Ff.e1);
        gv1 = gv1 + ge1; gv2 = gv2 + ge2; gv0 = gv0 - (ge1 + ge2);
        const uint32_t vi[3] = {(uint32_t)faces[3 * (size_t)f], (uint32_t)faces[3 * (size_t)f + 1], (uint32_t)faces[3 * (size_t)f + 2]};
        const V3 gv[3] = {gv0, gv1, gv2};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            uint32_t h = ((vi[c] * 2654435761u) >> 16) & (VSLOTS - 1);
            bool placed = false;
            for (int probe = 0; probe < 8 && !placed; probe++) {
                const uint32_t prev = atomicCAS(&v_key[h], 0xffffffffu, vi[c]);
                if (prev == 0xffffffffu || prev == vi[c]) {
                    atomicAdd(&v_acc[h][0], gv[c].x); atomicAdd(&v_acc[h][1], gv[c].y); atomicAdd(&v_acc[h][2], gv[c].z);
                    placed = true;
                } else {
                    h = (h + 1) & (VSLOTS - 1);
                }
            }
            if (!placed) {
                atomicAdd(dL_dverts + 3 * (size_t)vi[c], gv[c].x); atomicAdd(dL_dverts + 3 * (size_t)vi[c] + 1, gv[c].y);
                atomicAdd(dL_dverts + 3 * (size_t)vi[c] + 2, gv[c].z);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < VSLOTS * 3; i += blockDim.x) {
        const int sl = i / 3, c = i - sl * 3;
        if (v_key[sl] != 0xffffffffu) atomicAdd(dL_dverts + 3 * (size_t)v_key[sl] + c, v_acc[sl][c]);
    }
}

void launch_mesh_gaussians(int F, int G, const float* verts, const long long* faces, const float* bary,
                           const float* raw_scales, const float* raw_complex, float thickness, float min_scale,
                           float max_scale, const float* delta_t, const float* delta_r, float* points, float* scaling,
                           float* quats, float* clear, long long clear_n, hipStream_t st)
{
    if (G <= LPF) {
        const long long lanes = (long long)F * LPF;
        mesh_gaussians_fwd8_kernel<<<(unsigned)((lanes + 255) / 256), 256, 0, st>>>(F, G, verts, faces, bary, raw_scales, raw_complex,
                                                                                    thickness, min_scale, max_scale, delta_t, delta_r,
                                                                                    points, scaling, quats, clear, clear_n);
        return;
    }
    mesh_gaussians_fwd_kernel<<<(F + 127) / 128, 128, 0, st>>>(F, G, verts, faces, bary, raw_scales, raw_complex, thickness,
                                                              min_scale, max_scale, delta_t, delta_r, points, scaling, quats, clear,
                                                              clear_n);
}

void launch_mesh_gaussians_bwd(int F, int G, const float* verts, const long long* faces, const float* bary,
                               const float* raw_scales, const float* raw_complex, float min_scale, float max_scale,
                               const float* delta_r, const float* dL_dpoints, const float* dL_dscaling,
                               const float* dL_dquats, float* dL_dverts, float* dL_draw_scales, float* dL_draw_complex,
                               float* dL_ddelta_t, float* dL_ddelta_r, hipStream_t st)
{
    if (G <= LPF) {
        const long long lanes = (long long)F * LPF;
        mesh_gaussians_bwd8_kernel<<<(unsigned)((lanes + MBB - 1) / MBB), MBB, 0, st>>>(F, G, verts, faces, bary, raw_scales, raw_complex,
                                                                                    min_scale, max_scale, delta_r, dL_dpoints, dL_dscaling,
                                                                                    dL_dquats, dL_dverts, dL_draw_scales, dL_draw_complex,
                                                                                    dL_ddelta_t, dL_ddelta_r);
        return;
    }
    mesh_gaussians_bwd_kernel<<<(F + 127) / 128, 128, 0, st>>>(F, G, verts, faces, bary, raw_scales, raw_complex, min_scale,
                                                              max_scale, delta_r, dL_dpoints, dL_dscaling, dL_dquats, dL_dverts,
                                                              dL_draw_scales, dL_draw_complex, dL_ddelta_t, dL_ddelta_r);
}

// Zero fill of a small array (the vertex-gradient accumulator in front of the mesh producer's backward): hipMemsetAsync of
// 0.5 MB goes out as TWO runtime kernels (aligned body + remainder), 10 us on the stream for what one launch does in 2.
__global__ void __launch_bounds__(256) zero_f32_kernel(float* __restrict__ p, size_t n)
{
    const size_t n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t t = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n) p[t] = 0.f;
}

void launch_zero_f32(float* p, size_t n, hipStream_t st)
{
    if (n == 0) return;
    if (((uintptr_t)p & 15u) != 0) { (void)hipMemsetAsync(p, 0, n * sizeof(float), st); return; }
    size_t blocks = ((n >> 2) + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    zero_f32_kernel<<<(unsigned)blocks, 256, 0, st>>>(p, n);
}

// view == nullptr: rgb [P,3]; otherwise rgb + depth-as-colour [P, 3 + depth_channels] (gsr_sh_to_rgbd)
// rows of 25 coefficients (degree 4) stage 77 KB per workgroup: above the 64 KB a kernel may take without asking
static void sh_lds_limit(const void* fn, size_t bytes)
{
    if (bytes > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

void launch_sh_to_rgb(int P, int D, int M, const float* positions, const float* campos, const float* shs, const float* shs_rest,
                      const float* view, int depth_channels, float* out, const float* densities, float* opacity, hipStream_t st)
{
    sh_lds_limit(reinterpret_cast<const void*>(&sh_to_rgb_kernel), sh_stage_bytes(M, 4));
    sh_to_rgb_kernel<<<(P + 255) / 256, 256, sh_stage_bytes(M, 4), st>>>(P, D, M, positions, campos, shs, shs_rest, view, out,
                                                                          view ? 3 + depth_channels : 3, densities, opacity);
}

void launch_sh_to_rgb_bwd(int P, int D, int M, const float* positions, const float* campos, const float* shs,
                          const float* shs_rest, const float* view, int depth_channels, const float* dL_dout, float* dL_dsh,
                          float* dL_dsh_rest, float* dL_dpos, int accumulate_pos, const float* opacity, const float* dL_dopacity,
                          float* dL_ddensity, hipStream_t st)
{
    sh_lds_limit(reinterpret_cast<const void*>(&sh_to_rgb_bwd_kernel), sh_stage_bytes(M, 4));
    sh_to_rgb_bwd_kernel<<<(P + 255) / 256, 256, sh_stage_bytes(M, 4), st>>>(
        P, D, M, positions, campos, shs, shs_rest, view, dL_dout, view ? 3 + depth_channels : 3, dL_dsh, dL_dsh_rest, dL_dpos,
        accumulate_pos, opacity, dL_dopacity, dL_ddensity);
}

}  // namespace gsr
