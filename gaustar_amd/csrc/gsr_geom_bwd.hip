// gsr_geom_bwd.hip -- per-Gaussian backward, ONE kernel.
//
// Fuses what the reference runs as computeCov2DCUDA (DGR/cuda_rasterizer/backward.cu:144-274) and
// preprocessCUDA-backward (:346-396, with computeColorFromSH bwd :20-139 and computeCov3D bwd
// :278-341), plus the zero-fill of every gradient tensor that only this stage writes
// (DGR/rasterize_points.cu:151-159): each thread writes its Gaussian's dL_dmean3D / dL_dcov3D /
// dL_dscale / dL_drot / dL_dsh outright -- zeros when the Gaussian was culled (radii == 0).
//
// Input from the blend backward is the packed moment record grad_acc[P][12] (see gsr_blend_bwd.hip); this
// kernel applies the per-Gaussian linear maps that turn the moments into dL_dmean2D (pixel->NDC scale
// 0.5*W / 0.5*H, backward.cu:460-461), dL_dconic (the -0.5 factors, backward.cu:549-551), dL_dopacity and
// dL_dcolor, and writes those outputs too.  From the forward's geometry state only the conic/opacity
// record is read; the 3D covariance, the EWA rows and the SH basis are recomputed from the inputs (the
// reference re-reads cov3D and the `clamped` flags it stored, 27 B/Gaussian of state traffic each way).
#include "gsr_internal.h"

namespace gsr {

__device__ __forceinline__ void
geom_bwd_body(int idx, int P, int D, int M, int C, const float* __restrict__ means3D, float* __restrict__ my_sh,
                const float* __restrict__ scales, float scale_modifier, const float* __restrict__ rotations,
                const float* __restrict__ cov3D_precomp, const float* __restrict__ view,
                const float* __restrict__ proj, const float* __restrict__ campos, float tan_fovx, float tan_fovy,
                float focal_x, float focal_y, float half_w, float half_h, const int* __restrict__ radii,
                const float4* __restrict__ g0, const float4* __restrict__ g1, const float4* __restrict__ grad_acc,
                float* __restrict__ dL_dmean2D, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor,
                float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dscale,
              float* __restrict__ dL_drot)
{
    if (idx >= P) return;
    const size_t i = (size_t)idx;

    if (!(radii[idx] > 0)) {
        dL_dmean2D[3 * i] = 0.f; dL_dmean2D[3 * i + 1] = 0.f; dL_dmean2D[3 * i + 2] = 0.f;
        for (int ch = 0; ch < C; ch++) dL_dcolor[(size_t)C * i + ch] = 0.f;
        dL_dopacity[i] = 0.f;
        dL_dmean3D[3 * i] = 0.f; dL_dmean3D[3 * i + 1] = 0.f; dL_dmean3D[3 * i + 2] = 0.f;
        if (dL_dcov3D) {
#pragma unroll
            for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = 0.f;
        }
        if (dL_dscale) { dL_dscale[3 * i] = 0.f; dL_dscale[3 * i + 1] = 0.f; dL_dscale[3 * i + 2] = 0.f; }
        if (dL_drot) reinterpret_cast<float4*>(dL_drot)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (my_sh)
            for (int k = 0; k < 3 * M; k++) my_sh[k] = 0.f;
        return;
    }

    const Vec3 mean = load3(means3D, i);

    // ---------------- moments -> dL_dcolor, dL_dopacity, dL_dmean2D, dL_dconic
    static_assert(GRAD_RS == 12, "three float4 per record");
    const float4 f0 = grad_acc[3 * i], f1 = grad_acc[3 * i + 1], f2 = grad_acc[3 * i + 2];
    // record = {sum r, sum r dx, sum r dy, sum r dx^2 | sum r dx dy, sum r dy^2, c0, c1 | c2, c3, c4, c5},
    // c_k = sum w*dL_dpix_k (gsr_blend_bwd.hip)
    const float s_r = f0.x, s_x = f0.y, s_y = f0.z, s_xx = f0.w, s_xy = f1.x, s_yy = f1.y;
    const float cm[6] = {f1.z, f1.w, f2.x, f2.y, f2.z, f2.w};
    const float4 ga = g0[i], gb = g1[i];
    const float con_a = ga.z, con_b = ga.w, con_c = gb.x, op = gb.y;
    if (C == 3) { dL_dcolor[3 * i] = cm[0]; dL_dcolor[3 * i + 1] = cm[1]; dL_dcolor[3 * i + 2] = cm[2]; }
    else {
#pragma unroll
        for (int ch = 0; ch < 6; ch++) if (ch < C) dL_dcolor[(size_t)C * i + ch] = cm[ch];
    }
    dL_dopacity[i] = s_r;
    const float gx2 = -op * (con_a * s_x + con_b * s_y) * half_w;
    const float gy2 = -op * (con_c * s_y + con_b * s_x) * half_h;
    dL_dmean2D[3 * i] = gx2; dL_dmean2D[3 * i + 1] = gy2; dL_dmean2D[3 * i + 2] = 0.f;
    const float4 dcon = make_float4(-0.5f * op * s_xx, -0.5f * op * s_xy, 0.f, -0.5f * op * s_yy);   // (xx, xy, -, yy)

    // ---------------- conic -> cov2D -> {cov3D, view-space mean}  (backward.cu:144-274)
    float c3[6];
    float4 quat = make_float4(1.f, 0.f, 0.f, 0.f);
    Vec3 scl{1.f, 1.f, 1.f};
    if (cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) c3[k] = cov3D_precomp[6 * i + k];
    } else {
        quat = reinterpret_cast<const float4*>(rotations)[i];
        scl = load3(scales, i);
        cov3d_from_scale_rot(scl, scale_modifier, quat, c3);
    }
    const Ewa e = ewa_rows(mean, view, focal_x, focal_y, tan_fovx, tan_fovy);
    float v0[3], v1[3], a, b, c;
    cov2d_from(e, c3, v0, v1, a, b, c);

    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    const float x_grad_mul = (e.txtz < -limx || e.txtz > limx) ? 0.f : 1.f;
    const float y_grad_mul = (e.tytz < -limy || e.tytz > limy) ? 0.f : 1.f;

    const float denom = a * c - b * b;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (denom2inv != 0.f) {
        dL_da = denom2inv * (-c * c * dcon.x + 2 * b * c * dcon.y + (denom - a * c) * dcon.w);
        dL_dc = denom2inv * (-a * a * dcon.w + 2 * a * b * dcon.y + (denom - a * c) * dcon.x);
        dL_db = denom2inv * 2 * (b * c * dcon.x - (denom + 2 * b * b) * dcon.y + a * b * dcon.w);
        const float* a0 = e.a0; const float* a1 = e.a1;
        dcov[0] = a0[0] * a0[0] * dL_da + a0[0] * a1[0] * dL_db + a1[0] * a1[0] * dL_dc;
        dcov[3] = a0[1] * a0[1] * dL_da + a0[1] * a1[1] * dL_db + a1[1] * a1[1] * dL_dc;
        dcov[5] = a0[2] * a0[2] * dL_da + a0[2] * a1[2] * dL_db + a1[2] * a1[2] * dL_dc;
        dcov[1] = 2 * a0[0] * a0[1] * dL_da + (a0[0] * a1[1] + a0[1] * a1[0]) * dL_db + 2 * a1[0] * a1[1] * dL_dc;
        dcov[2] = 2 * a0[0] * a0[2] * dL_da + (a0[0] * a1[2] + a0[2] * a1[0]) * dL_db + 2 * a1[0] * a1[2] * dL_dc;
        dcov[4] = 2 * a0[2] * a0[1] * dL_da + (a0[1] * a1[2] + a0[2] * a1[1]) * dL_db + 2 * a1[1] * a1[2] * dL_dc;
    }
    if (dL_dcov3D) {   // (NULL when the covariances come from scales and rotations: nobody reads this gradient then)
#pragma unroll
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = dcov[k];
    }

    float dT0[3], dT1[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        dT0[k] = 2 * v0[k] * dL_da + v1[k] * dL_db;
        dT1[k] = 2 * v1[k] * dL_dc + v0[k] * dL_db;
    }
    const float dL_dJ00 = view[0] * dT0[0] + view[4] * dT0[1] + view[8] * dT0[2];
    const float dL_dJ02 = view[2] * dT0[0] + view[6] * dT0[1] + view[10] * dT0[2];
    const float dL_dJ11 = view[1] * dT1[0] + view[5] * dT1[1] + view[9] * dT1[2];
    const float dL_dJ12 = view[2] * dT1[0] + view[6] * dT1[1] + view[10] * dT1[2];
    const float tz = 1.f / e.t.z, tz2 = tz * tz, tz3 = tz2 * tz;
    const float dL_dtx = x_grad_mul * -focal_x * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -focal_y * tz2 * dL_dJ12;
    const float dL_dtz = -focal_x * tz2 * dL_dJ00 - focal_y * tz2 * dL_dJ11 + (2 * focal_x * e.t.x) * tz3 * dL_dJ02 +
                         (2 * focal_y * e.t.y) * tz3 * dL_dJ12;
    // W^T * dL_dt  (auxiliary.h:89-97)
    float gmx = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz;
    float gmy = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz;
    float gmz = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;

    // ---------------- mean2D -> mean3D through the perspective divide (backward.cu:370-387)
    {
        const float m_w = 1.0f / (xform4w(mean, proj) + 0.0000001f);
        const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
        gmx += (proj[0] * m_w - proj[3] * mul1) * gx2 + (proj[1] * m_w - proj[3] * mul2) * gy2;
        gmy += (proj[4] * m_w - proj[7] * mul1) * gx2 + (proj[5] * m_w - proj[7] * mul2) * gy2;
        gmz += (proj[8] * m_w - proj[11] * mul1) * gx2 + (proj[9] * m_w - proj[11] * mul2) * gy2;
    }

    // ---------------- colour -> SH coefficients and view direction (backward.cu:20-139)
    if (my_sh) {
        const float dcol[3] = {cm[0], cm[1], cm[2]};
        sh_colour_backward(D, M, mean, campos, my_sh, dcol, my_sh, gmx, gmy, gmz);
    }
    dL_dmean3D[3 * i] = gmx; dL_dmean3D[3 * i + 1] = gmy; dL_dmean3D[3 * i + 2] = gmz;

    // ---------------- cov3D -> scale, raw quaternion (backward.cu:278-341)
    if (!cov3D_precomp) {
        float R[3][3];
        quat_R(quat, R);
        const float s[3] = {scale_modifier * scl.x, scale_modifier * scl.y, scale_modifier * scl.z};
        float Mm[3][3];
#pragma unroll
        for (int r_ = 0; r_ < 3; r_++)
#pragma unroll
            for (int c_ = 0; c_ < 3; c_++) Mm[r_][c_] = s[r_] * R[c_][r_];
        const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
        float dM[3][3];
#pragma unroll
        for (int r_ = 0; r_ < 3; r_++)
#pragma unroll
            for (int c_ = 0; c_ < 3; c_++)
                dM[r_][c_] = 2.0f * (Mm[r_][0] * dS[0][c_] + Mm[r_][1] * dS[1][c_] + Mm[r_][2] * dS[2][c_]);
        float ds[3];
#pragma unroll
        for (int k = 0; k < 3; k++) ds[k] = R[0][k] * dM[k][0] + R[1][k] * dM[k][1] + R[2][k] * dM[k][2];
        dL_dscale[3 * i] = ds[0]; dL_dscale[3 * i + 1] = ds[1]; dL_dscale[3 * i + 2] = ds[2];
        float G[3][3];
#pragma unroll
        for (int r_ = 0; r_ < 3; r_++)
#pragma unroll
            for (int c_ = 0; c_ < 3; c_++) G[r_][c_] = s[r_] * dM[r_][c_];
        const float qr = quat.x, qx = quat.y, qy = quat.z, qz = quat.w;
        float4 dq;
        dq.x = 2 * qz * (G[0][1] - G[1][0]) + 2 * qy * (G[2][0] - G[0][2]) + 2 * qx * (G[1][2] - G[2][1]);
        dq.y = 2 * qy * (G[1][0] + G[0][1]) + 2 * qz * (G[2][0] + G[0][2]) + 2 * qr * (G[1][2] - G[2][1]) - 4 * qx * (G[2][2] + G[1][1]);
        dq.z = 2 * qx * (G[1][0] + G[0][1]) + 2 * qr * (G[2][0] - G[0][2]) + 2 * qz * (G[1][2] + G[2][1]) - 4 * qy * (G[2][2] + G[0][0]);
        dq.w = 2 * qr * (G[0][1] - G[1][0]) + 2 * qx * (G[2][0] + G[0][2]) + 2 * qy * (G[1][2] + G[2][1]) - 4 * qz * (G[1][1] + G[0][0]);
        reinterpret_cast<float4*>(dL_drot)[i] = dq;
    }
}


__global__ void __launch_bounds__(256)
geom_bwd_kernel(int P, int D, int M, int C, const float* __restrict__ means3D, const float* __restrict__ shs,
                const float* __restrict__ scales, float scale_modifier, const float* __restrict__ rotations,
                const float* __restrict__ cov3D_precomp, const float* __restrict__ view,
                const float* __restrict__ proj, const float* __restrict__ campos, float tan_fovx, float tan_fovy,
                float focal_x, float focal_y, float half_w, float half_h, const int* __restrict__ radii,
                const float4* __restrict__ g0, const float4* __restrict__ g1, const float4* __restrict__ grad_acc,
                float* __restrict__ dL_dmean2D, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor,
                float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
                float* __restrict__ dL_dscale, float* __restrict__ dL_drot)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    // SH mode: coefficient rows come in, and their gradients leave, through LDS (coalesced; see sh_stage_load).  The
    // gradient row overwrites the coefficient row in place, so no lane may leave before the wave's final store.
    extern __shared__ float sh_lds[];
    float* wave_rows = nullptr;
    float* my_sh = nullptr;
    if (shs) {
        wave_rows = sh_lds + (size_t)(threadIdx.x >> 6) * 64 * sh_row_stride(M);
        sh_stage_load(wave_rows, shs, (size_t)(idx - lane), P, M, lane);
        __builtin_amdgcn_wave_barrier();
        my_sh = wave_rows + lane * sh_row_stride(M);
    }
    geom_bwd_body(idx, P, D, M, C, means3D, my_sh, scales, scale_modifier, rotations, cov3D_precomp, view, proj, campos, tan_fovx,
                  tan_fovy, focal_x, focal_y, half_w, half_h, radii, g0, g1, grad_acc, dL_dmean2D, dL_dopacity, dL_dcolor,
                  dL_dmean3D, dL_dcov3D, dL_dscale, dL_drot);
    if (shs) {
        __builtin_amdgcn_wave_barrier();
        sh_stage_store(wave_rows, dL_dsh, (size_t)(idx - lane), P, M, lane);
    }
}

void launch_geom_bwd(int P, int D, int M, const float* means3D, const float* shs, const float* scales,
                     float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* view,
                     const float* proj, const float* campos, int W, int H, float tan_fovx, float tan_fovy,
                     const int* radii, GeomState g, int C, const float* grad_acc, float* dL_dmean2D, float* dL_dopacity,
                     float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                     float* dL_drot, hipStream_t st)
{
    const float focal_y = H / (2.0f * tan_fovy), focal_x = W / (2.0f * tan_fovx);   // rasterizer_impl.cu:381-382
    const size_t lds = shs ? sh_stage_bytes(M, 4) : 0;
    geom_bwd_kernel<<<(P + 255) / 256, 256, lds, st>>>(P, D, M, C, means3D, shs, scales, scale_modifier, rotations,
                                                     cov3D_precomp, view, proj, campos, tan_fovx, tan_fovy, focal_x,
                                                     focal_y, 0.5f * W, 0.5f * H, radii, g.g0, g.g1,
                                                     reinterpret_cast<const float4*>(grad_acc), dL_dmean2D,
                                                     dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
                                                     dL_drot);
}

}  // namespace gsr
