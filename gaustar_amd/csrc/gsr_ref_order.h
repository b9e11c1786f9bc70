// gsr_ref_order.h -- the per-Gaussian projection (projected centre, 3D and 2D covariance, conic, radius) in the OPERATION ORDER of
// the reference build.
//
// Why this file exists.  Whether a (pixel, Gaussian) pair is blended is decided by hard thresholds on alpha and T
// (DGR/cuda_rasterizer/forward.cu:336-351, backward.cu:496-502), and alpha is a function of the Gaussian's projected centre and
// conic.  Round 6 measured where the ~150 elements per 1080p view come from that sit outside the 1e-4 tolerance against the
// reference's own kernels (oracle/rig_parity.py::classify_flips, profiles/r06_parity_report.txt): 94 % are pairs that the two
// rasterizers decide differently because the per-Gaussian STATE differs in its last bits (80 % of the conics, 2 % of the centres:
// a centre that is one ulp off at x ~ 1000 px moves alpha by 3e-4 relative), not because alpha is evaluated in the exp2 domain.
// IEEE operations are deterministic: the same tree of operations on the same inputs gives the same bits.  So this file restates
// preprocessCUDA's arithmetic (forward.cu:74-152, :189-232; auxiliary.h:41-77) with EXACTLY the tree the reference's source
// compiles to under hipcc -O3 for gfx950 -- which product is fused into which sum, which quotient is a true division, the
// products with glm's structural zeros included (0 * x is not dropped without fast-math, and it decides the sign of a zero) --
// one IEEE operation per line, contraction off.  The tree was read off the reference build's assembly by tools/symfp.py
// (symbolic execution of the kernel's floating-point dataflow); tools/check_op_order.py runs the same tool over THIS library's
// preprocess kernel and compares the trees, so the restatement is checked without a GPU (tests/test_op_order.py pins it).
// Nothing here is smarter than the straightforward formulas of gsr_internal.h (which the backward keeps using: gradients are
// continuous in these values) -- it is the same mathematics in one particular rounding order.
//
// Do not simplify, reorder or "clean up" a line: every line is one rounding.
#pragma once
#include "gsr_internal.h"

namespace gsr {

// computeCov3D (forward.cu:118-152): Sigma = (S R)^T (S R) through glm's 3x3 products, upper triangle {xx, xy, xz, yy, yz, zz}.
// q = (r, x, y, z) raw (forward.cu:127), s = scale, mod = scale_modifier.
__device__ __forceinline__ void cov3d_ref_order(const Vec3 s, float mod, const float4 q, float c[6])
{
#pragma clang fp contract(off)
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    const float m0 = r * y;
    const float m1 = __builtin_fmaf(x, z, m0);
    const float m2 = 2.0f * m1;
    const float sz = mod * s.z;
    const float m4 = r * z;
    const float m5 = __builtin_fmaf(x, y, -m4);
    const float m6 = 2.0f * m5;
    const float m7 = y * y;
    const float m8 = z * z;
    const float m9 = m7 + m8;
    const float m10 = m9 + m9;
    const float m11 = 1.0f - m10;
    const float m12 = m11 * 0.0f;
    const float m13 = __builtin_fmaf(0.0f, m6, m12);
    const float m14 = __builtin_fmaf(m2, sz, m13);
    const float sx = mod * s.x;
    const float m16 = 0.0f * m6;
    const float m17 = __builtin_fmaf(m11, sx, m16);
    const float m18 = 0.0f * m2;
    const float m19 = m17 + m18;
    const float sy = mod * s.y;
    const float m21 = __builtin_fmaf(m6, sy, m12);
    const float m22 = m21 + m18;
    const float m23 = m22 * m22;
    const float m24 = __builtin_fmaf(m19, m19, m23);
    const float m25 = __builtin_fmaf(m14, m14, m24);
    const float m26 = r * x;
    const float m27 = __builtin_fmaf(y, z, -m26);
    const float m28 = m27 + m27;
    const float m29 = __builtin_fmaf(x, y, m4);
    const float m30 = m29 + m29;
    const float m31 = __builtin_fmaf(x, x, m8);
    const float m32 = m31 + m31;
    const float m33 = 1.0f - m32;
    const float m34 = m33 * 0.0f;
    const float m35 = __builtin_fmaf(m30, 0.0f, m34);
    const float m36 = __builtin_fmaf(m28, sz, m35);
    const float m37 = m33 * sy;
    const float m38 = __builtin_fmaf(m30, 0.0f, m37);
    const float m39 = __builtin_fmaf(m28, 0.0f, m38);
    const float m40 = x * y;
    const float m41 = __builtin_fmaf(r, z, m40);
    const float m42 = m41 + m41;
    const float m43 = __builtin_fmaf(m42, sx, m34);
    const float m44 = __builtin_fmaf(m28, 0.0f, m43);
    const float m45 = m19 * m44;
    const float m46 = __builtin_fmaf(m22, m39, m45);
    const float m47 = __builtin_fmaf(m36, m14, m46);
    const float m48 = __builtin_fmaf(x, x, m7);
    const float m49 = m48 + m48;
    const float m50 = 1.0f - m49;
    const float m51 = __builtin_fmaf(x, z, -m0);
    const float m52 = m51 + m51;
    const float m53 = __builtin_fmaf(y, z, m26);
    const float m54 = 2.0f * m53;
    const float m55 = 0.0f * m54;
    const float m56 = __builtin_fmaf(m52, 0.0f, m55);
    const float m57 = __builtin_fmaf(m50, sz, m56);
    const float m58 = __builtin_fmaf(m52, sx, m55);
    const float m59 = m50 * 0.0f;
    const float m60 = m58 + m59;
    const float m61 = m53 + m53;
    const float m62 = m52 * 0.0f;
    const float m63 = __builtin_fmaf(m61, sy, m62);
    const float m64 = m63 + m59;
    const float m65 = m64 * m22;
    const float m66 = __builtin_fmaf(m19, m60, m65);
    const float m67 = __builtin_fmaf(m57, m14, m66);
    const float m68 = m39 * m39;
    const float m69 = __builtin_fmaf(m44, m44, m68);
    const float m70 = __builtin_fmaf(m36, m36, m69);
    const float m71 = m64 * m39;
    const float m72 = __builtin_fmaf(m60, m44, m71);
    const float m73 = __builtin_fmaf(m57, m36, m72);
    const float m74 = __builtin_fmaf(m50, 0.0f, m63);
    const float m75 = m52 * sx;
    const float m76 = __builtin_fmaf(0.0f, m54, m75);
    const float m77 = __builtin_fmaf(m50, 0.0f, m76);
    const float m78 = m60 * m77;
    const float m79 = __builtin_fmaf(m64, m74, m78);
    const float m80 = __builtin_fmaf(m57, m57, m79);
    c[0] = m25;
    c[1] = m47;
    c[2] = m67;
    c[3] = m70;
    c[4] = m73;
    c[5] = m80;
}

struct Projected {
    float px, py;                    // pixel centre (points_xy_image)
    float cov_a, cov_b, cov_c, det;  // 2D covariance incl. the 0.3 low-pass, and its determinant
    float conic_a, conic_b, conic_c; // inverse (only meaningful when det != 0)
    float tz;                        // view-space z as computeCov2D forms it (NOT the bits of the depth key: view_depth())
};

// transformPoint4x4 + ndc2Pix (forward.cu:196-199, :233; auxiliary.h:41-44) and computeCov2D + the inversion (forward.cu:74-113, :214-222).
// c = the six numbers of the 3D covariance (computed above or handed in by the caller: cov3D_precomp).
__device__ __forceinline__ Projected project_ref_order(const Vec3 p, const float c[6], const float* __restrict__ view,
                                                       const float* __restrict__ proj, float fx, float fy, float tan_fovx,
                                                       float tan_fovy, int W, int H)
{
#pragma clang fp contract(off)
    const float u0 = p.y * proj[4];
    const float u1 = __builtin_fmaf(p.x, proj[0], u0);
    const float u2 = __builtin_fmaf(p.z, proj[8], u1);
    const float hom_x = u2 + proj[12];
    const float u4 = p.x * proj[3];
    const float u5 = __builtin_fmaf(p.y, proj[7], u4);
    const float u6 = p.z * proj[11];
    const float u7 = u5 + u6;
    const float u8 = u7 + proj[15];
    const float hom_w = u8 + 0.0000001f;
    const float p_w = 1.0f / hom_w;
    const float ndc_x = hom_x * p_w;
    const double u12 = (double)ndc_x;
    const double u13 = 1.0 + u12;
    const double u14 = (double)W;
    const double u15 = __builtin_fma(u13, u14, -1.0);
    const double u16 = 0.5 * u15;
    const float u17 = (float)u16;
    const float u18 = p.y * proj[5];
    const float u19 = __builtin_fmaf(p.x, proj[1], u18);
    const float u20 = __builtin_fmaf(p.z, proj[9], u19);
    const float hom_y = u20 + proj[13];
    const float ndc_y = hom_y * p_w;
    const double u23 = (double)ndc_y;
    const double u24 = 1.0 + u23;
    const double u25 = (double)H;
    const double u26 = __builtin_fma(u24, u25, -1.0);
    const double u27 = 0.5 * u26;
    const float u28 = (float)u27;
    const float u29 = p.x * view[1];
    const float u30 = __builtin_fmaf(p.y, view[5], u29);
    const float u31 = p.z * view[9];
    const float u32 = u30 + u31;
    const float ty = u32 + view[13];
    const float u34 = p.x * view[2];
    const float u35 = __builtin_fmaf(p.y, view[6], u34);
    const float u36 = p.z * view[10];
    const float u37 = u35 + u36;
    const float tz = u37 + view[14];
    const float tytz = ty / tz;
    const float limy = 1.3f * tan_fovy;
    const float u41 = fmaxf(tytz, -limy);
    const float u42 = fminf(u41, limy);
    const float u43 = u42 * -tz;
    const float u44 = fy * u43;
    const float tz2 = tz * tz;
    const float J12 = u44 / tz2;
    const float fy_tz = fy / tz;
    const float u48 = fy_tz * view[9];
    const float u49 = __builtin_fmaf(0.0f, view[8], u48);
    const float Ty2 = __builtin_fmaf(J12, view[10], u49);
    const float u51 = fy_tz * view[1];
    const float u52 = __builtin_fmaf(0.0f, view[0], u51);
    const float Ty0 = __builtin_fmaf(J12, view[2], u52);
    const float u54 = fy_tz * view[5];
    const float u55 = __builtin_fmaf(0.0f, view[4], u54);
    const float Ty1 = __builtin_fmaf(J12, view[6], u55);
    const float u57 = Ty1 * c[4];
    const float u58 = __builtin_fmaf(Ty0, c[2], u57);
    const float Vy2 = __builtin_fmaf(Ty2, c[5], u58);
    const float u60 = Ty0 * c[0];
    const float u61 = __builtin_fmaf(Ty1, c[1], u60);
    const float Vy0 = __builtin_fmaf(Ty2, c[2], u61);
    const float u63 = Ty0 * c[1];
    const float u64 = __builtin_fmaf(Ty1, c[3], u63);
    const float Vy1 = __builtin_fmaf(Ty2, c[4], u64);
    const float u66 = Ty1 * Vy1;
    const float u67 = __builtin_fmaf(Ty0, Vy0, u66);
    const float u68 = __builtin_fmaf(Ty2, Vy2, u67);
    const float cov_c = 0.3f + u68;
    const float fx_tz = fx / tz;
    const float u71 = 0.0f * view[9];
    const float u72 = __builtin_fmaf(fx_tz, view[8], u71);
    const float u73 = p.x * view[0];
    const float u74 = __builtin_fmaf(p.y, view[4], u73);
    const float u75 = p.z * view[8];
    const float u76 = u74 + u75;
    const float tx = u76 + view[12];
    const float txtz = tx / tz;
    const float limx = 1.3f * tan_fovx;
    const float u80 = fmaxf(txtz, -limx);
    const float u81 = fminf(u80, limx);
    const float u82 = u81 * -tz;
    const float u83 = fx * u82;
    const float J02 = u83 / tz2;
    const float u85 = J02 * view[10];
    const float Tx2 = u72 + u85;
    const float u87 = 0.0f * view[1];
    const float u88 = fx_tz * view[0];
    const float u89 = u87 + u88;
    const float Tx0 = __builtin_fmaf(J02, view[2], u89);
    const float u91 = 0.0f * view[5];
    const float u92 = fx_tz * view[4];
    const float u93 = u91 + u92;
    const float Tx1 = __builtin_fmaf(J02, view[6], u93);
    const float u95 = Tx1 * Vy1;
    const float u96 = __builtin_fmaf(Tx0, Vy0, u95);
    const float cov_b = __builtin_fmaf(Tx2, Vy2, u96);
    const float u98 = Tx1 * c[4];
    const float u99 = __builtin_fmaf(Tx0, c[2], u98);
    const float Vx2 = __builtin_fmaf(Tx2, c[5], u99);
    const float u101 = Tx1 * c[1];
    const float u102 = __builtin_fmaf(Tx0, c[0], u101);
    const float Vx0 = __builtin_fmaf(Tx2, c[2], u102);
    const float u104 = Tx1 * c[3];
    const float u105 = __builtin_fmaf(Tx0, c[1], u104);
    const float Vx1 = __builtin_fmaf(Tx2, c[4], u105);
    const float u107 = Vx1 * Tx1;
    const float u108 = __builtin_fmaf(Vx0, Tx0, u107);
    const float u109 = __builtin_fmaf(Tx2, Vx2, u108);
    const float cov_a = 0.3f + u109;
    const float u111 = cov_a * cov_c;
    const float det = __builtin_fmaf(cov_b, -cov_b, u111);
    const float det_inv = 1.0f / det;
    const float u114 = cov_c * det_inv;
    const float u115 = det_inv * -cov_b;
    const float u116 = cov_a * det_inv;
    Projected o;
    o.px = u17; o.py = u28;
    o.cov_a = cov_a; o.cov_b = cov_b; o.cov_c = cov_c; o.det = det;
    o.conic_a = u114; o.conic_b = u115; o.conic_c = u116;
    o.tz = tz;
    return o;
}

// Extent (forward.cu:224-232): radius = ceil(3 sqrt(larger eigenvalue)), eigenvalues from mid +- sqrt(max(0.1, mid^2 - det)).
__device__ __forceinline__ float radius_ref_order(float cov_a, float cov_c, float det)
{
#pragma clang fp contract(off)
    const float sum = cov_a + cov_c;
    const float mid = sum * 0.5f;
    const float s = sqrtf(fmaxf(0.1f, __builtin_fmaf(mid, mid, -det)));
    const float l2 = __builtin_fmaf(sum, 0.5f, -s);
    const float l1 = __builtin_fmaf(sum, 0.5f, s);
    return ceilf(3.0f * sqrtf(fmaxf(l2, l1)));
}

}  // namespace gsr
