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

namespace gsr {

__global__ void __launch_bounds__(256)
sh_to_rgb_kernel(int P, int D, int M, const float* __restrict__ positions, const float* __restrict__ campos,
                 const float* __restrict__ shs, float* __restrict__ rgb)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const size_t i = (size_t)idx;
    const Vec3 p = load3(positions, i);
    const float dx = p.x - campos[0], dy = p.y - campos[1], dz = p.z - campos[2];
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    float basis[16];
    sh_basis(D, dx * inv, dy * inv, dz * inv, basis);
    const int nb = (D + 1) * (D + 1);
    const float* sh = shs + i * M * 3;
    float cr = 0.f, cg = 0.f, cb = 0.f;
    for (int k = 0; k < nb; k++) { cr += basis[k] * sh[3 * k]; cg += basis[k] * sh[3 * k + 1]; cb += basis[k] * sh[3 * k + 2]; }
    rgb[3 * i] = fmaxf(cr + 0.5f, 0.0f);
    rgb[3 * i + 1] = fmaxf(cg + 0.5f, 0.0f);
    rgb[3 * i + 2] = fmaxf(cb + 0.5f, 0.0f);
}

__global__ void __launch_bounds__(256)
sh_to_rgb_bwd_kernel(int P, int D, int M, const float* __restrict__ positions, const float* __restrict__ campos,
                     const float* __restrict__ shs, const float* __restrict__ dL_drgb, float* __restrict__ dL_dsh,
                     float* __restrict__ dL_dpos)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const size_t i = (size_t)idx;
    const Vec3 p = load3(positions, i);
    const float dcol[3] = {dL_drgb[3 * i], dL_drgb[3 * i + 1], dL_drgb[3 * i + 2]};
    float gx = 0.f, gy = 0.f, gz = 0.f;
    sh_colour_backward(D, M, p, campos, shs + i * M * 3, dcol, dL_dsh + i * M * 3, gx, gy, gz);
    dL_dpos[3 * i] = gx; dL_dpos[3 * i + 1] = gy; dL_dpos[3 * i + 2] = gz;
}

void launch_sh_to_rgb(int P, int D, int M, const float* positions, const float* campos, const float* shs, float* rgb,
                      hipStream_t st)
{
    sh_to_rgb_kernel<<<(P + 255) / 256, 256, 0, st>>>(P, D, M, positions, campos, shs, rgb);
}

void launch_sh_to_rgb_bwd(int P, int D, int M, const float* positions, const float* campos, const float* shs,
                          const float* dL_drgb, float* dL_dsh, float* dL_dpos, hipStream_t st)
{
    sh_to_rgb_bwd_kernel<<<(P + 255) / 256, 256, 0, st>>>(P, D, M, positions, campos, shs, dL_drgb, dL_dsh, dL_dpos);
}

}  // namespace gsr
