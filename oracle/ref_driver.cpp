// ref_driver.cpp -- C-ABI shim around the REFERENCE rasterizer, for oracle/_ref only.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; it contains no reference code.  It is
// compiled by oracle/build_ref.sh together with the reference's own
// cuda_rasterizer/{forward,backward,rasterizer_impl}.cu (taken where they lie under
// /root/reference and translated to HIP by the image's hipify-perl in a temp dir) into
// oracle/_ref/libgsr_ref.so.  It calls CudaRasterizer::Rasterizer::{forward,backward,
// markVisible} (DGR/cuda_rasterizer/rasterizer.h:24-84) with raw device pointers, the
// way DGR/rasterize_points.cu:35-196 does, and exposes the scratch-buffer contents
// (rasterizer_impl.h:31-63) so golden fixtures can hold the intermediates.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <functional>
#include "rasterizer_impl.h"  // hipified copy in the build dir (GeometryState/ImageState/BinningState)

namespace {
struct Buf {
    char* p = nullptr;
    size_t n = 0;
    char* resize(size_t N)
    {
        if (N > n) {
            if (p) (void)hipFree(p);
            if (hipMalloc((void**)&p, N) != hipSuccess) { p = nullptr; n = 0; return nullptr; }
            n = N;
        }
        return p;
    }
    ~Buf() { if (p) (void)hipFree(p); }
};
struct State {
    Buf geom, binning, img;
    int P = 0, R = 0, W = 0, H = 0;
};
}  // namespace

extern "C" {

void* ref_create() { return new State(); }
void ref_destroy(void* s) { delete static_cast<State*>(s); }

// Absent optionals are passed as nullptr (rasterize_points.cu hands data_ptr() of empty tensors).
int ref_forward(void* sp, int P, int D, int M, const float* bg, int W, int H, const float* means3D, const float* shs,
                const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                const float* rotations, const float* cov3D_precomp, const float* view, const float* proj,
                const float* campos, float tanfovx, float tanfovy, int prefiltered, float* out_color, int* radii)
{
    State* s = static_cast<State*>(sp);
    s->P = P; s->W = W; s->H = H; s->R = 0;
    if (P == 0) return 0;
    std::function<char*(size_t)> g = [s](size_t N) { return s->geom.resize(N); };
    std::function<char*(size_t)> b = [s](size_t N) { return s->binning.resize(N); };
    std::function<char*(size_t)> i = [s](size_t N) { return s->img.resize(N); };
    s->R = CudaRasterizer::Rasterizer::forward(g, b, i, P, D, M, bg, W, H, means3D, shs, colors_precomp, opacities,
                                               scales, scale_modifier, rotations, cov3D_precomp, view, proj, campos,
                                               tanfovx, tanfovy, prefiltered != 0, out_color, radii, false);
    (void)hipDeviceSynchronize();
    return s->R;
}

void ref_backward(void* sp, int D, int M, const float* bg, const float* means3D, const float* shs,
                  const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                  const float* cov3D_precomp, const float* view, const float* proj, const float* campos, float tanfovx,
                  float tanfovy, const int* radii, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                  float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                  float* dL_dscale, float* dL_drot)
{
    State* s = static_cast<State*>(sp);
    if (s->P == 0) return;
    CudaRasterizer::Rasterizer::backward(s->P, D, M, s->R, bg, s->W, s->H, means3D, shs, colors_precomp, scales,
                                         scale_modifier, rotations, cov3D_precomp, view, proj, campos, tanfovx, tanfovy,
                                         radii, s->geom.p, s->binning.p, s->img.p, dL_dpix, dL_dmean2D, dL_dconic,
                                         dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, false);
    (void)hipDeviceSynchronize();
}

void ref_mark_visible(int P, float* means3D, float* view, float* proj, bool* present)
{
    if (P) CudaRasterizer::Rasterizer::markVisible(P, means3D, view, proj, present);
    (void)hipDeviceSynchronize();
}

// Device pointers into the scratch buffers (valid until the next ref_forward on this state).
// which: 0 depths[P] f32, 1 clamped[3P] u8, 2 means2D[2P] f32, 3 cov3D[6P] f32, 4 conic_opacity[4P] f32,
//        5 rgb[3P] f32, 6 tiles_touched[P] u32, 7 point_offsets[P] u32, 8 final_T[WH] f32, 9 n_contrib[WH] u32,
//        10 ranges[2*T] u32, 11 point_list[R] u32, 12 point_list_keys[R] u64
void* ref_state_ptr(void* sp, int which)
{
    State* s = static_cast<State*>(sp);
    if (s->P == 0) return nullptr;
    char* gp = s->geom.p;
    CudaRasterizer::GeometryState geo = CudaRasterizer::GeometryState::fromChunk(gp, s->P);
    char* ip = s->img.p;
    CudaRasterizer::ImageState img = CudaRasterizer::ImageState::fromChunk(ip, (size_t)s->W * s->H);
    switch (which) {
        case 0: return geo.depths;
        case 1: return geo.clamped;
        case 2: return geo.means2D;
        case 3: return geo.cov3D;
        case 4: return geo.conic_opacity;
        case 5: return geo.rgb;
        case 6: return geo.tiles_touched;
        case 7: return geo.point_offsets;
        case 8: return img.accum_alpha;
        case 9: return img.n_contrib;
        case 10: return img.ranges;
        default: break;
    }
    if (s->R == 0) return nullptr;
    char* bp = s->binning.p;
    CudaRasterizer::BinningState bin = CudaRasterizer::BinningState::fromChunk(bp, s->R);
    if (which == 11) return bin.point_list;
    if (which == 12) return bin.point_list_keys;
    return nullptr;
}

}  // extern "C"
