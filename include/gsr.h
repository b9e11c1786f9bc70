/*
 * gsr.h -- C ABI of the MI355X-native differentiable surface-Gaussian rasterizer.
 *
 * This is the drop-in boundary below the reference's Python API
 * (DGR = gaussian_splatting/submodules/diff-gaussian-rasterization).  Each entry
 * point names the reference interface it replaces.  Plain pointers, sizes and a HIP
 * stream only: no torch types, no C++ types.  All pointers are DEVICE pointers unless
 * marked [host].  All arrays are fp32 row-major and contiguous; matrices are the
 * reference's transposed (column-major) 4x4s (DGR/cuda_rasterizer/auxiliary.h:58-77).
 * "Absent" optional inputs are NULL (the reference passes data_ptr() of a 0-element
 * tensor, DGR/rasterize_points.cu:94-111).
 *
 * Ownership (as DGR/rasterize_points.cu:68-78): the caller owns every buffer including
 * the three scratch buffers, and backward re-derives its view of the scratch from
 * (P, R, W, H) alone (DGR/cuda_rasterizer/rasterizer_impl.cu:371-373).  The library itself
 * holds: per calling thread, a 64-byte pinned landing pad for the stage-1 totals and the
 * last error message; process-wide, a pool of HIP events for gsr_profile_*; and -- only
 * once gsr_forward_fused has been used -- one block of tile counters (<= 1 MB) per
 * (device, stream), exclusive to one call at a time (a concurrent call on the same
 * stream falls back to counters in its own image buffer) and freed by
 * gsr_release_stream_state().  Nothing else persists between calls.
 *
 * Every function returns 0 on success, non-zero on failure; gsr_last_error() then
 * holds a message for the calling thread.
 */
#ifndef GSR_H_INCLUDED
#define GSR_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gsr_stream_t; /* a hipStream_t; NULL = the default stream */

/* Growable-buffer callback: must return a device pointer to at least `bytes` bytes that
 * stays valid until the matching backward has run.  Mirrors the three
 * std::function<char*(size_t)> of CudaRasterizer::Rasterizer::forward
 * (DGR/cuda_rasterizer/rasterizer.h:32-34; DGR/rasterize_points.cu:27-33). */
typedef void* (*gsr_alloc_fn)(void* ctx, size_t bytes);

/* ABI version of this header (bumped on any signature change). */
int gsr_abi_version(void);

/* Message of the last failure on this thread ("" if none). */
const char* gsr_last_error(void);

/* Scratch sizes in bytes.  Replace CudaRasterizer::required<GeometryState|ImageState|
 * BinningState>() (DGR/cuda_rasterizer/rasterizer_impl.h:66-72). */
size_t gsr_geom_bytes(int P);
size_t gsr_image_bytes(int W, int H);
size_t gsr_binning_bytes(int R, int num_segments);
/* Same for a render with num_channels colour channels (3, 4 or 6, see gsr_forward_stage2_mt). */
size_t gsr_binning_bytes_mt(int R, int num_segments, int num_channels);
/* Backward-only scratch: one packed 48-byte accumulation record per Gaussian.  Takes the place of the
 * dL_dconic [P,2,2] work tensor the reference binding allocates (DGR/rasterize_points.cu:154). */
size_t gsr_grad_scratch_bytes(int P);

/* Forward, first half: per-Gaussian projection/culling/covariance/SH->RGB, tile counting and
 * the tile-offset scan; reads back num_rendered (= Gaussian x tile instances this library will
 * blend) -- the one host synchronisation of the forward, like
 * DGR/cuda_rasterizer/rasterizer_impl.cu:281.  Replaces rasterizer_impl.cu:198-281
 * (FORWARD::preprocess, forward.cu:155-256, + InclusiveSum).
 *   radii [P] int32 out (same values as the reference's), geom/image scratch sized by
 *   gsr_geom_bytes / gsr_image_bytes.  *num_rendered [host] out; *max_tile_instances [host] out =
 *   the longest per-tile list (lets stage 2 pick its LDS sort capacity without a second sync);
 *   *num_segments [host] out = number of (tile, list segment) work units of the backward pass, which
 *   also sizes the per-segment snapshot area of the binning buffer. */
int gsr_forward_stage1(int P, int D, int M, const float* means3D, const float* shs, const float* colors_precomp,
                       const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                       const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                       const float* campos, int W, int H, float tan_fovx, float tan_fovy, int prefiltered,
                       int* radii, void* geom_buffer, void* image_buffer, int* num_rendered,
                       int* max_tile_instances, int* num_segments, gsr_stream_t stream);

/* Forward, second half: instance scatter into per-tile buckets, per-tile depth sort, alpha
 * blend.  Replaces rasterizer_impl.cu:283-335 (duplicateWithKeys, SortPairs,
 * identifyTileRanges, FORWARD::render = forward.cu:261-374).
 *   out_color [3,H,W] planar; binning scratch sized by gsr_binning_bytes(R, num_segments).
 *   FORWARD-ONLY renders (no gsr_backward will follow): pass -num_segments here (binning scratch still sized with
 *   +num_segments): the per-segment snapshots the backward resumes from are not written.  With R > 0 the stage-1
 *   value of num_segments (or its negation) is REQUIRED -- it fixes the layout of the binning buffer -- and 0 is an
 *   error (ABI 9; earlier versions accepted 0 as "forward-only"). */
int gsr_forward_stage2(int P, int R, int max_tile_instances, int num_segments, int W, int H, const float* background,
                       const float* colors_precomp, void* geom_buffer, void* binning_buffer, void* image_buffer,
                       float* out_color, gsr_stream_t stream);

/* Multi-target extension (no reference counterpart: the reference fixes
 * NUM_CHANNELS = 3 at compile time, DGR/cuda_rasterizer/config.h:15, and GauSTAR renders RGB and
 * depth-as-colour in two full passes over identical geometry, gaustar_trainers/refine.py:552 and :607).
 * num_channels = 6 blends two 3-channel targets in ONE walk after ONE stage 1: colors_precomp is [P,6]
 * (required: no in-kernel SH for the extra channels), background [6], out_color [6,H,W] planar.  Channels 0-2
 * and 3-5 equal two separate 3-channel renders bit for bit.  num_channels = 4 is RGB + ONE scalar target (colors_precomp
 * [P,4], background [4], out_color [4,H,W]): GauSTAR's depth render carries the same value in its three channels and the
 * trainer reads only the first (refine.py:616), so channel 3 of a 4-channel render is that image at about the cost of a
 * 3-channel render.  num_channels = 3 is gsr_forward_stage2. */
int gsr_forward_stage2_mt(int P, int R, int max_tile_instances, int num_segments, int num_channels, int W, int H,
                          const float* background, const float* colors_precomp, void* geom_buffer, void* binning_buffer,
                          void* image_buffer, float* out_color, gsr_stream_t stream);

/* Both halves in one call over a binning buffer the caller sized IN ADVANCE from a guess (e.g. 1.25x what the last
 * view needed): when gsr_binning_bytes_mt(R, num_segments, num_channels) <= binning_capacity the library goes straight
 * from the stage-1 read-back into the stage-2 launches and sets *blended = 1 -- the GPU does not idle while the caller
 * allocates and re-enters (10-15 us per view through a Python binding).  Otherwise *blended = 0, nothing of stage 2 has
 * run, and the caller allocates exactly and calls gsr_forward_stage2[_mt] as usual.  need_backward = 0 renders
 * forward-only (see gsr_forward_stage2).  grad_scratch (may be NULL): gsr_grad_scratch_bytes(P) bytes that the
 * following gsr_backward_mt will use as its grad_scratch; when *blended = 1 (and need_backward) the forward blend has
 * cleared them on the side, and that backward may be called with grad_scratch_zeroed = 1 (once: the backward leaves
 * the contents undefined).  Same replaced reference code as the two stages
 * (DGR/cuda_rasterizer/rasterizer_impl.cu:198-335). */
int gsr_forward_fused(int P, int D, int M, int num_channels, int need_backward, const float* means3D, const float* shs,
                      const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                      const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                      const float* projmatrix, const float* campos, int W, int H, float tan_fovx, float tan_fovy,
                      int prefiltered, const float* background, int* radii, void* geom_buffer, void* image_buffer,
                      void* binning_buffer, size_t binning_capacity, void* grad_scratch, float* out_color,
                      int* num_rendered, int* max_tile_instances, int* num_segments, int* blended, gsr_stream_t stream);

/* PLANNED forward (ABI 13): gsr_forward_fused for a camera that is rendered again and again (a training rig: every camera
 * once per epoch, gaustar_trainers/refine.py:534-548).  The caller keeps, per camera, a device buffer of gsr_plan_bytes(W, H)
 * bytes (ABI 16: room for TWO plans -- the one in use and the one being built for the next view) and a plan_info block from gsr_plan_info_new() = {state, R_cap, U_cap, max_cap, slack level, ...} (all zero at first).  A call with plan_info[0] == 0
 * renders the exact way (as gsr_forward_fused) and leaves a PLAN behind (built by one extra workgroup of that view's forward
 * blend; nobody waits for it: the next call with this plan_info adopts it): per tile the place and capacity of its bucket of
 * instances (this view's count + (1/8 of it, at least 16, + 1/8 of what the largest of the tile's eight neighbours holds more -- the
 * tiles that outgrow a bucket are the few on the surface's silhouette, whose counts jump when it moves a few pixels their way) times
 * 2^level, rounded up to whole 64-entry units), the tile's first unit and a launch order; plan_info is updated.  A call with plan_info[0] == 1 bins BY THE PLAN: preprocess claims bucket slots and writes the
 * sort keys itself, the forward blend is queued right behind it, and the host only waits for preprocess's verdict -- no
 * tile-offset scan, no scatter pass and no host round trip between the stages; (ABI 16) such a view also RE-PLANS: the same extra
 * workgroup rides in its forward blend and builds the next view's plan from this view's own tile counts into the other half of the
 * plan buffer, so a camera's plan is never older than one visit however the Gaussians move between its visits (what replaces
 * DGR/cuda_rasterizer/rasterizer_impl.cu:277-317 -- InclusiveSum, the num_rendered read-back, duplicateWithKeys,
 * identifyTileRanges -- for such a view).  *planned = 1 then, and the sizes the backward needs are the plan's CAPACITIES:
 * *num_rendered = R_cap, *num_segments = U_cap (binning_capacity must hold gsr_binning_bytes_mt(R_cap, U_cap, num_channels),
 * otherwise the call takes the exact path); images are those of the exact path bit for bit, gradients to the order of the float atomics --
 * with one exception: a planned view is never SPLIT (gsr_blend_fwd.hip: lists above 1 024 entries blended in parts), whereas the exact
 * path decides that from the current view's longest list and R; a plan is only valid while its source view would not have split at
 * three quarters of its R, so the two differ only for a view whose R has fallen below that since, and then by the parts' rounding (a tile's list is
 * the same sorted list; only where it lies differs).  A view that does not fit its plan (a bucket overflows: the Gaussians
 * have moved since the plan was made) is detected by preprocess before anything is blended; the same call then renders it the
 * exact way and re-plans with the slack level raised by one, at most 3 (*planned = -1; 0: no plan was tried).  Only views whose longest list stays within the forward blend's own sort (2 048
 * entries incl. slack) and that would not be split are planned (plan_info[0] stays 0 otherwise).  A plan is a HINT: any plan of
 * the right image size is safe for any view -- a bad one costs the fallback, never a wrong pixel.  Needs the library's
 * per-stream counter block (gsr_forward_fused's), otherwise exact.  All other arguments as gsr_forward_fused. */
#define GSR_PLAN_INFO_INTS 32
size_t gsr_plan_bytes(int W, int H);
/* A plan_info block: GSR_PLAN_INFO_INTS ints of pinned, device-mapped host memory, zeroed (the builder of a plan writes its
 * header there from the device; ordinary host memory will not do).  [0] state: 0 no plan, 1 valid, -1 this camera's views cannot
 * be planned (reset to 0 to have the next view try again); [1..3] R_cap, U_cap, max_cap of a valid plan; [4] slack level
 * (0..3); the rest belongs to the library ([5] / [6]: the half of the plan buffer in use / being written, [8..17] the arriving header).  Free with gsr_plan_info_free once no call using it is in flight. */
int* gsr_plan_info_new(void);
void gsr_plan_info_free(int* plan_info);
int gsr_forward_planned(int P, int D, int M, int num_channels, int need_backward, const float* means3D, const float* shs,
                        const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                        const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                        const float* projmatrix, const float* campos, int W, int H, float tan_fovx, float tan_fovy,
                        int prefiltered, const float* background, int* radii, void* geom_buffer, void* image_buffer,
                        void* binning_buffer, size_t binning_capacity, void* grad_scratch, float* out_color,
                        int* num_rendered, int* max_tile_instances, int* num_segments, int* blended, void* plan_buffer,
                        int* plan_info, int* planned, gsr_stream_t stream);

/* CONTENT key of a camera (ABI 15): a 64-bit hash of the sixteen floats of its view matrix, read from wherever the caller
 * keeps them (row_stride / col_stride in elements: the reference hands over a TRANSPOSED view,
 * gaustar_scene/sugar_model.py:1149-1150).  What a caller keys its per-camera plans on when it cannot know the camera any other
 * way: the reference's caller builds a fresh view-matrix tensor on every render call (sugar_model.py:1149-1163), so neither the
 * tensor nor its address identifies the camera -- its contents do.  The sixteen floats are fetched by a one-wave kernel on a
 * library-owned NON-BLOCKING stream into the calling thread's pinned pad and the host waits for that kernel only: the caller's
 * stream is neither waited for nor delayed (work queued on it keeps running underneath).  The read is therefore not ordered
 * behind kernels of the caller's stream that may still be writing the matrix; a matrix uploaded from the host (the reference's
 * case: `.cuda()` synchronises) or resident since an earlier call is final.  A key of half-written contents is harmless -- a
 * plan is a hint (gsr_forward_planned) -- it merely names no camera. */
int gsr_camera_key(const float* viewmatrix, long long row_stride, long long col_stride, unsigned long long* key);
/* The same in two halves, so that the caller's own host work (allocating the view's buffers) hides the ~10 us round trip:
 * _begin launches the read, _end (same host thread, once per _begin) waits for it and returns the key. */
int gsr_camera_key_begin(const float* viewmatrix, long long row_stride, long long col_stride);
int gsr_camera_key_end(unsigned long long* key);

/* Frees what the library keeps for (current device, stream) -- the tile-counter block of gsr_forward_fused -- e.g.
 * before the stream is destroyed.  Fails if a gsr_forward_fused on that stream is in flight on another thread. */
int gsr_release_stream_state(gsr_stream_t stream);

/* One-call forward with the reference's allocator-callback shape.  Replaces
 * CudaRasterizer::Rasterizer::forward (DGR/cuda_rasterizer/rasterizer.h:31-55).
 * Returns num_rendered through *num_rendered [host]. */
int gsr_forward(gsr_alloc_fn geometry_buffer, gsr_alloc_fn binning_buffer, gsr_alloc_fn image_buffer, void* alloc_ctx,
                int P, int D, int M, const float* background, int W, int H, const float* means3D, const float* shs,
                const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                const float* campos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color, int* radii,
                int* num_rendered, gsr_stream_t stream);

/* Backward.  Replaces CudaRasterizer::Rasterizer::backward
 * (DGR/cuda_rasterizer/rasterizer.h:57-83; rasterizer_impl.cu:340-434; backward.cu).
 * Output gradient arrays need NOT be zeroed by the caller (the reference requires zeroed
 * tensors, DGR/rasterize_points.cu:151-159; here the fill is part of the call):
 *   grad_scratch: gsr_grad_scratch_bytes(P) bytes of work space (contents undefined afterwards);
 *   dL_dmean2D [P,3], dL_dopacity [P], dL_dcolor [P,3], dL_dmean3D [P,3], dL_dcov3D [P,6] (may be NULL when
 *   cov3D_precomp is NULL: the reference writes it regardless, rasterizer_impl.cu:401-416, and its binding then drops it),
 *   dL_dsh [P,M,3] (NULL when M == 0), dL_dscale [P,3], dL_drot [P,4] (both NULL when cov3D_precomp
 *   is given). */
int gsr_backward(int P, int D, int M, int R, int num_segments, const float* background, int W, int H,
                 const float* means3D, const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                 const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                 const void* geom_buffer, const void* binning_buffer, const void* image_buffer, const float* dL_dpix,
                 void* grad_scratch, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                 float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, gsr_stream_t stream);

/* Backward of a num_channels render (see gsr_forward_stage2_mt): dL_dpix [num_channels,H,W], dL_dcolor
 * [P,num_channels]; every other gradient is the sum over the targets, i.e. what autograd would accumulate from
 * the separate backward passes of the reference.  num_channels = 3 is gsr_backward.
 * Precision of the per-pixel sums: f32 products and f32 sums for every channel of every channel count (ABI 16; the four-channel
 * kernel of ABI 13-15 took dL_dpix of channels 2 and 3 into the matrix pipe with 16 mantissa bits).
 * grad_scratch_zeroed = 1: the caller guarantees grad_scratch is all zero (gsr_forward_fused cleared it) and the
 * library skips its own fill; 0: the fill is part of the call, as in gsr_backward. */
int gsr_backward_mt(int P, int D, int M, int R, int num_segments, int num_channels, const float* background, int W,
                    int H, const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                    float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                    const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                    const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                    const float* dL_dpix, void* grad_scratch, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor,
                    float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                    int grad_scratch_zeroed, gsr_stream_t stream);

/* Near-plane visibility test.  Replaces CudaRasterizer::Rasterizer::markVisible
 * (DGR/cuda_rasterizer/rasterizer.h:24-29; rasterizer_impl.cu:54-66, :141-153).
 * present [P] uint8 (bool) out. */
int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, gsr_stream_t stream);

/* Introspection for tests/benchmarks: copies library-internal per-stage results out of the
 * scratch buffers into caller-provided DEVICE arrays (any may be NULL):
 *   means2D [P,2], conic_opacity [P,4], depths [P], rgb [P,3] (SH mode only),
 *   tile_ranges [T,2] uint32, point_list [R] uint32, final_T [H*W], n_contrib [H*W] uint32.
 * No reference counterpart (the reference exposes its scratch only as opaque bytes). */
int gsr_debug_export(int P, int R, int num_segments, int W, int H, const void* geom_buffer, const void* binning_buffer,
                     const void* image_buffer, float* means2D, float* conic_opacity, float* depths, float* rgb,
                     uint32_t* tile_ranges, uint32_t* point_list, float* final_T, uint32_t* n_contrib,
                     gsr_stream_t stream);
/* Same for the per-pixel candidate words of the binning buffer: masks [num_segments][4][64] uint64 -- unit (= 64-entry
 * segment of a tile's list; units of a tile are consecutive, tiles in index order), 8x8 block 2*by + bx of the tile, pixel
 * 8*(y % 8) + (x % 8) of the block.  Bit i of a word <=> the instance at list position 64*segment + i CAN reach
 * alpha >= 1/255 at that pixel (a conservative superset of what the pixel blends; bits of positions past the end of the
 * list are zero).  Words of segments the forward never reached (every pixel of the tile saturated before) are undefined.
 * No reference counterpart. */
int gsr_debug_export_masks(int R, int num_segments, const void* binning_buffer, uint64_t* masks, gsr_stream_t stream);

/* ---- Producers of rasterizer inputs.
 * gsr_sh_to_rgb replaces SuGaR.get_points_rgb (gaustar_scene/sugar_model.py:674-718):
 *   rgb = clamp_min(eval_sh(D, sh[:, :(D+1)^2], normalize(positions - campos)) + 0.5, 0)
 * (eval_sh: gaustar_utils/spherical_harmonics.py:117-172); gsr_sh_to_rgb_backward is its autograd backward.
 *   positions [P,3], campos [3], shs [P,M,3] with (D+1)^2 <= M, D in 0..3; rgb / dL_drgb [P,3];
 *   dL_dsh [P,M,3] (zero above the active degree), dL_dpos [P,3] -- both written outright. */
int gsr_sh_to_rgb(int P, int D, int M, const float* positions, const float* campos, const float* shs, float* rgb,
                  gsr_stream_t stream);
int gsr_sh_to_rgb_backward(int P, int D, int M, const float* positions, const float* campos, const float* shs,
                           const float* dL_drgb, float* dL_dsh, float* dL_dpos, gsr_stream_t stream);

/* Two-target variant for the one-pass RGB + depth render (gsr_forward_stage2_mt with 6 channels): colors6 [P,6] =
 * {rgb as gsr_sh_to_rgb, z, z, z} with z = the view-space depth of the position, i.e. the `point_depth.expand(-1, 3)`
 * GauSTAR renders as colours (gaustar_trainers/refine.py:603-605: world-to-view transform of sugar.points, component 2).
 * viewmatrix [4,4] as handed to the rasterizer (row-vector convention: z = (p, 1) . column 2).  The backward adds the
 * depth channels' gradient to dL_dpos.  depth_channels = 3: [P,6] as above; 1: [P,4] = {rgb, z}, for the 4-channel render
 * (the trainer only reads channel 0 of its depth render, refine.py:616).  Replaces torch.cat + two skinny GEMMs + their
 * autograd mirror per iteration. */
int gsr_sh_to_rgbd(int P, int D, int M, const float* positions, const float* campos, const float* shs,
                   const float* viewmatrix, int depth_channels, float* colors6, gsr_stream_t stream);
int gsr_sh_to_rgbd_backward(int P, int D, int M, const float* positions, const float* campos, const float* shs,
                            const float* viewmatrix, int depth_channels, const float* dL_dcolors6, float* dL_dsh,
                            float* dL_dpos, gsr_stream_t stream);

/* The same producer reading SuGaR's coefficients where they live: `_sh_coordinates_dc` [P,1,3] and `_sh_coordinates_rest`
 * [P,M-1,3] (gaustar_scene/sugar_model.py:449-450 concatenates the two on EVERY access of `sh_coordinates`, and autograd
 * splits the gradient again: 53 MB written and read back each way per render at 491 520 Gaussians, sh_levels 3), and
 * taking SuGaR.strengths (sugar_model.py:442-447: sigmoid of `all_densities`) along.
 *   M = 1 + number of rest coefficients (sh_rest may be NULL when M == 1); viewmatrix NULL: colors [P,3], depth_channels
 *   must be 0; otherwise colors [P, 3 + depth_channels] as gsr_sh_to_rgbd.  densities / opacity [P]: both NULL or both
 *   given, opacity = 1 / (1 + exp(-density)).
 * Backward: dL_dsh_dc [P,1,3] and dL_dsh_rest [P,M-1,3] written outright; dL_dpos [P,3] written, or -- accumulate_pos != 0
 * -- ADDED to what the array holds (the rasterizer's gradient w.r.t. the same positions: the sum autograd would form with
 * one more kernel); opacity / dL_dopacity / dL_ddensities [P]: all NULL or all given, dL_ddensities = dL_dopacity (1 - o) o.
 * Results are bit-identical to gsr_sh_to_rgbd on the concatenated array + torch.sigmoid. */
int gsr_sh_colors_split(int P, int D, int M, const float* positions, const float* campos, const float* sh_dc,
                        const float* sh_rest, const float* viewmatrix, int depth_channels, const float* densities,
                        float* colors, float* opacity, gsr_stream_t stream);
int gsr_sh_colors_split_backward(int P, int D, int M, const float* positions, const float* campos, const float* sh_dc,
                                 const float* sh_rest, const float* viewmatrix, int depth_channels, const float* dL_dcolors,
                                 const float* opacity, const float* dL_dopacity, float* dL_dsh_dc, float* dL_dsh_rest,
                                 float* dL_dpos, int accumulate_pos, float* dL_ddensities, gsr_stream_t stream);

/* gsr_mesh_gaussians replaces the properties SuGaR.points / .scaling / .quaternions for Gaussians bound to a
 * triangle mesh (gaustar_scene/sugar_model.py:417-435, :457-476, :478-508; pytorch3d 0.7.4 face normals,
 * quaternion_to_matrix, matrix_to_quaternion): Gaussian n = f*G + g of face f gets
 *   points[n]  = sum_k bary[g][k] * verts[faces[f][k]]  (+ delta_t[n]),
 *   scaling[n] = {thickness, clamp(exp(raw_scales[n]), min_scale, max_scale)},
 *   quaternions[n] = normalize(matrix_to_quaternion(R(delta_r[n]) * [n_f | R1 | R2])),  (w, x, y, z),
 * with [R1 R2] the face's (first edge, normal x edge) basis turned by normalize(raw_complex[n]).
 *   verts [V,3] f32, faces [F,3] int64, bary [G,3], raw_scales / raw_complex [F*G,2]; delta_t [F*G,3] and
 *   delta_r [F*G,4] are the loose-bind offsets and may be NULL; min_scale / max_scale: -inf / +inf for "none".
 * gsr_mesh_gaussians_backward is its autograd backward: dL_dverts [V,3] is zeroed and accumulated inside;
 * dL_draw_scales, dL_draw_complex (and dL_ddelta_t, dL_ddelta_r when non-NULL) are written outright; any of
 * the three incoming gradients may be NULL (= zero).
 * (ABI 14) clear_dL_dverts / V of the forward: when non-NULL, the forward's launch also sets those [V,3] floats to zero -- the
 * accumulator of the backward to come, which is then called with dL_dverts_cleared = 1 and skips its own fill (a launch of its
 * own on the stream for 0.5 MB).  NULL / dL_dverts_cleared = 0: as before. */
int gsr_mesh_gaussians(int F, int G, const float* verts, const long long* faces, const float* bary,
                       const float* raw_scales, const float* raw_complex, float thickness, float min_scale,
                       float max_scale, const float* delta_t, const float* delta_r, float* points, float* scaling,
                       float* quaternions, float* clear_dL_dverts, int V, gsr_stream_t stream);
int gsr_mesh_gaussians_backward(int F, int G, int V, const float* verts, const long long* faces, const float* bary,
                                const float* raw_scales, const float* raw_complex, float min_scale, float max_scale,
                                const float* delta_r, const float* dL_dpoints, const float* dL_dscaling,
                                const float* dL_dquaternions, float* dL_dverts, float* dL_draw_scales,
                                float* dL_draw_complex, float* dL_ddelta_t, float* dL_ddelta_r, int dL_dverts_cleared,
                                gsr_stream_t stream);

/* ---- Image-space losses either side of the rasterizer.
 * gsr_l1_ssim replaces  (1 - f) * l1_loss(pred, gt) + f * (1 - ssim(pred, gt))  (gaustar_trainers/refine.py:451-453
 * over gaustar_utils/loss_utils.py:17-62: 11x11 Gaussian window, sigma 1.5, zero padding, mean over all
 * elements) AND its autograd backward w.r.t. pred, in two tiled passes.  Images are indexed [C,H,W] through
 * element strides (channel, row, column), so the reference's transposed views of [H,W,3] storage and its
 * margin crop (refine.py:584-594: pass pointers to the crop origin and the cropped H, W) need no copy.
 *   loss_out [3] device floats: {loss, l1 mean, ssim mean};  dL_dpred may be NULL (value only), else it receives
 *   d loss / d pred for the H x W region through its own strides (planar [C,H,W] is what gsr_backward reads).
 *   workspace: gsr_l1_ssim_workspace_bytes(C, H, W) bytes.  No host synchronisation. */
size_t gsr_l1_ssim_workspace_bytes(int C, int H, int W);
int gsr_l1_ssim(int C, int H, int W, const float* pred, long long pred_sc, long long pred_sy, long long pred_sx,
                const float* gt, long long gt_sc, long long gt_sy, long long gt_sx, float dssim_factor,
                void* workspace, float* loss_out, float* dL_dpred, long long grad_sc, long long grad_sy,
                long long grad_sx, gsr_stream_t stream);

/* Masked depth + silhouette L1 of gaustar_trainers/refine.py:634-660 (depth_alpha = False branch):
 *   depth_factor * mean_{gt < max_depth} |pred - gt|  +  mask_factor * mean_{gt > max_depth} |pred - max_depth|
 * on one [H,W] depth image (strided), with the gradient w.r.t. pred.
 *   loss_out [4] device floats: {depth term, mask term, #foreground, #background}. */
size_t gsr_depth_l1_workspace_bytes(void);
int gsr_depth_l1(int H, int W, const float* pred, long long pred_sy, long long pred_sx, const float* gt,
                 long long gt_sy, long long gt_sx, float max_depth, float depth_factor, float mask_factor,
                 void* workspace, float* loss_out, float* dL_dpred, long long grad_sy, long long grad_sx,
                 gsr_stream_t stream);

/* The same two losses with the gradient pass as a call of its own (ABI 12), for callers that learn the incoming
 * d(total)/d(loss) only in their backward pass (autograd's grad_output; refine.py:794 `loss.backward()`):
 *   gsr_l1_ssim(..., dL_dpred = NULL) / gsr_depth_l1(..., dL_dpred = NULL) / gsr_rgb_depth_loss evaluate the values and leave
 *   what the gradient needs in `workspace` resp. `loss_out`; gsr_l1_ssim_backward / gsr_depth_l1_backward then write
 *   grad_scale[0] * d loss / d pred, with grad_scale a DEVICE scalar (NULL = 1) -- no elementwise multiply over the image
 *   afterwards.  `workspace` / `stats` are the buffers of the value call, unmodified; pred / gt the same images.
 * gsr_rgb_depth_loss = the value passes of gsr_l1_ssim on an RGB image and of gsr_depth_l1 on a depth image with ONE
 * reduction kernel: loss_out [8] = {l1 + dssim loss, l1 mean, ssim mean, depth term, mask term, #fg, #bg, total of the
 * three terms}; loss_out + 3 is the `stats` argument of gsr_depth_l1_backward.  (ABI 14) The same eight floats are also left
 * in ssim_workspace at byte gsr_l1_ssim_workspace_bytes(C, H, W) - 256: gsr_rgb_depth_loss_backward's loss_out may point
 * there, so that a caller can hand loss_out itself to its user (who may modify it) and keep only the workspace. */
int gsr_l1_ssim_backward(int C, int H, int W, const float* pred, long long pred_sc, long long pred_sy, long long pred_sx,
                         const float* gt, long long gt_sc, long long gt_sy, long long gt_sx, float dssim_factor,
                         const void* workspace, const float* grad_scale, float* dL_dpred, long long grad_sc, long long grad_sy,
                         long long grad_sx, gsr_stream_t stream);
int gsr_depth_l1_backward(int H, int W, const float* pred, long long pred_sy, long long pred_sx, const float* gt,
                          long long gt_sy, long long gt_sx, float max_depth, float depth_factor, float mask_factor,
                          const float* stats, const float* grad_scale, float* dL_dpred, long long grad_sy, long long grad_sx,
                          gsr_stream_t stream);
int gsr_rgb_depth_loss(int C, int H, int W, const float* pred, long long pred_sc, long long pred_sy, long long pred_sx,
                       const float* gt, long long gt_sc, long long gt_sy, long long gt_sx, float dssim_factor,
                       void* ssim_workspace, int Hd, int Wd, const float* depth_pred, long long dpred_sy, long long dpred_sx,
                       const float* depth_gt, long long dgt_sy, long long dgt_sx, float max_depth, float depth_factor,
                       float mask_factor, void* depth_workspace, float* loss_out, gsr_stream_t stream);

/* Both gradient passes of gsr_rgb_depth_loss in ONE launch: loss_out is the [8] vector the value call wrote (its elements
 * 5 and 6 are the pixel counts the depth gradient divides by), ssim_workspace the value call's, unmodified. */
int gsr_rgb_depth_loss_backward(int C, int H, int W, const float* pred, long long pred_sc, long long pred_sy, long long pred_sx,
                                const float* gt, long long gt_sc, long long gt_sy, long long gt_sx, float dssim_factor,
                                const void* ssim_workspace, int Hd, int Wd, const float* depth_pred, long long dpred_sy,
                                long long dpred_sx, const float* depth_gt, long long dgt_sy, long long dgt_sx, float max_depth,
                                float depth_factor, float mask_factor, const float* loss_out, const float* grad_scale,
                                float* dL_dpred, long long grad_sc, long long grad_sy, long long grad_sx, float* dL_ddepth,
                                long long dgrad_sy, long long dgrad_sx, gsr_stream_t stream);

/* Tuning aid: when device_buffer is non-NULL (4*T uint64), the two blend kernels record the start/end wall
 * clock (100 MHz) of every workgroup: forward at [2*b], backward at [2*(T+b)], b = launch index.  NULL = off. */
int gsr_debug_set_trace(void* device_buffer);
/* Experiments: a launch order for the backward blend's units ([num_segments] unit ids by dispatch position; NULL: none). */
int gsr_debug_set_bwd_order(const void* device_order);
/* Tuning: workgroups per CU the runtime grants the exact / the planned preprocess kernel. */
int gsr_debug_preprocess_occupancy(int* exact, int* planned);

/* Per-kernel timing for benchmarks (no reference counterpart; the reference has no profiling hooks,
 * SURVEY.md section 5).  While enabled, every stage this thread launches is bracketed by HIP events
 * recorded on the launch stream.  gsr_profile_read waits for the recorded events and returns, per
 * stage (0 .. gsr_num_stages()-1, names from gsr_stage_name), the summed milliseconds and the number
 * of launches since the last reset.  ms, counts: [host] arrays of gsr_num_stages() entries. */
int gsr_num_stages(void);
const char* gsr_stage_name(int stage);
int gsr_profile_enable(int on);
int gsr_profile_read(float* ms, int* counts, int reset);

/* Host-side slack (benchmarks): nanoseconds this PROCESS has spent, summed over all threads, waiting for the stage-1
 * totals to arrive (the forward's one host synchronisation, rasterizer_impl.cu:281) since the last reset, and the number
 * of waits.  A step whose wait is near zero is bound by the host (Python, launches), not by the GPU.
 * wait_ns, waits: [host] out, may be NULL. */
int gsr_debug_host_wait(long long* wait_ns, long long* waits, int reset);

/* ---- Optimiser step.  gsr_adam_step replaces one parameter tensor's share of torch.optim.Adam.step() as GauSTAR
 * configures it (gaustar_scene/sugar_optimizer.py:87, :99-101; torch/optim/adam.py::_single_tensor_adam without weight
 * decay / amsgrad / maximize): in place on param, exp_avg, exp_avg_sq [n] f32 (16-byte aligned), grad [n] read only;
 * step = the 1-based count of this update (bias corrections 1 - beta^step).  Hyper-parameters are doubles, as the
 * reference holds them in Python floats: 1 - beta2 formed from a float32 beta2 would already be off by 1e-5. */
int gsr_adam_step(long long n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                  double beta2, double eps, int step, gsr_stream_t stream);
/* The same update for `count` tensors in one launch per 16 tensors (the optimiser of sugar_optimizer.py:67-87 holds eight
 * tensors, five of them a few MB: a launch each is mostly overhead).  numel, params, grads, exp_avgs, exp_avg_sqs, lrs:
 * [host] arrays of `count` entries (device pointers / element counts / per-tensor learning rates -- the groups differ in
 * nothing else); tensors with numel <= 0 are skipped.  Element for element the result of gsr_adam_step. */
int gsr_adam_step_multi(int count, const long long* numel, float* const* params, const float* const* grads,
                        float* const* exp_avgs, float* const* exp_avg_sqs, const double* lrs, double beta1, double beta2,
                        double eps, int step, gsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H_INCLUDED */
