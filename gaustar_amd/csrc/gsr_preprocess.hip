// gsr_preprocess.hip -- per-Gaussian forward stage + tile counting, and markVisible.
//
// Computes what the reference's preprocessCUDA (DGR/cuda_rasterizer/forward.cu:155-256)
// computes, with identical culling decisions and identical `radii`, but writes a different,
// smaller geometry state (44 B/Gaussian, +12 in SH mode, vs the reference's 79) laid out for the
// wave-per-8x8 blend kernels: two float4 records gathered per instance, plus the candidate tile
// rect so that the count pass (here) and the scatter pass walk the same tiles.
//
// Tile set actually binned = tiles of the reference rect (auxiliary.h:46-56, 3-sigma circle) in which
// the splat can reach alpha >= 1/255 at all (exact quadratic-vs-rectangle test, block_min_half_quad).
// Dropped tiles could only hold pairs the reference skips at forward.cu:340-342, so results are
// unchanged while num_rendered (library-internal) shrinks.  Per-tile counters are bumped with
// wave-aggregated atomics (one per distinct tile per wave round instead of one per lane).
#include "gsr_internal.h"
#include "gsr_ref_order.h"

namespace gsr {

// PLANNED (gsr_forward_planned): the tiles' buckets are laid out already (gsr_internal.h, "planned binning").  The walk is the
// same; the records stay in LDS ({lane, table slot, offset} in one word), the workgroup's one memory atomic per occupied slot
// goes to the tile's CURSOR and returns where the workgroup's span starts inside the tile's bucket, and the keys are written
// right here -- what scatter_kernel does behind a scan in the exact path.  A span that does not fit its bucket, a full table or
// a full record array raise the plan's flag (nothing is written outside a bucket).
// (at most 80 scalar registers: with 82-96 the hardware admits SEVEN 256-thread workgroups per CU where the occupancy API and the
// compiler say eight -- MI355X_MICROARCH.md "Residency" --, config C's 1 920 workgroups then run as 1 792 + a second generation of
// 128 that starts when the first ones leave: the planned kernel, at 85, ended at 31.6 us with its workgroups living 18)
template <bool PLANNED>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80)))
preprocess_kernel(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ shs,
                  const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                  const float* __restrict__ scales, float scale_modifier, const float* __restrict__ rotations,
                  const float* __restrict__ cov3D_precomp, const float* __restrict__ view,
                  const float* __restrict__ proj, const float* __restrict__ campos, int W, int H, float tan_fovx,
                  float tan_fovy, float focal_x, float focal_y, int gx, int gy, int* __restrict__ radii,
                  float4* __restrict__ g0, float4* __restrict__ g1, float* __restrict__ depth,
                  ushort4* __restrict__ rect, float* __restrict__ rgb, uint32_t* __restrict__ tile_count,
                  uint4* __restrict__ wg_recs, uint2* __restrict__ wg_tab, uint32_t* __restrict__ wg_nrec,
                  uint32_t* __restrict__ totals, uint32_t view_token, PlanRun plan)
{
    constexpr int AGG_SLOTS = WG_TAB_SLOTS;
    __shared__ uint32_t agg_key[AGG_SLOTS], agg_cnt[AGG_SLOTS];
    __shared__ uint32_t tab_full;   // set by the first group that finds no slot: later groups do not probe at all
    constexpr uint32_t WAVE_CAP = WG_REC_CAP / 4;
    __shared__ uint32_t rec_lds[PLANNED ? WG_REC_CAP : 1];   // (PLANNED) records {lane << 24 | slot << 8 | offset}, a quarter per wave
    __shared__ uint32_t depth_lds[PLANNED ? 256 : 1];        // (PLANNED) every thread's depth bits
    __shared__ uint32_t misfit;                              // (PLANNED) this workgroup found something that does not fit the plan
    for (int i = threadIdx.x; i < AGG_SLOTS; i += blockDim.x) { agg_key[i] = 0xffffffffu; agg_cnt[i] = 0u; }
    if (threadIdx.x == 0) { tab_full = 0u; misfit = 0u; }
    __syncthreads();
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = idx < P;
#ifdef GSR_PRE_PHASES   // devtool build (planned variant): thread 0's stamps go to the head of the workgroup's unused record area
    uint64_t* const ph = (PLANNED && threadIdx.x == 0) ? reinterpret_cast<uint64_t*>(wg_recs + (size_t)blockIdx.x * WG_REC_CAP) : nullptr;
    int ph_i = 0;
#define PRE_PHASE() do { if (ph) ph[ph_i++] = wall_clock64(); } while (0)
#else
#define PRE_PHASE() do { } while (0)
#endif
    PRE_PHASE();
    // SH mode: the wave's 64 coefficient rows come in through LDS (coalesced), see sh_stage_load
    extern __shared__ float sh_lds[];
    float* my_sh = nullptr;
    if (colors_precomp == nullptr) {
        float* wave_rows = sh_lds + (size_t)(threadIdx.x >> 6) * 64 * sh_row_stride(M);
        sh_stage_load(wave_rows, shs, (size_t)(idx - lane), P, M, lane);
        __builtin_amdgcn_wave_barrier();
        my_sh = wave_rows + lane * sh_row_stride(M);
    }

    int out_radius = 0;
    ushort4 out_rect = make_ushort4(0, 0, 0, 0);
    float px = 0.f, py = 0.f, conic_a = 1.f, conic_b = 0.f, conic_c = 1.f, tau = -1.f;
    float my_depth = 0.f;

    if (valid) {
        const Vec3 p = load3(means3D, idx);
        // (requested with the other inputs: read where it is used -- behind the near-plane, determinant and rect tests -- it
        // was a second dependent trip to memory in the middle of the kernel's one occupancy wave)
        const float op_in = opacities[idx];
        const float view_z = view_depth(p, view);
        // Near-plane cull only (auxiliary.h:154; the NDC side test is dead code there).
        if (view_z > NEAR_Z) {
            // Projected centre, covariance and conic in the reference build's operation order (gsr_ref_order.h): every
            // alpha >= 1/255 decision downstream is a function of THESE bits.
            float c3[6];
            if (cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; k++) c3[k] = cov3D_precomp[6 * (size_t)idx + k];
            } else {
                const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
                cov3d_ref_order(load3(scales, idx), scale_modifier, q, c3);
            }
            const Projected pr = project_ref_order(p, c3, view, proj, focal_x, focal_y, tan_fovx, tan_fovy, W, H);
            const float ca = pr.cov_a, cc = pr.cov_c;

            const float det = pr.det;
            if (det != 0.0f) {
                conic_a = pr.conic_a; conic_b = pr.conic_b; conic_c = pr.conic_c;
                const float my_radius = radius_ref_order(ca, cc, det);
                px = pr.px; py = pr.py;
                const int r = (int)my_radius;
                // Reference tile rect (C truncation toward zero, clamped to the grid).
                int rx0 = min(gx, max(0, (int)((px - r) / TILE)));
                int ry0 = min(gy, max(0, (int)((py - r) / TILE)));
                int rx1 = min(gx, max(0, (int)((px + r + TILE - 1) / TILE)));
                int ry1 = min(gy, max(0, (int)((py + r + TILE - 1) / TILE)));
                if ((rx1 - rx0) * (ry1 - ry0) != 0) {
                    out_radius = r;
                    const float op = op_in;
                    // alpha = min(0.99, op*exp(power)) >= 1/255  <=>  -power <= ln(255*op).  tau carries an
                    // absolute safety margin far above any rounding in exp or in the quadratic form.
                    if (op * 255.0f * 1.0001f >= 1.0f) {
                        tau = fmaxf(__logf(255.0f * op), 0.0f) * 1.0005f + 0.02f;
                        // shrink the candidate rect to the splat's alpha >= 1/255 bounding box first
                        const float hx = sqrtf(2.0f * tau * ca) * 1.0005f + 0.01f;
                        const float hy = sqrtf(2.0f * tau * cc) * 1.0005f + 0.01f;
                        const int ix0 = (int)ceilf(px - hx), ix1 = (int)floorf(px + hx);
                        const int iy0 = (int)ceilf(py - hy), iy1 = (int)floorf(py + hy);
                        if (ix1 < ix0 || iy1 < iy0 || ix1 < 0 || iy1 < 0) {
                            rx1 = rx0; ry1 = ry0;
                        } else {
                            rx0 = max(rx0, max(ix0, 0) / TILE);
                            ry0 = max(ry0, max(iy0, 0) / TILE);
                            rx1 = max(rx0, min(rx1, ix1 / TILE + 1));
                            ry1 = max(ry0, min(ry1, iy1 / TILE + 1));
                        }
                    } else {
                        rx1 = rx0; ry1 = ry0;   // can never reach alpha >= 1/255 anywhere
                    }
                    if (colors_precomp == nullptr) {
                        // SH -> RGB (forward.cu:20-71), +0.5 and clamp at 0.
                        const float dx = p.x - campos[0], dy = p.y - campos[1], dz = p.z - campos[2];
                        const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
                        float basis[16];
                        sh_basis(D, dx * inv, dy * inv, dz * inv, basis);
                        const int nb = (D + 1) * (D + 1);
                        const float* sh = my_sh;
                        float cr = 0.f, cg = 0.f, cbb = 0.f;
                        for (int k = 0; k < nb; k++) {
                            cr += basis[k] * sh[3 * k]; cg += basis[k] * sh[3 * k + 1]; cbb += basis[k] * sh[3 * k + 2];
                        }
                        rgb[3 * (size_t)idx + 0] = fmaxf(cr + 0.5f, 0.0f);
                        rgb[3 * (size_t)idx + 1] = fmaxf(cg + 0.5f, 0.0f);
                        rgb[3 * (size_t)idx + 2] = fmaxf(cbb + 0.5f, 0.0f);
                    }
                    g0[idx] = make_float4(px, py, conic_a, conic_b);
                    g1[idx] = make_float4(conic_c, op, tau, 0.0f);
                    depth[idx] = view_z;
                    my_depth = view_z;
                    out_rect = make_ushort4((unsigned short)rx0, (unsigned short)ry0, (unsigned short)rx1,
                                            (unsigned short)ry1);
                }
            }
        }
        radii[idx] = out_radius;
        rect[idx] = out_rect;
    }
    PRE_PHASE();   // inputs loaded, projected, state stored
    // All 64 lanes take part (lanes without work carry an empty rect).
    // The wave-level groups are merged once more per WORKGROUP in a small LDS table (open addressing on the tile id)
    // before they reach memory: the 256 Gaussians of a workgroup are neighbours on the mesh and hit the same dozen
    // tiles from all four waves and in every round, and the memory-side atomics are what bounds this kernel
    // (31.6 us with, 15.3 us without them).
    // The same table hands every instance its PLACE: the LDS counter of the tile's slot, read back by the group's leader,
    // is the group's offset inside the workgroup's span of that tile; the workgroup's one memory atomic per slot returns
    // where that span starts within (tile, shard).  Instances leave as records {gaussian, depth bits, slot, offset} --
    // complete the moment they are formed; each wave fills its own quarter of the workgroup's record array, a running
    // count in a scalar register -- and the table follows at the end: scatter then is one store per record instead of a
    // second tile walk with returning atomics (28 -> 13 us).  A workgroup whose table or record array does not suffice (close-ups: hundreds of tiles per
    // workgroup) marks the VIEW with its token: counting stays exact, and scatter falls back to walking the tiles.
    uint32_t* const my_row = tile_count + (size_t)(blockIdx.x & (NSHARD - 1)) * shard_stride(gx * gy);
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint4* const wave_recs = wg_recs + (size_t)blockIdx.x * WG_REC_CAP + (size_t)(threadIdx.x >> 6) * WAVE_CAP;
    uint32_t n_wave = 0;   // (wave-uniform) records of this wave so far
    // (No wave-level grouping of the lanes by tile here: with the LDS table in front of memory every lane simply takes its
    // own place with one returning LDS atomic -- same-address lanes serialise inside that one instruction, which is far
    // cheaper than the ballot loop that used to elect a leader per distinct tile.)
    // one round of the walk: every lane brings a reachable tile of the splat of lane `src` (or -1); -> whether any lane had one
    const auto visit = [&](int tile, uint32_t src, uint32_t src_idx, uint32_t src_depth) -> bool {
        const unsigned long long act = __ballot(tile >= 0);
        if (act == 0ull) return false;
        if (tile >= 0) {
            uint32_t slot = 0xffffffffu, off = 0u;
            // (multiplicative hash: a workgroup's ~90 tiles are runs of consecutive ids in rows gx apart, which
            // `tile & 255` folds onto each other into long probe chains)
            uint32_t h = ((uint32_t)tile * 0x9E3779B1u) >> (32 - WG_TAB_LOG2);   // (the top bits)
            const int max_probe = tab_full ? 0 : 24;
            for (int probe = 0; probe < max_probe; probe++) {
                const uint32_t prev = atomicCAS(&agg_key[h], 0xffffffffu, (uint32_t)tile);
                if (prev == 0xffffffffu || prev == (uint32_t)tile) { slot = h; break; }
                h = (h + 1) & (AGG_SLOTS - 1);
            }
            if (slot != 0xffffffffu) {
                off = atomicAdd(&agg_cnt[slot], 1u);
            } else if constexpr (PLANNED) {   // table full around this hash: the view does not fit its plan
                tab_full = 1u;
                misfit = 1u;
            } else {   // table full around this hash: count directly, flag the view
                atomicAdd(&my_row[tile], 1u);
                tab_full = 1u;
                totals[4] = view_token;
            }
            const uint32_t pos = n_wave + (uint32_t)__popcll(act & lt);
            if constexpr (PLANNED) {
                // (a Gaussian meets a tile once, so an offset inside a workgroup's span is below 256)
                // (an instance that found no table slot still takes its place in the record array -- as a record nobody acts on)
                if (pos < WAVE_CAP)
                    rec_lds[(threadIdx.x >> 6) * WAVE_CAP + pos] = slot != 0xffffffffu ? ((src << 24) | (slot << 8) | off) : 0xffffffffu;
            } else {
                if (pos < WAVE_CAP) wave_recs[pos] = make_uint4(src_idx, src_depth, slot, off);
            }
        }
        n_wave += (uint32_t)__popcll(act);
        return true;
    };
    // lanes with ordinary rects walk their own tiles in step; giant splats are then walked by the whole wave, one after the other
    const bool big = tau >= 0.0f && rect_is_big(out_rect);
    TileWalker walker(big ? make_ushort4(0, 0, 0, 0) : out_rect, px, py, conic_a, conic_b, conic_c, tau, gx, lane);
    while (visit(walker.next_tile(), (uint32_t)lane, (uint32_t)idx, __float_as_uint(my_depth))) {}
    for (unsigned long long bigs = __ballot(big); bigs != 0ull; bigs &= bigs - 1ull) {
        const int src = __ffsll((unsigned long long)bigs) - 1;
        const CoopSplat cs(out_rect, px, py, conic_a, conic_b, conic_c, tau, src);
        const uint32_t src_idx = (uint32_t)__shfl(idx, src, 64), src_depth = (uint32_t)__shfl((int)__float_as_uint(my_depth), src, 64);
        for (int base = 0; base < cs.n; base += 64) visit(cs.tile(base, lane, gx), (uint32_t)src, src_idx, src_depth);
    }
    PRE_PHASE();   // tiles walked (this wave)
    if constexpr (PLANNED) {
        depth_lds[threadIdx.x] = __float_as_uint(my_depth);
        if (lane == 0 && n_wave > WAVE_CAP) misfit = 1u;
        __syncthreads();
        PRE_PHASE();   // every wave has walked
        // one returning atomic per occupied slot on the tile's cursor: where this workgroup's span starts inside the bucket
        // (agg_key is reused for the absolute position of the span; 0xffffffff: the slot is empty or its span does not fit)
        for (int i = threadIdx.x; i < AGG_SLOTS; i += 256) {
            const uint32_t tile = agg_key[i];
            uint32_t at = 0xffffffffu;
            if (tile != 0xffffffffu) {
                const uint2 bucket = plan.ranges[tile];
                const uint32_t cnt = agg_cnt[i];
                const uint32_t old = atomicAdd(&plan.cursor[(size_t)tile * PLAN_CURSOR_STRIDE], cnt);
                if (old + cnt <= bucket.y) at = bucket.x + old;
                else misfit = 1u;
            }
            agg_key[i] = at;
        }
        __syncthreads();
        PRE_PHASE();   // cursor atomics returned
        {
            const uint32_t wv = threadIdx.x >> 6;
            const uint32_t nr = min(n_wave, WAVE_CAP);
            const uint32_t gbase = (uint32_t)(idx - lane);
            for (uint32_t i = lane; i < nr; i += 64u) {
                const uint32_t r = rec_lds[wv * WAVE_CAP + i];
                if (r == 0xffffffffu) continue;
                const uint32_t src = r >> 24, at = agg_key[(r >> 8) & 0xffffu];
                if (at != 0xffffffffu)
                    plan.keys[at + (r & 255u)] = ((uint64_t)depth_lds[wv * 64u + src] << 32) | (gbase + src);
            }
        }
        // A misfit raises the plan's flag (the token of this view: unique, so the word never has to be cleared).  Nobody reads it
        // before the next kernel: the forward blend, queued right behind, reports the verdict to the host and leaves the view alone.
        PRE_PHASE();   // keys stored (this wave)
        if (threadIdx.x == 0 && misfit) atomicExch(plan.sync + 9 * PLAN_SYNC_STRIDE, plan.token);
    } else {
    if (lane == 0) {
        wg_nrec[blockIdx.x * 4 + (threadIdx.x >> 6)] = n_wave;
        if (n_wave > WAVE_CAP) totals[4] = view_token;
    }
    __syncthreads();
    uint2* const tab = wg_tab + (size_t)blockIdx.x * WG_TAB_SLOTS;
    for (int i = threadIdx.x; i < AGG_SLOTS; i += blockDim.x) {
        uint2 e = make_uint2(0xffffffffu, 0u);
        if (agg_key[i] != 0xffffffffu) e = make_uint2(agg_key[i], atomicAdd(&my_row[agg_key[i]], agg_cnt[i]));
        tab[i] = e;
    }
    }
}

void launch_preprocess(int P, int D, int M, const float* means3D, const float* shs, const float* colors_precomp,
                       const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                       const float* cov3D_precomp, const float* view, const float* proj, const float* campos, int W,
                       int H, float tan_fovx, float tan_fovy, int* radii, GeomState g, ImageState im, uint32_t view_token,
                       hipStream_t st)
{
    const Tiles t = tiles_of(W, H);
    const float focal_y = H / (2.0f * tan_fovy), focal_x = W / (2.0f * tan_fovx);   // rasterizer_impl.cu:222-223
    const size_t lds = colors_precomp ? 0 : sh_stage_bytes(M, 4);
    preprocess_kernel<false><<<(P + 255) / 256, 256, lds, st>>>(P, D, M, means3D, shs, colors_precomp, opacities, scales,
                                                       scale_modifier, rotations, cov3D_precomp, view, proj, campos, W,
                                                       H, tan_fovx, tan_fovy, focal_x, focal_y, t.gx, t.gy, radii,
                                                       g.g0, g.g1, g.depth, g.rect, g.rgb, im.tile_count, g.wg_recs, g.wg_tab,
                                                       g.wg_nrec, im.totals, view_token, PlanRun{});
}

void launch_preprocess_planned(int P, int D, int M, const float* means3D, const float* shs, const float* colors_precomp,
                               const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                               const float* cov3D_precomp, const float* view, const float* proj, const float* campos, int W,
                               int H, float tan_fovx, float tan_fovy, int* radii, GeomState g, ImageState im, PlanRun plan,
                               hipStream_t st)
{
    const Tiles t = tiles_of(W, H);
    const float focal_y = H / (2.0f * tan_fovy), focal_x = W / (2.0f * tan_fovx);
    const size_t lds = colors_precomp ? 0 : sh_stage_bytes(M, 4);
    preprocess_kernel<true><<<(P + 255) / 256, 256, lds, st>>>(P, D, M, means3D, shs, colors_precomp, opacities, scales,
                                                      scale_modifier, rotations, cov3D_precomp, view, proj, campos, W,
                                                      H, tan_fovx, tan_fovy, focal_x, focal_y, t.gx, t.gy, radii,
                                                      g.g0, g.g1, g.depth, g.rect, g.rgb, im.tile_count, g.wg_recs, g.wg_tab,
                                                      g.wg_nrec, im.totals, plan.token, plan);
}

// (tuning: resident workgroups per CU the runtime computes for the two preprocess kernels)
void preprocess_occupancy(int* exact, int* planned)
{
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(exact, preprocess_kernel<false>, 256, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(planned, preprocess_kernel<true>, 256, 0);
}

// rasterizer_impl.cu:54-66 (checkFrustum): present = view-space z > 0.2.
__global__ void __launch_bounds__(256)
mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ view, uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    present[idx] = view_depth(load3(means3D, idx), view) > NEAR_Z ? 1 : 0;
}

void launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t st)
{
    mark_visible_kernel<<<(P + 255) / 256, 256, 0, st>>>(P, means3D, view, present);
}

}  // namespace gsr
