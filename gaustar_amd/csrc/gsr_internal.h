// gsr_internal.h -- scratch layout, launch prototypes and shared device math of the
// MI355X-native rasterizer.  gfx950 only (wave64, 256 CUs, 160 KB LDS/CU).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace gsr {

constexpr int TILE = 16;          // binning tile edge in pixels (matches the reference's BLOCK_X/Y so that the
                                  // reference's 3-sigma tile-rect truncation is reproduced exactly)
constexpr int SUB = 8;            // one wave64 blends an 8x8 pixel block (4 waves per tile)
constexpr float NEAR_Z = 0.2f;    // auxiliary.h:154
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float T_EPS = 0.0001f;

struct Tiles { int gx, gy, T; };
__host__ __device__ inline Tiles tiles_of(int W, int H)
{
    Tiles t; t.gx = (W + TILE - 1) / TILE; t.gy = (H + TILE - 1) / TILE; t.T = t.gx * t.gy; return t;
}

// ---------------------------------------------------------------- scratch carving
// 256-byte aligned sub-allocations; layout is a pure function of (P), (W,H), (R) so that
// backward can rebuild it (rasterizer_impl.cu:371-373 does the same with fromChunk).
__host__ __device__ inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct GeomState {           // per Gaussian
    float4* g0;              // {pix_x, pix_y, conic_a, conic_b}
    float4* g1;              // {conic_c, opacity, cull_tau = ln(255*opacity) + margin (< 0: never visible), 0}
    float* depth;            // view-space z (sort key)
    ushort4* rect;           // tile rect {min_x, min_y, max_x, max_y} actually binned (empty => no instances)
    float* rgb;              // SH-evaluated colours [P,3] (SH mode only, but always carved)
    // What preprocess workgroup b (256 Gaussians) leaves for scatter: its (Gaussian, tile) instances as records
    // {gaussian, depth bits, slot of the tile in the workgroup's table, position inside the workgroup's span of that tile}
    // and the table {tile, first rank of the workgroup's span within (tile, shard)} -- see gsr_preprocess.hip.
    uint4* wg_recs;          // [ceil(P / 256)][4 waves][WG_REC_CAP / 4]
    uint2* wg_tab;           // [ceil(P / 256)][WG_TAB_SLOTS]
    uint32_t* wg_nrec;       // [ceil(P / 256)][4] records each wave produced (more than its quarter: the view is flagged)
    size_t bytes;
};
// (table: open addressing on the tile id, a power of two and a multiple of the workgroup's 256 threads; records: a quarter per wave.
// 512 slots / 2 048 records since round 5: at 4K a workgroup's 256 Gaussians meet four times the tiles they meet at 1080p, and a
// view whose workgroups run out of either takes the slower tile-walking scatter and cannot be planned; config C: unchanged)
#ifndef GSR_WG_TAB_SLOTS
#define GSR_WG_TAB_SLOTS 512
#endif
#ifndef GSR_WG_REC_CAP
#define GSR_WG_REC_CAP 2048
#endif
constexpr int WG_REC_CAP = GSR_WG_REC_CAP, WG_TAB_SLOTS = GSR_WG_TAB_SLOTS;
static_assert(WG_TAB_SLOTS >= 256 && (WG_TAB_SLOTS & (WG_TAB_SLOTS - 1)) == 0 && WG_TAB_SLOTS <= 65536 && WG_REC_CAP % 256 == 0, "");
constexpr int WG_TAB_LOG2 = WG_TAB_SLOTS == 256 ? 8 : WG_TAB_SLOTS == 512 ? 9 : WG_TAB_SLOTS == 1024 ? 10 : WG_TAB_SLOTS == 2048 ? 11 : 12;
static_assert((1 << WG_TAB_LOG2) == WG_TAB_SLOTS, "table sizes 256 .. 4096");
inline GeomState carve_geom(void* base, int P)
{
    GeomState s; size_t o = 0; char* b = (char*)base;
    s.g0 = (float4*)(b + o); o = align_up(o + sizeof(float4) * (size_t)P);
    s.g1 = (float4*)(b + o); o = align_up(o + sizeof(float4) * (size_t)P);
    s.depth = (float*)(b + o); o = align_up(o + sizeof(float) * (size_t)P);
    s.rect = (ushort4*)(b + o); o = align_up(o + sizeof(ushort4) * (size_t)P);
    s.rgb = (float*)(b + o); o = align_up(o + sizeof(float) * 3 * (size_t)P);
    const size_t nwg = ((size_t)P + 255) / 256;
    s.wg_recs = (uint4*)(b + o); o = align_up(o + sizeof(uint4) * WG_REC_CAP * nwg);
    s.wg_tab = (uint2*)(b + o); o = align_up(o + sizeof(uint2) * WG_TAB_SLOTS * nwg);
    s.wg_nrec = (uint32_t*)(b + o); o = align_up(o + 16 * nwg);
    s.bytes = o + 256;
    return s;
}

// Per-tile counters are SHARDED: workgroup b updates shard b % NSHARD.  Device-scope atomics on one address are
// serialised at the memory side (~0.1 us each); a long tile list means hundreds of updates of its counter even
// after wave aggregation, which was what bounded preprocess and scatter.  Eight shards cut each chain by eight;
// the shard is a function of the Gaussian index only (same in both kernels), never of where a block runs.
constexpr int NSHARD = 8;
// Counters are stored SHARD-MAJOR, [shard][tile]: the eight shards of one tile used to share 32 bytes, so eight
// workgroups on eight XCDs bounced one cache line between them (the atomics alone were half of preprocess: 31.6 us
// with, 15.3 us without); shard-major keeps a line to one shard -- and a shard to one XCD under round-robin dispatch.
// The per-shard row is padded to a multiple of 64 tiles so that rows stay 256-byte aligned.
__host__ __device__ inline size_t shard_stride(int T) { return ((size_t)T + 63) & ~(size_t)63; }

struct ImageState {          // per pixel / per tile
    float* final_T;          // [H*W]
    uint32_t* n_contrib;     // [H*W] 1-based position in the tile list of the last blended instance
    uint2* ranges;           // [T] {start, end} into point_list
    uint32_t* tile_count;    // [NSHARD][Tp] instances per (shard, tile) (atomics in preprocess), Tp = shard_stride(T)
    uint32_t* tile_cursor;   // [NSHARD][Tp] scatter cursors: a tile's bucket is the concatenation of its shards
    uint32_t* totals;        // [8] {R, max tile count, number of non-empty tiles, U = number of list segments,
                             //      token of the view whose preprocess could not record every instance (scatter then
                             //      walks the tiles again), token of the current view, number of parts of long lists, -}
    uint32_t* order;         // [T] tile ids, longest instance lists first (32-entry buckets), empty tiles last:
                             // the blockIdx -> tile map of the per-tile kernels (longest-processing-time-first
                             // dispatch evens out the very uneven per-tile work of a surface seen in perspective)
    uint32_t* seg_off;       // [T+1] exclusive prefix of ceil(list length / SEG): first segment (unit) id of a tile
    size_t bytes;
};
inline ImageState carve_image(void* base, int W, int H)
{
    ImageState s; size_t o = 0; char* b = (char*)base;
    const size_t N = (size_t)W * H; const size_t T = (size_t)tiles_of(W, H).T;
    s.final_T = (float*)(b + o); o = align_up(o + 4 * N);
    s.n_contrib = (uint32_t*)(b + o); o = align_up(o + 4 * N);
    s.ranges = (uint2*)(b + o); o = align_up(o + 8 * T);
    s.tile_count = (uint32_t*)(b + o); o = align_up(o + 4 * shard_stride((int)T) * NSHARD);
    s.tile_cursor = (uint32_t*)(b + o); o = align_up(o + 4 * shard_stride((int)T) * NSHARD);
    s.totals = (uint32_t*)(b + o); o = align_up(o + 32);
    s.order = (uint32_t*)(b + o); o = align_up(o + 4 * T);
    s.seg_off = (uint32_t*)(b + o); o = align_up(o + 4 * (T + 1));
    s.bytes = o + 256;
    return s;
}

// A tile's depth-sorted list is cut into SEGMENTS of SEG = 64 instances (UNITS when counted over all tiles, U of them);
// a unit carries one 64-bit candidate word per pixel of its tile (gsr_mask.h) and one snapshot of the pixel's running
// (T, C), and is the work item of the backward blend.
constexpr int SEG = 64;

// List positions staged in LDS at a time by the forward blend (gsr_blend_fwd.hip).  A wave's trip count per chunk is
// the largest per-pixel candidate count within the chunk, so longer chunks mean fewer trips (config C: 45 trips per 8x8
// block at 256, 36 at 512; measured 0.088 -> 0.071 ms) at the price of LDS = resident workgroups.
#ifndef GSR_FWD_CHUNK
#define GSR_FWD_CHUNK 512
#endif
constexpr int FWD_CHUNK = GSR_FWD_CHUNK;
// The backward blend treats each (tile, SNAP_SEG list positions, 8x8 block) as an independent unit (bounded size: the
// dispatcher can balance them and nothing carries a long serial chain).  A pixel that blended instances beyond its unit
// resumes from a snapshot of its running (T, C): the forward blend stores one whenever a pixel moves on to a word of a
// new segment -- its state before the first candidate of that segment, which is also its state at every segment
// boundary it skipped since its last word.
#ifndef GSR_SNAP_SEG
#define GSR_SNAP_SEG 64
#endif
constexpr int SNAP_SEG = GSR_SNAP_SEG;
static_assert(SNAP_SEG % 64 == 0 && FWD_CHUNK % SNAP_SEG == 0, "segments are whole mask units and divide a chunk");

// Channel count of a render: 3 = the reference's NUM_CHANNELS (cuda_rasterizer/config.h:15); 6 = two targets
// sharing geometry, blended in one walk; 4 = RGB + one scalar target.  A snapshot is (T, C[0..C-1]) padded to whole float4s.
__host__ __device__ constexpr int snap_vecs(int C) { return (C + 4) / 4; }
__host__ __device__ constexpr bool channels_ok(int C) { return C == 3 || C == 4 || C == 6; }
constexpr int GRAD_RS = 12;  // floats per record of the backward accumulation table: 6 geometric moments + C <= 6 colours

__host__ __device__ constexpr size_t rec_tail_bytes(int C) { return C == 6 ? 16 : C == 4 ? 8 : 4; }   // == sizeof(RecTail<C>)

struct BinState {            // per instance / per segment
    uint64_t* keys;          // [R] (depth_bits << 32) | gaussian, bucketed by tile, unsorted within the bucket
    uint32_t* point_list;    // [R] gaussian ids, tile-major, depth-ascending, ties by ascending id
    float4* rec_a;           // [R] the forward's staged instance records in list order (make_rec: {x, y, a', b'},
    float4* rec_b;           // [R] {c', opacity, colour 0, colour 1}, conic in the exp2 domain), written by blend_fwd
    void* rec_c;             // [R] RecTail<C>: colours 2 .. C-1       for the backward's units (contiguous reads)
    uint4* unit_info;        // [U] {tile, first list entry of the tile, entries of the tile, first unit of the tile}
    uint2* masks;            // [U][4 blocks][64 lanes] per-pixel 64-bit words over the unit's 64 list positions
                             // {positions 0-31, positions 32-63}; block = 2*by + bx, lane = 8*(y % 8) + (x % 8)
    uint2* part_list;        // [part_capacity] {tile, first list position} of every part of a list blended in parts
    uint32_t* part_ticket;   // [part_capacity] at a tile's FIRST part: how many of its parts have finished (scatter zeroes it)
    uint32_t* part_last;     // [part_capacity][256] a part's per-pixel last contributor (bit 31: the part stopped)
    float4* part_fin;        // [part_capacity][256][snap_vecs(C)] a part's per-pixel {T, C...} from T = 1, C = 0
    float4* snap;            // [U][256][snap_vecs(C)] per-pixel {T, C0, C1, ...} BEFORE the first instance of segment
                             // u (u not the first segment of its tile; that slot holds the FINAL {T, C...} when the
                             // tile has more than one segment); pixel index = 16*(y - tile_y0) + (x - tile_x0)
    size_t bytes;
};
// Lists above LONG_LIST entries are sorted by kernels of their own (the forward blend sorts shorter ones itself, gsr_sort.h).
// (GSR_LONG_LIST: a build with a huge value walks every list serially -- the comparison build of the tests' tools.)
#ifndef GSR_LONG_LIST
#define GSR_LONG_LIST 2048
#endif
constexpr uint32_t LONG_LIST = GSR_LONG_LIST;
static_assert(LONG_LIST >= 2048, "lists up to 2 048 entries are sorted inside the forward kernel, whole");
// In a view that SPLITS, every list above PART_FROM entries is blended in PARTS of one forward chunk each, by workgroups of their
// own inside the forward blend's launch; the part that finishes last combines them (gsr_blend_fwd.hip).  part_capacity bounds
// the number of part slots (a part's slot: (first unit of the part / 8) + (first list entry of its tile / PART_FROM), see there).
#ifndef GSR_PART_FROM
#define GSR_PART_FROM 1024
#endif
constexpr uint32_t PART_FROM = GSR_PART_FROM > GSR_LONG_LIST ? GSR_LONG_LIST : GSR_PART_FROM;
static_assert(PART_FROM >= 512, "a part is one forward chunk");
__host__ __device__ inline size_t part_capacity(int R, int U) { return (size_t)U / 8 + (size_t)R / PART_FROM + 2; }
// A view splits only if its longest list would otherwise hold the kernel up (a pixel's walk is serial): the list has more than
// SPLIT_FROM entries AND more than 1/256 of all the view's instances -- then its one workgroup is still walking when the chip's
// 1 024 workgroup slots have finished everything else (config B: two 2 100-entry pole tiles of 368 k instances ran alone for
// 60 us of a 114 us launch).  Config C's longest lists (1 300 - 2 060 of ~800 k) end with the chip still busy, and their tiles
// are mostly OPAQUE: split, every part past a pixel's termination is wasted work and the terminating part is re-walked pixel by
// pixel from memory (measured: +140 us per split view) -- such views are not split.  0xffffffff: nothing is split.
// GSR_SPLIT_FROM overrides the first condition and drops the second (tests: 0 = every view with a list above PART_FROM).
constexpr uint32_t SPLIT_FROM = 1792;
inline uint32_t split_from()
{
    static const uint32_t from = getenv("GSR_SPLIT_FROM") ? ((uint32_t)atoi(getenv("GSR_SPLIT_FROM")) | 0x80000000u) : SPLIT_FROM;
    return from;   // (bit 31: set by the environment -- the share-of-R condition is off)
}
__host__ __device__ inline uint32_t split_threshold_from(uint32_t max_count, uint32_t R, uint32_t from)
{
    const bool forced = (from & 0x80000000u) != 0u;
    from &= 0x7fffffffu;
    const bool dominates = forced || (unsigned long long)max_count * 256ull > (unsigned long long)R;
    return max_count > from && max_count > PART_FROM && dominates ? PART_FROM : 0xffffffffu;
}
inline uint32_t split_threshold(uint32_t max_count, uint32_t R) { return split_threshold_from(max_count, R, split_from()); }
__host__ __device__ inline BinState carve_bin(void* base, int R, int U, int C = 3)
{
    // (everything up to and including `masks` sits at offsets that do not depend on C: the debug exports carve with C = 3)
    BinState s; size_t o = 0; char* b = (char*)base;
    s.keys = (uint64_t*)(b + o); o = align_up(o + 8 * (size_t)R);
    s.point_list = (uint32_t*)(b + o); o = align_up(o + 4 * (size_t)R);
    s.rec_a = (float4*)(b + o); o = align_up(o + 16 * (size_t)R);
    s.rec_b = (float4*)(b + o); o = align_up(o + 16 * (size_t)R);
    s.unit_info = (uint4*)(b + o); o = align_up(o + 16 * (size_t)U);
    s.masks = (uint2*)(b + o); o = align_up(o + sizeof(uint2) * 256 * (size_t)U);
    const size_t np = part_capacity(R, U);
    s.part_list = (uint2*)(b + o); o = align_up(o + sizeof(uint2) * np);
    s.part_ticket = (uint32_t*)(b + o); o = align_up(o + 4 * np);
    s.part_last = (uint32_t*)(b + o); o = align_up(o + 4 * 256 * np);
    s.part_fin = (float4*)(b + o); o = align_up(o + sizeof(float4) * 256 * (size_t)snap_vecs(C) * np);
    s.snap = (float4*)(b + o); o = align_up(o + sizeof(float4) * 256 * (size_t)snap_vecs(C) * (size_t)U);
    s.rec_c = (void*)(b + o); o = align_up(o + rec_tail_bytes(C) * (size_t)R);
    s.bytes = o + 256;
    return s;
}

// ---------------------------------------------------------------- planned binning (gsr_forward_planned)
// A PLAN fixes where every tile's bucket lies BEFORE the view is rendered: {first entry, capacity} per tile, the first UNIT of
// the tile (capacity / 64 units each) and a launch order -- built from the exact tile counts of an earlier view of the same
// camera (gsr_plan.h: one extra workgroup of that view's forward blend).  With a plan the forward has neither scan nor scatter nor a host
// round trip on its critical path: preprocess claims bucket slots with one returning atomic per (workgroup, tile) on the
// tile's cursor and writes the sort keys itself; a tile's list is [first, first + cursor) and the forward blend reads the
// cursor.  Units are numbered in BUCKET space (a tile owns capacity / 64 consecutive unit ids whatever its count turns out to
// be), so masks / snapshots / the backward's unit table need no prefix sum over the actual counts either; a unit past its
// tile's last entry is an empty work item of the backward (one scalar load, then the wave leaves).
// A view that does not fit its plan (a bucket overflows, a workgroup runs out of table or record space) raises the plan's
// flag; the forward blend behind it then does nothing but report that to the host -- which waits for this verdict only, written
// by the blend's first workgroup as it starts -- and the host renders the view the exact way (scan + scatter) and rebuilds the
// plan from it.
// Only views whose longest list stays in the forward blend's own sort (<= 2 048 entries, not split) are planned.
constexpr int PLAN_HDR = 16;              // header words: {valid, T, R_cap, U_cap, max_cap, non-empty tiles of the source view, R of it, -}
#ifndef GSR_PLAN_CURSOR_STRIDE
#define GSR_PLAN_CURSOR_STRIDE 32
#endif
// words between the cursors of neighbouring tiles: ONE cursor per 128-byte line.  Memory-side atomics serialise per line: config C's
// planned preprocess (190 k returning atomics, 23 per tile, all workgroups in that phase at once) takes 44.0 / 32.3 / 31.5 / 26.1 / 25.4 us
// at 2 / 8 / 16 / 32 / 64 words (HIP events, same box)
constexpr int PLAN_CURSOR_STRIDE = GSR_PLAN_CURSOR_STRIDE;
constexpr uint32_t PLAN_MAX_LIST = 2048;  // == SORT_SMALL_CAP: a planned list is sorted inside the forward blend
struct PlanState {
    uint32_t* header;        // [PLAN_HDR]
    uint2* ranges;           // [T] {first entry of the tile's bucket, capacity (a multiple of 64, 0: the tile may hold nothing)}
    uint32_t* seg_off;       // [T+1] first unit of the tile's bucket (exclusive prefix of capacity / 64)
    uint32_t* order;         // [T] launch order of the tiles (the source view's: longest lists first)
    size_t bytes;
};
// The caller's plan buffer holds TWO plans (round 6): the one the current view is binned by and the one that view's forward blend
// builds for the camera's next view (gsr_plan.h) -- a planned view re-plans from its own tile counts, so a plan is never older
// than one visit.  half = 0 / 1; bytes = the whole buffer (gsr_plan_bytes).
inline PlanState carve_plan(void* base, int W, int H, int half = 0)
{
    PlanState s; size_t o = 0;
    const size_t T = (size_t)tiles_of(W, H).T;
    const size_t o_hdr = o; o = align_up(o + 4 * PLAN_HDR);
    const size_t o_rng = o; o = align_up(o + 8 * T);
    const size_t o_seg = o; o = align_up(o + 4 * (T + 1));
    const size_t o_ord = o; o = align_up(o + 4 * T);
    const size_t half_bytes = o + 256;
    char* b = (char*)base + (half ? half_bytes : 0);
    s.header = (uint32_t*)(b + o_hdr);
    s.ranges = (uint2*)(b + o_rng);
    s.seg_off = (uint32_t*)(b + o_seg);
    s.order = (uint32_t*)(b + o_ord);
    s.bytes = 2 * half_bytes;
    return s;
}
// Words the planned forward keeps in the library's per-stream block behind the tile counters: the flag word (at
// 9 * PLAN_SYNC_STRIDE: the token of the last view that did not fit its plan; tokens are never 0 and never repeat).
constexpr int PLAN_SYNC_STRIDE = 32;
constexpr int PLAN_SYNC_WORDS = PLAN_SYNC_STRIDE * 10;
struct PlanRun {             // what the planned kernels get besides the exact path's arguments (by value)
    const uint2* ranges;     // plan
    const uint32_t* seg_off;
    const uint32_t* order;
    uint32_t* cursor;        // [T * PLAN_CURSOR_STRIDE] entries claimed per tile (library block, zero before preprocess; 1 MB at 1080p)
    uint32_t* cursor_other;  // the library's SECOND cursor block: the view before this one claimed there; this view's forward blend
                             //   hands it back zeroed tile by tile and leaves its own counts standing (the plan job reads them)
    uint32_t* sync;          // [PLAN_SYNC_WORDS] tickets + flag (library block)
    uint64_t* keys;          // the binning buffer's key array (capacity R_cap)
    uint4* unit_info;        // the binning buffer's unit table (capacity U_cap): written by the forward blend, tile by tile
    uint2* im_ranges;        // the image state's ranges / seg_off: the forward blend leaves {first, first + count} and the
    uint32_t* im_seg_off;    //   tile's first unit there as the exact path's scan would (debug exports read them)
    uint32_t* host_pad;      // pinned, device-mapped: {verdict (1 fits, 2 does not), sequence number}, written by the forward blend
    uint32_t host_seq;
    uint32_t token;          // this view's token
};
void launch_preprocess_planned(int P, int D, int M, const float* means3D, const float* shs, const float* colors_precomp,
                               const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                               const float* cov3D_precomp, const float* view, const float* proj, const float* campos, int W,
                               int H, float tan_fovx, float tan_fovy, int* radii, GeomState g, ImageState im, PlanRun plan,
                               hipStream_t st);
void launch_blend_fwd_planned(int C, int W, int H, const float* bg, const float* feats, GeomState g, ImageState im, BinState b,
                              float* out_color, bool keep_masks, void* zero_ptr, size_t zero_bytes, PlanRun plan, hipStream_t st,
                              const struct PlanJob* job = nullptr);

// producers of rasterizer inputs (gsr_producers.hip)
void launch_adam(long long n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float step_size,
                 float one_minus_b1, float b2, float one_minus_b2, float eps, float bc2s, hipStream_t st);
void launch_sh_to_rgb(int P, int D, int M, const float* positions, const float* campos, const float* shs, const float* shs_rest,
                      const float* view, int depth_channels, float* out, const float* densities, float* opacity, hipStream_t st);
void launch_sh_to_rgb_bwd(int P, int D, int M, const float* positions, const float* campos, const float* shs,
                          const float* shs_rest, const float* view, int depth_channels, const float* dL_dout, float* dL_dsh,
                          float* dL_dsh_rest, float* dL_dpos, int accumulate_pos, const float* opacity, const float* dL_dopacity,
                          float* dL_ddensity, hipStream_t st);
void launch_zero_f32(float* p, size_t n, hipStream_t st);
struct AdamTensor { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; long long n; float step_size; unsigned block0; };
constexpr int ADAM_BATCH = 16;
struct AdamBatch { AdamTensor t[ADAM_BATCH]; int count; unsigned blocks; };
void launch_adam_multi(const AdamBatch& b, float one_minus_b1, float b2, float one_minus_b2, float eps, float bc2s, hipStream_t st);

void launch_mesh_gaussians(int F, int G, const float* verts, const long long* faces, const float* bary,
                           const float* raw_scales, const float* raw_complex, float thickness, float min_scale,
                           float max_scale, const float* delta_t, const float* delta_r, float* points, float* scaling,
                           float* quats, float* clear, long long clear_n, hipStream_t st);
void launch_mesh_gaussians_bwd(int F, int G, const float* verts, const long long* faces, const float* bary,
                               const float* raw_scales, const float* raw_complex, float min_scale, float max_scale,
                               const float* delta_r, const float* dL_dpoints, const float* dL_dscaling,
                               const float* dL_dquats, float* dL_dverts, float* dL_draw_scales, float* dL_draw_complex,
                               float* dL_ddelta_t, float* dL_ddelta_r, hipStream_t st);

// image-space losses (gsr_loss.hip)
size_t l1_ssim_workspace_bytes(int C, int H, int W);
void launch_l1_ssim(int C, int H, int W, const float* pred, const long long* pred_strides, const float* gt,
                    const long long* gt_strides, float dssim_factor, void* workspace, float* loss_out, float* grad,
                    const long long* grad_strides, hipStream_t st);
void launch_l1_ssim_grad(int C, int H, int W, const float* pred, const long long* pred_strides, const float* gt,
                         const long long* gt_strides, float dssim_factor, const void* workspace, const float* scale, float* grad,
                         const long long* grad_strides, hipStream_t st);
size_t depth_l1_workspace_bytes();
void launch_depth_l1(int H, int W, const float* pred, const long long* pred_strides, const float* gt,
                     const long long* gt_strides, float max_depth, float depth_factor, float mask_factor,
                     void* workspace, float* loss_out, float* grad, const long long* grad_strides, hipStream_t st);

void launch_depth_l1_grad(int H, int W, const float* pred, const long long* pred_strides, const float* gt,
                          const long long* gt_strides, float max_depth, float depth_factor, float mask_factor,
                          const float* stats, const float* scale, float* grad, const long long* grad_strides, hipStream_t st);
void launch_rgb_depth_loss(int C, int H, int W, const float* pred, const long long* ps, const float* gt, const long long* gs_, float f,
                           void* ws_ssim, int Hd, int Wd, const float* dpred, const long long* dps, const float* dgt,
                           const long long* dgs, float max_depth, float depth_factor, float mask_factor, void* ws_depth,
                           float* out8, hipStream_t st);

void launch_rgb_depth_loss_grad(int C, int H, int W, const float* pred, const long long* ps, const float* gt, const long long* gs_, float f,
                                const void* ws_ssim, int Hd, int Wd, const float* dpred, const long long* dps, const float* dgt,
                                const long long* dgs, float max_depth, float depth_factor, float mask_factor, const float* stats,
                                const float* scale, float* grad, const long long* gstr, float* dgrad, const long long* dgstr,
                                hipStream_t st);

// Optional per-workgroup timeline for tuning (gsr_debug_set_trace): when non-null, the blend kernels store
// {start, end} of every workgroup (100 MHz wall clock) at trace[2*blockIdx] (forward) / trace[2*(T+blockIdx)].
extern uint64_t* g_trace;
extern const uint32_t* g_bwd_order;   // experiments: a launch order of the backward's units (gsr_debug_set_bwd_order)

// ---------------------------------------------------------------- kernel launchers (one per .hip file)
struct Camera {              // passed by value to kernels (lands in SGPRs / kernarg)
    float view[16];
    float proj[16];
    float campos[3];
    float tan_fovx, tan_fovy, focal_x, focal_y;
    int W, H;
};

void launch_preprocess(int P, int D, int M, const float* means3D, const float* shs, const float* colors_precomp,
                       const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                       const float* cov3D_precomp, const float* view, const float* proj, const float* campos, int W,
                       int H, float tan_fovx, float tan_fovy, int* radii, GeomState g, ImageState im, uint32_t view_token,
                       hipStream_t st);
void launch_tile_scan(ImageState im, int T, uint32_t* host_totals, uint32_t host_seq, uint32_t view_token, hipStream_t st);
void launch_scatter(int P, int W, int H, int R, uint32_t max_count, GeomState g, ImageState im, BinState b, hipStream_t st);
// the same launch before the host knows R, U and max_count (gsr_forward_fused): the kernel carves `binning_base` itself from
// the totals the scan left, and does nothing if the carve would not fit `capacity` bytes
void launch_scatter_early(int P, int W, int H, int C, GeomState g, ImageState im, void* binning_base, size_t capacity, hipStream_t st);
// -> true if lists of up to 2 048 entries were left for the forward blend to sort (gsr_sort.h)
bool launch_tile_sort(int W, int H, int R, int U, uint32_t max_count, ImageState im, BinState b, bool blend_sorts_small, hipStream_t st);
// whether launch_tile_sort will launch any kernel for such a view (the profiling bracket is skipped otherwise)
bool tile_sort_launches(int R, uint32_t max_count, bool blend_sorts_small);
// Launch positions [0, front_of_order(R, T)) of `order` hold every tile with 2 017 or more entries: they all fall into
// length class 0, which sits at the front, there are at most R / 2017 of them, and the snake only permutes within
// bands of 256.  Kernels that only concern such tiles are launched over this prefix instead of all T tiles.
inline int front_of_order(int R, int T)
{
    const long long bound = ((long long)(R > 0 ? R : 0) / 2017 + 1 + 255) / 256 * 256;
    return (int)(bound < (long long)T ? bound : (long long)T);
}
struct PlanJob;   // gsr_plan.h
void launch_blend_fwd(int C, int W, int H, int R, int U, uint32_t max_count, const float* bg, const float* feats, GeomState g, ImageState im,
                      BinState b, float* out_color, bool keep_masks, void* zero_ptr, size_t zero_bytes, uint32_t* counters,
                      bool sort_small, hipStream_t st, const PlanJob* job = nullptr, bool* job_rides = nullptr, int num_parts = -1);
void launch_blend_bwd(int C, int W, int H, int U, const float* bg, const float* feats, GeomState g, ImageState im, BinState b,
                      const float* dL_dpix, float* grad_acc, hipStream_t st);
void launch_geom_bwd(int P, int D, int M, const float* means3D, const float* shs, const float* scales,
                     float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* view,
                     const float* proj, const float* campos, int W, int H, float tan_fovx, float tan_fovy,
                     const int* radii, GeomState g, int C, const float* grad_acc, float* dL_dmean2D, float* dL_dopacity,
                     float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                     float* dL_drot, hipStream_t st);
void launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t st);
void preprocess_occupancy(int* exact, int* planned);   // tuning: resident workgroups per CU of the two preprocess kernels

// ---------------------------------------------------------------- shared device math
#ifdef __HIPCC__
// Spherical-harmonics constants (real SH, deg <= 3), as auxiliary.h:22-39.
__device__ constexpr float kSH0 = 0.28209479177387814f;
__device__ constexpr float kSH1 = 0.4886025119029199f;
__device__ constexpr float kSH2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                      -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float kSH3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                      0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                      -0.5900435899266435f};
// degree 4: only the Python-side producer reaches it (gaustar_utils/spherical_harmonics.py:23-33, :162-171); the
// rasterizer's in-kernel SH stops at degree 3 like the reference's (auxiliary.h:22-39)
__device__ constexpr float kSH4[9] = {2.5033429417967046f, -1.7701307697799304f, 0.9461746957575601f,
                                      -0.6690465435572892f, 0.10578554691520431f, -0.6690465435572892f,
                                      0.47308734787878004f, -1.7701307697799304f, 0.6258357354491761f};
constexpr int SH_MAX_BASIS = 25;

struct Vec3 { float x, y, z; };

__device__ __forceinline__ Vec3 load3(const float* p, size_t i) { return Vec3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

// Column-major 4x4 times point (auxiliary.h:58-77).
__device__ __forceinline__ Vec3 xform43(const Vec3 p, const float* m)
{
    return Vec3{m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
__device__ __forceinline__ float xform4w(const Vec3 p, const float* m)
{
    return m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
}

// View-space depth.  The per-tile blend order is a discontinuous function of these bits (near-coplanar
// surface splats tie or differ by one ulp), so the operation order is pinned -- contraction off, explicit
// fma -- to the one the reference's transformPoint4x3(...).z (auxiliary.h:58-66) compiles to in its
// gfx950 build: fma(m2, x, m6*y) + m10*z + m14.
__device__ __forceinline__ float view_depth(const Vec3 p, const float* m)
{
#pragma clang fp contract(off)
    const float yz = m[6] * p.y;
    const float zz = m[10] * p.z;
    return (__builtin_fmaf(m[2], p.x, yz) + zz) + m[14];
}

// Rotation of the RAW quaternion (r,x,y,z) -- callers normalise (forward.cu:127).
__device__ __forceinline__ void quat_R(const float4 q, float R[3][3])
{
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = R S^2 R^T, upper triangle {xx,xy,xz,yy,yz,zz} (forward.cu:118-152).
__device__ __forceinline__ void cov3d_from_scale_rot(const Vec3 s_in, float mod, const float4 q, float c[6])
{
    float R[3][3];
    quat_R(q, R);
    const float s[3] = {mod * s_in.x, mod * s_in.y, mod * s_in.z};
    float M[3][3];   // M = S R^T
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) M[i][j] = s[i] * R[j][i];
    c[0] = M[0][0] * M[0][0] + M[1][0] * M[1][0] + M[2][0] * M[2][0];
    c[1] = M[0][0] * M[0][1] + M[1][0] * M[1][1] + M[2][0] * M[2][1];
    c[2] = M[0][0] * M[0][2] + M[1][0] * M[1][2] + M[2][0] * M[2][2];
    c[3] = M[0][1] * M[0][1] + M[1][1] * M[1][1] + M[2][1] * M[2][1];
    c[4] = M[0][1] * M[0][2] + M[1][1] * M[1][2] + M[2][1] * M[2][2];
    c[5] = M[0][2] * M[0][2] + M[1][2] * M[1][2] + M[2][2] * M[2][2];
}

// EWA projection (forward.cu:74-113 / backward.cu:161-199): rows a0,a1 of J*R_w2c with the
// view-space point clamped to +-1.3*tanfov before the Jacobian is formed.
struct Ewa {
    float a0[3], a1[3];
    Vec3 t;            // clamped view-space mean
    float txtz, tytz;  // unclamped ratios (for the clamp-gradient masks)
};
__device__ __forceinline__ Ewa ewa_rows(const Vec3 mean, const float* view, float fx, float fy, float tan_fovx,
                                         float tan_fovy)
{
    Ewa e;
    Vec3 t = xform43(mean, view);
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    e.txtz = t.x / t.z;
    e.tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, e.txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, e.tytz)) * t.z;
    const float J00 = fx / t.z, J02 = -(fx * t.x) / (t.z * t.z);
    const float J11 = fy / t.z, J12 = -(fy * t.y) / (t.z * t.z);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        e.a0[k] = view[4 * k + 0] * J00 + view[4 * k + 2] * J02;
        e.a1[k] = view[4 * k + 1] * J11 + view[4 * k + 2] * J12;
    }
    e.t = t;
    return e;
}

// cov2D = A Sigma A^T + 0.3 I ; also returns v0 = Sigma a0, v1 = Sigma a1 (needed by backward).
__device__ __forceinline__ void cov2d_from(const Ewa& e, const float c[6], float v0[3], float v1[3], float& a, float& b,
                                            float& cc)
{
    v0[0] = c[0] * e.a0[0] + c[1] * e.a0[1] + c[2] * e.a0[2];
    v0[1] = c[1] * e.a0[0] + c[3] * e.a0[1] + c[4] * e.a0[2];
    v0[2] = c[2] * e.a0[0] + c[4] * e.a0[1] + c[5] * e.a0[2];
    v1[0] = c[0] * e.a1[0] + c[1] * e.a1[1] + c[2] * e.a1[2];
    v1[1] = c[1] * e.a1[0] + c[3] * e.a1[1] + c[4] * e.a1[2];
    v1[2] = c[2] * e.a1[0] + c[4] * e.a1[1] + c[5] * e.a1[2];
    a = e.a0[0] * v0[0] + e.a0[1] * v0[1] + e.a0[2] * v0[2] + 0.3f;
    b = e.a0[0] * v1[0] + e.a0[1] * v1[1] + e.a0[2] * v1[2];
    cc = e.a1[0] * v1[0] + e.a1[1] * v1[1] + e.a1[2] * v1[2] + 0.3f;
}

// Real-SH basis values for direction d (deg <= 4): b[0..(deg+1)^2); b must hold SH_MAX_BASIS entries when deg can be 4.
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float* b)
{
    b[0] = kSH0;
    if (deg > 0) {
        b[1] = -kSH1 * y; b[2] = kSH1 * z; b[3] = -kSH1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = kSH2[0] * xy; b[5] = kSH2[1] * yz; b[6] = kSH2[2] * (2.0f * zz - xx - yy);
            b[7] = kSH2[3] * xz; b[8] = kSH2[4] * (xx - yy);
            if (deg > 2) {
                b[9] = kSH3[0] * y * (3.0f * xx - yy); b[10] = kSH3[1] * xy * z;
                b[11] = kSH3[2] * y * (4.0f * zz - xx - yy); b[12] = kSH3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                b[13] = kSH3[4] * x * (4.0f * zz - xx - yy); b[14] = kSH3[5] * z * (xx - yy);
                b[15] = kSH3[6] * x * (xx - 3.0f * yy);
                if (deg > 3) {   // spherical_harmonics.py:162-171, term for term
                    b[16] = kSH4[0] * xy * (xx - yy);
                    b[17] = kSH4[1] * yz * (3.0f * xx - yy);
                    b[18] = kSH4[2] * xy * (7.0f * zz - 1.0f);
                    b[19] = kSH4[3] * yz * (7.0f * zz - 3.0f);
                    b[20] = kSH4[4] * (zz * (35.0f * zz - 30.0f) + 3.0f);
                    b[21] = kSH4[5] * xz * (7.0f * zz - 3.0f);
                    b[22] = kSH4[6] * (xx - yy) * (7.0f * zz - 1.0f);
                    b[23] = kSH4[7] * xz * (xx - 3.0f * yy);
                    b[24] = kSH4[8] * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
                }
            }
        }
    }
}

// Backward of colour = clamp_min(sum_k basis_k(dir) * sh_k + 0.5, 0), dir = normalize(mean - campos)
// (computeColorFromSH backward, backward.cu:20-139; dnormvdv, auxiliary.h:107-117): writes dL_dsh for all M
// coefficients (zero above the active degree) and ADDS the view-direction term to the mean's gradient.
__device__ __forceinline__ void sh_colour_backward(int D, int M, Vec3 mean, const float* __restrict__ campos,
                                                   const float* __restrict__ sh, const float (&dcol)[3],
                                                   float* __restrict__ dsh, float& gmx, float& gmy, float& gmz)
{
    const float ox = mean.x - campos[0], oy = mean.y - campos[1], oz = mean.z - campos[2];
    const float inv = 1.0f / sqrtf(ox * ox + oy * oy + oz * oz);
    const float x = ox * inv, y = oy * inv, z = oz * inv;
    float basis[SH_MAX_BASIS];
    sh_basis(D, x, y, z, basis);
    const int nb = (D + 1) * (D + 1);
    // recompute the clamp decision of the forward (forward.cu:63-70)
    float col[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < nb; k++) { col[0] += basis[k] * sh[3 * k]; col[1] += basis[k] * sh[3 * k + 1]; col[2] += basis[k] * sh[3 * k + 2]; }
    float dRGB[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) dRGB[ch] = (col[ch] + 0.5f < 0.f) ? 0.f : dcol[ch];
    // d(colour)/d(direction)
    float ddx = 0.f, ddy = 0.f, ddz = 0.f;
#define SHD(k) (sh[3 * (k)] * dRGB[0] + sh[3 * (k) + 1] * dRGB[1] + sh[3 * (k) + 2] * dRGB[2])
    if (D > 0) {
        ddx = -kSH1 * SHD(3); ddy = -kSH1 * SHD(1); ddz = kSH1 * SHD(2);
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            const float s4 = SHD(4), s5 = SHD(5), s6 = SHD(6), s7 = SHD(7), s8 = SHD(8);
            ddx += kSH2[0] * y * s4 + kSH2[2] * 2.f * -x * s6 + kSH2[3] * z * s7 + kSH2[4] * 2.f * x * s8;
            ddy += kSH2[0] * x * s4 + kSH2[1] * z * s5 + kSH2[2] * 2.f * -y * s6 + kSH2[4] * 2.f * -y * s8;
            ddz += kSH2[1] * y * s5 + kSH2[2] * 2.f * 2.f * z * s6 + kSH2[3] * x * s7;
            if (D > 2) {
                const float s9 = SHD(9), s10 = SHD(10), s11 = SHD(11), s12 = SHD(12), s13 = SHD(13), s14 = SHD(14),
                            s15 = SHD(15);
                ddx += kSH3[0] * s9 * 3.f * 2.f * xy + kSH3[1] * s10 * yz + kSH3[2] * s11 * -2.f * xy +
                       kSH3[3] * s12 * -3.f * 2.f * xz + kSH3[4] * s13 * (-3.f * xx + 4.f * zz - yy) +
                       kSH3[5] * s14 * 2.f * xz + kSH3[6] * s15 * 3.f * (xx - yy);
                ddy += kSH3[0] * s9 * 3.f * (xx - yy) + kSH3[1] * s10 * xz + kSH3[2] * s11 * (-3.f * yy + 4.f * zz - xx) +
                       kSH3[3] * s12 * -3.f * 2.f * yz + kSH3[4] * s13 * -2.f * xy + kSH3[5] * s14 * -2.f * yz +
                       kSH3[6] * s15 * -3.f * 2.f * xy;
                ddz += kSH3[1] * s10 * xy + kSH3[2] * s11 * 4.f * 2.f * yz + kSH3[3] * s12 * 3.f * (2.f * zz - xx - yy) +
                       kSH3[4] * s13 * 4.f * 2.f * xz + kSH3[5] * s14 * (xx - yy);
                if (D > 3) {   // partial derivatives of the nine degree-4 polynomials as the reference writes them (:162-171)
                    const float s16 = SHD(16), s17 = SHD(17), s18 = SHD(18), s19 = SHD(19), s20 = SHD(20), s21 = SHD(21),
                                s22 = SHD(22), s23 = SHD(23), s24 = SHD(24);
                    const float z7m1 = 7.f * zz - 1.f, z7m3 = 7.f * zz - 3.f, z21m3 = 21.f * zz - 3.f;
                    ddx += kSH4[0] * s16 * y * (3.f * xx - yy) + kSH4[1] * s17 * 6.f * xy * z + kSH4[2] * s18 * y * z7m1 +
                           kSH4[5] * s21 * z * z7m3 + kSH4[6] * s22 * 2.f * x * z7m1 + kSH4[7] * s23 * 3.f * z * (xx - yy) +
                           kSH4[8] * s24 * 4.f * x * (xx - 3.f * yy);
                    ddy += kSH4[0] * s16 * x * (xx - 3.f * yy) + kSH4[1] * s17 * 3.f * z * (xx - yy) + kSH4[2] * s18 * x * z7m1 +
                           kSH4[3] * s19 * z * z7m3 - kSH4[6] * s22 * 2.f * y * z7m1 - kSH4[7] * s23 * 6.f * xy * z +
                           kSH4[8] * s24 * 4.f * y * (yy - 3.f * xx);
                    ddz += kSH4[1] * s17 * y * (3.f * xx - yy) + kSH4[2] * s18 * 14.f * xy * z + kSH4[3] * s19 * y * z21m3 +
                           kSH4[4] * s20 * z * (140.f * zz - 60.f) + kSH4[5] * s21 * x * z21m3 + kSH4[6] * s22 * 14.f * z * (xx - yy) +
                           kSH4[7] * s23 * x * (xx - 3.f * yy);
                }
            }
        }
    }
#undef SHD
    // through the normalisation of the view direction (auxiliary.h:107-117)
    const float sum2 = ox * ox + oy * oy + oz * oz;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    gmx += ((sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * invsum32;
    gmy += (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * invsum32;
    gmz += (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * invsum32;
    // coefficient gradients last: every read of sh is done, so dsh may be the same (LDS-staged) row as sh
    for (int k = 0; k < M; k++) {
        const float bk = k < nb ? basis[k] : 0.f;
        dsh[3 * k] = bk * dRGB[0]; dsh[3 * k + 1] = bk * dRGB[1]; dsh[3 * k + 2] = bk * dRGB[2];
    }
}

// ---- coalesced access to per-Gaussian SH rows.  A thread that reads "its" 3M floats directly touches 64 different
// cache lines per load instruction (rows are 192 B apart for M = 16): SH-mode preprocess ran 8x slower per Gaussian
// than colour mode.  Instead a wave copies the contiguous 64 x 3M floats of its 64 Gaussians through LDS with
// lane-contiguous loads / stores; LDS rows are padded to an odd stride so lane-per-row access is conflict-free.
__host__ __device__ inline int sh_row_stride(int M) { return (3 * M) | 1; }
__host__ __device__ inline size_t sh_stage_bytes(int M, int waves) { return (size_t)waves * 64 * sh_row_stride(M) * sizeof(float); }
// global [g_base + g][3M] -> lds[g * stride + k], g = 0..63 (rows beyond P are left untouched)
// (general form: rows of W3 floats in global memory land at column col0 of LDS rows RS floats apart -- the coefficients
// of a Gaussian may live in two arrays, SuGaR's `_sh_coordinates_dc` [P,1,3] and `_sh_coordinates_rest` [P,M-1,3])
__device__ __forceinline__ void sh_stage_load_cols(float* __restrict__ lds, const float* __restrict__ src, size_t g_base, int P,
                                                   int W3, int RS, int col0, int lane)
{
    if (W3 <= 0) return;
    lds += col0;
    const long long rows = (long long)P - (long long)g_base;
    const int n_rows = (int)(rows >= 64 ? 64 : (rows > 0 ? rows : 0));
    const float* s = src + g_base * (size_t)W3;
    if ((W3 & 3) == 0) {   // rows are whole float4s (M = 4, 16): 16-byte loads, a row never shares one with its neighbour
        const int V = W3 >> 2, n_vec = n_rows * V;
        const float4* s4 = reinterpret_cast<const float4*>(s);   // g_base is a multiple of 64: 16-byte aligned
        int g = lane / V, k = lane - g * V;
        const int dg = 64 / V, dk = 64 - dg * V;
        // every 16-byte load of the wave's 64 rows is issued before the first one is parked in LDS (V <= 12 per lane for
        // M <= 16): as a load -> wait -> ds_write loop this was up to twelve dependent trips to memory per wave
        constexpr int MAXV = 12;
        float4 v[MAXV];
#pragma unroll
        for (int i = 0; i < MAXV; i++) {
            const int w = lane + 64 * i;
            if (i < V && w < n_vec) v[i] = s4[w];
        }
#pragma unroll
        for (int i = 0; i < MAXV; i++) {
            const int w = lane + 64 * i;
            if (i < V && w < n_vec) {
                float* d = lds + g * RS + 4 * k;
                d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
            }
            g += dg; k += dk;
            if (k >= V) { k -= V; g++; }
        }
        for (int w = lane + 64 * MAXV; w < n_vec; w += 64) {   // M > 16 (not used by the rasterizer): plain loop
            const float4 vv = s4[w];
            float* d = lds + g * RS + 4 * k;
            d[0] = vv.x; d[1] = vv.y; d[2] = vv.z; d[3] = vv.w;
            g += dg; k += dk;
            if (k >= V) { k -= V; g++; }
        }
        return;
    }
    const int n_words = n_rows * W3;
    int g = lane / W3, k = lane - g * W3;
    int w = lane;
    // same idea for rows that are not whole float4s (M = 1, 9, ...): eight scalar loads in flight per trip
    for (; w + 7 * 64 < n_words; w += 8 * 64) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = s[w + 64 * i];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            lds[g * RS + k] = t[i];
            k += 64;
            while (k >= W3) { k -= W3; g++; }
        }
    }
    for (; w < n_words; w += 64) {
        lds[g * RS + k] = s[w];
        k += 64;
        while (k >= W3) { k -= W3; g++; }
    }
}
__device__ __forceinline__ void sh_stage_load(float* __restrict__ lds, const float* __restrict__ src, size_t g_base, int P, int M,
                                              int lane)
{
    sh_stage_load_cols(lds, src, g_base, P, 3 * M, sh_row_stride(M), 0, lane);
}
__device__ __forceinline__ void sh_stage_store_cols(const float* __restrict__ lds, float* __restrict__ dst, size_t g_base, int P,
                                                    int W3, int RS, int col0, int lane)
{
    if (W3 <= 0) return;
    lds += col0;
    const long long rows = (long long)P - (long long)g_base;
    const int n_rows = (int)(rows >= 64 ? 64 : (rows > 0 ? rows : 0));
    float* d = dst + g_base * (size_t)W3;
    if ((W3 & 3) == 0) {
        const int V = W3 >> 2, n_vec = n_rows * V;
        float4* d4 = reinterpret_cast<float4*>(d);
        int g = lane / V, k = lane - g * V;
        const int dg = 64 / V, dk = 64 - dg * V;
        for (int w = lane; w < n_vec; w += 64) {
            const float* r = lds + g * RS + 4 * k;
            d4[w] = make_float4(r[0], r[1], r[2], r[3]);
            g += dg; k += dk;
            if (k >= V) { k -= V; g++; }
        }
        return;
    }
    const int n_words = n_rows * W3;
    int g = lane / W3, k = lane - g * W3;
    for (int w = lane; w < n_words; w += 64) {
        d[w] = lds[g * RS + k];
        k += 64;
        while (k >= W3) { k -= W3; g++; }
    }
}
__device__ __forceinline__ void sh_stage_store(const float* __restrict__ lds, float* __restrict__ dst, size_t g_base, int P, int M,
                                               int lane)
{
    sh_stage_store_cols(lds, dst, g_base, P, 3 * M, sh_row_stride(M), 0, lane);
}

// Gaussian exponent of one (pixel, splat) pair: power = -0.5*(a dx^2 + c dy^2) - b dx dy (forward.cu:334), evaluated in
// the exp2 domain: the kernels scale the conic ONCE per queued instance (conic_to_exp2) by -0.5*log2(e) / -log2(e), so a
// pair costs two multiplies and two fused steps and its result feeds v_exp_f32 directly (exp(power) = exp2(power*log2 e);
// the reference's __expf is that multiply followed by the same instruction).  sign(result) = sign(power), so the
// reference's `power > 0` skip reads the same.  Explicit fused steps with contraction disabled: the forward and the
// backward blend kernels must round identically (they have to agree on every alpha >= 1/255 decision).
__device__ __forceinline__ void conic_to_exp2(float& ca, float& cb, float& cc)
{
#pragma clang fp contract(off)
    ca *= -0.72134752044448170368f;   // -0.5 * log2(e)
    cb *= -1.44269504088896340736f;   // -log2(e)
    cc *= -0.72134752044448170368f;
}
__device__ __forceinline__ float pair_exp2_arg(float a2, float b2, float c2, float dx, float dy)
{
#pragma clang fp contract(off)
    const float u = __builtin_fmaf(a2, dx, b2 * dy);
    return __builtin_fmaf(c2 * dy, dy, u * dx);
}

// Exact culling primitive: min over the rectangle [X0,X1]x[Y0,Y1] (coordinates relative to the splat
// centre) of  f(x,y) = 0.5*(a x^2 + c y^2) + b x y  = -power.  A (splat, pixel block) pair can only reach
// alpha >= 1/255 if this minimum is <= ln(255*opacity).  f is convex with its minimum (0) at the centre,
// so the constrained minimum sits on one of the two edges facing the centre; both edge candidates are
// points of the rectangle, hence min(fx, fy) is exact in every case (centre inside: both are 0).
// Contraction is off: preprocess (tile counting) and scatter must take bit-identical decisions.
__device__ __forceinline__ float block_min_half_quad(float a, float b, float c, float X0, float X1, float Y0,
                                                     float Y1)
{
#pragma clang fp contract(off)
    const float xc = fminf(fmaxf(0.0f, X0), X1);
    const float yc = fminf(fmaxf(0.0f, Y0), Y1);
    // v_rcp_f32 instead of an IEEE division: any point of the rectangle is a valid (conservative) candidate,
    // so the ulp of the quotient is irrelevant; what matters is that it is one deterministic instruction.
    const float ys = fminf(fmaxf(-(b * xc) * __builtin_amdgcn_rcpf(c), Y0), Y1);
    const float xs = fminf(fmaxf(-(b * yc) * __builtin_amdgcn_rcpf(a), X0), X1);
    const float fx = 0.5f * ((a * xc) * xc + (c * ys) * ys) + (b * xc) * ys;
    const float fy = 0.5f * ((a * xs) * xs + (c * yc) * yc) + (b * xs) * yc;
    return fminf(fx, fy);
}

// Visit every tile of `rect` that the splat can reach (exact test above), one tile per lane per ROUND, with the
// visits of the 64 lanes of a wave grouped by tile id.  Once per round, all lanes (converged) call
//     f(tile, is_leader, group_size, rank, leader_lane)
// tile < 0 for lanes with nothing left; among the lanes that present the same tile exactly one is the leader
// and every lane knows its 0-based rank in the group.  Mesh-ordered surface splats make neighbouring lanes hit
// the same few tiles, so a wave issues ONE atomic instruction per round whose active lanes are the group
// leaders (count = group size) instead of one atomic per (lane, tile).  Must be called by all 64 lanes
// (inactive ones pass an empty rect).  Both callers pass the stored (rect, g0, g1) => identical tile sets.
struct TileVisit { int tile; bool is_leader; int group, rank, leader_lane; };
struct TileWalker {
    ushort4 rect; float px, py, ca, cb, cc, tau; int gx, lane, tx, ty; bool more;
    __device__ __forceinline__ TileWalker(ushort4 r, float px_, float py_, float ca_, float cb_, float cc_, float tau_,
                                          int gx_, int lane_)
        : rect(r), px(px_), py(py_), ca(ca_), cb(cb_), cc(cc_), tau(tau_), gx(gx_), lane(lane_), tx(r.x), ty(r.y),
          more(r.z > r.x && r.w > r.y && tau_ >= 0.0f) {}
    // This lane's next reachable tile, or -1.
    __device__ __forceinline__ int next_tile()
    {
        while (more) {
            const float X0 = (float)(tx * TILE) - px, Y0 = (float)(ty * TILE) - py;
            const bool hit = block_min_half_quad(ca, cb, cc, X0, X0 + (float)(TILE - 1), Y0, Y0 + (float)(TILE - 1)) <= tau;
            const int id = ty * gx + tx;
            if (++tx == rect.z) { tx = rect.x; if (++ty == rect.w) more = false; }
            if (hit) return id;
        }
        return -1;
    }
    // One round: every lane advances to its next reachable tile (or none), lanes are grouped by tile id.
    // Returns false (wave-uniformly) when no lane has a tile left; v.tile < 0 for lanes without one.
    __device__ __forceinline__ bool next_round(TileVisit& v)
    {
        const int cur = next_tile();
        unsigned long long active = __ballot(cur >= 0);
        v.tile = cur; v.is_leader = false; v.group = 0; v.rank = 0; v.leader_lane = lane;
        if (active == 0ull) return false;
        const unsigned long long lt = (1ull << lane) - 1ull;
        while (active) {   // group the lanes by tile value with ballots only (no memory traffic in here)
            const int leader = __ffsll((unsigned long long)active) - 1;
            const int t = __shfl(cur, leader, 64);
            const unsigned long long m = __ballot(cur == t);
            if (cur == t) { v.leader_lane = leader; v.group = __popcll(m); v.rank = __popcll(m & lt); }
            active &= ~m;
        }
        v.is_leader = cur >= 0 && lane == v.leader_lane;
        return true;
    }
};
// Rects above COOP_TILES tiles are not walked by their own lane but by the whole wave, lane k testing tile base + k of the rect:
// one lane testing a giant splat's thousands of tiles one after the other WAS the kernel for views with a few such splats
// (config C under solid-surface scales, sugar_model.py:1230-1232: preprocess 761 us, scatter 1 365 us; real captures always hold
// some).  The same test (block_min_half_quad against tau) on the same tiles as TileWalker, so counts and places agree.
constexpr int COOP_TILES = 32;
__device__ __forceinline__ bool rect_is_big(ushort4 r)
{
    return r.z > r.x && r.w > r.y && ((int)r.z - (int)r.x) * ((int)r.w - (int)r.y) > COOP_TILES;
}
struct CoopSplat {
    int x0, y0, w, n; float px, py, ca, cb, cc, tau;
    // the splat of lane `src`, to every lane of the wave
    __device__ __forceinline__ CoopSplat(ushort4 r, float px_, float py_, float ca_, float cb_, float cc_, float tau_, int src)
    {
        const int rx = __shfl((int)r.x, src, 64), ry = __shfl((int)r.y, src, 64), rz = __shfl((int)r.z, src, 64), rw = __shfl((int)r.w, src, 64);
        x0 = rx; y0 = ry; w = rz - rx; n = w * (rw - ry);
        px = __shfl(px_, src, 64); py = __shfl(py_, src, 64); ca = __shfl(ca_, src, 64); cb = __shfl(cb_, src, 64);
        cc = __shfl(cc_, src, 64); tau = __shfl(tau_, src, 64);
    }
    // this lane's tile of the round starting at rect position `base`, or -1
    __device__ __forceinline__ int tile(int base, int lane, int gx) const
    {
        const int k = base + lane;
        if (k >= n) return -1;
        const int dy = k / w, ty = y0 + dy, tx = x0 + (k - dy * w);
        const float X0 = (float)(tx * TILE) - px, Y0 = (float)(ty * TILE) - py;
        const bool hit = block_min_half_quad(ca, cb, cc, X0, X0 + (float)(TILE - 1), Y0, Y0 + (float)(TILE - 1)) <= tau;
        return hit ? ty * gx + tx : -1;
    }
};
template <class F>
__device__ __forceinline__ void for_each_tile_aggregated(ushort4 rect, float px, float py, float ca, float cb,
                                                         float cc, float tau, int gx, int lane, F f)
{
    TileWalker w(rect, px, py, ca, cb, cc, tau, gx, lane);
    TileVisit v;
    while (w.next_round(v)) f(v.tile, v.is_leader, v.group, v.rank, v.leader_lane);
}

// LDS image of the list instances the forward blend has staged (gsr_blend_fwd.hip), structure-of-arrays so that a
// pixel gathers an instance with two 16-byte reads and one short one:
//   ga[i] = {x, y, conic a, conic b},  gb[i] = {conic c, opacity, colour 0, colour 1},  gc[i] = colour 2 ..
// with the conic in the exp2 domain (conic_to_exp2: the forward and the backward must round identically).
// 36 B per instance for three channels (40 for four, 48 for six).
template <int C> struct RecTail { float c[C - 2]; };
template <> struct __attribute__((aligned(8))) RecTail<4> { float c[2]; };
template <> struct __attribute__((aligned(16))) RecTail<6> { float c[4]; };
static_assert(sizeof(RecTail<3>) == rec_tail_bytes(3) && sizeof(RecTail<4>) == rec_tail_bytes(4) && sizeof(RecTail<6>) == rec_tail_bytes(6), "");
template <int C> struct InstRec { float4 a, b; RecTail<C> t; };
template <int C>
__device__ __forceinline__ InstRec<C> make_rec(float4 a, float4 b, const float (&colour)[C])
{
    float a2 = a.z, b2 = a.w, c2 = b.x;
    conic_to_exp2(a2, b2, c2);
    InstRec<C> r;
    r.a = make_float4(a.x, a.y, a2, b2);
    r.b = make_float4(c2, b.y, colour[0], colour[1]);
#pragma unroll
    for (int ch = 2; ch < C; ch++) r.t.c[ch - 2] = colour[ch];
    return r;
}

// Snapshot record of one pixel: float4s {T, C0, C1, C2}, {C3, C4, C5, 0}, ...
template <int C>
__device__ __forceinline__ void store_snapshot(float4* dst, float T, const float (&c)[C])
{
    float v[4 * snap_vecs(C)];
    v[0] = T;
#pragma unroll
    for (int k = 1; k < 4 * snap_vecs(C); k++) v[k] = k - 1 < C ? c[k - 1 < C ? k - 1 : 0] : 0.f;
#pragma unroll
    for (int k = 0; k < snap_vecs(C); k++) dst[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
}
template <int C>
__device__ __forceinline__ void load_snapshot(const float4* src, float& T, float (&c)[C])
{
    float v[4 * snap_vecs(C)];
#pragma unroll
    for (int k = 0; k < snap_vecs(C); k++) {
        const float4 q = src[k];
        v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
    }
    T = v[0];
#pragma unroll
    for (int k = 0; k < C; k++) c[k] = v[k + 1];
}

// Pixel centre of an NDC coordinate; evaluated in double like auxiliary.h:41-44.
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * S - 1.0) * 0.5); }
#endif  // __HIPCC__

}  // namespace gsr
