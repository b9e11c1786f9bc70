// gsr_api.hip -- the C ABI (include/gsr.h): stage drivers, scratch sizing, error reporting.
// Host-side counterpart of CudaRasterizer::Rasterizer::{forward,backward,markVisible}
// (DGR/cuda_rasterizer/rasterizer_impl.cu:141-153, :198-336, :340-434).
#include "../../include/gsr.h"
#include "gsr_internal.h"
#include "gsr_plan.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <utility>

using namespace gsr;

namespace gsr { uint64_t* g_trace = nullptr; }

namespace {
thread_local std::string g_err;
thread_local uint32_t g_pinned_seq = 0;
thread_local uint32_t* g_pinned = nullptr;
// what stage 1 of this thread's last view told the host besides its three outputs: the number of PARTS the view's long lists
// are blended in (tile_scan counts them); stage 2 of the same view picks the forward blend's launch shape by it
thread_local struct { int R, maxc, nseg, parts; } g_last_stage1 = {-1, -1, -1, -1};
// words of the pad: [0..3] stage-1 totals, [4] their sequence number (tile_scan); [8] verdict of a planned preprocess, [9] its
// sequence number
constexpr int PAD_WORDS = 64, PAD_VERDICT = 8;
// ints of a plan_info block (include/gsr.h): [0..4] = {state, R_cap, U_cap, max_cap, slack level}; [PI_HEADER .. +7] the header of
// a plan as its builder wrote it, [PI_ARRIVED] the builder's sequence number (written behind the header), [PI_PENDING] the
// sequence number the host expects there (0: no plan under way)
constexpr int PI_HEADER = 8, PI_ARRIVED = 16, PI_PENDING = 17;
// [PI_HALF] which half of the caller's plan buffer holds the plan in use (gsr_internal.h carve_plan), [PI_PENDING_HALF] the half
// the builder on its way writes
constexpr int PI_HALF = 5, PI_PENDING_HALF = 6;
static_assert(PI_PENDING < GSR_PLAN_INFO_INTS, "plan_info block");
std::atomic<long long> g_wait_ns{0}, g_waits{0};   // gsr_debug_host_wait   // pinned, device-mapped landing pad for the stage-1 totals (written by tile_scan)

// ---- optional per-kernel timing (gsr_profile_*): HIP events on the launch stream around every stage.
enum Stage { ST_PREPROCESS = 0, ST_TILE_SCAN, ST_SCATTER, ST_TILE_SORT, ST_BLEND_FWD, ST_ZERO_FILL, ST_BLEND_BWD,
             ST_GEOM_BWD, ST_LOSS, ST_PRODUCERS, ST_OPTIM, ST_COUNT };
const char* const kStageNames[ST_COUNT] = {"preprocess_kernel", "tile_scan_kernel", "scatter_kernel",
                                           "tile_sort_kernel", "blend_fwd_kernel", "zero_fill", "blend_bwd_kernel",
                                           "geom_bwd_kernel", "loss_kernels", "producer_kernels", "optimizer_kernels"};
struct Rec { int stage; hipEvent_t a, b; };
// Process-wide (PyTorch runs backward on its own autograd thread), guarded by a mutex.
struct Profiler {
    std::atomic<bool> on{false};
    std::mutex mu;
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get()
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};
Profiler g_prof;
struct Scope {
    int stage; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    Scope(int stage_, hipStream_t st_) : stage(stage_), st(st_)   // stage < 0: no bracket
    {
        if (stage >= 0 && g_prof.on.load(std::memory_order_relaxed)) { a = g_prof.get(); b = g_prof.get(); (void)hipEventRecord(a, st); }
    }
    ~Scope()
    {
        if (a) {
            (void)hipEventRecord(b, st);
            std::lock_guard<std::mutex> lk(g_prof.mu);
            g_prof.recs.push_back(Rec{stage, a, b});
        }
    }
};

int fail(const char* where, hipError_t e)
{
    g_err = std::string(where) + ": " + hipGetErrorString(e);
    return 1;
}
int fail_msg(const char* msg)
{
    g_err = msg;
    return 2;
}
#define GSR_CHECK(expr)                                   \
    do {                                                  \
        hipError_t e__ = (expr);                          \
        if (e__ != hipSuccess) return fail(#expr, e__);   \
    } while (0)
#define GSR_CHECK_LAUNCH(name)                            \
    do {                                                  \
        hipError_t e__ = hipGetLastError();               \
        if (e__ != hipSuccess) return fail(name, e__);    \
    } while (0)
}  // namespace

extern "C" {

int gsr_abi_version(void) { return 16; }

const char* gsr_last_error(void) { return g_err.c_str(); }

size_t gsr_geom_bytes(int P) { return carve_geom(nullptr, P > 0 ? P : 0).bytes; }
size_t gsr_image_bytes(int W, int H) { return carve_image(nullptr, W, H).bytes; }
size_t gsr_binning_bytes_mt(int R, int num_segments, int num_channels)
{
    return carve_bin(nullptr, R > 0 ? R : 0, num_segments > 0 ? num_segments : 0,
                     channels_ok(num_channels) ? num_channels : 3).bytes;
}
size_t gsr_binning_bytes(int R, int num_segments) { return gsr_binning_bytes_mt(R, num_segments, 3); }
size_t gsr_grad_scratch_bytes(int P) { return (size_t)48 * (size_t)(P > 0 ? P : 0) + 256; }
size_t gsr_plan_bytes(int W, int H) { return (W > 0 && H > 0) ? carve_plan(nullptr, W, H).bytes : 0; }

// plan_info blocks: GSR_PLAN_INFO_INTS ints of pinned, device-mapped host memory each (the device writes a plan's header there),
// cut from slabs -- a hipHostMalloc per camera would cost a tenth of a millisecond at every camera's first view.
namespace {
std::mutex g_pi_mu;
std::vector<int*> g_pi_free;
}
int* gsr_plan_info_new(void)
{
    std::lock_guard<std::mutex> lk(g_pi_mu);
    if (g_pi_free.empty()) {
        constexpr int PER_SLAB = 512;
        int* slab = nullptr;
        if (hipHostMalloc((void**)&slab, sizeof(int) * GSR_PLAN_INFO_INTS * PER_SLAB,
                          hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        for (int i = PER_SLAB - 1; i >= 0; i--) g_pi_free.push_back(slab + (size_t)i * GSR_PLAN_INFO_INTS);
    }
    int* p = g_pi_free.back();
    g_pi_free.pop_back();
    memset(p, 0, sizeof(int) * GSR_PLAN_INFO_INTS);
    return p;
}
void gsr_plan_info_free(int* plan_info)
{
    if (!plan_info) return;
    std::lock_guard<std::mutex> lk(g_pi_mu);
    g_pi_free.push_back(plan_info);
}

namespace {
// ---- self-cleaning tile counters (gsr_forward_fused only).  The per-(shard, tile) counters and scatter cursors must be
// zero when preprocess starts; carved from the caller's (fresh) image buffer that costs a fill per view -- a 5 us blit
// plus its dispatch in front of a 23 us kernel.  The fused forward instead keeps them in a library-owned block per
// (device, stream) that the forward blend -- the last kernel of the forward, one workgroup per tile -- hands back zeroed.
// `clean` is host-side bookkeeping of that invariant: it is dropped while a call is in flight and only restored once the
// cleaning kernel has been launched (or the block has been re-filled), so any failure in between costs one fill later.
// `busy` makes a block exclusive to ONE call from acquire_counters() until its stage 2 has been launched (or the call
// has bailed out): a second host thread rendering on the same stream meanwhile gets nullptr and falls back to the
// counters in its own image buffer (one fill per view) instead of interleaving its preprocess with this call's scatter.
struct Counters {
    uint32_t* base = nullptr; size_t words = 0; bool clean = false; std::atomic<bool> busy{false};
    // The block: [PLAN_SYNC_WORDS words of the planned forward (the flag of a view that outgrew its plan)] [tile counters of the
    // exact path] [cursor block 0] [cursor block 1].  The sync words come FIRST: where they lie must not depend on the image
    // size -- a flag left behind at a small image's offset would read as a tile count of a larger image's view.
    // The two cursor blocks (round 6; one cursor per tile, PLAN_CURSOR_STRIDE words apart): a planned view claims on one of them
    // and its forward blend leaves the counts standing for the plan job riding in it, while zeroing the tile's cursor in the
    // OTHER block (the previous planned view's).  cur_dirty[b] = number of leading tiles of block b that may be non-zero --
    // host-side bookkeeping like `clean`: a block is only claimed on when it is 0, and a view of T tiles leaves the other
    // block clean up to T (views of different image sizes on one stream cost a fill now and then, never a wrong count).
    size_t cnt_words = 0, cur_words = 0;
    size_t cur_dirty[2] = {0, 0};
    uint32_t* sync() const { return base; }
    uint32_t* cnt() const { return base ? base + PLAN_SYNC_WORDS : nullptr; }
    uint32_t* cursors(int b) const { return base ? base + PLAN_SYNC_WORDS + cnt_words + (b ? cur_words : 0) : nullptr; }
};
std::mutex g_cnt_mu;
std::map<std::pair<int, hipStream_t>, Counters> g_cnt;

struct CountersLease {   // releases the block on every way out of gsr_forward_fused
    Counters* c;
    ~CountersLease() { if (c) c->busy.store(false, std::memory_order_release); }
};

// -> a zeroed block of at least `words` uint32 for (current device, st), marked in flight and busy; nullptr: use the image buffer
Counters* acquire_counters(hipStream_t st, size_t cnt_words, size_t cur_words)
{
    const size_t words = PLAN_SYNC_WORDS + cnt_words + 2 * cur_words;
    const char* e_own = getenv("GSR_OWN_COUNTERS");   // read per call: tools/ab_env.py flips it inside one process
    const bool off = e_own && e_own[0] == '0';
    int dev = 0;
    if (off || hipGetDevice(&dev) != hipSuccess) return nullptr;
    Counters* c;
    {
        // (the lease is taken under the lock: gsr_release_stream_state tests `busy` under the same lock before it frees
        // the block and erases the node, so a block can never be freed between the lookup and the exchange)
        std::lock_guard<std::mutex> lk(g_cnt_mu);
        c = &g_cnt[std::make_pair(dev, st)];   // std::map: the address stays valid
        if (c->busy.exchange(true, std::memory_order_acquire)) return nullptr;   // another thread's call owns it right now
    }
    if (c->cnt_words < cnt_words || c->cur_words < cur_words) {
        // (a larger image: the regions move, so everything is laid out and filled anew)
        if (c->base) { (void)hipStreamSynchronize(st); (void)hipFree(c->base); }
        c->base = nullptr; c->words = 0; c->clean = false;
        const size_t cw = std::max(cnt_words, c->cnt_words), uw = std::max(cur_words, c->cur_words);
        const size_t all = PLAN_SYNC_WORDS + cw + 2 * uw;
        if (hipMalloc((void**)&c->base, 4 * all) != hipSuccess) {
            (void)hipGetLastError(); c->base = nullptr; c->cnt_words = c->cur_words = 0; c->busy.store(false); return nullptr;
        }
        c->words = all; c->cnt_words = cw; c->cur_words = uw;
    }
    (void)words;
    if (!c->clean) {
        if (hipMemsetAsync(c->base, 0, 4 * c->words, st) != hipSuccess) {
            (void)hipGetLastError(); c->busy.store(false); return nullptr;
        }
        c->cur_dirty[0] = c->cur_dirty[1] = 0;
    }
    c->clean = false;   // in flight
    return c;
}

// Wait for a kernel's word in the pinned pad: normal waits are far below a millisecond and touch nothing but the pad.  The
// stream is only consulted (did it finish or fault?) once a wait has lasted 2 ms, then every 2 ms: hipStreamQuery takes
// runtime locks.  -> 0 and *seen = whether the word showed up (false: the stream retired without it), or an error code.
int wait_pad_word(const volatile uint32_t* flag, uint32_t seq, hipStream_t st, bool* seen)
{
    const auto t_wait0 = std::chrono::steady_clock::now();
    auto next_query = t_wait0 + std::chrono::milliseconds(2);
    for (uint32_t polls = 1; *flag != seq; polls++) {
        if ((polls & 1023u) == 0u && std::chrono::steady_clock::now() >= next_query) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) break;              // kernel retired: its stores are visible
            if (q != hipErrorNotReady) GSR_CHECK(q);   // a fault surfaces here
            next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
        }
        __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    *seen = *flag == seq;
    g_wait_ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_wait0).count(),
                        std::memory_order_relaxed);
    g_waits.fetch_add(1, std::memory_order_relaxed);
    return 0;
}

int ensure_pad()
{
    if (!g_pinned) {
        GSR_CHECK(hipHostMalloc((void**)&g_pinned, 4 * PAD_WORDS, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
        memset(g_pinned, 0, 4 * PAD_WORDS);
    }
    return 0;
}
uint32_t next_seq() { return ++g_pinned_seq ? g_pinned_seq : ++g_pinned_seq; }   // never 0 (the pad's initial value)
uint32_t next_view_token()
{
    // Token of a view (never 0, unique in the process): a preprocess workgroup that cannot record all of its instances
    // stores it in totals[4], the scan stores it in totals[5], and scatter walks the tiles again iff the two are equal.
    // No word has to be cleared between views, and a stale or uninitialised totals[4] can only select the slower path.
    static std::atomic<uint32_t> g_view_token{0};
    uint32_t view_token = ++g_view_token;
    if (view_token == 0) view_token = ++g_view_token;
    return view_token;
}

int check_stage1_args(int P, int D, int M, const float* means3D, const float* shs, const float* colors_precomp,
                      const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                      const float* viewmatrix, const float* projmatrix, const float* campos, int W, int H, const int* radii,
                      const void* geom_buffer, const void* image_buffer)
{
    if ((long long)tiles_of(W > 0 ? W : 1, H > 0 ? H : 1).T > 256ll * 1024)
        return fail_msg("gsr_forward_stage1: image too large (more than 262144 tiles)");
    if (W <= 0 || H <= 0) return fail_msg("gsr_forward_stage1: image size must be positive");
    if (P < 0) return fail_msg("gsr_forward_stage1: negative P");
    if (!image_buffer) return fail_msg("gsr_forward_stage1: image_buffer is null");
    if (P > 0) {
        if (!means3D || !opacities || !viewmatrix || !projmatrix || !campos || !radii || !geom_buffer)
            return fail_msg("gsr_forward_stage1: required pointer is null");
        if ((shs == nullptr) == (colors_precomp == nullptr))
            return fail_msg("gsr_forward_stage1: provide exactly one of shs / colors_precomp");
        if (shs && (D < 0 || D > 3 || (D + 1) * (D + 1) > M))
            return fail_msg("gsr_forward_stage1: sh degree must be 0..3 and fit in M coefficients");
        if ((cov3D_precomp == nullptr) == (scales == nullptr || rotations == nullptr))
            return fail_msg("gsr_forward_stage1: provide exactly one of scales+rotations / cov3D_precomp");
    }
    return 0;
}

int forward_stage1_impl(int P, int D, int M, const float* means3D, const float* shs, const float* colors_precomp,
                        const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                        const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                        const float* campos, int W, int H, float tan_fovx, float tan_fovy, int prefiltered, int* radii,
                        void* geom_buffer, void* image_buffer, int* num_rendered, int* max_tile_instances,
                        int* num_segments, uint32_t* counters, gsr_stream_t stream, void* early_bin = nullptr,
                        size_t early_capacity = 0, int early_C = 3)
{
    (void)prefiltered;   // the reference only uses it to trap on a culled point (auxiliary.h:156-160)
    g_err.clear();
    if (!num_rendered || !max_tile_instances || !num_segments)
        return fail_msg("gsr_forward_stage1: null output pointer");
    *num_rendered = 0;
    *max_tile_instances = 0;
    *num_segments = 0;
    if (const int bad = check_stage1_args(P, D, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                          viewmatrix, projmatrix, campos, W, H, radii, geom_buffer, image_buffer))
        return bad;
    hipStream_t st = (hipStream_t)stream;
    const Tiles t = tiles_of(W, H);
    ImageState im = carve_image(image_buffer, W, H);
    if (counters) {   // library-owned, already zero (acquire_counters)
        im.tile_count = counters;
        im.tile_cursor = counters + (size_t)shard_stride(t.T) * NSHARD;
    } else {
        // counters and cursors are adjacent in the image buffer: one fill covers both
        GSR_CHECK(hipMemsetAsync(im.tile_count, 0,
                                 (size_t)((char*)(im.tile_cursor + shard_stride(t.T) * NSHARD) - (char*)im.tile_count), st));
    }
    const uint32_t view_token = next_view_token();
    if (P > 0) {
        GeomState g = carve_geom(geom_buffer, P);
        {
            Scope sc(ST_PREPROCESS, st);
            launch_preprocess(P, D, M, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                              cov3D_precomp, viewmatrix, projmatrix, campos, W, H, tan_fovx, tan_fovy, radii, g, im, view_token,
                              st);
        }
        GSR_CHECK_LAUNCH("preprocess_kernel");
    }
    // Pinned landing pad, mapped into the device's address space: tile_scan stores the four totals and then this call's
    // sequence number into it.  The host polls the sequence word instead of blocking in hipStreamSynchronize (whose
    // interrupt wake-up costs tens of microseconds of idle GPU per view); every few thousand polls it asks the stream
    // whether it finished or faulted, so a failed kernel ends the wait with its error.  GSR_SYNC_SPIN=0 -> plain sync.
    if (const int bad = ensure_pad()) return bad;
    static const bool spin = !(getenv("GSR_SYNC_SPIN") && atoi(getenv("GSR_SYNC_SPIN")) == 0);
    const uint32_t seq = next_seq();
    {
        Scope sc(ST_TILE_SCAN, st);
        launch_tile_scan(im, t.T, g_pinned, seq, view_token, st);
    }
    GSR_CHECK_LAUNCH("tile_scan_kernel");
    if (early_bin && P > 0) {
        // The fused forward already holds a binning buffer: scatter goes out right behind the scan and resolves its
        // pointers from the totals on the device, instead of idling the GPU for the host's round trip (~4.6 us per view).
        {
            Scope sc(ST_SCATTER, st);
            launch_scatter_early(P, W, H, early_C, carve_geom(geom_buffer, P), im, early_bin, early_capacity, st);
        }
        GSR_CHECK_LAUNCH("scatter_kernel");
    }
    if (spin) {
        bool seen = false;
        if (const int bad = wait_pad_word(g_pinned + 4, seq, st, &seen)) return bad;
        if (!seen) {
            // The stream reports the kernel retired but its store into the pad has not shown up: never observed, but a
            // platform where device stores to mapped host memory are not coherent must not yield stale totals -- fetch
            // them with an ordinary copy.
            GSR_CHECK(hipMemcpyAsync(g_pinned, im.totals, 16, hipMemcpyDeviceToHost, st));
            g_pinned[5] = 0xffffffffu;   // (the number of parts did not arrive either: -1 = let stage 2 decide from the sizes)
            GSR_CHECK(hipStreamSynchronize(st));
        }
    } else {
        GSR_CHECK(hipStreamSynchronize(st));   // the forward's single host sync (cf. rasterizer_impl.cu:281)
    }
    *num_rendered = (int)g_pinned[0];
    *max_tile_instances = (int)g_pinned[1];
    *num_segments = (int)g_pinned[3];
    g_last_stage1 = {*num_rendered, *max_tile_instances, *num_segments, spin ? (int)g_pinned[5] : -1};
    return 0;
}
}  // namespace

int gsr_debug_host_wait(long long* wait_ns, long long* waits, int reset)
{
    if (wait_ns) *wait_ns = g_wait_ns.load(std::memory_order_relaxed);
    if (waits) *waits = g_waits.load(std::memory_order_relaxed);
    if (reset) { g_wait_ns.store(0); g_waits.store(0); }
    return 0;
}

int gsr_forward_stage1(int P, int D, int M, const float* means3D, const float* shs, const float* colors_precomp,
                       const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                       const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                       const float* campos, int W, int H, float tan_fovx, float tan_fovy, int prefiltered, int* radii,
                       void* geom_buffer, void* image_buffer, int* num_rendered, int* max_tile_instances,
                       int* num_segments, gsr_stream_t stream)
{
    return forward_stage1_impl(P, D, M, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                               cov3D_precomp, viewmatrix, projmatrix, campos, W, H, tan_fovx, tan_fovy, prefiltered, radii,
                               geom_buffer, image_buffer, num_rendered, max_tile_instances, num_segments, nullptr, stream);
}

namespace {
int forward_stage2_impl(int P, int R, int max_tile_instances, int num_segments, int num_channels, int W, int H,
                        const float* background, const float* colors_precomp, void* geom_buffer, void* binning_buffer,
                        void* image_buffer, float* out_color, void* grad_scratch, uint32_t* counters, gsr_stream_t stream,
                        bool scattered, const PlanJob* job = nullptr, bool* job_rides = nullptr);
}

namespace {
// One planned attempt (gsr_internal.h "planned binning"): preprocess claims bucket slots and writes the keys, the forward blend
// is queued right behind it, and the host waits for preprocess's verdict only.  *fit = 0: the view did not fit the plan --
// nothing was blended, the library's counter block is clean again once the queued blend has passed, and the caller takes the
// exact path.
int forward_planned_attempt(int P, int D, int M, int C, int need_backward, const float* means3D, const float* shs,
                            const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                            const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                            const float* projmatrix, const float* campos, int W, int H, float tan_fovx, float tan_fovy,
                            const float* background, int* radii, void* geom_buffer, void* image_buffer, void* binning_buffer,
                            void* grad_scratch, float* out_color, void* plan_buffer, const int* plan_info, uint32_t* cursors,
                            uint32_t* cursors_other, uint32_t* sync_words, hipStream_t st, int* fit, const PlanJob* job)
{
    *fit = 0;
    g_err.clear();
    if (const int bad = check_stage1_args(P, D, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                          viewmatrix, projmatrix, campos, W, H, radii, geom_buffer, image_buffer))
        return bad;
    if (!background || !out_color) return fail_msg("gsr_forward_stage2: required pointer is null");
    if (C != 3 && !colors_precomp)
        return fail_msg("gsr_forward_stage2: multi-target renders need precomputed colours [P, num_channels]");
    if (const int bad = ensure_pad()) return bad;
    const Tiles t = tiles_of(W, H);
    ImageState im = carve_image(image_buffer, W, H);
    GeomState g = carve_geom(geom_buffer, P);
    BinState b = carve_bin(binning_buffer, plan_info[1], plan_info[2], C);
    const PlanState pl = carve_plan(plan_buffer, W, H, plan_info[PI_HALF] & 1);
    PlanRun run;
    run.ranges = pl.ranges; run.seg_off = pl.seg_off; run.order = pl.order;
    run.cursor = cursors; run.cursor_other = cursors_other; run.sync = sync_words;
    run.keys = b.keys; run.unit_info = b.unit_info; run.im_ranges = im.ranges; run.im_seg_off = im.seg_off;
    run.host_pad = g_pinned + PAD_VERDICT; run.host_seq = next_seq(); run.token = next_view_token();
    (void)t;
    {
        Scope sc(ST_PREPROCESS, st);
        launch_preprocess_planned(P, D, M, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                                  cov3D_precomp, viewmatrix, projmatrix, campos, W, H, tan_fovx, tan_fovy, radii, g, im, run, st);
    }
    GSR_CHECK_LAUNCH("preprocess_kernel");
    const float* feats = colors_precomp ? colors_precomp : g.rgb;
    {
        Scope sc(ST_BLEND_FWD, st);
        launch_blend_fwd_planned(C, W, H, background, feats, g, im, b, out_color, need_backward != 0,
                                 need_backward ? grad_scratch : nullptr,
                                 need_backward && grad_scratch ? gsr_grad_scratch_bytes(P) : 0, run, st, job);
    }
    GSR_CHECK_LAUNCH("blend_fwd_kernel");
    bool seen = false;
    if (const int bad = wait_pad_word(g_pinned + PAD_VERDICT + 1, run.host_seq, st, &seen)) return bad;
    if (!seen) {   // (see forward_stage1_impl: never observed; the stream has retired, so a plain read-back is final)
        uint32_t flag = 0;
        GSR_CHECK(hipMemcpyAsync(&flag, run.sync + 9 * PLAN_SYNC_STRIDE, 4, hipMemcpyDeviceToHost, st));
        GSR_CHECK(hipStreamSynchronize(st));
        *fit = flag != run.token;
        return 0;
    }
    *fit = g_pinned[PAD_VERDICT] == 1u;
    return 0;
}

int forward_fused_impl(int P, int D, int M, int num_channels, int need_backward, const float* means3D, const float* shs,
                       const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                       const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                       const float* projmatrix, const float* campos, int W, int H, float tan_fovx, float tan_fovy,
                       int prefiltered, const float* background, int* radii, void* geom_buffer, void* image_buffer,
                       void* binning_buffer, size_t binning_capacity, void* grad_scratch, float* out_color,
                       int* num_rendered, int* max_tile_instances, int* num_segments, int* blended, void* plan_buffer,
                       int* plan_info, int* planned, gsr_stream_t stream)
{
    if (!blended) { g_err.clear(); return fail_msg("gsr_forward_fused: null output pointer"); }
    *blended = 0;
    if (planned) *planned = 0;
    if (!channels_ok(num_channels)) { g_err.clear(); return fail_msg("gsr_forward_fused: num_channels must be 3, 4 or 6"); }
    hipStream_t st = (hipStream_t)stream;
    const bool sane = W > 0 && H > 0 && (long long)tiles_of(W > 0 ? W : 1, H > 0 ? H : 1).T <= 256ll * 1024 && image_buffer && P > 0;
    const size_t cnt_words = sane ? 2 * (size_t)shard_stride(tiles_of(W, H).T) * NSHARD : 0;
    const bool keeps_plan = plan_buffer != nullptr && plan_info != nullptr && sane;
    // (a caller that keeps plans: two blocks of one cursor per tile, PLAN_CURSOR_STRIDE words apart, behind the exact counters)
    const size_t cur_words = keeps_plan ? (size_t)tiles_of(W, H).T * PLAN_CURSOR_STRIDE : 0;
    Counters* own = sane ? acquire_counters(st, cnt_words, cur_words) : nullptr;
    CountersLease lease{own};
    if (keeps_plan && plan_info[PI_PENDING] != 0) {
        // The plan an earlier view of this camera left behind: its header arrives in the caller's pinned words [PI_HEADER ..]
        // some tens of microseconds into that view's forward blend; nobody waited for it then.  Normally it is long there.
        bool seen = false;
        if (const int bad = wait_pad_word(reinterpret_cast<volatile uint32_t*>(plan_info) + PI_ARRIVED, (uint32_t)plan_info[PI_PENDING],
                                          st, &seen))
            return bad;
        const uint32_t* h = reinterpret_cast<const uint32_t*>(plan_info) + PI_HEADER;
        // (a header that is not this image's -- another size's, or not a header at all -- is no plan)
        if (seen && (h[1] != (uint32_t)tiles_of(W, H).T || h[3] != h[2] >> 6 || (h[2] & 63u) != 0u)) seen = false;
        plan_info[0] = seen ? (h[0] == 1u ? 1 : -1) : 0;
        if (seen) {
            plan_info[1] = (int)h[2]; plan_info[2] = (int)h[3]; plan_info[3] = (int)h[4];
            plan_info[PI_HALF] = plan_info[PI_PENDING_HALF] & 1;
        }
        plan_info[PI_PENDING] = 0;
    }
    // The plan workgroup of THIS view (exact or planned) writes the half of the plan buffer that is not in use; its header
    // lands in the caller's pinned words and is adopted by the camera's next call (above).
    const auto plan_job = [&](PlanJob& job) {
        const Tiles t = tiles_of(W, H);
        const int half = (plan_info[PI_HALF] & 1) ^ 1;
        const PlanState pl = carve_plan(plan_buffer, W, H, half);
        job.enabled = 1u; job.T = t.T; job.gx = t.gx; job.gy = t.gy;
        job.header = pl.header; job.ranges = pl.ranges; job.seg_off = pl.seg_off; job.order = pl.order;
        job.split_from_word = split_from();
        job.level = (uint32_t)(plan_info[4] < 0 ? 0 : plan_info[4] > 3 ? 3 : plan_info[4]);
        job.host_pad = reinterpret_cast<uint32_t*>(plan_info) + PI_HEADER;
        // (the sequence number belongs to the PLAN, not to the calling thread: a plan_info block may be used from several
        // host threads, and a thread-local counter that happens to equal the block's stale ARRIVED word would make the next
        // call adopt the OLD header while the device holds the new plan)
        job.host_seq = (uint32_t)plan_info[PI_ARRIVED] + 1u;
        if (job.host_seq == 0u) job.host_seq = 1u;
        job.cursor = nullptr;
        // (GSR_PLAN_DIAG builds: the plan being replaced, if the buffer has held one)
        job.prev_ranges = plan_info[1] > 0 ? carve_plan(plan_buffer, W, H, half ^ 1).ranges : nullptr;
        return half;
    };
    const char* e_rp = getenv("GSR_REPLAN");   // (0: round 5's behaviour; read per call: tools/ab_env.py flips it inside one process)
    const bool replan = !(e_rp && e_rp[0] == '0');
    if (keeps_plan && own && plan_info[0] == 1 && plan_info[1] > 0 && plan_info[2] > 0 && binning_buffer &&
        num_rendered && max_tile_instances && num_segments && (split_from() & 0x80000000u) == 0u &&
        gsr_binning_bytes_mt(plan_info[1], plan_info[2], num_channels) <= binning_capacity) {
        int fit = 0;
        // the cursor block this view claims on must be all zero; the other one is handed back zeroed by this view's forward blend
        const size_t Tn = (size_t)tiles_of(W, H).T;
        int blk = own->cur_dirty[0] == 0 ? 0 : own->cur_dirty[1] == 0 ? 1 : -1;
        if (blk < 0) {
            blk = 0;
            GSR_CHECK(hipMemsetAsync(own->cursors(0), 0, 4 * own->cur_dirty[0] * PLAN_CURSOR_STRIDE, st));
            own->cur_dirty[0] = 0;
        }
        PlanJob pjob{};
        int pjob_half = 0;
        if (replan) {
            pjob_half = plan_job(pjob);
            pjob.cursor = own->cursors(blk);
        }
        const int rc = forward_planned_attempt(P, D, M, num_channels, need_backward, means3D, shs, colors_precomp, opacities, scales,
                                               scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, W, H,
                                               tan_fovx, tan_fovy, background, radii, geom_buffer, image_buffer, binning_buffer,
                                               grad_scratch, out_color, plan_buffer, plan_info, own->cursors(blk),
                                               own->cursors(blk ^ 1), own->sync(), st, &fit, replan ? &pjob : nullptr);
        if (rc != 0) return rc;   // (own stays marked dirty: everything is re-filled on its next use)
        own->cur_dirty[blk] = Tn;
        if (own->cur_dirty[blk ^ 1] <= Tn) own->cur_dirty[blk ^ 1] = 0;
        if (fit) {
            if (replan) { plan_info[PI_PENDING] = (int)pjob.host_seq; plan_info[PI_PENDING_HALF] = pjob_half; }
            *num_rendered = plan_info[1];
            *num_segments = plan_info[2];
            *max_tile_instances = plan_info[3];
            *blended = 1;
            *planned = planned ? 1 : 0;
            own->clean = true;   // (the exact path's counters were not touched; the cursor blocks are accounted for above)
            return 0;
        }
        // the view outgrew its plan: the exact path below renders it (the queued blend leaves the block clean) and re-plans, with
        // twice the slack
        plan_info[0] = 0;
        plan_info[4] = plan_info[4] < 3 ? (plan_info[4] < 0 ? 0 : plan_info[4]) + 1 : 3;
        if (planned) *planned = -1;
    }
    static const bool early_ok = !(getenv("GSR_EARLY_SCATTER") && atoi(getenv("GSR_EARLY_SCATTER")) == 0);
    const bool early = early_ok && sane && binning_buffer != nullptr && binning_capacity > 0;
    const int rc = forward_stage1_impl(P, D, M, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                                       cov3D_precomp, viewmatrix, projmatrix, campos, W, H, tan_fovx, tan_fovy, prefiltered,
                                       radii, geom_buffer, image_buffer, num_rendered, max_tile_instances, num_segments,
                                       own ? own->cnt() : nullptr, stream, early ? binning_buffer : nullptr, binning_capacity,
                                       num_channels);
    if (rc != 0) return rc;   // (own stays marked dirty: re-filled on its next use)
    // The plan this view leaves behind is built by one extra workgroup of its forward blend (gsr_plan.h).  Nobody waits for
    // it: the header lands in the caller's pinned plan_info words and is adopted by the next call with this plan (above).
    // Not built when a plan exists (plan_info[0] == 1 was handled above; a misfit has reset it to 0), when the camera is known
    // to be unplannable (-1: the caller asks again by resetting it to 0), when this view's longest list cannot be planned, or
    // when stage 2 is left to the caller.
    PlanJob job{};
    bool job_rides = false;
    int job_half = 0;
    if (keeps_plan && plan_info[0] == 0) {
        if ((uint32_t)*max_tile_instances + 16u > PLAN_MAX_LIST) {
            plan_info[0] = -1;
        } else if (*num_rendered > 0) {
            job_half = plan_job(job);
        }
    }
    if (gsr_binning_bytes_mt(*num_rendered, *num_segments, num_channels) > binning_capacity || (*num_rendered > 0 && !binning_buffer)) {
        // the guess was too small: the caller allocates exactly and runs stage 2 itself -- over the image buffer's own
        // counters, so they get this call's counts (cursors are still zero) and the library's block is filled again
        if (own) {
            ImageState im = carve_image(image_buffer, W, H);
            GSR_CHECK(hipMemcpyAsync(im.tile_count, own->cnt(), 4 * cnt_words / 2, hipMemcpyDeviceToDevice, st));
            GSR_CHECK(hipMemsetAsync(im.tile_cursor, 0, 4 * cnt_words / 2, st));
            GSR_CHECK(hipMemsetAsync(own->cnt(), 0, 4 * cnt_words / 2, st));
            own->clean = true;
        }
        return 0;
    }
    const int rc2 = forward_stage2_impl(P, *num_rendered, *max_tile_instances, need_backward ? *num_segments : -*num_segments,
                                        num_channels, W, H, background, colors_precomp, geom_buffer, binning_buffer,
                                        image_buffer, out_color, need_backward ? grad_scratch : nullptr,
                                        own ? own->cnt() : nullptr, stream, early, job.enabled ? &job : nullptr, &job_rides);
    if (rc2 == 0) {
        *blended = 1;
        if (own) own->clean = true;   // the forward blend zeroes every tile's counters and cursors
    }
    if (rc2 != 0) return rc2;
    if (job_rides) { plan_info[PI_PENDING] = (int)job.host_seq; plan_info[PI_PENDING_HALF] = job_half; }
    return 0;
}
}  // namespace

int gsr_forward_fused(int P, int D, int M, int num_channels, int need_backward, const float* means3D, const float* shs,
                      const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                      const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                      const float* projmatrix, const float* campos, int W, int H, float tan_fovx, float tan_fovy,
                      int prefiltered, const float* background, int* radii, void* geom_buffer, void* image_buffer,
                      void* binning_buffer, size_t binning_capacity, void* grad_scratch, float* out_color,
                      int* num_rendered, int* max_tile_instances, int* num_segments, int* blended, gsr_stream_t stream)
{
    return forward_fused_impl(P, D, M, num_channels, need_backward, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
                              rotations, cov3D_precomp, viewmatrix, projmatrix, campos, W, H, tan_fovx, tan_fovy, prefiltered,
                              background, radii, geom_buffer, image_buffer, binning_buffer, binning_capacity, grad_scratch,
                              out_color, num_rendered, max_tile_instances, num_segments, blended, nullptr, nullptr, nullptr, stream);
}

int gsr_forward_planned(int P, int D, int M, int num_channels, int need_backward, const float* means3D, const float* shs,
                        const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                        const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                        const float* projmatrix, const float* campos, int W, int H, float tan_fovx, float tan_fovy,
                        int prefiltered, const float* background, int* radii, void* geom_buffer, void* image_buffer,
                        void* binning_buffer, size_t binning_capacity, void* grad_scratch, float* out_color,
                        int* num_rendered, int* max_tile_instances, int* num_segments, int* blended, void* plan_buffer,
                        int* plan_info, int* planned, gsr_stream_t stream)
{
    if (!plan_buffer || !plan_info || !planned) { g_err.clear(); return fail_msg("gsr_forward_planned: null plan pointer"); }
    return forward_fused_impl(P, D, M, num_channels, need_backward, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
                              rotations, cov3D_precomp, viewmatrix, projmatrix, campos, W, H, tan_fovx, tan_fovy, prefiltered,
                              background, radii, geom_buffer, image_buffer, binning_buffer, binning_capacity, grad_scratch,
                              out_color, num_rendered, max_tile_instances, num_segments, blended, plan_buffer, plan_info, planned,
                              stream);
}

int gsr_forward_stage2(int P, int R, int max_tile_instances, int num_segments, int W, int H, const float* background,
                       const float* colors_precomp, void* geom_buffer, void* binning_buffer, void* image_buffer,
                       float* out_color, gsr_stream_t stream)
{
    return gsr_forward_stage2_mt(P, R, max_tile_instances, num_segments, 3, W, H, background, colors_precomp,
                                 geom_buffer, binning_buffer, image_buffer, out_color, stream);
}

namespace {
// grad_scratch != nullptr: the backward's accumulation table, cleared by the forward blend on the side (gsr_forward_fused)
int forward_stage2_impl(int P, int R, int max_tile_instances, int num_segments, int num_channels, int W, int H,
                        const float* background, const float* colors_precomp, void* geom_buffer, void* binning_buffer,
                        void* image_buffer, float* out_color, void* grad_scratch, uint32_t* counters, gsr_stream_t stream,
                        bool scattered, const PlanJob* job, bool* job_rides)
{
    g_err.clear();
    if (W <= 0 || H <= 0) return fail_msg("gsr_forward_stage2: image size must be positive");
    if (!background || !out_color || !image_buffer) return fail_msg("gsr_forward_stage2: required pointer is null");
    if (R > 0 && (!binning_buffer || !geom_buffer)) return fail_msg("gsr_forward_stage2: scratch buffer is null");
    if (!channels_ok(num_channels)) return fail_msg("gsr_forward_stage2: num_channels must be 3, 4 or 6");
    if (num_channels != 3 && !colors_precomp)
        return fail_msg("gsr_forward_stage2: multi-target renders need precomputed colours [P, num_channels]");
    const int C = num_channels;
    hipStream_t st = (hipStream_t)stream;
    ImageState im = carve_image(image_buffer, W, H);
    if (counters) {   // library-owned counters of the fused forward; the blend hands them back zeroed
        im.tile_count = counters;
        im.tile_cursor = counters + (size_t)shard_stride(tiles_of(W, H).T) * NSHARD;
    }
    GeomState g = carve_geom(geom_buffer, P > 0 ? P : 0);
    // num_segments < 0: forward-only render (same buffer size as for |num_segments|; the blended-instance words the
    // backward would replay are not written back)
    const bool forward_only = num_segments < 0;
    if (num_segments < 0) num_segments = -num_segments;
    if (R > 0 && num_segments == 0) return fail_msg("gsr_forward_stage2: num_segments of stage 1 is required (it sizes the binning buffer)");
    BinState b = carve_bin(binning_buffer, R > 0 ? R : 0, num_segments, C);
    bool sort_in_blend = false;
    if (R > 0) {
        if (!scattered) {   // (the fused forward has launched it behind the scan already)
            Scope sc(ST_SCATTER, st);
            launch_scatter(P, W, H, R, (uint32_t)(max_tile_instances > 0 ? max_tile_instances : 0), g, im, b, st);
            GSR_CHECK_LAUNCH("scatter_kernel");
        }
        {
            const char* e_fuse = getenv("GSR_SORT_IN_BLEND");   // read per call (tools/ab_env.py); "0": separate sort kernel
            const bool fuse = !(e_fuse && e_fuse[0] == '0');
            // (no event bracket around a stage that launches nothing -- every list of a typical view is sorted inside the
            // forward blend; two back-to-back event records read as ~5 us of "kernel")
            const bool launches = tile_sort_launches(R, (uint32_t)(max_tile_instances > 0 ? max_tile_instances : 0), fuse);
            Scope sc(launches ? ST_TILE_SORT : -1, st);
            sort_in_blend = launch_tile_sort(W, H, R, num_segments, (uint32_t)(max_tile_instances > 0 ? max_tile_instances : 0), im, b, fuse, st);
        }
        GSR_CHECK_LAUNCH("tile_sort_kernel");
    }
    const float* feats = colors_precomp ? colors_precomp : g.rgb;
    const int num_parts = (g_last_stage1.R == R && g_last_stage1.maxc == max_tile_instances && g_last_stage1.nseg == num_segments)
                              ? g_last_stage1.parts : -1;   // (-1: stage 1 of another view, or of another thread)
    {
        Scope sc(ST_BLEND_FWD, st);
        launch_blend_fwd(C, W, H, R > 0 ? R : 0, num_segments, (uint32_t)(max_tile_instances > 0 ? max_tile_instances : 0), background, feats, g, im, b,
                         out_color, !forward_only, grad_scratch,
                         grad_scratch ? gsr_grad_scratch_bytes(P > 0 ? P : 0) : 0, counters, sort_in_blend, st, job, job_rides, num_parts);
    }
    GSR_CHECK_LAUNCH("blend_fwd_kernel");
    return 0;
}
}  // namespace

int gsr_forward_stage2_mt(int P, int R, int max_tile_instances, int num_segments, int num_channels, int W, int H,
                          const float* background, const float* colors_precomp, void* geom_buffer, void* binning_buffer,
                          void* image_buffer, float* out_color, gsr_stream_t stream)
{
    return forward_stage2_impl(P, R, max_tile_instances, num_segments, num_channels, W, H, background, colors_precomp,
                               geom_buffer, binning_buffer, image_buffer, out_color, nullptr, nullptr, stream, false);
}

int gsr_forward(gsr_alloc_fn geometry_buffer, gsr_alloc_fn binning_buffer, gsr_alloc_fn image_buffer, void* alloc_ctx,
                int P, int D, int M, const float* background, int W, int H, const float* means3D, const float* shs,
                const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                const float* campos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color, int* radii,
                int* num_rendered, gsr_stream_t stream)
{
    g_err.clear();
    if (!geometry_buffer || !binning_buffer || !image_buffer) return fail_msg("gsr_forward: null allocator callback");
    if (!num_rendered) return fail_msg("gsr_forward: num_rendered is null");
    void* geom = geometry_buffer(alloc_ctx, gsr_geom_bytes(P));
    void* img = image_buffer(alloc_ctx, gsr_image_bytes(W, H));
    if (!geom || !img) return fail_msg("gsr_forward: allocator callback returned null");
    int R = 0, maxc = 0, nseg = 0;
    int rc = gsr_forward_stage1(P, D, M, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                                cov3D_precomp, viewmatrix, projmatrix, campos, W, H, tan_fovx, tan_fovy, prefiltered,
                                radii, geom, img, &R, &maxc, &nseg, stream);
    if (rc) return rc;
    void* bin = binning_buffer(alloc_ctx, gsr_binning_bytes(R, nseg));
    if (!bin) return fail_msg("gsr_forward: allocator callback returned null");
    rc = gsr_forward_stage2(P, R, maxc, nseg, W, H, background, colors_precomp, geom, bin, img, out_color, stream);
    *num_rendered = R;
    return rc;
}

int gsr_backward(int P, int D, int M, int R, int num_segments, const float* background, int W, int H,
                 const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                 float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                 const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                 const void* geom_buffer, const void* binning_buffer, const void* image_buffer, const float* dL_dpix,
                 void* grad_scratch, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                 float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, gsr_stream_t stream)
{
    return gsr_backward_mt(P, D, M, R, num_segments, 3, background, W, H, means3D, shs, colors_precomp, scales,
                           scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy,
                           radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, grad_scratch, dL_dmean2D,
                           dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, 0, stream);
}

int gsr_backward_mt(int P, int D, int M, int R, int num_segments, int num_channels, const float* background, int W,
                    int H, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* campos, float tan_fovx, float tan_fovy, const int* radii, const void* geom_buffer,
                 const void* binning_buffer, const void* image_buffer, const float* dL_dpix, void* grad_scratch,
                 float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                 float* dL_dsh, float* dL_dscale, float* dL_drot, int grad_scratch_zeroed, gsr_stream_t stream)
{
    g_err.clear();
    if (P <= 0) return 0;
    if (W <= 0 || H <= 0) return fail_msg("gsr_backward: image size must be positive");
    if (!means3D || !viewmatrix || !projmatrix || !campos || !radii || !geom_buffer || !image_buffer || !dL_dpix ||
        !background)
        return fail_msg("gsr_backward: required pointer is null");
    if (!grad_scratch || !dL_dmean2D || !dL_dopacity || !dL_dcolor || !dL_dmean3D || (cov3D_precomp && !dL_dcov3D))
        return fail_msg("gsr_backward: required gradient pointer is null");
    if (shs && !dL_dsh) return fail_msg("gsr_backward: dL_dsh is null in SH mode");
    if (!cov3D_precomp && (!scales || !rotations || !dL_dscale || !dL_drot))
        return fail_msg("gsr_backward: scales/rotations and their gradients are required without cov3D_precomp");
    if (R > 0 && !binning_buffer) return fail_msg("gsr_backward: binning_buffer is null");
    if (R > 0 && num_segments <= 0)
        return fail_msg("gsr_backward: num_segments of the forward is required (a forward-only render cannot be differentiated)");
    if (!channels_ok(num_channels)) return fail_msg("gsr_backward: num_channels must be 3, 4 or 6");
    if (num_channels != 3 && !colors_precomp)
        return fail_msg("gsr_backward: multi-target renders need precomputed colours [P, num_channels]");
    const int C = num_channels;
    hipStream_t st = (hipStream_t)stream;
    ImageState im = carve_image(const_cast<void*>(image_buffer), W, H);
    GeomState g = carve_geom(const_cast<void*>(geom_buffer), P);
    BinState b = carve_bin(const_cast<void*>(binning_buffer), R > 0 ? R : 0, num_segments > 0 ? num_segments : 0, C);
    // Zero the packed moment records (the only atomic targets); every output tensor is written outright
    // by geom_bwd (cf. the nine zeroed tensors of rasterize_points.cu:151-159).
    float* grad_acc = static_cast<float*>(grad_scratch);
    if (!grad_scratch_zeroed) {   // otherwise the forward blend cleared it on the side (gsr_forward_fused)
        Scope sc(ST_ZERO_FILL, st);
        GSR_CHECK(hipMemsetAsync(grad_acc, 0, gsr_grad_scratch_bytes(P), st));
    }
    const float* feats = colors_precomp ? colors_precomp : g.rgb;
    if (R > 0) {
        {
            Scope sc(ST_BLEND_BWD, st);
            launch_blend_bwd(C, W, H, num_segments, background, feats, g, im, b, dL_dpix, grad_acc, st);
        }
        GSR_CHECK_LAUNCH("blend_bwd_kernel");
    }
    {
        Scope sc(ST_GEOM_BWD, st);
        launch_geom_bwd(P, D, M, means3D, shs, cov3D_precomp ? nullptr : scales, scale_modifier,
                        cov3D_precomp ? nullptr : rotations, cov3D_precomp, viewmatrix, projmatrix, campos, W, H,
                        tan_fovx, tan_fovy, radii, g, C, grad_acc, dL_dmean2D, dL_dopacity, dL_dcolor, dL_dmean3D,
                        dL_dcov3D, shs ? dL_dsh : nullptr, cov3D_precomp ? nullptr : dL_dscale, cov3D_precomp ? nullptr : dL_drot,
                        st);
    }
    GSR_CHECK_LAUNCH("geom_bwd_kernel");
    return 0;
}

int gsr_sh_to_rgb(int P, int D, int M, const float* positions, const float* campos, const float* shs, float* rgb,
                  gsr_stream_t stream)
{
    g_err.clear();
    if (P <= 0) return 0;
    if (!positions || !campos || !shs || !rgb) return fail_msg("gsr_sh_to_rgb: required pointer is null");
    if (D < 0 || D > 4 || (D + 1) * (D + 1) > M) return fail_msg("gsr_sh_to_rgb: sh degree must be 0..4 and fit in M coefficients");
    hipStream_t st = (hipStream_t)stream;
    {
        Scope sc(ST_PRODUCERS, st);
        launch_sh_to_rgb(P, D, M, positions, campos, shs, nullptr, nullptr, 0, rgb, nullptr, nullptr, st);
    }
    GSR_CHECK_LAUNCH("sh_to_rgb_kernel");
    return 0;
}

int gsr_sh_to_rgb_backward(int P, int D, int M, const float* positions, const float* campos, const float* shs,
                           const float* dL_drgb, float* dL_dsh, float* dL_dpos, gsr_stream_t stream)
{
    g_err.clear();
    if (P <= 0) return 0;
    if (!positions || !campos || !shs || !dL_drgb || !dL_dsh || !dL_dpos)
        return fail_msg("gsr_sh_to_rgb_backward: required pointer is null");
    if (D < 0 || D > 4 || (D + 1) * (D + 1) > M)
        return fail_msg("gsr_sh_to_rgb_backward: sh degree must be 0..4 and fit in M coefficients");
    hipStream_t st = (hipStream_t)stream;
    {
        Scope sc(ST_PRODUCERS, st);
        launch_sh_to_rgb_bwd(P, D, M, positions, campos, shs, nullptr, nullptr, 0, dL_drgb, dL_dsh, nullptr, dL_dpos, 0, nullptr, nullptr, nullptr, st);
    }
    GSR_CHECK_LAUNCH("sh_to_rgb_bwd_kernel");
    return 0;
}

int gsr_sh_to_rgbd(int P, int D, int M, const float* positions, const float* campos, const float* shs,
                   const float* viewmatrix, int depth_channels, float* colors6, gsr_stream_t stream)
{
    g_err.clear();
    if (depth_channels != 1 && depth_channels != 3) return fail_msg("gsr_sh_to_rgbd: depth_channels must be 1 or 3");
    if (P <= 0) return 0;
    if (!positions || !campos || !shs || !viewmatrix || !colors6) return fail_msg("gsr_sh_to_rgbd: required pointer is null");
    if (D < 0 || D > 4 || (D + 1) * (D + 1) > M) return fail_msg("gsr_sh_to_rgbd: sh degree must be 0..4 and fit in M coefficients");
    hipStream_t st = (hipStream_t)stream;
    {
        Scope sc(ST_PRODUCERS, st);
        launch_sh_to_rgb(P, D, M, positions, campos, shs, nullptr, viewmatrix, depth_channels, colors6, nullptr, nullptr, st);
    }
    GSR_CHECK_LAUNCH("sh_to_rgb_kernel");
    return 0;
}

int gsr_sh_to_rgbd_backward(int P, int D, int M, const float* positions, const float* campos, const float* shs,
                            const float* viewmatrix, int depth_channels, const float* dL_dcolors6, float* dL_dsh,
                            float* dL_dpos, gsr_stream_t stream)
{
    g_err.clear();
    if (depth_channels != 1 && depth_channels != 3) return fail_msg("gsr_sh_to_rgbd_backward: depth_channels must be 1 or 3");
    if (P <= 0) return 0;
    if (!positions || !campos || !shs || !viewmatrix || !dL_dcolors6 || !dL_dsh || !dL_dpos)
        return fail_msg("gsr_sh_to_rgbd_backward: required pointer is null");
    if (D < 0 || D > 4 || (D + 1) * (D + 1) > M)
        return fail_msg("gsr_sh_to_rgbd_backward: sh degree must be 0..4 and fit in M coefficients");
    hipStream_t st = (hipStream_t)stream;
    {
        Scope sc(ST_PRODUCERS, st);
        launch_sh_to_rgb_bwd(P, D, M, positions, campos, shs, nullptr, viewmatrix, depth_channels, dL_dcolors6, dL_dsh, nullptr, dL_dpos, 0, nullptr, nullptr, nullptr, st);
    }
    GSR_CHECK_LAUNCH("sh_to_rgb_bwd_kernel");
    return 0;
}

int gsr_sh_colors_split(int P, int D, int M, const float* positions, const float* campos, const float* sh_dc,
                        const float* sh_rest, const float* viewmatrix, int depth_channels, const float* densities,
                        float* colors, float* opacity, gsr_stream_t stream)
{
    g_err.clear();
    if (viewmatrix ? (depth_channels != 1 && depth_channels != 3) : depth_channels != 0)
        return fail_msg("gsr_sh_colors_split: depth_channels must be 1 or 3 with a view matrix, 0 without");
    if (P <= 0) return 0;
    if (!positions || !campos || !sh_dc || !colors || (M > 1 && !sh_rest)) return fail_msg("gsr_sh_colors_split: required pointer is null");
    if ((densities == nullptr) != (opacity == nullptr)) return fail_msg("gsr_sh_colors_split: densities and opacity go together");
    if (D < 0 || D > 4 || (D + 1) * (D + 1) > M) return fail_msg("gsr_sh_colors_split: sh degree must be 0..4 and fit in M coefficients");
    hipStream_t st = (hipStream_t)stream;
    {
        Scope sc(ST_PRODUCERS, st);
        // (M == 1: the one-array layout IS the dc array)
        launch_sh_to_rgb(P, D, M, positions, campos, sh_dc, M > 1 ? sh_rest : nullptr, viewmatrix, depth_channels, colors, densities,
                         opacity, st);
    }
    GSR_CHECK_LAUNCH("sh_to_rgb_kernel");
    return 0;
}

int gsr_sh_colors_split_backward(int P, int D, int M, const float* positions, const float* campos, const float* sh_dc,
                                 const float* sh_rest, const float* viewmatrix, int depth_channels, const float* dL_dcolors,
                                 const float* opacity, const float* dL_dopacity, float* dL_dsh_dc, float* dL_dsh_rest,
                                 float* dL_dpos, int accumulate_pos, float* dL_ddensities, gsr_stream_t stream)
{
    g_err.clear();
    if (viewmatrix ? (depth_channels != 1 && depth_channels != 3) : depth_channels != 0)
        return fail_msg("gsr_sh_colors_split_backward: depth_channels must be 1 or 3 with a view matrix, 0 without");
    if (P <= 0) return 0;
    if (!positions || !campos || !sh_dc || !dL_dcolors || !dL_dsh_dc || !dL_dpos || (M > 1 && (!sh_rest || !dL_dsh_rest)))
        return fail_msg("gsr_sh_colors_split_backward: required pointer is null");
    if ((opacity == nullptr) != (dL_dopacity == nullptr) || (opacity == nullptr) != (dL_ddensities == nullptr))
        return fail_msg("gsr_sh_colors_split_backward: opacity, dL_dopacity and dL_ddensities go together");
    if (D < 0 || D > 4 || (D + 1) * (D + 1) > M)
        return fail_msg("gsr_sh_colors_split_backward: sh degree must be 0..4 and fit in M coefficients");
    hipStream_t st = (hipStream_t)stream;
    {
        Scope sc(ST_PRODUCERS, st);
        launch_sh_to_rgb_bwd(P, D, M, positions, campos, sh_dc, M > 1 ? sh_rest : nullptr, viewmatrix, depth_channels, dL_dcolors,
                             dL_dsh_dc, M > 1 ? dL_dsh_rest : nullptr, dL_dpos, accumulate_pos ? 1 : 0, opacity, dL_dopacity,
                             dL_ddensities, st);
    }
    GSR_CHECK_LAUNCH("sh_to_rgb_bwd_kernel");
    return 0;
}

int gsr_adam_step_multi(int count, const long long* numel, float* const* params, const float* const* grads,
                        float* const* exp_avgs, float* const* exp_avg_sqs, const double* lrs, double beta1, double beta2,
                        double eps, int step, gsr_stream_t stream)
{
    g_err.clear();
    if (count <= 0) return 0;
    if (!numel || !params || !grads || !exp_avgs || !exp_avg_sqs || !lrs) return fail_msg("gsr_adam_step_multi: required pointer is null");
    if (step < 1) return fail_msg("gsr_adam_step_multi: step counts from 1");
    for (int i = 0; i < count; i++) {
        if (numel[i] <= 0) continue;
        if (!params[i] || !grads[i] || !exp_avgs[i] || !exp_avg_sqs[i]) return fail_msg("gsr_adam_step_multi: a tensor pointer is null");
        if (((uintptr_t)params[i] | (uintptr_t)grads[i] | (uintptr_t)exp_avgs[i] | (uintptr_t)exp_avg_sqs[i]) & 15u)
            return fail_msg("gsr_adam_step_multi: arrays must be 16-byte aligned");
    }
    const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
    hipStream_t st = (hipStream_t)stream;
    {
        Scope sc(ST_OPTIM, st);
        AdamBatch b;
        b.count = 0; b.blocks = 0;
        const auto flush = [&]() {
            if (b.count) launch_adam_multi(b, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)std::sqrt(bc2), st);
            b.count = 0; b.blocks = 0;
        };
        for (int i = 0; i < count; i++) {
            if (numel[i] <= 0) continue;
            // one 16-byte element per thread, 256 threads per workgroup (as gsr_adam_step); a launch takes up to 16 tensors
            const long long want = ((numel[i] >> 2) + 255) / 256;
            const unsigned nb = (unsigned)(want < 1 ? 1 : (want > 256 * 1024 ? 256 * 1024 : want));
            if (b.count == ADAM_BATCH || (unsigned long long)b.blocks + nb > 0x7fffffffull) flush();
            AdamTensor& t = b.t[b.count++];
            t.param = params[i]; t.grad = grads[i]; t.exp_avg = exp_avgs[i]; t.exp_avg_sq = exp_avg_sqs[i];
            t.n = numel[i]; t.step_size = (float)(lrs[i] / bc1); t.block0 = b.blocks;
            b.blocks += nb;
        }
        flush();
    }
    GSR_CHECK_LAUNCH("adam_multi_kernel");
    return 0;
}

int gsr_adam_step(long long n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                  double beta2, double eps, int step, gsr_stream_t stream)
{
    g_err.clear();
    if (n <= 0) return 0;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return fail_msg("gsr_adam_step: required pointer is null");
    if (step < 1) return fail_msg("gsr_adam_step: step counts from 1");
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15u)
        return fail_msg("gsr_adam_step: arrays must be 16-byte aligned");
    // bias corrections in double, as torch/optim/adam.py computes them in Python floats
    const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
    hipStream_t st = (hipStream_t)stream;
    {
        Scope sc(ST_OPTIM, st);
        launch_adam(n, param, grad, exp_avg, exp_avg_sq, (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2,
                    (float)(1.0 - beta2), (float)eps, (float)std::sqrt(bc2), st);
    }
    GSR_CHECK_LAUNCH("adam_kernel");
    return 0;
}

int gsr_mesh_gaussians(int F, int G, const float* verts, const long long* faces, const float* bary,
                       const float* raw_scales, const float* raw_complex, float thickness, float min_scale,
                       float max_scale, const float* delta_t, const float* delta_r, float* points, float* scaling,
                       float* quaternions, float* clear_dL_dverts, int V, gsr_stream_t stream)
{
    g_err.clear();
    if (V < 0) return fail_msg("gsr_mesh_gaussians: negative V");
    const long long clear_n = clear_dL_dverts ? 3ll * V : 0ll;
    if (F <= 0) {
        if (clear_n > 0) launch_zero_f32(clear_dL_dverts, (size_t)clear_n, (hipStream_t)stream);
        return 0;
    }
    if (G <= 0 || G > 64) return fail_msg("gsr_mesh_gaussians: Gaussians per face must be 1..64");
    if (!verts || !faces || !bary || !raw_scales || !raw_complex || !points || !scaling || !quaternions)
        return fail_msg("gsr_mesh_gaussians: required pointer is null");
    hipStream_t st = (hipStream_t)stream;
    {
        Scope sc(ST_PRODUCERS, st);
        launch_mesh_gaussians(F, G, verts, faces, bary, raw_scales, raw_complex, thickness, min_scale, max_scale, delta_t,
                              delta_r, points, scaling, quaternions, clear_dL_dverts, clear_n, st);
    }
    GSR_CHECK_LAUNCH("mesh_gaussians_fwd_kernel");
    return 0;
}

int gsr_mesh_gaussians_backward(int F, int G, int V, const float* verts, const long long* faces, const float* bary,
                                const float* raw_scales, const float* raw_complex, float min_scale, float max_scale,
                                const float* delta_r, const float* dL_dpoints, const float* dL_dscaling,
                                const float* dL_dquaternions, float* dL_dverts, float* dL_draw_scales,
                                float* dL_draw_complex, float* dL_ddelta_t, float* dL_ddelta_r, int dL_dverts_cleared,
                                gsr_stream_t stream)
{
    g_err.clear();
    if (V < 0) return fail_msg("gsr_mesh_gaussians_backward: negative V");
    if (V > 0 && !dL_dverts) return fail_msg("gsr_mesh_gaussians_backward: dL_dverts is null");
    hipStream_t st = (hipStream_t)stream;
    if (V > 0 && !dL_dverts_cleared) launch_zero_f32(dL_dverts, 3 * (size_t)V, st);
    if (F <= 0) return 0;
    if (G <= 0 || G > 64) return fail_msg("gsr_mesh_gaussians_backward: Gaussians per face must be 1..64");
    if (!verts || !faces || !bary || !raw_scales || !raw_complex || !dL_draw_scales || !dL_draw_complex)
        return fail_msg("gsr_mesh_gaussians_backward: required pointer is null");
    if (delta_r == nullptr && dL_ddelta_r != nullptr)
        return fail_msg("gsr_mesh_gaussians_backward: dL_ddelta_r given without delta_r");
    {
        Scope sc(ST_PRODUCERS, st);
        launch_mesh_gaussians_bwd(F, G, verts, faces, bary, raw_scales, raw_complex, min_scale, max_scale, delta_r,
                                  dL_dpoints, dL_dscaling, dL_dquaternions, dL_dverts, dL_draw_scales, dL_draw_complex,
                                  dL_ddelta_t, dL_ddelta_r, st);
    }
    GSR_CHECK_LAUNCH("mesh_gaussians_bwd_kernel");
    return 0;
}

size_t gsr_l1_ssim_workspace_bytes(int C, int H, int W)
{
    return l1_ssim_workspace_bytes(C > 0 ? C : 0, H > 0 ? H : 0, W > 0 ? W : 0);
}

int gsr_l1_ssim(int C, int H, int W, const float* pred, long long pred_sc, long long pred_sy, long long pred_sx,
                const float* gt, long long gt_sc, long long gt_sy, long long gt_sx, float dssim_factor,
                void* workspace, float* loss_out, float* dL_dpred, long long grad_sc, long long grad_sy,
                long long grad_sx, gsr_stream_t stream)
{
    g_err.clear();
    if (C <= 0 || H <= 0 || W <= 0) return fail_msg("gsr_l1_ssim: image size must be positive");
    if (C > 65535) return fail_msg("gsr_l1_ssim: too many channels");
    if (!pred || !gt || !workspace || !loss_out) return fail_msg("gsr_l1_ssim: required pointer is null");
    const long long ps[3] = {pred_sc, pred_sy, pred_sx}, gs_[3] = {gt_sc, gt_sy, gt_sx},
                    qs[3] = {grad_sc, grad_sy, grad_sx};
    hipStream_t st = (hipStream_t)stream;
    {
        Scope sc(ST_LOSS, st);
        launch_l1_ssim(C, H, W, pred, ps, gt, gs_, dssim_factor, workspace, loss_out, dL_dpred, qs, st);
    }
    GSR_CHECK_LAUNCH("l1_ssim kernels");
    return 0;
}

size_t gsr_depth_l1_workspace_bytes(void) { return depth_l1_workspace_bytes(); }

int gsr_depth_l1(int H, int W, const float* pred, long long pred_sy, long long pred_sx, const float* gt,
                 long long gt_sy, long long gt_sx, float max_depth, float depth_factor, float mask_factor,
                 void* workspace, float* loss_out, float* dL_dpred, long long grad_sy, long long grad_sx,
                 gsr_stream_t stream)
{
    g_err.clear();
    if (H <= 0 || W <= 0) return fail_msg("gsr_depth_l1: image size must be positive");
    if (!pred || !gt || !workspace || !loss_out) return fail_msg("gsr_depth_l1: required pointer is null");
    const long long ps[2] = {pred_sy, pred_sx}, gs_[2] = {gt_sy, gt_sx}, qs[2] = {grad_sy, grad_sx};
    hipStream_t st = (hipStream_t)stream;
    {
        Scope sc(ST_LOSS, st);
        launch_depth_l1(H, W, pred, ps, gt, gs_, max_depth, depth_factor, mask_factor, workspace, loss_out, dL_dpred, qs, st);
    }
    GSR_CHECK_LAUNCH("depth_l1 kernels");
    return 0;
}

int gsr_l1_ssim_backward(int C, int H, int W, const float* pred, long long pred_sc, long long pred_sy, long long pred_sx,
                         const float* gt, long long gt_sc, long long gt_sy, long long gt_sx, float dssim_factor,
                         const void* workspace, const float* grad_scale, float* dL_dpred, long long grad_sc, long long grad_sy,
                         long long grad_sx, gsr_stream_t stream)
{
    g_err.clear();
    if (C <= 0 || H <= 0 || W <= 0) return fail_msg("gsr_l1_ssim_backward: image size must be positive");
    if (C > 65535) return fail_msg("gsr_l1_ssim_backward: too many channels");
    if (!pred || !gt || !workspace || !dL_dpred) return fail_msg("gsr_l1_ssim_backward: required pointer is null");
    const long long ps[3] = {pred_sc, pred_sy, pred_sx}, gs_[3] = {gt_sc, gt_sy, gt_sx}, qs[3] = {grad_sc, grad_sy, grad_sx};
    hipStream_t st = (hipStream_t)stream;
    {
        Scope sc(ST_LOSS, st);
        launch_l1_ssim_grad(C, H, W, pred, ps, gt, gs_, dssim_factor, workspace, grad_scale, dL_dpred, qs, st);
    }
    GSR_CHECK_LAUNCH("ssim_grad_kernel");
    return 0;
}

int gsr_depth_l1_backward(int H, int W, const float* pred, long long pred_sy, long long pred_sx, const float* gt,
                          long long gt_sy, long long gt_sx, float max_depth, float depth_factor, float mask_factor,
                          const float* stats, const float* grad_scale, float* dL_dpred, long long grad_sy, long long grad_sx,
                          gsr_stream_t stream)
{
    g_err.clear();
    if (H <= 0 || W <= 0) return fail_msg("gsr_depth_l1_backward: image size must be positive");
    if (!pred || !gt || !stats || !dL_dpred) return fail_msg("gsr_depth_l1_backward: required pointer is null");
    const long long ps[2] = {pred_sy, pred_sx}, gs_[2] = {gt_sy, gt_sx}, qs[2] = {grad_sy, grad_sx};
    hipStream_t st = (hipStream_t)stream;
    {
        Scope sc(ST_LOSS, st);
        launch_depth_l1_grad(H, W, pred, ps, gt, gs_, max_depth, depth_factor, mask_factor, stats, grad_scale, dL_dpred, qs, st);
    }
    GSR_CHECK_LAUNCH("depth_grad_kernel");
    return 0;
}

int gsr_rgb_depth_loss(int C, int H, int W, const float* pred, long long pred_sc, long long pred_sy, long long pred_sx,
                       const float* gt, long long gt_sc, long long gt_sy, long long gt_sx, float dssim_factor,
                       void* ssim_workspace, int Hd, int Wd, const float* depth_pred, long long dpred_sy, long long dpred_sx,
                       const float* depth_gt, long long dgt_sy, long long dgt_sx, float max_depth, float depth_factor,
                       float mask_factor, void* depth_workspace, float* loss_out, gsr_stream_t stream)
{
    g_err.clear();
    if (C <= 0 || H <= 0 || W <= 0 || Hd <= 0 || Wd <= 0) return fail_msg("gsr_rgb_depth_loss: image size must be positive");
    if (C > 65534) return fail_msg("gsr_rgb_depth_loss: too many channels");
    if (!pred || !gt || !ssim_workspace || !depth_pred || !depth_gt || !depth_workspace || !loss_out)
        return fail_msg("gsr_rgb_depth_loss: required pointer is null");
    const long long ps[3] = {pred_sc, pred_sy, pred_sx}, gs_[3] = {gt_sc, gt_sy, gt_sx};
    const long long dps[2] = {dpred_sy, dpred_sx}, dgs[2] = {dgt_sy, dgt_sx};
    hipStream_t st = (hipStream_t)stream;
    {
        Scope sc(ST_LOSS, st);
        launch_rgb_depth_loss(C, H, W, pred, ps, gt, gs_, dssim_factor, ssim_workspace, Hd, Wd, depth_pred, dps, depth_gt, dgs,
                              max_depth, depth_factor, mask_factor, depth_workspace, loss_out, st);
    }
    GSR_CHECK_LAUNCH("rgb_depth_loss kernels");
    return 0;
}

int gsr_rgb_depth_loss_backward(int C, int H, int W, const float* pred, long long pred_sc, long long pred_sy, long long pred_sx,
                                const float* gt, long long gt_sc, long long gt_sy, long long gt_sx, float dssim_factor,
                                const void* ssim_workspace, int Hd, int Wd, const float* depth_pred, long long dpred_sy,
                                long long dpred_sx, const float* depth_gt, long long dgt_sy, long long dgt_sx, float max_depth,
                                float depth_factor, float mask_factor, const float* loss_out, const float* grad_scale,
                                float* dL_dpred, long long grad_sc, long long grad_sy, long long grad_sx, float* dL_ddepth,
                                long long dgrad_sy, long long dgrad_sx, gsr_stream_t stream)
{
    g_err.clear();
    if (C <= 0 || H <= 0 || W <= 0 || Hd <= 0 || Wd <= 0) return fail_msg("gsr_rgb_depth_loss_backward: image size must be positive");
    if (C > 65534) return fail_msg("gsr_rgb_depth_loss_backward: too many channels");
    if (!pred || !gt || !ssim_workspace || !depth_pred || !depth_gt || !loss_out || !dL_dpred || !dL_ddepth)
        return fail_msg("gsr_rgb_depth_loss_backward: required pointer is null");
    const long long ps[3] = {pred_sc, pred_sy, pred_sx}, gs_[3] = {gt_sc, gt_sy, gt_sx}, qs[3] = {grad_sc, grad_sy, grad_sx};
    const long long dps[2] = {dpred_sy, dpred_sx}, dgs[2] = {dgt_sy, dgt_sx}, dqs[2] = {dgrad_sy, dgrad_sx};
    hipStream_t st = (hipStream_t)stream;
    {
        Scope sc(ST_LOSS, st);
        launch_rgb_depth_loss_grad(C, H, W, pred, ps, gt, gs_, dssim_factor, ssim_workspace, Hd, Wd, depth_pred, dps, depth_gt, dgs,
                                   max_depth, depth_factor, mask_factor, loss_out + 3, grad_scale, dL_dpred, qs, dL_ddepth, dqs, st);
    }
    GSR_CHECK_LAUNCH("rgb_depth_loss gradient kernel");
    return 0;
}

int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     gsr_stream_t stream)
{
    (void)projmatrix;   // the reference's frustum test only uses the view matrix (auxiliary.h:154)
    g_err.clear();
    if (P <= 0) return 0;
    if (!means3D || !viewmatrix || !present) return fail_msg("gsr_mark_visible: required pointer is null");
    launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
    GSR_CHECK_LAUNCH("mark_visible_kernel");
    return 0;
}

int gsr_release_stream_state(gsr_stream_t stream)
{
    g_err.clear();
    hipStream_t st = (hipStream_t)stream;
    int dev = 0;
    GSR_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_cnt_mu);
    auto it = g_cnt.find(std::make_pair(dev, st));
    if (it == g_cnt.end()) return 0;
    if (it->second.busy.exchange(true)) return fail_msg("gsr_release_stream_state: a forward on this stream is in flight");
    if (it->second.base) { GSR_CHECK(hipStreamSynchronize(st)); GSR_CHECK(hipFree(it->second.base)); }
    g_cnt.erase(it);
    return 0;
}

// ---- gsr_camera_key: the sixteen floats of a view matrix to the host without touching the caller's stream.
namespace {
__global__ void __launch_bounds__(64) camera_key_kernel(const float* __restrict__ m, long long s0, long long s1, uint32_t* __restrict__ pad,
                                                        uint32_t seq)
{
    const int lane = threadIdx.x;
    if (lane < 16) pad[lane] = __float_as_uint(m[(long long)(lane >> 2) * s0 + (long long)(lane & 3) * s1]);
    __threadfence_system();   // (one wave: its wait covers every lane's store)
    if (lane == 0) __hip_atomic_store(&pad[16], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
std::mutex g_key_mu;
std::map<int, hipStream_t> g_key_stream;   // device -> the library's non-blocking side stream
constexpr int PAD_KEY = 32;                // words [32 .. 48] of the thread's pad: the matrix, then its sequence number
}  // namespace

namespace {
thread_local uint32_t g_key_seq = 0;          // sequence number of this thread's read under way (0: none)
thread_local hipStream_t g_key_side = nullptr;
}
int gsr_camera_key_begin(const float* viewmatrix, long long row_stride, long long col_stride)
{
    g_err.clear();
    g_key_seq = 0;
    if (!viewmatrix) return fail_msg("gsr_camera_key: null pointer");
    int dev = 0;
    GSR_CHECK(hipGetDevice(&dev));
    hipStream_t side = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_key_mu);
        auto it = g_key_stream.find(dev);
        if (it == g_key_stream.end()) {
            GSR_CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
            g_key_stream[dev] = side;
        } else {
            side = it->second;
        }
    }
    if (const int bad = ensure_pad()) return bad;
    static_assert(PAD_KEY + 17 <= PAD_WORDS, "pad");
    const uint32_t seq = next_seq();
    camera_key_kernel<<<1, 64, 0, side>>>(viewmatrix, row_stride, col_stride, g_pinned + PAD_KEY, seq);
    GSR_CHECK_LAUNCH("camera_key_kernel");
    g_key_seq = seq;
    g_key_side = side;
    return 0;
}

int gsr_camera_key_end(unsigned long long* key)
{
    if (!key) { g_err.clear(); return fail_msg("gsr_camera_key: null pointer"); }
    if (g_key_seq == 0u) { g_err.clear(); return fail_msg("gsr_camera_key_end: no read under way on this thread"); }
    const uint32_t seq = g_key_seq;
    g_key_seq = 0;
    bool seen = false;
    if (const int bad = wait_pad_word(g_pinned + PAD_KEY + 16, seq, g_key_side, &seen)) return bad;
    if (!seen) return fail_msg("gsr_camera_key: the kernel retired without its store showing up in the pinned pad");
    // FNV-1a over the sixteen words, then a finaliser (the low bits of a float's pattern are what differs between neighbouring
    // cameras of a rig)
    unsigned long long h = 1469598103934665603ull;
    for (int i = 0; i < 16; i++) {
        uint32_t w = g_pinned[PAD_KEY + i];
        if (w == 0x80000000u) w = 0u;   // -0.0 == 0.0: the same camera
        for (int b = 0; b < 4; b++) { h ^= (w >> (8 * b)) & 0xffu; h *= 1099511628211ull; }
    }
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33;
    *key = h;
    return 0;
}

int gsr_camera_key(const float* viewmatrix, long long row_stride, long long col_stride, unsigned long long* key)
{
    if (!key) { g_err.clear(); return fail_msg("gsr_camera_key: null pointer"); }
    if (const int bad = gsr_camera_key_begin(viewmatrix, row_stride, col_stride)) return bad;
    return gsr_camera_key_end(key);
}

int gsr_debug_preprocess_occupancy(int* exact, int* planned)
{
    gsr::preprocess_occupancy(exact, planned);
    return 0;
}

int gsr_debug_set_bwd_order(const void* device_order)
{
    gsr::g_bwd_order = static_cast<const uint32_t*>(device_order);
    return 0;
}

int gsr_debug_set_trace(void* device_buffer)
{
    gsr::g_trace = static_cast<uint64_t*>(device_buffer);
    return 0;
}

// ---- per-kernel timing --------------------------------------------------------------------------
int gsr_num_stages(void) { return ST_COUNT; }
const char* gsr_stage_name(int stage) { return (stage >= 0 && stage < ST_COUNT) ? kStageNames[stage] : ""; }

int gsr_profile_enable(int on)
{
    g_prof.on.store(on != 0);
    return 0;
}

int gsr_profile_read(float* ms, int* counts, int reset)
{
    g_err.clear();
    if (!ms || !counts) return fail_msg("gsr_profile_read: null output pointer");
    for (int i = 0; i < ST_COUNT; i++) { ms[i] = 0.f; counts[i] = 0; }
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (const Rec& r : g_prof.recs) {
        GSR_CHECK(hipEventSynchronize(r.b));
        float t = 0.f;
        GSR_CHECK(hipEventElapsedTime(&t, r.a, r.b));
        ms[r.stage] += t;
        counts[r.stage] += 1;
    }
    if (reset) {
        for (const Rec& r : g_prof.recs) { g_prof.pool.push_back(r.a); g_prof.pool.push_back(r.b); }
        g_prof.recs.clear();
    }
    return 0;
}

// ---- introspection -----------------------------------------------------------------------------
__global__ void export_geom_kernel(int P, const float4* g0, const float4* g1, const float* depth, const float* rgb_in,
                                   float* means2D, float* conic_opacity, float* depths, float* rgb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float4 a = g0[i], b = g1[i];
    if (means2D) { means2D[2 * i] = a.x; means2D[2 * i + 1] = a.y; }
    if (conic_opacity) {
        conic_opacity[4 * i] = a.z; conic_opacity[4 * i + 1] = a.w; conic_opacity[4 * i + 2] = b.x;
        conic_opacity[4 * i + 3] = b.y;
    }
    if (depths) depths[i] = depth[i];
    if (rgb) { rgb[3 * i] = rgb_in[3 * i]; rgb[3 * i + 1] = rgb_in[3 * i + 1]; rgb[3 * i + 2] = rgb_in[3 * i + 2]; }
}

int gsr_debug_export(int P, int R, int num_segments, int W, int H, const void* geom_buffer, const void* binning_buffer,
                     const void* image_buffer, float* means2D, float* conic_opacity, float* depths, float* rgb,
                     uint32_t* tile_ranges, uint32_t* point_list, float* final_T, uint32_t* n_contrib,
                     gsr_stream_t stream)
{
    g_err.clear();
    hipStream_t st = (hipStream_t)stream;
    const Tiles t = tiles_of(W, H);
    if (P > 0 && geom_buffer && (means2D || conic_opacity || depths || rgb)) {
        GeomState g = carve_geom(const_cast<void*>(geom_buffer), P);
        export_geom_kernel<<<(P + 255) / 256, 256, 0, st>>>(P, g.g0, g.g1, g.depth, g.rgb, means2D, conic_opacity,
                                                            depths, rgb);
        GSR_CHECK_LAUNCH("export_geom_kernel");
    }
    if (image_buffer) {
        ImageState im = carve_image(const_cast<void*>(image_buffer), W, H);
        const size_t N = (size_t)W * H;
        if (tile_ranges) GSR_CHECK(hipMemcpyAsync(tile_ranges, im.ranges, 8 * (size_t)t.T, hipMemcpyDeviceToDevice, st));
        if (final_T) GSR_CHECK(hipMemcpyAsync(final_T, im.final_T, 4 * N, hipMemcpyDeviceToDevice, st));
        if (n_contrib) GSR_CHECK(hipMemcpyAsync(n_contrib, im.n_contrib, 4 * N, hipMemcpyDeviceToDevice, st));
    }
    if (R > 0 && binning_buffer && point_list) {
        BinState b = carve_bin(const_cast<void*>(binning_buffer), R, num_segments > 0 ? num_segments : 0);
        GSR_CHECK(hipMemcpyAsync(point_list, b.point_list, 4 * (size_t)R, hipMemcpyDeviceToDevice, st));
    }
    return 0;
}

int gsr_debug_export_masks(int R, int num_segments, const void* binning_buffer, uint64_t* masks, gsr_stream_t stream)
{
    g_err.clear();
    if (R <= 0 || num_segments <= 0) return 0;
    if (!binning_buffer || !masks) return fail_msg("gsr_debug_export_masks: required pointer is null");
    BinState b = carve_bin(const_cast<void*>(binning_buffer), R, num_segments);
    GSR_CHECK(hipMemcpyAsync(masks, b.masks, 8 * 256 * (size_t)num_segments, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

}  // extern "C"
