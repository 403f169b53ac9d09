// vf_hip.hip -- C-ABI (include/vf_hip.h) over the gfx950 kernels in vf_kernels.h.
// Built by __graft_entry__.build():  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC + the -mllvm code-generation switches of HIPCC_TUNING
#include "../../include/vf_hip.h"
#include "vf_kernels.h"

#include <algorithm>
#include <cmath>
#include <vector>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include <string>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>
#include <dlfcn.h>
#include <sys/mman.h>
#include <rccl/rccl.h>     // types only: the functions are resolved at run time (resolve_rccl)

using namespace vf;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define VF_HIP_TRY(expr)                                                                           \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(VF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

double eotf_d(double s) { return s <= 0.04045 ? s / 12.92 : std::pow((s + 0.055) / 1.055, 2.4); }

struct SrgbTables {
    float decode[256];
    float thresh[256];
    SrgbTables()
    {
        for (int k = 0; k < 256; ++k) {
            decode[k] = (float)eotf_d((double)k / 255.0);
            thresh[k] = k == 0 ? -INFINITY : (float)eotf_d(((double)k - 0.5) / 255.0);
        }
    }
    uint32_t encode(float c) const
    {
        uint32_t k = 0;
        while (k < 255 && c >= thresh[k + 1]) ++k;
        return k;
    }
};
const SrgbTables &tables()
{
    static const SrgbTables t;
    return t;
}

bool is_pow2(uint32_t v) { return v && !(v & (v - 1)); }
// Tile-shard layouts are passed as one word, `skew | stripe_log2 << 16` (include/vf_hip.h, VF_TILE_LAYOUT): tile (tx, ty) belongs
// to rank ((tx >> stripe_log2) + skew * ty) % nranks.
uint32_t layout_skew(uint32_t layout) { return layout & 0xFFFFu; }
uint32_t layout_shift(uint32_t layout) { return (layout >> 16) & 0xFu; }
// Round 5: a layout word with bit 20 set names a REGISTERED stripe map (vf_tile_layout_register_map): an explicit owner per column
// stripe instead of the round-robin rule -- what a load-balanced deal of the stripes needs (vf_balance_stripes).  Bits 21..26: the
// map's number in the process-wide registry below; skew 0 by construction (column stripes).  Maps are immutable once registered.
constexpr uint32_t kLayoutMapBit = 1u << 20, kMaxStripeMaps = 64, kMaxStripes = 256;
struct StripeMap { std::vector<uint8_t> owner; uint32_t shift = 0, nranks = 0; };
std::mutex g_maps_mu;
std::vector<StripeMap> g_maps;                              // (entries are never removed or changed: a pointer into it stays valid ... while the vector does not grow:
const StripeMap *layout_map(uint32_t layout)                //  so it is reserved to its final capacity at the first registration)
{
    if (!(layout & kLayoutMapBit)) return nullptr;
    std::lock_guard<std::mutex> lk(g_maps_mu);
    const uint32_t id = (layout >> 21) & 0x3Fu;
    return id < g_maps.size() ? &g_maps[id] : nullptr;
}
bool layout_valid(uint32_t layout)
{
    if (layout & kLayoutMapBit) {
        const StripeMap *m = layout_map(layout);
        return m && (layout >> 27) == 0u && layout_skew(layout) == 0u && layout_shift(layout) == m->shift;
    }
    return (layout >> 20) == 0u;                            // skew < 65536, stripe_log2 <= 15, nothing above
}
// owner of tile (tx, ty) under a layout word (valid, see above)
uint32_t layout_owner(uint32_t layout, const StripeMap *m, uint32_t tx, uint32_t ty, uint32_t nranks)
{
    if (m) { const uint32_t sc = tx >> m->shift; return sc < m->owner.size() ? m->owner[sc] : 0u; }
    return ((tx >> layout_shift(layout)) + layout_skew(layout) * ty) % nranks;
}
uint32_t ilog2(uint32_t v)
{
    uint32_t s = 0;
    while ((1u << s) < v) ++s;
    return s;
}

} // namespace

// Page-locked host memory.  hipHostMalloc takes 7-9 ms for a C4 frame (64 MiB), the first copy into it another 11-16, hipHostFree 5:
// the runtime faults the buffer in 4 KiB page by page.  The same bytes as 2 MiB-aligned ordinary memory with MADV_HUGEPAGE, then
// hipHostRegister: 2.5-2.8 ms to make, copies at full rate (1.19 ms) from the first one, 2.5 ms to free (tools/micro/pin_cost.hip,
// round 5) -- cheaper than the page faults of ONE copy into fresh ordinary memory (4.4-4.9 ms), so even a one-shot caller's only
// read-back goes into such a buffer.  Small buffers and hosts without huge pages: hipHostMalloc.
struct PinnedRegistry {
    std::mutex mu;
    std::vector<void *> registered;                            // made by pinned_alloc's huge-page branch (freed by unregister + free)
    static PinnedRegistry &get() { static PinnedRegistry *r = new PinnedRegistry; return *r; }
};
static hipError_t pinned_alloc(void **out, size_t n)
{
    constexpr size_t kHuge = (size_t)2 << 20;
    *out = nullptr;
    if (n >= 2 * kHuge) {
        void *p = nullptr;
        const size_t r = (n + kHuge - 1) & ~(kHuge - 1);
        if (posix_memalign(&p, kHuge, r) == 0) {
            (void)madvise(p, r, MADV_HUGEPAGE);                 // (refused or unavailable: ordinary pages, still correct)
            if (hipHostRegister(p, r, hipHostRegisterDefault) == hipSuccess) {
                std::lock_guard<std::mutex> lk(PinnedRegistry::get().mu);
                PinnedRegistry::get().registered.push_back(p);
                *out = p;
                return hipSuccess;
            }
            (void)hipGetLastError();
            std::free(p);
        }
    }
    return hipHostMalloc(out, n, hipHostMallocDefault);
}
static void pinned_free(void *p)
{
    if (!p) return;
    bool ours = false;
    {
        PinnedRegistry &R = PinnedRegistry::get();
        std::lock_guard<std::mutex> lk(R.mu);
        auto it = std::find(R.registered.begin(), R.registered.end(), p);
        if (it != R.registered.end()) { R.registered.erase(it); ours = true; }
    }
    if (ours) { (void)hipHostUnregister(p); std::free(p); }
    else (void)hipHostFree(p);
}

struct vf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop;
    float *d_thresh = nullptr;   // 256 sRGB store thresholds
    // Streams the handles of this context borrow, made on first need: a stream costs 3-10 ms to create and 2-3 ms to destroy on this
    // runtime (round 5, measured), so they belong to the process-lifetime context, not to every handle.  side / side2: a frame's plan
    // chain and set-up pass (they overlap the previous frame's tile kernel; a handle's first frame needs neither); copy: batch read-back.
    // Handles that render at the same time share them: order is kept by events, sharing only serialises their plan chains.
    hipStream_t side = nullptr, side2 = nullptr, copy = nullptr;
    uint8_t *d_maps = nullptr;   // [kMaxStripeMaps][kMaxStripes] owner tables of the registered stripe maps, uploaded on first use (vf_stitch_tiles_device)
    uint64_t maps_uploaded = 0;  // bit k: map k is there
    // The members above that are made on first need are shared by every handle of the context, and the drop-in module keeps ONE context
    // and releases the GIL while it renders: two objects drawing their second frames on two threads would both find `side` missing.
    std::mutex lazy_mu;
};

// the context's plan streams, made once (3-10 ms each), under the context's lock
static hipError_t ctx_side_streams(vf_ctx *c, hipStream_t *side, hipStream_t *side2)
{
    std::lock_guard<std::mutex> lk(c->lazy_mu);
    hipError_t e = hipSuccess;
    if (!c->side) e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e == hipSuccess && !c->side2) e = hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking);
    *side = c->side; *side2 = c->side2;
    return e;
}
static hipError_t ctx_copy_stream(vf_ctx *c, hipStream_t *copy)
{
    std::lock_guard<std::mutex> lk(c->lazy_mu);
    hipError_t e = hipSuccess;
    if (!c->copy) e = hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking);
    *copy = c->copy;
    return e;
}

// What the second half of a frame (draw_frame) needs of the first (plan_frame)
struct FramePlan {
    FrameParams P;
    uint32_t set = 0, ntiles = 0;
    bool solo = false, motion_starts = false;
    bool sampled = true;                                         // timing level 3: this frame is one of those that carry events
    uint32_t *rc_lo = nullptr, *rc_hi = nullptr, *seg_count = nullptr;
};

struct vf_terrain {
    vf_ctx *ctx = nullptr;
    uint32_t W = 0, H = 0, n = 0;
    uint32_t nb = 0, nblocks = 0;        // 8x8-cell blocks per side / in total
    uint32_t ntx = 0, nty = 0;           // 64x64 screen tiles
    // shard
    uint32_t rank = 0, nranks = 1, band_h = kTileH, local_rows = 0, skew = 0;
    // uniforms
    float u[44];
    bool have_uniforms = false;
    float u_frame[44] = {};              // uniforms / shade mode of the frame vf_terrain_render drew last (visibility, fragment diagnostics)
    uint32_t shade_mode_frame = 0;
    bool have_frame = false;             // u_frame is valid: set by vf_terrain_render only, cleared when the shard layout changes
    uint32_t *d_rgba_scratch = nullptr;  // output of those diagnostic re-renders: the caller's frame is never overwritten
    uint32_t *d_diag = nullptr;          // [0] covered pixels (fragment-stage diagnostics)
    float u_drawn[32];                   // view + proj of the frame rendered last (is the camera moving?)
    bool have_drawn = false, camera_moving = false, was_moving = false;
    uint32_t frames_since_reset = 0;     // frames planned since create / set_shard (the feedback arrays were cleared then)
    // device state
    float *d_xs = nullptr, *d_sinx = nullptr, *d_cosz = nullptr;
    int32_t *d_txi = nullptr, *d_tyj = nullptr;
    float *d_height_own = nullptr;       // Scene::set_height_from_r32f's copy (its own allocation: it is replaced when the size changes)
    float *d_height_dummy = nullptr;     // the 1x1 zero texture of TerrainSpike (in the slab)
    const float *d_height = nullptr;
    // vf_terrain_render_batch_host: a ring of device frames the poses are drawn into while earlier ones travel to the host
    static constexpr uint32_t kBatchRing = 3;
    uint32_t *d_batch[kBatchRing] = { nullptr, nullptr, nullptr };
    hipEvent_t batch_drawn[kBatchRing] = { nullptr, nullptr, nullptr }, batch_copied[kBatchRing] = { nullptr, nullptr, nullptr };
    hipStream_t copy_stream = nullptr;
    uint8_t *slab = nullptr;             // ONE device allocation behind every fixed-size buffer of the handle (round 5: construction was ~40 hipMallocs)
    uint32_t tw = 1, th = 1;
    bool bounds_dirty = true;
    float2 *d_bounds = nullptr;          // per block: min/max displaced height
    float *d_hblk = nullptr;             // displaced-height cache: 81 floats per block
    // Per-frame plan state, kPlanStates times: frame f uses set f % kPlanStates.  The plan kernels of a frame run on `side` and touch
    // nothing else, so they overlap the previous frames' tile kernels (which still read the other sets) instead of waiting for them.
    // (VF_PLAN_STATES 3 and VF_OVERLAP_FRAMES=1 are round 4's experiment with consecutive frames on alternating streams and output
    //  buffers, whose tile kernels then overlap -- frame f + 1's persistent workgroups take the CUs frame f's tail leaves idle:
    //  measured +8 % on one GPU and +6 % on a rank of eight with two streams, -1 % on the rank with three; tools/exp_overlap.py.)
#ifndef VF_PLAN_STATES
#define VF_PLAN_STATES 2
#endif
    static constexpr uint32_t kPlanStates = VF_PLAN_STATES;
    struct PlanState {
        PixelBox *ranges = nullptr;      // per block: conservative pixel rectangle (from the block's height bounds)
        VertexRec *vtx = nullptr;        // per block 81 x {X, Y, 1/w, h} (k_block_setup)
        BlockRec *recs = nullptr;        // per block: alive masks, exact pixel box
        ulonglong2 *gen = nullptr;       // per block: primitives that need the generic path (valid where the record says so)
        uint32_t *seg_list = nullptr;    // 16-block segments with a block that reaches this shard's pixels (+ [nsegs] = their number)
        PixelBox *row_ranges = nullptr;  // per block row
        float4 *cap_seg = nullptr;       // per block: capsule axis (screen space)
        float *cap_rad = nullptr;        // per block: capsule radius
        uint32_t *rc = nullptr;          // per (tile column, block row): [lo | hi) block-column range, 2 * nb * ntx words
        uint2 *work = nullptr;           // busy tiles of the frame: (item, weight) as k_plan lists them
        uint2 *work_sorted = nullptr;    // ... heaviest first (k_plan_sort; the second half of the same allocation): what the tile kernel pulls from
        uint32_t *work_count = nullptr;  // [0] work items, [1] split budget used, [2] queue head, [3] items handed to the complete tile kernel
        uint32_t *redo = nullptr;        // those items (indices into work)
        uint32_t *background = nullptr;  // per local tile: bit 0 = no block row reaches it; bits 8.. = log2 of the strips it is cut into
        uint32_t *flags_new = nullptr;   // the same words as k_plan writes them; k_plan_sort moves them into `background`
        uint32_t *feedback = nullptr;    // time (10 ns ticks) per tile, added by this set's tile kernel, read two frames later; [ntiles] = split quantum;
                                         // then 64 words per tile: the time of each of its pieces (strip x slice)
        hipEvent_t planned = nullptr, drawn = nullptr, boxed = nullptr, set_up = nullptr;
        hipEvent_t pev[3] = { nullptr, nullptr, nullptr };   // timing: plan start, behind the block boxes, behind the plan (on the plan's stream), of this set's LAST plan
        bool pev_valid = false;
        uint8_t *slab = nullptr;         // one allocation per plan state
        float u_used[32] = {};           // view + proj of the frame whose tile times sit in `feedback` (the plan looks them up through the camera motion)
        bool have_u_used = false;
    } ps[kPlanStates];
    uint32_t cur_set = 0, last_set = 0;  // the set the next frame takes; the set of the frame rendered last
    const uint32_t *last_out = nullptr;  // output buffer of the frame rendered last
    hipStream_t side = nullptr;          // k_block_boxes -> k_plan -> k_plan_sort
    hipStream_t side2 = nullptr;         // k_block_setup (needs the block boxes only): beside the plan chain, both under the previous frame
    uint32_t frame_no = 0;
    float *d_lut = nullptr;              // 256*3 linear floats
    uint32_t *d_rgba_own = nullptr;
    uint8_t *d_png = nullptr, *h_png = nullptr;   // PNG scanlines of the last frame: device, pinned host
    uint8_t *h_stage = nullptr;                   // kStageSlots x kStageChunk pinned bytes: device -> pageable host copies go through here
    hipEvent_t stage_ev[4] = { nullptr, nullptr, nullptr, nullptr };
    // THE NEXT FRAME'S PLAN, QUEUED AHEAD (round 5).  A camera at rest draws the same plan again and again, and a caller that waits for
    // each frame (render_png, render_rgba: every call of the reference's API) used to pay the plan chain -- block boxes, set-up pass,
    // plan, sort: 0.2 ms at C4 -- in front of every tile kernel, because nothing is left to hide it under once the caller has waited.
    // So when a frame was drawn with the very inputs of the frame before it, the first half of the NEXT frame (plan_frame) is queued
    // right behind it: by the time the caller comes back the plan is there.  `inputs_gen` counts everything a plan depends on (uniforms,
    // heights, shard, shade mode, timing); a plan made ahead is used only for the generation it was made for, and otherwise thrown
    // away: the bookkeeping it advanced is rolled back and the frame is planned again (its kernels queue behind the stale ones).
    uint64_t inputs_gen = 1;
    struct PrePlan {
        bool valid = false;
        FramePlan K;
        uint64_t gen = 0;
        // what plan_frame advanced (restored when the plan is thrown away)
        uint32_t cur_set = 0, frame_no = 0, frames_since_reset = 0;
        bool camera_moving = false, was_moving = false, have_drawn = false;
        float u_drawn[32] = {};
    } pre;
    uint64_t last_drawn_gen = 0;         // inputs_gen of the frame drawn last (two frames of one generation: the camera is at rest)
    bool replan_fresh = false;           // a plan queued ahead was thrown away: the next plan takes the previous frame's tile times (drop_preplan)
    // Round 6: a caller that WAITS for every frame (every call of the reference's API) never makes the context's two plan streams (17-20 ms
    // once per process, which its second call used to pay): its plans run on its own stream, and the plan made ahead for a camera at
    // rest is queued BEHIND THE NEXT READ-BACK'S COPY (flush_preplan) -- it runs while the host unpacks / deflates / returns, not in
    // front of the copy the caller waits for.  The plan streams are made when a frame arrives while the previous one is still in
    // flight: a caller that streams frames, which is what they are for.
    bool preplan_pending = false;
    // ... unless that caller comes back faster than the plan runs: a tight loop of synchronous frames with nothing between them finds
    // the plan made ahead still running behind the copy (C4 render_rgba 2.0 -> 2.2 ms, C2 0.10 -> 0.14), where the plan streams would
    // have run it UNDER the frame.  Three such calls in a row and the handle takes the plan streams after all (17-20 ms, once).
    uint32_t tight_calls = 0;
    bool want_plan_streams = false;
    hipEvent_t copied = nullptr;         // behind a read-back's copy: the host waits for this, not for the stream (the plan made ahead follows it)
    uint32_t *d_tile_map = nullptr;      // tile shards: local tile -> tx | ty << 16
    uint8_t *d_stripe_owner = nullptr;   // tile shards with a registered stripe map: owner per column stripe (kMaxStripes bytes)
    bool use_map = false;
    bool shard_tiles = false;
    uint32_t shade_mode = 0;             // VF_SHADE_REFERENCE / VF_SHADE_SPEC_T32
    uint32_t precision = VF_PRECISION_FAST, precision_frame = VF_PRECISION_FAST;   // fragment arithmetic (of the frame vf_terrain_render drew last)
    uint32_t local_tiles = 0;            // tiles this handle renders (= ntx * local tile rows unless tile-sharded)
    uint32_t *d_rgba = nullptr;
    uint32_t *d_vis = nullptr;           // only allocated for vf_terrain_read_visibility
    uint32_t *d_stats = nullptr;         // [0] (tile, block) pairs rasterised
    // timing: a ring of (start, after block boxes, after plan, after tile) events, one set per rendered frame
    static constexpr int kTimingRing = 64;
    bool timing = false;
    bool stats_on = false;               // per-item statistics as well (vf_terrain_enable_timing(t, 1)); 2 = device times only
    uint32_t timing_every = 1;           // 3 = device times of every 4th frame only: two event records on the draw stream cost a frame 2 % (tools/exp_timing_cost.py)
    hipEvent_t ev[kTimingRing][5] = {};   // [3] after the tile kernels, [4] before the clear (caller's stream); [0..2] unused since round 6: the plan's events live with its plan state (pev)
    hipEvent_t entry = nullptr;          // caller's stream at render entry (orders a height-cache rebuild after the caller's work)
    uint32_t timed_frames = 0;           // frames recorded since timing was enabled
    hipStream_t last_stream = nullptr;
    bool rendered = false;
    // Which instantiation of the tile kernel draws the frames -- with or without line groups in the raster's line loop (vf_kernels.h,
    // raster_fast) -- is measured, like the frame plan: which one is faster depends on what the camera shows and on how the frame is
    // cut (C4: groups -7 % on the top-down camera at any rank count, -1.5 % on one GPU at the default camera, +5 % on a rank of eight,
    // whose items are narrow strips).  Never a difference in the picture.  groups_mode: -1 measure and choose, 0 / 1 fixed.
    int groups_mode = -1;
    // A probe = two events around a frame's kernels on the draw stream (k_clear + the tile launches); a sample = the time between them.
    // (Round 4 sampled the frame PERIOD between two consecutive frames of one variant, in runs AAABBB: right for a camera at rest, but a
    //  moving camera's frames differ in cost by +-20 % from pose to pose and a run of three A's sat on other poses than the B's -- on the
    //  C5 orbit the handle kept the slower variant in the bench, 0.326 instead of 0.305 ms per pose.  Round 5: every probed frame is a
    //  sample of its own, the variants alternate ABBA ABBA ..., so any linear drift of the cost over the window cancels exactly.)
    struct GroupProbe { hipEvent_t a = nullptr, b = nullptr; int variant = 0; uint32_t seq = 0, epoch = 0; bool pending = false, valid = false; } gprobe[16];
    uint32_t gprobe_head = 0, g_epoch_id = 0;
    float g_ms[2] = { 0.0f, 0.0f };      // frame period with each variant in this epoch (mean of the samples taken)
    uint32_t g_n[2] = { 0u, 0u };
    uint32_t g_epoch_frames = 0;         // frames since the epoch began (shard change, height upload, camera jump)
    int groups_now = 1;                  // the variant of the frame rendered last
    // vf_dist_exchange_bands: the chunks this rank receives in the all-to-all ([nranks][chunk_tiles] tile slots) and the band it stitches from them
    uint8_t *d_xrecv = nullptr, *d_xband = nullptr;
    size_t xrecv_bytes = 0, xband_bytes = 0;
};

extern "C" {

const char *vf_last_error(void) { return g_err.c_str(); }

int vf_device_count(int *count)
{
    if (!count) return fail(VF_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(VF_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n;
    return VF_OK;
}

static void fill_info(int ordinal, const hipDeviceProp_t &p, vf_device_info *out)
{
    std::memset(out, 0, sizeof *out);
    std::snprintf(out->name, sizeof out->name, "%s", p.name);
    std::snprintf(out->arch, sizeof out->arch, "%s", p.gcnArchName);
    out->device_ordinal = ordinal;
    out->compute_units = p.multiProcessorCount;
    out->wavefront_size = p.warpSize;
    out->clock_khz = p.clockRate;
    out->total_mem_bytes = p.totalGlobalMem;
    out->lds_bytes_per_cu = p.maxSharedMemoryPerMultiProcessor;
    out->pci_bus_id = p.pciBusID;
    out->pci_device_id = p.pciDeviceID;
}

int vf_device_query(int device_ordinal, vf_device_info *out)
{
    if (!out) return fail(VF_ERR_INVALID, "out is NULL");
    int n = 0;
    int rc = vf_device_count(&n);
    if (rc != VF_OK) return rc;
    if (device_ordinal < 0 || device_ordinal >= n) return fail(VF_ERR_NO_DEVICE, "No suitable GPU adapter");
    hipDeviceProp_t p;
    VF_HIP_TRY(hipGetDeviceProperties(&p, device_ordinal));
    fill_info(device_ordinal, p, out);
    return VF_OK;
}

int vf_ctx_create(int device_ordinal, vf_ctx **out)
{
    if (!out) return fail(VF_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0 || device_ordinal < 0 || device_ordinal >= n)
        return fail(VF_ERR_NO_DEVICE, "No suitable GPU adapter");   // reference string, src/terrain/mod.rs:285
    vf_ctx *c = new (std::nothrow) vf_ctx;
    if (!c) return fail(VF_ERR_NOMEM, "out of host memory");
    c->device = device_ordinal;
    hipError_t err = hipSetDevice(device_ordinal);
    if (err == hipSuccess) err = hipGetDeviceProperties(&c->prop, device_ordinal);
    if (err == hipSuccess) err = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipMalloc(&c->d_thresh, 256 * sizeof(float));
    if (err == hipSuccess) err = hipMemcpy(c->d_thresh, tables().thresh, 256 * sizeof(float), hipMemcpyHostToDevice);
    if (err != hipSuccess) {
        std::string m = std::string("context creation failed: ") + hipGetErrorString(err);
        vf_ctx_destroy(c);
        return fail(VF_ERR_HIP, m);
    }
    *out = c;
    return VF_OK;
}

void vf_ctx_destroy(vf_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_thresh) (void)hipFree(c->d_thresh);
    if (c->d_maps) (void)hipFree(c->d_maps);
    for (hipStream_t q : { c->side, c->side2, c->copy }) if (q) (void)hipStreamDestroy(q);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int vf_ctx_device_info(const vf_ctx *ctx, vf_device_info *out)
{
    if (!ctx || !out) return fail(VF_ERR_INVALID, "NULL argument");
    fill_info(ctx->device, ctx->prop, out);
    return VF_OK;
}

int vf_ctx_stream(const vf_ctx *ctx, void **stream)
{
    if (!ctx || !stream) return fail(VF_ERR_INVALID, "NULL argument");
    *stream = (void *)ctx->stream;
    return VF_OK;
}

// ------------------------------------------------------------------------------------------------

static uint32_t compute_local_rows(uint32_t H, uint32_t rank, uint32_t nranks, uint32_t band_h)
{
    if (nranks <= 1) return H;
    uint32_t rows = 0;
    for (uint32_t b = 0; b * band_h < H; ++b)
        if (b % nranks == rank) rows += (b + 1) * band_h <= H ? band_h : H - b * band_h;
    return rows;
}

static AxisTables axis(const vf_terrain *t)
{
    AxisTables A;
    A.xs = t->d_xs; A.sinx = t->d_sinx; A.cosz = t->d_cosz; A.txi = t->d_txi; A.tyj = t->d_tyj;
    return A;
}

// Buffers carved out of one device allocation, 256-byte aligned; those that must start at zero lie first and share one memset.
namespace {
struct Carver {
    struct Item { void **p; size_t bytes; };
    std::vector<Item> zeroed, plain;
    void add(void **p, size_t bytes, bool zero = false) { (zero ? zeroed : plain).push_back({ p, (bytes + 255u) & ~(size_t)255u }); }
    hipError_t commit(uint8_t **base, hipStream_t s)
    {
        size_t zero_bytes = 0, total = 0;
        for (const Item &i : zeroed) zero_bytes += i.bytes;
        total = zero_bytes;
        for (const Item &i : plain) total += i.bytes;
        hipError_t e = hipMalloc((void **)base, total);
        if (e != hipSuccess) return e;
        size_t off = 0;
        for (const Item &i : zeroed) { *i.p = *base + off; off += i.bytes; }
        for (const Item &i : plain) { *i.p = *base + off; off += i.bytes; }
        return zero_bytes ? hipMemsetAsync(*base, 0, zero_bytes, s) : hipSuccess;
    }
};
} // namespace

static hipError_t sync_sides(const vf_terrain *t)            // (the side streams exist from the handle's second frame on)
{
    hipError_t e = t->side ? hipStreamSynchronize(t->side) : hipSuccess;
    if (e == hipSuccess && t->side2) e = hipStreamSynchronize(t->side2);
    return e;
}

// A plan queued ahead of its frame (vf_terrain::pre) is thrown away: back to where the handle stood before it, and that set's segment list
// starts empty again (the stale k_block_boxes filled it; its k_clear, which would have emptied it, never runs).
static hipError_t drop_preplan(vf_terrain *t, hipStream_t next_plan_on = nullptr)
{
    t->preplan_pending = false;
    if (!t->pre.valid) return hipSuccess;
    const vf_terrain::PrePlan &R = t->pre;
    t->pre.valid = false;
    t->cur_set = R.cur_set; t->frame_no = R.frame_no; t->frames_since_reset = R.frames_since_reset;
    t->camera_moving = R.camera_moving; t->was_moving = R.was_moving; t->have_drawn = R.have_drawn;
    std::memcpy(t->u_drawn, R.u_drawn, sizeof t->u_drawn);
    // The stale plan's k_plan_sort has already zeroed this set's tile times and replaced its flag words: the frame that is planned in
    // its place reads the PREVIOUS frame's (complete: the caller was idle when the plan went out) -- plan_frame's `fresh` mode.
    t->replan_fresh = true;
    // The stale chain ran on `side`, the stale set-up pass on `side2` (a waiting caller's: both on the stream of its last read-back):
    // the frame planned next rewrites this set's block boxes, records and segment list, and must not do so while the stale set-up pass
    // still reads and writes them -- the stream its plan starts on waits for both.
    hipStream_t on = t->side ? t->side : (next_plan_on ? next_plan_on : (t->last_stream ? t->last_stream : t->ctx->stream));
    hipError_t e = hipStreamWaitEvent(on, t->ps[R.K.set].set_up, 0);
    if (e == hipSuccess) e = hipStreamWaitEvent(on, t->ps[R.K.set].planned, 0);
    if (e == hipSuccess) e = hipMemsetAsync(R.K.seg_count, 0, sizeof(uint32_t), on);
    return e;
}

// A plan state's buffers and events, made when the state is first used (frame 0: at construction; frame 1: by that frame).
static hipError_t ensure_plan_state(vf_terrain *t, uint32_t k, hipStream_t zero_on)
{
    vf_terrain::PlanState &S = t->ps[k];
    if (S.slab) return hipSuccess;
    const size_t all_tiles = (size_t)t->ntx * t->nty;
    const size_t nsegs = (size_t)t->nb * ((t->nb + kSegBlocks - 1) / kSegBlocks);
    Carver C;
    C.add((void **)&S.seg_list, (nsegs + 1) * sizeof(uint32_t), true);
    C.add((void **)&S.feedback, (all_tiles * 65 + 1) * sizeof(uint32_t), true);
    C.add((void **)&S.background, all_tiles * sizeof(uint32_t), true);
    C.add((void **)&S.ranges, t->nblocks * sizeof(PixelBox));
    C.add((void **)&S.vtx, (size_t)t->nblocks * kBlockStride * sizeof(VertexRec));
    C.add((void **)&S.recs, (size_t)t->nblocks * sizeof(BlockRec));
    C.add((void **)&S.gen, (size_t)t->nblocks * sizeof(ulonglong2));
    C.add((void **)&S.row_ranges, t->nb * sizeof(PixelBox));
    C.add((void **)&S.cap_seg, t->nblocks * sizeof(float4));
    C.add((void **)&S.cap_rad, t->nblocks * sizeof(float));
    C.add((void **)&S.rc, 2 * (size_t)t->nb * t->ntx * sizeof(uint32_t));
    C.add((void **)&S.work, 2 * (all_tiles + kSplitBudget + 16) * sizeof(uint2));   // tiles + strips created by splitting: as planned, and (second half) ordered
    C.add((void **)&S.work_count, 4 * sizeof(uint32_t));
    C.add((void **)&S.redo, (all_tiles + kSplitBudget) * sizeof(uint32_t));
    C.add((void **)&S.flags_new, all_tiles * sizeof(uint32_t));
    hipError_t err = C.commit(&S.slab, zero_on);          // (the stream of the state's first user: the plan chain)
    if (err != hipSuccess) { S.slab = nullptr; return err; }
    S.work_sorted = S.work + (all_tiles + kSplitBudget + 16);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&S.planned, hipEventDisableTiming);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&S.drawn, hipEventDisableTiming);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&S.boxed, hipEventDisableTiming);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&S.set_up, hipEventDisableTiming);
    return err;
}

static int refresh_tables(vf_terrain *t, hipStream_t s)
{
    hipLaunchKernelGGL(k_axis_tables, dim3((t->n + 255) / 256), dim3(256), 0, s, t->n, t->tw, t->th, t->d_xs, t->d_sinx,
                       t->d_cosz, t->d_txi, t->d_tyj);
    VF_HIP_TRY(hipGetLastError());
    t->bounds_dirty = true;
    return VF_OK;
}

int vf_terrain_create(vf_ctx *ctx, uint32_t width, uint32_t height, uint32_t grid, const uint8_t lut_rgba8[1024],
                      int lut_is_srgb, vf_terrain **out)
{
    if (!ctx || !out || !lut_rgba8) return fail(VF_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (width == 0 || height == 0 || width > 16384 || height > 16384) return fail(VF_ERR_INVALID, "width/height must be in 1..16384");
    uint32_t n = grid < 2 ? 2 : grid;   // `.max(2)`, src/terrain/mod.rs:260
    if (n > 8192) return fail(VF_ERR_INVALID, "grid must be <= 8192");
    VF_HIP_TRY(hipSetDevice(ctx->device));
    vf_terrain *t = new (std::nothrow) vf_terrain;
    if (!t) return fail(VF_ERR_NOMEM, "out of host memory");
    t->ctx = ctx; t->W = width; t->H = height; t->n = n;
    t->nb = (n - 1 + kBlockCells - 1) / kBlockCells;
    t->nblocks = t->nb * t->nb;
    t->ntx = (width + kTileW - 1) / kTileW;
    t->nty = (height + kTileH - 1) / kTileH;
    t->local_rows = height;
    t->local_tiles = t->ntx * t->nty;
    std::memset(t->u, 0, sizeof t->u);

    // linear LUT as the kernels stage it: 257 x {r, g, b, 0}, entry 256 = entry 255
    float lut[kLutFloats];
    for (int i = 0; i <= 256; ++i) {
        const int k = i < 256 ? i : 255;
        for (int ch = 0; ch < 3; ++ch)
            lut[kLutStride * i + ch] = lut_is_srgb ? tables().decode[lut_rgba8[4 * k + ch]] : (float)lut_rgba8[4 * k + ch] / 255.0f;
        lut[kLutStride * i + 3] = 0.0f;
    }
    // ONE device allocation for the handle's fixed-size buffers (the buffers that must start at zero lie first: one memset), one more per
    // plan state when that state is first used (ensure_plan_state) -- construction was ~40 hipMallocs and 4 memsets (round 4: unmeasured).
    hipError_t err = hipSuccess;
    const size_t all_tiles = (size_t)t->ntx * t->nty;
    Carver C;
    C.add((void **)&t->d_height_dummy, sizeof(float), true);      // 1x1 zero texture, src/terrain/mod.rs:342-378
    C.add((void **)&t->d_stats, (4 + 4 * (all_tiles + kSplitBudget) + 2 * kPhaseSlots + (t->nblocks + 31) / 32) * sizeof(uint32_t), true);   // (only [0..4) must be zero) + one bit per block: drawn this frame?
    C.add((void **)&t->d_xs, n * sizeof(float));
    C.add((void **)&t->d_sinx, n * sizeof(float));
    C.add((void **)&t->d_cosz, n * sizeof(float));
    C.add((void **)&t->d_txi, n * sizeof(int32_t));
    C.add((void **)&t->d_tyj, n * sizeof(int32_t));
    C.add((void **)&t->d_bounds, t->nblocks * sizeof(float2));
    C.add((void **)&t->d_hblk, (size_t)t->nblocks * kBlockStride * sizeof(float));
    C.add((void **)&t->d_lut, sizeof lut);
    C.add((void **)&t->d_rgba_own, all_tiles * kTileW * kTileH * sizeof(uint32_t));   // whole tiles: tile-major shards need the padding
    C.add((void **)&t->d_tile_map, all_tiles * sizeof(uint32_t));
    C.add((void **)&t->d_stripe_owner, kMaxStripes);
    err = C.commit(&t->slab, ctx->stream);
    // (the two side streams are made by the handle's SECOND frame: a stream costs ~2 ms to create and as much to destroy, and a
    //  handle's first frame -- the only one of the reference's construct / render_png once usage -- has nothing to overlap with)
    if (err == hipSuccess) err = hipMemcpyAsync(t->d_lut, lut, sizeof lut, hipMemcpyHostToDevice, ctx->stream);   // (pageable source: returns when the copy is staged)
    if (err == hipSuccess) err = hipEventCreateWithFlags(&t->entry, hipEventDisableTiming);
    // (both plan states now: a device allocation costs 0.03 ms whatever its size -- measured, round 5 -- while making the second one
    //  inside the handle's second frame cost that frame 0.7 ms)
    for (uint32_t k = 0; k < vf_terrain::kPlanStates && err == hipSuccess; ++k) err = ensure_plan_state(t, k, ctx->stream);
    if (err != hipSuccess) {
        std::string m = std::string("terrain allocation failed: ") + hipGetErrorString(err);
        vf_terrain_destroy(t);
        return fail(err == hipErrorOutOfMemory ? VF_ERR_NOMEM : VF_ERR_HIP, m);
    }
    t->d_rgba = t->d_rgba_own;
    t->d_height = t->d_height_dummy; t->tw = 1; t->th = 1;
    int rc = refresh_tables(t, ctx->stream);
    if (rc == VF_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(VF_ERR_HIP, "table setup failed");
    if (rc != VF_OK) { std::string keep = g_err; vf_terrain_destroy(t); g_err = keep; return rc; }
    *out = t;
    return VF_OK;
}

void vf_terrain_destroy(vf_terrain *t)
{
    if (!t) return;
    (void)hipSetDevice(t->ctx->device);
    (void)hipDeviceSynchronize();
    void *ptrs[] = { t->slab, t->d_height_own, t->d_vis, t->d_rgba_scratch, t->d_diag, t->d_xrecv, t->d_xband, t->d_batch[0], t->d_batch[1], t->d_batch[2] };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &e : t->batch_drawn) if (e) (void)hipEventDestroy(e);
    for (auto &e : t->batch_copied) if (e) (void)hipEventDestroy(e);

    for (auto &S : t->ps) {
        if (S.slab) (void)hipFree(S.slab);
        if (S.planned) (void)hipEventDestroy(S.planned);
        if (S.drawn) (void)hipEventDestroy(S.drawn);
        if (S.boxed) (void)hipEventDestroy(S.boxed);
        if (S.set_up) (void)hipEventDestroy(S.set_up);
        for (auto &e : S.pev) if (e) (void)hipEventDestroy(e);
    }
    // (side / side2 / copy_stream belong to the context)
    pinned_free(t->h_stage);
    for (auto &e : t->stage_ev) if (e) (void)hipEventDestroy(e);
    if (t->d_png) (void)hipFree(t->d_png);
    pinned_free(t->h_png);
    for (auto &f : t->ev) for (auto &e : f) if (e) (void)hipEventDestroy(e);
    if (t->entry) (void)hipEventDestroy(t->entry);
    if (t->copied) (void)hipEventDestroy(t->copied);
    for (auto &g : t->gprobe) { if (g.a) (void)hipEventDestroy(g.a); if (g.b) (void)hipEventDestroy(g.b); }
    delete t;
}

int vf_terrain_set_uniforms(vf_terrain *t, const float uniforms[44])
{
    if (!t || !uniforms) return fail(VF_ERR_INVALID, "NULL argument");
    if (!t->have_uniforms || std::memcmp(t->u, uniforms, sizeof t->u) != 0) t->inputs_gen++;     // (a plan made ahead was made for the old block)
    std::memcpy(t->u, uniforms, sizeof t->u);
    t->have_uniforms = true;
    return VF_OK;
}

static int set_height_common(vf_terrain *t, uint32_t tw, uint32_t th)
{
    t->inputs_gen++;
    VF_HIP_TRY(drop_preplan(t));
    t->g_epoch_frames = 0; t->g_epoch_id++; t->g_n[0] = t->g_n[1] = 0;     // other heights: which line loop is faster is measured again
    bool resized = tw != t->tw || th != t->th;
    t->tw = tw; t->th = th;
    if (resized) return refresh_tables(t, t->ctx->stream);
    t->bounds_dirty = true;
    return VF_OK;
}

int vf_terrain_set_height(vf_terrain *t, const float *host_height, uint32_t tw, uint32_t th)
{
    if (!t || !host_height) return fail(VF_ERR_INVALID, "NULL argument");
    if (tw == 0 || th == 0 || tw > 32768 || th > 32768) return fail(VF_ERR_INVALID, "height texture size must be in 1..32768");
    VF_HIP_TRY(hipSetDevice(t->ctx->device));
    VF_HIP_TRY(hipStreamSynchronize(t->last_stream ? t->last_stream : t->ctx->stream));
    VF_HIP_TRY(sync_sides(t));
    size_t bytes = (size_t)tw * th * sizeof(float);
    if ((size_t)t->tw * t->th != (size_t)tw * th || t->d_height != t->d_height_own) {
        float *fresh = nullptr;                            // allocate first: a failure leaves the handle as it was
        hipError_t e = hipMalloc(&fresh, bytes);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? VF_ERR_NOMEM : VF_ERR_HIP, std::string("height texture allocation failed: ") + hipGetErrorString(e));
        if (t->d_height_own) (void)hipFree(t->d_height_own);
        t->d_height_own = fresh;
    }
    t->d_height = t->d_height_own;
    VF_HIP_TRY(hipMemcpyAsync(t->d_height_own, host_height, bytes, hipMemcpyHostToDevice, t->ctx->stream));
    int rc = set_height_common(t, tw, th);
    if (rc == VF_OK) {
        // the per-block height cache and bounds depend on the texture alone: built here, behind the upload, not by the first frame
        // (a texture handed over in device memory may still be written by the caller's stream: that one is cached by the next frame)
        hipLaunchKernelGGL(k_height_blocks, dim3(t->nblocks), dim3(64), 0, t->ctx->stream, t->n, t->nb, t->tw, axis(t), t->d_height, t->d_hblk, t->d_bounds);
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) rc = fail(VF_ERR_HIP, std::string("k_height_blocks: ") + hipGetErrorString(le));   // (bounds stay dirty)
        else t->bounds_dirty = false;
    }
    VF_HIP_TRY(hipStreamSynchronize(t->ctx->stream));   // host buffer is only borrowed for this call
    return rc;
}

int vf_terrain_set_height_device(vf_terrain *t, const float *dev_height, uint32_t tw, uint32_t th)
{
    if (!t || !dev_height) return fail(VF_ERR_INVALID, "NULL argument");
    if (tw == 0 || th == 0 || tw > 32768 || th > 32768) return fail(VF_ERR_INVALID, "height texture size must be in 1..32768");
    VF_HIP_TRY(hipSetDevice(t->ctx->device));
    // a frame still in flight (caller's stream, or the plan / height-cache kernels on the side stream) reads the axis tables
    // and the old texture: let it finish before either changes
    VF_HIP_TRY(hipStreamSynchronize(t->last_stream ? t->last_stream : t->ctx->stream));
    VF_HIP_TRY(sync_sides(t));
    t->d_height = dev_height;
    int rc = set_height_common(t, tw, th);
    if (rc != VF_OK) return rc;
    VF_HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    return VF_OK;
}

int vf_terrain_set_shade_mode(vf_terrain *t, int mode)
{
    if (!t) return fail(VF_ERR_INVALID, "NULL argument");
    if (mode != VF_SHADE_REFERENCE && mode != VF_SHADE_SPEC_T32) return fail(VF_ERR_INVALID, "unknown shade mode");
    if (t->shade_mode != (uint32_t)mode) t->inputs_gen++;
    t->shade_mode = (uint32_t)mode;
    return VF_OK;
}

int vf_terrain_set_shade_precision(vf_terrain *t, int precision)
{
    if (!t) return fail(VF_ERR_INVALID, "NULL argument");
    if (precision != VF_PRECISION_EXACT && precision != VF_PRECISION_FAST) return fail(VF_ERR_INVALID, "unknown shade precision");
    if (t->precision != (uint32_t)precision) t->inputs_gen++;
    t->precision = (uint32_t)precision;
    return VF_OK;
}

int vf_terrain_set_raster_groups(vf_terrain *t, int mode)
{
    if (!t) return fail(VF_ERR_INVALID, "NULL argument");
    if (mode < -1 || mode > 1) return fail(VF_ERR_INVALID, "mode must be -1 (measure and choose), 0 or 1");
    t->groups_mode = mode;
    t->g_epoch_frames = 0; t->g_epoch_id++; t->g_n[0] = t->g_n[1] = 0; t->g_ms[0] = t->g_ms[1] = 0.0f;
    return VF_OK;
}

int vf_terrain_raster_groups(const vf_terrain *t, int *in_use, float ms[2])
{
    if (!t || !in_use) return fail(VF_ERR_INVALID, "NULL argument");
    *in_use = t->groups_now;
    if (ms) { ms[0] = t->g_n[0] ? t->g_ms[0] : 0.0f; ms[1] = t->g_n[1] ? t->g_ms[1] : 0.0f; }
    return VF_OK;
}

int vf_terrain_set_shard(vf_terrain *t, uint32_t rank, uint32_t nranks, uint32_t band_h)
{
    if (!t) return fail(VF_ERR_INVALID, "NULL argument");
    if (nranks == 0 || rank >= nranks) return fail(VF_ERR_INVALID, "rank must be < nranks");
    if (!is_pow2(band_h) || band_h < (uint32_t)kTileH) return fail(VF_ERR_INVALID, "band_h must be a power of two >= 64 (the tile height)");
    VF_HIP_TRY(hipSetDevice(t->ctx->device));
    VF_HIP_TRY(hipStreamSynchronize(t->last_stream ? t->last_stream : t->ctx->stream));
    t->inputs_gen++;
    VF_HIP_TRY(drop_preplan(t));
    t->rank = rank; t->nranks = nranks; t->band_h = band_h;
    t->shard_tiles = false; t->use_map = false;
    t->local_rows = compute_local_rows(t->H, rank, nranks, band_h);
    t->local_tiles = t->ntx * ((t->local_rows + kTileH - 1) / kTileH);
    t->rendered = false; t->have_frame = false;
    t->frames_since_reset = 0;
    // tile numbering changed: forget the scheduling feedback of the previous layout
    VF_HIP_TRY(sync_sides(t));
    for (auto &S : t->ps) {
        if (!S.slab) continue;                             // (a state not used yet starts at zero when it is made)
        VF_HIP_TRY(hipMemset(S.feedback, 0, ((size_t)t->ntx * t->nty * 65 + 1) * sizeof(uint32_t)));
        VF_HIP_TRY(hipMemset(S.background, 0, (size_t)t->ntx * t->nty * sizeof(uint32_t)));
        S.have_u_used = false;
    }
    return VF_OK;
}

int vf_tile_layout(uint32_t width, uint32_t height, uint32_t rank, uint32_t nranks, uint32_t skew, uint32_t *tiles, uint32_t capacity,
                   uint32_t *count)
{
    if (!count) return fail(VF_ERR_INVALID, "NULL argument");
    if (width == 0 || height == 0 || nranks == 0 || rank >= nranks) return fail(VF_ERR_INVALID, "empty frame or rank >= nranks");
    if (!layout_valid(skew)) return fail(VF_ERR_INVALID, "layout word is neither VF_TILE_LAYOUT(skew < 65536, stripe_log2 <= 15) nor a registered stripe map");
    const uint32_t ntx = (width + kTileW - 1) / kTileW, nty = (height + kTileH - 1) / kTileH;
    if (ntx > 0xFFFFu || nty > 0xFFFFu) return fail(VF_ERR_INVALID, "frame too large");
    const StripeMap *m = layout_map(skew);
    if (m && (m->nranks != nranks || ((ntx - 1u) >> m->shift) >= m->owner.size()))
        return fail(VF_ERR_INVALID, "the registered stripe map was made for another number of ranks or a narrower frame");
    uint32_t n = 0;
    for (uint32_t ty = 0; ty < nty; ++ty)
        for (uint32_t tx = 0; tx < ntx; ++tx)
            if (layout_owner(skew, m, tx, ty, nranks) == rank) {
                if (tiles && n < capacity) tiles[n] = tx | (ty << 16);
                ++n;
            }
    *count = n;
    return VF_OK;
}

int vf_tile_layout_register_map(const uint8_t *stripe_owner, uint32_t nstripes, uint32_t stripe_log2, uint32_t nranks, uint32_t *layout)
{
    if (!stripe_owner || !layout) return fail(VF_ERR_INVALID, "NULL argument");
    if (nstripes == 0 || nstripes > kMaxStripes || stripe_log2 > 15u || nranks == 0 || nranks > 255u) return fail(VF_ERR_INVALID, "1..256 stripes, stripe_log2 <= 15, 1..255 ranks");
    for (uint32_t k = 0; k < nstripes; ++k) if (stripe_owner[k] >= nranks) return fail(VF_ERR_INVALID, "a stripe's owner is not a rank");
    std::lock_guard<std::mutex> lk(g_maps_mu);
    g_maps.reserve(kMaxStripeMaps);                        // (never reallocated: layout_map hands out pointers)
    for (uint32_t id = 0; id < g_maps.size(); ++id)        // the same table again: the same word (every rank registers its own copy)
        if (g_maps[id].shift == stripe_log2 && g_maps[id].nranks == nranks && g_maps[id].owner.size() == nstripes &&
            std::memcmp(g_maps[id].owner.data(), stripe_owner, nstripes) == 0) { *layout = kLayoutMapBit | (id << 21) | (stripe_log2 << 16); return VF_OK; }
    if (g_maps.size() >= kMaxStripeMaps) return fail(VF_ERR_INVALID, "too many registered stripe maps (64 per process)");
    StripeMap m;
    m.owner.assign(stripe_owner, stripe_owner + nstripes); m.shift = stripe_log2; m.nranks = nranks;
    g_maps.push_back(std::move(m));
    *layout = kLayoutMapBit | ((uint32_t)(g_maps.size() - 1u) << 21) | (stripe_log2 << 16);
    return VF_OK;
}

int vf_balance_stripes(const float *stripe_ms, uint32_t nstripes, uint32_t nranks, uint8_t *stripe_owner)
{
    if (!stripe_ms || !stripe_owner) return fail(VF_ERR_INVALID, "NULL argument");
    if (nranks == 0 || nranks > 255u || nstripes == 0 || nstripes > kMaxStripes || nstripes % nranks) return fail(VF_ERR_INVALID, "the stripes must divide evenly among 1..255 ranks");
    // Heaviest stripe first, each to the least loaded rank that still has room (every rank ends up with nstripes / nranks stripes: the
    // slabs and the exchange's chunks keep their sizes); then stripes of the most loaded rank are swapped against lighter ones of the
    // others for as long as that lowers the heavier of the two loads; and the round-robin deal itself is kept when nothing beat it.
    // Ties by stripe number / rank number: the same input gives the same table on every rank.  Times that are not finite count as zero.
    std::vector<uint32_t> order(nstripes);
    for (uint32_t k = 0; k < nstripes; ++k) order[k] = k;
    auto cost = [&](uint32_t k) { const float v = stripe_ms[k]; return std::isfinite(v) && v > 0.0f ? (double)v : 0.0; };
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost(a) > cost(b); });
    std::vector<double> load(nranks, 0.0);
    std::vector<uint32_t> have(nranks, 0u);
    std::vector<uint8_t> own(nstripes, 0);
    const uint32_t room = nstripes / nranks;
    for (uint32_t k : order) {
        uint32_t best = nranks;
        for (uint32_t r = 0; r < nranks; ++r)
            if (have[r] < room && (best == nranks || load[r] < load[best])) best = r;
        own[k] = (uint8_t)best; load[best] += cost(k); have[best]++;
    }
    for (uint32_t round = 0; round < 4u * nstripes; ++round) {
        uint32_t hot = 0;
        for (uint32_t r = 1; r < nranks; ++r) if (load[r] > load[hot]) hot = r;
        double gain = 1e-12 * (load[hot] + 1.0);
        uint32_t sa = nstripes, sb = nstripes;
        for (uint32_t a = 0; a < nstripes; ++a) {
            if (own[a] != hot) continue;
            for (uint32_t b = 0; b < nstripes; ++b) {
                const uint32_t r = own[b];
                if (r == hot || !(cost(b) < cost(a))) continue;
                const double d = cost(a) - cost(b);         // moves from the hot rank to r
                const double worst = std::fmax(load[hot] - d, load[r] + d);
                if (load[hot] - worst > gain) { gain = load[hot] - worst; sa = a; sb = b; }
            }
        }
        if (sa == nstripes) break;
        const uint32_t r = own[sb];
        const double d = cost(sa) - cost(sb);
        load[hot] -= d; load[r] += d; own[sa] = (uint8_t)r; own[sb] = (uint8_t)hot;
    }
    std::vector<double> rr(nranks, 0.0);
    for (uint32_t k = 0; k < nstripes; ++k) rr[k % nranks] += cost(k);
    const bool keep_round_robin = *std::max_element(rr.begin(), rr.end()) <= *std::max_element(load.begin(), load.end());
    for (uint32_t k = 0; k < nstripes; ++k) stripe_owner[k] = keep_round_robin ? (uint8_t)(k % nranks) : own[k];
    return VF_OK;
}

int vf_terrain_tile_times(vf_terrain *t, float *ms, uint32_t capacity, uint32_t *count)
{
    if (!t || !count) return fail(VF_ERR_INVALID, "NULL argument");
    if (!t->rendered) return fail(VF_ERR_INVALID, "nothing rendered yet");
    int rc = vf_terrain_sync(t);
    if (rc != VF_OK) return rc;
    const uint32_t n = std::min(capacity, t->local_tiles);
    std::vector<uint32_t> ticks(n);
    if (n && ms) {
        VF_HIP_TRY(hipMemcpy(ticks.data(), t->ps[t->last_set].feedback, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint32_t k = 0; k < n; ++k) ms[k] = (float)ticks[k] * 1e-5f;      // 10 ns ticks of the 100 MHz clock
    }
    *count = t->local_tiles;
    return VF_OK;
}

int vf_terrain_set_tile_shard(vf_terrain *t, uint32_t rank, uint32_t nranks, uint32_t skew)
{
    if (!t) return fail(VF_ERR_INVALID, "NULL argument");
    if (nranks == 0 || rank >= nranks) return fail(VF_ERR_INVALID, "rank must be < nranks");
    if (!layout_valid(skew)) return fail(VF_ERR_INVALID, "layout word is neither VF_TILE_LAYOUT(skew < 65536, stripe_log2 <= 15) nor a registered stripe map");
    VF_HIP_TRY(hipSetDevice(t->ctx->device));
    VF_HIP_TRY(hipStreamSynchronize(t->last_stream ? t->last_stream : t->ctx->stream));
    std::vector<uint32_t> map((size_t)t->ntx * t->nty);
    uint32_t n = 0;
    int rc = vf_tile_layout(t->W, t->H, rank, nranks, skew, map.data(), (uint32_t)map.size(), &n);
    if (rc != VF_OK) return rc;
    VF_HIP_TRY(sync_sides(t));                                // (k_block_boxes of a frame in flight reads the owner table)
    const StripeMap *sm = layout_map(skew);
    if (sm) VF_HIP_TRY(hipMemcpy(t->d_stripe_owner, sm->owner.data(), sm->owner.size(), hipMemcpyHostToDevice));
    t->use_map = sm != nullptr;
    if (n) VF_HIP_TRY(hipMemcpy(t->d_tile_map, map.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
    t->inputs_gen++;
    VF_HIP_TRY(drop_preplan(t));
    t->shard_tiles = true; t->local_tiles = n;
    t->rank = rank; t->nranks = nranks; t->skew = skew;
    t->local_rows = 0;                                   // row-oriented accessors do not apply to a tile-major buffer
    t->rendered = false; t->have_frame = false;
    t->frames_since_reset = 0;
    VF_HIP_TRY(sync_sides(t));
    for (auto &S : t->ps) {
        if (!S.slab) continue;                             // (a state not used yet starts at zero when it is made)
        VF_HIP_TRY(hipMemset(S.feedback, 0, ((size_t)t->ntx * t->nty * 65 + 1) * sizeof(uint32_t)));
        VF_HIP_TRY(hipMemset(S.background, 0, (size_t)t->ntx * t->nty * sizeof(uint32_t)));
        S.have_u_used = false;
    }
    return VF_OK;
}

int vf_terrain_local_tiles(const vf_terrain *t, uint32_t *tiles)
{
    if (!t || !tiles) return fail(VF_ERR_INVALID, "NULL argument");
    *tiles = t->local_tiles;
    return VF_OK;
}

int vf_terrain_read_tiles(vf_terrain *t, uint8_t *dst, uint32_t first, uint32_t count)
{
    if (!t || !dst) return fail(VF_ERR_INVALID, "NULL argument");
    if (!t->shard_tiles) return fail(VF_ERR_INVALID, "handle is not tile-sharded");
    if (!t->rendered) return fail(VF_ERR_INVALID, "nothing rendered yet");
    if ((uint64_t)first + count > t->local_tiles) return fail(VF_ERR_INVALID, "tile range outside the local tiles");
    int rc = vf_terrain_sync(t);
    if (rc != VF_OK) return rc;
    const size_t tile_px = (size_t)kTileW * kTileH;
    VF_HIP_TRY(hipMemcpy(dst, t->d_rgba + first * tile_px, count * tile_px * 4, hipMemcpyDeviceToHost));
    return VF_OK;
}

int vf_terrain_local_rows(const vf_terrain *t, uint32_t *rows)
{
    if (!t || !rows) return fail(VF_ERR_INVALID, "NULL argument");
    *rows = t->local_rows;
    return VF_OK;
}

int vf_terrain_set_output_device(vf_terrain *t, void *dev_rgba)
{
    if (!t) return fail(VF_ERR_INVALID, "NULL argument");
    t->d_rgba = dev_rgba ? (uint32_t *)dev_rgba : t->d_rgba_own;
    return VF_OK;
}

int vf_terrain_rgba_device(const vf_terrain *t, void **dev_rgba)
{
    if (!t || !dev_rgba) return fail(VF_ERR_INVALID, "NULL argument");
    *dev_rgba = t->d_rgba;
    return VF_OK;
}

static void build_params(const vf_terrain *t, FrameParams &P)
{
    const float *u = t->u;
    std::memcpy(P.view, u, 64);
    std::memcpy(P.proj, u + 16, 64);
    P.spacing = std::fmax(u[36], 1e-8f);       // terrain.wgsl:46
    P.exag = u[38];
    P.h_range = std::fmax(u[37], 1e-8f);       // terrain.wgsl:71
    P.exposure = u[35];
    {   // L = normalize(sun), terrain.wgsl:83 -- uniform per frame, evaluated once with the same IEEE ops
        float sx = u[32], sy = u[33], sz = u[34];
        float inv = 1.0f / std::sqrt(std::fmaf(sz, sz, std::fmaf(sy, sy, sx * sx)));
        P.Lx = sx * inv; P.Ly = sy * inv; P.Lz = sz * inv;
    }
    P.hw = 0.5f * (float)t->W; P.hh = 0.5f * (float)t->H;
    P.step = (2.0f * 1.5f) / ((float)t->n - 1.0f);
    P.n = t->n; P.nm1 = t->n - 1; P.nb = t->nb;
    P.W = t->W; P.H = t->H; P.ntx = t->ntx; P.nty = t->nty; P.tw = t->tw; P.th = t->th;
    P.rank = t->rank; P.nranks = t->nranks; P.band_h = t->band_h; P.band_shift = ilog2(t->band_h);
    P.local_rows = t->local_rows;
    P.shard_tiles = t->shard_tiles ? 1u : 0u; P.tile_map = t->d_tile_map; P.skew = layout_skew(t->skew); P.stripe_shift = layout_shift(t->skew);
    P.stripe_owner = t->shard_tiles && t->use_map ? t->d_stripe_owner : nullptr;
    P.shade_mode = t->shade_mode; P.tex = t->d_height;
    P.inv2hr = 1.0f / (2.0f * P.h_range);
    {   // cell / nm1 as mulhi(cell, m) >> s, exact for cell < 2^26 (nm1 < 2^13): s = 31 + ceil(log2 nm1) - 32, m = ceil(2^(s + 32) / nm1)
        const uint32_t d = P.nm1;
        if (d <= 1u) { P.div_m = 0u; P.div_s = 0u; }       // one cell: row 0
        else {
            const uint32_t L = ilog2(d);                   // ceil(log2 d)
            P.div_s = L - 1u;
            P.div_m = (uint32_t)((((uint64_t)1 << (31u + L)) + d - 1u) / d);
        }
    }
    const SrgbTables &T = tables();
    P.clear_rgba = T.encode(0.02f) | (T.encode(0.02f) << 8) | (T.encode(0.03f) << 16) | 0xFF000000u;   // src/terrain/mod.rs:421
}

// How far, in pixels, the terrain's footprint moved on the screen between two cameras (corners of the xz square at y = 0).
// Scheduling feedback is per screen tile: once the picture has moved by about a tile, an older frame's tile times describe
// other content.  A corner behind either camera counts as "moved".
static float camera_shift_px(const vf_terrain *t, const float *a, const float *b)
{
    const float ext = 1.5f * std::fmax(t->u[36], 1e-8f);
    float worst = 0.0f;
    for (int c = 0; c < 4; ++c) {
        const float p[4] = { (c & 1) ? ext : -ext, 0.0f, (c & 2) ? ext : -ext, 1.0f };
        float xy[2][2];
        for (int k = 0; k < 2; ++k) {
            const float *u = k ? b : a;                    // column-major view (u[0..16)) and proj (u[16..32))
            float v[4], q[4];
            for (int r = 0; r < 4; ++r) v[r] = u[r] * p[0] + u[4 + r] * p[1] + u[8 + r] * p[2] + u[12 + r] * p[3];
            for (int r = 0; r < 4; ++r) q[r] = u[16 + r] * v[0] + u[20 + r] * v[1] + u[24 + r] * v[2] + u[28 + r] * v[3];
            if (!(q[3] > 1e-6f)) return 1e9f;
            xy[k][0] = q[0] / q[3] * 0.5f * (float)t->W; xy[k][1] = q[1] / q[3] * 0.5f * (float)t->H;
        }
        worst = std::fmax(worst, std::fmax(std::fabs(xy[0][0] - xy[1][0]), std::fabs(xy[0][1] - xy[1][1])));
    }
    return worst;
}
// Homography of the ground plane y = 0 from the screen of camera `now` to the screen of camera `then` (both: view u[0..16), proj
// u[16..32), column-major; pixels): a point (X, 0, Z) of the plane projects to A G (X, Z, 1) with G the columns x, z, w / rows x, y, w
// of proj * view and A the viewport; the map is (A G_then) (A G_now)^-1.  False when the plane is (nearly) edge-on for `now`.
static bool motion_map(const vf_terrain *t, const float *then, const float *now, MotionMap &M)
{
    const double hw = 0.5 * t->W, hh = 0.5 * t->H;
    auto plane = [&](const float *u, double g[9]) {
        double vp[16];                                     // proj * view, column-major
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) {
                double a = 0.0;
                for (int k = 0; k < 4; ++k) a += (double)u[16 + 4 * k + r] * (double)u[4 * c + k];
                vp[4 * c + r] = a;
            }
        const int cols[3] = { 0, 2, 3 };
        for (int j = 0; j < 3; ++j) {
            const double cx = vp[4 * cols[j] + 0], cy = vp[4 * cols[j] + 1], cw = vp[4 * cols[j] + 3];
            g[0 + j] = hw * cx + hw * cw; g[3 + j] = -hh * cy + hh * cw; g[6 + j] = cw;      // row-major 3 x 3
        }
    };
    double a[9], b[9];
    plane(then, a); plane(now, b);
    const double det = b[0] * (b[4] * b[8] - b[5] * b[7]) - b[1] * (b[3] * b[8] - b[5] * b[6]) + b[2] * (b[3] * b[7] - b[4] * b[6]);
    double scale = 0.0;
    for (double v : b) scale = std::fmax(scale, std::fabs(v));
    if (!(std::fabs(det) > 1e-12 * scale * scale * scale)) return false;
    const double inv[9] = { (b[4] * b[8] - b[5] * b[7]) / det, (b[2] * b[7] - b[1] * b[8]) / det, (b[1] * b[5] - b[2] * b[4]) / det,
                            (b[5] * b[6] - b[3] * b[8]) / det, (b[0] * b[8] - b[2] * b[6]) / det, (b[2] * b[3] - b[0] * b[5]) / det,
                            (b[3] * b[7] - b[4] * b[6]) / det, (b[1] * b[6] - b[0] * b[7]) / det, (b[0] * b[4] - b[1] * b[3]) / det };
    double m[9], big = 0.0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { m[3 * r + c] = a[3 * r] * inv[c] + a[3 * r + 1] * inv[3 + c] + a[3 * r + 2] * inv[6 + c]; big = std::fmax(big, std::fabs(m[3 * r + c])); }
    if (!(big > 0.0) || !std::isfinite(big)) return false;
    // (homogeneous: any POSITIVE scale -- w' = w_then / w_now is positive for plane points in front of both cameras, which is what
    //  k_plan asks of a landing point; keep the floats in range)
    for (int k = 0; k < 9; ++k) M.m[k] = (float)(m[k] / big);
    M.on = 1u;
    return true;
}
constexpr float kFreshFeedbackPx = 24.0f;   // from here on (3/8 of a tile per frame) the plan waits for the previous frame's feedback

#ifndef VF_GROUPS_MAX_RANKS
#define VF_GROUPS_MAX_RANKS 8    // handles sharded over this many ranks or more take the tile kernel without line groups
#endif
// the fast fragment path exists for fs_main as coded; the documented-only SPEC_T32 stage always takes the exact arithmetic
static bool fast_shading(const vf_terrain *t) { return t->precision == VF_PRECISION_FAST && t->shade_mode == VF_SHADE_REFERENCE; }

// A frame in two halves (round 5).  plan_frame: everything up to the plan's last kernel -- block boxes, set-up pass, plan, sort -- on the
// side streams (a handle's first frame: on `s`); it touches plan state only.  draw_frame: the kernels on the caller's stream.  What the
// second half needs of the first travels in a FramePlan, so that the first half of the NEXT frame can be queued ahead of its call
// (vf_terrain::pre, render_impl).
static int plan_frame(vf_terrain *t, hipStream_t s, FramePlan &K, bool ahead = false, bool streaming = false)
{
    FrameParams &P = K.P;
    build_params(t, P);
    AxisTables A = axis(t);
    const uint32_t ntiles = t->local_tiles;
    const uint32_t set = t->cur_set;
    t->cur_set = (set + 1u) % vf_terrain::kPlanStates;
    t->frame_no++;
    // A handle's FIRST frame runs on the caller's stream alone, plan and set-up included: its chain is serial anyway (the static plan
    // estimate reads the set-up pass's records) and there is no earlier frame to hide anything under.  The side streams and the second
    // plan chain's events come into play with the second frame.
    // ... and so do the frames of a caller that waits for each of them (round 6): the plan streams are made for the caller that
    // streams -- a frame arriving while the previous one is in flight -- or borrowed when the context already has them.
    if (!t->side && t->frame_no > 1u) {
        bool have = false;
        { std::lock_guard<std::mutex> lk(t->ctx->lazy_mu); have = t->ctx->side && t->ctx->side2; }
        if (have || streaming || t->want_plan_streams) VF_HIP_TRY(ctx_side_streams(t->ctx, &t->side, &t->side2));   // (the context's: made once per process)
    }
    const bool solo = !t->side;
    VF_HIP_TRY(ensure_plan_state(t, set, solo ? s : t->side));
    vf_terrain::PlanState &S = t->ps[set], &O = t->ps[t->last_set];       // this frame's plan state, the previous frame's
    // A camera at rest (or moving slowly) plans under the previous frame's tile kernel with the feedback of the frame before it;
    // a camera that moves the picture by a good part of a tile per frame waits for the previous frame instead and uses ITS feedback:
    // the plan then costs its own time (k_plan + k_plan_sort after the tile kernel), stale feedback costs more (64-pose orbit
    // at 1920x1080, grid 2048: 0.80 -> 0.61 ms per pose together with the dilated weights in k_plan; tools/exp_orbit.py).
    float shift = 0.0f;
    if (t->have_drawn) {
        shift = camera_shift_px(t, t->u_drawn, t->u);
        if (shift > kFreshFeedbackPx) t->camera_moving = true;                 // hysteresis: frames that alternate between the two
        else if (shift < 0.5f * kFreshFeedbackPx) t->camera_moving = false;    // modes get the worst of both
    }
    // (one more frame after the motion stops: the frame before last still shows the old view, the last one the new)
    // ... and the second frame of a handle: its own plan state has no times yet, the first frame's has
    // ... and the frames of a handle whose own plan state has no times yet (the sets take turns): the previous frame's has
    // Round 5: a moving camera's plan no longer waits either when this set's times can be looked up THROUGH the motion (k_plan,
    // MotionMap): the homography of the ground plane between this frame's screen and the screen of the frame before last.  What is
    // left of the waiting mode: a handle's second frame, sharded handles (their feedback is per local tile), a jump cut (the view of
    // two frames ago shares little with this one: the previous frame's times, dilated, are the better guess), an edge-on plane.
    const bool young = t->frames_since_reset >= 1 && t->frames_since_reset < vf_terrain::kPlanStates;
    MotionMap M;
    std::memset(&M, 0, sizeof M);
    if ((t->camera_moving || t->was_moving) && !young && t->frames_since_reset >= vf_terrain::kPlanStates && S.have_u_used && t->nranks == 1u && !t->shard_tiles &&
        camera_shift_px(t, S.u_used, t->u) < 0.4f * (float)std::max(t->W, t->H))
        (void)motion_map(t, S.u_used, t->u, M);
#ifdef VF_EXPERIMENTS
    if (std::getenv("VF_NO_MOTION_MAP")) M.on = 0u;
#endif
    const bool after_drop = t->replan_fresh && t->frames_since_reset >= vf_terrain::kPlanStates;   // (a handle's first frames have their own rules)
    t->replan_fresh = false;
    if (after_drop) M.on = 0u;                              // (the motion map reads this set's times too)
    const bool fresh = young || after_drop || ((t->camera_moving || t->was_moving) && !M.on);
#ifdef VF_EXPERIMENTS
    const bool first = t->frames_since_reset == 0 && !std::getenv("VF_NO_STATIC_PLAN");
#else
    const bool first = t->frames_since_reset == 0;
#endif   // no tile times at all yet: a static estimate stands in (k_plan_estimate)
    t->frames_since_reset++;
    const bool motion_starts = t->camera_moving && !t->was_moving;
    t->was_moving = t->camera_moving;
    const bool dilate = fresh || shift > 0.5f * kFreshFeedbackPx;     // slower motion: still overlapped, but the tile weights spread to the neighbours
    std::memcpy(t->u_drawn, t->u, sizeof t->u_drawn);
    t->have_drawn = true;
    hipStream_t side = solo ? s : t->side, side2 = solo ? s : t->side2;
    // (the plan's three timing events belong to its plan state, not to a slot of the frame ring: a plan queued ahead of its call has no
    //  place in the ring yet -- advisor, round 5 -- and the ring's events sit on the draw stream, where every record costs the frame)
    K.sampled = t->timing_every <= 1u || t->frame_no % t->timing_every == 0u;
    (void)ahead;
    // (not at timing level 3: an event record between two kernels of the plan chain delays the chain, and a plan that is late holds its frame's
    //  tile kernel back -- three records per plan cost a C4 frame 2.5 %, on the plan streams as on the draw stream)
    const bool timed = t->timing && t->timing_every <= 1u && S.pev[0] != nullptr;
    hipEvent_t *ev = S.pev;
    S.pev_valid = timed;
    // ---- plan, on the side stream: needs this set back from the frame before last, then touches plan state only ----
    VF_HIP_TRY(hipStreamWaitEvent(side, S.drawn, 0));
    if (t->bounds_dirty) {
        // the height texture may have been produced by work queued on the caller's stream: order the cache rebuild after it
        VF_HIP_TRY(hipEventRecord(t->entry, s));
        VF_HIP_TRY(hipStreamWaitEvent(side, t->entry, 0));
        hipLaunchKernelGGL(k_height_blocks, dim3(t->nblocks), dim3(64), 0, side, t->n, t->nb, t->tw, A, t->d_height, t->d_hblk, t->d_bounds);
        VF_HIP_TRY(hipGetLastError());
        t->bounds_dirty = false;
    }
    if (timed) VF_HIP_TRY(hipEventRecord(ev[0], side));
    const size_t rc_n = (size_t)t->nb * t->ntx;
    uint32_t *rc_lo = S.rc, *rc_hi = S.rc + rc_n;
    // the split quantum comes from the same tile times the plan will read: summed by an extra workgroup of k_block_boxes, or --
    // when those times belong to the frame still being drawn -- by a kernel of its own after the wait below
    uint32_t *quantum = S.feedback + (size_t)t->ntx * t->nty;
    const uint32_t nsegs_all = t->nb * ((t->nb + kSegBlocks - 1) / kSegBlocks);
    uint32_t *seg_count = S.seg_list + nsegs_all;
    hipLaunchKernelGGL(k_block_boxes, dim3(t->nb + 1), dim3(t->nb > 256 ? 512 : 256), 0, side, P, t->d_bounds, S.ranges, S.row_ranges, S.cap_seg, S.cap_rad, rc_lo, rc_hi,
                       fresh || first ? (const uint32_t *)nullptr : S.feedback, t->ntx * t->nty, quantum, S.work_count, S.recs, S.seg_list, seg_count);
    // vertex stage + tile-independent culling, once per frame (streams ~1.3 KB per block into this frame's plan state): needs the
    // block boxes only, so it runs on a second stream beside k_plan / k_plan_sort -- all of them under the previous frame's tile kernel
    VF_HIP_TRY(hipEventRecord(S.boxed, side));
    VF_HIP_TRY(hipStreamWaitEvent(side2, S.boxed, 0));                  // (orders it after S.drawn and the height cache too)
    // (one short-lived workgroup per possible segment -- those beyond the list's length leave at once: workgroups that come and go
    //  share the CUs with the previous frame's tile kernel more smoothly than a few long-lived ones)
    // (experiments only, VF_DBG_NO_SETUP=1: a camera at rest re-creates the same records in the same buffers, so after the first frames
    //  the pass can be left out to see what the frame costs without it -- the picture stays right, the time is a lower bound)
#ifdef VF_EXPERIMENTS
    static const bool dbg_no_setup = std::getenv("VF_DBG_NO_SETUP") != nullptr;
#else
    constexpr bool dbg_no_setup = false;
#endif
    if (!(dbg_no_setup && t->frames_since_reset > 6))
    hipLaunchKernelGGL(k_block_setup, dim3(nsegs_all), dim3(kSetupThreads), 0, side2,
                       P, t->d_hblk, S.ranges, S.vtx, S.recs, S.gen, S.seg_list, seg_count);
    VF_HIP_TRY(hipEventRecord(S.set_up, side2));
    if (timed) VF_HIP_TRY(hipEventRecord(ev[1], side));
    if (ntiles) {
        if (fresh) {
            VF_HIP_TRY(hipStreamWaitEvent(side, O.drawn, 0));         // (the block boxes above did not need to wait)
            hipLaunchKernelGGL(k_quantum, dim3(1), dim3(512), 0, side, O.feedback, t->ntx * t->nty, quantum);
        } else if (first) {
            VF_HIP_TRY(hipStreamWaitEvent(side, S.set_up, 0));       // the estimate reads the set-up pass's block records
            hipLaunchKernelGGL(k_plan_estimate, dim3(ntiles), dim3(256), 0, side, P, S.row_ranges, S.recs, S.cap_seg, S.cap_rad, rc_lo, rc_hi, S.feedback);
            hipLaunchKernelGGL(k_quantum, dim3(1), dim3(512), 0, side, S.feedback, t->ntx * t->nty, quantum);
        }
        const vf_terrain::PlanState &F = fresh ? O : S;               // whose tile times steer this frame
        hipLaunchKernelGGL(k_plan, dim3(ntiles), dim3(256), 0, side, P, S.row_ranges, S.flags_new, S.work, S.work_count,
                           F.feedback, quantum, S.work_count + 1, rc_lo, rc_hi, dilate ? 1u : 0u, F.background, M);
        hipLaunchKernelGGL(k_plan_sort, dim3(1), dim3(256), 0, side, S.work, S.work_sorted, S.work_count, S.feedback, t->ntx * t->nty,
                           S.flags_new, S.background, ntiles);
    }
    VF_HIP_TRY(hipGetLastError());                              // a failed plan launch is reported here: the probe block below clears hipEventQuery's "not ready"
    if (timed) VF_HIP_TRY(hipEventRecord(ev[2], side));
    VF_HIP_TRY(hipEventRecord(S.planned, side));
    K.set = set; K.ntiles = ntiles; K.solo = solo; K.motion_starts = motion_starts; K.rc_lo = rc_lo; K.rc_hi = rc_hi; K.seg_count = seg_count;
    return VF_OK;
}

static int draw_frame(vf_terrain *t, hipStream_t s, const FramePlan &K, bool write_vis)
{
    const FrameParams &P = K.P;
    vf_terrain::PlanState &S = t->ps[K.set];
    const uint32_t ntiles = K.ntiles, set = K.set;
    const bool motion_starts = K.motion_starts, solo = K.solo;
    uint32_t *const rc_lo = K.rc_lo, *const rc_hi = K.rc_hi, *const seg_count = K.seg_count;
    // (a plan queued ahead of its call may have been made before timing was switched on, or for another position of the ring: its three
    //  events are then recorded here -- valid, if not telling -- so that the frame's entry in the ring is complete)
    const uint32_t slot_now = t->timed_frames % (uint32_t)vf_terrain::kTimingRing;
    hipEvent_t *ev = t->ev[slot_now];
    const bool timing_now = t->timing && K.sampled;
    (void)solo;
    // ---- draw, on the caller's stream: everything that touches the output buffers ----
    uint32_t *stats = t->timing && t->stats_on ? t->d_stats : nullptr;
    const uint32_t nstats = (uint32_t)(4 + 4 * ((size_t)t->ntx * t->nty + kSplitBudget) + 2 * kPhaseSlots + (t->nblocks + 31) / 32);   // zeroed by k_clear (no memset dispatch)
    VF_HIP_TRY(hipStreamWaitEvent(s, S.planned, 0));
    VF_HIP_TRY(hipStreamWaitEvent(s, S.set_up, 0));
    // The previous frame may have been drawn on another stream of the caller's: this frame waits for it (same output / statistics
    // buffers; and when it went to another output buffer, letting the two tile kernels overlap costs more than it gains -- the
    // experiment behind VF_OVERLAP_FRAMES, tools/exp_overlap.py).
#ifdef VF_EXPERIMENTS
    static const bool overlap_frames = std::getenv("VF_OVERLAP_FRAMES") != nullptr;
#else
    constexpr bool overlap_frames = false;
#endif
    if (t->last_stream && t->last_stream != s && t->rendered && (!overlap_frames || t->last_out == t->d_rgba || stats || write_vis || !t->last_out))
        VF_HIP_TRY(hipStreamWaitEvent(s, t->ps[t->last_set].drawn, 0));
    if (timing_now) VF_HIP_TRY(hipEventRecord(ev[4], s));
    // line groups in the raster's line loop (vf_kernels.h, raster_fast): whole frames and shards of few ranks -- wide items, triangles
    // with many lines -- gain from them (C4: one GPU -2 %, top-down camera -7 %); a rank of many mostly draws narrow strips, whose
    // triangles have a handful of lines, and is better off with the leaner kernel (VF_GROUPS=0 / 1 overrides)
#ifdef VF_EXPERIMENTS
    static const int groups_env = std::getenv("VF_GROUPS") ? std::atoi(std::getenv("VF_GROUPS")) : -1;
#else
    constexpr int groups_env = -1;
#endif
    const int forced = groups_env >= 0 ? groups_env : t->groups_mode;
    // the variant by default: groups for whole frames and shards of few ranks, none for a rank of many (mostly narrow strips)
    const int guess = t->nranks < (uint32_t)VF_GROUPS_MAX_RANKS ? 1 : 0;
    int pick = guess;
    if (forced >= 0) pick = forced != 0;
    else {
        // probes that have completed (frames behind us: never a wait): every probed frame is a sample -- the time its kernels took on
        // the draw stream.  (hipEventQuery's "not ready" is cleared below: the plan launches' errors were collected above.)
        for (auto &g : t->gprobe)
            if (g.pending && hipEventQuery(g.b) == hipSuccess) {
                g.pending = false;
                float ms = 0.0f;
                if (g.epoch == t->g_epoch_id && hipEventElapsedTime(&ms, g.a, g.b) == hipSuccess && ms > 0.0f) {
                    // the probe window's samples count alike; a later look weighs as much as all before it (the view may have drifted)
                    const uint32_t n = ++t->g_n[g.variant];
                    t->g_ms[g.variant] += (ms - t->g_ms[g.variant]) / (float)(n <= 8u ? n : 2u);
                }
            }
        (void)hipGetLastError();
        // what was measured belongs to another layout, or to the view before the camera started to move (a camera that keeps moving
        // keeps its choice and is looked at again every kAgain frames)
        if (t->frames_since_reset <= 1 || motion_starts) { t->g_epoch_frames = 0; t->g_epoch_id++; t->g_n[0] = t->g_n[1] = 0; t->g_ms[0] = t->g_ms[1] = 0.0f; }
        // the plan settles for four frames on the default variant; then sixteen frames ABBA ABBA ABBA ABBA (both variants see the same
        // mean position in the window: a drift of the frame cost -- the plan still settling, a camera under way -- cancels); then the
        // faster one, looked at again now and then (four frames, ABBA)
        const uint32_t e = t->g_epoch_frames++;
        constexpr uint32_t kSettle = 4, kProbe = 16, kAgain = 128;
        auto abba = [](uint32_t k) -> int { return (int)(((k & 3u) == 1u || (k & 3u) == 2u) ? 1u : 0u); };
        if (e < kSettle) pick = guess;
        else if (e < kSettle + kProbe) pick = guess ^ abba(e - kSettle);
        else if (t->g_n[0] && t->g_n[1]) {
            // (round 6: the default variant stays unless the other one measured CLEARLY faster, 3 %.  Where the choice matters the two are
            //  5-17 % apart -- profiles/r06_line_loops.log -- but a probed frame carries two event records and reads 10 % high, and two
            //  noisy means 0.01 % apart once made a C4 handle draw with the strip variant: 0.787 ms instead of 0.723)
            pick = t->g_ms[guess ^ 1] < 0.97f * t->g_ms[guess] ? (guess ^ 1) : guess;
            if (e % kAgain >= kAgain - 4u) pick ^= abba(e % kAgain - (kAgain - 4u)) ^ 1;      // B A A B seen from the variant in use: two frames of the other one
        }
        // (no samples yet -- a host that queues frames faster than the GPU draws them is past the window before its first probe
        //  completes: the default stays until they arrive; they are taken whenever they complete)
    }
    t->groups_now = pick;
    const bool groups = pick != 0;
    // timed: the frames of the probe window (from the last settle frame on), and now and then two frames of each variant (one event each)
    const uint32_t e_now = t->g_epoch_frames ? t->g_epoch_frames - 1u : 0u;
    const bool probe = ntiles && forced < 0 && ((e_now >= 4u && e_now < 20u) || (e_now >= 20u && e_now % 128u >= 124u));
    vf_terrain::GroupProbe *gp = nullptr;
    if (probe) { gp = &t->gprobe[t->gprobe_head++ % 16]; if (gp->pending) gp = nullptr; }    // (ring full: the frame goes unmeasured)
    if (gp && !gp->b && (hipEventCreate(&gp->a) != hipSuccess || hipEventCreate(&gp->b) != hipSuccess)) gp = nullptr;   // (made on first use: a one-shot handle never probes)
    if (gp && (!gp->a || !gp->b)) gp = nullptr;
    if (gp) (void)hipEventRecord(gp->a, s);
    if (ntiles) {
        uint32_t *vis = write_vis ? t->d_vis : nullptr;
        hipLaunchKernelGGL(k_clear, dim3(ntiles), dim3(256), 0, s, P, S.background, t->d_rgba, vis, stats, nstats, seg_count);
        // one persistent workgroup per CU (a 1024-thread workgroup with 70 KB of LDS fills one), the fast variant, works through
        // the items; then a handful of persistent workgroups of the complete variant take the items that met a clipped or
        // oversized primitive (normally none)
        const dim3 per_cu(std::min<uint32_t>((uint32_t)std::max(1, t->ctx->prop.multiProcessorCount) * (1024u / (uint32_t)kTileThreads), ntiles + kSplitBudget)),
                   few(std::min<uint32_t>(64u, ntiles + kSplitBudget)), threads(kTileThreads);
        const SetupView V = { S.vtx, t->d_hblk, S.recs, S.gen };
#define VF_TILE_ARGS P, V, S.row_ranges, S.cap_seg, S.cap_rad, t->d_lut, t->ctx->d_thresh, S.work_sorted, S.work_count, \
                     rc_lo, rc_hi, t->d_rgba, vis, stats, S.feedback, S.redo
        const bool fast = fast_shading(t);
        // (the complete variant redraws the rare items that met a clipped primitive: always the plain loop)
#define VF_TILE_LAUNCH(WV, FS)                                                                                         \
        do {                                                                                                           \
            if (groups) hipLaunchKernelGGL((k_tile<WV, false, FS, true>), per_cu, threads, 0, s, VF_TILE_ARGS);        \
            else hipLaunchKernelGGL((k_tile<WV, false, FS, false>), per_cu, threads, 0, s, VF_TILE_ARGS);              \
            hipLaunchKernelGGL((k_tile<WV, true, FS, false>), few, threads, 0, s, VF_TILE_ARGS);                       \
            if (gp) { (void)hipEventRecord(gp->b, s); gp->variant = groups ? 1 : 0; gp->seq = e_now; gp->epoch = t->g_epoch_id; gp->pending = true; gp->valid = true; gp = nullptr; } \
        } while (0)
        if (write_vis && fast) VF_TILE_LAUNCH(true, true);
        else if (write_vis) VF_TILE_LAUNCH(true, false);
        else if (fast) VF_TILE_LAUNCH(false, true);
        else VF_TILE_LAUNCH(false, false);
#undef VF_TILE_LAUNCH
#undef VF_TILE_ARGS
    }
    else VF_HIP_TRY(hipMemsetAsync(seg_count, 0, sizeof(uint32_t), s));   // (a shard without tiles: what k_clear does on its way in)
    if (timing_now) { VF_HIP_TRY(hipEventRecord(ev[3], s)); t->timed_frames++; }
    VF_HIP_TRY(hipEventRecord(S.drawn, s));
    VF_HIP_TRY(hipGetLastError());
    std::memcpy(S.u_used, t->u, sizeof S.u_used);           // the camera this set's tile times (being measured now) belong to
    S.have_u_used = true;
    t->last_stream = s;
    t->last_set = set;
    t->last_out = t->d_rgba;
    t->rendered = true;
    return VF_OK;
}

// The first half of the NEXT frame, queued ahead of its call (vf_terrain::pre).
static void plan_ahead(vf_terrain *t, hipStream_t s)
{
    vf_terrain::PrePlan &R = t->pre;
    R.cur_set = t->cur_set; R.frame_no = t->frame_no; R.frames_since_reset = t->frames_since_reset;
    R.camera_moving = t->camera_moving; R.was_moving = t->was_moving; R.have_drawn = t->have_drawn;
    std::memcpy(R.u_drawn, t->u_drawn, sizeof R.u_drawn);
    if (plan_frame(t, s, R.K, true) == VF_OK) { R.valid = true; R.gen = t->inputs_gen; }
    else {                                                  // (a failed launch: the next call plans for itself and reports it)
        t->cur_set = R.cur_set; t->frame_no = R.frame_no; t->frames_since_reset = R.frames_since_reset;
        (void)hipGetLastError();
    }
}

// A read-back has queued its copy on `s`: the plan a waiting caller's next frame will want goes out now, behind the copy (the caller
// waits for the copy's event, not for the stream), if the frame was drawn from inputs that still stand.
static void flush_preplan(vf_terrain *t, hipStream_t s)
{
    if (!t->preplan_pending) return;
    t->preplan_pending = false;
    if (t->pre.valid || t->side || t->last_drawn_gen != t->inputs_gen || t->bounds_dirty || s != t->last_stream) return;
    plan_ahead(t, s);
}

// ... and the wait that goes with it: for the copy's event when a plan was queued behind it, for the stream otherwise
static hipError_t finish_readback(vf_terrain *t, hipStream_t s)
{
    if (!t->preplan_pending) return hipStreamSynchronize(s);
    hipError_t e = t->copied ? hipSuccess : hipEventCreateWithFlags(&t->copied, hipEventDisableTiming);
    if (e != hipSuccess) { t->preplan_pending = false; return hipStreamSynchronize(s); }
    e = hipEventRecord(t->copied, s);
    if (e != hipSuccess) return e;
    flush_preplan(t, s);
    return hipEventSynchronize(t->copied);
}

static int render_impl(vf_terrain *t, hipStream_t s, bool write_vis)
{
    FramePlan K;
    bool planned = false;
    // Was the GPU done with the previous frame when this call came in?  Then the caller waits between frames (render_png, render_rgba,
    // anything that reads a frame back) and the next frame's plan is worth queuing ahead; a caller that streams frames brings the next
    // plan itself, at the same moment, and a plan queued ahead would only be one more after the last frame.
    const bool idle_at_entry = t->rendered && hipEventQuery(t->ps[t->last_set].drawn) == hipSuccess;
    (void)hipGetLastError();                                // ("not ready" is not an error)
    const bool streaming = t->rendered && !idle_at_entry;
    t->preplan_pending = false;                             // (no read-back came in between: this call plans for itself)
    if (t->pre.valid) {
        t->pre.valid = false;
        if (t->pre.gen == t->inputs_gen && !write_vis) {
            K = t->pre.K; planned = true;
            if (!t->side) {                                 // a waiting caller's plan, queued behind its last read-back: done by now?
                const bool ready = hipEventQuery(t->ps[K.set].planned) == hipSuccess;
                (void)hipGetLastError();
                t->tight_calls = ready ? 0u : t->tight_calls + 1u;
                if (t->tight_calls >= 3u) t->want_plan_streams = true;
            }
        }
        else { t->pre.valid = true; VF_HIP_TRY(drop_preplan(t, s)); }     // made for other inputs
    }
    if (!planned) { const int rc = plan_frame(t, s, K, false, streaming); if (rc != VF_OK) return rc; }
    const bool again = t->last_drawn_gen == t->inputs_gen;      // the frame before this one was drawn from the same inputs
    const int rc = draw_frame(t, s, K, write_vis);
    if (rc != VF_OK) return rc;
    t->last_drawn_gen = t->inputs_gen;
    // the camera is at rest (two frames from one set of inputs), the handle is past its first frames, nothing diagnostic is going on:
    // the next frame's plan goes out now
    // (round 6: also for a caller that streams frames of a resting camera -- its next plan goes out one call early, which changes nothing while
    //  the frames keep coming and has the plan ready for the first frame after it has waited once)
    if (again && (idle_at_entry || t->side) && !write_vis && !t->bounds_dirty && !t->camera_moving && !t->was_moving && t->frames_since_reset > vf_terrain::kPlanStates && !(t->timing && t->stats_on)) {
        if (t->side) plan_ahead(t, s);                      // on the plan streams, under this frame's tile kernel
        else t->preplan_pending = true;                     // no plan streams (a waiting caller): behind the next read-back's copy, flush_preplan
    }
    return VF_OK;
}

int vf_terrain_render(vf_terrain *t, void *stream)
{
    if (!t) return fail(VF_ERR_INVALID, "NULL argument");
    if (!t->have_uniforms) return fail(VF_ERR_INVALID, "uniforms not set");
    VF_HIP_TRY(hipSetDevice(t->ctx->device));
    std::memcpy(t->u_frame, t->u, sizeof t->u_frame);
    t->shade_mode_frame = t->shade_mode; t->precision_frame = t->precision;
    t->have_frame = true;
    return render_impl(t, stream ? (hipStream_t)stream : t->ctx->stream, false);
}

int vf_terrain_render_batch(vf_terrain *t, const float *uniforms, uint32_t n, void *const *dev_rgba, void *stream)
{
    if (!t || (!uniforms && n)) return fail(VF_ERR_INVALID, "NULL argument");
    VF_HIP_TRY(hipSetDevice(t->ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : t->ctx->stream;
    for (uint32_t k = 0; k < n; ++k) {
        if (dev_rgba && !dev_rgba[k]) return fail(VF_ERR_INVALID, "dev_rgba holds a NULL output buffer");
        if (!t->have_uniforms || std::memcmp(t->u, uniforms + 44u * k, sizeof t->u) != 0) t->inputs_gen++;
        std::memcpy(t->u, uniforms + 44u * k, sizeof t->u);
        t->have_uniforms = true;
        if (dev_rgba) t->d_rgba = (uint32_t *)dev_rgba[k];
        std::memcpy(t->u_frame, t->u, sizeof t->u_frame);
        t->shade_mode_frame = t->shade_mode; t->precision_frame = t->precision;
        t->have_frame = true;
        const int rc = render_impl(t, s, false);
        if (rc != VF_OK) return rc;
    }
    return VF_OK;
}

int vf_terrain_render_batch_host(vf_terrain *t, const float *uniforms, uint32_t n, uint8_t *const *host_rgba)
{
    if (!t || !uniforms || !host_rgba) return fail(VF_ERR_INVALID, "NULL argument");
    if (t->shard_tiles || t->local_rows != t->H) return fail(VF_ERR_INVALID, "batch read-back needs the whole frame on one handle (pose-parallel ranks are replicas)");
    for (uint32_t k = 0; k < n; ++k) if (!host_rgba[k]) return fail(VF_ERR_INVALID, "host_rgba holds a NULL destination");
    VF_HIP_TRY(hipSetDevice(t->ctx->device));
    constexpr uint32_t R = vf_terrain::kBatchRing;
    const size_t frame_bytes = (size_t)t->W * t->H * 4, slot_bytes = (size_t)t->ntx * t->nty * kTileW * kTileH * 4;
    VF_HIP_TRY(ctx_copy_stream(t->ctx, &t->copy_stream));
    for (uint32_t r = 0; r < R && r < n; ++r) {
        if (!t->d_batch[r]) VF_HIP_TRY(hipMalloc(&t->d_batch[r], slot_bytes));
        if (!t->batch_drawn[r]) VF_HIP_TRY(hipEventCreateWithFlags(&t->batch_drawn[r], hipEventDisableTiming));
        if (!t->batch_copied[r]) VF_HIP_TRY(hipEventCreateWithFlags(&t->batch_copied[r], hipEventDisableTiming));
    }
    hipStream_t s = t->ctx->stream;
    uint32_t *const out_before = t->d_rgba;
    int rc = VF_OK;
    hipError_t err = hipSuccess;
    for (uint32_t k = 0; k < n && rc == VF_OK && err == hipSuccess; ++k) {
        const uint32_t r = k % R;
        if (k >= R) err = hipStreamWaitEvent(s, t->batch_copied[r], 0);       // pose k - R has left the slot
        if (err != hipSuccess) break;
        rc = vf_terrain_render_batch(t, uniforms + 44u * k, 1, (void *const *)&t->d_batch[r], s);
        if (rc != VF_OK) break;
        err = hipEventRecord(t->batch_drawn[r], s);
        if (err == hipSuccess) err = hipStreamWaitEvent(t->copy_stream, t->batch_drawn[r], 0);
        // (a pinned destination -- vf_host_alloc -- makes this one asynchronous DMA under the next pose's kernels; a pageable one is
        //  staged by the runtime and holds the host until it is done: correct, slower)
        if (err == hipSuccess) err = hipMemcpyAsync(host_rgba[k], t->d_batch[r], frame_bytes, hipMemcpyDeviceToHost, t->copy_stream);
        if (err == hipSuccess) err = hipEventRecord(t->batch_copied[r], t->copy_stream);
    }
    const hipError_t e2 = hipStreamSynchronize(t->copy_stream);
    const hipError_t e3 = hipStreamSynchronize(s);
    t->d_rgba = out_before;
    t->rendered = false;                                    // the handle's own output buffer does not hold the last pose: read-backs must render first
    if (rc != VF_OK) return rc;
    if (err != hipSuccess) return fail(VF_ERR_HIP, std::string("batch read-back: ") + hipGetErrorString(err));
    if (e2 != hipSuccess || e3 != hipSuccess) return fail(VF_ERR_HIP, std::string("batch read-back: ") + hipGetErrorString(e2 != hipSuccess ? e2 : e3));
    return VF_OK;
}

// Re-render the frame vf_terrain_render drew last with the visibility store enabled, into scratch buffers: uniforms set since,
// the output buffer (also a caller's, vf_terrain_set_output_device), the stream later calls synchronise with, the timing ring
// and the camera-motion state are as before afterwards.  What the extra frame does leave behind: it uses one of the two plan
// states and adds its tile times to that state's scheduling feedback -- the same view as the frame before it, so the feedback
// stays valid -- and the next frame plans with the other state.
static int render_visibility(vf_terrain *t)
{
    const size_t npx = (size_t)t->ntx * t->nty * kTileW * kTileH;
    if (!t->d_vis) VF_HIP_TRY(hipMalloc(&t->d_vis, npx * sizeof(uint32_t)));
    if (!t->d_rgba_scratch) VF_HIP_TRY(hipMalloc(&t->d_rgba_scratch, npx * sizeof(uint32_t)));
    int rc = vf_terrain_sync(t);
    if (rc != VF_OK) return rc;
    VF_HIP_TRY(sync_sides(t));
    float u_now[44], u_drawn[32];
    std::memcpy(u_now, t->u, sizeof u_now); std::memcpy(u_drawn, t->u_drawn, sizeof u_drawn);
    uint32_t *const out_now = t->d_rgba;
    const hipStream_t stream_now = t->last_stream;
    const bool timing = t->timing, have_drawn = t->have_drawn, moving = t->camera_moving, was_moving = t->was_moving;
    const uint32_t mode_now = t->shade_mode, prec_now = t->precision, since = t->frames_since_reset;
    const bool rendered = t->rendered;
    if (t->have_frame) { std::memcpy(t->u, t->u_frame, sizeof t->u); t->shade_mode = t->shade_mode_frame; t->precision = t->precision_frame; }
    t->d_rgba = t->d_rgba_scratch; t->timing = false;
    rc = render_impl(t, t->ctx->stream, true);
    hipError_t e = hipStreamSynchronize(t->ctx->stream);
    std::memcpy(t->u, u_now, sizeof u_now); std::memcpy(t->u_drawn, u_drawn, sizeof u_drawn);
    t->d_rgba = out_now; t->last_stream = stream_now; t->timing = timing; t->have_drawn = have_drawn;
    t->camera_moving = moving; t->was_moving = was_moving; t->shade_mode = mode_now; t->precision = prec_now; t->frames_since_reset = since;
    t->rendered = rendered;                                 // the diagnostic frame went to scratch buffers: the caller's output is as it was
    if (rc != VF_OK) return rc;
    if (e != hipSuccess) return fail(VF_ERR_HIP, std::string("visibility render: ") + hipGetErrorString(e));
    return VF_OK;
}

int vf_terrain_sync(vf_terrain *t)
{
    if (!t) return fail(VF_ERR_INVALID, "NULL argument");
    VF_HIP_TRY(hipSetDevice(t->ctx->device));
    VF_HIP_TRY(hipStreamSynchronize(t->last_stream ? t->last_stream : t->ctx->stream));
    return VF_OK;
}

constexpr size_t kStageChunk = 8u << 20;
constexpr size_t kStageSlots = 4;
static unsigned copy_threads()
{
    if (const char *e = std::getenv("VF_COPY_THREADS")) { const int v = std::atoi(e); if (v >= 1) return (unsigned)std::min(v, 16); }
    const unsigned hw = std::thread::hardware_concurrency();
    return std::max(1u, std::min(4u, hw > 1 ? hw - 1 : 1u));
}
// Device -> pageable host memory through a ring of pinned chunks: the DMA engine fills chunk k + 1 .. k + 3 while host threads
// move chunk k out (a frame-sized destination is usually fresh memory: the copy out is page-fault bound, which is why it is
// spread over a few threads).  The calling thread only orchestrates: it enqueues a chunk once every thread is done with the
// chunk that used the slot before.  The ring is owned by the handle: nothing is allocated or registered per call (the
// reference maps a fresh buffer per call, src/terrain/mod.rs:446-451).
static bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // (ordinary pageable memory: "invalid value")
    return a.type == hipMemoryTypeHost;
}
// Device -> page-locked host memory, queued on s.  Where the device can address the destination the bytes travel by a kernel's
// stores (k_copy_to_host) -- the same PCIe time as the copy engine's transfer (C4 frame: 1.35 against 1.30 ms), without the 8-10 ms
// the engine's FIRST transfer of a process costs, which a one-shot caller (construct, render once) would pay in its only read-back.
// The batch read-back keeps the copy engine: its copies run beside the next poses' kernels.
static hipError_t copy_to_pinned(void *host, const void *dev, size_t n, hipStream_t s)
{
    void *view = nullptr;
    if (n >= 4096 && hipHostGetDevicePointer(&view, host, 0) == hipSuccess && view && (((uintptr_t)view | (uintptr_t)dev) & 15u) == 0) {
        const size_t n16 = n / 16;
        const unsigned grid = (unsigned)std::min<size_t>(1024, (n16 + 255) / 256);
        hipLaunchKernelGGL(k_copy_to_host, dim3(grid), dim3(256), 0, s, (const uint4 *)dev, (uint4 *)view, n16, n);
        return hipGetLastError();
    }
    (void)hipGetLastError();
    return hipMemcpyAsync(host, dev, n, hipMemcpyDeviceToHost, s);
}
static int copy_to_host_staged(vf_terrain *t, uint8_t *dst, const uint8_t *src, size_t n, hipStream_t s)
{
    // a destination in pinned host memory (vf_host_alloc, or the caller's own hipHostMalloc / hipHostRegister): one transfer, nothing to stage
    if (is_pinned_host(dst)) { VF_HIP_TRY(copy_to_pinned(dst, src, n, s)); VF_HIP_TRY(finish_readback(t, s)); return VF_OK; }
    if (n < kStageChunk) { VF_HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, s)); VF_HIP_TRY(finish_readback(t, s)); return VF_OK; }
    if (!t->h_stage) VF_HIP_TRY(pinned_alloc((void **)&t->h_stage, kStageSlots * kStageChunk));
    for (auto &e : t->stage_ev) if (!e) VF_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const size_t nchunks = (n + kStageChunk - 1) / kStageChunk;
    const unsigned nthreads = copy_threads();
    auto chunk_len = [&](size_t k) { return k + 1 == nchunks ? n - k * kStageChunk : kStageChunk; };
    std::atomic<size_t> enqueued{0};                       // chunks whose copy + event are on the stream
    std::unique_ptr<std::atomic<uint32_t>[]> done(new std::atomic<uint32_t>[nchunks]);
    for (size_t k = 0; k < nchunks; ++k) done[k].store(0, std::memory_order_relaxed);
    std::atomic<int> failed{0};
    const int device = t->ctx->device;
    auto worker = [&](unsigned w) {
        (void)hipSetDevice(device);
        for (size_t k = 0; k < nchunks; ++k) {
            while (enqueued.load(std::memory_order_acquire) <= k) {
                if (failed.load(std::memory_order_relaxed)) return;
                std::this_thread::yield();
            }
            if (hipEventSynchronize(t->stage_ev[k % kStageSlots]) != hipSuccess) failed.store(1);
            else {
                const size_t len = chunk_len(k);
                const size_t lo = (len * w / nthreads) & ~(size_t)4095, hi = w + 1 == nthreads ? len : (len * (w + 1) / nthreads) & ~(size_t)4095;
                if (hi > lo) std::memcpy(dst + k * kStageChunk + lo, t->h_stage + (k % kStageSlots) * kStageChunk + lo, hi - lo);
            }
            done[k].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(nthreads);
    for (unsigned w = 0; w < nthreads; ++w) pool.emplace_back(worker, w);
    hipError_t err = hipSuccess;
    for (size_t i = 0; i < nchunks && err == hipSuccess && !failed.load(); ++i) {
        if (i >= kStageSlots)
            while (done[i - kStageSlots].load(std::memory_order_acquire) < nthreads) std::this_thread::yield();
        err = copy_to_pinned(t->h_stage + (i % kStageSlots) * kStageChunk, src + i * kStageChunk, chunk_len(i), s);
        if (err == hipSuccess) err = hipEventRecord(t->stage_ev[i % kStageSlots], s);
        if (err == hipSuccess) enqueued.store(i + 1, std::memory_order_release);
    }
    if (err == hipSuccess) flush_preplan(t, s);            // (behind the last chunk's copy: the threads above wait for the chunks' events)
    if (err != hipSuccess) failed.store(1);
    for (auto &th : pool) th.join();
    if (err != hipSuccess) return fail(VF_ERR_HIP, hipGetErrorString(err));
    if (failed.load()) return fail(VF_ERR_HIP, "device -> host copy failed");
    return VF_OK;
}

int vf_terrain_read_rgba(vf_terrain *t, uint8_t *dst, uint32_t y0, uint32_t rows)
{
    if (!t || !dst) return fail(VF_ERR_INVALID, "NULL argument");
    if (!t->rendered) return fail(VF_ERR_INVALID, "nothing rendered yet");
    if (t->shard_tiles) return fail(VF_ERR_INVALID, "tile-sharded handle: read with vf_terrain_read_tiles");
    if ((uint64_t)y0 + rows > t->local_rows) return fail(VF_ERR_INVALID, "row range outside the local rows");
    int rc = vf_terrain_sync(t);
    if (rc != VF_OK) return rc;
    return copy_to_host_staged(t, dst, (const uint8_t *)(t->d_rgba + (size_t)y0 * t->W), (size_t)rows * t->W * 4,
                               t->last_stream ? t->last_stream : t->ctx->stream);
}

int vf_host_alloc(size_t bytes, void **host)
{
    if (!host || bytes == 0) return fail(VF_ERR_INVALID, "NULL argument or zero size");
    *host = nullptr;
    hipError_t e = pinned_alloc(host, bytes);
    if (e != hipSuccess) { *host = nullptr; return fail(e == hipErrorOutOfMemory ? VF_ERR_NOMEM : VF_ERR_HIP, std::string("page-locked allocation: ") + hipGetErrorString(e)); }
    return VF_OK;
}
void vf_host_free(void *host) { pinned_free(host); }

int vf_terrain_read_png_scanlines(vf_terrain *t, const uint8_t **host_scanlines, size_t *nbytes)
{
    if (!t || !host_scanlines || !nbytes) return fail(VF_ERR_INVALID, "NULL argument");
    if (!t->rendered) return fail(VF_ERR_INVALID, "nothing rendered yet");
    if (t->shard_tiles || t->local_rows != t->H) return fail(VF_ERR_INVALID, "PNG read-back needs the whole frame on one handle");
    VF_HIP_TRY(hipSetDevice(t->ctx->device));
    const size_t n = ((size_t)t->W * 4 + 1) * t->H;
    if (!t->d_png) VF_HIP_TRY(hipMalloc(&t->d_png, n));
    hipStream_t s = t->last_stream ? t->last_stream : t->ctx->stream;
    hipLaunchKernelGGL(k_png_filter, dim3(t->H), dim3(256), 0, s, (const uint32_t *)t->d_rgba, t->W, t->d_png);
    VF_HIP_TRY(hipGetLastError());
    // the scanlines land in a page-locked buffer of the handle (pinned_alloc: 2.6 ms to make for a C4 frame, then one DMA per frame):
    // persistent, where the reference allocates per call (:446-451)
    if (!t->h_png) VF_HIP_TRY(pinned_alloc((void **)&t->h_png, n));
    uint8_t *host = t->h_png;
    VF_HIP_TRY(copy_to_pinned(host, t->d_png, n, s));
    VF_HIP_TRY(finish_readback(t, s));                      // (a waiting caller's next plan runs while the host deflates)
    *host_scanlines = host;
    *nbytes = n;
    return VF_OK;
}

int vf_terrain_read_visibility(vf_terrain *t, uint32_t *dst)
{
    if (!t || !dst) return fail(VF_ERR_INVALID, "NULL argument");
    if (!t->have_uniforms) return fail(VF_ERR_INVALID, "uniforms not set");
    if (t->shard_tiles) return fail(VF_ERR_INVALID, "visibility read-back needs a row-oriented handle (vf_terrain_set_shard)");
    VF_HIP_TRY(hipSetDevice(t->ctx->device));
    // the visibility tile normally lives and dies in LDS: the last frame is rendered again with the debug store enabled
    int rc = render_visibility(t);
    if (rc != VF_OK) return rc;
    VF_HIP_TRY(hipMemcpy(dst, t->d_vis, (size_t)t->local_rows * t->W * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return VF_OK;
}

int vf_terrain_debug_fragment_stage(vf_terrain *t, uint32_t repeats, vf_fragment_timing *out)
{
    if (!t || !out) return fail(VF_ERR_INVALID, "NULL argument");
    if (!t->have_uniforms) return fail(VF_ERR_INVALID, "uniforms not set");
    if (t->shard_tiles || t->nranks != 1) return fail(VF_ERR_INVALID, "fragment-stage diagnostics need a whole-frame handle");
    if (repeats == 0) repeats = 1;
    VF_HIP_TRY(hipSetDevice(t->ctx->device));
    int rc = render_visibility(t);                          // d_vis + the frame as the tile kernel shades it (d_rgba_scratch)
    if (rc != VF_OK) return rc;
    const size_t npx = (size_t)t->W * t->H;
    uint32_t *d_out = nullptr;
    VF_HIP_TRY(hipMalloc(&d_out, npx * sizeof(uint32_t)));
    if (!t->d_diag) { hipError_t e = hipMalloc(&t->d_diag, 4 * sizeof(uint32_t)); if (e != hipSuccess) { (void)hipFree(d_out); return fail(VF_ERR_NOMEM, "diagnostics allocation failed"); } }
    uint32_t redo = 0;                                      // did the frame hold clipped / oversized primitives?  (normally not)
    hipError_t err = hipMemcpy(&redo, t->ps[t->last_set].work_count + 3, sizeof redo, hipMemcpyDeviceToHost);
    FrameParams P;
    float u_now[44];
    std::memcpy(u_now, t->u, sizeof u_now);
    const uint32_t mode_now = t->shade_mode, prec_now = t->precision;
    if (t->have_frame) { std::memcpy(t->u, t->u_frame, sizeof t->u); t->shade_mode = t->shade_mode_frame; t->precision = t->precision_frame; }
    build_params(t, P);
    const bool fast = fast_shading(t);
    std::memcpy(t->u, u_now, sizeof u_now); t->shade_mode = mode_now; t->precision = prec_now;
    hipStream_t s = t->ctx->stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (err == hipSuccess) err = hipEventCreate(&e0);
    if (err == hipSuccess) err = hipEventCreate(&e1);
    // 16-byte HBM accesses when the rows allow them and the frame holds no clipped primitive (VF_RESOLVE_PER_PIXEL=1: the per-pixel form)
    const bool quads = !redo && t->W % 4u == 0 && !std::getenv("VF_RESOLVE_PER_PIXEL");
    // Persistent workgroups, a multiple of 8 of them (one share per XCD).  FEW per CU on purpose: the record gathers of more waves
    // than these evict each other's lines from the 32 KB L1 (C4 fill camera, k_resolve4: 3 per CU 0.130 ms, 4: 0.144, 8: 0.148)
    const uint32_t per_cu = std::getenv("VF_RESOLVE_PER_CU") ? (uint32_t)std::max(1, std::atoi(std::getenv("VF_RESOLVE_PER_CU"))) : (quads ? 3u : 4u);
    const uint32_t cus = (uint32_t)std::max(8, t->ctx->prop.multiProcessorCount) / 8u * 8u;
    const dim3 grid(std::min<uint32_t>((((t->W + 31u) / 32u) * ((t->H + 7u) / 8u) + 7u) / 8u * 8u, cus * per_cu)), threads(256);   // 32 x 8 pixel regions
    const vf_terrain::PlanState &S = t->ps[t->last_set];     // the set-up of the frame just rendered
    const SetupView V = { S.vtx, t->d_hblk, S.recs, S.gen };
    // four pixels per lane when the rows allow 16-byte accesses and the frame holds no clipped primitive (VF_RESOLVE_PER_PIXEL=1: the per-pixel form)
    constexpr uint32_t RQ = 8u, RY = 32u;                  // k_resolve4's region: 8 quads x 32 rows
    const dim3 grid4(std::min<uint32_t>((((t->W / 4u + RQ - 1u) / RQ) * ((t->H + RY - 1u) / RY) + 7u) / 8u * 8u, cus * per_cu));
    auto launch = [&](uint32_t *covered) {
        if (quads && fast) hipLaunchKernelGGL((k_resolve4<true>), grid4, threads, 0, s, P, V, t->d_lut, t->ctx->d_thresh, (const uint4 *)t->d_vis, (uint4 *)d_out, covered);
        else if (quads) hipLaunchKernelGGL((k_resolve4<false>), grid4, threads, 0, s, P, V, t->d_lut, t->ctx->d_thresh, (const uint4 *)t->d_vis, (uint4 *)d_out, covered);
        else if (redo && fast) hipLaunchKernelGGL((k_resolve<true, true>), grid, threads, 0, s, P, V, t->d_lut, t->ctx->d_thresh, t->d_vis, d_out, covered);
        else if (redo) hipLaunchKernelGGL((k_resolve<true, false>), grid, threads, 0, s, P, V, t->d_lut, t->ctx->d_thresh, t->d_vis, d_out, covered);
        else if (fast) hipLaunchKernelGGL((k_resolve<false, true>), grid, threads, 0, s, P, V, t->d_lut, t->ctx->d_thresh, t->d_vis, d_out, covered);
        else hipLaunchKernelGGL((k_resolve<false, false>), grid, threads, 0, s, P, V, t->d_lut, t->ctx->d_thresh, t->d_vis, d_out, covered);
    };
    if (err == hipSuccess) err = hipMemsetAsync(t->d_diag, 0, 4 * sizeof(uint32_t), s);
    if (err == hipSuccess) { launch(t->d_diag); err = hipGetLastError(); }       // warm-up launch, counts the covered pixels
    if (err == hipSuccess) err = hipEventRecord(e0, s);
    for (uint32_t k = 0; k < repeats && err == hipSuccess; ++k) { launch(nullptr); err = hipGetLastError(); }
    if (err == hipSuccess) err = hipEventRecord(e1, s);
    if (err == hipSuccess) err = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    uint32_t covered = 0;
    if (err == hipSuccess) err = hipMemcpy(&covered, t->d_diag, sizeof covered, hipMemcpyDeviceToHost);
    // the resolved frame against the one the tile kernel produced (row-major whole frame in both buffers)
    uint32_t equal = 0;
    if (err == hipSuccess) {
        std::vector<uint32_t> a(npx), b(npx);
        err = hipMemcpy(a.data(), d_out, npx * 4, hipMemcpyDeviceToHost);
        if (err == hipSuccess) err = hipMemcpy(b.data(), t->d_rgba_scratch, npx * 4, hipMemcpyDeviceToHost);
        if (err == hipSuccess) equal = std::memcmp(a.data(), b.data(), npx * 4) == 0 ? 1u : 0u;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_out);
    if (err != hipSuccess) return fail(VF_ERR_HIP, std::string("fragment-stage diagnostics: ") + hipGetErrorString(err));
    out->resolve_ms = ms / (float)repeats; out->covered_pixels = covered; out->repeats = repeats; out->equal_to_frame = equal;
    return VF_OK;
}

int vf_terrain_debug_item_stats(vf_terrain *t, uint32_t *dst, uint32_t max_items, uint32_t *count)
{
    if (!t || !dst || !count) return fail(VF_ERR_INVALID, "NULL argument");
    if (!t->timing || !t->stats_on || !t->rendered) return fail(VF_ERR_INVALID, "statistics not enabled or nothing rendered");
    int rc = vf_terrain_sync(t);
    if (rc != VF_OK) return rc;
    uint32_t n = 0;
    VF_HIP_TRY(hipMemcpy(&n, t->ps[t->last_set].work_count, sizeof n, hipMemcpyDeviceToHost));
    if (n > max_items) n = max_items;
    if (n) VF_HIP_TRY(hipMemcpy(dst, t->d_stats + 4, 4 * (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    *count = n;
    return VF_OK;
}

int vf_terrain_debug_phase_cycles(vf_terrain *t, uint64_t *dst, uint32_t n)
{
    if (!t || !dst) return fail(VF_ERR_INVALID, "NULL argument");
#ifndef VF_PHASE_PROF
    (void)n;
    return fail(VF_ERR_INVALID, "library built without -DVF_PHASE_PROF");
#else
    if (!t->timing || !t->stats_on || !t->rendered) return fail(VF_ERR_INVALID, "statistics not enabled or nothing rendered");
    int rc = vf_terrain_sync(t);
    if (rc != VF_OK) return rc;
    if (n > kPhaseSlots) n = kPhaseSlots;
    VF_HIP_TRY(hipMemcpy(dst, t->d_stats + 4 + 4 * ((size_t)t->ntx * t->nty + kSplitBudget), n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return VF_OK;
#endif
}

int vf_terrain_enable_timing(vf_terrain *t, int enable)
{
    if (!t) return fail(VF_ERR_INVALID, "NULL argument");
    if (enable != 0 && !t->ev[0][0]) {                      // the event ring is made when timing is first asked for
        VF_HIP_TRY(hipSetDevice(t->ctx->device));
        for (auto &f : t->ev) for (auto &e : f) VF_HIP_TRY(hipEventCreate(&e));
    }
    if (enable != 0) for (auto &S : t->ps) for (auto &e : S.pev) if (!e) VF_HIP_TRY(hipEventCreate(&e));
    if (enable < 0 || enable > 3) return fail(VF_ERR_INVALID, "enable must be 0 .. 3");
    t->timing = enable != 0;
    t->stats_on = enable == 1;   // 2, 3: HIP events only -- the tile kernel runs exactly as it does untimed (no per-item statistics)
    t->timing_every = enable == 3 ? 4u : 1u;
    t->timed_frames = 0;     // (re)start the averaging window
    return VF_OK;
}

int vf_terrain_timings(vf_terrain *t, vf_timings *out)
{
    if (!t || !out) return fail(VF_ERR_INVALID, "NULL argument");
    if (!t->timing || !t->rendered || t->timed_frames == 0) return fail(VF_ERR_INVALID, "timing not enabled or nothing rendered");
    int rc = vf_terrain_sync(t);
    if (rc != VF_OK) return rc;
    // average over the frames recorded since vf_terrain_enable_timing (at most the last kTimingRing)
    const uint32_t nf = t->timed_frames < (uint32_t)vf_terrain::kTimingRing ? t->timed_frames : (uint32_t)vf_terrain::kTimingRing;
    double ranges = 0, plan = 0, tile = 0, total = 0;
    uint32_t nplan = 0;
    for (uint32_t f = 0; f < nf; ++f) {
        float c = 0;
        VF_HIP_TRY(hipEventSynchronize(t->ev[f][3]));
        VF_HIP_TRY(hipEventElapsedTime(&c, t->ev[f][4], t->ev[f][3]));      // clear + tile kernels, on the caller's stream
        tile += c; total += c;
    }
    // the plan chain: the last plan of each plan state (the frames drawn last, or the plan queued ahead of the next call)
    VF_HIP_TRY(sync_sides(t));
    for (auto &S : t->ps) {
        if (!S.pev_valid || !S.pev[2]) continue;
        float a = 0, b = 0;
        if (hipEventSynchronize(S.pev[2]) != hipSuccess || hipEventElapsedTime(&a, S.pev[0], S.pev[1]) != hipSuccess ||
            hipEventElapsedTime(&b, S.pev[1], S.pev[2]) != hipSuccess) { (void)hipGetLastError(); continue; }
        ranges += a; plan += b; ++nplan;
    }
    if (nf == 1u && t->ps[t->last_set].pev_valid) {         // a single timed frame: plan start -> RGBA8 complete
        float d = 0;
        if (hipEventElapsedTime(&d, t->ps[t->last_set].pev[0], t->ev[0][3]) == hipSuccess && d > 0.0f) total = d;
        else (void)hipGetLastError();
    }
    out->ranges_ms = nplan ? (float)(ranges / nplan) : 0.0f; out->plan_ms = nplan ? (float)(plan / nplan) : 0.0f; out->tile_ms = (float)(tile / nf);
    out->total_ms = (float)(total / nf);
    if (nf >= 2 && t->timed_frames <= (uint32_t)vf_terrain::kTimingRing) {
        // frames rendered back to back overlap (frame f+1 plans while frame f draws): the frame period is what a frame costs
        float span = 0;
        VF_HIP_TRY(hipEventElapsedTime(&span, t->ev[0][3], t->ev[nf - 1][3]));
        out->total_ms = span / (float)((nf - 1) * t->timing_every);
    }
    out->frames = nf;
    out->blocks_rasterised = 0; out->blocks_distinct = 0;
    out->tiles = t->local_tiles;
    if (t->stats_on) {
        uint32_t c[4];
        VF_HIP_TRY(hipMemcpy(c, t->d_stats, sizeof c, hipMemcpyDeviceToHost));
        out->blocks_rasterised = c[0];   // distinct blocks behind those pairs (last frame): the bitmap behind the per-item statistics
        std::vector<uint32_t> bits((t->nblocks + 31) / 32);
        VF_HIP_TRY(hipMemcpy(bits.data(), t->d_stats + 4 + 4 * ((size_t)t->ntx * t->nty + kSplitBudget) + 2 * kPhaseSlots, bits.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint32_t n = 0;
        for (uint32_t w : bits) n += (uint32_t)__builtin_popcount(w);
        out->blocks_distinct = n;
    }
    return VF_OK;
}

int vf_terrain_frame_times(vf_terrain *t, float *tile_ms, float *period_ms, uint32_t max_frames, uint32_t *count)
{
    if (!t || !count) return fail(VF_ERR_INVALID, "NULL argument");
    if (!t->timing || !t->rendered || t->timed_frames == 0) return fail(VF_ERR_INVALID, "timing not enabled or nothing rendered");
    int rc = vf_terrain_sync(t);
    if (rc != VF_OK) return rc;
    const uint32_t ring = (uint32_t)vf_terrain::kTimingRing;
    const uint32_t nf = std::min(std::min(t->timed_frames, ring), max_frames);
    const uint32_t first = t->timed_frames - nf;            // frame numbers first .. timed_frames - 1 live in ring slot (number % ring)
    for (uint32_t k = 0; k < nf; ++k) {
        hipEvent_t *e = t->ev[(first + k) % ring];
        VF_HIP_TRY(hipEventSynchronize(e[3]));
        if (tile_ms) VF_HIP_TRY(hipEventElapsedTime(&tile_ms[k], e[4], e[3]));
        if (period_ms) {
            period_ms[k] = 0.0f;
            if (k) { VF_HIP_TRY(hipEventElapsedTime(&period_ms[k], t->ev[(first + k - 1u) % ring][3], e[3])); period_ms[k] /= (float)t->timing_every; }
        }
    }
    *count = nf;
    return VF_OK;
}

// ------------------------------------------------------------------------------------------------

int vf_grid_generate_device(vf_ctx *ctx, uint32_t nx, uint32_t nz, float dx, float dy, float *dev_xy, float *dev_uv,
                            uint32_t *dev_idx, void *stream)
{
    if (!ctx || !dev_xy || !dev_uv || !dev_idx) return fail(VF_ERR_INVALID, "NULL argument");
    if (nx < 2 || nz < 2) return fail(VF_ERR_INVALID, "nx and nz must be >= 2");
    if ((uint64_t)nx * nz > 0xFFFFFFFFull) return fail(VF_ERR_INVALID, "vertex count exceeds u32 indices");
    VF_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    size_t nv = (size_t)nx * nz, nc = (size_t)(nx - 1) * (nz - 1);
    hipLaunchKernelGGL(k_grid_vertices, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, s, nx, nz, dx, dy, (float2 *)dev_xy, (float2 *)dev_uv);
    hipLaunchKernelGGL(k_grid_indices, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, nx, nz, dev_idx);
    VF_HIP_TRY(hipGetLastError());
    return VF_OK;
}

int vf_grid_generate(vf_ctx *ctx, uint32_t nx, uint32_t nz, float dx, float dy, float *xy, float *uv, uint32_t *idx)
{
    if (!ctx || !xy || !uv || !idx) return fail(VF_ERR_INVALID, "NULL argument");
    if (nx < 2 || nz < 2) return fail(VF_ERR_INVALID, "nx and nz must be >= 2");
    VF_HIP_TRY(hipSetDevice(ctx->device));
    size_t nv = (size_t)nx * nz, ni = 6 * (size_t)(nx - 1) * (nz - 1);
    float *d_xy = nullptr, *d_uv = nullptr;
    uint32_t *d_idx = nullptr;
    hipError_t err = hipMalloc(&d_xy, nv * 8);
    if (err == hipSuccess) err = hipMalloc(&d_uv, nv * 8);
    if (err == hipSuccess) err = hipMalloc(&d_idx, ni * 4);
    int rc = VF_OK;
    if (err != hipSuccess) rc = fail(VF_ERR_NOMEM, std::string("grid_generate allocation failed: ") + hipGetErrorString(err));
    if (rc == VF_OK) rc = vf_grid_generate_device(ctx, nx, nz, dx, dy, d_xy, d_uv, d_idx, ctx->stream);
    if (rc == VF_OK) {
        err = hipStreamSynchronize(ctx->stream);
        if (err == hipSuccess) err = hipMemcpy(xy, d_xy, nv * 8, hipMemcpyDeviceToHost);
        if (err == hipSuccess) err = hipMemcpy(uv, d_uv, nv * 8, hipMemcpyDeviceToHost);
        if (err == hipSuccess) err = hipMemcpy(idx, d_idx, ni * 4, hipMemcpyDeviceToHost);
        if (err != hipSuccess) rc = fail(VF_ERR_HIP, std::string("grid_generate readback failed: ") + hipGetErrorString(err));
    }
    if (d_xy) (void)hipFree(d_xy);
    if (d_uv) (void)hipFree(d_uv);
    if (d_idx) (void)hipFree(d_idx);
    return rc;
}

int vf_triangle_render(vf_ctx *ctx, uint32_t width, uint32_t height, uint8_t *rgba_host)
{
    if (!ctx || !rgba_host) return fail(VF_ERR_INVALID, "NULL argument");
    if (width == 0 || height == 0 || width > 16384 || height > 16384) return fail(VF_ERR_INVALID, "width/height must be in 1..16384");
    VF_HIP_TRY(hipSetDevice(ctx->device));
    size_t npx = (size_t)width * height;
    uint32_t *d = nullptr;
    VF_HIP_TRY(hipMalloc(&d, npx * 4));
    hipLaunchKernelGGL(k_triangle, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, ctx->stream, width, height, ctx->d_thresh, d);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    if (err == hipSuccess) err = hipMemcpy(rgba_host, d, npx * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (err != hipSuccess) return fail(VF_ERR_HIP, std::string("triangle render failed: ") + hipGetErrorString(err));
    return VF_OK;
}

// ---- Renderer DEM path -------------------------------------------------------------------------------------
} // extern "C"   (struct definition below is C++)

struct vf_dem {
    vf_ctx *ctx = nullptr;
    float *d_h = nullptr;        // heights (already multiplied by the exaggeration), w*h
    float *d_tex = nullptr;      // R32F "texture" written by upload_height_r32f
    void *d_stage = nullptr;     // staging for the ingest (raw f32 / f64 samples)
    size_t stage_bytes = 0;
    uint32_t w = 0, h = 0, tex_w = 0, tex_h = 0;
    size_t cap = 0, tex_cap = 0;
    uint32_t *d_mm = nullptr;    // min / max as ordered ints
    double *d_partial = nullptr; // one partial sum per reduction block
};
static constexpr int kDemBlocks = 2048;

extern "C" {

int vf_dem_create(vf_ctx *ctx, vf_dem **out)
{
    if (!ctx || !out) return fail(VF_ERR_INVALID, "NULL argument");
    VF_HIP_TRY(hipSetDevice(ctx->device));
    vf_dem *d = new (std::nothrow) vf_dem;
    if (!d) return fail(VF_ERR_NOMEM, "out of host memory");
    d->ctx = ctx;
    hipError_t e = hipMalloc(&d->d_mm, 2 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&d->d_partial, kDemBlocks * sizeof(double));
    if (e != hipSuccess) { vf_dem_destroy(d); return fail(VF_ERR_NOMEM, std::string("dem allocation failed: ") + hipGetErrorString(e)); }
    *out = d;
    return VF_OK;
}

void vf_dem_destroy(vf_dem *d)
{
    if (!d) return;
    (void)hipSetDevice(d->ctx->device);
    (void)hipDeviceSynchronize();
    void *ptrs[] = { d->d_h, d->d_tex, d->d_stage, d->d_mm, d->d_partial };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete d;
}

} // extern "C"

template <typename T>
static int dem_ingest(vf_dem *d, const T *host, uint32_t w, uint32_t h, float exaggeration)
{
    if (!d || !host) return fail(VF_ERR_INVALID, "NULL argument");
    if (w == 0 || h == 0) return fail(VF_ERR_INVALID, "heightmap cannot be empty");
    VF_HIP_TRY(hipSetDevice(d->ctx->device));
    const size_t n = (size_t)w * h;
    if (n > d->cap) {
        if (d->d_h) VF_HIP_TRY(hipFree(d->d_h));
        d->d_h = nullptr; d->cap = 0;
        VF_HIP_TRY(hipMalloc(&d->d_h, n * sizeof(float)));
        d->cap = n;
    }
    if (n * sizeof(T) > d->stage_bytes) {
        if (d->d_stage) VF_HIP_TRY(hipFree(d->d_stage));
        d->d_stage = nullptr; d->stage_bytes = 0;
        VF_HIP_TRY(hipMalloc(&d->d_stage, n * sizeof(T)));
        d->stage_bytes = n * sizeof(T);
    }
    hipStream_t s = d->ctx->stream;
    VF_HIP_TRY(hipMemcpyAsync(d->d_stage, host, n * sizeof(T), hipMemcpyHostToDevice, s));
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_dem_ingest<T>, dim3(blocks), dim3(256), 0, s, (const T *)d->d_stage, d->d_h, n, exaggeration);
    VF_HIP_TRY(hipGetLastError());
    VF_HIP_TRY(hipStreamSynchronize(s));     // the host buffer is only borrowed for this call
    d->w = w; d->h = h;
    return VF_OK;
}

extern "C" {

int vf_dem_set_heights_f32(vf_dem *d, const float *host, uint32_t w, uint32_t h, float ex) { return dem_ingest<float>(d, host, w, h, ex); }
int vf_dem_set_heights_f64(vf_dem *d, const double *host, uint32_t w, uint32_t h, float ex) { return dem_ingest<double>(d, host, w, h, ex); }

static int dem_stats_impl(vf_dem *d, float out[4])
{
    if (!d->d_h || d->w == 0) return fail(VF_ERR_INVALID, "no terrain uploaded; call add_terrain() first");
    VF_HIP_TRY(hipSetDevice(d->ctx->device));
    const size_t n = (size_t)d->w * d->h;
    hipStream_t s = d->ctx->stream;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, (size_t)kDemBlocks);
    const uint32_t init[2] = { 0xFFFFFFFFu, 0u };
    VF_HIP_TRY(hipMemcpyAsync(d->d_mm, init, sizeof init, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_dem_minmaxsum, dim3(blocks), dim3(256), 0, s, d->d_h, n, d->d_mm, d->d_partial);
    VF_HIP_TRY(hipGetLastError());
    std::vector<double> part(blocks);
    uint32_t mm[2];
    VF_HIP_TRY(hipMemcpyAsync(part.data(), d->d_partial, blocks * sizeof(double), hipMemcpyDeviceToHost, s));
    VF_HIP_TRY(hipMemcpyAsync(mm, d->d_mm, sizeof mm, hipMemcpyDeviceToHost, s));
    VF_HIP_TRY(hipStreamSynchronize(s));
    double sum = 0.0;
    for (double v : part) sum += v;
    const float mean = (float)(sum / (double)n);
    hipLaunchKernelGGL(k_dem_sqdev, dim3(blocks), dim3(256), 0, s, d->d_h, n, mean, d->d_partial);
    VF_HIP_TRY(hipGetLastError());
    VF_HIP_TRY(hipMemcpyAsync(part.data(), d->d_partial, blocks * sizeof(double), hipMemcpyDeviceToHost, s));
    VF_HIP_TRY(hipStreamSynchronize(s));
    double var = 0.0;
    for (double v : part) var += v;
    auto unorder = [](uint32_t u) { uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; float f; std::memcpy(&f, &b, 4); return f; };
    // an all-NaN map leaves the init pattern; the reference would report its first element (NaN)
    out[0] = mm[0] == 0xFFFFFFFFu ? NAN : unorder(mm[0]);
    out[1] = mm[1] == 0u ? NAN : unorder(mm[1]);
    out[2] = mean;
    out[3] = std::sqrt((float)(var / (double)n));
    return VF_OK;
}
int vf_dem_stats(vf_dem *d, float out[4])
{
    if (!d || !out) return fail(VF_ERR_INVALID, "NULL argument");
    return dem_stats_impl(d, out);
}

int vf_dem_percentile_range(vf_dem *d, float *p1, float *p99)
{
    if (!d || !p1 || !p99) return fail(VF_ERR_INVALID, "NULL argument");
    if (!d->d_h || d->w == 0) return fail(VF_ERR_INVALID, "no terrain uploaded; call add_terrain() first");
    VF_HIP_TRY(hipSetDevice(d->ctx->device));
    const size_t n = (size_t)d->w * d->h, SAMPLE = 65536;
    const size_t step = n > SAMPLE ? n / SAMPLE : 1;                 // src/terrain_stats.rs:24-29
    const size_t m = (n + step - 1) / step;                          // iter().step_by(step).count()
    std::vector<float> buf(m);
    hipStream_t s = d->ctx->stream;
    if (step == 1) {
        VF_HIP_TRY(hipMemcpyAsync(buf.data(), d->d_h, n * sizeof(float), hipMemcpyDeviceToHost, s));
    } else {
        if (m * sizeof(float) > d->stage_bytes) {
            if (d->d_stage) VF_HIP_TRY(hipFree(d->d_stage));
            d->d_stage = nullptr; d->stage_bytes = 0;
            VF_HIP_TRY(hipMalloc(&d->d_stage, m * sizeof(float)));
            d->stage_bytes = m * sizeof(float);
        }
        hipLaunchKernelGGL(k_dem_sample, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, d->d_h, n, step, (float *)d->d_stage, m);
        VF_HIP_TRY(hipGetLastError());
        VF_HIP_TRY(hipMemcpyAsync(buf.data(), d->d_stage, m * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    VF_HIP_TRY(hipStreamSynchronize(s));
    // the sample is at most ~131071 values: ordering it is host logic, exactly like the reference's sort_by
    std::stable_sort(buf.begin(), buf.end(), [](float a, float b) { return a < b; });
    *p1 = buf[(size_t)((float)buf.size() * 0.01f)];
    *p99 = buf[(size_t)((float)buf.size() * 0.99f)];
    return VF_OK;
}

int vf_dem_normalize(vf_dem *d, int mode, float lo, float hi, float eps)
{
    if (!d) return fail(VF_ERR_INVALID, "NULL argument");
    if (mode != 0 && mode != 1) return fail(VF_ERR_INVALID, "mode must be 'minmax' or 'zscore'");
    float st[4];
    int rc = dem_stats_impl(d, st);
    if (rc != VF_OK) return rc;
    const size_t n = (size_t)d->w * d->h;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    hipStream_t s = d->ctx->stream;
    if (mode == 0) {
        const float denom = std::fmax(std::fabs(st[1] - st[0]), eps);    // src/lib.rs:938-939
        const float scale = (hi - lo) / denom;
        hipLaunchKernelGGL(k_dem_normalize, dim3(blocks), dim3(256), 0, s, d->d_h, n, 0, st[0], scale, lo);
    } else {
        const float denom = std::fmax(st[3], eps);                       // :945
        hipLaunchKernelGGL(k_dem_normalize, dim3(blocks), dim3(256), 0, s, d->d_h, n, 1, st[2], denom, 0.0f);
    }
    VF_HIP_TRY(hipGetLastError());
    VF_HIP_TRY(hipStreamSynchronize(s));
    return VF_OK;
}

int vf_dem_upload_height(vf_dem *d)
{
    if (!d) return fail(VF_ERR_INVALID, "NULL argument");
    if (!d->d_h || d->w == 0) return fail(VF_ERR_INVALID, "no terrain uploaded; call add_terrain() first");
    VF_HIP_TRY(hipSetDevice(d->ctx->device));
    const size_t n = (size_t)d->w * d->h;
    if (n > d->tex_cap) {
        if (d->d_tex) VF_HIP_TRY(hipFree(d->d_tex));
        d->d_tex = nullptr; d->tex_cap = 0;
        VF_HIP_TRY(hipMalloc(&d->d_tex, n * sizeof(float)));
        d->tex_cap = n;
    }
    VF_HIP_TRY(hipMemcpyAsync(d->d_tex, d->d_h, n * sizeof(float), hipMemcpyDeviceToDevice, d->ctx->stream));
    VF_HIP_TRY(hipStreamSynchronize(d->ctx->stream));
    d->tex_w = d->w; d->tex_h = d->h;
    return VF_OK;
}

int vf_dem_texture_size(const vf_dem *d, uint32_t *w, uint32_t *h)
{
    if (!d || !w || !h) return fail(VF_ERR_INVALID, "NULL argument");
    *w = d->tex_w; *h = d->tex_h;
    return VF_OK;
}

int vf_dem_read_patch(vf_dem *d, uint32_t x, uint32_t y, uint32_t w, uint32_t h, float *dst)
{
    if (!d || !dst) return fail(VF_ERR_INVALID, "NULL argument");
    if (w == 0 || h == 0) return fail(VF_ERR_INVALID, "patch dimensions must be > 0");
    if (!d->d_tex) return fail(VF_ERR_INVALID, "no height texture uploaded; call upload_height_r32f() first");
    if ((uint64_t)x + w > d->tex_w)
        return fail(VF_ERR_INVALID, "requested patch exceeds texture bounds in x: x+w (" + std::to_string((uint64_t)x + w) + ") > width (" + std::to_string(d->tex_w) + ")");
    if ((uint64_t)y + h > d->tex_h)
        return fail(VF_ERR_INVALID, "requested patch exceeds texture bounds in y: y+h (" + std::to_string((uint64_t)y + h) + ") > height (" + std::to_string(d->tex_h) + ")");
    VF_HIP_TRY(hipSetDevice(d->ctx->device));
    VF_HIP_TRY(hipMemcpy2D(dst, (size_t)w * 4, d->d_tex + (size_t)y * d->tex_w + x, (size_t)d->tex_w * 4, (size_t)w * 4, h, hipMemcpyDeviceToHost));
    return VF_OK;
}

int vf_stitch_bands_device(vf_ctx *ctx, const void *dev_gathered, void *dev_image, uint32_t width, uint32_t height,
                           uint32_t nranks, uint32_t band_h, void *stream)
{
    if (!ctx || !dev_gathered || !dev_image) return fail(VF_ERR_INVALID, "NULL argument");
    if (!is_pow2(band_h) || nranks == 0) return fail(VF_ERR_INVALID, "band_h must be a power of two, nranks > 0");
    if (width % 4 != 0) return fail(VF_ERR_INVALID, "width must be a multiple of 4");
    if (height % (band_h * nranks) != 0) return fail(VF_ERR_INVALID, "height must be a multiple of band_h*nranks");
    VF_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    uint32_t row_vec4 = width / 4;
    size_t nvec = (size_t)height * row_vec4;
    hipLaunchKernelGGL(k_stitch_bands, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, s, (const uint4 *)dev_gathered,
                       (uint4 *)dev_image, row_vec4, height, nranks, ilog2(band_h), band_h, height / nranks);
    VF_HIP_TRY(hipGetLastError());
    return VF_OK;
}

int vf_stitch_tiles_device(vf_ctx *ctx, const void *dev_gathered, void *dev_image, uint32_t width, uint32_t height,
                           uint32_t nranks, uint32_t skew, uint32_t stride_tiles, void *stream)
{
    if (!ctx || !dev_gathered || !dev_image) return fail(VF_ERR_INVALID, "NULL argument");
    if (width == 0 || height == 0 || nranks == 0) return fail(VF_ERR_INVALID, "empty frame or nranks == 0");
    const uint32_t ntx = (width + kTileW - 1) / kTileW, nty = (height + kTileH - 1) / kTileH;
    uint32_t most = 0;
    for (uint32_t r = 0; r < nranks; ++r) {
        uint32_t n = 0;
        int rc = vf_tile_layout(width, height, r, nranks, skew, nullptr, 0, &n);
        if (rc != VF_OK) return rc;
        most = n > most ? n : most;
    }
    if (stride_tiles < most) return fail(VF_ERR_INVALID, "stride_tiles is smaller than the largest shard");
    VF_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    // persistent, about one small workgroup per CU (VF_STITCH_WGS overrides): it has to fit beside the next frame's tile kernel
    uint32_t wgs = (uint32_t)std::max(1, ctx->prop.multiProcessorCount);
    if (const char *e = std::getenv("VF_STITCH_WGS")) wgs = (uint32_t)std::max(1, std::atoi(e));
    const uint8_t *d_owner = nullptr;
    if (const StripeMap *m = layout_map(skew)) {            // a registered stripe map: its owner table on this device (uploaded once per context)
        const uint32_t id = (skew >> 21) & 0x3Fu;
        std::lock_guard<std::mutex> lk(ctx->lazy_mu);
        if (!ctx->d_maps) VF_HIP_TRY(hipMalloc(&ctx->d_maps, (size_t)kMaxStripeMaps * kMaxStripes));
        if (!((ctx->maps_uploaded >> id) & 1ull)) {
            VF_HIP_TRY(hipMemcpy(ctx->d_maps + (size_t)id * kMaxStripes, m->owner.data(), m->owner.size(), hipMemcpyHostToDevice));
            ctx->maps_uploaded |= 1ull << id;
        }
        d_owner = ctx->d_maps + (size_t)id * kMaxStripes;
    }
    hipLaunchKernelGGL(k_stitch_tiles, dim3(std::min(ntx * nty, wgs)), dim3(256), 0, s, (const uint32_t *)dev_gathered, (uint32_t *)dev_image, width, height,
                       ntx, nty, nranks, layout_skew(skew), layout_shift(skew), stride_tiles, d_owner);
    VF_HIP_TRY(hipGetLastError());
    return VF_OK;
}

} // extern "C"

// ---- multi-GPU exchange over RCCL ---------------------------------------------------------------------------------------
namespace {
struct Rccl {
    bool ok = false;
    std::string why;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;   // (optional)
    ncclResult_t (*GetVersion)(int *) = nullptr;                       // (optional)
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
// librccl.so.1 as already loaded in the process (PyTorch brings its own copy; a second one would not share its state), else
// the ROCm installation's through this library's run path
const Rccl &resolve_rccl()
{
    static const Rccl r = [] {
        Rccl x;
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) { x.why = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return x; }
        bool all = true;
        auto sym = [&](const char *name) { void *p = dlsym(h, name); if (!p) { all = false; x.why = std::string("RCCL lacks ") + name; } return p; };
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
        x.CommCount = reinterpret_cast<decltype(x.CommCount)>(sym("ncclCommCount"));
        x.CommUserRank = reinterpret_cast<decltype(x.CommUserRank)>(sym("ncclCommUserRank"));
        x.CommCuDevice = reinterpret_cast<decltype(x.CommCuDevice)>(dlsym(h, "ncclCommCuDevice"));
        x.GetVersion = reinterpret_cast<decltype(x.GetVersion)>(dlsym(h, "ncclGetVersion"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
        x.Send = reinterpret_cast<decltype(x.Send)>(sym("ncclSend"));
        x.Recv = reinterpret_cast<decltype(x.Recv)>(sym("ncclRecv"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
        x.ok = all;
        return x;
    }();
    return r;
}
#define VF_RCCL_TRY(R, expr)                                                                                   \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return fail(VF_ERR_HIP, std::string(#expr) + ": " + (R).GetErrorString(r_));   \
    } while (0)

// rank and size of the communicator must be the handle's shard: a mismatch would exchange the wrong slabs silently
int check_comm(const Rccl &R, const vf_terrain *t, void *comm, int root)
{
    if (!R.ok) return fail(VF_ERR_HIP, R.why);
    if (!comm) return fail(VF_ERR_INVALID, "communicator is NULL");
    int n = 0, r = -1;
    VF_RCCL_TRY(R, R.CommCount((ncclComm_t)comm, &n));
    VF_RCCL_TRY(R, R.CommUserRank((ncclComm_t)comm, &r));
    if ((uint32_t)n != t->nranks || (uint32_t)r != t->rank) return fail(VF_ERR_INVALID, "communicator rank/size differ from the handle's shard");
    if (root < 0 || root >= n) return fail(VF_ERR_INVALID, "root must be a rank of the communicator");
    return VF_OK;
}
} // namespace

extern "C" {

int vf_dist_available(void) { return resolve_rccl().ok ? 1 : 0; }

int vf_dist_version(int *version)
{
    if (!version) return fail(VF_ERR_INVALID, "NULL argument");
    *version = 0;
    const Rccl &R = resolve_rccl();
    if (!R.ok) return fail(VF_ERR_HIP, R.why);
    if (!R.GetVersion) return fail(VF_ERR_HIP, "RCCL lacks ncclGetVersion");
    VF_RCCL_TRY(R, R.GetVersion(version));
    return VF_OK;
}

int vf_dist_unique_id(uint8_t id[VF_DIST_UNIQUE_ID_BYTES])
{
    if (!id) return fail(VF_ERR_INVALID, "NULL argument");
    const Rccl &R = resolve_rccl();
    if (!R.ok) return fail(VF_ERR_HIP, R.why);
    static_assert(sizeof(ncclUniqueId) == VF_DIST_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    VF_RCCL_TRY(R, R.GetUniqueId(&u));
    std::memcpy(id, u.internal, sizeof u.internal);
    return VF_OK;
}

int vf_dist_comm_init(vf_ctx *ctx, const uint8_t id[VF_DIST_UNIQUE_ID_BYTES], int rank, int nranks, void **comm)
{
    if (!ctx || !id || !comm) return fail(VF_ERR_INVALID, "NULL argument");
    *comm = nullptr;
    if (nranks <= 0 || rank < 0 || rank >= nranks) return fail(VF_ERR_INVALID, "rank must be < nranks");
    const Rccl &R = resolve_rccl();
    if (!R.ok) return fail(VF_ERR_HIP, R.why);
    VF_HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, sizeof u.internal);
    ncclComm_t c = nullptr;
    VF_RCCL_TRY(R, R.CommInitRank(&c, nranks, u, rank));
    *comm = c;
    return VF_OK;
}

void vf_dist_comm_destroy(void *comm)
{
    const Rccl &R = resolve_rccl();
    if (!R.ok || !comm) return;
    int dev = -1, was = -1;                                 // the communicator's own device, whatever the calling thread has current ...
    const bool have_was = hipGetDevice(&was) == hipSuccess;
    if (R.CommCuDevice && R.CommCuDevice((ncclComm_t)comm, &dev) == ncclSuccess && dev >= 0) (void)hipSetDevice(dev);
    (void)R.CommDestroy((ncclComm_t)comm);
    if (have_was && was >= 0) (void)hipSetDevice(was);     // ... which is the caller's again afterwards
}

int vf_dist_gather_tiles(vf_terrain *t, void *rccl_comm, int root, void *dev_gathered, uint32_t stride_tiles, void *stream)
{
    if (!t) return fail(VF_ERR_INVALID, "NULL argument");
    if (!t->shard_tiles) return fail(VF_ERR_INVALID, "handle is not tile-sharded");
    if (!t->rendered) return fail(VF_ERR_INVALID, "nothing rendered yet");
    const Rccl &R = resolve_rccl();
    int rc = check_comm(R, t, rccl_comm, root);
    if (rc != VF_OK) return rc;
    VF_HIP_TRY(hipSetDevice(t->ctx->device));
    ncclComm_t comm = (ncclComm_t)rccl_comm;
    hipStream_t s = stream ? (hipStream_t)stream : t->ctx->stream;
    if (s != t->last_stream && t->last_stream) {            // the slab is being drawn on another stream: order the exchange after it
        VF_HIP_TRY(hipStreamWaitEvent(s, t->ps[t->last_set].drawn, 0));
    }
    const size_t tile_bytes = (size_t)kTileW * kTileH * 4;
    const bool is_root = (uint32_t)root == t->rank;
    // Checked on EVERY rank, from what every rank can compute locally, before anything is posted: a rank that returned here
    // while the others had queued their ncclSend would leave them waiting for a receive that never comes.  (The one check only
    // the root can make -- its gather buffer -- must hold by construction: a root that passes NULL fails alone and the senders
    // hang; vf_hip.h says so.)
    for (uint32_t r = 0; r < t->nranks; ++r) {
        uint32_t n = 0;
        rc = vf_tile_layout(t->W, t->H, r, t->nranks, t->skew, nullptr, 0, &n);
        if (rc != VF_OK) return rc;
        if (n > stride_tiles) return fail(VF_ERR_INVALID, "stride_tiles is smaller than the largest shard");
    }
    if (is_root && !dev_gathered) return fail(VF_ERR_INVALID, "the root needs the gather buffer");
    uint8_t *const slots = (uint8_t *)dev_gathered;
    uint8_t *const own_slot = is_root ? slots + (size_t)root * stride_tiles * tile_bytes : nullptr;
    const bool in_place = is_root && (uint8_t *)t->d_rgba == own_slot;
    VF_RCCL_TRY(R, R.GroupStart());
    ncclResult_t res = ncclSuccess;
    if (is_root) {
        for (uint32_t r = 0; r < t->nranks && res == ncclSuccess; ++r) {
            if (r == t->rank && in_place) continue;         // the root drew straight into its slot
            uint32_t n = 0;
            (void)vf_tile_layout(t->W, t->H, r, t->nranks, t->skew, nullptr, 0, &n);
            if (n) res = R.Recv(slots + (size_t)r * stride_tiles * tile_bytes, (size_t)n * tile_bytes, ncclUint8, (int)r, comm, s);
        }
    }
    if (res == ncclSuccess && (!is_root || !in_place) && t->local_tiles)
        res = R.Send(t->d_rgba, (size_t)t->local_tiles * tile_bytes, ncclUint8, root, comm, s);
    const ncclResult_t end = R.GroupEnd();
    if (res != ncclSuccess) return fail(VF_ERR_HIP, std::string("ncclSend/ncclRecv: ") + R.GetErrorString(res));
    if (end != ncclSuccess) return fail(VF_ERR_HIP, std::string("ncclGroupEnd: ") + R.GetErrorString(end));
    return VF_OK;
}

int vf_dist_gather_bands(vf_terrain *t, void *rccl_comm, int root, void *dev_image, void *stream)
{
    if (!t) return fail(VF_ERR_INVALID, "NULL argument");
    if (t->shard_tiles) return fail(VF_ERR_INVALID, "handle is tile-sharded: gather with vf_dist_gather_tiles");
    if (!t->rendered) return fail(VF_ERR_INVALID, "nothing rendered yet");
    const Rccl &R = resolve_rccl();
    int rc = check_comm(R, t, rccl_comm, root);
    if (rc != VF_OK) return rc;
    VF_HIP_TRY(hipSetDevice(t->ctx->device));
    ncclComm_t comm = (ncclComm_t)rccl_comm;
    hipStream_t s = stream ? (hipStream_t)stream : t->ctx->stream;
    if (s != t->last_stream && t->last_stream) VF_HIP_TRY(hipStreamWaitEvent(s, t->ps[t->last_set].drawn, 0));
    const bool is_root = (uint32_t)root == t->rank;
    if (is_root && !dev_image) return fail(VF_ERR_INVALID, "the root needs the image buffer");
    const size_t row_bytes = (size_t)t->W * 4;
    std::vector<uint32_t> local(t->nranks, 0u);             // rows a rank has stored so far, band by band
    // the root's own bands first, as plain device copies on the stream: nothing but RCCL calls inside the group
    if (is_root) {
        uint32_t ly0 = 0;
        for (uint32_t b = t->rank; b * t->band_h < t->H; b += t->nranks) {
            const uint32_t y0 = b * t->band_h, rows = std::min(t->band_h, t->H - y0);
            uint8_t *dst = (uint8_t *)dev_image + (size_t)y0 * row_bytes;
            const uint8_t *src = (const uint8_t *)t->d_rgba + (size_t)ly0 * row_bytes;
            if ((const uint8_t *)dst != src) VF_HIP_TRY(hipMemcpyAsync(dst, src, rows * row_bytes, hipMemcpyDeviceToDevice, s));
            ly0 += rows;
        }
    }
    VF_RCCL_TRY(R, R.GroupStart());
    ncclResult_t res = ncclSuccess;
    for (uint32_t b = 0; b * t->band_h < t->H && res == ncclSuccess; ++b) {
        const uint32_t y0 = b * t->band_h, rows = std::min(t->band_h, t->H - y0), owner = b % t->nranks, ly0 = local[owner];
        local[owner] += rows;
        uint8_t *dst = is_root ? (uint8_t *)dev_image + (size_t)y0 * row_bytes : nullptr;
        const uint8_t *src = (const uint8_t *)t->d_rgba + (size_t)ly0 * row_bytes;
        if (owner == t->rank && is_root) continue;          // copied above
        else if (is_root) res = R.Recv(dst, rows * row_bytes, ncclUint8, (int)owner, comm, s);
        else if (owner == t->rank) res = R.Send(src, rows * row_bytes, ncclUint8, root, comm, s);
    }
    const ncclResult_t end = R.GroupEnd();
    if (res != ncclSuccess) return fail(VF_ERR_HIP, std::string("ncclSend/ncclRecv: ") + R.GetErrorString(res));
    if (end != ncclSuccess) return fail(VF_ERR_HIP, std::string("ncclGroupEnd: ") + R.GetErrorString(end));
    return VF_OK;
}

// Tile shards (column stripes: skew 0) -> the row-major frame on rank `root` without any rank copying the whole frame: the stitch is
// sharded like the rendering.  (The steps are those of vulkan_forge_amd/dist.py::BandStitchExchange, here on RCCL directly.)
int vf_dist_exchange_bands(vf_terrain *t, void *rccl_comm, int root, void *dev_image, void *stream)
{
    if (!t) return fail(VF_ERR_INVALID, "NULL argument");
    if (!t->shard_tiles) return fail(VF_ERR_INVALID, "handle is not tile-sharded");
    if (!t->rendered) return fail(VF_ERR_INVALID, "nothing rendered yet");
    const Rccl &R = resolve_rccl();
    int rc = check_comm(R, t, rccl_comm, root);
    if (rc != VF_OK) return rc;
    const uint32_t N = t->nranks;
    // every rank can check the layout for itself, before anything is posted: all ranks fail together
    if (layout_skew(t->skew) != 0u) return fail(VF_ERR_INVALID, "the band exchange needs column stripes (tile shard with skew 0)");
    if (t->W % (uint32_t)kTileW || t->H % (uint32_t)kTileH || t->ntx % (N << layout_shift(t->skew)) || t->nty % N)
        return fail(VF_ERR_INVALID, "the band exchange needs whole tiles, stripes that divide the tile columns evenly among the ranks and tile rows that divide by the number of ranks");
    const bool is_root = (uint32_t)root == t->rank;
    if (is_root && !dev_image) return fail(VF_ERR_INVALID, "the root needs the image buffer");
    VF_HIP_TRY(hipSetDevice(t->ctx->device));
    ncclComm_t comm = (ncclComm_t)rccl_comm;
    hipStream_t s = stream ? (hipStream_t)stream : t->ctx->stream;
    if (s != t->last_stream && t->last_stream) VF_HIP_TRY(hipStreamWaitEvent(s, t->ps[t->last_set].drawn, 0));
    const size_t tile_bytes = (size_t)kTileW * kTileH * 4;
    const uint32_t band_tile_rows = t->nty / N, chunk_tiles = band_tile_rows * (t->ntx / N);   // tiles one rank holds of one band
    const size_t chunk_bytes = (size_t)chunk_tiles * tile_bytes;
    const uint32_t band_rows = t->H / N;
    const size_t band_bytes = (size_t)band_rows * t->W * 4;
    // the handle's own staging: what this rank receives ([N][chunk_tiles] tile slots) and, off the root, the band it stitches.
    // Reused by every call: calls on one handle are ordered on one stream (or by the caller's events).
    if (t->xrecv_bytes < chunk_bytes * N) {
        if (t->d_xrecv) { VF_HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(t->d_xrecv); t->d_xrecv = nullptr; t->xrecv_bytes = 0; }
        hipError_t e = hipMalloc(&t->d_xrecv, chunk_bytes * N);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? VF_ERR_NOMEM : VF_ERR_HIP, std::string("exchange buffer: ") + hipGetErrorString(e));
        t->xrecv_bytes = chunk_bytes * N;
    }
    if (!is_root && t->xband_bytes < band_bytes) {
        if (t->d_xband) { VF_HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(t->d_xband); t->d_xband = nullptr; t->xband_bytes = 0; }
        hipError_t e = hipMalloc(&t->d_xband, band_bytes);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? VF_ERR_NOMEM : VF_ERR_HIP, std::string("exchange buffer: ") + hipGetErrorString(e));
        t->xband_bytes = band_bytes;
    }
    // 1. all-to-all: a rank's slab lists its tiles row-major by (ty, tx), so the tiles of band b are the contiguous chunk b; it goes
    //    to rank b -- 1 / N of a slab per peer, every xGMI link busy in both directions, none a hot spot.  (Own chunk: a device copy.)
    const uint8_t *slab = (const uint8_t *)t->d_rgba;
    // (a one-rank communicator sends its only chunk through RCCL to itself: the loop-back test then exercises the very calls N ranks make)
    if (N > 1u) VF_HIP_TRY(hipMemcpyAsync(t->d_xrecv + (size_t)t->rank * chunk_bytes, slab + (size_t)t->rank * chunk_bytes, chunk_bytes, hipMemcpyDeviceToDevice, s));
    {
        VF_RCCL_TRY(R, R.GroupStart());
        ncclResult_t res = ncclSuccess;
        for (uint32_t r = 0; r < N && res == ncclSuccess; ++r) {
            if (r == t->rank && N > 1u) continue;
            res = R.Send(slab + (size_t)r * chunk_bytes, chunk_bytes, ncclUint8, (int)r, comm, s);
            if (res == ncclSuccess) res = R.Recv(t->d_xrecv + (size_t)r * chunk_bytes, chunk_bytes, ncclUint8, (int)r, comm, s);
        }
        const ncclResult_t end = R.GroupEnd();
        if (res != ncclSuccess) return fail(VF_ERR_HIP, std::string("ncclSend/ncclRecv: ") + R.GetErrorString(res));
        if (end != ncclSuccess) return fail(VF_ERR_HIP, std::string("ncclGroupEnd: ") + R.GetErrorString(end));
    }
    // 2. every rank stitches ITS band (a frame of H / N rows whose tile (tx, ty) sits in slot [tx % N][ty * (ntx / N) + tx / N]):
    //    1 / N of the copy each; the root writes its band straight into the image
    uint8_t *const band = is_root ? (uint8_t *)dev_image + (size_t)t->rank * band_bytes : t->d_xband;
    rc = vf_stitch_tiles_device(t->ctx, t->d_xrecv, band, t->W, band_rows, N, t->skew, chunk_tiles, s);
    if (rc != VF_OK) return rc;
    // 3. the bands are contiguous slabs of the final image: the root receives them in place
    if (N > 1u) {
        VF_RCCL_TRY(R, R.GroupStart());
        ncclResult_t res = ncclSuccess;
        if (is_root) {
            for (uint32_t r = 0; r < N && res == ncclSuccess; ++r)
                if (r != t->rank) res = R.Recv((uint8_t *)dev_image + (size_t)r * band_bytes, band_bytes, ncclUint8, (int)r, comm, s);
        } else res = R.Send(band, band_bytes, ncclUint8, root, comm, s);
        const ncclResult_t end = R.GroupEnd();
        if (res != ncclSuccess) return fail(VF_ERR_HIP, std::string("ncclSend/ncclRecv: ") + R.GetErrorString(res));
        if (end != ncclSuccess) return fail(VF_ERR_HIP, std::string("ncclGroupEnd: ") + R.GetErrorString(end));
    }
    return VF_OK;
}

} // extern "C"
