/*
 * vf_hip.h -- C-ABI of the MI355X-native terrain rasteriser (libvf_hip.so).
 *
 * This is the drop-in boundary underneath the reference's `_vulkan_forge` extension module:
 * plain pointers and sizes, opaque handles, int status codes; no C++/torch/pybind types.
 * The reference has no C-ABI of its own (its boundary is PyO3, src/lib.rs:961-976); each entry
 * point below cites the reference code it replaces.  A Rust host (as BASELINE.json's north_star
 * words it) binds this header with `extern "C"` unchanged -- see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns VF_OK (0) or a negative VF_ERR_* code; vf_last_error() returns a
 *     thread-local, NUL-terminated description of the last failure on the calling thread.
 *   - host buffers are caller-owned and only borrowed for the duration of the call; device
 *     buffers passed to *_device entry points are borrowed until replaced or the object dies.
 *   - `stream` arguments are `hipStream_t` passed as `void*`; NULL = the object's own stream (a non-blocking stream of the context:
 *     NOT the legacy default stream -- a host that orders its own work by "the default stream", whose raw handle is 0 in HIP and in
 *     torch.cuda, must create an explicit stream and pass that, or synchronise with vf_terrain_sync).
 *     Calls on one handle are not re-entrant.  Rendering is asynchronous on the stream; the
 *     read_* entry points synchronise.
 *   - no CPU fallback exists: without a HIP device vf_ctx_create fails with VF_ERR_NO_DEVICE.
 */
#ifndef VF_HIP_H
#define VF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VF_OK 0
#define VF_ERR_NO_DEVICE (-1) /* no HIP device / runtime: reference message "No suitable GPU adapter" (src/terrain/mod.rs:285) */
#define VF_ERR_HIP (-2)       /* a HIP runtime call failed */
#define VF_ERR_INVALID (-3)   /* bad argument */
#define VF_ERR_NOMEM (-4)

typedef struct vf_ctx vf_ctx;         /* one HIP device + default stream; replaces the wgpu Instance/Adapter/Device/Queue
                                         (src/terrain/mod.rs:277-294, src/scene/mod.rs:62-70, src/lib.rs:23-61) */
typedef struct vf_terrain vf_terrain; /* render target + pipeline state + mesh + bind groups of TerrainSpike/Scene
                                         (src/terrain/mod.rs:221-253, src/scene/mod.rs:25-55) */

typedef struct vf_device_info {
    char name[256];
    char arch[64];          /* gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
    int32_t device_ordinal;
    int32_t compute_units;
    int32_t wavefront_size;
    int32_t clock_khz;
    uint64_t total_mem_bytes;
    uint64_t lds_bytes_per_cu;
    int32_t pci_bus_id;
    int32_t pci_device_id;
} vf_device_info;

/* device time (HIP events), averaged over the frames rendered since vf_terrain_enable_timing(t, 1 or 2) -- at most the last 64.
 * A frame's plan kernels run on a stream of the handle's own and overlap the previous frame's tile kernel, so the first two
 * figures are elapsed times on that stream (they include waiting for free CUs when frames are rendered back to back). */
typedef struct vf_timings {
    float ranges_ms;    /* k_block_boxes (per-block pixel boxes and capsules) up to the launch of k_block_setup (the plan's stream, elapsed) */
    float plan_ms;      /* k_plan + k_plan_sort: background flags, busy-tile list, heaviest first (the plan's stream, elapsed); both: the last
                         * plan of each of the handle's two plan states (round 6: a plan may be queued ahead of its frame's call) */
    float tile_ms;      /* k_clear + k_tile (both variants) on the caller's stream: background clear, LDS raster from the set-up pass's records, fragment stage */
    float total_ms;     /* frame period when >= 2 frames were timed back to back, else plan start -> RGBA8 complete */
    uint32_t blocks_rasterised; /* (tile, block) pairs the tile kernel processed (after early-out); 0 at timing level 2 */
    uint32_t tiles;             /* workgroups launched = owned screen tiles */
    uint32_t frames;            /* frames averaged */
    uint32_t blocks_distinct;   /* distinct grid blocks behind blocks_rasterised in the last frame (of grid blocks in total: ceil((n-1)/8)^2) */
} vf_timings;

const char *vf_last_error(void);

/* ---- context ----------------------------------------------------------------------------- */
/* replaces wgpu adapter enumeration (src/lib.rs:746-777) */
int vf_device_count(int *count);
int vf_device_query(int device_ordinal, vf_device_info *out);
/* replaces Instance::new + request_adapter + request_device (src/terrain/mod.rs:277-294) */
int vf_ctx_create(int device_ordinal, vf_ctx **out);
void vf_ctx_destroy(vf_ctx *ctx);
int vf_ctx_device_info(const vf_ctx *ctx, vf_device_info *out);
/* The context's own stream (what a NULL `stream` argument means), as a hipStream_t: for a host that wants to order its own work
 * -- events, copies, collectives -- on the stream the frames are drawn on without creating one more (torch:
 * torch.cuda.ExternalStream(handle)).  Streams are a scarce resource here: the HIP runtime maps them onto GPU_MAX_HW_QUEUES
 * hardware queues (default 4) round-robin, a handle uses three (this one and two of its own for the next frame's plan and set-up
 * pass), and two streams that land on one queue run their kernels one after the other -- the next frame's set-up then no longer
 * hides under this frame's tile kernel (C4: 0.87 -> 0.99 ms per frame, measured).  A host with streams of its own should raise
 * GPU_MAX_HW_QUEUES (environment, before the HIP runtime starts; bench.py sets 8). */
int vf_ctx_stream(const vf_ctx *ctx, void **stream);

/* ---- terrain object ---------------------------------------------------------------------- */
/*
 * Replaces TerrainSpike::new / Scene::new minus the Python-side validation
 * (src/terrain/mod.rs:259-407, src/scene/mod.rs:60-206): colour target W x H (Rgba8UnormSrgb),
 * fixed pipeline state (src/terrain/pipeline.rs:97-139), the procedural n x n grid of
 * build_grid_xyuv (src/terrain/mod.rs:553-598; never materialised), the 256x1 LUT
 * (ColormapLUT::new, src/terrain/mod.rs:31-110).  lut_is_srgb = 1: bytes are sRGB-encoded and
 * decoded per texel before filtering (Rgba8UnormSrgb); 0: bytes are used as UNORM (the
 * VF_FORCE_LUT_UNORM fallback).  The height texture starts as a 1x1 zero texel
 * (src/terrain/mod.rs:342-378); uniforms start zeroed -- call vf_terrain_set_uniforms.
 * grid < 2 is raised to 2 like the reference (`.max(2)`).
 */
int vf_terrain_create(vf_ctx *ctx, uint32_t width, uint32_t height, uint32_t grid,
                      const uint8_t lut_rgba8[1024], int lut_is_srgb, vf_terrain **out);
void vf_terrain_destroy(vf_terrain *t);

/* queue.write_buffer(ubo, TerrainUniforms) -- 44 floats, layout of src/terrain/mod.rs:114-123 */
int vf_terrain_set_uniforms(vf_terrain *t, const float uniforms[44]);

/* Scene::set_height_from_r32f (src/scene/mod.rs:226-276): R32F texture (th rows x tw cols, tightly
 * packed float32, no 256-byte row padding needed), nearest, clamp-to-edge.  Copies host -> HBM. */
int vf_terrain_set_height(vf_terrain *t, const float *host_height, uint32_t tw, uint32_t th);
/* same, borrowing a texture already resident in HBM (e.g. a torch tensor's data_ptr) */
int vf_terrain_set_height_device(vf_terrain *t, const float *dev_height, uint32_t tw, uint32_t th);

/* Fragment-stage variant.  VF_SHADE_REFERENCE (default) is fs_main as coded (src/shaders/terrain.wgsl:69-91): analytic
 * normals, no tonemap -- the only mode for which parity with the reference is claimed.  VF_SHADE_SPEC_T32 is the stage the
 * reference documents but never implemented (ROADMAP.md:421-436 forward-difference normals from the height texture;
 * README.md:128,174-175 Reinhard in linear before the sRGB store); it is validated against this build's oracle only. */
#define VF_SHADE_REFERENCE 0
#define VF_SHADE_SPEC_T32 1
int vf_terrain_set_shade_mode(vf_terrain *t, int mode);
/* Fragment-stage arithmetic (new; the reference leaves it to the driver's shader compiler, src/shaders/terrain.wgsl:69-91).
 * VF_PRECISION_FAST (default): hardware reciprocal / rsq / sin / cos / log / exp and fused multiply-adds -- visibility is
 * unaffected, RGBA stays within 1 LSB per channel of VF_PRECISION_EXACT (BASELINE.json: "RGBA within +-1 LSB of reference"),
 * which evaluates every operation as a correctly rounded IEEE binary32 one in a fixed order and reproduces the CPU oracle bit
 * for bit.  VF_SHADE_SPEC_T32 frames are always drawn with the exact arithmetic. */
#define VF_PRECISION_EXACT 0
#define VF_PRECISION_FAST 1
int vf_terrain_set_shade_precision(vf_terrain *t, int precision);
/* The raster stage's line loop exists twice (round 4): with and without a first pass that tests a triangle's lines in groups of four
 * against the final-pixel masks.  Which one is faster depends on the view and on how the frame is cut (wide items with many-line
 * triangles gain, a multi-GPU rank's narrow strips lose); the picture never differs.  mode -1 (default): the handle times both on
 * its own frames (one HIP event at the end of a frame's work on the draw stream; the time between two consecutive frames of one
 * variant -- the frame period -- read back frames later, never a wait) and keeps the faster one per view; re-measured when the shard
 * layout, the heights or (at the start of a motion) the camera change;
 * 0 / 1: fixed.  No reference counterpart: the fixed-function raster behind draw_indexed, src/terrain/mod.rs:435. */
int vf_terrain_set_raster_groups(vf_terrain *t, int mode);
/* which variant drew the last frame (0 / 1) and, when measured, the frame period with each in this view (ms; 0 = not measured) */
int vf_terrain_raster_groups(const vf_terrain *t, int *in_use, float ms[2]);

/* Multi-GPU screen split (new; the reference is single-device).  Pixel row y belongs to this
 * object iff ((y / band_h) % nranks) == rank; owned rows are stored densely ("local rows") in
 * band order.  band_h must be a power of two.  Default: rank 0 of 1 (all rows). */
int vf_terrain_set_shard(vf_terrain *t, uint32_t rank, uint32_t nranks, uint32_t band_h);
int vf_terrain_local_rows(const vf_terrain *t, uint32_t *rows);

/* Multi-GPU screen split by interleaved 64 x 64 tiles ("screen-tile split", BASELINE.json configs[3]; new, the reference
 * is single-device).  The `layout` argument of vf_terrain_set_tile_shard, vf_tile_layout and vf_stitch_tiles_device is ONE WORD,
 * VF_TILE_LAYOUT(skew, stripe_log2) = skew | stripe_log2 << 16:
 *     tile (tx, ty) belongs to rank ((tx >> stripe_log2) + skew * ty) % nranks
 * -- column stripes 2^stripe_log2 tiles wide dealt round-robin to the ranks, each tile row shifted by `skew` stripes.  A plain
 * number < 65536 is therefore a skew with single-tile stripes (the meaning the argument had before round 4).  skew 0 = column
 * stripes (what vf_dist_exchange_bands needs, and bench.py's default with a period of eight tile columns: stripe_log2 2 / 1 / 0
 * for 2 / 4 / 8 ranks).  skew < 65536 and stripe_log2 <= 15; a word with any higher bit set is refused (VF_ERR_INVALID).
 * A rank stores its tiles densely, row-major by (ty, tx), each as 64 x 64 RGBA8 words (edge tiles keep the full slot): local
 * tile k starts at byte k * 16384.  vf_tile_layout lists a rank's tiles (tx | ty << 16; pure host arithmetic, no device needed;
 * tiles may be NULL to count).  A tile-sharded handle is read with vf_terrain_read_tiles; rows come back after
 * vf_stitch_tiles_device. */
#define VF_TILE_LAYOUT(skew, stripe_log2) ((uint32_t)(skew) | ((uint32_t)(stripe_log2) << 16))
int vf_terrain_set_tile_shard(vf_terrain *t, uint32_t rank, uint32_t nranks, uint32_t layout);
/* Load-balanced stripes (round 5; SURVEY.md 8(e) "a load-balanced map").  The round-robin deal gives every rank the same NUMBER of
 * stripes, not the same work: at C4's top-down camera the slowest of 4 ranks carries 15 % more than the fastest.  Three pieces:
 *   vf_terrain_tile_times      the time (ms) the handle's last frame spent on each of its local tiles (what its own frame plan
 *                              feeds on); summed per stripe and added up over the ranks by the host (an all-reduce of <= 256 floats)
 *                              they are the per-stripe cost every rank agrees on.  *count = local tiles; ms may be NULL to count.
 *   vf_balance_stripes         the deterministic rule: heaviest stripe first, each to the least loaded rank that still has room --
 *                              every rank keeps nstripes / nranks stripes, so slabs and exchanged chunks keep their sizes; the same
 *                              times give the same table on every rank.
 *   vf_tile_layout_register_map  an owner table (owner[sc] for the stripe sc = tx >> stripe_log2) -> a layout WORD (bit 20 set) that
 *                              every entry point taking a layout accepts: vf_terrain_set_tile_shard, vf_tile_layout,
 *                              vf_stitch_tiles_device, and through the handle vf_dist_exchange_bands / vf_dist_gather_tiles.  Tables
 *                              are immutable and process-wide (64 at most); registering the same table again returns the same word.
 * No reference counterpart (single device: src/terrain/mod.rs:277-294). */
int vf_terrain_tile_times(vf_terrain *t, float *ms, uint32_t capacity, uint32_t *count);
int vf_balance_stripes(const float *stripe_ms, uint32_t nstripes, uint32_t nranks, uint8_t *stripe_owner);
int vf_tile_layout_register_map(const uint8_t *stripe_owner, uint32_t nstripes, uint32_t stripe_log2, uint32_t nranks, uint32_t *layout);
int vf_terrain_local_tiles(const vf_terrain *t, uint32_t *tiles);
int vf_terrain_read_tiles(vf_terrain *t, uint8_t *dst, uint32_t first, uint32_t count);
int vf_tile_layout(uint32_t width, uint32_t height, uint32_t rank, uint32_t nranks, uint32_t layout, uint32_t *tiles,
                   uint32_t capacity, uint32_t *count);

/* Render into a caller-provided device buffer of local_rows*W*4 bytes (tile shards: local_tiles*16384 bytes);
 * NULL = internal buffer. */
int vf_terrain_set_output_device(vf_terrain *t, void *dev_rgba);
int vf_terrain_rgba_device(const vf_terrain *t, void **dev_rgba);

/*
 * The render pass of render_png (src/terrain/mod.rs:412-437, src/scene/mod.rs:280-298): clear to
 * linear (0.02,0.02,0.03,1), one indexed draw of 6(n-1)^2 indices, vs_main / raster / fs_main
 * (src/shaders/terrain.wgsl:44-91), sRGB store.  Leaves tightly packed RGBA8 (local rows) in HBM.
 * Asynchronous on `stream` (NULL = the context's stream): the output is complete when work queued on `stream` after this
 * call runs.  How the frame's work is ordered and split follows the times measured on this handle's earlier frames -- that
 * steers speed only, never a pixel.  Where the planning kernels run (round 6): for a caller that WAITS for every frame, on
 * `stream` itself -- and the plan of its next frame, when the inputs have not changed, behind the copy of its next read-back
 * (vf_terrain_read_rgba / _read_png_scanlines return when the copy is done, not when that plan is); for a caller that submits
 * a frame while the previous one is in flight, on two streams of the context (made once per process, 17-20 ms), under the
 * previous frame's tile kernel.
 */
int vf_terrain_render(vf_terrain *t, void *stream);
/* BASELINE.json configs[4] ("batch of 64 camera look-ats over one terrain"): n render passes with n uniform blocks over the handle's
 * terrain, queued back to back on `stream` -- the loop Scene.set_camera_look_at + render_png runs per pose (src/scene/mod.rs:208-224,
 * :278-335) without the per-pose FFI round trips and read-backs.  uniforms: n x 44 floats (vf_terrain_set_uniforms layout);
 * dev_rgba: n device buffers, frame k goes to dev_rgba[k] (NULL: every frame into the current output buffer, the last one stays).
 * Every pose is planned from the tile times of the pose before last, looked up through the camera motion between the two (the
 * homography of the ground plane): no pose waits for its predecessor's times, the plan and the set-up pass of pose k + 1 run under
 * pose k's tile kernel.  Afterwards the handle's uniforms are the last pose's and its output buffer is dev_rgba[n - 1].
 * Multi-GPU (pose-parallel replicas, no collective): rank r passes the poses k = r mod N. */
int vf_terrain_render_batch(vf_terrain *t, const float *uniforms, uint32_t n, void *const *dev_rgba, void *stream);
/* The same batch with every frame read back (the per-pose copy_texture_to_buffer + map of src/terrain/mod.rs:439-485): pose k is drawn
 * into one of three device frames owned by the handle and copied to host_rgba[k] (H*W*4 bytes each; page-locked destinations --
 * vf_host_alloc -- travel by DMA while the next poses are drawn).  Returns when every frame has arrived.  Whole-frame handles only.
 * The handle's own output buffer is left untouched (and "nothing rendered yet" holds for it afterwards). */
int vf_terrain_render_batch_host(vf_terrain *t, const float *uniforms, uint32_t n, uint8_t *const *host_rgba);
int vf_terrain_sync(vf_terrain *t);

/* copy_texture_to_buffer + map + un-pad (src/terrain/mod.rs:439-485): local rows [y0, y0+rows)
 * into dst (rows*W*4 bytes).  Waits for the last render and for the copy (work the library queues behind the copy for the
 * caller's NEXT frame may still be running when this returns: later calls on the handle are ordered after it). */
int vf_terrain_read_rgba(vf_terrain *t, uint8_t *dst, uint32_t y0, uint32_t rows);
/* Page-locked host memory for read-back destinations (new; the reference maps a fresh staging buffer per call,
 * src/terrain/mod.rs:446-451).  vf_terrain_read_rgba into such a buffer is ONE DMA transfer (C4: 64 MiB in 1.2 ms); into ordinary
 * pageable memory it goes through the handle's ring of pinned chunks and a copy by host threads, and a frame-sized fresh
 * destination is page-fault bound (2-5 ms).  Any hipHostMalloc / hipHostRegister memory of the caller's is recognised as well.
 * Buffers of 4 MiB and more are 2 MiB-aligned MADV_HUGEPAGE memory registered with the runtime (2.6 ms to make for 64 MiB, where
 * hipHostMalloc takes 9 and its first copy another 11-16); free them with vf_host_free only. */
int vf_host_alloc(size_t bytes, void **host);
void vf_host_free(void *host);
/* Read-back for render_png (src/terrain/mod.rs:439-490): the last frame as PNG scanlines -- per row one filter-type byte
 * (row-adaptive: the filter with the smallest sum of absolute residuals, chosen on the GPU) followed by W*4 filtered
 * bytes -- in pinned host memory owned by the handle (allocated once, not per call as the reference's read-back buffer
 * :446-451), valid until the next call on the handle.  The host only deflates.  Whole-frame handles only. */
int vf_terrain_read_png_scanlines(vf_terrain *t, const uint8_t **host_scanlines, size_t *nbytes);
/* debug/parity: per-pixel visible primitive id + 1 (0 = background), local rows.  The visibility tile never leaves LDS in a
 * normal frame, so this draws a frame again into scratch buffers: the frame vf_terrain_render drew last (its uniforms and
 * shade mode, whatever was set since), or -- before the first vf_terrain_render on this shard layout -- the current uniforms.
 * The caller's output buffer and the `rendered` state are left as they were. */
int vf_terrain_read_visibility(vf_terrain *t, uint32_t *dst);

/* enable: 0 off; 1 device times (HIP events around the frame's kernels) and the per-item statistics behind
 * vf_terrain_debug_item_stats / vf_timings.blocks_* (the tile kernel then writes them: a few atomics per drawn block);
 * 2 device times only -- the kernels run exactly as they do untimed (blocks_* read 0);
 * 3 as 2, for every FOURTH frame only (what bench.py times: the two event records a timed frame puts on the draw stream keep
 *   the next kernel from being launched under the previous one and cost a C4 frame 2 %, tools/exp_timing_cost.py); the frame
 *   periods vf_timings.total_ms / vf_terrain_frame_times report are then per frame, from events four frames apart; the plan chain
 *   carries no events at this level (ranges_ms / plan_ms read 0: measure them at level 2). */
int vf_terrain_enable_timing(vf_terrain *t, int enable);
int vf_terrain_timings(vf_terrain *t, vf_timings *out);
/* The same HIP events frame by frame (timing enabled; the frames since vf_terrain_enable_timing, at most the last 64, oldest
 * first): tile_ms[f] = k_clear + k_tile of frame f on the caller's stream, period_ms[f] = end of frame f - 1 -> end of frame f
 * (frames rendered back to back overlap: the period is what a frame costs; period_ms[0] = 0).  Either array may be NULL.
 * For the median / p95 report of python/tools/perf_sanity.py:45-69. */
int vf_terrain_frame_times(vf_terrain *t, float *tile_ms, float *period_ms, uint32_t max_frames, uint32_t *count);
/* diagnostics (timing enabled): per work item of the last frame (a busy tile, or one column strip of a heavy tile), in
 * launch order, 4 words: item code (local tile | strip << 20 | log2(strips) << 24 | depth slice << 27 | log2(slices) << 29), candidate blocks processed,
 * raster-phase time, raster+fragment time (10 ns ticks of the constant 100 MHz clock).  *count = items written. */
int vf_terrain_debug_item_stats(vf_terrain *t, uint32_t *dst, uint32_t max_items, uint32_t *count);
/* diagnostics, only in libraries built with -DVF_PHASE_PROF (VF_ERR_INVALID otherwise): shader-clock cycles summed over
 * all waves of the last frame's tile kernel, per phase (set-up, pull/cull, vertex stage, classification, span raster,
 * completion/rescan, end-of-chunk wait, fragment stage), then 8 event counts, then up to 8 parts of the set-up phase; n <= 32 */
int vf_terrain_debug_phase_cycles(vf_terrain *t, uint64_t *dst, uint32_t n);

/* ---- grid_generate ----------------------------------------------------------------------- */
/* make_grid (src/terrain/mesh.rs:35-90) computed on the GPU; outputs as the PyO3 wrapper returns
 * them (:149-203): xy (nx*nz,2) f32, uv (nx*nz,2) f32, idx 6(nx-1)(nz-1) u32.  Argument validation
 * (ValueError strings) is done by the host layer.  Host-pointer form copies back; device form
 * leaves results in HBM. */
int vf_grid_generate(vf_ctx *ctx, uint32_t nx, uint32_t nz, float dx, float dy,
                     float *xy, float *uv, uint32_t *idx);
int vf_grid_generate_device(vf_ctx *ctx, uint32_t nx, uint32_t nz, float dx, float dy,
                            float *dev_xy, float *dev_uv, uint32_t *dev_idx, void *stream);

/* ---- triangle smoke path ------------------------------------------------------------------ */
/* Renderer::render_triangle_rgba (src/lib.rs:286-309, :685-721; src/shaders/triangle.wgsl):
 * white clear, one colour-varying triangle, tightly packed (H,W,4) u8 into host memory. */
int vf_triangle_render(vf_ctx *ctx, uint32_t width, uint32_t height, uint8_t *rgba_host);

/* ---- Renderer DEM path ------------------------------------------------- */
/* HBM-resident heightmap of Renderer::add_terrain and its statistics / normalisation / R32F texture
 * round trip (src/lib.rs:336-682, 905-951; src/renderer.rs; src/terrain_stats.rs).  Not connected to any draw in
 * the reference either. */
typedef struct vf_dem vf_dem;
int vf_dem_create(vf_ctx *ctx, vf_dem **out);
void vf_dem_destroy(vf_dem *d);
/* add_terrain ingest (src/lib.rs:351-388): heights = (f32)src * exaggeration, row-major h rows x w cols */
int vf_dem_set_heights_f32(vf_dem *d, const float *host, uint32_t w, uint32_t h, float exaggeration);
int vf_dem_set_heights_f64(vf_dem *d, const double *host, uint32_t w, uint32_t h, float exaggeration);
/* terrain_stats -> dem_stats_from_slice (src/lib.rs:905-932): out = {min, max, mean, std} */
int vf_dem_stats(vf_dem *d, float out[4]);
/* terrain_stats::min_max(data, clamp=true) (src/terrain_stats.rs:11-35): 1st / 99th percentile, stride-sampled above 65536 */
int vf_dem_percentile_range(vf_dem *d, float *p1, float *p99);
/* normalize_terrain -> normalize_in_place (src/lib.rs:934-951): mode 0 = minmax to [lo, hi], 1 = zscore */
int vf_dem_normalize(vf_dem *d, int mode, float lo, float hi, float eps);
/* upload_height_r32f (src/lib.rs:496-571): heights -> R32F texture (device copy, no row padding) */
int vf_dem_upload_height(vf_dem *d);
int vf_dem_texture_size(const vf_dem *d, uint32_t *w, uint32_t *h);   /* 0 x 0 before the first upload */
/* debug_read_height_patch (src/lib.rs:574-666): texture sub-rectangle -> dst (h rows x w cols) */
int vf_dem_read_patch(vf_dem *d, uint32_t x, uint32_t y, uint32_t w, uint32_t h, float *dst);

/* ---- multi-GPU helper --------------------------------------------------------------------- */
/* De-interleave a rank-major gather buffer [nranks][local_rows][W][4] into the final (H,W,4)
 * image on the device (used after an RCCL all-gather/gather when receiving in place is not
 * possible).  All pointers are device pointers. */
int vf_stitch_bands_device(vf_ctx *ctx, const void *dev_gathered, void *dev_image, uint32_t width,
                           uint32_t height, uint32_t nranks, uint32_t band_h, void *stream);
/* Same for tile shards: gather buffer [nranks][stride_tiles][64][64][4] (stride_tiles >= the largest shard) -> (H,W,4);
 * `layout` = the VF_TILE_LAYOUT word the shards were made with. */
int vf_stitch_tiles_device(vf_ctx *ctx, const void *dev_gathered, void *dev_image, uint32_t width, uint32_t height,
                           uint32_t nranks, uint32_t layout, uint32_t stride_tiles, void *stream);

/* ---- multi-GPU exchange over RCCL (new: the reference creates one device per object,
 * src/terrain/mod.rs:277-294, and has no exchange) -------------------------------------------------------------
 * One process per GPU.  The library resolves RCCL at run time (librccl.so.1: the copy already loaded in the process,
 * e.g. PyTorch's, else the ROCm one); nothing here needs torch.  A communicator is an `ncclComm_t` passed as void*:
 * either one the host already owns (torch.distributed's, a Rust host's own binding) or one made by vf_dist_comm_init.
 *
 *   rank 0:  vf_dist_unique_id(id)            -> hand the 128 bytes to every rank (any host channel)
 *   all:     vf_dist_comm_init(ctx, id, rank, nranks, &comm)
 *   frame:   vf_terrain_set_tile_shard(t, rank, nranks, layout); vf_terrain_set_output_device(t, slab); vf_terrain_render(t, s);
 *            vf_dist_gather_tiles(t, comm, 0, gathered, stride_tiles, s);      (all ranks; `gathered` used on the root)
 *            root: vf_stitch_tiles_device(ctx, gathered, image, W, H, nranks, layout, stride_tiles, s);
 *   or, with column stripes (skew 0) and a tile grid that divides by nranks -- the default of bench.py:
 *            vf_dist_exchange_bands(t, comm, 0, image, s);                     (all ranks; `image` used on the root; no root stitch)
 */
#define VF_DIST_UNIQUE_ID_BYTES 128
int vf_dist_available(void);                                   /* 1 when RCCL could be resolved in this process */
int vf_dist_version(int *version);                             /* ncclGetVersion of the RCCL this process resolved (e.g. 22703) */
int vf_dist_unique_id(uint8_t id[VF_DIST_UNIQUE_ID_BYTES]);    /* ncclGetUniqueId */
int vf_dist_comm_init(vf_ctx *ctx, const uint8_t id[VF_DIST_UNIQUE_ID_BYTES], int rank, int nranks, void **comm);   /* ncclCommInitRank on ctx's device */
void vf_dist_comm_destroy(void *comm);                         /* ncclCommDestroy on the communicator's own device */
/* Every rank of the communicator calls the gather with the SAME root, stride_tiles and shard layout.  What a rank can check
 * locally (handle sharded and rendered, communicator rank/size == the handle's shard, root in range, stride_tiles >= the
 * largest shard of the layout) is checked on every rank before anything is posted, so all ranks fail together.  The root's
 * buffer pointer can only be checked on the root: a root that passes NULL returns VF_ERR_INVALID while the other ranks have
 * queued their sends and wait -- pass a valid buffer.
 * Tile shards -> rank `root`, point to point (every sender on its own xGMI link to the root, no ring): one
 * ncclGroupStart / ncclSend | ncclRecv x (nranks - 1) / ncclGroupEnd on `stream`, ordered after the render queued there.
 * The handle must be tile-sharded (vf_terrain_set_tile_shard); it sends exactly its local tiles (local_tiles * 16384
 * bytes) from its current output buffer.  On the root `dev_gathered` is [nranks][stride_tiles][64][64][4]: rank r's slab lands in
 * slot r; the root's own slab is not moved when it rendered straight into slot `root` (vf_terrain_set_output_device), and
 * goes through RCCL to itself otherwise.  `dev_gathered` is ignored on the other ranks. */
int vf_dist_gather_tiles(vf_terrain *t, void *rccl_comm, int root, void *dev_gathered, uint32_t stride_tiles, void *stream);
/* Band shards (vf_terrain_set_shard) -> rank `root`: every band is a contiguous band_h * W * 4-byte slab of the final image,
 * so the root receives each remote band in place in `dev_image` ((H, W, 4), no stitch pass) and copies its own. */
int vf_dist_gather_bands(vf_terrain *t, void *rccl_comm, int root, void *dev_image, void *stream);
/* Tile shards in column stripes (vf_terrain_set_tile_shard with skew 0, any stripe width) -> the row-major (H, W, 4) frame in `dev_image` on rank
 * `root`, with the stitch sharded like the rendering -- no rank copies the whole frame (round 3 measured +40..107 us on the root
 * of eight for a whole-frame stitch, +11 us per rank this way).  One call per frame on every rank, all queued on `stream`:
 *   1. all-to-all (one ncclGroupStart / ncclSend + ncclRecv per peer / ncclGroupEnd): the frame is cut into nranks horizontal
 *      bands of whole tile rows; a rank's slab holds the tiles of band b contiguously and sends that chunk to rank b;
 *   2. k_stitch_tiles on the rank's band (H / nranks rows; the root stitches straight into `dev_image`);
 *   3. the bands are contiguous slabs of the frame: one more group moves them to the root in place.
 * Needs W, H multiples of 64, tile columns that divide by nranks * stripe width and tile rows that divide by nranks (C4: 2, 4, 8
 * ranks with stripes of 4, 2, 1 tiles); every rank checks that, the
 * communicator (rank / size == the handle's shard) and `root` before anything is posted, so all ranks fail together.  The
 * receive chunks and (off the root) the band live in the handle and are reused by the next call: calls on one handle are
 * ordered on one stream or by the caller's events.  `dev_image` is ignored off the root.
 * There is no reference counterpart (single device: src/terrain/mod.rs:277-294); the module surface it extends is
 * src/lib.rs:961-976. */
int vf_dist_exchange_bands(vf_terrain *t, void *rccl_comm, int root, void *dev_image, void *stream);

/* ---- diagnostics: the fragment stage on its own (BASELINE.json north_star names it) ----------------------------
 * Renders the frame vf_terrain_render drew last (the current uniforms before the first render) once with the visibility store enabled (into scratch buffers: the handle's output, feedback
 * and timing state stay as they were), then runs `repeats` launches of the resolve kernel -- visibility (H, W) u32 ->
 * RGBA8 through fs_main + sRGB store (src/shaders/terrain.wgsl:69-91), the same device code the tile kernel runs on its
 * LDS tile -- and reports the average launch time (HIP events) and the number of covered pixels.  Whole-frame handles only.
 * Algorithmic bytes of a launch: 4 W H (visibility) + 4 W H (RGBA8) + 4 Tw Th (heights). */
typedef struct vf_fragment_timing {
    float resolve_ms;        /* average of `repeats` launches */
    uint32_t covered_pixels; /* pixels with a visible primitive */
    uint32_t repeats;
    uint32_t equal_to_frame; /* 1 when the resolved RGBA8 equals the frame the tile kernel produced, byte for byte */
} vf_fragment_timing;
int vf_terrain_debug_fragment_stage(vf_terrain *t, uint32_t repeats, vf_fragment_timing *out);

#ifdef __cplusplus
}
#endif
#endif /* VF_HIP_H */
