// vf_kernels.h -- the gfx950 kernels of the terrain raster path.
//
//   k_axis_tables     once per (grid, texture size): per-column / per-row vertex-shader terms
//   k_height_blocks   once per height upload: displaced-height cache (9x9 heights per 8x8-cell block) + block min/max
//   per frame, on the handle's side streams (they touch plan state only and run under the previous frame's tile kernel):
//   k_block_boxes     conservative pixel rectangle + capsule of every block, block ranges per (block row, tile column); on a
//                     shard: drops the blocks that reach none of the rank's tiles, lists the 16-block segments still needed
//   k_block_setup     vs_main + the tile-independent part of primitive assembly / culling, once per block: 81 vertex records
//                     {X, Y, 1/w, h} and a block record (exact pixel box, alive masks) per block
//   k_plan, k_plan_sort   busy tiles -> work items weighted by last frame's measured times, heavy tiles cut into column
//                     strips, heaviest first
//   per frame, on the caller's stream:
//   k_clear           background tiles <- clear colour
//   k_tile            persistent workgroups pull work items (a 64x64 screen tile or a strip of one).  Per item: the block rows
//                     that can touch the tile in DESCENDING primitive order, in chunks; waves pull candidate blocks, load
//                     their records, classify the alive primitives against the tile and its final-pixel masks and rasterise
//                     the survivors with the FP32-first exact span solver of vf_raster.h into an LDS visibility tile
//                     (ds_max_u32).  After every block row the covered pixels are final; final-pixel masks cull occluded
//                     blocks, lines and pixels, and a fully final tile stops early.  The fragment stage then runs on the LDS
//                     tile and stores RGBA8 -- no framebuffer-sized intermediate ever touches HBM.  Two instantiations:
//                     the fast one (no clipping code) and the complete one for the rare items that need it.
//   k_resolve         diagnostics: the fragment stage as a launch of its own
//   k_grid_*          grid_generate (bit-exact make_grid)
//   k_triangle        the triangle smoke path
//   k_dem_*           Renderer DEM path: ingest, statistics, normalisation, sampling
//   k_stitch_bands / k_stitch_tiles   multi-GPU de-interleave
//   k_png_filter      render_png read-back: PNG scanline filtering
//
// Painter's order: the reference pipeline has no depth buffer (src/terrain/pipeline.rs:133), so the
// visible fragment is the LAST covering front-facing primitive in index order == max primitive id.
// Primitive ids are cell-row major, so every primitive of block row r+1 beats every primitive of
// block row r: after a block row has been rasterised, covered pixels are final.
//
// Build-time switches: VF_TILE_MIN_WAVES (vf_device.h: the tile kernel's register cap) and two diagnostics builds -- VF_PHASE_PROF
// (per-phase cycle counters, tools/exp_phases.py) and VF_DIAG_ITEM=1 / 2 / 3 (what an item reports in its statistics word: ticks inside
// its block loops / the plan's weight / its start time; tools/exp_toptiles.py, exp_gantt.py).  The experiment knobs of rounds 3-5
// (depth slices, tile priorities, alternative candidate passes, ...) live in tools/experiments/r06_kernel_laboratory.patch: the
// constants below are their measured optima at C4 (EXPERIMENTS.md has the sweeps).
#pragma once
#include "vf_device.h"
#include "vf_raster.h"

namespace vf {

// ---------------------------------------------------------------------------------------------
__global__ void k_axis_tables(uint32_t n, uint32_t tw, uint32_t th, float *xs, float *sinx, float *cosz,
                              int32_t *txi, int32_t *tyj)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float scale = 1.5f;
    const float nm1f = (float)n - 1.0f;
    const float step = (2.0f * scale) / nm1f;          // src/terrain/mod.rs:559-560
    float x = -scale + (float)i * step;                // :566-567 (grid_coord() repeats exactly this)
    float uvc = (float)i / nm1f;                       // :568-569
    xs[i] = x;
    sinx[i] = det_sin(x * 1.3f);                       // terrain.wgsl:40
    cosz[i] = det_cos(x * 1.1f);
    int tx = (int)floorf(uvc * (float)tw), ty = (int)floorf(uvc * (float)th);   // nearest, clamp-to-edge
    txi[i] = min(max(tx, 0), (int)tw - 1);
    tyj[i] = min(max(ty, 0), (int)th - 1);
}

// one wave per block, once per height upload: the block's 9x9 displaced heights -> contiguous cache + exact min/max
__global__ __launch_bounds__(64) void k_height_blocks(uint32_t n, uint32_t nb, uint32_t tw, AxisTables A,
                                                      const float *__restrict__ tex, float *__restrict__ hblk,
                                                      float2 *__restrict__ bounds)
{
    const uint32_t bx = blockIdx.x % nb, by = blockIdx.x / nb;
    const uint32_t i0 = bx * kBlockCells, j0 = by * kBlockCells;
    float lo = INFINITY, hi = -INFINITY;
    bool bad = false;
    for (int v = threadIdx.x; v < kBlockVerts * kBlockVerts; v += 64) {
        uint32_t lj = v / kBlockVerts, li = v - lj * kBlockVerts;
        uint32_t i = i0 + li, j = j0 + lj;
        float h = 0.0f;
        if (i < n && j < n) {
            h = displaced_height(A, tex, tw, i, j);
            bad |= !isfinite(h);
            lo = fminf(lo, h); hi = fmaxf(hi, h);
        }
        hblk[(size_t)blockIdx.x * kBlockStride + v] = h;
    }
    bad = __any(bad);
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if (bad) { lo = -INFINITY; hi = INFINITY; }
    if (threadIdx.x == 0) bounds[blockIdx.x] = make_float2(lo, hi);
}

// ---------------------------------------------------------------------------------------------
// per frame: which pixels can a block touch?  One workgroup per block row, one thread per block.
// The block's vertices all lie in the box [x0,x1] x [hmin,hmax] x [z0,z1]; when its 8 corners are
// regular (finite, w > 0, inside near/far with a margin) their screen bbox bounds every vertex.
// ---------------------------------------------------------------------------------------------
//
// Besides the axis-aligned pixel box the kernel emits a capsule: every vertex (x, h, z) of the block projects onto
// the screen segment between the projections of (x, hmin, z) and (x, hmax, z); those end points lie in the hulls of
// the projected bottom / top faces, so the whole block lies within `rad` of the segment joining the two face centres
// (rad = largest centre-to-corner distance + 1 px).  For oblique views the streak a block sweeps is long and thin
// and the capsule rejects most of the tiles its bounding box crosses.
// Split quantum of a frame: kSplitQuantumX4 quarter-shares of the time the tiles took in the frame whose feedback steers this
// one (sum over its tiles / kTargetItems); 0 = no feedback yet, nothing is split.  One workgroup.
constexpr uint32_t kTargetItems = 1024;                    // work items that keep 256 CUs busy (4 per CU)
__device__ __forceinline__ void publish_quantum(const uint32_t *__restrict__ feedback, uint32_t ntiles, uint32_t *__restrict__ quantum)
{
    __shared__ unsigned long long s_sum;
    if (threadIdx.x == 0) s_sum = 0ull;
    __syncthreads();
    unsigned long long sum = 0;
    for (uint32_t k = threadIdx.x; k < ntiles; k += blockDim.x) sum += feedback[k];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63u) == 0 && sum) atomicAdd(&s_sum, sum);
    __syncthreads();
    if (threadIdx.x == 0) *quantum = s_sum ? (uint32_t)max(1ull, (unsigned long long)kSplitQuantumX4 * s_sum / (4ull * kTargetItems)) : 0u;
}
__global__ __launch_bounds__(512) void k_quantum(const uint32_t *__restrict__ feedback, uint32_t ntiles, uint32_t *__restrict__ quantum)
{
    publish_quantum(feedback, ntiles, quantum);
}

// One workgroup per block row, plus one (index nb) that publishes the frame's split quantum when `feedback` is given.
__global__ __launch_bounds__(512) void k_block_boxes(FrameParams P, const float2 *__restrict__ bounds,
                                                     PixelBox *__restrict__ boxes, PixelBox *__restrict__ row_boxes,
                                                     float4 *__restrict__ cap_seg, float *__restrict__ cap_rad,
                                                     uint32_t *__restrict__ rc_lo, uint32_t *__restrict__ rc_hi,
                                                     const uint32_t *__restrict__ feedback, uint32_t ntiles_all, uint32_t *__restrict__ quantum,
                                                     uint32_t *__restrict__ work_count, BlockRec *__restrict__ recs,
                                                     uint32_t *__restrict__ seg_list, uint32_t *__restrict__ seg_count)
{
    if (blockIdx.x == P.nb) {                               // (uniform per workgroup)
        // this frame's queue words start at zero: [0] work items, [1] split budget used, [2] queue head, [3] items for the complete
        // kernel (k_plan, next on this stream, is their first user; a memset here would be one more dispatch in a latency-bound chain)
        if (threadIdx.x < 4u) work_count[threadIdx.x] = 0u;
        if (feedback) publish_quantum(feedback, ntiles_all, quantum);
        return;
    }
    __shared__ int s_rr[4];   // x0, y0 (min) / x1, y1 (max)
    __shared__ uint32_t s_lo[kMaxTileCols], s_hi[kMaxTileCols];   // this block row's [first, last+1) block per tile column
    __shared__ uint32_t s_seg[(1024 + 15) / 16];                  // per 16-block segment of the row: does any block reach this shard's pixels?
    const uint32_t by = blockIdx.x;
    if (threadIdx.x == 0) { s_rr[0] = 0x7FFF; s_rr[1] = 0x7FFF; s_rr[2] = -1; s_rr[3] = -1; }
    for (uint32_t tc = threadIdx.x; tc < P.ntx; tc += blockDim.x) { s_lo[tc] = 0xFFFFFFFFu; s_hi[tc] = 0u; }
    for (uint32_t k = threadIdx.x; k < (1024u + 15u) / 16u; k += blockDim.x) s_seg[k] = 0u;
    __syncthreads();
    for (uint32_t bx = threadIdx.x; bx < P.nb; bx += blockDim.x) {
        const uint32_t b = by * P.nb + bx;
        const float2 hb = bounds[b];
        const uint32_t i0 = bx * kBlockCells, j0 = by * kBlockCells;
        const uint32_t i1 = min(i0 + kBlockCells, P.n - 1), j1 = min(j0 + kBlockCells, P.n - 1);
        const float x0 = grid_coord(P, i0), x1 = grid_coord(P, i1), z0 = grid_coord(P, j0), z1 = grid_coord(P, j1);   // = AxisTables::xs, bit for bit
        float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
        float cxs[8], cys[8];
        int regular = 0, out_near = 0, out_far = 0;
        const bool hfinite = isfinite(hb.x) && isfinite(hb.y);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            cxs[c] = 0.0f; cys[c] = 0.0f;
            float x = (c & 1) ? x1 : x0, z = (c & 2) ? z1 : z0, h = (c & 4) ? hb.y : hb.x;
            float vp[4], cp[4];
            mat_vec(P.view, x * P.spacing, h * P.exag, z * P.spacing, 1.0f, vp);
            mat_vec(P.proj, vp[0], vp[1], vp[2], vp[3], cp);
            bool fin = finite4(cp[0], cp[1], cp[2], cp[3]);
            // margins keep the classification conservative against rounding at the clip planes
            const float margin = 1e-3f * fmaxf(1.0f, fabsf(cp[3]));
            out_near += fin && cp[2] < -margin;
            out_far += fin && cp[2] - cp[3] > margin;
            if (fin && cp[3] > 0.0f && cp[2] >= margin && cp[3] - cp[2] >= margin) {
                float rw = 1.0f / cp[3];
                float xf = fmaf(cp[0] * rw, P.hw, P.hw), yf = fmaf(-(cp[1] * rw), P.hh, P.hh);
                if (isfinite(xf) && isfinite(yf)) {
                    ++regular;
                    cxs[c] = xf; cys[c] = yf;
                    xmin = fminf(xmin, xf); xmax = fmaxf(xmax, xf);
                    ymin = fminf(ymin, yf); ymax = fmaxf(ymax, yf);
                }
            }
        }
        PixelBox r;
        const int16_t W1 = (int16_t)(P.W - 1), H1 = (int16_t)(P.H - 1);
        if (hfinite && (out_near == 8 || out_far == 8)) {
            r.x0 = 1; r.y0 = 1; r.x1 = 0; r.y1 = 0;            // clip z is affine in position: the whole block is clipped away
        } else if (regular == 8) {
            // one pixel of slack covers the rounding difference between corner and vertex arithmetic
            if (xmax < -1.0f || ymax < -1.0f || xmin > (float)P.W + 1.0f || ymin > (float)P.H + 1.0f) {
                r.x0 = 1; r.y0 = 1; r.x1 = 0; r.y1 = 0;
            } else {
                r.x0 = (int16_t)fmaxf(floorf(xmin) - 1.0f, 0.0f); r.x1 = (int16_t)fminf(ceilf(xmax) + 1.0f, (float)W1);
                r.y0 = (int16_t)fmaxf(floorf(ymin) - 1.0f, 0.0f); r.y1 = (int16_t)fminf(ceilf(ymax) + 1.0f, (float)H1);
            }
        } else {
            r.x0 = 0; r.y0 = 0; r.x1 = W1; r.y1 = H1;         // no bound available: the whole target
        }
        // A shard draws only its own tiles (or row bands): a block whose box meets none of them is dropped here, before the set-up
        // pass, the block ranges and the tile kernel ever see it -- geometry work is sharded with the pixels.
        if (P.nranks > 1u && r.x0 <= r.x1) {
            bool mine = false;
            if (P.shard_tiles) {
                const uint32_t tx0 = ((uint32_t)r.x0 / kTileW) >> P.stripe_shift, tx1 = ((uint32_t)r.x1 / kTileW) >> P.stripe_shift;     // stripes (groups of tile columns)
                const uint32_t ty0 = (uint32_t)r.y0 / kTileH, ty1 = (uint32_t)r.y1 / kTileH;
                if (P.stripe_owner) {                                       // a registered stripe map: an owner per stripe (column stripes: rows do not matter)
                    for (uint32_t sc = tx0; sc <= tx1 && !mine; ++sc) mine = P.stripe_owner[sc] == P.rank;
                }
                else if (tx1 - tx0 + 1u >= P.nranks) mine = true;        // a full period of stripes: every rank owns one in each row
                for (uint32_t ty = ty0; ty <= ty1 && !mine && !P.stripe_owner; ++ty) {
                    // owner(tx, ty) = ((tx >> stripe_shift) + skew ty) % nranks: the first stripe >= tx0 this rank owns in row ty
                    const uint32_t want = (P.rank + P.nranks - (P.skew * ty) % P.nranks) % P.nranks;
                    const uint32_t first = tx0 + (want + P.nranks - tx0 % P.nranks) % P.nranks;
                    mine = first <= tx1;
                }
            } else {
                for (uint32_t bd = (uint32_t)r.y0 >> P.band_shift; bd <= ((uint32_t)r.y1 >> P.band_shift) && !mine; ++bd) mine = bd % P.nranks == P.rank;
            }
            if (!mine) { r.x0 = 1; r.y0 = 1; r.x1 = 0; r.y1 = 0; }
        }
        boxes[b] = r;
        if (r.x0 <= r.x1) s_seg[bx / 16u] = 1u;            // the set-up pass works through the segments that hold such a block ...
        else {                                             // ... and never sees the others: their records say "nothing to draw" from here
            BlockRec e;
            e.box = PixelBox{ 1, 1, 0, 0 }; e.flags = 0u; e.count = 0u; e.alive_even = 0ull; e.alive_odd = 0ull;
            recs[b] = e;
        }
        float4 seg = make_float4(0.f, 0.f, 0.f, 0.f);
        float rad = INFINITY;                                   // no capsule bound unless all 8 corners are regular
        if (regular == 8) {
            // corners 0..3 have h = hmin (bottom face), 4..7 h = hmax (top face)
            const float ax = 0.25f * ((cxs[0] + cxs[1]) + (cxs[2] + cxs[3])), ay = 0.25f * ((cys[0] + cys[1]) + (cys[2] + cys[3]));
            const float bx2 = 0.25f * ((cxs[4] + cxs[5]) + (cxs[6] + cxs[7])), by2 = 0.25f * ((cys[4] + cys[5]) + (cys[6] + cys[7]));
            float r2 = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                r2 = fmaxf(r2, (cxs[c] - ax) * (cxs[c] - ax) + (cys[c] - ay) * (cys[c] - ay));
                r2 = fmaxf(r2, (cxs[4 + c] - bx2) * (cxs[4 + c] - bx2) + (cys[4 + c] - by2) * (cys[4 + c] - by2));
            }
            seg = make_float4(ax, ay, bx2, by2);
            rad = sqrtf(r2) * 1.0001f + 1.5f;                  // + rounding slack between corner and vertex arithmetic
        }
        cap_seg[b] = seg; cap_rad[b] = rad;
        if (r.x0 <= r.x1) {
            // which blocks of this row can reach tile column tc?  [rc_lo, rc_hi) lets the tile kernel test a handful of
            // blocks per row instead of all nb.  One workgroup owns the whole block row, so the ranges are reduced in LDS
            // (global atomics on these 32 K words cost more than the rest of the kernel).
            for (uint32_t tc = (uint32_t)r.x0 / kTileW; tc <= (uint32_t)r.x1 / kTileW; ++tc) {
                atomicMin(&s_lo[tc], bx);
                atomicMax(&s_hi[tc], bx + 1u);
            }
            atomicMin(&s_rr[0], (int)r.x0); atomicMin(&s_rr[1], (int)r.y0);
            atomicMax(&s_rr[2], (int)r.x1); atomicMax(&s_rr[3], (int)r.y1);
        }
    }
    __syncthreads();
    for (uint32_t tc = threadIdx.x; tc < P.ntx; tc += blockDim.x) { rc_lo[tc * P.nb + by] = s_lo[tc]; rc_hi[tc * P.nb + by] = s_hi[tc]; }   // [tile column][block row]: a tile reads its column's rows contiguously
    for (uint32_t k = threadIdx.x; k < (P.nb + 15u) / 16u; k += blockDim.x)
        if (s_seg[k]) {                                    // (order is irrelevant; the counter is zeroed by the frame's k_clear)
            // clamped: a frame that failed between this kernel and its k_clear leaves the counter where it was, and the next use of
            // this plan state must not append past the list (the counter itself sits right behind it)
            const uint32_t at = atomicAdd(seg_count, 1u);
            if (at < P.nb * ((P.nb + 15u) / 16u)) seg_list[at] = by | (k << 16);
        }
    if (threadIdx.x == 0) {
        PixelBox rr;
        if (s_rr[2] < 0) { rr.x0 = 1; rr.y0 = 1; rr.x1 = 0; rr.y1 = 0; }
        else { rr.x0 = (int16_t)s_rr[0]; rr.y0 = (int16_t)s_rr[1]; rr.x1 = (int16_t)s_rr[2]; rr.y1 = (int16_t)s_rr[3]; }
        row_boxes[by] = rr;
    }
}

// ---------------------------------------------------------------------------------------------
// per frame: vs_main (src/shaders/terrain.wgsl:44-66) for every vertex of every block that can reach the target, once, and the
// tile-independent part of primitive assembly / culling (src/terrain/pipeline.rs:124-132: front = CCW, cull back; clip to
// 0 <= z <= w).  One 256-thread workgroup per segment of 16 blocks of a block row:
//   1. the segment's 9 x 129 vertices, each once (neighbouring blocks share their edge columns): displaced height from the cache
//      -> clip coordinates -> 24.8 snapped X, Y, 1/w and clip flags, kept in LDS;
//   2. one wave per block, lane = cell: both primitives are classified -- dead (non-finite, outside near/far, not projectable,
//      bounding box without a pixel centre of the target, back-facing or degenerate), generic (needs clipping / oversized) or alive;
//      alive masks by ballot, exact union of the alive primitives' pixel boxes by wave reduction -> BlockRec;
//   3. the block's 81 vertices as {X, Y, 1/w, h} records (1296 B, coalesced) for the tile kernel's raster and fragment stages.
// Streaming: 324 B read, ~1.3 KB written per block.  The tile kernel then loads instead of recomputing -- each block is looked at by
// 1.9 tiles on average, by up to 16 strips of a heavy tile on a multi-GPU rank -- and never touches a block without alive primitives.
// It needs at most 64 vector registers and 21 KB of LDS: workgroups of the NEXT frame's set-up (side stream) fit on a CU beside the
// tile kernel's workgroup and run in the issue slots it leaves idle.
// ---------------------------------------------------------------------------------------------
constexpr int kSegBlocks = 16;                             // blocks of one block row per workgroup
constexpr int kSegCols = kSegBlocks * kBlockCells + 1;     // 129 vertex columns
constexpr int kSegStride = 136;                            // LDS row pitch (8-byte words): rows 16 banks apart, conflict-free cell reads
constexpr int kSetupThreads = 256;
struct VertexRec { int32_t X, Y; float rw, h; };           // 24.8 snapped position, 1/w, displaced height (the `height` varying)
static_assert(sizeof(VertexRec) == 16, "VertexRec is moved as one 16-byte word");

typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b)
{
    const short2v r = __builtin_elementwise_min(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b));
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b)
{
    const short2v r = __builtin_elementwise_max(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b));
    return __builtin_bit_cast(uint32_t, r);
}

// wave-wide minimum / maximum of packed (x, y) int16 pairs by DPP (no LDS round trips): quad, half row, row, then across the rows;
// the result is valid in lane 63
template <bool MAX>
__device__ __forceinline__ uint32_t wave_reduce_pk_i16(uint32_t v)
{
    auto op = [](uint32_t a, uint32_t b) { return MAX ? pk_max_i16(a, b) : pk_min_i16(a, b); };
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));    // row_half_mirror
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));    // row_mirror
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false));    // row_bcast:15 into rows 1 and 3
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false));    // row_bcast:31 into rows 2 and 3
    return v;
}

// inclusive prefix sum across the wave by DPP (no LDS round trips): within the rows of 16, then row totals into the rows behind
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);     // row_shr:1 (lanes without a source add 0)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);     // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);     // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);     // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);     // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);     // row_bcast:31 into rows 2 and 3
    return v;
}

// inclusive running maximum across the wave (values >= 0), the same DPP ladder
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v)
{
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
    return v;
}

__global__ __launch_bounds__(kSetupThreads) void k_block_setup(FrameParams P, const float *__restrict__ hblk,
                                                               const PixelBox *__restrict__ cons_boxes, VertexRec *__restrict__ vtx,
                                                               BlockRec *__restrict__ recs, ulonglong2 *__restrict__ gen,
                                                               const uint32_t *__restrict__ seg_list, const uint32_t *__restrict__ seg_count)
{
    __shared__ int2 sXY[kBlockVerts][kSegStride];
    __shared__ uint8_t sF[kBlockVerts][kSegStride];
    __shared__ uint32_t s_flagged, s_need;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    // the segments k_block_boxes listed: those with a block that can reach this shard's part of the target
    const uint32_t nseg = min(*seg_count, P.nb * ((P.nb + (uint32_t)kSegBlocks - 1u) / (uint32_t)kSegBlocks));
    for (uint32_t item = blockIdx.x; item < nseg; item += gridDim.x) {
    const uint32_t code = seg_list[item];
    const uint32_t by = code & 0xFFFFu, bx0 = (code >> 16) * kSegBlocks;
    const uint32_t nblk = min((uint32_t)kSegBlocks, P.nb - bx0);
    const uint32_t b0 = by * P.nb + bx0;
    __syncthreads();                                       // the previous segment's LDS reads are done
    if (wave == 0) {                                       // which blocks of the segment have anything to draw (k_block_boxes' conservative box)?
        bool need = false;
        if (lane < nblk) { const PixelBox cb = cons_boxes[b0 + lane]; need = cb.x0 <= cb.x1; }
        const unsigned long long m = __ballot(need);
        if (lane == 0) { s_need = (uint32_t)m; s_flagged = 0u; }
    }
    __syncthreads();
    const uint32_t need = s_need;
    // ---- 1. vertex stage: 9 rows x (8 nblk + 1) columns, every vertex once; its record {X, Y, 1/w, h} goes straight to the block(s)
    //         it belongs to (an edge column to two), 16 bytes per lane -- the tile kernel's raster and fragment stages read these ----
    const uint32_t i0 = bx0 * kBlockCells, j0 = by * kBlockCells, ncols = nblk * kBlockCells + 1u;
    bool flagged = false;
    for (uint32_t v = tid; v < (uint32_t)(kBlockVerts * kSegCols); v += kSetupThreads) {
        const uint32_t r = v / (uint32_t)kSegCols, c = v - r * (uint32_t)kSegCols;
        if (c >= ncols) continue;
        const uint32_t i = i0 + c, j = j0 + r;
        const uint32_t k = min(c >> 3, nblk - 1u), li = c - 8u * k;    // the cache is per block too: column 8 k' is also column 8 of block k' - 1
        VertexRec o;
        o.X = 0; o.Y = 0; o.rw = 0.0f; o.h = 0.0f;
        uint32_t fl = F_BAD;
        if (i < P.n && j < P.n) {
            o.h = hblk[(size_t)(b0 + k) * kBlockStride + r * kBlockVerts + li];
            const ClipVert cv = vertex_shader(P, grid_coord(P, i), grid_coord(P, j), o.h);
            fl = vertex_flags(cv);
            if (!(fl & F_BAD) && !snap_vertex(cv.x, cv.y, cv.w, P.hw, P.hh, o.X, o.Y, o.rw)) fl |= F_NOSNAP;
            flagged |= fl != 0u;
        }
        sXY[r][c] = make_int2(o.X, o.Y); sF[r][c] = (uint8_t)fl;
        if ((need >> k) & 1u) vtx[(size_t)(b0 + k) * kBlockStride + r * kBlockVerts + li] = o;
        if (li == 0u && k > 0u && ((need >> (k - 1u)) & 1u)) vtx[(size_t)(b0 + k - 1u) * kBlockStride + r * kBlockVerts + 8u] = o;
    }
    if (__any(flagged) && lane == 0) s_flagged = 1u;
    __syncthreads();
    const bool any_flag = s_flagged != 0u;                 // (uniform) false for every ordinary view: the flag tests drop out
    // ---- 2. one wave per block, lane = cell ----
    for (uint32_t k = wave; k < nblk; k += kSetupThreads / 64) {
        const uint32_t b = b0 + k, cbase = k * kBlockCells;
        BlockRec rec;
        rec.box = PixelBox{ 1, 1, 0, 0 }; rec.flags = 0u; rec.count = 0u; rec.alive_even = 0ull; rec.alive_odd = 0ull;
        if (!((need >> k) & 1u)) {                         // (uniform) clipped away or off the target: nothing to draw
            if (lane == 0) recs[b] = rec;
            continue;
        }
        const uint32_t lj = lane >> 3, li = lane & 7u;
        int c0 = 0, c1 = 0;                                // 0 dead, 1 alive, 2 generic
        uint32_t lo = 0x7FFF7FFFu, hi = 0x80008000u;       // packed (x, y) int16: running min of (px0, py0), max of (px1, py1)
        if (i0 + cbase + li < P.nm1 && j0 + lj < P.nm1) {
            const int2 pa = sXY[lj][cbase + li], pb = sXY[lj][cbase + li + 1u], pc = sXY[lj + 1u][cbase + li], pd = sXY[lj + 1u][cbase + li + 1u];
            uint32_t fa = 0, fb = 0, fc = 0, fd = 0;
            if (any_flag) { fa = sF[lj][cbase + li]; fb = sF[lj][cbase + li + 1u]; fc = sF[lj + 1u][cbase + li]; fd = sF[lj + 1u][cbase + li + 1u]; }
            // Straight-line on purpose: nearly every primitive reaches the facing test, so early exits would only add divergent
            // branches (scalar instructions cost twice a vector one here); the flag tests are skipped as a whole in ordinary views.
            auto classify = [&](uint32_t f0, uint32_t f1, uint32_t f2, int2 q0, int2 q1, int2 q2) -> int {
                int forced = -1;                                       // a verdict from the clip flags, if any
                if (any_flag) {
                    const uint32_t any = f0 | f1 | f2, all = f0 & f1 & f2;
                    if (any & F_BAD) forced = 0;                      // non-finite clip coordinate: primitive dropped
                    else if (all & (F_NEAR | F_FAR)) forced = 0;      // entirely outside the near or the far plane
                    else if (any & (F_NEAR | F_FAR)) forced = 2;      // needs clipping
                    else if (any & F_NOSNAP) forced = 0;              // a vertex could not be projected (w <= 0)
                }
                const int32_t xmin = min(q0.x, min(q1.x, q2.x)), xmax = max(q0.x, max(q1.x, q2.x));
                const int32_t ymin = min(q0.y, min(q1.y, q2.y)), ymax = max(q0.y, max(q1.y, q2.y));
                const bool oversized = (uint32_t)xmax - (uint32_t)xmin >= (uint32_t)kFastExtent || (uint32_t)ymax - (uint32_t)ymin >= (uint32_t)kFastExtent;
                const int32_t px0 = max((xmin + 127) >> 8, 0), px1 = min((xmax - 128) >> 8, (int32_t)P.W - 1);
                const int32_t py0 = max((ymin + 127) >> 8, 0), py1 = min((ymax - 128) >> 8, (int32_t)P.H - 1);
                const bool holds_centre = px0 <= px1 && py0 <= py1;    // a pixel centre of the target inside the bounding box
                // facing: extents < 2^24 (else oversized), so both products are exact in FP64 and so is their difference (an int64
                // multiply-add costs four times the issue cycles on this chip)
                const double area2 = fma((double)(q1.x - q0.x), (double)(q2.y - q0.y), -((double)(q1.y - q0.y) * (double)(q2.x - q0.x)));
                const bool alive = forced < 0 && !oversized && holds_centre && area2 < 0.0;   // front-facing, not degenerate
                const uint32_t plo = (uint32_t)px0 | ((uint32_t)py0 << 16), phi = (uint32_t)px1 | ((uint32_t)py1 << 16);
                lo = alive ? pk_min_i16(lo, plo) : lo;
                hi = alive ? pk_max_i16(hi, phi) : hi;
                return forced >= 0 ? forced : (oversized ? 2 : (alive ? 1 : 0));
            };
            c0 = classify(fa, fc, fb, pa, pc, pb);        // (a, c, b)
            c1 = classify(fb, fc, fd, pb, pc, pd);        // (b, c, d)
        }
        const unsigned long long a0 = __ballot(c0 == 1), a1 = __ballot(c1 == 1), g0 = __ballot(c0 == 2), g1 = __ballot(c1 == 2);
        lo = wave_reduce_pk_i16<false>(lo);                // (valid in lane 63, which writes the record)
        hi = wave_reduce_pk_i16<true>(hi);
        rec.alive_even = a0; rec.alive_odd = a1;
        rec.count = (uint32_t)(__popcll(a0) + __popcll(a1));
        if (g0 | g1) { rec.flags = kRecGeneric; rec.box = cons_boxes[b]; }   // generic primitives: only the conservative bound holds
        else if (rec.count) rec.box = PixelBox{ (int16_t)(lo & 0xFFFFu), (int16_t)(lo >> 16), (int16_t)(hi & 0xFFFFu), (int16_t)(hi >> 16) };
        if (lane == 63u) {
            recs[b] = rec;
            if (g0 | g1) gen[b] = make_ulonglong2(g0, g1);
        }
    }
    }
}

// ---------------------------------------------------------------------------------------------
// tile kernel
// ---------------------------------------------------------------------------------------------
struct TileCtx {
    uint32_t *vis;            // LDS, kTileW*kTileH words, skewed (see vis_index)
    const uint32_t *colfin;   // LDS, [64][2]: per tile column, bit r set = pixel (column, r) is final
    const uint32_t *rowfin;   // LDS, [64][2]: per tile row, bit c set
    const uint32_t *colfin4;  // LDS, [16][2]: the AND of the masks of columns 4 g .. 4 g + 3 (refresh_fin4: rebuilt behind every rescan, may lag, never lies)
    const uint32_t *rowfin4;  // LDS, [16][2]: ... of rows 4 g .. 4 g + 3
    int32_t px_lo, px_hi;     // inclusive pixel rectangle of the tile, clipped to the target
    int32_t py_lo, py_hi;
};


// LDS layout of the visibility tile: rotate each row by its row number so that a walk down a pixel
// column visits all 32 banks (plain row-major would keep a column in one bank: 64-word row stride).
__device__ __forceinline__ uint32_t vis_index(int32_t lx, int32_t ly) { return (uint32_t)ly * kTileW + (uint32_t)((lx + ly) & (kTileW - 1)); }

__device__ __forceinline__ uint64_t load_mask(const uint32_t *m, int32_t k) { return (uint64_t)m[2 * k] | ((uint64_t)m[2 * k + 1] << 32); }
__device__ __forceinline__ uint64_t bit_range(int32_t lo, int32_t hi)   // bits lo..hi inclusive, 0 <= lo <= hi <= 63
{
    return (~0ull >> (63 - (hi - lo))) << lo;
}

// Exact rasterisation of one unclipped, front-facing triangle restricted to the tile: for each line of the SHORTER bbox axis the
// covered span along the longer one, by the span solver of vf_raster.h -- three FP32 crossings with a proven error bound decide
// nearly every line (stage 1: a span that contains the true one, rejected if it holds no open pixel -- five lines in six of the
// sub-pixel-wide slivers a noise terrain is made of; stage 2: no pixel centre within the error of a crossing => that span is exact),
// and the rare line they cannot decide is solved in FP64 on the integer coordinates.  Lines whose candidate pixels are all final
// (owned by a higher block row) are skipped unsolved, and only non-final pixels are touched.
// `sub` / `nsub`: the lines of one triangle are dealt round-robin to nsub cooperating lanes (all of them run the set-up).
#ifdef VF_PHASE_PROF
struct RasterCounts { uint32_t tris, lines, solved, painted, paint_lines, trips, w_iter, w_s1, w_s2, w_paint, w_cls, c_reach, l_iter, nsurv, iters, live, empty, alive, apass, act; };   // w_*: wave-level executions, the others lane-level: their ratio is the lanes a wave keeps busy there
#define VF_RC_ARG , RasterCounts &RC
#define VF_RC(...) __VA_ARGS__
#else
#define VF_RC_ARG
#define VF_RC(...)
#endif
// floor(x) as int32 in one instruction (V_CVT_FLR_I32_F32; the compiler emits v_floor_f32 + v_cvt_i32_f32 for (int)floorf(x))
__device__ __forceinline__ int32_t floor_to_int(float x)
{
    int32_t k;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(k) : "v"(x));
    return k;
}
// `vb` / `vcode`: where the three vertices came from (this wave's LDS copy of the block, local indices v0 | v1 << 8 | v2 << 16): the
// exact solver reloads them from there instead of keeping six more registers alive through the line loop for a rare event.
// GROUPS (round 4): nineteen lines in twenty end in stage 1 -- their span holds no open pixel -- and a wave walks them at
// the pace of its busiest lane.  So the lanes of a triangle first test its lines four at a time: one bound for the spans of four
// adjacent lines (span_group: the same three crossings, taken at the end of the group each edge's slope points away from) against
// the AND of their four final-pixel masks.  A group without an open pixel inside its bound cannot paint and is dropped; the verdicts
// travel between the triangle's lanes by ballot, and only the lines of the groups that are left are dealt to the lanes and solved as
// before.  The pixels painted are the same by construction (a dropped line would have failed its own stage-1 test).
// `first`: the wave's first lane working on this triangle (lanes first .. first + nsub - 1 do, all of them active here).
template <bool GROUPS>
__device__ __forceinline__ void raster_fast(const TileCtx &T, uint32_t word, const int2 *vb, uint32_t vcode, int32_t sub, int32_t nsub, uint32_t first VF_RC_ARG)
{
    VF_RC(RC.tris++;)
    int32_t px0, py0, n_outer, n_inner, o_base, i_base;
    bool cols;
    SpanSetup S;
    {
        const int2 q0 = vb[vcode & 0xFFu], q1 = vb[(vcode >> 8) & 0xFFu], q2 = vb[vcode >> 16];
        const int32_t xmin = min(q0.x, min(q1.x, q2.x)), xmax = max(q0.x, max(q1.x, q2.x));
        const int32_t ymin = min(q0.y, min(q1.y, q2.y)), ymax = max(q0.y, max(q1.y, q2.y));
        px0 = max((xmin + 127) >> 8, T.px_lo); py0 = max((ymin + 127) >> 8, T.py_lo);
        const int32_t px1 = min((xmax - 128) >> 8, T.px_hi), py1 = min((ymax - 128) >> 8, T.py_hi);
        if (px0 > px1 || py0 > py1) return;
        cols = (px1 - px0) <= (py1 - py0);                 // iterate the short axis, solve spans along the long one
        const int32_t U[3] = { cols ? q0.x : q0.y, cols ? q1.x : q1.y, cols ? q2.x : q2.y };     // outer / inner coordinates of the vertices
        const int32_t V[3] = { cols ? q0.y : q0.x, cols ? q1.y : q1.x, cols ? q2.y : q2.x };
        n_outer = cols ? px1 - px0 : py1 - py0; n_inner = cols ? py1 - py0 : px1 - px0;
        span_setup(U, V, !cols, (cols ? px0 : py0) * 256 + 128, (cols ? py0 : px0) * 256 + 128, n_outer, S);
    }
    o_base = cols ? px0 - T.px_lo : py0 - T.py_lo;         // tile-local index of outer line 0
    i_base = cols ? py0 - T.py_lo : px0 - T.px_lo;         // tile-local index of inner offset 0
    const uint32_t *fin = cols ? T.colfin : T.rowfin;
    // (one loop header or the other: the body below is shared)
    uint32_t gmask = 0, gleft = 0;
    int32_t passed = 0, nlines_left = 0;
    const int32_t g_first = o_base >> 2, o_skew = o_base & 3;           // (GROUPS) the tile's four-line group of the box's first line, and that line's place in it
    if constexpr (GROUPS) {
    // ---- stage 0: the groups of four lines, one per lane and trip; every lane of the triangle takes the same trips ----
    // (groups are the tile's own: lines 4 G .. 4 G + 3 of the tile, G from the one that holds the box's first line -- one load of the
    //  four-line mask per group; the first and the last group may reach beyond the box: bits of `gmask` stay relative to the first)
    const int32_t ng = ((o_base + n_outer) >> 2) - g_first + 1;          // at most 17
    const uint32_t *fin4 = cols ? T.colfin4 : T.rowfin4;
    // (round 4 let a triangle with no more than one line per lane skip the test, its groups all counting as open; since a group's test is
    //  one LDS word -- round 5 -- every triangle is tested: C4 -0.4 %)
    const bool test_groups = n_outer + 1 > 0;
    gmask = test_groups ? 0u : (1u << ng) - 1u;
    const uint32_t lanes_mask = nsub >= 32 ? 0xFFFFFFFFu : (1u << nsub) - 1u;
    for (int32_t kb = 0; test_groups && kb < ng; kb += nsub) {
        const int32_t g = kb + sub;
        bool open = false;
        if (g < ng) {
            const int32_t oa = max(4 * g - o_skew, 0), ob = min(4 * g + 3 - o_skew, n_outer);     // the group's lines inside the box
            const uint64_t done4 = load_mask(fin4, g_first + g);
            int32_t glo, ghi;
            span_group(S, oa, ob, n_inner, glo, ghi);
            if (!S.regular) { glo = 0; ghi = n_inner; }
            open = glo <= ghi && (bit_range(i_base + min(glo, n_inner), i_base + max(ghi, 0)) & ~done4) != 0ull;
            VF_RC(RC.lines += (uint32_t)(ob - oa + 1); RC.l_iter++;)
        }
        VF_RC(if ((int)(threadIdx.x & 63u) == __builtin_ctzll(__ballot(1))) RC.w_iter++;)
        const unsigned long long votes = __ballot(open);
        gmask |= ((uint32_t)(votes >> first) & lanes_mask) << kb;     // (groups beyond ng voted no; other triangles' lanes are masked out)
    }
    // ---- the lines of the groups that are left, dealt to the lanes in order ----
    gleft = gmask;                                         // groups not yet passed by this lane; `passed` of them are behind it
    nlines_left = 4 * (int32_t)__popc(gmask);
    }
    // GROUPS: idx runs over the lines of the open groups; otherwise over all lines (idx = o)
    for (int32_t idx = sub; GROUPS ? idx < nlines_left : idx <= n_outer; idx += nsub) {
        int32_t o = idx;
        if constexpr (GROUPS) {
            while (passed < (idx >> 2)) { gleft &= gleft - 1u; ++passed; }
            o = 4 * (int32_t)__builtin_ctz(gleft) + (idx & 3) - o_skew;
            if (o < 0 || o > n_outer) continue;
        }
        const uint64_t done = load_mask(fin, o_base + o);
        if constexpr (!GROUPS) { VF_RC(RC.lines++; RC.l_iter++; if ((int)(threadIdx.x & 63u) == __builtin_ctzll(__ballot(1))) RC.w_iter++;) }
        // ---- stage 1 (straight-line): a span that contains the true one; does it hold an open pixel? ----
        int32_t F[3], lo, hi;
        span_line(S, o, n_inner, F, lo, hi);
        if (!S.regular) { lo = 0; hi = n_inner; }          // (rare) no FP32 form for this triangle: every open line goes to the exact solver
        const uint64_t cand = bit_range(i_base + min(lo, n_inner), i_base + max(hi, 0)) & ~done;   // (garbage when lo > hi: tested first)
        VF_RC(if (~done) { RC.trips++; if ((int)(threadIdx.x & 63u) == __builtin_ctzll(__ballot(1))) RC.w_s1++; })
        if (lo > hi || cand == 0ull) continue;             // nothing this line could still change
        // ---- stage 2: the span is exact unless a pixel centre lies within the FP32 error of a crossing ----
        VF_RC(RC.solved++; if ((int)(threadIdx.x & 63u) == __builtin_ctzll(__ballot(1))) RC.w_s2++;)
        uint64_t bits = cand;
        if (!S.regular || !span_confirm(S, o, n_inner, F)) {
            // (about one line in 10^4) exact: from the integer coordinates, reloaded -- nothing of this is kept or hoisted out of the loop
            asm volatile("" ::: "memory");
            const int2 q0 = vb[vcode & 0xFFu], q1 = vb[(vcode >> 8) & 0xFFu], q2 = vb[vcode >> 16];
            const int32_t U[3] = { cols ? q0.x : q0.y, cols ? q1.x : q1.y, cols ? q2.x : q2.y };
            const int32_t V[3] = { cols ? q0.y : q0.x, cols ? q1.y : q1.x, cols ? q2.y : q2.x };
            span_exact(U, V, !cols, (cols ? px0 : py0) * 256 + 128, (cols ? py0 : px0) * 256 + 128, o, n_inner, lo, hi);
            if (lo > hi) continue;
            bits = bit_range(i_base + lo, i_base + hi) & ~done;
        }
        const int32_t ol = o_base + o;
        VF_RC(RC.painted += (uint32_t)__popcll(bits); RC.paint_lines += bits ? 1u : 0u;)
        // vis_index(ol, k) for a column, vis_index(k, ol) for a row: (cols ? k : ol) * 64 + ((ol + k) & 63) -- as arithmetic: the compiler
        // made the choice a divergent branch inside this innermost loop (ten scalar instructions per painted pixel)
        // (byte offsets: line * 256 + ((ol + k) & 63) * 4)
        static_assert(kTileW == 64, "the paint loop's index arithmetic");
        const uint32_t ol4 = (uint32_t)ol << 2, line4 = cols ? 0u : ol4 << 6;
        while (bits) {
            const uint32_t k = (uint32_t)__builtin_ctzll(bits);
            bits &= bits - 1;
            VF_RC(if ((int)(threadIdx.x & 63u) == __builtin_ctzll(__ballot(1))) RC.w_paint++;)
            const uint32_t off = __umul24(k, cols ? 256u : 0u) + line4 + (((k << 2) + ol4) & 252u);
            atomicMax(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(T.vis) + off), word);
        }
    }
}

// primitive -> clip-space vertices with varyings (used by the clipped path and the fragment stage)
__device__ __forceinline__ GVert load_vertex(const FrameParams &P, const float *__restrict__ hblk, uint32_t i, uint32_t j)
{
    const float x = grid_coord(P, i), z = grid_coord(P, j);
    const ClipVert c = vertex_shader(P, x, z, cached_height(hblk, P.nb, i, j));
    GVert v;
    v.x = c.x; v.y = c.y; v.z = c.z; v.w = c.w;
    v.a[0] = c.h; v.a[1] = x; v.a[2] = z;                  // varyings: height, xz (terrain.wgsl:63-64)
    return v;
}
__device__ __forceinline__ void load_prim(const FrameParams &P, const float *__restrict__ hblk, uint32_t prim,
                                          GVert &v0, GVert &v1, GVert &v2)
{
    // indices [a,c,b, b,c,d] (src/terrain/mod.rs:578-582): even = (a, c, b), odd = (b, c, d)
    const uint32_t cell = prim >> 1, odd = prim & 1u;
    const uint32_t j = cell / P.nm1, i = cell - j * P.nm1;
    v0 = load_vertex(P, hblk, odd ? i + 1 : i, j);
    v1 = load_vertex(P, hblk, i, j + 1);
    v2 = load_vertex(P, hblk, i + 1, odd ? j + 1 : j);
}

// clipped or oversized primitives: clip, fan, and scan each piece's bbox inside the tile with the
// int64 coverage test.  Rare (primitives crossing the near plane, or > 65536 px across).
__device__ __noinline__ void raster_generic(const GVert v[3], float hw, float hh, uint32_t W, uint32_t H, uint32_t *vis,
                                            int32_t px_lo, int32_t px_hi, int32_t py_lo, int32_t py_hi, uint32_t word)
{
    GVert poly[8];
    const int np = clip_primitive(v, poly);
    for (int f = 1; f + 1 < np; ++f) {
        TriSetup S;
        if (!setup_triangle(poly[0], poly[f], poly[f + 1], hw, hh, W, H, S)) continue;
        const int32_t px0 = max(S.px0, px_lo), px1 = min(S.px1, px_hi);
        const int32_t py0 = max(S.py0, py_lo), py1 = min(S.py1, py_hi);
        for (int32_t py = py0; py <= py1; ++py)
            for (int32_t px = px0; px <= px1; ++px) {
                int64_t e[3];
                if (covers(S, px, py, e)) atomicMax(&vis[vis_index(px - px_lo, py - py_lo)], word);
            }
    }
}

// Tile-dependent part of the classification of a primitive k_block_setup found alive (no clipping needed, front-facing, a pixel
// centre of the target inside its bounding box): does it hold a pixel centre of this tile that is still open?
// `lines`: how many lines raster_fast will walk for it (the shorter side of its box inside the tile).
template <bool FOUR>      // FOUR: the occlusion loop reads the four-line masks (the kernel instantiation for wide items; narrow strips keep their lines' own)
__device__ __forceinline__ bool classify_alive(const TileCtx &T, int32_t X0, int32_t Y0, int32_t X1, int32_t Y1, int32_t X2, int32_t Y2, uint32_t &lines VF_RC_ARG)
{
    const int32_t xmin = min(X0, min(X1, X2)), xmax = max(X0, max(X1, X2));
    const int32_t ymin = min(Y0, min(Y1, Y2)), ymax = max(Y0, max(Y1, Y2));
    const int32_t px0 = max((xmin + 127) >> 8, T.px_lo), px1 = min((xmax - 128) >> 8, T.px_hi);
    const int32_t py0 = max((ymin + 127) >> 8, T.py_lo), py1 = min((ymax - 128) >> 8, T.py_hi);
    if (px0 > px1 || py0 > py1) return false;             // no pixel centre of the tile inside the bbox
    // occlusion: every candidate pixel already final
    const bool cols = (px1 - px0) <= (py1 - py0);
    lines = (uint32_t)(min(px1 - px0, py1 - py0) + 1);
    const uint32_t *fin = cols ? T.colfin : T.rowfin;
    const int32_t o0 = cols ? px0 - T.px_lo : py0 - T.py_lo, o1 = cols ? px1 - T.px_lo : py1 - T.py_lo;
    const uint64_t seg = cols ? bit_range(py0 - T.py_lo, py1 - T.py_lo) : bit_range(px0 - T.px_lo, px1 - T.px_lo);
    VF_RC(RC.c_reach++;)
    // four lines per step: the loop is a chain of LDS latencies (load, test, branch), not of arithmetic
    constexpr int kClsLines = 4;
    if constexpr (FOUR) {
    // (not in the instantiation for narrow strips: a sliver one or two pixels wide in a 4-pixel strip would answer for the strip's other
    //  columns as well -- a rank of eight at the default camera +1 %)
    const uint32_t *fin4 = cols ? T.colfin4 : T.rowfin4;
    for (int32_t g = o0 >> 2; g <= (o1 >> 2); ++g) {
        VF_RC(if ((int)(threadIdx.x & 63u) == __builtin_ctzll(__ballot(1))) RC.w_cls++;)
        if (~load_mask(fin4, g) & seg) return true;
    }
    return false;
    } else {
    for (int32_t o = o0; o <= o1; o += kClsLines) {
        VF_RC(if ((int)(threadIdx.x & 63u) == __builtin_ctzll(__ballot(1))) RC.w_cls++;)
        uint64_t all = load_mask(fin, o);
#pragma unroll
        for (int d = 1; d < kClsLines; ++d) all &= load_mask(fin, min(o + d, o1));
        if (~all & seg) return true;
    }
    return false;
    }
}

// ---- fragment stage ---------------------------------------------------------------------------
// LDS: the linear LUT as 257 x {r, g, b, -} (entry 256 repeats entry 255: the fast path reads texel i and i + 1 as two 16-byte words
// without a clamp), 256 sRGB thresholds.  (Kept this small on purpose: the tile kernel's LDS has to stay below half a CU's 160 KB
// or its register budget -- VF_TILE_MIN_WAVES -- is silently dropped.)
constexpr int kLutStride = 4, kLutFloats = 257 * kLutStride;
struct ShadeTables { const float *lut; const float *thresh; };

// fs_main (terrain.wgsl:69-91) + Rgba8UnormSrgb store
__device__ __forceinline__ uint32_t fragment_shader(const FrameParams &P, const ShadeTables &S, const float attr[3])
{
    const float height = attr[0], x = attr[1], z = attr[2];
    float t = 0.5f + height / (2.0f * P.h_range);
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    float c = t * 256.0f - 0.5f;
    float i0f = floorf(c);
    float f = c - i0f;
    int i0 = (int)i0f, i1 = i0 + 1;
    i0 = min(max(i0, 0), 255); i1 = min(max(i1, 0), 255);
    float nx, ny, nz;
    if (P.shade_mode == 0u) {
        float dhdx = 1.3f * det_cos(x * 1.3f) * 0.25f;
        float dhdz = -1.1f * det_sin(z * 1.1f) * 0.25f;
        float d = fmaf(dhdz, dhdz, fmaf(dhdx, dhdx, 1.0f));
        float inv = 1.0f / sqrtf(d);
        nx = -dhdx * inv; ny = inv; nz = -dhdz * inv;
    } else {
        // SPEC_T32: the fragment stage the reference documents but does not implement (ROADMAP.md:421-436, README.md:128,
        // 174-175) -- forward-difference normals from the height texture, Reinhard in linear.  There is no reference code
        // to follow; the choices are listed in DESIGN.md section 4 and the CPU checker restates them in the same order.
        const float third = 1.0f / 3.0f;
        const float uu = fmaf(x, third, 0.5f), vv = fmaf(z, third, 0.5f);
        const float du = 1.0f / (float)(max(P.tw, 2u) - 1u), dv = 1.0f / (float)(max(P.th, 2u) - 1u);
        const int mx = (int)P.tw - 1, my = (int)P.th - 1;
        const int tx0 = min(max((int)floorf(uu * (float)P.tw), 0), mx), tx1 = min(max((int)floorf((uu + du) * (float)P.tw), 0), mx);
        const int ty0 = min(max((int)floorf(vv * (float)P.th), 0), my), ty1 = min(max((int)floorf((vv + dv) * (float)P.th), 0), my);
        const float h0 = P.tex[(size_t)ty0 * P.tw + tx0], hx = P.tex[(size_t)ty0 * P.tw + tx1], hy = P.tex[(size_t)ty1 * P.tw + tx0];
        const float ax = (hx - h0) * P.exag, az = (hy - h0) * P.exag, sp = P.spacing;
        const float vx = -(ax * sp), vy = sp * sp, vz = -(sp * az);
        const float d = fmaf(vz, vz, fmaf(vy, vy, vx * vx));
        const float inv = 1.0f / sqrtf(d);
        nx = vx * inv; ny = vy * inv; nz = vz * inv;
    }
    float ndl = fmaf(nz, P.Lz, fmaf(ny, P.Ly, nx * P.Lx));
    float lambert = fminf(fmaxf(ndl, 0.0f), 1.0f);
    float shade = 0.15f * (1.0f - lambert) + lambert;
    uint32_t out = 0xFF000000u;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float l0 = S.lut[i0 * kLutStride + ch], l1 = S.lut[i1 * kLutStride + ch];
        float lc = fmaf(f, l1 - l0, l0);
        float v = lc * P.exposure * shade;
        if (P.shade_mode != 0u) v = v / (1.0f + v);          // Reinhard (tests/test_tonemap.py:7-8), before the sRGB store
        out |= srgb_encode(v, S.thresh) << (8 * ch);
    }
    return out;
}

// ---- the fast fragment path (vf_terrain_set_shade_precision(VF_PRECISION_FAST), the default) -------------------------------
// The same formulas (terrain.wgsl:69-91 and the interpolation conventions of DESIGN.md section 4) evaluated the way the hardware
// likes them: v_rcp / v_rsq / v_sin / v_cos / v_log / v_exp (1 ulp) instead of IEEE division, 1/sqrt, the Cody-Waite polynomials and the
// threshold search; fused multiply-adds; one division for the three barycentrics and the perspective sum together.  Visibility is not
// touched by any of this; colours stay within 1 LSB of the exact path (BASELINE.json: "RGBA within +-1 LSB"), and nearly always equal:
// every quantity below is continuous in its inputs and carries a relative error of a few 2^-23, against 2^-8 between two bytes.
// (The sRGB byte is round(255 oetf(c)) from the hardware log/exp estimate WITHOUT the exact path's fix-up against the threshold table.)
// Every fused multiply-add below is written out (the file is compiled with -ffp-contract=off): the compiler may not fuse or
// split anything on its own, so all instantiations -- tile kernel variants, strips, the resolve kernels -- give the same bits.
__device__ __forceinline__ uint32_t srgb_encode_fast(float c)
{
    const float cc = __builtin_amdgcn_fmed3f(c, 0.0f, 1.0f);
    const float nl = fmaf(1.055f, __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(cc) * (1.0f / 2.4f)), -0.055f);
    const float est = cc <= 0.0031308f ? 12.92f * cc : nl;
    return (uint32_t)fmaf(est, 255.0f, 0.5f);                          // est in [0, 1]: the conversion truncates, 255.5 -> 255
}
__device__ __forceinline__ uint32_t fragment_shader_fast(const FrameParams &P, const ShadeTables &S, float height, float x, float z)
{
    const float t = __builtin_amdgcn_fmed3f(fmaf(height, P.inv2hr, 0.5f), 0.0f, 1.0f);
    const float c = fmaxf(fmaf(t, 256.0f, -0.5f), 0.0f);              // (below 0 both texels are entry 0: the same colour as c = 0)
    const float i0f = floorf(c);
    const float f = c - i0f;
    const uint32_t i0 = (uint32_t)i0f;                                  // 0..255
    const float4 l0 = *reinterpret_cast<const float4 *>(S.lut + i0 * kLutStride);
    const float4 l1 = *reinterpret_cast<const float4 *>(S.lut + i0 * kLutStride + kLutStride);
    // v_sin_f32 / v_cos_f32 take revolutions
    const float dhdx = 0.325f * __builtin_amdgcn_cosf(x * (1.3f * 0.15915494309189535f));       // 1.3 cos(1.3 x) / 4
    const float dhdz = -0.275f * __builtin_amdgcn_sinf(z * (1.1f * 0.15915494309189535f));      // -1.1 sin(1.1 z) / 4
    const float inv = __builtin_amdgcn_rsqf(fmaf(dhdz, dhdz, fmaf(dhdx, dhdx, 1.0f)));
    // n = (-dhdx, 1, -dhdz) inv;  n . L
    const float ndl = inv * fmaf(-dhdz, P.Lz, fmaf(-dhdx, P.Lx, P.Ly));
    const float lambert = __builtin_amdgcn_fmed3f(ndl, 0.0f, 1.0f);
    const float es = P.exposure * fmaf(lambert, 0.85f, 0.15f);       // 0.15 (1 - l) + l
    const uint32_t r = srgb_encode_fast(fmaf(f, l1.x - l0.x, l0.x) * es);
    const uint32_t g = srgb_encode_fast(fmaf(f, l1.y - l0.y, l0.y) * es);
    const uint32_t b = srgb_encode_fast(fmaf(f, l1.z - l0.z, l0.z) * es);
    return 0xFF000000u | r | (g << 8) | (b << 16);
}

__device__ __noinline__ bool clipped_attributes(const GVert v[3], float hw, float hh, uint32_t W, uint32_t H, int32_t px, int32_t py,
                                                float attr[3])
{
    GVert poly[8];
    const int np = clip_primitive(v, poly);
    bool hit = false;
    for (int f = 1; f + 1 < np; ++f) {      // the last covering piece wins, as in the draw order
        TriSetup T;
        int64_t e[3];
        if (setup_triangle(poly[0], poly[f], poly[f + 1], hw, hh, W, H, T) && covers(T, px, py, e)) { interpolate(T, e, attr); hit = true; }
    }
    return hit;
}

__device__ __forceinline__ bool vertex_plain(const GVert &v) { return finite4(v.x, v.y, v.z, v.w) && !(v.z < 0.0f) && !(v.z > v.w); }

// What the fragment stage needs of the frame's set-up (k_block_setup): snapped vertices, 1/w, displaced heights, block records.
struct SetupView {
    const VertexRec *vtx;       // per block 81 x {X, Y (24.8 fixed point), 1/w, displaced height}
    const float *hblk;          // per block 81 displaced heights (generic path: the vertex stage is run again there)
    const BlockRec *recs;
    const ulonglong2 *gen;      // per block: which primitives need the generic path (valid where kRecGeneric is set)
};

// Varyings + fs_main for the primitive that owns pixel (px, py).  The three vertices come from the set-up arrays -- the vertex
// stage ran once per frame -- so this is loads, three exact edge functions (FP64: operands are integers < 2^25, every product and
// sum stays below 2^53) and the perspective-correct interpolation of (height, x, z).
// fs_main's inputs from the three vertex records of the visible primitive: exact edge weights, perspective-correct varyings.
__device__ __forceinline__ uint32_t shade_from_records(const FrameParams &P, const ShadeTables &S, uint32_t i, uint32_t j, uint32_t odd,
                                                       const VertexRec &r0, const VertexRec &r1, const VertexRec &r2, int32_t px, int32_t py)
{
    const float rw0 = r0.rw, rw1 = r1.rw, rw2 = r2.rw, h0 = r0.h, h1 = r1.h, h2 = r2.h;
    // varyings xz (terrain.wgsl:64): vertex 0 = (i + odd, j), vertex 1 = (i, j + 1), vertex 2 = (i + 1, j + odd)
    const float x0 = grid_coord(P, i + odd), x1 = grid_coord(P, i), x2 = grid_coord(P, i + 1u);
    const float z0 = grid_coord(P, j), z1 = grid_coord(P, j + 1u), z2 = grid_coord(P, j + odd);
    // inside-positive edge weights at the pixel centre (covers() / edge_fn() in int64, here exactly the same values in FP64)
    const double Px = (double)(px * 256 + 128), Py = (double)(py * 256 + 128);
    const double X0 = r0.X, Y0 = r0.Y, X1 = r1.X, Y1 = r1.Y, X2 = r2.X, Y2 = r2.Y;
    const double e0 = -fma(X2 - X1, Py - Y1, -((Y2 - Y1) * (Px - X1)));
    const double e1 = -fma(X0 - X2, Py - Y2, -((Y0 - Y2) * (Px - X2)));
    const double e2 = -fma(X1 - X0, Py - Y0, -((Y1 - Y0) * (Px - X0)));
    const double area2 = fma(X1 - X0, Y2 - Y0, -((Y1 - Y0) * (X2 - X0)));
    // interpolate(): lambda_i = e_i / -area2 in float, perspective q_i = lambda_i / w_i
    const float fA = (float)(-area2);
    const float la0 = (float)e0 / fA, la1 = (float)e1 / fA, la2 = (float)e2 / fA;
    const float q0 = la0 * rw0, q1 = la1 * rw1, q2 = la2 * rw2;
    const float rQ = 1.0f / ((q0 + q1) + q2);
    float attr[3];
    attr[0] = fmaf(q2, h2, fmaf(q1, h1, q0 * h0)) * rQ;
    attr[1] = fmaf(q2, x2, fmaf(q1, x1, q0 * x0)) * rQ;
    attr[2] = fmaf(q2, z2, fmaf(q1, z1, q0 * z0)) * rQ;
    return fragment_shader(P, S, attr);
}

// The fast path's version: the three edge weights from the vertices relative to the pixel centre.  a, b, c = v0, v1, v2 - P are
// integers below 2^24 in magnitude (the fast raster path only takes primitives less than 2^24 across, and P lies inside the bounding
// box), so they convert exactly; e0 = c x b, e1 = a x c, e2 = b x a are differences of two 48-bit products: one product rounded, its
// rounding error recovered exactly by an fma, the other folded into the difference by a second fma -- each weight is good to a few
// 2^-24 of ITSELF, however thin the sliver (a plain FP32 difference of the products would be good to 2^-24 of the PRODUCTS, which
// for a pixel far from a sub-pixel-wide primitive's vertices is more than the whole area).  Barycentrics, perspective weights and
// the division by their sum collapse into one reciprocal: attr = sum(e_i rw_i a_i) / sum(e_i rw_i).
__device__ __forceinline__ uint32_t shade_from_records_fast(const FrameParams &P, const ShadeTables &S, uint32_t i, uint32_t j, uint32_t odd,
                                                            const VertexRec &r0, const VertexRec &r1, const VertexRec &r2, int32_t px, int32_t py)
{
    const int32_t Px = px * 256 + 128, Py = py * 256 + 128;
    const float ax = (float)(r0.X - Px), ay = (float)(r0.Y - Py), bx = (float)(r1.X - Px), by = (float)(r1.Y - Py);
    const float cx = (float)(r2.X - Px), cy = (float)(r2.Y - Py);
    auto cross = [](float ux, float uy, float vx, float vy) -> float {     // ux vy - uy vx
        const float q = uy * vx;
        return fmaf(ux, vy, -q) - fmaf(uy, vx, -q);
    };
    const float q0 = cross(cx, cy, bx, by) * r0.rw, q1 = cross(ax, ay, cx, cy) * r1.rw, q2 = cross(bx, by, ax, ay) * r2.rw;
    const float rQ = __builtin_amdgcn_rcpf((q0 + q1) + q2);
    const float height = fmaf(q2, r2.h, fmaf(q1, r1.h, q0 * r0.h)) * rQ;
    // xz (terrain.wgsl:64): vertex 0 = (i + odd, j), vertex 1 = (i, j + 1), vertex 2 = (i + 1, j + odd), one grid pitch apart
    const float xi = fmaf((float)i, P.step, -1.5f), zj = fmaf((float)j, P.step, -1.5f);
    const float x = fmaf(P.step * rQ, (odd ? q0 : 0.0f) + q2, xi);
    const float z = fmaf(P.step * rQ, q1 + (odd ? q2 : 0.0f), zj);
    return fragment_shader_fast(P, S, height, x, z);
}

// cell / nm1 without a division: cell < 2^26, nm1 < 2^13; the host chose div_m, div_s so that the product's high word shifted is exact
__device__ __forceinline__ uint32_t cell_row(const FrameParams &P, uint32_t cell) { return __umulhi(cell, P.div_m) >> P.div_s; }

// CLIPPED = false: the caller never put a near/far-clipped primitive into the visibility tile (the fast tile kernel), so the
// clipping code -- calls, stack arrays, scratch memory, the vertex shader -- is not compiled in at all.
// FAST: the fast fragment path above (the host picks it for VF_PRECISION_FAST frames in REFERENCE shade mode).
template <bool CLIPPED, bool FAST>
__device__ inline uint32_t shade_pixel(const FrameParams &P, const SetupView &V, const ShadeTables &S, uint32_t prim, int32_t px, int32_t py)
{
    // indices [a,c,b, b,c,d] (src/terrain/mod.rs:578-582): even = (a, c, b), odd = (b, c, d)
    const uint32_t cell = prim >> 1, odd = prim & 1u;
    const uint32_t j = cell_row(P, cell), i = cell - j * P.nm1;
    const uint32_t li = i & 7u, lj = j & 7u;
    const size_t b = (size_t)(j >> 3) * P.nb + (i >> 3);
    if constexpr (CLIPPED) {
        if (V.recs[b].flags & kRecGeneric) {
            const ulonglong2 g = V.gen[b];
            if (((odd ? g.y : g.x) >> (lj * 8u + li)) & 1ull) {
                GVert v[3];                                // only the clipped path keeps the vertices in memory
                load_prim(P, V.hblk, prim, v[0], v[1], v[2]);
                float attr[3] = { 0.f, 0.f, 0.f };
                if (!clipped_attributes(v, P.hw, P.hh, P.W, P.H, px, py, attr)) return P.clear_rgba;   // unreachable when the visibility tile is consistent
                return FAST ? fragment_shader_fast(P, S, attr[0], attr[1], attr[2]) : fragment_shader(P, S, attr);
            }
        }
    }
    const uint32_t va = lj * kBlockVerts + li;
    const uint32_t l0 = odd ? va + 1u : va, l1 = va + kBlockVerts, l2 = odd ? va + kBlockVerts + 1u : va + 1u;
    const size_t base = b * kBlockStride;
    const VertexRec r0 = V.vtx[base + l0], r1 = V.vtx[base + l1], r2 = V.vtx[base + l2];   // three 16-byte loads: all a vertex contributes
    if constexpr (FAST) return shade_from_records_fast(P, S, i, j, odd, r0, r1, r2, px, py);
    else return shade_from_records(P, S, i, j, odd, r0, r1, r2, px, py);
}

// does the block's pixel box touch the tile, and is any pixel of the overlap still open (not final)?
// conservative: does the segment (a, b) come within `rad` of the tile rectangle?  (slab test on the tile grown by rad)
__device__ __forceinline__ bool capsule_hits_rect(const float4 &seg, float rad, int32_t px_lo, int32_t px_hi, int32_t py_lo, int32_t py_hi)
{
    if (!(rad < 1e30f)) return true;
    const float ex0 = (float)px_lo - rad, ex1 = (float)(px_hi + 1) + rad;
    const float ey0 = (float)py_lo - rad, ey1 = (float)(py_hi + 1) + rad;
    const float dx = seg.z - seg.x, dy = seg.w - seg.y;
    float t0 = 0.0f, t1 = 1.0f;
    if (fabsf(dx) < 1e-6f) { if (seg.x < ex0 || seg.x > ex1) return false; }
    else { const float inv = 1.0f / dx; const float a = (ex0 - seg.x) * inv, b = (ex1 - seg.x) * inv; t0 = fmaxf(t0, fminf(a, b)); t1 = fminf(t1, fmaxf(a, b)); }
    if (fabsf(dy) < 1e-6f) { if (seg.y < ey0 || seg.y > ey1) return false; }
    else { const float inv = 1.0f / dy; const float a = (ey0 - seg.y) * inv, b = (ey1 - seg.y) * inv; t0 = fmaxf(t0, fminf(a, b)); t1 = fminf(t1, fmaxf(a, b)); }
    return t0 <= t1 + 1e-4f;
}
__device__ __forceinline__ bool capsule_hits_tile(const float4 &seg, float rad, const TileCtx &T)
{
    return capsule_hits_rect(seg, rad, T.px_lo, T.px_hi, T.py_lo, T.py_hi);
}

template <bool FOUR>
__device__ __forceinline__ bool block_is_candidate(const PixelBox &b, const float4 &cap_seg, float cap_rad, const TileCtx &T)
{
    if (b.x0 > b.x1) return false;
    const int32_t x0 = max((int32_t)b.x0, T.px_lo), x1 = min((int32_t)b.x1, T.px_hi);
    const int32_t y0 = max((int32_t)b.y0, T.py_lo), y1 = min((int32_t)b.y1, T.py_hi);
    if (x0 > x1 || y0 > y1) return false;
    if (!capsule_hits_tile(cap_seg, cap_rad, T)) return false;
    // four lines of the shorter side per step, as in classify_prim: a chain of LDS latencies otherwise
    const bool cols = x1 - x0 <= y1 - y0;
    const uint32_t *fin = cols ? T.colfin : T.rowfin;
    const int32_t o0 = cols ? x0 - T.px_lo : y0 - T.py_lo, o1 = cols ? x1 - T.px_lo : y1 - T.py_lo;
    const uint64_t seg = cols ? bit_range(y0 - T.py_lo, y1 - T.py_lo) : bit_range(x0 - T.px_lo, x1 - T.px_lo);
    if constexpr (FOUR) {
        const uint32_t *fin4 = cols ? T.colfin4 : T.rowfin4;
        for (int32_t g = o0 >> 2; g <= (o1 >> 2); ++g)
            if (~load_mask(fin4, g) & seg) return true;
        return false;
    } else {
        for (int32_t o = o0; o <= o1; o += 4) {
            const uint64_t all4 = load_mask(fin, o) & load_mask(fin, min(o + 1, o1)) & load_mask(fin, min(o + 2, o1)) & load_mask(fin, min(o + 3, o1));
            if (~all4 & seg) return true;
        }
        return false;
    }
}

// FOUR-LINE MASKS (round 5).  Every occlusion test in this kernel asks the same thing of the final-pixel masks: "is there an open
// pixel in this range on ANY of these adjacent lines" -- the block and triangle tests four lines per step, the line groups of the
// raster four lines per group -- and each step was four 64-bit LDS loads, three min() for the range's end, three ANDs.  The AND of
// the masks of lines 4 g .. 4 g + 3 is kept beside them instead, rebuilt by the wave that has just rescanned (32 lanes, four loads
// each): one load per step.  The steps are aligned to multiples of four lines of the tile, not to the box: a step at the box's end
// also sees up to three lines outside it -- their open pixels can only make the answer "open", the conservative side.
__device__ __forceinline__ void refresh_fin4(const uint32_t *colfin, const uint32_t *rowfin, uint32_t *colfin4, uint32_t *rowfin4, uint32_t lane)
{
    if (lane < 32u) {
        const uint32_t g = lane & 15u;
        const uint32_t *src = lane < 16u ? colfin : rowfin;
        uint32_t *dst = lane < 16u ? colfin4 : rowfin4;
        const uint32_t lo = src[8u * g] & src[8u * g + 2u] & src[8u * g + 4u] & src[8u * g + 6u];
        const uint32_t hi = src[8u * g + 1u] & src[8u * g + 3u] & src[8u * g + 5u] & src[8u * g + 7u];
        dst[2u * g] = lo; dst[2u * g + 1u] = hi;
    }
}

// One wave rescans the whole tile: which pixels are owned by a primitive with (id + 1) >= first_id?  Those can never
// change again (only lower ids remain to be drawn).  Row masks by ballot, column masks by OR.  Returns the count.
// Frontier rescan by one wave while the others keep rasterising: like rescan_final over the whole tile, but rows whose
// pixels were all final at an earlier publication are not looked at again (final stays final) -- as a tile fills up, a
// rescan costs less and less.  `row_full` = the row mask of a fully final row (the low `width` bits).
__device__ __forceinline__ uint32_t rescan_open_rows(uint32_t *vis, uint32_t *colfin, uint32_t *rowfin, uint32_t lane, uint32_t first_id,
                                                     uint64_t row_full, int32_t height)
{
    const uint64_t mine = (int32_t)lane < height ? load_mask(rowfin, (int32_t)lane) : row_full;     // lane r looks after row r
    unsigned long long todo = __ballot(mine != row_full);
    uint64_t colbits = 0;
    const bool in_row = (row_full >> lane) & 1ull;
    while (todo) {                                         // four rows per trip: their LDS reads are in flight together
        int32_t ly[4];
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ly[k] = todo ? __builtin_ctzll(todo) : -1;
            todo &= todo - 1;                              // (0 stays 0)
            v[k] = ly[k] >= 0 ? vis[vis_index((int32_t)lane, ly[k])] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (ly[k] < 0) break;                          // uniform
            const bool fin = in_row && v[k] >= first_id;
            const unsigned long long rm = __ballot(fin);
            if (lane == 0) { rowfin[2 * ly[k]] = (uint32_t)rm; rowfin[2 * ly[k] + 1] = (uint32_t)(rm >> 32); }
            colbits |= (uint64_t)(fin ? 1u : 0u) << ly[k];
        }
    }
    if ((uint32_t)colbits) atomicOr(&colfin[2 * lane], (uint32_t)colbits);
    if ((uint32_t)(colbits >> 32)) atomicOr(&colfin[2 * lane + 1], (uint32_t)(colbits >> 32));
    __builtin_amdgcn_wave_barrier();
    uint32_t nfinal = (int32_t)lane < height ? (uint32_t)__popcll(load_mask(rowfin, (int32_t)lane)) : 0u;   // rows as they stand now
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nfinal += __shfl_xor(nfinal, o);
    return nfinal;
}

// The same for a narrow column strip (a heavy tile cut into 4, 8 or 16 strips is 16, 8 or 4 pixels wide): lane = ROW, one trip per four
// columns instead of one per four open rows -- a 4-pixel strip is rescanned with one trip of LDS reads instead of up to sixteen, and on
// a multi-GPU rank most heavy items are such strips (completion + rescans were a fifth of a rank's wave time).  Column masks by
// ballot, row masks per lane.  Final stays final (the first id only falls), so the masks are simply rebuilt.
__device__ __forceinline__ uint32_t rescan_strip(uint32_t *vis, uint32_t *colfin, uint32_t *rowfin, uint32_t lane, uint32_t first_id,
                                                 int32_t width, int32_t height)
{
    const bool row_ok = (int32_t)lane < height;
    uint32_t rowbits = 0;                                  // (width <= 16: the row mask fits the low word)
    for (int32_t c0 = 0; c0 < width; c0 += 4) {            // uniform
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (row_ok && c0 + k < width) ? vis[vis_index(c0 + k, (int32_t)lane)] : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (c0 + k >= width) break;                    // uniform
            const bool fin = row_ok && v[k] >= first_id;
            const unsigned long long cm = __ballot(fin);
            if (lane == 0) { colfin[2 * (c0 + k)] = (uint32_t)cm; colfin[2 * (c0 + k) + 1] = (uint32_t)(cm >> 32); }
            rowbits |= (fin ? 1u : 0u) << (c0 + k);
        }
    }
    if (row_ok) { rowfin[2 * lane] = rowbits; rowfin[2 * lane + 1] = 0u; }
    uint32_t nfinal = (uint32_t)__popc(rowbits);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nfinal += __shfl_xor(nfinal, o);
    return nfinal;
}

__device__ __forceinline__ uint32_t rescan_final(uint32_t *vis, uint32_t *colfin, uint32_t *rowfin, uint32_t lane, uint32_t first_id,
                                                 int32_t row_begin, int32_t row_end)
{
    uint32_t nfinal = 0;
    uint64_t colbits = 0;
    for (int32_t ly = row_begin; ly < row_end; ++ly) {
        const bool fin = lane < (uint32_t)kTileW && vis[vis_index((int32_t)lane, ly)] >= first_id;
        const unsigned long long rm = __ballot(fin);
        if (lane == 0) { rowfin[2 * ly] = (uint32_t)rm; rowfin[2 * ly + 1] = (uint32_t)(rm >> 32); }
        colbits |= (uint64_t)(fin ? 1u : 0u) << ly;
        nfinal += (uint32_t)__popcll(rm);
    }
    if (lane < (uint32_t)kTileW) {
        if ((uint32_t)colbits) atomicOr(&colfin[2 * lane], (uint32_t)colbits);
        if ((uint32_t)(colbits >> 32)) atomicOr(&colfin[2 * lane + 1], (uint32_t)(colbits >> 32));
    }
    return nfinal;
}

// ---------------------------------------------------------------------------------------------
// Frame plan: one workgroup per owned tile decides whether any block row can touch it.  Background tiles (the
// majority with the reference's default camera) are cleared right here with full-width stores; busy tiles are
// appended to a work list with a weight (number of block rows in reach) and k_plan_sort orders the list
// heaviest first, so the long-running tiles of the frame start first and the tail of the launch stays short.
// ---------------------------------------------------------------------------------------------
// local tile -> pixel rectangle, tile column, and where its pixels go: pixel (lx, ly) of the tile is stored at
// out_base + ly * out_stride + lx.  Band shards keep their rows densely packed (row-major, stride W); tile shards keep
// their tiles densely packed (tile-major, 64 x 64 words each), which is what the multi-GPU exchange moves as one slab.
struct TilePlace { uint32_t tx; size_t out_base; uint32_t out_stride; };
__device__ __forceinline__ TilePlace tile_rect(const FrameParams &P, uint32_t tile, int32_t &px_lo, int32_t &px_hi, int32_t &py_lo, int32_t &py_hi)
{
    TilePlace tp;
    if (P.shard_tiles) {
        const uint32_t m = P.tile_map[tile];
        tp.tx = m & 0xFFFFu;
        py_lo = (int32_t)((m >> 16) * kTileH);
        tp.out_base = (size_t)tile * (kTileW * kTileH);
        tp.out_stride = kTileW;
    } else {
        tp.tx = tile % P.ntx;
        const uint32_t lty = tile / P.ntx;
        py_lo = (int32_t)global_row(P, lty * kTileH);                  // band_h is a multiple of kTileH
        tp.out_base = (size_t)lty * kTileH * P.W + (size_t)tp.tx * kTileW;
        tp.out_stride = P.W;
    }
    px_lo = (int32_t)(tp.tx * kTileW); px_hi = min(px_lo + kTileW, (int32_t)P.W) - 1;
    py_hi = min(py_lo + kTileH, (int32_t)P.H) - 1;
    return tp;
}

// work item = tile | part << 20 | log2(parts) << 24 | slice << 27 | log2(slices) << 29: a heavy tile is cut into 2, 4, 8 or 16
// column strips (blocks are only a few pixels wide, so narrow strips share little work), and the heaviest ones also into 2 or 4
// DEPTH SLICES -- consecutive parts of the tile's descending block-row list.  Every piece is a workgroup's item.  The slices of a
// strip meet in the tile's words of the merge buffer (painter's order is a max over primitive ids, so the slices' tiles combine by
// atomic max) and the slice that arrives last runs the fragment stage.  A slice culls against its own final pixels only: what
// the slices in front of it cover it does not see, so slices repeat occluded work the way strips repeat block work -- the two
// cuts together reach 64 pieces at about the repeated work of 16 strips (tools/exp_slices.py).
// Depth slices are not in this file any more (round 6: tools/experiments/r06_kernel_laboratory.patch holds them): on a rank of eight at
// C4 they shorten the longest item (0.21 -> 0.16 ms) and still lengthen the frame (0.225 -> 0.249 ms) -- the ~25 us every item costs
// before its first block (row list, candidate tests, list fill, fragment stage) times the extra items outweighs the shorter critical
// path.  The item word keeps their two fields (always zero).
constexpr uint32_t kSplitBudget = 2048;                    // extra work items a frame may create by splitting
__device__ __forceinline__ uint32_t work_tile(uint32_t code) { return code & 0xFFFFFu; }
__device__ __forceinline__ uint32_t work_part(uint32_t code) { return (code >> 20) & 15u; }
__device__ __forceinline__ uint32_t work_slice(uint32_t code) { return (code >> 27) & 3u; }
__device__ __forceinline__ uint32_t work_lg_slices(uint32_t code) { return (code >> 29) & 3u; }
__device__ __forceinline__ void work_strip(uint32_t code, int32_t &px_lo, int32_t &px_hi)
{
    const uint32_t lg = (code >> 24) & 7u, part = (code >> 20) & 15u;
    const int32_t w = kTileW >> lg;
    const int32_t lo = px_lo + (int32_t)part * w;
    px_hi = min(px_hi, lo + w - 1);
    px_lo = lo;
}

// Weights are FEEDBACK: the time (10 ns ticks, summed over its strips) k_tile spent on the tile in the previous frame.  They
// only steer scheduling -- order and strip splitting -- never the result, so a stale value after a camera jump costs time, not
// correctness.  A handle's FIRST frame has no times yet: k_plan_estimate stands in for them with a static estimate from the
// block ranges (below), so that the one-shot render of the reference's usage (construct, render_png once: src/terrain/mod.rs:410-491)
// is planned -- cut into strips, heaviest first -- like any other frame.

// Block rows that reach tile column `tcol` within pixel rows [py_lo, py_hi]: their number (returned; 0 = background tile) and the
// number of blocks in their [rc_lo, rc_hi) ranges (*est_out) -- the candidates the tile kernel will test, a little more than the
// (tile, block) pairs it will draw.  One workgroup of 256 threads; both results are workgroup-uniform after the call.
__device__ __forceinline__ uint32_t tile_reach(const FrameParams &P, const PixelBox *__restrict__ row_boxes, const uint32_t *__restrict__ rc_lo,
                                               const uint32_t *__restrict__ rc_hi, uint32_t tcol, int32_t py_lo, int32_t py_hi,
                                               uint32_t *s_hits, uint32_t *s_est)
{
    if (threadIdx.x == 0) { *s_hits = 0; *s_est = 0; }
    __syncthreads();
    uint32_t hits = 0, est = 0;
    for (uint32_t r = threadIdx.x; r < P.nb; r += 256) {
        const PixelBox rr = row_boxes[r];
        const uint32_t lo = rc_lo[tcol * P.nb + r], hi = rc_hi[tcol * P.nb + r];
        if (lo < hi && rr.x0 <= rr.x1 && rr.y1 >= py_lo && rr.y0 <= py_hi) { ++hits; est += hi - lo; }
    }
    for (int o = 32; o > 0; o >>= 1) { hits += __shfl_xor(hits, o); est += __shfl_xor(est, o); }
    if ((threadIdx.x & 63u) == 0 && hits) { atomicAdd(s_hits, hits); atomicAdd(s_est, est); }
    __syncthreads();
    return *s_hits;
}
// ticks (10 ns) a tile with `blocks` candidate blocks is expected to cost: 0.25 us per block (C4: 0.2 .. 0.35 us per pair at the
// default camera, more for near blocks) -- only ratios between tiles matter, the split quantum is derived from the same numbers
__device__ __forceinline__ uint32_t static_ticks(uint32_t blocks) { return max(blocks * 25u, 1u); }

// First frame of a handle: a static estimate written where k_plan expects the previous frame's tile times (and the time of the
// tile's only piece); k_quantum then derives the split quantum from them.  What a tile costs is the (tile, block) pairs it draws
// before all its pixels are final, so the estimate walks the tile's block rows the way the tile kernel does -- nearest first, in
// passes of 32 rows, 8 lanes per row -- and counts the blocks whose EXACT pixel box (the set-up pass's block record: on a first
// frame nothing is there to overlap the plan with, so it may wait for the set-up) and capsule meet the tile, until the tile's
// top strip is buried: every one of its eight 8 x 8-pixel cells inside the boxes of kEstDepth blocks.  Boxes are not coverage (a
// noise terrain fills a tenth of a block's box), hence a depth, not a single layer; and the TOP strip because painter's order
// draws the near rows first and they sit lowest on the screen: a tile on the silhouette keeps sky in its top strip and is
// counted to its last row -- those are the frame's heaviest tiles -- while an interior tile stops after the rows that bury it.
__device__ __forceinline__ bool capsule_hits_rect(const float4 &seg, float rad, int32_t px_lo, int32_t px_hi, int32_t py_lo, int32_t py_hi);
__global__ __launch_bounds__(256) void k_plan_estimate(FrameParams P, const PixelBox *__restrict__ row_boxes, const BlockRec *__restrict__ recs,
                                                       const float4 *__restrict__ cap_seg, const float *__restrict__ cap_rad,
                                                       const uint32_t *__restrict__ rc_lo, const uint32_t *__restrict__ rc_hi,
                                                       uint32_t *__restrict__ feedback)
{
    constexpr uint32_t kEstDepth = 24;
    __shared__ uint32_t s_est, s_top[8], s_stop;
    int32_t px_lo, px_hi, py_lo, py_hi;
    const TilePlace tp = tile_rect(P, blockIdx.x, px_lo, px_hi, py_lo, py_hi);
    if (threadIdx.x < 8u) s_top[threadIdx.x] = 0u;
    if (threadIdx.x == 0) { s_est = 0; s_stop = 0; }
    __syncthreads();
    const uint32_t sub = threadIdx.x & 7u, rsub = threadIdx.x >> 3;          // 8 lanes share a row, 32 rows per pass
    uint32_t est = 0;
    for (uint32_t base = 0; base < P.nb; base += 32u) {
        const uint32_t k = base + rsub;
        if (k < P.nb) {
            const uint32_t r = P.nb - 1u - k;                                   // descending rows = descending primitive ids = the tile kernel's order
            const PixelBox rr = row_boxes[r];
            const uint32_t lo = rc_lo[tp.tx * P.nb + r], hi = rc_hi[tp.tx * P.nb + r];
            if (lo < hi && rr.x0 <= rr.x1 && rr.y1 >= py_lo && rr.y0 <= py_hi)
                for (uint32_t bx = lo + sub; bx < hi; bx += 8u) {
                    const uint4 rw = *reinterpret_cast<const uint4 *>(recs + r * P.nb + bx);        // box, flags, count
                    const PixelBox b = PixelBox{ (int16_t)(rw.x & 0xFFFFu), (int16_t)(rw.x >> 16), (int16_t)(rw.y & 0xFFFFu), (int16_t)(rw.y >> 16) };
                    if (!(b.x0 <= b.x1 && b.x1 >= px_lo && b.x0 <= px_hi && b.y1 >= py_lo && b.y0 <= py_hi)) continue;
                    if (!capsule_hits_rect(cap_seg[r * P.nb + bx], cap_rad[r * P.nb + bx], px_lo, px_hi, py_lo, py_hi)) continue;
                    // what the pair costs: one round trip for the block (2 ticks of the workgroup's time) + its alive primitives that
                    // reach the tile -- the share of the block's box that lies inside the tile -- at 1.8 ticks each (C4, measured)
                    const float ov = (float)((min((int32_t)b.x1, px_hi) - max((int32_t)b.x0, px_lo) + 1) * (min((int32_t)b.y1, py_hi) - max((int32_t)b.y0, py_lo) + 1));
                    const float area = (float)(((int32_t)b.x1 - (int32_t)b.x0 + 1) * ((int32_t)b.y1 - (int32_t)b.y0 + 1));
                    est += 32u + (uint32_t)(29.0f * (float)rw.w * ov / area);                          // sixteenths of a tick
                    if ((int32_t)b.y0 <= py_lo && (int32_t)b.y1 >= py_lo + 7)      // spans the top strip: its cells inside the box
                        for (int32_t c = max(((int32_t)b.x0 - px_lo + 7) >> 3, 0); c <= min(((int32_t)b.x1 - px_lo - 7) >> 3, 7); ++c) atomicAdd(&s_top[c], 1u);
                }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t least = s_top[0];
            for (int c = 1; c < 8; ++c) least = min(least, s_top[c]);
            s_stop = least >= kEstDepth ? 1u : 0u;
        }
        __syncthreads();
        if (s_stop) break;                                                      // (uniform)
    }
    for (int o = 32; o > 0; o >>= 1) est += __shfl_xor(est, o);
    if ((threadIdx.x & 63u) == 0 && est) atomicAdd(&s_est, est);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t ticks = max(s_est / 16u, 1u);         // (never 0: a tile without a time falls back to k_plan's looser count)
        feedback[blockIdx.x] = ticks;
        feedback[(size_t)P.ntx * P.nty + 1u + (size_t)blockIdx.x * 64u] = ticks;
    }
}

// `flags_out`: this frame's per-tile word (bit 0 background, bits 8.. the cut).  Not the `background` array itself: when the plan is
// not fresh, `last_flags` IS this plan state's `background` (the frame two back), and other workgroups of this launch still read
// their neighbours' old entries -- k_plan_sort, which runs after every workgroup of this kernel, moves the new words over.
__global__ __launch_bounds__(256) void k_plan(FrameParams P, const PixelBox *__restrict__ row_boxes, uint32_t *__restrict__ flags_out,
                                              uint2 *__restrict__ work, uint32_t *__restrict__ work_count,
                                              const uint32_t *__restrict__ last_blocks, const uint32_t *__restrict__ last_mean,
                                              uint32_t *__restrict__ split_budget, const uint32_t *__restrict__ rc_lo,
                                              const uint32_t *__restrict__ rc_hi, uint32_t moving, const uint32_t *last_flags, MotionMap M)
{
    __shared__ uint32_t s_hits, s_est;
    int32_t px_lo, px_hi, py_lo, py_hi;
    const TilePlace tp = tile_rect(P, blockIdx.x, px_lo, px_hi, py_lo, py_hi);
    const uint32_t total = tile_reach(P, row_boxes, rc_lo, rc_hi, tp.tx, py_lo, py_hi, &s_hits, &s_est);
    if (total) {
        if (threadIdx.x == 0) {
            // A tile's time is the sum over its strips, and strips repeat block work: cut in 16 it reports about twice what it
            // would whole.  Taken at face value that keeps a tile cut that was cut once -- the plan has several fixed points, and
            // which one a handle sits in depends on its history (1920 x 1080, grid 2048: 0.50 ms per frame from a cold start, 0.56
            // for good after five frames of another view).  So an unsharded handle brings the time back to "as one item" first:
            // / (1 + log2(strips) / 4), strips as recorded in the flags word of the frame the time comes from (one fixed point:
            // 0.52 ms whatever came before).  Shards keep the face value: their cold-start fixed point is the better one for a
            // GPU with few tiles (emulated 2 / 4 ranks: 0.74 / 0.52 ms against 0.84 / 0.55), and their camera rarely moves.
            const bool as_one = P.nranks == 1u && !P.shard_tiles;
            auto tile_time = [&](uint32_t idx) -> uint32_t {
                uint32_t t = last_blocks[idx];
                const uint32_t f = last_flags[idx];            // bits 8..11 log2(strips), 12..13 log2(slices) of the frame the time comes from
                const uint32_t lg = (f >> 8) & 15u;
                // A tile cut into strips is as heavy as its HEAVIEST strip makes it (round 5): a silhouette that crosses a corner of the tile
                // leaves one strip with most of the work, and the tile's sum then says "cut in 8" where that strip alone outlasts the frame's
                // even share several times -- at 1920 x 1080 (few tiles, a half-empty GPU) the frame waited for such strips with 47-65 % of
                // the workgroups idle (tools/exp_gantt.py, VF_C5=1).  The strips' own times are in the 64 words behind the tile's.
                const uint32_t lgp = lg + ((f >> 12) & 3u);     // log2 of the pieces the tile was cut into (strips x depth slices): <= 6, 64 words
                if (t && lgp && lgp <= 6u) {
                    const uint32_t *pc = last_blocks + (size_t)P.ntx * P.nty + 1u + (size_t)idx * 64u;
                    uint32_t pk = 0;
                    for (uint32_t p = 0; p < (1u << lgp); ++p) pk = max(pk, pc[p]);
                    t = max(t, (uint32_t)min((unsigned long long)pk << lgp, 0xFFFFFFFFull));
                }
                return as_one ? (uint32_t)(((unsigned long long)t * 4ull) / (4ull + lg + ((f >> 12) & 3u))) : t;
            };
            uint32_t seen = tile_time(blockIdx.x);
            const uint32_t mean = *last_mean;
            bool own_time = true;                          // `seen` is this very tile's time (its pieces' times then apply too)
            if (M.on && as_one) {
                // MOTION-COMPENSATED FEEDBACK (round 5).  The tile times were measured under another camera (the frame before last of a
                // moving camera): where on THAT screen was what this tile shows now?  The host hands over the homography of the ground
                // plane y = 0 between the two screens; the tile's centre goes through it, and the tile takes the heaviest time within one
                // tile of where it lands (the silhouette's tiles show terrain above the plane: they land a little off).  The plan of a
                // moving camera therefore no longer waits for the previous frame's times (vf_hip.hip, render_impl): it runs under the
                // previous frame's tile kernel like a resting camera's.  Scheduling only -- never a pixel.
                const float cx = 0.5f * (float)(px_lo + px_hi + 1), cy = 0.5f * (float)(py_lo + py_hi + 1);
                const float ox = fmaf(M.m[0], cx, fmaf(M.m[1], cy, M.m[2])), oy = fmaf(M.m[3], cx, fmaf(M.m[4], cy, M.m[5])), ow = fmaf(M.m[6], cx, fmaf(M.m[7], cy, M.m[8]));
                seen = 0u; own_time = false;
                if (ow > 1e-6f) {
                    const float fx = ox / ow, fy = oy / ow;
                    if (fx > -64.0f && fy > -64.0f && fx < (float)P.W + 64.0f && fy < (float)P.H + 64.0f) {
                        // (measured on the C5 orbit, tools/exp_c5.py: the heaviest of 3 x 3 tiles 0.302 ms per pose; the heaviest of the 2 x 2 nearest
                        //  0.372, their bilinear mean 0.42 -- under-cutting a heavy tile costs the frame's critical path; 5 x 5 or the
                        //  times scaled up 1.5 .. 3 x: 0.306 .. 0.32; the waiting plan of round 4: 0.321)
                        const int32_t tx = (int32_t)floorf(fx * (1.0f / kTileW)), ty = (int32_t)floorf(fy * (1.0f / kTileH));
                        for (int32_t dy = -1; dy <= 1; ++dy)
                            for (int32_t dx = -1; dx <= 1; ++dx) {
                                const int32_t x = tx + dx, y = ty + dy;
                                if (x >= 0 && y >= 0 && x < (int32_t)P.ntx && y < (int32_t)P.nty) seen = max(seen, tile_time((uint32_t)y * P.ntx + (uint32_t)x));
                            }
                    }
                }
                // The plane is the ground: the silhouette's tiles show terrain ABOVE it -- at the horizon their centres lie beyond the plane's
                // vanishing line and land nowhere, or on the light tiles under the ridge -- and those are the frame's heaviest tiles: left
                // whole at the end of the queue one of them ran 0.5 ms after everything else had finished (C5 orbit, pose 16: 0.63 ms
                // instead of 0.2).  So a tile also takes the heaviest time around its OWN place on the old screen (a ridge moves sideways
                // under an orbiting camera, hardly up or down: two tiles to either side, one up and down) ...
                // (C5 orbit: 0.302 -> 0.292 ms per pose; a floor from this frame's own block ranges -- the static estimate of a first frame --
                //  under the cut or under the queue order: 0.300 .. 0.306, not kept)
                {
                    const int32_t tx = (int32_t)tp.tx, ty = (int32_t)(blockIdx.x / P.ntx);
                    for (int32_t dy = -1; dy <= 1; ++dy)
                        for (int32_t dx = -2; dx <= 2; ++dx) {
                            const int32_t x = tx + dx, y = ty + dy;
                            if (x >= 0 && y >= 0 && x < (int32_t)P.ntx && y < (int32_t)P.nty) seen = max(seen, tile_time((uint32_t)y * P.ntx + (uint32_t)x));
                        }
                }
            }
            else if ((seen == 0u || moving) && mean && P.nranks == 1u && !P.shard_tiles) {
                // The camera moved.  A tile that is busy now but was background in the frame the feedback comes from would sort last
                // and never be split, and what moved in is most likely what a neighbour held (the silhouette's heavy tiles
                // wander): it takes the heaviest tile within two tiles' distance.  With a fast camera (`moving`: the plan runs
                // after the previous frame, vf_hip.hip) every tile takes the heaviest of its 3 x 3 neighbourhood -- cutting a
                // tile too fine costs some repeated block work, leaving a heavy one whole costs the frame's critical path.
                const int32_t tx = (int32_t)tp.tx, ty = (int32_t)(blockIdx.x / P.ntx);
                const int32_t reach = seen ? 1 : 2;
                for (int32_t dy = -reach; dy <= reach; ++dy)
                    for (int32_t dx = -reach; dx <= reach; ++dx) {
                        const int32_t x = tx + dx, y = ty + dy;
                        if (x >= 0 && y >= 0 && x < (int32_t)P.ntx && y < (int32_t)P.nty) seen = max(seen, tile_time((uint32_t)y * P.ntx + (uint32_t)x));
                    }
            }
            const uint32_t weight = seen ? seen : static_ticks(s_est);     // no time from any frame, nor from a neighbour: the blocks in reach
            // strips: 1, 2, 4, 8 or 16.  `mean` holds the split quantum published by k_plan_sort: four times the work one
            // item would carry if last frame's blocks were spread evenly over kTargetItems workgroups -- so a lightly loaded
            // GPU (one rank of a multi-GPU frame) cuts its few heavy tiles finer than a fully loaded one.
            // Beyond 8 strips the next cuts are depth slices (2, then 4), and only then the 16th strip: 8-pixel strips in 2 slices
            // repeat less work than 4-pixel strips, and their items are half as long.
            uint32_t lg = 0, lgs = 0;
            if (mean && px_hi - px_lo + 1 == kTileW) {
                const uint32_t q = seen / mean;
                lg = q >= 16u ? 4u : q >= 8u ? 3u : q >= 4u ? 2u : q >= 2u ? 1u : 0u;
                // the launch holds ntiles + kSplitBudget workgroups: reserve the extra items, fall back to fewer pieces
                while (lg + lgs && atomicAdd(split_budget, (1u << (lg + lgs)) - 1u) + (1u << (lg + lgs)) - 1u > kSplitBudget) {
                    atomicSub(split_budget, (1u << (lg + lgs)) - 1u);
                    if (lgs) --lgs; else --lg;
                }
            }
            // The pieces' weights order the launch (heaviest first).  Cut the way it was in the frame its time comes from, a tile
            // hands every piece the time that piece took then (pieces differ: the slice in front draws more than the one behind,
            // the strip over the ridge more than its neighbour); cut differently, the tile's time is shared evenly.
            const uint32_t cut = lg | (lgs << 4);
            const bool same_cut = own_time && seen == tile_time(blockIdx.x) && seen != 0u && ((last_flags[blockIdx.x] >> 8) & 0x3Fu) == cut;
            const uint32_t *piece_time = last_blocks + (size_t)P.ntx * P.nty + 1u + (size_t)blockIdx.x * 64u;
            flags_out[blockIdx.x] = cut << 8;               // busy; the pieces it is cut into (read back with its time, two frames on)
            const uint32_t parts = 1u << lg, slices = 1u << lgs;
            const uint32_t at = atomicAdd(work_count, parts * slices);
            for (uint32_t p = 0; p < parts; ++p)
                for (uint32_t sl = 0; sl < slices; ++sl)
                    work[at + p * slices + sl] = make_uint2(blockIdx.x | (p << 20) | (lg << 24) | (sl << 27) | (lgs << 29),
                                                            same_cut ? max(piece_time[p * slices + sl], 1u) : weight >> (lg + lgs));
        }
        return;
    }
    // background tile: only flagged here.  The plan kernels touch nothing but plan state, so they may run on a side stream
    // while the previous frame is still being drawn; k_clear, on the frame's own stream, does the clearing.
    if (threadIdx.x == 0) flags_out[blockIdx.x] = 1u;
}

// Background tiles: clear colour (src/terrain/mod.rs:421).  Whole tiles in a 16-byte aligned layout take 16 bytes per lane
// (this pass is pure HBM write bandwidth: 49 MiB of the C4 default frame); edge tiles and odd widths go pixel by pixel.
__global__ __launch_bounds__(256) void k_clear(FrameParams P, const uint32_t *__restrict__ background, uint32_t *__restrict__ rgba,
                                               uint32_t *__restrict__ vis_out, uint32_t *__restrict__ stats, uint32_t nstats,
                                               uint32_t *__restrict__ seg_count)
{
    // the frame's set-up pass has read its segment list (this launch waits for it): empty it for the frame after next, whose
    // k_block_boxes fills it again after this frame is drawn
    if (blockIdx.x == 0 && threadIdx.x == 0) *seg_count = 0u;
    // diagnostics (timing enabled): this frame's per-item statistics start at zero -- spread over the launch, no memset dispatch
    if (stats) for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < nstats; k += gridDim.x * 256u) stats[k] = 0u;
    if ((background[blockIdx.x] & 1u) == 0u) return;
    int32_t px_lo, px_hi, py_lo, py_hi;
    const TilePlace tp = tile_rect(P, blockIdx.x, px_lo, px_hi, py_lo, py_hi);
    const int32_t w = px_hi - px_lo + 1, h = py_hi - py_lo + 1;
    if (w == kTileW && (tp.out_stride & 3u) == 0u && (tp.out_base & 3u) == 0u && (reinterpret_cast<uintptr_t>(rgba) & 15u) == 0u &&
        (!vis_out || (reinterpret_cast<uintptr_t>(vis_out) & 15u) == 0u)) {
        const uint4 c4 = make_uint4(P.clear_rgba, P.clear_rgba, P.clear_rgba, P.clear_rgba), z4 = make_uint4(0u, 0u, 0u, 0u);
        for (int32_t k = threadIdx.x; k < (kTileW / 4) * h; k += 256) {
            const int32_t ly = k / (kTileW / 4), q = k - ly * (kTileW / 4);
            const size_t o = tp.out_base + (size_t)ly * tp.out_stride + 4u * (uint32_t)q;
            *reinterpret_cast<uint4 *>(rgba + o) = c4;
            if (vis_out) *reinterpret_cast<uint4 *>(vis_out + o) = z4;
        }
        return;
    }
    for (int32_t k = threadIdx.x; k < w * h; k += 256) {
        const int32_t ly = k / w, lx = k - ly * w;
        const size_t o = tp.out_base + (size_t)ly * tp.out_stride + (uint32_t)lx;
        rgba[o] = P.clear_rgba;
        if (vis_out) vis_out[o] = 0u;
    }
}

// single workgroup: order the work items by descending weight and clear this plan state's per-tile feedback counters (the plan has
// read them).  The order only steers scheduling -- heaviest first, so that the long items start first and the launch's tail stays
// short -- and needs no more than that: a counting sort on a 10-bit logarithmic key (the weight's exponent and five mantissa bits:
// 2 % steps), four barriers per run of 4096 items instead of the 66 passes of the bitonic sort this replaces (round 4: 25 us alone
// -> 5 us; it is on the critical path of a handle's first frames and of a moving camera, whose plan waits for the previous frame).
// Out of place (work -> sorted), four waves, nothing held in registers across a barrier: the kernel has to find a place BESIDE a
// resident tile-kernel workgroup and the set-up pass's workgroups that come and go -- as sixteen waves that need four free slots on
// every SIMD of one CU at the same moment it waited for most of the frame to be launched at all (570 us "long" in the trace, 5 us
// alone).  Items of one key keep no particular order.  Lists longer than
// 4096 are ordered in independent 4096-item runs, which is all the scheduler needs.
__global__ __launch_bounds__(256) void k_plan_sort(const uint2 *__restrict__ work, uint2 *__restrict__ sorted, const uint32_t *__restrict__ work_count,
                                                   uint32_t *__restrict__ last_blocks, uint32_t ntiles,
                                                   const uint32_t *__restrict__ flags_new, uint32_t *__restrict__ background, uint32_t nlocal)
{
    constexpr uint32_t kKeys = 1024, kT = 256;             // four waves: one per SIMD finds a place at once beside whatever is resident
    __shared__ uint32_t s_hist[kKeys];                     // items per key, then the key's first position
    __shared__ uint32_t s_kr[4096];                        // per item of the run: key | rank within the key << 10
    __shared__ uint32_t s_wsum[kT / 64];
    const uint32_t n = *work_count, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t k = tid; k < ntiles; k += kT) last_blocks[k] = 0u;         // this frame's tile kernel adds its times
    for (uint32_t k = tid; k < nlocal; k += kT) background[k] = flags_new[k];     // the plan is done reading the old words
    for (uint32_t base = 0; base < n; base += 4096) {
        const uint32_t m = min(4096u, n - base);
        for (uint32_t k = tid; k < kKeys; k += kT) s_hist[k] = 0u;
        __syncthreads();
#pragma unroll 1
        for (uint32_t k = tid; k < m; k += kT) {
            // heaviest first: key 0 = the largest weights.  (float)w >> 18 = exponent and five mantissa bits; w >= 1
            const uint32_t f = (__float_as_uint((float)max(work[base + k].y, 1u)) >> 18) - (127u << 5);
            const uint32_t key = (kKeys - 1u) - min(f, kKeys - 1u);
            s_kr[k] = key | (atomicAdd(&s_hist[key], 1u) << 10);
        }
        __syncthreads();
        // exclusive prefix sum over the 1024 keys: four keys per thread, DPP scan per wave, the four wave totals by every thread
        const uint32_t a0 = s_hist[4u * tid], a1 = s_hist[4u * tid + 1u], a2 = s_hist[4u * tid + 2u], a3 = s_hist[4u * tid + 3u];
        const uint32_t mine = a0 + a1 + a2 + a3;
        const uint32_t inc = wave_scan_add(mine);
        if (lane == 63u) s_wsum[wave] = inc;
        __syncthreads();
        uint32_t before = 0;
#pragma unroll
        for (uint32_t w = 0; w < kT / 64; ++w) before += w < wave ? s_wsum[w] : 0u;
        const uint32_t first = before + inc - mine;
        s_hist[4u * tid] = first; s_hist[4u * tid + 1u] = first + a0; s_hist[4u * tid + 2u] = first + a0 + a1; s_hist[4u * tid + 3u] = first + a0 + a1 + a2;
        __syncthreads();
#pragma unroll 1
        for (uint32_t k = tid; k < m; k += kT) {
            const uint32_t kr = s_kr[k];
            sorted[base + s_hist[kr & (kKeys - 1u)] + (kr >> 10)] = work[base + k];
        }
        __syncthreads();
    }
}

// Tile kernel.  Structure per tile (one workgroup, kTileThreads / 64 waves):
//   1. bitmask of the block rows whose pixel box touches the tile;
//   2. chunks of up to kMaxSteps block rows in DESCENDING order: every wave tests the blocks of "its" rows against the
//      tile (box, capsule, open pixels) and the hits form one work list (a row = one "step");
//   3. waves pull blocks from the list with an LDS counter and rasterise them independently -- no barrier: painting is
//      an atomic max, so the order inside the list does not matter for the result;
//   4. a wave that finishes the last block of a step tries to advance the "final step" frontier: once every block of
//      steps 0..s is done, pixels owned by ids >= first id of step s are final; the frontier owner rescans the tile and
//      publishes the final-pixel masks that later blocks, lines and pixels are culled against (stale masks are merely
//      conservative); a fully final tile stops early;
//   5. fragment stage on the LDS tile.
#ifdef VF_PHASE_PROF   // diagnostics build: per-phase shader-clock cycles and event counts (vf_terrain_debug_phase_cycles)
struct PhaseClock {
    uint64_t acc[20], last;                                // 8..15: parts of the set-up; 16..19: parts of `vertex`
    __device__ __forceinline__ void start() { for (int k = 0; k < 20; ++k) acc[k] = 0; last = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void tick(int p) { const uint64_t now = __builtin_readcyclecounter(); acc[p] += now - last; last = now; }
};
#define VF_PH_INIT PhaseClock PH; PH.start();
#define VF_PH(p) PH.tick(p);
#define VF_PH_ARG , PhaseClock &PH
#define VF_PH_PASS , PH
#else
#define VF_PH_INIT
#define VF_PH(p)
#define VF_PH_ARG
#define VF_PH_PASS
#endif
// ---- k_tile's constants, its LDS and its phases --------------------------------------------------------------------------------
namespace tile {
constexpr int kWaves = kTileThreads / 64;
constexpr int kNV = kBlockVerts * kBlockVerts;             // 81
constexpr uint32_t kChunk = 4096;                          // work-list entries per chunk (>= one full block row: nb <= 1024)
constexpr int kMaxSteps = 128;                             // block rows per chunk
constexpr int kHitWords = 16;                              // 64-bit ballots per block row (nb <= 1024)
constexpr int kRescanEvery = 1;                            // publish new masks when the frontier moved this many steps
constexpr int kBlockPrims = 2 * kBlockCells * kBlockCells;
constexpr uint32_t kWideLines = 28, kBalGain = 2;          // lane dealing by line counts: looked at when a survivor has more lines than this / taken when it saves this many trips
// The waves' private arrays of the block loop share their LDS with the ballots of the list building (`hit`): the ballots are dead
// from the barrier behind the list fill to the barrier behind the block loop, the private arrays live only between the two.  The
// kernel's LDS falls from 80.5 to 64 KB: two of its workgroups (this frame's last, the next frame's first) then leave a CU 32 KB for
// the set-up pass that runs beside them, where they left 2.8 -- C4 -1.4 %, top-down camera -0.9 % (round 5; spent on chunks of 192
// block rows instead, the 16 KB lose: 0.7189 -> 0.7239 ms).
struct WaveLds {
    int2 xy[kWaves][kNV];                                  // per wave: snapped vertices of the current block (from k_block_setup)
    uint8_t alive[kWaves][kBlockPrims];                    // per wave: the block's alive primitives (cell << 1 | odd), compacted
    uint8_t surv[kWaves][kBlockPrims];                     // per wave: those of them that survive against this tile
    uint8_t lines[kWaves][kBlockPrims];                    // per wave: ... and how many lines each of them has inside the tile
};
constexpr size_t kHitBytes = sizeof(unsigned long long) * kMaxSteps * kHitWords;
constexpr size_t kOverlayBytes = sizeof(WaveLds) > kHitBytes ? sizeof(WaveLds) : kHitBytes;
// The workgroup's LDS as the phases below see it: pointers to the kernel's __shared__ arrays (every phase is inlined into the kernel,
// where they ARE the arrays again -- one struct in LDS instead would cost the compiler the knowledge that the arrays do not overlap).
struct Lds {
    uint32_t *vis;                                         // the 64 x 64 visibility tile (rows skewed: vis_index)
    unsigned long long (*hit)[kHitWords];                  // 64-bit ballots per block row of the chunk (list building)
    int2 (*xy)[kNV]; uint8_t (*alive)[kBlockPrims]; uint8_t (*surv)[kBlockPrims]; uint8_t (*lines)[kBlockPrims];   // WaveLds (block loop)
    uint32_t *list;                                        // bx | by << 10 | step << 20
    uint32_t *cnt;                                         // candidates per step
    uint16_t *words;                                       // per step: first | end << 8 of the 64-block groups its ballots were taken for
    uint32_t *pending;                                     // blocks of the step not finished yet
    uint32_t *firstid;
    uint16_t *allrows;                                     // block rows that reach the tile, descending (nb <= 1024)
    uint32_t *rc;                                          // per block row: first | end << 16 of the blocks that can reach this tile column
    uint32_t *colfin, *rowfin, *colfin4, *rowfin4;         // final-pixel masks per line, per four lines (refresh_fin4)
    uint32_t *part;
    unsigned long long *rows;                              // bit r of word w: block row 64 w + r reaches this tile
    uint32_t *next, *lock, *done, *frontier, *published, *blocks, *redo;
    const uint32_t *per;                                   // [survivors]: lanes per survivor | ceil(2^16 / that) << 7 | survivors per round << 24
};
// (the frontier words are read while other waves write them: relaxed atomic loads on the LDS variables themselves -- a `volatile`
//  pointer loses the address space, the loads become FLAT ones and their 64-bit generic addresses live in (spilled) vector registers)
__device__ __forceinline__ uint32_t lds_peek(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// An item's block rows: those whose pixel box touches the tile (most tiles of a frame see none: background), as a bit mask and then
// as a list, highest row first (= descending primitive id).  Returns their number.  Two workgroup barriers inside.
__device__ __forceinline__ uint32_t list_block_rows(const FrameParams &P, const Lds &L, const TileCtx &T, const PixelBox *__restrict__ row_boxes,
                                                    const uint32_t *__restrict__ rc_lo, const uint32_t *__restrict__ rc_hi, uint32_t tcol,
                                                    uint32_t tid, uint32_t lane, uint32_t wave VF_PH_ARG)
{
    // ---- block rows whose box touches the tile (most tiles of a frame see none: background) ----
    for (uint32_t base = 0; base < P.nb; base += kTileThreads) {
        const uint32_t r = base + tid;
        bool hit = false;
        if (r < P.nb) {
            const PixelBox rr = row_boxes[r];
            const uint32_t lo = rc_lo[tcol * P.nb + r], hi = rc_hi[tcol * P.nb + r];
            L.rc[r] = lo < hi ? lo | (hi << 16) : 1u;                  // (1 = the empty range [1, 0))
            hit = lo < hi && rr.x0 <= rr.x1 && rr.x1 >= T.px_lo && rr.x0 <= T.px_hi && rr.y1 >= T.py_lo && rr.y0 <= T.py_hi;
        }
        const unsigned long long m = __ballot(hit);
        if (lane == 0 && m) L.rows[r >> 6] = m;            // r is a multiple of 64 for lane 0
    }
    __syncthreads();
    VF_PH(14)                                              // row mask

    // ---- the hit rows as a list, highest first (= descending primitive id): wave w expands words 15 - w, 15 - w - kWaves, ... ----
    uint32_t nrows_total = 0;
    {
        uint32_t cnt[16];
#pragma unroll
        for (uint32_t w = 0; w < 16u; ++w) { cnt[w] = (uint32_t)__popcll(L.rows[w]); nrows_total += cnt[w]; }
        for (uint32_t word = 15u - wave; word < 16u; word -= (uint32_t)kWaves) {     // (unsigned wrap ends the loop)
            uint32_t above = 0;                            // hit rows in the words above this one
#pragma unroll
            for (uint32_t w = 0; w < 16u; ++w) above += w > word ? cnt[w] : 0u;
            const unsigned long long m = L.rows[word];
            const uint32_t b = 63u - lane;                 // lane 0 takes the highest row of the word
            if ((m >> b) & 1ull) L.allrows[above + (uint32_t)__popcll(b == 63u ? 0ull : m >> (b + 1u))] = (uint16_t)(word * 64u + b);
        }
    }
    __syncthreads();

    return nrows_total;
}

// A chunk's candidate tests for narrow strips (the kernel instantiation without line groups): which blocks of the chunk's rows can still
// draw into the tile?  Ballots per row in `hit` (kept for the list fill), counts in `cnt`.
// Lanes = (row, block) pairs: a row's range holds a dozen blocks in the far field and fewer elsewhere, so a wave takes EIGHT of
// the chunk's rows at once, eight lanes each (round 4; before: four rows per pass, 64 lanes per row, ten of them busy -- two
// passes, i.e. two memory round trips, and four times the instructions per chunk).  Two blocks per lane and trip are requested
// together: the latency of the bounds -- the whole cost of this phase -- is paid once per sixteen blocks of a row.  Hits are
// OR-ed into the row's ballot words (the list fill reads the same words as before: same list, same order).
template <bool GROUPS>
__device__ __forceinline__ void test_candidates_pairs(const FrameParams &P, const SetupView &V, const Lds &L, const TileCtx &T, const float4 *__restrict__ cap_seg,
                                                      const float *__restrict__ cap_rad, uint32_t cursor, uint32_t nrowsteps, uint32_t lane, uint32_t wave)
{
    const uint32_t rsub = lane >> 3, bsub = lane & 7u;
    uint32_t *const hit32 = reinterpret_cast<uint32_t *>(&L.hit[0][0]);
    for (uint32_t kbase = 0; kbase < nrowsteps; kbase += 8u * kWaves) {            // (one trip for a chunk of <= 128 rows)
        const uint32_t k = kbase + wave + rsub * kWaves;
        const bool valid = k < nrowsteps;
        uint32_t by = 0, lo = 1, hi = 0;
        if (valid) { by = L.allrows[cursor + k]; const uint32_t range = L.rc[by]; lo = range & 0xFFFFu; hi = range >> 16; }
        const uint32_t g_first = lo >> 6, g_last = lo < hi ? ((hi - 1u) >> 6) + 1u : g_first;
        if (valid && bsub == 0u) for (uint32_t g = g_first; g < g_last; ++g) L.hit[k][g] = 0ull;
        __builtin_amdgcn_wave_barrier();            // (LDS operations of one wave complete in order: the zeros are behind us)
        uint32_t cnt = 0;
        for (uint32_t b0 = lo + bsub; b0 < hi; b0 += 16u) {
            const uint32_t b1 = b0 + 8u;
            const bool in1 = b1 < hi;
            const uint32_t i0 = by * P.nb + b0, i1 = by * P.nb + (in1 ? b1 : b0);
            const PixelBox box0 = V.recs[i0].box, box1 = V.recs[i1].box;
            const float4 seg0 = cap_seg[i0], seg1 = cap_seg[i1];
            const float rad0 = cap_rad[i0], rad1 = cap_rad[i1];
            if (block_is_candidate<GROUPS>(box0, seg0, rad0, T)) { atomicOr(&hit32[k * (2u * kHitWords) + (b0 >> 5)], 1u << (b0 & 31u)); ++cnt; }
            if (in1 && block_is_candidate<GROUPS>(box1, seg1, rad1, T)) { atomicOr(&hit32[k * (2u * kHitWords) + (b1 >> 5)], 1u << (b1 & 31u)); ++cnt; }
        }
        // the row's eight lanes: quad, the other quad (DPP, no LDS round trips)
        cnt += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cnt, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
        cnt += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x4E, 0xF, 0xF, false);      // quad_perm [2,3,0,1]
        cnt += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x141, 0xF, 0xF, false);     // row_half_mirror
        // (the list fill reads the ballots of groups g_first .. g_last - 1 only: the others are never written)
        if (valid && bsub == 0u) { L.cnt[k] = cnt; L.words[k] = (uint16_t)(g_first < g_last ? g_first | (g_last << 8) : 0u); L.firstid[k] = 2u * (by * kBlockCells * P.nm1) + 1u; }   // smallest (id + 1) of the row
    }
}

// ... and for wide items (the instantiation with line groups; its items have a dozen rows): lanes = the blocks of a row.
// Four rows per pass: their bounds (pixel box, capsule) are fetched together, so the global-memory latency -- the
// whole cost of this phase -- is paid once per pass instead of once per row and array.
template <bool GROUPS>
__device__ __forceinline__ void test_candidates_rows(const FrameParams &P, const SetupView &V, const Lds &L, const TileCtx &T, const float4 *__restrict__ cap_seg,
                                                     const float *__restrict__ cap_rad, uint32_t cursor, uint32_t nrowsteps, uint32_t hit_words, uint32_t lane, uint32_t wave)
{
    constexpr int kRowsAtOnce = 4;
    for (uint32_t k0 = wave; k0 < nrowsteps; k0 += kRowsAtOnce * kWaves) {
        uint32_t by[kRowsAtOnce], bx_lo[kRowsAtOnce], bx_hi[kRowsAtOnce], cnt[kRowsAtOnce];
#pragma unroll
        for (int r = 0; r < kRowsAtOnce; ++r) {
            const uint32_t k = k0 + (uint32_t)r * kWaves;
            const bool valid = k < nrowsteps;
            by[r] = valid ? (uint32_t)__builtin_amdgcn_readfirstlane((int)L.allrows[cursor + k]) : 0u;
            // only blocks [bx_lo, bx_hi) of the row can reach the tile column; an absent row gets an empty range
            const uint32_t range = valid ? (uint32_t)__builtin_amdgcn_readfirstlane((int)L.rc[by[r]]) : 1u;
            bx_lo[r] = range & 0xFFFFu;
            bx_hi[r] = range >> 16;
            cnt[r] = 0u;
        }
        uint32_t g_first = hit_words, g_last = 0;                 // groups of 64 blocks that hold any block of the four ranges
#pragma unroll
        for (int r = 0; r < kRowsAtOnce; ++r)
            if (bx_lo[r] < bx_hi[r]) { g_first = min(g_first, bx_lo[r] >> 6); g_last = max(g_last, ((bx_hi[r] - 1u) >> 6) + 1u); }
        for (uint32_t g = g_first; g < g_last; ++g) {
            const uint32_t bx = g * 64u + lane;
            PixelBox box[kRowsAtOnce];
            float4 seg[kRowsAtOnce];
            float rad[kRowsAtOnce];
            bool in[kRowsAtOnce];
#pragma unroll
            for (int r = 0; r < kRowsAtOnce; ++r) {
                in[r] = bx >= bx_lo[r] && bx < bx_hi[r];
                box[r] = PixelBox{ 1, 1, 0, 0 }; seg[r] = make_float4(0.f, 0.f, 0.f, 0.f); rad[r] = 0.0f;
                if (in[r]) {
                    const uint32_t bidx = by[r] * P.nb + bx;
                    box[r] = V.recs[bidx].box; seg[r] = cap_seg[bidx]; rad[r] = cap_rad[bidx];
                }
            }
#pragma unroll
            for (int r = 0; r < kRowsAtOnce; ++r) {
                const uint32_t k = k0 + (uint32_t)r * kWaves;
                if (k >= nrowsteps) continue;                                   // uniform
                const unsigned long long m = __ballot(in[r] && block_is_candidate<GROUPS>(box[r], seg[r], rad[r], T));
                if (lane == 0) L.hit[k][g] = m;
                cnt[r] += (uint32_t)__popcll(m);
            }
        }
#pragma unroll
        for (int r = 0; r < kRowsAtOnce; ++r) {
            const uint32_t k = k0 + (uint32_t)r * kWaves;
            // (the list fill reads the ballots of groups g_first .. g_last - 1 only: the others are never written)
            if (lane == 0 && k < nrowsteps) { L.cnt[k] = cnt[r]; L.words[k] = (uint16_t)(g_first < g_last ? g_first | (g_last << 8) : 0u); L.firstid[k] = 2u * (by[r] * kBlockCells * P.nm1) + 1u; }   // smallest (id + 1) of the row
        }
    }
}

// A chunk's work list from the kept ballots: how many of its rows fit the list (`nsteps`), how many blocks they hold (`nlist`).
__device__ __forceinline__ void fill_work_list(const Lds &L, uint32_t cursor, uint32_t nrowsteps, uint32_t lane, uint32_t wave, uint32_t &nsteps, uint32_t &nlist)
{
    // ---- chunk set-up 2: every wave scans the row counts for itself (64 rows at a time, DPP), so all agree on the list offsets
    //      and on how many rows fit the list without another barrier; rows that do not fit wait for the next chunk ----
    constexpr int kParts = kMaxSteps / 64;              // the rows of a chunk, 64 (one per lane) at a time
    static_assert(kMaxSteps % 64 == 0 && kParts >= 1 && kParts <= 4, "the offset scan below works on 64-row parts");
    uint32_t c[kParts], inc[kParts];
    unsigned long long fm[kParts];
#pragma unroll
    for (int p = 0; p < kParts; ++p) {
        c[p] = lane + 64u * p < nrowsteps ? L.cnt[lane + 64u * p] : 0u;
        inc[p] = wave_scan_add(c[p]);
        if (p) inc[p] += (uint32_t)__builtin_amdgcn_readlane((int)inc[p - 1], 63);
        fm[p] = __ballot(lane + 64u * p < nrowsteps && inc[p] <= kChunk);
    }
    // rows are admitted in order: stop at the first one that does not fit (a single row always fits: nb <= 1024 < kChunk)
    nsteps = 64u * kParts;
#pragma unroll
    for (int p = kParts - 1; p >= 0; --p) if (fm[p] != ~0ull) nsteps = 64u * p + (uint32_t)__builtin_ctzll(~fm[p]);
    // wave-uniform values are read with readlane / readfirstlane so that they, and the addresses derived from them, live in
    // scalar registers: the vector register file is the scarce resource of this kernel
    auto row_value = [&](const uint32_t (&v)[kParts], uint32_t k) -> uint32_t {      // v of row k (k wave-uniform)
        uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)v[0], (int)(k & 63u));
#pragma unroll
        for (int p = 1; p < kParts; ++p) if ((k >> 6) == (uint32_t)p) r = (uint32_t)__builtin_amdgcn_readlane((int)v[p], (int)(k & 63u));
        return r;
    };
    nlist = row_value(inc, nsteps - 1u);
    // ---- chunk set-up 3: fill the work list from the kept ballots ----
    for (uint32_t k = wave; k < nsteps; k += kWaves) {
        const uint32_t by = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.allrows[cursor + k]);
        uint32_t pos = row_value(inc, k) - row_value(c, k);
        uint32_t cnt = 0;
        const uint32_t words = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.words[k]);
        for (uint32_t g = words & 0xFFu; g < (words >> 8); ++g) {
            const unsigned long long m = L.hit[k][g];
            if ((m >> lane) & 1ull) L.list[pos + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (g * 64u + lane) | (by << 10) | (k << 20);
            const uint32_t c = (uint32_t)__popcll(m);
            pos += c; cnt += c;
        }
        if (lane == 0) L.pending[k] = cnt;
    }
}

// Every block of the chunk is done: exact masks for the next chunk; returns the number of final pixels.  One workgroup barrier inside.
template <bool GROUPS>
__device__ __forceinline__ uint32_t publish_chunk_masks(const Lds &L, uint32_t nsteps, uint32_t lane, uint32_t wave)
{
    // ---- end of chunk: every block of the chunk is done; publish exact masks for the next chunk ----
    {
        constexpr int kRowsPerWave = kTileH / kWaves;
        // once per chunk: hide `lane` from the optimiser here, or it computes this unrolled loop's LDS addresses at kernel
        // entry and keeps them in (in fact: spills them from) vector registers for the whole kernel
        uint32_t lane_here = lane;
        asm volatile("" : "+v"(lane_here));
        const uint32_t nfinal = rescan_final(L.vis, L.colfin, L.rowfin, lane_here, L.firstid[nsteps - 1], (int32_t)wave * kRowsPerWave,
                                             (int32_t)(wave + 1) * kRowsPerWave);
        if (lane == 0) L.part[wave] = nfinal;
    }
    __syncthreads();
    if constexpr (GROUPS)
        if (wave == 0) refresh_fin4(L.colfin, L.rowfin, L.colfin4, L.rowfin4, lane);   // (the other waves may start the next chunk on the old four-line masks: they lag, they do not lie)
    uint32_t all = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) all += L.part[w];
    return all;
}

// The fragment stage on the LDS tile; one wave writes one 256-byte row segment.
template <bool WRITE_VIS, bool COMPLETE, bool FAST>
__device__ __forceinline__ void shade_item(const FrameParams &P, const SetupView &V, const ShadeTables &S, const Lds &L, const TileCtx &T, const TilePlace &tp, int32_t tile_x0,
                                           uint32_t *__restrict__ rgba, uint32_t *__restrict__ vis_out, uint32_t tid)
{
    const int32_t item_w = T.px_hi - T.px_lo + 1;
    // ---- fragment stage on the LDS tile; one wave writes one 256-byte row segment ----
    if (item_w > 32 || (item_w & (item_w - 1)) != 0) {     // whole tiles, and tiles the target's right edge cuts
        for (int k = tid; k < kTileW * kTileH; k += kTileThreads) {
            const int32_t lx = k & (kTileW - 1), ly = k / kTileW;
            const int32_t px = T.px_lo + lx, py = T.py_lo + ly;
            if (px > T.px_hi || py > T.py_hi) continue;
            const uint32_t id = L.vis[vis_index(lx, ly)];
            const size_t o = tp.out_base + (size_t)ly * tp.out_stride + (uint32_t)(px - tile_x0);
            rgba[o] = id ? shade_pixel<COMPLETE, FAST>(P, V, S, id - 1u, px, py) : P.clear_rgba;
            if (WRITE_VIS) vis_out[o] = id;
        }
    } else {
        // a strip (32, 16, 8 or 4 pixels wide): lanes run over the pixels of the strip, not of the 64 x 64 tile -- a 4-pixel strip is
        // shaded by four waves in one trip, not by four lanes of every wave in four
        const int32_t w_shift = 31 - __builtin_clz(item_w);
        for (int k = tid; k < ((T.py_hi - T.py_lo + 1) << w_shift); k += kTileThreads) {
            const int32_t ly = k >> w_shift, lx = k - (ly << w_shift);
            const int32_t px = T.px_lo + lx, py = T.py_lo + ly;
            const uint32_t id = L.vis[vis_index(lx, ly)];
            const size_t o = tp.out_base + (size_t)ly * tp.out_stride + (uint32_t)(px - tile_x0);
            rgba[o] = id ? shade_pixel<COMPLETE, FAST>(P, V, S, id - 1u, px, py) : P.clear_rgba;
            if (WRITE_VIS) vis_out[o] = id;
        }
    }
}
} // namespace tile

// Two variants.  COMPLETE = false is the frame's main launch, one workgroup per planned item: it draws every primitive that
// needs no clipping and is not oversized -- all of them in ordinary views -- and files an item that met one of the others
// in `redo`.  COMPLETE = true is a small persistent launch that renders the filed items again, this time with the generic
// path (Sutherland-Hodgman clipping, per-pixel int64 coverage).  Keeping that path -- non-inlined calls, stack arrays -- out
// of the main kernel is what lets it live in 96 vector registers (VF_TILE_MIN_WAVES, vf_device.h: the next frame's set-up kernel
// shares the CUs with it).  At that cap it spills 64 bytes per lane (profiles/r03_isa_stats.txt, tools/isa_stats.py): the stores at
// kernel entry and in the item loop, the reloads in the item and chunk loops -- and ONE reload per pulled block (loop depth 3, the
// block pull loop); none in pass A, pass B, the line loop or the paint loop.
// GROUPS: the line loop of raster_fast tests a triangle's lines in groups of four first (above).  A launch-time choice, not a per-item
// one: both loops in one kernel cost the narrow strips of a multi-GPU rank 6 % (registers: the block loop's spills), so the host picks
// the instantiation per handle -- whole frames and wide shards with groups, many-rank shards (mostly strips) without (vf_hip.hip).
template <bool WRITE_VIS, bool COMPLETE, bool FAST, bool GROUPS>
__global__ __launch_bounds__(kTileThreads, VF_TILE_MIN_WAVES) void k_tile(FrameParams P, SetupView V, const PixelBox *__restrict__ row_boxes,
                                                       const float4 *__restrict__ cap_seg, const float *__restrict__ cap_rad,
                                                       const float *__restrict__ lut_linear, const float *__restrict__ thresh,
                                                       const uint2 *__restrict__ work, uint32_t *__restrict__ work_count,
                                                       const uint32_t *__restrict__ rc_lo, const uint32_t *__restrict__ rc_hi,
                                                       uint32_t *__restrict__ rgba, uint32_t *__restrict__ vis_out, uint32_t *stats,
                                                       uint32_t *__restrict__ last_blocks, uint32_t *__restrict__ redo)
{
    using namespace tile;
    uint32_t *const redo_count = work_count + 3;           // items handed to the complete variant
    // ---- LDS (64 KB): separate arrays, bundled into `L` for the phases (tile::Lds) ----
    __shared__ uint32_t s_vis[kTileW * kTileH];
    __shared__ __attribute__((aligned(16))) unsigned char s_overlay[kOverlayBytes];   // list-building ballots, then the waves' private arrays (WaveLds)
    __shared__ uint32_t s_list[kChunk];
    __shared__ uint32_t s_cnt[kMaxSteps];
    __shared__ uint16_t s_words[kMaxSteps];
    __shared__ uint32_t s_pending[kMaxSteps];
    __shared__ uint32_t s_firstid[kMaxSteps];
    __shared__ uint16_t s_allrows[1024];
    __shared__ uint32_t s_rc[1024];
    __shared__ uint32_t s_colfin[kTileW * 2];
    __shared__ uint32_t s_rowfin[kTileH * 2];
    __shared__ uint32_t s_colfin4[kTileW / 4 * 2], s_rowfin4[kTileH / 4 * 2];
    __shared__ uint32_t s_part[kWaves];
    __shared__ unsigned long long s_rows[16];
    __shared__ uint32_t s_next, s_lock, s_done, s_frontier, s_published, s_blocks, s_redo, s_item;
    __shared__ __attribute__((aligned(16))) float s_lut[kLutFloats];
    __shared__ float s_thr[256];
    __shared__ uint32_t s_per[65];
    WaveLds &s_wave = *reinterpret_cast<WaveLds *>(s_overlay);
    int2 (&sXY)[kWaves][kNV] = s_wave.xy;
    uint8_t (&sC)[kWaves][kBlockPrims] = s_wave.alive;
    uint8_t (&sS)[kWaves][kBlockPrims] = s_wave.surv;
    uint8_t (&sL)[kWaves][kBlockPrims] = s_wave.lines;
    Lds L;
    L.vis = s_vis; L.hit = reinterpret_cast<unsigned long long (*)[kHitWords]>(s_overlay);
    L.xy = s_wave.xy; L.alive = s_wave.alive; L.surv = s_wave.surv; L.lines = s_wave.lines;
    L.list = s_list; L.cnt = s_cnt; L.words = s_words; L.pending = s_pending; L.firstid = s_firstid; L.allrows = s_allrows; L.rc = s_rc;
    L.colfin = s_colfin; L.rowfin = s_rowfin; L.colfin4 = s_colfin4; L.rowfin4 = s_rowfin4; L.part = s_part; L.rows = s_rows;
    L.next = &s_next; L.lock = &s_lock; L.done = &s_done; L.frontier = &s_frontier; L.published = &s_published; L.blocks = &s_blocks; L.redo = &s_redo;
    L.per = s_per;

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));   // wave-uniform: keep it (and what derives from it) scalar
    VF_PH_INIT
    VF_RC(RasterCounts RC = {};)
    uint32_t redo_at = blockIdx.x;                         // COMPLETE: position in the list of items to render again
    if (COMPLETE && redo_at >= *redo_count) return;        // (normally the case for every workgroup of that launch)
    for (int k = tid; k < kLutFloats; k += kTileThreads) s_lut[k] = lut_linear[k];
    for (int k = tid; k < 256; k += kTileThreads) s_thr[k] = thresh[k];
    if (tid <= 64u) { const uint32_t per = tid ? 64u / tid : 64u; s_per[tid] = per | (((65536u + per - 1u) / per) << 7) | ((64u / per) << 24); }
    const ShadeTables S = { s_lut, s_thr };
    const uint32_t nwork = *work_count;
    uint32_t pulled = 0;
    if (!COMPLETE) {
        if (tid == 0) s_item = atomicAdd(work_count + 2, 1u);
        __syncthreads();
        pulled = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_item);
        if (pulled >= nwork) return;
    }
    for (;;) {                                             // ---- the item loop: a persistent workgroup draws items until the queue is empty ----
    VF_PH(12)                                              // hand-over: wait for the other waves, pull the next item
    const uint32_t item_idx = COMPLETE ? redo[redo_at] : pulled;
    const uint64_t t_start = __builtin_amdgcn_s_memrealtime();   // 100 MHz wall clock: scheduling feedback + diagnostics
    // tile: work-list entry -> (tile column, local tile row) -> pixel rectangle of this shard
    const uint32_t item = work[item_idx].x;
    const uint32_t tile = work_tile(item);
    TileCtx T;
    T.vis = s_vis; T.colfin = s_colfin; T.rowfin = s_rowfin; T.colfin4 = s_colfin4; T.rowfin4 = s_rowfin4;
    const TilePlace tp = tile_rect(P, tile, T.px_lo, T.px_hi, T.py_lo, T.py_hi);
    const uint32_t tcol = tp.tx;
    const int32_t tile_x0 = T.px_lo;                       // the tile's left edge (T.px_lo becomes the strip's below)
    work_strip(item, T.px_lo, T.px_hi);                    // heavy tiles arrive as 2..16 column strips
    const uint32_t tile_pixels = (uint32_t)(T.px_hi - T.px_lo + 1) * (uint32_t)(T.py_hi - T.py_lo + 1);
    const uint64_t row_full = ~0ull >> (63 - (T.px_hi - T.px_lo));   // row mask of a fully final row of this tile / strip

    for (int k = tid; k < kTileW * kTileH; k += kTileThreads) s_vis[k] = 0u;
    for (int k = tid; k < kTileW * 2; k += kTileThreads) s_colfin[k] = 0u;
    for (int k = tid; k < kTileH * 2; k += kTileThreads) s_rowfin[k] = 0u;
    if (tid < kTileW / 4 * 2) { s_colfin4[tid] = 0u; s_rowfin4[tid] = 0u; }
    if (tid < 16) s_rows[tid] = 0ull;
    if (tid == 0) { s_done = 0; s_blocks = 0; s_redo = 0; }
    __syncthreads();
    VF_PH(13)                                              // item record, tile state
    const uint32_t nrows_total = list_block_rows(P, L, T, row_boxes, rc_lo, rc_hi, tcol, tid, lane, wave VF_PH_PASS);
    VF_PH(8)                                                // item start, tile state, row list
    uint32_t dbg_loop = 0;                                 // (VF_DIAG_ITEM=1 only; dead code otherwise)
    const uint32_t hit_words = (P.nb + 63u) / 64u;
    for (uint32_t cursor = 0; cursor < nrows_total;) {     // ---- chunks of <= kMaxSteps block rows, nearest first; uniform: nothing (left) to draw ends the loop ----
        // chunk set-up 1: each wave tests the blocks of its rows (the next <= kMaxSteps of the list) against the tile; the ballots are kept
        const uint32_t nrowsteps = min((uint32_t)kMaxSteps, nrows_total - cursor);
        if (tid == 0) { s_next = 0; s_lock = 0; s_frontier = 0; s_published = 0; }
        // (lanes = (row, block) pairs in the kernel instantiation WITHOUT line groups only -- the one a handle ends up with when its items
        //  are narrow strips of far-field tiles, hundreds of rows each: C4, a rank of eight 0.214 -> 0.199 ms; the other instantiation's
        //  wide items have a dozen rows and lose 1 % to the longer code: measured both ways, EXPERIMENTS.md)
        if (!GROUPS) test_candidates_pairs<GROUPS>(P, V, L, T, cap_seg, cap_rad, cursor, nrowsteps, lane, wave);
        else test_candidates_rows<GROUPS>(P, V, L, T, cap_seg, cap_rad, cursor, nrowsteps, hit_words, lane, wave);
        VF_PH(9)                                            // candidate tests
        __syncthreads();
        VF_PH(10)                                           // ... waiting for the slowest wave
        // chunk set-up 2 + 3: list offsets, the work list
        uint32_t nsteps, nlist;
        fill_work_list(L, cursor, nrowsteps, lane, wave, nsteps, nlist);
        cursor += nsteps;
        VF_PH(11)                                           // scan + list fill
        __syncthreads();
        VF_PH(0)

        // ---- asynchronous raster: waves pull blocks until the list is empty or the tile is final ----
        uint32_t my_blocks = 0;
        const uint64_t dbg_t0 = kDiagItem == 1 ? __builtin_amdgcn_s_memrealtime() : 0ull;
        for (;;) {
            // (a finished tile pushes the list counter past the list's end: one LDS round trip tells "nothing left" and "tile final" apart
            //  from "here is your block" -- a separate look at a done flag was one more dependent round trip per block: -1.6 %)
            uint32_t idx = 0;
            if (lane == 0) idx = atomicAdd(&s_next, 1u);
            idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
            if (idx >= nlist) break;
            const uint32_t entry = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_list[idx]);
            const uint32_t bx = entry & 0x3FFu, by = (entry >> 10) & 0x3FFu, stepidx = entry >> 20;
            // WAVE PRIORITY (round 4).  What every wave of the tile culls against -- blocks, triangles, lines, pixels -- is the set of FINAL
            // pixels, and that set grows when the oldest unfinished step completes and its completer has rescanned the tile.  So the
            // wave that holds a block of that step goes first on its SIMD (s_setprio 2), the step behind it next (1), everything
            // younger last (0), and the rescan itself above all of them (3, below): the masks come out earlier and everybody else does
            // less.  Scheduling only -- the same pixels.  C4: one GPU -2 %, top-down camera -5.7 %, a rank of eight at that camera -4 %.
            // In the instantiation for wide items only (GROUPS): the narrow strips of a many-rank shard and the C5 orbit, which the
            // other instantiation draws, gain nothing (a rank of eight at the default camera +1 %, C5 +-0).
            if (GROUPS) {
                const uint32_t fr_now = lds_peek(&s_frontier);
                if (stepidx <= fr_now) __builtin_amdgcn_s_setprio(2); else if (stepidx == fr_now + 1u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            }
            // One round trip for everything the block needs from HBM: its record (pixel box, alive masks) and its 81 snapped
            // vertices are requested together, before the box decides whether the block is still worth drawing -- a dependent
            // second trip costs a wave more than the vertices of the blocks that turn out culled.
            const uint32_t bidx = by * P.nb + bx;
            // (wave-uniform address, data the set-up pass wrote before this launch: read through the scalar cache into scalar registers --
            //  the box arithmetic of the late cull below then runs on the scalar unit)
            // SCALAR-CACHE COHERENCE, the assumption this rests on: k_block_setup wrote the record with VECTOR stores, on another
            // stream, into a buffer the tile kernel of two frames ago read through the scalar cache at the same address.  The scalar
            // cache is not coherent with vector stores; what makes the read safe is the kernel boundary: every dispatch begins with an
            // acquire that invalidates the scalar (and vector L1) caches of the CUs it lands on, and the producer's release at its end
            // writes the record back to L2 before the event this launch waits on signals.  A record can therefore only be stale if it
            // were rewritten WHILE this launch runs -- and the plan state this launch reads is not written again before its `drawn`
            // event (render_impl: the next user of the set waits for it).  tests/test_gpu_parity.py::
            // test_records_rewritten_every_frame_are_never_read_stale rewrites both sets with different records 20 frames in a row.
            typedef uint32_t __attribute__((ext_vector_type(8))) rec_words;
            const rec_words rw = *(const __attribute__((address_space(4))) rec_words *)(uintptr_t)(V.recs + bidx);
            const uint4 r_lo = make_uint4(rw[0], rw[1], rw[2], rw[3]), r_hi = make_uint4(rw[4], rw[5], rw[6], rw[7]);   // box (2 words), flags, count | alive_even, alive_odd
            const uint4 *vsrc = reinterpret_cast<const uint4 *>(V.vtx + (size_t)bidx * kBlockStride);
            const uint4 va4 = vsrc[lane];
            uint4 vb4 = make_uint4(0u, 0u, 0u, 0u);
            if (lane < (uint32_t)(kNV - 64)) vb4 = vsrc[64u + lane];
            const int2 xa = make_int2((int32_t)va4.x, (int32_t)va4.y), xb = make_int2((int32_t)vb4.x, (int32_t)vb4.y);   // the raster needs X, Y only
            // late culling against the masks published since the list was built (one column / row per lane)
            bool live;
            {
                const int32_t bx0 = (int32_t)(int16_t)(r_lo.x & 0xFFFFu), by0 = (int32_t)(int16_t)(r_lo.x >> 16);
                const int32_t bx1 = (int32_t)(int16_t)(r_lo.y & 0xFFFFu), by1 = (int32_t)(int16_t)(r_lo.y >> 16);
                const int32_t x0 = max(bx0, T.px_lo) - T.px_lo, x1 = min(bx1, T.px_hi) - T.px_lo;
                const int32_t y0 = max(by0, T.py_lo) - T.py_lo, y1 = min(by1, T.py_hi) - T.py_lo;
                const bool in = (int32_t)lane >= x0 && (int32_t)lane <= x1;
                const uint64_t open = in ? (~load_mask(s_colfin, (int32_t)lane) & bit_range(y0, y1)) : 0ull;
                live = __ballot(open != 0ull) != 0ull;
            }
            VF_PH(1)
            if (live) {
                ++my_blocks;
#ifndef VF_PHASE_PROF   // (the phase build measures waits: a device-scope atomic per pair in the middle of one would be what it measures)
                if (stats && lane == 0)                    // diagnostics: which blocks were drawn by at least one tile this frame
                    atomicOr(&stats[4u + 4u * (P.ntx * P.nty + kSplitBudget) + 2u * kPhaseSlots + (bidx >> 5)], 1u << (bidx & 31u));
#endif
                const uint32_t i0 = bx * kBlockCells, j0 = by * kBlockCells;
                // ---- set-up stage, once per frame in k_block_setup: here the block's snapped vertices and alive masks are loads ----
                const unsigned long long alive_e = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)r_hi.x) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)r_hi.y) << 32);
                const unsigned long long alive_o = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)r_hi.z) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)r_hi.w) << 32);
                const uint32_t rflags = (uint32_t)__builtin_amdgcn_readfirstlane((int)r_lo.z);
                VF_PH(16)                                  // (diagnostics: the record is here)
                VF_PH(17)                                  // (diagnostics: nothing in between -- what one time stamp costs)
#ifdef VF_PHASE_PROF
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                VF_PH(18)                                  // (diagnostics: the vertex records are here)
#endif
                sXY[wave][lane] = xa;
                if (lane < (uint32_t)(kNV - 64)) sXY[wave][64u + lane] = xb;
#ifdef VF_PHASE_PROF
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                VF_PH(19)                                  // (diagnostics: ... and staged in LDS)
#endif
                if (!COMPLETE && (rflags & kRecGeneric)) { if (lane == 0) s_redo = 1u; }   // rare: clipped / oversized -> the COMPLETE launch
                // the alive primitives as a dense list: cell c's even primitive sits at popcount(alive_e below c), its odd one behind all
                // the even ones -- every lane of the classification below then holds a primitive that can draw
                const uint32_t n_even = (uint32_t)__popcll(alive_e), n_alive = n_even + (uint32_t)__popcll(alive_o);
                {
                    const uint32_t pe = __builtin_amdgcn_mbcnt_hi((uint32_t)(alive_e >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)alive_e, 0u));
                    const uint32_t po = __builtin_amdgcn_mbcnt_hi((uint32_t)(alive_o >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)alive_o, 0u));
                    if ((alive_e >> lane) & 1ull) sC[wave][pe] = (uint8_t)(2u * lane);
                    if ((alive_o >> lane) & 1ull) sC[wave][n_even + po] = (uint8_t)(2u * lane + 1u);
                }
                __builtin_amdgcn_wave_barrier();   // LDS ops of one wave complete in order; keep the compiler from reordering
                VF_PH(2)
                if constexpr (COMPLETE) if (rflags & kRecGeneric) {
                    // lane = cell: the primitives k_block_setup marked generic go through clipping and the per-pixel int64 test
                    const ulonglong2 g = V.gen[bidx];
                    const uint32_t lj = lane >> 3, li = lane & 7u;
                    const uint32_t prim = 2u * ((j0 + lj) * P.nm1 + (i0 + li));
                    GVert gv[3];                                               // in memory only on this rare path
                    if ((g.x >> lane) & 1ull) { load_prim(P, V.hblk, prim, gv[0], gv[1], gv[2]); raster_generic(gv, P.hw, P.hh, P.W, P.H, T.vis, T.px_lo, T.px_hi, T.py_lo, T.py_hi, prim + 1u); }
                    if ((g.y >> lane) & 1ull) { load_prim(P, V.hblk, prim + 1u, gv[0], gv[1], gv[2]); raster_generic(gv, P.hw, P.hh, P.W, P.H, T.vis, T.px_lo, T.px_hi, T.py_lo, T.py_hi, prim + 2u); }
                }
                // ---- pass A: lane = alive primitive: bbox against the tile, occlusion; compact the survivors (ballot + prefix popcount) ----
                uint32_t nsurv = 0;
                bool any_wide = false;                     // a survivor with more than kWideLines lines: only then is the lane dealing below looked at
                // ONE classification pass for narrow strips (round 6).  Half of a noise terrain's primitives face the camera: a block's alive
                // list holds 64 +- 6 of them, so nearly every second block used to pay a second pass for a handful of primitives.  Pass B does
                // not need them classified -- raster_fast clamps a triangle's box to the tile itself and culls its lines against the final
                // masks -- so in the strip instantiation, where a primitive beyond the strip leaves the raster at once, the primitives
                // beyond the wave's 64 lanes go straight to the survivor list: a rank of eight 0.1945 -> 0.187 ms, never a pixel.  (Wide
                // items keep the second pass: their unclassified primitives are real candidates and cost the line loop more trips than the
                // pass -- one GPU default camera -1 %, top-down +3 %.)
                uint32_t n_cls = n_alive;
                if constexpr (!GROUPS) n_cls = min(n_alive, 64u);
                for (uint32_t k0 = 0; k0 < n_cls; k0 += 64u) {
                    const uint32_t k = k0 + lane;
                    bool keep = false;
                    uint32_t code = 0, nlines = 0;
                    if (k < n_cls) {
                        code = sC[wave][k];
                        const uint32_t cell = code >> 1, odd = code & 1u;
                        const uint32_t va = (cell >> 3) * kBlockVerts + (cell & 7u);
                        const int2 q0 = sXY[wave][odd ? va + 1u : va], q1 = sXY[wave][va + kBlockVerts], q2 = sXY[wave][odd ? va + kBlockVerts + 1u : va + 1u];
                        keep = classify_alive<GROUPS>(T, q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, nlines VF_RC(, RC));
                    }
                    const unsigned long long m = __ballot(keep);
                    any_wide = any_wide || __ballot(keep && nlines > kWideLines) != 0ull;
                    if (keep) {
                        const uint32_t at = nsurv + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        sS[wave][at] = (uint8_t)code;
                        sL[wave][at] = (uint8_t)nlines;
                    }
                    nsurv += (uint32_t)__popcll(m);
                }
                if constexpr (!GROUPS) if (n_cls < n_alive) {      // (uniform) a few primitives beyond the wave's 64 lanes: straight to the raster, unclassified
                    const uint32_t extra = n_alive - n_cls;
                    if (lane < extra) { sS[wave][nsurv + lane] = sC[wave][n_cls + lane]; sL[wave][nsurv + lane] = (uint8_t)8; }
                    nsurv += extra;
                }
                __builtin_amdgcn_wave_barrier();
                VF_RC(if (lane == 0) { RC.nsurv += nsurv; RC.live++; RC.empty += nsurv ? 0u : 1u; RC.alive += n_alive; RC.apass += (n_alive + 63u) / 64u; })
                VF_PH(3)
                // ---- pass B: the survivors share the wave: with few of them, 2..64 lanes split the lines of one triangle ----
                {
                    // lanes per survivor: floor(64 / nsurv) -- any number, not only powers of two (17 survivors: 3 lanes each, not 2);
                    // lane / per by a 16-bit reciprocal from a table in LDS (exact for lane < 64)
                    const uint32_t pe = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_per[min(nsurv, 64u)]);
                    const uint32_t per = pe & 0x7Fu, group = pe >> 24;          // group = survivors per round = 64 / per
                    const uint32_t q = (lane * ((pe >> 7) & 0x1FFFFu)) >> 16, sub = lane - q * per;
                    // Even shares leave the wave waiting for its widest triangle (6.9 trips through the line loop per pair at C4's default
                    // camera where the lines would fill 3.5; 27 against 13 with the fill camera).  When that costs two trips or more, the
                    // lanes are dealt in proportion to the line counts instead: chunks of C = ceil(lines / (64 - survivors)) lines,
                    // survivor k gets ceil(L_k / C) lanes (at most 64 in all), found by a running maximum over the lanes.
                    bool balanced = false;
                    uint32_t b_mine = 0, b_used = 0;                               // balanced: code | first lane << 8 | lanes << 16 of this lane's survivor
                    if (any_wide && nsurv > 1u && nsurv <= 56u) {                  // (uniform)
                        const uint32_t Lk = lane < nsurv ? (uint32_t)sL[wave][lane] : 0u;
                        const uint32_t incL = wave_scan_add(Lk), maxL = wave_scan_max(Lk);
                        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incL, 63), widest = (uint32_t)__builtin_amdgcn_readlane((int)maxL, 63);
                        const uint32_t spare = 64u - nsurv;
                        // (uniform integer divisions by small numbers: the +0.5 keeps the 1-ulp reciprocal on the right side of exact quotients)
                        uint32_t C = (uint32_t)(((float)(total + spare - 1u) + 0.5f) * __builtin_amdgcn_rcpf((float)spare));      // always fits: sum of ceil(L / C) <= 64
                        {   // rounding up costs half a lane per survivor on average, not a whole one: try the chunk size that assumes so, keep it if it fits
                            const uint32_t spare2 = 64u - (nsurv >> 1) - 2u;
                            const uint32_t C2 = (uint32_t)(((float)(total + spare2 - 1u) + 0.5f) * __builtin_amdgcn_rcpf((float)spare2));
                            const uint32_t m2 = lane < nsurv ? (uint32_t)(((float)(Lk + C2 - 1u) + 0.5f) * __builtin_amdgcn_rcpf((float)C2)) : 0u;
                            if ((uint32_t)__builtin_amdgcn_readlane((int)wave_scan_add(m2), 63) <= 64u) C = C2;
                        }
                        const uint32_t trips_now = (uint32_t)(((float)(widest + per - 1u) + 0.5f) * __builtin_amdgcn_rcpf((float)per));
                        if (__builtin_amdgcn_readfirstlane((int)(trips_now >= C + kBalGain ? 1u : 0u))) {     // (the same in every lane: keep the branch scalar)
                            balanced = true;
                            const uint32_t mk = lane < nsurv ? (uint32_t)(((float)(Lk + C - 1u) + 0.5f) * __builtin_amdgcn_rcpf((float)C)) : 0u;   // lanes for survivor `lane`
                            const uint32_t inc = wave_scan_add(mk), start = inc - mk;
                            // owner of lane l = the last survivor whose first lane is <= l: marks at the first lanes, running maximum
                            sC[wave][lane] = 0;                                    // (the alive list is done with: its bytes carry the marks)
                            __builtin_amdgcn_wave_barrier();
                            if (lane < nsurv) sC[wave][start] = (uint8_t)(lane + 1u);
                            __builtin_amdgcn_wave_barrier();
                            const uint32_t owner = wave_scan_max((uint32_t)sC[wave][lane]) - 1u;
                            const uint32_t packed = (uint32_t)sS[wave][min(lane, nsurv - 1u)] | (start << 8) | (mk << 16);   // of survivor `lane`
                            b_mine = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)packed);
                            b_used = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                        }
                    }
                    for (uint32_t sbase = 0; sbase < nsurv; sbase += group) {      // (balanced: one round)
                        const uint32_t sidx = sbase + q;
                        const bool act = balanced ? lane < b_used : (q < group && sidx < nsurv);
                        VF_RC({ const uint32_t na = (uint32_t)__popcll(__ballot(act)); if (lane == 0) { RC.iters++; RC.act += na; } })
                        if (act) {
                            const uint32_t code = balanced ? (b_mine & 0xFFu) : (uint32_t)sS[wave][sidx];
                            const uint32_t my_sub = balanced ? lane - ((b_mine >> 8) & 0xFFu) : sub, my_n = balanced ? b_mine >> 16 : per;
                            const uint32_t cell = code >> 1, odd = code & 1u;
                            const uint32_t lj = cell >> 3, li = cell & 7u;
                            const uint32_t va = lj * kBlockVerts + li, vb = va + 1, vc = va + kBlockVerts, vd = vc + 1;
                            const uint32_t v0 = odd ? vb : va, v1 = vc, v2 = odd ? vd : vb;
                            const uint32_t prim = 2u * ((j0 + lj) * P.nm1 + (i0 + li)) + odd;
                            raster_fast<GROUPS>(T, prim + 1u, sXY[wave], v0 | (v1 << 8) | (v2 << 16), (int32_t)my_sub, (int32_t)my_n, lane - my_sub VF_RC(, RC));
                        }
                        if (balanced) break;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                VF_PH(4)
            }
            // ---- completion: the DS queue of a wave is FIFO, so this decrement is ordered after the block's paints ----
            uint32_t old = 0;
            if (lane == 0) old = atomicSub(&s_pending[stepidx], 1u);
            old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
            if (old != 1u) { VF_PH(5) continue; }
            // the step is complete: try to advance the frontier (first step that still has unfinished blocks)
            uint32_t got = 0;
            if (lane == 0) got = atomicCAS(&s_lock, 0u, 1u) == 0u ? 1u : 0u;
            got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            if (!got) { VF_PH(5) continue; }                           // somebody else is publishing; masks may lag, never lie
            if (GROUPS) __builtin_amdgcn_s_setprio(3);   // the publisher of new final-pixel masks: before everything else (above)
            uint32_t fr = lds_peek(&s_frontier);
            while (fr < nsteps && lds_peek(&s_pending[fr]) == 0u) ++fr;
            const uint32_t pub = lds_peek(&s_published);
            if (fr > pub && (fr - pub >= (uint32_t)kRescanEvery || fr == nsteps)) {
                // steps 0 .. fr-1 are complete: everything owned by ids >= first id of step fr-1 is final
                const int32_t sw = T.px_hi - T.px_lo + 1, sh = T.py_hi - T.py_lo + 1;
                const uint32_t nfinal = sw <= 16 ? rescan_strip(s_vis, s_colfin, s_rowfin, lane, s_firstid[fr - 1], sw, sh)
                                                 : rescan_open_rows(s_vis, s_colfin, s_rowfin, lane, s_firstid[fr - 1], row_full, sh);
                if constexpr (GROUPS) {
                    __builtin_amdgcn_wave_barrier();                       // (this wave's mask stores are in the DS queue: the loads below come behind them)
                    refresh_fin4(s_colfin, s_rowfin, s_colfin4, s_rowfin4, lane);
                }
                if (lane == 0) { s_published = fr; if (nfinal >= tile_pixels) { s_done = 1u; atomicOr(&s_next, 0x40000000u); } }
            }
            if (lane == 0) { s_frontier = fr; __threadfence_block(); atomicExch(&s_lock, 0u); }
            __builtin_amdgcn_s_setprio(0);
            VF_PH(5)
        }
        VF_PH(1)
        __builtin_amdgcn_s_setprio(0);                                 // (the last block's priority ends with the block loop)
        if (lane == 0 && my_blocks) atomicAdd(&s_blocks, my_blocks);
        __syncthreads();
        if (kDiagItem == 1) dbg_loop += (uint32_t)(__builtin_amdgcn_s_memrealtime() - dbg_t0);
        if (s_done) break;                                             // uniform
        const uint32_t all = publish_chunk_masks<GROUPS>(L, nsteps, lane, wave);
        VF_PH(6)
        if (all >= tile_pixels) break;                                 // uniform: the whole tile is final
    }
    __syncthreads();
    VF_PH(6)
    if (stats && tid == 0) {
        atomicAdd(&stats[0], s_blocks);
        // (diagnostics builds, VF_DIAG_ITEM: 1 ticks inside the block loops, 2 the plan's weight of the item, 3 when the item started --
        //  10 ns ticks of the 100 MHz clock, tools/exp_gantt.py -- instead of the blocks drawn)
        stats[4 + 4 * item_idx] = item;
        stats[5 + 4 * item_idx] = kDiagItem == 1 ? dbg_loop : kDiagItem == 2 ? work[item_idx].y : kDiagItem == 3 ? (uint32_t)t_start : s_blocks;
        stats[6 + 4 * item_idx] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_start);       // raster phase, 10 ns ticks
    }
    shade_item<WRITE_VIS, COMPLETE, FAST>(P, V, S, L, T, tp, tile_x0, rgba, vis_out, tid);
    VF_PH(7)
    if (tid == 0) {
        const uint32_t ticks = (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_start);
        atomicAdd(&last_blocks[tile], max(ticks, 1u));        // feedback for the next frame's plan: time this tile cost (10 ns ticks)
        if (!COMPLETE)                                        // ... and this piece of it (behind the tile times and the quantum word)
            last_blocks[(size_t)P.ntx * P.nty + 1u + (size_t)tile * 64u + work_part(item)] = max(ticks, 1u);
        if (stats) stats[7 + 4 * item_idx] = ticks;        // raster + fragment phase
        if (!COMPLETE && s_redo) redo[atomicAdd(redo_count, 1u)] = item_idx;   // this item met a primitive the fast path skips
    }
    __syncthreads();                                       // the next item re-initialises the tile state
    if constexpr (COMPLETE) {
        redo_at += gridDim.x;
        if (redo_at >= *redo_count) break;
    }
    else {
        if (tid == 0) s_item = atomicAdd(work_count + 2, 1u);
        __syncthreads();
        pulled = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_item);
        if (pulled >= nwork) break;
    }
    }
#ifdef VF_PHASE_PROF   // one flush per workgroup, after its last item (per-item flushes would perturb the hand-over they measure)
    if (stats) {
        unsigned long long *ph = reinterpret_cast<unsigned long long *>(stats + 4 + 4 * ((size_t)P.ntx * P.nty + kSplitBudget));
        if (lane == 0) {
            for (int p = 0; p < 8; ++p) atomicAdd(&ph[p], (unsigned long long)PH.acc[p]);
            for (int p = 8; p < 16; ++p) atomicAdd(&ph[8 + p], (unsigned long long)PH.acc[p]);
            atomicAdd(&ph[30], (unsigned long long)PH.acc[16]); atomicAdd(&ph[31], (unsigned long long)PH.acc[17]);
            atomicAdd(&ph[32], (unsigned long long)PH.acc[18]); atomicAdd(&ph[33], (unsigned long long)PH.acc[19]);
            atomicAdd(&ph[8], (unsigned long long)RC.nsurv); atomicAdd(&ph[9], (unsigned long long)RC.iters);
            atomicAdd(&ph[10], (unsigned long long)RC.empty); atomicAdd(&ph[15], (unsigned long long)RC.live);
            atomicAdd(&ph[35], (unsigned long long)RC.alive); atomicAdd(&ph[36], (unsigned long long)RC.apass); atomicAdd(&ph[37], (unsigned long long)RC.act);
        }
        {   // lanes in the line loop's first step (group tests with line groups, line trips without)
            uint32_t li = RC.l_iter;
            for (int o = 32; o > 0; o >>= 1) li += __shfl_xor(li, o);
            if (lane == 0) atomicAdd(&ph[34], (unsigned long long)li);
        }
        uint32_t rcs[4] = { RC.lines, RC.solved, RC.painted, RC.trips };   // (paint_lines gave way to: lines with an open pixel in their bounding range)
        for (int c = 0; c < 4; ++c) {
            for (int o = 32; o > 0; o >>= 1) rcs[c] += __shfl_xor(rcs[c], o);
            if (lane == 0) atomicAdd(&ph[11 + c], (unsigned long long)rcs[c]);
        }
        uint32_t wcs[6] = { RC.w_iter, RC.w_s1, RC.w_s2, RC.w_paint, RC.w_cls, RC.c_reach };   // wave-level executions of the line loop's parts
        for (int c = 0; c < 6; ++c) {
            for (int o = 32; o > 0; o >>= 1) wcs[c] += __shfl_xor(wcs[c], o);
            if (lane == 0) atomicAdd(&ph[24 + c], (unsigned long long)wcs[c]);
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Diagnostics: the fragment stage as a launch of its own (vf_terrain_debug_fragment_stage).  Visibility (H, W) u32 in HBM ->
// RGBA8 through the same shade_pixel the tile kernel runs on its LDS tile: fs_main + sRGB store
// (src/shaders/terrain.wgsl:69-91).  Streaming: 4 B read + 4 B written per pixel plus the records of the visible primitives.
//
// What bounds it is neither arithmetic nor HBM bandwidth but the L1's miss handling (rocprofv3 --pmc on the row-segment form
// this replaces, fill camera: 52 distinct cache lines per 64-lane record gather, 64 % of them L1 misses, the L1 stalled on
// pending L2 data 65 % of the launch, 344 cycles per L2 round trip -- and halving the instruction count moved the time by 4 %).
// A pixel's three vertex records sit in two 144-byte vertex rows of its block; pixels next to each other ALONG A ROW share
// few of them on a noise terrain (the visible primitive jumps between cell rows), pixels in a small 2-D neighbourhood share
// many.  So a wave takes an 8 x 8 pixel tile (four of them, side by side, per workgroup: every visibility / RGBA row segment
// of the 32 x 8 region is one whole 128-byte line), and the regions are dealt to the workgroups so that each XCD -- each L2 --
// owns a contiguous band of region columns: vertically and horizontally adjacent regions meet in the same L2.
// Persistent workgroups, tables staged once, the next region's visibility words requested before the current one is shaded.
// ---------------------------------------------------------------------------------------------
// Which region a workgroup takes next.  Workgroups are dealt to the XCDs round-robin (workgroup b runs on XCD b % 8, each XCD with its
// own L2): the regions are grouped into super-tiles of kSuper x kSuper regions, super-tile (sx, sy) belongs to XCD (sx + 3 sy) % 8 --
// every XCD gets an eighth of every part of the frame (a terrain that covers the middle of the picture loads them all alike) -- and
// the workgroups of an XCD walk its super-tiles region by region, so that neighbouring regions are shaded at the same time from the
// same L2.  Frames whose region columns do not divide into 8 super-tile columns are walked in plain row-major order.
struct RegionWalk {
    static constexpr uint32_t kSuper = 4;
    uint32_t nrx, nry, per_row, xcd, stride, k, total;   // per_row: super-tiles of one XCD per super-tile row (0: plain walk)
    __device__ __forceinline__ void init(uint32_t nrx_, uint32_t nry_)
    {
        nrx = nrx_; nry = nry_;
        const bool banded = nrx % (8u * kSuper) == 0u && gridDim.x % 8u == 0u;
        per_row = banded ? nrx / (8u * kSuper) : 0u;
        xcd = banded ? blockIdx.x % 8u : 0u;
        stride = banded ? gridDim.x / 8u : gridDim.x;
        k = banded ? blockIdx.x / 8u : blockIdx.x;
        total = banded ? per_row * ((nry + kSuper - 1u) / kSuper) * kSuper * kSuper : nrx * nry;
    }
    __device__ __forceinline__ bool valid() const { return k < total; }
    // (rx, ry) of sequence number kk; ry may lie beyond nry in the last super-tile row (the callers' bounds checks skip those)
    __device__ __forceinline__ void at(uint32_t kk, uint32_t &rx, uint32_t &ry) const
    {
        if (per_row == 0u) { ry = kk / nrx; rx = kk - ry * nrx; return; }
        const uint32_t st = kk / (kSuper * kSuper), in = kk % (kSuper * kSuper);
        const uint32_t sy = st / per_row, m = st - sy * per_row;
        const uint32_t sx = ((xcd + 8u * 3u - (3u * sy) % 8u) % 8u) + 8u * m;      // (sx + 3 sy) % 8 == xcd
        rx = sx * kSuper + in % kSuper; ry = sy * kSuper + in / kSuper;
    }
};

template <bool CLIPPED, bool FAST>
__global__ __launch_bounds__(256) void k_resolve(FrameParams P, SetupView V, const float *__restrict__ lut_linear,
                                                 const float *__restrict__ thresh, const uint32_t *__restrict__ vis,
                                                 uint32_t *__restrict__ rgba, uint32_t *__restrict__ covered)
{
    __shared__ __attribute__((aligned(16))) float s_lut[kLutFloats];
    __shared__ float s_thr[256];
    for (int k = threadIdx.x; k < kLutFloats; k += 256) s_lut[k] = lut_linear[k];
    s_thr[threadIdx.x] = thresh[threadIdx.x];
    __syncthreads();
    const ShadeTables S = { s_lut, s_thr };
    // lane -> pixel of the workgroup's 32 x 8 region: wave w holds the 8 x 8 tile at x = 8 w
    const uint32_t lx = (threadIdx.x >> 6) * 8u + (threadIdx.x & 7u), ly = (threadIdx.x >> 3) & 7u;
    RegionWalk R;
    R.init((P.W + 31u) / 32u, (P.H + 7u) / 8u);
    auto fetch = [&](uint32_t kk) -> uint32_t {
        uint32_t rx, ry;
        R.at(kk, rx, ry);
        const uint32_t px = rx * 32u + lx, py = ry * 8u + ly;
        return px < P.W && py < P.H ? vis[(size_t)py * P.W + px] : 0u;
    };
    uint32_t ncov = 0;
    uint32_t id_next = R.valid() ? fetch(R.k) : 0u;
    while (R.valid()) {
        const uint32_t id = id_next;
        const uint32_t kn = R.k + R.stride;
        if (kn < R.total) id_next = fetch(kn);
        uint32_t rx, ry;
        R.at(R.k, rx, ry);
        const uint32_t px = rx * 32u + lx, py = ry * 8u + ly;
        if (px < P.W && py < P.H) rgba[(size_t)py * P.W + px] = id ? shade_pixel<CLIPPED, FAST>(P, V, S, id - 1u, (int32_t)px, (int32_t)py) : P.clear_rgba;
        ncov += (uint32_t)__popcll(__ballot(id != 0u));
        R.k = kn;
    }
    if (covered && (threadIdx.x & 63u) == 0u && ncov) atomicAdd(covered, ncov);
}

// The same for rows of a multiple of four pixels (and no clipped primitives in the frame): the HBM side moves 16 bytes per lane -- a lane
// loads the visibility words of four consecutive pixels of a row and stores their four colours, a wave covers 16 x 16 pixels, the
// workgroup's four waves a 32 x 32 region -- while the shading still runs on compact 8 x 8 tiles: a wave that holds any covered
// pixel passes its 256 words through LDS (row pitch 20 words: 16-byte aligned rows, at most two-way bank conflicts) and shades the four
// 8 x 8 quarters of its area one pixel per lane, as above.  Background regions (84 % of C4's default frame) cost one 16-byte
// load and one 16-byte store per lane; covered ones keep the 2-D locality that the L1 needs.
template <bool FAST>
__global__ __launch_bounds__(256) void k_resolve4(FrameParams P, SetupView V, const float *__restrict__ lut_linear,
                                                  const float *__restrict__ thresh, const uint4 *__restrict__ vis,
                                                  uint4 *__restrict__ rgba, uint32_t *__restrict__ covered)
{
    constexpr uint32_t kPitch = 20;
    __shared__ __attribute__((aligned(16))) float s_lut[kLutFloats];
    __shared__ float s_thr[256];
    __shared__ __attribute__((aligned(16))) uint32_t s_px[4][16 * kPitch];
    for (int k = threadIdx.x; k < kLutFloats; k += 256) s_lut[k] = lut_linear[k];
    s_thr[threadIdx.x] = thresh[threadIdx.x];
    __syncthreads();
    const ShadeTables S = { s_lut, s_thr };
    const uint32_t W4 = P.W / 4u;
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // HBM side: lane -> quad (lane & 3) of row (lane >> 2) of the wave's 16 x 16 area; the waves sit 2 x 2 in the region
    const uint32_t lq = (wv & 1u) * 4u + (lane & 3u), ly = (wv >> 1) * 16u + (lane >> 2);
    uint32_t *const sw = s_px[wv];
    uint32_t *const mine4 = sw + (lane >> 2) * kPitch + (lane & 3u) * 4u;
    RegionWalk R;
    R.init((W4 + 7u) / 8u, (P.H + 31u) / 32u);
    auto fetch = [&](uint32_t kk) -> uint4 {
        uint32_t rx, ry;
        R.at(kk, rx, ry);
        const uint32_t q = rx * 8u + lq, py = ry * 32u + ly;
        if (!(q < W4 && py < P.H)) return make_uint4(0u, 0u, 0u, 0u);
        return vis[(size_t)py * W4 + q];
    };
    uint32_t ncov = 0;
    // visibility words are requested kAhead regions ahead: with the few waves per CU that suit the record gathers (below), one
    // 16-byte load in flight per lane would leave HBM idle (bytes in flight = bandwidth x latency)
    constexpr int kAhead = 3;
    uint4 ring[kAhead];
#pragma unroll
    for (int a = 0; a < kAhead; ++a) ring[a] = R.k + (uint32_t)a * R.stride < R.total ? fetch(R.k + (uint32_t)a * R.stride) : make_uint4(0u, 0u, 0u, 0u);
    while (R.valid()) {
        const uint4 id = ring[0];
        const uint32_t kn = R.k + R.stride, kf = R.k + (uint32_t)kAhead * R.stride;
#pragma unroll
        for (int a = 0; a + 1 < kAhead; ++a) ring[a] = ring[a + 1];
        if (kf < R.total) ring[kAhead - 1] = fetch(kf);
        uint32_t rx, ry;
        R.at(R.k, rx, ry);
        const uint32_t q = rx * 8u + lq, py = ry * 32u + ly;
        uint4 out = make_uint4(P.clear_rgba, P.clear_rgba, P.clear_rgba, P.clear_rgba);
        if (__ballot((id.x | id.y | id.z | id.w) != 0u) != 0ull) {            // (wave-uniform)
            *reinterpret_cast<uint4 *>(mine4) = id;
            __builtin_amdgcn_wave_barrier();               // LDS operations of one wave complete in order; keep the compiler from reordering
            const int32_t ax = (int32_t)((rx * 8u + (wv & 1u) * 4u) * 4u), ay = (int32_t)(ry * 32u + (wv >> 1) * 16u);   // the wave's area
#pragma unroll 1
            for (uint32_t t8 = 0; t8 < 4u; ++t8) {         // its four 8 x 8 quarters, one pixel per lane
                const uint32_t sx = (t8 & 1u) * 8u + (lane & 7u), sy = (t8 >> 1) * 8u + (lane >> 3);
                uint32_t *const w = sw + sy * kPitch + sx;
                const uint32_t pid = *w;
                *w = pid ? shade_pixel<false, FAST>(P, V, S, pid - 1u, ax + (int32_t)sx, ay + (int32_t)sy) : P.clear_rgba;
            }
            __builtin_amdgcn_wave_barrier();
            out = *reinterpret_cast<const uint4 *>(mine4);
            __builtin_amdgcn_wave_barrier();               // (the next region's words go to the same place)
        }
        if (q < W4 && py < P.H) {
            rgba[(size_t)py * W4 + q] = out;
        }
        ncov += (id.x ? 1u : 0u) + (id.y ? 1u : 0u) + (id.z ? 1u : 0u) + (id.w ? 1u : 0u);
        R.k = kn;
    }
    for (int o = 32; o > 0; o >>= 1) ncov += __shfl_xor(ncov, o);
    if (covered && (threadIdx.x & 63u) == 0u && ncov) atomicAdd(covered, ncov);
}

// ---------------------------------------------------------------------------------------------
// grid_generate: make_grid src/terrain/mesh.rs:35-90 (bit-exact: no FMA contraction, IEEE divide)
// ---------------------------------------------------------------------------------------------
__global__ void k_grid_vertices(uint32_t w, uint32_t h, float dx, float dy, float2 *__restrict__ xy, float2 *__restrict__ uv)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (size_t)w * h) return;
    uint32_t y = (uint32_t)(k / w), x = (uint32_t)(k - (size_t)y * w);
    float cx = ((float)w - 1.0f) * 0.5f * dx;            // :46
    float cy = ((float)h - 1.0f) * 0.5f * dy;            // :47
    float wx = (float)x * dx - cx;                       // :53
    float wy = (float)y * dy - cy;                       // :50
    float u = (float)x / ((float)w - 1.0f);              // :54
    float v = (float)y / ((float)h - 1.0f);              // :51
    xy[k] = make_float2(wx, wy);
    uv[k] = make_float2(u, v);
}
__global__ void k_grid_indices(uint32_t w, uint32_t h, uint32_t *__restrict__ idx)
{
    size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t ncell = (size_t)(w - 1) * (h - 1);
    if (c >= ncell) return;
    uint32_t y = (uint32_t)(c / (w - 1)), x = (uint32_t)(c - (size_t)y * (w - 1));
    uint32_t i0 = y * w + x, i1 = i0 + 1, i2 = i0 + w, i3 = i2 + 1;   // :64-73
    uint2 *o = reinterpret_cast<uint2 *>(idx + 6 * c);                 // 24-byte records are 8-byte aligned
    o[0] = make_uint2(i0, i1); o[1] = make_uint2(i2, i2); o[2] = make_uint2(i1, i3);
}

// ---------------------------------------------------------------------------------------------
// triangle smoke path (src/lib.rs:72-91, src/shaders/triangle.wgsl): one primitive, per-pixel test
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_triangle(uint32_t W, uint32_t H, const float *__restrict__ thresh, uint32_t *__restrict__ rgba)
{
    __shared__ float s_thr[256];
    s_thr[threadIdx.x] = thresh[threadIdx.x];
    __syncthreads();
    VF_RESERVE_VGPR(16);   // 17 registers, not 16: see vf_device.h (the int64 -> float conversions below shift by a register amount)
    size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (size_t)W * H) return;
    int32_t py = (int32_t)(p / W), px = (int32_t)(p - (size_t)py * W);
    GVert v[3] = {
        { -0.8f, -0.8f, 0.0f, 1.0f, { 1.0f, 0.2f, 0.2f } },
        { 0.8f, -0.8f, 0.0f, 1.0f, { 0.2f, 1.0f, 0.2f } },
        { 0.0f, 0.8f, 0.0f, 1.0f, { 0.2f, 0.2f, 1.0f } },
    };
    uint32_t out = 0xFFFFFFFFu;   // clear WHITE (src/lib.rs:19)
    TriSetup T;
    int64_t e[3];
    if (setup_triangle(v[0], v[1], v[2], 0.5f * (float)W, 0.5f * (float)H, W, H, T) && covers(T, px, py, e)) {
        float attr[3];
        interpolate(T, e, attr);
        out = 0xFF000000u;
        for (int ch = 0; ch < 3; ++ch) out |= srgb_encode(attr[ch], s_thr) << (8 * ch);
    }
    rgba[p] = out;
}

// ---------------------------------------------------------------------------------------------
// Renderer DEM path: add_terrain / terrain_stats / normalize_terrain on HBM-resident heights.
// All four kernels are streaming and HBM-bound (4 B read [+ 4 B written] per sample).
// ---------------------------------------------------------------------------------------------
// add_terrain ingest (src/lib.rs:351-388): heights[k] = (f32)src[k] * exaggeration
template <typename T>
__global__ void k_dem_ingest(const T *__restrict__ src, float *__restrict__ dst, size_t n, float exaggeration)
{
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x)
        dst[k] = (float)src[k] * exaggeration;
}

// pass 1 of dem_stats_from_slice (src/lib.rs:905-932): min, max, sum.  The reference adds in f32 in index order; a
// parallel sum cannot reproduce that rounding, so the sum is carried in FP64 (closer to the true mean) -- DESIGN.md.
// out: [0] min bits (ordered-int trick), [1] max bits, then one double per block in `partial`.
__device__ __forceinline__ uint32_t float_order(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float float_unorder(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

__global__ __launch_bounds__(256) void k_dem_minmaxsum(const float *__restrict__ h, size_t n, uint32_t *__restrict__ mm, double *__restrict__ partial)
{
    __shared__ double s_sum[4];
    __shared__ uint32_t s_lo[4], s_hi[4];
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    double sum = 0.0;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
        const float v = h[k];
        const uint32_t o = float_order(v);
        if (v == v) { lo = min(lo, o); hi = max(hi, o); }     // NaN never wins a `<` / `>` comparison in the reference loop either
        sum += (double)v;
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, o)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, o));
        sum += __shfl_xor(sum, o);
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; s_sum[threadIdx.x >> 6] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&mm[0], min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3])));
        atomicMax(&mm[1], max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3])));
        partial[blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
    }
}
// pass 2: sum of squared deviations from the (f32) mean, as the reference forms them: diff = h - mean (f32), diff*diff
__global__ __launch_bounds__(256) void k_dem_sqdev(const float *__restrict__ h, size_t n, float mean, double *__restrict__ partial)
{
    __shared__ double s_sum[4];
    double sum = 0.0;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
        const float d = h[k] - mean;
        sum += (double)(d * d);
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
}
// normalize_in_place (src/lib.rs:934-951): v = (v - a) * scale + lo   |   v = (v - a) / denom
__global__ void k_dem_normalize(float *__restrict__ h, size_t n, int zscore, float a, float scale_or_denom, float lo)
{
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
        const float v = h[k];
        h[k] = zscore ? (v - a) / scale_or_denom : (v - a) * scale_or_denom + lo;
    }
}
// stride sample for the 1-99 percentile clamp of terrain_stats::min_max (src/terrain_stats.rs:24-29)
__global__ void k_dem_sample(const float *__restrict__ h, size_t n, size_t step, float *__restrict__ out, size_t nout)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nout && k * step < n) out[k] = h[k * step];
}

// ---------------------------------------------------------------------------------------------
// multi-GPU: [nranks][local_rows][W] rank-major gather buffer -> (H, W) image
// ---------------------------------------------------------------------------------------------
__global__ void k_stitch_bands(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t row_vec4, uint32_t H,
                               uint32_t nranks, uint32_t band_shift, uint32_t band_h, uint32_t local_rows)
{
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (size_t)H * row_vec4) return;
    uint32_t y = (uint32_t)(p / row_vec4), xq = (uint32_t)(p - (size_t)y * row_vec4);
    uint32_t b = y >> band_shift, r = b % nranks;
    uint32_t ly = ((b / nranks) << band_shift) + (y & (band_h - 1u));
    dst[p] = src[((size_t)r * local_rows + ly) * row_vec4 + xq];
}

// ---------------------------------------------------------------------------------------------
// render_png read-back (src/terrain/mod.rs:439-490): PNG scanline filtering on the device, so that the host only deflates.
// One workgroup per row: pass 1 sums |signed residual| for the five PNG filters (None, Sub, Up, Average, Paeth), the
// smallest wins (first on ties -- the same rule as the host encoder), pass 2 writes filter byte + residuals.  HBM-bound:
// the row, its upper neighbour and the output once each.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t png_paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (uint32_t)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}
__device__ __forceinline__ uint32_t png_residual(int f, uint32_t x, uint32_t a, uint32_t b, uint32_t c)   // one byte
{
    switch (f) {
    case 0: return x;
    case 1: return (x - a) & 255u;
    case 2: return (x - b) & 255u;
    case 3: return (x - ((a + b) >> 1)) & 255u;
    default: return (x - png_paeth((int)a, (int)b, (int)c)) & 255u;
    }
}
__global__ __launch_bounds__(256) void k_png_filter(const uint32_t *__restrict__ rgba, uint32_t W, uint8_t *__restrict__ out)
{
    __shared__ uint32_t s_sum[5][4];
    __shared__ uint32_t s_best;
    const uint32_t y = blockIdx.x, tid = threadIdx.x;
    const uint32_t *cur = rgba + (size_t)y * W, *up = y ? cur - W : nullptr;
    uint32_t sums[5] = { 0, 0, 0, 0, 0 };
    for (uint32_t x = tid; x < W; x += 256) {
        const uint32_t px = cur[x], pa = x ? cur[x - 1] : 0u, pb = up ? up[x] : 0u, pc = (up && x) ? up[x - 1] : 0u;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const uint32_t v = (px >> (8 * ch)) & 255u, a = (pa >> (8 * ch)) & 255u, b = (pb >> (8 * ch)) & 255u, c = (pc >> (8 * ch)) & 255u;
#pragma unroll
            for (int f = 0; f < 5; ++f) { const uint32_t r = png_residual(f, v, a, b, c); sums[f] += r < 128u ? r : 256u - r; }
        }
    }
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        uint32_t v = sums[f];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((tid & 63u) == 0) s_sum[f][tid >> 6] = v;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t best = 0, best_sum = 0xFFFFFFFFu;          // row sums stay below 2^32: 4 * 65536 * 128 at the widest frame
        for (int f = 0; f < 5; ++f) {
            const uint32_t v = s_sum[f][0] + s_sum[f][1] + s_sum[f][2] + s_sum[f][3];
            if (v < best_sum) { best_sum = v; best = (uint32_t)f; }
        }
        s_best = best;
    }
    __syncthreads();
    const int f = (int)s_best;
    uint8_t *dst = out + (size_t)y * ((size_t)W * 4 + 1);
    if (tid == 0) dst[0] = (uint8_t)f;
    for (uint32_t x = tid; x < W; x += 256) {
        const uint32_t px = cur[x], pa = x ? cur[x - 1] : 0u, pb = up ? up[x] : 0u, pc = (up && x) ? up[x - 1] : 0u;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
            dst[1 + 4 * (size_t)x + ch] = (uint8_t)png_residual(f, (px >> (8 * ch)) & 255u, (pa >> (8 * ch)) & 255u, (pb >> (8 * ch)) & 255u,
                                                               (pc >> (8 * ch)) & 255u);
    }
}

// Device -> page-locked host memory by stores (16 bytes per lane, whole 64-byte lines per quarter wave), for read-backs whose
// destination the device can address (vf_host_alloc / hipHostRegister memory): the copy engine's first transfer of a process costs
// 8-10 ms on this runtime, a kernel's stores cost the PCIe time only (round 5, tools/exp_second_call.py).  n16: 16-byte words.
__global__ __launch_bounds__(256) void k_copy_to_host(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16, size_t nbytes)
{
    const size_t stride = (size_t)gridDim.x * 256u;
    for (size_t k = (size_t)blockIdx.x * 256u + threadIdx.x; k < n16; k += stride) dst[k] = src[k];
    if (blockIdx.x == 0 && threadIdx.x < (nbytes & 15u))
        reinterpret_cast<uint8_t *>(dst)[n16 * 16u + threadIdx.x] = reinterpret_cast<const uint8_t *>(src)[n16 * 16u + threadIdx.x];
}

// multi-GPU, tile shards: [nranks][stride_tiles][64][64] rank-major gather buffer -> (H, W) image.  Tile (tx, ty) belongs
// to rank (tx + skew * ty) % nranks; a rank numbers its tiles row-major.  Persistent: a few hundred workgroups walk the screen tiles
// with a grid stride, 16 bytes per lane where the frame allows it.  Few workgroups on purpose: on rank 0 this copy of the whole frame
// runs on a side stream BESIDE the next frame's tile kernel, whose 1024-thread workgroups need most of a CU to start -- a launch of
// one workgroup per tile fills every CU's wave slots and holds them back until it has drained (a rank of eight, emulated: frame
// period +80 us instead of the +35 the copy is worth).
// tiles of the stripes a rank owns in one tile row that lie left of column `tx_end`: the owned stripes are g = want, want + nranks, ...
// (g = tx >> shift), each 1 << shift tiles wide
__device__ __forceinline__ uint32_t owned_left_of(uint32_t tx_end, uint32_t want, uint32_t nranks, uint32_t shift)
{
    const uint32_t g_end = tx_end >> shift, rem = tx_end & ((1u << shift) - 1u);
    const uint32_t full = g_end > want ? (g_end - 1u - want) / nranks + 1u : 0u;          // owned stripes wholly left of tx_end
    return (full << shift) + (g_end % nranks == want ? rem : 0u);
}
// ... and under a registered stripe map (`owner` per column stripe)
__device__ __forceinline__ uint32_t owned_left_of_map(const uint8_t *__restrict__ owner, uint32_t tx_end, uint32_t r, uint32_t shift)
{
    uint32_t n = 0;
    for (uint32_t g = 0; (g << shift) < tx_end; ++g)
        if (owner[g] == r) n += min((g + 1u) << shift, tx_end) - (g << shift);
    return n;
}
__global__ __launch_bounds__(256) void k_stitch_tiles(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, uint32_t W, uint32_t H,
                                                      uint32_t ntx, uint32_t nty, uint32_t nranks, uint32_t skew, uint32_t shift, uint32_t stride_tiles,
                                                      const uint8_t *__restrict__ owner)
{
    __shared__ uint32_t s_local;
    const bool wide = (W & 3u) == 0u && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0u;
    for (uint32_t tile = blockIdx.x; tile < ntx * nty; tile += gridDim.x) {
        const uint32_t tx = tile % ntx, ty = tile / ntx;
        const uint32_t r = owner ? owner[tx >> shift] : ((tx >> shift) + skew * ty) % nranks;
        __syncthreads();                                   // (the previous tile's s_local has been read)
        if (threadIdx.x == 0) {
            uint32_t before = 0;                           // tiles of rank r in the rows above
            if (owner) s_local = ty * owned_left_of_map(owner, ntx, r, shift) + owned_left_of_map(owner, tx, r, shift);
            else {
            for (uint32_t t = 0; t < ty; ++t) before += owned_left_of(ntx, (r + nranks - (skew * t) % nranks) % nranks, nranks, shift);
            s_local = before + owned_left_of(tx, (r + nranks - (skew * ty) % nranks) % nranks, nranks, shift);
            }
        }
        __syncthreads();
        const uint32_t *tp = src + ((size_t)r * stride_tiles + s_local) * (kTileW * kTileH);
        const uint32_t x0 = tx * kTileW, y0 = ty * kTileH;
        if (wide && x0 + kTileW <= W) {                    // whole-width tile: 16 lanes x 16 bytes per row
            const uint4 *t4 = reinterpret_cast<const uint4 *>(tp);
            for (uint32_t k = threadIdx.x; k < (uint32_t)(kTileW / 4 * kTileH); k += 256) {
                const uint32_t lq = k % (kTileW / 4), ly = k / (kTileW / 4);
                if (y0 + ly < H) *reinterpret_cast<uint4 *>(dst + (size_t)(y0 + ly) * W + x0 + 4u * lq) = t4[k];
            }
        } else {
            for (uint32_t k = threadIdx.x; k < (uint32_t)(kTileW * kTileH); k += 256) {
                const uint32_t lx = k % kTileW, ly = k / kTileW;
                if (x0 + lx < W && y0 + ly < H) dst[(size_t)(y0 + ly) * W + x0 + lx] = tp[k];
            }
        }
    }
}

} // namespace vf
