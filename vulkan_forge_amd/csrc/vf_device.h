// vf_device.h -- device-side building blocks of the gfx950 terrain rasteriser.
//
// Numerics contract (DESIGN.md "Raster conventions"): every float operation below is an IEEE-754
// binary32 add/mul/div/sqrt/fma in a fixed order (compiled with -ffp-contract=off; HIP's default
// correctly-rounded divide/sqrt), so results are reproducible bit for bit on any conforming
// implementation.  sin/cos use a fixed Cody-Waite + minimax-polynomial algorithm instead of the
// hardware transcendental units for the same reason.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// MI355X executes v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64 wrongly when the 32-bit shift amount sits in the LAST register of
// the wave's VGPR allocation (v15 of 16, v23 of 24, ...): ~2 % of the lanes get a result shifted by a wrong amount
// (tools/probes/shift64_top.hip, profiles/r04_hw_shift64_probe.log).  The compiler does not avoid that form, so it can appear in any
// kernel whose register count is a multiple of the granule (8).  tools/isa_lint.py finds it in the built library and build() refuses
// such a library; where it does appear, naming the next register as clobbered moves the count off the multiple at no cost in
// occupancy for a small kernel:  VF_RESERVE_VGPR(16)  in a kernel that would otherwise use exactly v0..v15.
#define VF_RESERVE_VGPR(n) asm volatile("" ::: "v" #n)

namespace vf {

// ---- constants shared by the kernels -------------------------------------------------------
constexpr int kBlockCells = 8;                  // grid block = 8 x 8 cells (128 primitives) = the work item of one wave
constexpr int kBlockVerts = kBlockCells + 1;    // 9 x 9 vertices incl. the shared edges
constexpr int kTileW = 64, kTileH = 64;         // screen tile held in LDS (power of two, <= 64 each: masks are 64-bit)
// strip splitting quantum, in quarters of the even per-item share of last frame's work (kTargetItems items = 4 per CU).  A tile stays
// whole up to two quanta, i.e. up to kSplitQuantumX4 / 8 of a CU's even share of the frame: at 8 the heaviest whole tile alone
// would take as long as a perfectly balanced frame, beyond it sets the frame time (C4: 6 -> 1.273 ms, 8 -> 1.258, 10 -> 1.45)
constexpr uint32_t kSplitQuantumX4 = 7;
// Minimum waves per SIMD the tile kernel's register allocation leaves room for.  Its 16 waves per CU are 4 per SIMD; asking for 5
// caps it at 96 vector registers, which leaves 128 per SIMD free: the set-up kernel of the NEXT frame (side stream, 56 registers)
// can then run on the same CUs under the tile kernel and fill the issue slots it leaves idle, instead of waiting for its tail.
#ifndef VF_TILE_MIN_WAVES
#define VF_TILE_MIN_WAVES 5
#endif
// What a work item reports in its statistics word (timing + statistics enabled): 0 the blocks it drew (the product build),
// 1 / 2 / 3 diagnostics builds for tools/exp_toptiles.py and tools/exp_gantt.py (vf_kernels.h, the end of k_tile's raster phase).
#ifndef VF_DIAG_ITEM
#define VF_DIAG_ITEM 0
#endif
constexpr int kDiagItem = VF_DIAG_ITEM;
constexpr int kMaxTileCols = 256;                 // frame width <= 16384 (vf_terrain_create)
constexpr int kPhaseSlots = 40;                   // u64 diagnostic accumulators behind the per-tile stats (VF_PHASE_PROF builds)
constexpr int kTileThreads = 1024;              // waves per tile workgroup = kTileThreads / 64 (each wave rasterises one block at a time)
constexpr int kFastExtent = 1 << 24;            // fast path: triangle extent < 65536 px (24.8 fixed point)

constexpr uint32_t F_NEAR = 1u, F_FAR = 2u, F_BAD = 4u, F_NOSNAP = 8u;

struct FrameParams {
    float view[16];
    float proj[16];
    float spacing, exag;            // max(u[36],1e-8), u[38]          (terrain.wgsl:46-47)
    float h_range, exposure;        // max(u[37],1e-8), u[35]          (terrain.wgsl:71,85)
    float Lx, Ly, Lz;               // normalize(sun)                   (terrain.wgsl:83)
    float hw, hh;                   // 0.5*W, 0.5*H
    float step;                     // grid pitch 3/(n-1): x_i = -1.5 + i*step (src/terrain/mod.rs:559-567), as k_axis_tables forms it
    uint32_t n, nm1;                // grid vertices per side, cells per side
    uint32_t nb;                    // blocks per side = ceil(nm1 / kBlockCells)
    uint32_t W, H;
    uint32_t ntx, nty;              // screen tiles per row / column
    uint32_t tw, th;
    uint32_t rank, nranks, band_shift, band_h;
    uint32_t local_rows;
    uint32_t shard_tiles;           // 0: whole frame / row bands (local rows, row-major); 1: interleaved tiles (tile-major)
    uint32_t skew;                  // shard_tiles: tile (tx, ty) belongs to rank ((tx >> stripe_shift) + skew * ty) % nranks
    uint32_t stripe_shift;          // log2 of the stripe width in tiles (0: single tile columns)
    const uint32_t *tile_map;       // shard_tiles: local tile -> tx | ty << 16
    const uint8_t *stripe_owner;    // shard_tiles with a registered stripe map (vf_tile_layout_register_map): owner per column stripe; else NULL
    uint32_t clear_rgba;            // packed sRGB8 clear colour
    uint32_t shade_mode;            // 0 REFERENCE (terrain.wgsl as coded), 1 SPEC_T32 (the documented fragment stage)
    const float *tex;               // height texture (SPEC_T32 normals)
    float inv2hr;                   // 1 / (2 h_range): the fast fragment path multiplies where terrain.wgsl:71 divides
    uint32_t div_m, div_s;          // cell / nm1 == mulhi(cell, div_m) >> div_s for every cell < 2^26 (host: cell_divider)
};

// Screen of this frame -> screen of the frame the plan's tile times come from, for points of the ground plane y = 0 (k_plan):
// (x', y', w') = m * (x, y, 1) in pixels.  on = 0: the times are looked up where they were measured (camera at rest).
struct MotionMap { float m[9]; uint32_t on; };

// inclusive pixel rectangle (clamped to the target) a grid block, or a whole block row, may touch; x0 > x1 = empty
struct PixelBox { int16_t x0, y0, x1, y1; };

// Per block and frame, written by k_block_setup (the vertex stage and every test of a primitive that does not depend on the
// screen tile run once per frame there, not once per (tile, block) pair): which of the block's 128 primitives can draw at all.
//   alive_even / alive_odd   bit c = cell c (row-major 8 x 8) of the block: primitive (a, c, b) / (b, c, d) needs no clipping, is
//                            front-facing and its bounding box holds a pixel centre of the target -> the tile kernel's fast path
//   flags                    kRecGeneric: some primitive of the block needs the generic path (near/far clipping, oversized); which
//                            ones is in the block's entry of the `gen` array, read by the complete tile kernel only
//   box                      pixel centres the alive primitives can cover (exact union of their boxes; the conservative corner
//                            box when the block holds generic primitives); x0 > x1 = nothing to draw
struct BlockRec {
    PixelBox box;
    uint32_t flags;
    uint32_t count;            // popcount(alive_even) + popcount(alive_odd)
    uint64_t alive_even, alive_odd;
};
constexpr uint32_t kRecGeneric = 1u;
static_assert(sizeof(BlockRec) == 32, "BlockRec is read with wide loads");

// ---- deterministic sin / cos ---------------------------------------------------------------
__device__ __forceinline__ float sin_poly(float r)
{
    float r2 = r * r;
    float p = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    p = fmaf(r2, p, -1.6666654611e-1f);
    return fmaf(r * r2, p, r);
}
__device__ __forceinline__ float cos_poly(float r)
{
    float r2 = r * r;
    float p = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    p = fmaf(r2, p, 4.166664568298827e-2f);
    return fmaf(r2 * r2, p, fmaf(-0.5f, r2, 1.0f));
}
__device__ __forceinline__ float reduce_pio2(float x, int &q)
{
    float k = rintf(x * 0.636619772f);
    float r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188e-8f, r);
    q = ((int)k) & 3;
    return r;
}
__device__ __forceinline__ float det_sin(float x)
{
    int q; float r = reduce_pio2(x, q);
    float s = sin_poly(r), c = cos_poly(r);
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
__device__ __forceinline__ float det_cos(float x)
{
    int q; float r = reduce_pio2(x, q);
    float s = sin_poly(r), c = cos_poly(r);
    float v = (q & 1) ? -s : c;
    return (q & 2) ? -v : v;
}

// ---- sRGB store ----------------------------------------------------------------------------
// byte = #{k in 1..255 : c >= T[k]}.  A hardware log/exp estimate lands within one step; the
// comparison against the threshold table makes the result exact and implementation-independent.
__device__ __forceinline__ uint32_t srgb_encode(float c, const float *T /* LDS, 256 */)
{
    float cc = fminf(fmaxf(c, 0.0f), 1.0f);
    float est = cc <= 0.0031308f ? 12.92f * cc
                                 : 1.055f * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(cc) * (1.0f / 2.4f)) - 0.055f;
    int k = (int)(est * 255.0f + 0.5f);
    k = k < 0 ? 0 : (k > 255 ? 255 : k);
    while (k < 255 && c >= T[k + 1]) ++k;
    while (k > 0 && !(c >= T[k])) --k;
    return (uint32_t)k;
}

// ---- vertex stage --------------------------------------------------------------------------
// proj * (view * (wx, wy, wz, 1)): each row is  m0*x, then fma(m1,y,.), fma(m2,z,.), fma(m3,w,.)
__device__ __forceinline__ void mat_vec(const float *m, float x, float y, float z, float w, float r[4])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float acc = m[k] * x;
        acc = fmaf(m[4 + k], y, acc);
        acc = fmaf(m[8 + k], z, acc);
        acc = fmaf(m[12 + k], w, acc);
        r[k] = acc;
    }
}

struct AxisTables {          // per-axis values of build_grid_xyuv + vs_main that depend on i or j only
    const float *xs;         // x_i = -1.5 + i*step                  (also z_j: the grid is square)
    const float *sinx;       // det_sin(x_i * 1.3)
    const float *cosz;       // det_cos(z_j * 1.1)
    const int32_t *txi;      // texel column for u_i = i/(n-1)
    const int32_t *tyj;      // texel row for v_j
};

// Displaced-height cache, rebuilt by k_height_blocks after every height upload: for each 8x8-cell block
// its 9x9 vertex heights h = h_tex + h_ana (terrain.wgsl:50-55) stored contiguously (81 floats = 324 B), so a
// wave fetches a whole block with two coalesced loads and no dependent texel-index lookups.  Shared
// edges are duplicated (1.27x the texture size).  kBlockStride floats per block.
constexpr int kBlockStride = 81;
__device__ __forceinline__ float cached_height(const float *__restrict__ hblk, uint32_t nb, uint32_t i, uint32_t j)
{
    const uint32_t bx = min(i >> 3, nb - 1u), by = min(j >> 3, nb - 1u);
    return hblk[(size_t)(by * nb + bx) * kBlockStride + (j - 8u * by) * 9u + (i - 8u * bx)];
}

struct ClipVert { float x, y, z, w, h; };

// x_i (= z_j) recomputed with the very operations k_axis_tables used for AxisTables::xs: cheaper than a table load in the
// tile kernel's vertex stage, identical bits
__device__ __forceinline__ float grid_coord(const FrameParams &P, uint32_t i) { return -1.5f + (float)i * P.step; }

// h for grid vertex (i, j) exactly as vs_main forms it (terrain.wgsl:50-55)
__device__ __forceinline__ float displaced_height(const AxisTables &A, const float *__restrict__ tex, uint32_t tw, uint32_t i, uint32_t j)
{
    float h_tex = tex[(size_t)A.tyj[j] * tw + A.txi[i]];
    float h_ana = A.sinx[i] * 0.25f + A.cosz[j] * 0.25f;       // terrain.wgsl:39-41
    return h_tex + h_ana;                                      // :55
}

// vs_main from the cached displaced height
__device__ __forceinline__ ClipVert vertex_shader(const FrameParams &P, float x, float z, float h)
{
    float vp[4], cp[4];
    mat_vec(P.view, x * P.spacing, h * P.exag, z * P.spacing, 1.0f, vp);
    mat_vec(P.proj, vp[0], vp[1], vp[2], vp[3], cp);
    ClipVert o;
    o.x = cp[0]; o.y = cp[1]; o.z = cp[2]; o.w = cp[3]; o.h = h;
    return o;
}

__device__ __forceinline__ bool finite4(float a, float b, float c, float d)
{
    return isfinite(a) && isfinite(b) && isfinite(c) && isfinite(d);
}

// viewport transform + 24.8 snap.  Returns false when the vertex cannot be rasterised (w <= 0 or overflow).
__device__ __forceinline__ bool snap_vertex(float cx, float cy, float cw, float hw, float hh, int32_t &X, int32_t &Y, float &rw)
{
    if (!(cw > 0.0f)) return false;
    rw = 1.0f / cw;
    float xf = fmaf(cx * rw, hw, hw);
    float yf = fmaf(-(cy * rw), hh, hh);
    if (!isfinite(xf) || !isfinite(yf)) return false;
    xf = fminf(fmaxf(xf, -4194304.0f), 4194304.0f);
    yf = fminf(fmaxf(yf, -4194304.0f), 4194304.0f);
    X = (int32_t)rintf(xf * 256.0f);
    Y = (int32_t)rintf(yf * 256.0f);
    return true;
}

__device__ __forceinline__ uint32_t vertex_flags(const ClipVert &c)
{
    uint32_t f = 0;
    if (!finite4(c.x, c.y, c.z, c.w)) f |= F_BAD;
    if (c.z < 0.0f) f |= F_NEAR;
    if (c.z > c.w) f |= F_FAR;
    return f;
}

// ---- shard row mapping ---------------------------------------------------------------------
__device__ __forceinline__ bool row_owned(const FrameParams &P, uint32_t y)
{
    return P.nranks <= 1u || ((y >> P.band_shift) % P.nranks) == P.rank;
}
__device__ __forceinline__ uint32_t local_row(const FrameParams &P, uint32_t y)
{
    if (P.nranks <= 1u) return y;
    uint32_t b = y >> P.band_shift;
    return ((b / P.nranks) << P.band_shift) + (y & (P.band_h - 1u));
}
__device__ __forceinline__ uint32_t global_row(const FrameParams &P, uint32_t ly)
{
    if (P.nranks <= 1u) return ly;
    uint32_t lb = ly >> P.band_shift;
    return (((lb * P.nranks) + P.rank) << P.band_shift) + (ly & (P.band_h - 1u));
}

// ---- generic (clipped / large) primitive handling --------------------------------------------
struct GVert { float x, y, z, w; float a[3]; };
struct SVert { int32_t X, Y; float rw; float a[3]; };

__device__ __forceinline__ float plane_dist(const GVert &v, int plane) { return plane == 0 ? v.z : v.w - v.z; }

// Sutherland-Hodgman against z >= 0 then z <= w; crossing points from the inside vertex outwards.
// Returns the polygon size (0 = nothing to draw).
__device__ inline int clip_primitive(const GVert v[3], GVert poly[8])
{
    for (int k = 0; k < 3; ++k)
        if (!finite4(v[k].x, v[k].y, v[k].z, v[k].w)) return 0;
    int out_near = 0, out_far = 0;
    for (int k = 0; k < 3; ++k) { out_near += v[k].z < 0.0f; out_far += v[k].z > v[k].w; }
    if (out_near == 3 || out_far == 3) return 0;
    for (int k = 0; k < 3; ++k) poly[k] = v[k];
    if (out_near == 0 && out_far == 0) return 3;
    GVert tmp[8];
    int n = 3;
    for (int plane = 0; plane < 2; ++plane) {
        int m = 0;
        for (int k = 0; k < n; ++k) {
            const GVert &cur = poly[k];
            const GVert &nxt = poly[(k + 1) % n];
            float dc = plane_dist(cur, plane), dn = plane_dist(nxt, plane);
            bool cin = dc >= 0.0f, nin = dn >= 0.0f;
            if (cin) tmp[m++] = cur;
            if (cin != nin) {
                const GVert &in = cin ? cur : nxt;
                const GVert &ou = cin ? nxt : cur;
                float di = cin ? dc : dn, dou = cin ? dn : dc;
                float t = di / (di - dou);
                GVert r;
                r.x = fmaf(t, ou.x - in.x, in.x);
                r.y = fmaf(t, ou.y - in.y, in.y);
                r.z = fmaf(t, ou.z - in.z, in.z);
                r.w = fmaf(t, ou.w - in.w, in.w);
                for (int a = 0; a < 3; ++a) r.a[a] = fmaf(t, ou.a[a] - in.a[a], in.a[a]);
                tmp[m++] = r;
            }
        }
        n = m;
        for (int k = 0; k < n; ++k) poly[k] = tmp[k];
        if (n < 3) return 0;
    }
    return n;
}

struct TriSetup {
    SVert s[3];
    int64_t area2;
    int32_t px0, px1, py0, py1;
    bool tl0, tl1, tl2;
};

// snap + cull + pixel-centre bbox (clamped to the target).  false = nothing to rasterise.
__device__ inline bool setup_triangle(const GVert &v0, const GVert &v1, const GVert &v2, float hw, float hh,
                                      uint32_t W, uint32_t H, TriSetup &T)
{
    // no pointer tables here: the vertices must stay in registers on the hot fragment path
    if (!snap_vertex(v0.x, v0.y, v0.w, hw, hh, T.s[0].X, T.s[0].Y, T.s[0].rw)) return false;
    if (!snap_vertex(v1.x, v1.y, v1.w, hw, hh, T.s[1].X, T.s[1].Y, T.s[1].rw)) return false;
    if (!snap_vertex(v2.x, v2.y, v2.w, hw, hh, T.s[2].X, T.s[2].Y, T.s[2].rw)) return false;
#pragma unroll
    for (int a = 0; a < 3; ++a) { T.s[0].a[a] = v0.a[a]; T.s[1].a[a] = v1.a[a]; T.s[2].a[a] = v2.a[a]; }
    const SVert *s = T.s;
    T.area2 = (int64_t)(s[1].X - s[0].X) * (s[2].Y - s[0].Y) - (int64_t)(s[1].Y - s[0].Y) * (s[2].X - s[0].X);
    if (T.area2 >= 0) return false;
    int32_t xmin = min(s[0].X, min(s[1].X, s[2].X)), xmax = max(s[0].X, max(s[1].X, s[2].X));
    int32_t ymin = min(s[0].Y, min(s[1].Y, s[2].Y)), ymax = max(s[0].Y, max(s[1].Y, s[2].Y));
    T.px0 = max((xmin + 127) >> 8, 0);
    T.px1 = min((xmax - 128) >> 8, (int32_t)W - 1);
    T.py0 = max((ymin + 127) >> 8, 0);
    T.py1 = min((ymax - 128) >> 8, (int32_t)H - 1);
    if (T.px0 > T.px1 || T.py0 > T.py1) return false;
    const int32_t a0 = s[2].Y - s[1].Y, b0 = -(s[2].X - s[1].X);
    const int32_t a1 = s[0].Y - s[2].Y, b1 = -(s[0].X - s[2].X);
    const int32_t a2 = s[1].Y - s[0].Y, b2 = -(s[1].X - s[0].X);
    T.tl0 = a0 > 0 || (a0 == 0 && b0 > 0);
    T.tl1 = a1 > 0 || (a1 == 0 && b1 > 0);
    T.tl2 = a2 > 0 || (a2 == 0 && b2 > 0);
    return true;
}

__device__ __forceinline__ int64_t edge_fn(const SVert &a, const SVert &b, int64_t Px, int64_t Py)
{
    return (int64_t)(b.X - a.X) * (Py - a.Y) - (int64_t)(b.Y - a.Y) * (Px - a.X);
}

// coverage of pixel (px,py) with the top-left rule; e[] = inside-positive edge weights
__device__ __forceinline__ bool covers(const TriSetup &T, int32_t px, int32_t py, int64_t e[3])
{
    int64_t Px = (int64_t)px * 256 + 128, Py = (int64_t)py * 256 + 128;
    e[0] = -edge_fn(T.s[1], T.s[2], Px, Py);
    e[1] = -edge_fn(T.s[2], T.s[0], Px, Py);
    e[2] = -edge_fn(T.s[0], T.s[1], Px, Py);
    return (e[0] > 0 || (e[0] == 0 && T.tl0)) && (e[1] > 0 || (e[1] == 0 && T.tl1)) && (e[2] > 0 || (e[2] == 0 && T.tl2));
}

// perspective-correct varyings at a covered pixel
__device__ __forceinline__ void interpolate(const TriSetup &T, const int64_t e[3], float attr[3])
{
    const float fA = (float)(-T.area2);
    float l0 = (float)e[0] / fA, l1 = (float)e[1] / fA, l2 = (float)e[2] / fA;
    float q0 = l0 * T.s[0].rw, q1 = l1 * T.s[1].rw, q2 = l2 * T.s[2].rw;
    float rQ = 1.0f / ((q0 + q1) + q2);
#pragma unroll
    for (int k = 0; k < 3; ++k)
        attr[k] = fmaf(q2, T.s[2].a[k], fmaf(q1, T.s[1].a[k], q0 * T.s[0].a[k])) * rQ;
}

} // namespace vf
