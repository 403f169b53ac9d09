// vf_kernels.h -- the gfx950 kernels of the terrain raster path.
//
//   k_axis_tables     once per (grid, texture size): per-column / per-row vertex-shader terms
//   k_block_bounds    once per height upload: min/max displaced height of every 16x16-cell block
//   k_block_ranges    per frame: conservative screen-tile rectangle of every block (+ per block row)
//   k_tile            per frame: one workgroup per 64x64 screen tile.  Walks the block rows that can
//                     touch the tile in DESCENDING primitive order, rebuilds each candidate block's
//                     17x17 vertex tile in LDS, sets its 512 triangles up and rasterises them with
//                     exact FP64 span solving into an LDS visibility tile (ds_max_u32), stops as soon
//                     as every pixel of the tile is final, then runs the fragment stage on the LDS
//                     tile and stores RGBA8 -- no framebuffer-sized intermediate ever touches HBM.
//   k_grid_*          grid_generate (bit-exact make_grid)
//   k_triangle        the triangle smoke path
//   k_stitch_bands    multi-GPU de-interleave
//
// Painter's order: the reference pipeline has no depth buffer (src/terrain/pipeline.rs:133), so the
// visible fragment is the LAST covering front-facing primitive in index order == max primitive id.
// Primitive ids are cell-row major, so every primitive of block row r+1 beats every primitive of
// block row r: after a block row has been rasterised, covered pixels are final.
#pragma once
#include "vf_device.h"

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
    float x = -scale + (float)i * step;                // :566-567
    float uvc = (float)i / nm1f;                       // :568-569
    xs[i] = x;
    sinx[i] = det_sin(x * 1.3f);                       // terrain.wgsl:40
    cosz[i] = det_cos(x * 1.1f);
    int tx = (int)floorf(uvc * (float)tw), ty = (int)floorf(uvc * (float)th);   // nearest, clamp-to-edge
    txi[i] = min(max(tx, 0), (int)tw - 1);
    tyj[i] = min(max(ty, 0), (int)th - 1);
}

// one workgroup per block: exact min/max of h = h_tex + h_ana over its 17x17 vertices
__global__ __launch_bounds__(64) void k_block_bounds(uint32_t n, uint32_t nb, uint32_t tw, AxisTables A,
                                                     const float *__restrict__ tex, float2 *__restrict__ bounds)
{
    const uint32_t bx = blockIdx.x % nb, by = blockIdx.x / nb;
    const uint32_t i0 = bx * kBlockCells, j0 = by * kBlockCells;
    float lo = INFINITY, hi = -INFINITY;
    bool bad = false;
    for (int v = threadIdx.x; v < kBlockVerts * kBlockVerts; v += 64) {
        uint32_t lj = v / kBlockVerts, li = v - lj * kBlockVerts;
        uint32_t i = i0 + li, j = j0 + lj;
        if (i < n && j < n) {
            float h = tex[(size_t)A.tyj[j] * tw + A.txi[i]] + (A.sinx[i] * 0.25f + A.cosz[j] * 0.25f);
            bad |= !isfinite(h);
            lo = fminf(lo, h); hi = fmaxf(hi, h);
        }
    }
    if (bad) { lo = -INFINITY; hi = INFINITY; }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if (threadIdx.x == 0) bounds[blockIdx.x] = make_float2(lo, hi);
}

// ---------------------------------------------------------------------------------------------
// per frame: which screen tiles can a block touch?  One workgroup per block row, one thread per block.
// The block's vertices all lie in the box [x0,x1] x [hmin,hmax] x [z0,z1]; when its 8 corners are
// regular (finite, w > 0, inside near/far) their screen bbox bounds every vertex of the block.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_block_ranges(FrameParams P, AxisTables A, const float2 *__restrict__ bounds,
                                                      TileRange *__restrict__ ranges, TileRange *__restrict__ row_ranges)
{
    __shared__ int s_rr[4];   // x0, y0 (min) / x1, y1 (max)
    const uint32_t by = blockIdx.x;
    if (threadIdx.x == 0) { s_rr[0] = 0xFFFF; s_rr[1] = 0xFFFF; s_rr[2] = -1; s_rr[3] = -1; }
    __syncthreads();
    for (uint32_t bx = threadIdx.x; bx < P.nb; bx += 256) {
        const uint32_t b = by * P.nb + bx;
        const float2 hb = bounds[b];
        const uint32_t i0 = bx * kBlockCells, j0 = by * kBlockCells;
        const uint32_t i1 = min(i0 + kBlockCells, P.n - 1), j1 = min(j0 + kBlockCells, P.n - 1);
        float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
        int regular = 0, out_near = 0, out_far = 0;
        const bool hfinite = isfinite(hb.x) && isfinite(hb.y);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float x = A.xs[(c & 1) ? i1 : i0], z = A.xs[(c & 2) ? j1 : j0], h = (c & 4) ? hb.y : hb.x;
            float vp[4], cp[4];
            mat_vec(P.view, x * P.spacing, h * P.exag, z * P.spacing, 1.0f, vp);
            mat_vec(P.proj, vp[0], vp[1], vp[2], vp[3], cp);
            bool fin = finite4(cp[0], cp[1], cp[2], cp[3]);
            // margins keep the whole-block rejection conservative against rounding at the clip planes
            const float margin = 1e-3f * fmaxf(1.0f, fabsf(cp[3]));
            out_near += fin && cp[2] < -margin;
            out_far += fin && cp[2] - cp[3] > margin;
            if (fin && cp[3] > 0.0f && cp[2] >= 0.0f && cp[2] <= cp[3]) {
                float rw = 1.0f / cp[3];
                float xf = fmaf(cp[0] * rw, P.hw, P.hw), yf = fmaf(-(cp[1] * rw), P.hh, P.hh);
                if (isfinite(xf) && isfinite(yf)) {
                    ++regular;
                    xmin = fminf(xmin, xf); xmax = fmaxf(xmax, xf);
                    ymin = fminf(ymin, yf); ymax = fmaxf(ymax, yf);
                }
            }
        }
        TileRange r;
        if (hfinite && (out_near == 8 || out_far == 8)) {
            r.x0 = 1; r.y0 = 1; r.x1 = 0; r.y1 = 0;            // clip z is affine in position: the whole block is clipped away
        } else if (regular == 8) {
            // one pixel of slack covers the rounding difference between corner and vertex arithmetic
            if (xmax < -1.0f || ymax < -1.0f || xmin > (float)P.W + 1.0f || ymin > (float)P.H + 1.0f) {
                r.x0 = 1; r.y0 = 1; r.x1 = 0; r.y1 = 0;
            } else {
                int px0 = (int)fmaxf(floorf(xmin) - 1.0f, 0.0f), px1 = (int)fminf(ceilf(xmax) + 1.0f, (float)P.W - 1.0f);
                int py0 = (int)fmaxf(floorf(ymin) - 1.0f, 0.0f), py1 = (int)fminf(ceilf(ymax) + 1.0f, (float)P.H - 1.0f);
                r.x0 = (uint16_t)(px0 / kTileW); r.x1 = (uint16_t)(px1 / kTileW);
                r.y0 = (uint16_t)(py0 / kTileH); r.y1 = (uint16_t)(py1 / kTileH);
            }
        } else {
            r.x0 = 0; r.y0 = 0; r.x1 = (uint16_t)(P.ntx - 1); r.y1 = (uint16_t)(P.nty - 1);   // no bound available: every tile
        }
        ranges[b] = r;
        if (r.x0 <= r.x1) {
            atomicMin(&s_rr[0], (int)r.x0); atomicMin(&s_rr[1], (int)r.y0);
            atomicMax(&s_rr[2], (int)r.x1); atomicMax(&s_rr[3], (int)r.y1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        TileRange rr;
        if (s_rr[2] < 0) { rr.x0 = 1; rr.y0 = 1; rr.x1 = 0; rr.y1 = 0; }
        else { rr.x0 = (uint16_t)s_rr[0]; rr.y0 = (uint16_t)s_rr[1]; rr.x1 = (uint16_t)s_rr[2]; rr.y1 = (uint16_t)s_rr[3]; }
        row_ranges[by] = rr;
    }
}

// ---------------------------------------------------------------------------------------------
// tile kernel
// ---------------------------------------------------------------------------------------------
struct TileCtx {
    uint32_t *vis;            // LDS, kTileW*kTileH words, skewed (see vis_index)
    int32_t px_lo, px_hi;     // inclusive pixel rectangle of the tile, clipped to the target
    int32_t py_lo, py_hi;
};

// LDS layout of the visibility tile: rotate each row by its row number so that a walk down a pixel
// column visits all 32 banks (plain row-major would keep a column in one bank: 64-word row stride).
__device__ __forceinline__ uint32_t vis_index(int32_t lx, int32_t ly) { return (uint32_t)ly * kTileW + (uint32_t)((lx + ly) & (kTileW - 1)); }

__device__ __forceinline__ int32_t clamp_d2i(double v, int32_t lo, int32_t hi)
{
    v = fmax(v, (double)lo); v = fmin(v, (double)hi);
    return (int32_t)v;
}

// Exact rasterisation of one unclipped front-facing-or-not triangle restricted to the tile.
// Edge functions are evaluated in FP64: all operands are integers < 2^25 and every product/sum stays
// below 2^53, so the arithmetic is exact; spans along the longer bbox axis are solved with one FP64
// division per edge, exact for quotients below 2^20 (larger ones are clamped away) -- DESIGN.md.
__device__ __forceinline__ void raster_fast(const TileCtx &T, uint32_t word, int32_t X0, int32_t Y0, int32_t X1, int32_t Y1,
                                            int32_t X2, int32_t Y2)
{
    const int32_t xmin = min(X0, min(X1, X2)), xmax = max(X0, max(X1, X2));
    const int32_t ymin = min(Y0, min(Y1, Y2)), ymax = max(Y0, max(Y1, Y2));
    const int32_t px0 = max((xmin + 127) >> 8, T.px_lo), px1 = min((xmax - 128) >> 8, T.px_hi);
    const int32_t py0 = max((ymin + 127) >> 8, T.py_lo), py1 = min((ymax - 128) >> 8, T.py_hi);
    if (px0 > px1 || py0 > py1) return;
    const double dX0 = X0, dY0 = Y0, dX1 = X1, dY1 = Y1, dX2 = X2, dY2 = Y2;
    const double area2 = fma(dX1 - dX0, dY2 - dY0, -((dY1 - dY0) * (dX2 - dX0)));
    if (area2 >= 0.0) return;                              // back-facing or degenerate
    // inside-positive edge functions e_i(P) = A_i (Px - Xr_i) + B_i (Py - Yr_i); covered iff e_i + t_i > 0
    const double A[3] = { dY2 - dY1, dY0 - dY2, dY1 - dY0 };
    const double B[3] = { -(dX2 - dX1), -(dX0 - dX2), -(dX1 - dX0) };
    const double XR[3] = { dX1, dX2, dX0 }, YR[3] = { dY1, dY2, dY0 };
    const bool cols = (px1 - px0) <= (py1 - py0);          // iterate the short axis, solve spans along the long one
    const int32_t n_outer = cols ? px1 - px0 : py1 - py0, n_inner = cols ? py1 - py0 : px1 - px0;
    // f_i(o, r) = base_i + SO_i*o + SI_i*r   with o/r = outer/inner pixel offsets from (px0, py0)
    double base[3], SO[3], SI[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double tl = (A[i] > 0.0 || (A[i] == 0.0 && B[i] > 0.0)) ? 1.0 : 0.0;
        base[i] = fma(A[i], (double)(px0 * 256 + 128) - XR[i], fma(B[i], (double)(py0 * 256 + 128) - YR[i], tl));
        SO[i] = 256.0 * (cols ? A[i] : B[i]);
        SI[i] = 256.0 * (cols ? B[i] : A[i]);
    }
    for (int32_t o = 0; o <= n_outer; ++o) {
        int32_t lo = 0, hi = n_inner;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double alpha = fma(SO[i], (double)o, base[i]);   // f_i at inner offset 0
            const double beta = SI[i];
            if (beta > 0.0) {            // need r > -alpha/beta
                if (alpha <= 0.0) lo = max(lo, clamp_d2i(floor(-alpha / beta) + 1.0, 0, n_inner + 1));
            } else if (beta < 0.0) {     // need r < alpha/(-beta)  <=>  r <= floor((alpha-1)/(-beta))
                if (alpha <= 0.0) hi = -1;
                else hi = min(hi, clamp_d2i(floor((alpha - 1.0) / -beta), -1, n_inner));
            } else if (alpha <= 0.0) hi = -1;
        }
        if (cols) {
            const int32_t lx = px0 + o - T.px_lo;
            for (int32_t r = lo; r <= hi; ++r) atomicMax(&T.vis[vis_index(lx, py0 + r - T.py_lo)], word);
        } else {
            const int32_t ly = py0 + o - T.py_lo;
            for (int32_t r = lo; r <= hi; ++r) atomicMax(&T.vis[vis_index(px0 + r - T.px_lo, ly)], word);
        }
    }
}

// primitive -> clip-space vertices with varyings (used by the clipped path and the fragment stage)
__device__ inline void load_prim(const FrameParams &P, const AxisTables &A, const float *__restrict__ tex, uint32_t prim, GVert v[3])
{
    uint32_t vi[3], vj[3];
    prim_vertices(prim, P.nm1, vi, vj);
    for (int k = 0; k < 3; ++k) {
        float x, z;
        ClipVert c = vertex_shader(P, A, tex, vi[k], vj[k], x, z);
        v[k].x = c.x; v[k].y = c.y; v[k].z = c.z; v[k].w = c.w;
        v[k].a[0] = c.h; v[k].a[1] = x; v[k].a[2] = z;     // varyings: height, xz (terrain.wgsl:63-64)
    }
}

// clipped or oversized primitives: clip, fan, and scan each piece's bbox inside the tile with the
// int64 coverage test.  Rare (primitives crossing the near plane, or > 65536 px across).
__device__ __noinline__ void raster_generic(const FrameParams &P, const AxisTables &A, const float *__restrict__ tex,
                                            const TileCtx &T, uint32_t prim)
{
    GVert v[3], poly[8];
    load_prim(P, A, tex, prim, v);
    const int np = clip_primitive(v, poly);
    for (int f = 1; f + 1 < np; ++f) {
        TriSetup S;
        if (!setup_triangle(poly[0], poly[f], poly[f + 1], P.hw, P.hh, P.W, P.H, S)) continue;
        const int32_t px0 = max(S.px0, T.px_lo), px1 = min(S.px1, T.px_hi);
        const int32_t py0 = max(S.py0, T.py_lo), py1 = min(S.py1, T.py_hi);
        for (int32_t py = py0; py <= py1; ++py)
            for (int32_t px = px0; px <= px1; ++px) {
                int64_t e[3];
                if (covers(S, px, py, e)) atomicMax(&T.vis[vis_index(px - T.px_lo, py - T.py_lo)], prim + 1u);
            }
    }
}

__device__ __forceinline__ void raster_prim(const FrameParams &P, const AxisTables &A, const float *__restrict__ tex, const TileCtx &T,
                                            uint32_t prim, uint32_t fl0, uint32_t fl1, uint32_t fl2, int32_t X0, int32_t Y0,
                                            int32_t X1, int32_t Y1, int32_t X2, int32_t Y2)
{
    const uint32_t any = fl0 | fl1 | fl2, all = fl0 & fl1 & fl2;
    if (any & F_BAD) return;                              // non-finite clip coordinate: primitive dropped
    if (all & (F_NEAR | F_FAR)) return;                   // entirely outside the near or the far plane
    if (any & (F_NEAR | F_FAR)) { raster_generic(P, A, tex, T, prim); return; }   // needs clipping
    if (any & F_NOSNAP) return;                           // a vertex could not be projected (w <= 0)
    const uint32_t ex = (uint32_t)max(X0, max(X1, X2)) - (uint32_t)min(X0, min(X1, X2));
    const uint32_t ey = (uint32_t)max(Y0, max(Y1, Y2)) - (uint32_t)min(Y0, min(Y1, Y2));
    if (ex >= (uint32_t)kFastExtent || ey >= (uint32_t)kFastExtent) { raster_generic(P, A, tex, T, prim); return; }
    raster_fast(T, prim + 1u, X0, Y0, X1, Y1, X2, Y2);
}

// ---- fragment stage ---------------------------------------------------------------------------
struct ShadeTables { const float *lut; const float *thresh; };   // LDS: 256*3 linear LUT, 256 sRGB thresholds

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
    float dhdx = 1.3f * det_cos(x * 1.3f) * 0.25f;
    float dhdz = -1.1f * det_sin(z * 1.1f) * 0.25f;
    float d = fmaf(dhdz, dhdz, fmaf(dhdx, dhdx, 1.0f));
    float inv = 1.0f / sqrtf(d);
    float nx = -dhdx * inv, ny = inv, nz = -dhdz * inv;
    float ndl = fmaf(nz, P.Lz, fmaf(ny, P.Ly, nx * P.Lx));
    float lambert = fminf(fmaxf(ndl, 0.0f), 1.0f);
    float shade = 0.15f * (1.0f - lambert) + lambert;
    uint32_t out = 0xFF000000u;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float l0 = S.lut[i0 * 3 + ch], l1 = S.lut[i1 * 3 + ch];
        float lc = fmaf(f, l1 - l0, l0);
        float v = lc * P.exposure * shade;
        out |= srgb_encode(v, S.thresh) << (8 * ch);
    }
    return out;
}

__device__ __noinline__ bool clipped_attributes(const FrameParams &P, const GVert v[3], int32_t px, int32_t py, float attr[3])
{
    GVert poly[8];
    const int np = clip_primitive(v, poly);
    bool hit = false;
    for (int f = 1; f + 1 < np; ++f) {      // the last covering piece wins, as in the draw order
        TriSetup T;
        int64_t e[3];
        if (setup_triangle(poly[0], poly[f], poly[f + 1], P.hw, P.hh, P.W, P.H, T) && covers(T, px, py, e)) { interpolate(T, e, attr); hit = true; }
    }
    return hit;
}

__device__ inline uint32_t shade_pixel(const FrameParams &P, const AxisTables &A, const float *__restrict__ tex,
                                       const ShadeTables &S, uint32_t prim, int32_t px, int32_t py)
{
    GVert v[3];
    load_prim(P, A, tex, prim, v);
    bool plain = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) plain &= finite4(v[k].x, v[k].y, v[k].z, v[k].w) && !(v[k].z < 0.0f) && !(v[k].z > v[k].w);
    float attr[3] = { 0.f, 0.f, 0.f };
    bool hit = false;
    if (plain) {
        TriSetup T;
        int64_t e[3];
        if (setup_triangle(v[0], v[1], v[2], P.hw, P.hh, P.W, P.H, T) && covers(T, px, py, e)) { interpolate(T, e, attr); hit = true; }
    } else {
        hit = clipped_attributes(P, v, px, py, attr);
    }
    if (!hit) return P.clear_rgba;   // unreachable when the visibility tile is consistent
    return fragment_shader(P, S, attr);
}

__device__ __forceinline__ bool range_hits(const TileRange &r, uint32_t tx, uint32_t ty)
{
    return r.x0 <= r.x1 && tx >= r.x0 && tx <= r.x1 && ty >= r.y0 && ty <= r.y1;
}

template <bool WRITE_VIS>
__global__ __launch_bounds__(kTileThreads) void k_tile(FrameParams P, AxisTables A, const float *__restrict__ tex,
                                                       const TileRange *__restrict__ ranges, const TileRange *__restrict__ row_ranges,
                                                       const float *__restrict__ lut_linear, const float *__restrict__ thresh,
                                                       uint32_t *__restrict__ rgba, uint32_t *__restrict__ vis_out, uint32_t *stats)
{
    __shared__ uint32_t s_vis[kTileW * kTileH];
    __shared__ int32_t sX[kBlockVerts * kBlockVerts];
    __shared__ int32_t sY[kBlockVerts * kBlockVerts];
    __shared__ uint8_t sF[kBlockVerts * kBlockVerts];
    __shared__ uint16_t s_cand[1024];
    __shared__ uint32_t s_ncand;
    __shared__ float s_lut[256 * 3];
    __shared__ float s_thr[256];
    __shared__ uint32_t s_part[8];   // per-wave covered-pixel counts, double-buffered by row parity

    const uint32_t tid = threadIdx.x;
    // tile coordinates: blockIdx -> (tile column, local tile row) -> global tile row of this shard
    const uint32_t ttx = blockIdx.x % P.ntx, lty = blockIdx.x / P.ntx;
    const uint32_t gy0 = global_row(P, lty * kTileH);              // band_h is a multiple of kTileH
    const uint32_t tty = gy0 / kTileH;
    TileCtx T;
    T.vis = s_vis;
    T.px_lo = (int32_t)(ttx * kTileW); T.px_hi = min(T.px_lo + kTileW, (int32_t)P.W) - 1;
    T.py_lo = (int32_t)gy0;            T.py_hi = min(T.py_lo + kTileH, (int32_t)P.H) - 1;
    const uint32_t tile_pixels = (uint32_t)(T.px_hi - T.px_lo + 1) * (uint32_t)(T.py_hi - T.py_lo + 1);

    for (int k = tid; k < kTileW * kTileH; k += kTileThreads) s_vis[k] = 0u;
    for (int k = tid; k < 768; k += kTileThreads) s_lut[k] = lut_linear[k];
    s_thr[tid] = thresh[tid];
    if (tid == 0) s_ncand = 0;
    __syncthreads();

    uint32_t blocks_done = 0;
    uint32_t cand_total = 0;   // running value of the monotonic candidate counter (uniform)
    for (int32_t by = (int32_t)P.nb - 1; by >= 0; --by) {
        const TileRange rr = row_ranges[by];
        if (!range_hits(rr, ttx, tty)) continue;                   // uniform
        // ---- candidate blocks of this block row (order inside a row is irrelevant: atomicMax) ----
        for (uint32_t base = 0; base < P.nb; base += kTileThreads) {
            const uint32_t bx = base + tid;
            bool hit = false;
            if (bx < P.nb) hit = range_hits(ranges[(uint32_t)by * P.nb + bx], ttx, tty);
            const unsigned long long m = __ballot(hit);
            uint32_t wbase = 0;
            if ((tid & 63u) == 0 && m) wbase = atomicAdd(&s_ncand, (uint32_t)__popcll(m));
            wbase = __shfl(wbase, 0);
            if (hit) s_cand[(wbase + __popcll(m & ((1ull << (tid & 63u)) - 1ull))) & 1023u] = (uint16_t)bx;
        }
        __syncthreads();
        const uint32_t cand_end = s_ncand;
        const uint32_t nc = cand_end - cand_total;
        for (uint32_t c = 0; c < nc; ++c) {
            const uint32_t bx = s_cand[(cand_total + c) & 1023u];
            const uint32_t i0 = bx * kBlockCells, j0 = (uint32_t)by * kBlockCells;
            // ---- vertex stage: 17 x 17 vertices -> snapped screen coordinates in LDS ----
            for (int v = tid; v < kBlockVerts * kBlockVerts; v += kTileThreads) {
                const uint32_t lj = v / kBlockVerts, li = v - lj * kBlockVerts;
                const uint32_t i = i0 + li, j = j0 + lj;
                int32_t X = 0, Y = 0;
                uint32_t fl = F_BAD;
                if (i < P.n && j < P.n) {
                    float x, z, rw;
                    ClipVert cv = vertex_shader(P, A, tex, i, j, x, z);
                    fl = vertex_flags(cv);
                    if (!(fl & F_BAD) && !snap_vertex(cv.x, cv.y, cv.w, P.hw, P.hh, X, Y, rw)) fl |= F_NOSNAP;
                }
                sX[v] = X; sY[v] = Y; sF[v] = (uint8_t)fl;
            }
            __syncthreads();
            // ---- primitive stage: thread = cell, both triangles ----
            {
                const uint32_t lj = tid >> 4, li = tid & 15u;
                const uint32_t i = i0 + li, j = j0 + lj;
                if (i < P.nm1 && j < P.nm1) {
                    const uint32_t va = lj * kBlockVerts + li, vb = va + 1, vc = va + kBlockVerts, vd = vc + 1;
                    const int32_t Xa = sX[va], Ya = sY[va], Xb = sX[vb], Yb = sY[vb];
                    const int32_t Xc = sX[vc], Yc = sY[vc], Xd = sX[vd], Yd = sY[vd];
                    const uint32_t fa = sF[va], fb = sF[vb], fc = sF[vc], fd = sF[vd];
                    const uint32_t prim = 2u * (j * P.nm1 + i);
                    raster_prim(P, A, tex, T, prim, fa, fc, fb, Xa, Ya, Xc, Yc, Xb, Yb);          // (a, c, b)
                    raster_prim(P, A, tex, T, prim + 1u, fb, fc, fd, Xb, Yb, Xc, Yc, Xd, Yd);     // (b, c, d)
                }
            }
            __syncthreads();
        }
        blocks_done += nc;
        cand_total = cand_end;
        // ---- early out: pixels covered so far are final (lower block rows only hold smaller ids) ----
        uint32_t covered = 0;
        if (nc) for (int k = tid; k < kTileW * kTileH; k += kTileThreads) covered += s_vis[k] != 0u;
        for (int o = 32; o > 0; o >>= 1) covered += __shfl_xor(covered, o);
        if ((tid & 63u) == 0) s_part[(by & 1) * 4 + (tid >> 6)] = covered;
        __syncthreads();   // also orders every thread's read of s_ncand before the next row's atomics
        const uint32_t *part = s_part + (by & 1) * 4;
        if (part[0] + part[1] + part[2] + part[3] >= tile_pixels) break;   // uniform
    }
    if (stats && tid == 0) atomicAdd(&stats[0], blocks_done);

    // ---- fragment stage on the LDS tile; one wave writes one 256-byte row segment ----
    ShadeTables S = { s_lut, s_thr };
    for (int k = tid; k < kTileW * kTileH; k += kTileThreads) {
        const int32_t lx = k & (kTileW - 1), ly = k >> 6;
        const int32_t px = T.px_lo + lx, py = T.py_lo + ly;
        if (px > T.px_hi || py > T.py_hi) continue;
        const uint32_t id = s_vis[vis_index(lx, ly)];
        const size_t o = (size_t)(lty * kTileH + (uint32_t)ly) * P.W + (uint32_t)px;
        rgba[o] = id ? shade_pixel(P, A, tex, S, id - 1u, px, py) : P.clear_rgba;
        if (WRITE_VIS) vis_out[o] = id;
    }
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

} // namespace vf
