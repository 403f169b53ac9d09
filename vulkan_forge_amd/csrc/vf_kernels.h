// vf_kernels.h -- the gfx950 kernels of the terrain raster path.
//
//   k_axis_tables     once per (grid, texture size): per-column / per-row vertex-shader terms
//   k_block_bounds    once per height upload: min/max displaced height of every 32x32-cell block
//   k_geometry        per frame: block cull -> 33x33 vertex tile in LDS -> triangle setup ->
//                     small-triangle raster with atomicMax(primitive id) into the visibility buffer
//   k_generic         per frame: clipped / large primitives, one workgroup slice per primitive
//   k_resolve         per frame: fragment stage, visibility -> RGBA8 (sRGB)
//   k_grid_generate   grid_generate (bit-exact make_grid)
//   k_triangle        the triangle smoke path
//   k_stitch_bands    multi-GPU de-interleave
//
// Painter's order: the reference pipeline has no depth buffer (src/terrain/pipeline.rs:133), so the
// visible fragment is the LAST covering front-facing primitive in index order == max primitive id.
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

// one workgroup per 32x32-cell block: exact min/max of h = h_tex + h_ana over its 33x33 vertices
__global__ __launch_bounds__(256) void k_block_bounds(uint32_t n, uint32_t nbx, uint32_t tw, AxisTables A,
                                                      const float *__restrict__ tex, float2 *__restrict__ bounds)
{
    __shared__ float smin[4], smax[4];
    const uint32_t bx = blockIdx.x % nbx, by = blockIdx.x / nbx;
    const uint32_t i0 = bx * kBlockCells, j0 = by * kBlockCells;
    float lo = INFINITY, hi = -INFINITY;
    bool bad = false;
    for (int v = threadIdx.x; v < kBlockVerts * kBlockVerts; v += 256) {
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
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        hi = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
        bounds[blockIdx.x] = make_float2(lo, hi);
    }
}

// ---------------------------------------------------------------------------------------------
// geometry + small-triangle raster
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void push_slow(uint32_t prim, uint32_t *slow_list, uint32_t *counters)
{
    uint32_t k = atomicAdd(&counters[0], 1u);
    slow_list[k] = prim;   // capacity == total primitive count: cannot overflow
}

__device__ __forceinline__ void raster_small(const FrameParams &P, uint32_t prim, uint32_t fl0, uint32_t fl1, uint32_t fl2,
                                             int32_t X0, int32_t Y0, int32_t X1, int32_t Y1, int32_t X2, int32_t Y2,
                                             uint32_t *__restrict__ vis, uint32_t *slow_list, uint32_t *counters)
{
    const uint32_t any = fl0 | fl1 | fl2, all = fl0 & fl1 & fl2;
    if (any & F_BAD) return;                              // non-finite clip coordinate: primitive dropped
    if (all & (F_NEAR | F_FAR)) return;                   // entirely outside the near or the far plane
    if (any & (F_NEAR | F_FAR)) { push_slow(prim, slow_list, counters); return; }   // needs clipping
    if (any & 8u) return;                                 // a vertex could not be projected (w <= 0)

    const int32_t xmin = min(X0, min(X1, X2)), xmax = max(X0, max(X1, X2));
    const int32_t ymin = min(Y0, min(Y1, Y2)), ymax = max(Y0, max(Y1, Y2));
    int32_t px0 = max((xmin + 127) >> 8, 0), px1 = min((xmax - 128) >> 8, (int32_t)P.W - 1);
    int32_t py0 = max((ymin + 127) >> 8, 0), py1 = min((ymax - 128) >> 8, (int32_t)P.H - 1);
    if (px0 > px1 || py0 > py1) return;                   // no pixel centre inside the bbox (the common case)
    if ((uint32_t)xmax - (uint32_t)xmin >= (uint32_t)kSmallExtent || (uint32_t)ymax - (uint32_t)ymin >= (uint32_t)kSmallExtent) {
        push_slow(prim, slow_list, counters);
        return;
    }
    // extents < 2^14: every product below fits in 32 bits
    const int32_t area2 = (X1 - X0) * (Y2 - Y0) - (Y1 - Y0) * (X2 - X0);
    if (area2 >= 0) return;                               // back-facing or degenerate
    if ((px1 - px0 + 1) * (py1 - py0 + 1) > kSmallPixels) { push_slow(prim, slow_list, counters); return; }
    const int32_t a0 = Y2 - Y1, b0 = -(X2 - X1);
    const int32_t a1 = Y0 - Y2, b1 = -(X0 - X2);
    const int32_t a2 = Y1 - Y0, b2 = -(X1 - X0);
    // top-left rule as an integer bias: covered iff e + bias > 0  (bias 1 on top/left edges, else 0) -> e >= 0 / e > 0
    const int32_t t0 = (a0 > 0 || (a0 == 0 && b0 > 0)) ? 1 : 0;
    const int32_t t1 = (a1 > 0 || (a1 == 0 && b1 > 0)) ? 1 : 0;
    const int32_t t2 = (a2 > 0 || (a2 == 0 && b2 > 0)) ? 1 : 0;
    const uint32_t word = P.tag | (prim + 1u);
    for (int32_t py = py0; py <= py1; ++py) {
        if (!row_owned(P, (uint32_t)py)) continue;
        const int32_t Py = py * 256 + 128;
        uint32_t *row = vis + (size_t)local_row(P, (uint32_t)py) * P.W;
        for (int32_t px = px0; px <= px1; ++px) {
            const int32_t Px = px * 256 + 128;
            // e_i = -E_jk(P), all differences < 2^15 in magnitude
            const int32_t e0 = (Y2 - Y1) * (Px - X1) - (X2 - X1) * (Py - Y1);
            const int32_t e1 = (Y0 - Y2) * (Px - X2) - (X0 - X2) * (Py - Y2);
            const int32_t e2 = (Y1 - Y0) * (Px - X0) - (X1 - X0) * (Py - Y0);
            if (e0 + t0 > 0 && e1 + t1 > 0 && e2 + t2 > 0) atomicMax(row + px, word);
        }
    }
}

__global__ __launch_bounds__(kGeomThreads) void k_geometry(FrameParams P, AxisTables A, const float *__restrict__ tex,
                                                           const float2 *__restrict__ bounds, uint32_t nbx,
                                                           uint32_t *__restrict__ vis, uint32_t *slow_list, uint32_t *counters)
{
    __shared__ int32_t sX[kBlockVerts * kBlockVerts];
    __shared__ int32_t sY[kBlockVerts * kBlockVerts];
    __shared__ uint8_t sF[kBlockVerts * kBlockVerts];
    __shared__ int s_cull;

    const uint32_t tid = threadIdx.x;
    const uint32_t bx = blockIdx.x % nbx, by = blockIdx.x / nbx;
    const uint32_t i0 = bx * kBlockCells, j0 = by * kBlockCells;

    // ---- block cull: project the 8 corners of the block's (x, h, z) bounding box ----------------
    if (tid < 64) {
        bool keep = true;       // conservative default
        float xf = 0.f, yf = 0.f;
        bool ok = false;
        if (tid < 8) {
            const float2 hb = bounds[blockIdx.x];
            uint32_t i1 = min(i0 + kBlockCells, P.n - 1), j1 = min(j0 + kBlockCells, P.n - 1);
            float x = A.xs[(tid & 1) ? i1 : i0], z = A.xs[(tid & 2) ? j1 : j0], h = (tid & 4) ? hb.y : hb.x;
            float vp[4], cp[4];
            mat_vec(P.view, x * P.spacing, h * P.exag, z * P.spacing, 1.0f, vp);
            mat_vec(P.proj, vp[0], vp[1], vp[2], vp[3], cp);
            ok = finite4(cp[0], cp[1], cp[2], cp[3]) && cp[3] > 0.0f && cp[2] >= 0.0f && cp[2] <= cp[3];
            if (ok) {
                float rw = 1.0f / cp[3];
                xf = fmaf(cp[0] * rw, P.hw, P.hw);
                yf = fmaf(-(cp[1] * rw), P.hh, P.hh);
                ok = isfinite(xf) && isfinite(yf);
            }
        }
        // all 8 corners must be regular for the bound to hold
        unsigned long long okmask = __ballot(ok);
        if ((okmask & 0xFFull) == 0xFFull) {
            float xmin = tid < 8 ? xf : INFINITY, xmax = tid < 8 ? xf : -INFINITY;
            float ymin = tid < 8 ? yf : INFINITY, ymax = tid < 8 ? yf : -INFINITY;
            for (int o = 4; o > 0; o >>= 1) {
                xmin = fminf(xmin, __shfl_xor(xmin, o)); xmax = fmaxf(xmax, __shfl_xor(xmax, o));
                ymin = fminf(ymin, __shfl_xor(ymin, o)); ymax = fmaxf(ymax, __shfl_xor(ymax, o));
            }
            if (tid == 0) {
                // one pixel of slack covers the rounding difference between corner and vertex arithmetic
                if (xmax < -1.0f || ymax < -1.0f || xmin > (float)P.W + 1.0f || ymin > (float)P.H + 1.0f) keep = false;
                else if (P.nranks > 1u) {
                    int32_t ylo = (int32_t)fmaxf(floorf(ymin) - 1.0f, 0.0f);
                    int32_t yhi = (int32_t)fminf(ceilf(ymax) + 1.0f, (float)P.H - 1.0f);
                    uint32_t blo = (uint32_t)ylo >> P.band_shift, bhi = (uint32_t)yhi >> P.band_shift;
                    if (bhi - blo + 1u < P.nranks) {
                        keep = false;
                        for (uint32_t b = blo; b <= bhi; ++b) keep |= (b % P.nranks) == P.rank;
                    }
                }
            }
        }
        if (tid == 0) {
            s_cull = keep ? 0 : 1;
            if (!keep) atomicAdd(&counters[1], 1u);
        }
    }
    __syncthreads();
    if (s_cull) return;

    // ---- vertex stage: 33 x 33 vertices -> snapped screen coordinates in LDS ---------------------
    for (int v = tid; v < kBlockVerts * kBlockVerts; v += kGeomThreads) {
        uint32_t lj = v / kBlockVerts, li = v - lj * kBlockVerts;
        uint32_t i = i0 + li, j = j0 + lj;
        int32_t X = 0, Y = 0;
        uint32_t fl = F_BAD;
        if (i < P.n && j < P.n) {
            float x, z;
            ClipVert c = vertex_shader(P, A, tex, i, j, x, z);
            fl = vertex_flags(c);
            float rw;
            if (!(fl & F_BAD) && !snap_vertex(c.x, c.y, c.w, P.hw, P.hh, X, Y, rw)) fl |= 8u;
        }
        sX[v] = X; sY[v] = Y; sF[v] = (uint8_t)fl;
    }
    __syncthreads();

    // ---- primitive stage: 2 triangles per cell, 4 cells per thread -------------------------------
#pragma unroll 1
    for (int k = 0; k < (kBlockCells * kBlockCells) / kGeomThreads; ++k) {
        const uint32_t c = tid + kGeomThreads * k;
        const uint32_t lj = c >> 5, li = c & 31u;
        const uint32_t i = i0 + li, j = j0 + lj;
        if (i >= P.nm1 || j >= P.nm1) continue;
        const uint32_t va = lj * kBlockVerts + li, vb = va + 1, vc = va + kBlockVerts, vd = vc + 1;
        const int32_t Xa = sX[va], Ya = sY[va], Xb = sX[vb], Yb = sY[vb];
        const int32_t Xc = sX[vc], Yc = sY[vc], Xd = sX[vd], Yd = sY[vd];
        const uint32_t fa = sF[va], fb = sF[vb], fc = sF[vc], fd = sF[vd];
        const uint32_t prim = 2u * (j * P.nm1 + i);
        raster_small(P, prim, fa, fc, fb, Xa, Ya, Xc, Yc, Xb, Yb, vis, slow_list, counters);        // (a, c, b)
        raster_small(P, prim + 1u, fb, fc, fd, Xb, Yb, Xc, Yc, Xd, Yd, vis, slow_list, counters);   // (b, c, d)
    }
}

// ---------------------------------------------------------------------------------------------
// generic path: clipped or large primitives.  item = (list entry, row part)
// ---------------------------------------------------------------------------------------------
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

__global__ __launch_bounds__(256) void k_generic(FrameParams P, AxisTables A, const float *__restrict__ tex,
                                                 uint32_t *__restrict__ vis, const uint32_t *slow_list, const uint32_t *counters)
{
    const uint32_t count = min(counters[0], P.slow_cap);
    const uint64_t items = (uint64_t)count * kGenericSplit;
    for (uint64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const uint32_t prim = slow_list[item / kGenericSplit];
        const uint32_t part = (uint32_t)(item % kGenericSplit);
        GVert v[3], poly[8];
        load_prim(P, A, tex, prim, v);
        const int np = clip_primitive(v, poly);
        const uint32_t word = P.tag | (prim + 1u);
        for (int f = 1; f + 1 < np; ++f) {
            TriSetup T;
            if (!setup_triangle(poly[0], poly[f], poly[f + 1], P.hw, P.hh, P.W, P.H, T)) continue;
            const int32_t wpx = T.px1 - T.px0 + 1;
            for (int32_t py = T.py0 + (int32_t)part; py <= T.py1; py += kGenericSplit) {
                if (!row_owned(P, (uint32_t)py)) continue;
                uint32_t *row = vis + (size_t)local_row(P, (uint32_t)py) * P.W;
                for (int32_t dx = threadIdx.x; dx < wpx; dx += 256) {
                    int64_t e[3];
                    if (covers(T, T.px0 + dx, py, e)) atomicMax(row + T.px0 + dx, word);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fragment stage: visibility -> RGBA8
// ---------------------------------------------------------------------------------------------
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

__device__ inline uint32_t shade_pixel(const FrameParams &P, const AxisTables &A, const float *__restrict__ tex,
                                       const ShadeTables &S, uint32_t prim, int32_t px, int32_t py)
{
    GVert v[3];
    load_prim(P, A, tex, prim, v);
    bool plain = true;
    for (int k = 0; k < 3; ++k) plain &= finite4(v[k].x, v[k].y, v[k].z, v[k].w) && !(v[k].z < 0.0f) && !(v[k].z > v[k].w);
    float attr[3] = { 0.f, 0.f, 0.f };
    bool hit = false;
    if (plain) {
        TriSetup T;
        int64_t e[3];
        if (setup_triangle(v[0], v[1], v[2], P.hw, P.hh, P.W, P.H, T) && covers(T, px, py, e)) { interpolate(T, e, attr); hit = true; }
    } else {
        GVert poly[8];
        const int np = clip_primitive(v, poly);
        for (int f = 1; f + 1 < np; ++f) {      // the last covering piece wins, as in the draw order
            TriSetup T;
            int64_t e[3];
            if (setup_triangle(poly[0], poly[f], poly[f + 1], P.hw, P.hh, P.W, P.H, T) && covers(T, px, py, e)) { interpolate(T, e, attr); hit = true; }
        }
    }
    if (!hit) return P.clear_rgba;   // unreachable when the visibility buffer is consistent
    return fragment_shader(P, S, attr);
}

template <int PPT>
__global__ __launch_bounds__(256) void k_resolve(FrameParams P, AxisTables A, const float *__restrict__ tex,
                                                 const float *__restrict__ lut_linear, const float *__restrict__ thresh,
                                                 const uint32_t *__restrict__ vis, uint32_t *__restrict__ rgba, uint32_t *counters)
{
    __shared__ float s_lut[256 * 3];
    __shared__ float s_thr[256];
    for (int k = threadIdx.x; k < 768; k += 256) s_lut[k] = lut_linear[k];
    s_thr[threadIdx.x] = thresh[threadIdx.x];
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // publish this frame's statistics, re-arm for the next frame
        counters[2] = counters[0]; counters[3] = counters[1];
        counters[0] = 0; counters[1] = 0;
    }
    ShadeTables S = { s_lut, s_thr };
    const size_t npx = (size_t)P.local_rows * P.W;
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * PPT;
    if (base >= npx) return;
    uint32_t v[PPT], o[PPT];
    if (PPT == 4) { uint4 q = *reinterpret_cast<const uint4 *>(vis + base); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
    else v[0] = vis[base];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        uint32_t word = v[k];
        uint32_t id = word & kPrimMask;
        bool live = P.tag ? ((word & ~kPrimMask) == P.tag && id != 0u) : (word != 0u);
        if (!P.tag) id = word;
        if (!live) { o[k] = P.clear_rgba; continue; }
        size_t p = base + k;
        uint32_t ly = (uint32_t)(p / P.W), px = (uint32_t)(p - (size_t)ly * P.W);
        o[k] = shade_pixel(P, A, tex, S, id - 1u, (int32_t)px, (int32_t)global_row(P, ly));
    }
    if (PPT == 4) *reinterpret_cast<uint4 *>(rgba + base) = make_uint4(o[0], o[1], o[2], o[3]);
    else rgba[base] = o[0];
}

// decode the tagged visibility words into prim+1 / 0 (debug + parity tests)
__global__ void k_decode_vis(FrameParams P, const uint32_t *__restrict__ vis, uint32_t *__restrict__ out)
{
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (size_t)P.local_rows * P.W) return;
    uint32_t word = vis[p];
    if (P.tag) out[p] = ((word & ~kPrimMask) == P.tag) ? (word & kPrimMask) : 0u;
    else out[p] = word;
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
