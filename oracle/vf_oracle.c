/*
 * vf_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference's headless terrain-raster hot path
 * (milos-agathon/vulkan-forge @ 2025-08-15).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (vulkan_forge_amd/csrc + host) never links, imports or calls it.
 *
 * PARITY STATUS
 *   - grid_generate, camera matrices, uniform block, LUT bytes, clear colour:
 *     PINNED by the reference's own known-answer tests (tests/test_grid_generate.py,
 *     tests/test_camera.py, tests/test_t31_integration.py, src/terrain/mesh.rs:92-129,
 *     src/terrain/mod.rs:699-732) -- see tests/test_oracle_pins.py.
 *   - rendered pixels: PARITY UNPINNED.  The reference holds no golden image and cannot
 *     be built or run here (no Rust toolchain, no Vulkan adapter).  Rasterisation,
 *     texture sampling and sRGB conversion execute inside wgpu 0.19.4 + the platform
 *     driver (Cargo.lock:1114-1211), which are not in the tree; their published rules
 *     (WebGPU/Vulkan: top-left fill rule, pixel-centre sampling, primitive order,
 *     perspective-correct interpolation, sRGB transfer functions) are restated below
 *     with the fixed conventions listed in DESIGN.md "Raster conventions".
 *
 * Every function cites the reference file:line it follows.
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; no fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define VFO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * Deterministic elementary functions.
 * WGSL leaves sin/cos precision implementation-defined (driver); we fix one algorithm so
 * that the oracle and the HIP kernels agree bit-for-bit: Cody-Waite reduction by pi/2 in
 * three parts + degree-7/8 minimax polynomials (Cephes single-precision coefficients),
 * every multiply-add an explicit IEEE fmaf.  Max error vs libm ~1 ulp on |x| < 100.
 * ---------------------------------------------------------------------------------------- */
static inline float o_sin_poly(float r)
{
    float r2 = r * r;
    float p = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    p = fmaf(r2, p, -1.6666654611e-1f);
    return fmaf(r * r2, p, r);
}
static inline float o_cos_poly(float r)
{
    float r2 = r * r;
    float p = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    p = fmaf(r2, p, 4.166664568298827e-2f);
    return fmaf(r2 * r2, p, fmaf(-0.5f, r2, 1.0f));
}
static inline float o_reduce(float x, int *q)
{
    float k = rintf(x * 0.636619772f); /* 2/pi */
    float r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188e-8f, r);
    *q = ((int)k) & 3;
    return r;
}
static float o_sinf(float x)
{
    int q; float r = o_reduce(x, &q);
    float s = o_sin_poly(r), c = o_cos_poly(r);
    switch (q) { case 0: return s; case 1: return c; case 2: return -s; default: return -c; }
}
static float o_cosf(float x)
{
    int q; float r = o_reduce(x, &q);
    float s = o_sin_poly(r), c = o_cos_poly(r);
    switch (q) { case 0: return c; case 1: return -s; case 2: return -c; default: return s; }
}
VFO_API void vfo_sincos(const float *x, float *s, float *c, int n)
{
    for (int i = 0; i < n; ++i) { s[i] = o_sinf(x[i]); c[i] = o_cosf(x[i]); }
}

/* ------------------------------------------------------------------------------------------
 * sRGB transfer functions.
 *   decode: what sampling an Rgba8UnormSrgb texture does per texel (src/terrain/mod.rs:50-54).
 *   encode: what storing to the Rgba8UnormSrgb target does (src/terrain/mod.rs:219,304) --
 *           the ideal conversion round(255*oetf(c)), expressed through 255 float thresholds:
 *           byte = #{k in 1..255 : c >= T[k]}, T[k] = float(eotf((k-0.5)/255)).
 * ---------------------------------------------------------------------------------------- */
static float g_srgb_decode[256];
static float g_srgb_thresh[256]; /* [0] unused (= -inf) */
static int g_tables_ready = 0;
static double eotf_d(double s) { return s <= 0.04045 ? s / 12.92 : pow((s + 0.055) / 1.055, 2.4); }
static void init_tables(void)
{
    if (g_tables_ready) return;
    for (int k = 0; k < 256; ++k) {
        g_srgb_decode[k] = (float)eotf_d((double)k / 255.0);
        g_srgb_thresh[k] = k == 0 ? -INFINITY : (float)eotf_d(((double)k - 0.5) / 255.0);
    }
    g_tables_ready = 1;
}
static inline uint8_t srgb_encode(float c)
{
    /* binary search for the number of thresholds <= c (NaN -> 0, like a clamp through max/min) */
    int lo = 0, hi = 255; /* invariant: T[lo] <= c < T[hi+1] */
    if (!(c >= g_srgb_thresh[1])) return 0;
    lo = 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (c >= g_srgb_thresh[mid]) lo = mid; else hi = mid - 1;
    }
    return (uint8_t)lo;
}
VFO_API void vfo_srgb_tables(float *decode256, float *thresh256)
{
    init_tables();
    memcpy(decode256, g_srgb_decode, sizeof g_srgb_decode);
    memcpy(thresh256, g_srgb_thresh, sizeof g_srgb_thresh);
}
VFO_API void vfo_srgb_encode(const float *c, uint8_t *out, int n)
{
    init_tables();
    for (int i = 0; i < n; ++i) out[i] = srgb_encode(c[i]);
}

/* src/colormap/mod.rs:59-79  to_linear_u8_rgba (the VF_FORCE_LUT_UNORM fallback bytes). */
VFO_API void vfo_lut_to_linear_u8(const uint8_t *src, uint8_t *dst, int ntexels)
{
    for (int i = 0; i < ntexels; ++i) {
        for (int ch = 0; ch < 3; ++ch) {
            float s = (float)src[4 * i + ch] / 255.0f;
            float l = s <= 0.04045f ? s / 12.92f : powf((s + 0.055f) / 1.055f, 2.4f);
            l = l < 0.0f ? 0.0f : (l > 1.0f ? 1.0f : l);
            dst[4 * i + ch] = (uint8_t)(l * 255.0f + 0.5f);
        }
        dst[4 * i + 3] = src[4 * i + 3];
    }
}

/* ------------------------------------------------------------------------------------------
 * Camera math: glam 0.24.2 (Cargo.lock:297-298; crate not vendored -- published formulas,
 * scalar f32, no FMA) as called from src/camera.rs.
 * Matrices are column-major float[16] like glam::Mat4::to_cols_array.
 * ---------------------------------------------------------------------------------------- */
static const char *ERR_FOVY = "fovy_deg must be finite and in (0, 180)";     /* src/camera.rs:24 */
static const char *ERR_NEAR = "znear must be finite and > 0";                /* :25 */
static const char *ERR_FAR = "zfar must be finite and > znear";              /* :26 */
static const char *ERR_ASPECT = "aspect must be finite and > 0";             /* :27 */
static const char *ERR_VECFINITE = "eye/target/up components must be finite";/* :28 */
static const char *ERR_UPCOLINEAR = "up vector must not be colinear with view direction"; /* :29 */
static const char *ERR_CLIP = "clip_space must be 'wgpu' or 'gl'";           /* :30 */

typedef struct { float x, y, z; } v3;
static inline v3 v3sub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline float v3dot(v3 a, v3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
static inline v3 v3cross(v3 a, v3 b)
{
    v3 r = { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y };
    return r;
}
static inline v3 v3scale(v3 a, float s) { v3 r = { a.x * s, a.y * s, a.z * s }; return r; }
static inline v3 v3normalize(v3 a) { return v3scale(a, 1.0f / sqrtf(v3dot(a, a))); }
static inline v3 v3normalize_or_zero(v3 a)
{
    float rcp = 1.0f / sqrtf(v3dot(a, a));
    if (isfinite(rcp) && rcp > 0.0f) return v3scale(a, rcp);
    v3 z = { 0, 0, 0 };
    return z;
}
static inline int v3finite(v3 a) { return isfinite(a.x) && isfinite(a.y) && isfinite(a.z); }

/* glam Mat4::look_at_rh -> look_to_rh(eye, center - eye, up) */
static void look_at_rh(v3 eye, v3 center, v3 up, float m[16])
{
    v3 f = v3normalize(v3sub(center, eye));
    v3 s = v3normalize(v3cross(f, up));
    v3 u = v3cross(s, f);
    m[0] = s.x; m[1] = u.x; m[2] = -f.x; m[3] = 0.0f;
    m[4] = s.y; m[5] = u.y; m[6] = -f.y; m[7] = 0.0f;
    m[8] = s.z; m[9] = u.z; m[10] = -f.z; m[11] = 0.0f;
    m[12] = -v3dot(eye, s); m[13] = -v3dot(eye, u); m[14] = v3dot(eye, f); m[15] = 1.0f;
}
/* glam Mat4::perspective_rh_gl */
static void perspective_rh_gl(float fovy, float aspect, float zn, float zf, float m[16])
{
    float inv_length = 1.0f / (zn - zf);
    float f = 1.0f / tanf(0.5f * fovy);
    float a = f / aspect;
    float b = (zn + zf) * inv_length;
    float c = (2.0f * zn * zf) * inv_length;
    memset(m, 0, 16 * sizeof(float));
    m[0] = a; m[5] = f; m[10] = b; m[11] = -1.0f; m[14] = c;
}
/* glam Mat4 * Mat4: column j = ((A.c0*b.x + A.c1*b.y) + A.c2*b.z) + A.c3*b.w */
static void mat4_mul(const float A[16], const float B[16], float out[16])
{
    float r[16];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r[4 * j + i] = ((A[i] * B[4 * j] + A[4 + i] * B[4 * j + 1]) + A[8 + i] * B[4 * j + 2]) + A[12 + i] * B[4 * j + 3];
    memcpy(out, r, sizeof r);
}
/* src/camera.rs:14-21 -- the literal is fed to from_cols_array, i.e. it is read column by
 * column: columns (1,0,0,0),(0,1,0,0),(0,0,.5,.5),(0,0,0,1).  Effect: z' = .5 z, w' = w + .5 z
 * (SURVEY.md finding 2; NOT the conventional depth remap -- reproduced as coded). */
static const float GL_TO_WGPU[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0.5f, 0.5f, 0, 0, 0, 1 };
/* src/camera.rs:218-221 */
static void perspective_wgpu(float fovy_rad, float aspect, float zn, float zf, float m[16])
{
    float gl[16];
    perspective_rh_gl(fovy_rad, aspect, zn, zf, gl);
    mat4_mul(GL_TO_WGPU, gl, m);
}
static inline float to_radians(float deg) { return deg * (3.14159265358979323846f / 180.0f); }

/* src/camera.rs:33-91 validators, in the order validate_camera_params applies them (:224-240) */
static const char *validate_vectors(v3 eye, v3 target, v3 up)
{
    if (!v3finite(eye) || !v3finite(target) || !v3finite(up)) return ERR_VECFINITE;
    v3 d = v3normalize_or_zero(v3sub(target, eye));
    v3 un = v3normalize_or_zero(up);
    v3 c = v3cross(d, un);
    if (v3dot(c, c) < 1e-6f) return ERR_UPCOLINEAR;
    return NULL;
}
static const char *validate_fovy(float f) { return (!isfinite(f) || f <= 0.0f || f >= 180.0f) ? ERR_FOVY : NULL; }
static const char *validate_near(float n) { return (!isfinite(n) || n <= 0.0f) ? ERR_NEAR : NULL; }
static const char *validate_far(float f, float n) { return (!isfinite(f) || f <= n) ? ERR_FAR : NULL; }
static const char *validate_aspect(float a) { return (!isfinite(a) || a <= 0.0f) ? ERR_ASPECT : NULL; }

/* transposes column-major glam data into the (4,4) row-major array mat4_to_numpy returns (:94-112) */
static void to_rowmajor(const float m[16], float out[16])
{
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[4 * r + c] = m[4 * c + r];
}

/* src/camera.rs:117-135.  Returns NULL or the exact error string (raised as RuntimeError). */
VFO_API const char *vfo_camera_look_at(const float eye[3], const float target[3], const float up[3], float out_rm[16])
{
    v3 e = { eye[0], eye[1], eye[2] }, t = { target[0], target[1], target[2] }, u = { up[0], up[1], up[2] };
    const char *err = validate_vectors(e, t, u);
    if (err) return err;
    float m[16];
    look_at_rh(e, t, u, m);
    to_rowmajor(m, out_rm);
    return NULL;
}
/* src/camera.rs:140-169.  clip: 0 = "wgpu", 1 = "gl", anything else -> ERR_CLIP */
VFO_API const char *vfo_camera_perspective(float fovy_deg, float aspect, float zn, float zf, int clip, float out_rm[16])
{
    const char *err;
    if ((err = validate_fovy(fovy_deg))) return err;
    if ((err = validate_aspect(aspect))) return err;
    if ((err = validate_near(zn))) return err;
    if ((err = validate_far(zf, zn))) return err;
    if (clip != 0 && clip != 1) return ERR_CLIP;
    float m[16];
    if (clip == 1) perspective_rh_gl(to_radians(fovy_deg), aspect, zn, zf, m);
    else perspective_wgpu(to_radians(fovy_deg), aspect, zn, zf, m);
    to_rowmajor(m, out_rm);
    return NULL;
}
/* src/camera.rs:174-215 */
VFO_API const char *vfo_camera_view_proj(const float eye[3], const float target[3], const float up[3],
                                         float fovy_deg, float aspect, float zn, float zf, int clip, float out_rm[16])
{
    v3 e = { eye[0], eye[1], eye[2] }, t = { target[0], target[1], target[2] }, u = { up[0], up[1], up[2] };
    const char *err = validate_vectors(e, t, u);
    if (err) return err;
    if ((err = validate_fovy(fovy_deg))) return err;
    if ((err = validate_aspect(aspect))) return err;
    if ((err = validate_near(zn))) return err;
    if ((err = validate_far(zf, zn))) return err;
    if (clip != 0 && clip != 1) return ERR_CLIP;
    float v[16], p[16], vp[16];
    look_at_rh(e, t, u, v);
    if (clip == 1) perspective_rh_gl(to_radians(fovy_deg), aspect, zn, zf, p);
    else perspective_wgpu(to_radians(fovy_deg), aspect, zn, zf, p);
    mat4_mul(p, v, vp);
    to_rowmajor(vp, out_rm);
    return NULL;
}

/* ------------------------------------------------------------------------------------------
 * Uniform block: TerrainUniforms (src/terrain/mod.rs:114-175), Globals (:178-215).
 * kind 0 = TerrainSpike defaults (build_view_matrices :681-691, sun override :325-327),
 * kind 1 = Scene defaults (src/scene/mod.rs:17-23,118-122).
 * ---------------------------------------------------------------------------------------- */
static void fill_uniforms(const float view[16], const float proj[16], v3 sun, float u[44])
{
    memcpy(u, view, 64);
    memcpy(u + 16, proj, 64);
    u[32] = sun.x; u[33] = sun.y; u[34] = sun.z; u[35] = 1.0f;          /* exposure 1 (:192) */
    u[36] = 1.0f; u[37] = 0.5f - (-0.5f); u[38] = 1.0f; u[39] = 0.0f;   /* spacing, h_max-h_min, exag (:193-198) */
    u[40] = u[41] = u[42] = u[43] = 0.0f;
}
static v3 default_sun(int kind)
{
    v3 spike = { 0.5f, 1.0f, 0.3f }, scene = { 0.5f, 0.8f, 0.6f };
    return v3normalize(kind == 0 ? spike : scene);
}
VFO_API void vfo_default_uniforms(int kind, uint32_t W, uint32_t H, float u[44])
{
    float view[16], proj[16];
    v3 eye = { 3, 2, 3 }, zero = { 0, 0, 0 }, up = { 0, 1, 0 };
    look_at_rh(eye, zero, up, view);
    perspective_wgpu(to_radians(45.0f), (float)W / (float)H, 0.1f, 100.0f, proj);
    fill_uniforms(view, proj, default_sun(kind), u);
}
/* set_camera_look_at: src/terrain/mod.rs:498-535, src/scene/mod.rs:208-224 (aspect = W/H) */
VFO_API const char *vfo_look_at_uniforms(int kind, uint32_t W, uint32_t H, const float eye[3], const float target[3],
                                         const float up[3], float fovy_deg, float zn, float zf, float u[44])
{
    v3 e = { eye[0], eye[1], eye[2] }, t = { target[0], target[1], target[2] }, up3 = { up[0], up[1], up[2] };
    const char *err = validate_vectors(e, t, up3);
    if (err) return err;
    if ((err = validate_fovy(fovy_deg))) return err;
    if ((err = validate_near(zn))) return err;
    if ((err = validate_far(zf, zn))) return err;
    float view[16], proj[16];
    look_at_rh(e, t, up3, view);
    perspective_wgpu(to_radians(fovy_deg), (float)W / (float)H, zn, zf, proj);
    fill_uniforms(view, proj, default_sun(kind), u);
    return NULL;
}

/* ------------------------------------------------------------------------------------------
 * grid_generate: make_grid (src/terrain/mesh.rs:35-90) + the PyO3 wrapper's validation
 * (:157-173).  xy/uv are (nx*nz,2) float32 row-major, idx is 6*(nx-1)*(nz-1) uint32.
 * Returns NULL or the exact ValueError string.
 * ---------------------------------------------------------------------------------------- */
VFO_API const char *vfo_grid_generate(uint32_t nx, uint32_t nz, float dx, float dy, const char *origin,
                                      float *xy, float *uv, uint32_t *idx)
{
    if (nx < 2 || nz < 2) return "nx and nz must be >= 2";
    if (!isfinite(dx) || !isfinite(dy) || dx <= 0.0f || dy <= 0.0f) return "spacing components must be finite and > 0";
    if (origin && strcmp(origin, "center") != 0) return "origin must be 'center'";
    size_t w = nx, h = nz;
    float cx = ((float)w - 1.0f) * 0.5f * dx;
    float cy = ((float)h - 1.0f) * 0.5f * dy;
    for (size_t y = 0; y < h; ++y) {
        float wy = (float)y * dy - cy;
        float v = (float)y / ((float)h - 1.0f);
        for (size_t x = 0; x < w; ++x) {
            float wx = (float)x * dx - cx;
            float uu = (float)x / ((float)w - 1.0f);
            size_t k = y * w + x;
            xy[2 * k] = wx; xy[2 * k + 1] = wy;
            uv[2 * k] = uu; uv[2 * k + 1] = v;
        }
    }
    size_t o = 0;
    for (size_t y = 0; y + 1 < h; ++y) {
        size_t row = y * w;
        for (size_t x = 0; x + 1 < w; ++x) {
            uint32_t i0 = (uint32_t)(row + x), i1 = i0 + 1, i2 = (uint32_t)(row + x + w), i3 = i2 + 1;
            idx[o++] = i0; idx[o++] = i1; idx[o++] = i2; idx[o++] = i2; idx[o++] = i1; idx[o++] = i3;
        }
    }
    return NULL;
}
/* u16/u32 index-width switch, src/terrain/mesh.rs:30-32 (known answer :123-129) */
VFO_API int vfo_grid_uses_u16(uint32_t nx, uint32_t nz) { return (size_t)nx * nz <= 65535u; }

/* Render mesh: build_grid_xyuv src/terrain/mod.rs:553-598 (twin: src/scene/mod.rs:85-116).
 * verts = n*n*[x,z,u,v], idx = 6*(n-1)^2 u32 with per-cell order [a,c,b,b,c,d]. */
VFO_API void vfo_build_grid_xyuv(uint32_t n_in, float *verts, uint32_t *idx)
{
    size_t n = n_in < 2 ? 2 : n_in, w = n, h = n;
    float scale = 1.5f;
    float step_x = (2.0f * scale) / ((float)w - 1.0f);
    float step_z = (2.0f * scale) / ((float)h - 1.0f);
    for (size_t j = 0; j < h; ++j)
        for (size_t i = 0; i < w; ++i) {
            float *p = verts + 4 * (j * w + i);
            p[0] = -scale + (float)i * step_x;
            p[1] = -scale + (float)j * step_z;
            p[2] = (float)i / ((float)w - 1.0f);
            p[3] = (float)j / ((float)h - 1.0f);
        }
    if (!idx) return;
    size_t o = 0;
    for (size_t j = 0; j + 1 < h; ++j)
        for (size_t i = 0; i + 1 < w; ++i) {
            uint32_t a = (uint32_t)(j * w + i), b = a + 1, c = (uint32_t)((j + 1) * w + i), d = c + 1;
            idx[o++] = a; idx[o++] = c; idx[o++] = b; idx[o++] = b; idx[o++] = c; idx[o++] = d;
        }
}

/* ------------------------------------------------------------------------------------------
 * Software rasteriser restating what `draw_indexed` asks of the driver
 * (src/terrain/mod.rs:412-437, src/scene/mod.rs:280-298, src/lib.rs:693-719) under the fixed
 * state of src/terrain/pipeline.rs:97-139: TriangleList, front = CCW, cull Back, no depth,
 * no blend, MSAA 1, full-target viewport, clip volume 0 <= z <= w.
 * ---------------------------------------------------------------------------------------- */
typedef struct { float x, y, z, w; float a[3]; } cvert;           /* clip-space vertex + 3 varyings */
typedef struct { int32_t X, Y; float rw; float a[3]; } svert;      /* snapped screen vertex */

typedef struct {
    uint32_t W, H;
    uint8_t *rgba;        /* W*H*4 */
    uint32_t *vis;        /* W*H  (prim+1; 0 = background) */
    int pass;             /* 0: visibility pass (which primitive's fragment survives at each pixel)
                             1: fragment pass (evaluate fs_main for the surviving fragment only) */
    int atomic_vis;       /* visibility pass runs on several threads: merge with an atomic max */
    int32_t sc_x0, sc_x1, sc_y0, sc_y1;   /* inclusive scissor (fragment pass: the one pixel being shaded) */
    /* row ownership (multi-GPU band split, DESIGN.md "Sharding"): pixel row y is rendered iff
       ((y / band_h) % nranks) == rank.  nranks == 1 -> every row. */
    uint32_t rank, nranks, band_h;
    /* fragment constants */
    int mode;             /* 0 terrain fs_main, 1 triangle fs_main */
    float h_range, exposure, Lx, Ly, Lz;
    float lut[256][3];
    /* SPEC_T32 fragment mode (no reference implementation, see frag_terrain) */
    int shade_mode;       /* 0 REFERENCE = what terrain.wgsl does, 1 SPEC_T32 = what ROADMAP.md T3.2 / README.md describe */
    const float *tex; uint32_t tw, th;
    float spacing, exag;
} rtarget;

typedef void (*frag_fn)(const rtarget *, const float attr[3], uint8_t out[4]);

/* fs_main src/shaders/terrain.wgsl:69-91 + the Rgba8UnormSrgb store */
static void frag_terrain(const rtarget *rt, const float attr[3], uint8_t out[4])
{
    float height = attr[0], x = attr[1], z = attr[2];
    float t = 0.5f + height / (2.0f * rt->h_range);                       /* :72-73 */
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    /* textureSampleLevel(lut, linear clamp, (t,.5)) :76 -- texel centres at (i+.5)/256 */
    float c = t * 256.0f - 0.5f;
    float i0f = floorf(c);
    float f = c - i0f;
    int i0 = (int)i0f, i1 = i0 + 1;
    i0 = i0 < 0 ? 0 : (i0 > 255 ? 255 : i0);
    i1 = i1 < 0 ? 0 : (i1 > 255 ? 255 : i1);
    float nx, ny, nz;
    if (rt->shade_mode == 0) {
        float dhdx = 1.3f * o_cosf(x * 1.3f) * 0.25f;                      /* :79 */
        float dhdz = -1.1f * o_sinf(z * 1.1f) * 0.25f;                     /* :80 */
        float d = fmaf(dhdz, dhdz, fmaf(dhdx, dhdx, 1.0f));
        float inv = 1.0f / sqrtf(d);                                       /* normalize :81 */
        nx = -dhdx * inv; ny = inv; nz = -dhdz * inv;
    } else {
        /* SPEC_T32 -- the fragment stage the reference DOCUMENTS but does not implement (ROADMAP.md:421-436,
         * README.md:128,174-175): forward-difference normals from the height texture, Lambert + ambient, LUT, Reinhard in
         * linear, sRGB store.  No reference code exists for it; these are this build's choices, checked only HIP <-> oracle:
         *  - uv from the interpolated xz varying (the mesh spans [-1.5, 1.5]: u = x/3 + 1/2);
         *  - h, hx, hy = nearest texels at uv, uv + (1/(Tw-1), 0), uv + (0, 1/(Th-1)) (ROADMAP.md:427-429), texture only;
         *  - tangents (spacing, (hx-h)*exag, 0) and (0, (hy-h)*exag, spacing) in the Y-up world, normal = their upward
         *    cross product (the ROADMAP snippet's cross(dpy, dpx) points down in its own z-up frame; "sun from the east
         *    lights the east slopes" :433 needs the upward one). */
        const float third = 1.0f / 3.0f;
        float uu = fmaf(x, third, 0.5f), vv = fmaf(z, third, 0.5f);
        float du = 1.0f / (float)((rt->tw > 2u ? rt->tw : 2u) - 1u), dv = 1.0f / (float)((rt->th > 2u ? rt->th : 2u) - 1u);
        int tx0 = (int)floorf(uu * (float)rt->tw), tx1 = (int)floorf((uu + du) * (float)rt->tw);
        int ty0 = (int)floorf(vv * (float)rt->th), ty1 = (int)floorf((vv + dv) * (float)rt->th);
        int mx = (int)rt->tw - 1, my = (int)rt->th - 1;
        tx0 = tx0 < 0 ? 0 : (tx0 > mx ? mx : tx0); tx1 = tx1 < 0 ? 0 : (tx1 > mx ? mx : tx1);
        ty0 = ty0 < 0 ? 0 : (ty0 > my ? my : ty0); ty1 = ty1 < 0 ? 0 : (ty1 > my ? my : ty1);
        float h0 = rt->tex[(size_t)ty0 * rt->tw + tx0], hx = rt->tex[(size_t)ty0 * rt->tw + tx1], hy = rt->tex[(size_t)ty1 * rt->tw + tx0];
        float ax = (hx - h0) * rt->exag, az = (hy - h0) * rt->exag, sp = rt->spacing;
        float vx = -(ax * sp), vy = sp * sp, vz = -(sp * az);
        float d = fmaf(vz, vz, fmaf(vy, vy, vx * vx));
        float inv = 1.0f / sqrtf(d);
        nx = vx * inv; ny = vy * inv; nz = vz * inv;
    }
    float ndl = fmaf(nz, rt->Lz, fmaf(ny, rt->Ly, nx * rt->Lx));
    float lambert = fminf(fmaxf(ndl, 0.0f), 1.0f);                         /* :84 */
    float shade = 0.15f * (1.0f - lambert) + lambert;                      /* mix(.15,1,lambert) :88 */
    for (int ch = 0; ch < 3; ++ch) {
        float l0 = rt->lut[i0][ch], l1 = rt->lut[i1][ch];
        float lc = fmaf(f, l1 - l0, l0);
        float v = lc * rt->exposure * shade;                               /* :90 */
        if (rt->shade_mode != 0) v = v / (1.0f + v);                       /* Reinhard, tests/test_tonemap.py:7-8, in linear (README.md:128) */
        out[ch] = srgb_encode(v);
    }
    out[3] = 255;
}
/* fs_main src/shaders/triangle.wgsl:20-23 */
static void frag_triangle(const rtarget *rt, const float attr[3], uint8_t out[4])
{
    (void)rt;
    for (int ch = 0; ch < 3; ++ch) out[ch] = srgb_encode(attr[ch]);
    out[3] = 255;
}

static inline int row_owned(const rtarget *rt, int y)
{
    return rt->nranks <= 1 || (((uint32_t)y / rt->band_h) % rt->nranks) == rt->rank;
}

/* Painter's order without depth or blending: fragments reach a pixel in primitive order and the
 * last one survives.  fs_main has no side effects, so only the survivor's colour is observable:
 * pass 0 records which primitive survives (a later primitive simply overwrites; with several
 * threads "later" = larger id, merged by atomic max), pass 1 re-draws that primitive under a
 * one-pixel scissor with the very same code and evaluates fs_main for its fragment. */
static inline void store_visibility(const rtarget *rt, int px, int py, uint32_t prim)
{
    size_t o = (size_t)py * rt->W + px;
    if (rt->atomic_vis) {
        uint32_t k = prim + 1, cur = __atomic_load_n(&rt->vis[o], __ATOMIC_RELAXED);
        while (k > cur && !__atomic_compare_exchange_n(&rt->vis[o], &cur, k, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    } else {
        rt->vis[o] = prim + 1;
    }
}

static inline int64_t edge_fn(const svert *a, const svert *b, int64_t Px, int64_t Py)
{
    return (int64_t)(b->X - a->X) * (Py - a->Y) - (int64_t)(b->Y - a->Y) * (Px - a->X);
}
static inline int32_t iclamp(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* One unclipped triangle: viewport, snap, cull, scan, interpolate, shade. */
static void raster_triangle(const rtarget *rt, frag_fn fs, const cvert v[3], uint32_t prim)
{
    svert s[3];
    const float hw = 0.5f * (float)rt->W, hh = 0.5f * (float)rt->H;
    for (int k = 0; k < 3; ++k) {
        if (!(v[k].w > 0.0f)) return;
        float rw = 1.0f / v[k].w;
        float xf = fmaf(v[k].x * rw, hw, hw);
        float yf = fmaf(-(v[k].y * rw), hh, hh);
        if (!isfinite(xf) || !isfinite(yf)) return;
        xf = fminf(fmaxf(xf, -4194304.0f), 4194304.0f);   /* guard-band saturation, DESIGN.md */
        yf = fminf(fmaxf(yf, -4194304.0f), 4194304.0f);
        s[k].X = (int32_t)rintf(xf * 256.0f);              /* 8 sub-pixel bits, round-half-even */
        s[k].Y = (int32_t)rintf(yf * 256.0f);
        s[k].rw = rw;
        memcpy(s[k].a, v[k].a, sizeof s[k].a);
    }
    int64_t area2 = (int64_t)(s[1].X - s[0].X) * (s[2].Y - s[0].Y) - (int64_t)(s[1].Y - s[0].Y) * (s[2].X - s[0].X);
    if (area2 >= 0) return; /* back-facing (CW in Y-up NDC) or degenerate: culled */
    int32_t xmin = s[0].X < s[1].X ? s[0].X : s[1].X; if (s[2].X < xmin) xmin = s[2].X;
    int32_t xmax = s[0].X > s[1].X ? s[0].X : s[1].X; if (s[2].X > xmax) xmax = s[2].X;
    int32_t ymin = s[0].Y < s[1].Y ? s[0].Y : s[1].Y; if (s[2].Y < ymin) ymin = s[2].Y;
    int32_t ymax = s[0].Y > s[1].Y ? s[0].Y : s[1].Y; if (s[2].Y > ymax) ymax = s[2].Y;
    /* pixel centres (px+.5) inside [min,max]:  px >= (min-128)/256,  px <= (max-128)/256 */
    int32_t px0 = (xmin + 127) >> 8, px1 = (xmax - 128) >> 8;
    int32_t py0 = (ymin + 127) >> 8, py1 = (ymax - 128) >> 8;
    px0 = px0 < 0 ? 0 : px0; py0 = py0 < 0 ? 0 : py0;
    if (px1 > (int32_t)rt->W - 1) px1 = (int32_t)rt->W - 1;
    if (py1 > (int32_t)rt->H - 1) py1 = (int32_t)rt->H - 1;
    if (px0 < rt->sc_x0) px0 = rt->sc_x0;
    if (py0 < rt->sc_y0) py0 = rt->sc_y0;
    if (px1 > rt->sc_x1) px1 = rt->sc_x1;
    if (py1 > rt->sc_y1) py1 = rt->sc_y1;
    if (px0 > px1 || py0 > py1) return;
    const float fA = (float)(-area2);
    /* inside-positive edge weights e_i = -E_{jk}(P); top-left rule on their gradients */
    const int32_t a0 = s[2].Y - s[1].Y, b0 = -(s[2].X - s[1].X);
    const int32_t a1 = s[0].Y - s[2].Y, b1 = -(s[0].X - s[2].X);
    const int32_t a2 = s[1].Y - s[0].Y, b2 = -(s[1].X - s[0].X);
    const int tl0 = a0 > 0 || (a0 == 0 && b0 > 0);
    const int tl1 = a1 > 0 || (a1 == 0 && b1 > 0);
    const int tl2 = a2 > 0 || (a2 == 0 && b2 > 0);
    for (int32_t py = py0; py <= py1; ++py) {
        if (!row_owned(rt, py)) continue;
        int64_t Py = (int64_t)py * 256 + 128;
        for (int32_t px = px0; px <= px1; ++px) {
            int64_t Px = (int64_t)px * 256 + 128;
            int64_t e0 = -edge_fn(&s[1], &s[2], Px, Py);
            int64_t e1 = -edge_fn(&s[2], &s[0], Px, Py);
            int64_t e2 = -edge_fn(&s[0], &s[1], Px, Py);
            if (!(e0 > 0 || (e0 == 0 && tl0))) continue;
            if (!(e1 > 0 || (e1 == 0 && tl1))) continue;
            if (!(e2 > 0 || (e2 == 0 && tl2))) continue;
            if (rt->pass == 0) { store_visibility(rt, px, py, prim); continue; }
            float l0 = (float)e0 / fA, l1 = (float)e1 / fA, l2 = (float)e2 / fA;
            float q0 = l0 * s[0].rw, q1 = l1 * s[1].rw, q2 = l2 * s[2].rw;
            float rQ = 1.0f / ((q0 + q1) + q2);
            float attr[3];
            for (int k = 0; k < 3; ++k)
                attr[k] = fmaf(q2, s[2].a[k], fmaf(q1, s[1].a[k], q0 * s[0].a[k])) * rQ;
            fs(rt, attr, rt->rgba + 4 * ((size_t)py * rt->W + px));
        }
    }
}

/* Clip against 0 <= z (near) and z <= w (far) -- `unclipped_depth: false`
 * (src/terrain/pipeline.rs:128).  Sutherland-Hodgman; the crossing point is always computed
 * from the inside vertex towards the outside one so shared edges clip identically. */
static inline float plane_dist(const cvert *v, int plane) { return plane == 0 ? v->z : v->w - v->z; }
static void lerp_vert(const cvert *in, const cvert *out, float t, cvert *r)
{
    r->x = fmaf(t, out->x - in->x, in->x);
    r->y = fmaf(t, out->y - in->y, in->y);
    r->z = fmaf(t, out->z - in->z, in->z);
    r->w = fmaf(t, out->w - in->w, in->w);
    for (int k = 0; k < 3; ++k) r->a[k] = fmaf(t, out->a[k] - in->a[k], in->a[k]);
}
static void draw_primitive(const rtarget *rt, frag_fn fs, const cvert v[3], uint32_t prim)
{
    for (int k = 0; k < 3; ++k)
        if (!isfinite(v[k].x) || !isfinite(v[k].y) || !isfinite(v[k].z) || !isfinite(v[k].w)) return;
    int out_near = 0, out_far = 0;
    for (int k = 0; k < 3; ++k) { out_near += v[k].z < 0.0f; out_far += v[k].z > v[k].w; }
    if (out_near == 3 || out_far == 3) return;
    if (out_near == 0 && out_far == 0) { raster_triangle(rt, fs, v, prim); return; }
    cvert poly[8], tmp[8];
    int n = 3;
    memcpy(poly, v, 3 * sizeof(cvert));
    for (int plane = 0; plane < 2; ++plane) {
        int m = 0;
        for (int k = 0; k < n; ++k) {
            const cvert *cur = &poly[k], *nxt = &poly[(k + 1) % n];
            float dc = plane_dist(cur, plane), dn = plane_dist(nxt, plane);
            int cin = dc >= 0.0f, nin = dn >= 0.0f;
            if (cin) tmp[m++] = *cur;
            if (cin != nin) {
                const cvert *in = cin ? cur : nxt, *ou = cin ? nxt : cur;
                float di = cin ? dc : dn, dou = cin ? dn : dc;
                float t = di / (di - dou);
                lerp_vert(in, ou, t, &tmp[m++]);
            }
        }
        n = m;
        memcpy(poly, tmp, (size_t)n * sizeof(cvert));
        if (n < 3) return;
    }
    for (int k = 1; k + 1 < n; ++k) { /* fan; every piece keeps the primitive's id */
        cvert tri[3] = { poly[0], poly[k], poly[k + 1] };
        raster_triangle(rt, fs, tri, prim);
    }
}

/* vs_main src/shaders/terrain.wgsl:44-66 for grid vertex (i,j) of build_grid_xyuv */
typedef struct {
    const float *u; uint32_t n; const float *tex; uint32_t tw, th;
    float spacing, exag, step;
} vsctx;
static inline void mat_vec(const float *m, float x, float y, float z, float w, float r[4])
{
    for (int k = 0; k < 4; ++k) {
        float acc = m[k] * x;
        acc = fmaf(m[4 + k], y, acc);
        acc = fmaf(m[8 + k], z, acc);
        acc = fmaf(m[12 + k], w, acc);
        r[k] = acc;
    }
}
static void vs_terrain(const vsctx *c, uint32_t i, uint32_t j, cvert *o)
{
    const float scale = 1.5f;
    float x = -scale + (float)i * c->step;                 /* src/terrain/mod.rs:566-567 */
    float z = -scale + (float)j * c->step;
    float uu = (float)i / ((float)c->n - 1.0f);            /* :568-569 */
    float vv = (float)j / ((float)c->n - 1.0f);
    /* textureSampleLevel(height, nearest clamp, uv, 0).r  terrain.wgsl:50 */
    int tx = (int)floorf(uu * (float)c->tw), ty = (int)floorf(vv * (float)c->th);
    tx = iclamp(tx, 0, (int)c->tw - 1); ty = iclamp(ty, 0, (int)c->th - 1);
    float h_tex = c->tex[(size_t)ty * c->tw + tx];
    float h_ana = o_sinf(x * 1.3f) * 0.25f + o_cosf(z * 1.1f) * 0.25f;   /* :39-41,53 */
    float h = h_tex + h_ana;                                               /* :55 */
    float wx = x * c->spacing, wy = h * c->exag, wz = z * c->spacing;      /* :58 */
    float vp[4], cp[4];
    mat_vec(c->u, wx, wy, wz, 1.0f, vp);                                   /* view * world :61 */
    mat_vec(c->u + 16, vp[0], vp[1], vp[2], vp[3], cp);                    /* proj * (...) */
    o->x = cp[0]; o->y = cp[1]; o->z = cp[2]; o->w = cp[3];
    o->a[0] = h; o->a[1] = x; o->a[2] = z;                                 /* height, xz :63-64 (uv unused by fs) */
}

static void setup_target(rtarget *rt, uint32_t W, uint32_t H, uint8_t *rgba, uint32_t *vis,
                         uint32_t rank, uint32_t nranks, uint32_t band_h)
{
    memset(rt, 0, sizeof *rt);
    rt->W = W; rt->H = H; rt->rgba = rgba; rt->vis = vis;
    rt->rank = rank; rt->nranks = nranks ? nranks : 1; rt->band_h = band_h ? band_h : 1;
    rt->sc_x0 = 0; rt->sc_y0 = 0; rt->sc_x1 = (int32_t)W - 1; rt->sc_y1 = (int32_t)H - 1;
}
static void clear_target(const rtarget *rt, const float clear_linear[3])
{
    uint8_t c[4] = { srgb_encode(clear_linear[0]), srgb_encode(clear_linear[1]), srgb_encode(clear_linear[2]), 255 };
    size_t npx = (size_t)rt->W * rt->H;
    for (size_t o = 0; o < npx; ++o) {
        memcpy(rt->rgba + 4 * o, c, 4);
        rt->vis[o] = 0;
    }
}

/* primitive id -> its three grid vertices in index order [a,c,b, b,c,d] (src/terrain/mod.rs:578-582) */
static void terrain_prim(const vsctx *vc, uint32_t prim, cvert t[3])
{
    uint32_t nm1 = vc->n - 1, cell = prim >> 1, j = cell / nm1, i = cell - j * nm1;
    if ((prim & 1u) == 0) { vs_terrain(vc, i, j, &t[0]); vs_terrain(vc, i, j + 1, &t[1]); vs_terrain(vc, i + 1, j, &t[2]); }
    else { vs_terrain(vc, i + 1, j, &t[0]); vs_terrain(vc, i, j + 1, &t[1]); vs_terrain(vc, i + 1, j + 1, &t[2]); }
}

/*
 * Full terrain frame = render_png's render pass (src/terrain/mod.rs:412-437 / src/scene/mod.rs:280-298)
 * up to the RGBA8 texture contents.  `height`: the bound R32F texture (the class's dummy when the
 * caller never uploaded one).  Rows not owned by (rank,nranks,band_h) keep the clear colour.
 * nthreads <= 1: primitives are drawn strictly in index order on one thread (literal painter's loop);
 * nthreads > 1: grid rows are spread over OpenMP threads and merged by max primitive id.
 * vis (W*H u32, required): surviving primitive id + 1 per pixel, 0 = background.
 * Returns 0, or -1 on allocation failure.
 */
VFO_API int vfo_render_terrain_mode(const float u[44], uint32_t W, uint32_t H, uint32_t grid,
                                    const float *height, uint32_t tw, uint32_t th,
                                    const uint8_t lut_rgba8[1024], int lut_is_srgb,
                                    uint32_t rank, uint32_t nranks, uint32_t band_h,
                                    uint8_t *rgba, uint32_t *vis, int nthreads, int shade_mode);
VFO_API int vfo_render_terrain(const float u[44], uint32_t W, uint32_t H, uint32_t grid,
                               const float *height, uint32_t tw, uint32_t th,
                               const uint8_t lut_rgba8[1024], int lut_is_srgb,
                               uint32_t rank, uint32_t nranks, uint32_t band_h,
                               uint8_t *rgba, uint32_t *vis, int nthreads)
{
    return vfo_render_terrain_mode(u, W, H, grid, height, tw, th, lut_rgba8, lut_is_srgb, rank, nranks, band_h, rgba, vis, nthreads, 0);
}
VFO_API int vfo_render_terrain_mode(const float u[44], uint32_t W, uint32_t H, uint32_t grid,
                                    const float *height, uint32_t tw, uint32_t th,
                                    const uint8_t lut_rgba8[1024], int lut_is_srgb,
                                    uint32_t rank, uint32_t nranks, uint32_t band_h,
                                    uint8_t *rgba, uint32_t *vis, int nthreads, int shade_mode)
{
    init_tables();
    uint32_t n = grid < 2 ? 2 : grid;
    if (nthreads < 1) nthreads = 1;
    rtarget rt;
    setup_target(&rt, W, H, rgba, vis, rank, nranks, band_h);
    rt.mode = 0;
    rt.shade_mode = shade_mode; rt.tex = height; rt.tw = tw; rt.th = th;
    rt.spacing = fmaxf(u[36], 1e-8f); rt.exag = u[38];
    rt.h_range = fmaxf(u[37], 1e-8f);                                    /* terrain.wgsl:71 */
    rt.exposure = u[35];
    {   /* L = normalize(sun) terrain.wgsl:83 */
        float sx = u[32], sy = u[33], sz = u[34];
        float inv = 1.0f / sqrtf(fmaf(sz, sz, fmaf(sy, sy, sx * sx)));
        rt.Lx = sx * inv; rt.Ly = sy * inv; rt.Lz = sz * inv;
    }
    for (int i = 0; i < 256; ++i)
        for (int ch = 0; ch < 3; ++ch)
            rt.lut[i][ch] = lut_is_srgb ? g_srgb_decode[lut_rgba8[4 * i + ch]] : (float)lut_rgba8[4 * i + ch] / 255.0f;
    const float clear[3] = { 0.02f, 0.02f, 0.03f };                       /* src/terrain/mod.rs:421 */
    clear_target(&rt, clear);

    vsctx vc;
    vc.u = u; vc.n = n; vc.tex = height; vc.tw = tw; vc.th = th;
    vc.spacing = fmaxf(u[36], 1e-8f);                                     /* terrain.wgsl:46 */
    vc.exag = u[38];
    vc.step = (2.0f * 1.5f) / ((float)n - 1.0f);                          /* src/terrain/mod.rs:559-560 */

    /* ---- pass 0: which primitive's fragment survives at each pixel ---- */
    rt.pass = 0;
    rt.atomic_vis = nthreads > 1;
    int fail = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        cvert *row0 = (cvert *)malloc((size_t)n * sizeof(cvert));
        cvert *row1 = (cvert *)malloc((size_t)n * sizeof(cvert));
        if (!row0 || !row1) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            fail = 1;
        } else {
            long last_j = -2;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 8)
#endif
            for (long j = 0; j < (long)n - 1; ++j) {
                if (last_j == j - 1) { cvert *t = row0; row0 = row1; row1 = t; }
                else for (uint32_t i = 0; i < n; ++i) vs_terrain(&vc, i, (uint32_t)j, &row0[i]);
                for (uint32_t i = 0; i < n; ++i) vs_terrain(&vc, i, (uint32_t)j + 1, &row1[i]);
                last_j = j;
                for (uint32_t i = 0; i + 1 < n; ++i) {
                    /* indices [a,c,b, b,c,d] src/terrain/mod.rs:578-582 */
                    uint32_t cell = (uint32_t)j * (n - 1) + i;
                    cvert t0[3] = { row0[i], row1[i], row0[i + 1] };
                    cvert t1[3] = { row0[i + 1], row1[i], row1[i + 1] };
                    draw_primitive(&rt, frag_terrain, t0, 2 * cell);
                    draw_primitive(&rt, frag_terrain, t1, 2 * cell + 1);
                }
            }
        }
        free(row0); free(row1);
    }
    if (fail) return -1;

    /* ---- pass 1: fs_main for the surviving fragment of every covered pixel ---- */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
#endif
    for (long py = 0; py < (long)H; ++py) {
        rtarget r1 = rt;
        r1.pass = 1; r1.atomic_vis = 0;
        for (uint32_t px = 0; px < W; ++px) {
            uint32_t id = vis[(size_t)py * W + px];
            if (!id) continue;
            cvert t[3];
            terrain_prim(&vc, id - 1, t);
            r1.sc_x0 = r1.sc_x1 = (int32_t)px; r1.sc_y0 = r1.sc_y1 = (int32_t)py;
            draw_primitive(&r1, frag_terrain, t, id - 1);
        }
    }
    return 0;
}

/* Triangle smoke path: src/lib.rs:72-91 (geometry), :685-721 (pass; clear WHITE :19),
 * src/shaders/triangle.wgsl (pos = (x,y,0,1), colour varying). */
VFO_API int vfo_render_triangle(uint32_t W, uint32_t H, uint8_t *rgba)
{
    init_tables();
    uint32_t *vis = (uint32_t *)malloc((size_t)W * H * sizeof(uint32_t));
    if (!vis) return -1;
    rtarget rt;
    setup_target(&rt, W, H, rgba, vis, 0, 1, 1);
    rt.mode = 1;
    const float clear[3] = { 1.0f, 1.0f, 1.0f };
    clear_target(&rt, clear);
    cvert v[3] = {
        { -0.8f, -0.8f, 0.0f, 1.0f, { 1.0f, 0.2f, 0.2f } },
        { 0.8f, -0.8f, 0.0f, 1.0f, { 0.2f, 1.0f, 0.2f } },
        { 0.0f, 0.8f, 0.0f, 1.0f, { 0.2f, 0.2f, 1.0f } },
    };
    rt.pass = 0; draw_primitive(&rt, frag_triangle, v, 0);
    rt.pass = 1; draw_primitive(&rt, frag_triangle, v, 0);   /* one primitive: every covered pixel is its own */
    free(vis);
    return 0;
}

/* Convention test hook: rasterise arbitrary clip-space triangles (xyzw per vertex) in index order and
 * return the surviving primitive id + 1 per pixel -- exercises the fill rule, culling and clipping. */
VFO_API int vfo_raster_triangles(const float *clip_xyzw, uint32_t ntris, uint32_t W, uint32_t H, uint32_t *vis)
{
    init_tables();
    uint8_t *rgba = (uint8_t *)malloc((size_t)W * H * 4);
    if (!rgba) return -1;
    rtarget rt;
    setup_target(&rt, W, H, rgba, vis, 0, 1, 1);
    const float clear[3] = { 0.0f, 0.0f, 0.0f };
    clear_target(&rt, clear);
    rt.pass = 0;
    for (uint32_t t = 0; t < ntris; ++t) {
        cvert v[3];
        for (int k = 0; k < 3; ++k) {
            const float *p = clip_xyzw + (size_t)(3 * t + k) * 4;
            v[k].x = p[0]; v[k].y = p[1]; v[k].z = p[2]; v[k].w = p[3];
            v[k].a[0] = v[k].a[1] = v[k].a[2] = 0.0f;
        }
        draw_primitive(&rt, frag_triangle, v, t);
    }
    free(rgba);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Renderer DEM path, sequential f32 loops exactly as the reference writes them.
 * ---------------------------------------------------------------------------------------- */
/* add_terrain ingest, src/lib.rs:351-388 */
VFO_API void vfo_dem_ingest_f32(const float *src, float *dst, size_t n, float exaggeration)
{
    for (size_t k = 0; k < n; ++k) dst[k] = src[k] * exaggeration;
}
VFO_API void vfo_dem_ingest_f64(const double *src, float *dst, size_t n, float exaggeration)
{
    for (size_t k = 0; k < n; ++k) dst[k] = (float)src[k] * exaggeration;
}
/* dem_stats_from_slice, src/lib.rs:905-932: out = {min, max, mean, std} */
VFO_API void vfo_dem_stats(const float *h, size_t n, float out[4])
{
    if (n == 0) { out[0] = out[1] = out[2] = out[3] = 0.0f; return; }
    float mn = h[0], mx = h[0], sum = 0.0f;
    for (size_t k = 0; k < n; ++k) { if (h[k] < mn) mn = h[k]; if (h[k] > mx) mx = h[k]; sum += h[k]; }
    float mean = sum / (float)n;
    float vs = 0.0f;
    for (size_t k = 0; k < n; ++k) { float d = h[k] - mean; vs += d * d; }
    out[0] = mn; out[1] = mx; out[2] = mean; out[3] = sqrtf(vs / (float)n);
}
/* normalize_in_place, src/lib.rs:934-951; stats = {min,max,mean,std} */
VFO_API void vfo_dem_normalize(float *h, size_t n, int zscore, float eps, float lo, float hi, const float stats[4])
{
    if (!zscore) {
        float denom = fmaxf(fabsf(stats[1] - stats[0]), eps);
        float scale = (hi - lo) / denom;
        for (size_t k = 0; k < n; ++k) h[k] = (h[k] - stats[0]) * scale + lo;
    } else {
        float denom = fmaxf(stats[3], eps);
        for (size_t k = 0; k < n; ++k) h[k] = (h[k] - stats[2]) / denom;
    }
}
/* terrain_stats::min_max(data, clamp = true), src/terrain_stats.rs:11-35 */
static int cmp_f32(const void *a, const void *b) { float x = *(const float *)a, y = *(const float *)b; return (x > y) - (x < y); }
VFO_API int vfo_dem_percentile_range(const float *h, size_t n, float *p1, float *p99)
{
    const size_t SAMPLE = 65536;
    size_t step = n > SAMPLE ? n / SAMPLE : 1;
    size_t m = (n + step - 1) / step;
    float *buf = (float *)malloc(m * sizeof(float));
    if (!buf) return -1;
    for (size_t k = 0; k < m; ++k) buf[k] = h[k * step];
    qsort(buf, m, sizeof(float), cmp_f32);
    *p1 = buf[(size_t)((float)m * 0.01f)];
    *p99 = buf[(size_t)((float)m * 0.99f)];
    free(buf);
    return 0;
}

VFO_API int vfo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
