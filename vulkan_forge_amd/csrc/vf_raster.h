// vf_raster.h -- exact span solving for one triangle on the lines of a screen tile: FP32 first, exact arithmetic only where FP32
// cannot decide.
//
// The rasteriser walks the SHORT axis of a triangle's bounding box ("outer", line o) and needs, per line, the set of pixel centres
// along the long axis ("inner", offset r) that the triangle covers.  With the edge functions of DESIGN.md section 4 (inside-positive,
// top-left rule folded in as +1 on top/left edges) the covered offsets of line o are the integers r with
//        alpha_i(o) + beta_i * r > 0      for the three edges i,     alpha_i, beta_i integers,
// i.e. r > r*_i for edges with beta_i > 0 ("lower" edges) and r < r*_i for beta_i < 0 ("upper" edges), r*_i = -alpha_i / beta_i:
//        lo = max over lower edges of floor(r*_i) + 1,      hi = min over upper edges of ceil(r*_i) - 1.
// r*_i(o) is affine in o.  It is evaluated in FP32 as  t_i = fma(o, s_i, k_i)  with a per-edge error bound eps_i derived below
// (|t_i - r*_i| < eps_i, eps_i < 1/4); upper edges are kept negated (t'_i = -r*_i) so that both kinds go through the same floor.
//   stage 1:  F_i = floor(t_i - eps_i)  gives a span that CONTAINS the true one -- most lines of the thin slivers a noise terrain
//             is made of are rejected here because that span holds no open pixel;
//   stage 2:  if floor(t_i + eps_i) == F_i for all three edges, no integer lies inside any error interval and the stage-1 span IS
//             the exact one; otherwise (a crossing within eps of a pixel centre: about one line in 10^4) the line is solved
//             exactly in FP64 from the integer vertex coordinates (span_exact: every product and sum below 2^53).
// Triangles the FP32 form cannot describe (an edge parallel to the lines, error bound too large) take span_exact on every line.
//
// A vector FP64 FMA costs two FP32 ones on gfx950, an int64 multiply-add eight, a divergent branch a handful of scalar
// instructions at twice a vector one (tools/micro/valu_rates.hip): the common path below is straight-line FP32 / int32 code.
//
// Host-compilable (tests/cpp/raster_fuzz.cpp checks it against a brute-force int64 rasteriser): no HIP-only construct outside
// the VF_RASTER_DEVICE blocks.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIP_DEVICE_COMPILE__)
#define VF_RASTER_DEVICE 1
#else
#define VF_RASTER_DEVICE 0
#endif
#if defined(__HIPCC__)
#define VF_HD __host__ __device__ __forceinline__         // (hipcc's host pass sees the device callers too)
#else
#define VF_HD static inline
#endif

namespace vf {

#if VF_RASTER_DEVICE
VF_HD float rs_rcp(float x) { return __builtin_amdgcn_rcpf(x); }                         // 1 ulp
VF_HD float rs_med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
VF_HD int32_t rs_floor_i(float x)                                                        // V_CVT_FLR_I32_F32: one instruction
{
    int32_t k;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(k) : "v"(x));
    return k;
}
VF_HD float rs_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
VF_HD double rs_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
#else
#ifndef VF_RASTER_RCP_ULPS
#define VF_RASTER_RCP_ULPS 0        // the fuzz harness perturbs the reciprocal by -1 / +1 ulp to cover the hardware's 1-ulp estimate
#endif
static inline float rs_rcp(float x)
{
    float r = 1.0f / x;
    if (VF_RASTER_RCP_ULPS > 0) r = nextafterf(r, INFINITY);
    if (VF_RASTER_RCP_ULPS < 0) r = nextafterf(r, -INFINITY);
    return r;
}
static inline float rs_med3(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline int32_t rs_floor_i(float x) { return (int32_t)floorf(x); }
static inline float rs_fma(float a, float b, float c) { return fmaf(a, b, c); }
static inline double rs_fma(double a, double b, double c) { return fma(a, b, c); }
#endif

constexpr int32_t kSpanBig = 1 << 20;

// One triangle against the lines of a tile, prepared once.  Per edge i (upper edges negated, so both kinds take a floor):
//     F_i = floor(fma(o, s_i, km_i))                    (km carries -eps_i: the conservative side)
//     G_i = F_i ^ x_i,   c_i = G_i + c1_i               x = 0, c1 = 1 for a lower edge:  c = F + 1          (lower bound on r)
//                                                       x = -1, c1 = -kSpanBig for an upper one:  c = ~F - big,  ~F = -F - 1 = ceil(r*) - 1
//     lo = max(0, max_i c_i),   hi = min(n_inner, min_i c_i + kSpanBig)
// (a lower edge leaves the minimum alone and an upper edge the maximum: |F| stays inside the window, far below kSpanBig).
struct SpanSetup {
    float s[3], km[3], eps2[3];   // eps2 = 2 eps_i: stage 2 looks at floor(t + eps) = floor((t - eps) + eps2)
    int32_t x[3], c1[3];
    int32_t zD0;                  // an edge parallel to the lines (beta = 0) has no crossing: line o lies inside it iff
    int32_t zflags;               //   D = zD0 + 256 o  is 0 and the edge is top/left, or has the sign of its u coefficient (bit 0: edge
                                  //   present, bit 1: coefficient > 0, bit 2: top/left); its slot above is a lower edge far below the window
    bool regular;                 // false: every line is solved by span_exact
};

// Edge i runs from vertex i+1 to vertex i+2 (mod 3), as everywhere in this code base: e_0 = (v1, v2), e_1 = (v2, v0), e_2 = (v0, v1).
// (U, V) are the snapped vertex coordinates along the outer / inner axis; (u0c, v0c) the centre of the pixel at line 0, offset 0
// (24.8 fixed point); swapped = the outer axis is y (mirrors the orientation: the coefficients change sign).
struct EdgeInts { int32_t Cu, Cv, UR, VR, tl; };
VF_HD EdgeInts edge_ints(int i, const int32_t U[3], const int32_t V[3], bool swapped)
{
    const int a = i == 0 ? 1 : (i == 1 ? 2 : 0), b = i == 0 ? 2 : (i == 1 ? 0 : 1);
    const int32_t dU = U[b] - U[a], dV = V[b] - V[a];
    EdgeInts e;
    e.Cu = swapped ? -dV : dV;            // A = dY, B = -dX in screen axes; outer = x: (Cu, Cv) = (A, B), outer = y: (B, A)
    e.Cv = swapped ? dU : -dU;
    e.UR = U[a]; e.VR = V[a];
    const int32_t A = swapped ? e.Cv : e.Cu, B = swapped ? e.Cu : e.Cv;
    e.tl = (A > 0 || (A == 0 && B > 0)) ? 1 : 0;      // top-left rule on the inside-positive edge function
    return e;
}

VF_HD float rs_bits(uint32_t u) { union { uint32_t u; float f; } c; c.u = u; return c.f; }
VF_HD uint32_t rs_ubits(float f) { union { uint32_t u; float f; } c; c.f = f; return c.u; }

VF_HD void span_setup(const int32_t U[3], const int32_t V[3], bool swapped, int32_t u0c, int32_t v0c, int32_t n_outer, SpanSetup &S)
{
    uint32_t good = 0, flat = 0, kinds = 0;                // per-edge bits: error bound small enough | parallel to the lines; kinds seen
    const float span = (float)(n_outer + 1);
    const int32_t sw = swapped ? -1 : 0;                   // all ones when the orientation is mirrored
#if VF_RASTER_DEVICE
#pragma unroll
#endif
    for (int i = 0; i < 3; ++i) {
        const int a = i == 0 ? 1 : (i == 1 ? 2 : 0), b = i == 0 ? 2 : (i == 1 ? 0 : 1);
        const int32_t dU = U[b] - U[a], dV = V[b] - V[a];
        // (Cu, Cv) = +-(dV, -dU), the sign is the mirroring's: the slope s = -Cu / Cv does not see it, the kind of the edge does.
        // r*(o) = (VR - v0c)/256 - tl/(256 Cv)  +  s * ((u0c - UR)/256 + o)            (alpha + beta r* = 0)
        const float rc = rs_rcp((float)dU);                                // |dU| < 2^24: the conversion is exact (dU = 0: fixed up below)
        const float s = (float)dV * rc;                                    // = -Cu / Cv
        const float gg = (float)(u0c - U[a]) * (1.0f / 256.0f);
        const float cc = (float)(V[a] - v0c) * (1.0f / 256.0f);
        const float k = rs_fma(gg, s, cc);
        // Error bound of t = fma(o, s, k) against r*: s carries 1.6 * 2^-23 relative (1-ulp reciprocal, one product) on |s| (|gg| + o);
        // the conversions of gg and cc, the fma that forms k, k - eps, the line's own fma and + 2 eps round at 2^-24 of values
        // bounded by |cc|, |k| <= |cc| + |gg||s| and the window (|t| < 2^7; outside the window nothing depends on the fraction):
        //     |t - r*| < ((|gg| + span) |s| * 2.1 + (2 |cc| + |gg||s|) + 2^7) * 2^-24  <  ((2 |gg| + span) |s| + |cc|) 2^-21 + 2^-16.
        // The top-left rule's +1 (on top / left edges) moves the crossing by 1 / (256 |Cv|): FP32 cannot carry that next to cc, so
        // the shift is part of every edge's bound instead (a crossing that close to a pixel centre takes the exact route).
        const float eps = rs_fma(rs_fma(rs_fma(fabsf(gg), 2.0f, span), fabsf(s), fabsf(cc)), 0x1.0p-21f, rs_fma(fabsf(rc), 0x1.0p-8f, 0x1.0p-16f));
        good |= (eps < 0.25f ? 1u : 0u) << i;                              // (NaN / inf fail the comparison)
        flat |= (dU == 0 ? 1u : 0u) << i;
        // kind: Cv = -dU (mirrored: +dU) > 0 is a lower edge; upper edges are negated -- a sign-bit flip -- and get x = -1
        const int32_t up = ((dU ^ sw) >> 31) ^ -1;                         // all ones for an upper edge (dU != 0)
        const uint32_t sm = (uint32_t)up & 0x80000000u;
        kinds |= (uint32_t)(up & 1) + 1u;                                  // bit 0: a lower edge seen, bit 1: an upper one
        S.s[i] = rs_bits(rs_ubits(s) ^ sm);
        S.km[i] = rs_bits(rs_ubits(k) ^ sm) - eps;
        S.eps2[i] = eps + eps;
        S.x[i] = up; S.c1[i] = (up & (-kSpanBig - 1)) + 1;
    }
    S.zD0 = 0; S.zflags = 0;
    if (flat) {
        // (rare) an edge parallel to the lines has no crossing: its slot becomes a lower edge far below the window, and a per-line sign
        // test of D = (u0c - UR) + 256 o stands in for it (span_line).  At most one edge of a triangle with area can be flat.
        kinds = 0;
#if VF_RASTER_DEVICE
#pragma unroll
#endif
        for (int i = 0; i < 3; ++i) {
            const int a = i == 0 ? 1 : (i == 1 ? 2 : 0), b = i == 0 ? 2 : (i == 1 ? 0 : 1);
            if ((flat >> i) & 1u) {
                const int32_t dV = V[b] - V[a];
                const int32_t Cu = swapped ? -dV : dV, dY = swapped ? 0 : dV, dX = swapped ? dV : 0;       // (dU = 0)
                const bool tl = dY > 0 || (dY == 0 && dX < 0);                                              // top-left in screen axes
                S.s[i] = 0.0f; S.km[i] = -0x1.0p20f; S.eps2[i] = 0.0f; S.x[i] = 0; S.c1[i] = 1;
                S.zD0 = u0c - U[a]; S.zflags = 1 | (Cu > 0 ? 2 : 0) | (tl ? 4 : 0);
            } else {
                kinds |= S.x[i] ? 2u : 1u;
            }
        }
    }
    S.regular = ((good | flat) == 7u) && kinds == 3u;
}

// Stage 1: a span [lo, hi] that contains the true one (lo > hi: no pixel on this line).  F[] feeds span_confirm.
VF_HD void span_line(const SpanSetup &S, int32_t o, int32_t n_inner, int32_t F[3], int32_t &lo, int32_t &hi)
{
    // one window for both kinds of edge (upper edges are negated): beyond +-(n_inner + 4) only "no constraint" / "empty" matter
    const float of = (float)o, w_hi = (float)(n_inner + 4), w_lo = -w_hi;
    int32_t c[3];
#if VF_RASTER_DEVICE
#pragma unroll
#endif
    for (int i = 0; i < 3; ++i) {
        F[i] = rs_floor_i(rs_med3(rs_fma(of, S.s[i], S.km[i]), w_lo, w_hi));
        c[i] = (F[i] ^ S.x[i]) + S.c1[i];
    }
    int32_t cmax = c[0] > c[1] ? c[0] : c[1], cmin = c[0] < c[1] ? c[0] : c[1];
    cmax = cmax > c[2] ? cmax : c[2]; cmin = cmin < c[2] ? cmin : c[2];
    int32_t l = cmax > 0 ? cmax : 0, h = cmin + kSpanBig;
    h = h < n_inner ? h : n_inner;
    if (S.zflags) {                                        // (rare) the edge parallel to the lines: exact integer sign test
        const int32_t D = S.zD0 + 256 * o;
        const bool in = D == 0 ? (S.zflags & 4) != 0 : ((D > 0) == ((S.zflags & 2) != 0));
        if (!in) h = -1;
    }
    lo = l; hi = h;
}

// Stage 0 (round 4): one bound for a GROUP of adjacent lines oa .. ob: a span [lo, hi] that contains the stage-1 span of every line of
// the group -- when it holds no open pixel of any of them, none of those lines can paint and the group is skipped untested.
// Why it contains them: the stage-1 bounds of line o are built from F_i(o) = floor(med3(fma(o, s_i, km_i))), and both lo (through the
// lower edges' c_i = F_i + 1) and hi (through the upper edges' -F_i - 1) move the CONSERVATIVE way when F_i is replaced by something
// smaller.  F_i is monotone in o (the exact o s_i + km_i is, and rounding, med3 and floor keep order), so its minimum over the group
// is F_i at oa when the stored slope s_i is >= 0 and at ob otherwise: the same instruction sequence as span_line at that end point,
// bit for bit.  (The per-line test of an edge parallel to the lines is left to span_line: ignoring it here only widens the bound.)
VF_HD void span_group(const SpanSetup &S, int32_t oa, int32_t ob, int32_t n_inner, int32_t &lo, int32_t &hi)
{
    const float fa = (float)oa, fb = (float)ob, w_hi = (float)(n_inner + 4), w_lo = -w_hi;
    int32_t c[3];
#if VF_RASTER_DEVICE
#pragma unroll
#endif
    for (int i = 0; i < 3; ++i) {
        const float of = S.s[i] < 0.0f ? fb : fa;
        const int32_t F = rs_floor_i(rs_med3(rs_fma(of, S.s[i], S.km[i]), w_lo, w_hi));
        c[i] = (F ^ S.x[i]) + S.c1[i];
    }
    int32_t cmax = c[0] > c[1] ? c[0] : c[1], cmin = c[0] < c[1] ? c[0] : c[1];
    cmax = cmax > c[2] ? cmax : c[2]; cmin = cmin < c[2] ? cmin : c[2];
    const int32_t l = cmax > 0 ? cmax : 0, h = cmin + kSpanBig;
    lo = l; hi = h < n_inner ? h : n_inner;
}

// Stage 2: true when no pixel centre lies within the FP32 error of any crossing -- the stage-1 span is then the exact one.
VF_HD bool span_confirm(const SpanSetup &S, int32_t o, int32_t n_inner, const int32_t F[3])
{
    const float of = (float)o, w_hi = (float)(n_inner + 4), w_lo = -w_hi;
    int32_t diff = 0;                                      // straight-line: no short-circuit branches
#if VF_RASTER_DEVICE
#pragma unroll
#endif
    for (int i = 0; i < 3; ++i) {
        const float tp = rs_fma(of, S.s[i], S.km[i]) + S.eps2[i];
        diff |= rs_floor_i(rs_med3(tp, w_lo, w_hi)) ^ F[i];
    }
    return diff == 0;
}

// The exact span of line o from the integer vertex coordinates: FP64 on integers below 2^25, every product and sum below 2^53.
// Independent of the FP32 quantities above; handles every edge direction (also beta = 0).
VF_HD void span_exact(const int32_t U[3], const int32_t V[3], bool swapped, int32_t u0c, int32_t v0c, int32_t o, int32_t n_inner,
                      int32_t &lo, int32_t &hi)
{
    int32_t l = 0, h = n_inner;
#if VF_RASTER_DEVICE
#pragma unroll
#endif
    for (int i = 0; i < 3; ++i) {
        const EdgeInts e = edge_ints(i, U, V, swapped);
        const double alpha = rs_fma((double)e.Cu, (double)(u0c + 256 * o - e.UR), rs_fma((double)e.Cv, (double)(v0c - e.VR), (double)e.tl));
        const double beta = 256.0 * (double)e.Cv;
        if (e.Cv == 0) {
            if (!(alpha > 0.0)) h = -1;
            continue;
        }
        // g(r) = alpha + beta r.  Start from the rounded quotient inside a window around the line, then walk to the exact boundary:
        // the quotient is off by less than one, the window clamp by any amount -- the walks below give up after three steps and
        // report "beyond the window", which the [0, n_inner] clamp of the caller turns into the right answer.
        double q = -alpha / beta;
        q = q < -3.0 ? -3.0 : (q > (double)(n_inner + 3) ? (double)(n_inner + 3) : q);
        int32_t r = (int32_t)floor(q);
        if (e.Cv > 0) {                                    // smallest r with g(r) > 0
            int steps = 0;
            if (rs_fma(beta, (double)r, alpha) > 0.0) { while (steps < 3 && rs_fma(beta, (double)(r - 1), alpha) > 0.0) { --r; ++steps; } if (steps == 3) r = -kSpanBig; }
            else { do { ++r; ++steps; } while (steps < 3 && !(rs_fma(beta, (double)r, alpha) > 0.0)); if (!(rs_fma(beta, (double)r, alpha) > 0.0)) r = kSpanBig; }
            l = r > l ? r : l;
        } else {                                           // largest r with g(r) > 0
            int steps = 0;
            if (rs_fma(beta, (double)r, alpha) > 0.0) { while (steps < 3 && rs_fma(beta, (double)(r + 1), alpha) > 0.0) { ++r; ++steps; } if (steps == 3) r = kSpanBig; }
            else { do { --r; ++steps; } while (steps < 3 && !(rs_fma(beta, (double)r, alpha) > 0.0)); if (!(rs_fma(beta, (double)r, alpha) > 0.0)) r = -kSpanBig; }
            h = r < h ? r : h;
        }
    }
    lo = l; hi = h;
}

} // namespace vf
