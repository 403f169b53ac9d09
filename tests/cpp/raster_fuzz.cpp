// raster_fuzz.cpp -- host check of vulkan_forge_amd/csrc/vf_raster.h (the span solver of the tile kernel's fast raster path)
// against a brute-force int64 evaluation of the coverage rule of DESIGN.md section 4 (pixel centres, top-left rule on
// inside-positive edge functions).  Built and run by tests/test_raster_spans.py:
//     g++ -O2 -ffp-contract=off [-DVF_RASTER_RCP_ULPS=-1|0|1] tests/cpp/raster_fuzz.cpp -o raster_fuzz && ./raster_fuzz <cases> <seed>
// For every random triangle x tile window x line it requires
//   - stage 1 (span_line) to contain the true span,
//   - stage 2 (span_confirm), when it accepts, to make the stage-1 span equal to the true span,
//   - span_exact to equal the true span always,
//   - the group bound (span_group) of every run of up to four adjacent lines to contain the stage-1 span of each of its lines.
// It also reports how often stage 2 falls back (the FP32 path must decide nearly every line by itself).
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <random>
#include "../../vulkan_forge_amd/csrc/vf_raster.h"

using namespace vf;

static bool covered(const int32_t X[3], const int32_t Y[3], int32_t px, int32_t py)
{
    const int64_t Px = (int64_t)px * 256 + 128, Py = (int64_t)py * 256 + 128;
    for (int i = 0; i < 3; ++i) {
        const int a = i == 0 ? 1 : (i == 1 ? 2 : 0), b = i == 0 ? 2 : (i == 1 ? 0 : 1);
        const int64_t A = (int64_t)Y[b] - Y[a], B = -((int64_t)X[b] - X[a]);
        const int64_t e = A * (Px - X[a]) + B * (Py - Y[a]);
        const bool tl = A > 0 || (A == 0 && B > 0);
        if (!(e > 0 || (e == 0 && tl))) return false;
    }
    return true;
}

int main(int argc, char **argv)
{
    const long cases = argc > 1 ? atol(argv[1]) : 200000;
    std::mt19937_64 rng(argc > 2 ? (uint64_t)atoll(argv[2]) : 1);
    auto uni = [&](int64_t lo, int64_t hi) { return (int64_t)(lo + (int64_t)(rng() % (uint64_t)(hi - lo + 1))); };
    long lines = 0, nonempty = 0, fallback = 0, irregular_tris = 0, tris = 0, bad = 0;
    long k_tris[10] = {0}, k_irr[10] = {0}, k_lines[10] = {0}, k_fb[10] = {0};
    long group_lines = 0, group_slack = 0;
    for (long c = 0; c < cases && bad < 10; ++c) {
        int32_t X[3], Y[3];
        const int kind = (int)uni(0, 9);
        const int32_t cx = (int32_t)uni(-20000, 1100000), cy = (int32_t)uni(-20000, 1100000);    // 24.8: around a 4096^2 target
        if (kind <= 4) {                                   // sliver: long, thin, any direction (what a noise terrain is made of)
            const double ang = (double)uni(0, 6283185) * 1e-6, len = (double)uni(256, 80000), wid = (double)uni(1, 400);
            const double dx = cos(ang), dy = sin(ang);
            X[0] = cx; Y[0] = cy;
            X[1] = cx + (int32_t)(len * dx); Y[1] = cy + (int32_t)(len * dy);
            X[2] = cx + (int32_t)(0.5 * len * dx - wid * dy); Y[2] = cy + (int32_t)(0.5 * len * dy + wid * dx);
            if (kind == 0) { X[1] = X[0]; }                // an edge exactly parallel to y
            if (kind == 1) { Y[1] = Y[0]; }                // ... to x
        } else if (kind <= 6) {                            // general triangle up to ~300 px
            for (int k = 0; k < 3; ++k) { X[k] = cx + (int32_t)uni(-40000, 40000); Y[k] = cy + (int32_t)uni(-40000, 40000); }
        } else if (kind == 7) {                            // vertices and edges through pixel centres (the top-left rule decides)
            for (int k = 0; k < 3; ++k) { X[k] = ((cx >> 8) + (int32_t)uni(-6, 6)) * 256 + 128; Y[k] = ((cy >> 8) + (int32_t)uni(-6, 6)) * 256 + 128; }
        } else if (kind == 8) {                            // small, sub-pixel scale
            for (int k = 0; k < 3; ++k) { X[k] = cx + (int32_t)uni(-600, 600); Y[k] = cy + (int32_t)uni(-600, 600); }
        } else {                                           // huge: extents just below the fast path's limit (2^24)
            for (int k = 0; k < 3; ++k) { X[k] = cx + (int32_t)uni(-8000000, 8000000); Y[k] = cy + (int32_t)uni(-8000000, 8000000); }
        }
        int64_t area2 = (int64_t)(X[1] - X[0]) * (Y[2] - Y[0]) - (int64_t)(Y[1] - Y[0]) * (X[2] - X[0]);
        if (area2 == 0) continue;
        if (area2 > 0) { std::swap(X[1], X[2]); std::swap(Y[1], Y[2]); }     // front-facing = negative area in y-down pixels
        const int32_t xmin = std::min(X[0], std::min(X[1], X[2])), xmax = std::max(X[0], std::max(X[1], X[2]));
        const int32_t ymin = std::min(Y[0], std::min(Y[1], Y[2])), ymax = std::max(Y[0], std::max(Y[1], Y[2]));
        if ((uint32_t)xmax - (uint32_t)xmin >= (1u << 24) || (uint32_t)ymax - (uint32_t)ymin >= (1u << 24)) continue;
        // a tile (or strip) window that meets the bounding box
        const int32_t bx0 = (xmin + 127) >> 8, bx1 = (xmax - 128) >> 8, by0 = (ymin + 127) >> 8, by1 = (ymax - 128) >> 8;
        if (bx0 > bx1 || by0 > by1) continue;
        const int32_t tw = (int32_t)(1 << uni(2, 6)), th = 64;
        const int32_t tx_lo = (int32_t)uni(bx0 - tw + 1, bx1), ty_lo = (int32_t)uni(by0 - th + 1, by1);
        const int32_t tx_hi = tx_lo + tw - 1, ty_hi = ty_lo + (int32_t)uni(0, th - 1);
        const int32_t px0 = std::max(bx0, tx_lo), px1 = std::min(bx1, tx_hi), py0 = std::max(by0, ty_lo), py1 = std::min(by1, ty_hi);
        if (px0 > px1 || py0 > py1) continue;
        ++tris;
        const bool cols = (px1 - px0) <= (py1 - py0);
        const int32_t U[3] = { cols ? X[0] : Y[0], cols ? X[1] : Y[1], cols ? X[2] : Y[2] };
        const int32_t V[3] = { cols ? Y[0] : X[0], cols ? Y[1] : X[1], cols ? Y[2] : X[2] };
        const int32_t n_outer = cols ? px1 - px0 : py1 - py0, n_inner = cols ? py1 - py0 : px1 - px0;
        const int32_t u0c = (cols ? px0 : py0) * 256 + 128, v0c = (cols ? py0 : px0) * 256 + 128;
        SpanSetup S;
        span_setup(U, V, !cols, u0c, v0c, n_outer, S);
        irregular_tris += S.regular ? 0 : 1;
        k_tris[kind]++; k_irr[kind] += S.regular ? 0 : 1;
        if (S.regular)                                     // stage 0: the bound of lines oa .. oa + 3 holds every one of their stage-1 spans
            for (int32_t oa = 0; oa <= n_outer && bad < 10; oa += 4) {
                const int32_t ob = std::min(oa + 3, n_outer);
                int32_t glo, ghi;
                span_group(S, oa, ob, n_inner, glo, ghi);
                for (int32_t o = oa; o <= ob; ++o) {
                    int32_t F[3], lo, hi;
                    span_line(S, o, n_inner, F, lo, hi);
                    ++group_lines;
                    if (lo <= hi && (glo > lo || ghi < hi)) { printf("group bound cuts a line: case %ld lines %d..%d line %d: [%d,%d] vs group [%d,%d]\n", c, oa, ob, o, lo, hi, glo, ghi); ++bad; break; }
                    group_slack += (lo <= hi) ? (lo - glo) + (ghi - hi) : 0;
                }
            }
        for (int32_t o = 0; o <= n_outer; ++o) {
            ++lines; k_lines[kind]++;
            int32_t tlo = n_inner + 1, thi = -1;           // brute force: first / last covered offset (coverage along a line is an interval)
            int ncov = 0;
            for (int32_t r = 0; r <= n_inner; ++r) {
                const int32_t px = cols ? px0 + o : px0 + r, py = cols ? py0 + r : py0 + o;
                if (covered(X, Y, px, py)) { tlo = std::min(tlo, r); thi = std::max(thi, r); ++ncov; }
            }
            if (ncov && ncov != thi - tlo + 1) { printf("NOT AN INTERVAL?! case %ld\n", c); ++bad; break; }
            nonempty += ncov ? 1 : 0;
            int32_t F[3], lo, hi;
            span_line(S, o, n_inner, F, lo, hi);
            bool use_exact = !S.regular;
            if (S.regular) {
                if (ncov && (lo > tlo || hi < thi || lo > hi)) { printf("stage 1 cuts the span: case %ld line %d: [%d,%d] vs true [%d,%d]\n", c, o, lo, hi, tlo, thi); ++bad; break; }
                if (lo <= hi) {
                    if (span_confirm(S, o, n_inner, F)) {
                        const bool same = ncov ? (lo == tlo && hi == thi) : false;
                        if (!same) { printf("stage 2 accepts a wrong span: case %ld line %d: [%d,%d] vs true [%d,%d] (ncov %d)\n", c, o, lo, hi, tlo, thi, ncov); ++bad; break; }
                    } else { use_exact = true; ++fallback; k_fb[kind]++; }
                }
            }
            int32_t elo, ehi;
            span_exact(U, V, !cols, u0c, v0c, o, n_inner, elo, ehi);
            elo = std::max(elo, 0); ehi = std::min(ehi, n_inner);
            const bool eok = ncov ? (elo == tlo && ehi == thi) : (elo > ehi);
            if (!eok) { printf("span_exact wrong: case %ld line %d: [%d,%d] vs true [%d,%d] (ncov %d)\n", c, o, elo, ehi, tlo, thi, ncov); ++bad; break; }
            (void)use_exact;
        }
    }
    printf("triangles %ld (irregular %ld = %.3f %%)  lines %ld  non-empty %ld  stage-2 fallbacks %ld (%.4f %% of lines)  failures %ld\n",
           tris, irregular_tris, 100.0 * irregular_tris / (tris ? tris : 1), lines, nonempty, fallback, 100.0 * fallback / (lines ? lines : 1), bad);
    printf("group bounds: %ld lines checked, mean slack %.2f pixels per non-empty line\n", group_lines, (double)group_slack / (group_lines ? group_lines : 1));
    for (int k = 0; k < 10; ++k) printf("  kind %d: triangles %ld irregular %.3f %%  lines %ld fallback %.4f %%\n", k, k_tris[k], 100.0 * k_irr[k] / (k_tris[k] ? k_tris[k] : 1), k_lines[k], 100.0 * k_fb[k] / (k_lines[k] ? k_lines[k] : 1));
    return bad ? 1 : 0;
}
