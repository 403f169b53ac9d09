/*
 * walk_model.c -- ANALYSIS TOOL (VERDICT r05 "next round" item 1, step 0), driven by tests/walk_model/walk_step0.py.  Not product, not the oracle.
 *
 * A CPU model of an IMAGE-ORDER visibility resolver: per pixel, enumerate the grid cells whose primitives can cover the
 * pixel centre in DESCENDING primitive id (cell rows descending, cells of a row descending, odd primitive before even) and
 * stop at the first exact hit -- the reference draws without depth (src/terrain/pipeline.rs:133), ids are cell-row major
 * (src/terrain/mod.rs:578-582), so the largest covering id is what the frame shows.
 *
 * Which cells CAN cover pixel P?  A snapped triangle that covers P is, up to eps px per vertex (1/512 px of snap + the
 * roundings of the divide / viewport fma), the exact projection of a world triangle whose vertices lie within rho of the
 * grid's (rho: the FP32 rounding of the two mat-vecs, pulled back through the 3x3 part of proj*view).  So the thin pyramid
 * { |x_ndc - x_P| <= eps/hw, |y_ndc - y_P| <= eps/hh } -- four planes through the eye -- meets the cell's prism
 * [x_i, x_i+1] x [hmin, hmax] x [z_j, z_j+1] grown by rho.  Each plane is tested on its own (necessary, cheap, and valid for
 * any box: block, 64 x 64 super-block, the whole grid), which gives the hierarchy its skips.
 *
 * The model counts, per pixel: row steps and box tests per level, cells that reach the exact edge test, and checks the
 * winner against the oracle's visibility buffer.  Vertex arithmetic comes from the oracle's own functions (included below).
 */
#include "../../oracle/vf_oracle.c"

typedef struct { double a, b, c, d; } plane;   /* a X + b Y + c Z + d <= 0, world units */

typedef struct {
    uint32_t n, nm1, nb, ns;          /* vertices per side, cells, 8-blocks, 64-super-blocks */
    uint32_t W, H;
    float *h;                          /* n*n displaced heights */
    int32_t *X, *Y;                    /* n*n snapped */
    uint8_t *fl;                       /* n*n vertex flags: 1 near, 2 far, 4 bad, 8 nosnap */
    float *cmin, *cmax;                /* nm1*nm1 per cell (world y) */
    float *bmin, *bmax;                /* nb*nb */
    float *smin, *smax;                /* ns*ns */
    double r0[4], r1[4], r3[4];        /* rows of proj*view over world (X, Y, Z, 1) */
    double spacing, exag, step;
    double rho[3];                     /* prism growth, world units */
    double ex, ey;                     /* eps / hw, eps / hh */
    double gy0, gy1;                   /* global world-y range */
} model;

static model G;

enum { C_SROW, C_STEST, C_BROW, C_BTEST, C_CROW, C_CTEST, C_EXACT, C_TRI, C_NCOUNT };

static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline double dmax(double a, double b) { return a > b ? a : b; }

static inline double plane_min(const plane *p, double x0, double x1, double y0, double y1, double z0, double z1)
{
    return p->d + dmin(p->a * x0, p->a * x1) + dmin(p->b * y0, p->b * y1) + dmin(p->c * z0, p->c * z1);
}
static inline int box_hits(const plane pl[4], double x0, double x1, double y0, double y1, double z0, double z1)
{
    for (int k = 0; k < 4; ++k) if (plane_min(&pl[k], x0, x1, y0, y1, z0, z1) > 0.0) return 0;
    return 1;
}
/* X interval of the beam for Y in [y0,y1], Z in [z0,z1]; returns 0 when empty */
static inline int x_interval(const plane pl[4], double y0, double y1, double z0, double z1, double *xlo, double *xhi)
{
    double lo = -1e300, hi = 1e300;
    for (int k = 0; k < 4; ++k) {
        const plane *p = &pl[k];
        double R = -(p->d + dmin(p->b * y0, p->b * y1) + dmin(p->c * z0, p->c * z1));   /* a X <= R */
        if (p->a > 0.0) hi = dmin(hi, R / p->a);
        else if (p->a < 0.0) lo = dmax(lo, R / p->a);
        else if (R < 0.0) return 0;
    }
    *xlo = lo; *xhi = hi;
    return lo <= hi;
}
static inline int z_interval(const plane pl[4], double x0, double x1, double y0, double y1, double *zlo, double *zhi)
{
    double lo = -1e300, hi = 1e300;
    for (int k = 0; k < 4; ++k) {
        const plane *p = &pl[k];
        double R = -(p->d + dmin(p->a * x0, p->a * x1) + dmin(p->b * y0, p->b * y1));
        if (p->c > 0.0) hi = dmin(hi, R / p->c);
        else if (p->c < 0.0) lo = dmax(lo, R / p->c);
        else if (R < 0.0) return 0;
    }
    *zlo = lo; *zhi = hi;
    return lo <= hi;
}

static inline double wcoord(uint32_t i) { return (double)(-1.5f + (float)i * (float)G.step) * G.spacing; }

/* grid index range [ia, ib] (clamped to [0, cnt-1]) of groups of `g` cells that a world interval [lo, hi] can touch */
static inline int idx_range(double lo, double hi, uint32_t g, uint32_t cnt, int32_t *ia, int32_t *ib)
{
    double cell = G.step * G.spacing * (double)g;
    double a = floor((lo + 1.5 * G.spacing) / cell) - 1.0, b = floor((hi + 1.5 * G.spacing) / cell) + 1.0;
    if (b < 0.0 || a > (double)cnt - 1.0) return 0;
    *ia = a < 0.0 ? 0 : (int32_t)a;
    *ib = b > (double)cnt - 1.0 ? (int32_t)cnt - 1 : (int32_t)b;
    return 1;
}

/* exact coverage of pixel centre by primitive (cell, odd): 1 hit, 0 miss, -1 generic (needs the clipped path) */
static int prim_covers(uint32_t i, uint32_t j, int odd, int32_t px, int32_t py)
{
    const uint32_t n = G.n;
    size_t va = (size_t)j * n + i, vb = va + 1, vc = va + n, vd = vc + 1;
    size_t k0 = odd ? vb : va, k1 = vc, k2 = odd ? vd : vb;
    uint8_t f0 = G.fl[k0], f1 = G.fl[k1], f2 = G.fl[k2];
    uint8_t any = f0 | f1 | f2, all = f0 & f1 & f2;
    if (any & 4) return 0;
    if (all & 3) return 0;
    if (any & 3) return -1;
    if (any & 8) return 0;
    int64_t X0 = G.X[k0], Y0 = G.Y[k0], X1 = G.X[k1], Y1 = G.Y[k1], X2 = G.X[k2], Y2 = G.Y[k2];
    int64_t xmin = X0 < X1 ? X0 : X1; if (X2 < xmin) xmin = X2;
    int64_t xmax = X0 > X1 ? X0 : X1; if (X2 > xmax) xmax = X2;
    int64_t ymin = Y0 < Y1 ? Y0 : Y1; if (Y2 < ymin) ymin = Y2;
    int64_t ymax = Y0 > Y1 ? Y0 : Y1; if (Y2 > ymax) ymax = Y2;
    if (xmax - xmin >= (1 << 24) || ymax - ymin >= (1 << 24)) return -1;
    int64_t area2 = (X1 - X0) * (Y2 - Y0) - (Y1 - Y0) * (X2 - X0);
    if (area2 >= 0) return 0;
    int64_t Px = (int64_t)px * 256 + 128, Py = (int64_t)py * 256 + 128;
    int64_t e0 = -((X2 - X1) * (Py - Y1) - (Y2 - Y1) * (Px - X1));
    int64_t e1 = -((X0 - X2) * (Py - Y2) - (Y0 - Y2) * (Px - X2));
    int64_t e2 = -((X1 - X0) * (Py - Y0) - (Y1 - Y0) * (Px - X0));
    int64_t a0 = Y2 - Y1, b0 = -(X2 - X1), a1 = Y0 - Y2, b1 = -(X0 - X2), a2 = Y1 - Y0, b2 = -(X1 - X0);
    int tl0 = a0 > 0 || (a0 == 0 && b0 > 0), tl1 = a1 > 0 || (a1 == 0 && b1 > 0), tl2 = a2 > 0 || (a2 == 0 && b2 > 0);
    if (!(e0 > 0 || (e0 == 0 && tl0))) return 0;
    if (!(e1 > 0 || (e1 == 0 && tl1))) return 0;
    if (!(e2 > 0 || (e2 == 0 && tl2))) return 0;
    return 1;
}

/* returns prim + 1, 0 = background, 0xFFFFFFFF = met a generic primitive first */
static uint32_t walk_pixel(int32_t px, int32_t py, uint32_t cnt[C_NCOUNT])
{
    const double hw = 0.5 * G.W, hh = 0.5 * G.H;
    const double xn = ((double)px + 0.5 - hw) / hw, yn = -((double)py + 0.5 - hh) / hh;
    plane pl[4];
    for (int k = 0; k < 4; ++k) {
        const double *r = k < 2 ? G.r0 : G.r1;
        const double t = k < 2 ? xn : yn, e = k < 2 ? G.ex : G.ey;
        const double s = (k & 1) ? -1.0 : 1.0;                 /* s (r - (t + s e) r3) . p <= 0 */
        const double q = t + s * e;
        pl[k].a = s * (r[0] - q * G.r3[0]); pl[k].b = s * (r[1] - q * G.r3[1]);
        pl[k].c = s * (r[2] - q * G.r3[2]); pl[k].d = s * (r[3] - q * G.r3[3]);
    }
    const double gx0 = -1.5 * G.spacing - G.rho[0], gx1 = 1.5 * G.spacing + G.rho[0];
    const double gz0 = -1.5 * G.spacing - G.rho[2], gz1 = 1.5 * G.spacing + G.rho[2];
    const double gy0 = G.gy0 - G.rho[1], gy1 = G.gy1 + G.rho[1];
    if (!box_hits(pl, gx0, gx1, gy0, gy1, gz0, gz1)) return 0;
    double zlo, zhi;
    if (!z_interval(pl, gx0, gx1, gy0, gy1, &zlo, &zhi)) return 0;
    int32_t sja, sjb;
    if (!idx_range(zlo - G.rho[2], zhi + G.rho[2], 64, G.ns, &sja, &sjb)) return 0;
    for (int32_t sj = sjb; sj >= sja; --sj) {
        cnt[C_SROW]++;
        const uint32_t j_lo = (uint32_t)sj * 64u, j_hi = j_lo + 64u < G.nm1 ? j_lo + 64u : G.nm1;
        const double z0 = wcoord(j_lo) - G.rho[2], z1 = wcoord(j_hi) + G.rho[2];
        double xlo, xhi;
        if (!x_interval(pl, gy0, gy1, z0, z1, &xlo, &xhi)) continue;
        int32_t sia, sib;
        if (!idx_range(xlo - G.rho[0], xhi + G.rho[0], 64, G.ns, &sia, &sib)) continue;
        /* which super-blocks of this super-row pass?  (interval hull of the passing ones) */
        int32_t spa = 1 << 30, spb = -1;
        double sy0 = 1e300, sy1 = -1e300;
        for (int32_t si = sib; si >= sia; --si) {
            cnt[C_STEST]++;
            const size_t s = (size_t)sj * G.ns + si;
            const uint32_t i_lo = (uint32_t)si * 64u, i_hi = i_lo + 64u < G.nm1 ? i_lo + 64u : G.nm1;
            if (box_hits(pl, wcoord(i_lo) - G.rho[0], wcoord(i_hi) + G.rho[0], G.smin[s] - G.rho[1], G.smax[s] + G.rho[1], z0, z1)) {
                if (si < spa) spa = si;
                if (si > spb) spb = si;
                sy0 = dmin(sy0, G.smin[s] - G.rho[1]); sy1 = dmax(sy1, G.smax[s] + G.rho[1]);
            }
        }
        if (spb < 0) continue;
        const uint32_t bj_hi = (j_hi + 7u) / 8u;
        for (int32_t bj = (int32_t)bj_hi - 1; bj >= (int32_t)(j_lo / 8u); --bj) {
            cnt[C_BROW]++;
            const uint32_t bj_lo_c = (uint32_t)bj * 8u, bj_hi_c = bj_lo_c + 8u < G.nm1 ? bj_lo_c + 8u : G.nm1;
            const double bz0 = wcoord(bj_lo_c) - G.rho[2], bz1 = wcoord(bj_hi_c) + G.rho[2];
            if (!x_interval(pl, sy0, sy1, bz0, bz1, &xlo, &xhi)) continue;
            int32_t bia, bib;
            if (!idx_range(xlo - G.rho[0], xhi + G.rho[0], 8, G.nb, &bia, &bib)) continue;
            if (bia < spa * 8) bia = spa * 8;
            if (bib > spb * 8 + 7) bib = spb * 8 + 7;
            int32_t bpa = 1 << 30, bpb = -1;
            double by0 = 1e300, by1 = -1e300;
            for (int32_t bi = bib; bi >= bia; --bi) {
                cnt[C_BTEST]++;
                const size_t b = (size_t)bj * G.nb + bi;
                const uint32_t i_lo = (uint32_t)bi * 8u, i_hi = i_lo + 8u < G.nm1 ? i_lo + 8u : G.nm1;
                if (box_hits(pl, wcoord(i_lo) - G.rho[0], wcoord(i_hi) + G.rho[0], G.bmin[b] - G.rho[1], G.bmax[b] + G.rho[1], bz0, bz1)) {
                    if (bi < bpa) bpa = bi;
                    if (bi > bpb) bpb = bi;
                    by0 = dmin(by0, G.bmin[b] - G.rho[1]); by1 = dmax(by1, G.bmax[b] + G.rho[1]);
                }
            }
            if (bpb < 0) continue;
            for (int32_t j = (int32_t)bj_hi_c - 1; j >= (int32_t)bj_lo_c; --j) {
                cnt[C_CROW]++;
                const double cz0 = wcoord((uint32_t)j) - G.rho[2], cz1 = wcoord((uint32_t)j + 1u) + G.rho[2];
                if (!x_interval(pl, by0, by1, cz0, cz1, &xlo, &xhi)) continue;
                int32_t ia, ib;
                if (!idx_range(xlo - G.rho[0], xhi + G.rho[0], 1, G.nm1, &ia, &ib)) continue;
                if (ia < bpa * 8) ia = bpa * 8;
                if (ib > bpb * 8 + 7) ib = bpb * 8 + 7;
                if (ib > (int32_t)G.nm1 - 1) ib = (int32_t)G.nm1 - 1;
                for (int32_t i = ib; i >= ia; --i) {
                    cnt[C_CTEST]++;
                    const size_t c = (size_t)j * G.nm1 + i;
                    if (!box_hits(pl, wcoord((uint32_t)i) - G.rho[0], wcoord((uint32_t)i + 1u) + G.rho[0],
                                  G.cmin[c] - G.rho[1], G.cmax[c] + G.rho[1], cz0, cz1)) continue;
                    cnt[C_EXACT]++;
                    for (int odd = 1; odd >= 0; --odd) {
                        cnt[C_TRI]++;
                        int r = prim_covers((uint32_t)i, (uint32_t)j, odd, px, py);
                        if (r > 0) return 2u * (uint32_t)c + (uint32_t)odd + 1u;
                        if (r < 0) return 0xFFFFFFFFu;
                    }
                }
            }
        }
    }
    return 0;
}

static void invert3(const double A[9], double inv[9])
{
    double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    double id = 1.0 / det;
    inv[0] = (A[4] * A[8] - A[5] * A[7]) * id; inv[1] = (A[2] * A[7] - A[1] * A[8]) * id; inv[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    inv[3] = (A[5] * A[6] - A[3] * A[8]) * id; inv[4] = (A[0] * A[8] - A[2] * A[6]) * id; inv[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    inv[6] = (A[3] * A[7] - A[4] * A[6]) * id; inv[7] = (A[1] * A[6] - A[0] * A[7]) * id; inv[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

VFO_API int wm_setup(const float u[44], uint32_t W, uint32_t H, uint32_t grid, const float *tex, uint32_t tw, uint32_t th, double eps_px)
{
    init_tables();
    model *g = &G;
    free(g->h); free(g->X); free(g->Y); free(g->fl); free(g->cmin); free(g->cmax); free(g->bmin); free(g->bmax); free(g->smin); free(g->smax);
    memset(g, 0, sizeof *g);
    uint32_t n = grid < 2 ? 2 : grid;
    g->n = n; g->nm1 = n - 1; g->nb = (n - 1 + 7) / 8; g->ns = (n - 1 + 63) / 64; g->W = W; g->H = H;
    size_t nv = (size_t)n * n, nc = (size_t)(n - 1) * (n - 1);
    g->h = malloc(nv * 4); g->X = malloc(nv * 4); g->Y = malloc(nv * 4); g->fl = malloc(nv);
    g->cmin = malloc(nc * 4); g->cmax = malloc(nc * 4);
    g->bmin = malloc((size_t)g->nb * g->nb * 4); g->bmax = malloc((size_t)g->nb * g->nb * 4);
    g->smin = malloc((size_t)g->ns * g->ns * 4); g->smax = malloc((size_t)g->ns * g->ns * 4);
    vsctx vc;
    vc.u = u; vc.n = n; vc.tex = tex; vc.tw = tw; vc.th = th;
    vc.spacing = fmaxf(u[36], 1e-8f); vc.exag = u[38]; vc.step = (2.0f * 1.5f) / ((float)n - 1.0f);
    g->spacing = vc.spacing; g->exag = vc.exag; g->step = vc.step;
    const float hwf = 0.5f * (float)W, hhf = 0.5f * (float)H;
    double gy0 = 1e300, gy1 = -1e300;
#pragma omp parallel for schedule(static) reduction(min : gy0) reduction(max : gy1)
    for (long j = 0; j < (long)n; ++j)
        for (uint32_t i = 0; i < n; ++i) {
            cvert v; vs_terrain(&vc, i, (uint32_t)j, &v);
            size_t k = (size_t)j * n + i;
            g->h[k] = v.a[0];
            double wy = (double)(v.a[0] * vc.exag);
            gy0 = dmin(gy0, wy); gy1 = dmax(gy1, wy);
            uint8_t fl = 0;
            int32_t X = 0, Y = 0;
            if (!isfinite(v.x) || !isfinite(v.y) || !isfinite(v.z) || !isfinite(v.w)) fl |= 4;
            if (v.z < 0.0f) fl |= 1;
            if (v.z > v.w) fl |= 2;
            if (!(fl & 4)) {
                if (!(v.w > 0.0f)) fl |= 8;
                else {
                    float rw = 1.0f / v.w;
                    float xf = fmaf(v.x * rw, hwf, hwf), yf = fmaf(-(v.y * rw), hhf, hhf);
                    if (!isfinite(xf) || !isfinite(yf)) fl |= 8;
                    else {
                        xf = fminf(fmaxf(xf, -4194304.0f), 4194304.0f); yf = fminf(fmaxf(yf, -4194304.0f), 4194304.0f);
                        X = (int32_t)rintf(xf * 256.0f); Y = (int32_t)rintf(yf * 256.0f);
                    }
                }
            }
            g->X[k] = X; g->Y[k] = Y; g->fl[k] = fl;
        }
    g->gy0 = gy0; g->gy1 = gy1;
    const double e = g->exag;
#pragma omp parallel for schedule(static)
    for (long j = 0; j < (long)n - 1; ++j)
        for (uint32_t i = 0; i + 1 < n; ++i) {
            size_t a = (size_t)j * n + i;
            double h0 = (double)(g->h[a] * (float)e), h1 = (double)(g->h[a + 1] * (float)e), h2 = (double)(g->h[a + n] * (float)e), h3 = (double)(g->h[a + n + 1] * (float)e);
            g->cmin[(size_t)j * (n - 1) + i] = (float)dmin(dmin(h0, h1), dmin(h2, h3));
            g->cmax[(size_t)j * (n - 1) + i] = (float)dmax(dmax(h0, h1), dmax(h2, h3));
        }
    for (uint32_t bj = 0; bj < g->nb; ++bj)
        for (uint32_t bi = 0; bi < g->nb; ++bi) {
            float lo = INFINITY, hi = -INFINITY;
            for (uint32_t j = bj * 8; j < bj * 8 + 8 && j < n - 1; ++j)
                for (uint32_t i = bi * 8; i < bi * 8 + 8 && i < n - 1; ++i) {
                    lo = fminf(lo, g->cmin[(size_t)j * (n - 1) + i]); hi = fmaxf(hi, g->cmax[(size_t)j * (n - 1) + i]);
                }
            g->bmin[(size_t)bj * g->nb + bi] = lo; g->bmax[(size_t)bj * g->nb + bi] = hi;
        }
    for (uint32_t sj = 0; sj < g->ns; ++sj)
        for (uint32_t si = 0; si < g->ns; ++si) {
            float lo = INFINITY, hi = -INFINITY;
            for (uint32_t bj = sj * 8; bj < sj * 8 + 8 && bj < g->nb; ++bj)
                for (uint32_t bi = si * 8; bi < si * 8 + 8 && bi < g->nb; ++bi) {
                    lo = fminf(lo, g->bmin[(size_t)bj * g->nb + bi]); hi = fmaxf(hi, g->bmax[(size_t)bj * g->nb + bi]);
                }
            g->smin[(size_t)sj * g->ns + si] = lo; g->smax[(size_t)sj * g->ns + si] = hi;
        }
    /* proj * view over world coordinates, in double from the float entries */
    double VP[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += (double)u[16 + 4 * k + r] * (double)u[4 * c + k];
            VP[4 * r + c] = s;
        }
    for (int c = 0; c < 4; ++c) { g->r0[c] = VP[c]; g->r1[c] = VP[4 + c]; g->r3[c] = VP[12 + c]; }
    /* FP32 rounding of the two mat-vecs: |clip_fp32 - clip_exact| per row, then pulled back to the world */
    const double uu = ldexp(1.0, -24);
    double wmax[4] = { 1.5 * g->spacing * (1 + 1e-6), dmax(fabs(gy0), fabs(gy1)), 1.5 * g->spacing * (1 + 1e-6), 1.0 };
    double S1[4], E1[4], E2[4];
    for (int k = 0; k < 4; ++k) {
        S1[k] = 0.0;
        for (int c = 0; c < 4; ++c) S1[k] += fabs((double)u[4 * c + k]) * wmax[c];
        E1[k] = 6.0 * uu * S1[k];
    }
    for (int r = 0; r < 4; ++r) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < 4; ++k) { a += fabs((double)u[16 + 4 * k + r]) * E1[k]; b += fabs((double)u[16 + 4 * k + r]) * S1[k]; }
        E2[r] = 2.0 * (a + 5.0 * uu * b);
    }
    double A[9] = { VP[0], VP[1], VP[2], VP[4], VP[5], VP[6], VP[12], VP[13], VP[14] }, Ai[9];
    invert3(A, Ai);
    const double ev[3] = { E2[0], E2[1], E2[3] };
    for (int c = 0; c < 3; ++c) g->rho[c] = fabs(Ai[3 * c]) * ev[0] + fabs(Ai[3 * c + 1]) * ev[1] + fabs(Ai[3 * c + 2]) * ev[2];
    g->ex = eps_px / (0.5 * W); g->ey = eps_px / (0.5 * H);
    return 0;
}

VFO_API void wm_info(double out[8])
{
    out[0] = G.rho[0]; out[1] = G.rho[1]; out[2] = G.rho[2]; out[3] = G.gy0; out[4] = G.gy1; out[5] = G.step * G.spacing;
}

/* walks the pixels of [x0,x1) x [y0,y1); vis_out and counts are (y1-y0) x (x1-x0) row-major; counts has C_NCOUNT u32 per pixel */
VFO_API void wm_walk_rect(int32_t x0, int32_t y0, int32_t x1, int32_t y1, uint32_t *vis_out, uint32_t *counts)
{
    const int32_t w = x1 - x0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int32_t y = y0; y < y1; ++y)
        for (int32_t x = x0; x < x1; ++x) {
            uint32_t cnt[C_NCOUNT];
            memset(cnt, 0, sizeof cnt);
            size_t o = (size_t)(y - y0) * w + (x - x0);
            vis_out[o] = walk_pixel(x, y, cnt);
            if (counts) memcpy(counts + o * C_NCOUNT, cnt, sizeof cnt);
        }
}
