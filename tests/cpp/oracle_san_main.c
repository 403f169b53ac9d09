/* AddressSanitizer + UBSan run of the CPU oracle (tests/test_sanitizers.py): a few small frames that take every path of
 * oracle/vf_oracle.c -- clipping against the near plane (camera inside the terrain), ragged frame sizes, a band shard, several threads,
 * the SPEC_T32 fragment mode, the triangle path, grid_generate, the DEM helpers.  Test infrastructure. */
#include "../../oracle/vf_oracle.c"
#include <stdio.h>

static void frame(uint32_t W, uint32_t H, uint32_t grid, const float eye[3], const float tgt[3], float fovy, float zn, uint32_t tw, uint32_t th,
                  uint32_t rank, uint32_t nranks, int threads, int shade_mode)
{
    float u[44];
    const float up[3] = { 0.f, 1.f, 0.f };
    const char *err = vfo_look_at_uniforms(1, W, H, eye, tgt, up, fovy, zn, 100.0f, u);
    if (err) { fprintf(stderr, "%s\n", err); exit(2); }
    u[38] = 1.3f;
    float *tex = (float *)malloc((size_t)tw * th * sizeof(float));
    uint32_t s = 12345u + W + 31u * grid;
    for (size_t k = 0; k < (size_t)tw * th; ++k) { s = s * 1664525u + 1013904223u; tex[k] = (float)(s >> 8) / 16777216.0f * 0.5f - 0.25f; }
    uint8_t lut[1024];
    for (int k = 0; k < 1024; ++k) lut[k] = (uint8_t)(k * 7);
    uint8_t *rgba = (uint8_t *)malloc((size_t)W * H * 4);
    uint32_t *vis = (uint32_t *)malloc((size_t)W * H * 4);
    if (vfo_render_terrain_mode(u, W, H, grid, tex, tw, th, lut, 1, rank, nranks, 64, rgba, vis, threads, shade_mode) != 0) exit(3);
    size_t covered = 0;
    for (size_t k = 0; k < (size_t)W * H; ++k) covered += vis[k] != 0;
    printf("%ux%u grid %u: %zu covered\n", W, H, grid, covered);
    free(tex); free(rgba); free(vis);
}

int main(void)
{
    const float e0[3] = { 3.f, 2.f, 3.f }, o[3] = { 0.f, 0.f, 0.f }, inside[3] = { 0.2f, 0.3f, 0.4f }, low[3] = { 0.5f, 0.05f, 0.5f }, t2[3] = { 0.f, 0.2f, 0.f };
    frame(160, 120, 48, e0, o, 45.f, 0.1f, 48, 48, 0, 1, 1, 0);
    frame(131, 77, 37, e0, o, 45.f, 0.1f, 17, 9, 0, 1, 4, 0);
    frame(200, 150, 32, inside, o, 70.f, 0.1f, 32, 32, 0, 1, 2, 0);       /* near-plane clipping */
    frame(160, 120, 16, low, t2, 90.f, 0.05f, 1, 1, 0, 1, 1, 1);         /* grazing, SPEC_T32 */
    frame(128, 192, 24, e0, o, 45.f, 0.1f, 8, 8, 1, 2, 3, 0);            /* band shard */
    frame(64, 48, 2, e0, o, 45.f, 0.1f, 2, 2, 0, 1, 1, 0);               /* two big triangles */
    uint8_t *tri = (uint8_t *)malloc(96 * 64 * 4);
    if (vfo_render_triangle(96, 64, tri) != 0) return 4;
    free(tri);
    float *xy = (float *)malloc(5 * 4 * 2 * 4), *uv = (float *)malloc(5 * 4 * 2 * 4);
    uint32_t *idx = (uint32_t *)malloc(6 * 4 * 3 * 4);
    if (vfo_grid_generate(5, 4, 2.f, 1.f, "center", xy, uv, idx)) return 5;
    free(xy); free(uv); free(idx);
    float h[100], st[4], p1, p99;
    for (int k = 0; k < 100; ++k) h[k] = (float)((k * 37) % 11) - 5.f;
    vfo_dem_stats(h, 100, st);
    vfo_dem_normalize(h, 100, 0, 1e-8f, 0.f, 1.f, st);
    vfo_dem_stats(h, 100, st);
    vfo_dem_normalize(h, 100, 1, 1e-8f, 0.f, 1.f, st);
    if (vfo_dem_percentile_range(h, 100, &p1, &p99)) return 6;
    puts("oracle under ASan + UBSan: ok");
    return 0;
}
