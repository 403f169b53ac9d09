// AddressSanitizer + UBSan run of the host side's header-only pieces (tests/test_sanitizers.py): the camera / uniform arithmetic of
// vulkan_forge_amd/host/camera.hpp and the PNG writer (filter choice, parallel deflate runs, chunk assembly) of png_writer.hpp.
#include "../../vulkan_forge_amd/host/camera.hpp"
#include "../../vulkan_forge_amd/host/png_writer.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace vfh;

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const Mat4 v = look_at_rh({ 3.f, 2.f, 3.f }, { 0.f, 0.f, 0.f }, { 0.f, 1.f, 0.f });
    const Mat4 p = perspective_wgpu(to_radians(45.f), 4.f / 3.f, 0.1f, 100.f);
    Globals g;
    const Uniforms u = to_uniforms(g, v, p);
    if (!(u[0] == v[0]) || !(u[16] == p[0])) return 2;
    int thrown = 0;
    try { validate_camera_params({ 0.f, 0.f, 0.f }, { 0.f, 1.f, 0.f }, { 0.f, 1.f, 0.f }, 45.f, 0.1f, 100.f); } catch (const std::runtime_error &) { ++thrown; }
    try { validate_camera_params({ 3.f, 2.f, 3.f }, { 0.f, 0.f, 0.f }, { 0.f, 1.f, 0.f }, 200.f, 0.1f, 100.f); } catch (const std::runtime_error &) { ++thrown; }
    if (thrown != 2) return 3;
    // frames that exercise every filter type, one row, one column, a width that is no multiple of anything, and several deflate runs
    const uint32_t sizes[][2] = { { 1, 1 }, { 257, 1 }, { 1, 300 }, { 250, 131 }, { 1024, 700 } };
    for (const auto &wh : sizes) {
        const uint32_t W = wh[0], H = wh[1];
        std::vector<uint8_t> px((size_t)W * H * 4);
        uint32_t s = 99u + W;
        for (size_t k = 0; k < px.size(); ++k) { s = s * 1664525u + 1013904223u; px[k] = (k / (W * 4)) % 3 == 0 ? (uint8_t)(s >> 24) : (uint8_t)(k / 4); }
        const std::vector<uint8_t> png = encode_png_rgba8(px.data(), W, H);
        if (png.size() < 57 || png[1] != 'P') return 4;
        write_png_rgba8(dir + "/san_" + std::to_string(W) + "x" + std::to_string(H) + ".png", px.data(), W, H);
        const std::vector<uint8_t> scan = filter_scanlines_cpu(px.data(), W, H, 3);
        if (scan.size() != ((size_t)W * 4 + 1) * H) return 5;
        write_png_scanlines(dir + "/san_scan.png", scan.data(), W, H);
    }
    std::puts("host headers under ASan + UBSan: ok");
    return 0;
}
