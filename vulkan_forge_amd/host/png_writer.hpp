// png_writer.hpp -- RGBA8 PNG encoder on host zlib.
// Stands in for image::RgbaImage::save (src/terrain/mod.rs:487-489, src/lib.rs:329-332): 8-bit RGBA,
// non-interlaced, per-row adaptive filter (minimum sum of absolute differences), fast deflate level --
// any decoder returns the exact pixels that were rendered.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace vfh {

inline void put_be32(std::vector<uint8_t> &o, uint32_t v)
{
    o.push_back(uint8_t(v >> 24)); o.push_back(uint8_t(v >> 16)); o.push_back(uint8_t(v >> 8)); o.push_back(uint8_t(v));
}
inline void put_chunk(std::vector<uint8_t> &o, const char type[4], const uint8_t *data, size_t len)
{
    put_be32(o, (uint32_t)len);
    size_t start = o.size();
    o.insert(o.end(), type, type + 4);
    if (len) o.insert(o.end(), data, data + len);
    put_be32(o, (uint32_t)crc32(0L, o.data() + start, (uInt)(len + 4)));
}
inline uint8_t paeth(int a, int b, int c)
{
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return uint8_t((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}

inline std::vector<uint8_t> encode_png_rgba8(const uint8_t *rgba, uint32_t W, uint32_t H, int level = 2)
{
    const size_t stride = (size_t)W * 4;
    std::vector<uint8_t> raw((stride + 1) * H);
    std::vector<uint8_t> cand(stride);
    const std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < H; ++y) {
        const uint8_t *cur = rgba + y * stride;
        const uint8_t *up = y ? cur - stride : zero.data();
        uint64_t best = ~0ull;
        int best_f = 0;
        uint8_t *dst = raw.data() + y * (stride + 1);
        for (int f = 0; f < 5; ++f) {
            uint64_t sum = 0;
            for (size_t x = 0; x < stride; ++x) {
                int a = x >= 4 ? cur[x - 4] : 0, b = up[x], c = x >= 4 ? up[x - 4] : 0;
                uint8_t v;
                switch (f) {
                case 0: v = cur[x]; break;
                case 1: v = uint8_t(cur[x] - a); break;
                case 2: v = uint8_t(cur[x] - b); break;
                case 3: v = uint8_t(cur[x] - ((a + b) >> 1)); break;
                default: v = uint8_t(cur[x] - paeth(a, b, c)); break;
                }
                cand[x] = v;
                sum += v < 128 ? v : 256 - v;
            }
            if (sum < best) {
                best = sum; best_f = f;
                dst[0] = uint8_t(f);
                std::copy(cand.begin(), cand.end(), dst + 1);
            }
        }
        (void)best_f;
    }
    uLongf zcap = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zcap);
    if (compress2(z.data(), &zcap, raw.data(), (uLong)raw.size(), level) != Z_OK) throw std::runtime_error("PNG deflate failed");
    std::vector<uint8_t> out = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    uint8_t ihdr[13];
    ihdr[0] = uint8_t(W >> 24); ihdr[1] = uint8_t(W >> 16); ihdr[2] = uint8_t(W >> 8); ihdr[3] = uint8_t(W);
    ihdr[4] = uint8_t(H >> 24); ihdr[5] = uint8_t(H >> 16); ihdr[6] = uint8_t(H >> 8); ihdr[7] = uint8_t(H);
    ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;   // 8-bit, RGBA, deflate, adaptive, no interlace
    put_chunk(out, "IHDR", ihdr, 13);
    put_chunk(out, "IDAT", z.data(), zcap);
    put_chunk(out, "IEND", nullptr, 0);
    return out;
}

inline void write_png_rgba8(const std::string &path, const uint8_t *rgba, uint32_t W, uint32_t H)
{
    std::vector<uint8_t> png = encode_png_rgba8(rgba, W, H);
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("failed to open '" + path + "' for writing");
    size_t n = std::fwrite(png.data(), 1, png.size(), f);
    int rc = std::fclose(f);
    if (n != png.size() || rc != 0) throw std::runtime_error("failed to write '" + path + "'");
}

} // namespace vfh
