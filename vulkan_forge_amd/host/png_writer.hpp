// png_writer.hpp -- RGBA8 PNG encoder on host zlib, parallel over rows.
// Stands in for image::RgbaImage::save (src/terrain/mod.rs:487-489, src/lib.rs:329-332): 8-bit RGBA, non-interlaced,
// per-row adaptive filter (minimum sum of absolute differences) -- any decoder returns the exact pixels that were rendered.
// The deflate stage dominates render_png end to end, so it runs pigz-style: the scanlines are cut into
// runs of rows, every run is deflated independently (raw deflate, sync-flushed, so the pieces concatenate into one valid
// zlib stream) and becomes its own IDAT chunk with its own CRC; the Adler-32 of the whole stream is combined from the
// runs'.  The filter stage is done by the GPU for terrain frames (vf_terrain_read_png_scanlines) and by
// filter_scanlines_cpu for everything else.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace vfh {

inline unsigned png_threads()
{
    if (const char *e = std::getenv("VF_PNG_THREADS")) { int v = std::atoi(e); if (v > 0) return (unsigned)v; }
    unsigned hc = std::thread::hardware_concurrency();
    return std::max(1u, std::min(hc ? hc : 1u, 16u));
}

template <typename F>
inline void parallel_for(size_t n, unsigned threads, F &&body)
{
    threads = (unsigned)std::min<size_t>(threads, n);
    if (threads <= 1) { for (size_t i = 0; i < n; ++i) body(i); return; }
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    std::vector<std::thread> pool;
    auto run = [&] {
        try { for (size_t i; (i = next.fetch_add(1)) < n;) body(i); } catch (...) { failed = true; }
    };
    for (unsigned k = 1; k < threads; ++k) pool.emplace_back(run);
    run();
    for (auto &th : pool) th.join();
    if (failed) throw std::runtime_error("PNG encoder worker failed");
}

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

// one scanline: filter-type byte + filtered bytes; the filter with the smallest sum of |signed residual| wins, first on ties
inline void filter_row(const uint8_t *cur, const uint8_t *up /* NULL above row 0 */, size_t stride, uint8_t *dst, uint8_t *scratch)
{
    uint64_t best = ~0ull;
    for (int f = 0; f < 5; ++f) {
        uint64_t sum = 0;
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= 4 ? cur[x - 4] : 0, b = up ? up[x] : 0, c = (up && x >= 4) ? up[x - 4] : 0;
            uint8_t v;
            switch (f) {
            case 0: v = cur[x]; break;
            case 1: v = uint8_t(cur[x] - a); break;
            case 2: v = uint8_t(cur[x] - b); break;
            case 3: v = uint8_t(cur[x] - ((a + b) >> 1)); break;
            default: v = uint8_t(cur[x] - paeth(a, b, c)); break;
            }
            scratch[x] = v;
            sum += v < 128 ? v : 256 - v;
        }
        if (sum < best) {
            best = sum;
            dst[0] = uint8_t(f);
            std::copy(scratch, scratch + stride, dst + 1);
        }
    }
}

inline std::vector<uint8_t> filter_scanlines_cpu(const uint8_t *rgba, uint32_t W, uint32_t H, unsigned threads)
{
    const size_t stride = (size_t)W * 4;
    std::vector<uint8_t> raw((stride + 1) * H);
    const size_t rows_per_task = 16;
    parallel_for((H + rows_per_task - 1) / rows_per_task, threads, [&](size_t task) {
        std::vector<uint8_t> scratch(stride);
        const uint32_t y1 = (uint32_t)std::min<size_t>(H, (task + 1) * rows_per_task);
        for (uint32_t y = (uint32_t)(task * rows_per_task); y < y1; ++y)
            filter_row(rgba + y * stride, y ? rgba + (y - 1) * stride : nullptr, stride, raw.data() + y * (stride + 1), scratch.data());
    });
    return raw;
}

// PNG file as a list of byte pieces (signature + IHDR, one IDAT per run of rows, IEND) from H scanlines of 4W+1 bytes
inline std::vector<std::vector<uint8_t>> png_pieces_from_scanlines(const uint8_t *scan, uint32_t W, uint32_t H, int level, unsigned threads)
{
    const size_t row_bytes = (size_t)W * 4 + 1;
    const size_t rows_per_run = std::max<size_t>(1, std::min<size_t>((1u << 20) / row_bytes, 1u << 16));   // ~1 MiB of scanlines per run
    const size_t nruns = std::max<size_t>(1, (H + rows_per_run - 1) / rows_per_run);
    std::vector<std::vector<uint8_t>> idat(nruns);
    std::vector<uLong> adler(nruns), length(nruns);
    parallel_for(nruns, threads, [&](size_t r) {
        const size_t y0 = r * rows_per_run, y1 = std::min<size_t>(H, y0 + rows_per_run);
        const uint8_t *src = scan + y0 * row_bytes;
        const size_t n = (y1 - y0) * row_bytes;
        z_stream zs{};
        // Z_RLE: matches at distance one only -- what adaptive filtering leaves of a rendered frame is runs (background, flat shading) and
        // noise, and on those the run-length strategy is both faster and smaller than Z_FILTERED's hash chains (1024² terrain frame, one
        // thread: 13.3 ms / 211 KB against 16.6 ms / 232 KB at level 2; round 5)
        (void)level;
        if (deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_RLE) != Z_OK) throw std::runtime_error("deflateInit2 failed");
        std::vector<uint8_t> &o = idat[r];
        const bool first = r == 0, last = r + 1 == nruns;
        // the output starts at a quarter of the input and doubles when deflate fills it: sized for the worst case (deflateBound) the 64
        // runs of a C4 frame zero-filled and faulted in 67 MB per call -- 8-10 ms of a cold process's first render_png calls (round 5)
        constexpr size_t kTail = 16;                                       // room kept behind the stream: Adler-32, CRC
        o.resize(8 + 2 + std::max<size_t>(n / 4, (size_t)1 << 16) + kTail);
        size_t at = 8;                                                     // chunk length + type are filled in below
        if (first) { o[at++] = 0x78; o[at++] = 0x01; }                       // zlib header: deflate, 32 KiB window, fastest
        const size_t stream_at = at;
        zs.next_in = const_cast<Bytef *>(src); zs.avail_in = (uInt)n;
        zs.next_out = o.data() + at; zs.avail_out = (uInt)(o.size() - at - kTail);
        bool ok = false;
        for (;;) {
            const int rc = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) break;
            if (last ? rc == Z_STREAM_END : (zs.avail_in == 0 && zs.avail_out != 0)) { ok = true; break; }
            if (zs.avail_out != 0) break;                                  // (no progress with room left: cannot happen)
            const size_t used = (size_t)(zs.next_out - o.data());
            o.resize(o.size() * 2);
            zs.next_out = o.data() + used; zs.avail_out = (uInt)(o.size() - used - kTail);
        }
        at = stream_at + zs.total_out;
        deflateEnd(&zs);
        if (!ok) throw std::runtime_error("PNG deflate failed");
        adler[r] = adler32(adler32(0L, Z_NULL, 0), src, (uInt)n);
        length[r] = (uLong)n;
        o.resize(at + (last ? 4 : 0) + 4);                                   // (Adler-32 of the whole stream +) CRC
    });
    uLong total = adler[0];
    for (size_t r = 1; r < nruns; ++r) total = adler32_combine(total, adler[r], (z_off_t)length[r]);
    for (size_t r = 0; r < nruns; ++r) {
        std::vector<uint8_t> &o = idat[r];
        size_t end = o.size() - 4;
        if (r + 1 == nruns) { o[end - 4] = uint8_t(total >> 24); o[end - 3] = uint8_t(total >> 16); o[end - 2] = uint8_t(total >> 8); o[end - 1] = uint8_t(total); }
        const uint32_t len = (uint32_t)(end - 8);
        o[0] = uint8_t(len >> 24); o[1] = uint8_t(len >> 16); o[2] = uint8_t(len >> 8); o[3] = uint8_t(len);
        o[4] = 'I'; o[5] = 'D'; o[6] = 'A'; o[7] = 'T';
    }
    parallel_for(nruns, threads, [&](size_t r) {
        std::vector<uint8_t> &o = idat[r];
        const size_t end = o.size() - 4;
        const uint32_t c = (uint32_t)crc32(0L, o.data() + 4, (uInt)(end - 4));
        o[end] = uint8_t(c >> 24); o[end + 1] = uint8_t(c >> 16); o[end + 2] = uint8_t(c >> 8); o[end + 3] = uint8_t(c);
    });
    std::vector<std::vector<uint8_t>> pieces;
    std::vector<uint8_t> head = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    uint8_t ihdr[13];
    ihdr[0] = uint8_t(W >> 24); ihdr[1] = uint8_t(W >> 16); ihdr[2] = uint8_t(W >> 8); ihdr[3] = uint8_t(W);
    ihdr[4] = uint8_t(H >> 24); ihdr[5] = uint8_t(H >> 16); ihdr[6] = uint8_t(H >> 8); ihdr[7] = uint8_t(H);
    ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;   // 8-bit, RGBA, deflate, adaptive, no interlace
    put_chunk(head, "IHDR", ihdr, 13);
    pieces.push_back(std::move(head));
    for (auto &o : idat) pieces.push_back(std::move(o));
    std::vector<uint8_t> tail;
    put_chunk(tail, "IEND", nullptr, 0);
    pieces.push_back(std::move(tail));
    return pieces;
}

inline std::vector<uint8_t> encode_png_rgba8(const uint8_t *rgba, uint32_t W, uint32_t H, int level = 2)
{
    const unsigned threads = png_threads();
    std::vector<uint8_t> raw = filter_scanlines_cpu(rgba, W, H, threads);
    std::vector<uint8_t> out;
    for (auto &p : png_pieces_from_scanlines(raw.data(), W, H, level, threads)) out.insert(out.end(), p.begin(), p.end());
    return out;
}

inline void write_pieces(const std::string &path, const std::vector<std::vector<uint8_t>> &pieces)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("failed to open '" + path + "' for writing");
    bool ok = true;
    for (auto &p : pieces) ok = ok && std::fwrite(p.data(), 1, p.size(), f) == p.size();
    const int rc = std::fclose(f);
    if (!ok || rc != 0) throw std::runtime_error("failed to write '" + path + "'");
}

// scanlines already filtered (the GPU path of render_png)
inline void write_png_scanlines(const std::string &path, const uint8_t *scan, uint32_t W, uint32_t H, int level = 2)
{
    write_pieces(path, png_pieces_from_scanlines(scan, W, H, level, png_threads()));
}

inline void write_png_rgba8(const std::string &path, const uint8_t *rgba, uint32_t W, uint32_t H, int level = 2)
{
    const unsigned threads = png_threads();
    std::vector<uint8_t> raw = filter_scanlines_cpu(rgba, W, H, threads);
    write_pieces(path, png_pieces_from_scanlines(raw.data(), W, H, level, threads));
}

} // namespace vfh
