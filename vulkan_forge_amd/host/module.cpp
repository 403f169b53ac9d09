// module.cpp -- CPython extension `_vulkan_forge`: the C++ host side above the C-ABI (include/vf_hip.h).
//
// Mirrors the Python-visible surface the reference registers in src/lib.rs:961-976 (names, signatures,
// return types, exception classes and messages) for the terrain-raster hot path:
//   classes   Renderer (triangle path), TerrainSpike, Scene
//   functions enumerate_adapters, device_probe, grid_generate, colormap_supported,
//             camera_look_at, camera_perspective, camera_view_proj
// All rendering goes through libvf_hip.so; there is no CPU fallback -- without a HIP device every
// object that needs the GPU raises RuntimeError("No suitable GPU adapter") like the reference.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include <sys/mman.h>

#include "../../include/vf_hip.h"
#include "../data/colormaps_rgba8.h"
#include "camera.hpp"
#include "png_writer.hpp"

namespace py = pybind11;
using namespace vfh;

namespace {

// ---- colormap registry: src/colormap/mod.rs --------------------------------------------------
const char *const kSupported[3] = { "viridis", "magma", "terrain" };   // :7 (case-sensitive)

std::string unknown_colormap(const std::string &name)
{
    return "Unknown colormap '" + name + "'. Supported: viridis, magma, terrain";   // :16,:32,:39
}
const uint8_t *resolve_lut(const std::string &name)
{
    if (name == "viridis") return VF_LUT_VIRIDIS;
    if (name == "magma") return VF_LUT_MAGMA;
    if (name == "terrain") return VF_LUT_TERRAIN;
    throw std::runtime_error(unknown_colormap(name));
}
// to_linear_u8_rgba, src/colormap/mod.rs:59-79 (the Rgba8Unorm fallback bytes)
void to_linear_u8_rgba(const uint8_t *src, uint8_t *dst)
{
    for (int i = 0; i < 256; ++i) {
        for (int ch = 0; ch < 3; ++ch) {
            float s = (float)src[4 * i + ch] / 255.0f;
            float l = s <= 0.04045f ? s / 12.92f : std::pow((s + 0.055f) / 1.055f, 2.4f);
            l = l < 0.0f ? 0.0f : (l > 1.0f ? 1.0f : l);
            dst[4 * i + ch] = (uint8_t)(l * 255.0f + 0.5f);
        }
        dst[4 * i + 3] = src[4 * i + 3];
    }
}

// ---- HIP context: one per process and device, like the reference's OnceCell (src/lib.rs:23-61) ----
[[noreturn]] void raise_vf(int rc)
{
    if (rc == VF_ERR_NO_DEVICE) throw std::runtime_error("No suitable GPU adapter");
    throw std::runtime_error(std::string(vf_last_error()));
}
void check(int rc) { if (rc != VF_OK) raise_vf(rc); }

int device_ordinal()
{
    const char *e = std::getenv("VF_HIP_DEVICE");
    return e ? std::atoi(e) : 0;
}

struct Ctx {
    vf_ctx *c = nullptr;
    ~Ctx() { /* process lifetime: the HIP runtime may already be torn down at exit */ }
};
vf_ctx *global_ctx()
{
    static std::mutex mu;
    static Ctx ctx;
    std::lock_guard<std::mutex> lk(mu);
    if (!ctx.c) check(vf_ctx_create(device_ordinal(), &ctx.c));
    return ctx.c;
}

// ---- frame-sized read-back buffers in page-locked host memory (round 5) ---------------------------------------------------------
// render_rgba returns a NumPy array OVER such a buffer instead of copying the frame into fresh pageable memory: the device writes it
// with one DMA transfer (C4: 64 MiB in 1.2 ms) where the staged copy into untouched pages was page-fault bound (2-5 ms; the
// reference maps a fresh buffer per call, src/terrain/mod.rs:446-451).  The array owns its buffer through a capsule; when the array
// dies the buffer goes back to the pool, so a render loop alternates between two buffers and never allocates again.  A buffer is
// 2 MiB-aligned huge-page memory registered with the runtime (vf_host_alloc): 2.6 ms to make for a C4 frame, where hipHostMalloc took 9.
class PinnedPool {
public:
    static PinnedPool &get() { static PinnedPool *p = new PinnedPool; return *p; }   // (process lifetime: never destroyed, like the context)
    void *take(size_t bytes)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t k = 0; k < idle.size(); ++k)
                if (idle[k].second == bytes) { void *p = idle[k].first; idle.erase(idle.begin() + (long)k); return p; }
        }
        void *p = nullptr;
        check(vf_host_alloc(bytes, &p));
        { std::lock_guard<std::mutex> lk(mu); made_bytes += bytes; }
        return p;
    }
    // ... for a caller that can do without: nullptr once `max_out` bytes of page-locked memory are out with arrays the caller keeps.
    // render_rgba allows three frames (round 6): a loop that DROPS its frames alternates between two pooled buffers for good, a loop
    // that KEEPS them gets ordinary arrays from the fourth on -- filled through the handle's ring of pinned chunks by four host threads,
    // 2.1 ms for a C4 frame, where a page-locked buffer of its own per frame costs 2.6 ms to make on top of the 1.3 ms transfer (round
    // 5: 4.9 ms per kept frame) and cannot be swapped.  render_batch allows kMaxOutBytes.
    void *take_or_null(size_t bytes, size_t max_out)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            bool have = false;
            size_t held = 0;
            for (auto &b : idle) { have |= b.second == bytes; held += b.second; }
            if (!have && (made_bytes - held) + bytes > max_out) return nullptr;
        }
        return take(bytes);
    }
    static constexpr size_t kMaxOutBytes = (size_t)1 << 30;
    void give(void *p, size_t bytes)
    {
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> lk(mu);
            idle.emplace_back(p, bytes);
            size_t held = 0;
            for (auto &b : idle) held += b.second;
            while (idle.size() > kMaxIdle || held > kMaxIdleBytes) { held -= idle.front().second; made_bytes -= idle.front().second; drop.push_back(idle.front().first); idle.erase(idle.begin()); }
        }
        for (void *d : drop) vf_host_free(d);
    }
private:
    static constexpr size_t kMaxIdle = 4, kMaxIdleBytes = (size_t)1 << 30;
    size_t made_bytes = 0;                                      // page-locked and not freed: idle buffers + those out with arrays
    std::mutex mu;
    std::vector<std::pair<void *, size_t>> idle;
};
struct PinnedLease { void *p; size_t bytes; };
py::capsule pinned_owner(void *p, size_t bytes)
{
    return py::capsule(new PinnedLease{ p, bytes }, [](void *v) { auto *l = static_cast<PinnedLease *>(v); PinnedPool::get().give(l->p, l->bytes); delete l; });
}

// PyO3's `&mut self` borrow (Threading): a method entered while another thread is inside one of the same object
// raises RuntimeError("Already borrowed") instead of racing.  It matters here because render / read-back release the GIL:
// per-handle state (plan sets, pinned read-back buffers, frame counters) is not synchronised below the C-ABI ("calls on one
// handle are not re-entrant", include/vf_hip.h).
class Borrow {
public:
    explicit Borrow(std::atomic<bool> &flag, const char *what = "Already borrowed") : f(flag)
    {
        bool expected = false;
        if (!f.compare_exchange_strong(expected, true, std::memory_order_acquire)) throw std::runtime_error(what);
    }
    ~Borrow() { f.store(false, std::memory_order_release); }
    Borrow(const Borrow &) = delete;
    Borrow &operator=(const Borrow &) = delete;
private:
    std::atomic<bool> &f;
};

Vec3 v3(const std::tuple<float, float, float> &t) { return { std::get<0>(t), std::get<1>(t), std::get<2>(t) }; }

// mat4_to_numpy, src/camera.rs:94-112: (4,4) float32 C-contiguous in mathematical (row, col) indexing
py::array_t<float> mat4_to_numpy(const Mat4 &m)
{
    py::array_t<float> a({ 4, 4 });
    auto r = a.mutable_unchecked<2>();
    for (int row = 0; row < 4; ++row)
        for (int col = 0; col < 4; ++col) r(row, col) = m[4 * col + row];
    return a;
}

// ---- TerrainSpike / Scene -----------------------------------------------------------------------
class TerrainObject {
public:
    // kind 0: TerrainSpike::new (src/terrain/mod.rs:259-407); kind 1: Scene::new (src/scene/mod.rs:60-206)
    TerrainObject(int kind, uint32_t width, uint32_t height, py::object grid, py::object colormap) : W(width), H(height)
    {
        uint32_t g = grid.is_none() ? 128u : grid.cast<uint32_t>();
        n = g < 2 ? 2 : g;
        std::string cmap = colormap.is_none() ? "viridis" : colormap.cast<std::string>();
        bool known = false;
        for (const char *s : kSupported) known |= cmap == s;
        if (!known) throw std::runtime_error(unknown_colormap(cmap));
        const uint8_t *srgb_bytes = resolve_lut(cmap);
        // ColormapLUT::new format selection (src/terrain/mod.rs:45-60): sRGB sampling is always
        // available here, so only VF_FORCE_LUT_UNORM selects the CPU-linearised UNORM bytes.
        bool force_unorm = std::getenv("VF_FORCE_LUT_UNORM") != nullptr;
        uint8_t lin[1024];
        if (force_unorm) to_linear_u8_rgba(srgb_bytes, lin);
        lut_format = force_unorm ? "Rgba8Unorm" : "Rgba8UnormSrgb";

        check(vf_terrain_create(global_ctx(), W, H, n, force_unorm ? lin : srgb_bytes, force_unorm ? 0 : 1, &t));

        // build_view_matrices (src/terrain/mod.rs:681-691) / SceneGlobals::default (src/scene/mod.rs:17-23,119)
        view = look_at_rh({ 3.f, 2.f, 3.f }, { 0.f, 0.f, 0.f }, { 0.f, 1.f, 0.f });
        proj = perspective_wgpu(to_radians(45.0f), (float)W / (float)H, 0.1f, 100.0f);
        if (kind == 0) globals.sun_dir = normalize({ 0.5f, 1.0f, 0.3f });   // R4 override, src/terrain/mod.rs:325-327
        push_uniforms();
        if (kind == 1) {
            const float dummy[4] = { 0.00f, 0.25f, 0.50f, 0.75f };   // 2x2 gradient, src/scene/mod.rs:142-189
            check(vf_terrain_set_height(t, dummy, 2, 2));
        }
    }
    ~TerrainObject() { if (t) vf_terrain_destroy(t); }
    TerrainObject(const TerrainObject &) = delete;
    TerrainObject &operator=(const TerrainObject &) = delete;

    // src/terrain/mod.rs:498-535, src/scene/mod.rs:208-224
    void set_camera_look_at(std::tuple<float, float, float> eye, std::tuple<float, float, float> target,
                            std::tuple<float, float, float> up, float fovy_deg, float znear, float zfar)
    {
        Borrow b(busy);
        float aspect = (float)W / (float)H;
        validate_camera_params(v3(eye), v3(target), v3(up), fovy_deg, znear, zfar);
        view = look_at_rh(v3(eye), v3(target), v3(up));
        proj = perspective_wgpu(to_radians(fovy_deg), aspect, znear, zfar);
        push_uniforms();
    }

    // Scene::set_height_from_r32f, src/scene/mod.rs:226-276
    void set_height_from_r32f(py::object obj)
    {
        if (!py::isinstance<py::array>(obj)) throw py::type_error("argument 'height_r32f': expected a 2-D numpy.ndarray of float32");
        py::array arr = py::reinterpret_borrow<py::array>(obj);
        if (arr.ndim() != 2 || !arr.dtype().is(py::dtype::of<float>()))
            throw py::type_error("argument 'height_r32f': expected a 2-D numpy.ndarray of float32");
        if (!(arr.flags() & py::array::c_style)) throw std::runtime_error("height must be C-contiguous float32[H,W]");
        uint32_t h = (uint32_t)arr.shape(0), w = (uint32_t)arr.shape(1);
        Borrow b(busy);
        check(vf_terrain_set_height(t, static_cast<const float *>(arr.data()), w, h));
    }

    void render_into(uint8_t *dst, uint32_t rows)          // caller holds the borrow
    {
        py::gil_scoped_release nogil;
        int rc = vf_terrain_render(t, nullptr);
        if (rc == VF_OK) rc = vf_terrain_read_rgba(t, dst, 0, rows);
        if (rc != VF_OK) { py::gil_scoped_acquire gil; raise_vf(rc); }
    }
    std::vector<uint8_t> render_pixels()                    // caller holds the borrow
    {
        uint32_t rows = 0;
        check(vf_terrain_local_rows(t, &rows));
        std::unique_ptr<uint8_t[]> raw(new uint8_t[(size_t)rows * W * 4]);     // not value-initialised: every byte is overwritten
        render_into(raw.get(), rows);
        return std::vector<uint8_t>(raw.get(), raw.get() + (size_t)rows * W * 4);
    }

    // render_png, src/terrain/mod.rs:409-491, src/scene/mod.rs:278-335
    void render_png(const std::string &path)
    {
        Borrow b(busy);
        uint32_t rows = 0;
        check(vf_terrain_local_rows(t, &rows));
        if (rows != H) {                                       // band-sharded handle: its rows only, filtered on the host
            std::vector<uint8_t> px = render_pixels();
            py::gil_scoped_release nogil;
            write_png_rgba8(path, px.data(), W, rows);
            return;
        }
        // whole frame: the GPU filters the scanlines into pinned host memory, the host deflates them in parallel
        py::gil_scoped_release nogil;
        const uint8_t *scan = nullptr;
        size_t nbytes = 0;
        int rc = vf_terrain_render(t, nullptr);
        if (rc == VF_OK) rc = vf_terrain_read_png_scanlines(t, &scan, &nbytes);
        if (rc != VF_OK) { py::gil_scoped_acquire gil; raise_vf(rc); }
        write_png_scanlines(path, scan, W, H);
    }

    // extension (not in the reference): the frame as (H, W, 4) uint8 without the PNG round trip
    py::array_t<uint8_t> render_rgba()
    {
        Borrow b(busy);
        uint32_t rows = 0;
        check(vf_terrain_local_rows(t, &rows));
        const size_t bytes = (size_t)rows * W * 4;
        // small frames: an ordinary array.  (Round 5 kept an object's FIRST frame-sized result in ordinary memory too, when page-locking
        // a C4 frame cost 9 ms; huge pages + registration make the buffer in 2.6 ms, less than the page faults of one copy into fresh
        // memory -- vf_host_alloc, tools/micro/pin_cost.hip.)
        if (bytes < ((size_t)4 << 20) || std::getenv("VF_RGBA_PAGEABLE")) {
            py::array_t<uint8_t> a({ (py::ssize_t)rows, (py::ssize_t)W, (py::ssize_t)4 });
            render_into(a.mutable_data(), rows);
            return a;
        }
        // frame-sized: the array lives in page-locked memory from the pool (above) -- one DMA transfer, no host copy
        void *p = PinnedPool::get().take_or_null(bytes, 3 * bytes);
        if (!p) {                                                // (the caller keeps its frames: ordinary arrays from here on)
            // ... in huge pages where the host has them: the copy out of the pinned ring into a fresh array is page-fault bound, and a
            // 2 MiB page takes one fault where 4 KiB pages take 512
            constexpr size_t kHuge = (size_t)2 << 20;
            void *q = nullptr;
            const size_t r = (bytes + kHuge - 1) & ~(kHuge - 1);
            if (posix_memalign(&q, kHuge, r) == 0) {
                (void)madvise(q, r, MADV_HUGEPAGE);
                py::capsule owner(q, [](void *v) { std::free(v); });
                render_into(static_cast<uint8_t *>(q), rows);
                return py::array_t<uint8_t>({ (py::ssize_t)rows, (py::ssize_t)W, (py::ssize_t)4 }, static_cast<uint8_t *>(q), owner);
            }
            py::array_t<uint8_t> a({ (py::ssize_t)rows, (py::ssize_t)W, (py::ssize_t)4 });
            render_into(a.mutable_data(), rows);
            return a;
        }
        py::capsule owner = pinned_owner(p, bytes);             // (returns the buffer if anything below throws)
        render_into(static_cast<uint8_t *>(p), rows);
        return py::array_t<uint8_t>({ (py::ssize_t)rows, (py::ssize_t)W, (py::ssize_t)4 }, static_cast<uint8_t *>(p), owner);
    }
    // extension (BASELINE config 5, "batch of camera look-ats over one terrain"): every pose is what set_camera_look_at + render
    // does per pose (src/scene/mod.rs:208-224, :278-335), queued back to back on the GPU; poses = sequence of
    // (eye, target, up, fovy_deg, znear, zfar).  Returns the frames as a list of (H, W, 4) uint8 arrays, or -- with `paths`, one per
    // pose -- writes PNG files and returns None.  The camera afterwards is the last pose's, as after the per-pose loop.
    py::object render_batch(py::sequence poses, py::object paths)
    {
        Borrow b(busy);
        uint32_t rows = 0;
        check(vf_terrain_local_rows(t, &rows));
        if (rows != H) throw std::runtime_error("render_batch needs the whole frame on one object (a band shard renders its rows with render_rgba)");
        const size_t n = (size_t)py::len(poses);
        std::vector<std::string> files;
        if (!paths.is_none()) {
            files = paths.cast<std::vector<std::string>>();
            if (files.size() != n) throw py::value_error("paths must hold one file name per pose");
        }
        const float aspect = (float)W / (float)H;
        std::vector<float> blocks(44 * n);
        Mat4 v{}, p{};
        for (size_t k = 0; k < n; ++k) {
            auto pose = poses[k].cast<std::tuple<std::tuple<float, float, float>, std::tuple<float, float, float>, std::tuple<float, float, float>, float, float, float>>();
            validate_camera_params(v3(std::get<0>(pose)), v3(std::get<1>(pose)), v3(std::get<2>(pose)), std::get<3>(pose), std::get<4>(pose), std::get<5>(pose));
            v = look_at_rh(v3(std::get<0>(pose)), v3(std::get<1>(pose)), v3(std::get<2>(pose)));
            p = perspective_wgpu(to_radians(std::get<3>(pose)), aspect, std::get<4>(pose), std::get<5>(pose));
            const Uniforms u = to_uniforms(globals, v, p);
            std::memcpy(blocks.data() + 44 * k, u.data(), 44 * sizeof(float));
        }
        if (n == 0) return files.empty() && paths.is_none() ? py::object(py::list()) : py::object(py::none());
        const size_t bytes = (size_t)H * W * 4;
        view = v; proj = p;                                     // the camera of the last pose stays, as after a loop of set_camera_look_at
        if (!files.empty()) {
            // PNG files: the poses go through in groups over a few pooled page-locked buffers -- every group one batch call, its frames
            // encoded before the next group is drawn -- so the page-locked footprint is kGroup frames whatever the number of poses
            // (advisor, round 5: n buffers up front were 4 GiB of unswappable memory and n registrations for 64 poses at 4096^2)
            constexpr size_t kGroup = 4;
            std::vector<py::capsule> owners;
            std::vector<uint8_t *> dst;
            for (size_t k = 0; k < std::min(n, kGroup); ++k) { void *q = PinnedPool::get().take(bytes); owners.push_back(pinned_owner(q, bytes)); dst.push_back(static_cast<uint8_t *>(q)); }
            {
                py::gil_scoped_release nogil;
                for (size_t k0 = 0; k0 < n; k0 += kGroup) {
                    const size_t m = std::min(kGroup, n - k0);
                    const int rc = vf_terrain_render_batch_host(t, blocks.data() + 44 * k0, (uint32_t)m, dst.data());
                    if (rc != VF_OK) { py::gil_scoped_acquire gil; raise_vf(rc); }
                    for (size_t k = 0; k < m; ++k) write_png_rgba8(files[k0 + k], dst[k], W, H);
                }
            }
            push_uniforms();
            return py::none();
        }
        // arrays: page-locked while the pool's allowance lasts (one DMA each, beside the next poses' kernels), ordinary memory beyond
        // (the runtime stages those copies: correct, slower)
        py::list frames;
        std::vector<uint8_t *> dst(n);
        for (size_t k = 0; k < n; ++k) {
            void *q = bytes >= ((size_t)4 << 20) ? PinnedPool::get().take_or_null(bytes, PinnedPool::kMaxOutBytes) : nullptr;
            if (q) {
                py::capsule owner = pinned_owner(q, bytes);
                frames.append(py::array_t<uint8_t>({ (py::ssize_t)H, (py::ssize_t)W, (py::ssize_t)4 }, static_cast<uint8_t *>(q), owner));
                dst[k] = static_cast<uint8_t *>(q);
            } else {
                py::array_t<uint8_t> a({ (py::ssize_t)H, (py::ssize_t)W, (py::ssize_t)4 });
                dst[k] = a.mutable_data();
                frames.append(a);
            }
        }
        {
            py::gil_scoped_release nogil;
            const int rc = vf_terrain_render_batch_host(t, blocks.data(), (uint32_t)n, dst.data());
            if (rc != VF_OK) { py::gil_scoped_acquire gil; raise_vf(rc); }
        }
        push_uniforms();
        return frames;
    }
    // extension: visible primitive id + 1 per pixel of the last render (0 = background)
    py::array_t<uint32_t> debug_visibility()
    {
        Borrow b(busy);
        uint32_t rows = 0;
        check(vf_terrain_local_rows(t, &rows));
        py::array_t<uint32_t> a({ (py::ssize_t)rows, (py::ssize_t)W });
        check(vf_terrain_read_visibility(t, a.mutable_data()));
        return a;
    }
    // extension: multi-GPU band ownership (DESIGN.md "Sharding")
    void set_shard(uint32_t rank, uint32_t nranks, uint32_t band_h) { Borrow b(busy); check(vf_terrain_set_shard(t, rank, nranks, band_h)); }
    // extension: "reference" = fs_main as coded; "spec_t32" = the documented-only stage (forward-difference normals + Reinhard)
    void set_shade_mode(const std::string &mode)
    {
        Borrow b(busy);
        if (mode == "reference") check(vf_terrain_set_shade_mode(t, VF_SHADE_REFERENCE));
        else if (mode == "spec_t32") check(vf_terrain_set_shade_mode(t, VF_SHADE_SPEC_T32));
        else throw py::value_error("shade mode must be 'reference' or 'spec_t32'");
    }
    // extension: "fast" (default) = hardware rcp / rsq / sin / cos / log / exp in the fragment stage, within 1 LSB of "exact"
    // (IEEE binary32 in a fixed order, the CPU oracle bit for bit); visibility is identical
    void set_shade_precision(const std::string &precision)
    {
        Borrow b(busy);
        if (precision == "exact") check(vf_terrain_set_shade_precision(t, VF_PRECISION_EXACT));
        else if (precision == "fast") check(vf_terrain_set_shade_precision(t, VF_PRECISION_FAST));
        else throw py::value_error("shade precision must be 'exact' or 'fast'");
    }
    py::dict last_timings()
    {
        Borrow b(busy);
        vf_timings tm;
        check(vf_terrain_timings(t, &tm));
        py::dict d;
        d["ranges_ms"] = tm.ranges_ms; d["plan_ms"] = tm.plan_ms; d["tile_ms"] = tm.tile_ms; d["total_ms"] = tm.total_ms;
        d["blocks_rasterised"] = tm.blocks_rasterised; d["tiles"] = tm.tiles; d["frames"] = tm.frames; d["blocks_distinct"] = tm.blocks_distinct;
        return d;
    }
    void enable_timing(bool on) { Borrow b(busy); check(vf_terrain_enable_timing(t, on ? 1 : 0)); }

    // src/terrain/mod.rs:537-546
    py::array_t<float> debug_uniforms_f32() const
    {
        if (busy.load(std::memory_order_acquire)) throw std::runtime_error("Already mutably borrowed");   // a `&self` method, PyO3-style
        py::array_t<float> a(44);
        std::memcpy(a.mutable_data(), last.data(), 44 * sizeof(float));
        return a;
    }
    std::string debug_lut_format() const { return lut_format; }   // src/terrain/mod.rs:493-496

private:
    void push_uniforms()
    {
        last = to_uniforms(globals, view, proj);
        check(vf_terrain_set_uniforms(t, last.data()));
    }
    uint32_t W, H, n = 128;
    vf_terrain *t = nullptr;
    Globals globals;
    Mat4 view{}, proj{};
    Uniforms last{};
    std::string lut_format;
    mutable std::atomic<bool> busy{false};
};

class TerrainSpike : public TerrainObject {
public:
    TerrainSpike(uint32_t w, uint32_t h, py::object grid, py::object colormap) : TerrainObject(0, w, h, grid, colormap) {}
};
class Scene : public TerrainObject {
public:
    Scene(uint32_t w, uint32_t h, py::object grid, py::object colormap) : TerrainObject(1, w, h, grid, colormap) {}
};

// ---- Renderer: the triangle smoke path (src/lib.rs:245-334, 685-721) + the DEM path (:336-682) ---------------------
class Renderer {
public:
    Renderer(uint32_t w, uint32_t h) : W(w), H(h) { global_ctx(); }
    ~Renderer() { if (dem) vf_dem_destroy(dem); }
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;
    std::string info() const { return "Renderer " + std::to_string(W) + "x" + std::to_string(H) + ", format=Rgba8UnormSrgb"; }
    py::array_t<uint8_t> render_triangle_rgba()
    {
        py::array_t<uint8_t> a({ (py::ssize_t)H, (py::ssize_t)W, (py::ssize_t)4 });
        check(vf_triangle_render(global_ctx(), W, H, a.mutable_data()));
        return a;
    }
    void render_triangle_png(const std::string &path)
    {
        std::vector<uint8_t> px((size_t)W * H * 4);
        check(vf_triangle_render(global_ctx(), W, H, px.data()));
        write_png_rgba8(path, px.data(), W, H);
    }

    // add_terrain, src/lib.rs:336-421
    void add_terrain(py::object heightmap, std::tuple<float, float> spacing, float exaggeration, const std::string &colormap)
    {
        if (std::get<0>(spacing) <= 0.0f || std::get<1>(spacing) <= 0.0f) throw std::runtime_error("spacing components must be > 0");
        if (exaggeration <= 0.0f) throw std::runtime_error("exaggeration must be > 0");
        const char *kind = "heightmap must be a 2-D NumPy array of dtype float32 or float64";
        if (!py::isinstance<py::array>(heightmap)) throw std::runtime_error(kind);
        py::array arr = py::reinterpret_borrow<py::array>(heightmap);
        const bool f32 = arr.dtype().is(py::dtype::of<float>()), f64 = arr.dtype().is(py::dtype::of<double>());
        if (arr.ndim() != 2 || !(f32 || f64)) throw std::runtime_error(kind);
        // a non-contiguous float32 array fails the reference's f32 attempt and then its f64 downcast: the dtype message
        // (src/lib.rs:351-372); only a non-contiguous float64 array reports the layout (:374-376)
        if (!(arr.flags() & py::array::c_style)) throw std::runtime_error(f32 ? kind : "heightmap must be C-contiguous (row-major)");
        const uint32_t h = (uint32_t)arr.shape(0), w = (uint32_t)arr.shape(1);
        if (w == 0 || h == 0) throw std::runtime_error("heightmap cannot be empty");
        // staged: the uploaded terrain replaces the current one only when the whole call succeeds (src/lib.rs:409-416 assigns
        // self.terrain last); the height range alone is stored before the colormap check, as in the reference (:394)
        struct Staged {
            vf_dem *d = nullptr;
            ~Staged() { if (d) vf_dem_destroy(d); }
        } fresh;
        check(vf_dem_create(global_ctx(), &fresh.d));
        if (f32) check(vf_dem_set_heights_f32(fresh.d, static_cast<const float *>(arr.data()), w, h, exaggeration));
        else check(vf_dem_set_heights_f64(fresh.d, static_cast<const double *>(arr.data()), w, h, exaggeration));
        // TerrainMeta::compute_and_store_h_range (src/renderer.rs:26-30): 1-99 percentile clamp
        float p1 = 0, p99 = 1;
        check(vf_dem_percentile_range(fresh.d, &p1, &p99));
        h_min = p1; h_max = std::fmax(p99, p1 + 1e-5f);
        bool known = false;
        for (const char *s : kSupported) known |= colormap == s;
        if (!known) throw std::runtime_error(unknown_colormap(colormap));    // after the range, like the reference (:398-407)
        std::swap(dem, fresh.d);                                             // (the previous terrain, if any, is released by `fresh`)
        has_terrain = true;
        terrain_colormap = colormap;
    }
    void need_terrain() const { if (!has_terrain) throw std::runtime_error("no terrain uploaded; call add_terrain() first"); }
    std::tuple<float, float, float, float> terrain_stats()                  // src/lib.rs:423-429
    {
        need_terrain();
        float st[4];
        check(vf_dem_stats(dem, st));
        return { st[0], st[1], st[2], st[3] };
    }
    void set_height_range(float mn, float mx)                               // src/renderer.rs:32-42
    {
        if (!std::isfinite(mn) || !std::isfinite(mx)) throw py::value_error("min/max must be finite floats");
        if (mn >= mx) throw py::value_error("min must be < max");
        h_min = mn; h_max = mx;
    }
    void set_sun(float elevation_deg, float azimuth_deg)                    // src/lib.rs:441-462
    {
        if (!std::isfinite(elevation_deg) || !std::isfinite(azimuth_deg)) throw py::value_error("angles must be finite");
        const float k = 3.14159265358979323846f / 180.0f;
        const float el = elevation_deg * k, az = azimuth_deg * k;
        globals.sun_dir = normalize_or_zero({ std::cos(el) * std::cos(az), std::sin(el), std::cos(el) * std::sin(az) });
    }
    void set_exposure(float exposure)                                       // src/lib.rs:464-473
    {
        if (!std::isfinite(exposure) || exposure <= 0.0f) throw py::value_error("exposure must be > 0");
        globals.exposure = exposure;
    }
    void normalize_terrain(const std::string &mode, py::object range, py::object eps)   // src/lib.rs:476-493
    {
        need_terrain();
        std::string m = mode;
        for (auto &c : m) c = (char)std::tolower((unsigned char)c);
        int code = m == "minmax" ? 0 : (m == "zscore" ? 1 : -1);
        if (code < 0) throw std::runtime_error("mode must be 'minmax' or 'zscore'");
        float e = eps.is_none() ? 1e-8f : eps.cast<float>();
        std::tuple<float, float> r = range.is_none() ? std::make_tuple(0.0f, 1.0f) : range.cast<std::tuple<float, float>>();
        check(vf_dem_normalize(dem, code, std::get<0>(r), std::get<1>(r), e));
    }
    void upload_height_r32f() { need_terrain(); check(vf_dem_upload_height(dem)); }     // src/lib.rs:495-571
    py::array_t<float> debug_read_height_patch(uint32_t x, uint32_t y, uint32_t w, uint32_t h)   // :573-666
    {
        if (w == 0 || h == 0) throw std::runtime_error("patch dimensions must be > 0");
        py::array_t<float> out({ (py::ssize_t)h, (py::ssize_t)w });
        uint32_t tw = 0, th = 0;
        if (dem) check(vf_dem_texture_size(dem, &tw, &th));
        if (tw == 0) { std::memset(out.mutable_data(), 0, (size_t)w * h * sizeof(float)); return out; }   // no texture yet: zeros (:580-588)
        check(vf_dem_read_patch(dem, x, y, w, h, out.mutable_data()));
        return out;
    }
    py::array_t<float> read_full_height_texture()                          // src/lib.rs:668-681
    {
        need_terrain();
        uint32_t tw = 0, th = 0;
        check(vf_dem_texture_size(dem, &tw, &th));
        if (tw == 0) throw std::runtime_error("no height texture uploaded; call upload_height_r32f() first");
        return debug_read_height_patch(0, 0, tw, th);
    }
private:
    uint32_t W, H;
    vf_dem *dem = nullptr;
    bool has_terrain = false;
    float h_min = 0.0f, h_max = 1.0f;          // TerrainMeta (src/renderer.rs:7-22)
    Globals globals;
    std::string terrain_colormap;
};

// ---- module functions ------------------------------------------------------------------------------
// grid_generate, src/terrain/mesh.rs:149-203 (validation strings :161-173 raised as ValueError)
py::tuple grid_generate(uint32_t nx, uint32_t nz, std::tuple<float, float> spacing, py::object origin)
{
    if (nx < 2 || nz < 2) throw py::value_error("nx and nz must be >= 2");
    float dx = std::get<0>(spacing), dy = std::get<1>(spacing);
    if (!std::isfinite(dx) || !std::isfinite(dy) || dx <= 0.0f || dy <= 0.0f) throw py::value_error("spacing components must be finite and > 0");
    std::string o = origin.is_none() ? "center" : origin.cast<std::string>();
    if (o != "center") throw py::value_error("origin must be 'center'");
    py::ssize_t nv = (py::ssize_t)nx * nz, ni = 6 * (py::ssize_t)(nx - 1) * (nz - 1);
    py::array_t<float> xy({ nv, (py::ssize_t)2 }), uv({ nv, (py::ssize_t)2 });
    py::array_t<uint32_t> idx(ni);
    check(vf_grid_generate(global_ctx(), nx, nz, dx, dy, xy.mutable_data(), uv.mutable_data(), idx.mutable_data()));
    return py::make_tuple(xy, uv, idx);
}

std::vector<std::string> colormap_supported() { return { "viridis", "magma", "terrain" }; }   // src/colormap/mod.rs:44-47
// extension: the 256 x 1 RGBA8 texels the registry resolves a name to (src/colormap/mod.rs:50-56), for callers that drive the
// C-ABI themselves (vf_terrain_create takes the bytes)
py::array_t<uint8_t> colormap_rgba8(const std::string &name)
{
    const uint8_t *lut = resolve_lut(name);
    py::array_t<uint8_t> a({ (py::ssize_t)256, (py::ssize_t)4 });
    std::memcpy(a.mutable_data(), lut, 1024);
    return a;
}

py::array_t<float> camera_look_at(std::tuple<float, float, float> eye, std::tuple<float, float, float> target,
                                  std::tuple<float, float, float> up)
{
    validate_vectors(v3(eye), v3(target), v3(up));
    return mat4_to_numpy(look_at_rh(v3(eye), v3(target), v3(up)));
}
py::array_t<float> camera_perspective(float fovy_deg, float aspect, float znear, float zfar, py::object clip_space)
{
    std::string clip = clip_space.is_none() ? "wgpu" : clip_space.cast<std::string>();
    validate_fovy(fovy_deg); validate_aspect(aspect); validate_near(znear); validate_far(zfar, znear);
    bool gl = clip_is_gl(clip);
    Mat4 p = gl ? perspective_rh_gl(to_radians(fovy_deg), aspect, znear, zfar) : perspective_wgpu(to_radians(fovy_deg), aspect, znear, zfar);
    return mat4_to_numpy(p);
}
py::array_t<float> camera_view_proj(std::tuple<float, float, float> eye, std::tuple<float, float, float> target,
                                    std::tuple<float, float, float> up, float fovy_deg, float aspect, float znear, float zfar,
                                    py::object clip_space)
{
    std::string clip = clip_space.is_none() ? "wgpu" : clip_space.cast<std::string>();
    validate_vectors(v3(eye), v3(target), v3(up));
    validate_fovy(fovy_deg); validate_aspect(aspect); validate_near(znear); validate_far(zfar, znear);
    bool gl = clip_is_gl(clip);
    Mat4 v = look_at_rh(v3(eye), v3(target), v3(up));
    Mat4 p = gl ? perspective_rh_gl(to_radians(fovy_deg), aspect, znear, zfar) : perspective_wgpu(to_radians(fovy_deg), aspect, znear, zfar);
    return mat4_to_numpy(mat_mul(p, v));
}

// enumerate_adapters / device_probe (src/lib.rs:744-845) reported from hipGetDeviceProperties
py::dict adapter_dict(const vf_device_info &di)
{
    py::dict d;
    d["name"] = std::string(di.name);
    d["backend"] = "HIP";
    d["device_type"] = "DiscreteGpu";
    d["vendor_id"] = 0x1002u;
    d["device_id"] = (uint32_t)di.pci_device_id;
    d["features"] = std::string("arch=") + di.arch + " wavefront=" + std::to_string(di.wavefront_size);
    d["limits"] = "compute_units=" + std::to_string(di.compute_units) + " total_mem_bytes=" + std::to_string(di.total_mem_bytes) +
                  " lds_bytes_per_cu=" + std::to_string(di.lds_bytes_per_cu);
    return d;
}
py::list enumerate_adapters()
{
    py::list out;
    int n = 0;
    if (vf_device_count(&n) != VF_OK) return out;
    for (int i = 0; i < n; ++i) {
        vf_device_info di;
        if (vf_device_query(i, &di) == VF_OK) out.append(adapter_dict(di));
    }
    return out;
}
py::dict device_probe(py::object backend)
{
    std::string b = backend.is_none() ? "AUTO" : backend.cast<std::string>();
    for (auto &ch : b) ch = (char)std::toupper((unsigned char)ch);
    py::dict d;
    d["backend_request"] = b;
    auto t0 = std::chrono::steady_clock::now();
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    if (b != "AUTO" && b != "HIP") {   // only one backend exists in this build
        d["status"] = "unsupported"; d["message"] = "No suitable GPU adapter"; d["millis"] = ms();
        return d;
    }
    vf_device_info di;
    int rc = vf_device_query(device_ordinal(), &di);
    if (rc != VF_OK) {
        d["status"] = "unsupported"; d["message"] = "No suitable GPU adapter"; d["millis"] = ms();
        return d;
    }
    py::dict a = adapter_dict(di);
    d["adapter_name"] = a["name"];
    for (const char *k : { "backend", "device_type", "vendor_id", "device_id", "features", "limits" }) d[k] = a[k];
    vf_ctx *c = nullptr;
    rc = vf_ctx_create(device_ordinal(), &c);
    if (rc != VF_OK) {
        d["status"] = "error"; d["message"] = std::string("request_device failed: ") + vf_last_error(); d["millis"] = ms();
        return d;
    }
    vf_ctx_destroy(c);
    d["status"] = "ok"; d["millis"] = ms();
    return d;
}

// testing hook: the PNG encoder behind render_png / render_triangle_png, without touching the GPU
py::bytes py_encode_png(py::array_t<uint8_t, py::array::c_style | py::array::forcecast> img)
{
    if (img.ndim() != 3 || img.shape(2) != 4) throw py::value_error("expected (H, W, 4) uint8");
    std::vector<uint8_t> png = vfh::encode_png_rgba8(img.data(), (uint32_t)img.shape(1), (uint32_t)img.shape(0));
    return py::bytes(reinterpret_cast<const char *>(png.data()), png.size());
}

template <class T>
py::class_<T> bind_terrain(py::module_ &m, const char *name)
{
    return py::class_<T>(m, name)
        .def(py::init<uint32_t, uint32_t, py::object, py::object>(), py::arg("width"), py::arg("height"), py::arg("grid") = 128,
             py::arg("colormap") = "viridis")
        .def("render_png", &T::render_png, py::arg("path"))
        .def("render_rgba", &T::render_rgba)
        .def("render_batch", &T::render_batch, py::arg("poses"), py::arg("paths") = py::none())
        .def("set_camera_look_at", &T::set_camera_look_at, py::arg("eye"), py::arg("target"), py::arg("up"), py::arg("fovy_deg"),
             py::arg("znear"), py::arg("zfar"))
        .def("debug_uniforms_f32", &T::debug_uniforms_f32)
        .def("debug_lut_format", &T::debug_lut_format)
        .def("debug_visibility", &T::debug_visibility)
        .def("set_shard", &T::set_shard, py::arg("rank"), py::arg("nranks"), py::arg("band_h") = 64)
        .def("set_shade_mode", &T::set_shade_mode, py::arg("mode"))
        .def("set_shade_precision", &T::set_shade_precision, py::arg("precision"))
        .def("enable_timing", &T::enable_timing, py::arg("on") = true)
        .def("last_timings", &T::last_timings);
}

} // namespace

PYBIND11_MODULE(_vulkan_forge, m)
{
    m.doc() = "MI355X-native drop-in for vulkan-forge's _vulkan_forge extension (terrain raster hot path)";
    py::class_<Renderer>(m, "Renderer")
        .def(py::init<uint32_t, uint32_t>(), py::arg("width"), py::arg("height"))
        .def("info", &Renderer::info)
        .def("render_triangle_rgba", &Renderer::render_triangle_rgba)
        .def("render_triangle_png", &Renderer::render_triangle_png, py::arg("path"))
        .def("add_terrain", &Renderer::add_terrain, py::arg("heightmap"), py::arg("spacing"), py::arg("exaggeration") = 1.0f,
             py::arg("colormap") = "viridis")
        .def("terrain_stats", &Renderer::terrain_stats)
        .def("set_height_range", &Renderer::set_height_range, py::arg("min"), py::arg("max"))
        .def("set_sun", &Renderer::set_sun, py::arg("elevation_deg"), py::arg("azimuth_deg"))
        .def("set_exposure", &Renderer::set_exposure, py::arg("exposure"))
        .def("normalize_terrain", &Renderer::normalize_terrain, py::arg("mode"), py::arg("range") = py::none(), py::arg("eps") = py::none())
        .def("upload_height_r32f", &Renderer::upload_height_r32f)
        .def("debug_read_height_patch", &Renderer::debug_read_height_patch, py::arg("x"), py::arg("y"), py::arg("w"), py::arg("h"))
        .def("read_full_height_texture", &Renderer::read_full_height_texture);
    bind_terrain<TerrainSpike>(m, "TerrainSpike");
    bind_terrain<Scene>(m, "Scene").def("set_height_from_r32f", &Scene::set_height_from_r32f, py::arg("height_r32f"));
    m.def("enumerate_adapters", &enumerate_adapters);
    m.def("device_probe", &device_probe, py::arg("backend") = py::none());
    m.def("grid_generate", &grid_generate, py::arg("nx"), py::arg("nz"), py::arg("spacing") = std::make_tuple(1.0f, 1.0f),
          py::arg("origin") = "center");
    m.def("colormap_supported", &colormap_supported);
    m.def("colormap_rgba8", &colormap_rgba8, py::arg("name"));
    m.def("_encode_png_rgba8", &py_encode_png, py::arg("rgba"));
    m.def("camera_look_at", &camera_look_at, py::arg("eye"), py::arg("target"), py::arg("up"));
    m.def("camera_perspective", &camera_perspective, py::arg("fovy_deg"), py::arg("aspect"), py::arg("znear"), py::arg("zfar"),
          py::arg("clip_space") = "wgpu");
    m.def("camera_view_proj", &camera_view_proj, py::arg("eye"), py::arg("target"), py::arg("up"), py::arg("fovy_deg"),
          py::arg("aspect"), py::arg("znear"), py::arg("zfar"), py::arg("clip_space") = "wgpu");
}
