// camera.hpp -- host-side camera math behind camera_look_at / camera_perspective / camera_view_proj and
// the uniform block of TerrainSpike / Scene.
//
// Mirrors src/camera.rs of the reference: right-handed, Y-up, -Z forward look-at; GL projection optionally
// multiplied by the GL->WGPU matrix *as the reference codes it* (src/camera.rs:14-21: the literal is consumed
// column by column by glam's from_cols_array, so z' = z/2 and w' = w + z/2 -- reproduced on purpose, the
// rendered picture depends on it).  glam 0.24.2 formulas (scalar f32): look_at_rh, perspective_rh_gl.
#pragma once
#include <array>
#include <cmath>
#include <stdexcept>
#include <string>

namespace vfh {

using Mat4 = std::array<float, 16>;   // column-major, like glam::Mat4::to_cols_array
struct Vec3 { float x, y, z; };

// exact error strings, src/camera.rs:24-30
constexpr const char *kErrFovy = "fovy_deg must be finite and in (0, 180)";
constexpr const char *kErrNear = "znear must be finite and > 0";
constexpr const char *kErrFar = "zfar must be finite and > znear";
constexpr const char *kErrAspect = "aspect must be finite and > 0";
constexpr const char *kErrVecFinite = "eye/target/up components must be finite";
constexpr const char *kErrUpColinear = "up vector must not be colinear with view direction";
constexpr const char *kErrClip = "clip_space must be 'wgpu' or 'gl'";

inline Vec3 sub(Vec3 a, Vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
inline Vec3 cross(Vec3 a, Vec3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
inline Vec3 mul(Vec3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline Vec3 normalize(Vec3 a) { return mul(a, 1.0f / std::sqrt(dot(a, a))); }
inline Vec3 normalize_or_zero(Vec3 a)
{
    float rcp = 1.0f / std::sqrt(dot(a, a));
    return (std::isfinite(rcp) && rcp > 0.0f) ? mul(a, rcp) : Vec3{ 0.f, 0.f, 0.f };
}
inline bool finite(Vec3 a) { return std::isfinite(a.x) && std::isfinite(a.y) && std::isfinite(a.z); }

inline Mat4 look_at_rh(Vec3 eye, Vec3 center, Vec3 up)
{
    Vec3 f = normalize(sub(center, eye));
    Vec3 s = normalize(cross(f, up));
    Vec3 u = cross(s, f);
    return { s.x, u.x, -f.x, 0.f, s.y, u.y, -f.y, 0.f, s.z, u.z, -f.z, 0.f, -dot(eye, s), -dot(eye, u), dot(eye, f), 1.f };
}

inline Mat4 perspective_rh_gl(float fovy_rad, float aspect, float zn, float zf)
{
    float inv_length = 1.0f / (zn - zf);
    float f = 1.0f / std::tan(0.5f * fovy_rad);
    float a = f / aspect;
    float b = (zn + zf) * inv_length;
    float c = (2.0f * zn * zf) * inv_length;
    return { a, 0.f, 0.f, 0.f, 0.f, f, 0.f, 0.f, 0.f, 0.f, b, -1.f, 0.f, 0.f, c, 0.f };
}

inline Mat4 mat_mul(const Mat4 &A, const Mat4 &B)
{
    Mat4 r;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r[4 * j + i] = ((A[i] * B[4 * j] + A[4 + i] * B[4 * j + 1]) + A[8 + i] * B[4 * j + 2]) + A[12 + i] * B[4 * j + 3];
    return r;
}

// src/camera.rs:14-21, read as glam reads it (column-major)
inline Mat4 gl_to_wgpu() { return { 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.5f, 0.5f, 0.f, 0.f, 0.f, 1.f }; }

// src/camera.rs:218-221
inline Mat4 perspective_wgpu(float fovy_rad, float aspect, float zn, float zf)
{
    return mat_mul(gl_to_wgpu(), perspective_rh_gl(fovy_rad, aspect, zn, zf));
}

inline float to_radians(float deg) { return deg * (3.14159265358979323846f / 180.0f); }

// validators (src/camera.rs:33-91); all raise RuntimeError on the Python side
inline void validate_vectors(Vec3 eye, Vec3 target, Vec3 up)
{
    if (!finite(eye) || !finite(target) || !finite(up)) throw std::runtime_error(kErrVecFinite);
    Vec3 c = cross(normalize_or_zero(sub(target, eye)), normalize_or_zero(up));
    if (dot(c, c) < 1e-6f) throw std::runtime_error(kErrUpColinear);
}
inline void validate_fovy(float f) { if (!std::isfinite(f) || f <= 0.0f || f >= 180.0f) throw std::runtime_error(kErrFovy); }
inline void validate_near(float n) { if (!std::isfinite(n) || n <= 0.0f) throw std::runtime_error(kErrNear); }
inline void validate_far(float f, float n) { if (!std::isfinite(f) || f <= n) throw std::runtime_error(kErrFar); }
inline void validate_aspect(float a) { if (!std::isfinite(a) || a <= 0.0f) throw std::runtime_error(kErrAspect); }
inline bool clip_is_gl(const std::string &clip)
{
    if (clip == "gl") return true;
    if (clip == "wgpu") return false;
    throw std::runtime_error(kErrClip);
}
// src/camera.rs:224-240
inline void validate_camera_params(Vec3 eye, Vec3 target, Vec3 up, float fovy_deg, float zn, float zf)
{
    validate_vectors(eye, target, up);
    validate_fovy(fovy_deg);
    validate_near(zn);
    validate_far(zf, zn);
}

// ---- uniform block: TerrainUniforms (src/terrain/mod.rs:114-175) from Globals (:178-215) --------
struct Globals {
    Vec3 sun_dir = normalize({ 0.5f, 0.8f, 0.6f });
    float exposure = 1.0f, spacing = 1.0f, h_min = -0.5f, h_max = 0.5f, exaggeration = 1.0f;
};
using Uniforms = std::array<float, 44>;

inline Uniforms to_uniforms(const Globals &g, const Mat4 &view, const Mat4 &proj)
{
    Uniforms u{};
    for (int k = 0; k < 16; ++k) { u[k] = view[k]; u[16 + k] = proj[k]; }
    u[32] = g.sun_dir.x; u[33] = g.sun_dir.y; u[34] = g.sun_dir.z; u[35] = g.exposure;
    u[36] = g.spacing; u[37] = g.h_max - g.h_min; u[38] = g.exaggeration; u[39] = 0.0f;
    return u;
}

} // namespace vfh
