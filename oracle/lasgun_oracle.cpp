// oracle/lasgun_oracle.cpp -- TEST INFRASTRUCTURE. NOT PART OF THE PRODUCT.
//
// CPU restatement (C++17, scalar f64, no FMA: build with -ffp-contract=off) of
// nfrasser/lasgun's per-pixel ray-trace path, written to follow the reference
// source file by file so that every function can be checked against the Rust it
// restates.  Citations are file:line under /root/reference.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// the library built from this file.  The product (lasgun_amd/) never does.
//
// Parity status: PINNED ONLY BY THE REFERENCE'S 17 INLINE KATs (src/shape/sphere.rs:137-173,
// src/shape/cuboid.rs:137-246, src/shape/triangle.rs:411-454, src/interaction/surface.rs:194-200).
// The reference is Rust and there is no Rust toolchain in this pipeline, so the
// original cannot be executed; BVH build, transforms, camera, shading, lights,
// background and quantisation are restated from source and are "parity unpinned"
// beyond those KATs.  Third-party arithmetic restated from published sources:
// cgmath ^0.17 (vector/matrix op order), obj ^0.10 (OBJ text -> index tuples),
// partition ^0.1 (in-place two-pointer partition).
//
// Transcendentals: default mode calls glibc libm exactly where the Rust calls
// f64::atan2/acos/sin/cos (sphere.rs:99-114); orc_set_trig_mode(1) switches to the
// portable algorithm of oracle/lg_trig.h (the algorithm the GPU kernels use), so
// GPU radiance can be compared bit-for-bit.

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include <atomic>
#include <cctype>
#include <mutex>

#include "lg_trig.h"

namespace orc {

static const double PI = 3.14159265358979323846264338327950288;       // std::f64::consts::PI
static const double FRAC_1_PI = 0.318309886183790671537767526745028724; // std::f64::consts::FRAC_1_PI
static const double F64_MAX = std::numeric_limits<double>::max();
static const double F64_INF = std::numeric_limits<double>::infinity();

static std::atomic<int> g_trig_mode{0}; // 0 = libm (what Rust does), 1 = portable

static inline double t_atan2(double y, double x) { return g_trig_mode.load(std::memory_order_relaxed) ? orc_atan2(y, x) : std::atan2(y, x); }
static inline double t_acos(double x) { return g_trig_mode.load(std::memory_order_relaxed) ? orc_acos(x) : std::acos(x); }
static inline double t_sin(double x) { return g_trig_mode.load(std::memory_order_relaxed) ? orc_sin(x) : std::sin(x); }
static inline double t_cos(double x) { return g_trig_mode.load(std::memory_order_relaxed) ? orc_cos(x) : std::cos(x); }
// sin and cos of ONE argument in one expression: a native build of the reference calls glibc's sincos() there (LLVM turns an
// fsin / fcos pair on the same operand into the sincos libcall on *-linux-gnu, as GCC does for the rotation matrices
// below), and sincos(x) is not always bit-identical to (sin(x), cos(x)) -- 71.69565646291534 degrees is an example
static inline void t_sincos(double x, double &s, double &c) {
    if (g_trig_mode.load(std::memory_order_relaxed)) { s = orc_sin(x); c = orc_cos(x); }
    else ::sincos(x, &s, &c);
}

// ---- Rust float semantics -----------------------------
static inline double fmin_(double a, double b) { return std::fmin(a, b); } // f64::min: NaN-ignoring
static inline double fmax_(double a, double b) { return std::fmax(a, b); }
static inline double signum(double x) { // f64::signum: 1.0 for +0.0, -1.0 for -0.0, NaN for NaN
    if (x != x) return x;
    return std::signbit(x) ? -1.0 : 1.0;
}
static inline uint32_t as_u32(double v) { // Rust `as u32`: saturating, NaN -> 0
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}
static inline uint8_t as_u8(double v) {
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}
// the local `min`/`max` of space/bounds.rs:171-178 and space/transform.rs:309-310
static inline double bmin(double a, double b) { return a < b ? a : b; }
static inline double bmax(double a, double b) { return a < b ? b : a; }

// ---- cgmath ^0.17 restated -----------------------------
struct V3 {
    double x, y, z;
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
static inline V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
static inline V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
static inline V3 operator/(V3 a, double s) { return V3{a.x / s, a.y / s, a.z / s}; }
static inline V3 mul_ew(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline V3 div_ew(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
static inline double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline double magnitude2(V3 a) { return dot(a, a); }
static inline double magnitude(V3 a) { return std::sqrt(dot(a, a)); }
static inline V3 normalize(V3 a) { return a * (1.0 / magnitude(a)); }
static inline bool eq(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
static inline bool ne(V3 a, V3 b) { return !eq(a, b); }
static const V3 ZERO3 = {0.0, 0.0, 0.0};

struct V4 { double x, y, z, w; };
static inline V4 operator*(V4 a, double s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }
static inline V4 operator+(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }

// Column-major 4x4: c[i] is column i (cgmath Matrix4 {x,y,z,w})
struct M4 {
    V4 c[4];
    double at(int col, int row) const {
        const V4 &v = c[col];
        return row == 0 ? v.x : (row == 1 ? v.y : (row == 2 ? v.z : v.w));
    }
};
static inline M4 m4_identity() {
    return M4{{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}};
}
static inline V4 m4_mul_v4(const M4 &m, V4 v) { // Matrix4 * Vector4
    return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z + m.c[3] * v.w;
}
static inline M4 m4_mul(const M4 &l, const M4 &r) { // Matrix4 * Matrix4 (column combination)
    M4 o;
    for (int j = 0; j < 4; ++j)
        o.c[j] = l.c[0] * r.c[j].x + l.c[1] * r.c[j].y + l.c[2] * r.c[j].z + l.c[3] * r.c[j].w;
    return o;
}
static inline M4 m4_transpose(const M4 &m) {
    M4 o;
    o.c[0] = V4{m.c[0].x, m.c[1].x, m.c[2].x, m.c[3].x};
    o.c[1] = V4{m.c[0].y, m.c[1].y, m.c[2].y, m.c[3].y};
    o.c[2] = V4{m.c[0].z, m.c[1].z, m.c[2].z, m.c[3].z};
    o.c[3] = V4{m.c[0].w, m.c[1].w, m.c[2].w, m.c[3].w};
    return o;
}
static inline V3 m4_transform_vector(const M4 &m, V3 v) { // (M * v.extend(0)).truncate()
    V4 h = m4_mul_v4(m, V4{v.x, v.y, v.z, 0.0});
    return V3{h.x, h.y, h.z};
}
static inline V3 m4_transform_point(const M4 &m, V3 p) { // Point3::from_homogeneous(M * p.to_homogeneous())
    V4 h = m4_mul_v4(m, V4{p.x, p.y, p.z, 1.0});
    double s = 1.0 / h.w;
    return V3{h.x * s, h.y * s, h.z * s};
}
static inline M4 m4_from_translation(V3 d) {
    M4 m = m4_identity();
    m.c[3] = V4{d.x, d.y, d.z, 1.0};
    return m;
}
static inline M4 m4_from_scale(double x, double y, double z) {
    return M4{{{x, 0, 0, 0}, {0, y, 0, 0}, {0, 0, z, 0}, {0, 0, 0, 1}}};
}
static inline double deg_to_rad(double deg) { return deg * (PI / 180.0); } // Rad::from(Deg)
static inline M4 m4_from_angle_x(double deg) {
    double t = deg_to_rad(deg), s, c;
    ::sincos(t, &s, &c); // Rad::sin_cos (cgmath): one glibc sincos() call in a native build, which is what g++ -O2 made of a sin / cos pair anyway
    return M4{{{1, 0, 0, 0}, {0, c, s, 0}, {0, -s, c, 0}, {0, 0, 0, 1}}};
}
static inline M4 m4_from_angle_y(double deg) {
    double t = deg_to_rad(deg), s, c;
    ::sincos(t, &s, &c); // Rad::sin_cos (cgmath): one glibc sincos() call in a native build, which is what g++ -O2 made of a sin / cos pair anyway
    return M4{{{c, 0, -s, 0}, {0, 1, 0, 0}, {s, 0, c, 0}, {0, 0, 0, 1}}};
}
static inline M4 m4_from_angle_z(double deg) {
    double t = deg_to_rad(deg), s, c;
    ::sincos(t, &s, &c); // Rad::sin_cos (cgmath): one glibc sincos() call in a native build, which is what g++ -O2 made of a sin / cos pair anyway
    return M4{{{c, s, 0, 0}, {-s, c, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}};
}
static inline M4 m4_from_axis_angle(V3 a, double deg) {
    double t = deg_to_rad(deg), s, c;
    ::sincos(t, &s, &c); // Rad::sin_cos (cgmath): one glibc sincos() call in a native build, which is what g++ -O2 made of a sin / cos pair anyway
    double k = 1.0 - c;
    return M4{{{k * a.x * a.x + c, k * a.x * a.y + s * a.z, k * a.x * a.z - s * a.y, 0.0},
               {k * a.x * a.y - s * a.z, k * a.y * a.y + c, k * a.y * a.z + s * a.x, 0.0},
               {k * a.x * a.z + s * a.y, k * a.y * a.z - s * a.x, k * a.z * a.z + c, 0.0},
               {0.0, 0.0, 0.0, 1.0}}};
}

// ---- src/space/ray.rs -------------------------------------------------------
struct Ray {
    V3 origin, d, dinv;
};
static inline Ray ray_new(V3 origin, V3 d) { // ray.rs:28-33
    return Ray{origin, d, V3{1.0 / d.x, 1.0 / d.y, 1.0 / d.z}};
}

// ---- src/space/bounds.rs ----------------------------------------------------
struct Bounds {
    V3 min, max;
};
static inline Bounds bounds_new(V3 p0, V3 p1) { // bounds.rs:36-42
    return Bounds{V3{bmin(p0.x, p1.x), bmin(p0.y, p1.y), bmin(p0.z, p1.z)},
                  V3{bmax(p0.x, p1.x), bmax(p0.y, p1.y), bmax(p0.z, p1.z)}};
}
static inline Bounds bounds_union(const Bounds &a, const Bounds &b) { // bounds.rs:55-60
    return Bounds{V3{bmin(a.min.x, b.min.x), bmin(a.min.y, b.min.y), bmin(a.min.z, b.min.z)},
                  V3{bmax(a.max.x, b.max.x), bmax(a.max.y, b.max.y), bmax(a.max.z, b.max.z)}};
}
static inline Bounds bounds_point_union(const Bounds &a, V3 p) { // bounds.rs:64-69
    return Bounds{V3{bmin(a.min.x, p.x), bmin(a.min.y, p.y), bmin(a.min.z, p.z)},
                  V3{bmax(a.max.x, p.x), bmax(a.max.y, p.y), bmax(a.max.z, p.z)}};
}
static inline Bounds bounds_none() { // bounds.rs:152-157
    return Bounds{V3{F64_MAX, F64_MAX, F64_MAX}, V3{-F64_MAX, -F64_MAX, -F64_MAX}};
}
static inline double bounds_surface_area(const Bounds &b) { // bounds.rs:110-114
    V3 d = b.max - b.min;
    double half = d.x * d.y + d.x * d.z + d.y * d.z;
    return half + half;
}
static inline int bounds_maximum_extent(const Bounds &b) { // bounds.rs:125-130 (bug kept: never 0)
    V3 d = b.max - b.min;
    if (d.x > d.y && d.z > d.z) return 0;
    else if (d.y > d.z) return 1;
    else return 2;
}
static inline V3 bounds_offset(const Bounds &b, V3 p) { // bounds.rs:133-139
    V3 o = p - b.min;
    if (b.max.x > b.min.x) o.x /= b.max.x - b.min.x;
    if (b.max.y > b.min.y) o.y /= b.max.y - b.min.y;
    if (b.max.z > b.min.z) o.z /= b.max.z - b.min.z;
    return o;
}

// ---- src/space/normal.rs ----------------------------------------------------
static inline V3 face_forward(V3 n, V3 v) { return dot(n, v) < 0.0 ? -n : n; } // normal.rs:37-40

// ---- src/space/mod.rs -------------------------------------------------------
static inline V3 vabs(V3 v) { return V3{std::fabs(v.x), std::fabs(v.y), std::fabs(v.z)}; }
static inline double lerp(double t, double p0, double p1) { return p0 * (1.0 - t) + p1 * t; } // mod.rs:28-30
static inline int max_dimension(V3 v) { // mod.rs:33-36
    if (v.x > v.y) { return v.x > v.z ? 0 : 2; }
    else { return v.y > v.z ? 1 : 2; }
}
static inline void coordinate_system(V3 v1, V3 &v2, V3 &v3o) { // mod.rs:39-47
    if (std::fabs(v1.x) > std::fabs(v1.y))
        v2 = V3{-v1.z, 0.0, v1.x} / std::sqrt(v1.x * v1.x + v1.z * v1.z);
    else
        v2 = V3{0.0, v1.z, -v1.y} / std::sqrt(v1.y * v1.y + v1.z * v1.z);
    v3o = cross(v1, v2);
}

// ---- src/core/math.rs -------------------------------------------------------
static inline int quad_roots(double a, double b, double c, double roots[2]) { // math.rs:7-30
    const double NaN = std::numeric_limits<double>::quiet_NaN();
    if (a == 0.0) {
        if (b == 0.0) { roots[0] = NaN; roots[1] = NaN; return 0; }
        roots[0] = -c / b; roots[1] = NaN; return 1;
    }
    double d = b * b - 4.0 * a * c;
    if (d < 0.0) { roots[0] = NaN; roots[1] = NaN; return 0; }
    double q = -(b + signum(b) * std::sqrt(d)) / 2.0;
    double q_over_a = q / a;
    roots[0] = q_over_a;
    roots[1] = (q == 0.0) ? q_over_a : c / q;
    return 2;
}

// ---- src/material/mod.rs ----------------------------------------------------
enum MatKind { MATTE = 0, PLASTIC = 1, METAL = 2, GLASS = 3, MIRROR = 4 };
struct Material { // C-ABI POD: kind + 10 doubles
    int32_t kind;
    double p[10];
};
static inline Material material_matte(const double kd[3], double sigma) { // matte.rs:14-16
    Material m{}; m.kind = MATTE;
    m.p[0] = kd[0]; m.p[1] = kd[1]; m.p[2] = kd[2];
    m.p[3] = fmin_(fmax_(sigma, 0.0), 90.0);
    return m;
}
static inline Material material_default() { // material/mod.rs:15-17
    const double kd[3] = {0.5, 0.5, 0.5};
    return material_matte(kd, 0.0);
}

// ---- src/interaction/surface.rs --------------------------------------------
struct Shading { V3 dpdu, dpdv; };
struct RayIntersection {
    double t;
    double uv[2];
    Shading geometry, surface;
    Material material;
    bool has_n;
    V3 n;
};
static inline RayIntersection isect_new(double t, double u, double v, V3 dpdu, V3 dpdv) { // surface.rs:57-62
    RayIntersection r;
    r.t = t; r.uv[0] = u; r.uv[1] = v;
    r.geometry = Shading{dpdu, dpdv};
    r.surface = r.geometry;
    r.material = material_default();
    r.has_n = false; r.n = ZERO3;
    return r;
}
static inline RayIntersection isect_default() { return isect_new(F64_INF, 0.0, 0.0, ZERO3, ZERO3); } // surface.rs:65-72
static inline void isect_swap_backface(RayIntersection &i) { // surface.rs:88-99
    V3 a = i.geometry.dpdu, b = i.geometry.dpdv;
    i.geometry.dpdu = b; i.geometry.dpdv = a;
    a = i.surface.dpdu; b = i.surface.dpdv;
    i.surface.dpdu = b; i.surface.dpdv = a;
    if (i.has_n) i.n = -i.n;
}
static inline V3 isect_ng(const RayIntersection &i) { return normalize(cross(i.geometry.dpdu, i.geometry.dpdv)); } // :107-109
static inline V3 isect_ns(const RayIntersection &i) { // :112-118
    if (i.has_n) return normalize(i.n);
    return normalize(cross(i.surface.dpdu, i.surface.dpdv));
}
struct SurfaceInteraction {
    V3 p, p_err, wo, ng, ns;
    Shading geometry, surface;
};
static inline SurfaceInteraction si_from(const Ray &ray, const RayIntersection &isect) { // surface.rs:158-183
    SurfaceInteraction s;
    s.wo = -normalize(ray.d);
    s.ng = face_forward(isect_ng(isect), s.wo);
    s.ns = isect_ns(isect);
    double err = 2.220446049250313e-16 * 65536.0; // N::epsilon() * 2.powi(16)
    s.p = ray.origin + ray.d * isect.t;
    s.p_err = s.ng * err;
    s.geometry = Shading{normalize(isect.geometry.dpdu), normalize(isect.geometry.dpdv)};
    s.surface = Shading{normalize(isect.surface.dpdu), normalize(isect.surface.dpdv)};
    return s;
}

// ---- src/space/transform.rs -------------------------------------------------
struct Transform {
    M4 m, minv;
};
static inline Transform tr_identity() { return Transform{m4_identity(), m4_identity()}; }
static inline void tr_concat_self(Transform &self, const Transform &other) { // transform.rs:191-197
    M4 m = m4_mul(other.m, self.m);
    M4 minv = m4_mul(self.minv, other.minv);
    self.m = m; self.minv = minv;
}
static inline V3 tr_transform_normal(const Transform &t, V3 n) { // transform.rs:202-209
    const M4 &mi = t.minv;
    return V3{mi.at(0, 0) * n.x + mi.at(0, 1) * n.y + mi.at(0, 2) * n.z,
              mi.at(1, 0) * n.x + mi.at(1, 1) * n.y + mi.at(1, 2) * n.z,
              mi.at(2, 0) * n.x + mi.at(2, 1) * n.y + mi.at(2, 2) * n.z};
}
static inline V3 tr_inverse_transform_normal(const Transform &t, V3 n) { // transform.rs:267-274
    const M4 &m = t.m;
    return V3{m.at(0, 0) * n.x + m.at(0, 1) * n.y + m.at(0, 2) * n.z,
              m.at(1, 0) * n.x + m.at(1, 1) * n.y + m.at(1, 2) * n.z,
              m.at(2, 0) * n.x + m.at(2, 1) * n.y + m.at(2, 2) * n.z};
}
static inline Bounds tr_transform_bounds(const Transform &t, const Bounds &b) { // transform.rs:219-240
    V4 xa = t.m.c[0] * b.min.x, xb = t.m.c[0] * b.max.x;
    V4 ya = t.m.c[1] * b.min.y, yb = t.m.c[1] * b.max.y;
    V4 za = t.m.c[2] * b.min.z, zb = t.m.c[2] * b.max.z;
    V3 mn = V3{bmin(xa.x, xb.x), bmin(xa.y, xb.y), bmin(xa.z, xb.z)} +
            V3{bmin(ya.x, yb.x), bmin(ya.y, yb.y), bmin(ya.z, yb.z)} +
            V3{bmin(za.x, zb.x), bmin(za.y, zb.y), bmin(za.z, zb.z)};
    V3 mx = V3{bmax(xa.x, xb.x), bmax(xa.y, xb.y), bmax(xa.z, xb.z)} +
            V3{bmax(ya.x, yb.x), bmax(ya.y, yb.y), bmax(ya.z, yb.z)} +
            V3{bmax(za.x, zb.x), bmax(za.y, zb.y), bmax(za.z, zb.z)};
    const V4 &w = t.m.c[3];
    V3 pmin{mn.x + w.x, mn.y + w.y, mn.z + w.z};
    V3 pmax{mx.x + w.x, mx.y + w.y, mx.z + w.z};
    return bounds_new(pmin, pmax);
}
static inline Ray tr_inverse_transform_ray(const Transform &t, const Ray &r) { // transform.rs:279-283
    V3 o = m4_transform_point(t.minv, r.origin);
    V3 d = m4_transform_vector(t.minv, r.d);
    return ray_new(o, d);
}
static inline RayIntersection tr_transform_isect(const Transform &t, const RayIntersection &i) { // transform.rs:243-264
    V3 dpdu = m4_transform_vector(t.m, i.geometry.dpdu);
    V3 dpdv = m4_transform_vector(t.m, i.geometry.dpdv);
    RayIntersection o = isect_new(i.t, i.uv[0], i.uv[1], dpdu, dpdv);
    o.material = i.material;
    if (ne(i.geometry.dpdu, i.surface.dpdu) || ne(i.geometry.dpdv, i.surface.dpdv)) {
        o.surface.dpdu = m4_transform_vector(t.m, i.surface.dpdu);
        o.surface.dpdv = m4_transform_vector(t.m, i.surface.dpdv);
    }
    if (i.has_n) { o.has_n = true; o.n = tr_transform_normal(t, i.n); }
    return o;
}
static inline RayIntersection tr_inverse_transform_isect(const Transform &t, const RayIntersection &i) { // transform.rs:286-305
    V3 dpdu = m4_transform_vector(t.minv, i.geometry.dpdu);
    V3 dpdv = m4_transform_vector(t.minv, i.geometry.dpdv);
    RayIntersection o = isect_new(i.t, i.uv[0], i.uv[1], dpdu, dpdv); // material NOT carried (as in the reference)
    if (ne(i.geometry.dpdu, i.surface.dpdu) || ne(i.geometry.dpdv, i.surface.dpdv)) {
        o.surface.dpdu = m4_transform_vector(t.minv, i.surface.dpdu);
        o.surface.dpdv = m4_transform_vector(t.minv, i.surface.dpdv);
    }
    if (i.has_n) { o.has_n = true; o.n = tr_inverse_transform_normal(t, i.n); }
    return o;
}

// ---- instrumentation (defines the roofline's algorithmic byte count) -------
struct Stats {
    uint64_t primary_rays = 0, shadow_rays = 0, secondary_rays = 0;
    uint64_t nodes_tested = 0, spheres_tested = 0, cuboids_tested = 0, triangles_tested = 0;
    uint64_t accel_entries = 0, hits = 0;
    void add(const Stats &o) {
        primary_rays += o.primary_rays; shadow_rays += o.shadow_rays; secondary_rays += o.secondary_rays;
        nodes_tested += o.nodes_tested; spheres_tested += o.spheres_tested; cuboids_tested += o.cuboids_tested;
        triangles_tested += o.triangles_tested; accel_entries += o.accel_entries; hits += o.hits;
    }
};
static thread_local Stats tl_stats;

// ---- src/primitive/mod.rs ---------------------------------------------------
struct Primitive {
    virtual ~Primitive() {}
    virtual Bounds bound() const = 0;
    virtual const Primitive *intersect(const Ray &ray, RayIntersection &isect) const = 0;
    virtual bool material(Material &out) const { (void)out; return false; }
};

// ---- src/shape/cuboid.rs ----------------------------------------------------
static const V3 CUBE_DIFF[3][2] = { // cuboid.rs:126-130
    {{0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}},
    {{0.0, 0.0, 1.0}, {1.0, 0.0, 0.0}},
    {{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}}};

static inline bool bounds_intersect(const Bounds &b, const Ray &ray, RayIntersection &isect) { // cuboid.rs:55-102
    double tnear = -F64_INF, tfar = F64_INF;
    V3 near0 = CUBE_DIFF[0][0], near1 = CUBE_DIFF[0][1];
    V3 far0 = CUBE_DIFF[0][0], far1 = CUBE_DIFF[0][1];
    for (int i = 0; i < 3; ++i) {
        V3 dpa = CUBE_DIFF[i][0], dpb = CUBE_DIFF[i][1];
        double t1 = (b.min[i] - ray.origin[i]) * ray.dinv[i];
        double t2 = (b.max[i] - ray.origin[i]) * ray.dinv[i];
        double tmin, tmax; V3 dp0, dp1;
        if (t1 < t2) { tmin = t1; tmax = t2; dp0 = dpb; dp1 = dpa; }
        else { tmin = t2; tmax = t1; dp0 = dpa; dp1 = dpb; }
        if (tmin > tnear) { near0 = dp0; near1 = dp1; }
        if (tmax < tfar) { far0 = dp1; far1 = dp0; }
        tnear = fmax_(tnear, tmin);
        tfar = fmin_(tfar, tmax);
    }
    if (tnear > tfar || tfar <= 0.0) return false;
    double t; V3 d0, d1;
    if (tnear <= 0.0) { t = tfar; d0 = far0; d1 = far1; }
    else { t = tnear; d0 = near0; d1 = near1; }
    if (t >= isect.t) return false;
    isect = isect_new(t, 0.0, 0.0, d0, d1);
    isect.has_n = true;
    isect.n = face_forward(cross(d0, d1), -ray.d);
    return true;
}
static inline bool bounds_intersects(const Bounds &b, const Ray &ray) { // cuboid.rs:104-121
    double tnear = -F64_INF, tfar = F64_INF;
    for (int i = 0; i < 3; ++i) {
        double t1 = (b.min[i] - ray.origin[i]) * ray.dinv[i];
        double t2 = (b.max[i] - ray.origin[i]) * ray.dinv[i];
        double tmin = fmin_(t1, t2), tmax = fmax_(t1, t2);
        tnear = fmax_(tnear, tmin);
        tfar = fmin_(tfar, tmax);
    }
    return tnear <= tfar && tfar > 0.0;
}
struct Cuboid : Primitive {
    Bounds bounds;
    Material mat;
    Bounds bound() const override { return bounds; }
    const Primitive *intersect(const Ray &ray, RayIntersection &isect) const override { // cuboid.rs:38-44
        tl_stats.cuboids_tested++;
        return bounds_intersect(bounds, ray, isect) ? this : nullptr;
    }
    bool material(Material &out) const override { out = mat; return true; }
};

// ---- src/shape/sphere.rs ----------------------------------------------------
struct Sphere : Primitive {
    V3 origin;
    double radius;
    Material mat;
    Bounds bound() const override { // sphere.rs:73-77
        V3 r{radius, radius, radius};
        return bounds_new(origin - r, origin + r);
    }
    void intersect_t(const Ray &ray, double &t, bool &inside) const { // sphere.rs:30-69
        V3 d = ray.d;
        V3 l = ray.origin - origin;
        double a = dot(d, d);
        double b = 2.0 * dot(d, l);
        double c = dot(l, l) - radius * radius;
        double roots[2];
        int n = quad_roots(a, b, c, roots);
        if (n == 2) {
            double t0 = fmin_(roots[0], roots[1]), t1 = fmax_(roots[0], roots[1]);
            if (t0 < 0.0) { t = t1; inside = true; } else { t = t0; inside = false; }
        } else if (n == 1) { t = roots[0]; inside = false; }
        else { t = -F64_INF; inside = false; }
    }
    const Primitive *intersect(const Ray &ray, RayIntersection &isect) const override { // sphere.rs:79-123
        tl_stats.spheres_tested++;
        double t; bool inside;
        intersect_t(ray, t, inside);
        if (t < 0.0) return nullptr;
        if (t >= isect.t) return nullptr;
        V3 p = ray.origin + ray.d * t - origin;
        if (p.x == 0.0 && p.y == 0.0) p.x = 1e-5 * radius;
        double phi = t_atan2(p.y, p.x);
        if (phi < 0.0) phi += 2.0 * PI;
        double theta = t_acos(fmin_(fmax_(p.z / radius, -1.0), 1.0));
        V3 dpdu{-2.0 * PI * p.y, 2.0 * PI * p.x, 0.0};
        double sin_phi, cos_phi;
        t_sincos(phi, sin_phi, cos_phi); // phi.cos(), phi.sin() (sphere.rs:107-108)
        V3 dpdv = PI * V3{p.z * cos_phi, p.z * sin_phi, -radius * t_sin(theta)};
        if (inside) isect = isect_new(t, 0.0, 0.0, dpdu, dpdv);
        else isect = isect_new(t, 0.0, 0.0, dpdv, dpdu);
        return this;
    }
    bool material(Material &out) const override { out = mat; return true; }
};

// ---- OBJ text -> index tuples (third-party `obj ^0.10`, restated) -----------
struct Obj {
    std::vector<float> position; // 3 per vertex (f32, widened on every access: triangle.rs:40-43)
    std::vector<float> texture;  // 2 per vt
    std::vector<float> normal;   // 3 per vn
    // one entry per `f` line, in file order == objects -> groups -> polys order
    // (triangle.rs:315-371); only the first three index tuples are ever read.
    struct Tuple { uint32_t v; int32_t t, n; }; // t/n = -1 when absent
    std::vector<Tuple> tri; // 3 per face
};
static bool parse_index(const char *s, const char *e, long count, long &out) {
    if (s == e) return false;
    char *endp = nullptr;
    std::string tmp(s, e);
    long v = std::strtol(tmp.c_str(), &endp, 10);
    if (*endp != 0) return false;
    if (v < 0) v = count + v; else v = v - 1;
    if (v < 0 || v >= count) return false;
    out = v;
    return true;
}
static bool parse_f32(const std::string &w, float &out) {
    if (w.empty()) return false;
    char *endp = nullptr;
    out = std::strtof(w.c_str(), &endp);
    return *endp == 0;
}
static int parse_obj_text(const char *text, size_t len, Obj &obj, std::string &err) {
    size_t pos = 0; int lineno = 0;
    bool group_open = false; size_t group_polys = 0; // to reject what would panic in TriangleIterator
    while (pos <= len) {
        size_t eol = pos;
        while (eol < len && text[eol] != '\n') ++eol;
        std::string line(text + pos, text + eol);
        pos = eol + 1; ++lineno;
        std::vector<std::string> words;
        {
            size_t i = 0;
            while (i < line.size()) {
                while (i < line.size() && std::isspace((unsigned char)line[i])) ++i;
                size_t j = i;
                while (j < line.size() && !std::isspace((unsigned char)line[j])) ++j;
                if (j > i) words.emplace_back(line.substr(i, j - i));
                i = j;
            }
        }
        if (words.empty()) { if (eol >= len) break; continue; }
        const std::string &cmd = words[0];
        if (cmd == "v" || cmd == "vn") {
            float f[3];
            if (words.size() < 4 || !parse_f32(words[1], f[0]) || !parse_f32(words[2], f[1]) || !parse_f32(words[3], f[2])) {
                err = "obj: bad vertex on line " + std::to_string(lineno); return 1;
            }
            auto &dst = (cmd == "v") ? obj.position : obj.normal;
            dst.push_back(f[0]); dst.push_back(f[1]); dst.push_back(f[2]);
        } else if (cmd == "vt") {
            float f[2];
            if (words.size() < 3 || !parse_f32(words[1], f[0]) || !parse_f32(words[2], f[1])) {
                err = "obj: bad vt on line " + std::to_string(lineno); return 1;
            }
            obj.texture.push_back(f[0]); obj.texture.push_back(f[1]);
        } else if (cmd == "f") {
            if (words.size() < 4) { err = "obj: face with <3 vertices on line " + std::to_string(lineno); return 1; }
            for (int k = 1; k <= 3; ++k) {
                const std::string &w = words[k];
                size_t s1 = w.find('/');
                size_t s2 = (s1 == std::string::npos) ? std::string::npos : w.find('/', s1 + 1);
                const char *b = w.c_str();
                Obj::Tuple tp{0, -1, -1};
                long idx;
                const char *ve = (s1 == std::string::npos) ? b + w.size() : b + s1;
                if (!parse_index(b, ve, (long)obj.position.size() / 3, idx)) { err = "obj: bad face index on line " + std::to_string(lineno); return 1; }
                tp.v = (uint32_t)idx;
                if (s1 != std::string::npos) {
                    const char *ts = b + s1 + 1;
                    const char *te = (s2 == std::string::npos) ? b + w.size() : b + s2;
                    if (te > ts) {
                        if (!parse_index(ts, te, (long)obj.texture.size() / 2, idx)) { err = "obj: bad vt index on line " + std::to_string(lineno); return 1; }
                        tp.t = (int32_t)idx;
                    }
                    if (s2 != std::string::npos) {
                        const char *ns = b + s2 + 1, *ne_ = b + w.size();
                        if (ne_ > ns) {
                            if (!parse_index(ns, ne_, (long)obj.normal.size() / 3, idx)) { err = "obj: bad vn index on line " + std::to_string(lineno); return 1; }
                            tp.n = (int32_t)idx;
                        }
                    }
                }
                obj.tri.push_back(tp);
            }
            group_open = true; group_polys++;
        } else if (cmd == "g" || cmd == "o") {
            (void)group_open; (void)group_polys; // grouping does not change the f-line order
        } else if (cmd == "mtllib" || cmd == "usemtl" || cmd == "s") {
        } else if (cmd[0] == '#') {
        } else { err = "obj: unexpected command '" + cmd + "' on line " + std::to_string(lineno); return 1; }
        if (eol >= len) break;
    }
    return 0;
}

// ---- src/shape/triangle.rs --------------------------------------------------
struct Triangle : Primitive {
    const Obj *obj;
    uint32_t face;
    V3 pos(int k) const { // p0/p1/p2: f32 -> f64 `.into()` (triangle.rs:39-55)
        const float *v = &obj->position[3 * (size_t)obj->tri[3 * (size_t)face + k].v];
        return V3{(double)v[0], (double)v[1], (double)v[2]};
    }
    V3 nrm(int k) const {
        const float *v = &obj->normal[3 * (size_t)obj->tri[3 * (size_t)face + k].n];
        return V3{(double)v[0], (double)v[1], (double)v[2]};
    }
    bool has_n() const { return !obj->normal.empty(); }
    bool has_uv() const { return !obj->texture.empty(); }
    void uv(double out[3][2]) const { // triangle.rs:80-114
        if (has_uv()) {
            for (int k = 0; k < 3; ++k) {
                const float *t = &obj->texture[2 * (size_t)obj->tri[3 * (size_t)face + k].t];
                out[k][0] = (double)t[0]; out[k][1] = (double)t[1];
            }
        } else {
            out[0][0] = 0.0; out[0][1] = 0.0; out[1][0] = 1.0; out[1][1] = 0.0; out[2][0] = 1.0; out[2][1] = 1.0;
        }
    }
    Bounds bound() const override { return bounds_point_union(bounds_new(pos(0), pos(1)), pos(2)); } // :157-159
    const Primitive *intersect(const Ray &ray, RayIntersection &isect) const override { // triangle.rs:161-307
        tl_stats.triangles_tested++;
        V3 p0 = pos(0), p1 = pos(1), p2 = pos(2);
        V3 p0t = p0 - ray.origin, p1t = p1 - ray.origin, p2t = p2 - ray.origin;
        int kz = max_dimension(vabs(ray.d));
        int kx = (kz + 1) % 3;
        int ky = (kx + 1) % 3;
        V3 d{ray.d[kx], ray.d[ky], ray.d[kz]};
        p0t = V3{p0t[kx], p0t[ky], p0t[kz]};
        p1t = V3{p1t[kx], p1t[ky], p1t[kz]};
        p2t = V3{p2t[kx], p2t[ky], p2t[kz]};
        double sx = -d.x / d.z, sy = -d.y / d.z, sz = 1.0 / d.z;
        p0t.x += sx * p0t.z; p0t.y += sy * p0t.z;
        p1t.x += sx * p1t.z; p1t.y += sy * p1t.z;
        p2t.x += sx * p2t.z; p2t.y += sy * p2t.z;
        double e0 = p1t.x * p2t.y - p1t.y * p2t.x;
        double e1 = p2t.x * p0t.y - p2t.y * p0t.x;
        double e2 = p0t.x * p1t.y - p0t.y * p1t.x;
        if ((e0 < 0.0 || e1 < 0.0 || e2 < 0.0) && (e0 > 0.0 || e1 > 0.0 || e2 > 0.0)) return nullptr;
        double det = e0 + e1 + e2;
        if (det == 0.0) return nullptr;
        p0t.z *= sz; p1t.z *= sz; p2t.z *= sz;
        double tscaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
        if ((det < 0.0 && tscaled >= 0.0) || (det > 0.0 && tscaled <= 0.0)) return nullptr;
        double invdet = 1.0 / det;
        double b0 = e0 * invdet, b1 = e1 * invdet, b2 = e2 * invdet;
        double t = tscaled * invdet;
        if (t >= isect.t) return nullptr;

        double uvs[3][2]; uv(uvs);
        double duv02x = uvs[0][0] - uvs[2][0], duv02y = uvs[0][1] - uvs[2][1];
        double duv12x = uvs[1][0] - uvs[2][0], duv12y = uvs[1][1] - uvs[2][1];
        V3 dp02 = p0 - p2, dp12 = p1 - p2;
        double determinant = (duv02x * duv12y) - (duv02y * duv12x);
        V3 dpdu, dpdv;
        if (determinant == 0.0) {
            coordinate_system(cross(p2 - p1, p1 - p0), dpdu, dpdv);
        } else {
            double inv = 1.0 / determinant;
            dpdu = (duv12y * dp02 - duv02y * dp12) * inv;
            dpdv = (-duv12x * dp02 - duv02x * dp12) * inv;
        }
        // triangle.rs:276 parses as b0 * (uv0 + b1*uv1 + b2*uv2); uv is never read afterwards
        double hu = b0 * ((uvs[0][0] + b1 * uvs[1][0]) + b2 * uvs[2][0]);
        double hv = b0 * ((uvs[0][1] + b1 * uvs[1][1]) + b2 * uvs[2][1]);
        isect = isect_new(t, hu, hv, dpdu, dpdv);
        if (has_n()) {
            V3 n0 = nrm(0), n1 = nrm(1), n2 = nrm(2);
            V3 ns = b0 * n0 + b1 * n1 + b2 * n2;
            V3 ss = isect.geometry.dpdu;
            V3 ts = cross(ns, ss);
            if (magnitude2(ts) > 0.0) { ss = cross(ts, ns); }
            else { coordinate_system(ns, ss, ts); }
            isect.has_n = true; isect.n = ns;
            isect.surface.dpdu = ss; isect.surface.dpdv = ts;
        } else {
            isect.has_n = true;
            isect.n = face_forward(cross(dp02, dp12), -ray.d);
        }
        return this;
    }
};

// ---- src/light/point.rs, src/material/background.rs, src/camera.rs ----------
struct PointLight { V3 position, intensity; double falloff[3]; };
struct Background {
    V3 inner, outer; double scale;
    V3 bg(V3 d) const { // background.rs:25-34 (powf(2.) == x*x: LLVM folds pow(x,2) to a multiply)
        double a = std::fabs(dot(V3{0.0, 0.0, 1.0}, d));
        double t = fmin_(std::sqrt(1.0 - a * a) / scale, 1.0);
        return V3{lerp(t, inner.x, outer.x), lerp(t, inner.y, outer.y), lerp(t, inner.z, outer.z)};
    }
};
struct Img { uint32_t w, h; double winv, hinv, aspect; }; // film.rs:36-45
struct Camera {
    V3 origin{0, 0, 0}, view{0, 0, 1}, up{0, 1, 0}, aux{1, 0, 0};
    bool perspective = true;
    double param = 45.0; // fov (deg) or orthographic height
    size_t ss_root = 1; double ss_distance = 1.0;
    double aperture_radius = 0.0;
    double image_plane_height = 0.0, pixel_separation = 0.0;
    double plane_height(double focal) const { // camera.rs:158-164
        if (perspective) return focal * std::tan(param * PI / 360.) * 2.;
        return param;
    }
    void init(bool persp, double p) { // camera.rs:61-73
        *this = Camera();
        perspective = persp; param = p;
        image_plane_height = plane_height(1.);
        pixel_separation = persp ? 0. : 1.;
    }
    void look_at(V3 o, V3 look, V3 upv) { // camera.rs:85-94
        V3 v = look - o;
        V3 a = cross(v, upv);
        origin = o;
        up = normalize(cross(a, v));
        aux = normalize(a);
        view = v;
        image_plane_height = plane_height(magnitude(v));
    }
    void set_supersampling(uint8_t base) { ss_root = (size_t)base + 1; ss_distance = 1. / (double)ss_root; } // :189-193
    size_t num_samples() const { return ss_root * ss_root; }
    void sample(uint32_t x, uint32_t y, const Img &img, Ray *rays) const { // camera.rs:113-146
        double img_plane_height = image_plane_height;
        double img_plane_width = img_plane_height * img.aspect;
        double pixel_size = img_plane_height * img.hinv;
        double sample_separation = ss_distance * pixel_size;
        double sox = ((double)x * img.winv - 0.5) * img_plane_width;
        double soy = (0.5 - (double)(y + 1u) * img.hinv) * img_plane_height;
        V3 o = origin + ((soy * pixel_separation) * up) + ((sox * pixel_separation) * aux);
        V3 d = view + (soy * up) + (sox * aux);
        V3 updiff = up * sample_separation;
        V3 auxdiff = aux * sample_separation;
        V3 halfdiff = updiff * 0.5 + auxdiff * 0.5;
        size_t dim = ss_root;
        for (size_t i = 0; i < dim; ++i)
            for (size_t j = 0; j < dim; ++j) {
                size_t idx = i * dim + j;
                V3 dd = d + ((double)j * updiff) + ((double)i * auxdiff) + halfdiff;
                rays[idx] = ray_new(o, dd);
            }
    }
};

// ---- src/scene.rs, src/scene/node.rs ----------------------------------------
struct Aggregate;
struct SceneNode {
    enum Kind { SPHERE, CUBE, CUBOID, MESH, GROUP } kind;
    double a[3], b[3]; // sphere: a=center, b[0]=radius; cube: a=origin, b[0]=dim; cuboid: a=min, b=max
    Material mat; bool has_mat;
    uint32_t obj;
    std::unique_ptr<Aggregate> group;
};
struct Aggregate {
    std::vector<SceneNode> contents;
    Transform transform = tr_identity();
    bool swap_backface = false;
};
struct Scene {
    std::unique_ptr<Aggregate> root{new Aggregate()};
    Camera camera;
    Background background{ZERO3, ZERO3, 1.0};
    V3 ambient = ZERO3;
    bool smoothing = true;
    uint32_t recursion = 3;
    size_t threads = 0;
    std::vector<PointLight> lights;
    std::vector<std::unique_ptr<Obj>> meshes;
    Scene() { camera.init(true, 45.); }
};

// ---- src/accelerators/bvh.rs ------------------------------------------------
struct LinearNode { Bounds bounds; bool leaf; uint32_t a; uint32_t b; }; // leaf: (prim_offset, nprims u16) ; interior: (axis, second)
struct BuildNode { bool leaf; size_t first, n; int axis; BuildNode *c0, *c1; Bounds bounds; };
struct MortonPrim { size_t index; uint32_t code; };
struct PrimInfo { size_t number; Bounds bounds; V3 centroid; };

static inline uint32_t left_shift_3(uint32_t x) { // bvh.rs:590-598
    if (x == (1u << 10)) x -= 1;
    x = (x | (x << 16)) & 0b00000011000000000000000011111111u;
    x = (x | (x << 8)) & 0b00000011000000001111000000001111u;
    x = (x | (x << 4)) & 0b00000011000011000011000011000011u;
    x = (x | (x << 2)) & 0b00001001001001001001001001001001u;
    return x;
}
static inline uint32_t encode_morton_3(V3 v) { // bvh.rs:575-579 (z,y,z -- x never contributes; kept)
    return (left_shift_3(as_u32(v.z)) << 2) | (left_shift_3(as_u32(v.y)) << 1) | left_shift_3(as_u32(v.z));
}
static void radix_sort(std::vector<MortonPrim> &v) { // bvh.rs:600-635
    std::vector<MortonPrim> temp(v.size(), MortonPrim{0, 0});
    const uint32_t BITS = 6, NPASS = 5, NB = 64, MASK = 63;
    for (uint32_t pass = 0; pass < NPASS; ++pass) {
        uint32_t lowbit = pass * BITS;
        std::vector<MortonPrim> &in = (pass & 1) == 0 ? v : temp;
        std::vector<MortonPrim> &out = (pass & 1) == 0 ? temp : v;
        size_t count[NB] = {0};
        for (const auto &mp : in) count[(mp.code >> lowbit) & MASK]++;
        size_t idx[NB]; idx[0] = 0;
        for (uint32_t i = 1; i < NB; ++i) idx[i] = idx[i - 1] + count[i - 1];
        for (const auto &mp : in) out[idx[(mp.code >> lowbit) & MASK]++] = mp;
    }
    if (NPASS & 1) v.swap(temp);
}
// third-party `partition ^0.1`: in-place, unstable, two-pointer swap partition (parity unpinned)
template <class T, class P> static size_t partition_slice(T *data, size_t len, P pred) {
    if (len == 0) return 0;
    size_t l = 0, r = len - 1;
    for (;;) {
        while (l < len && pred(data[l])) ++l;
        while (r > 0 && !pred(data[r])) --r;
        if (l >= r) return l;
        std::swap(data[l], data[r]);
    }
}

struct BuildError { std::string msg; };

struct BVHAccel : Primitive {
    const Scene *scene = nullptr;
    std::vector<std::unique_ptr<Primitive>> primitives;
    std::vector<LinearNode> nodes;
    const Transform *transform = nullptr;
    std::vector<size_t> order;
    bool has_material = false; Material mat{};
    uint8_t max_prims_per_node = 0;
    bool swap_backface = false;
    std::vector<std::unique_ptr<BuildNode>> arena;

    BuildNode *alloc() { arena.emplace_back(new BuildNode{true, 0, 0, 0, nullptr, nullptr, bounds_none()}); return arena.back().get(); }

    void init(const Scene *sc, std::vector<std::unique_ptr<Primitive>> prims, const Transform *tr, bool has_mat, Material m,
              size_t max_prims, bool swap) { // bvh.rs:164-202
        scene = sc; primitives = std::move(prims); transform = tr; has_material = has_mat; mat = m;
        max_prims_per_node = (uint8_t)(max_prims < 255 ? max_prims : 255);
        swap_backface = swap;
        size_t nprims = primitives.size();
        if (nprims == 0) throw BuildError{"empty aggregate: the reference recurses without bound in build_upper_sah (bvh.rs:355-424)"};
        std::vector<PrimInfo> info(nprims);
        for (size_t i = 0; i < nprims; ++i) {
            Bounds b = primitives[i]->bound();
            info[i] = PrimInfo{i, b, 0.5 * b.min + 0.5 * b.max}; // bvh.rs:525-533
        }
        order.assign(nprims, (size_t)-1);
        size_t total = 0;
        BuildNode *root = build(info, total);
        nodes.assign(total, LinearNode{bounds_none(), true, 0, 0});
        size_t off = 0;
        flatten(root, off);
        arena.clear();
    }

    BuildNode *build(const std::vector<PrimInfo> &info, size_t &total_nodes) { // bvh.rs:205-273
        Bounds bounds = bounds_none();
        for (const auto &i : info) bounds = bounds_union(bounds, i.bounds);
        std::vector<MortonPrim> mp(info.size());
        for (size_t i = 0; i < info.size(); ++i) {
            V3 off = bounds_offset(bounds, info[i].centroid);
            mp[i] = MortonPrim{info[i].number, encode_morton_3(off * 1024.0)};
        }
        radix_sort(mp);
        std::vector<BuildNode *> treelets;
        size_t start = 0, ordered_off = 0, total = 0;
        for (size_t end = 1; end <= mp.size(); ++end) {
            const uint32_t mask = 0b00111111111111000000000000000000u;
            if (end == mp.size() || ((mp[start].code & mask) != (mp[end].code & mask))) {
                size_t created = 0;
                BuildNode *n = emit_lbvh(&mp[start], end - start, info, created, ordered_off, 29 - 12);
                total += created;
                treelets.push_back(n);
                start = end;
            }
        }
        total_nodes += total;
        return build_upper_sah(treelets.data(), treelets.size(), total_nodes, 0);
    }

    BuildNode *emit_lbvh(const MortonPrim *mp, size_t nprims, const std::vector<PrimInfo> &info, size_t &total_nodes,
                         size_t &ordered_off, int bit_index) { // bvh.rs:278-347
        if (bit_index == -1 || nprims < (size_t)max_prims_per_node) {
            size_t first = ordered_off;
            BuildNode *node = alloc();
            ordered_off += nprims;
            total_nodes += 1;
            Bounds b = bounds_none();
            for (size_t i = 0; i < nprims; ++i) {
                size_t pi = mp[i].index;
                order[first + i] = pi;
                b = bounds_union(b, info[pi].bounds);
            }
            node->leaf = true; node->first = first; node->n = nprims; node->bounds = b;
            return node;
        }
        uint32_t mask = 1u << bit_index;
        if ((mp[0].code & mask) == (mp[nprims - 1].code & mask))
            return emit_lbvh(mp, nprims, info, total_nodes, ordered_off, bit_index - 1);
        size_t s = 0, e = nprims - 1;
        while (s + 1 != e) {
            size_t mid = (s + e) / 2;
            if ((mp[s].code & mask) == (mp[mid].code & mask)) s = mid; else e = mid;
        }
        size_t split = e;
        BuildNode *node = alloc();
        total_nodes += 1;
        BuildNode *l0 = emit_lbvh(mp, split, info, total_nodes, ordered_off, bit_index - 1);
        BuildNode *l1 = emit_lbvh(mp + split, nprims - split, info, total_nodes, ordered_off, bit_index - 1);
        node->leaf = false; node->axis = bit_index % 3; node->c0 = l0; node->c1 = l1;
        node->bounds = bounds_union(l0->bounds, l1->bounds);
        return node;
    }

    BuildNode *build_upper_sah(BuildNode **roots, size_t ncount, size_t &total_nodes, int depth) { // bvh.rs:350-427
        if (ncount == 1) return roots[0];
        if (ncount == 0 || depth > 4096)
            throw BuildError{"degenerate upper-SAH split: the reference recurses without bound here (bvh.rs:414-424)"};
        BuildNode *node = alloc();
        total_nodes += 1;
        Bounds bounds = bounds_none();
        for (size_t i = 0; i < ncount; ++i) bounds = bounds_union(bounds, roots[i]->bounds);
        Bounds cb = bounds_none();
        for (size_t i = 0; i < ncount; ++i) {
            V3 c = 0.5 * (roots[i]->bounds.min + roots[i]->bounds.max);
            cb = bounds_point_union(cb, c);
        }
        int dim = bounds_maximum_extent(cb);
        const int NB = 12;
        struct Bucket { size_t count; Bounds b; } buckets[NB];
        for (int i = 0; i < NB; ++i) buckets[i] = Bucket{0, bounds_none()};
        auto bucket_of = [&](const BuildNode *r) {
            double centroid = (r->bounds.min[dim] + r->bounds.max[dim]) * 0.5;
            double b0 = (centroid - cb.min[dim]) / (cb.max[dim] - cb.min[dim]);
            size_t b = (size_t)as_u32((double)NB * b0);
            if (b == (size_t)NB) b = NB - 1;
            return b;
        };
        for (size_t i = 0; i < ncount; ++i) {
            size_t b = bucket_of(roots[i]);
            if (b >= (size_t)NB) throw BuildError{"SAH bucket index out of range (the reference would panic)"};
            buckets[b].count += 1;
            buckets[b].b = bounds_union(buckets[b].b, roots[i]->bounds);
        }
        double cost[NB];
        for (int i = 0; i < NB; ++i) {
            Bounds b0 = bounds_none(); size_t c0 = 0;
            for (int j = 0; j <= i; ++j) { b0 = bounds_union(b0, buckets[j].b); c0 += buckets[j].count; }
            Bounds b1 = bounds_none(); size_t c1 = 0;
            for (int j = i + 1; j < NB; ++j) { b1 = bounds_union(b1, buckets[j].b); c1 += buckets[j].count; }
            cost[i] = 0.125 + ((double)c0 * bounds_surface_area(b0) + (double)c1 * bounds_surface_area(b1)) / bounds_surface_area(bounds);
        }
        size_t split = 0;
        for (int i = 0; i < NB; ++i) if (cost[i] < cost[split]) split = (size_t)i;
        size_t mid = partition_slice(roots, ncount, [&](BuildNode *n) { return bucket_of(n) <= split; });
        BuildNode *lo = build_upper_sah(roots, mid, total_nodes, depth + 1);
        BuildNode *hi = build_upper_sah(roots + mid, ncount - mid, total_nodes, depth + 1);
        node->leaf = false; node->axis = dim; node->c0 = lo; node->c1 = hi;
        node->bounds = bounds_union(lo->bounds, hi->bounds);
        return node;
    }

    size_t flatten(const BuildNode *n, size_t &offset) { // bvh.rs:430-453
        size_t my = offset++;
        nodes[my].bounds = n->bounds;
        if (n->leaf) {
            nodes[my].leaf = true; nodes[my].a = (uint32_t)n->first; nodes[my].b = (uint32_t)(uint16_t)n->n;
        } else {
            flatten(n->c0, offset);
            size_t second = flatten(n->c1, offset);
            nodes[my].leaf = false; nodes[my].a = (uint32_t)(uint8_t)n->axis; nodes[my].b = (uint32_t)second;
        }
        return my;
    }

    Bounds bound() const override { return tr_transform_bounds(*transform, nodes[0].bounds); } // bvh.rs:457-459

    const Primitive *intersect(const Ray &wray, RayIntersection &isect) const override { // bvh.rs:461-522
        tl_stats.accel_entries++;
        Ray ray = tr_inverse_transform_ray(*transform, wray);
        bool dir_is_neg[3] = {ray.dinv.x < 0.0, ray.dinv.y < 0.0, ray.dinv.z < 0.0};
        RayIntersection isect_inv = tr_inverse_transform_isect(*transform, isect);
        const Primitive *hit = nullptr;
        size_t to_visit = 0, cur = 0;
        size_t stack[64];
        for (;;) {
            const LinearNode &node = nodes[cur];
            tl_stats.nodes_tested++;
            if (!bounds_intersects(node.bounds, ray)) {
                if (to_visit == 0) break;
                cur = stack[--to_visit];
                continue;
            }
            if (node.leaf) {
                for (uint32_t i = 0; i < node.b; ++i) {
                    size_t pi = order[node.a + i];
                    const Primitive *p = primitives[pi]->intersect(ray, isect_inv);
                    if (p) hit = p;
                }
                if (to_visit == 0) break;
                cur = stack[--to_visit];
            } else {
                if (to_visit >= 64) { std::fprintf(stderr, "oracle: BVH stack overflow (the reference would panic, bvh.rs:497)\n"); std::abort(); }
                if (dir_is_neg[node.a]) { stack[to_visit] = cur + 1; cur = node.b; }
                else { stack[to_visit] = node.b; cur = cur + 1; }
                to_visit++;
            }
        }
        if (hit) {
            isect = tr_transform_isect(*transform, isect_inv);
            if (has_material) isect.material = mat;
            if (swap_backface) isect_swap_backface(isect);
        }
        return hit;
    }
};

static const Transform ID_TRANSFORM = tr_identity();

static std::unique_ptr<BVHAccel> accel_from_mesh(const Scene *scene, uint32_t mesh, bool has_mat, Material mat) { // bvh.rs:141-148
    if (mesh >= scene->meshes.size()) throw BuildError{"mesh handle out of range (the reference panics, bvh.rs:142)"};
    const Obj *obj = scene->meshes[mesh].get();
    size_t nfaces = obj->tri.size() / 3;
    std::vector<std::unique_ptr<Primitive>> tris;
    tris.reserve(nfaces);
    for (size_t f = 0; f < nfaces; ++f) {
        for (int k = 0; k < 3; ++k) {
            const auto &tp = obj->tri[3 * f + k];
            if (!obj->normal.empty() && tp.n < 0) throw BuildError{"mesh has normals but a face lacks a vn index (the reference panics, triangle.rs:60)"};
            if (!obj->texture.empty() && tp.t < 0) throw BuildError{"mesh has vt but a face lacks a vt index (the reference panics, triangle.rs:96)"};
        }
        auto *t = new Triangle(); t->obj = obj; t->face = (uint32_t)f;
        tris.emplace_back(t);
    }
    std::unique_ptr<BVHAccel> a(new BVHAccel());
    size_t per_node = tris.size();
    a->init(scene, std::move(tris), &ID_TRANSFORM, has_mat, mat, per_node, false);
    return a;
}
static std::unique_ptr<BVHAccel> accel_from_aggregate(const Scene *scene, const Aggregate *agg) { // bvh.rs:150-162, 563-572
    std::vector<std::unique_ptr<Primitive>> prims;
    for (const SceneNode &n : agg->contents) {
        switch (n.kind) {
        case SceneNode::SPHERE: {
            auto *s = new Sphere(); s->origin = V3{n.a[0], n.a[1], n.a[2]}; s->radius = n.b[0]; s->mat = n.mat;
            prims.emplace_back(s); break; }
        case SceneNode::CUBE: { // Cuboid::cube cuboid.rs:24-30
            auto *c = new Cuboid(); V3 o{n.a[0], n.a[1], n.a[2]};
            c->bounds = bounds_new(o, o + V3{n.b[0], n.b[0], n.b[0]}); c->mat = n.mat;
            prims.emplace_back(c); break; }
        case SceneNode::CUBOID: {
            auto *c = new Cuboid();
            c->bounds = bounds_new(V3{n.a[0], n.a[1], n.a[2]}, V3{n.b[0], n.b[1], n.b[2]}); c->mat = n.mat;
            prims.emplace_back(c); break; }
        case SceneNode::MESH: prims.emplace_back(accel_from_mesh(scene, n.obj, n.has_mat, n.mat).release()); break;
        case SceneNode::GROUP: prims.emplace_back(accel_from_aggregate(scene, n.group.get()).release()); break;
        }
    }
    std::unique_ptr<BVHAccel> a(new BVHAccel());
    size_t per_node = prims.size();
    a->init(scene, std::move(prims), &agg->transform, false, Material{}, per_node, agg->swap_backface);
    return a;
}

// ---- src/core/bxdf/* --------------------------------------------------------
namespace bx {
static inline double cos_theta(V3 w) { return w.z; }
static inline double cos2_theta(V3 w) { return w.z * w.z; }
static inline double abs_cos_theta(V3 w) { return std::fabs(w.z); }
static inline double sin2_theta(V3 w) { return fmax_(1.0 - cos2_theta(w), 0.0); }
static inline double sin_theta(V3 w) { return std::sqrt(sin2_theta(w)); }
static inline double tan_theta(V3 w) { return sin_theta(w) / cos_theta(w); }
static inline double tan2_theta(V3 w) { return sin2_theta(w) / cos2_theta(w); }
static inline double cos_phi(V3 w) { double s = sin_theta(w); return s == 0.0 ? 1.0 : fmin_(fmax_(w.x / s, -1.0), 1.0); }
static inline double sin_phi(V3 w) { double s = sin_theta(w); return s == 0.0 ? 0.0 : fmin_(fmax_(w.y / s, -1.0), 1.0); }
static inline double cos2_phi(V3 w) { return cos_phi(w) * cos_phi(w); }
static inline double sin2_phi(V3 w) { return sin_phi(w) * sin_phi(w); }
static inline V3 reflect(V3 wo, V3 n) { return -1.0 * wo + 2.0 * dot(wo, n) * n; } // bxdf/mod.rs:269-271
static inline bool refract(V3 wi, V3 n, double eta, V3 &out) { // bxdf/mod.rs:276-288
    double cos_theta_i = dot(n, wi);
    double sin2_theta_i = fmax_(1.0 - cos_theta_i * cos_theta_i, 0.0);
    double sin2_theta_t = eta * eta * sin2_theta_i;
    if (sin2_theta_t >= 1.0) return false;
    double cos_theta_t = std::sqrt(1.0 - sin2_theta_t);
    out = eta * -1.0 * wi + (eta * cos_theta_i - cos_theta_t) * n;
    return true;
}
// fresnel.rs
enum SubKind { DIELECTRIC, CONDUCTOR, NOOP };
struct Substance { SubKind kind; double eta_i, eta_t; V3 ci, ct, k; };
static inline double dielectric(double cos_theta_i, double eta_i, double eta_t) { // fresnel.rs:37-64
    cos_theta_i = fmin_(fmax_(cos_theta_i, -1.0), 1.0);
    bool entering = cos_theta_i > 0.0;
    if (!entering) { std::swap(eta_i, eta_t); cos_theta_i = std::fabs(cos_theta_i); }
    double sin_theta_i = std::sqrt(fmax_(1.0 - cos_theta_i * cos_theta_i, 0.0));
    double sin_theta_t = eta_i / eta_t * sin_theta_i;
    if (sin_theta_t >= 1.0) return 1.0;
    double cos_theta_t = std::sqrt(fmax_(1.0 - sin_theta_t * sin_theta_t, 0.0));
    double r_parl = ((eta_t * cos_theta_i) - (eta_i * cos_theta_t)) / ((eta_t * cos_theta_i) + (eta_i * cos_theta_t));
    double r_perp = ((eta_i * cos_theta_i) - (eta_t * cos_theta_t)) / ((eta_i * cos_theta_i) + (eta_t * cos_theta_t));
    return (r_parl * r_parl + r_perp * r_perp) * 0.5;
}
static inline V3 vsqrt(V3 v) { return V3{std::sqrt(v.x), std::sqrt(v.y), std::sqrt(v.z)}; }
static inline V3 splat(double v) { return V3{v, v, v}; }
static inline V3 conductor(double cos_theta_i, V3 eta_i, V3 eta_t, V3 k) { // fresnel.rs:69-91
    cos_theta_i = fmin_(fmax_(cos_theta_i, -1.0), 1.0);
    V3 eta = div_ew(eta_t, eta_i);
    V3 etak = div_ew(k, eta_i);
    double c2 = cos_theta_i * cos_theta_i;
    double s2 = 1.0 - c2;
    V3 eta2 = mul_ew(eta, eta), etak2 = mul_ew(etak, etak);
    V3 t0 = eta2 - etak2 - splat(s2);
    V3 a2plusb2 = vsqrt(mul_ew(t0, t0) + 4.0 * mul_ew(eta2, etak2));
    V3 t1 = a2plusb2 + splat(c2);
    V3 a = vsqrt(0.5 * (a2plusb2 + t0));
    V3 t2 = 2.0 * cos_theta_i * a;
    V3 rs = div_ew(t1 - t2, t1 + t2);
    V3 t3 = c2 * a2plusb2 + splat(s2 * s2);
    V3 t4 = t2 * s2;
    V3 rp = div_ew(mul_ew(rs, t3 - t4), t3 + t4);
    return 0.5 * (rp + rs);
}
static inline V3 substance_evaluate(const Substance &s, double cos_theta_i) { // fresnel.rs:20-28
    switch (s.kind) {
    case DIELECTRIC: return splat(dielectric(cos_theta_i, s.eta_i, s.eta_t));
    case CONDUCTOR: return conductor(cos_theta_i, s.ci, s.ct, s.k);
    default: return splat(1.0);
    }
}
// microfacet.rs Distribution
struct Distribution { double alphax, alphay; };
static inline double tr_d(const Distribution &m, V3 wh) { // microfacet.rs:31-40
    double tan2 = tan2_theta(wh);
    if (std::isinf(tan2)) return 0.0;
    double cos4 = cos2_theta(wh) * cos2_theta(wh);
    double e = (cos2_phi(wh) / (m.alphax * m.alphax) + sin2_phi(wh) / (m.alphay * m.alphay)) * tan2;
    return 1.0 / (PI * m.alphax * m.alphay * cos4 * (1.0 + e) * (1.0 + e));
}
static inline double tr_lambda(const Distribution &m, V3 w) { // microfacet.rs:55-66
    double abs_tan = std::fabs(tan_theta(w));
    if (std::isinf(abs_tan)) return 0.0;
    double alpha = std::sqrt(cos2_phi(w) * m.alphax * m.alphax + sin2_phi(w) * m.alphay * m.alphay);
    double a2t2 = (alpha * abs_tan) * (alpha * abs_tan);
    return (std::sqrt(1.0 + a2t2) - 1.0) / 2.0;
}
static inline double tr_g(const Distribution &m, V3 wo, V3 wi) { return 1.0 / (1.0 + tr_lambda(m, wo) + tr_lambda(m, wi)); } // :44-46

enum BxKind { CONSTANT, SPEC_REFL, SPEC_TRANS, QUICK_DIFFUSE, DIFFUSE, MICRO_REFL, MICRO_TRANS };
enum { T_REFLECTION = 1, T_TRANSMISSION = 2, T_DIFFUSE = 4, T_GLOSSY = 8, T_SPECULAR = 16 };
struct BxDF {
    BxKind kind;
    V3 r;                 // reflection / transmission spectrum
    Substance substance;
    Distribution dist;
    double eta_a, eta_b;  // specular transmission
    double on_a, on_b;    // Oren-Nayar
    int type() const { // bxdf/mod.rs:139-151
        switch (kind) {
        case CONSTANT: return 0;
        case SPEC_REFL: return T_REFLECTION | T_SPECULAR;
        case SPEC_TRANS: return T_TRANSMISSION | T_SPECULAR;
        case QUICK_DIFFUSE: case DIFFUSE: return T_REFLECTION | T_DIFFUSE;
        case MICRO_REFL: return T_REFLECTION | T_GLOSSY;
        default: return T_TRANSMISSION | T_GLOSSY;
        }
    }
    bool matches(int flags) const { int t = type(); return (t & flags) == t; }
    bool has_t(int flags) const { return (type() & flags) != 0; }
    V3 f(V3 wo, V3 wi) const { // bxdf/mod.rs:164-174
        switch (kind) {
        case CONSTANT: return r;
        case QUICK_DIFFUSE: return r * FRAC_1_PI; // diffuse.rs:14
        case DIFFUSE: { // diffuse.rs:36-56
            double sin_i = sin_theta(wi), sin_o = sin_theta(wo);
            double max_cos = 0.0;
            if (sin_i > 1e-4 && sin_o > 1e-4) {
                double sp_i = sin_phi(wi), cp_i = cos_phi(wi), sp_o = sin_phi(wo), cp_o = cos_phi(wo);
                double d_cos = cp_i * cp_o + sp_i * sp_o;
                max_cos = fmax_(d_cos, 0.0);
            }
            double sin_alpha, tan_beta;
            if (abs_cos_theta(wi) > abs_cos_theta(wo)) { sin_alpha = sin_o; tan_beta = sin_i / abs_cos_theta(wi); }
            else { sin_alpha = sin_i; tan_beta = sin_o / abs_cos_theta(wo); }
            return r * FRAC_1_PI * (on_a + on_b * max_cos * sin_alpha * tan_beta);
        }
        case MICRO_REFL: { // microfacet.rs:101-115
            double cos_o = abs_cos_theta(wo), cos_i = abs_cos_theta(wi);
            V3 wh = wi + wo;
            if (cos_i == 0.0 || cos_o == 0.0) return ZERO3;
            if (wh.x == 0.0 && wh.y == 0.0 && wh.z == 0.0) return ZERO3;
            wh = normalize(wh);
            V3 spectrum = substance_evaluate(substance, dot(wi, wh));
            return mul_ew(r * tr_d(dist, wh) * tr_g(dist, wo, wi), spectrum) / (4.0 * cos_i * cos_o);
        }
        case MICRO_TRANS: return ZERO3; // never constructed: Material::glass passes roughness 0 (material/mod.rs:39-40)
        default: return ZERO3;
        }
    }
};
struct LightSample { V3 spectrum, wi; double pdf; };
static inline LightSample ls_zero() { return LightSample{ZERO3, ZERO3, 0.0}; }
static inline LightSample bxdf_sample_f(const BxDF &b, V3 wo) { // specular.rs:17-24, 43-63 (other kinds unreachable from li)
    if (b.kind == SPEC_REFL) {
        V3 wi{-wo.x, -wo.y, wo.z};
        V3 spectrum = mul_ew(substance_evaluate(b.substance, cos_theta(wi)), b.r) / abs_cos_theta(wi);
        return LightSample{spectrum, wi, 1.0};
    }
    if (b.kind == SPEC_TRANS) {
        bool entering = cos_theta(wo) > 0.0;
        double eta_i = entering ? b.eta_a : b.eta_b, eta_t = entering ? b.eta_b : b.eta_a;
        V3 wi;
        if (refract(wo, V3{0.0, 0.0, 1.0}, eta_i / eta_t, wi)) {
            V3 spectrum = mul_ew(b.r, splat(1.0) - substance_evaluate(b.substance, cos_theta(wi))) / abs_cos_theta(wi);
            return LightSample{spectrum, wi, 1.0};
        }
        return ls_zero();
    }
    std::fprintf(stderr, "oracle: non-specular sample_f is unreachable from the Whitted integrator\n");
    std::abort();
}
} // namespace bx

// ---- src/interaction/bsdf.rs ------------------------------------------------
struct BSDF {
    double eta;
    V3 ng, ns, ss, ts;
    bx::BxDF bxdfs[8];
    size_t num = 0;
    void init(const SurfaceInteraction &si) { // bsdf.rs:29-46
        eta = 1.0; ng = si.ng; ns = si.ns; ss = si.surface.dpdu; ts = cross(ns, ss); num = 0;
    }
    void add(const bx::BxDF &b) { bxdfs[num++] = b; }
    V3 to_local(V3 v) const { return V3{dot(v, ss), dot(v, ts), dot(v, ns)}; } // :155-161
    V3 to_world(V3 v) const { // :165-171
        return V3{ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z};
    }
    V3 f(V3 wo, V3 wi) const { // bsdf.rs:73-92
        bool reflect = dot(wi, ng) * dot(wo, ng) > 0.0;
        V3 wo_l = to_local(wo), wi_l = to_local(wi);
        if (wo_l.z == 0.0) return ZERO3;
        V3 f = ZERO3;
        for (size_t i = 0; i < num; ++i) {
            const bx::BxDF &b = bxdfs[i];
            if ((reflect && b.has_t(bx::T_REFLECTION)) || (!reflect && b.has_t(bx::T_TRANSMISSION))) f = f + b.f(wo_l, wi_l);
        }
        return f;
    }
    bx::LightSample sample_f(V3 wo, double sx, double sy, int flags) const { // bsdf.rs:94-145
        (void)sy;
        size_t matching = 0;
        for (size_t i = 0; i < num; ++i) if (bxdfs[i].matches(flags)) matching++;
        if (matching == 0) return bx::ls_zero();
        size_t comp = (size_t)std::floor(sx * (double)matching);
        if (comp > matching - 1) comp = matching - 1;
        const bx::BxDF *bxdf = nullptr; size_t seen = 0;
        for (size_t i = 0; i < num; ++i) if (bxdfs[i].matches(flags)) { if (seen == comp) { bxdf = &bxdfs[i]; break; } seen++; }
        V3 wo_l = to_local(wo);
        if (wo_l.z == 0.0) return bx::ls_zero();
        bx::LightSample fs = bx::bxdf_sample_f(*bxdf, wo_l);
        if (fs.pdf == 0.0) return fs;
        V3 wi = to_world(fs.wi);
        // chosen component is always SPECULAR here (flags always contain SPECULAR and matches() is a subset test)
        V3 sp = fs.spectrum;
        sp = V3{fmin_(fmax_(sp.x, 0.0), 1.0), fmin_(fmax_(sp.y, 0.0), 1.0), fmin_(fmax_(sp.z, 0.0), 1.0)};
        double pdf = fs.pdf / (double)matching;
        return bx::LightSample{sp, wi, pdf};
    }
};

// ---- src/material/*.rs ------------------------------------------------------
static inline bx::BxDF mk_quick_diffuse(V3 r) { bx::BxDF b{}; b.kind = bx::QUICK_DIFFUSE; b.r = r; return b; }
static void material_scattering(const Material &m, const SurfaceInteraction &si, BSDF &bsdf) {
    bsdf.init(si);
    switch (m.kind) {
    case MATTE: { // matte.rs:18-26
        V3 kd{m.p[0], m.p[1], m.p[2]}; double sigma = m.p[3];
        if (sigma == 0.0) bsdf.add(mk_quick_diffuse(kd));
        else { // diffuse.rs:29-35
            bx::BxDF b{}; b.kind = bx::DIFFUSE; b.r = kd;
            double s = deg_to_rad(sigma), s2 = s * s;
            b.on_a = 1.0 - (s2 / 2.0 * (s2 + 0.33));
            b.on_b = 0.45 * s2 / (s2 + 0.09);
            bsdf.add(b);
        }
        break; }
    case PLASTIC: { // plastic.rs:20-37
        V3 kd{m.p[0], m.p[1], m.p[2]}, ks{m.p[3], m.p[4], m.p[5]}; double rough = m.p[6];
        if (ne(kd, ZERO3)) bsdf.add(mk_quick_diffuse(kd));
        if (ne(ks, ZERO3)) {
            bx::BxDF b{}; b.kind = bx::MICRO_REFL; b.r = ks;
            b.substance = bx::Substance{bx::DIELECTRIC, 1.0, 1.5, ZERO3, ZERO3, ZERO3};
            b.dist = bx::Distribution{rough, rough};
            bsdf.add(b);
        }
        break; }
    case METAL: { // metal.rs:17-26
        V3 eta{m.p[0], m.p[1], m.p[2]}, k{m.p[3], m.p[4], m.p[5]};
        bx::BxDF b{}; b.kind = bx::MICRO_REFL; b.r = V3{1.0, 1.0, 1.0};
        b.substance = bx::Substance{bx::CONDUCTOR, 0, 0, V3{1.0, 1.0, 1.0}, eta, k};
        b.dist = bx::Distribution{m.p[6], m.p[7]};
        bsdf.add(b);
        break; }
    case GLASS: { // glass.rs:33-56 (distribution is always None: material/mod.rs:39-40)
        V3 kr{m.p[0], m.p[1], m.p[2]}, kt{m.p[3], m.p[4], m.p[5]}; double eta = m.p[6];
        if (ne(kr, ZERO3)) {
            bx::BxDF b{}; b.kind = bx::SPEC_REFL; b.r = kr;
            b.substance = bx::Substance{bx::DIELECTRIC, 1.0, eta, ZERO3, ZERO3, ZERO3};
            bsdf.add(b);
        }
        if (ne(kt, ZERO3)) {
            bx::BxDF b{}; b.kind = bx::SPEC_TRANS; b.r = kt; b.eta_a = 1.0; b.eta_b = eta;
            b.substance = bx::Substance{bx::DIELECTRIC, 1.0, eta, ZERO3, ZERO3, ZERO3};
            bsdf.add(b);
        }
        break; }
    case MIRROR: { // mirror.rs:15-17
        bx::BxDF b{}; b.kind = bx::SPEC_REFL; b.r = V3{m.p[0], m.p[1], m.p[2]};
        b.substance = bx::Substance{bx::NOOP, 0, 0, ZERO3, ZERO3, ZERO3};
        bsdf.add(b);
        break; }
    }
}

// ---- src/integrate/integrate.rs --------------------------------------------
struct Accel { Scene const *scene; std::unique_ptr<BVHAccel> root; };

static V3 li(const Accel &acc, const Ray &ray, uint32_t depth);

static V3 specular_reflect(const Accel &acc, const SurfaceInteraction &si, const BSDF &bsdf, uint32_t depth) { // :82-106
    V3 wo = si.wo;
    bx::LightSample s = bsdf.sample_f(wo, 0.5, 0.5, bx::T_REFLECTION | bx::T_SPECULAR);
    V3 ns = si.ns;
    if (s.pdf <= 0.0 || eq(s.spectrum, ZERO3) || dot(s.wi, ns) <= 0.0) return ZERO3;
    V3 wr = bx::reflect(wo, ns);
    Ray r = ray_new(si.p + si.p_err, wr);
    tl_stats.secondary_rays++;
    V3 l = li(acc, r, depth + 1);
    return mul_ew(s.spectrum, l);
}
static V3 specular_transmit(const Accel &acc, const SurfaceInteraction &si, const BSDF &bsdf, uint32_t depth) { // :108-132
    V3 wo = si.wo;
    bx::LightSample s = bsdf.sample_f(wo, 0.5, 0.5, bx::T_TRANSMISSION | bx::T_SPECULAR);
    V3 ns = si.ns;
    if (s.pdf <= 0.0 || eq(s.spectrum, ZERO3) || std::fabs(dot(s.wi, ns)) == 0.0) return ZERO3;
    Ray r = ray_new(si.p - si.p_err, s.wi);
    tl_stats.secondary_rays++;
    V3 l = li(acc, r, depth + 1);
    return mul_ew(s.spectrum, l) * std::fabs(dot(s.wi, ns)) / s.pdf;
}
static V3 li(const Accel &acc, const Ray &ray, uint32_t depth) { // integrate.rs:23-80
    const Scene &scene = *acc.scene;
    RayIntersection isect = isect_default();
    const Primitive *shape = acc.root->intersect(ray, isect);
    if (!shape) return scene.background.bg(normalize(ray.d));
    tl_stats.hits++;
    Material material;
    if (!shape->material(material)) material = isect.material;
    SurfaceInteraction si = si_from(ray, isect);
    V3 n = si.ns, wo = si.wo, p = si.p + si.p_err;
    BSDF bsdf;
    material_scattering(material, si, bsdf);
    V3 output = ZERO3;
    for (const PointLight &light : scene.lights) {
        // PointLight::sample (point.rs:42-54): un-normalised shadow ray, full closest hit, occluded iff t < 1
        V3 dl = light.position - p;
        Ray sray = ray_new(p, dl);
        RayIntersection sis = isect_default();
        tl_stats.shadow_rays++;
        acc.root->intersect(sray, sis);
        if (sis.t < 1.0) continue;
        V3 wi = light.position - p;
        double d = magnitude(wi);
        double f_att = light.falloff[0] + light.falloff[1] * d + light.falloff[2] * d * d;
        if (f_att == 0.0) continue;
        wi = normalize(wi);
        double wi_dot_n = dot(wi, n);
        V3 f = bsdf.f(wo, wi);
        output = output + (mul_ew(PI * light.intensity, f) * wi_dot_n / f_att);
    }
    output = output + mul_ew(scene.ambient, bsdf.f(wo, n));
    V3 refracted = ZERO3, reflected = ZERO3;
    if (depth < scene.recursion) {
        refracted = specular_transmit(acc, si, bsdf, depth);
        reflected = specular_reflect(acc, si, bsdf, depth);
    }
    return output + reflected + refracted;
}
static V3 integrate(const Accel &acc, const Ray *samples, size_t n, double weight) { // integrate.rs:16-20
    V3 color = ZERO3;
    for (size_t i = 0; i < n; ++i) { tl_stats.primary_rays++; color = color + li(acc, samples[i], 0); }
    return color * weight;
}

// ---- src/img.rs, src/film.rs, src/lib.rs -----------------------------------
static inline uint8_t to_byte(double c) { return as_u8(std::round(fmin_(fmax_(c, 0.0), 1.0) * 255.0)); } // img.rs:65-67
struct Film {
    Img img;
    std::vector<uint8_t> owned;
    uint8_t *px;
};
static void capture_subset_impl(size_t k, size_t n, const Accel &acc, const Img &img, uint8_t *px, double *radiance) { // lib.rs:110-162
    size_t width = img.w, height = img.h, area = width * height;
    std::vector<Ray> samples(acc.scene->camera.num_samples());
    double weight = 1. / (double)samples.size();
    for (size_t offset = k; offset < area; offset += n) {
        uint32_t x = (uint32_t)(offset % width), y = (uint32_t)(offset / width);
        acc.scene->camera.sample(x, y, img, samples.data());
        V3 color = integrate(acc, samples.data(), samples.size(), weight);
        if (px) {
            uint8_t *p = px + 4 * offset;
            p[0] = to_byte(color.x); p[1] = to_byte(color.y); p[2] = to_byte(color.z); p[3] = 255;
        }
        if (radiance) { double *r = radiance + 3 * offset; r[0] = color.x; r[1] = color.y; r[2] = color.z; }
    }
}

static thread_local std::string tl_error;
static Stats g_stats_total;
static std::mutex *g_stats_mutex = new std::mutex();
static void flush_stats() {
    std::lock_guard<std::mutex> g(*g_stats_mutex);
    g_stats_total.add(tl_stats);
    tl_stats = Stats();
}

} // namespace orc

// ============================ C API (orc_*) =================================
// Same shape as the product's include/lasgun_hip.h so that one parametrised
// Python binding (prefix "lg_" vs "orc_") drives both sides in the tests.
using namespace orc;

extern "C" {

typedef struct { int32_t kind; double p[10]; } orc_material;
typedef struct {
    uint64_t primary_rays, shadow_rays, secondary_rays, nodes_tested, spheres_tested, cuboids_tested, triangles_tested, accel_entries, hits;
} orc_stats;

static Material to_mat(const orc_material *m) { Material r; r.kind = m->kind; std::memcpy(r.p, m->p, sizeof r.p); return r; }
static orc_material from_mat(const Material &m) { orc_material r; r.kind = m.kind; std::memcpy(r.p, m.p, sizeof r.p); return r; }

const char *orc_last_error(void) { return tl_error.c_str(); }
void orc_set_trig_mode(int portable) { g_trig_mode.store(portable ? 1 : 0); }

orc_material orc_material_default(void) { return from_mat(material_default()); }
orc_material orc_material_matte(const double kd[3], double sigma) { return from_mat(material_matte(kd, sigma)); }
orc_material orc_material_plastic(const double kd[3], const double ks[3], double roughness) {
    Material m{}; m.kind = PLASTIC; for (int i = 0; i < 3; ++i) { m.p[i] = kd[i]; m.p[3 + i] = ks[i]; } m.p[6] = roughness; return from_mat(m);
}
orc_material orc_material_metal(const double eta[3], const double k[3], double u, double v) {
    Material m{}; m.kind = METAL; for (int i = 0; i < 3; ++i) { m.p[i] = eta[i]; m.p[3 + i] = k[i]; } m.p[6] = u; m.p[7] = v; return from_mat(m);
}
orc_material orc_material_glass(const double kr[3], const double kt[3], double eta) {
    Material m{}; m.kind = GLASS; for (int i = 0; i < 3; ++i) { m.p[i] = kr[i]; m.p[3 + i] = kt[i]; } m.p[6] = eta; return from_mat(m);
}
orc_material orc_material_mirror(const double kr[3]) {
    Material m{}; m.kind = MIRROR; for (int i = 0; i < 3; ++i) m.p[i] = kr[i]; return from_mat(m);
}

void *orc_scene_new(void) { return new Scene(); }
void orc_scene_free(void *s) { delete (Scene *)s; }
void orc_scene_set_perspective_camera(void *s, double fov) { ((Scene *)s)->camera.init(true, fov); }
void orc_scene_set_orthographic_camera(void *s, double h) { ((Scene *)s)->camera.init(false, h); }
void orc_camera_look_at(void *s, const double o[3], const double l[3], const double u[3]) {
    ((Scene *)s)->camera.look_at(V3{o[0], o[1], o[2]}, V3{l[0], l[1], l[2]}, V3{u[0], u[1], u[2]});
}
void orc_camera_set_supersampling(void *s, uint8_t base) { ((Scene *)s)->camera.set_supersampling(base); }
void orc_camera_set_aperture_radius(void *s, double r) { ((Scene *)s)->camera.aperture_radius = r; }
void orc_scene_set_solid_background(void *s, const double c[3]) { ((Scene *)s)->background = Background{V3{c[0], c[1], c[2]}, V3{c[0], c[1], c[2]}, 1.0}; }
void orc_scene_set_radial_background(void *s, const double i[3], const double o[3], double scale) {
    ((Scene *)s)->background = Background{V3{i[0], i[1], i[2]}, V3{o[0], o[1], o[2]}, scale};
}
void orc_scene_set_ambient_light(void *s, const double c[3]) { ((Scene *)s)->ambient = V3{c[0], c[1], c[2]}; }
void orc_scene_set_mesh_smoothing(void *s, int e) { ((Scene *)s)->smoothing = e != 0; }
void orc_scene_set_max_recursion_depth(void *s, uint32_t d) { ((Scene *)s)->recursion = d; }
void orc_scene_set_threads(void *s, size_t t) { ((Scene *)s)->threads = t; }
void orc_scene_add_point_light(void *s, const double p[3], const double i[3], const double f[3]) {
    ((Scene *)s)->lights.push_back(PointLight{V3{p[0], p[1], p[2]}, V3{i[0], i[1], i[2]}, {f[0], f[1], f[2]}});
}
int orc_scene_parse_obj(void *s, const char *text, size_t len, uint32_t *out_ref) { // scene.rs:109-123
    Scene *sc = (Scene *)s;
    std::unique_ptr<Obj> obj(new Obj());
    std::string err;
    if (parse_obj_text(text, len, *obj, err)) { tl_error = err; return 1; }
    if (!sc->smoothing) obj->normal.clear();
    *out_ref = (uint32_t)sc->meshes.size();
    sc->meshes.push_back(std::move(obj));
    return 0;
}
int orc_scene_load_obj(void *s, const char *path, uint32_t *out_ref) {
    FILE *f = std::fopen(path, "rb");
    if (!f) { tl_error = std::string("cannot open ") + path; return 1; }
    std::string buf; char tmp[65536]; size_t n;
    while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.append(tmp, n);
    std::fclose(f);
    return orc_scene_parse_obj(s, buf.data(), buf.size(), out_ref);
}
void *orc_scene_root(void *s) { return ((Scene *)s)->root.get(); }
void orc_scene_set_root(void *s, void *agg) { ((Scene *)s)->root.reset((Aggregate *)agg); }

void *orc_aggregate_new(void) { return new Aggregate(); }
void orc_aggregate_free(void *a) { delete (Aggregate *)a; }
static SceneNode mk_node(SceneNode::Kind k) { SceneNode n; n.kind = k; n.has_mat = false; n.obj = 0; n.mat = material_default(); for (int i = 0; i < 3; ++i) { n.a[i] = 0; n.b[i] = 0; } return n; }
void orc_aggregate_add_group(void *a, void *child) { SceneNode n = mk_node(SceneNode::GROUP); n.group.reset((Aggregate *)child); ((Aggregate *)a)->contents.push_back(std::move(n)); }
void orc_aggregate_add_sphere(void *a, const double c[3], double r, const orc_material *m) {
    SceneNode n = mk_node(SceneNode::SPHERE); for (int i = 0; i < 3; ++i) n.a[i] = c[i]; n.b[0] = r; n.mat = to_mat(m); n.has_mat = true; ((Aggregate *)a)->contents.push_back(std::move(n));
}
void orc_aggregate_add_cube(void *a, const double o[3], double dim, const orc_material *m) {
    SceneNode n = mk_node(SceneNode::CUBE); for (int i = 0; i < 3; ++i) n.a[i] = o[i]; n.b[0] = dim; n.mat = to_mat(m); n.has_mat = true; ((Aggregate *)a)->contents.push_back(std::move(n));
}
void orc_aggregate_add_box(void *a, const double mn[3], const double mx[3], const orc_material *m) {
    SceneNode n = mk_node(SceneNode::CUBOID); for (int i = 0; i < 3; ++i) { n.a[i] = mn[i]; n.b[i] = mx[i]; } n.mat = to_mat(m); n.has_mat = true; ((Aggregate *)a)->contents.push_back(std::move(n));
}
void orc_aggregate_add_obj(void *a, uint32_t ref) { SceneNode n = mk_node(SceneNode::MESH); n.obj = ref; ((Aggregate *)a)->contents.push_back(std::move(n)); }
void orc_aggregate_add_obj_of(void *a, uint32_t ref, const orc_material *m) {
    SceneNode n = mk_node(SceneNode::MESH); n.obj = ref; n.mat = to_mat(m); n.has_mat = true; ((Aggregate *)a)->contents.push_back(std::move(n));
}
void orc_aggregate_swap_backface(void *a) { ((Aggregate *)a)->swap_backface = !((Aggregate *)a)->swap_backface; }
void orc_aggregate_translate(void *a, const double d[3]) { // node.rs:85-88, transform.rs:94-99
    V3 v{d[0], d[1], d[2]};
    Transform t{m4_from_translation(v), m4_from_translation(-v)};
    tr_concat_self(((Aggregate *)a)->transform, t);
}
void orc_aggregate_scale(void *a, double x, double y, double z) { // transform.rs:101-108
    Transform t{m4_from_scale(x, y, z), m4_from_scale(1.0 / x, 1.0 / y, 1.0 / z)};
    tr_concat_self(((Aggregate *)a)->transform, t);
}
void orc_aggregate_rotate_x(void *a, double th) { M4 m = m4_from_angle_x(th); Transform t{m, m4_transpose(m)}; tr_concat_self(((Aggregate *)a)->transform, t); }
void orc_aggregate_rotate_y(void *a, double th) { M4 m = m4_from_angle_y(th); Transform t{m, m4_transpose(m)}; tr_concat_self(((Aggregate *)a)->transform, t); }
void orc_aggregate_rotate_z(void *a, double th) { M4 m = m4_from_angle_z(th); Transform t{m, m4_transpose(m)}; tr_concat_self(((Aggregate *)a)->transform, t); }
void orc_aggregate_rotate(void *a, double th, const double ax[3]) {
    M4 m = m4_from_axis_angle(V3{ax[0], ax[1], ax[2]}, th); Transform t{m, m4_transpose(m)}; tr_concat_self(((Aggregate *)a)->transform, t);
}
void orc_aggregate_get_transform(void *a, double m[16], double minv[16]) { // test hook
    const Transform &t = ((Aggregate *)a)->transform;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) { m[4 * c + r] = t.m.at(c, r); minv[4 * c + r] = t.minv.at(c, r); }
}

static Img mk_img(uint32_t w, uint32_t h) { return Img{w, h, 1. / (double)w, 1. / (double)h, (double)w / (double)h}; }
void *orc_film_new(uint32_t w, uint32_t h) { Film *f = new Film(); f->img = mk_img(w, h); f->owned.assign((size_t)w * h * 4, 0); f->px = f->owned.data(); return f; }
void *orc_film_wrap(uint32_t w, uint32_t h, uint8_t *rgba) { Film *f = new Film(); f->img = mk_img(w, h); f->px = rgba; return f; }
uint8_t *orc_film_pixels(void *f) { return ((Film *)f)->px; }
uint32_t orc_film_width(void *f) { return ((Film *)f)->img.w; }
uint32_t orc_film_height(void *f) { return ((Film *)f)->img.h; }
void orc_film_free(void *f) { delete (Film *)f; }

void *orc_accel_from(const void *s) {
    try {
        Accel *a = new Accel();
        a->scene = (const Scene *)s;
        a->root = accel_from_aggregate(a->scene, a->scene->root.get());
        return a;
    } catch (const BuildError &e) { tl_error = e.msg; return nullptr; }
}
void orc_accel_free(void *a) { delete (Accel *)a; }

int orc_capture_subset(size_t k, size_t n, const void *accel, void *film) {
    if (n == 0) { tl_error = "n must be > 0"; return 1; }
    Film *f = (Film *)film;
    capture_subset_impl(k, n, *(const Accel *)accel, f->img, f->px, nullptr);
    flush_stats();
    return 0;
}
static size_t max_threads() { unsigned t = std::thread::hardware_concurrency(); return t ? t : 1; }
int orc_capture(const void *scene, void *film) { // lib.rs:55-104
    const Scene *sc = (const Scene *)scene;
    size_t barrels = sc->threads == 0 ? max_threads() : sc->threads;
    void *av = orc_accel_from(scene);
    if (!av) return 1;
    Accel *acc = (Accel *)av;
    Film *f = (Film *)film;
    std::vector<std::thread> th;
    for (size_t i = 1; i < barrels; ++i)
        th.emplace_back([=]() { capture_subset_impl(i, barrels, *acc, f->img, f->px, nullptr); flush_stats(); });
    capture_subset_impl(0, barrels, *acc, f->img, f->px, nullptr);
    flush_stats();
    for (auto &t : th) t.join();
    delete acc;
    return 0;
}
void *orc_render(const void *scene, uint32_t w, uint32_t h) { void *f = orc_film_new(w, h); if (orc_capture(scene, f)) { orc_film_free(f); return nullptr; } return f; }

// f64 radiance (pre-quantisation) for the pixels of subset (k, n); rgb has w*h*3 doubles, untouched pixels are left alone.
int orc_capture_radiance(size_t k, size_t n, const void *accel, uint32_t w, uint32_t h, double *rgb, size_t nthreads) {
    if (n == 0) { tl_error = "n must be > 0"; return 1; }
    Img img = mk_img(w, h);
    const Accel *acc = (const Accel *)accel;
    if (nthreads <= 1) { capture_subset_impl(k, n, *acc, img, nullptr, rgb); flush_stats(); return 0; }
    std::vector<std::thread> th;
    for (size_t i = 0; i < nthreads; ++i)
        th.emplace_back([=]() { capture_subset_impl(k + i * n, n * nthreads, *acc, img, nullptr, rgb); flush_stats(); });
    for (auto &t : th) t.join();
    return 0;
}
// Multi-threaded subset capture used by the timed CPU baseline: pixels {k + i*n}, split over `nthreads` threads.
int orc_capture_subset_mt(size_t k, size_t n, const void *accel, void *film, size_t nthreads) {
    if (n == 0 || nthreads == 0) { tl_error = "n and nthreads must be > 0"; return 1; }
    Film *f = (Film *)film;
    const Accel *acc = (const Accel *)accel;
    std::vector<std::thread> th;
    for (size_t i = 0; i < nthreads; ++i)
        th.emplace_back([=]() { capture_subset_impl(k + i * n, n * nthreads, *acc, f->img, f->px, nullptr); flush_stats(); });
    for (auto &t : th) t.join();
    return 0;
}
// Any list of pixel offsets (y * w + x) of a w x h film, results compact and in list order (samples and crops of
// films too large to render whole on the CPU): rgba_out[4*i..] and / or rgb_out[3*i..] for offsets[i].
// The per-pixel body is capture_subset_impl's (lib.rs:110-162).
int orc_capture_pixels(const void *accel, uint32_t w, uint32_t h, const uint64_t *offsets, size_t count, uint8_t *rgba_out, double *rgb_out, size_t nthreads) {
    const Accel *acc = (const Accel *)accel;
    const Img img = mk_img(w, h);
    const size_t area = (size_t)w * h;
    for (size_t i = 0; i < count; ++i) if (offsets[i] >= area) { tl_error = "pixel offset outside the film"; return 1; }
    if (nthreads == 0) nthreads = 1;
    auto work = [=](size_t t) {
        std::vector<Ray> samples(acc->scene->camera.num_samples());
        double weight = 1. / (double)samples.size();
        for (size_t i = t; i < count; i += nthreads) {
            size_t offset = (size_t)offsets[i];
            uint32_t x = (uint32_t)(offset % w), y = (uint32_t)(offset / w);
            acc->scene->camera.sample(x, y, img, samples.data());
            V3 color = integrate(*acc, samples.data(), samples.size(), weight);
            if (rgba_out) { uint8_t *p = rgba_out + 4 * i; p[0] = to_byte(color.x); p[1] = to_byte(color.y); p[2] = to_byte(color.z); p[3] = 255; }
            if (rgb_out) { double *r = rgb_out + 3 * i; r[0] = color.x; r[1] = color.y; r[2] = color.z; }
        }
        flush_stats();
    };
    std::vector<std::thread> th;
    for (size_t t = 0; t < nthreads; ++t) th.emplace_back(work, t);
    for (auto &t : th) t.join();
    return 0;
}
void orc_stats_reset(void) { std::lock_guard<std::mutex> g(*g_stats_mutex); g_stats_total = Stats(); tl_stats = Stats(); }
void orc_stats_read(orc_stats *o) {
    flush_stats();
    std::lock_guard<std::mutex> g(*g_stats_mutex);
    const Stats &s = g_stats_total;
    *o = orc_stats{s.primary_rays, s.shadow_rays, s.secondary_rays, s.nodes_tested, s.spheres_tested, s.cuboids_tested, s.triangles_tested, s.accel_entries, s.hits};
}

// ---- structure dump of the BVH tree (for build-parity tests) ----------------
// Pre-order walk over every BVHAccel: for each accel emits its node table and
// order[]; lets tests compare the product's host builder with the oracle's.
static void dump_accel(const BVHAccel *a, std::vector<double> &f, std::vector<int64_t> &i) {
    i.push_back((int64_t)a->nodes.size()); i.push_back((int64_t)a->order.size());
    i.push_back(a->has_material ? 1 : 0); i.push_back(a->swap_backface ? 1 : 0);
    for (const auto &n : a->nodes) {
        f.push_back(n.bounds.min.x); f.push_back(n.bounds.min.y); f.push_back(n.bounds.min.z);
        f.push_back(n.bounds.max.x); f.push_back(n.bounds.max.y); f.push_back(n.bounds.max.z);
        i.push_back(n.leaf ? 1 : 0); i.push_back(n.a); i.push_back(n.b);
    }
    for (size_t o : a->order) i.push_back((int64_t)o);
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) f.push_back(a->transform->m.at(c, r));
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) f.push_back(a->transform->minv.at(c, r));
    for (const auto &p : a->primitives) if (auto *child = dynamic_cast<const BVHAccel *>(p.get())) dump_accel(child, f, i);
}
static thread_local std::vector<double> tl_dump_f;
static thread_local std::vector<int64_t> tl_dump_i;
int orc_accel_dump(const void *accel, const double **f, size_t *nf, const int64_t **i, size_t *ni) {
    tl_dump_f.clear(); tl_dump_i.clear();
    dump_accel(((const Accel *)accel)->root.get(), tl_dump_f, tl_dump_i);
    *f = tl_dump_f.data(); *nf = tl_dump_f.size(); *i = tl_dump_i.data(); *ni = tl_dump_i.size();
    return 0;
}

// ---- known-answer hooks for the reference's 17 inline tests -----------------
// kind 0: sphere (params = cx,cy,cz,r); kind 1: cuboid (params = min xyz, max xyz);
// kind 2: every triangle of `obj_text` in TriangleIterator order (triangle.rs:425-427).
// out = { hit(0/1), t, ng.xyz, ns.xyz }
int orc_kat_intersect(int kind, const double *params, const char *obj_text, size_t obj_len, const double o[3], const double d[3], double out[8]) {
    Ray ray = ray_new(V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]});
    RayIntersection isect = isect_default();
    bool hit = false;
    if (kind == 0) {
        Sphere s; s.origin = V3{params[0], params[1], params[2]}; s.radius = params[3]; s.mat = material_default();
        hit = s.intersect(ray, isect) != nullptr;
    } else if (kind == 1) {
        Cuboid c; c.bounds = bounds_new(V3{params[0], params[1], params[2]}, V3{params[3], params[4], params[5]}); c.mat = material_default();
        hit = c.intersect(ray, isect) != nullptr;
    } else if (kind == 2) {
        Obj obj; std::string err;
        if (parse_obj_text(obj_text, obj_len, obj, err)) { tl_error = err; return 1; }
        for (size_t f = 0; f < obj.tri.size() / 3; ++f) {
            Triangle t; t.obj = &obj; t.face = (uint32_t)f;
            if (t.intersect(ray, isect)) hit = true;
        }
    } else { tl_error = "bad kind"; return 1; }
    out[0] = hit ? 1.0 : 0.0; out[1] = isect.t;
    V3 ng = isect_ng(isect), ns = isect_ns(isect);
    out[2] = ng.x; out[3] = ng.y; out[4] = ng.z; out[5] = ns.x; out[6] = ns.y; out[7] = ns.z;
    return 0;
}
// surface.rs:194-200: SurfaceInteraction::from(ray, RayIntersection::new(t, (0,0), dpdu, dpdv)).ng()
int orc_kat_surface_interaction(const double o[3], const double d[3], double t, const double dpdu[3], const double dpdv[3], double out_ng[3]) {
    Ray ray = ray_new(V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]});
    RayIntersection isect = isect_new(t, 0.0, 0.0, V3{dpdu[0], dpdu[1], dpdu[2]}, V3{dpdv[0], dpdv[1], dpdv[2]});
    SurfaceInteraction si = si_from(ray, isect);
    out_ng[0] = si.ng.x; out_ng[1] = si.ng.y; out_ng[2] = si.ng.z;
    return 0;
}
// elementary math hooks (sqrt/div/trig) so GPU arithmetic can be compared op by op
int orc_math_eval(int op, size_t n, const double *a, const double *b, double *out) {
    for (size_t i = 0; i < n; ++i) {
        switch (op) {
        case 0: out[i] = std::sqrt(a[i]); break;
        case 1: out[i] = a[i] / b[i]; break;
        case 2: out[i] = orc_sin(a[i]); break;
        case 3: out[i] = orc_cos(a[i]); break;
        case 4: out[i] = orc_atan2(a[i], b[i]); break;
        case 5: out[i] = orc_acos(a[i]); break;
        case 6: out[i] = std::fmin(a[i], b[i]); break;
        case 7: out[i] = std::fmax(a[i], b[i]); break;
        case 8: out[i] = (double)to_byte(a[i]); break;
        default: return 1;
        }
    }
    return 0;
}

} // extern "C"
