// lasgun_amd/csrc/vecmath.h -- f64 vector / matrix substrate shared by the host builder and the
// HIP kernels of the product.
//
// Numerical contract: every operation below restates the order of
// operations of the reference's math layer -- cgmath ^0.17 as used by
// /root/reference/src/space/{mod,ray,bounds,normal,transform}.rs -- so that the device
// computes bit-identical f64 values.  The whole product is compiled with -ffp-contract=off
// (hipcc contracts by default; Rust never does).
#pragma once

#include <cstdint>
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LG_HD __host__ __device__ __forceinline__
#else
#define LG_HD inline
#endif

namespace lg {

constexpr double PI = 3.14159265358979323846264338327950288;        // std::f64::consts::PI
constexpr double FRAC_1_PI = 0.318309886183790671537767526745028724; // std::f64::consts::FRAC_1_PI
constexpr double F64_MAX = 1.7976931348623157e308;

// ---- Rust float semantics ---------------------------------------------------------------
LG_HD double fmin_(double a, double b) { return fmin(a, b); } // f64::min (NaN-ignoring)
LG_HD double fmax_(double a, double b) { return fmax(a, b); }
LG_HD double bmin(double a, double b) { return a < b ? a : b; } // the local `min` of bounds.rs:171-178
LG_HD double bmax(double a, double b) { return a < b ? b : a; }
LG_HD double signum(double x) { // f64::signum
    if (x != x) return x;
    return __builtin_signbit(x) ? -1.0 : 1.0;
}
LG_HD uint32_t as_u32(double v) { // `as u32`: saturating, NaN -> 0
    if (!(v == v)) return 0u;
    if (v <= 0.0) return 0u;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}
LG_HD uint8_t as_u8(double v) {
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

// ---- Vector3 / Point3 -------------------------------------------------------------------
struct V3 {
    double x, y, z;
};
LG_HD V3 v3(double x, double y, double z) { return V3{x, y, z}; }
LG_HD double comp(const V3 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
LG_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
LG_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
LG_HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
LG_HD V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
LG_HD V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
LG_HD V3 operator/(V3 a, double s) { return V3{a.x / s, a.y / s, a.z / s}; }
LG_HD V3 mul_ew(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
LG_HD V3 div_ew(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
LG_HD double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
LG_HD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
LG_HD double magnitude2(V3 a) { return dot(a, a); }
LG_HD double magnitude(V3 a) { return sqrt(dot(a, a)); }
LG_HD V3 normalize(V3 a) { return a * (1.0 / magnitude(a)); }
LG_HD bool veq(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
LG_HD bool vne(V3 a, V3 b) { return !veq(a, b); }
LG_HD V3 vzero() { return V3{0.0, 0.0, 0.0}; }
LG_HD V3 splat(double v) { return V3{v, v, v}; }
LG_HD V3 vabs(V3 v) { return V3{fabs(v.x), fabs(v.y), fabs(v.z)}; }
LG_HD V3 vsqrt(V3 v) { return V3{sqrt(v.x), sqrt(v.y), sqrt(v.z)}; }
LG_HD V3 face_forward(V3 n, V3 v) { return dot(n, v) < 0.0 ? -n : n; } // normal.rs:37-40
LG_HD double lerp(double t, double p0, double p1) { return p0 * (1.0 - t) + p1 * t; } // space/mod.rs:28-30
LG_HD int max_dimension(V3 v) { // space/mod.rs:33-36
    if (v.x > v.y) return v.x > v.z ? 0 : 2;
    return v.y > v.z ? 1 : 2;
}
LG_HD void coordinate_system(V3 v1, V3 &v2, V3 &v3o) { // space/mod.rs:39-47
    if (fabs(v1.x) > fabs(v1.y)) v2 = V3{-v1.z, 0.0, v1.x} / sqrt(v1.x * v1.x + v1.z * v1.z);
    else v2 = V3{0.0, v1.z, -v1.y} / sqrt(v1.y * v1.y + v1.z * v1.z);
    v3o = cross(v1, v2);
}

// ---- affine part of a cgmath Matrix4 ------------------------------------------------------
// Column-major: c[col][row], rows 0..2.  Every transform the reference's API can build
// (translate / scale / rotate and their products, scene/node.rs:85-114) has the bottom row
// (+-0, +-0, +-0, 1), for which transform_point's homogeneous w is exactly 1 for finite
// inputs and the `* (1/w)` is the identity, so only the 3x4 part is kept on the device.
struct Affine {
    double c[4][3];
};
// Matrix4 * (v, 0) then truncate: ((c0*x + c1*y) + c2*z) + c3*0 per component.
LG_HD V3 xf_vector(const Affine &m, V3 v) {
    return V3{((m.c[0][0] * v.x + m.c[1][0] * v.y) + m.c[2][0] * v.z) + m.c[3][0] * 0.0,
              ((m.c[0][1] * v.x + m.c[1][1] * v.y) + m.c[2][1] * v.z) + m.c[3][1] * 0.0,
              ((m.c[0][2] * v.x + m.c[1][2] * v.y) + m.c[2][2] * v.z) + m.c[3][2] * 0.0};
}
// Matrix4 * (p, 1), w == 1.
LG_HD V3 xf_point(const Affine &m, V3 p) {
    return V3{((m.c[0][0] * p.x + m.c[1][0] * p.y) + m.c[2][0] * p.z) + m.c[3][0] * 1.0,
              ((m.c[0][1] * p.x + m.c[1][1] * p.y) + m.c[2][1] * p.z) + m.c[3][1] * 1.0,
              ((m.c[0][2] * p.x + m.c[1][2] * p.y) + m.c[2][2] * p.z) + m.c[3][2] * 1.0};
}
// transform.rs:202-209 -- rows of the transpose of the *other* matrix (m[i][j] = column i, row j)
LG_HD V3 xf_normal(const Affine &inv, V3 n) {
    return V3{inv.c[0][0] * n.x + inv.c[0][1] * n.y + inv.c[0][2] * n.z,
              inv.c[1][0] * n.x + inv.c[1][1] * n.y + inv.c[1][2] * n.z,
              inv.c[2][0] * n.x + inv.c[2][1] * n.y + inv.c[2][2] * n.z};
}

// ---- src/space/ray.rs ---------------------------------------------------------------------
struct Ray {
    V3 o, d, dinv;
};
LG_HD Ray ray_new(V3 o, V3 d) { return Ray{o, d, V3{1.0 / d.x, 1.0 / d.y, 1.0 / d.z}}; } // ray.rs:28-33
LG_HD Ray ray_to_local(const Affine &minv, const Ray &r) { // transform.rs:279-283
    return ray_new(xf_point(minv, r.o), xf_vector(minv, r.d));
}

// ---- src/core/math.rs ---------------------------------------------------------------------
// quad_roots + Sphere::intersect_t (sphere.rs:30-69) collapsed: returns t (or -inf) and `inside`.
// `a` = dot(d, d) depends on the ray alone; callers that test many spheres against one ray pass it
// in (same f64 value as recomputing it).
LG_HD double sphere_t_a(const Ray &ray, double a, V3 cen, double rad, bool &inside) {
    V3 d = ray.d;
    V3 l = ray.o - cen;
    double b = 2.0 * dot(d, l);
    double c = dot(l, l) - rad * rad;
    inside = false;
    const double NEG_INF = -INFINITY;
    if (a == 0.0) {
        if (b == 0.0) return NEG_INF; // 0 roots
        return -c / b;               // 1 root
    }
    double disc = b * b - 4.0 * a * c;
    if (disc < 0.0) return NEG_INF;
    double q = -(b + signum(b) * sqrt(disc)) / 2.0;
    double r0 = q / a;
    double r1 = (q == 0.0) ? r0 : c / q;
    double t0 = fmin_(r0, r1), t1 = fmax_(r0, r1);
    if (t0 < 0.0) { inside = true; return t1; }
    return t0;
}
LG_HD double sphere_t(const Ray &ray, V3 cen, double rad, bool &inside) { return sphere_t_a(ray, dot(ray.d, ray.d), cen, rad, inside); }

} // namespace lg
