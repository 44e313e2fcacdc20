// lasgun_amd/csrc/walk.h -- device code of the per-pixel ray-trace path (CDNA4, gfx950): primitives, the traversals, hit resolution.
//
// One lane = one pixel (an 8x8 pixel tile per 64-wide wavefront, fetched from a global tile
// counter so the persistent grid drains evenly).  Per lane: camera ray -> nested-BVH traversal
// with a short per-lane stack in LDS -> hit resolution -> Whitted shading with one any-hit
// shadow traversal per point light -> optional specular recursion through an explicit frame
// stack -> RGBA8.  All arithmetic is f64 in the reference's order of operations (vecmath.h);
// compile with -ffp-contract=off.  No MFMA: this is branchy traversal, not a contraction.
//
// Kernels:
//   trace_kernel<STATS, FAST, LDSS, PRUNE>    the whole of li() per lane (scenes with glass / mirror over big meshes, light
//                                             scenes, small films; STATS = the counting variant behind lg_capture_stats)
//   wf_trace_kernel<FAST, SHADOW, LDSS, L0, PRUNE>  wavefront pipeline, traversal only: closest hit of a level's rays (hits
//                                             compacted, frames parked) or per-light any-hit of its hits;
//                                             LDSS = scene tables resident in LDS, one 1024-lane workgroup per CU
//   wf_shade_kernel<KIND, L0>, wf_combine_kernel   radiance of a level's hits, specular children queued; levels combined bottom-up
//   queue_kernel<LDSS, PRUNE>                 every recursion level in one persistent launch (k_queue.hip)
//   trace_pixel_kernel<FAST>                  one pixel by one lane, with an event log of the walk (lg_trace_pixel)
//   kat_kernel, kat_si_kernel, math_kernel    probes behind the test hooks of the C ABI
//
// One traversal per mode: traverse_ref<LDSS, FAST, PRUNE, COUNT> (reference tree; FAST: the fast trees one node per step, an
// A/B) and traverse_fast<COUNT> (fast trees, wide records), both behind walk<>.
//
// What is restated from where (file:line under /root/reference):
//   pixel loop / quantisation   src/lib.rs:110-162, src/img.rs:56-67
//   camera rays                 src/camera.rs:113-146
//   traversal                   src/accelerators/bvh.rs:461-522, src/shape/cuboid.rs:104-121
//   sphere / box / triangle     src/shape/sphere.rs:30-123, cuboid.rs:55-102, triangle.rs:161-307
//   hit records, transforms     src/interaction/surface.rs:57-183, src/space/transform.rs:243-264
//   materials / BxDFs           src/material/*.rs, src/core/bxdf/*.rs, src/interaction/bsdf.rs:73-145
//   integrator, lights, bg      src/integrate/integrate.rs:16-132, src/light/point.rs:42-54,
//                               src/material/background.rs:25-34
//
// Legal restructurings (each leaves every produced f64 bit-identical):
//   * traversal only tracks (t, primitive, accel) of the best hit; dpdu/dpdv/normals of the
//     WINNING primitive are computed once afterwards (the reference overwrites them on every
//     closer hit, so only the last accepted ones survive: sphere.rs:120, bvh.rs:510);
//   * shadow rays stop at the first accepted hit with t < 1 (point.rs:49 only tests isect.t < 1.0
//     and t only ever decreases);
//   * a lane's visit order is exactly the reference's (near child first by dir_is_neg[axis],
//     leaf primitives in order[]), which is what breaks ties between equal t.
#pragma once
#include "kcommon.h"

namespace lg {

// ------------------------------------------------------------------------------------------
// primitives
// ------------------------------------------------------------------------------------------
// Bounds::intersects (cuboid.rs:104-121): slab test, fmin/fmax absorb the NaN of 0*inf.
__device__ __forceinline__ bool slab_intersects(const double bmin[3], const double bmax[3], const Ray &r) {
    double t1 = (bmin[0] - r.o.x) * r.dinv.x, t2 = (bmax[0] - r.o.x) * r.dinv.x;
    double tnear = fmax_(-INFINITY, fmin_(t1, t2));
    double tfar = fmin_(INFINITY, fmax_(t1, t2));
    t1 = (bmin[1] - r.o.y) * r.dinv.y; t2 = (bmax[1] - r.o.y) * r.dinv.y;
    tnear = fmax_(tnear, fmin_(t1, t2));
    tfar = fmin_(tfar, fmax_(t1, t2));
    t1 = (bmin[2] - r.o.z) * r.dinv.z; t2 = (bmax[2] - r.o.z) * r.dinv.z;
    tnear = fmax_(tnear, fmin_(t1, t2));
    tfar = fmin_(tfar, fmax_(t1, t2));
    return tnear <= tfar && tfar > 0.0;
}

// The same test without the two clamps.  fmax / fmin ignore a NaN operand, so the chains above yield the largest / smallest
// non-NaN term, with -inf / +inf standing in when every term is NaN (0 * inf on all three axes).  fmin(t1, t2) is NaN exactly
// when fmax(t1, t2) is, so tnear is NaN exactly when tfar is; the clamped test then reads -inf <= +inf && +inf > 0 = true, and
// the negated comparisons below read !(NaN > NaN) && !(NaN <= 0) = true as well; on numbers they are the same comparisons.
__device__ __forceinline__ bool slab_intersects_nc(const double bmin[3], const double bmax[3], const Ray &r) {
    double t1 = (bmin[0] - r.o.x) * r.dinv.x, t2 = (bmax[0] - r.o.x) * r.dinv.x;
    double tnear = fmin_(t1, t2), tfar = fmax_(t1, t2);
    t1 = (bmin[1] - r.o.y) * r.dinv.y; t2 = (bmax[1] - r.o.y) * r.dinv.y;
    tnear = fmax_(tnear, fmin_(t1, t2));
    tfar = fmin_(tfar, fmax_(t1, t2));
    t1 = (bmin[2] - r.o.z) * r.dinv.z; t2 = (bmax[2] - r.o.z) * r.dinv.z;
    tnear = fmax_(tnear, fmin_(t1, t2));
    tfar = fmin_(tfar, fmax_(t1, t2));
    return !(tnear > tfar) && !(tfar <= 0.0);
}

__device__ __forceinline__ bool slab_intersects_nc_t(const double bmin[3], const double bmax[3], const Ray &r, double &tnear_out, double &tfar_out) {
    double t1 = (bmin[0] - r.o.x) * r.dinv.x, t2 = (bmax[0] - r.o.x) * r.dinv.x;
    double tnear = fmin_(t1, t2), tfar = fmax_(t1, t2);
    t1 = (bmin[1] - r.o.y) * r.dinv.y; t2 = (bmax[1] - r.o.y) * r.dinv.y;
    tnear = fmax_(tnear, fmin_(t1, t2));
    tfar = fmin_(tfar, fmax_(t1, t2));
    t1 = (bmin[2] - r.o.z) * r.dinv.z; t2 = (bmax[2] - r.o.z) * r.dinv.z;
    tnear = fmax_(tnear, fmin_(t1, t2));
    tfar = fmin_(tfar, fmax_(t1, t2));
    tnear_out = tnear; tfar_out = tfar;
    return !(tnear > tfar) && !(tfar <= 0.0);
}

// slab_intersects_nc, also handing back the three per-axis ENTRY parameters min(t1, t2) (the pruned walk compares each with its
// own limit).  fmin ignores a NaN operand and yields NaN only when both are NaN (0 * inf twice: never for a box with min < max).
__device__ __forceinline__ bool slab_intersects_nc_axes(const double bmin[3], const double bmax[3], const Ray &r, double &tx, double &ty, double &tz) {
    double t1 = (bmin[0] - r.o.x) * r.dinv.x, t2 = (bmax[0] - r.o.x) * r.dinv.x;
    tx = fmin_(t1, t2);
    double tnear = tx, tfar = fmax_(t1, t2);
    t1 = (bmin[1] - r.o.y) * r.dinv.y; t2 = (bmax[1] - r.o.y) * r.dinv.y;
    ty = fmin_(t1, t2);
    tnear = fmax_(tnear, ty);
    tfar = fmin_(tfar, fmax_(t1, t2));
    t1 = (bmin[2] - r.o.z) * r.dinv.z; t2 = (bmax[2] - r.o.z) * r.dinv.z;
    tz = fmin_(t1, t2);
    tnear = fmax_(tnear, tz);
    tfar = fmin_(tfar, fmax_(t1, t2));
    return !(tnear > tfar) && !(tfar <= 0.0);
}

// slab_intersects_nc for a ray whose sign triple is known at compile time (SG: bit a set <=> dinv[a] < 0) -- the "plain" case:
// every box coordinate finite and every box ordered, bmin <= bmax (the host checks the scene: DParams::boxes_finite), the ray's origin finite, dinv finite and not
// zero on every axis (ray_signs_plain).  Then fl(bmin - o) <= fl(bmax - o) (rounding is monotone; both may overflow to the same
// infinity, neither is NaN), and multiplying by a finite non-zero dinv keeps (dinv > 0) or reverses (dinv < 0) that order, again
// without a NaN (inf * 0 needs dinv = 0): min(t1, t2) and max(t1, t2) of cuboid.rs:113-117 ARE (t1, t2) or (t2, t1), up to the
// sign of a zero when both are zeros -- and tnear / tfar only ever meet comparisons, which do not see that sign.  Six of the
// ten min / max of a node step become register names.  Also hands back the per-axis entry parameters (the pruned walk's).
template <int SG>
__device__ __forceinline__ bool slab_intersects_sg(const double bmin[3], const double bmax[3], const Ray &r, double &tx, double &ty, double &tz) {
    tx = (((SG & 1) ? bmax[0] : bmin[0]) - r.o.x) * r.dinv.x;
    double tfar = (((SG & 1) ? bmin[0] : bmax[0]) - r.o.x) * r.dinv.x;
    ty = (((SG & 2) ? bmax[1] : bmin[1]) - r.o.y) * r.dinv.y;
    const double fy = (((SG & 2) ? bmin[1] : bmax[1]) - r.o.y) * r.dinv.y;
    double tnear = fmax_(tx, ty);
    tfar = fmin_(tfar, fy);
    tz = (((SG & 4) ? bmax[2] : bmin[2]) - r.o.z) * r.dinv.z;
    const double fz = (((SG & 4) ? bmin[2] : bmax[2]) - r.o.z) * r.dinv.z;
    tnear = fmax_(tnear, tz);
    tfar = fmin_(tfar, fz);
    return !(tnear > tfar) && !(tfar <= 0.0);
}

// the same test, also handing back its tnear (used by the fast mode's front-to-back pruning)
__device__ __forceinline__ bool slab_intersects_t(const double bmin[3], const double bmax[3], const Ray &r, double &tnear_out, double &tfar_out) {
    double t1 = (bmin[0] - r.o.x) * r.dinv.x, t2 = (bmax[0] - r.o.x) * r.dinv.x;
    double tnear = fmax_(-INFINITY, fmin_(t1, t2));
    double tfar = fmin_(INFINITY, fmax_(t1, t2));
    t1 = (bmin[1] - r.o.y) * r.dinv.y; t2 = (bmax[1] - r.o.y) * r.dinv.y;
    tnear = fmax_(tnear, fmin_(t1, t2));
    tfar = fmin_(tfar, fmax_(t1, t2));
    t1 = (bmin[2] - r.o.z) * r.dinv.z; t2 = (bmax[2] - r.o.z) * r.dinv.z;
    tnear = fmax_(tnear, fmin_(t1, t2));
    tfar = fmin_(tfar, fmax_(t1, t2));
    tnear_out = tnear; tfar_out = tfar;
    return tnear <= tfar && tfar > 0.0;
}

__device__ __forceinline__ V3 cube_diff(int axis, int which) { // CUBE_DIFFERENTIALS cuboid.rs:126-130
    // axis 0: (y, z)   axis 1: (z, x)   axis 2: (x, y)
    int a = which == 0 ? (axis + 1) % 3 : (axis + 2) % 3;
    return V3{a == 0 ? 1.0 : 0.0, a == 1 ? 1.0 : 0.0, a == 2 ? 1.0 : 0.0};
}

// Bounds::intersect (cuboid.rs:55-102).  Returns false on a miss; on a hit t is the cuboid's t
// (NOT yet compared with the current best).  With FULL also the differentials of the hit face.
template <bool FULL>
__device__ __forceinline__ bool cuboid_hit(const double mn[3], const double mx[3], const Ray &r, double &t, V3 &d0, V3 &d1) {
    double tnear = -INFINITY, tfar = INFINITY;
    // codes: axis*2 + flipped  (flipped: the pair is (dp.1, dp.0))
    int near_code = 0, far_code = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double o = comp(r.o, i), di = comp(r.dinv, i);
        double t1 = (mn[i] - o) * di;
        double t2 = (mx[i] - o) * di;
        double tmin, tmax;
        bool lt = t1 < t2;
        if (lt) { tmin = t1; tmax = t2; } else { tmin = t2; tmax = t1; }
        if (FULL) {
            // (dp0, dp1) = lt ? (dp.1, dp.0) : (dp.0, dp.1); near = (dp0, dp1); far = (dp1, dp0)
            if (tmin > tnear) near_code = i * 2 + (lt ? 1 : 0);
            if (tmax < tfar) far_code = i * 2 + (lt ? 0 : 1);
        }
        tnear = fmax_(tnear, tmin);
        tfar = fmin_(tfar, tmax);
    }
    if (tnear > tfar || tfar <= 0.0) return false;
    int code;
    if (tnear <= 0.0) { t = tfar; code = far_code; } else { t = tnear; code = near_code; }
    if (FULL) {
        int axis = code >> 1, flipped = code & 1;
        d0 = cube_diff(axis, flipped ? 1 : 0);
        d1 = cube_diff(axis, flipped ? 0 : 1);
    }
    return true;
}

struct TriHit {
    double t, b0, b1, b2;
};
// Triangle::intersect up to the `t >= isect.t` test (triangle.rs:161-251).
__device__ __forceinline__ bool triangle_t(V3 p0, V3 p1, V3 p2, const Ray &ray, TriHit &h) {
    V3 p0t = p0 - ray.o, p1t = p1 - ray.o, p2t = p2 - ray.o;
    int kz = max_dimension(vabs(ray.d));
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    V3 d{comp(ray.d, kx), comp(ray.d, ky), comp(ray.d, kz)};
    p0t = V3{comp(p0t, kx), comp(p0t, ky), comp(p0t, kz)};
    p1t = V3{comp(p1t, kx), comp(p1t, ky), comp(p1t, kz)};
    p2t = V3{comp(p2t, kx), comp(p2t, ky), comp(p2t, kz)};
    double sx = -d.x / d.z, sy = -d.y / d.z, sz = 1.0 / d.z;
    p0t.x += sx * p0t.z; p0t.y += sy * p0t.z;
    p1t.x += sx * p1t.z; p1t.y += sy * p1t.z;
    p2t.x += sx * p2t.z; p2t.y += sy * p2t.z;
    double e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    double e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    double e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if ((e0 < 0.0 || e1 < 0.0 || e2 < 0.0) && (e0 > 0.0 || e1 > 0.0 || e2 > 0.0)) return false;
    double det = e0 + e1 + e2;
    if (det == 0.0) return false;
    p0t.z *= sz; p1t.z *= sz; p2t.z *= sz;
    double tscaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if ((det < 0.0 && tscaled >= 0.0) || (det > 0.0 && tscaled <= 0.0)) return false;
    double invdet = 1.0 / det;
    h.b0 = e0 * invdet; h.b1 = e1 * invdet; h.b2 = e2 * invdet;
    h.t = tscaled * invdet;
    return true;
}

// The ray-only part of Triangle::intersect (triangle.rs:186-201): permutation and shear
// constants.  They depend on the ray alone, so they are computed once per mesh-accel entry
// instead of once per triangle; the values are the same f64s the reference recomputes.
struct TriSetup {
    int kz;
    double sx, sy, sz;
};
__device__ __forceinline__ TriSetup tri_setup(const Ray &ray) {
    TriSetup s;
    s.kz = max_dimension(vabs(ray.d));
    int kx = s.kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    double dx = comp(ray.d, kx), dy = comp(ray.d, ky), dz = comp(ray.d, s.kz);
    s.sx = -dx / dz; s.sy = -dy / dz;
    s.sz = comp(ray.dinv, s.kz); // 1.0 / dz: the quotient Ray::new already formed (ray.rs:28-33; every Ray here comes from ray_new)
    return s;
}
template <int KZ> __device__ __forceinline__ V3 permute_kz(V3 v) { // (kx, ky, kz) = (KZ+1, KZ+2, KZ) mod 3
    if (KZ == 0) return V3{v.y, v.z, v.x};
    if (KZ == 1) return V3{v.z, v.x, v.y};
    return v;
}
// triangle_t with the setup hoisted and the permutation resolved at compile time
template <int KZ>
__device__ __forceinline__ bool triangle_t_pre(V3 p0, V3 p1, V3 p2, V3 o, double sx, double sy, double sz, TriHit &h) {
    V3 p0t = permute_kz<KZ>(p0 - o), p1t = permute_kz<KZ>(p1 - o), p2t = permute_kz<KZ>(p2 - o);
    p0t.x += sx * p0t.z; p0t.y += sy * p0t.z;
    p1t.x += sx * p1t.z; p1t.y += sy * p1t.z;
    p2t.x += sx * p2t.z; p2t.y += sy * p2t.z;
    double e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    double e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    double e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if ((e0 < 0.0 || e1 < 0.0 || e2 < 0.0) && (e0 > 0.0 || e1 > 0.0 || e2 > 0.0)) return false;
    double det = e0 + e1 + e2;
    if (det == 0.0) return false;
    p0t.z *= sz; p1t.z *= sz; p2t.z *= sz;
    double tscaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if ((det < 0.0 && tscaled >= 0.0) || (det > 0.0 && tscaled <= 0.0)) return false;
    double invdet = 1.0 / det;
    h.b0 = e0 * invdet; h.b1 = e1 * invdet; h.b2 = e2 * invdet;
    h.t = tscaled * invdet;
    return true;
}

__device__ __forceinline__ V3 load_f3(const float *base, uint32_t idx) {
    const float *p = base + 3ull * idx;
    return V3{(double)p[0], (double)p[1], (double)p[2]};
}

// ------------------------------------------------------------------------------------------
// hit record (surface.rs:33-119)
// ------------------------------------------------------------------------------------------
struct Isect {
    double t;
    V3 gu, gv; // geometry dpdu / dpdv
    V3 su, sv; // surface dpdu / dpdv
    V3 n;
    bool has_n;
};
__device__ __forceinline__ void isect_set(Isect &i, double t, V3 dpdu, V3 dpdv) { // RayIntersection::new
    i.t = t; i.gu = dpdu; i.gv = dpdv; i.su = dpdu; i.sv = dpdv; i.has_n = false; i.n = vzero();
}

// Sphere::intersect (sphere.rs:79-123), for an accepted t
// (as a function call instead of inline code -- measured in round 6 -- every kernel loses 10-25 %: the call's register convention spills the walk's state)
__device__ __forceinline__ void sphere_full(const DSphere &s, const Ray &ray, double t, bool inside, Isect &is) {
    V3 cen{s.cx, s.cy, s.cz};
    V3 p = ray.o + ray.d * t - cen;
    if (p.x == 0.0 && p.y == 0.0) p.x = 1e-5 * s.r;
    double phi = p_atan2(p.y, p.x);
    if (phi < 0.0) phi += 2.0 * PI;
    double theta = p_acos(fmin_(fmax_(p.z / s.r, -1.0), 1.0));
    V3 dpdu{-2.0 * PI * p.y, 2.0 * PI * p.x, 0.0};
    double sin_phi, cos_phi;
    // ONE inlined instance of the double-double sincos for phi and theta -- a two-trip loop -- instead of two (p_sin(theta) is the sine that
    // p_sincos(theta) returns: one function).  Measured, variants in turn (profiles/r06_ab_trig_shared.jsonl): the megakernel of the mesh configs
    // compiles better around one instance -- config 4 33.58 -> 33.30 ms, 4m 13.22 -> 12.47 (the old 3-ulp algorithm: 33.11 / 12.59) -- at
    // +1.5 % on simple.rs 9 spp and nothing on the headline.
    double sin_theta = 0.0;
    sin_phi = 0.0; cos_phi = 0.0;
#pragma clang loop unroll(disable)
    for (int i = 0; i < 2; ++i) {
        double sn, cs;
        p_sincos(i == 0 ? phi : theta, sn, cs);
        if (i == 0) { sin_phi = sn; cos_phi = cs; } else sin_theta = sn;
    }
    V3 dpdv = PI * V3{p.z * cos_phi, p.z * sin_phi, -s.r * sin_theta};
    if (inside) isect_set(is, t, dpdu, dpdv);
    else isect_set(is, t, dpdv, dpdu);
}

// Triangle::intersect from the partial derivatives on (triangle.rs:257-304)
__device__ __forceinline__ void triangle_full(const DParams &P, uint32_t tri, uint32_t aflags, const Ray &ray, Isect &is) {
    const uint32_t *vi = P.tri_v + 3ull * tri;
    V3 p0 = load_f3(P.vpos, vi[0]), p1 = load_f3(P.vpos, vi[1]), p2 = load_f3(P.vpos, vi[2]);
    TriHit h;
    triangle_t(p0, p1, p2, ray, h);
    double uv[3][2];
    if (aflags & AF_HAS_UV) {
        const uint32_t *ti = P.tri_t + 3ull * tri;
#pragma unroll
        for (int k = 0; k < 3; ++k) { uv[k][0] = (double)P.vtex[2ull * ti[k]]; uv[k][1] = (double)P.vtex[2ull * ti[k] + 1]; }
    } else {
        uv[0][0] = 0.0; uv[0][1] = 0.0; uv[1][0] = 1.0; uv[1][1] = 0.0; uv[2][0] = 1.0; uv[2][1] = 1.0;
    }
    double duv02x = uv[0][0] - uv[2][0], duv02y = uv[0][1] - uv[2][1];
    double duv12x = uv[1][0] - uv[2][0], duv12y = uv[1][1] - uv[2][1];
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
    isect_set(is, h.t, dpdu, dpdv);
    if (aflags & AF_HAS_N) {
        const uint32_t *ni = P.tri_n + 3ull * tri;
        V3 n0 = load_f3(P.vnorm, ni[0]), n1 = load_f3(P.vnorm, ni[1]), n2 = load_f3(P.vnorm, ni[2]);
        V3 ns = h.b0 * n0 + h.b1 * n1 + h.b2 * n2;
        V3 ss = is.gu;
        V3 ts = cross(ns, ss);
        if (magnitude2(ts) > 0.0) ss = cross(ts, ns);
        else coordinate_system(ns, ss, ts);
        is.has_n = true; is.n = ns;
        is.su = ss; is.sv = ts;
    } else {
        is.has_n = true;
        is.n = face_forward(cross(dp02, dp12), -ray.d);
    }
}

// ------------------------------------------------------------------------------------------
// traversal
// ------------------------------------------------------------------------------------------
// event log of one traced ray (lg_trace_pixel, counting instantiations only): code, a, b, c
__device__ __forceinline__ void dbg_event(const DParams &P, double code, double a, double b, double c) {
    if (!P.dbg_log) return;
    const uint32_t n = (uint32_t)P.dbg_log[0];
    if (n >= 4000u) return;
    double *e = P.dbg_log + 1 + 4 * (size_t)n;
    e[0] = code; e[1] = a; e[2] = b; e[3] = c;
    P.dbg_log[0] = (double)(n + 1u);
}
struct Counters {
    uint32_t primary, shadow, secondary, nodes, spheres, cuboids, triangles, entries, hits;
    // audit of the pruned walk (counting instantiations with DParams::audit; see audit_prim): nodes / runs it skipped although the
    // reference's own box test passed, primitives below them put to the reference's tests, those the reference would have ACCEPTED
    // (violations of property (P), DESIGN.md 3.4: must be 0), and the smallest (t - limit) / margin over the others
    uint32_t a_nodes = 0u, a_runs = 0u, a_prims = 0u, a_viol = 0u;
    double a_slack_n = INFINITY, a_slack_r = INFINITY;
    double a_used_n = 0.0; // nodes: the largest (slab entry parameter - t) / margin over the skipped primitives: the share of the margin (P) needed
};

__device__ __forceinline__ Affine load_affine(const Affine *p) { return *p; }

// ray in the local space of `accel`: apply minv of every accel on the chain root..accel, in
// order, exactly as the nested BVHAccel::intersect calls do (bvh.rs:462).
__device__ __forceinline__ Ray local_ray(const DParams &P, const Ray &wray, uint32_t accel) {
    const DAccel *a = P.accels + accel;
    uint32_t n = a->nchain;
    Ray r = wray;
    for (uint32_t i = 0; i < n; ++i) r = ray_to_local(P.accels[a->chain[i]].minv, r);
    return r;
}

// The same record through the SCALAR cache: for an address every active lane shares (the caller checks).  The scene tables are
// never written while a kernel runs, so the constant address space's promise holds; one 48- or 64-byte scalar fetch then stands
// for 64 lanes' fetches of the same line through the vector L1.
typedef uint32_t lg_u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) lg_u32x4 *lg_const_u4;
__device__ __forceinline__ uint4 load_const_u4(lg_const_u4 q, int i) {
    const lg_u32x4 v = q[i];
    return uint4{v.x, v.y, v.z, v.w};
}
// The accel records in LDS (`arec` below), read through a pointer that SAYS it is LDS.  A kernel that may find them in LDS or in the DAccel
// table (the 256-lane forms: `arec` is a run-time value) otherwise gets both alternatives merged into one generic pointer and read with
// flat_load -- which waits on the vector-memory AND the LDS counters and takes the long way round (round 4, read in the ISA: every
// field of an accel record on entering and leaving a nested accel went that way).
typedef const __attribute__((address_space(3))) lg_u32x4 *lg_lds_u4;
typedef double lg_f64x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) lg_f64x2 *lg_lds_d2;
typedef const __attribute__((address_space(3))) uint32_t *lg_lds_u32;
__device__ __forceinline__ uint4 lds_u4(const uint4 *p) {
    const lg_u32x4 v = *(lg_lds_u4)p;
    return uint4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ double2 lds_d2(const void *p) {
    const lg_f64x2 v = *(lg_lds_d2)p;
    return double2{v.x, v.y};
}
// ---- scene table access: HBM/L2 tables, or (LDSS) the copy a 1024-lane workgroup holds in LDS.
// Lanes of a wave read DIFFERENT records, 56 bytes per node visit: through the vector L1 that is
// 64 B/clk per CU and the traversal kernels were bound by it as much as by VALU issue; the LDS
// delivers 256 B/clk per CU at a third of the latency.  Image: nodes padded to 80 B, primrefs, and one 48-byte
// leaf record per primref slot (sphere c, r / cuboid min, max / triangle positions): 16 consecutive records of
// either kind start in 16 different bank groups.
struct NodeRec {
    double bmin[3], bmax[3];
    uint32_t link, meta, parent, pad;
};
__device__ __forceinline__ double u2d(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }
template <bool LDSS>
__device__ __forceinline__ NodeRec load_node(const DParams &P, const uint4 *scn, uint32_t idx) {
    NodeRec n;
    if (LDSS) {
        static_assert(LDS_NODE_STRIDE == 5u, "idx * 5 below is written as a shift and an add");
        const uint4 *q4 = scn + (P.lds_node_off + ((idx << 2) + idx)); // no quarter-rate 32-bit multiply
        const double2 *q = reinterpret_cast<const double2 *>(q4);
        const double2 a = q[0], b = q[1], c = q[2]; // three ds_read_b128
        const uint2 d = *reinterpret_cast<const uint2 *>(q4 + 3);
        n.bmin[0] = a.x; n.bmin[1] = a.y; n.bmin[2] = b.x;
        n.bmax[0] = b.y; n.bmax[1] = c.x; n.bmax[2] = c.y;
        n.link = d.x; n.meta = d.y; n.parent = 0u; n.pad = 0u;
    } else {
        // one 64-byte record = four 16-byte loads from a single line, all issued before the slab test
        const DNode *nd = P.nodes + idx;
        n.bmin[0] = nd->bmin[0]; n.bmin[1] = nd->bmin[1]; n.bmin[2] = nd->bmin[2];
        n.bmax[0] = nd->bmax[0]; n.bmax[1] = nd->bmax[1]; n.bmax[2] = nd->bmax[2];
        n.link = nd->link; n.meta = nd->meta; n.parent = nd->parent; n.pad = nd->pad;
    }
    return n;
}
__device__ __forceinline__ NodeRec load_node_uniform(const DParams &P, uint32_t idx) { // (idx: the same for every active lane)
    lg_const_u4 q = (lg_const_u4)(uintptr_t)(P.nodes + idx);
    const uint4 a = load_const_u4(q, 0), b = load_const_u4(q, 1), c = load_const_u4(q, 2), d = load_const_u4(q, 3);
    NodeRec n;
    n.bmin[0] = u2d(a.x, a.y); n.bmin[1] = u2d(a.z, a.w); n.bmin[2] = u2d(b.x, b.y);
    n.bmax[0] = u2d(b.z, b.w); n.bmax[1] = u2d(c.x, c.y); n.bmax[2] = u2d(c.z, c.w);
    n.link = d.x; n.meta = d.y; n.parent = d.z; n.pad = d.w;
    return n;
}
template <bool LDSS>
__device__ __forceinline__ uint32_t load_primref(const DParams &P, const uint4 *scn, uint32_t i) {
    if (LDSS) return reinterpret_cast<const uint32_t *>(scn + P.lds_prim_off)[i];
    return P.primref[i];
}
// LDSS: the geometry comes from the leaf-ordered record of the SLOT (same index as its primref), not from the
// per-kind table: the primref and its geometry are fetched side by side instead of one after the other
template <bool LDSS>
__device__ __forceinline__ DSphere load_sphere(const DParams &P, const uint4 *scn, uint32_t idx, uint32_t slot) {
    if (LDSS) {
        const uint4 *q = scn + (P.lds_soup_off + slot * 3u);
        uint4 a = q[0], b = q[1];
        return DSphere{u2d(a.x, a.y), u2d(a.z, a.w), u2d(b.x, b.y), u2d(b.z, b.w)};
    }
    return P.spheres[idx];
}
template <bool LDSS>
__device__ __forceinline__ DCuboid load_cuboid(const DParams &P, const uint4 *scn, uint32_t idx, uint32_t slot) {
    if (LDSS) {
        const uint4 *q = scn + (P.lds_soup_off + slot * 3u);
        uint4 a = q[0], b = q[1], c = q[2];
        DCuboid cb;
        cb.mn[0] = u2d(a.x, a.y); cb.mn[1] = u2d(a.z, a.w); cb.mn[2] = u2d(b.x, b.y);
        cb.mx[0] = u2d(b.z, b.w); cb.mx[1] = u2d(c.x, c.y); cb.mx[2] = u2d(c.z, c.w);
        return cb;
    }
    return P.cuboids[idx];
}

// the same ray from the ROOT accel's local ray (chain[0] already applied): the continuation of the same
// sequence of transforms, so bit-identical to local_ray() -- and one transform cheaper per call
__device__ __forceinline__ Ray level_ray(const DParams &P, V3 root_o, V3 root_d, uint32_t accel) {
    const DAccel *a = P.accels + accel;
    uint32_t n = a->nchain;
    Ray r = ray_new(root_o, root_d);
    for (uint32_t i = 1; i < n; ++i) r = ray_to_local(P.accels[a->chain[i]].minv, r);
    return r;
}

struct Best {
    double t;
    uint32_t ref;   // primref of the closest accepted primitive, NO_HIT if none
    uint32_t accel; // accel instance it was hit in
};

// ---- fast mode's candidate check ----------------------------------------------------------------
// The fast tree finds primitives quickly, but WHICH primitives a ray is tested against is the reference
// tree's decision: the reference tests a primitive iff every box from its root down to the primitive's
// leaf passes the slab test (and likewise for every nested accel on the way up), and near a box face that
// decision is made by the last bit.  The fast walk's WINNER therefore counts only after the same boxes
// have been put to the same test, leaf to root, level by level: one check per ray with a hit, after the
// walk.  A winner that fails it -- like an exact tie -- sends the ray to the reference walk.  (If the
// winner passes, it is the reference's winner: the fast walk tests every primitive the reference tests
// and hits -- its boxes are the same primitive boxes, pushed out by 1e-9 of the accel's extent -- so
// nothing the reference accepts is closer, and the winner is one of the reference's candidates.)
__device__ __forceinline__ bool ref_path_hit(const DParams &P, uint32_t node_base, uint32_t leaf, const Ray &ray) {
    uint32_t n = leaf;
    if (n == NO_HIT) return false; // a primitive beyond its leaf's u16 count: the reference never reaches it
    for (;;) {
        // (neighbouring rays mostly hit primitives of the same reference leaf and then climb the same nodes: one scalar fetch)
        const uint32_t i = node_base + n, i0 = __builtin_amdgcn_readfirstlane(i);
        NodeRec nd;
        if (__builtin_amdgcn_ballot_w64(i != i0) == 0ull) nd = load_node_uniform(P, i0);
        else nd = load_node<false>(P, nullptr, i);
        n = nd.parent; // same 64-byte record: one fetch per step
        if (!slab_intersects(nd.bmin, nd.bmax, ray)) return false;
        if (n == NO_HIT) return true;
    }
}
__device__ __forceinline__ bool ref_candidate(const DParams &P, const Ray &wray, const Best &best) {
    const uint32_t kind = best.ref >> 30, idx = best.ref & PRIM_INDEX_MASK;
    const uint32_t leaf = kind == PK_SPHERE ? P.sphere_ref_leaf[idx] : kind == PK_CUBOID ? P.cuboid_ref_leaf[idx] : P.tri_ref_leaf[idx];
    uint32_t a = best.accel;
    if (!ref_path_hit(P, P.accels[a].node_base, leaf, local_ray(P, wray, a))) return false;
    while (a != 0u) {
        const uint32_t parent = (uint32_t)P.accels[a].parent;
        if (!ref_path_hit(P, P.accels[parent].node_base, P.accel_ref_leaf[a], local_ray(P, wray, parent))) return false;
        a = parent;
    }
    return true;
}

// leaf-ordered 48-byte geometry records: three 16-byte loads per slot
struct LeafRec {
    uint4 a, b, c;
};
__device__ __forceinline__ LeafRec load_rec(const DParams &P, uint32_t slot) {
    const uint4 *q = reinterpret_cast<const uint4 *>(P.leaf_soup + slot);
    return LeafRec{q[0], q[1], q[2]};
}
__device__ __forceinline__ double rec_f64(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }
__device__ __forceinline__ double rec_f32(uint32_t w) { return (double)__uint_as_float(w); } // f32 -> f64 `.into()`

// ---- fast mode (opt-in; NOT the reference's traversal): the limit beyond which its walk skips a node's subtree
__device__ __forceinline__ double prune_limit(double tbest, bool anyhit) {
#ifdef LG_FAST_NOPRUNE
    return INFINITY;
#endif
    double lim = anyhit ? 1.0 : tbest;
    return lim + 1e-5 * (fabs(lim) + 1.0); // +inf stays +inf
}


// ------------------------------------------------------------------------------------------
// BVHAccel::intersect over the whole nested scene graph (bvh.rs:461-522), one lane = one ray.
// `stack` is this lane's LDS stack: entry i lives at stack[i * stride].  A lane's own visit sequence (near child
// first by dir_is_neg[axis], leaf primitives in order[], nested accels entered in place) is exactly the reference's,
// which is what decides ties between equal t.
//
// How the divergence is written down matters (round 1 kept the per-lane state in a handful of bools and nested
// per-lane loops; hipcc turned each into lane masks in SGPR pairs merged with s_and / s_andn2 / s_or triplets: 87
// scalar and 86 vector instructions per node or primitive step, of which 26 are the slab test).  Here
//   * a lane's phase is ONE integer (ST_NODE / ST_LEAF / ST_ENTER / ST_LEVEL_DONE / ST_DONE);
//   * every loop is WAVE-UNIFORM (`while (any lane is in this phase)`: one ballot and one scalar branch per
//     trip) around a flat predicated step;
//   * the node step has no branch at all: the far child is stored above the stack top unconditionally (it
//     only counts if sp advances), the entry below the top is fetched at the top of the step together with
//     the node record (so a pop costs no extra LDS round trip), and near / far / pop are selects;
//   * the ROOT accel's ray stays in registers, and entering an accel whose inverse transform is exactly the
//     identity (every mesh: BVHAccel::from_mesh uses transform::ID, bvh.rs:147) keeps the ray as it is when
//     all six components are finite and not -0 -- ((1*x + 0*y) + 0*z) + 0*w is then x, bit for bit -- and
//     the matching return keeps it too; other returns recompute the parent's ray from the root's, through
//     the same sequence of transforms (bit-identical, as before).
// ------------------------------------------------------------------------------------------
enum : uint32_t { ST_NODE = 0u, ST_LEAF = 1u, ST_LEVEL_DONE = 2u, ST_DONE = 3u, ST_ENTER = 4u };
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// finite and not -0: not one of sNaN, qNaN, -inf, -0, +inf (v_cmp_class_f64)
__device__ __forceinline__ bool f64_plain(double x) { return !__builtin_amdgcn_class(x, 0x001 | 0x002 | 0x004 | 0x020 | 0x200); }
__device__ __forceinline__ bool ray_plain(const Ray &r) {
    return f64_plain(r.o.x) && f64_plain(r.o.y) && f64_plain(r.o.z) && f64_plain(r.d.x) && f64_plain(r.d.y) && f64_plain(r.d.z);
}
__device__ __forceinline__ uint32_t neg_mask(const Ray &r) {
    return (r.dinv.x < 0.0 ? 1u : 0u) | (r.dinv.y < 0.0 ? 2u : 0u) | (r.dinv.z < 0.0 ? 4u : 0u);
}
// neg_mask, with bit 3 set when the ray is not "plain" in the sense of slab_intersects_sg: origin finite, dinv finite and non-zero
// (v_cmp_class: normal or subnormal of either sign)
constexpr uint32_t SIGNS_NOT_PLAIN = 8u;
__device__ __forceinline__ uint32_t neg_mask_x(const Ray &r) {
    const bool plain = __builtin_amdgcn_class(r.dinv.x, 0x008 | 0x010 | 0x080 | 0x100) && __builtin_amdgcn_class(r.dinv.y, 0x008 | 0x010 | 0x080 | 0x100) &&
                       __builtin_amdgcn_class(r.dinv.z, 0x008 | 0x010 | 0x080 | 0x100) && __builtin_amdgcn_class(r.o.x, 0x1F8) &&
                       __builtin_amdgcn_class(r.o.y, 0x1F8) && __builtin_amdgcn_class(r.o.z, 0x1F8);
    return neg_mask(r) | (plain ? 0u : SIGNS_NOT_PLAIN);
}
template <int N> struct IntC { static constexpr int value = N; };
constexpr uint32_t FRAME_SAME_RAY = 0x80000000u; // level frame, third word: the level was entered without changing the ray

struct Lvl { // the accel level a lane is walking
    uint32_t accel, node_base, prim_base, soup_delta, flags;
};
// Node cursor of the second formulation.  Global tables: the node's index in P.nodes (64-byte DNode records).
// LDS image: the node's BYTE offset in the image -- every record of the image carries "walk words" made by the host
// (accel.cpp, the image builder): an interior node its second child's cursor and 1 << axis, a leaf its first slot, NODE_LEAF
// and its last slot + 1 -- so a step forms the record's address with one add, never multiplies or shifts, and takes a
// leaf's slot range as it is.
constexpr uint32_t LDS_NODE_BYTES = LDS_NODE_STRIDE * 16u;
constexpr uint32_t LDS_NODE_WALK_OFF = 64u; // words 16..19 of the 80-byte record
// What the walk needs of a DAccel: from the LDS image (LDS_ACCEL_UNITS) or from the table in HBM / L2
// `arec`: the accel records in LDS (LDS_ACCEL_UNITS units each) -- part of the LDS-resident scene image (LDSS), or, for a scene whose
// tables stay in L2, the small image of just these records that the 256-lane kernels copy in behind their stacks (DParams::
// accel_image; its unit [6] carries global node / primref bases); nullptr: the DAccel table in HBM / L2.  Entering and leaving
// nested accels is a chain of DEPENDENT fetches of these fields: from L2 that was 37 % of the walk's cycles on config 4m.
template <bool LDSS, bool FAST = false>
__device__ __forceinline__ void lvl_set(const DParams &P, const uint4 *arec, Lvl &L, uint32_t accel) {
    L.accel = accel;
    if (LDSS || (!FAST && arec)) {
        const uint4 info = lds_u4(arec + (accel * LDS_ACCEL_UNITS + 6u));
        L.node_base = info.x; L.prim_base = info.y; L.soup_delta = info.z; L.flags = info.w;
    } else {
        const DAccel *A = P.accels + accel;
        L.node_base = FAST ? A->fnode_base : A->node_base; L.prim_base = FAST ? A->fprim_base : A->prim_base; L.soup_delta = 0u; L.flags = A->flags;
    }
}
template <bool LDSS>
__device__ __forceinline__ Ray accel_local_ray(const DParams &P, const uint4 *arec, uint32_t accel, const Ray &r) { // inverse_transform_ray (bvh.rs:462)
    if (LDSS || arec) {
        const uint4 *q = arec + accel * LDS_ACCEL_UNITS;
        const double2 a = lds_d2(q), b = lds_d2(q + 1), c = lds_d2(q + 2), d = lds_d2(q + 3), e = lds_d2(q + 4), f = lds_d2(q + 5);
        Affine m;
        m.c[0][0] = a.x; m.c[0][1] = a.y; m.c[0][2] = b.x; m.c[1][0] = b.y; m.c[1][1] = c.x; m.c[1][2] = c.y;
        m.c[2][0] = d.x; m.c[2][1] = d.y; m.c[2][2] = e.x; m.c[3][0] = e.y; m.c[3][1] = f.x; m.c[3][2] = f.y;
        return ray_to_local(m, r);
    }
    return ray_to_local(P.accels[accel].minv, r);
}
// One fat mesh leaf [li, le) of the second formulation: the reference's leaf loop (bvh.rs:483-488) over the leaf-ordered
// 48-byte f32 position records, two triangles per trip in ping-pong (while one record is tested the next is in flight and
// neither is ever copied), one address add per triangle; returns true when an any-hit ray is done.
// (Records pre-widened to f64 -- 80 bytes, nine conversions fewer per triangle -- were measured again this round: 138 -> 170 ms
// on config 4.  Lanes of incoherent rays read different triangles and the loop then waits on the vector L1, not on the VALU.)
// The sign test of the edge functions (triangle.rs:224-230) is written with v_cmp_class: "negative" = -normal, -subnormal,
// -inf and "positive" likewise -- exactly `e < 0.0` / `e > 0.0` (zeros and NaNs are neither) -- which keeps hipcc from
// turning the six comparisons into a min / max chain with canonicalising moves.
__device__ __forceinline__ bool f64_neg(double x) { return __builtin_amdgcn_class(x, 0x004 | 0x008 | 0x010); }
__device__ __forceinline__ bool f64_pos(double x) { return __builtin_amdgcn_class(x, 0x080 | 0x100 | 0x200); }
__device__ __forceinline__ LeafRec load_rec_at(const char *base, uint32_t off) {
    const uint4 *q = reinterpret_cast<const uint4 *>(base + off);
    return LeafRec{q[0], q[1], q[2]};
}
__device__ __forceinline__ LeafRec load_rec_uniform(const char *base, uint32_t off) {
    lg_const_u4 q = (lg_const_u4)(uintptr_t)(base + off);
    return LeafRec{load_const_u4(q, 0), load_const_u4(q, 1), load_const_u4(q, 2)};
}
template <int KZ>
__device__ __forceinline__ bool tri_rec_t(const LeafRec &r, V3 o, double sx, double sy, double sz, TriHit &h) {
    const V3 p0{rec_f32(r.a.x), rec_f32(r.a.y), rec_f32(r.a.z)}, p1{rec_f32(r.a.w), rec_f32(r.b.x), rec_f32(r.b.y)},
        p2{rec_f32(r.b.z), rec_f32(r.b.w), rec_f32(r.c.x)};
    V3 p0t = permute_kz<KZ>(p0 - o), p1t = permute_kz<KZ>(p1 - o), p2t = permute_kz<KZ>(p2 - o);
    p0t.x += sx * p0t.z; p0t.y += sy * p0t.z;
    p1t.x += sx * p1t.z; p1t.y += sy * p1t.z;
    p2t.x += sx * p2t.z; p2t.y += sy * p2t.z;
    double e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    double e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    double e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if ((f64_neg(e0) || f64_neg(e1) || f64_neg(e2)) && (f64_pos(e0) || f64_pos(e1) || f64_pos(e2))) return false;
    double det = e0 + e1 + e2;
    if (det == 0.0) return false;
    p0t.z *= sz; p1t.z *= sz; p2t.z *= sz;
    double tscaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if ((det < 0.0 && tscaled >= 0.0) || (det > 0.0 && tscaled <= 0.0)) return false;
    double invdet = 1.0 / det;
    h.b0 = e0 * invdet; h.b1 = e1 * invdet; h.b2 = e2 * invdet;
    h.t = tscaled * invdet;
    return true;
}
// What the pruned walk knows when it opens a fat leaf: the limit (already scaled by 1 + PRUNE_LIMIT_REL), the dominant axis's
// margin in t (eps * |1/d_kz|) and d . d.
struct LeafCull {
    double lb, ekz, dd;
    uint32_t records;      // DNode::pad of the leaf: first record | number of records << 24
    float dhx, dhy, dhz;   // the ray's direction over its length, in f32, every rounding towards a shorter vector (per leaf, not per record)
};
// One culling record (DChunk) against the ray: true = none of its triangles (<= 32 for a run) can be accepted, the tests are skipped.
// (t1, t2) per axis are the slab test's own expressions on the record's box.  Dominant axis: an accepted t is a convex
// combination of plane parameters that lie between the box's two (section 3.4), so it lies in [tmin_kz - ekz, tmax_kz + ekz]: outside
// [0, limit] nothing is accepted.  All axes, when the ray meets every triangle of the record at an angle of sine >= sigma > 0
// and the triangles are not degenerate at its distance: the accepted hit point lies within m = CHUNK_K0 * R^2 * g2 / sigma^3 of the
// triangle (the edge functions' rounding, 48 u R^2, moves the projected origin by at most that over an altitude), so t lies in
// every axis's [tmin_i - m |1/d_i|, tmax_i + m |1/d_i|]: an empty intersection, one beyond the limit or one before 0 accepts nothing.
struct ChunkRec { uint4 a, b, c, d; }; // one DChunk as loaded: four 16-byte words of one line
__device__ __forceinline__ ChunkRec load_chunk(const DChunk *rec) {
    const uint4 *q = reinterpret_cast<const uint4 *>(rec);
    return ChunkRec{q[0], q[1], q[2], q[3]};
}
__device__ __forceinline__ ChunkRec load_chunk_uniform(const DChunk *rec) { // (see load_rec_uniform)
    lg_const_u4 q = (lg_const_u4)(uintptr_t)rec;
    return ChunkRec{load_const_u4(q, 0), load_const_u4(q, 1), load_const_u4(q, 2), load_const_u4(q, 3)};
}
template <int KZ>
__device__ __forceinline__ bool chunk_culled(const ChunkRec &k, const Ray &ray, const LeafCull &lc, uint32_t &run_start, uint32_t &run_count ) {
    const uint4 a = k.a, b = k.b, c = k.c;
    run_start = k.d.x; run_count = k.d.y;
    const double bmin[3] = {rec_f32(a.x), rec_f32(a.y), rec_f32(a.z)}, bmax[3] = {rec_f32(a.w), rec_f32(b.x), rec_f32(b.y)};
    const float cos_t = __uint_as_float(c.y), g2 = __uint_as_float(c.z), hmin = __uint_as_float(c.w), sin_t = __uint_as_float(k.d.z);
    const double ox = bmin[0] - ray.o.x, px = bmax[0] - ray.o.x, oy = bmin[1] - ray.o.y, py = bmax[1] - ray.o.y, oz = bmin[2] - ray.o.z, pz = bmax[2] - ray.o.z;
    double t1 = ox * ray.dinv.x, t2 = px * ray.dinv.x;
    const double nx = fmin_(t1, t2), fx = fmax_(t1, t2);
    t1 = oy * ray.dinv.y; t2 = py * ray.dinv.y;
    const double ny = fmin_(t1, t2), fy = fmax_(t1, t2);
    t1 = oz * ray.dinv.z; t2 = pz * ray.dinv.z;
    const double nz = fmin_(t1, t2), fz = fmax_(t1, t2);
    const double nk = KZ == 0 ? nx : KZ == 1 ? ny : nz, fk = KZ == 0 ? fx : KZ == 1 ? fy : fz;
    bool skip = nk > lc.lb + lc.ekz || fk < -lc.ekz; // (NaN parameters compare false)
    // sigma: a lower bound of |n . d| / |d| over the record's triangles -- cos(alpha + theta) with cos(alpha) = |axis . d| / |d|, in
    // f32 with every rounding pushed towards a smaller sigma (a larger margin)
    // (round 4: the unit direction is the leaf's -- LeafCull -- not the record's: three conversions and an f64 dot product less per record;
    // the f32 dot product's own rounding, ~2e-7, sits inside the 4e-6: config 4m 14.7 -> 14.2 ms, 5 54.2 -> 53.8.  The level's S in place of
    // the record's own R was measured too: 5 % more triangle tests -- the margin grows with R^2 -- and no time gained.)
#ifdef LG_DIR_PER_RECORD // (k_queue.hip: three more registers live across the leaf loop cost that kernel more than the conversions -- config 4 36.9 -> 37.5 ms)
    const V3 ax{rec_f32(b.z), rec_f32(b.w), rec_f32(c.x)};
    const float ca = fminf(fabsf((float)dot(ax, ray.d)) * __frsqrt_rn((float)lc.dd) * (1.0f - 4e-6f), 1.0f);
#else
    const float ca = fminf(fabsf((__uint_as_float(b.z) * lc.dhx + __uint_as_float(b.w) * lc.dhy) + __uint_as_float(c.x) * lc.dhz), 1.0f);
#endif
    // (v_sqrt_f32 and v_rcp_f32 as the hardware has them -- 1 ulp, 1.2e-7 -- instead of the correctly rounded library forms, ~18 and ~12
    // instructions each: both results are padded by 1e-6 and more towards a smaller sigma; a denormal argument of the root comes back as
    // 0 or as itself, and the + 1e-6 covers sqrt(1.2e-38))
#ifdef LG_LIB_SQRT // (A/B: the library forms)
    const float sa = sqrtf(fmaxf(1.0f - ca * ca, 0.0f)) * (1.0f + 4e-6f) + 1e-6f;
#else
    const float sa = __builtin_amdgcn_sqrtf(fmaxf(1.0f - ca * ca, 0.0f)) * (1.0f + 4e-6f) + 1e-6f;
#endif
    const float sigma = (ca * cos_t - sa * sin_t) - 1e-5f;
    const float Rf = (float)((fmax_(fabs(ox), fabs(px)) + fmax_(fabs(oy), fabs(py))) + fmax_(fabs(oz), fabs(pz))) * (1.0f + 1e-6f); // >= the 1-norm distance to any vertex
    if (sigma >= CHUNK_SIGMA_MIN && hmin * hmin * sigma >= CHUNK_HGATE * Rf * Rf) {
#ifdef LG_LIB_SQRT
        const float inv = __frcp_rn(sigma) * (1.0f + 1e-6f);
#else
        const float inv = __builtin_amdgcn_rcpf(sigma) * (1.0f + 1e-6f);
#endif
        const double m = (double)(((CHUNK_K0 * g2) * (inv * inv * inv)) * (Rf * Rf) * (1.0f + 1e-5f));
        const double ex = m * fabs(ray.dinv.x), ey = m * fabs(ray.dinv.y), ez = m * fabs(ray.dinv.z);
        const double tn = fmax_(fmax_(nx - ex, ny - ey), nz - ez), tf = fmin_(fmin_(fx + ex, fy + ey), fz + ez); // (inf - inf = NaN: ignored)
        skip = skip || tn > tf || tf < 0.0 || tn > lc.lb;
    }
    return skip;
}
// ---- audit of the pruned walk (diagnostic, counting instantiations only; lg_audit_prune) -----------------------------------------
// Property (P) of DESIGN.md 3.4 on real data: a primitive under a node or in a run the pruned walk SKIPS must be one the reference
// would have rejected at that moment -- `t >= isect.t` for a closest-hit ray (an exact tie inside the same fat leaf goes to the
// lower original slot, as in mesh_leaf2), t >= 1 for a shadow ray (point.rs:49).  One that would have been accepted is a
// violation; for the others (t - limit) / margin says how much of the shipped margin was needed.
__device__ __forceinline__ void audit_prim(bool valid, double t, double limit, double margin, bool anyhit, bool tie_wins, Counters &cnt, double &slack) {
    if (!valid) return;
    cnt.a_prims++;
    const bool accepted = anyhit ? !(t >= 1.0) : (!(t >= limit) || tie_wins);
    if (accepted) { cnt.a_viol++; return; }
    const double s = (t - limit) / margin;
    if (s < slack) slack = s; // (margin == 0 or inf: +inf / 0 / NaN never lower the minimum)
}
// every primitive the reference reaches below node `top` of level L (its own box has passed the reference's test): no nested
// accel can be among them (NODE_NOPRUNE keeps such nodes from being skipped)
// `entry` / `emargin`: the skipped node's slab entry parameter and margin on each axis that said "beyond" (margin +inf elsewhere): the
// rule relies on t >= entry_i - emargin_i for every primitive below the node; (entry_i - t) / emargin_i is the share of the margin a
// primitive actually needed (<= 0: none; >= 1 would be a violation of (P))
__device__ __noinline__ void audit_subtree(const DParams &P, uint32_t node_base, uint32_t prim_base, uint32_t top, const Ray &ray, double dd, double four_a,
                                           double limit, double margin, bool anyhit, Counters &cnt, V3 entry, V3 emargin) {
    uint32_t st[64];
    int sp = 0;
    uint32_t n = top;
    cnt.a_nodes++;
    for (;;) {
        const DNode *nd = P.nodes + n;
        const bool h = n == top || slab_intersects_nc(nd->bmin, nd->bmax, ray); // cuboid.rs:104-121, as the reference walks it
        if (h) {
            if (nd->meta & NODE_LEAF) {
                const uint32_t first = prim_base + nd->link, count = nd->meta & 0xFFFFu;
                for (uint32_t slot = first; slot < first + count; ++slot) {
                    const uint32_t ref = P.primref[slot], kind = ref >> 30;
                    const uint4 *q = reinterpret_cast<const uint4 *>(P.leaf_soup + slot);
                    const LeafRec g{q[0], q[1], q[2]};
                    bool valid = false;
                    double t = 0.0;
                    if (kind == PK_SPHERE) { // as in traverse_ref
                        const V3 cen{rec_f64(g.a.x, g.a.y), rec_f64(g.a.z, g.a.w), rec_f64(g.b.x, g.b.y)};
                        const V3 l = ray.o - cen;
                        const double b = 2.0 * dot(ray.d, l), c = dot(l, l) - rec_f64(g.c.x, g.c.y);
                        if (dd == 0.0) { if (b != 0.0) { t = -c / b; valid = true; } }
                        else {
                            const double disc = b * b - four_a * c;
                            if (!(disc < 0.0)) {
                                const double qq = -(b + signum(b) * sqrt(disc)) / 2.0;
                                const double r0 = qq / dd, r1 = (qq == 0.0) ? r0 : c / qq;
                                const double t0 = fmin_(r0, r1), t1 = fmax_(r0, r1);
                                t = t0 < 0.0 ? t1 : t0;
                                valid = true;
                            }
                        }
                        valid = valid && !(t < 0.0);
                    } else if (kind == PK_CUBOID) {
                        double mn[3] = {rec_f64(g.a.x, g.a.y), rec_f64(g.a.z, g.a.w), rec_f64(g.b.x, g.b.y)};
                        double mx[3] = {rec_f64(g.b.z, g.b.w), rec_f64(g.c.x, g.c.y), rec_f64(g.c.z, g.c.w)};
                        V3 d0, d1;
                        valid = cuboid_hit<false>(mn, mx, ray, t, d0, d1);
                    } else if (kind == PK_TRIANGLE) {
                        const V3 p0{rec_f32(g.a.x), rec_f32(g.a.y), rec_f32(g.a.z)}, p1{rec_f32(g.a.w), rec_f32(g.b.x), rec_f32(g.b.y)},
                            p2{rec_f32(g.b.z), rec_f32(g.b.w), rec_f32(g.c.x)};
                        TriHit hh;
                        valid = triangle_t(p0, p1, p2, ray, hh);
                        t = hh.t;
                    } else { cnt.a_viol++; } // a nested accel below a skipped node: the host's NODE_NOPRUNE marking failed
                    audit_prim(valid, t, limit, margin, anyhit, false, cnt, cnt.a_slack_n);
                    if (valid) {
                        const double used = fmax_(fmax_((entry.x - t) / emargin.x, (entry.y - t) / emargin.y), (entry.z - t) / emargin.z); // (x / inf = 0)
                        if (used > cnt.a_used_n) cnt.a_used_n = used;
                    }
                }
            } else {
                if (sp == 64) { cnt.a_viol++; break; } // deeper than the audit's own stack: reported as a violation, never an overrun (the walk's depth is data-dependent)
                st[sp++] = node_base + nd->link; // second child
                n = n + 1u;                      // first child
                continue;
            }
        }
        if (sp == 0) break;
        n = st[--sp];
    }
}
// the triangles of the strip entries [e0, e1) (a culled run, or the runs of a culled group) against the ray, the reference's way: each
// from its own record in leaf_soup
template <int KZ>
__device__ __noinline__ void audit_run(const DParams &P, uint32_t e0, uint32_t e1, V3 o, TriSetup tri, double best_t, uint32_t best_ref, uint32_t leaf_slot,
                                       double ekz, bool anyhit, Counters &cnt) {
    cnt.a_runs++;
    for (uint32_t e = e0; e < e1; ++e) {
        const uint32_t code = P.strips[e].code;
        if (!(code & STRIP_TRI)) continue;
        const uint32_t slot = code & STRIP_SLOT_MASK;
        const uint4 *q = reinterpret_cast<const uint4 *>(P.leaf_soup + slot);
        const LeafRec r{q[0], q[1], q[2]};
        TriHit hh;
        const bool valid = tri_rec_t<KZ>(r, o, tri.sx, tri.sy, tri.sz, hh);
        const bool tie_wins = valid && hh.t == best_t && best_ref != NO_HIT && leaf_slot != NO_HIT && slot < leaf_slot;
        audit_prim(valid, hh.t, anyhit ? 1.0 : best_t, ekz, anyhit, tie_wins, cnt, cnt.a_slack_r);
    }
}

template <int KZ, bool LDSS, bool FAST = false, bool COUNT = false, bool PRUNE = false>
__device__ __forceinline__ bool mesh_leaf2(const DParams &P, const uint4 *scn, const Ray &ray, const TriSetup tri, uint32_t li, const uint32_t le,
                                           const uint32_t soup_delta, const uint32_t accel, const bool anyhit, Best &best, bool &tie, Counters &cnt,
                                           LeafCull lc) {
    const V3 o = ray.o;
    const char *base = reinterpret_cast<const char *>(P.leaf_soup);
    constexpr uint32_t REC = (uint32_t)sizeof(DLeafRec);
#define LG_TRI(R, SLOT)                                                                                                  \
    do {                                                                                                                 \
        TriHit h_;                                                                                                       \
        if (COUNT) cnt.triangles++;                                                                                      \
        if (tri_rec_t<KZ>(R, o, tri.sx, tri.sy, tri.sz, h_)) {                                                           \
            if (FAST && ((h_.t == best.t && best.ref != NO_HIT) || h_.t != h_.t)) tie = true; /* visit order decides */  \
            if (!(h_.t >= best.t)) {                                                                                     \
                best.t = h_.t; best.ref = load_primref<LDSS>(P, scn, (SLOT)); best.accel = accel;                        \
                if (COUNT) dbg_event(P, 6.0, (double)best.ref, h_.t, (double)accel);                                     \
                if (anyhit && h_.t < 1.0) return true; /* point.rs:49 */                                                 \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)
    if (PRUNE && lc.ekz < INFINITY && (lc.records >> 24) != 0u) { // (a level or ray outside the stated ranges, or a leaf the host gave no records: the plain loop below, in the reference's order)
        // The leaf in runs of <= 32 triangles that are neighbours in space (one culling record per run; host.cpp, build_chunks): a run
        // whose record is culled is stepped over.  The reference scans the leaf in order[] sequence and keeps the FIRST of several
        // triangles with exactly the same t (triangle.rs:251: `t >= isect.t` rejects); scanning in another order gives the same
        // winner when a tie goes to the lower original slot -- the final hit is the lexicographic minimum of (t, slot) either way.
        // (No t is NaN inside the stated ranges.)
        //
        // A kept run is read as TRIANGLE STRIPS (dscene.h, DStrip): one entry per new vertex.  Per entry the vertex is transformed as
        // triangle.rs:186-201 transforms it (translate, permute, shear in x and y) and the edge function across it and its predecessor
        // formed (triangle.rs:204-206: a.x * b.y - a.y * b.x); a triangle entry forms the third edge function and asks the reference's
        // first question -- are the three of mixed sign (triangle.rs:224-230)?  The three values are the doubles the reference would
        // compute for that triangle, or all three negated (cross(b, a) = 0 - cross(a, b) bit for bit, zeros staying zeros), in
        // some rotation: the answer is the same.  Only a triangle that passes (about one in thirty) is put to the reference's whole
        // formula, from its own 48-byte record, after the run (at most one is parked per lane; a second one flushes the first).
        const char *sbase = reinterpret_cast<const char *>(P.strips);
        const char *rbase = reinterpret_cast<const char *>(P.leaf_soup);
        constexpr uint32_t SREC = (uint32_t)sizeof(DStrip);
        uint32_t rec = lc.records & 0x00FFFFFFu;                 // the next record to look at
        const uint32_t rec_end = rec + (lc.records >> 24);
        uint32_t s = 0, run_end = 0;                              // the run being tested: strip entries [s, run_end)
        uint32_t leaf_slot = NO_HIT; // original slot of the hit this leaf has given so far
        uint32_t pend = NO_HIT;      // a triangle that passed the sign test and waits for the reference's whole formula
        double ax = 0.0, ay = 0.0, bx = 0.0, by = 0.0, cab = 0.0; // the last two transformed vertices of the strip, the edge function across them
        bool done = false;
#define LG_TRI_FULL(SLOT)                                                                                                \
    do {                                                                                                                 \
        const uint32_t from_ = (SLOT);                                                                                   \
        const LeafRec r_ = load_rec_at(rbase, from_ * REC);                                                              \
        TriHit h_;                                                                                                       \
        if (tri_rec_t<KZ>(r_, o, tri.sx, tri.sy, tri.sz, h_)) {                                                          \
            if (h_.t < best.t || (h_.t == best.t && leaf_slot != NO_HIT && from_ < leaf_slot)) {                         \
                best.t = h_.t; best.ref = load_primref<LDSS>(P, scn, from_ - soup_delta); best.accel = accel;            \
                leaf_slot = from_;                                                                                       \
                if (!anyhit) lc.lb = h_.t + h_.t * PRUNE_LIMIT_REL;                                                      \
                if (COUNT) dbg_event(P, 6.0, (double)best.ref, h_.t, (double)accel);                                     \
                if (anyhit && h_.t < 1.0) done = true; /* point.rs:49 */                                                 \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)
#define LG_STRIP(E)                                                                                                      \
    do {                                                                                                                 \
        const V3 pt_ = permute_kz<KZ>(V3{rec_f32((E).x), rec_f32((E).y), rec_f32((E).z)} - o);                           \
        const double vx_ = pt_.x + tri.sx * pt_.z, vy_ = pt_.y + tri.sy * pt_.z;                                         \
        const double cbc_ = bx * vy_ - by * vx_;                                                                         \
        if ((E).w & STRIP_TRI) {                                                                                         \
            const double cca_ = vx_ * ay - vy_ * ax;                                                                     \
            if (COUNT) cnt.triangles++;                                                                                  \
            if (!((f64_neg(cab) || f64_neg(cbc_) || f64_neg(cca_)) && (f64_pos(cab) || f64_pos(cbc_) || f64_pos(cca_)))) { \
                if (pend != NO_HIT) LG_TRI_FULL(pend);                                                                   \
                pend = (E).w & STRIP_SLOT_MASK;                                                                          \
            }                                                                                                            \
        }                                                                                                                \
        ax = bx; ay = by; bx = vx_; by = vy_; cab = cbc_;                                                                \
    } while (0)
        // Two wave-uniform phases, like the walk itself: every lane first steps over culled runs until it stands in one that
        // survives (or its leaf ends), then the lanes that stand in a run test it, two entries per trip.  A lane's own loop nest
        // would make the whole wave pay for every run that ANY lane keeps.
        // When every lane that takes a step stands at the same record (the lanes of a coherent wave in the same leaf mostly do),
        // the record comes through the scalar cache (load_*_uniform) instead of 64 times through the vector L1.
        bool in_run = false;
        uint32_t off = 0;
#ifdef LG_STAMPS
        unsigned long long ml_cnt[6] = {0, 0, 0, 0, 1ull, (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true))};
#endif
        for (;;) {
            bool seeks = !in_run && rec < rec_end; // (carried like at_node_l in traverse_ref)
            bool seeking = wave_any(seeks);
            while (seeking) {
#ifdef LG_STAMPS
                ml_cnt[0] += 1; ml_cnt[1] += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(!in_run && rec < rec_end));
#endif
                if (seeks) {
                    if (COUNT) cnt.nodes++; // (a record test is counted with the node tests)
                    uint32_t start, count;
                    const uint32_t rec0 = __builtin_amdgcn_readfirstlane(rec);
                    ChunkRec ck;
                    bool culled_;
#ifdef LG_NO_UNIFORM_DUP
                    if (__builtin_amdgcn_ballot_w64(rec != rec0) == 0ull) ck = load_chunk_uniform(P.chunks + rec0);
                    else ck = load_chunk(P.chunks + rec);
                    culled_ = chunk_culled<KZ>(ck, ray, lc, start, count);
#else
                    // (the test is written out once per load path: merged behind the two loads, the scalar path's sixteen words are first
                    // copied into vector registers -- 16 of the trip's ~125 vector instructions; here its arithmetic reads them as they are)
                    if (__builtin_amdgcn_ballot_w64(rec != rec0) == 0ull) { ck = load_chunk_uniform(P.chunks + rec0); culled_ = chunk_culled<KZ>(ck, ray, lc, start, count); }
                    else { ck = load_chunk(P.chunks + rec); culled_ = chunk_culled<KZ>(ck, ray, lc, start, count); }
#endif
                    if (COUNT && P.audit && culled_) { // what the skipped run (or the runs of the skipped group) would have given the reference
                        uint32_t e0 = ck.d.w, e1 = ck.d.w + (count >> 8);
                        if (start == CHUNK_IS_GROUP) { e0 = P.chunks[rec + 1u].pad; e1 = P.chunks[rec + count].pad + (P.chunks[rec + count].count >> 8); }
                        audit_run<KZ>(P, e0, e1, o, tri, best.t, best.ref, leaf_slot, lc.ekz, anyhit, cnt);
                    }
                    ++rec;
                    if (start == CHUNK_IS_GROUP) { if (culled_) rec += count; } // a group record: culled, its runs are stepped over; kept, they come next
                    else if (!culled_) { in_run = true; s = ck.d.w; run_end = s + (count >> 8); off = s * SREC; } // (< 2^32 / 16 entries: checked by the host)
                }
                seeks = !in_run && rec < rec_end;
                seeking = wave_any(seeks);
            }
            bool testing = wave_any(in_run);
            if (!testing) break;
            while (testing) {
#ifdef LG_STAMPS
                ml_cnt[2] += 1; ml_cnt[3] += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(in_run));
#endif
                if (in_run) {
                    const bool two = s + 1u < run_end;
                    const uint32_t off0 = __builtin_amdgcn_readfirstlane(off);
                    uint4 ea, eb; // two entries per trip (two spare entries behind the last: always readable)
#ifdef LG_STRIP_DUP
                    if (__builtin_amdgcn_ballot_w64(off != off0) == 0ull) {
                        lg_const_u4 q = (lg_const_u4)(uintptr_t)(sbase + off0);
                        ea = load_const_u4(q, 0); eb = load_const_u4(q, 1);
                        LG_STRIP(ea);
                        if (two) LG_STRIP(eb);
                    } else {
                        const uint4 *q = reinterpret_cast<const uint4 *>(sbase + off);
                        ea = q[0]; eb = q[1];
                        LG_STRIP(ea);
                        if (two) LG_STRIP(eb);
                    }
#else
                    if (__builtin_amdgcn_ballot_w64(off != off0) == 0ull) {
                        lg_const_u4 q = (lg_const_u4)(uintptr_t)(sbase + off0);
                        ea = load_const_u4(q, 0); eb = load_const_u4(q, 1);
                    } else {
                        const uint4 *q = reinterpret_cast<const uint4 *>(sbase + off);
                        ea = q[0]; eb = q[1];
                    }
                    LG_STRIP(ea);
                    if (two) LG_STRIP(eb);
#endif
                    s += two ? 2u : 1u;
                    off += 2u * SREC;
                    if (s >= run_end) in_run = false;
                }
                testing = wave_any(in_run);
            }
            // the triangles that passed the sign test: the reference's whole formula, once per lane and round
            if (wave_any(pend != NO_HIT)) {
                if (pend != NO_HIT) { LG_TRI_FULL(pend); pend = NO_HIT; }
            }
            if (done) rec = rec_end; // an occluded any-hit ray: nothing more to find
        }
#ifdef LG_STAMPS
        if (P.stamp_counts && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true)))
            for (int i = 0; i < 6; ++i) atomicAdd(P.stamp_counts + 9 + i, ml_cnt[i]);
#endif
        return done;
#undef LG_STRIP
#undef LG_TRI_FULL
        return false;
    }
    uint32_t off = (li + soup_delta) * REC; // (the array holds < 2^32 / 48 slots: checked by the host)
    LeafRec ra = load_rec_at(base, off);
    for (; li + 1u < le; li += 2u) {
        const LeafRec rb = load_rec_at(base, off + REC);
        LG_TRI(ra, li);
        off += 2u * REC;
        ra = load_rec_at(base, off); // (two spare records behind the last slot: always readable)
        LG_TRI(rb, li + 1u);
    }
    if (li < le) LG_TRI(ra, li);
#undef LG_TRI
    return false;
}

// Diagnostic build (-DLG_STAMPS, never shipped): cycles a wave spends in each phase of the walk, summed into P.stats
// (nine 64-bit words: setup, A nodes, B mesh leaves, B leaf slots, enter, C returns, trips, -, -), and wave-level trip
// counts and lane sums (lanes stepping / lanes already done, per node trip and per leaf-slot trip) into P.stamp_counts.  tools/stamp_phases.py reads them.
#ifdef LG_STAMPS
#define LG_STAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); stamp_acc[i] += now_ - stamp_t; stamp_t = now_; } while (0)
#else
#define LG_STAMP(i) do { } while (0)
#endif
// FAST (lg_accel_set_mode(1), opt-in, NOT the reference's traversal): the same walk over the binned-SAH trees with one primitive
// per leaf, near child first by dir_is_neg[axis] as before, and a node is skipped when its slab tnear lies beyond the best hit so
// far (closest) or beyond the light (any-hit) -- margins as in prune_limit().  Exact ties in t (and NaN t), where the reference's
// visit order decides, raise `tie`; the caller (walk() below) then puts the winner to the reference tree's own box tests
// (ref_candidate) and re-traces with the reference walk when either fails.
//
// PRUNE (the reference tree, the reference's visit order; DESIGN.md section 3.4 has the derivations): a node is skipped when,
// on some axis, the ray enters its slab only at a parameter beyond the limit -- the best accepted t so far (closest hit) or 1
// (any-hit) -- by more than that axis's margin.  Every primitive below such a node would be rejected by the reference's own
// `t >= isect.t` (sphere.rs:86, cuboid.rs:95, triangle.rs:251) or could not bring isect.t below 1 (point.rs:49), so the lane's
// sequence of accepted hits is the reference's.  What makes that a statement about COMPUTED values: an accepted hit point
// o + t*d lies within eps = e0 + S*(PRUNE_E1 + e2*S) of the primitive's bounds box, S = |o - centre|_1 + size of the level --
// for a sphere because the computed root satisfies the sphere's equation to 114 u W^2, for a box because t IS one of its plane
// parameters, for a triangle on the ray's dominant axis kz only (its t is a convex combination of the vertices' plane
// parameters along kz, computed by the slab test's own expression; the other axes promise nothing for a triangle seen edge-on)
// -- hence t >= (entry parameter on the axis) - eps * |1/d_axis|.  Nodes over a nested accel are never skipped (NODE_NOPRUNE),
// levels or rays outside the stated magnitude range are walked unpruned (eps = +inf).
// COUNT: the counting instantiation (lg_capture_stats, lg_trace_pixel): the same walk, plus the deterministic work
// counters and, for lg_trace_pixel, an event log -- 2.x node tested (.1 = taken), 3.x primitive tested (.1 = accepted),
// 4 accel entered, 5 returned to the parent, 6 triangle accepted.
// REFILL (round 6; the headline's shadow pass, k_wavefront.hip): the walk as a PERSISTENT one.  A lane whose ray is done does not wait for
// the wave's slowest lane: when at least RF::threshold() lanes are done, each of them hands its result to the caller's `rf` and is given
// its next ray (rf.next(done, ...), called by EVERY lane in wave-uniform control flow: same lane, same stack, a fresh walk), until the caller has none left.  Which lane walks which ray never
// changes a result -- a lane's visit sequence is its ray's alone.  `live`: the lane starts with a ray (every lane of the wave enters the
// walk, so that idle ones can be given work inside it).
struct NoRefill {
    static constexpr bool enabled = false;
    __device__ __forceinline__ uint32_t threshold() const { return 65u; }
    __device__ __forceinline__ bool next(bool, const Best &, Ray &) { return false; }
};
template <bool LDSS, bool FAST = false, bool PRUNE = false, bool COUNT = false, class RF = NoRefill>
__device__ __forceinline__ void traverse_ref(const DParams &P, const Ray &wray, const bool anyhit, uint32_t *stack, const uint32_t stride,
                                             Best &best, const uint4 *scn, bool &tie, Counters &cnt, const uint4 *arec_in = nullptr,
                                             RF *rf = nullptr, const bool live = true) {
    static_assert(!(FAST && LDSS), "the LDS-resident scene holds the reference tree only");
    const uint4 *const arec = LDSS ? scn + P.lds_accel_off : (FAST ? nullptr : arec_in); // the accel records in LDS, if they are there (lvl_set)
    static_assert(!(FAST && PRUNE), "the fast mode prunes its own trees by its own rule");
#ifdef LG_STAMPS
    unsigned long long stamp_acc[7] = {0, 0, 0, 0, 0, 0, 0}, stamp_t = __builtin_readcyclecounter();
    unsigned long long stamp_cnt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ret_acc[7] = {0, 0, 0, 0, 0, 0, 0}; // inside phase C: loop entry, frame + parent fetch, level record, ray, prune constants, next state; iterations
#endif
    best.t = INFINITY; best.ref = NO_HIT; best.accel = 0;
    if (COUNT) cnt.entries++; // the root accel
    uint32_t *const stk = stack + stride; // entry -1 of an empty stack is fetched (never used): one guard entry below
    Lvl L;
    lvl_set<LDSS, FAST>(P, arec, L, 0u);
    double limit = prune_limit(INFINITY, anyhit); // FAST: nodes whose tnear lies beyond this are skipped
    TriSetup tri;                                  // FAST: per mesh level (its leaves hold one triangle: per leaf the three divides would dominate)
    tri.kz = 0; tri.sx = 0.0; tri.sy = 0.0; tri.sz = 0.0;
    // ---- the root accel's local ray (bvh.rs:462), kept for the returns
    Ray root = wray;
    if (!((L.flags & AF_IDENTITY) && ray_plain(wray))) root = accel_local_ray<LDSS>(P, arec, 0u, wray);
    Ray ray = root;
    double dd = dot(ray.d, ray.d);      // a of every sphere's quadratic at this level
    double four_a = 4.0 * dd;           // 4.0 * a of its discriminant b*b - 4.0*a*c (core/math.rs:16: (4.0 * a) * c)
    uint32_t negmask = neg_mask_x(ray); // dir_is_neg (bvh.rs:463), + SIGNS_NOT_PLAIN
    uint32_t sp = 0, base = 0, cur = L.node_base, li = 0, le = 0, enter = 0, lcb = 0;
    uint32_t state = (RF::enabled && !live) ? ST_DONE : ST_NODE;
    // ---- PRUNE: per-axis limits of the level the lane is in, and the level's margin
    V3 plim{INFINITY, INFINITY, INFINITY};
    double peps = INFINITY;
    auto prune_limits = [&](const double limit_in) { // limit >= 0 (every accepted t is), or +inf before the first hit
        // (audit == 2, counting instantiations only: the audit's own test -- a deliberately UNSOUND limit, half the real one, so
        // that primitives the reference would accept do get skipped and lg_audit_prune must report them)
        const double limit = (COUNT && P.audit == 2u) ? limit_in * 0.5 : limit_in;
        const double lb = limit + limit * PRUNE_LIMIT_REL;
        V3 m{lb + peps * fabs(ray.dinv.x), lb + peps * fabs(ray.dinv.y), lb + peps * fabs(ray.dinv.z)}; // (an axis with d == 0: +inf)
        if (L.flags & AF_MESH) { // triangles: the dominant axis alone (max_dimension as in tri_setup, triangle.rs:186)
            const int kz = max_dimension(vabs(ray.d));
            if (kz != 0) m.x = INFINITY;
            if (kz != 1) m.y = INFINITY;
            if (kz != 2) m.z = INFINITY;
        }
        plim = m;
    };
    auto prune_level = [&]() { // after L and ray have changed
        double c[6];
        if (LDSS || arec) {
            const uint4 *q = arec + (L.accel * LDS_ACCEL_UNITS + 10u);
            const double2 a = lds_d2(q), b = lds_d2(q + 1), e = lds_d2(q + 2);
            c[0] = a.x; c[1] = a.y; c[2] = b.x; c[3] = b.y; c[4] = e.x; c[5] = e.y;
        } else {
            const double *q = P.accels[L.accel].prune;
            c[0] = q[0]; c[1] = q[1]; c[2] = q[2]; c[3] = q[3]; c[4] = q[4]; c[5] = q[5];
        }
        const double S = ((fabs(ray.o.x - c[0]) + fabs(ray.o.y - c[1])) + fabs(ray.o.z - c[2])) + c[3];
        const double idm = fmin_(fmin_(fabs(ray.dinv.x), fabs(ray.dinv.y)), fabs(ray.dinv.z)); // 1 / max |d|
        const bool in_range = S <= PRUNE_RANGE && idm >= 1.0 / PRUNE_RANGE && idm <= PRUNE_RANGE; // (NaN: false)
        peps = in_range ? c[4] + S * (PRUNE_E1 + c[5] * S) : INFINITY;
        prune_limits(anyhit ? 1.0 : best.t);
    };
    if (PRUNE) prune_level();
    LG_STAMP(0);
    for (;;) {
        // ---- phase A: interior nodes (bvh.rs:471-505), until no lane of the wave is at a node
        // (loops are written with their wave-uniform condition in a variable tested at the bottom: hipcc then keeps the
        // loop-carried state in place instead of copying it in and out of the loop on every trip)
        // The step comes in nine forms: SG = 0..7 for a wave whose lanes at a node all carry the same sign triple of dinv and are
        // "plain" (slab_intersects_sg: six of the slab test's ten min / max are then decided by the signs and cost nothing), SG = 8
        // the reference's formula as written.  Which one runs is a scalar decision per entry into this phase: a lane's ray, and
        // with it its signs, only changes between phases.
        auto node_phase = [&](auto sgc) __attribute__((always_inline)) {
        constexpr int SG = decltype(sgc)::value;
        // (the lane's "at a node" predicate is carried from the bottom of one trip to the top of the next: written as two tests of `state`,
        // hipcc compares twice per trip -- one vector instruction of ~35)
        bool more_nodes = true;
        bool at_node_l = state == ST_NODE;
        while (more_nodes) {
#ifdef LG_STAMPS
            stamp_cnt[5] += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == ST_NODE));
#endif
            if (at_node_l) {
                // the record's walk words: interior -> (second child's cursor, 1 << split axis, -), leaf -> (first slot, NODE_LEAF, last slot + 1)
                double bmin[3], bmax[3];
                uint32_t w_link, w_meta, w_end, w_chunk;
                if (LDSS) {
                    const char *rec = reinterpret_cast<const char *>(scn) + cur;
                    const double2 *q = reinterpret_cast<const double2 *>(rec);
                    const double2 a = q[0], b = q[1], c = q[2]; // four ds_read_b128
                    const uint4 d = *reinterpret_cast<const uint4 *>(rec + LDS_NODE_WALK_OFF);
                    bmin[0] = a.x; bmin[1] = a.y; bmin[2] = b.x; bmax[0] = b.y; bmax[1] = c.x; bmax[2] = c.y;
                    w_link = d.x; w_meta = d.y; w_end = d.z; w_chunk = d.w;
                } else {
                    // (lanes of a coherent wave are mostly at the same node near the root: one scalar fetch then)
                    const uint32_t cur0 = __builtin_amdgcn_readfirstlane(cur);
                    NodeRec nd;
                    if (__builtin_amdgcn_ballot_w64(cur != cur0) == 0ull) nd = load_node_uniform(P, cur0);
                    else nd = load_node<false>(P, scn, cur);
                    bmin[0] = nd.bmin[0]; bmin[1] = nd.bmin[1]; bmin[2] = nd.bmin[2]; bmax[0] = nd.bmax[0]; bmax[1] = nd.bmax[1]; bmax[2] = nd.bmax[2];
                    const bool lf = (nd.meta & NODE_LEAF) != 0u;
                    w_link = (lf ? L.prim_base : L.node_base) + nd.link;
                    w_meta = (lf ? NODE_LEAF : 1u << (nd.meta & 3u)) | (nd.meta & NODE_NOPRUNE);
                    w_end = w_link + (nd.meta & 0xFFFFu);
                    w_chunk = nd.pad;
                }
                const uint32_t popped = stk[(int)(sp - 1u) * (int)stride];
                bool hit;
                if (FAST) {
                    // a primitive's computed t can undershoot its box's tnear by the error of its own formula: for a sphere
                    // the quadratic's cancellation, ~sqrt(eps) of the distance to its centre, which lies inside the box
                    double tn, tf;
                    hit = slab_intersects_nc_t(bmin, bmax, ray, tn, tf);
                    hit = hit && !(tn - 4e-8 * fabs(tf) > limit);
                } else if (PRUNE) {
                    // the reference's test, and the node is skipped as well when on some axis the ray reaches its slab only beyond
                    // the limit (+ that axis's margin); never a node over a nested accel
                    double tx, ty, tz;
                    if (SG < 8) hit = slab_intersects_sg<SG & 7>(bmin, bmax, ray, tx, ty, tz);
                    else hit = slab_intersects_nc_axes(bmin, bmax, ray, tx, ty, tz);
                    const bool beyond = tx > plim.x || ty > plim.y || tz > plim.z; // (a NaN entry parameter compares false)
                    if (COUNT && !LDSS && P.audit && hit && beyond && (w_meta & NODE_NOPRUNE) == 0u) { // skipped although the reference would walk it
                        double margin = INFINITY; // the smallest margin among the axes that said "beyond"
                        V3 em{INFINITY, INFINITY, INFINITY};
                        if (tx > plim.x) { em.x = peps * fabs(ray.dinv.x); margin = fmin_(margin, em.x); }
                        if (ty > plim.y) { em.y = peps * fabs(ray.dinv.y); margin = fmin_(margin, em.y); }
                        if (tz > plim.z) { em.z = peps * fabs(ray.dinv.z); margin = fmin_(margin, em.z); }
                        audit_subtree(P, L.node_base, L.prim_base, cur, ray, dd, four_a, anyhit ? 1.0 : best.t, margin, anyhit, cnt, V3{tx, ty, tz}, em);
                    }
                    hit = hit && !(beyond && (w_meta & NODE_NOPRUNE) == 0u);
                } else if (SG < 8) {
                    double tx, ty, tz;
                    hit = slab_intersects_sg<SG & 7>(bmin, bmax, ray, tx, ty, tz);
                } else hit = slab_intersects_nc(bmin, bmax, ray);
                if (COUNT) { cnt.nodes++; dbg_event(P, 2.0 + (hit ? 0.1 : 0.0), (double)L.accel, (double)cur, (double)w_meta); }
                const bool leaf = (int32_t)w_meta < 0;            // n_primitives > 0 (bvh.rs:475): the builder emits no empty leaf
                const bool neg = ((SG < 8 ? (uint32_t)SG : negmask) & w_meta) != 0u; // dir_is_neg[axis] (bvh.rs:496); w_meta carries 1 << axis, never bit 3
                const uint32_t first = cur + (LDSS ? LDS_NODE_BYTES : 1u), second = w_link; // the two children (interior nodes)
                const uint32_t near_node = neg ? second : first, far_node = neg ? first : second;
                const bool leaf_hit = hit && leaf, interior_hit = hit != leaf_hit;
                const bool pop = !hit, can_pop = sp != base;
                stk[sp * stride] = far_node; // counts only if sp advances (bvh.rs:493-504)
                cur = interior_hit ? near_node : popped;
                sp = sp + (interior_hit ? 1u : 0u) - (pop && can_pop ? 1u : 0u);
                li = w_link; le = w_end; // (read in ST_LEAF only)
                if (PRUNE) lcb = w_chunk;
                state = leaf_hit ? ST_LEAF : (pop && !can_pop) ? ST_LEVEL_DONE : ST_NODE;
            }
#ifdef LG_STAMPS
            stamp_cnt[0] += 1; // (lanes that took this step: those whose state was ST_NODE when it began -- counted after it as "not idle")
            stamp_cnt[6] += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == ST_DONE));
#endif
            at_node_l = state == ST_NODE;
            more_nodes = wave_any(at_node_l);
        }
        };
        {
            const unsigned long long at_node = __builtin_amdgcn_ballot_w64(state == ST_NODE);
            if (at_node != 0ull) {
                uint32_t sg = SIGNS_NOT_PLAIN;
                if (!FAST && P.boxes_finite) {
                    const uint32_t n0 = (uint32_t)__builtin_amdgcn_readlane((int)negmask, (int)__builtin_ctzll(at_node));
                    if (__builtin_amdgcn_ballot_w64(state == ST_NODE && negmask != n0) == 0ull) sg = n0; // (>= 8: a ray that is not plain)
                }
#ifdef LG_NO_SG // (A/B: one form of the node phase -- 8 loop bodies less in the instruction cache)
                sg = SIGNS_NOT_PLAIN;
#endif
                switch (sg) {
#ifndef LG_NO_SG
                case 0u: node_phase(IntC<0>{}); break;
                case 1u: node_phase(IntC<1>{}); break;
                case 2u: node_phase(IntC<2>{}); break;
                case 3u: node_phase(IntC<3>{}); break;
                case 4u: node_phase(IntC<4>{}); break;
                case 5u: node_phase(IntC<5>{}); break;
                case 6u: node_phase(IntC<6>{}); break;
                case 7u: node_phase(IntC<7>{}); break;
#endif
                default: node_phase(IntC<8>{}); break;
                }
            }
        }
        LG_STAMP(1);
        // ---- phase B: leaf primitives in order[] sequence (bvh.rs:481-488)
        const bool mesh = (L.flags & AF_MESH) != 0u;
#ifdef LG_STAMPS
        if (wave_any(state == ST_LEAF && mesh)) stamp_cnt[1] += 1;
#endif
        if (state == ST_LEAF && mesh) { // every slot of a mesh accel is a triangle
            if (!FAST) tri = tri_setup(ray); // per fat leaf: amortises the three divides (triangle.rs:186-201)
            bool done;
            LeafCull lc{INFINITY, INFINITY, dd, lcb, 0.0f, 0.0f, 0.0f};
            if (PRUNE) { // the level's limits, as prune_limits made them: the dominant axis carries lb + eps * |1/d_kz|
                const double lim = anyhit ? 1.0 : best.t;
                lc.lb = lim + lim * PRUNE_LIMIT_REL;
                lc.ekz = peps * fabs(tri.kz == 0 ? ray.dinv.x : tri.kz == 1 ? ray.dinv.y : ray.dinv.z);
#ifndef LG_DIR_PER_RECORD
                const float inv_len = __frsqrt_rn((float)dd) * (1.0f - 4e-6f);
                lc.dhx = (float)ray.d.x * inv_len; lc.dhy = (float)ray.d.y * inv_len; lc.dhz = (float)ray.d.z * inv_len;
#endif
            }
            if (tri.kz == 0) done = mesh_leaf2<0, LDSS, FAST, COUNT, PRUNE>(P, scn, ray, tri, li, le, L.soup_delta, L.accel, anyhit, best, tie, cnt, lc);
            else if (tri.kz == 1) done = mesh_leaf2<1, LDSS, FAST, COUNT, PRUNE>(P, scn, ray, tri, li, le, L.soup_delta, L.accel, anyhit, best, tie, cnt, lc);
            else done = mesh_leaf2<2, LDSS, FAST, COUNT, PRUNE>(P, scn, ray, tri, li, le, L.soup_delta, L.accel, anyhit, best, tie, cnt, lc);
            if (FAST) limit = prune_limit(best.t, anyhit);
            if (PRUNE && !anyhit) prune_limits(best.t);
            if (done) state = ST_DONE;
            else if (sp != base) { --sp; cur = stk[sp * stride]; state = ST_NODE; }
            else state = ST_LEVEL_DONE;
        }
        LG_STAMP(2);
        bool at_slot_l = state == ST_LEAF; // (carried like at_node_l above)
        bool more_prims = wave_any(at_slot_l);
        while (more_prims) {
#ifdef LG_STAMPS
            stamp_cnt[7] += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == ST_LEAF));
            stamp_cnt[8] += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == ST_DONE));
#endif
            if (at_slot_l) {
                const uint32_t slot = li;
                const uint32_t ref = load_primref<LDSS>(P, scn, slot);
                LeafRec g;
                if (LDSS) { const uint4 *q = scn + (P.lds_soup_off + __umul24(slot, 3u)); g = LeafRec{q[0], q[1], q[2]}; }
                else g = load_rec(P, slot);
                const uint32_t popped = stk[(int)(sp - 1u) * (int)stride];
                li = slot + 1u;
                const uint32_t kind = ref >> 30, idx = ref & PRIM_INDEX_MASK;
                bool accepted = false;
                double t = 0.0;
                // (the slots of a non-mesh accel are spheres, boxes and nested accels -- triangles live in mesh accels, whose leaves
                // mesh_leaf2 walks; the sphere is asked for first, with one compare on the primref: the common case passes one branch)
                if (ref < (1u << 30)) { // PK_SPHERE: Sphere::intersect_t + quad_roots (sphere.rs:30-69, core/math.rs) == sphere_t_a
                    const V3 cen{rec_f64(g.a.x, g.a.y), rec_f64(g.a.z, g.a.w), rec_f64(g.b.x, g.b.y)};
                    const V3 l = ray.o - cen;
                    const double b = 2.0 * dot(ray.d, l);
                    const double c = dot(l, l) - rec_f64(g.c.x, g.c.y); // rad * rad, formed by the host
                    bool has = false;
                    if (dd == 0.0) {
                        if (b != 0.0) { t = -c / b; has = true; }
                    } else {
                        const double disc = b * b - four_a * c;
                        if (!(disc < 0.0)) {
                            const double q = -(b + signum(b) * sqrt(disc)) / 2.0;
                            const double r0 = q / dd;
                            const double r1 = (q == 0.0) ? r0 : c / q;
                            const double t0 = fmin_(r0, r1), t1 = fmax_(r0, r1);
                            t = t0 < 0.0 ? t1 : t0;
                            has = true;
                        }
                    }
                    accepted = has && !(t < 0.0) && !(t >= best.t);
                } else if (ref >= (3u << 30)) { // PK_ACCEL
                    // nested BVHAccel (Group / Mesh): entered below, outside this loop -- the ray and the level are
                    // loop-invariant here, which keeps them out of the loop's register shuffles
                    enter = idx;
                    state = ST_ENTER;
                } else { // PK_CUBOID (a triangle slot outside a mesh accel cannot be built: host.cpp, Flattener::aggregate)
                    double mn[3] = {rec_f64(g.a.x, g.a.y), rec_f64(g.a.z, g.a.w), rec_f64(g.b.x, g.b.y)};
                    double mx[3] = {rec_f64(g.b.z, g.b.w), rec_f64(g.c.x, g.c.y), rec_f64(g.c.z, g.c.w)};
                    V3 d0, d1;
                    if (cuboid_hit<false>(mn, mx, ray, t, d0, d1)) accepted = !(t >= best.t);
                }
                if (COUNT) {
                    if (kind == PK_SPHERE) cnt.spheres++; else if (kind == PK_CUBOID) cnt.cuboids++; else if (kind == PK_ACCEL) cnt.entries++; else cnt.triangles++;
                    if (kind != PK_ACCEL) dbg_event(P, 3.0 + (accepted ? 0.1 : 0.0), (double)ref, t, (double)L.accel);
                    else dbg_event(P, 4.0, (double)idx, (double)sp, (double)base);
                }
                if (FAST && kind != PK_ACCEL && ((!accepted && t == best.t && best.ref != NO_HIT) || t != t)) tie = true; // visit order decides
                if (accepted) {
                    best.t = t; best.ref = ref; best.accel = L.accel;
                    if (FAST) limit = prune_limit(t, anyhit);
                    if (PRUNE && !anyhit) prune_limits(t);
                    if (anyhit && t < 1.0) state = ST_DONE; // occluded: point.rs:49 only asks isect.t < 1.0
                }
                if (state == ST_LEAF && li >= le) { // leaf exhausted: next pending node of this level, or the level is done
                    if (sp != base) { --sp; cur = popped; state = ST_NODE; }
                    else state = ST_LEVEL_DONE;
                }
            }
            at_slot_l = state == ST_LEAF;
            more_prims = wave_any(at_slot_l);
#ifdef LG_STAMPS
            stamp_cnt[2] += 1;
#endif
        }
        LG_STAMP(3);
        // ---- a leaf slot that is a nested BVHAccel (Group / Mesh): park this level, re-express the ray (bvh.rs:462)
#ifdef LG_STAMPS
        if (wave_any(state == ST_ENTER)) stamp_cnt[3] += 1;
        if (wave_any(state == ST_LEVEL_DONE)) stamp_cnt[4] += 1;
#endif
        if (state == ST_ENTER) {
            lvl_set<LDSS, FAST>(P, arec, L, enter);
            const bool same = (L.flags & AF_IDENTITY) != 0u && ray_plain(ray);
            stk[sp * stride] = li; stk[(sp + 1u) * stride] = le; stk[(sp + 2u) * stride] = base | (same ? FRAME_SAME_RAY : 0u);
            sp += 3u; base = sp;
            if (!same) {
                ray = accel_local_ray<LDSS>(P, arec, enter, ray);
                dd = dot(ray.d, ray.d);
                four_a = 4.0 * dd;
                negmask = neg_mask_x(ray);
            }
            if (FAST && (L.flags & AF_MESH)) tri = tri_setup(ray);
            if (PRUNE) prune_level();
            cur = L.node_base;
            state = ST_NODE; // node 0 is tested when visited (bvh.rs:472-473)
        }
        LG_STAMP(4);
        // ---- phase C: this nested BVHAccel is exhausted: resume the parent's leaf loop (bvh.rs:483-488)
#ifdef LG_STAMPS
        unsigned long long rt_t = __builtin_readcyclecounter();
#define LG_RSTAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); ret_acc[i] += now_ - rt_t; rt_t = now_; } while (0)
#else
#define LG_RSTAMP(i) do { } while (0)
#endif
        while (state == ST_LEVEL_DONE) { // (a lane comes back through every level that is exhausted with it)
            LG_RSTAMP(0);
            if (L.accel == 0u) state = ST_DONE;
            else {
                const uint32_t w2 = stk[(sp - 1u) * stride];
                le = stk[(sp - 2u) * stride]; li = stk[(sp - 3u) * stride];
                sp -= 3u; base = w2 & ~FRAME_SAME_RAY;
                if (COUNT) dbg_event(P, 5.0, (double)L.accel, (double)li, (double)le);
                uint32_t parent, nchain;
                if (LDSS || arec) {
                    parent = lds_u4(arec + (L.accel * LDS_ACCEL_UNITS + 7u)).x;
                    nchain = lds_u4(arec + (parent * LDS_ACCEL_UNITS + 7u)).y;
                } else {
                    parent = (uint32_t)P.accels[L.accel].parent;
                    nchain = P.accels[parent].nchain;
                }
                LG_RSTAMP(1);
                lvl_set<LDSS, FAST>(P, arec, L, parent);
                LG_RSTAMP(2);
                if (!(w2 & FRAME_SAME_RAY)) { // the parent's ray again: from the root's, through the same transforms
                    ray = root;
                    for (uint32_t i = 1; i < nchain; ++i) {
                        uint32_t c, cflags; // the parent's root -> self chain, one accel at a time
                        if (LDSS || arec) {
                            c = ((lg_lds_u32)(arec + (parent * LDS_ACCEL_UNITS + 8u)))[i];
                            cflags = lds_u4(arec + (c * LDS_ACCEL_UNITS + 6u)).w;
                        } else {
                            c = P.accels[parent].chain[i];
                            cflags = P.accels[c].flags;
                        }
                        if (!((cflags & AF_IDENTITY) && ray_plain(ray))) ray = accel_local_ray<LDSS>(P, arec, c, ray);
                    }
                    dd = dot(ray.d, ray.d);
                    four_a = 4.0 * dd;
                    negmask = neg_mask_x(ray);
                }
                LG_RSTAMP(3);
                if (PRUNE) prune_level();
                LG_RSTAMP(4);
                if (li < le) state = ST_LEAF;
                else if (sp != base) { --sp; cur = stk[sp * stride]; state = ST_NODE; }
                else state = ST_LEVEL_DONE; // the parent level is exhausted as well
                LG_RSTAMP(5);
#ifdef LG_STAMPS
                ret_acc[6] += 1;
#endif
            }
        }
#undef LG_RSTAMP
        LG_STAMP(5);
#ifdef LG_STAMPS
        stamp_acc[6] += 1;
#endif
        if (RF::enabled) { // (a wave-uniform decision, and EVERY lane calls rf.next -- it keeps wave-wide state, which only uniform control flow can keep in step; the lanes that are done say so)
            if ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == ST_DONE)) >= rf->threshold()) {
                {
                    Ray nw = wray;
                    if (rf->next(state == ST_DONE, best, nw)) { // a fresh walk for this lane: everything the head of this function set up
                        best.t = INFINITY; best.ref = NO_HIT; best.accel = 0;
                        lvl_set<LDSS, FAST>(P, arec, L, 0u);
                        root = nw;
                        if (!((L.flags & AF_IDENTITY) && ray_plain(nw))) root = accel_local_ray<LDSS>(P, arec, 0u, nw);
                        ray = root;
                        dd = dot(ray.d, ray.d);
                        four_a = 4.0 * dd;
                        negmask = neg_mask_x(ray);
                        sp = 0; base = 0; cur = L.node_base; li = 0; le = 0; enter = 0; lcb = 0;
                        state = ST_NODE;
                        if (PRUNE) prune_level();
                    }
                }
            }
        }
        if (!wave_any(state != ST_DONE)) break;
    }
#ifdef LG_STAMPS
    for (int i = 0; i < 7; ++i) // phase C runs lane by lane: the wave's figure is its busiest lane's
        for (int off = 32; off > 0; off >>= 1) { const unsigned long long o_ = __shfl_xor(ret_acc[i], off); ret_acc[i] = o_ > ret_acc[i] ? o_ : ret_acc[i]; }
    if ((threadIdx.x & 63u) == 0u && P.stats) {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(P.stats);
        for (int i = 0; i < 7; ++i) atomicAdd(dst + i, stamp_acc[i]);
        atomicAdd(dst + 7, 1ull);
        unsigned long long *cnt = P.stamp_counts;
        if (cnt) for (int i = 0; i < 9; ++i) atomicAdd(cnt + i, stamp_cnt[i]);
        for (int i = 0; i < 7; ++i) atomicAdd(dst + 8 + i, ret_acc[i]); // (words 8..14 of the first record; the lane that has seen most)
    }
#endif
}

// Fast mode's walk over WIDE records (DNode4, dscene.h): `cur` is an interior node of the fast tree whose own box is known to be
// hit; one 128-byte record holds the boxes and link words of up to four nodes below it, so a step tests four boxes per dependent
// fetch -- the fast walk waits on those fetches, not on the VALU.  The nearest hit child (by slab tnear) is taken, the others are
// pushed (one word each, the record's link word: a leaf carries its slot range, so a popped leaf needs no fetch of its own).
// Everything else -- levels, leaves, pruning margins, tie / NaN flags -- is traverse_ref<false, true>.

// the root node of a level: its own box, once (bvh.rs:472-473 for node 0)
template <bool COUNT>
__device__ __forceinline__ void fast_level_root(const DParams &P, const Lvl &L, const Ray &ray, const double limit, uint32_t &state, uint32_t &cur,
                                                uint32_t &li, uint32_t &le, Counters &cnt) {
    if (COUNT) cnt.nodes++;
    const NodeRec nd = load_node<false>(P, nullptr, L.node_base);
    double tn, tf;
    bool hit = slab_intersects_nc_t(nd.bmin, nd.bmax, ray, tn, tf);
    hit = hit && !(tn - 4e-8 * fabs(tf) > limit);
    cur = L.node_base;
    if (!hit) state = ST_LEVEL_DONE;
    else if (nd.meta & NODE_LEAF) { li = L.prim_base + nd.link; le = li + (nd.meta & 0xFFFFu); state = ST_LEAF; }
    else state = ST_NODE;
}
template <bool COUNT = false>
__device__ __forceinline__ void traverse_fast(const DParams &P, const Ray &wray, const bool anyhit, uint32_t *stack, const uint32_t stride,
                                             Best &best, const uint4 *scn, bool &tie, Counters &cnt) {
    constexpr bool LDSS = false, FAST = true;
    best.t = INFINITY; best.ref = NO_HIT; best.accel = 0;
    if (COUNT) cnt.entries++; // the root accel
    uint32_t *const stk = stack + stride; // entry -1 of an empty stack is fetched (never used): one guard entry below
    Lvl L;
    lvl_set<LDSS, FAST>(P, scn, L, 0u);
    double limit = prune_limit(INFINITY, anyhit); // FAST: nodes whose tnear lies beyond this are skipped
    TriSetup tri;                                  // FAST: per mesh level (its leaves hold one triangle: per leaf the three divides would dominate)
    tri.kz = 0; tri.sx = 0.0; tri.sy = 0.0; tri.sz = 0.0;
    // ---- the root accel's local ray (bvh.rs:462), kept for the returns
    Ray root = wray;
    if (!((L.flags & AF_IDENTITY) && ray_plain(wray))) root = accel_local_ray<LDSS>(P, scn, 0u, wray);
    Ray ray = root;
    double dd = dot(ray.d, ray.d);      // a of every sphere's quadratic at this level
    double four_a = 4.0 * dd;           // 4.0 * a of its discriminant b*b - 4.0*a*c (core/math.rs:16: (4.0 * a) * c)
    uint32_t negmask = neg_mask(ray); (void)negmask; // (the pair walk orders children by tnear; kept for the level bookkeeping shared with traverse_ref)
    uint32_t sp = 0, base = 0, cur = L.node_base, li = 0, le = 0, enter = 0;
    uint32_t state = ST_NODE;
    fast_level_root<COUNT>(P, L, ray, limit, state, cur, li, le, cnt);
    auto take = [&](const uint32_t e) { // a pending child comes off the stack
        if (e & WIDE_LEAF) { li = L.prim_base + (e & WIDE_START_MASK); le = li + ((e >> WIDE_COUNT_SHIFT) & 7u); state = ST_LEAF; }
        else { cur = L.node_base + e; state = ST_NODE; }
    };
    for (;;) {
        // ---- phase A: interior nodes, two children per step, until no lane of the wave is at a node
        // (loops are written with their wave-uniform condition in a variable tested at the bottom: hipcc then keeps the
        // loop-carried state in place instead of copying it in and out of the loop on every trip)
        bool more_nodes = wave_any(state == ST_NODE);
        while (more_nodes) {
            if (state == ST_NODE) {
                // one wide record: four child boxes (f32, rounded outward by the host; widened exactly) and their link words
                // (lanes of a coherent wave are mostly at the same record near the root: it then comes through the scalar cache)
                uint4 w0, w1, w2, w3, w4, w5, lk;
                const uint32_t cur0 = __builtin_amdgcn_readfirstlane(cur);
                if (__builtin_amdgcn_ballot_w64(cur != cur0) == 0ull) {
                    lg_const_u4 q = (lg_const_u4)(uintptr_t)(P.nodes4 + cur0);
                    w0 = load_const_u4(q, 0); w1 = load_const_u4(q, 1); w2 = load_const_u4(q, 2); w3 = load_const_u4(q, 3);
                    w4 = load_const_u4(q, 4); w5 = load_const_u4(q, 5); lk = load_const_u4(q, 6);
                } else {
                    const uint4 *q = reinterpret_cast<const uint4 *>(P.nodes4 + cur);
                    w0 = q[0]; w1 = q[1]; w2 = q[2]; w3 = q[3]; w4 = q[4]; w5 = q[5]; lk = q[6];
                }
                const uint32_t popped = stk[(int)(sp - 1u) * (int)stride];
                const uint32_t bw[24] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w,
                                         w3.x, w3.y, w3.z, w3.w, w4.x, w4.y, w4.z, w4.w, w5.x, w5.y, w5.z, w5.w};
                const uint32_t link[WIDE] = {lk.x, lk.y, lk.z, lk.w};
                if (COUNT) cnt.nodes += (uint32_t)WIDE;
                double tn[WIDE];
                bool hit[WIDE];
#pragma unroll
                for (int k = 0; k < WIDE; ++k) {
                    const double mn[3] = {(double)__uint_as_float(bw[6 * k]), (double)__uint_as_float(bw[6 * k + 1]), (double)__uint_as_float(bw[6 * k + 2])};
                    const double mx[3] = {(double)__uint_as_float(bw[6 * k + 3]), (double)__uint_as_float(bw[6 * k + 4]), (double)__uint_as_float(bw[6 * k + 5])};
                    double tf;
                    hit[k] = slab_intersects_nc_t(mn, mx, ray, tn[k], tf);
                    // (a primitive's computed t can undershoot its box's tnear by the error of its own formula: for a sphere the
                    // quadratic's cancellation, ~sqrt(eps) of the distance to its centre, which lies inside the box)
                    // (audit == 2, counting instantiations only: lg_audit_fast's own test -- a deliberately UNSOUND limit, half the real one, so that
                    // hits the reference finds are skipped and the audit must report them)
                    hit[k] = hit[k] && !(tn[k] - 4e-8 * fabs(tf) > ((COUNT && P.audit == 2u) ? 0.5 * limit : limit)) && link[k] != NO_HIT;
                }
                // the nearest hit child is taken, the others are pushed in record order
                int near = -1;
                double near_t = INFINITY;
#pragma unroll
                for (int k = 0; k < WIDE; ++k) {
                    const bool better = hit[k] && (near < 0 || tn[k] < near_t);
                    near = better ? k : near;
                    near_t = better ? tn[k] : near_t;
                }
                const bool any = near >= 0, can_pop = sp != base;
                uint32_t taken = popped;
#pragma unroll
                for (int k = 0; k < WIDE; ++k) {
                    const bool push = hit[k] && k != near;
                    stk[sp * stride] = link[k]; // counts only if sp advances
                    sp += push ? 1u : 0u;
                    taken = (k == near) ? link[k] : taken;
                }
                sp -= (!any && can_pop) ? 1u : 0u;
                if (!any && !can_pop) state = ST_LEVEL_DONE;
                else if (taken & WIDE_LEAF) { li = L.prim_base + (taken & WIDE_START_MASK); le = li + ((taken >> WIDE_COUNT_SHIFT) & 7u); state = ST_LEAF; }
                else cur = L.node_base + taken;
            }
            more_nodes = wave_any(state == ST_NODE);
        }
        // ---- phase B: leaf primitives in order[] sequence (bvh.rs:481-488)
        const bool mesh = (L.flags & AF_MESH) != 0u;
        if (state == ST_LEAF && mesh) { // every slot of a mesh accel is a triangle
            if (!FAST) tri = tri_setup(ray); // per fat leaf: amortises the three divides (triangle.rs:186-201)
            bool done;
            const LeafCull lc{INFINITY, INFINITY, dd, 0u, 0.0f, 0.0f, 0.0f};
            if (tri.kz == 0) done = mesh_leaf2<0, LDSS, FAST, COUNT>(P, scn, ray, tri, li, le, L.soup_delta, L.accel, anyhit, best, tie, cnt, lc);
            else if (tri.kz == 1) done = mesh_leaf2<1, LDSS, FAST, COUNT>(P, scn, ray, tri, li, le, L.soup_delta, L.accel, anyhit, best, tie, cnt, lc);
            else done = mesh_leaf2<2, LDSS, FAST, COUNT>(P, scn, ray, tri, li, le, L.soup_delta, L.accel, anyhit, best, tie, cnt, lc);
            if (FAST) limit = prune_limit(best.t, anyhit);
            if (done) state = ST_DONE;
            else if (sp != base) { --sp; take(stk[sp * stride]); }
            else state = ST_LEVEL_DONE;
        }
        bool more_prims = wave_any(state == ST_LEAF);
        while (more_prims) {
            if (state == ST_LEAF) {
                const uint32_t slot = li;
                const uint32_t ref = load_primref<LDSS>(P, scn, slot);
                LeafRec g;
                if (LDSS) { const uint4 *q = scn + (P.lds_soup_off + __umul24(slot, 3u)); g = LeafRec{q[0], q[1], q[2]}; }
                else g = load_rec(P, slot);
                const uint32_t popped = stk[(int)(sp - 1u) * (int)stride];
                li = slot + 1u;
                const uint32_t kind = ref >> 30, idx = ref & PRIM_INDEX_MASK;
                bool accepted = false;
                double t = 0.0;
                if (kind == PK_SPHERE) { // Sphere::intersect_t + quad_roots (sphere.rs:30-69, core/math.rs) == sphere_t_a
                    const V3 cen{rec_f64(g.a.x, g.a.y), rec_f64(g.a.z, g.a.w), rec_f64(g.b.x, g.b.y)};
                    const V3 l = ray.o - cen;
                    const double b = 2.0 * dot(ray.d, l);
                    const double c = dot(l, l) - rec_f64(g.c.x, g.c.y); // rad * rad, formed by the host
                    bool has = false;
                    if (dd == 0.0) {
                        if (b != 0.0) { t = -c / b; has = true; }
                    } else {
                        const double disc = b * b - four_a * c;
                        if (!(disc < 0.0)) {
                            const double q = -(b + signum(b) * sqrt(disc)) / 2.0;
                            const double r0 = q / dd;
                            const double r1 = (q == 0.0) ? r0 : c / q;
                            const double t0 = fmin_(r0, r1), t1 = fmax_(r0, r1);
                            t = t0 < 0.0 ? t1 : t0;
                            has = true;
                        }
                    }
                    accepted = has && !(t < 0.0) && !(t >= best.t);
                } else if (kind == PK_CUBOID) {
                    double mn[3] = {rec_f64(g.a.x, g.a.y), rec_f64(g.a.z, g.a.w), rec_f64(g.b.x, g.b.y)};
                    double mx[3] = {rec_f64(g.b.z, g.b.w), rec_f64(g.c.x, g.c.y), rec_f64(g.c.z, g.c.w)};
                    V3 d0, d1;
                    if (cuboid_hit<false>(mn, mx, ray, t, d0, d1)) accepted = !(t >= best.t);
                } else if (kind == PK_ACCEL) {
                    // nested BVHAccel (Group / Mesh): entered below, outside this loop -- the ray and the level are
                    // loop-invariant here, which keeps them out of the loop's register shuffles
                    enter = idx;
                    state = ST_ENTER;
                } else { // a triangle outside a mesh accel cannot be built by the scene API; kept for completeness
                    const uint32_t *vi = P.tri_v + 3ull * idx;
                    TriHit h;
                    if (triangle_t(load_f3(P.vpos, vi[0]), load_f3(P.vpos, vi[1]), load_f3(P.vpos, vi[2]), ray, h)) { t = h.t; accepted = !(t >= best.t); }
                }
                if (COUNT) {
                    if (kind == PK_SPHERE) cnt.spheres++; else if (kind == PK_CUBOID) cnt.cuboids++; else if (kind == PK_ACCEL) cnt.entries++; else cnt.triangles++;
                    if (kind != PK_ACCEL) dbg_event(P, 3.0 + (accepted ? 0.1 : 0.0), (double)ref, t, (double)L.accel);
                    else dbg_event(P, 4.0, (double)idx, (double)sp, (double)base);
                }
                if (FAST && kind != PK_ACCEL && ((!accepted && t == best.t && best.ref != NO_HIT) || t != t)) tie = true; // visit order decides
                if (accepted) {
                    best.t = t; best.ref = ref; best.accel = L.accel;
                    if (FAST) limit = prune_limit(t, anyhit);
                    if (anyhit && t < 1.0) state = ST_DONE; // occluded: point.rs:49 only asks isect.t < 1.0
                }
                if (state == ST_LEAF && li >= le) { // leaf exhausted: next pending node of this level, or the level is done
                    if (sp != base) { --sp; take(popped); }
                    else state = ST_LEVEL_DONE;
                }
            }
            more_prims = wave_any(state == ST_LEAF);
        }
        // ---- a leaf slot that is a nested BVHAccel (Group / Mesh): park this level, re-express the ray (bvh.rs:462)
        if (state == ST_ENTER) {
            lvl_set<LDSS, FAST>(P, scn, L, enter);
            const bool same = (L.flags & AF_IDENTITY) != 0u && ray_plain(ray);
            stk[sp * stride] = li; stk[(sp + 1u) * stride] = le; stk[(sp + 2u) * stride] = base | (same ? FRAME_SAME_RAY : 0u);
            sp += 3u; base = sp;
            if (!same) {
                ray = accel_local_ray<LDSS>(P, scn, enter, ray);
                dd = dot(ray.d, ray.d);
                four_a = 4.0 * dd;
                negmask = neg_mask(ray);
            }
            if (FAST && (L.flags & AF_MESH)) tri = tri_setup(ray);
            fast_level_root<COUNT>(P, L, ray, limit, state, cur, li, le, cnt);
        }
        // ---- phase C: this nested BVHAccel is exhausted: resume the parent's leaf loop (bvh.rs:483-488)
        while (state == ST_LEVEL_DONE) { // (a lane comes back through every level that is exhausted with it)
            if (L.accel == 0u) state = ST_DONE;
            else {
                const uint32_t w2 = stk[(sp - 1u) * stride];
                le = stk[(sp - 2u) * stride]; li = stk[(sp - 3u) * stride];
                sp -= 3u; base = w2 & ~FRAME_SAME_RAY;
                if (COUNT) dbg_event(P, 5.0, (double)L.accel, (double)li, (double)le);
                uint32_t parent, nchain;
                const uint32_t *chain;
                if (LDSS) {
                    const uint4 *rec = scn + (P.lds_accel_off + L.accel * LDS_ACCEL_UNITS); // (never taken: LDSS is false here)
                    parent = rec[7].x;
                    const uint4 *prec = scn + (P.lds_accel_off + parent * LDS_ACCEL_UNITS);
                    nchain = prec[7].y; chain = reinterpret_cast<const uint32_t *>(prec + 8);
                } else {
                    parent = (uint32_t)P.accels[L.accel].parent;
                    nchain = P.accels[parent].nchain; chain = P.accels[parent].chain;
                }
                lvl_set<LDSS, FAST>(P, scn, L, parent);
                if (!(w2 & FRAME_SAME_RAY)) { // the parent's ray again: from the root's, through the same transforms
                    ray = root;
                    for (uint32_t i = 1; i < nchain; ++i) {
                        const uint32_t c = chain[i];
                        const uint32_t cflags = LDSS ? scn[P.lds_accel_off + c * LDS_ACCEL_UNITS + 6u].w : P.accels[c].flags;
                        if (!((cflags & AF_IDENTITY) && ray_plain(ray))) ray = accel_local_ray<LDSS>(P, scn, c, ray);
                    }
                    dd = dot(ray.d, ray.d);
                    four_a = 4.0 * dd;
                    negmask = neg_mask(ray);
                }
                if (li < le) state = ST_LEAF;
                else if (sp != base) { --sp; take(stk[sp * stride]); }
                else state = ST_LEVEL_DONE; // the parent level is exhausted as well
            }
        }
        if (!wave_any(state != ST_DONE)) break;
    }
}


// One ray through the scene in the accel's mode.  Reference mode: the reference walk.  Fast mode: the fast walk, then
//   * closest hit: the winner counts if no exact tie (or NaN) was met and the reference tree would have tested it (ref_candidate);
//   * any-hit: an occluder counts if the reference tree would have tested it (the reference then finds it or one before it);
//     "not occluded" stands unless a tie / NaN makes the reference's own answer depend on its visit order;
// otherwise the ray is traced again with the reference walk over the tables in HBM / L2.
template <bool LDSS, bool FAST, bool PRUNE = false, bool COUNT = false>
__device__ __forceinline__ void walk(const DParams &P, const Ray &ray, const bool anyhit, uint32_t *stack, const uint32_t stride, Best &best,
                                     const uint4 *scn, Counters &cnt, const uint4 *arec = nullptr) {
    bool tie = false;
#ifndef LG_FAST_ONE_NODE
    if (FAST) traverse_fast<COUNT>(P, ray, anyhit, stack, stride, best, scn, tie, cnt);
    else
#endif
    traverse_ref<LDSS, FAST, PRUNE, COUNT>(P, ray, anyhit, stack, stride, best, scn, tie, cnt, arec);
    if (!FAST) return;
    if (COUNT) dbg_event(P, 9.0, tie ? 1.0 : 0.0, best.t, (double)best.ref);
    bool redo;
    if (anyhit && !(best.t < 1.0)) redo = tie;
    else {
        redo = anyhit ? false : tie;
#ifndef LG_NO_REFCHECK
        if (!redo && best.ref != NO_HIT) redo = !ref_candidate(P, ray, best);
#endif
    }
    if (COUNT && redo) cnt.a_runs++; // (lg_audit_fast: rays the fast walk itself hands to the reference walk -- an exact tie, a winner the reference tree would not have tested)
    if (redo) traverse_ref<false, false, false, COUNT>(P, ray, anyhit, stack, stride, best, nullptr, tie, cnt, arec);
}
// lg_audit_fast (counting instantiations in fast mode, DParams::audit): the ray once more with the reference walk, and whether fast mode's
// answer is the reference's -- the same primitive in the same accel at the same t (bit for bit) for a closest-hit ray, the same verdict
// `isect.t < 1.0` for a shadow ray (point.rs:49: which occluder is found first does not matter).
__device__ __forceinline__ void audit_fast_ray(const DParams &P, const Ray &ray, const bool anyhit, uint32_t *stack, const uint32_t stride, const Best &got,
                                               Counters &cnt, const uint4 *arec) {
    Best want;
    bool tie = false;
    Counters unused = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    traverse_ref<false, false, false, false>(P, ray, anyhit, stack, stride, want, nullptr, tie, unused, arec);
    cnt.a_prims++;
    const bool same = anyhit ? ((got.t < 1.0) == (want.t < 1.0))
                             : (got.ref == want.ref && (got.ref == NO_HIT || (got.accel == want.accel && __double_as_longlong(got.t) == __double_as_longlong(want.t))));
    if (!same) cnt.a_viol++;
}

// ------------------------------------------------------------------------------------------
// hit resolution: the winning primitive's RayIntersection carried back to world space
// (primitive intersect, then bvh.rs:509-519 / transform.rs:243-264 for every accel on the way up)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t resolve_hit(const DParams &P, const Ray &wray, const Best &best, Isect &is) {
    Ray lr = local_ray(P, wray, best.accel);
    uint32_t kind = best.ref >> 30, idx = best.ref & PRIM_INDEX_MASK;
    int32_t prim_mat = -1;
    if (kind == PK_SPHERE) {
        DSphere s = P.spheres[idx];
        bool inside;
        double t = sphere_t(lr, V3{s.cx, s.cy, s.cz}, s.r, inside);
        sphere_full(s, lr, t, inside, is);
        prim_mat = P.sphere_mat[idx];
    } else if (kind == PK_CUBOID) {
        DCuboid c = P.cuboids[idx];
        double t; V3 d0, d1;
        cuboid_hit<true>(c.mn, c.mx, lr, t, d0, d1);
        isect_set(is, t, d0, d1);
        is.has_n = true;
        is.n = face_forward(cross(d0, d1), -lr.d);
        prim_mat = P.cuboid_mat[idx];
    } else {
        triangle_full(P, idx, P.accels[best.accel].flags, lr, is);
    }
    int32_t isect_mat = P.default_material; // RayIntersection::new -> Material::default()
    int32_t a = (int32_t)best.accel;
    while (a >= 0) {
        const DAccel *A = P.accels + a;
        // transform_ray_intersection (transform.rs:243-264)
        V3 gu = xf_vector(A->m, is.gu), gv = xf_vector(A->m, is.gv);
        if (vne(is.gu, is.su) || vne(is.gv, is.sv)) {
            is.su = xf_vector(A->m, is.su); is.sv = xf_vector(A->m, is.sv);
        } else {
            is.su = gu; is.sv = gv;
        }
        is.gu = gu; is.gv = gv;
        if (is.has_n) is.n = xf_normal(A->minv, is.n);
        if (A->material >= 0) isect_mat = A->material; // bvh.rs:513-515
        if (A->flags & AF_SWAP_BACKFACE) {             // surface.rs:88-99
            V3 tmp = is.gu; is.gu = is.gv; is.gv = tmp;
            tmp = is.su; is.su = is.sv; is.sv = tmp;
            if (is.has_n) is.n = -is.n;
        }
        a = A->parent;
    }
    return prim_mat >= 0 ? prim_mat : isect_mat; // integrate.rs:30
}


} // namespace lg
