// lasgun_amd/csrc/kernels.hip -- CDNA4 (gfx950) kernels of the per-pixel ray-trace path.
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
//   stream_packet_kernel<SHADOW, LDSS>        opt-in: ONE tree walk per wavefront (ballot / vote), stream_fixup_kernel<SHADOW>
//                                             re-traces the lanes it flagged, stream_shade_kernel finishes the pixels
//   trace_pixel_kernel<FAST>                  one pixel by one lane, with an event log of the walk (lg_trace_pixel)
//   kat_kernel, kat_si_kernel, math_kernel    probes behind the test hooks of the C ABI
//
// One traversal per mode: traverse_ref<LDSS, FAST, PRUNE, COUNT> (reference tree; FAST: the fast trees one node per step, an
// A/B) and traverse_fast<COUNT> (fast trees, child pairs), both behind walk<>; traverse_packet<LDSS> for the packet kernels.
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
#include <hip/hip_runtime.h>

#include "dscene.h"
#include "trig.h"

#ifndef LG_TRAV_WAVES_PER_SIMD
#define LG_TRAV_WAVES_PER_SIMD 4 // register budget of the streaming pipeline's traversal kernels
#endif
#define LG_BLOCK 256 // threads per workgroup; also the per-entry stride (in dwords) of the LDS stacks
#ifndef LG_WAVES_PER_SIMD
#define LG_WAVES_PER_SIMD 4 // register budget: 512 / 4 = 128 VGPRs per lane
#endif

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
__device__ __forceinline__ void sphere_full(const DSphere &s, const Ray &ray, double t, bool inside, Isect &is) {
    V3 cen{s.cx, s.cy, s.cz};
    V3 p = ray.o + ray.d * t - cen;
    if (p.x == 0.0 && p.y == 0.0) p.x = 1e-5 * s.r;
    double phi = p_atan2(p.y, p.x);
    if (phi < 0.0) phi += 2.0 * PI;
    double theta = p_acos(fmin_(fmax_(p.z / s.r, -1.0), 1.0));
    V3 dpdu{-2.0 * PI * p.y, 2.0 * PI * p.x, 0.0};
    double sin_phi, cos_phi;
    p_sincos(phi, sin_phi, cos_phi);
    V3 dpdv = PI * V3{p.z * cos_phi, p.z * sin_phi, -s.r * p_sin(theta)};
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

// ---- scene table access: HBM/L2 tables, or (LDSS) the copy a 1024-lane workgroup holds in LDS.
// Lanes of a wave read DIFFERENT records, 56 bytes per node visit: through the vector L1 that is
// 64 B/clk per CU and the traversal kernels were bound by it as much as by VALU issue; the LDS
// delivers 256 B/clk per CU at a third of the latency.  Image: nodes padded to 80 B, primrefs, and one 48-byte
// leaf record per primref slot (sphere c, r / cuboid min, max / triangle positions): 16 consecutive records of
// either kind start in 16 different bank groups.
struct NodeRec {
    double bmin[3], bmax[3];
    uint32_t link, meta;
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
        n.link = d.x; n.meta = d.y;
    } else {
        // one 64-byte record = four 16-byte loads from a single line, all issued before the slab test
        const DNode *nd = P.nodes + idx;
        n.bmin[0] = nd->bmin[0]; n.bmin[1] = nd->bmin[1]; n.bmin[2] = nd->bmin[2];
        n.bmax[0] = nd->bmax[0]; n.bmax[1] = nd->bmax[1]; n.bmax[2] = nd->bmax[2];
        n.link = nd->link; n.meta = nd->meta;
    }
    return n;
}
template <bool LDSS>
__device__ __forceinline__ void load_node_link(const DParams &P, const uint4 *scn, uint32_t idx, uint32_t &link, uint32_t &meta) {
    if (LDSS) {
        uint2 d = *reinterpret_cast<const uint2 *>(scn + (P.lds_node_off + ((idx << 2) + idx) + 3u));
        link = d.x; meta = d.y;
    } else {
        const DNode *nd = P.nodes + idx;
        link = nd->link; meta = nd->meta;
    }
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
        const DNode *nd = P.nodes + (node_base + n);
        const double bmin[3] = {nd->bmin[0], nd->bmin[1], nd->bmin[2]}, bmax[3] = {nd->bmax[0], nd->bmax[1], nd->bmax[2]};
        n = nd->parent; // same 64-byte record: one fetch per step
        if (!slab_intersects(bmin, bmax, ray)) return false;
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
constexpr uint32_t FRAME_SAME_RAY = 0x80000000u; // level frame, third word: the level was entered without changing the ray

struct Lvl { // the accel level a lane is walking
    uint32_t accel, node_base, prim_base, soup_delta, flags;
};
// Node cursor of the second formulation.  Global tables: the node's index in P.nodes (64-byte DNode records).
// LDS image: the node's BYTE offset in the image -- every record of the image carries "walk words" made by the host
// (capi.cpp, the image builder): an interior node its second child's cursor and 1 << axis, a leaf its first slot, NODE_LEAF
// and its last slot + 1 -- so a step forms the record's address with one add, never multiplies or shifts, and takes a
// leaf's slot range as it is.
constexpr uint32_t LDS_NODE_BYTES = LDS_NODE_STRIDE * 16u;
constexpr uint32_t LDS_NODE_WALK_OFF = 64u; // words 16..19 of the 80-byte record
// What the walk needs of a DAccel: from the LDS image (LDS_ACCEL_UNITS) or from the table in HBM / L2
template <bool LDSS, bool FAST = false>
__device__ __forceinline__ void lvl_set(const DParams &P, const uint4 *scn, Lvl &L, uint32_t accel) {
    L.accel = accel;
    if (LDSS) {
        const uint4 info = scn[P.lds_accel_off + accel * LDS_ACCEL_UNITS + 6u];
        L.node_base = info.x; L.prim_base = info.y; L.soup_delta = info.z; L.flags = info.w;
    } else {
        const DAccel *A = P.accels + accel;
        L.node_base = FAST ? A->fnode_base : A->node_base; L.prim_base = FAST ? A->fprim_base : A->prim_base; L.soup_delta = 0u; L.flags = A->flags;
    }
}
template <bool LDSS>
__device__ __forceinline__ Ray accel_local_ray(const DParams &P, const uint4 *scn, uint32_t accel, const Ray &r) { // inverse_transform_ray (bvh.rs:462)
    if (LDSS) {
        const double2 *q = reinterpret_cast<const double2 *>(scn + (P.lds_accel_off + accel * LDS_ACCEL_UNITS));
        const double2 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5];
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
template <int KZ, bool LDSS, bool FAST = false, bool COUNT = false>
__device__ __forceinline__ bool mesh_leaf2(const DParams &P, const uint4 *scn, const V3 o, const TriSetup tri, uint32_t li, const uint32_t le,
                                           const uint32_t soup_delta, const uint32_t accel, const bool anyhit, Best &best, bool &tie, Counters &cnt) {
    const char *base = reinterpret_cast<const char *>(P.leaf_soup);
    constexpr uint32_t REC = (uint32_t)sizeof(DLeafRec);
    uint32_t off = (li + soup_delta) * REC; // (the array holds < 2^32 / 48 slots: checked by the host)
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
// FAST (lg_accel_set_mode(1), opt-in, NOT the reference's traversal): the same walk over the binned-SAH trees with <= 4 primitives
// per leaf, near child first by dir_is_neg[axis] as before, and a node is skipped when its slab tnear lies beyond the best hit so
// far (closest) or beyond the light (any-hit) -- margins as in prune_limit().  Exact ties in t (and NaN t), where the reference's
// visit order decides, raise `tie`; the caller (walk() below) then puts the winner to the reference tree's own box tests
// (ref_candidate) and re-traces with the reference walk when either fails.
//
// PRUNE (the reference tree, the reference's visit order; DESIGN.md section 3.5 has the derivations): a node is skipped when,
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
template <bool LDSS, bool FAST = false, bool PRUNE = false, bool COUNT = false>
__device__ __forceinline__ void traverse_ref(const DParams &P, const Ray &wray, const bool anyhit, uint32_t *stack, const uint32_t stride,
                                             Best &best, const uint4 *scn, bool &tie, Counters &cnt) {
    static_assert(!(FAST && LDSS), "the LDS-resident scene holds the reference tree only");
    static_assert(!(FAST && PRUNE), "the fast mode prunes its own trees by its own rule");
#ifdef LG_STAMPS
    unsigned long long stamp_acc[7] = {0, 0, 0, 0, 0, 0, 0}, stamp_t = __builtin_readcyclecounter();
    unsigned long long stamp_cnt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    best.t = INFINITY; best.ref = NO_HIT; best.accel = 0;
    if (COUNT) cnt.entries++; // the root accel
    uint32_t *const stk = stack + stride; // entry -1 of an empty stack is fetched (never used): one guard entry below
    Lvl L;
    lvl_set<LDSS, FAST>(P, scn, L, 0u);
    double limit = prune_limit(INFINITY, anyhit); // FAST: nodes whose tnear lies beyond this are skipped
    TriSetup tri;                                  // FAST: per mesh level (its leaves hold <= 4 triangles: per leaf the three divides would dominate)
    tri.kz = 0; tri.sx = 0.0; tri.sy = 0.0; tri.sz = 0.0;
    // ---- the root accel's local ray (bvh.rs:462), kept for the returns
    Ray root = wray;
    if (!((L.flags & AF_IDENTITY) && ray_plain(wray))) root = accel_local_ray<LDSS>(P, scn, 0u, wray);
    Ray ray = root;
    double dd = dot(ray.d, ray.d);      // a of every sphere's quadratic at this level
    double four_a = 4.0 * dd;           // 4.0 * a of its discriminant b*b - 4.0*a*c (core/math.rs:16: (4.0 * a) * c)
    uint32_t negmask = neg_mask(ray);   // dir_is_neg (bvh.rs:463)
    uint32_t sp = 0, base = 0, cur = L.node_base, li = 0, le = 0, enter = 0;
    uint32_t state = ST_NODE;
    // ---- PRUNE: per-axis limits of the level the lane is in, and the level's margin
    V3 plim{INFINITY, INFINITY, INFINITY};
    double peps = INFINITY;
    auto prune_limits = [&](const double limit) { // limit >= 0 (every accepted t is), or +inf before the first hit
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
        if (LDSS) {
            const double2 *q = reinterpret_cast<const double2 *>(scn + (P.lds_accel_off + L.accel * LDS_ACCEL_UNITS + 10u));
            const double2 a = q[0], b = q[1], e = q[2];
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
        bool more_nodes = wave_any(state == ST_NODE);
        while (more_nodes) {
#ifdef LG_STAMPS
            stamp_cnt[5] += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == ST_NODE));
#endif
            if (state == ST_NODE) {
                // the record's walk words: interior -> (second child's cursor, 1 << split axis, -), leaf -> (first slot, NODE_LEAF, last slot + 1)
                double bmin[3], bmax[3];
                uint32_t w_link, w_meta, w_end;
                if (LDSS) {
                    const char *rec = reinterpret_cast<const char *>(scn) + cur;
                    const double2 *q = reinterpret_cast<const double2 *>(rec);
                    const double2 a = q[0], b = q[1], c = q[2]; // four ds_read_b128
                    const uint4 d = *reinterpret_cast<const uint4 *>(rec + LDS_NODE_WALK_OFF);
                    bmin[0] = a.x; bmin[1] = a.y; bmin[2] = b.x; bmax[0] = b.y; bmax[1] = c.x; bmax[2] = c.y;
                    w_link = d.x; w_meta = d.y; w_end = d.z;
                } else {
                    const NodeRec nd = load_node<false>(P, scn, cur);
                    bmin[0] = nd.bmin[0]; bmin[1] = nd.bmin[1]; bmin[2] = nd.bmin[2]; bmax[0] = nd.bmax[0]; bmax[1] = nd.bmax[1]; bmax[2] = nd.bmax[2];
                    const bool lf = (nd.meta & NODE_LEAF) != 0u;
                    w_link = (lf ? L.prim_base : L.node_base) + nd.link;
                    w_meta = (lf ? NODE_LEAF : 1u << (nd.meta & 3u)) | (nd.meta & NODE_NOPRUNE);
                    w_end = w_link + (nd.meta & 0xFFFFu);
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
                    hit = slab_intersects_nc_axes(bmin, bmax, ray, tx, ty, tz);
                    const bool beyond = tx > plim.x || ty > plim.y || tz > plim.z; // (a NaN entry parameter compares false)
                    hit = hit && !(beyond && (w_meta & NODE_NOPRUNE) == 0u);
                } else hit = slab_intersects_nc(bmin, bmax, ray);
                if (COUNT) { cnt.nodes++; dbg_event(P, 2.0 + (hit ? 0.1 : 0.0), (double)L.accel, (double)cur, (double)w_meta); }
                const bool leaf = (int32_t)w_meta < 0;            // n_primitives > 0 (bvh.rs:475): the builder emits no empty leaf
                const bool neg = (negmask & w_meta) != 0u;        // dir_is_neg[axis] (bvh.rs:496)
                const uint32_t first = cur + (LDSS ? LDS_NODE_BYTES : 1u), second = w_link; // the two children (interior nodes)
                const uint32_t near_node = neg ? second : first, far_node = neg ? first : second;
                const bool leaf_hit = hit && leaf, interior_hit = hit != leaf_hit;
                const bool pop = !hit, can_pop = sp != base;
                stk[sp * stride] = far_node; // counts only if sp advances (bvh.rs:493-504)
                cur = interior_hit ? near_node : popped;
                sp = sp + (interior_hit ? 1u : 0u) - (pop && can_pop ? 1u : 0u);
                li = w_link; le = w_end; // (read in ST_LEAF only)
                state = leaf_hit ? ST_LEAF : (pop && !can_pop) ? ST_LEVEL_DONE : ST_NODE;
            }
#ifdef LG_STAMPS
            stamp_cnt[0] += 1; // (lanes that took this step: those whose state was ST_NODE when it began -- counted after it as "not idle")
            stamp_cnt[6] += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == ST_DONE));
#endif
            more_nodes = wave_any(state == ST_NODE);
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
            if (tri.kz == 0) done = mesh_leaf2<0, LDSS, FAST, COUNT>(P, scn, ray.o, tri, li, le, L.soup_delta, L.accel, anyhit, best, tie, cnt);
            else if (tri.kz == 1) done = mesh_leaf2<1, LDSS, FAST, COUNT>(P, scn, ray.o, tri, li, le, L.soup_delta, L.accel, anyhit, best, tie, cnt);
            else done = mesh_leaf2<2, LDSS, FAST, COUNT>(P, scn, ray.o, tri, li, le, L.soup_delta, L.accel, anyhit, best, tie, cnt);
            if (FAST) limit = prune_limit(best.t, anyhit);
            if (PRUNE && !anyhit) prune_limits(best.t);
            if (done) state = ST_DONE;
            else if (sp != base) { --sp; cur = stk[sp * stride]; state = ST_NODE; }
            else state = ST_LEVEL_DONE;
        }
        LG_STAMP(2);
        bool more_prims = wave_any(state == ST_LEAF);
        while (more_prims) {
#ifdef LG_STAMPS
            stamp_cnt[7] += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == ST_LEAF));
            stamp_cnt[8] += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == ST_DONE));
#endif
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
                    if (PRUNE && !anyhit) prune_limits(t);
                    if (anyhit && t < 1.0) state = ST_DONE; // occluded: point.rs:49 only asks isect.t < 1.0
                }
                if (state == ST_LEAF && li >= le) { // leaf exhausted: next pending node of this level, or the level is done
                    if (sp != base) { --sp; cur = popped; state = ST_NODE; }
                    else state = ST_LEVEL_DONE;
                }
            }
            more_prims = wave_any(state == ST_LEAF);
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
            if (PRUNE) prune_level();
            cur = L.node_base;
            state = ST_NODE; // node 0 is tested when visited (bvh.rs:472-473)
        }
        LG_STAMP(4);
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
                    const uint4 *rec = scn + (P.lds_accel_off + L.accel * LDS_ACCEL_UNITS);
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
                if (PRUNE) prune_level();
                if (li < le) state = ST_LEAF;
                else if (sp != base) { --sp; cur = stk[sp * stride]; state = ST_NODE; }
                else state = ST_LEVEL_DONE; // the parent level is exhausted as well
            }
        }
        LG_STAMP(5);
#ifdef LG_STAMPS
        stamp_acc[6] += 1;
#endif
        if (!wave_any(state != ST_DONE)) break;
    }
#ifdef LG_STAMPS
    if ((threadIdx.x & 63u) == 0u && P.stats) {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(P.stats);
        for (int i = 0; i < 7; ++i) atomicAdd(dst + i, stamp_acc[i]);
        atomicAdd(dst + 7, 1ull);
        unsigned long long *cnt = P.stamp_counts;
        if (cnt) for (int i = 0; i < 9; ++i) atomicAdd(cnt + i, stamp_cnt[i]);
    }
#endif
}

// Fast mode's walk with CHILD-PAIR records (nodes2): `cur` is an interior node whose own box is known to be hit; one 128-byte
// record holds both children's boxes and leaf words, so a step tests two boxes per dependent fetch -- the fast walk waits on
// those fetches, not on the VALU (half the fetches of the one-node-per-step form).  The nearer hit child (by slab tnear) is
// taken, the other is pushed (one word: node index, bit 31 = it is a leaf); a popped leaf entry fetches its slot range
// (ST_OPEN).  Everything else -- levels, leaves, pruning margins, tie / NaN flags -- is traverse_ref<false, true>.
constexpr uint32_t ST_OPEN = 5u;
constexpr uint32_t STK_LEAF = 0x80000000u;
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
    TriSetup tri;                                  // FAST: per mesh level (its leaves hold <= 4 triangles: per leaf the three divides would dominate)
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
    for (;;) {
        // ---- phase A: interior nodes, two children per step, until no lane of the wave is at a node
        // (loops are written with their wave-uniform condition in a variable tested at the bottom: hipcc then keeps the
        // loop-carried state in place instead of copying it in and out of the loop on every trip)
        bool more_nodes = wave_any(state == ST_NODE);
        while (more_nodes) {
            if (state == ST_NODE) {
                const DNode2 *nd = P.nodes2 + cur;
                const double b0min[3] = {nd->b0min[0], nd->b0min[1], nd->b0min[2]}, b0max[3] = {nd->b0max[0], nd->b0max[1], nd->b0max[2]};
                const double b1min[3] = {nd->b1min[0], nd->b1min[1], nd->b1min[2]}, b1max[3] = {nd->b1max[0], nd->b1max[1], nd->b1max[2]};
                const uint32_t link0 = nd->link0, meta0 = nd->meta0, link1 = nd->link1, meta1 = nd->meta1, second = L.node_base + nd->second;
                const uint32_t popped = stk[(int)(sp - 1u) * (int)stride];
                if (COUNT) cnt.nodes += 2u; // both children's boxes
                // a primitive's computed t can undershoot its box's tnear by the error of its own formula: for a sphere
                // the quadratic's cancellation, ~sqrt(eps) of the distance to its centre, which lies inside the box
                double tn0, tf0, tn1, tf1;
                bool hit0 = slab_intersects_nc_t(b0min, b0max, ray, tn0, tf0);
                bool hit1 = slab_intersects_nc_t(b1min, b1max, ray, tn1, tf1);
                hit0 = hit0 && !(tn0 - 4e-8 * fabs(tf0) > limit);
                hit1 = hit1 && !(tn1 - 4e-8 * fabs(tf1) > limit);
                const bool swap = hit1 && (!hit0 || tn1 < tn0); // the nearer hit child first
                const bool any = hit0 || hit1, both = hit0 && hit1;
                const uint32_t first = cur + 1u;
                const uint32_t near_idx = swap ? second : first, far_idx = swap ? first : second;
                const uint32_t near_link = swap ? link1 : link0, near_meta = swap ? meta1 : meta0, far_meta = swap ? meta0 : meta1;
                const bool near_leaf = (near_meta & NODE_LEAF) != 0u;
                const bool can_pop = sp != base;
                stk[sp * stride] = far_idx | ((far_meta & NODE_LEAF) ? STK_LEAF : 0u); // counts only if sp advances
                const uint32_t next = any ? near_idx : (popped & ~STK_LEAF);
                sp = sp + (both ? 1u : 0u) - (!any && can_pop ? 1u : 0u);
                li = L.prim_base + near_link; le = li + (near_meta & 0xFFFFu); // (read in ST_LEAF only)
                cur = next;
                state = any ? (near_leaf ? ST_LEAF : ST_NODE) : !can_pop ? ST_LEVEL_DONE : (popped & STK_LEAF) ? ST_OPEN : ST_NODE;
            }
            more_nodes = wave_any(state == ST_NODE);
        }
        // ---- phase B: leaf primitives in order[] sequence (bvh.rs:481-488)
        const bool mesh = (L.flags & AF_MESH) != 0u;
        if (state == ST_LEAF && mesh) { // every slot of a mesh accel is a triangle
            if (!FAST) tri = tri_setup(ray); // per fat leaf: amortises the three divides (triangle.rs:186-201)
            bool done;
            if (tri.kz == 0) done = mesh_leaf2<0, LDSS, FAST, COUNT>(P, scn, ray.o, tri, li, le, L.soup_delta, L.accel, anyhit, best, tie, cnt);
            else if (tri.kz == 1) done = mesh_leaf2<1, LDSS, FAST, COUNT>(P, scn, ray.o, tri, li, le, L.soup_delta, L.accel, anyhit, best, tie, cnt);
            else done = mesh_leaf2<2, LDSS, FAST, COUNT>(P, scn, ray.o, tri, li, le, L.soup_delta, L.accel, anyhit, best, tie, cnt);
            if (FAST) limit = prune_limit(best.t, anyhit);
            if (done) state = ST_DONE;
            else if (sp != base) { --sp; const uint32_t e = stk[sp * stride]; cur = e & ~STK_LEAF; state = (e & STK_LEAF) ? ST_OPEN : ST_NODE; }
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
                    if (sp != base) { --sp; cur = popped & ~STK_LEAF; state = (popped & STK_LEAF) ? ST_OPEN : ST_NODE; }
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
        // ---- a pending child that is a leaf (pushed with its box already tested): its slot range
        if (state == ST_OPEN) {
            uint32_t link, meta;
            load_node_link<false>(P, scn, cur, link, meta);
            li = L.prim_base + link; le = li + (meta & 0xFFFFu);
            state = ST_LEAF;
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
                    const uint4 *rec = scn + (P.lds_accel_off + L.accel * LDS_ACCEL_UNITS);
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
                else if (sp != base) { --sp; const uint32_t e = stk[sp * stride]; cur = e & ~STK_LEAF; state = (e & STK_LEAF) ? ST_OPEN : ST_NODE; }
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
                                     const uint4 *scn, Counters &cnt) {
    bool tie = false;
#ifndef LG_FAST_ONE_NODE
    if (FAST) traverse_fast<COUNT>(P, ray, anyhit, stack, stride, best, scn, tie, cnt);
    else
#endif
    traverse_ref<LDSS, FAST, PRUNE, COUNT>(P, ray, anyhit, stack, stride, best, scn, tie, cnt);
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
    if (redo) traverse_ref<false, false, false, COUNT>(P, ray, anyhit, stack, stride, best, nullptr, tie, cnt);
}

// ------------------------------------------------------------------------------------------
// Packet traversal (streaming pipeline, reference tree): ONE walk per wavefront.
//
// The 64 rays of an 8x8 tile (primary) or of its hit points towards one light (shadow) visit
// almost the same nodes.  Instead of 64 private walks -- private node fetches, private stacks,
// private near/far selects -- the wave walks the UNION of its lanes' node sets once: the node and
// primitive records are fetched with wave-uniform addresses (one request, broadcast), the stack
// is one small per-wave array of (node, lane mask) in LDS, the near child is chosen by a vote of
// the lanes that hit the node (ballot + popcount on dir_is_neg[axis], bvh.rs:496), and a lane
// simply drops out of the mask at a node whose box it misses.
//
// Exactness.  A lane takes part in a primitive test iff every box on the path from the root to
// that leaf passed ITS OWN slab test -- exactly the reference's candidate set for that ray, since
// the reference never culls by t (cuboid.rs:120) -- and every test is the same arithmetic on the
// same operands.  The closest hit is the minimum of the accepted t over that set, which does not
// depend on the visiting order except (a) between candidates with exactly equal t, where the
// reference keeps the first one it visits, and (b) after a NaN t, which the reference's
// comparisons accept and which then accepts everything after it.  Both are detected per lane
// (`tie`) and that lane is re-traced with its private reference-order walk.  An occluded any-hit
// ray needs no re-trace: "some accepted t < 1 exists" is order-independent (NaN aside, as in
// the private walk).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ bool lane_in(unsigned long long m, uint32_t lane) { return ((m >> lane) & 1ull) != 0ull; }
constexpr uint32_t PKT_ENTRY = 4u; // dwords per wave-stack entry: {a, b, mask lo, mask hi}

__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
    return (unsigned long long)uni((uint32_t)v) | ((unsigned long long)uni((uint32_t)(v >> 32)) << 32);
}
// Records of the packet walk: the same LDS image as the private walks (every lane reads the SAME address here,
// so its padding is irrelevant), or the tables in HBM / L2 with wave-uniform addresses: one request per record and wave.
template <bool LDSS>
__device__ __forceinline__ NodeRec pkt_node(const DParams &P, const uint4 *scn, uint32_t idx) { return load_node<LDSS>(P, scn, idx); }
struct SlotRec { // one leaf slot: its primref and its 48-byte geometry record
    uint32_t ref;
    LeafRec g;
};
template <bool LDSS>
__device__ __forceinline__ SlotRec pkt_slot(const DParams &P, const uint4 *scn, uint32_t slot, bool mesh, uint32_t soup_delta) {
    SlotRec r;
    if (LDSS) {
        r.ref = reinterpret_cast<const uint32_t *>(scn + P.lds_prim_off)[slot];
        if (mesh) { // triangle records are not part of the LDS image
            r.g = load_rec(P, slot + soup_delta);
        } else {
            const uint4 *q = scn + (P.lds_soup_off + slot * 3u);
            r.g = LeafRec{q[0], q[1], q[2]};
        }
    } else {
        r.ref = P.primref[slot];
        r.g = load_rec(P, slot);
    }
    return r;
}
// The wave-uniform state lives in plain locals and every value that comes back from memory goes through
// readfirstlane, so that the compiler keeps it in SGPRs and branches on it with scalar branches.  (No
// by-reference lambdas here: state that round-trips through a private-memory capture is treated as divergent.)
#define PKT_SET_LEVEL(ACC, LOCAL)                                                                                      \
    do {                                                                                                               \
        const DAccel *A_ = P.accels + (ACC);                                                                           \
        accel = (ACC);                                                                                                 \
        ray = (LOCAL);                                                                                                 \
        dd = dot(ray.d, ray.d);                                                                                        \
        node_base = uni(LDSS ? A_->lnode_base : A_->node_base);                                                        \
        prim_base = uni(LDSS ? A_->lprim_base : A_->prim_base);                                                        \
        soup_delta = uni(LDSS ? A_->prim_base - A_->lprim_base : 0u);                                                  \
        mesh = (uni(A_->flags) & AF_MESH) != 0u;                                                                       \
        negbits = (ray.dinv.x < 0.0 ? 1u : 0u) | (ray.dinv.y < 0.0 ? 2u : 0u) | (ray.dinv.z < 0.0 ? 4u : 0u);          \
    } while (0)
/* every lane stores the same four words to the same address: one LDS write, no exec juggling */
#define PKT_PUSH(SP, A, B, M)                                                                                          \
    do {                                                                                                               \
        *reinterpret_cast<uint4 *>(ws + (SP) * PKT_ENTRY) = uint4{(A), (B), (uint32_t)(M), (uint32_t)((M) >> 32)};     \
    } while (0)
#define PKT_MASK(SP) ((unsigned long long)uni(ws[(SP) * PKT_ENTRY + 2u]) | ((unsigned long long)uni(ws[(SP) * PKT_ENTRY + 3u]) << 32))
// one candidate of this lane: tie / NaN bookkeeping, acceptance, any-hit exit
#define PKT_CANDIDATE(VALID, T, REF)                                                                                   \
    do {                                                                                                               \
        if (VALID) {                                                                                                   \
            const double t_ = (T);                                                                                     \
            if ((t_ == best.t && best.ref != NO_HIT) || t_ != t_) tie = true;                                          \
            if (!(t_ >= best.t)) {                                                                                     \
                best.t = t_; best.ref = (REF); best.accel = accel;                                                     \
                if (anyhit && t_ < 1.0) alive = false; /* point.rs:49 */                                               \
            }                                                                                                          \
        }                                                                                                              \
    } while (0)

template <bool LDSS>
__device__ __forceinline__ void traverse_packet(const DParams &P, const uint4 *scn, const Ray &wray, bool alive, const bool anyhit,
                                                uint32_t *ws, const uint32_t lane, Best &best, bool &tie) {
    best.t = INFINITY; best.ref = NO_HIT; best.accel = 0;
    tie = false;
    unsigned long long alive_m = __ballot(alive);
    if (alive_m == 0ull) return;
    // ---- level state: wave-uniform except the rays
    uint32_t accel = 0u, node_base = 0u, prim_base = 0u, soup_delta = 0u;
    bool mesh = false;
    Ray ray;
    double dd = 0.0;
    uint32_t negbits = 0u;                 // per lane: bit a set <=> dinv[a] < 0 (dir_is_neg, bvh.rs:463)
    const Ray root = ray_to_local(P.accels->minv, wray); // the world ray is not kept (see level_ray)
    const V3 root_o = root.o, root_d = root.d;
    PKT_SET_LEVEL(0u, root);
    uint32_t sp = 0u, base = 0u;           // wave stack, in entries
    uint32_t cur = 0u;                     // node to visit ...
    unsigned long long m = alive_m;        // ... by these lanes
    bool have = true;                      // (cur, m) is pending
    uint32_t li = 0u, le = 0u;             // leaf cursor (slots of this level's numbering) ...
    unsigned long long lm = 0ull;          // ... and the lanes inside the leaf
    bool leaf_open = false;
    for (;;) {
        // ---- phase A: interior nodes, one record fetch and one slab test per step for the whole wave
        // (readfirstlane pins: no-ops in hardware terms, they tell the compiler this state is wave-uniform)
        have = uni((uint32_t)have) != 0u; leaf_open = uni((uint32_t)leaf_open) != 0u;
        sp = uni(sp); base = uni(base); accel = uni(accel); node_base = uni(node_base); prim_base = uni(prim_base);
        alive_m = uni64(alive_m);
        while (have) {
            cur = uni(cur); m = uni64(m); sp = uni(sp);
            have = false;
            m &= alive_m;
            if (m == 0ull) break;
            const NodeRec nd = pkt_node<LDSS>(P, scn, node_base + cur);
            const uint32_t link = uni(nd.link), meta = uni(nd.meta);
            const bool inside_box = slab_intersects(nd.bmin, nd.bmax, ray); // every lane computes it: no exec juggling
            const unsigned long long hm = __ballot(inside_box) & m;
            if (hm == 0ull) break;
            if (meta & NODE_LEAF) {
                const uint32_t count = meta & 0xFFFFu;
                if (count != 0u) { li = prim_base + link; le = li + count; lm = hm; leaf_open = true; } // nprims as u16 == 0: nothing
                break;
            }
            const unsigned long long ng = __ballot(((negbits >> (meta & 3u)) & 1u) != 0u); // dir_is_neg[axis] of every lane
            const bool neg_first = 2 * __popcll(hm & ng) > __popcll(hm); // the vote: most lanes' near child first (bvh.rs:496)
            const uint32_t near_node = neg_first ? link : cur + 1u, far_node = neg_first ? cur + 1u : link;
            PKT_PUSH(sp, far_node, 0u, hm);
            ++sp;
            cur = near_node; m = hm; have = true;
        }
        // ---- phase B: the leaf's slots li .. le for the lanes lm, in order[] sequence, one slot ahead
        if (leaf_open) {
            leaf_open = false;
            li = uni(li); le = uni(le); lm = uni64(lm);
            lm &= alive_m;
            if (lm != 0ull) {
                TriSetup tri{2, 0.0, 0.0, 0.0};
                unsigned long long kz0 = 0ull, kz1 = 0ull, kz2 = 0ull;
                if (mesh) { // shear constants and dominant axis per fat leaf (as the private walk does)
                    tri = tri_setup(ray);
                    kz0 = __ballot(tri.kz == 0); kz1 = __ballot(tri.kz == 1); kz2 = __ballot(tri.kz == 2);
                }
                const uint32_t last = le - 1u;
                SlotRec nxt = pkt_slot<LDSS>(P, scn, li, mesh, soup_delta);
                while (li < le) {
                    li = uni(li); lm = uni64(lm);
                    const SlotRec s = nxt;
                    const uint32_t slot = li;
                    ++li;
                    nxt = pkt_slot<LDSS>(P, scn, li < last ? li : last, mesh, soup_delta); // prefetch (clamped: always a valid slot)
                    const uint32_t ref = uni(s.ref);
                    const uint32_t kind = ref >> 30, idx = ref & PRIM_INDEX_MASK;
                    const bool in = lane_in(lm, lane);
                    (void)slot;
                    if (kind == PK_TRIANGLE) {
                        const V3 p0{rec_f32(s.g.a.x), rec_f32(s.g.a.y), rec_f32(s.g.a.z)}, p1{rec_f32(s.g.a.w), rec_f32(s.g.b.x), rec_f32(s.g.b.y)},
                            p2{rec_f32(s.g.b.z), rec_f32(s.g.b.w), rec_f32(s.g.c.x)};
                        TriHit h;
                        if (mesh) { // the permutation is a per-ray property: one pass per dominant axis present among the leaf's lanes
                            if ((lm & kz0) != 0ull) { if (lane_in(lm & kz0, lane)) { const bool ok = triangle_t_pre<0>(p0, p1, p2, ray.o, tri.sx, tri.sy, tri.sz, h); PKT_CANDIDATE(ok, h.t, ref); } }
                            if ((lm & kz1) != 0ull) { if (lane_in(lm & kz1, lane)) { const bool ok = triangle_t_pre<1>(p0, p1, p2, ray.o, tri.sx, tri.sy, tri.sz, h); PKT_CANDIDATE(ok, h.t, ref); } }
                            if ((lm & kz2) != 0ull) { if (lane_in(lm & kz2, lane)) { const bool ok = triangle_t_pre<2>(p0, p1, p2, ray.o, tri.sx, tri.sy, tri.sz, h); PKT_CANDIDATE(ok, h.t, ref); } }
                        } else if (in) { // a triangle outside a mesh accel cannot be built by the scene API; kept for completeness
                            const bool ok = triangle_t(p0, p1, p2, ray, h);
                            PKT_CANDIDATE(ok, h.t, ref);
                        }
                    } else if (kind == PK_SPHERE) {
                        const V3 cen{rec_f64(s.g.a.x, s.g.a.y), rec_f64(s.g.a.z, s.g.a.w), rec_f64(s.g.b.x, s.g.b.y)};
                        const double rad = rec_f64(s.g.b.z, s.g.b.w);
                        // the discriminant for every lane first; the roots (sqrt, two divides) only if some lane of the leaf needs them
                        const V3 l = ray.o - cen;
                        const double b = 2.0 * dot(ray.d, l);
                        const double c = dot(l, l) - rad * rad;
                        const double disc = b * b - 4.0 * dd * c;
                        const bool need = in && (dd == 0.0 || !(disc < 0.0));
                        if (__ballot(need) != 0ull) {
                            if (need) {
                                bool inside;
                                const double t = sphere_t_a(ray, dd, cen, rad, inside);
                                PKT_CANDIDATE(!(t < 0.0), t, ref);
                            }
                        }
                    } else if (kind == PK_CUBOID) {
                        if (in) {
                            double mn[3] = {rec_f64(s.g.a.x, s.g.a.y), rec_f64(s.g.a.z, s.g.a.w), rec_f64(s.g.b.x, s.g.b.y)};
                            double mx[3] = {rec_f64(s.g.b.z, s.g.b.w), rec_f64(s.g.c.x, s.g.c.y), rec_f64(s.g.c.z, s.g.c.w)};
                            V3 d0, d1;
                            double t = 0.0;
                            const bool ok = cuboid_hit<false>(mn, mx, ray, t, d0, d1);
                            PKT_CANDIDATE(ok, t, ref);
                        }
                    } else { // PK_ACCEL -- nested BVHAccel: every lane of the leaf enters it (bvh.rs:483-488, 462)
                        PKT_PUSH(sp, li, le, lm);
                        PKT_PUSH(sp + 1u, base, 0u, 0ull);
                        sp += 2u; base = sp;
                        PKT_SET_LEVEL(idx, ray_to_local(P.accels[idx].minv, ray));
                        cur = 0u; m = lm; have = true;
                        break;
                    }
                    if (anyhit) {
                        alive_m = __ballot(alive);
                        lm &= alive_m;
                        if (lm == 0ull) break;
                    }
                }
            }
        }
        // ---- phase C: next pending (node, mask) of this level, or back to the parent's leaf
        if (!have) {
            if (sp != base) {
                --sp;
                cur = uni(ws[sp * PKT_ENTRY]); m = PKT_MASK(sp);
                have = true;
            } else {
                if (accel == 0u) break;
                sp -= 2u;                  // level frame: {li, le, leaf mask} {previous base}
                li = uni(ws[sp * PKT_ENTRY]); le = uni(ws[sp * PKT_ENTRY + 1u]); lm = PKT_MASK(sp);
                base = uni(ws[(sp + 1u) * PKT_ENTRY]);
                const uint32_t parent = uni((uint32_t)P.accels[accel].parent);
                PKT_SET_LEVEL(parent, level_ray(P, root_o, root_d, parent)); // recomputed, bit-identical to the first computation
                leaf_open = li < le;
            }
        }
    }
}
#undef PKT_SET_LEVEL
#undef PKT_PUSH
#undef PKT_MASK
#undef PKT_CANDIDATE

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

// ------------------------------------------------------------------------------------------
// BxDFs (core/bxdf/*.rs) in shading space
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double cos2_theta(V3 w) { return w.z * w.z; }
__device__ __forceinline__ double abs_cos_theta(V3 w) { return fabs(w.z); }
__device__ __forceinline__ double sin2_theta(V3 w) { return fmax_(1.0 - cos2_theta(w), 0.0); }
__device__ __forceinline__ double sin_theta(V3 w) { return sqrt(sin2_theta(w)); }
__device__ __forceinline__ double tan_theta(V3 w) { return sin_theta(w) / w.z; }
__device__ __forceinline__ double tan2_theta(V3 w) { return sin2_theta(w) / cos2_theta(w); }
__device__ __forceinline__ double cos_phi(V3 w) { double s = sin_theta(w); return s == 0.0 ? 1.0 : fmin_(fmax_(w.x / s, -1.0), 1.0); }
__device__ __forceinline__ double sin_phi(V3 w) { double s = sin_theta(w); return s == 0.0 ? 0.0 : fmin_(fmax_(w.y / s, -1.0), 1.0); }

__device__ __forceinline__ double fr_dielectric(double cos_i, double eta_i, double eta_t) { // fresnel.rs:37-64
    cos_i = fmin_(fmax_(cos_i, -1.0), 1.0);
    bool entering = cos_i > 0.0;
    if (!entering) { double tmp = eta_i; eta_i = eta_t; eta_t = tmp; cos_i = fabs(cos_i); }
    double sin_i = sqrt(fmax_(1.0 - cos_i * cos_i, 0.0));
    double sin_t = eta_i / eta_t * sin_i;
    if (sin_t >= 1.0) return 1.0;
    double cos_t = sqrt(fmax_(1.0 - sin_t * sin_t, 0.0));
    double r_parl = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
    double r_perp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
    return (r_parl * r_parl + r_perp * r_perp) * 0.5;
}
__device__ __forceinline__ V3 fr_conductor(double cos_i, V3 eta_i, V3 eta_t, V3 k) { // fresnel.rs:69-91
    cos_i = fmin_(fmax_(cos_i, -1.0), 1.0);
    V3 eta = div_ew(eta_t, eta_i);
    V3 etak = div_ew(k, eta_i);
    double c2 = cos_i * cos_i;
    double s2 = 1.0 - c2;
    V3 eta2 = mul_ew(eta, eta), etak2 = mul_ew(etak, etak);
    V3 t0 = eta2 - etak2 - splat(s2);
    V3 a2plusb2 = vsqrt(mul_ew(t0, t0) + 4.0 * mul_ew(eta2, etak2));
    V3 t1 = a2plusb2 + splat(c2);
    V3 a = vsqrt(0.5 * (a2plusb2 + t0));
    V3 t2 = 2.0 * cos_i * a;
    V3 rs = div_ew(t1 - t2, t1 + t2);
    V3 t3 = c2 * a2plusb2 + splat(s2 * s2);
    V3 t4 = t2 * s2;
    V3 rp = div_ew(mul_ew(rs, t3 - t4), t3 + t4);
    return 0.5 * (rp + rs);
}
__device__ __forceinline__ double tr_d(double ax, double ay, V3 wh) { // microfacet.rs:31-40
    double tan2 = tan2_theta(wh);
    if (isinf(tan2)) return 0.0;
    double cos4 = cos2_theta(wh) * cos2_theta(wh);
    double cp = cos_phi(wh), sp = sin_phi(wh);
    double e = ((cp * cp) / (ax * ax) + (sp * sp) / (ay * ay)) * tan2;
    return 1.0 / (PI * ax * ay * cos4 * (1.0 + e) * (1.0 + e));
}
__device__ __forceinline__ double tr_lambda(double ax, double ay, V3 w) { // microfacet.rs:55-66
    double abs_tan = fabs(tan_theta(w));
    if (isinf(abs_tan)) return 0.0;
    double cp = cos_phi(w), sp = sin_phi(w);
    double alpha = sqrt((cp * cp) * ax * ax + (sp * sp) * ay * ay);
    double a2t2 = (alpha * abs_tan) * (alpha * abs_tan);
    return (sqrt(1.0 + a2t2) - 1.0) / 2.0;
}
// microfacet::Reflection::f (microfacet.rs:101-115); conductor selects Substance::Conductor(1, eta, k)
__device__ __forceinline__ V3 microfacet_f(V3 r, bool conductor, double d_eta_i, double d_eta_t, V3 c_eta, V3 c_k, double ax, double ay, V3 wo, V3 wi) {
    double cos_o = abs_cos_theta(wo), cos_i = abs_cos_theta(wi);
    V3 wh = wi + wo;
    if (cos_i == 0.0 || cos_o == 0.0) return vzero();
    if (wh.x == 0.0 && wh.y == 0.0 && wh.z == 0.0) return vzero();
    wh = normalize(wh);
    double ci = dot(wi, wh);
    V3 spectrum = conductor ? fr_conductor(ci, splat(1.0), c_eta, c_k) : splat(fr_dielectric(ci, d_eta_i, d_eta_t));
    double g = 1.0 / (1.0 + tr_lambda(ax, ay, wo) + tr_lambda(ax, ay, wi));
    return mul_ew(r * tr_d(ax, ay, wh) * g, spectrum) / (4.0 * cos_i * cos_o);
}
__device__ __forceinline__ V3 oren_nayar_f(V3 r, double sigma_deg, V3 wo, V3 wi) { // diffuse.rs:29-56
    double s = sigma_deg * (PI / 180.0), s2 = s * s;
    double A = 1.0 - (s2 / 2.0 * (s2 + 0.33));
    double B = 0.45 * s2 / (s2 + 0.09);
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
    return r * FRAC_1_PI * (A + B * max_cos * sin_alpha * tan_beta);
}

// Shading frame of one hit: what Material::scattering + BSDF::new keep (bsdf.rs:29-46)
struct Shade {
    V3 praw;   // interaction.p = ray.origin + ray.d * t (surface.rs:169)
    V3 p;      // interaction.p + interaction.p_err (integrate.rs:40)
    V3 pm;     // interaction.p - interaction.p_err (integrate.rs:127)
    V3 wo, ng, ns, ss, ts;
    int32_t mat;
};

// BSDF::f (bsdf.rs:73-92) with the BxDF list of Material::scattering (material/*.rs) inlined
__device__ __forceinline__ V3 bsdf_f(const DMaterial &m, const Shade &sh, V3 wo, V3 wi) {
    bool reflect = dot(wi, sh.ng) * dot(wo, sh.ng) > 0.0;
    V3 wo_l{dot(wo, sh.ss), dot(wo, sh.ts), dot(wo, sh.ns)};
    V3 wi_l{dot(wi, sh.ss), dot(wi, sh.ts), dot(wi, sh.ns)};
    if (wo_l.z == 0.0) return vzero();
    V3 f = vzero();
    switch (m.kind) {
    case MAT_MATTE: { // matte.rs:18-26 -- REFLECTION | DIFFUSE
        if (reflect) {
            V3 kd{m.p[0], m.p[1], m.p[2]};
            f = f + (m.p[3] == 0.0 ? kd * FRAC_1_PI : oren_nayar_f(kd, m.p[3], wo_l, wi_l));
        }
        break;
    }
    case MAT_PLASTIC: { // plastic.rs:20-37 -- Lambertian then microfacet reflection, both REFLECTION
        if (reflect) {
            V3 kd{m.p[0], m.p[1], m.p[2]}, ks{m.p[3], m.p[4], m.p[5]};
            if (vne(kd, vzero())) f = f + kd * FRAC_1_PI;
            if (vne(ks, vzero())) f = f + microfacet_f(ks, false, 1.0, 1.5, vzero(), vzero(), m.p[6], m.p[6], wo_l, wi_l);
        }
        break;
    }
    case MAT_METAL: { // metal.rs:17-26
        if (reflect) {
            V3 eta{m.p[0], m.p[1], m.p[2]}, k{m.p[3], m.p[4], m.p[5]};
            f = f + microfacet_f(splat(1.0), true, 0.0, 0.0, eta, k, m.p[6], m.p[7], wo_l, wi_l);
        }
        break;
    }
    case MAT_GLASS: { // glass.rs:33-56: specular BxDFs evaluate to zero (bxdf/mod.rs:172)
        V3 kr{m.p[0], m.p[1], m.p[2]}, kt{m.p[3], m.p[4], m.p[5]};
        if (reflect && vne(kr, vzero())) f = f + vzero();
        if (!reflect && vne(kt, vzero())) f = f + vzero();
        break;
    }
    default: // MAT_MIRROR, mirror.rs:15-17
        if (reflect) f = f + vzero();
        break;
    }
    return f;
}

struct Sample { // bxdf::LightSample
    V3 spectrum, wi;
    double pdf;
};
__device__ __forceinline__ V3 to_world(const Shade &sh, V3 v) { // bsdf.rs:165-171
    return V3{sh.ss.x * v.x + sh.ts.x * v.y + sh.ns.x * v.z, sh.ss.y * v.x + sh.ts.y * v.y + sh.ns.y * v.z,
              sh.ss.z * v.x + sh.ts.z * v.y + sh.ns.z * v.z};
}
__device__ __forceinline__ V3 clamp01(V3 v) {
    return V3{fmin_(fmax_(v.x, 0.0), 1.0), fmin_(fmax_(v.y, 0.0), 1.0), fmin_(fmax_(v.z, 0.0), 1.0)};
}
// BSDF::sample_f(wo, (0.5, 0.5), REFLECTION | SPECULAR) (bsdf.rs:94-145, specular.rs:17-24):
// only Mirror and Glass(kr != 0) own a matching component; exactly one, so comp = 0 and pdf / 1.
__device__ __forceinline__ bool sample_specular_reflection(const DMaterial &m, const Shade &sh, Sample &s) {
    bool glass = m.kind == MAT_GLASS;
    if (!(m.kind == MAT_MIRROR || glass)) return false;
    V3 r{m.p[0], m.p[1], m.p[2]};
    if (glass && !vne(r, vzero())) return false; // component not present
    V3 wo_l{dot(sh.wo, sh.ss), dot(sh.wo, sh.ts), dot(sh.wo, sh.ns)};
    if (wo_l.z == 0.0) return false; // LightSample::zero(): pdf 0 -> caller returns zero
    V3 wi_l{-wo_l.x, -wo_l.y, wo_l.z};
    V3 fr = glass ? splat(fr_dielectric(wi_l.z, 1.0, m.p[6])) : splat(1.0);
    V3 spectrum = mul_ew(fr, r) / abs_cos_theta(wi_l);
    s.wi = to_world(sh, wi_l);
    s.spectrum = clamp01(spectrum);
    s.pdf = 1.0 / 1.0;
    return true;
}
// BSDF::sample_f(wo, (0.5, 0.5), TRANSMISSION | SPECULAR) (specular.rs:43-63, bxdf/mod.rs:276-288)
__device__ __forceinline__ bool sample_specular_transmission(const DMaterial &m, const Shade &sh, Sample &s) {
    if (m.kind != MAT_GLASS) return false;
    V3 kt{m.p[3], m.p[4], m.p[5]};
    if (!vne(kt, vzero())) return false;
    double eta_a = 1.0, eta_b = m.p[6];
    V3 wo_l{dot(sh.wo, sh.ss), dot(sh.wo, sh.ts), dot(sh.wo, sh.ns)};
    if (wo_l.z == 0.0) return false;
    bool entering = wo_l.z > 0.0;
    double eta_i = entering ? eta_a : eta_b, eta_t = entering ? eta_b : eta_a;
    double eta = eta_i / eta_t;
    // refract(wo, n = (0,0,1), eta)
    V3 n{0.0, 0.0, 1.0};
    double cos_i = dot(n, wo_l);
    double sin2_i = fmax_(1.0 - cos_i * cos_i, 0.0);
    double sin2_t = eta * eta * sin2_i;
    if (sin2_t >= 1.0) return false; // total internal reflection: LightSample::zero()
    double cos_t = sqrt(1.0 - sin2_t);
    V3 wi_l = eta * -1.0 * wo_l + (eta * cos_i - cos_t) * n;
    V3 spectrum = mul_ew(kt, splat(1.0) - splat(fr_dielectric(wi_l.z, eta_a, eta_b))) / abs_cos_theta(wi_l);
    s.wi = to_world(sh, wi_l);
    s.spectrum = clamp01(spectrum);
    s.pdf = 1.0 / 1.0;
    return true;
}

// ------------------------------------------------------------------------------------------
// the render kernel
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t to_byte(double c) { // img.rs:65-67
    return (uint32_t)as_u8(round(fmin_(fmax_(c, 0.0), 1.0) * 255.0));
}
__device__ __forceinline__ V3 background(const DParams &P, V3 d) { // background.rs:25-34; powf(2.) == x*x
    double a = fabs(dot(V3{0.0, 0.0, 1.0}, d));
    double t = fmin_(sqrt(1.0 - a * a) / P.bg_scale, 1.0);
    return V3{lerp(t, P.bg_inner.x, P.bg_outer.x), lerp(t, P.bg_inner.y, P.bg_outer.y), lerp(t, P.bg_inner.z, P.bg_outer.z)};
}

// frame fields
enum { FR_ACC = 0, FR_SPEC_R = 3, FR_STATE = 6, FR_TO = 7, FR_TD = 10, FR_SPEC_T = 13, FR_A = 16, FR_PDF = 17 };
__device__ __forceinline__ double &frame_at(const DParams &P, uint32_t depth, int field, unsigned long long gtid) {
    return P.frames[((unsigned long long)depth * FRAME_DOUBLES + field) * P.frame_threads + gtid];
}
__device__ __forceinline__ void frame_put3(const DParams &P, uint32_t depth, int field, unsigned long long g, V3 v) {
    frame_at(P, depth, field, g) = v.x; frame_at(P, depth, field + 1, g) = v.y; frame_at(P, depth, field + 2, g) = v.z;
}
__device__ __forceinline__ V3 frame_get3(const DParams &P, uint32_t depth, int field, unsigned long long g) {
    return V3{frame_at(P, depth, field, g), frame_at(P, depth, field + 1, g), frame_at(P, depth, field + 2, g)};
}

// Which pixel a work item (tile, lane) renders and where its result goes (lib.rs:110-162 addresses pixels by
// offset = y * w + x; the three modes are three ways of enumerating offsets).
struct Pixel {
    uint32_t x, y;
    unsigned long long pix; // index into out_rgba (x4 bytes) / out_radiance (x3 doubles)
    bool active;
};
__device__ __forceinline__ Pixel pixel_of(const DParams &P, uint32_t tile, uint32_t lane) {
    Pixel px;
    if (P.mode == 0) {
        uint32_t tx = tile % P.tiles_x, ty = tile / P.tiles_x;
        px.x = P.x0 + tx * 8u + (lane & 7u);
        const uint32_t vy = P.y0 + ty * 8u + (lane >> 3); // row of the output buffer's addressing (== y unless rows are interleaved)
        px.active = px.x < P.x1 && vy < P.y1;
        px.y = P.ilv_n > 1u ? ((vy / P.ilv_b) * P.ilv_n + P.ilv_r) * P.ilv_b + vy % P.ilv_b : vy;
        px.pix = (unsigned long long)(vy - P.out_row0) * P.out_pitch + (px.x - P.out_x0);
    } else {
        unsigned long long i = (unsigned long long)tile * 64ull + lane;
        px.active = i < P.sub_count;
        // mode 1: the strided subset {k + i*n} (lib.rs:152); mode 2: an explicit list of pixel offsets
        unsigned long long off = P.mode == 1 ? P.sub_k + i * P.sub_n : (px.active ? P.pixel_list[i] : 0ull);
        px.x = (uint32_t)(off % P.w);
        px.y = (uint32_t)(off / P.w);
        px.pix = P.out_compact ? i : off;
    }
    return px;
}

extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];

// Shading frame of a hit from the ray that found it (resolve_hit + SurfaceInteraction::from,
// surface.rs:158-183).  Pure function of (ray, best): recomputed after each shadow traversal
// instead of being kept in registers across it, which is what lets 4-5 waves share a SIMD.
__device__ __forceinline__ void shade_frame(const DParams &P, const Ray &ray, const Best &best, Shade &sh) {
    Isect is;
    sh.mat = resolve_hit(P, ray, best, is);
    sh.wo = -normalize(ray.d);
    sh.ng = face_forward(normalize(cross(is.gu, is.gv)), sh.wo);
    sh.ns = is.has_n ? normalize(is.n) : normalize(cross(is.su, is.sv));
    const double err = 2.220446049250313e-16 * 65536.0; // N::epsilon() * 2^16
    V3 p = ray.o + ray.d * is.t;
    V3 p_err = sh.ng * err;
    sh.praw = p;
    sh.p = p + p_err;
    sh.pm = p - p_err;
    sh.ss = normalize(is.su);    // si.surface.dpdu (bsdf.rs:34)
    sh.ts = cross(sh.ns, sh.ss); // bsdf.rs:35
}

// Park / restore the shading frame in HBM across the shadow traversals ([field][lane]: coalesced).
// What is stored are the very f64s shade_frame produced; wo, ts, p +- p_err are re-derived by
// the same expressions, so the restored frame is bit-identical to a recomputed one.
__device__ __forceinline__ void stash_put(const DParams &P, unsigned long long g, const Shade &sh) {
    const unsigned long long n = P.frame_threads;
    V3 p = sh.praw;
    double *s = P.stash + g;
    s[0 * n] = p.x; s[1 * n] = p.y; s[2 * n] = p.z;
    s[3 * n] = sh.ng.x; s[4 * n] = sh.ng.y; s[5 * n] = sh.ng.z;
    s[6 * n] = sh.ns.x; s[7 * n] = sh.ns.y; s[8 * n] = sh.ns.z;
    s[9 * n] = sh.ss.x; s[10 * n] = sh.ss.y; s[11 * n] = sh.ss.z;
    s[12 * n] = (double)sh.mat;
}
__device__ __forceinline__ void stash_get(const DParams &P, unsigned long long g, Shade &sh, const Ray &ray) {
    const unsigned long long n = P.frame_threads;
    const double *s = P.stash + g;
    V3 p{s[0 * n], s[1 * n], s[2 * n]};
    sh.ng = V3{s[3 * n], s[4 * n], s[5 * n]};
    sh.ns = V3{s[6 * n], s[7 * n], s[8 * n]};
    sh.ss = V3{s[9 * n], s[10 * n], s[11 * n]};
    sh.mat = (int32_t)s[12 * n];
    sh.wo = -normalize(ray.d);
    const double err = 2.220446049250313e-16 * 65536.0;
    V3 p_err = sh.ng * err;
    sh.praw = p;
    sh.p = p + p_err;
    sh.pm = p - p_err;
    sh.ts = cross(sh.ns, sh.ss);
}

#define LG_LDSS_BLOCK 1024
// LDSS (reference traversal only): one 1024-lane workgroup per CU with the scene's node / primref / sphere /
// cuboid tables copied into LDS behind the stacks (see load_node); otherwise 256-lane workgroups and L1/L2.
template <bool STATS, bool FAST, bool LDSS, bool PRUNE = false>
__global__ void __launch_bounds__(LDSS ? LG_LDSS_BLOCK : LG_BLOCK, LG_WAVES_PER_SIMD) trace_kernel(const DParams P) {
    static_assert(!(PRUNE && FAST), "the fast mode prunes its own trees by its own rule");
    static_assert(!(FAST && LDSS) && !(STATS && LDSS), "LDS-resident scene: plain reference traversal only");
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const unsigned long long gtid = (unsigned long long)blockIdx.x * blockDim.x + tid;
    uint32_t *stack = lds_stack + tid; // entry i at stack[i * stride]: bank = tid % 32 for every i
    constexpr uint32_t stride = LDSS ? LG_LDSS_BLOCK : LG_BLOCK;
    const uint4 *scn = nullptr;
    if (LDSS) {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_stack + P.stack_depth * stride);
        const uint4 *src = reinterpret_cast<const uint4 *>(P.lds_image);
        for (uint32_t i = tid; i < P.lds_image_n16; i += stride) dst[i] = src[i];
        __syncthreads(); // the only workgroup-wide step; every wave reaches it before pulling tiles
        scn = dst;
    }
    Counters cnt = {0, 0, 0, 0, 0, 0, 0, 0, 0};

    for (;;) {
        // ---- fetch the next 64-pixel tile for this wavefront
        uint32_t tile = 0;
        if (lane == 0) tile = atomicAdd(P.tile_counter, 1u);
        tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
        if (tile >= P.ntiles) break; // every wave reaches this exit

        const Pixel px = pixel_of(P, tile, lane);
        const uint32_t x = px.x, y = px.y;
        const bool active = px.active;
        if (!active) continue; // lanes past the edge idle for this tile

        // ---- Camera::sample (camera.rs:113-146)
        double img_plane_height = P.image_plane_height;
        double img_plane_width = img_plane_height * P.aspect;
        double pixel_size = img_plane_height * P.hinv;
        double sample_separation = P.ss_distance * pixel_size;
        double sox = ((double)x * P.winv - 0.5) * img_plane_width;
        double soy = (0.5 - (double)(y + 1u) * P.hinv) * img_plane_height;
        const uint32_t dim = P.ss_root;
        const uint32_t nsamples = dim * dim;
        const double weight = 1. / (double)nsamples;

        V3 color = vzero(); // integrate.rs:17
        for (uint32_t sidx = 0; sidx < nsamples; ++sidx) {
            Ray pray; // the ray of the li() invocation being evaluated
            {
                V3 cam_o = P.cam_origin + ((soy * P.pixel_separation) * P.cam_up) + ((sox * P.pixel_separation) * P.cam_aux);
                V3 cam_d = P.cam_view + (soy * P.cam_up) + (sox * P.cam_aux);
                V3 updiff = P.cam_up * sample_separation;
                V3 auxdiff = P.cam_aux * sample_separation;
                V3 halfdiff = updiff * 0.5 + auxdiff * 0.5;
                uint32_t si = sidx / dim, sj = sidx % dim;
                V3 dd = cam_d + ((double)sj * updiff) + ((double)si * auxdiff) + halfdiff;
                pray = ray_new(cam_o, dd);
            }
            if (STATS) cnt.primary++;

            // ---- li() (integrate.rs:23-80) as a state machine with ONE traversal call site:
            // job 0 = closest hit along `pray`; job 1 = any-hit shadow ray for light `light`.
            uint32_t depth = 0, light = 0;
            bool shadow_job = false;
            Best pbest;              // closest hit of pray
            V3 output = vzero();     // running sum over lights (integrate.rs:47-66)
            V3 value = vzero();
            Ray tray = pray;         // the ray handed to the traversal
            for (;;) {
                Best b;
                {
                    Counters before = cnt;
                    walk<LDSS, FAST, PRUNE, STATS>(P, tray, shadow_job, stack, stride, b, scn, cnt);
                    if (STATS && P.stats_filter != 0u && (P.stats_filter == 2u) != shadow_job) { // not the kind being counted
                        cnt.nodes = before.nodes; cnt.spheres = before.spheres; cnt.cuboids = before.cuboids;
                        cnt.triangles = before.triangles; cnt.entries = before.entries;
                    }
                }
                bool have_value = false, need_shade = false, visible = false;
                if (!shadow_job) {
                    if (b.ref == NO_HIT) {
                        value = background(P, normalize(pray.d)); // integrate.rs:26-28
                        have_value = true;
                    } else {
                        if (STATS) cnt.hits++;
                        pbest = b;
                        output = vzero();
                        need_shade = true;
                    }
                } else {
                    visible = !(b.t < 1.0); // point.rs:49
                    need_shade = visible || (light + 1 == P.nlights);
                    if (!need_shade) {
                        ++light;
                        const DLight L = P.lights[light];
                        V3 hit_p = tray.o; // interaction.p + p_err: every shadow ray of this hit starts there
                        tray = ray_new(hit_p, V3{L.pos[0], L.pos[1], L.pos[2]} - hit_p);
                        if (STATS) cnt.shadow++;
                        continue;
                    }
                }
                if (need_shade) {
                    Shade sh;
                    if (!shadow_job) {
                        shade_frame(P, pray, pbest, sh);
                        if (P.nlights > 0) stash_put(P, gtid, sh);
                    } else {
                        stash_get(P, gtid, sh, pray);
                    }
                    const DMaterial m = P.materials[sh.mat];
                    V3 n = sh.ns;
                    if (shadow_job && visible) { // integrate.rs:53-65
                        const DLight L = P.lights[light];
                        V3 wi = V3{L.pos[0], L.pos[1], L.pos[2]} - sh.p;
                        double d = magnitude(wi);
                        double f_att = L.falloff[0] + L.falloff[1] * d + L.falloff[2] * d * d;
                        if (f_att != 0.0) {
                            wi = normalize(wi);
                            double wi_dot_n = dot(wi, n);
                            V3 f = bsdf_f(m, sh, sh.wo, wi);
                            V3 li_col{L.intensity[0], L.intensity[1], L.intensity[2]};
                            output = output + (mul_ew(PI * li_col, f) * wi_dot_n / f_att);
                        }
                    }
                    uint32_t next_light = shadow_job ? light + 1 : 0;
                    if (next_light < P.nlights) {
                        light = next_light;
                        shadow_job = true;
                        const DLight L = P.lights[light];
                        tray = ray_new(sh.p, V3{L.pos[0], L.pos[1], L.pos[2]} - sh.p); // point.rs:43-44
                        if (STATS) cnt.shadow++;
                        continue;
                    }
                    // ---- all lights done
                    output = output + mul_ew(P.ambient, bsdf_f(m, sh, sh.wo, n)); // integrate.rs:67
                    // specular children (integrate.rs:69-77,82-132)
                    bool has_r = false, has_t = false;
                    Sample sr, st;
                    if (depth < P.recursion && (m.kind == MAT_GLASS || m.kind == MAT_MIRROR)) {
                        if (sample_specular_transmission(m, sh, st))
                            has_t = !(st.pdf <= 0.0 || veq(st.spectrum, vzero()) || fabs(dot(st.wi, sh.ns)) == 0.0);
                        if (sample_specular_reflection(m, sh, sr))
                            has_r = !(sr.pdf <= 0.0 || veq(sr.spectrum, vzero()) || dot(sr.wi, sh.ns) <= 0.0);
                    }
                    if (!has_r && !has_t) {
                        value = output + vzero() + vzero(); // integrate.rs:79
                        have_value = true;
                    } else {
                        // push a frame; the reflected child is evaluated first because the sum is
                        // (output + reflected) + refracted
                        if (has_t) {
                            frame_put3(P, depth, FR_TO, gtid, sh.pm);
                            frame_put3(P, depth, FR_TD, gtid, st.wi);
                            frame_put3(P, depth, FR_SPEC_T, gtid, st.spectrum);
                            frame_at(P, depth, FR_A, gtid) = fabs(dot(st.wi, sh.ns));
                            frame_at(P, depth, FR_PDF, gtid) = st.pdf;
                        }
                        if (has_r) {
                            frame_put3(P, depth, FR_ACC, gtid, output);
                            frame_put3(P, depth, FR_SPEC_R, gtid, sr.spectrum);
                            frame_at(P, depth, FR_STATE, gtid) = has_t ? 1.0 : 2.0;
                            V3 wr = -1.0 * sh.wo + 2.0 * dot(sh.wo, sh.ns) * sh.ns; // bxdf::util::reflect (integrate.rs:100)
                            pray = ray_new(sh.p, wr);
                        } else {
                            frame_put3(P, depth, FR_ACC, gtid, output + vzero());
                            frame_at(P, depth, FR_STATE, gtid) = 3.0;
                            pray = ray_new(sh.pm, st.wi);
                        }
                        if (STATS) cnt.secondary++;
                        depth += 1;
                        shadow_job = false;
                        tray = pray;
                    }
                }
                // ---- return `value` up the frame stack
                bool finished = false;
                while (have_value) {
                    if (depth == 0) { finished = true; break; }
                    uint32_t fd = depth - 1;
                    double state = frame_at(P, fd, FR_STATE, gtid);
                    if (state == 3.0) {
                        V3 acc = frame_get3(P, fd, FR_ACC, gtid);
                        V3 spec = frame_get3(P, fd, FR_SPEC_T, gtid);
                        double a = frame_at(P, fd, FR_A, gtid), pdf = frame_at(P, fd, FR_PDF, gtid);
                        V3 refracted = mul_ew(spec, value) * a / pdf; // integrate.rs:129
                        value = acc + refracted;
                        depth = fd;
                    } else {
                        V3 acc = frame_get3(P, fd, FR_ACC, gtid);
                        V3 spec = frame_get3(P, fd, FR_SPEC_R, gtid);
                        V3 reflected = mul_ew(spec, value); // integrate.rs:103
                        acc = acc + reflected;
                        if (state == 1.0) {
                            frame_put3(P, fd, FR_ACC, gtid, acc);
                            frame_at(P, fd, FR_STATE, gtid) = 3.0;
                            pray = ray_new(frame_get3(P, fd, FR_TO, gtid), frame_get3(P, fd, FR_TD, gtid));
                            if (STATS) cnt.secondary++;
                            shadow_job = false;
                            tray = pray;
                            have_value = false; // trace the transmitted child at depth fd + 1
                        } else {
                            value = acc + vzero();
                            depth = fd;
                        }
                    }
                }
                if (finished) break;
            }
            color = color + value;
        }
        color = color * weight; // integrate.rs:19

        // ---- Img::set (img.rs:46-67)
        const unsigned long long pix = px.pix;
        if (P.out_rgba) {
            uint32_t rgba = to_byte(color.x) | (to_byte(color.y) << 8) | (to_byte(color.z) << 16) | (255u << 24);
            reinterpret_cast<uint32_t *>(P.out_rgba)[pix] = rgba;
        }
        if (P.out_radiance) {
            P.out_radiance[3 * pix] = color.x; P.out_radiance[3 * pix + 1] = color.y; P.out_radiance[3 * pix + 2] = color.z;
        }
    }

    if (STATS) {
        atomicAdd(&P.stats->primary_rays, (unsigned long long)cnt.primary);
        atomicAdd(&P.stats->shadow_rays, (unsigned long long)cnt.shadow);
        atomicAdd(&P.stats->secondary_rays, (unsigned long long)cnt.secondary);
        atomicAdd(&P.stats->nodes_tested, (unsigned long long)cnt.nodes);
        atomicAdd(&P.stats->spheres_tested, (unsigned long long)cnt.spheres);
        atomicAdd(&P.stats->cuboids_tested, (unsigned long long)cnt.cuboids);
        atomicAdd(&P.stats->triangles_tested, (unsigned long long)cnt.triangles);
        atomicAdd(&P.stats->accel_entries, (unsigned long long)cnt.entries);
        atomicAdd(&P.stats->hits, (unsigned long long)cnt.hits);
    }
}

// ------------------------------------------------------------------------------------------
// Streaming pipeline: the same li() for scenes WITHOUT glass / mirror (no recursion), cut into
// three kernels so that traversal (wants occupancy, 128 VGPRs) and shading (wants registers: trig,
// microfacet, Fresnel) each get their own register allocation.  Per work item (pixel) the state
// between kernels lives in HBM, SoA, indexed by widx = tile * 64 + lane:
//   K1 primary   camera ray -> closest hit -> shade_frame           -> hit_ref, frame[13][n]
//   K2 shadow    one any-hit traversal per light from frame.p       -> vis bits
//   K3 shade     lights in order, ambient, sample sum, Img::set     -> film
// (the shading frame used to be a kernel of its own between K1 and K2; see park_frame)
// Every f64 is produced by the same expressions as in the megakernel; only their placement in
// kernels differs.  Up to 32 lights; scenes with more use the megakernel.
// ------------------------------------------------------------------------------------------
// Camera::sample for sample `sidx` of pixel (x, y) (camera.rs:113-146)
__device__ __forceinline__ Ray camera_ray(const DParams &P, uint32_t x, uint32_t y, uint32_t sidx) {
    double img_plane_height = P.image_plane_height;
    double img_plane_width = img_plane_height * P.aspect;
    double pixel_size = img_plane_height * P.hinv;
    double sample_separation = P.ss_distance * pixel_size;
    double sox = ((double)x * P.winv - 0.5) * img_plane_width;
    double soy = (0.5 - (double)(y + 1u) * P.hinv) * img_plane_height;
    V3 cam_o = P.cam_origin + ((soy * P.pixel_separation) * P.cam_up) + ((sox * P.pixel_separation) * P.cam_aux);
    V3 cam_d = P.cam_view + (soy * P.cam_up) + (sox * P.cam_aux);
    V3 updiff = P.cam_up * sample_separation;
    V3 auxdiff = P.cam_aux * sample_separation;
    V3 halfdiff = updiff * 0.5 + auxdiff * 0.5;
    const uint32_t dim = P.ss_root;
    uint32_t si = sidx / dim, sj = sidx % dim;
    V3 dd = cam_d + ((double)sj * updiff) + ((double)si * auxdiff) + halfdiff;
    return ray_new(cam_o, dd);
}

// The shading frame of a primary hit, parked for the shadow and shade passes.  It is computed at the end of
// the primary traversal pass (after the walk, so its registers are not live during it) rather than in a pass
// of its own: one launch and one round trip of the hit record less (measured -2.7 % / -5 % per frame).
__device__ __forceinline__ void park_frame(const DParams &P, unsigned long long widx, const Ray &ray, const Best &b) {
    if (b.ref == NO_HIT) return;
    Shade sh;
    shade_frame(P, ray, b, sh);
    const unsigned long long n = P.n_items;
    double *f = P.frame + widx;
    f[0 * n] = sh.praw.x; f[1 * n] = sh.praw.y; f[2 * n] = sh.praw.z;
    f[3 * n] = sh.ng.x; f[4 * n] = sh.ng.y; f[5 * n] = sh.ng.z;
    f[6 * n] = sh.ns.x; f[7 * n] = sh.ns.y; f[8 * n] = sh.ns.z;
    f[9 * n] = sh.ss.x; f[10 * n] = sh.ss.y; f[11 * n] = sh.ss.z;
    f[12 * n] = (double)sh.mat;
}

// Fix-up pass of the packet organisation (persistent, tile counter, per-lane LDS stack): only the tiles listed in
// P.tie_tiles, and in them only the lanes (and lights) flagged in P.tie_flag, are re-traced with the private reference walk.
// (Round 1's three-kernel pipeline ran its two traversal passes through this kernel's plain forms; the wavefront pipeline
// replaced it in round 2 and those forms were retired in round 3.)
template <bool SHADOW>
__global__ void __launch_bounds__(LG_BLOCK, LG_TRAV_WAVES_PER_SIMD) stream_fixup_kernel(const DParams P) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint32_t *stack = lds_stack + tid;
    constexpr uint32_t stride = LG_BLOCK;
    Counters cnt = {0, 0, 0, 0, 0, 0, 0, 0, 0}; (void)cnt;
    for (;;) {
        uint32_t tile = 0;
        if (lane == 0) tile = atomicAdd(P.tile_counter + 2, 1u);
        tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
        if (tile >= P.tile_counter[1]) break; // number of listed tiles (written by the packet pass); every wave reaches this exit
        tile = P.tie_tiles[tile];
        Pixel px = pixel_of(P, tile, lane);
        if (!px.active) continue;
        const unsigned long long widx = (unsigned long long)tile * 64ull + lane;
        const uint32_t redo = P.tie_flag[widx]; // bit l = light l (shadow) / bit 0 (primary) must be re-traced
        if (redo == 0u) continue;
        if (!SHADOW) {
            Ray ray = camera_ray(P, px.x, px.y, P.sample_index);
            Best b;
            walk<false, false>(P, ray, false, stack, stride, b, nullptr, cnt);
            P.hit_ref[widx] = b.ref; // (t and the accel instance are consumed by park_frame right here)
            park_frame(P, widx, ray, b);
        } else {
            if (P.hit_ref[widx] == NO_HIT) continue;
            const unsigned long long n = P.n_items;
            // interaction.p + p_err, recomputed from the parked frame exactly as stash_get does
            V3 praw{P.frame[0 * n + widx], P.frame[1 * n + widx], P.frame[2 * n + widx]};
            V3 ng{P.frame[3 * n + widx], P.frame[4 * n + widx], P.frame[5 * n + widx]};
            const double err = 2.220446049250313e-16 * 65536.0;
            V3 hit_p = praw + ng * err;
            uint32_t vis = P.vis[widx];
            for (uint32_t l = 0; l < P.nlights; ++l) {
                if (!((redo >> l) & 1u)) continue;
                const DLight L = P.lights[l];
                Ray sray = ray_new(hit_p, V3{L.pos[0], L.pos[1], L.pos[2]} - hit_p); // point.rs:43-44
                Best b;
                walk<false, false>(P, sray, true, stack, stride, b, nullptr, cnt);
                if (!(b.t < 1.0)) vis |= 1u << l; // point.rs:49
                else vis &= ~(1u << l);
            }
            P.vis[widx] = vis;
        }
    }
}

// K1' / K2': the packet organisation of the same two traversal passes -- one tree walk per wavefront
// (traverse_packet).  Lanes whose walk met an exact tie (or a NaN t) get their bit set in P.tie_flag and
// their tile appended to P.tie_tiles; stream_fixup_kernel re-traces just those.
// LDS: [per-wave stacks][scene image (LDSS)].
template <bool SHADOW, bool LDSS>
__global__ void __launch_bounds__(LDSS ? LG_LDSS_BLOCK : LG_BLOCK, LG_TRAV_WAVES_PER_SIMD) stream_packet_kernel(const DParams P) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    constexpr uint32_t block = LDSS ? LG_LDSS_BLOCK : LG_BLOCK;
    uint32_t *ws = lds_stack + (tid >> 6) * P.stack_depth * PKT_ENTRY; // this wave's stack
    const uint4 *scn = nullptr;
    if (LDSS) {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_stack + (block / 64u) * P.stack_depth * PKT_ENTRY);
        const uint4 *src = reinterpret_cast<const uint4 *>(P.lds_image);
        for (uint32_t i = tid; i < P.lds_image_n16; i += block) dst[i] = src[i];
        __syncthreads();
        scn = dst;
    }
    for (;;) {
        uint32_t tile = 0;
        if (lane == 0) tile = atomicAdd(P.tile_counter, 1u);
        tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
        if (tile >= P.ntiles) break; // every wave reaches this exit
        // all 64 lanes stay together; lanes without a pixel (or without a hit) are dead from the start
        const Pixel px = pixel_of(P, tile, lane);
        const unsigned long long widx = (unsigned long long)tile * 64ull + lane;
        uint32_t ties = 0u;
        if (!SHADOW) {
            const Ray ray = camera_ray(P, px.x, px.y, P.sample_index);
            Best b;
            bool tie = false;
            traverse_packet<LDSS>(P, scn, ray, px.active, false, ws, lane, b, tie);
            if (px.active) P.hit_ref[widx] = b.ref;
            ties = px.active && tie ? 1u : 0u;
            if (px.active && !tie) park_frame(P, widx, ray, b); // a tie lane's frame comes from the fix-up pass
        } else {
            const bool has = px.active && P.hit_ref[widx] != NO_HIT;
            const unsigned long long n = P.n_items;
            V3 hit_p = vzero();
            if (has) {
                V3 praw{P.frame[0 * n + widx], P.frame[1 * n + widx], P.frame[2 * n + widx]};
                V3 ng{P.frame[3 * n + widx], P.frame[4 * n + widx], P.frame[5 * n + widx]};
                const double err = 2.220446049250313e-16 * 65536.0;
                hit_p = praw + ng * err;
            }
            uint32_t vis = 0;
            for (uint32_t l = 0; l < P.nlights; ++l) {
                const DLight L = P.lights[l];
                const Ray sray = ray_new(hit_p, V3{L.pos[0], L.pos[1], L.pos[2]} - hit_p); // point.rs:43-44
                Best b;
                bool tie = false;
                traverse_packet<LDSS>(P, scn, sray, has, true, ws, lane, b, tie);
                if (!(b.t < 1.0)) vis |= 1u << l; // point.rs:49
                if (has && tie && !(b.t < 1.0)) ties |= 1u << l; // an occluded ray's answer is order-independent
            }
            if (has) P.vis[widx] = vis;
        }
        if (px.active) P.tie_flag[widx] = ties;
        if (__ballot(ties != 0u) != 0ull && lane == 0u) P.tie_tiles[atomicAdd(P.tile_counter + 1, 1u)] = tile;
    }
}

// K3: li() of a non-specular hit from the parked frame and the visibility bits, then the per-pixel
// sample sum and Img::set (integrate.rs:16-80, img.rs:46-67).  3 waves per SIMD: measured best (0.97 -> 0.87 ms).
__global__ void __launch_bounds__(LG_BLOCK, 3) stream_shade_kernel(const DParams P) {
    const unsigned long long widx = (unsigned long long)blockIdx.x * LG_BLOCK + threadIdx.x;
    if (widx >= P.n_items) return;
    Pixel px = pixel_of(P, (uint32_t)(widx >> 6), (uint32_t)(widx & 63u));
    if (!px.active) return;
    Ray ray = camera_ray(P, px.x, px.y, P.sample_index);
    V3 value;
    if (P.hit_ref[widx] == NO_HIT) {
        value = background(P, normalize(ray.d)); // integrate.rs:26-28
    } else {
        const unsigned long long n = P.n_items;
        const double *f = P.frame + widx;
        Shade sh;
        V3 p{f[0 * n], f[1 * n], f[2 * n]};
        sh.ng = V3{f[3 * n], f[4 * n], f[5 * n]};
        sh.ns = V3{f[6 * n], f[7 * n], f[8 * n]};
        sh.ss = V3{f[9 * n], f[10 * n], f[11 * n]};
        sh.mat = (int32_t)f[12 * n];
        sh.wo = -normalize(ray.d);
        const double err = 2.220446049250313e-16 * 65536.0;
        V3 p_err = sh.ng * err;
        sh.praw = p; sh.p = p + p_err; sh.pm = p - p_err;
        sh.ts = cross(sh.ns, sh.ss);
        const DMaterial m = P.materials[sh.mat];
        const uint32_t vis = P.vis[widx];
        V3 nrm = sh.ns;
        V3 output = vzero();
        for (uint32_t l = 0; l < P.nlights; ++l) { // integrate.rs:47-66
            if (!((vis >> l) & 1u)) continue;
            const DLight L = P.lights[l];
            V3 wi = V3{L.pos[0], L.pos[1], L.pos[2]} - sh.p;
            double d = magnitude(wi);
            double f_att = L.falloff[0] + L.falloff[1] * d + L.falloff[2] * d * d;
            if (f_att == 0.0) continue;
            wi = normalize(wi);
            double wi_dot_n = dot(wi, nrm);
            V3 fr = bsdf_f(m, sh, sh.wo, wi);
            V3 li_col{L.intensity[0], L.intensity[1], L.intensity[2]};
            output = output + (mul_ew(PI * li_col, fr) * wi_dot_n / f_att);
        }
        output = output + mul_ew(P.ambient, bsdf_f(m, sh, sh.wo, nrm)); // integrate.rs:67
        value = output + vzero() + vzero();                              // integrate.rs:79 (no specular children)
    }
    // integrate(): color = sum over samples, then * weight
    const uint32_t nsamples = P.ss_root * P.ss_root;
    V3 color = vzero();
    if (P.sample_index > 0) color = V3{P.accum[widx], P.accum[P.n_items + widx], P.accum[2 * P.n_items + widx]};
    color = color + value;
    if (P.sample_index + 1 < nsamples) {
        P.accum[widx] = color.x; P.accum[P.n_items + widx] = color.y; P.accum[2 * P.n_items + widx] = color.z;
        return;
    }
    const double weight = 1. / (double)nsamples;
    color = color * weight;
    const unsigned long long pix = px.pix;
    if (P.out_rgba) {
        uint32_t rgba = to_byte(color.x) | (to_byte(color.y) << 8) | (to_byte(color.z) << 16) | (255u << 24);
        reinterpret_cast<uint32_t *>(P.out_rgba)[pix] = rgba;
    }
    if (P.out_radiance) {
        P.out_radiance[3 * pix] = color.x; P.out_radiance[3 * pix + 1] = color.y; P.out_radiance[3 * pix + 2] = color.z;
    }
}

// ------------------------------------------------------------------------------------------
// Wavefront pipeline: li() (integrate.rs:23-132) level by level, for every scene -- with or without glass / mirror.
//
// The streaming pipeline above covers scenes without specular materials with three kernels over the pixels.  This
// is its generalisation: level d holds the rays of recursion depth d (level 0 = the pixels of a chunk of the film,
// in their dense 8x8-tile order; deeper levels = queues of secondary rays), and per level
//   W1 closest  one closest-hit walk per ray.  A miss is finished on the spot (background).  Hits are COMPACTED: the
//               lanes that hit take consecutive slots of the level's hit queue (ballot + popcount prefix, one atomic
//               per wavefront) and park their shading frame there, so that W2 / W3 run full waves however sparse
//               the hits are (README sphere: 4 % of the pixels; secondary rays of a glass object: a few %).
//   W2 shadow   one any-hit walk per hit and light -> visibility bits.
//   W3 shade    radiance of the hit (lights in order, ambient); with a level below: the specular children
//               (BSDF::sample_f, integrate.rs:82-132) are appended to the next level's ray queue -- again one
//               atomic per wavefront and kind -- and their weights are left with the parent.
// then bottom-up, W4 combine: li = (output + spec_r * li[child_r]) + spec_t * li[child_t] * |wi.n| / pdf, the order of
// integrate.rs:79 / 103 / 129; the level-0 pass quantises (Img::set).  Every f64 is produced by the same expression
// as in the megakernel; what differs is where the intermediate values wait (HBM, SoA by ray index of the level).
// Queue capacities are worst case (level d: 2^d rays per pixel of the chunk), so nothing can overflow; the host
// sizes the chunk of the film to its memory budget (capi.cpp).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) { // number of set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// one slot per lane of `mask` in a queue whose fill count is *counter: consecutive slots, one atomic per wavefront
__device__ __forceinline__ uint32_t wave_append(uint32_t *counter, bool want) {
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(want);
    if (mask == 0ull) return 0u;
    uint32_t base = 0u;
    const uint32_t leader = (uint32_t)__builtin_ctzll(mask);
    if ((threadIdx.x & 63u) == leader) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(mask));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
    return base + lanes_below(mask);
}
// integrate(): sum over the pixel's samples, then * weight; Img::set (integrate.rs:16-20, img.rs:46-67)
__device__ __forceinline__ void finish_pixel(const DParams &P, const Pixel &px, unsigned long long widx, V3 value) {
    const uint32_t nsamples = P.ss_root * P.ss_root;
    V3 color = vzero();
    if (P.sample_index > 0) color = V3{P.accum[widx], P.accum[P.n_items + widx], P.accum[2 * P.n_items + widx]};
    color = color + value;
    if (P.sample_index + 1 < nsamples) {
        P.accum[widx] = color.x; P.accum[P.n_items + widx] = color.y; P.accum[2 * P.n_items + widx] = color.z;
        return;
    }
    const double weight = 1. / (double)nsamples;
    color = color * weight;
    const unsigned long long pix = px.pix;
    if (P.out_rgba) {
        uint32_t rgba = to_byte(color.x) | (to_byte(color.y) << 8) | (to_byte(color.z) << 16) | (255u << 24);
        reinterpret_cast<uint32_t *>(P.out_rgba)[pix] = rgba;
    }
    if (P.out_radiance) {
        P.out_radiance[3 * pix] = color.x; P.out_radiance[3 * pix + 1] = color.y; P.out_radiance[3 * pix + 2] = color.z;
    }
}
__device__ __forceinline__ Ray wf_load_ray(const DParams &P, unsigned long long j) {
    const unsigned long long n = P.wf_cap;
    const double *q = P.wf_q + j;
    return ray_new(V3{q[0 * n], q[1 * n], q[2 * n]}, V3{q[3 * n], q[4 * n], q[5 * n]}); // Ray3::new (ray.rs:28-33)
}
// The hit queue of a level has two parts.  A wavefront most of whose lanes hit (>= WF_FULL_MIN) keeps its hits where its
// rays are: slot = ray index, no atomic, holes marked WF_NONE -- the block IS the 8x8 tile (or the 64 consecutive queue rays),
// so the shadow pass walks the same coherent rays, and packing 61 + 3 lanes of two tiles into one wave would cost more than
// three idle lanes.  Every other wavefront marks its block empty and appends just its hits, compacted (ballot + popcount
// prefix, one atomic per wavefront), behind the dense part: sparse hits -- a small object in front of the background, the
// secondary rays of a glass object -- become full waves for the shadow and shade passes.
constexpr uint32_t WF_FULL_MIN = 48u;
// Rays per work tile of a level (a hook: a level with few, incoherent rays could be cut into tiles of fewer rays).
__device__ __forceinline__ uint32_t wf_lanes_per_tile(unsigned long long rays) {
    (void)rays;
    return 64u; // measured: narrower tiles (8 .. 32 rays per wave for levels of < 2^19 rays) do not help the mesh configs and cost the small scenes 30-80 %
}
struct HitSlots { // work tile t of a pass over the hit queue -> hit index of this lane (valid or not)
    unsigned long long n_rays, n_part;
    uint32_t tiles_dense, tiles, lpt;
};
__device__ __forceinline__ unsigned long long wf_level_rays(const DParams &P, uint32_t level) {
    return level == 0u ? (unsigned long long)P.ntiles * 64ull : P.wf_counts[level];
}
__device__ __forceinline__ HitSlots hit_slots(const DParams &P, uint32_t level, uint32_t lpt) {
    HitSlots s;
    s.lpt = lpt;
    s.n_part = P.wf_counts[P.wf_levels + level];
    s.n_rays = wf_level_rays(P, level);
    s.tiles_dense = (uint32_t)((s.n_rays + lpt - 1u) / lpt);
    s.tiles = s.tiles_dense + (uint32_t)((s.n_part + lpt - 1u) / lpt);
    return s;
}
__device__ __forceinline__ bool hit_of(const DParams &P, const HitSlots &s, uint32_t tile, uint32_t lane, unsigned long long &h) {
    if (lane >= s.lpt) return false;
    if (tile < s.tiles_dense) {
        h = (unsigned long long)tile * s.lpt + lane;
        return h < s.n_rays && P.wf_hq[h] != WF_NONE;
    }
    const unsigned long long k = (unsigned long long)(tile - s.tiles_dense) * s.lpt + lane;
    h = P.wf_hit_cap + k;
    return k < s.n_part;
}

// W1 / W2: persistent traversal kernels of the wavefront pipeline (tile counter, per-lane LDS stack; LDSS as above).
// L0: the launch is level 0's (rays from the camera, work items = the chunk's pixels in 8x8 tiles).
template <bool FAST, bool SHADOW, bool LDSS, bool L0, bool PRUNE = false>
__global__ void __launch_bounds__(LDSS ? LG_LDSS_BLOCK : LG_BLOCK, LG_TRAV_WAVES_PER_SIMD) wf_trace_kernel(const DParams P) {
    static_assert(!(FAST && LDSS), "the LDS-resident scene holds the reference tree only");
    static_assert(!(FAST && PRUNE), "the fast mode prunes its own trees by its own rule");
    static_assert(!(SHADOW && L0), "the shadow pass has one form for every level");
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t level = L0 ? 0u : P.wf_level;
    // work items of this launch: the chunk's pixels (level 0, closest), this level's rays, or this level's hits
    unsigned long long n_work = 0;
    HitSlots hs;
    uint32_t ntiles;
    const uint32_t lpt = L0 ? 64u : wf_lanes_per_tile(wf_level_rays(P, level)); // rays per wave (level 0's closest pass: 8x8 pixel tiles)
    if (SHADOW) { hs = hit_slots(P, level, lpt); ntiles = hs.tiles; }
    else {
        n_work = L0 ? (unsigned long long)P.ntiles * 64ull : P.wf_counts[level];
        ntiles = L0 ? P.ntiles : (uint32_t)((n_work + lpt - 1u) / lpt);
    }
    if (ntiles == 0u) return; // (uniform: before the LDS copy and its barrier)
    uint32_t *stack = lds_stack + tid;
    constexpr uint32_t stride = LDSS ? LG_LDSS_BLOCK : LG_BLOCK;
    const uint4 *scn = nullptr;
    if (LDSS) {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_stack + P.stack_depth * stride);
        const uint4 *src = reinterpret_cast<const uint4 *>(P.lds_image);
        for (uint32_t i = tid; i < P.lds_image_n16; i += stride) dst[i] = src[i];
        __syncthreads(); // the only workgroup-wide step; every wave reaches it before pulling tiles
        scn = dst;
    }
    Counters cnt = {0, 0, 0, 0, 0, 0, 0, 0, 0}; (void)cnt;
    for (;;) {
        uint32_t tile = 0;
        if (lane == 0) tile = atomicAdd(P.tile_counter, 1u);
        tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
        if (tile >= ntiles) break; // every wave reaches this exit
        if (!SHADOW) {
            const unsigned long long i = (unsigned long long)tile * lpt + lane;
            Pixel px;
            px.active = false;
            Ray ray = ray_new(V3{0.0, 0.0, 0.0}, V3{0.0, 0.0, 1.0});
            if (L0) {
                px = pixel_of(P, P.tile0 + tile, lane);
                if (px.active) ray = camera_ray(P, px.x, px.y, P.sample_index);
            } else if (lane < lpt && i < n_work) {
                px.active = true;
                ray = wf_load_ray(P, i);
            }
            const bool active = px.active;
            Best b;
            b.ref = NO_HIT; b.t = INFINITY; b.accel = 0u;
            if (active) {
                bool tie = false;
                walk<LDSS, FAST, PRUNE>(P, ray, false, stack, stride, b, scn, cnt);
                (void)tie;
            }
            const bool hit = active && b.ref != NO_HIT;
            // ---- this wave's slots in the level's hit queue
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
            const uint32_t nhit = (uint32_t)__builtin_popcountll(mask);
            unsigned long long h = i; // most lanes hit: the hits stay where the rays are
            const bool mine = L0 || lane < lpt; // this lane stands for slot i of the dense part (a ray of the level or a slot past its last ray)
            if (nhit * 64u < WF_FULL_MIN * lpt) { // few hits (or none): the block is marked empty, the hits are appended behind the dense part
                if (mine) P.wf_hq[i] = WF_NONE;
                if (nhit != 0u) {
                    uint32_t base_v = 0u;
                    if (lane == 0u) base_v = atomicAdd(P.wf_counts + P.wf_levels + level, nhit);
                    h = P.wf_hit_cap + (uint32_t)__builtin_amdgcn_readfirstlane((int)base_v) + lanes_below(mask);
                }
            } else if (!hit && mine) P.wf_hq[i] = WF_NONE; // a hole of a dense block
            if (hit) {
                P.wf_hq[h] = (uint32_t)i;
                Shade sh;
                shade_frame(P, ray, b, sh);
                const unsigned long long n = P.wf_hit_stride;
                double *f = P.frame + h;
                f[0 * n] = sh.praw.x; f[1 * n] = sh.praw.y; f[2 * n] = sh.praw.z;
                f[3 * n] = sh.ng.x; f[4 * n] = sh.ng.y; f[5 * n] = sh.ng.z;
                f[6 * n] = sh.ns.x; f[7 * n] = sh.ns.y; f[8 * n] = sh.ns.z;
                f[9 * n] = sh.ss.x; f[10 * n] = sh.ss.y; f[11 * n] = sh.ss.z;
                f[12 * n] = (double)sh.mat;
            } else if (active) { // integrate.rs:26-28
                const V3 value = background(P, normalize(ray.d));
                if (P.wf_levels == 1u) finish_pixel(P, px, i, value);
                else {
                    const unsigned long long n = P.wf_cap;
                    P.wf_out[i] = value.x; P.wf_out[n + i] = value.y; P.wf_out[2 * n + i] = value.z;
                    if (level + 1u < P.wf_levels) P.wf_child[i] = WF_MISS;
                }
            }
        } else {
            unsigned long long h;
            if (!hit_of(P, hs, tile, lane, h)) continue;
            const unsigned long long n = P.wf_hit_stride;
            // interaction.p + p_err, recomputed from the parked frame exactly as stash_get does
            V3 praw{P.frame[0 * n + h], P.frame[1 * n + h], P.frame[2 * n + h]};
            V3 ng{P.frame[3 * n + h], P.frame[4 * n + h], P.frame[5 * n + h]};
            const double err = 2.220446049250313e-16 * 65536.0;
            V3 hit_p = praw + ng * err;
            uint32_t vis = 0u;
            for (uint32_t l = 0; l < P.nlights; ++l) {
                const DLight L = P.lights[l];
                Ray sray = ray_new(hit_p, V3{L.pos[0], L.pos[1], L.pos[2]} - hit_p); // point.rs:43-44
                Best b;
                bool tie = false;
                walk<LDSS, FAST, PRUNE>(P, sray, true, stack, stride, b, scn, cnt);
                (void)tie;
                if (!(b.t < 1.0)) vis |= 1u << l; // point.rs:49
            }
            P.vis[h] = vis;
        }
    }
}

// W3: li() of a hit up to its specular children (integrate.rs:30-77).
// KIND 0: the scene has no recursion at all (one level): li = output + 0 + 0 goes straight to the film (L0 by construction);
// KIND 1: a level with a level below: the specular children are appended to the next level's ray queue;
// KIND 2: the deepest level of a recursive scene: li = output + 0 + 0 is stored for the level above.
template <int KIND, bool L0>
__global__ void __launch_bounds__(LG_BLOCK, 3) wf_shade_kernel(const DParams P) {
    const uint32_t level = L0 ? 0u : P.wf_level;
    const HitSlots hs = hit_slots(P, level, 64u);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // (whole waves stay together: the appends of KIND 1 are wave-wide)
    // (level 0: the grid covers the chunk's dense tiles AND the most hits that can be appended behind them, one wave per
    // tile, no loop -- the loop costs this kernel 30 spilled registers; deeper levels stride)
    const unsigned long long t_step = L0 ? ~0ull >> 1 : (unsigned long long)gridDim.x * (LG_BLOCK / 64u);
    for (unsigned long long t = (unsigned long long)blockIdx.x * (LG_BLOCK / 64u) + wave; t < hs.tiles; t += t_step) {
        unsigned long long h = 0;
        const bool valid = hit_of(P, hs, (uint32_t)t, lane, h);
        unsigned long long j = 0;
        Pixel px;
        px.active = false;
        bool has_r = false, has_t = false;
        Sample sr, st;
        Shade sh;
        V3 output = vzero();
        if (valid) {
            j = P.wf_hq[h];
            Ray ray;
            if (L0) {
                px = pixel_of(P, P.tile0 + (uint32_t)(j >> 6), (uint32_t)(j & 63u));
                ray = camera_ray(P, px.x, px.y, P.sample_index);
            } else ray = wf_load_ray(P, j);
            const unsigned long long n = P.wf_hit_stride;
            const double *f = P.frame + h;
            V3 p{f[0 * n], f[1 * n], f[2 * n]};
            sh.ng = V3{f[3 * n], f[4 * n], f[5 * n]};
            sh.ns = V3{f[6 * n], f[7 * n], f[8 * n]};
            sh.ss = V3{f[9 * n], f[10 * n], f[11 * n]};
            sh.mat = (int32_t)f[12 * n];
            sh.wo = -normalize(ray.d);
            const double err = 2.220446049250313e-16 * 65536.0;
            V3 p_err = sh.ng * err;
            sh.praw = p; sh.p = p + p_err; sh.pm = p - p_err;
            sh.ts = cross(sh.ns, sh.ss);
            const DMaterial m = P.materials[sh.mat];
            const uint32_t vis = P.nlights ? P.vis[h] : 0u;
            V3 nrm = sh.ns;
            for (uint32_t l = 0; l < P.nlights; ++l) { // integrate.rs:47-66
                if (!((vis >> l) & 1u)) continue;
                const DLight L = P.lights[l];
                V3 wi = V3{L.pos[0], L.pos[1], L.pos[2]} - sh.p;
                double d = magnitude(wi);
                double f_att = L.falloff[0] + L.falloff[1] * d + L.falloff[2] * d * d;
                if (f_att == 0.0) continue;
                wi = normalize(wi);
                double wi_dot_n = dot(wi, nrm);
                V3 fr = bsdf_f(m, sh, sh.wo, wi);
                V3 li_col{L.intensity[0], L.intensity[1], L.intensity[2]};
                output = output + (mul_ew(PI * li_col, fr) * wi_dot_n / f_att);
            }
            output = output + mul_ew(P.ambient, bsdf_f(m, sh, sh.wo, nrm)); // integrate.rs:67
            if (KIND == 1 && (m.kind == MAT_GLASS || m.kind == MAT_MIRROR)) { // depth < max recursion (integrate.rs:69-77)
                if (sample_specular_transmission(m, sh, st))
                    has_t = !(st.pdf <= 0.0 || veq(st.spectrum, vzero()) || fabs(dot(st.wi, sh.ns)) == 0.0);
                if (sample_specular_reflection(m, sh, sr))
                    has_r = !(sr.pdf <= 0.0 || veq(sr.spectrum, vzero()) || dot(sr.wi, sh.ns) <= 0.0);
            }
        }
        if (KIND == 0) { // integrate.rs:79 with no children, straight to the film
            if (valid) finish_pixel(P, px, j, output + vzero() + vzero());
            continue;
        }
        if (KIND == 2) {
            if (valid) {
                const V3 value = output + vzero() + vzero();
                const unsigned long long n = P.wf_cap;
                P.wf_out[j] = value.x; P.wf_out[n + j] = value.y; P.wf_out[2 * n + j] = value.z;
            }
            continue;
        }
        // children: consecutive slots of the next level's ray queue per wavefront and kind
        const uint32_t cr = wave_append(P.wf_counts + level + 1u, has_r);
        const uint32_t ct = wave_append(P.wf_counts + level + 1u, has_t);
        if (!valid) continue;
        const unsigned long long n = P.wf_cap, nn = P.wf_cap_next;
        P.wf_out[j] = output.x; P.wf_out[n + j] = output.y; P.wf_out[2 * n + j] = output.z;
        P.wf_child[j] = has_r ? cr : WF_NONE;
        P.wf_child[n + j] = has_t ? ct : WF_NONE;
        double *sp = P.wf_spec + j;
        if (has_r) {
            sp[0 * n] = sr.spectrum.x; sp[1 * n] = sr.spectrum.y; sp[2 * n] = sr.spectrum.z;
            const V3 wr = -1.0 * sh.wo + 2.0 * dot(sh.wo, sh.ns) * sh.ns; // bxdf::util::reflect (integrate.rs:100)
            double *q = P.wf_q_next + cr;
            q[0 * nn] = sh.p.x; q[1 * nn] = sh.p.y; q[2 * nn] = sh.p.z; q[3 * nn] = wr.x; q[4 * nn] = wr.y; q[5 * nn] = wr.z;
        }
        if (has_t) {
            sp[3 * n] = st.spectrum.x; sp[4 * n] = st.spectrum.y; sp[5 * n] = st.spectrum.z;
            sp[6 * n] = fabs(dot(st.wi, sh.ns)); sp[7 * n] = st.pdf;
            double *q = P.wf_q_next + ct;
            q[0 * nn] = sh.pm.x; q[1 * nn] = sh.pm.y; q[2 * nn] = sh.pm.z; q[3 * nn] = st.wi.x; q[4 * nn] = st.wi.y; q[5 * nn] = st.wi.z;
        }
    }
}

// W4: li of this level's rays from their children's (integrate.rs:79, 103, 129); level 0 also quantises
__global__ void __launch_bounds__(LG_BLOCK) wf_combine_kernel(const DParams P) {
    const uint32_t level = P.wf_level;
    const unsigned long long n_work = level == 0u ? (unsigned long long)P.ntiles * 64ull : P.wf_counts[level];
    const unsigned long long n = P.wf_cap, nn = P.wf_cap_next;
    for (unsigned long long j = (unsigned long long)blockIdx.x * LG_BLOCK + threadIdx.x; j < n_work; j += (unsigned long long)gridDim.x * LG_BLOCK) {
        Pixel px;
        if (level == 0u) {
            px = pixel_of(P, P.tile0 + (uint32_t)(j >> 6), (uint32_t)(j & 63u));
            if (!px.active) continue;
        }
        V3 value{P.wf_out[j], P.wf_out[n + j], P.wf_out[2 * n + j]};
        const uint32_t c0 = P.wf_child[j];
        if (c0 != WF_MISS) {
            const uint32_t c1 = P.wf_child[n + j];
            const double *sp = P.wf_spec + j;
            V3 reflected = vzero(), refracted = vzero();
            if (c0 != WF_NONE) {
                const V3 l{P.wf_out_next[c0], P.wf_out_next[nn + c0], P.wf_out_next[2 * nn + c0]};
                reflected = mul_ew(V3{sp[0 * n], sp[1 * n], sp[2 * n]}, l); // integrate.rs:103
            }
            if (c1 != WF_NONE) {
                const V3 l{P.wf_out_next[c1], P.wf_out_next[nn + c1], P.wf_out_next[2 * nn + c1]};
                refracted = mul_ew(V3{sp[3 * n], sp[4 * n], sp[5 * n]}, l) * sp[6 * n] / sp[7 * n]; // integrate.rs:129
            }
            value = value + reflected + refracted; // integrate.rs:79
            if (level != 0u) { P.wf_out[j] = value.x; P.wf_out[n + j] = value.y; P.wf_out[2 * n + j] = value.z; }
        }
        if (level == 0u) finish_pixel(P, px, j, value);
    }
}

// ------------------------------------------------------------------------------------------
// known-answer / arithmetic probe kernels (one thread; test hooks of the C ABI)
// ------------------------------------------------------------------------------------------
// kind 0 sphere (cx,cy,cz,r), 1 cuboid (min,max), 2 every triangle of a mesh in order.
// out = { hit, t, ng.xyz, ns.xyz } -- what the reference's inline tests assert on.
__global__ void kat_kernel(int kind, const double *params, const float *vpos, const uint32_t *tri_v, uint32_t ntri, V3 o, V3 d,
                           double *out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Ray ray = ray_new(o, d);
    Isect is;
    isect_set(is, INFINITY, vzero(), vzero());
    bool hit = false;
    if (kind == 0) {
        DSphere s{params[0], params[1], params[2], params[3]};
        bool inside;
        double t = sphere_t(ray, V3{s.cx, s.cy, s.cz}, s.r, inside);
        if (!(t < 0.0) && !(t >= is.t)) { sphere_full(s, ray, t, inside, is); hit = true; }
    } else if (kind == 1) {
        double mn[3] = {params[0], params[1], params[2]}, mx[3] = {params[3], params[4], params[5]};
        double t; V3 d0, d1;
        if (cuboid_hit<true>(mn, mx, ray, t, d0, d1) && !(t >= is.t)) {
            isect_set(is, t, d0, d1);
            is.has_n = true; is.n = face_forward(cross(d0, d1), -ray.d);
            hit = true;
        }
    } else {
        DParams P{};
        P.vpos = vpos; P.tri_v = tri_v;
        for (uint32_t f = 0; f < ntri; ++f) {
            const uint32_t *vi = tri_v + 3ull * f;
            TriHit h;
            if (!triangle_t(load_f3(vpos, vi[0]), load_f3(vpos, vi[1]), load_f3(vpos, vi[2]), ray, h)) continue;
            if (h.t >= is.t) continue;
            triangle_full(P, f, 0u, ray, is);
            hit = true;
        }
    }
    V3 ng = normalize(cross(is.gu, is.gv));
    V3 ns = is.has_n ? normalize(is.n) : normalize(cross(is.su, is.sv));
    out[0] = hit ? 1.0 : 0.0; out[1] = is.t;
    out[2] = ng.x; out[3] = ng.y; out[4] = ng.z; out[5] = ns.x; out[6] = ns.y; out[7] = ns.z;
}
// surface.rs:194-200
__global__ void kat_si_kernel(V3 o, V3 d, double t, V3 dpdu, V3 dpdv, double *out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    V3 wo = -normalize(d);
    V3 ng = face_forward(normalize(cross(dpdu, dpdv)), wo);
    (void)o; (void)t;
    out[0] = ng.x; out[1] = ng.y; out[2] = ng.z;
}
// One pixel, traced by lane 0 with the private walk (test hook lg_trace_pixel): the primary hit, then for each
// light the shadow ray's result.  out = { t, primref, accel, nlights, then per light: t, primref; then the shadow rays' origin }.
template <bool FAST>
__global__ void trace_pixel_kernel(const DParams P, uint32_t x, uint32_t y, double *out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t *stack = lds_stack;
    Counters cnt = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const Ray ray = camera_ray(P, x, y, 0u);
    Best b;
    walk<false, FAST, false, true>(P, ray, false, stack, 1u, b, nullptr, cnt); // (the counting instantiation writes the event log)
    if (!FAST) dbg_event(P, 9.0, 0.0, b.t, (double)b.ref);
    out[0] = b.t; out[1] = (double)b.ref; out[2] = (double)b.accel; out[3] = (double)P.nlights;
    if (b.ref == NO_HIT) return;
    Shade sh;
    shade_frame(P, ray, b, sh);
    DParams Q = P;
    Q.dbg_log = nullptr; // the log is the primary ray's
    for (uint32_t l = 0; l < P.nlights; ++l) {
        const DLight L = P.lights[l];
        const Ray sray = ray_new(sh.p, V3{L.pos[0], L.pos[1], L.pos[2]} - sh.p);
        Best sb;
        walk<false, FAST, false, true>(Q, sray, true, stack, 1u, sb, nullptr, cnt);
        out[4 + 2 * l] = sb.t; out[5 + 2 * l] = (double)sb.ref;
    }
    out[4 + 2 * P.nlights] = sh.p.x; out[5 + 2 * P.nlights] = sh.p.y; out[6 + 2 * P.nlights] = sh.p.z; // origin of the shadow rays
}
__global__ void math_kernel(int op, size_t n, const double *a, const double *b, double *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r;
    switch (op) {
    case 0: r = sqrt(a[i]); break;
    case 1: r = a[i] / b[i]; break;
    case 2: r = p_sin(a[i]); break;
    case 3: r = p_cos(a[i]); break;
    case 4: r = p_atan2(a[i], b[i]); break;
    case 5: r = p_acos(a[i]); break;
    case 6: r = fmin_(a[i], b[i]); break;
    case 7: r = fmax_(a[i], b[i]); break;
    case 8: r = (double)to_byte(a[i]); break;
    default: r = 0.0; break;
    }
    out[i] = r;
}

// ------------------------------------------------------------------------------------------
// rate probes (lg_probe_rate): the two memory denominators of the roofline bookkeeping, measured on the box
// ------------------------------------------------------------------------------------------
// 16 bytes per lane, grid-stride: the float4 copy the HBM figure of MI355X_MICROARCH.md is quoted on
__global__ void __launch_bounds__(256) probe_copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// every lane streams conflict-free 16-byte reads from a 64 KB LDS window (ds_read_b128, the instruction the
// LDS-resident scene is walked with); `sink` is written only if the xor of everything read is a magic value
__global__ void __launch_bounds__(1024) probe_lds_kernel(uint32_t iters, uint32_t *sink) {
    uint4 *lds = reinterpret_cast<uint4 *>(lds_stack);
    for (uint32_t i = threadIdx.x; i < 4096u; i += 1024u) lds[i] = uint4{i, i + 1u, i + 2u, i + 3u};
    __syncthreads();
    uint4 acc{0u, 0u, 0u, 0u};
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (uint32_t j = 0; j < 16u; ++j) {
            const uint4 v = lds[(threadIdx.x + ((it + j) & 3u) * 1024u) & 4095u];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) sink[blockIdx.x] = acc.x;
}
hipError_t launch_probe_copy(const void *src, void *dst, size_t bytes, hipStream_t stream) {
    hipLaunchKernelGGL(probe_copy_kernel, dim3(256 * 8), dim3(256), 0, stream, (const uint4 *)src, (uint4 *)dst, bytes / 16);
    return hipGetLastError();
}
hipError_t launch_probe_lds(uint32_t blocks, uint32_t iters, uint32_t *sink, hipStream_t stream) {
    hipLaunchKernelGGL(probe_lds_kernel, dim3(blocks), dim3(1024), 65536, stream, iters, sink);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// host-callable launchers (used by capi.cpp)
// ------------------------------------------------------------------------------------------
hipError_t launch_trace(const DParams &P, bool stats, bool fast, uint32_t blocks, uint32_t stack_depth, hipStream_t stream) {
    const bool prune = P.prune && !fast;
    if (P.lds_image && !stats && !fast) { // LDS-resident scene: `blocks` = one 1024-lane workgroup per CU
        size_t lds = (size_t)P.stack_depth * LG_LDSS_BLOCK * sizeof(uint32_t) + (size_t)P.lds_image_n16 * 16u;
        if (prune) hipLaunchKernelGGL((trace_kernel<false, false, true, true>), dim3(blocks), dim3(LG_LDSS_BLOCK), lds, stream, P);
        else hipLaunchKernelGGL((trace_kernel<false, false, true>), dim3(blocks), dim3(LG_LDSS_BLOCK), lds, stream, P);
        return hipGetLastError();
    }
    size_t lds = (size_t)stack_depth * LG_BLOCK * sizeof(uint32_t);
    if (prune) {
        if (stats) hipLaunchKernelGGL((trace_kernel<true, false, false, true>), dim3(blocks), dim3(LG_BLOCK), lds, stream, P);
        else hipLaunchKernelGGL((trace_kernel<false, false, false, true>), dim3(blocks), dim3(LG_BLOCK), lds, stream, P);
        return hipGetLastError();
    }
    if (fast) {
        if (stats) hipLaunchKernelGGL((trace_kernel<true, true, false>), dim3(blocks), dim3(LG_BLOCK), lds, stream, P);
        else hipLaunchKernelGGL((trace_kernel<false, true, false>), dim3(blocks), dim3(LG_BLOCK), lds, stream, P);
    } else {
        if (stats) hipLaunchKernelGGL((trace_kernel<true, false, false>), dim3(blocks), dim3(LG_BLOCK), lds, stream, P);
        else hipLaunchKernelGGL((trace_kernel<false, false, false>), dim3(blocks), dim3(LG_BLOCK), lds, stream, P);
    }
    return hipGetLastError();
}
hipError_t launch_stream_fixup(const DParams &P, bool shadow, uint32_t blocks, hipStream_t stream) {
    size_t lds = (size_t)P.stack_depth * LG_BLOCK * sizeof(uint32_t);
    if (shadow) hipLaunchKernelGGL((stream_fixup_kernel<true>), dim3(blocks), dim3(LG_BLOCK), lds, stream, P);
    else hipLaunchKernelGGL((stream_fixup_kernel<false>), dim3(blocks), dim3(LG_BLOCK), lds, stream, P);
    return hipGetLastError();
}
hipError_t launch_stream_packet(const DParams &P, bool shadow, uint32_t blocks, hipStream_t stream) {
    const bool ldss = P.lds_image != nullptr;
    const uint32_t block = ldss ? LG_LDSS_BLOCK : LG_BLOCK;
    size_t lds = (size_t)(block / 64u) * P.stack_depth * PKT_ENTRY * sizeof(uint32_t) + (ldss ? (size_t)P.lds_image_n16 * 16u : 0u);
    if (ldss) { if (shadow) hipLaunchKernelGGL((stream_packet_kernel<true, true>), dim3(blocks), dim3(block), lds, stream, P);
                else hipLaunchKernelGGL((stream_packet_kernel<false, true>), dim3(blocks), dim3(block), lds, stream, P); }
    else { if (shadow) hipLaunchKernelGGL((stream_packet_kernel<true, false>), dim3(blocks), dim3(block), lds, stream, P);
           else hipLaunchKernelGGL((stream_packet_kernel<false, false>), dim3(blocks), dim3(block), lds, stream, P); }
    return hipGetLastError();
}
hipError_t stream_packet_occupancy(uint32_t stack_depth, int *blocks_per_cu) { // the 256-lane form (scene in L1/L2)
    size_t lds = (size_t)(LG_BLOCK / 64u) * stack_depth * PKT_ENTRY * sizeof(uint32_t);
    int a = 0, b = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, stream_packet_kernel<false, false>, LG_BLOCK, lds);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, stream_packet_kernel<true, false>, LG_BLOCK, lds);
    *blocks_per_cu = a < b ? a : b;
    return e;
}
// raise the dynamic-LDS limit of the LDS-resident-scene variants to `bytes`
hipError_t stream_trace_ldss_prepare(size_t bytes) {
    const void *fns[10] = {reinterpret_cast<const void *>(trace_kernel<false, false, true>),
                          reinterpret_cast<const void *>(trace_kernel<false, false, true, true>),
                          reinterpret_cast<const void *>(wf_trace_kernel<false, false, true, true, true>),
                          reinterpret_cast<const void *>(wf_trace_kernel<false, true, true, false, true>),
                          reinterpret_cast<const void *>(wf_trace_kernel<false, false, true, false, true>),
                          reinterpret_cast<const void *>(wf_trace_kernel<false, false, true, true>),
                          reinterpret_cast<const void *>(wf_trace_kernel<false, true, true, false>),
                          reinterpret_cast<const void *>(wf_trace_kernel<false, false, true, false>),
                          reinterpret_cast<const void *>(stream_packet_kernel<false, true>),
                          reinterpret_cast<const void *>(stream_packet_kernel<true, true>)};
    for (const void *f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
hipError_t launch_wf_trace(const DParams &P, bool fast, bool shadow, uint32_t blocks, uint32_t stack_depth, hipStream_t stream) {
    const bool ldss = P.lds_image && !fast; // LDS-resident scene: `blocks` = one workgroup per CU
    const bool l0 = !shadow && P.wf_level == 0u;
    const uint32_t block = ldss ? LG_LDSS_BLOCK : LG_BLOCK;
    const uint32_t depth = fast ? stack_depth : P.stack_depth;
    size_t lds = (size_t)depth * block * sizeof(uint32_t) + (ldss ? (size_t)P.lds_image_n16 * 16u : 0u);
#define LG_LAUNCH(F, S, L, Z) hipLaunchKernelGGL((wf_trace_kernel<F, S, L, Z>), dim3(blocks), dim3(block), lds, stream, P)
#define LG_LAUNCH_PRUNED(S, L, Z) hipLaunchKernelGGL((wf_trace_kernel<false, S, L, Z, true>), dim3(blocks), dim3(block), lds, stream, P)
    if (P.prune && !fast) {
        if (ldss) { if (shadow) LG_LAUNCH_PRUNED(true, true, false); else if (l0) LG_LAUNCH_PRUNED(false, true, true); else LG_LAUNCH_PRUNED(false, true, false); }
        else { if (shadow) LG_LAUNCH_PRUNED(true, false, false); else if (l0) LG_LAUNCH_PRUNED(false, false, true); else LG_LAUNCH_PRUNED(false, false, false); }
    } else
    if (fast) { if (shadow) LG_LAUNCH(true, true, false, false); else if (l0) LG_LAUNCH(true, false, false, true); else LG_LAUNCH(true, false, false, false); }
    else if (ldss) { if (shadow) LG_LAUNCH(false, true, true, false); else if (l0) LG_LAUNCH(false, false, true, true); else LG_LAUNCH(false, false, true, false); }
    else { if (shadow) LG_LAUNCH(false, true, false, false); else if (l0) LG_LAUNCH(false, false, false, true); else LG_LAUNCH(false, false, false, false); }
#undef LG_LAUNCH
#undef LG_LAUNCH_PRUNED
    return hipGetLastError();
}
hipError_t launch_wf_shade(const DParams &P, uint32_t blocks, hipStream_t stream) {
    const bool l0 = P.wf_level == 0u;
    if (P.wf_levels == 1u) hipLaunchKernelGGL((wf_shade_kernel<0, true>), dim3(blocks), dim3(LG_BLOCK), 0, stream, P);
    else if (P.wf_level + 1u >= P.wf_levels) hipLaunchKernelGGL((wf_shade_kernel<2, false>), dim3(blocks), dim3(LG_BLOCK), 0, stream, P);
    else if (l0) hipLaunchKernelGGL((wf_shade_kernel<1, true>), dim3(blocks), dim3(LG_BLOCK), 0, stream, P);
    else hipLaunchKernelGGL((wf_shade_kernel<1, false>), dim3(blocks), dim3(LG_BLOCK), 0, stream, P);
    return hipGetLastError();
}
hipError_t launch_wf_combine(const DParams &P, uint32_t blocks, hipStream_t stream) {
    hipLaunchKernelGGL(wf_combine_kernel, dim3(blocks), dim3(LG_BLOCK), 0, stream, P);
    return hipGetLastError();
}
hipError_t wf_trace_occupancy(uint32_t stack_depth, bool fast, int *blocks_per_cu) {
    size_t lds = (size_t)stack_depth * LG_BLOCK * sizeof(uint32_t);
    int a = 0, b = 0;
    hipError_t e;
    if (fast) {
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, wf_trace_kernel<true, false, false, true>, LG_BLOCK, lds);
        if (e != hipSuccess) return e;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, wf_trace_kernel<true, true, false, false>, LG_BLOCK, lds);
    } else {
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, wf_trace_kernel<false, false, false, true>, LG_BLOCK, lds);
        if (e != hipSuccess) return e;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, wf_trace_kernel<false, true, false, false>, LG_BLOCK, lds);
    }
    *blocks_per_cu = a < b ? a : b;
    return e;
}
hipError_t launch_stream_shade(const DParams &P, hipStream_t stream) {
    uint32_t blocks = (uint32_t)((P.n_items + LG_BLOCK - 1) / LG_BLOCK);
    hipLaunchKernelGGL(stream_shade_kernel, dim3(blocks), dim3(LG_BLOCK), 0, stream, P);
    return hipGetLastError();
}
hipError_t trace_occupancy(uint32_t stack_depth, bool fast, int *blocks_per_cu) {
    size_t lds = (size_t)stack_depth * LG_BLOCK * sizeof(uint32_t);
    if (fast) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, trace_kernel<false, true, false>, LG_BLOCK, lds);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, trace_kernel<false, false, false>, LG_BLOCK, lds);
}
hipError_t trace_set_lds_limit(size_t bytes) {
    const void *fns[17] = {reinterpret_cast<const void *>(wf_trace_kernel<false, false, false, true>), reinterpret_cast<const void *>(wf_trace_kernel<false, true, false, false>),
                          reinterpret_cast<const void *>(wf_trace_kernel<false, false, false, true, true>), reinterpret_cast<const void *>(wf_trace_kernel<false, true, false, false, true>),
                          reinterpret_cast<const void *>(wf_trace_kernel<false, false, false, false, true>), reinterpret_cast<const void *>(trace_kernel<false, false, false, true>),
                          reinterpret_cast<const void *>(wf_trace_kernel<true, false, false, true>), reinterpret_cast<const void *>(wf_trace_kernel<true, true, false, false>),
                          reinterpret_cast<const void *>(wf_trace_kernel<false, false, false, false>), reinterpret_cast<const void *>(wf_trace_kernel<true, false, false, false>),
                          reinterpret_cast<const void *>(trace_kernel<false, false, false>), reinterpret_cast<const void *>(trace_kernel<true, false, false>),
                          reinterpret_cast<const void *>(trace_kernel<false, true, false>), reinterpret_cast<const void *>(trace_kernel<true, true, false>),
                          reinterpret_cast<const void *>(trace_kernel<true, false, false, true>),
                          reinterpret_cast<const void *>(stream_fixup_kernel<false>), reinterpret_cast<const void *>(stream_fixup_kernel<true>)};
    for (const void *f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
hipError_t launch_kat(int kind, const double *params, const float *vpos, const uint32_t *tri_v, uint32_t ntri, V3 o, V3 d, double *out,
                      hipStream_t stream) {
    hipLaunchKernelGGL(kat_kernel, dim3(1), dim3(64), 0, stream, kind, params, vpos, tri_v, ntri, o, d, out);
    return hipGetLastError();
}
hipError_t launch_kat_si(V3 o, V3 d, double t, V3 dpdu, V3 dpdv, double *out, hipStream_t stream) {
    hipLaunchKernelGGL(kat_si_kernel, dim3(1), dim3(64), 0, stream, o, d, t, dpdu, dpdv, out);
    return hipGetLastError();
}
hipError_t launch_trace_pixel(const DParams &P, bool fast, uint32_t stack_depth, uint32_t x, uint32_t y, double *out, hipStream_t stream) {
    size_t lds = (size_t)stack_depth * sizeof(uint32_t) + 64;
    if (fast) hipLaunchKernelGGL((trace_pixel_kernel<true>), dim3(1), dim3(64), lds, stream, P, x, y, out);
    else hipLaunchKernelGGL((trace_pixel_kernel<false>), dim3(1), dim3(64), lds, stream, P, x, y, out);
    return hipGetLastError();
}
hipError_t launch_math(int op, size_t n, const double *a, const double *b, double *out, hipStream_t stream) {
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(math_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, op, n, a, b, out);
    return hipGetLastError();
}

} // namespace lg
