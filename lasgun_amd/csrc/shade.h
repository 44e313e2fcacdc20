// lasgun_amd/csrc/shade.h -- BxDFs, materials, the shading frame of a hit, camera rays, pixel addressing, quantisation.
#pragma once
#include "walk.h"

namespace lg {
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
#ifndef LG_NO_LATTICE // (A/B: what this branch costs the kernels it is inlined into)
    } else if (P.mode == 4 || P.mode == 5) {
        // A strided subset {k + i*n} (lib.rs:152) -- mode 4 -- or several of one n (lg_capture_subsets) -- mode 5 -- as the LATTICE it is: its
        // pixels sit at x = (k - y*w) mod n + n*c in row y, one per period of n.  A tile is 64 (mode 5: sub_rows = 64 / m) consecutive rows of one
        // lattice column c -- a window of <= 64 rows x <= n pixels, the densest 64 pixels of the subset there are -- instead of 64 consecutive i
        // (for the progressive front end's n = 100 on a 4096-wide film: a strip 6,400 pixels long and a row and a half high, whose rays share no
        // node of a BVH).  Which lane takes which pixel of the subset never changes a pixel.  No division here (this function is inlined into
        // kernels that live at their register budget: the first form, with 64-bit remainders, cost configs 4 / 4m / 5 3-6 %): the host tabulates
        // floor(y*w / n) and (y*w) mod n per row (sub_rowtab, one table per (w, h, n) and launch context) and k mod n, k / n per subset.
        const uint32_t ty = tile / P.sub_cols, c = tile - ty * P.sub_cols;
        uint32_t r = lane, j = 0u, kk = P.sub_kk, kdiv = P.sub_kdiv;
        if (P.mode == 5) {
            r = lane / P.sub_m; j = lane - r * P.sub_m;
            const unsigned long long kd = P.pixel_list[P.sub_m + j]; // (k_j mod n) | (k_j / n) << 32, behind the m values of k
            kk = (uint32_t)kd; kdiv = (uint32_t)(kd >> 32);
        }
        const uint32_t y = ty * P.sub_rows + r;
        const bool in = r < P.sub_rows && y < P.h;
        const DRowTab rt = P.sub_rowtab[in ? y : 0u]; // (floor(y*w / n), (y*w) mod n)
        int32_t ph = (int32_t)kk - (int32_t)rt.rem;
        const uint32_t wrapped = ph < 0 ? 1u : 0u;
        if (wrapped) ph += (int32_t)P.sub_n;
        const uint32_t x = (uint32_t)ph + (uint32_t)P.sub_n * c;
        const long long q = (long long)rt.base + (long long)c + (long long)wrapped - (long long)kdiv; // the pixel's place in its subset (negative: before k)
        px.active = in && x < P.w && q >= 0;
        px.x = px.active ? x : 0u;
        px.y = px.active ? y : 0u;
        const unsigned long long place = P.mode == 5 ? (unsigned long long)q * P.sub_m + j : (unsigned long long)q; // (mode 3's work item q * m + j)
        px.pix = px.active ? (P.out_compact ? place : (unsigned long long)y * P.w + x) : 0ull;
#endif
    } else {
        unsigned long long i = (unsigned long long)tile * 64ull + lane;
        px.active = i < P.sub_count;
        // mode 1: the strided subset {k + i*n} (lib.rs:152); mode 2: an explicit list of pixel offsets; mode 3: SEVERAL strided subsets of one
        // n in one launch (lg_capture_subsets) -- work item i = q * m + j is pixel ks[j] + q*n, ks = pixel_list[0 .. m) ascending: with every
        // k of 0 .. n-1 listed that is pixel i itself, the row-major film (the host keeps sub_count below 2^32 for this mode)
        unsigned long long off;
        if (P.mode == 1) off = P.sub_k + i * P.sub_n;
        else if (P.mode == 3) {
            const uint32_t q = (uint32_t)i / P.sub_m, j = (uint32_t)i - q * P.sub_m;
            off = P.pixel_list[j] + (unsigned long long)q * P.sub_n;
            px.active = px.active && off < (unsigned long long)P.w * P.h; // (the last period of a subset may end before the others')
            if (!px.active) off = 0ull;
        } else off = px.active ? P.pixel_list[i] : 0ull;
        px.x = (uint32_t)(off % P.w);
        px.y = (uint32_t)(off / P.w);
        px.pix = P.out_compact ? i : off;
    }
    return px;
}

extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];

// 256-lane kernels (scene tables in L2): the accel records copied in behind the per-lane stacks (walk.h, lvl_set); nullptr when the
// scene has too many accels for that.  Called by every wave of the workgroup before it pulls work (one barrier).
__device__ __forceinline__ const uint4 *load_accel_image(const DParams &P, uint32_t stack_words) {
    if (!P.accel_image) return nullptr;
    uint4 *dst = reinterpret_cast<uint4 *>(lds_stack + stack_words);
    const uint4 *src = reinterpret_cast<const uint4 *>(P.accel_image);
    copy_to_lds(dst, src, P.accel_image_n16, threadIdx.x, blockDim.x);
    __syncthreads();
    return dst;
}

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

// the s-th tile a launch hands out (DParams::tile_rev): 0 = in order, 1 = from the last tile down, 2 = from the middle outwards
__device__ __forceinline__ uint32_t tile_in_order(const DParams &P, uint32_t s) {
    if (P.tile_rev == 0u) return s;
    if (P.tile_rev == 1u) return P.ntiles - 1u - s;
    const uint32_t mid = P.ntiles >> 1;
    return (s & 1u) ? mid - 1u - (s >> 1) : mid + (s >> 1);
}
// level 0 of the level-by-level pipeline: the pixel tile and the sample a work tile stands for (DParams::ss_par)
__device__ __forceinline__ uint32_t l0_tile(const DParams &P, uint32_t vt, uint32_t &sample) {
    if (P.ss_par <= 1u) { sample = P.sample_index; return vt; }
    const uint32_t t = vt / P.ss_par;
    sample = vt - t * P.ss_par;
    return t;
}
// the film word / radiance triple of a finished pixel (img.rs:46-67)
__device__ __forceinline__ void write_pixel(const DParams &P, const Pixel &px, V3 color) {
    const unsigned long long pix = px.pix;
    if (P.out_rgba) {
        uint32_t rgba = to_byte(color.x) | (to_byte(color.y) << 8) | (to_byte(color.z) << 16) | (255u << 24);
        reinterpret_cast<uint32_t *>(P.out_rgba)[pix] = rgba;
    }
    if (P.out_radiance) {
        P.out_radiance[3 * pix] = color.x; P.out_radiance[3 * pix + 1] = color.y; P.out_radiance[3 * pix + 2] = color.z;
    }
}
// integrate(): sum over the pixel's samples, then * weight; Img::set (integrate.rs:16-20, img.rs:46-67)
__device__ __forceinline__ void finish_pixel(const DParams &P, const Pixel &px, unsigned long long widx, V3 value) {
    const uint32_t nsamples = P.ss_root * P.ss_root;
    if (P.ss_par > 1u) { // samples side by side: this sample's li() is parked; the resolve pass sums the pixel's samples in their order
        P.accum[widx] = value.x; P.accum[P.n_items + widx] = value.y; P.accum[2 * P.n_items + widx] = value.z;
        return;
    }
    V3 color = vzero();
    if (P.sample_index > 0) color = V3{P.accum[widx], P.accum[P.n_items + widx], P.accum[2 * P.n_items + widx]};
    color = color + value;
    if (P.sample_index + 1 < nsamples) {
        P.accum[widx] = color.x; P.accum[P.n_items + widx] = color.y; P.accum[2 * P.n_items + widx] = color.z;
        return;
    }
    const double weight = 1. / (double)nsamples;
    color = color * weight;
    write_pixel(P, px, color);
}
// li() of a hit up to its specular children: the lights in order, then the ambient term (integrate.rs:47-67).  `vis` bit l:
// light l is visible from the hit (PointLight::sample, point.rs:49).  Shared by the level-by-level shade pass and the queue kernel.
__device__ __forceinline__ V3 shade_lights(const DParams &P, const DMaterial &m, const Shade &sh, const uint32_t vis) {
    V3 output = vzero();
    const V3 nrm = sh.ns;
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
    return output + mul_ew(P.ambient, bsdf_f(m, sh, sh.wo, nrm)); // integrate.rs:67
}

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

} // namespace lg
