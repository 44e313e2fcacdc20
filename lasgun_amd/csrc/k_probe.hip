// lasgun_amd/csrc/k_probe.hip -- probes behind the test hooks of the C ABI (known-answer tests, IEEE checks, copy / LDS rates).
#include "shade.h"

namespace lg {

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


// ---- host-callable launchers (used by launch.cpp, capi.cpp)
hipError_t launch_kat(int kind, const double *params, const float *vpos, const uint32_t *tri_v, uint32_t ntri, V3 o, V3 d, double *out,
                      hipStream_t stream) {
    hipLaunchKernelGGL(kat_kernel, dim3(1), dim3(64), 0, stream, kind, params, vpos, tri_v, ntri, o, d, out);
    return hipGetLastError();
}
hipError_t launch_kat_si(V3 o, V3 d, double t, V3 dpdu, V3 dpdv, double *out, hipStream_t stream) {
    hipLaunchKernelGGL(kat_si_kernel, dim3(1), dim3(64), 0, stream, o, d, t, dpdu, dpdv, out);
    return hipGetLastError();
}
hipError_t launch_math(int op, size_t n, const double *a, const double *b, double *out, hipStream_t stream) {
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(math_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, op, n, a, b, out);
    return hipGetLastError();
}

} // namespace lg
