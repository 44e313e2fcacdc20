// lasgun_amd/csrc/k_packet.hip -- the packet organisation (opt-in, lg_accel_set_packet): one walk per wavefront, fix-up pass, shade.
#include "packet.h"
#include "shade.h"

namespace lg {

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

// ---- host-callable launchers (used by capi.cpp)
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
hipError_t launch_stream_shade(const DParams &P, hipStream_t stream) {
    uint32_t blocks = (uint32_t)((P.n_items + LG_BLOCK - 1) / LG_BLOCK);
    hipLaunchKernelGGL(stream_shade_kernel, dim3(blocks), dim3(LG_BLOCK), 0, stream, P);
    return hipGetLastError();
}
// raise the dynamic-LDS limit of this file's kernels to `bytes` (ldss: the LDS-resident-scene forms; otherwise the 256-lane forms)
hipError_t packet_set_lds_limit(size_t bytes, bool ldss) {
    const void *ldss_fns[] = {
        reinterpret_cast<const void *>(stream_packet_kernel<false, true>),
        reinterpret_cast<const void *>(stream_packet_kernel<true, true>)};
    const void *plain_fns[] = {
        reinterpret_cast<const void *>(stream_fixup_kernel<false>),
        reinterpret_cast<const void *>(stream_fixup_kernel<true>)};
    const void *const *fns = ldss ? ldss_fns : plain_fns;
    const size_t n = ldss ? sizeof ldss_fns / sizeof ldss_fns[0] : sizeof plain_fns / sizeof plain_fns[0];
    for (size_t i = 0; i < n; ++i) {
        hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

} // namespace lg
