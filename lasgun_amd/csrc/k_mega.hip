// lasgun_amd/csrc/k_mega.hip -- the megakernel: the whole of li() per lane (and its counting variant, and lg_trace_pixel).
#include "shade.h"

namespace lg {

// LDSS (reference traversal only): one 1024-lane workgroup per CU with the scene's node / primref / sphere /
// cuboid tables copied into LDS behind the stacks (see load_node); otherwise 256-lane workgroups and L1/L2.
// LB: lanes of the LDS-resident form's workgroup.  1024 = four waves per SIMD under 128 registers, what the traversal-only kernels want; 768 =
// three waves under 168 registers and hardly a spill, which this kernel -- the walk AND the shading code in one register allocation -- prefers
// on scenes of few primitives (simple.rs at 9 spp 512^2: 0.67 -> 0.61 ms, 1024^2: 1.84 -> 1.33; Cornell glass 1024^2: 1.14 -> 1.00; the
// 1024-sphere scene loses 8 %: more walk than shading, and the walk wants the fourth wave).  The host chooses (DParams::mega_lanes).
constexpr int LG_MEGA_NARROW = 768;
template <bool STATS, bool FAST, bool LDSS, bool PRUNE = false, int LB = LG_LDSS_BLOCK>
__global__ void __launch_bounds__(LDSS ? LB : LG_BLOCK, (LDSS && LB == LG_MEGA_NARROW) ? 3 : LG_WAVES_PER_SIMD) trace_kernel(const DParams P) {
    static_assert(!(PRUNE && FAST), "the fast mode prunes its own trees by its own rule");
    static_assert(!(FAST && LDSS) && !(STATS && LDSS), "LDS-resident scene: plain reference traversal only");
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const unsigned long long gtid = (unsigned long long)blockIdx.x * blockDim.x + tid;
    uint32_t *stack = lds_stack + tid; // entry i at stack[i * stride]: bank = tid % 32 for every i
    constexpr uint32_t stride = LDSS ? (uint32_t)LB : LG_BLOCK;
    const uint4 *scn = nullptr;
    if (LDSS) {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_stack + P.stack_depth * stride);
        const uint4 *src = reinterpret_cast<const uint4 *>(P.lds_image);
        copy_to_lds(dst, src, P.lds_image_n16, tid, stride);
        __syncthreads(); // the only workgroup-wide step; every wave reaches it before pulling tiles
        scn = dst;
    }
    const uint4 *const arec = (LDSS || FAST) ? nullptr : load_accel_image(P, P.stack_depth * LG_BLOCK);
    Counters cnt = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!wave_has_work(P.ntiles)) return; // (after the workgroup's barrier: a small film leaves most of the grid nothing to claim)
#ifdef LG_QIDLE // diagnostic build (tools/queue_idle.py): when this wave started and left, in 100 MHz ticks -> how much of the launch its waves sit out
    const unsigned long long idle_t0 = wall_clock64();
#endif

    for (bool final = false; !final;) {
        // ---- fetch the next 64-pixel tile for this wavefront.  (One head word for the whole chip here: the per-XCD bands of
        // claim_tile, which buy the traversal-only kernels 9 %, cost this kernel 5-12 % on every config it runs -- measured, and again
        // in round 4 with the cheap launch end of kcommon.h: 1b 0.71 -> 0.78 ms, Cornell glass 0.80 -> 0.88, only one-sphere scenes gain.)
        const uint32_t tile = claim_tile_single(P.tile_counter, P.ntiles, final);
        if (tile == NO_TILE) break; // (a wave leaves here, or after one of the launch's last tiles: kcommon.h)

        // (claimed from the last tile down: DParams::tile_rev; samples side by side: a tile is a pixel tile at ONE of its samples, DParams::ss_par)
        // (a small launch may hand a tile out in 2^split_shift parts of 64 >> split_shift lanes each, DParams::split_shift: scalar shifts and one
        // compare -- written with a division by the part count this cost the kernel 5-8 % on configs 4, 4m and 5 through its register allocation)
        uint32_t vtile = tile_in_order(P, tile);
#ifndef LG_NO_SPLIT
        const uint32_t part = vtile & ((1u << P.split_shift) - 1u);
        vtile >>= P.split_shift;
#endif
        uint32_t s_first;
        const Pixel px = pixel_of(P, l0_tile(P, vtile, s_first), lane);
        const uint32_t x = px.x, y = px.y;
#ifndef LG_NO_SPLIT
        const bool active = px.active && (lane >> (6u - P.split_shift)) == part;
#else
        const bool active = px.active;
#endif
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
        const uint32_t s_end = P.ss_par > 1u ? s_first + 1u : nsamples;
        for (uint32_t sidx = s_first; sidx < s_end; ++sidx) {
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
                    walk<LDSS, FAST, PRUNE, STATS>(P, tray, shadow_job, stack, stride, b, scn, cnt, arec);
                    if (STATS && FAST && P.audit) audit_fast_ray(P, tray, shadow_job, stack, stride, b, cnt, arec); // lg_audit_fast
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
        if (P.ss_par > 1u) { // this sample's li() (0 + value) parked; wf_resolve_kernel sums the pixel's samples in their order
            const unsigned long long i = (unsigned long long)vtile * 64ull + lane;
            P.accum[i] = color.x; P.accum[P.n_items + i] = color.y; P.accum[2 * P.n_items + i] = color.z;
            continue;
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

#ifdef LG_QIDLE
    if (!STATS && lane == 0u && P.stats) {
        const unsigned long long t1 = wall_clock64();
        atomicMax(&P.stats->primary_rays, (1ull << 62) - idle_t0); // the earliest start
        atomicMax(&P.stats->shadow_rays, t1);                       // the last exit
        atomicAdd(&P.stats->secondary_rays, t1);                    // sum of the exits
        atomicAdd(&P.stats->nodes_tested, 1ull);                    // waves
    }
#endif
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
        if (P.audit) {
            atomicAdd(&P.stats->audit_nodes, (unsigned long long)cnt.a_nodes);
            atomicAdd(&P.stats->audit_runs, (unsigned long long)cnt.a_runs);
            atomicAdd(&P.stats->audit_prims, (unsigned long long)cnt.a_prims);
            atomicAdd(&P.stats->audit_violations, (unsigned long long)cnt.a_viol);
            // (the slacks of non-violating primitives are >= 0, so their f64 bit patterns order like the numbers; the record starts
            // zeroed, so the MINIMUM is kept as the maximum of the complemented bits: 0 = no sample)
            atomicMax(&P.stats->audit_slack_nodes, ~(unsigned long long)__double_as_longlong(fmax_(cnt.a_slack_n, 0.0)));
            atomicMax(&P.stats->audit_slack_runs, ~(unsigned long long)__double_as_longlong(fmax_(cnt.a_slack_r, 0.0)));
            atomicMax(&P.stats->audit_used_nodes, (unsigned long long)__double_as_longlong(fmax_(cnt.a_used_n, 0.0)));
        }
    }
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

// ---- host-callable launchers (used by launch.cpp, capi.cpp)
// ------------------------------------------------------------------------------------------
// host-callable launchers (used by launch.cpp, capi.cpp)
// ------------------------------------------------------------------------------------------
hipError_t launch_trace(const DParams &P, bool stats, bool fast, uint32_t blocks, uint32_t stack_depth, hipStream_t stream) {
    const bool prune = P.prune && !fast;
    if (P.lds_image && !stats && !fast) { // LDS-resident scene: `blocks` = one workgroup per CU, of 1024 or of 768 lanes
        const bool narrow = P.mega_lanes == (uint32_t)LG_MEGA_NARROW;
        const size_t lds = (size_t)P.stack_depth * (narrow ? LG_MEGA_NARROW : LG_LDSS_BLOCK) * sizeof(uint32_t) + (size_t)P.lds_image_n16 * 16u;
        if (narrow) {
            if (prune) hipLaunchKernelGGL((trace_kernel<false, false, true, true, LG_MEGA_NARROW>), dim3(blocks), dim3(LG_MEGA_NARROW), lds, stream, P);
            else hipLaunchKernelGGL((trace_kernel<false, false, true, false, LG_MEGA_NARROW>), dim3(blocks), dim3(LG_MEGA_NARROW), lds, stream, P);
        } else if (prune) hipLaunchKernelGGL((trace_kernel<false, false, true, true>), dim3(blocks), dim3(LG_LDSS_BLOCK), lds, stream, P);
        else hipLaunchKernelGGL((trace_kernel<false, false, true>), dim3(blocks), dim3(LG_LDSS_BLOCK), lds, stream, P);
        return hipGetLastError();
    }
    size_t lds = (size_t)stack_depth * LG_BLOCK * sizeof(uint32_t) + (!fast && P.accel_image ? (size_t)P.accel_image_n16 * 16u : 0u);
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
hipError_t trace_occupancy(uint32_t stack_depth, bool fast, size_t extra_lds, int *blocks_per_cu) {
    size_t lds = (size_t)stack_depth * LG_BLOCK * sizeof(uint32_t) + (fast ? 0u : extra_lds);
    if (fast) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, trace_kernel<false, true, false>, LG_BLOCK, lds);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, trace_kernel<false, false, false>, LG_BLOCK, lds);
}
hipError_t launch_trace_pixel(const DParams &P, bool fast, uint32_t stack_depth, uint32_t x, uint32_t y, double *out, hipStream_t stream) {
    size_t lds = (size_t)stack_depth * sizeof(uint32_t) + 64;
    if (fast) hipLaunchKernelGGL((trace_pixel_kernel<true>), dim3(1), dim3(64), lds, stream, P, x, y, out);
    else hipLaunchKernelGGL((trace_pixel_kernel<false>), dim3(1), dim3(64), lds, stream, P, x, y, out);
    return hipGetLastError();
}
// raise the dynamic-LDS limit of this file's kernels to `bytes` (ldss: the LDS-resident-scene forms; otherwise the 256-lane forms)
hipError_t mega_set_lds_limit(size_t bytes, bool ldss) {
    const void *ldss_fns[] = {
        reinterpret_cast<const void *>(trace_kernel<false, false, true>),
        reinterpret_cast<const void *>(trace_kernel<false, false, true, true>),
        reinterpret_cast<const void *>(trace_kernel<false, false, true, false, LG_MEGA_NARROW>),
        reinterpret_cast<const void *>(trace_kernel<false, false, true, true, LG_MEGA_NARROW>)};
    const void *plain_fns[] = {
        reinterpret_cast<const void *>(trace_kernel<false, false, false>),
        reinterpret_cast<const void *>(trace_kernel<true, false, false>),
        reinterpret_cast<const void *>(trace_kernel<false, true, false>),
        reinterpret_cast<const void *>(trace_kernel<true, true, false>),
        reinterpret_cast<const void *>(trace_kernel<false, false, false, true>),
        reinterpret_cast<const void *>(trace_kernel<true, false, false, true>)};
    const void *const *fns = ldss ? ldss_fns : plain_fns;
    const size_t n = ldss ? sizeof ldss_fns / sizeof ldss_fns[0] : sizeof plain_fns / sizeof plain_fns[0];
    for (size_t i = 0; i < n; ++i) {
        hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

} // namespace lg
