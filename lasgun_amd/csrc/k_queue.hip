// lasgun_amd/csrc/k_queue.hip -- the queue organisation: li() for every recursion level of a chunk of the film in ONE persistent launch.
#include "shade.h"

namespace lg {

// ------------------------------------------------------------------------------------------
// Why a third organisation.  On scenes whose rays are long, uneven walks -- a 100k-triangle mesh in the
// reference's 254-triangle leaves -- a wave is busy for a millisecond or more with one 8x8 tile.  The megakernel gives every lane
// its pixel's whole ray tree (integrate.rs:23-132): deep levels run with a few lanes of the wave, the recursion state lives in
// 300-500 spilled registers, and their scratch traffic competes with the scene tables for the L2.  The level-by-level pipeline
// packs every level's rays into full waves, but each of its ~15 launches per chunk ends with a tail as long as its slowest wave.
// Here ONE launch does both: its waves pull 64-ray PACKETS from per-level queues -- level 0's packets are the chunk's 8x8 pixel
// tiles, level d + 1's are filled by level d's specular hits (ballot + prefix appends, as in wf_shade_kernel) -- and a packet is
// taken through closest hit -> shading frame -> per-light any-hit -> radiance -> children by the wave that claimed it, with ONE
// traversal call site in a wave-uniform job loop: what is live across a walk is a ray and a few words.  Deepest non-empty level
// first, so the expensive secondary rays start early and the launch has one tail, not fifteen.  The levels are then combined
// bottom-up by wf_combine_kernel (integrate.rs:79, 103, 129), exactly as in the level-by-level pipeline: same arrays, same order
// of operations -- every f64 comes from the same expression in all three organisations.
//
// Scheduling state (DParams::q_ctl, QC_*; every hot word on a 128-byte line of its own): per level a ray count (appends), a claim
// counter and one 64-bit word (packets + 1) << 32 | packets done; per packet of the levels >= 1 a ready word.  A packet may be
// claimed when its ready word says all its rays are written: 64, or QR_LAST | n for the last, partial packet of a level, set by
// the wave that saw the level above complete.  Hand-off of the ray data between waves on different CUs / XCDs (MI355X_MICROARCH.md,
// inter-workgroup visibility): the producer stores the rays WRITE-THROUGH (agent-scope relaxed atomic stores: global_store sc1;
// a release fence would write back the XCD's whole dirty L2 -- megabytes of parked frames and results -- per packet), waits for
// them (s_waitcnt vmcnt(0)), then raises the ready words with agent atomics; the consumer polls with relaxed agent loads, claims
// by compare-and-swap, runs an agent-scope acquire and reads the rays.  Exactly one atomic operation on a level's 64-bit word
// observes "count final and every packet done"; that wave marks the next level's last packet and publishes its packet count.
// Level 0's packets are counted per wave and flushed when the wave turns to a deeper level or finds the tiles exhausted: one
// atomic per tile (the claim), as in the megakernel.  Every wave leaves through QC_FINISHED (or, should the protocol ever stall,
// through the poll limit with QC_ERROR set -- reported by the host, never silent).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t q_lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t q_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void q_store_wt(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } // write-through (sc1)
__device__ __forceinline__ uint32_t *q_level(const DParams &P, uint32_t d) { return P.q_ctl + QC_LEVEL0 + QC_LEVEL_WORDS * d; }
__device__ __forceinline__ uint32_t *q_ready(const DParams &P, uint32_t d) { // ready words of level d >= 1: the packets of levels 1 .. d-1 (+ slack each) before them
    return P.q_ready + (size_t)(P.n_items >> 6) * ((1u << d) - 2u) + (size_t)(d - 1u) * QR_SLACK;
}
constexpr uint32_t Q_EXIT = 0xFFFFFFFEu;
constexpr uint32_t Q_POLL_LIMIT = 1u << 21; // polls of an idle wave (>= 30 microseconds each with the back-off) before it gives up: a minute or so

// a level is complete: the next level's count is final; cascades through empty levels; the last level sets QC_FINISHED
__device__ __forceinline__ void q_level_complete(const DParams &P, uint32_t d) {
    for (;;) {
        if (d + 1u >= P.wf_levels) { __hip_atomic_store(P.q_ctl + QC_FINISHED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        uint32_t *L = q_level(P, d + 1u);
        // every append of level d's packets was performed before the "done" that led here (the appending lane waited for its
        // atomics before counting its packet): the count is final, and so is the last packet's ready word
        const uint32_t c = q_load(L + QC_COUNT);
        const uint32_t npk = (c + 63u) >> 6;
        if (c & 63u) atomicOr(q_ready(P, d + 1u) + (c >> 6), QR_LAST);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(L + QC_STATE), (unsigned long long)(npk + 1u) << 32);
        if ((uint32_t)old != npk) return; // packets of level d + 1 still out: the last of them will find the count final
        ++d;
    }
}
// `n` packets of level d have been taken through: count them; the one call that completes the level publishes the next
__device__ __forceinline__ void q_packets_done(const DParams &P, uint32_t d, uint32_t n) {
    const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(q_level(P, d) + QC_STATE), (unsigned long long)n);
    const uint32_t done = (uint32_t)old + n, target1 = (uint32_t)(old >> 32);
    const bool complete = d == 0u ? done == P.ntiles : (target1 != 0u && done == target1 - 1u);
    if (complete) q_level_complete(P, d);
}

template <bool LDSS, bool PRUNE>
__global__ void __launch_bounds__(LDSS ? LG_LDSS_BLOCK : LG_BLOCK, LG_WAVES_PER_SIMD) queue_kernel(const DParams P) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const unsigned long long gtid = (unsigned long long)blockIdx.x * blockDim.x + tid;
    uint32_t *stack = lds_stack + tid;
    constexpr uint32_t stride = LDSS ? LG_LDSS_BLOCK : LG_BLOCK;
    const uint4 *scn = nullptr;
    if (LDSS) {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_stack + P.stack_depth * stride);
        const uint4 *src = reinterpret_cast<const uint4 *>(P.lds_image);
        for (uint32_t i = tid; i < P.lds_image_n16; i += stride) dst[i] = src[i];
        __syncthreads(); // the only workgroup-wide step; every wave reaches it before pulling packets
        scn = dst;
    }
    Counters cnt = {0, 0, 0, 0, 0, 0, 0, 0, 0}; (void)cnt;
    uint32_t *const ctl = P.q_ctl;
    const uint32_t levels = P.wf_levels;
    uint32_t backoff = 1u, idle = 0u;
    bool tiles_open = true;  // level 0 still has tiles (as far as this wave knows)
    uint32_t done0 = 0u;     // level-0 packets this wave has taken through and not yet counted
    // Packets of the levels >= 1 are handed out by TICKET: a wave holds at most one ticket per level -- a fetch-add on the level's
    // claim counter gives it packet number t, for good -- and looks at that packet's own ready word whenever it wants work; packet t
    // is taken through by the holder of ticket t and nobody else.  (A compare-and-swap on "the next ready packet" lets one wave
    // through per round trip and sends the losers away: 270k deep packets of a glass torus then cost a microsecond EACH, serially.
    // Tickets are one atomic per packet on the shared word and the polls go to words no other wave reads.)  A ticket beyond the
    // level's last packet is void; its holder learns that from the level's final packet count when it runs out of other work.
    constexpr uint32_t NO_TICKET = 0xFFFFFFFFu;
    uint32_t tk[QC_MAX_LEVELS];
#pragma unroll
    for (uint32_t d = 0u; d < QC_MAX_LEVELS; ++d) tk[d] = NO_TICKET;
    uint32_t closed = 0u; // bit d: level d has handed out its last packet: no more tickets
    for (;;) {
        // ---- claim a packet: the deepest level whose ticket has come up (so that the expensive secondary rays start early and the
        // launch ends with one short tail), else the next pixel tile
        uint32_t lvl = NO_TILE, pkt = 0u, nrays = 64u;
        if (lane == 0u) {
            if (levels > 1u) {
                uint32_t rr[QC_MAX_LEVELS];
#pragma unroll
                for (uint32_t d = 1u; d < QC_MAX_LEVELS; ++d)
                    if (d < levels && tk[d] == NO_TICKET && !((closed >> d) & 1u)) tk[d] = atomicAdd(q_level(P, d) + QC_CLAIMED, 1u);
#pragma unroll
                for (uint32_t d = 1u; d < QC_MAX_LEVELS; ++d) {
                    const uint32_t cap_pk = (uint32_t)(P.n_items >> 6) << d; // a ticket beyond the level's capacity looks at the spare word behind it (never raised)
                    rr[d] = (d < levels && tk[d] != NO_TICKET) ? q_load(q_ready(P, d) + (tk[d] < cap_pk ? tk[d] : cap_pk)) : 0u;
                }
#pragma unroll
                for (uint32_t d = QC_MAX_LEVELS - 1u; d >= 1u; --d) {
                    if (lvl != NO_TILE || d >= levels) continue;
                    const uint32_t r = rr[d];
                    if (!(r == 64u || (r & QR_LAST) != 0u)) continue; // (its rays are still being written, or it does not exist yet)
                    lvl = d; pkt = tk[d]; nrays = r & 0xFFu;
                    tk[d] = NO_TICKET;
                }
            }
            if (lvl == NO_TILE && tiles_open) {
                const uint32_t k = atomicAdd(q_level(P, 0u) + QC_CLAIMED, 1u);
                if (k < P.ntiles) { lvl = 0u; pkt = k; }
                else tiles_open = false;
            }
            if (lvl != 0u && done0 != 0u) { q_packets_done(P, 0u, done0); done0 = 0u; } // (this wave leaves level 0, for now or for good)
            if (lvl == NO_TILE) { // nothing to do right now: are the tickets still good?  is everything done?
#pragma unroll
                for (uint32_t d = 1u; d < QC_MAX_LEVELS; ++d) {
                    if (d >= levels || tk[d] == NO_TICKET) continue;
                    const uint32_t target1 = q_load(q_level(P, d) + QC_STATE + 1u); // high half: packets of the level + 1 once its count is final
                    if (target1 != 0u && tk[d] >= target1 - 1u) { tk[d] = NO_TICKET; closed |= 1u << d; }
                }
                if (q_load(ctl + QC_FINISHED) != 0u) lvl = Q_EXIT;
            }
        }
        lvl = (uint32_t)__builtin_amdgcn_readfirstlane((int)lvl);
        if (lvl == Q_EXIT) break; // every wave reaches this exit (or the poll limit below)
        if (lvl == NO_TILE) {
            for (uint32_t s = 0; s < backoff; ++s) __builtin_amdgcn_s_sleep(127);
            backoff = backoff < 8u ? backoff * 2u : 8u;
            if (++idle > Q_POLL_LIMIT) {
                if (lane == 0u) __hip_atomic_store(ctl + QC_ERROR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            continue;
        }
        backoff = 1u; idle = 0u;
        pkt = (uint32_t)__builtin_amdgcn_readfirstlane((int)pkt);
        nrays = (uint32_t)__builtin_amdgcn_readfirstlane((int)nrays);
        const uint32_t d = lvl;
        const unsigned long long cap = P.n_items << d; // SoA stride of level d's arrays
        const unsigned long long i = (unsigned long long)pkt * 64ull + lane; // this lane's ray of level d

        // ---- the ray
        Pixel px;
        px.active = false; px.x = 0u; px.y = 0u; px.pix = 0ull;
        Ray ray = ray_new(V3{0.0, 0.0, 0.0}, V3{0.0, 0.0, 1.0});
        bool valid;
        if (d == 0u) {
            px = pixel_of(P, P.tile0 + pkt, lane);
            valid = px.active;
            if (valid) ray = camera_ray(P, px.x, px.y, P.sample_index);
        } else {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // the rays were written by other waves, on any CU / XCD
            valid = lane < nrays;
            if (valid) {
                const double *q = P.q_rays[d] + i;
                ray = ray_new(V3{q[0 * cap], q[1 * cap], q[2 * cap]}, V3{q[3 * cap], q[4 * cap], q[5 * cap]}); // Ray3::new (ray.rs:28-33)
            }
        }

        // ---- li() of the packet (integrate.rs:23-80): job 0 = closest hit, job 1 + l = any-hit towards light l (one call site)
        bool hit = false;
        uint32_t vis = 0u;
        V3 hit_p = vzero();
        Shade sh;
        sh.mat = 0;
        for (uint32_t job = 0u; job <= P.nlights; ++job) {
            const bool shadow = job != 0u;
            if (shadow && !wave_any(hit)) break;
            Ray tray = ray;
            if (shadow && hit) {
                const DLight L = P.lights[job - 1u];
                tray = ray_new(hit_p, V3{L.pos[0], L.pos[1], L.pos[2]} - hit_p); // point.rs:43-44
            }
            Best b;
            b.ref = NO_HIT; b.t = INFINITY; b.accel = 0u;
            if (shadow ? hit : valid) walk<LDSS, false, PRUNE>(P, tray, shadow, stack, stride, b, scn, cnt);
            if (!shadow) {
                hit = valid && b.ref != NO_HIT;
                if (hit) {
                    shade_frame(P, ray, b, sh); // resolve_hit + SurfaceInteraction::from, after the walk
                    hit_p = sh.p;               // interaction.p + p_err (integrate.rs:40)
                    if (P.nlights > 0u) stash_put(P, gtid, sh); // parked across the shadow walks
                } else if (valid) { // integrate.rs:26-28
                    const V3 value = background(P, normalize(ray.d));
                    if (levels == 1u) finish_pixel(P, px, i, value);
                    else {
                        double *o = P.q_out[d] + i;
                        o[0] = value.x; o[cap] = value.y; o[2 * cap] = value.z;
                        if (d + 1u < levels) P.q_child[d][i] = WF_MISS;
                    }
                }
            } else if (hit && !(b.t < 1.0)) vis |= 1u << (job - 1u); // point.rs:49
        }

        // ---- radiance of the hits, specular children (integrate.rs:47-77, 82-132)
        bool has_r = false, has_t = false;
        Sample sr, st;
        V3 output = vzero();
        if (hit) {
            if (P.nlights > 0u) stash_get(P, gtid, sh, ray);
            const DMaterial m = P.materials[sh.mat];
            output = shade_lights(P, m, sh, vis);
            if (d + 1u < levels && (m.kind == MAT_GLASS || m.kind == MAT_MIRROR)) { // depth < max recursion (integrate.rs:69-77)
                if (sample_specular_transmission(m, sh, st))
                    has_t = !(st.pdf <= 0.0 || veq(st.spectrum, vzero()) || fabs(dot(st.wi, sh.ns)) == 0.0);
                if (sample_specular_reflection(m, sh, sr))
                    has_r = !(sr.pdf <= 0.0 || veq(sr.spectrum, vzero()) || dot(sr.wi, sh.ns) <= 0.0);
            }
            if (levels == 1u) finish_pixel(P, px, i, output + vzero() + vzero()); // integrate.rs:79 with no children
            else if (d + 1u >= levels) {
                const V3 value = output + vzero() + vzero();
                double *o = P.q_out[d] + i;
                o[0] = value.x; o[cap] = value.y; o[2 * cap] = value.z;
            }
        }
        if (d + 1u < levels) {
            // children: one reservation per wave in the next level's queue, reflected rays first
            const unsigned long long mr = __builtin_amdgcn_ballot_w64(has_r), mt = __builtin_amdgcn_ballot_w64(has_t);
            const uint32_t nr = (uint32_t)__builtin_popcountll(mr), nt = (uint32_t)__builtin_popcountll(mt);
            uint32_t base = 0u;
            uint32_t *LN = q_level(P, d + 1u);
            if (nr + nt != 0u) {
                if (lane == 0u) base = atomicAdd(LN + QC_COUNT, nr + nt);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            }
            const uint32_t cr = base + q_lanes_below(mr), ct = base + nr + q_lanes_below(mt);
            if (hit) {
                const unsigned long long nn = cap << 1;
                double *o = P.q_out[d] + i;
                o[0] = output.x; o[cap] = output.y; o[2 * cap] = output.z;
                P.q_child[d][i] = has_r ? cr : WF_NONE;
                P.q_child[d][cap + i] = has_t ? ct : WF_NONE;
                double *sp = P.q_spec[d] + i;
                if (has_r) {
                    sp[0 * cap] = sr.spectrum.x; sp[1 * cap] = sr.spectrum.y; sp[2 * cap] = sr.spectrum.z;
                    const V3 wr = -1.0 * sh.wo + 2.0 * dot(sh.wo, sh.ns) * sh.ns; // bxdf::util::reflect (integrate.rs:100)
                    double *q = P.q_rays[d + 1u] + cr;
                    q_store_wt(q + 0 * nn, sh.p.x); q_store_wt(q + 1 * nn, sh.p.y); q_store_wt(q + 2 * nn, sh.p.z);
                    q_store_wt(q + 3 * nn, wr.x); q_store_wt(q + 4 * nn, wr.y); q_store_wt(q + 5 * nn, wr.z);
                }
                if (has_t) {
                    sp[3 * cap] = st.spectrum.x; sp[4 * cap] = st.spectrum.y; sp[5 * cap] = st.spectrum.z;
                    sp[6 * cap] = fabs(dot(st.wi, sh.ns)); sp[7 * cap] = st.pdf;
                    double *q = P.q_rays[d + 1u] + ct;
                    q_store_wt(q + 0 * nn, sh.pm.x); q_store_wt(q + 1 * nn, sh.pm.y); q_store_wt(q + 2 * nn, sh.pm.z);
                    q_store_wt(q + 3 * nn, st.wi.x); q_store_wt(q + 4 * nn, st.wi.y); q_store_wt(q + 5 * nn, st.wi.z);
                }
            }
            if (nr + nt != 0u) {
                // publish: the write-through stores above have left this wave, then the ready words of the packets the reservation
                // [base, base + nr + nt) touches (at most three) are raised by what it put into each; the packet is counted only
                // after those atomics have been performed (the count must never overtake them)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0u) {
                    uint32_t *rd = q_ready(P, d + 1u);
                    uint32_t lo = base;
                    const uint32_t end = base + nr + nt;
                    while (lo < end) {
                        const uint32_t p = lo >> 6, hi = (p + 1u) << 6 < end ? (p + 1u) << 6 : end;
                        atomicAdd(rd + p, hi - lo);
                        lo = hi;
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        // ---- the packet is through
        if (lane == 0u) {
            if (d == 0u) ++done0; // counted when this wave next leaves level 0
            else q_packets_done(P, d, 1u);
        }
    }
}

// ---- host-callable launchers (used by capi.cpp)
hipError_t launch_queue(const DParams &P, uint32_t blocks, hipStream_t stream) {
    if (P.lds_image) { // LDS-resident scene: `blocks` = one 1024-lane workgroup per CU
        const size_t lds = (size_t)P.stack_depth * LG_LDSS_BLOCK * sizeof(uint32_t) + (size_t)P.lds_image_n16 * 16u;
        if (P.prune) hipLaunchKernelGGL((queue_kernel<true, true>), dim3(blocks), dim3(LG_LDSS_BLOCK), lds, stream, P);
        else hipLaunchKernelGGL((queue_kernel<true, false>), dim3(blocks), dim3(LG_LDSS_BLOCK), lds, stream, P);
        return hipGetLastError();
    }
    const size_t lds = (size_t)P.stack_depth * LG_BLOCK * sizeof(uint32_t);
    if (P.prune) hipLaunchKernelGGL((queue_kernel<false, true>), dim3(blocks), dim3(LG_BLOCK), lds, stream, P);
    else hipLaunchKernelGGL((queue_kernel<false, false>), dim3(blocks), dim3(LG_BLOCK), lds, stream, P);
    return hipGetLastError();
}
hipError_t queue_occupancy(uint32_t stack_depth, int *blocks_per_cu) {
    const size_t lds = (size_t)stack_depth * LG_BLOCK * sizeof(uint32_t);
    int a = 0, b = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, queue_kernel<false, false>, LG_BLOCK, lds);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, queue_kernel<false, true>, LG_BLOCK, lds);
    *blocks_per_cu = a < b ? a : b;
    return e;
}
hipError_t queue_set_lds_limit(size_t bytes, bool ldss) {
    const void *ldss_fns[] = {reinterpret_cast<const void *>(queue_kernel<true, false>), reinterpret_cast<const void *>(queue_kernel<true, true>)};
    const void *plain_fns[] = {reinterpret_cast<const void *>(queue_kernel<false, false>), reinterpret_cast<const void *>(queue_kernel<false, true>)};
    const void *const *fns = ldss ? ldss_fns : plain_fns;
    for (size_t i = 0; i < 2; ++i) {
        hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

} // namespace lg
