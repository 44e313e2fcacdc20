// lasgun_amd/csrc/k_queue.hip -- the queue organisation: li() for every recursion level of a chunk of the film in ONE persistent launch.
#define LG_DIR_PER_RECORD 1 // (walk.h, chunk_culled: this kernel keeps the per-record form)
#include "shade.h"

namespace lg {

// ------------------------------------------------------------------------------------------
// Why a third organisation.  On scenes whose rays are long, uneven walks -- a 100k-triangle mesh in the
// reference's 254-triangle leaves -- a wave is busy for a millisecond or more with one 8x8 tile.  The megakernel gives every lane
// its pixel's whole ray tree (integrate.rs:23-132): deep levels run with the few lanes of the tile that hit glass, the recursion
// state lives in 300-500 spilled registers.  The level-by-level pipeline packs every level's rays into full waves, but each of its
// ~15 launches per chunk ends with a tail as long as its slowest wave, and a queue filled by whoever appends next packs rays from
// all over the film into one wave: incoherent, and the walk's wave-uniform phases then serialise (measured, round 4: such packets
// cost 1.4 x the megakernel's sparse ones).  Here ONE launch does it: its waves pull work from per-level queues, deepest level
// first -- level 0's work items are UNITS of a few neighbouring 8x8 pixel tiles, deeper levels' are 64-ray PACKETS -- and
//   * a unit / packet is taken through closest hit -> shading frame -> per-light any-hit -> radiance -> children by the wave that
//     claimed it, with ONE traversal call site in a wave-uniform job loop: what is live across a walk is a ray and a few words;
//   * the specular children of a unit are COMPACTED BY THE WAVE ITSELF (ballot + prefix, as in wf_shade_kernel) into packets it
//     owns until they are published: reflected and refracted rays apart, neighbours together -- full waves of coherent rays
//     wherever a 32 x 8 pixel strip is mostly glass; a deeper packet's children go out as that packet's own one or two packets;
//   * every published packet is complete (its ready word carries its ray count), so no wave ever waits for another's appends.
// The levels are then combined bottom-up by wf_combine_kernel (integrate.rs:79, 103, 129), exactly as in the level-by-level
// pipeline: same arrays, same order of operations -- every f64 comes from the same expression in all three organisations.
//
// Scheduling state (DParams::q_ctl, QC_*; every hot word on a 128-byte line of its own): per level a packet count (reservations), a
// ticket counter and one 64-bit word (packets + 1) << 32 | packets done; per packet of the levels >= 1 a ready word, QR_LAST | rays
// once its rays are written.  Hand-off of the ray data between waves on different CUs / XCDs (MI355X_MICROARCH.md,
// inter-workgroup visibility): the producer stores the rays WRITE-THROUGH (agent-scope relaxed atomic stores: global_store sc1;
// a release fence would write back the XCD's whole dirty L2 -- megabytes of parked frames and results -- per packet), waits for
// them (s_waitcnt vmcnt(0)), then stores the ready word the same way; the consumer polls with relaxed agent loads, runs an
// agent-scope acquire and reads the rays.  Exactly one atomic operation on a level's 64-bit word observes "count final and every
// packet done"; that wave publishes the next level's packet count.  Level 0's units are counted per wave and flushed when the
// wave turns to a deeper level or finds the units exhausted: one atomic per unit (the claim).  Every wave leaves through
// QC_FINISHED (or, should the protocol ever stall, through the poll limit with DParams::q_err set -- reported by the host, never silent).
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

// a level is complete: the next level's packet count is final; cascades through empty levels; the last level sets QC_FINISHED
__device__ __forceinline__ void q_level_complete(const DParams &P, uint32_t d) {
    for (;;) {
        if (d + 1u >= P.wf_levels) { __hip_atomic_store(P.q_ctl + QC_FINISHED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        uint32_t *L = q_level(P, d + 1u);
        // every reservation of level d's work items was performed before the "done" that led here (a reservation returns its
        // packet number to the lane that then counts the item): the count is final
        const uint32_t npk = q_load(L + QC_COUNT);
        const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(L + QC_STATE), (unsigned long long)(npk + 1u) << 32);
        if ((uint32_t)old != npk) return; // packets of level d + 1 still out: the last of them will find the count final
        ++d;
    }
}
// `n` work items of level d have been taken through: count them; the one call that completes the level publishes the next
__device__ __forceinline__ void q_items_done(const DParams &P, uint32_t d, uint32_t n) {
    const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(q_level(P, d) + QC_STATE), (unsigned long long)n);
    const uint32_t done = (uint32_t)old + n, target1 = (uint32_t)(old >> 32);
    const bool complete = d == 0u ? done == P.q_units : (target1 != 0u && done == target1 - 1u);
    if (complete) q_level_complete(P, d);
}

// Level 0's tile sequence.  Which tiles are in flight together decides what the L2s must hold: 4096 waves on 4096 consecutive tiles
// in row order cover a band of the film 64 pixels high and as wide as the film -- across the whole mesh -- and every XCD's L2 sees
// all of it.  q_order 1: the film in blocks of 32 x 32 tiles (256 x 256 pixels), Morton order inside a block, and the sequence in
// eight contiguous bands, one per XCD (HW_REG_XCC_ID; a wave whose band is exhausted moves on to the next for good): the 512 waves
// of an XCD then work on half a block, a compact patch of the scene that its own 4 MiB L2 keeps.  Where a tile is rendered never
// changes what is rendered.
__device__ __forceinline__ uint32_t q_compact_bits(uint32_t x) { // bits 0, 2, 4, ... of x, packed
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    return (x | (x >> 8)) & 0x0000FFFFu;
}
__device__ __forceinline__ uint32_t q_seq_tile(const DParams &P, uint32_t s) { // sequence index -> tile of the chunk (NO_TILE: a hole of the block grid)
    if (P.q_order == 0u) return s < P.ntiles ? tile_in_order(P, s) : NO_TILE; // (tile_rev: from the last tile down, or from the middle outwards)
    const uint32_t block = s >> 10, w = s & 1023u;
    const uint32_t tx = (block % P.q_blocks_x) * 32u + q_compact_bits(w), ty = (block / P.q_blocks_x) * 32u + q_compact_bits(w >> 1);
    return tx < P.tiles_x && ty < P.q_tiles_y ? ty * P.tiles_x + tx : NO_TILE;
}
// the next unit of level 0 for this wave (lane 0 only); `band` / `left`: the wave's place among the XCD bands
__device__ __forceinline__ uint32_t q_claim_unit(const DParams &P, uint32_t &band, uint32_t &left) {
    if (P.q_order == 0u) {
        if (left == 0u) return NO_TILE;
        const uint32_t k = atomicAdd(q_level(P, 0u) + QC_CLAIMED, 1u);
        if (k < P.q_units) return k;
        left = 0u;
        return NO_TILE;
    }
    while (left != 0u) {
        const uint32_t lo = (uint32_t)(((unsigned long long)band * P.q_units) / TILE_HEADS), hi = (uint32_t)(((unsigned long long)(band + 1u) * P.q_units) / TILE_HEADS);
        const uint32_t t = atomicAdd(P.q_ctl + QC_HEADS + band * 32u, 1u);
        if (t < hi - lo) return lo + t;
        band = (band + 1u) & (TILE_HEADS - 1u); // this band is done (for every wave: its head only grows)
        --left;
    }
    return NO_TILE;
}

// A packet of the next level that this wave is filling: it owns the slot from the reservation to the publication.
struct OpenPacket {
    uint32_t pkt;  // its number in the level (NO_TILE: none open)
    uint32_t fill; // rays placed so far
};
__device__ __forceinline__ uint32_t q_reserve(const DParams &P, uint32_t level, uint32_t lane) {
    uint32_t k = 0u;
    if (lane == 0u) k = atomicAdd(q_level(P, level) + QC_COUNT, 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
}
// Slots in `level` for the lanes of `mask` (at most 64): the open packet first, a newly reserved one for what does not fit.
// Returns this lane's ray index in the level (meaningful for the lanes of `mask`); `full` = the packet that this call filled up
// (NO_TILE: none) -- the caller publishes it once the rays are stored.
__device__ __forceinline__ uint32_t q_place(const DParams &P, uint32_t level, OpenPacket &op, unsigned long long mask, uint32_t lane, uint32_t &full) {
    const uint32_t n = (uint32_t)__builtin_popcountll(mask);
    full = NO_TILE;
    if (n == 0u) return 0u;
    if (op.pkt == NO_TILE) { op.pkt = q_reserve(P, level, lane); op.fill = 0u; }
    const uint32_t room = 64u - op.fill, rank = q_lanes_below(mask);
    uint32_t idx = op.pkt * 64u + op.fill + rank;
    if (n < room) op.fill += n;
    else {
        full = op.pkt;
        if (n > room) {
            const uint32_t k = q_reserve(P, level, lane);
            if (rank >= room) idx = k * 64u + (rank - room);
            op.pkt = k; op.fill = n - room;
        } else { op.pkt = NO_TILE; op.fill = 0u; }
    }
    return idx;
}
// the rays of packet `pkt` of `level` are written and have left the wave (the caller waited for them): raise its ready word
__device__ __forceinline__ void q_publish(const DParams &P, uint32_t level, uint32_t pkt, uint32_t rays, uint32_t lane) {
    if (pkt != NO_TILE && rays != 0u && lane == 0u)
        __hip_atomic_store(q_ready(P, level) + pkt, QR_LAST | rays, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        copy_to_lds(dst, src, P.lds_image_n16, tid, stride);
        __syncthreads(); // the only workgroup-wide step; every wave reaches it before pulling work
        scn = dst;
    }
    const uint4 *const arec = LDSS ? nullptr : load_accel_image(P, P.stack_depth * LG_BLOCK);
    Counters cnt = {0, 0, 0, 0, 0, 0, 0, 0, 0}; (void)cnt;
    uint32_t *const ctl = P.q_ctl;
    const uint32_t levels = P.wf_levels;
    // (a small film: level 0 has q_units units and a unit's family has at most 2^d packets in flight at level d -- more waves than
    // that find nothing to do, yet would take tickets and poll until the launch ends; they leave before touching a control word)
    if (!wave_has_work((unsigned long long)P.q_units << (levels - 1u))) return;
    uint32_t backoff = 1u, idle = 0u;
    uint32_t band = xcc_id(), bands_left = TILE_HEADS; // level 0 still has units while bands_left != 0 (as far as this wave knows)
    uint32_t done0 = 0u;     // level-0 units this wave has taken through and not yet counted
    // Packets of the levels >= 1 are handed out by TICKET: a wave holds at most one ticket per level -- a fetch-add on the level's
    // ticket counter gives it packet number t, for good -- and looks at that packet's own ready word whenever it wants work; packet t
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
        // ---- claim work: the deepest level whose ticket has come up (so that the expensive secondary rays start early and the
        // launch ends with one short tail), else the next unit of pixel tiles
        uint32_t lvl = NO_TILE, item = 0u, nrays = 64u;
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
                    if ((r & QR_LAST) == 0u) continue; // (not published yet, or it does not exist)
                    lvl = d; item = tk[d]; nrays = r & 0xFFu;
                    tk[d] = NO_TICKET;
                }
            }
            if (lvl == NO_TILE && bands_left != 0u) {
                const uint32_t k = q_claim_unit(P, band, bands_left);
                if (k != NO_TILE) { lvl = 0u; item = k; }
            }
            if (lvl != 0u && done0 != 0u) { q_items_done(P, 0u, done0); done0 = 0u; } // (this wave leaves level 0, for now or for good)
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
#ifdef LG_QIDLE // diagnostic build (tools/queue_idle.py): the time this wave spends with nothing to claim, in 100 MHz ticks
            const unsigned long long idle_t0 = wall_clock64();
#endif
            for (uint32_t s = 0; s < backoff; ++s) __builtin_amdgcn_s_sleep(127);
#ifdef LG_QIDLE
            if (lane == 0u) atomicAdd(reinterpret_cast<unsigned long long *>(ctl + QC_ERROR), wall_clock64() - idle_t0);
#endif
            backoff = backoff < 8u ? backoff * 2u : 8u;
            if (++idle > Q_POLL_LIMIT) {
                if (lane == 0u && P.q_err) __hip_atomic_store(P.q_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // sticky, in host memory: no later launch clears it
                break;
            }
            continue;
        }
        backoff = 1u; idle = 0u;
        item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
        nrays = (uint32_t)__builtin_amdgcn_readfirstlane((int)nrays);
        const uint32_t d = lvl;
        const unsigned long long cap = P.n_items << d; // SoA stride of level d's arrays
        // level 0: the unit's tiles one after the other; deeper: the one packet
        const uint32_t first = d == 0u ? item * P.q_unit_tiles : item;
        const uint32_t last = d == 0u ? (first + P.q_unit_tiles < P.q_seq_len ? first + P.q_unit_tiles : P.q_seq_len) : item + 1u;
        OpenPacket op_r{NO_TILE, 0u}, op_t{NO_TILE, 0u}; // the next level's packets this wave is filling: reflected / refracted children
        if (d != 0u) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // the rays were written by another wave, on any CU / XCD

        for (uint32_t seq = first; seq < last; ++seq) {
            const uint32_t pkt = d == 0u ? q_seq_tile(P, seq) : seq; // level 0: the tile behind this place of the sequence
            if (pkt == NO_TILE) continue;
            const unsigned long long i = (unsigned long long)pkt * 64ull + lane; // this lane's ray of level d
            // ---- the ray
            Pixel px;
            px.active = false; px.x = 0u; px.y = 0u; px.pix = 0ull;
            Ray ray = ray_new(V3{0.0, 0.0, 0.0}, V3{0.0, 0.0, 1.0});
            bool valid;
            if (d == 0u) {
                uint32_t sample; // (samples side by side: a level-0 tile is a pixel tile at ONE of its samples, DParams::ss_par; a small launch's tiles
                // in parts: 64 >> split_shift of its lanes, DParams::split_shift -- the children's packets are then as narrow as their parents)
                px = pixel_of(P, P.tile0 + l0_tile(P, pkt >> P.split_shift, sample), lane);
                valid = px.active && (lane >> (6u - P.split_shift)) == (pkt & ((1u << P.split_shift) - 1u));
                if (valid) ray = camera_ray(P, px.x, px.y, sample);
            } else {
                valid = lane < nrays;
                if (valid) {
                    const double *q = P.q_rays[d] + i;
                    ray = ray_new(V3{q[0 * cap], q[1 * cap], q[2 * cap]}, V3{q[3 * cap], q[4 * cap], q[5 * cap]}); // Ray3::new (ray.rs:28-33)
                } else if (d + 1u < levels) P.q_child[d][i] = WF_MISS; // a slot past the packet's rays: nothing for the combine pass to follow
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
                if (shadow ? hit : valid) walk<LDSS, false, PRUNE>(P, tray, shadow, stack, stride, b, scn, cnt, arec);
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
                // children into the packets this wave is filling at the next level: reflected and refracted rays apart
                const unsigned long long mr = __builtin_amdgcn_ballot_w64(has_r), mt = __builtin_amdgcn_ballot_w64(has_t);
                uint32_t full_r, full_t;
                const uint32_t cr = q_place(P, d + 1u, op_r, mr, lane, full_r), ct = q_place(P, d + 1u, op_t, mt, lane, full_t);
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
                if (full_r != NO_TILE || full_t != NO_TILE) { // a packet filled up: its rays have left the wave, then its ready word
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    q_publish(P, d + 1u, full_r, 64u, lane);
                    q_publish(P, d + 1u, full_t, 64u, lane);
                }
            }
        }
        // ---- the unit / packet is through: what is left in the open packets goes out as it is, then the item is counted (the
        // reservations -- atomics that returned their packet numbers -- came first)
        if (d + 1u < levels && (op_r.fill != 0u || op_t.fill != 0u)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            q_publish(P, d + 1u, op_r.pkt, op_r.fill, lane);
            q_publish(P, d + 1u, op_t.pkt, op_t.fill, lane);
        }
        if (lane == 0u) {
            if (d == 0u) ++done0; // counted when this wave next leaves level 0
            else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the ready words first)
                q_items_done(P, d, 1u);
            }
        }
    }
}

// ---- host-callable launchers (used by launch.cpp, capi.cpp)
hipError_t launch_queue(const DParams &P, uint32_t blocks, hipStream_t stream) {
    if (P.lds_image) { // LDS-resident scene: `blocks` = one 1024-lane workgroup per CU
        const size_t lds = (size_t)P.stack_depth * LG_LDSS_BLOCK * sizeof(uint32_t) + (size_t)P.lds_image_n16 * 16u;
        if (P.prune) hipLaunchKernelGGL((queue_kernel<true, true>), dim3(blocks), dim3(LG_LDSS_BLOCK), lds, stream, P);
        else hipLaunchKernelGGL((queue_kernel<true, false>), dim3(blocks), dim3(LG_LDSS_BLOCK), lds, stream, P);
        return hipGetLastError();
    }
    const size_t lds = (size_t)P.stack_depth * LG_BLOCK * sizeof(uint32_t) + (P.accel_image ? (size_t)P.accel_image_n16 * 16u : 0u);
    if (P.prune) hipLaunchKernelGGL((queue_kernel<false, true>), dim3(blocks), dim3(LG_BLOCK), lds, stream, P);
    else hipLaunchKernelGGL((queue_kernel<false, false>), dim3(blocks), dim3(LG_BLOCK), lds, stream, P);
    return hipGetLastError();
}
hipError_t queue_occupancy(uint32_t stack_depth, size_t extra_lds, int *blocks_per_cu) {
    const size_t lds = (size_t)stack_depth * LG_BLOCK * sizeof(uint32_t) + extra_lds;
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
