// lasgun_amd/csrc/k_wavefront.hip -- the wavefront pipeline: li() level by level (closest / shadow / shade / combine).
#include <cstdlib>

#include "shade.h"

namespace lg {

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
// sizes the chunk of the film to its memory budget (launch.cpp).
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
    if (level == 0u) return (unsigned long long)P.ntiles * 64ull;
    return P.q_ctl ? (unsigned long long)P.q_ctl[QC_LEVEL0 + QC_LEVEL_WORDS * level + QC_COUNT] * 64ull : P.wf_counts[level]; // (the queue organisation counts 64-ray packets)
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

// (An experiment of round 6, kept as an opt-in for its measurement -- LASGUN_REFILL=1; see launch_wf_trace for what it showed.)
// The shadow pass as ONE persistent walk per wave (walk.h, REFILL): a lane whose shadow ray is done takes the next hit of the wave's current
// 64-slot tile (a new tile is claimed when that one is used up) instead of waiting for the slowest of 64 rays -- the any-hit walks of a
// tile end anywhere between the first box and the whole tree.  Lanes refill when at least LG_REFILL_MIN of them are done: a refill is
// the frame loads and the ray set-up (three divisions) for however many lanes take part, so it wants company.  Which lane walks which
// hit never changes a visibility bit.
#ifndef LG_REFILL_MIN
#define LG_REFILL_MIN 16
#endif
struct ShadowRefill {
    static constexpr bool enabled = true;
    static constexpr unsigned long long NONE = ~0ull;
    const DParams &P;
    const HitSlots &hs;
    uint32_t band, left;             // the wave's place among the tile heads (claim_tile_partial)
    uint32_t tile = NO_TILE, slot = 64u; // the tile being handed out and its next slot (wave-uniform)
    bool exhausted = false;          // no tile is left (wave-uniform)
    unsigned long long h = NONE;     // this lane's hit, its light, the bits so far, the point the shadow rays leave from
    uint32_t light = 0u, vis = 0u;
    V3 hit_p{0.0, 0.0, 0.0};
    __device__ __forceinline__ ShadowRefill(const DParams &p, const HitSlots &s, uint32_t b) : P(p), hs(s), band(b), left(TILE_HEADS) {}
    __device__ __forceinline__ uint32_t threshold() const { return exhausted ? (P.nlights > 1u ? 1u : 65u) : (uint32_t)LG_REFILL_MIN; }
    __device__ __forceinline__ Ray shadow_ray() const {
        const DLight L = P.lights[light];
        return ray_new(hit_p, V3{L.pos[0], L.pos[1], L.pos[2]} - hit_p); // point.rs:43-44
    }
    __device__ __forceinline__ void deliver(const Best &b) {
        if (h == NONE) return;
        if (!(b.t < 1.0)) vis |= 1u << light; // point.rs:49
    }
    __device__ __forceinline__ void finish(const Best &b) { // after the walk: a lane's last result
        if (h == NONE) return;
        deliver(b);
        P.vis[h] = vis;
        h = NONE;
    }
    // called by EVERY lane of the wave, in wave-uniform control flow (tile, slot, band, left, exhausted are the wave's: a lane that sat a call
    // out would keep stale copies); `done`: this lane's walk is done (or it has no ray yet).  true = `nw` is this lane's next ray
    __device__ __forceinline__ bool next(const bool done, const Best &b, Ray &nw) {
        bool more_lights = false;
        if (done && h != NONE) {
            deliver(b);
            if (++light < P.nlights) more_lights = true;
            else { P.vis[h] = vis; h = NONE; }
        }
        bool need = done && h == NONE;
        const bool wanted = need;
        while (!exhausted && wave_any(need)) {
            if (slot >= 64u) { // (wave-uniform)
                tile = claim_tile_partial(P.tile_counter, hs.tiles, band, left);
                if (tile == NO_TILE) { exhausted = true; break; }
                slot = 0u;
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(need);
            const uint32_t want = (uint32_t)__builtin_popcountll(m), room = 64u - slot, take = want < room ? want : room;
            const uint32_t rank = lanes_below(m);
            if (need && rank < take) {
                unsigned long long hh;
                if (hit_of(P, hs, tile, slot + rank, hh)) { // (a hole of a dense block, a slot past the count: the lane asks again)
                    h = hh;
                    const unsigned long long n = P.wf_hit_stride;
                    // interaction.p + p_err, recomputed from the parked frame exactly as stash_get does
                    const V3 praw{P.frame[0 * n + h], P.frame[1 * n + h], P.frame[2 * n + h]};
                    const V3 ng{P.frame[3 * n + h], P.frame[4 * n + h], P.frame[5 * n + h]};
                    const double err = 2.220446049250313e-16 * 65536.0;
                    hit_p = praw + ng * err;
                    light = 0u; vis = 0u;
                    need = false;
                }
            }
            slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)(slot + take));
        }
        const bool got = more_lights || (wanted && !need); // the next light of the same hit, or a hit taken just now
        if (got) nw = shadow_ray();
        return got;
    }
};

// W1 / W2: persistent traversal kernels of the wavefront pipeline (tile counter, per-lane LDS stack; LDSS as above).
// L0: the launch is level 0's (rays from the camera, work items = the chunk's pixels in 8x8 tiles).
template <bool FAST, bool SHADOW, bool LDSS, bool L0, bool PRUNE = false, bool REFILL = false>
__global__ void __launch_bounds__(LDSS ? LG_LDSS_BLOCK : LG_BLOCK, LG_TRAV_WAVES_PER_SIMD) wf_trace_kernel(const DParams P) {
    static_assert(!REFILL || (SHADOW && LDSS && !FAST), "the refilling walk: the LDS-resident shadow pass");
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
        copy_to_lds(dst, src, P.lds_image_n16, tid, stride);
        __syncthreads(); // the only workgroup-wide step; every wave reaches it before pulling tiles
        scn = dst;
    }
    const uint4 *const arec = (LDSS || FAST) ? nullptr : load_accel_image(P, P.stack_depth * LG_BLOCK);
    Counters cnt = {0, 0, 0, 0, 0, 0, 0, 0, 0}; (void)cnt;
    if (!wave_has_work(ntiles)) return; // (after the workgroup's barrier; a deep level or a small film leaves most of the grid nothing to claim)
    // Tiles are claimed XCD by XCD (kcommon.h, claim_tile) where the scene sits in LDS and the claim itself is what waves queue on:
    // headline frame 3.39 + 3.52 -> 3.10 + 3.19 ms for the two traversal passes.  With the tables in L2 (mesh scenes) the bands
    // measured no better than one head word (config 5: 68.4 vs 69.4 ms), so those forms keep the single word.
    if (REFILL) { // one persistent walk per wave: its lanes take hits until none is left (ShadowRefill above)
        ShadowRefill rf(P, hs, xcc_id());
        Best b;
        b.ref = NO_HIT; b.t = INFINITY; b.accel = 0u;
        Ray first = ray_new(V3{0.0, 0.0, 0.0}, V3{0.0, 0.0, 1.0});
        const bool live = rf.next(true, b, first); // every lane asks for its first hit
        bool tie = false;
        traverse_ref<LDSS, false, PRUNE, false, ShadowRefill>(P, first, true, stack, stride, b, scn, tie, cnt, arec, &rf, live);
        rf.finish(b);
        return;
    }
    uint32_t band = LDSS ? xcc_id() : 0u, bands_left = TILE_HEADS;
    for (bool final = false; !final;) {
        uint32_t tile;
        if (LDSS) tile = claim_tile(P.tile_counter, ntiles, band, bands_left, final);
        else tile = claim_tile_single(P.tile_counter, ntiles, final);
        if (tile == NO_TILE) break; // (a wave leaves here, or after one of the launch's last tiles: kcommon.h)
        if (!SHADOW) {
            const unsigned long long i = (unsigned long long)tile * lpt + lane;
            Pixel px;
            px.active = false;
            Ray ray = ray_new(V3{0.0, 0.0, 0.0}, V3{0.0, 0.0, 1.0});
            if (L0) {
                uint32_t sample;
                const uint32_t ptile = l0_tile(P, tile, sample);
                px = pixel_of(P, P.tile0 + ptile, lane);
                if (px.active) ray = camera_ray(P, px.x, px.y, sample);
            } else if (lane < lpt && i < n_work) {
                px.active = true;
                ray = wf_load_ray(P, i);
            }
            const bool active = px.active;
            Best b;
            b.ref = NO_HIT; b.t = INFINITY; b.accel = 0u;
            if (active) {
                bool tie = false;
                walk<LDSS, FAST, PRUNE>(P, ray, false, stack, stride, b, scn, cnt, arec);
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
                walk<LDSS, FAST, PRUNE>(P, sray, true, stack, stride, b, scn, cnt, arec);
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
                uint32_t sample;
                const uint32_t ptile = l0_tile(P, (uint32_t)(j >> 6), sample);
                px = pixel_of(P, P.tile0 + ptile, (uint32_t)(j & 63u));
                ray = camera_ray(P, px.x, px.y, sample);
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
            output = shade_lights(P, m, sh, vis); // integrate.rs:47-67
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
    const unsigned long long n_work = wf_level_rays(P, level);
    const unsigned long long n = P.wf_cap, nn = P.wf_cap_next;
    for (unsigned long long j = (unsigned long long)blockIdx.x * LG_BLOCK + threadIdx.x; j < n_work; j += (unsigned long long)gridDim.x * LG_BLOCK) {
        Pixel px;
        if (level == 0u) {
            uint32_t sample;
            const uint32_t vt = (uint32_t)(j >> 6), l = (uint32_t)(j & 63u);
            if ((l >> (6u - P.split_shift)) != (vt & ((1u << P.split_shift) - 1u))) continue; // (the queue organisation's tiles in parts: this lane belongs to another part)
            px = pixel_of(P, P.tile0 + l0_tile(P, vt >> P.split_shift, sample), l);
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


// W5 (samples side by side, DParams::ss_par): a pixel's samples summed in their order, * weight, quantised (integrate.rs:16-20, img.rs:46-67).
// P.ntiles = the chunk's PIXEL tiles; the parked li() of pixel tile t, sample s, lane l sits at accum[(t * ss_par + s) * 64 + l].
__global__ void __launch_bounds__(LG_BLOCK) wf_resolve_kernel(const DParams P) {
    const unsigned long long n_pix = (unsigned long long)P.ntiles * 64ull;
    const uint32_t S = P.ss_par;
    const double weight = 1. / (double)(P.ss_root * P.ss_root);
    for (unsigned long long q = (unsigned long long)blockIdx.x * LG_BLOCK + threadIdx.x; q < n_pix; q += (unsigned long long)gridDim.x * LG_BLOCK) {
        const uint32_t t = (uint32_t)(q >> 6), l = (uint32_t)(q & 63u);
        const Pixel px = pixel_of(P, P.tile0 + t, l);
        if (!px.active) continue;
        V3 color = vzero();
        for (uint32_t sidx = 0; sidx < S; ++sidx) {
            const unsigned long long i = ((((unsigned long long)t * S + sidx) << P.split_shift) + (l >> (6u - P.split_shift))) * 64ull + l; // (split_shift: the queue organisation's tiles in parts)
            color = color + V3{P.accum[i], P.accum[P.n_items + i], P.accum[2 * P.n_items + i]};
        }
        write_pixel(P, px, color * weight);
    }
}

// ---- host-callable launchers (used by launch.cpp, capi.cpp)
hipError_t launch_wf_resolve(const DParams &P, uint32_t blocks, hipStream_t stream) {
    hipLaunchKernelGGL(wf_resolve_kernel, dim3(blocks), dim3(LG_BLOCK), 0, stream, P);
    return hipGetLastError();
}
hipError_t launch_wf_trace(const DParams &P, bool fast, bool shadow, uint32_t blocks, uint32_t stack_depth, hipStream_t stream) {
    const bool ldss = P.lds_image && !fast; // LDS-resident scene: `blocks` = one workgroup per CU
    const bool l0 = !shadow && P.wf_level == 0u;
    const uint32_t block = ldss ? LG_LDSS_BLOCK : LG_BLOCK;
    const uint32_t depth = fast ? stack_depth : P.stack_depth;
    size_t lds = (size_t)depth * block * sizeof(uint32_t) + (ldss ? (size_t)P.lds_image_n16 * 16u : (!fast && P.accel_image ? (size_t)P.accel_image_n16 * 16u : 0u));
#define LG_LAUNCH(F, S, L, Z) hipLaunchKernelGGL((wf_trace_kernel<F, S, L, Z>), dim3(blocks), dim3(block), lds, stream, P)
#define LG_LAUNCH_PRUNED(S, L, Z) hipLaunchKernelGGL((wf_trace_kernel<false, S, L, Z, true>), dim3(blocks), dim3(block), lds, stream, P)
    if (P.prune && !fast) {
        if (ldss) { if (shadow) LG_LAUNCH_PRUNED(true, true, false); else if (l0) LG_LAUNCH_PRUNED(false, true, true); else LG_LAUNCH_PRUNED(false, true, false); }
        else { if (shadow) LG_LAUNCH_PRUNED(true, false, false); else if (l0) LG_LAUNCH_PRUNED(false, false, true); else LG_LAUNCH_PRUNED(false, false, false); }
    } else
    if (fast) { if (shadow) LG_LAUNCH(true, true, false, false); else if (l0) LG_LAUNCH(true, false, false, true); else LG_LAUNCH(true, false, false, false); }
    else if (ldss) {
        // the refilling shadow pass (ShadowRefill): level 0 of a big launch (every wave of the grid has several tiles to go through: the final-tile
        // rules of small launches do not apply to it).  MEASURED and OFF unless LASGUN_REFILL=1 (round 6, profiles/r06_ab_refill*): same film, but the
        // headline's shadow pass issues 2.65e9 VALU wave-instructions instead of 1.61e9 for the same lane-cycles (8.5e10 against 8.6e10) -- lane use
        // falls from 81 % to ~50 % and the frame goes from 7.13 to > 7.7 ms (the measured choice then takes the megakernel).  The walk's
        // wave-uniform phases live on the lanes of a wave being at the SAME phase: 64 rays that start together through one tile stay roughly in
        // step, lanes restarted at the root while their neighbours are deep in the tree do not, and every phase then runs for a few lanes.
        static const bool refill_on = [] { const char *e = std::getenv("LASGUN_REFILL"); return e && e[0] == '1'; }();
        const bool refill = refill_on && shadow && P.wf_level == 0u && (unsigned long long)P.ntiles >= 4ull * blocks * (block / 64u);
        if (shadow) { if (refill) hipLaunchKernelGGL((wf_trace_kernel<false, true, true, false, false, true>), dim3(blocks), dim3(block), lds, stream, P); else LG_LAUNCH(false, true, true, false); }
        else if (l0) LG_LAUNCH(false, false, true, true); else LG_LAUNCH(false, false, true, false);
    }
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
hipError_t wf_trace_occupancy(uint32_t stack_depth, bool fast, size_t extra_lds, int *blocks_per_cu) {
    size_t lds = (size_t)stack_depth * LG_BLOCK * sizeof(uint32_t) + (fast ? 0u : extra_lds);
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
// raise the dynamic-LDS limit of this file's kernels to `bytes` (ldss: the LDS-resident-scene forms; otherwise the 256-lane forms)
hipError_t wf_set_lds_limit(size_t bytes, bool ldss) {
    const void *ldss_fns[] = {
        reinterpret_cast<const void *>(wf_trace_kernel<false, false, true, true>),
        reinterpret_cast<const void *>(wf_trace_kernel<false, true, true, false>),
        reinterpret_cast<const void *>(wf_trace_kernel<false, false, true, false>),
        reinterpret_cast<const void *>(wf_trace_kernel<false, false, true, true, true>),
        reinterpret_cast<const void *>(wf_trace_kernel<false, true, true, false, true>),
        reinterpret_cast<const void *>(wf_trace_kernel<false, false, true, false, true>),
        reinterpret_cast<const void *>(wf_trace_kernel<false, true, true, false, false, true>)};
    const void *plain_fns[] = {
        reinterpret_cast<const void *>(wf_trace_kernel<false, false, false, true>),
        reinterpret_cast<const void *>(wf_trace_kernel<false, true, false, false>),
        reinterpret_cast<const void *>(wf_trace_kernel<false, false, false, false>),
        reinterpret_cast<const void *>(wf_trace_kernel<false, false, false, true, true>),
        reinterpret_cast<const void *>(wf_trace_kernel<false, true, false, false, true>),
        reinterpret_cast<const void *>(wf_trace_kernel<false, false, false, false, true>),
        reinterpret_cast<const void *>(wf_trace_kernel<true, false, false, true>),
        reinterpret_cast<const void *>(wf_trace_kernel<true, true, false, false>),
        reinterpret_cast<const void *>(wf_trace_kernel<true, false, false, false>)};
    const void *const *fns = ldss ? ldss_fns : plain_fns;
    const size_t n = ldss ? sizeof ldss_fns / sizeof ldss_fns[0] : sizeof plain_fns / sizeof plain_fns[0];
    for (size_t i = 0; i < n; ++i) {
        hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

} // namespace lg
