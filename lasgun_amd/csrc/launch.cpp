// lasgun_amd/csrc/launch.cpp -- one render enqueued on a stream (internal.h): the launch contexts, the three kernel organisations
// (level by level, queue, megakernel), the fitted rule and the measured choice between them (tune.cpp does the measuring), and the
// addressing modes of capture_subset (strided subsets, batches, lattice tiling).
#include "internal.h"

// The launch context of `stream` (at most MAX_LAUNCH_CTXS are kept; the least recently used one is recycled after a
// device-wide synchronise).  Caller holds a.mtx and has made the accel's device current.
static lg_accel::LaunchCtx &ctx_for(const lg_accel &a, hipStream_t stream) {
    for (auto &c : a.ctxs)
        if (c->key == stream) { c->last_use = ++a.ctx_clock; return *c; }
    lg_accel::LaunchCtx *c = nullptr;
    if (a.ctxs.size() < MAX_LAUNCH_CTXS) {
        a.ctxs.emplace_back(new lg_accel::LaunchCtx());
        c = a.ctxs.back().get();
        c->tile_counter.alloc(TILE_COUNTER_WORDS);
        HIP_TRY(hipMemset(c->tile_counter.p, 0, TILE_COUNTER_WORDS * sizeof(uint32_t)));
    } else {
        c = a.ctxs[0].get();
        for (auto &x : a.ctxs) if (x->last_use < c->last_use) c = x.get();
        HIP_TRY(hipDeviceSynchronize()); // nothing may still be using the recycled buffers
    }
    c->key = stream;
    c->last_use = ++a.ctx_clock;
    return *c;
}

// The queue organisation's error word: a wave that gave up waiting for work that never came (a scheduler bug) must fail a call, not
// leave a half-rendered film behind.  The word is sticky -- the device only ever sets it -- and is cleared here, once reported.
// Looked at after every synchronise of the accel's stream, at the head of every enqueue and in lg_accel_synchronize: a launch on a
// CALLER's stream that stalled is reported by the first of those that follows its end.  Caller holds a.mtx.
void check_queue_error(const lg_accel &a) {
    if (!a.q_err) return;
    volatile uint32_t *w = a.q_err;
    if (*w == 0u) return;
    *w = 0u;
    throw Error("queue organisation: a wave gave up waiting for work (scheduler stalled); the film is incomplete");
}
void sync_checked(const lg_accel &a) {
    HIP_TRY(hipStreamSynchronize(a.stream));
    check_queue_error(a);
}

// ------------------------------------------------------------------------------------------
DParams base_params(const lg_accel &a, uint32_t w, uint32_t h) {
    const Scene &s = *a.scene;
    DParams P{};
    P.nodes = a.nodes.p; P.nodes4 = a.nodes4.p; P.primref = a.primref.p; P.spheres = a.spheres.p; P.sphere_mat = a.sphere_mat.p;
    P.cuboids = a.cuboids.p; P.cuboid_mat = a.cuboid_mat.p; P.tri_v = a.tri_v.p; P.tri_n = a.tri_n.p; P.tri_t = a.tri_t.p;
    P.vpos = a.vpos.p; P.vnorm = a.vnorm.p; P.vtex = a.vtex.p; P.leaf_soup = a.leaf_soup.p; P.chunks = a.chunks.p; P.strips = a.strips.p; P.sphere_ref_leaf = a.sphere_ref_leaf.p; P.cuboid_ref_leaf = a.cuboid_ref_leaf.p;
    P.tri_ref_leaf = a.tri_ref_leaf.p; P.accel_ref_leaf = a.accel_ref_leaf.p; P.accels = a.accels.p; P.materials = a.materials.p;
    P.lights = a.lights.p;
    P.nlights = (uint32_t)a.flat.lights.size();
    P.recursion = s.recursion;
    P.default_material = a.flat.default_material;
    P.stack_depth = a.stack_depth;
    {   // LASGUN_ACCEL_LDS=0 (A/B): the accel records from the DAccel table in L2
        static const bool accel_lds = [] { const char *e = std::getenv("LASGUN_ACCEL_LDS"); return !(e && e[0] == '0'); }();
        P.accel_image = a.accel_image_n16 && accel_lds ? a.accel_image.p : nullptr; P.accel_image_n16 = a.accel_image_n16;
    }
    {   // LASGUN_PRUNE=0|1 replaces the scene-dependent DEFAULT (test suites run whole under either); lg_accel_set_prune still wins
        static const int env_default = [] { const char *e = std::getenv("LASGUN_PRUNE"); return e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : -1; }();
        const bool dflt = env_default < 0 ? a.prune_default : env_default != 0;
        P.prune = !a.fast && (a.prune < 0 ? dflt : a.prune != 0) ? 1u : 0u;
    }
    P.cam_origin = s.camera.origin; P.cam_view = s.camera.view; P.cam_up = s.camera.up; P.cam_aux = s.camera.aux;
    P.image_plane_height = s.camera.image_plane_height;
    P.pixel_separation = s.camera.pixel_separation;
    P.ss_distance = s.camera.ss_distance;
    P.ss_root = s.camera.ss_root;
    {   // LASGUN_SLAB_SIGNS=0: the reference's slab formula as written in every node step (A/B, tests)
        static const bool signs = [] { const char *e = std::getenv("LASGUN_SLAB_SIGNS"); return !(e && e[0] == '0'); }();
        P.boxes_finite = a.flat.boxes_finite && signs ? 1u : 0u;
    }
    P.bg_inner = s.bg_inner; P.bg_outer = s.bg_outer; P.bg_scale = s.bg_scale;
    P.ambient = s.ambient;
    P.w = w; P.h = h;
    P.winv = 1. / (double)w; P.hinv = 1. / (double)h; P.aspect = (double)w / (double)h; // film.rs:40-42
    return P;
}

// the accel's internal streams (bands of a big wavefront launch, bands of a whole-film capture).  Caller holds a.mtx.
void ensure_aux_streams(const lg_accel &a, unsigned n) {
    while (a.aux_streams.size() < n) {
        hipStream_t st = g_streams.take(a.device); hipEvent_t ev = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        a.aux_streams.push_back(st); a.aux_done.push_back(ev);
    }
    if (!a.aux_fork) HIP_TRY(hipEventCreateWithFlags(&a.aux_fork, hipEventDisableTiming));
}

// The wavefront pipeline (k_wavefront.hip): per chunk of the film and per supersample, levels 0 .. L-1 top-down (closest,
// shadow, shade), then the combine passes bottom-up.  Queue capacities are worst case (level d holds at most 2^d rays per
// pixel of the chunk), so the chunk is sized to the memory budget of the launch context: nothing can overflow.
constexpr size_t WF_FULL_MIN_HOST = 48; // == WF_FULL_MIN of k_wavefront.hip
constexpr uint32_t MEGA_SPLIT = 4;      // the parts a small launch's tiles are handed out in where the measured choice found that faster (enqueue_mega, enqueue_queue)
static void enqueue_wavefront(const lg_accel &a, DParams &P0, lg_accel::LaunchCtx &c, hipStream_t stream) {
    const uint32_t levels = (a.flat.has_specular && P0.recursion > 0) ? P0.recursion + 1u : 1u;
    const uint32_t nsamples = P0.ss_root * P0.ss_root;
    // A supersampled launch runs its samples SIDE BY SIDE (DParams::ss_par): level 0 holds pixels x samples work items, one chain of
    // launches per chunk instead of one per sample, and a resolve pass sums each pixel's samples in their order.  The levels of a 9-sample
    // frame are nine times as wide -- a 512^2 film of glass fills the machine at its deep levels, which one sample at a time does not.
    // lg_accel_set_sample_order(1) / LASGUN_SS_SERIAL=1 (A/B): one chain per sample, summed as they come.
    static const bool ss_serial = [] { const char *e = std::getenv("LASGUN_SS_SERIAL"); return e && e[0] == '1'; }();
    const uint32_t S = nsamples > 1 && !ss_serial && a.sample_order != 1 ? nsamples : 1u; // level-0 work items per pixel
    // bytes per level-0 work item of a chunk
    auto level_bytes = [&](uint32_t d) -> size_t {
        size_t b = 0;
        if (d >= 1) b += 6 * 8;                       // ray queue
        if (levels > 1) b += 3 * 8;                   // output / li
        if (d + 1 < levels) b += 8 * 8 + 2 * 4;       // children's weights and indices
        return b;
    };
    size_t per_item = (nsamples > 1 ? 3 * 8 : 0);
    for (uint32_t d = 0; d < levels; ++d) per_item += level_bytes(d) << d;
    per_item += ((size_t)(4 + STASH_DOUBLES * 8 + 4) << (levels - 1)) * 7 / 4; // hit queue, frame, visibility of the widest level: dense part + appended part
    const size_t per_pixel = per_item * S;
    if (a.wf_budget == 0) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        // per launch CONTEXT, and an accel keeps up to MAX_LAUNCH_CTXS of them plus the band contexts (a caller with
        // four frames in flight uses five): a sixteenth of the free memory, at most 8 GiB each (the headline frame
        // needs 3.3 GB and stays one chunk; an allocation that fails anyway halves the chunk below)
        size_t budget = free_b / 16;
        const char *env = std::getenv("LASGUN_WF_BUDGET_MB");
        if (env && std::atoll(env) > 0) budget = (size_t)std::atoll(env) << 20;
        else if (budget > (8ull << 30)) budget = 8ull << 30;
        a.wf_budget = budget < (64ull << 20) ? (64ull << 20) : budget;
    }
    unsigned long long chunk_tiles = a.wf_budget / (per_pixel * 64);
    const unsigned long long cap_limit = (0xFFFFFFF0ull >> (levels - 1)) / (64ull * S); // ray indices are 32-bit
    if (chunk_tiles > cap_limit) chunk_tiles = cap_limit;
    if (chunk_tiles < 1) chunk_tiles = 1;
    if (chunk_tiles > P0.ntiles) chunk_tiles = P0.ntiles;
    // Bands on internal streams: opt-in (lg_accel_set_wf_split, or LASGUN_WF_SPLIT=n as the default), launches of 2 Mpixel and
    // more.  Measured: one headline frame at a time 7.79 -> 7.50 ms with 4 bands, but 7.20 -> 7.46 ms
    // when the caller already keeps four frames in flight -- which is why it is not the default.
    static const unsigned split_env = [] { const char *e = std::getenv("LASGUN_WF_SPLIT"); return e && std::atoi(e) > 0 ? (unsigned)std::atoi(e) : 1u; }();
    const unsigned want = a.wf_split ? a.wf_split : split_env;
    // (at most MAX_WF_BANDS bands: their contexts and the callers' streams share the accel's MAX_LAUNCH_CTXS slots, and a
    // context that has to be recycled costs a device-wide synchronise)
    const unsigned split = (unsigned long long)P0.ntiles * 64ull >= (1ull << 21) ? std::min(want, MAX_WF_BANDS) : 1u;
    if (split > 1) chunk_tiles = std::min<unsigned long long>(chunk_tiles, (P0.ntiles + split - 1) / split);
    unsigned long long nchunks = (P0.ntiles + chunk_tiles - 1) / chunk_tiles;
    const unsigned nstreams = split > 1 && nchunks > 1 ? (unsigned)std::min<unsigned long long>(split, nchunks) : 0u; // 0: everything on the caller's stream
    if (nstreams) ensure_aux_streams(a, nstreams);
    unsigned long long n0 = 0;
    size_t need = 0, hit_cap = 0, hit_len = 0;
    const uint32_t nlaunch = 4 * levels;
    const uint32_t CL = TILE_COUNTER_WORDS; // the queue counts (3 per level) in the first block, then a block of tile heads per launch (one head per XCD, each on a line of its own)
    auto size_chunk = [&] {
        n0 = chunk_tiles * 64ull * S;
        need = (size_t)n0 * per_item + 4096 * (3 * levels + 4);
        hit_cap = (size_t)n0 << (levels - 1);
        hit_len = hit_cap + hit_cap / 64 * (WF_FULL_MIN_HOST - 1); // appended part: fewer than WF_FULL_MIN hits per block of 64 rays
        nchunks = (P0.ntiles + chunk_tiles - 1) / chunk_tiles;
    };
    size_chunk();
    struct Carved {
        std::vector<double *> q, out, spec;
        std::vector<uint32_t *> child;
        uint32_t *hq = nullptr, *vis = nullptr, *counters = nullptr;
        double *frame = nullptr, *accum = nullptr;
    };
    auto carve = [&](lg_accel::LaunchCtx &cx) { // this context's arrays for one chunk (256-byte aligned)
        if (cx.wf_mem.n < need) { HIP_TRY(hipDeviceSynchronize()); cx.wf_mem.alloc(need); }
        if (cx.wf_counters.n < CL * (1 + nlaunch)) { HIP_TRY(hipDeviceSynchronize()); cx.wf_counters.alloc(CL * (1 + nlaunch)); }
        Carved k;
        k.q.assign(levels, nullptr); k.out.assign(levels, nullptr); k.spec.assign(levels, nullptr); k.child.assign(levels, nullptr);
        uint8_t *cur = cx.wf_mem.p;
        auto take = [&](size_t bytes) { uint8_t *p = cur; cur += (bytes + 255) & ~(size_t)255; return p; };
        for (uint32_t d = 0; d < levels; ++d) {
            const size_t cap = (size_t)n0 << d;
            if (d >= 1) k.q[d] = (double *)take(cap * 6 * 8);
            if (levels > 1) k.out[d] = (double *)take(cap * 3 * 8);
            if (d + 1 < levels) { k.spec[d] = (double *)take(cap * 8 * 8); k.child[d] = (uint32_t *)take(cap * 2 * 4); }
        }
        k.hq = (uint32_t *)take(hit_len * 4);
        k.frame = (double *)take(hit_len * STASH_DOUBLES * 8);
        k.vis = (uint32_t *)take(hit_len * 4);
        k.accum = nsamples > 1 ? (double *)take((size_t)n0 * 3 * 8) : nullptr;
        k.counters = cx.wf_counters.p;
        return k;
    };
    std::vector<Carved> carved;
    std::vector<hipStream_t> lanes;
    for (;;) { // memory that is not there (other contexts, other accels, other processes): halve the chunk and carve again
        try {
            carved.clear(); lanes.clear();
            if (nstreams) for (unsigned j = 0; j < nstreams; ++j) { lanes.push_back(a.aux_streams[j]); carved.push_back(carve(ctx_for(a, a.aux_streams[j]))); }
            else { lanes.push_back(stream); carved.push_back(carve(c)); }
            break;
        } catch (const Error &e) {
            if (std::string(e.what()).find("hipMalloc") == std::string::npos || chunk_tiles <= 1) throw;
            (void)hipGetLastError(); // (the failed allocation's error must not be what the next launch reports)
            chunk_tiles = (chunk_tiles + 1) / 2;
            a.wf_budget = std::max<size_t>(a.wf_budget / 2, 64ull << 20); // (later launches start from what fitted)
            size_chunk();
            if (std::getenv("LASGUN_DEBUG")) std::fprintf(stderr, "[lasgun] wavefront: %s -- chunks of %llu tiles instead\n", e.what(), chunk_tiles);
        }
    }
    if (nstreams) {
        HIP_TRY(hipEventRecord(a.aux_fork, stream)); // the bands start after whatever the caller's stream holds (a film clear, the previous frame's copy)
        for (unsigned j = 0; j < nstreams; ++j) HIP_TRY(hipStreamWaitEvent(a.aux_streams[j], a.aux_fork, 0));
    }

    const bool ldss = !a.fast && a.lds_scene && a.ldss_blocks;
    const uint32_t depth = a.fast ? a.stack_depth_fast1 : a.stack_depth;
    const uint32_t trace_cap = ldss ? a.ldss_blocks : (a.fast ? a.wf_blocks_fast : a.wf_blocks);
    const uint32_t flat_cap = a.cus * 16u;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (a.profiling) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventRecord(e0, stream)); }
    hipStream_t ls = stream; // the stream of the chunk being enqueued
    auto timed = [&](int kind, auto &&launch) { // HIP events around ONE kernel on its launch stream
        hipEvent_t k0 = nullptr, k1 = nullptr;
        if (a.profiling) { HIP_TRY(hipEventCreate(&k0)); HIP_TRY(hipEventCreate(&k1)); HIP_TRY(hipEventRecord(k0, ls)); }
        HIP_TRY(launch());
        if (a.profiling) { HIP_TRY(hipEventRecord(k1, ls)); a.kind_events[kind].emplace_back(k0, k1); }
    };
    if (std::getenv("LASGUN_DEBUG"))
        std::fprintf(stderr, "[lasgun] wavefront: levels %u, %llu tiles in chunks of %llu on %u stream(s) (%.1f MiB per context), trace grid %u x %u, stack %u, max_blocks %u\n", levels,
                     (unsigned long long)P0.ntiles, chunk_tiles, nstreams ? nstreams : 1u, need / 1048576.0, trace_cap, ldss ? 1024u : 256u, depth, a.max_blocks);
    // A SMALL frame's chain is a dozen dependent launches of a few microseconds each (Cornell glass 512^2: 16 launches for 0.4 ms), and
    // what separates them on a stream is the runtime's launch path per kernel.  The chain has no host decision in it -- fixed grids, counts
    // on the device -- so it is captured ONCE into a HIP graph and replayed: frames of <= 2^20 work items, one chunk, the caller's own
    // stream (not the null stream), not profiling; captured when the same chain (a hash of every parameter: scene tables, camera, film
    // pointer, carved arrays, grids) comes a second time in a row on the context, so a program that renders one frame never pays
    // for a capture, and re-captured at most MAX_GRAPH_CAPTURES times per context (a caller that changes the film every frame gains nothing
    // and stops paying).  The bytes are the same launches' bytes.
    // MEASURED, and OFF unless LASGUN_GRAPH=1 (profiles/r06_small_frames.jsonl, tools/ab_small_frames.sh, variants in turn on one box): the
    // replay is SLOWER where it was meant to pay -- the README sphere at 512^2 0.086 -> 0.093 ms alone and 0.072 -> 0.081 back to back (4 nodes),
    // Cornell plastic 0.119 -> 0.129 / 0.104 -> 0.116 -- and within +-2 % on every longer chain (simple.rs 9 spp, Cornell glass at 256^2 / 512^2,
    // spooky.rs, playground.rs, simplecows.rs: 16 - 28 nodes).  On this runtime (ROCm 7.2) a graph launch costs more than the stream launches it
    // replaces and the gaps between dependent kernels do not shrink.
    static const bool graphs_on = [] { const char *e = std::getenv("LASGUN_GRAPH"); return e && e[0] == '1'; }();
    constexpr unsigned MAX_GRAPH_CAPTURES = 8;
    bool capturing = false;
    if (graphs_on && stream != nullptr && nstreams == 0 && nchunks == 1 && !a.profiling && (unsigned long long)P0.ntiles * 64ull * S <= (1ull << 20)) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusActive; }
        if (cs == hipStreamCaptureStatusNone) {
            uint64_t sig = 1469598103934665603ull;
            auto mix = [&sig](const void *q, size_t n) { const uint8_t *b = (const uint8_t *)q; for (size_t i = 0; i < n; ++i) { sig ^= b[i]; sig *= 1099511628211ull; } };
            mix(&P0, sizeof P0);
            const Carved &K0 = carved[0];
            mix(&K0.hq, sizeof K0.hq); mix(&K0.counters, sizeof K0.counters); mix(&K0.frame, sizeof K0.frame); mix(&K0.accum, sizeof K0.accum);
            const uint64_t shape[8] = {levels, S, n0, chunk_tiles, trace_cap, depth, (uint64_t)a.fast | ((uint64_t)ldss << 1), (uint64_t)(uintptr_t)a.lds_image.p};
            mix(shape, sizeof shape);
            if (sig == 0) sig = 1;
            if (c.wf_graph && c.wf_graph_sig == sig) {
                const hipError_t ge = hipGraphLaunch(c.wf_graph, stream);
                if (ge == hipSuccess) return;
                (void)hipGetLastError(); // a replay that is refused: the plain chain below
                (void)hipGraphExecDestroy(c.wf_graph); c.wf_graph = nullptr; c.wf_graph_sig = 0; c.wf_graph_captures = MAX_GRAPH_CAPTURES;
            } else if (c.wf_last_sig == sig && c.wf_graph_captures < MAX_GRAPH_CAPTURES) {
                if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) == hipSuccess) { capturing = true; c.wf_graph_captures++; }
                else (void)hipGetLastError();
            }
            c.wf_last_sig = sig;
        }
    }
    const uint64_t chain_sig = c.wf_last_sig;
    auto run_chain = [&] {
    unsigned long long chunk_no = 0;
    for (unsigned long long t0 = 0; t0 < P0.ntiles; t0 += chunk_tiles, ++chunk_no) {
        const Carved &K = carved[chunk_no % carved.size()];
        ls = lanes[chunk_no % lanes.size()];
        const std::vector<double *> &q = K.q, &out = K.out, &spec = K.spec;
        const std::vector<uint32_t *> &child = K.child;
        uint32_t *const hq = K.hq, *const vis = K.vis;
        double *const frame = K.frame, *const accum = K.accum;
        DParams P = P0;
        P.tile0 = (uint32_t)t0;
        const uint32_t pixel_tiles = (uint32_t)std::min<unsigned long long>(chunk_tiles, P0.ntiles - t0);
        P.ntiles = pixel_tiles * S; // level 0's work tiles
        P.ss_par = S;
        P.n_items = n0; // stride of the sample accumulator
        P.accum = accum;
        P.wf_levels = levels;
        P.wf_counts = K.counters;
        P.wf_hit_cap = hit_cap; P.wf_hit_stride = hit_len; P.wf_hq = hq; P.frame = frame; P.vis = vis;
#ifdef LG_STAMPS
        P.stats = a.stats.p;
        P.stamp_counts = reinterpret_cast<unsigned long long *>(a.stats.p + 1);
#endif
        if (ldss) {
            P.lds_image = a.lds_image.p; P.lds_image_n16 = a.lds_image_n16;
            P.lds_node_off = a.lds_node_off; P.lds_prim_off = a.lds_prim_off; P.lds_soup_off = a.lds_soup_off; P.lds_accel_off = a.lds_accel_off;
        }
        const uint32_t tiles_needed = (P.ntiles + 3u) / 4u;
        const uint32_t trace_blocks0 = ldss ? trace_cap : std::min(trace_cap, tiles_needed);
        const uint32_t flat_blocks0 = (uint32_t)(((unsigned long long)P.ntiles * 64ull + 255ull) / 256ull); // level 0: one thread per pixel
        // level-0 shade: one wave per dense tile and per tile the appended hits can fill (< WF_FULL_MIN of every 64 rays)
        const uint32_t shade_blocks0 = (uint32_t)(((unsigned long long)P.ntiles + ((unsigned long long)P.ntiles * (WF_FULL_MIN_HOST - 1) + 63ull) / 64ull + 3ull) / 4ull);
        for (uint32_t sidx = 0; sidx < nsamples / S; ++sidx) {
            P.sample_index = sidx;
            HIP_TRY(hipMemsetAsync(K.counters, 0, CL * (1 + nlaunch) * sizeof(uint32_t), ls));
            uint32_t launch_no = 0;
            auto level_params = [&](uint32_t d) {
                P.wf_level = d;
                P.wf_cap = (unsigned long long)n0 << d; P.wf_cap_next = (unsigned long long)n0 << (d + 1);
                P.wf_q = q[d]; P.wf_out = out[d]; P.wf_spec = spec[d]; P.wf_child = child[d];
                P.wf_q_next = d + 1 < levels ? q[d + 1] : nullptr;
                P.wf_out_next = d + 1 < levels ? out[d + 1] : nullptr;
                P.tile_counter = K.counters + CL * (1 + launch_no++);
            };
            for (uint32_t d = 0; d < levels; ++d) {
                // (deeper levels: the number of rays is only known on the device; grids are sized for a full level 0, which
                // every deeper level may exceed only in waves, never in work per wave)
                const uint32_t tb = d == 0 ? trace_blocks0 : trace_cap, fb = d == 0 ? shade_blocks0 : flat_cap;
                level_params(d);
                timed(0, [&] { return launch_wf_trace(P, a.fast, false, tb, depth, ls); });
                if (P.nlights > 0) {
                    level_params(d);
                    timed(2, [&] { return launch_wf_trace(P, a.fast, true, tb, depth, ls); });
                }
                level_params(d);
                timed(3, [&] { return launch_wf_shade(P, fb, ls); });
            }
            for (uint32_t d = levels - 1; d-- > 0;) {
                level_params(d);
                timed(1, [&] { return launch_wf_combine(P, d == 0 ? flat_blocks0 : flat_cap, ls); });
            }
        }
        if (S > 1) {
            DParams R = P;
            R.ntiles = pixel_tiles;
            timed(1, [&] { return launch_wf_resolve(R, (uint32_t)(((unsigned long long)pixel_tiles * 64ull + 255ull) / 256ull), ls); });
        }
    }
    };
    if (capturing) { // record the chain, replay it; a capture that was begun is always ended (a stream left in capture mode is lost to its owner)
        hipGraph_t g = nullptr;
        try { run_chain(); } catch (...) { (void)hipStreamEndCapture(stream, &g); if (g) (void)hipGraphDestroy(g); (void)hipGetLastError(); throw; }
        bool launched = false;
        if (hipStreamEndCapture(stream, &g) == hipSuccess && g) {
            if (c.wf_graph) { (void)hipGraphExecDestroy(c.wf_graph); c.wf_graph = nullptr; c.wf_graph_sig = 0; }
            hipGraphExec_t x = nullptr;
            if (hipGraphInstantiate(&x, g, nullptr, nullptr, 0) == hipSuccess && x) {
                if (hipGraphLaunch(x, stream) == hipSuccess) { c.wf_graph = x; c.wf_graph_sig = chain_sig; launched = true; }
                else (void)hipGraphExecDestroy(x);
            }
        }
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        if (launched) return; // (no bands, no profiling on this path)
        c.wf_graph_captures = MAX_GRAPH_CAPTURES; // captured but not launched: the frame still has to be rendered, plainly, and this context stops trying
    }
    run_chain();
    for (unsigned j = 0; j < nstreams; ++j) { // join: the caller's stream continues when every band is done
        HIP_TRY(hipEventRecord(a.aux_done[j], a.aux_streams[j]));
        HIP_TRY(hipStreamWaitEvent(stream, a.aux_done[j], 0));
    }
    if (a.profiling) { HIP_TRY(hipEventRecord(e1, stream)); a.events.emplace_back(e0, e1); }
}

// The queue organisation (k_queue.hip): per chunk of the film and per supersample ONE persistent launch that runs every recursion
// level -- its waves pull 64-ray packets from per-level queues, deepest level first -- then the combine passes bottom-up, shared
// with the level-by-level pipeline.  Queue capacities are worst case (level d: 2^d rays per pixel of the chunk), so nothing can
// overflow; a recursive scene may take a large share of the HBM for it (a 4096^2 frame at recursion 3: 26 GB of 288) and keeps ONE
// chunk in flight per launch context.
static void enqueue_queue(const lg_accel &a, DParams &P0, lg_accel::LaunchCtx &c, bool split, hipStream_t stream) {
    const uint32_t levels = (a.flat.has_specular && P0.recursion > 0) ? P0.recursion + 1u : 1u;
    const uint32_t nsamples = P0.ss_root * P0.ss_root;
    static const bool ss_serial = [] { const char *e = std::getenv("LASGUN_SS_SERIAL"); return e && e[0] == '1'; }();
    // level 0's tiles in parts (enqueue_mega, DParams::split_shift): the children's packets are then as narrow as their parents -- a small
    // launch's recursion chains are walked by four times the waves, 16 lanes each
    const uint32_t parts = a.tile_parts >= 1 ? (uint32_t)a.tile_parts : split ? MEGA_SPLIT : 1u, split_shift = parts == 8u ? 3u : parts == 4u ? 2u : parts == 2u ? 1u : 0u;
    const uint32_t S = (nsamples > 1 && !ss_serial && a.sample_order != 1 ? nsamples : 1u) * parts; // samples side by side (enqueue_wavefront) x parts: level-0 tiles per pixel tile
    auto level_bytes = [&](uint32_t d) -> size_t { // per ray of level d
        size_t b = 0;
        if (d >= 1) b += 6 * 8;                       // ray queue
        if (levels > 1) b += 3 * 8;                   // output / li
        if (d + 1 < levels) b += 8 * 8 + 2 * 4;       // children's weights and indices
        return b;
    };
    size_t per_item = (nsamples > 1 ? 3 * 8 : 0) + 1;
    for (uint32_t d = 0; d < levels; ++d) per_item += level_bytes(d) << d;
    const size_t per_pixel = per_item * S;
    if (a.queue_budget == 0) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        // a recursive scene: up to a quarter of the free memory (at most 48 GiB) per launch context, so that a 4096^2 frame is one
        // launch; others need a few bytes per pixel and take the wavefront pipeline's share (an allocation that fails halves the chunk)
        size_t budget = levels > 1 ? free_b / 4 : free_b / 16;
        const size_t cap = levels > 1 ? (48ull << 30) : (8ull << 30);
        const char *env = std::getenv("LASGUN_QUEUE_BUDGET_MB");
        const bool from_env = env && std::atoll(env) > 0;
        if (from_env) budget = (size_t)std::atoll(env) << 20; // (as given: tests cut small films into many chunks with it)
        else if (budget > cap) budget = cap;
        a.queue_budget = !from_env && budget < (64ull << 20) ? (64ull << 20) : budget;
    }
    unsigned long long chunk_tiles = a.queue_budget / (per_pixel * 64);
    const unsigned long long cap_limit = (0xFFFFFF00ull >> (levels - 1)) / (64ull * S); // ray indices are 32-bit
    if (chunk_tiles > cap_limit) chunk_tiles = cap_limit;
    if (chunk_tiles < 1) chunk_tiles = 1;
    if (chunk_tiles > P0.ntiles) chunk_tiles = P0.ntiles;
    if (chunk_tiles < P0.ntiles && P0.mode == 0u && P0.tiles_x != 0u && chunk_tiles >= (unsigned long long)P0.tiles_x * 32ull)
        chunk_tiles -= chunk_tiles % ((unsigned long long)P0.tiles_x * 32ull); // whole rows of 32 x 32-tile blocks: the block order applies to every chunk
    // level 0's work items: units of consecutive 8x8 tiles whose specular children the wave compacts into packets of its own.  Default
    // 1 (measured, config 4 / 4m in ms: 1 tile 39.0 / 16.4, 2: 40.0 / 17.0, 4: 41.0 / 18.5, 8: 44.6 / 22.2, 16: 51.7 / 31.6 -- a mesh tile is a
    // millisecond of work, so longer units lengthen the launch's tail by more than fuller packets save); LASGUN_QUEUE_UNIT: A/B
    static const uint32_t unit_tiles = [] { const char *e = std::getenv("LASGUN_QUEUE_UNIT"); const int v = e ? std::atoi(e) : 0; return v >= 1 && v <= 64 ? (uint32_t)v : 1u; }();
    // LASGUN_QUEUE_ORDER=1 (A/B): 32 x 32-tile blocks in Morton order, claimed XCD by XCD -- measured no better than row order with one
    // claim counter (config 4 / 4m / 5: 39.4 / 16.7 / 63.7 against 38.4 / 16.0 / 64.8 ms): which tiles are in flight together does not
    // move these kernels, as round 3 found for the other organisations
    static const bool order_blocks = [] { const char *e = std::getenv("LASGUN_QUEUE_ORDER"); return e && e[0] == '1'; }();
    const bool ldss = a.lds_scene && a.ldss_blocks;
    const uint32_t blocks_cap = ldss ? a.ldss_blocks : a.queue_blocks;
    const unsigned long long threads = (unsigned long long)blocks_cap * (ldss ? 1024ull : 256ull);
    unsigned long long n0 = 0;
    size_t need = 0, nready = 0;
    struct Carved {
        std::vector<double *> q, out, spec;
        std::vector<uint32_t *> child;
        double *accum = nullptr;
    } K;
    for (;;) { // memory that is not there: halve the chunk and carve again
        try {
            n0 = chunk_tiles * 64ull * S;
            need = (size_t)n0 * per_item + 4096 * (4 * levels + 4);
            nready = (size_t)chunk_tiles * S * ((1ull << levels) - 2ull) + (size_t)levels * QR_SLACK; // one word per packet of the levels >= 1, + slack per level
            if (c.wf_mem.n < need) { HIP_TRY(hipDeviceSynchronize()); c.wf_mem.alloc(need); }
            if (c.wf_counters.n < QC_WORDS + nready) { HIP_TRY(hipDeviceSynchronize()); c.wf_counters.alloc(QC_WORDS + nready); }
            if (P0.nlights > 0 && c.stash.n < (size_t)threads * STASH_DOUBLES) { HIP_TRY(hipDeviceSynchronize()); c.stash.alloc((size_t)threads * STASH_DOUBLES); }
            break;
        } catch (const Error &e) {
            if (std::string(e.what()).find("hipMalloc") == std::string::npos || chunk_tiles <= 1) throw;
            (void)hipGetLastError();
            chunk_tiles = (chunk_tiles + 1) / 2;
            a.queue_budget = std::max<size_t>(a.queue_budget / 2, 64ull << 20);
            if (std::getenv("LASGUN_DEBUG")) std::fprintf(stderr, "[lasgun] queue: %s -- chunks of %llu tiles instead\n", e.what(), chunk_tiles);
        }
    }
    {
        K.q.assign(levels, nullptr); K.out.assign(levels, nullptr); K.spec.assign(levels, nullptr); K.child.assign(levels, nullptr);
        uint8_t *cur = c.wf_mem.p;
        auto take = [&](size_t bytes) { uint8_t *p = cur; cur += (bytes + 255) & ~(size_t)255; return p; };
        for (uint32_t d = 0; d < levels; ++d) {
            const size_t cap = (size_t)n0 << d;
            if (d >= 1) K.q[d] = (double *)take(cap * 6 * 8);
            if (levels > 1) K.out[d] = (double *)take(cap * 3 * 8);
            if (d + 1 < levels) { K.spec[d] = (double *)take(cap * 8 * 8); K.child[d] = (uint32_t *)take(cap * 2 * 4); }
        }
        K.accum = nsamples > 1 ? (double *)take((size_t)n0 * 3 * 8) : nullptr;
    }
    if (!a.q_err) a.q_err = g_err_words.take();
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (a.profiling) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventRecord(e0, stream)); }
    auto timed = [&](int kind, auto &&launch) { // HIP events around ONE kernel on its launch stream
        hipEvent_t k0 = nullptr, k1 = nullptr;
        if (a.profiling) { HIP_TRY(hipEventCreate(&k0)); HIP_TRY(hipEventCreate(&k1)); HIP_TRY(hipEventRecord(k0, stream)); }
        HIP_TRY(launch());
        if (a.profiling) { HIP_TRY(hipEventRecord(k1, stream)); a.kind_events[kind].emplace_back(k0, k1); }
    };
    if (std::getenv("LASGUN_DEBUG"))
        std::fprintf(stderr, "[lasgun] queue: levels %u, %llu tiles in chunks of %llu (%.1f MiB), grid %u x %u, stack %u\n", levels,
                     (unsigned long long)P0.ntiles, chunk_tiles, need / 1048576.0, blocks_cap, ldss ? 1024u : 256u, a.stack_depth);
    const uint32_t flat_cap = a.cus * 16u;
    for (unsigned long long t0 = 0; t0 < P0.ntiles; t0 += chunk_tiles) {
        DParams P = P0;
        P.tile0 = (uint32_t)t0;
        const uint32_t pixel_tiles = (uint32_t)std::min<unsigned long long>(chunk_tiles, P0.ntiles - t0);
        P.ntiles = pixel_tiles * S; // level 0's tiles
        P.ss_par = S / parts; P.split_shift = split_shift;
        P.n_items = n0; // SoA stride of level 0's arrays and of the sample accumulator
        P.accum = K.accum;
        P.wf_levels = levels;
        P.q_ctl = c.wf_counters.p; P.q_ready = c.wf_counters.p + QC_WORDS; P.q_err = a.q_err;
        P.q_unit_tiles = levels > 1 ? unit_tiles : 1u;
        // the tile sequence: rectangles whose chunk is whole tile rows go block by block, XCD by XCD (k_queue.hip, q_seq_tile)
        P.q_order = (order_blocks && !ldss && S == 1u && P.mode == 0u && P.tiles_x != 0u && t0 % P.tiles_x == 0u && P.ntiles % P.tiles_x == 0u) ? 1u : 0u;
        P.q_tiles_y = P.q_order ? P.ntiles / P.tiles_x : 0u;
        P.q_blocks_x = P.q_order ? (P.tiles_x + 31u) / 32u : 0u;
        P.q_seq_len = P.q_order ? P.q_blocks_x * ((P.q_tiles_y + 31u) / 32u) * 1024u : P.ntiles;
        P.q_units = (P.q_seq_len + P.q_unit_tiles - 1u) / P.q_unit_tiles;
        for (uint32_t d = 0; d < levels; ++d) { P.q_rays[d] = K.q[d]; P.q_out[d] = K.out[d]; P.q_spec[d] = K.spec[d]; P.q_child[d] = K.child[d]; }
        P.stash = c.stash.p; P.frame_threads = threads;
        if (ldss) {
            P.lds_image = a.lds_image.p; P.lds_image_n16 = a.lds_image_n16;
            P.lds_node_off = a.lds_node_off; P.lds_prim_off = a.lds_prim_off; P.lds_soup_off = a.lds_soup_off; P.lds_accel_off = a.lds_accel_off;
        }
        const uint32_t blocks = ldss ? blocks_cap : std::min(blocks_cap, (P.q_units + 3u) / 4u);
        const size_t nready_now = nready;
        for (uint32_t sidx = 0; sidx < nsamples / (S / parts); ++sidx) {
            P.sample_index = sidx;
            HIP_TRY(hipMemsetAsync(c.wf_counters.p, 0, (QC_WORDS + (levels > 1 ? nready_now : 0)) * sizeof(uint32_t), stream));
            timed(4, [&] { return launch_queue(P, blocks, stream); });
            for (uint32_t d = levels - 1; d-- > 0;) { // bottom-up: li of level d's rays from their children's (integrate.rs:79, 103, 129)
                P.wf_level = d;
                P.wf_cap = (unsigned long long)n0 << d; P.wf_cap_next = (unsigned long long)n0 << (d + 1);
                P.wf_out = K.out[d]; P.wf_spec = K.spec[d]; P.wf_child = K.child[d]; P.wf_out_next = K.out[d + 1];
                const uint32_t flat_blocks0 = (uint32_t)(((unsigned long long)P.ntiles * 64ull + 255ull) / 256ull);
                timed(1, [&] { return launch_wf_combine(P, d == 0 ? flat_blocks0 : flat_cap, stream); });
            }
        }
        if (S / parts > 1) { // a pixel's samples summed in their order (k_wavefront.hip, wf_resolve_kernel)
            DParams R = P;
            R.ntiles = pixel_tiles;
            timed(1, [&] { return launch_wf_resolve(R, (uint32_t)(((unsigned long long)pixel_tiles * 64ull + 255ull) / 256ull), stream); });
        }
    }
    if (a.profiling) { HIP_TRY(hipEventRecord(e1, stream)); a.events.emplace_back(e0, e1); }
}

// ---- which organisation renders a launch -----------------------------------------------------------------
enum Org : int { ORG_MEGA = 0, ORG_WAVEFRONT = 1, ORG_QUEUE = 2 };
static uint32_t levels_of(const lg_accel &a, const DParams &P) { return (a.flat.has_specular && P.recursion > 0) ? P.recursion + 1u : 1u; }
// what each organisation can take: the queue organisation the reference traversal with <= 32 lights and <= 8 recursion levels, the
// level-by-level pipeline any scene with <= 32 lights; neither the counting variant
static bool org_possible(const lg_accel &a, const DParams &P, bool stats, Org org) {
    if (org == ORG_QUEUE) return !stats && !a.fast && P.nlights <= 32 && levels_of(a, P) <= QC_MAX_LEVELS;
    if (org == ORG_WAVEFRONT) return !stats && P.nlights <= 32 && P.recursion < 20;
    return true;
}
// The FITTED rule of rounds 2-4 (primitive count, glass / mirror, pixels per launch, samples per pixel): what a launch gets when
// nothing has been measured for its kind -- LASGUN_AUTOTUNE=0, or the first candidate the measurement below starts from.
//   * queue organisation: glass / mirror over a big mesh (long uneven walks, sparse deep levels), launches of 2^16 pixels and more;
//   * level by level, for a scene resident in LDS (round 4, once a launch no longer ended in 75-90 us of failed tile claims): glass /
//     mirror frames up to 2^20 pixels (Cornell glass 0.41 against 0.81 ms at 512^2, 0.99 / 1.18 at 1024^2, 1.81 / 1.64 at 1536^2), and
//     frames from 2^18 pixels of few primitives at one sample per pixel (README sphere 2.2 / 3.5 ms at 4096^2); and wherever node and
//     sphere tests dominate (>= 512 spheres / boxes) from 2^21 pixels;
//   * the megakernel otherwise (supersampled frames of small scenes: its 768-lane form is ahead at every size).
static Org org_by_rule(const lg_accel &a, const DParams &P, bool stats) {
    const unsigned long long items = (unsigned long long)P.ntiles * 64ull;
    if (org_possible(a, P, stats, ORG_QUEUE) && a.streaming && a.queue_default && items >= a.queue_min_items) return ORG_QUEUE;
    const bool lds_resident = !a.fast && a.lds_scene && a.ldss_blocks, specular = a.flat.has_specular && P.recursion > 0;
    // (level 0's work items: with a pixel's samples side by side -- round 5 -- a 9-sample frame is nine times as wide as its film;
    // profiles/r05_ss_par.jsonl: Cornell glass at 9 spp goes level by level at 256^2 and in the megakernel from 512^2, like its
    // one-sample frames of nine times the pixels; simple.rs at 16 spp level by level at every size)
    const unsigned long long work = items * (a.sample_order != 1 ? (unsigned long long)P.ss_root * P.ss_root : 1ull);
    const bool small_specular = lds_resident && specular && work <= a.specular_small_items;
    const bool light_scene = lds_resident && !specular && !a.streaming_pays && (P.ss_root == 1u || a.sample_order != 1) && work >= (1ull << 18);
    if (a.streaming && org_possible(a, P, stats, ORG_WAVEFRONT) && (small_specular || light_scene || (a.streaming_pays && work >= a.streaming_min_items)))
        return ORG_WAVEFRONT;
    return ORG_MEGA;
}

// the megakernel (k_mega.hip): the whole of li() per lane
// The megakernel with a pixel's samples SIDE BY SIDE (DParams::ss_par, enqueue_wavefront): the launch hands out (tile, sample) pairs,
// so that a wave's share is 1 / samples of what it was and the launch's tail with it; the samples are parked (24 bytes each) and summed
// in their order by the resolve pass.  Measured (tools/ss_probe.py, profiles/r05_ss_par.jsonl): 4- and 9-sample frames of 256^2 .. 1024^2
// 1.2 - 10 x faster (a 512^2 film is one tile per wave of the grid: nine samples in a row on each, or nine times the tiles); frames of
// 1024^2 and more of a cheap scene 30-50 % SLOWER (nine times the claims on one head word, 8 ns each).  So: possible while the parked
// samples fit 1 GiB, the rule below where nothing is measured, and one more thing the measured choice times.
constexpr unsigned long long SS_PAR_WAVES = 8; // the rule: side by side below this many pixel tiles per wave of the grid (9 of 12 scenes faster at 1024^2, none at 2048^2)
static bool mega_par_possible(const DParams &P, bool stats) {
    const unsigned long long nsamples = (unsigned long long)P.ss_root * P.ss_root;
    return nsamples > 1 && !stats && (unsigned long long)P.ntiles * 64ull * nsamples * 24ull <= (1ull << 30);
}
static bool mega_par_by_rule(const lg_accel &a, const DParams &P, bool stats) {
    static const int ss_mega = [] { const char *e = std::getenv("LASGUN_SS_MEGA"); return e ? std::atoi(e) : -1; }(); // A/B: 0 never, 1 always
    if (!mega_par_possible(P, stats) || a.sample_order == 1) return false;
    if (a.sample_order == 0) return true;
    const bool lds_form = !a.fast && a.lds_scene && a.ldss_blocks;
    const unsigned long long grid_waves = lds_form ? (unsigned long long)a.ldss_blocks * (a.mega_narrow ? 12u : 16u) : (unsigned long long)(a.fast ? a.max_blocks_fast : a.max_blocks) * 4ull;
    return ss_mega >= 0 ? ss_mega == 1 : P.ntiles < SS_PAR_WAVES * grid_waves;
}
// A SMALL launch may hand its tiles out in QUARTERS (DParams::split_shift: 16 lanes of a tile per wave, four times the waves at work): a frame of
// fewer tiles than the grid has waves is as slow as its slowest tile's recursion tree, and a quarter of a tile is a shorter tree walked by
// fewer diverging lanes.  Measured (tools/split_probe.py, profiles/r05_ab_split.jsonl): the kitchen sink at 512^2 2.40 -> 1.73 ms, the
// 100k-triangle metal torus at 256^2 2.23 -> 1.82; cheap scenes and frames of 1024^2 and more lose (idle lanes are then lost throughput).
// One more candidate of the measured choice; never by rule.
static bool mega_split_possible(const lg_accel &a, const DParams &P, bool stats) {
    const bool lds_form = !a.fast && a.lds_scene && a.ldss_blocks;
    const unsigned long long grid_waves = lds_form ? (unsigned long long)a.ldss_blocks * (a.mega_narrow ? 12u : 16u) : (unsigned long long)(a.fast ? a.max_blocks_fast : a.max_blocks) * 4ull;
    const unsigned long long work = (unsigned long long)P.ntiles * (mega_par_by_rule(a, P, stats) ? (unsigned long long)P.ss_root * P.ss_root : 1ull);
    return !stats && P.ntiles >= 2u && work <= 4ull * grid_waves;
}
static void enqueue_mega(const lg_accel &a, DParams &P, lg_accel::LaunchCtx &c, bool par, bool split, bool stats, hipStream_t stream) {
    const uint32_t nsamples = P.ss_root * P.ss_root;
    par = par && mega_par_possible(P, stats);
    if (par) {
        const size_t n_items = (size_t)P.ntiles * 64ull * nsamples, need = n_items * 3 * 8;
        if (c.wf_mem.n < need) { HIP_TRY(hipDeviceSynchronize()); c.wf_mem.alloc(need); }
        P.accum = reinterpret_cast<double *>(c.wf_mem.p); P.n_items = n_items; P.ss_par = nsamples; P.ntiles *= nsamples;
    }
    { // tiles handed out in parts: the measured choice's candidate, or LASGUN_MEGA_SPLIT=2|4|8 (A/B)
        static const uint32_t split_env = [] { const char *e = std::getenv("LASGUN_MEGA_SPLIT"); const int v = e ? std::atoi(e) : 0; return v == 2 || v == 4 || v == 8 ? (uint32_t)v : 1u; }();
        const uint32_t parts = a.tile_parts >= 1 ? (uint32_t)a.tile_parts : split_env > 1u ? split_env : split ? MEGA_SPLIT : 1u;
        if (parts > 1u && !stats && (unsigned long long)P.ntiles * parts < (1ull << 31)) { P.split_shift = parts == 2u ? 1u : parts == 4u ? 2u : 3u; P.ntiles *= parts; }
    }
    uint32_t cap = a.fast ? a.max_blocks_fast : a.max_blocks;
    uint32_t blocks = (P.ntiles + 3u) / 4u;
    if (blocks > cap) blocks = cap;
    uint32_t maxb = a.max_blocks > a.max_blocks_fast ? a.max_blocks : a.max_blocks_fast;
    if (!stats && !a.fast && a.lds_scene && a.ldss_blocks) { // scene tables resident in LDS: one 1024-lane workgroup per CU
        P.lds_image = a.lds_image.p; P.lds_image_n16 = a.lds_image_n16;
        P.lds_node_off = a.lds_node_off; P.lds_prim_off = a.lds_prim_off; P.lds_soup_off = a.lds_soup_off; P.lds_accel_off = a.lds_accel_off;
        P.mega_lanes = a.mega_narrow ? 768u : 1024u; // (k_mega.hip: three waves per SIMD and 168 registers where shading weighs more than walking)
        blocks = a.ldss_blocks;
    }
    if (maxb < a.ldss_blocks * 4u) maxb = a.ldss_blocks * 4u; // per-lane slots below: 1024 lanes per LDS-scene workgroup
    // Whitted frames: one slot per resident lane and recursion level, only for glass / mirror scenes
    if (a.flat.has_specular && P.recursion > 0) {
        unsigned long long threads = (unsigned long long)maxb * 256ull;
        size_t need = (size_t)threads * P.recursion * FRAME_DOUBLES;
        if (c.frames.n < need) {
            HIP_TRY(hipDeviceSynchronize()); // (re)allocation: nothing may still use the old buffer
            c.frames.alloc(need);
        }
        P.frames = c.frames.p;
        P.frame_threads = threads;
    }
    if (P.nlights > 0) { // shading frame parked across the shadow traversals
        unsigned long long threads = (unsigned long long)maxb * 256ull;
        size_t need = (size_t)threads * STASH_DOUBLES;
        if (c.stash.n < need) {
            HIP_TRY(hipDeviceSynchronize()); // (re)allocation: nothing may still use the old buffer
            c.stash.alloc(need);
        }
        P.stash = c.stash.p;
        P.frame_threads = threads;
    }
    if (stats) {
        P.stats = a.stats.p;
        HIP_TRY(hipMemsetAsync(a.stats.p, 0, sizeof(DStats), stream));
    }
    HIP_TRY(hipMemsetAsync(c.tile_counter.p, 0, TILE_COUNTER_WORDS * sizeof(uint32_t), stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (a.profiling) {
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, stream));
    }
#ifdef LG_QIDLE // diagnostic build: the waves' start / exit times of this launch (k_mega.hip), read by lg_debug_stats
    if (!stats) { P.stats = a.stats.p; HIP_TRY(hipMemsetAsync(a.stats.p, 0, sizeof(DStats), stream)); }
#endif
    HIP_TRY(launch_trace(P, stats, a.fast, blocks, a.fast ? a.stack_depth_fast1 : a.stack_depth, stream));
    if (par) {
        DParams R = P;
        R.ntiles = (P.ntiles >> P.split_shift) / nsamples; R.tile_rev = 0u; R.split_shift = 0u;
        HIP_TRY(launch_wf_resolve(R, (uint32_t)(((unsigned long long)R.ntiles * 64ull + 255ull) / 256ull), stream));
    }
    if (a.profiling) {
        HIP_TRY(hipEventRecord(e1, stream));
        a.events.emplace_back(e0, e1);
    }
}
static void enqueue_org(const lg_accel &a, DParams P, lg_accel::LaunchCtx &c, Org org, int dir, bool ss_serial, bool split, bool stats, hipStream_t stream) { // (P by value: an organisation fills in its own fields)
    P.tile_counter = c.tile_counter.p;
    P.tile_rev = org != ORG_WAVEFRONT ? (uint32_t)dir : 0u; // 0 top-down, 1 bottom-up, 2 from the middle outwards (the level-by-level passes are short and alike: one direction)
    if (org == ORG_QUEUE) enqueue_queue(a, P, c, split, stream);
    else if (org == ORG_WAVEFRONT) enqueue_wavefront(a, P, c, stream);
    else enqueue_mega(a, P, c, !ss_serial, split, stats, stream);
}

// The MEASURED choice (round 5; the rule above was a fit to eight scenes and wrong by 6-22 % on the first scene that was not among
// them; round 6: the table and the race live in tune.cpp, this file supplies the kind, the candidates and how one is launched).
// Every organisation renders the same bytes, so which one runs is a question of time alone, and the answer is taken from the clock: the
// second API call that launches a KIND in the process (the first gets the rule's choice at no cost) -- the scene's shape (table sizes,
// materials, lights, recursion, samples per pixel, traversal mode, LDS residency), the device, the launch's size class (log2 of its pixels)
// and addressing mode -- renders the launch with every CANDIDATE that can take it (a warm-up pass, then three timed passes over the
// candidates in turn, HIP events on the caller's stream, the HOST WAITING -- the one place where a *_device entry point blocks; never on a
// stream that is being captured), keeps the fastest (the rule's own choice unless another beats it by 1 %) and remembers it for
// the process: capture() rebuilds its accel for every frame (lib.rs:64), so the memory is keyed by the scene's shape, not by the accel.
// A candidate that cannot run (no memory for its buffers) drops out of the race instead of failing the caller's render.
// A candidate is an organisation and, for the megakernel and the queue organisation, the DIRECTION the launch's tiles are claimed in:
// a launch ends with the recursion trees of its last tiles, and whether the film's top or its bottom should come last is the scene's
// and the camera's business -- simple.rs at 9 spp and the metal torus gain 6-9 % from the bottom up, the glass torus loses 3 %
// (profiles/r05_ab_tile_order.jsonl); which tile is rendered when never changes a pixel.  (The kind does not know the camera: a
// direction measured for one view is kept for the next.)  The launch itself is then enqueued as usual; what the measurement rendered
// into the caller's film are the same pixels.  Overridden by lg_accel_set_streaming(0 / 2 / 3) and lg_accel_set_tile_order
// (lg_accel_last_organisation says what a launch ran as); LASGUN_AUTOTUNE=0 keeps the rule and the middle-out direction;
// lg_tune_export / lg_tune_import / lg_tune_clear read, pin and forget choices.
namespace {
using TuneKey = lg::tune::Key;
constexpr int TUNE_REV = 16;    // a remembered choice: organisation | TUNE_REV when the tiles go bottom-up
constexpr int TUNE_MID = 64;    //   | TUNE_MID when they go from the middle row outwards
static int dir_bits(int dir) { return dir == 1 ? TUNE_REV : dir == 2 ? TUNE_MID : 0; }
static int dir_of(int choice) { return (choice & TUNE_REV) ? 1 : (choice & TUNE_MID) ? 2 : 0; }
// The direction a launch's tiles are claimed in when nothing is forced or measured: from the middle row outwards.  What a frame shows
// tends to sit in its middle, and a launch should END on cheap tiles: config 4 in the megakernel 36.2 -> 32.8 ms, 4m 13.1 -> 12.7,
// simple.rs 0.55 -> 0.53, nothing slower among the configs (profiles/r05_ab_tile_middle.jsonl).
constexpr int DIR_DEFAULT = 2;
static int dir_unmeasured(const lg_accel &a, Org org) { return org == ORG_WAVEFRONT ? 0 : a.tile_order >= 0 ? a.tile_order : DIR_DEFAULT; }
constexpr int TUNE_SPLIT = 128; //   | TUNE_SPLIT when the megakernel hands a small launch's tiles out in quarters (enqueue_mega)
constexpr int TUNE_SERIAL = 32; //   | TUNE_SERIAL when the megakernel takes a pixel's samples one after the other (enqueue_mega)
// LASGUN_AUTOTUNE (tune.cpp: mode()): 0 = never measure (the fitted rule), 1 (default) = measure a kind at the second API CALL that
// launches it, 2 = at the first.  A program that renders one frame and exits (every example of the reference) gets the rule's choice at no
// cost -- timing seven candidates three times over costs 30-50 frames' worth; whatever renders a kind twice (an animation, the progressive
// front end's hundred subsets, a benchmark) is measured from then on.  What counts is the CALL, not the launch: one lg_capture of a big
// film launches its kind four times (row bands), lg_multi_* once per share of a device (round 5 counted launches and measured inside
// the first frame: ADVICE r5).
bool autotune_enabled() { return lg::tune::mode() != 0; }
// Which API call is running: bumped when a call enters the library from outside (CallScope in guarded(), lg_capture, lg_multi_*:
// calls nested in it, on this thread or on the threads it starts, belong to it).
std::atomic<uint64_t> g_call_serial{1};
std::atomic<int> g_call_depth{0};
} // namespace
CallScope::CallScope() { if (g_call_depth.fetch_add(1) == 0) g_call_serial.fetch_add(1); }
CallScope::~CallScope() { g_call_depth.fetch_sub(1); }
namespace {
} // namespace
extern "C" void lg_internal_call_scope(int enter) { // (multi.cpp: one lg_multi_capture* is one call, whatever its shares launch)
    if (enter) { if (g_call_depth.fetch_add(1) == 0) g_call_serial.fetch_add(1); }
    else g_call_depth.fetch_sub(1);
}
static TuneKey tune_key(const lg_accel &a, const DParams &P) {
    const FlatScene &f = a.flat;
    const unsigned long long items = (unsigned long long)P.ntiles * 64ull;
    uint64_t cls = 0;
    while ((items >> cls) > 1ull) ++cls;
    TuneKey k{};
    k.v[0] = f.nodes.size(); k.v[1] = f.primref.size(); k.v[2] = f.spheres.size(); k.v[3] = f.cuboids.size(); k.v[4] = f.tri_v.size();
    k.v[5] = f.accels.size(); k.v[6] = ((uint64_t)f.max_stack << 32) | (uint64_t)f.lights.size();
    k.v[7] = ((uint64_t)P.recursion << 32) | ((uint64_t)P.ss_root << 8) | (f.has_specular ? 1u : 0u);
    k.v[8] = ((uint64_t)P.prune << 2) | (a.fast ? 2u : 0u) | (a.lds_scene && a.ldss_blocks ? 1u : 0u);
    {   // what the primitives are made of decides how many rays have children: two scenes of one shape (config 4's glass torus, 4m's metal one) are two kinds
        uint64_t hsh = 1469598103934665603ull;
        auto mix = [&hsh](const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; ++i) { hsh ^= b[i]; hsh *= 1099511628211ull; } };
        for (const DMaterial &m : f.materials) mix(&m.kind, sizeof m.kind);
        for (const DAccel &A : f.accels) { mix(&A.material, sizeof A.material); mix(&A.flags, sizeof A.flags); }
        if (!f.sphere_mat.empty()) mix(f.sphere_mat.data(), f.sphere_mat.size() * sizeof f.sphere_mat[0]);
        if (!f.cuboid_mat.empty()) mix(f.cuboid_mat.data(), f.cuboid_mat.size() * sizeof f.cuboid_mat[0]);
        k.v[9] = hsh ^ ((uint64_t)a.device << 56);
    }
    k.v[10] = cls;
    k.v[11] = (P.mode == 0u ? 0u : 1u) | (a.tile_order >= 0 ? 2u + (uint64_t)a.tile_order : 0u) | ((uint64_t)(a.sample_order + 1) << 4) | ((uint64_t)(a.tile_parts + 1) << 8); // (a forced direction is a kind of its own: only the organisations race)
    return k;
}
// a remembered choice (measured here, or pinned by lg_tune_import for a kind this build may see differently) that the launch cannot take
// falls back to the rule's
static int rule_choice(const lg_accel &a, const DParams &P) {
    const Org rule = org_by_rule(a, P, false);
    return (int)rule | dir_bits(P.ntiles < 2u ? 0 : dir_unmeasured(a, rule)) | (rule == ORG_MEGA && !mega_par_by_rule(a, P, false) ? TUNE_SERIAL : 0);
}
static bool choice_possible(const lg_accel &a, const DParams &P, int choice) {
    const int org = choice & (TUNE_REV - 1);
    if (org < 0 || org > (int)ORG_QUEUE || !org_possible(a, P, false, (Org)org)) return false;
    if ((choice & TUNE_SPLIT) && !mega_split_possible(a, P, false)) return false;
    if (org == (int)ORG_MEGA && !(choice & TUNE_SERIAL) && P.ss_root > 1 && !mega_par_possible(P, false)) return false;
    return true;
}
static int tuned_choice(const lg_accel &a, const DParams &P, lg_accel::LaunchCtx &c, hipStream_t stream) {
    const Org rule = org_by_rule(a, P, false);
    const TuneKey key = tune_key(a, P);
    int known;
    if (lg::tune::lookup(key, &known)) return choice_possible(a, P, known) ? known : rule_choice(a, P);
    if (lg::tune::mode() == 1 && lg::tune::first_call_of_kind(key, g_call_serial.load())) return rule_choice(a, P); // the first call that launches the kind: the rule's choice, at no cost
    {   // a stream that is being captured into a graph cannot be waited on: no race there (the next plain launch of the kind measures)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
        else if (cs != hipStreamCaptureStatusNone) return rule_choice(a, P);
    }
    // candidates: [organisation][samples side by side, one after the other (megakernel only)][top-down, bottom-up, middle-out]
    constexpr int NC = 20, K_SPLIT = 18, K_QSPLIT = 19; // (+ the megakernel / the queue organisation with their tiles in quarters: sample order by the rule, middle-out)
    lg::tune::Candidate cand[NC];
    const unsigned long long items = (unsigned long long)P.ntiles * 64ull;
    for (int k = 0; k < NC; ++k) {
        if (k == K_SPLIT) {
            cand[k].choice = (int)ORG_MEGA | dir_bits(DIR_DEFAULT) | (!mega_par_by_rule(a, P, false) ? TUNE_SERIAL : 0) | TUNE_SPLIT;
            cand[k].in_race = mega_split_possible(a, P, false) && a.tile_parts < 0 && (a.tile_order < 0 || a.tile_order == DIR_DEFAULT);
            continue;
        }
        if (k == K_QSPLIT) {
            cand[k].choice = (int)ORG_QUEUE | dir_bits(DIR_DEFAULT) | TUNE_SPLIT;
            cand[k].in_race = org_possible(a, P, false, ORG_QUEUE) && items >= 4096ull && mega_split_possible(a, P, false) && a.tile_parts < 0 && (a.tile_order < 0 || a.tile_order == DIR_DEFAULT);
            continue;
        }
        const int org = k / 6, ser = (k / 3) & 1, dir = k % 3;
        cand[k].choice = org | dir_bits(dir) | (ser ? TUNE_SERIAL : 0);
        cand[k].in_race = org_possible(a, P, false, (Org)org) &&
                     (ser ? org == ORG_MEGA : (org != ORG_MEGA || mega_par_possible(P, false))) &&
                     !(org == ORG_MEGA && mega_par_possible(P, false) && a.sample_order >= 0 && ser != a.sample_order) && // (lg_accel_set_sample_order) // (one form of the megakernel for a frame of one sample per pixel: the serial one)
                     !(org == ORG_QUEUE && items < 4096ull && rule != ORG_QUEUE) && // (a persistent scheduler for a handful of tiles: never ahead)
                     !(dir != 0 && (org == ORG_WAVEFRONT || P.ntiles < 2u)) &&      // (one direction for the level-by-level passes and for a single tile)
                     (a.tile_order < 0 || org == ORG_WAVEFRONT || dir == a.tile_order); // (lg_accel_set_tile_order: only the organisations race)
    }
    const int rule_k = (int)rule * 6 + (rule == ORG_MEGA && !mega_par_by_rule(a, P, false) ? 3 : 0) + (P.ntiles < 2u ? 0 : dir_unmeasured(a, rule));
    float best_ms[NC];
    for (float &m : best_ms) m = INFINITY;
    const bool was_profiling = a.profiling;
    a.profiling = false; // (the measurement's launches are not the caller's: lg_profile_read must not count them)
    struct Restore { const lg_accel &a; bool was; ~Restore() { a.profiling = was; } } restore{a, was_profiling};
    // (LASGUN_TUNE_FAIL=<organisation 0..2>: test hook -- every candidate of that organisation throws in the race, as one whose buffers do not fit would)
    static const int fail_org = [] { const char *e = std::getenv("LASGUN_TUNE_FAIL"); return e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : -1; }();
    const int choice = lg::tune::race(key, cand, NC, rule_k, stream, [&](int k) {
        if (fail_org >= 0 && (cand[k].choice & (TUNE_REV - 1)) == fail_org) throw Error("LASGUN_TUNE_FAIL: injected failure of a candidate");
        if (k == K_SPLIT) enqueue_org(a, P, c, ORG_MEGA, DIR_DEFAULT, !mega_par_by_rule(a, P, false), true, false, stream);
        else if (k == K_QSPLIT) enqueue_org(a, P, c, ORG_QUEUE, DIR_DEFAULT, false, true, false, stream);
        else enqueue_org(a, P, c, (Org)(k / 6), k % 3, ((k / 3) & 1) != 0, false, false, stream);
    }, best_ms);
    check_queue_error(a);
    if (std::getenv("LASGUN_DEBUG")) {
        std::fprintf(stderr, "[lasgun] measured for %llu pixels (top-down / bottom-up / middle-out): megakernel %.3f / %.3f / %.3f ms (samples in a row: %.3f / %.3f / %.3f), level by level %.3f ms, queue %.3f / %.3f / %.3f ms -> choice %d (rule: %d)\n",
                     items, best_ms[0], best_ms[1], best_ms[2], best_ms[3], best_ms[4], best_ms[5], best_ms[6], best_ms[12], best_ms[13], best_ms[14], // (one sample per pixel: "in a row" is the megakernel)
                     choice, (int)rule);
        if (std::isfinite(best_ms[K_SPLIT]) || std::isfinite(best_ms[K_QSPLIT])) std::fprintf(stderr, "[lasgun]   (tiles in quarters: megakernel %.3f ms, queue %.3f ms)\n", best_ms[K_SPLIT], best_ms[K_QSPLIT]);
    }
    return choice_possible(a, P, choice) ? choice : rule_choice(a, P);
}

// Enqueue one render on `stream`.  Caller holds a.mtx.
void enqueue(const lg_accel &a, DParams &P, bool stats, hipStream_t stream) {
    if (P.ntiles == 0) return;
    check_queue_error(a); // (an earlier launch on a caller's stream that stalled: reported here at the latest)
    lg_accel::LaunchCtx &c = ctx_for(a, stream);
    Org org;
    int dir = -1; // lg_accel_set_tile_order; -1: from the middle outwards unless measured otherwise (dir_unmeasured)
    bool ss_serial = !mega_par_by_rule(a, P, stats); // the megakernel's samples: by the rule unless measured
    bool split = false;                              // its tiles in quarters: only as measured
    if (stats) { org = ORG_MEGA; dir = 0; }                                                      // the counting variant
    else if (a.queue == 1) org = org_possible(a, P, stats, ORG_QUEUE) ? ORG_QUEUE : org_by_rule(a, P, stats); // lg_accel_set_streaming(3)
    else if (!a.streaming) org = ORG_MEGA;                                                       // lg_accel_set_streaming(0)
    else if (a.streaming_forced) org = org_possible(a, P, stats, ORG_WAVEFRONT) ? ORG_WAVEFRONT : ORG_MEGA; // lg_accel_set_streaming(2)
    else if (a.queue == 0 || !autotune_enabled()) {                                              // the fitted rule (queue ruled out by set_streaming(0 .. 2))
        org = org_by_rule(a, P, stats);
        if (a.queue == 0 && org == ORG_QUEUE) org = ORG_MEGA;
    } else {
        const int choice = tuned_choice(a, P, c, stream);
        org = (Org)(choice & (TUNE_REV - 1));
        dir = dir_of(choice);
        ss_serial = (choice & TUNE_SERIAL) != 0;
        split = (choice & TUNE_SPLIT) != 0;
    }
    if (dir < 0) dir = P.ntiles < 2u ? 0 : dir_unmeasured(a, org);
    if (org == ORG_WAVEFRONT) dir = 0;
    a.last_org = (int)org | dir_bits(dir) | (org == ORG_MEGA && ss_serial && P.ss_root > 1 ? TUNE_SERIAL : 0) | (org != ORG_WAVEFRONT && (split || a.tile_parts > 1) ? TUNE_SPLIT : 0);
    enqueue_org(a, P, c, org, dir, ss_serial, split, stats, stream);
}

void set_rect(DParams &P, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
    P.mode = 0; P.x0 = x0; P.y0 = y0; P.x1 = x1; P.y1 = y1;
    P.tiles_x = (x1 - x0 + 7u) / 8u;
    uint32_t tiles_y = (y1 - y0 + 7u) / 8u;
    P.ntiles = P.tiles_x * tiles_y;
    P.ilv_n = 1; P.ilv_r = 0; P.ilv_b = 1;
    P.out_x0 = 0; P.out_pitch = P.w;
}
// the row table of the lattice addressing for (w, h, n): made once per launch context and kept while the caller stays with that film and period
// (the progressive front end's hundred calls share it)
constexpr size_t MAX_ROW_TABLES = 4;
static const DRowTab *lattice_rows(const lg_accel &a, hipStream_t stream, uint32_t w, uint32_t h, unsigned long long n) {
    lg_accel::LaunchCtx &c = ctx_for(a, stream);
    using RowTable = lg_accel::LaunchCtx::RowTable;
    for (auto &r : c.rowtabs)
        if (r->w == w && r->h == h && r->n == n) { r->last_use = ++c.rowtab_clock; return r->buf.p; }
    RowTable *r = nullptr;
    if (c.rowtabs.size() < MAX_ROW_TABLES) {
        c.rowtabs.emplace_back(new RowTable());
        r = c.rowtabs.back().get();
    } else { // the least recently used table makes room: launches that read it are ahead of the new copy in stream order, unless its buffer must grow
        r = c.rowtabs[0].get();
        for (auto &x : c.rowtabs) if (x->last_use < r->last_use) r = x.get();
        if (r->up) HIP_TRY(hipEventSynchronize(r->up)); // (its staging is rewritten below)
        if (r->buf.n < h) HIP_TRY(hipStreamSynchronize(stream)); // (a buffer goes back to the pool only when nothing can still read it)
    }
    r->w = 0; r->h = 0; r->n = 0; // (not a table of anything until the copy below is enqueued)
    if (r->buf.n < h) r->buf.alloc(h);
    r->stage.need((size_t)h * sizeof(DRowTab));
    DRowTab *t = static_cast<DRowTab *>(r->stage.p);
    for (uint32_t y = 0; y < h; ++y) { const unsigned long long o = (unsigned long long)y * w; t[y] = DRowTab{(uint32_t)(o / n), (uint32_t)(o % n)}; }
    HIP_TRY(hipMemcpyAsync(r->buf.p, t, (size_t)h * sizeof(DRowTab), hipMemcpyHostToDevice, stream));
    if (!r->up) HIP_TRY(hipEventCreateWithFlags(&r->up, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(r->up, stream));
    r->w = w; r->h = h; r->n = n; r->last_use = ++c.rowtab_clock;
    return r->buf.p;
}
// the pixels of the subset {k + i*n} of an `area`-pixel film (k < area, n > 0), without forming area - k + n - 1 (which wraps for n near SIZE_MAX)
unsigned long long subset_count(unsigned long long area, unsigned long long k, unsigned long long n) { return k < area ? 1ull + (area - 1ull - k) / n : 0ull; }
void set_subset(const lg_accel &a, hipStream_t stream, DParams &P, size_t k, size_t n, uint32_t w, uint32_t h) {
    unsigned long long area = (unsigned long long)w * h;
    P.mode = 1; P.sub_k = k; P.sub_n = n;
    P.sub_count = subset_count(area, k, n);
    P.ntiles = (uint32_t)((P.sub_count + 63ull) / 64ull);
    // The subset tile by lattice column (mode 4, shade.h: 64 rows x <= n pixels per tile instead of 64 consecutive i) where that is the denser
    // window: a period shorter than the film's width and longer than a tile's 64 pixels in a row would be.  LASGUN_SUBSET_LATTICE=0: never (A/B).
    static const bool lattice = [] { const char *e = std::getenv("LASGUN_SUBSET_LATTICE"); return !(e && e[0] == '0'); }();
    const unsigned long long cols = (w + n - 1) / n, tiles4 = ((unsigned long long)h + 63ull) / 64ull * cols;
    if (lattice && P.sub_count != 0 && n >= 8 && n <= w && h >= 16 && area < (1ull << 32) && tiles4 < (1ull << 31) && tiles4 <= 2ull * P.ntiles + 8ull) {
        P.mode = 4; P.sub_cols = (uint32_t)cols; P.sub_rows = 64u; P.ntiles = (uint32_t)tiles4;
        P.sub_kk = (uint32_t)(k % n); P.sub_kdiv = (uint32_t)(k / n);
        P.sub_rowtab = lattice_rows(a, stream, w, h, n);
    }
}

SubsetBatch make_batch(const size_t *ks, size_t count, size_t n, uint32_t w, uint32_t h) {
    if (n == 0) throw Error("n must be > 0");
    if (count != 0 && !ks) throw Error("ks is NULL");
    const unsigned long long area = (unsigned long long)w * h;
    SubsetBatch b;
    b.n = n;
    for (size_t j = 0; j < count; ++j) if (ks[j] < area) b.ks.push_back(ks[j]); // (a subset that starts behind the film has no pixel: lib.rs:152)
    std::sort(b.ks.begin(), b.ks.end());
    b.ks.erase(std::unique(b.ks.begin(), b.ks.end()), b.ks.end());
    for (unsigned long long k : b.ks) b.periods = std::max(b.periods, subset_count(area, k, n));
    b.items = (unsigned long long)b.ks.size() * b.periods;
    if (b.items >= 0xFFFFFFFFull) throw Error("too many pixels for one batch of subsets (2^32 work items)");
    b.whole = b.ks.size() == n;
    for (size_t j = 0; b.whole && j < b.ks.size(); ++j) b.whole = b.ks[j] == j;
    return b;
}
// the batch as addressing mode 3 on `stream` (its k table lives in the stream's launch context).  Caller holds a.mtx.
void set_subsets(const lg_accel &a, DParams &P, const SubsetBatch &b, hipStream_t stream) {
    lg_accel::LaunchCtx &c = ctx_for(a, stream);
    for (size_t i = 0; i < c.ks_live.size();) { // tables whose launch is through go back to the pool
        if (c.ks_live[i]->done && hipEventQuery(c.ks_live[i]->done) == hipSuccess) {
            (void)hipEventDestroy(c.ks_live[i]->done);
            c.ks_live.erase(c.ks_live.begin() + (long)i);
        } else ++i;
    }
    (void)hipGetLastError(); // (hipEventQuery's "not ready" is not an error of this call)
    // (a table whose launch was never enqueued -- the call failed between set_subsets and subsets_enqueued -- has no event to wait for:
    // subsets_abandoned() below takes it out again on that path)
    c.ks_live.emplace_back(new lg_accel::LaunchCtx::KsTable());
    lg_accel::LaunchCtx::KsTable &t = *c.ks_live.back();
    const size_t m_ = b.ks.size();
    t.stage.need(2 * m_ * sizeof(unsigned long long));
    unsigned long long *tab = static_cast<unsigned long long *>(t.stage.p); // the m values of k, then (k mod n) | (k / n) << 32 of each (the lattice form, shade.h mode 5)
    for (size_t j = 0; j < m_; ++j) { tab[j] = b.ks[j]; tab[m_ + j] = (b.ks[j] % b.n) | ((b.ks[j] / b.n) << 32); }
    t.buf.alloc(std::max<size_t>(2 * m_, 128));
    HIP_TRY(hipMemcpyAsync(t.buf.p, tab, 2 * m_ * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
    P.mode = 3; P.pixel_list = t.buf.p; P.sub_m = (uint32_t)b.ks.size(); P.sub_n = b.n; P.sub_k = 0; P.sub_count = b.items;
    P.ntiles = (uint32_t)((b.items + 63ull) / 64ull);
    // the batch tile by lattice column (mode 5, shade.h: 64 / m rows x <= n pixels per tile instead of 64 consecutive work items -- 64 / m
    // periods of one row) where that wastes few lanes; LASGUN_SUBSET_LATTICE=0: never (A/B)
    static const bool lattice = [] { const char *e = std::getenv("LASGUN_SUBSET_LATTICE"); return !(e && e[0] == '0'); }();
    const unsigned long long m = b.ks.size(), rows = m != 0 && m <= 64 ? 64ull / m : 0ull, cols = (P.w + b.n - 1) / b.n;
    const unsigned long long tiles5 = rows ? ((unsigned long long)P.h + rows - 1ull) / rows * cols : ~0ull;
    if (lattice && rows >= 2 && b.n >= 8 && b.n <= P.w && (unsigned long long)P.w * P.h < (1ull << 32) && tiles5 < (1ull << 31) && tiles5 * 3ull <= (unsigned long long)P.ntiles * 4ull + 24ull) {
        P.mode = 5; P.sub_cols = (uint32_t)cols; P.sub_rows = (uint32_t)rows; P.ntiles = (uint32_t)tiles5;
        P.sub_rowtab = lattice_rows(a, stream, P.w, P.h, b.n);
    }
}
// ... and once the batch's launch is enqueued: the event that releases its table
void subsets_enqueued(const lg_accel &a, hipStream_t stream) {
    lg_accel::LaunchCtx &c = ctx_for(a, stream);
    if (c.ks_live.empty() || c.ks_live.back()->done) return;
    HIP_TRY(hipEventCreateWithFlags(&c.ks_live.back()->done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c.ks_live.back()->done, stream));
}

// ... and when the call fails before its launch is enqueued: the table set_subsets made has no launch that reads it and no event that
// would ever release it (the copy into it may still be in flight: the stream is drained first)
void subsets_abandoned(const lg_accel &a, hipStream_t stream) {
    for (auto &c : a.ctxs)
        if (c->key == stream && !c->ks_live.empty() && !c->ks_live.back()->done) {
            (void)hipStreamSynchronize(stream);
            (void)hipGetLastError();
            c->ks_live.pop_back();
        }
}
