// lasgun_amd/csrc/accel.cpp -- Accel::from (src/accelerators/bvh.rs:135-202 on the host, host.cpp) as device tables: flatten, the LDS images,
// grids and budgets from the device's occupancy, one upload; and the tables once more with what was left out of them (fast trees, leaf records).
#include "internal.h"

static void build_and_upload(lg_accel *a, bool with_fast) {
    a->ldss_blocks = 0; a->lds_image_n16 = 0; a->fast_available = true;
        static const bool times = std::getenv("LASGUN_DEBUG_TIMES") != nullptr; // (where lg_accel_from's time goes: flatten / upload / derived)
        const auto t_begin = std::chrono::steady_clock::now();
        // The culling records and strips of the pruned walk's mesh leaves are built when that walk will run: by default from PRUNE_MIN_TRIS
        // triangles in a mesh (below: on), when LASGUN_PRUNE=1 or lg_accel_set_prune(1) ask for it (rebuild_tables).
        size_t big_mesh_tris = 0;
        for (const auto &m : a->scene->meshes) if (m && m->tri.size() / 3 > big_mesh_tris) big_mesh_tris = m->tri.size() / 3;
        static const int prune_env = [] { const char *e = std::getenv("LASGUN_PRUNE"); return e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : -1; }();
        const bool with_records = a->prune == 1 || (a->prune < 0 && (prune_env == 1 || (prune_env < 0 && big_mesh_tris >= PRUNE_MIN_TRIS)));
        flatten_scene(*a->scene, a->flat, with_fast, with_records); // host HLBVH build + flatten (throws on what the reference would panic on)
        const auto t_flat = std::chrono::steady_clock::now();
        use_device(a->device);
        const FlatScene &f = a->flat;
        TableStage stage; // (committed at the end: one allocation, one copy)
        stage.add(a->nodes, f.nodes); stage.add(a->nodes4, f.nodes4); stage.add(a->primref, f.primref); stage.add(a->spheres, f.spheres); stage.add(a->sphere_mat, f.sphere_mat);
        stage.add(a->cuboids, f.cuboids); stage.add(a->cuboid_mat, f.cuboid_mat); stage.add(a->tri_v, f.tri_v); stage.add(a->tri_n, f.tri_n);
        stage.add(a->tri_t, f.tri_t); stage.add(a->leaf_soup, f.leaf_soup); stage.add(a->chunks, f.chunks); stage.add(a->strips, f.strips); stage.add(a->sphere_ref_leaf, f.sphere_ref_leaf); stage.add(a->cuboid_ref_leaf, f.cuboid_ref_leaf);
        stage.add(a->tri_ref_leaf, f.tri_ref_leaf); stage.add(a->accel_ref_leaf, f.accel_ref_leaf); stage.add(a->vpos, f.vpos); stage.add(a->vnorm, f.vnorm); stage.add(a->vtex, f.vtex);
        stage.add(a->materials, f.materials); stage.add(a->lights, f.lights); // (the accel records: below, once their compact bases are known)
        {   // the counters' two records, zeroed (the second: iteration counters of the diagnostic build)
            static const std::vector<DStats> zero(2);
            stage.add(a->stats, zero);
        }
        const auto t_up = std::chrono::steady_clock::now();
        a->device_bytes = f.nodes.size() * sizeof(DNode) + f.nodes4.size() * sizeof(DNode4) + f.primref.size() * 4 + f.spheres.size() * sizeof(DSphere) +
                          f.cuboids.size() * sizeof(DCuboid) + f.tri_v.size() * 12 + f.vpos.size() * 4 + f.vnorm.size() * 4 + f.leaf_soup.size() * sizeof(DLeafRec) +
                          f.strips.size() * sizeof(DStrip) + f.chunks.size() * sizeof(DChunk) +
                          f.accels.size() * sizeof(DAccel) + f.materials.size() * sizeof(DMaterial);
        if (!a->stream) a->stream = g_streams.take(a->device);
        // per-lane LDS stack: worst case of this scene graph, +2 guard entries
        a->stack_depth = f.max_stack + 2;
        // the fast kernel falls back to the reference traversal on exact ties, so its stack must hold either
        a->stack_depth_fast1 = (f.max_stack > f.max_stack_fast1 ? f.max_stack : f.max_stack_fast1) + 2;
        const size_t LDS_MAX = 160 * 1024;
        if ((size_t)a->stack_depth * 256 * 4 > LDS_MAX)
            throw Error("BVH too deep for the LDS traversal stack (" + std::to_string(a->stack_depth) + " entries per lane; the reference panics beyond 64 per level, bvh.rs:497)");
        a->fast_available = (size_t)a->stack_depth_fast1 * 256 * 4 <= LDS_MAX;
        // The fast tree's tight boxes are only meaningful if every accel's `minv` (which moves the rays) really is the
        // inverse of its `m` (which moved the boxes).  Transform3::rotate(theta, axis) takes the transpose for the inverse
        // without normalising the axis (transform.rs:144-148), so a non-unit axis gives a pair that is not: the reference
        // still renders *something* through its fat, overlapping leaves, and the reference traversal reproduces that
        // bit for bit, but the fast mode is refused for such a scene.
        for (const DAccel &A : f.accels) {
            double worst = 0.0;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 4; ++c) {
                    double v = (c == 3 ? A.m.c[3][r] : 0.0);
                    for (int k = 0; k < 3; ++k) v += A.m.c[k][r] * (c == 3 ? A.minv.c[3][k] : A.minv.c[c][k]);
                    const double want = (c < 3 && r == c) ? 1.0 : 0.0;
                    if (!(std::fabs(v - want) <= worst)) worst = std::fabs(v - want);
                }
            if (!(worst <= 1e-11)) { // two orders below the 1e-9 the fast tree's boxes are pushed out by
                a->fast_available = false;
                a->fast_refusal = "fast mode unavailable: an aggregate's transform and inverse do not match (rotate() about a non-unit axis?)";
            }
        }
        // The fast mode cannot be made exact for meshes in principle: a ray that lies within rounding of a FAR triangle's
        // plane is accepted by the reference wherever it passes (its fat leaves test every triangle), and a tight tree never visits that
        // triangle.  The band in which that happens is ~ 64 u R^2 / edge wide: negligible for a tessellated surface seen from nearby,
        // not for a mesh whose coordinates dwarf its small triangles (round 4's progression_soup_scene: triangles at 1e9 beside
        // triangles of 0.05 -- 5 wrong pixels in 4,100 scenes).  Such a mesh is refused, like a transform that does not invert.
        for (const auto &m : a->scene->meshes) { // (a question about the fast mode: asked when its trees are built -- every lg_accel_set_mode(1) goes through such a build first)
            if (!m || !with_fast) continue;
            double max_abs = 0.0, min_edge = INFINITY;
            for (float v : m->position) if (std::isfinite(v)) max_abs = std::fmax(max_abs, std::fabs((double)v));
            for (size_t t = 0; t + 2 < m->tri.size(); t += 3) {
                double longest = 0.0;
                for (int e = 0; e < 3; ++e) {
                    const size_t i = (size_t)m->tri[t + e].v, j = (size_t)m->tri[t + (e + 1) % 3].v;
                    double d2 = 0.0;
                    for (int c = 0; c < 3; ++c) { const double d = (double)m->position[3 * i + c] - (double)m->position[3 * j + c]; d2 += d * d; }
                    longest = std::fmax(longest, std::sqrt(d2));
                }
                if (longest > 0.0 && std::isfinite(longest)) min_edge = std::fmin(min_edge, longest);
            }
            // (LASGUN_FAST_NO_GATE=1: measurement only -- lg_audit_fast on exactly the meshes the gate refuses, tools/fast_adversarial.py)
            static const bool no_gate = [] { const char *e = std::getenv("LASGUN_FAST_NO_GATE"); return e && e[0] == '1'; }();
            if (!no_gate && std::isfinite(min_edge) && max_abs > min_edge * 0x1p20) {
                a->fast_available = false;
                a->fast_refusal = "fast mode unavailable: a mesh whose coordinates exceed 2^20 times its smallest triangle (a ray in a far triangle's plane is accepted by the reference wherever it passes)";
            }
        }
        if (!a->fast_available) a->stack_depth_fast1 = a->stack_depth;
        // Scenes whose tables stay in L2: the accel records (13 x 16 bytes each) go into LDS behind the stacks of the 256-lane
        // kernels when that keeps four workgroups on a CU -- entering and leaving nested accels is a chain of dependent fetches of
        // these records (37 % of the walk's cycles on config 4m when they come from L2)
        a->accel_image_n16 = 0;
        {
            const size_t img = f.accels.size() * LDS_ACCEL_UNITS * 16;
            if (f.accels.size() <= 64 && ((size_t)a->stack_depth * 256 * 4 + img) * 4 <= LDS_MAX) a->accel_image_n16 = (uint32_t)(f.accels.size() * LDS_ACCEL_UNITS);
        }
        const size_t extra_lds = (size_t)a->accel_image_n16 * 16;
        size_t lds = (size_t)std::max(a->stack_depth, a->stack_depth_fast1) * 256 * 4 + extra_lds;
        if (lds > 64 * 1024) { HIP_TRY(mega_set_lds_limit(lds, false)); HIP_TRY(wf_set_lds_limit(lds, false)); HIP_TRY(queue_set_lds_limit(lds, false)); }
        int per_cu = 0, cus = 0;
        HIP_TRY(trace_occupancy(a->stack_depth, false, extra_lds, &per_cu));
        int per_cu_fast = 0;
        HIP_TRY(trace_occupancy(a->stack_depth_fast1, true, 0, &per_cu_fast));
        if (per_cu_fast < 1) per_cu_fast = 1;
        a->max_blocks_fast = (uint32_t)per_cu_fast;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, a->device));
        if (per_cu < 1) per_cu = 1;
        a->max_blocks = (uint32_t)(per_cu * cus);
        a->max_blocks_fast *= (uint32_t)cus;
        int wb = 0, wbf = 0;
        HIP_TRY(wf_trace_occupancy(a->stack_depth, false, extra_lds, &wb));
        HIP_TRY(wf_trace_occupancy(a->stack_depth_fast1, true, 0, &wbf));
        a->wf_blocks = (uint32_t)((wb < 1 ? 1 : wb) * cus);
        a->wf_blocks_fast = (uint32_t)((wbf < 1 ? 1 : wbf) * cus);
        int qb = 0;
        HIP_TRY(queue_occupancy(a->stack_depth, extra_lds, &qb));
        if (const char *e = std::getenv("LASGUN_QUEUE_BLOCKS_PER_CU")) { const int v = std::atoi(e); if (v >= 1 && v < qb) qb = v; } // (diagnostic: how much the kernel gains from each resident workgroup)
        a->queue_blocks = (uint32_t)((qb < 1 ? 1 : qb) * cus);
        a->cus = (uint32_t)cus;
        // LDS-resident scene: the REFERENCE tree's nodes (56 of 64 bytes, padded to 80 when that fits),
        // its primrefs, the spheres (padded to 48 when that fits) and cuboids, behind 1024 per-lane
        // stacks, all within one CU's LDS.  The flat tables interleave reference and fast trees per
        // accel, so the image renumbers the reference trees compactly (DAccel::lnode_base / lprim_base).
        {
            FlatScene &fm = a->flat;
            std::vector<uint32_t> nb, pb; // start offsets of every tree / primref run, both kinds
            for (const DAccel &A : fm.accels) { nb.push_back(A.node_base); nb.push_back(A.fnode_base); pb.push_back(A.prim_base); pb.push_back(A.fprim_base); }
            std::sort(nb.begin(), nb.end()); nb.erase(std::unique(nb.begin(), nb.end()), nb.end());
            std::sort(pb.begin(), pb.end()); pb.erase(std::unique(pb.begin(), pb.end()), pb.end());
            auto extent = [](const std::vector<uint32_t> &starts, uint32_t b, size_t total) {
                auto it = std::upper_bound(starts.begin(), starts.end(), b);
                return (uint32_t)((it == starts.end() ? total : (size_t)*it) - b);
            };
            std::vector<std::pair<uint32_t, uint32_t>> nruns, pruns; // (global base, compact base) of each reference tree, once
            uint32_t nn = 0, np = 0;
            // compact numbering: the non-mesh accels first (their slots get a leaf record in the image), then the meshes
            uint32_t np_soup = 0; // slots of the non-mesh accels
            for (int pass = 0; pass < 2; ++pass) {
              if (pass == 1) np_soup = np;
              for (DAccel &A : fm.accels) {
                if (((A.flags & AF_MESH) != 0u) != (pass == 1)) continue;
                auto fn = std::find_if(nruns.begin(), nruns.end(), [&](auto &r) { return r.first == A.node_base; });
                if (fn == nruns.end()) { nruns.emplace_back(A.node_base, nn); A.lnode_base = nn; nn += extent(nb, A.node_base, fm.nodes.size()); }
                else A.lnode_base = fn->second;
                auto fp = std::find_if(pruns.begin(), pruns.end(), [&](auto &r) { return r.first == A.prim_base; });
                if (fp == pruns.end()) { pruns.emplace_back(A.prim_base, np); A.lprim_base = np; np += extent(pb, A.prim_base, fm.primref.size()); }
                else A.lprim_base = fp->second;
              }
            }
            const size_t stack_bytes = (size_t)a->stack_depth * 1024 * 4; // per-lane stacks of the private walks
            const size_t prim16 = ((size_t)np + 3) / 4;
            // image: [nodes, LDS_NODE_STRIDE units each][primrefs][leaf records, 3 units per slot][accel records]
            const size_t accel16 = fm.accels.size() * LDS_ACCEL_UNITS;
            const size_t n16 = (size_t)nn * LDS_NODE_STRIDE + prim16 + (size_t)np_soup * 3 + accel16;
            if (stack_bytes + n16 * 16 <= LDS_MAX) {
                std::vector<uint32_t> img(n16 * 4, 0u);
                a->lds_node_off = 0;
                for (auto &r : nruns)
                    for (uint32_t i = 0, e = extent(nb, r.first, fm.nodes.size()); i < e; ++i) {
                        uint32_t *rec = &img[((size_t)(r.second + i) * LDS_NODE_STRIDE) * 4];
                        std::memcpy(rec, &fm.nodes[r.first + i], 56);
                    }
                a->lds_prim_off = nn * LDS_NODE_STRIDE;
                a->lds_soup_off = a->lds_prim_off + (uint32_t)prim16;
                for (auto &r : pruns)
                    for (uint32_t i = 0, e = extent(pb, r.first, fm.primref.size()); i < e; ++i) {
                        img[(size_t)a->lds_prim_off * 4 + r.second + i] = fm.primref[r.first + i];
                        if (r.second + i < np_soup)
                            std::memcpy(&img[((size_t)a->lds_soup_off + (size_t)(r.second + i) * 3) * 4], &fm.leaf_soup[r.first + i], 48);
                    }
                // walk words of every record (words 16..19; walk.h, traverse_ref): the second formulation of the reference walk
                // addresses nodes by their byte offset in the image and takes a leaf's slot range ready-made
                for (const DAccel &A : fm.accels) {
                    const uint32_t tree0 = a->lds_node_off * 16u + A.lnode_base * LDS_NODE_STRIDE * 16u;
                    for (uint32_t i = 0, e = extent(nb, A.node_base, fm.nodes.size()); i < e; ++i) {
                        uint32_t *rec = &img[((size_t)(A.lnode_base + i) * LDS_NODE_STRIDE) * 4];
                        const DNode &nd = fm.nodes[A.node_base + i];
                        if (nd.meta & NODE_LEAF) { rec[16] = A.lprim_base + nd.link; rec[17] = NODE_LEAF; rec[18] = rec[16] + (nd.meta & 0xFFFFu); rec[19] = nd.pad; }
                        else { rec[16] = tree0 + nd.link * LDS_NODE_STRIDE * 16u; rec[17] = 1u << (nd.meta & 3u); rec[18] = 0u; }
                        rec[17] |= nd.meta & NODE_NOPRUNE;
                    }
                }
                a->lds_accel_off = a->lds_soup_off + np_soup * 3u;
                for (size_t i = 0; i < fm.accels.size(); ++i) {
                    const DAccel &A = fm.accels[i];
                    uint32_t *rec = &img[((size_t)a->lds_accel_off + i * LDS_ACCEL_UNITS) * 4];
                    std::memcpy(rec, &A.minv, 96);
                    rec[24] = a->lds_node_off * 16u + A.lnode_base * LDS_NODE_STRIDE * 16u;
                    rec[25] = A.lprim_base; rec[26] = A.prim_base - A.lprim_base; rec[27] = A.flags;
                    rec[28] = (uint32_t)A.parent; rec[29] = A.nchain;
                    for (int k = 0; k < MAX_CHAIN; ++k) rec[32 + k] = A.chain[k];
                    std::memcpy(rec + 40, A.prune, sizeof A.prune);
                }
                stage.add(a->lds_image, img);
                a->lds_image_n16 = (uint32_t)n16;
                HIP_TRY(mega_set_lds_limit(LDS_MAX, true)); HIP_TRY(wf_set_lds_limit(LDS_MAX, true)); HIP_TRY(queue_set_lds_limit(LDS_MAX, true));
                a->ldss_blocks = (uint32_t)cus;
            }
            stage.add(a->accels, fm.accels); // with the compact bases
            if (a->accel_image_n16) { // the accel records alone, global bases in unit [6] (the LDS-resident image carries compact ones)
                std::vector<uint32_t> img((size_t)a->accel_image_n16 * 4, 0u);
                for (size_t i = 0; i < fm.accels.size(); ++i) {
                    const DAccel &A = fm.accels[i];
                    uint32_t *rec = &img[i * LDS_ACCEL_UNITS * 4];
                    std::memcpy(rec, &A.minv, 96);
                    rec[24] = A.node_base; rec[25] = A.prim_base; rec[26] = 0u; rec[27] = A.flags;
                    rec[28] = (uint32_t)A.parent; rec[29] = A.nchain;
                    for (int k = 0; k < MAX_CHAIN; ++k) rec[32 + k] = A.chain[k];
                    std::memcpy(rec + 40, A.prune, sizeof A.prune);
                }
                stage.add(a->accel_image, img);
            }
        }
        stage.commit(a->arena);
        // Which organisation is the default (measured, tools/threshold_sweep.py): the megakernel unless the scene has
        // so many spheres / boxes that BVH-node and sphere tests dominate a ray (>= 512: with the scene tables in LDS
        // the megakernel keeps up to ~50 node + primitive tests per ray; beyond that the traversal kernels' lower
        // register pressure outweighs the per-pixel state traffic); scenes that also carry a big mesh have long,
        // uneven tiles and need more of them per wave to balance.
        {
            size_t big_mesh = 0;
            for (const auto &m : a->scene->meshes) if (m && m->tri.size() / 3 > big_mesh) big_mesh = m->tri.size() / 3;
            // Scenes with glass / mirror run level by level in the wavefront pipeline under the same criterion (tools/bench_configs.py
            // --org=...): where node and sphere tests dominate.  Small specular scenes are a wash (Cornell glass
            // 512^2: 0.81 ms level by level, 0.80 ms in the megakernel since both walk with traverse_ref), and with a big mesh the deeper
            // levels are few, long, incoherent walks through 254-triangle leaves whose slowest wave sets each launch's length
            // (100k-triangle glass torus: 226 against 136 ms): those stay in the megakernel, where other tiles fill the gaps.
            // the reference's mesh leaves hold up to 254 triangles (bvh.rs:187,289): skipping one pays for many node steps -- from PRUNE_MIN_TRIS
            // triangles (tools/prune_threshold_probe.py, profiles/r05_prune_threshold.jsonl: below that the pruned walk is 5-30 % SLOWER on
            // 1024^2 frames and its records are a third to a half of the accel build; rounds 3-5 had 256)
            a->prune_default = big_mesh >= PRUNE_MIN_TRIS;
            a->streaming_pays = f.spheres.size() + f.cuboids.size() >= 512 && !(f.has_specular && big_mesh >= 4096);
            a->mega_narrow = f.spheres.size() + f.cuboids.size() < 512;
            if (const char *e = std::getenv("LASGUN_MEGA_LANES")) a->mega_narrow = std::atoi(e) == 768; // (A/B)
            // a big mesh of glass / mirror: the queue organisation (round 4; config 4: 38.7 against the megakernel's 40.8 ms and the
            // level-by-level pipeline's 80; a metal mesh beside a small mirror -- config 4m -- stays in the megakernel: 14.7 / 16.4)
            bool specular_mesh = false; // a mesh of >= 4096 triangles that is itself glass / mirror: every hit on it spawns secondary rays
            for (const DAccel &A : f.accels)
                if ((A.flags & AF_MESH) && A.material >= 0) {
                    const int kind = f.materials[(size_t)A.material].kind;
                    size_t tris = 0;
                    for (const auto &m : a->scene->meshes) if (m && m->tri.size() / 3 > tris) tris = m->tri.size() / 3; // (an upper bound: the largest mesh)
                    specular_mesh = specular_mesh || ((kind == MAT_GLASS || kind == MAT_MIRROR) && tris >= 4096);
                }
            a->queue_default = f.has_specular && big_mesh >= 4096 && specular_mesh;
            a->streaming_min_items = big_mesh >= 4096 ? (1ull << 23) : (1ull << 21); // (config 3's scene at 1024^2: 0.84 ms in the megakernel, 0.99 level by level; at 2048^2: 2.20 / 2.11)
        }
        if (times) {
            const auto t_end = std::chrono::steady_clock::now();
            auto ms = [](auto a0, auto a1) { return std::chrono::duration<double, std::milli>(a1 - a0).count(); };
            std::fprintf(stderr, "[lasgun] accel build: flatten %.3f ms, table uploads %.3f ms, streams / occupancy / LDS images %.3f ms\n", ms(t_begin, t_flat), ms(t_flat, t_up), ms(t_up, t_end));
        }
}

// First request for the fast mode: its trees are built and every table is uploaded into a SECOND accel; only when all of
// that has succeeded are the tables swapped in (a failure -- a HIP error, out of memory -- leaves the accel as it was).
// The reference trees must come out as they did at lg_accel_from: a scene modified since is an error, not a silent
// change of the parity tables.  Caller holds a->mtx.
static void swap_tables(lg_accel &x, lg_accel &y) {
    using std::swap;
    swap(x.flat, y.flat);
    swap(x.arena, y.arena); swap(x.stats, y.stats); // (the counters' record is a view into the arena like the small tables)
    swap(x.nodes, y.nodes); swap(x.nodes4, y.nodes4); swap(x.primref, y.primref); swap(x.spheres, y.spheres); swap(x.sphere_mat, y.sphere_mat);
    swap(x.cuboids, y.cuboids); swap(x.cuboid_mat, y.cuboid_mat); swap(x.tri_v, y.tri_v); swap(x.tri_n, y.tri_n); swap(x.tri_t, y.tri_t);
    swap(x.vpos, y.vpos); swap(x.vnorm, y.vnorm); swap(x.vtex, y.vtex); swap(x.leaf_soup, y.leaf_soup); swap(x.chunks, y.chunks); swap(x.strips, y.strips);
    swap(x.sphere_ref_leaf, y.sphere_ref_leaf); swap(x.cuboid_ref_leaf, y.cuboid_ref_leaf); swap(x.tri_ref_leaf, y.tri_ref_leaf); swap(x.accel_ref_leaf, y.accel_ref_leaf);
    swap(x.accels, y.accels); swap(x.materials, y.materials); swap(x.lights, y.lights);
    swap(x.lds_image, y.lds_image); swap(x.accel_image, y.accel_image); swap(x.accel_image_n16, y.accel_image_n16);
    swap(x.lds_image_n16, y.lds_image_n16); swap(x.lds_node_off, y.lds_node_off); swap(x.lds_prim_off, y.lds_prim_off);
    swap(x.lds_soup_off, y.lds_soup_off); swap(x.lds_accel_off, y.lds_accel_off);
    swap(x.ldss_blocks, y.ldss_blocks); swap(x.cus, y.cus);
    swap(x.stack_depth, y.stack_depth); swap(x.stack_depth_fast1, y.stack_depth_fast1); swap(x.max_blocks, y.max_blocks); swap(x.max_blocks_fast, y.max_blocks_fast);
    swap(x.wf_blocks, y.wf_blocks); swap(x.wf_blocks_fast, y.wf_blocks_fast); swap(x.queue_blocks, y.queue_blocks);
    swap(x.queue_default, y.queue_default); swap(x.prune_default, y.prune_default); swap(x.queue_min_items, y.queue_min_items); swap(x.specular_small_items, y.specular_small_items);
    swap(x.device_bytes, y.device_bytes); swap(x.fast_available, y.fast_available); swap(x.fast_refusal, y.fast_refusal);
    swap(x.streaming_pays, y.streaming_pays); swap(x.streaming_min_items, y.streaming_min_items); swap(x.mega_narrow, y.mega_narrow);
}
// the tables once more, with what was left out of them: the fast mode's trees (lg_accel_set_mode(1)), the pruned walk's leaf records (lg_accel_set_prune(1), lg_audit_prune)
void rebuild_tables(const lg_accel *ca, bool fast) {
    if (fast && ca->flat.has_fast) return;
    lg_accel *a = const_cast<lg_accel *>(ca);
    use_device(a->device);
    std::unique_ptr<lg_accel> next(new lg_accel());
    next->scene = a->scene;
    next->device = a->device;
    next->prune = a->prune == 1 || a->flat.has_records ? 1 : a->prune; // (what the tables hold stays in them)
    build_and_upload(next.get(), fast || a->flat.has_fast); // throws: `a` is untouched
    next->prune = a->prune;
    // (bit patterns, not values: a NaN bound of a degenerate scene equals itself here)
    if (next->flat.dump_f.size() != a->flat.dump_f.size() ||
        (!a->flat.dump_f.empty() && std::memcmp(next->flat.dump_f.data(), a->flat.dump_f.data(), a->flat.dump_f.size() * sizeof(a->flat.dump_f[0])) != 0) ||
        next->flat.dump_i != a->flat.dump_i)
        throw Error("the scene was modified after lg_accel_from: the accel's reference trees no longer match it (build a new accel)");
    HIP_TRY(hipDeviceSynchronize()); // nothing may still be reading the tables that are about to be replaced
    swap_tables(*a, *next);
    std::swap(a->flat.dump_f, next->flat.dump_f); // same contents; keeps the storage lg_accel_dump's callers point into
    std::swap(a->flat.dump_i, next->flat.dump_i);
    // `next` (the old tables) is released here; its stream was never created for launches
}

lg_accel *accel_from_on(const lg_scene *s, int device) {
    lg_accel *a = nullptr;
    int rc = guarded([&] {
        a = new lg_accel();
        a->scene = &s->s;
        a->device = device;
        build_and_upload(a, false);
    });
    if (rc) { delete a; return nullptr; }
    return a;
}
