// lasgun_amd/csrc/dscene.h -- the flattened scene as the HIP kernels see it in HBM.
//
// The reference walks a tree of boxed `dyn Primitive`s (one BVHAccel per Aggregate and per
// mesh, /root/reference/src/accelerators/bvh.rs:45-70,150-162).  The device gets flat tables:
//
//   DNode[]     64-byte AoS records {bmin[3], bmax[3], link, meta}: a lane fetches one whole
//               node with four 16-byte loads that fall in ONE cache line.  Lanes of a wave walk
//               different nodes, so a per-field SoA split would cost seven lines per visit
//               instead of one; coalescing across lanes happens whenever neighbouring pixels
//               visit the same node (same address -> one request).
//   primref[]   per accel, in the reference's `order[]` sequence: kind<<30 | index.
//   DSphere[] / DCuboid[] 32/48-byte records; triangles as u32 index triples into f32 vertex
//               tables (f32 is what obj::ObjData holds; widened on every access exactly like
//               `.into()`, src/shape/triangle.rs:40-43).
//   leaf_soup[] the geometry of every primitive again, one 48-byte record per primref slot in
//               LEAF ORDER (triangle: 9 x f32 positions; sphere: c, r as 4 x f64; cuboid: min, max
//               as 6 x f64).  The traversal streams a leaf with three aligned 16-byte loads per
//               slot, issued together with the primref load and one slot ahead, instead of the
//               primref -> index -> vertex dependent-load chain (the reference builds leaves of
//               up to 254 triangles, bvh.rs:187,289).
//   DAccel[]    one per BVHAccel instance: 3x4 m / minv, parent link, root->self chain,
//               default material, swap_backface.  Mesh instances share nodes and primrefs.
#pragma once
#include "vecmath.h"

namespace lg {

enum PrimKind : uint32_t { PK_SPHERE = 0, PK_CUBOID = 1, PK_TRIANGLE = 2, PK_ACCEL = 3 };
constexpr uint32_t PRIM_INDEX_MASK = 0x3FFFFFFFu;
constexpr uint32_t NO_HIT = 0xFFFFFFFFu;
constexpr uint32_t WF_NONE = 0xFFFFFFFFu, WF_MISS = 0xFFFFFFFEu; // wavefront pipeline: no such child / the ray hit nothing
constexpr int MAX_CHAIN = 8;      // scene-graph nesting levels (root = 1)
// tile counters of a persistent kernel (kcommon.h, claim_tile): word [0] the single-head forms' next tile, word [TILE_GONE] one bit per XCD band
// that is known to be exhausted (a line no claim touches), then one head word per XCD band from [TILE_HEAD0] on, 128 bytes -- one L2 line -- apart
constexpr uint32_t TILE_HEADS = 8u, TILE_HEAD_STRIDE = 32u;
constexpr uint32_t TILE_HEAD0 = 32u, TILE_GONE = 16u;
constexpr uint32_t TILE_COUNTER_WORDS = TILE_HEAD0 + TILE_HEADS * TILE_HEAD_STRIDE;
constexpr uint32_t NO_TILE = 0xFFFFFFFFu;
constexpr uint32_t NODE_LEAF = 0x80000000u;
// reference trees: a leaf below this node holds a nested BVHAccel -- the pruned walk never skips such a
// node, because its bounds on a primitive's t are stated per level, for spheres, boxes and triangles only
constexpr uint32_t NODE_NOPRUNE = 0x40000000u;
// Margins of the pruned walk, in units of the level's size S = |o - centre|_1 + (sum of the root box's extents):
// a primitive's accepted hit point o + t*d lies within  e0 + S*(PRUNE_E1 + e2*S)  of its bounds box (per axis; triangles: on the
// ray's dominant axis only).  u = 2^-53; the derivations give 114 u W^2 / r for a sphere, 4 u (|b| + |o|) for a
// box, 12 u S for a triangle; the constants below keep a factor >= 8 above them.
constexpr double PRUNE_E1 = 0x1p-46;          // per unit of S (boxes: 4u, triangles: 12u)
constexpr double PRUNE_E0_PER_COORD = 0x1p-46; // e0 = this * (largest |coordinate| of the level's boxes): box rounding u|c -+ r|, 8u|b|
constexpr double PRUNE_E2_TIMES_RMIN = 0x1p-42; // e2 = this / (smallest sphere radius of the level); 114 u = 2^-46.2
constexpr double PRUNE_LIMIT_REL = 0x1p-40;    // the limit itself is taken as limit * (1 + this): covers the slab's own 3 roundings
constexpr double PRUNE_RANGE = 0x1p100;        // magnitudes beyond [1/this, this] (scene or ray): the level is walked unpruned
// LDS-resident scene image: node stride in 16-byte units.  80-byte nodes (and the 48-byte leaf records) make 16
// consecutive records start in 16 different bank groups; a compile-time constant so that indexing is a shift and an add.
constexpr uint32_t LDS_NODE_STRIDE = 5u;
// LDS image, per accel (10 units = 160 bytes): [0..5] minv (12 doubles), [6] {byte offset of its tree in the image, compact prim
// base, leaf_soup slot delta, flags}, [7] {parent, nchain, -, -}, [8..9] chain[8].  Entering and leaving nested accels is
// a chain of DEPENDENT fetches of these fields; from LDS each link costs ~64 cycles instead of an L2 round trip.
// [10..12] the pruned walk's constants of the level: {centre x, y}, {centre z, size}, {e0, e2} (DAccel::prune).
constexpr uint32_t LDS_ACCEL_UNITS = 13u;

struct alignas(64) DNode {
    double bmin[3];
    double bmax[3];
    uint32_t link; // leaf: offset of its first primref (relative to the accel's prim_base); interior: second child (relative to node_base)
    uint32_t meta; // leaf: NODE_LEAF | nprims (u16, bvh.rs:440) ; interior: split axis
    uint32_t parent; // reference trees: the parent node (relative to node_base), NO_HIT at the root -- fast mode's candidate check walks leaf -> root
    uint32_t pad;    // leaves of a mesh's reference tree: index of the leaf's first culling record (DChunk) | number of its records << 24
};
static_assert(sizeof(DNode) == 64, "DNode must be one 64-byte line");

// Wide record of an INTERIOR node of a FAST tree (same index as its DNode; only the nodes a wide node was collapsed from are
// filled in): up to WIDE children -- the node's two children with the larger ones opened in turn (host.cpp, wide_records) --
// each with its box in f32 rounded OUTWARD from the fast tree's (already inflated) f64 box, and a link word: bit 31 = leaf,
// then bits 28-30 = its primitive count and bits 0-27 its first slot (relative to the accel's prim_base), else the child's node
// index (relative to node_base); NO_HIT = no such child.  One 128-byte fetch feeds four slab tests and the walk is half as
// deep in dependent fetches as a binary walk.
constexpr int WIDE = 4;
constexpr uint32_t WIDE_LEAF = 0x80000000u, WIDE_COUNT_SHIFT = 28u, WIDE_START_MASK = 0x0FFFFFFFu;
struct alignas(128) DNode4 {
    float box[WIDE][6]; // min x y z, max x y z
    uint32_t link[WIDE];
    uint32_t pad[4];
};
static_assert(sizeof(DNode4) == 128, "DNode4 must be 128 bytes");

struct alignas(32) DSphere {
    double cx, cy, cz, r;
};
struct alignas(16) DCuboid {
    double mn[3], mx[3];
};
struct alignas(16) DLeafRec { // 48-byte leaf-ordered geometry record (see header comment)
    uint32_t w[12];
};
// Pruned walk, inside the reference's fat mesh leaves (up to 254 triangles, bvh.rs:187,289): one record per run of <= 32
// triangles of the leaf, made by the host from those triangles -- their bounds, a cone around their normals and two shape numbers.
// (The reference orders a leaf's triangles by a Morton code that ignores x, bvh.rs:575-579: sixteen consecutive ones are no
// neighbours.  The host regroups each leaf's triangles into spatial runs -- host.h, leaf_soup2, never uploaded -- and the device reads a run
// as triangle strips, DStrip below; the leaf loop decides exact ties in t by the ORIGINAL slot number, which is what the reference's
// first-come rule amounts to.)  The walk skips a run's triangle tests when NONE of them could be accepted:
// always by the ray's dominant axis; on all three axes with a margin that grows with 1 / sigma^3, sigma a lower bound (from the
// cone) of the sine of the angle at which the ray crosses the record's triangles -- never for a ray that may lie in a triangle's plane.
struct alignas(64) DChunk {
    float bmin[3], bmax[3]; // union of the triangles' bounds (vertex coordinates are f32: exact)
    float axis[3];          // unit vector; every triangle normal of the record is within theta of it (or of its opposite)
    float cos_t;            // cos(theta), rounded down; -1 = no lateral culling for this record (wide cone, degenerate triangle)
    float g2;               // max over the triangles of (longest edge)^2 / (smallest altitude)^3, rounded up
    float hmin;             // smallest altitude of any triangle of the record, rounded down
    uint32_t start, count;  // run record: first slot of the run in the host's run-ordered soup, triangles | strip entries << 8; group record: CHUNK_IS_GROUP, runs behind it
    float sin_t;            // sin(theta), rounded up
    uint32_t pad;           // run record: its first strip entry (DParams::strips)
};
static_assert(sizeof(DChunk) == 64, "DChunk is one 64-byte line");
// The triangles of a run again, as TRIANGLE STRIPS (round 4): one 16-byte entry per NEW vertex.  A strip is a sequence of triangles
// in which each shares the last two vertices of the one before, so an entry carries one vertex (f32 position, as in the soup) and
// a code word: STRIP_TRI = this vertex completes a triangle with the two before it (low 28 bits: the triangle's slot in leaf_soup);
// otherwise it only (re)starts a strip.  What the walk does with them (walk.h, mesh_leaf2): the reference transforms the three
// vertices of every triangle and forms three edge functions (triangle.rs:186-222) -- per vertex and per edge the SAME doubles for
// every triangle that shares them, and an edge function seen from the neighbouring triangle is the same value negated -- and its
// first test is "all three of one sign" (triangle.rs:224-230), which does not change when all three are negated or permuted.  Along
// a strip one vertex transform and two products-and-a-difference per triangle answer that test exactly; only a triangle that
// passes it (about one in thirty) is put to the reference's whole formula, from its own record.
struct alignas(16) DStrip {
    float x, y, z;
    uint32_t code;
};
constexpr uint32_t STRIP_TRI = 0x80000000u, STRIP_SLOT_MASK = 0x0FFFFFFFu;
// a run record's `count` word: triangles of the run | strip entries of the run << 8; its `pad` word: first strip entry
constexpr uint32_t CHUNK_SHIFT = 5u;            // at most 32 slots per run (measured, config 4 / 4m / 5 in ms: 16 slots in groups of 4: 44.2 / 16.1 / 62.7;
                                                // 32 in groups of 2: 42.0 / 15.6 / 61.7; 8 in 4: 48.7 / 17.6 / 64.8; a binary hierarchy of groups: no change)
constexpr uint32_t CHUNK_GROUP = 2u;            // runs per group record
constexpr uint32_t CHUNK_IS_GROUP = 0xFFFFFFFFu; // DChunk::start of a group record; its count = the run records behind it
// Lateral culling of a record: with sigma <= |n . d| / |d| for every triangle of the record (from the cone and the ray:
// cos(alpha + theta), alpha the angle between the ray and the cone's axis), the accepted hit point lies within
// m = CHUNK_K0 * R^2 * g2 / sigma^3 of the triangle (derivation: 4608 u = 5.1e-13, x 7), R the 1-norm distance from the ray's
// origin to the record's far corner; used when sigma >= CHUNK_SIGMA_MIN and hmin^2 * sigma >= CHUNK_HGATE * R^2 (the projected
// triangle's area must dominate the edge functions' rounding: 384 u = 4.3e-14).
constexpr float CHUNK_K0 = 0x1p-38f;
constexpr float CHUNK_SIGMA_MIN = 1e-3f;
constexpr float CHUNK_HGATE = 0x1p-40f;

struct DMaterial { // == lg_material of include/lasgun_hip.h
    int32_t kind;
    int32_t pad;
    double p[10];
};
enum MatKind : int32_t { MAT_MATTE = 0, MAT_PLASTIC = 1, MAT_METAL = 2, MAT_GLASS = 3, MAT_MIRROR = 4 };

struct DLight {
    double pos[3], intensity[3], falloff[3];
};

enum AccelFlags : uint32_t { AF_SWAP_BACKFACE = 1, AF_MESH = 2, AF_HAS_N = 4, AF_HAS_UV = 8, AF_IDENTITY = 16 /* minv is exactly the identity */ };

struct alignas(16) DAccel {
    Affine m;    // 96 B
    Affine minv; // 96 B
    uint32_t node_base;
    uint32_t prim_base;
    int32_t parent;   // -1 for the root
    int32_t material; // default material id (bvh.rs:63), -1 = None
    uint32_t flags;
    uint32_t nchain;  // number of accels on the path root..self
    uint32_t chain[MAX_CHAIN];
    uint32_t fnode_base; // the optional fast tree (binned SAH, one primitive per leaf) over the SAME primitives
    uint32_t fprim_base;
    uint32_t lnode_base; // reference tree again, in the compact numbering of the LDS-resident scene image (DParams::lds_image)
    uint32_t lprim_base;
    uint32_t pad[2];
    double prune[6]; // pruned walk: centre of the root box (3), sum of its extents, e0, e2 (e0 = +inf: this level is never pruned)
};

// Control words of the queue organisation (DParams::q_ctl).  Every word that many waves hammer sits on a 128-byte line of its own
// (agent-scope atomics and sc1 polls are served per line, ~88 per microsecond: MI355X_MICROARCH.md): header [QC_FINISHED] every
// level is done ([QC_ERROR]: unused since round 5 -- the error word is DParams::q_err, outside everything a launch clears); then QC_LEVEL_WORDS
// words per level d at QC_LEVEL0 + QC_LEVEL_WORDS * d: [QC_COUNT] packets reserved so far (level 0: unused), [QC_CLAIMED] tickets handed out (level 0: units),
// [QC_STATE, +1] one 64-bit word (packets of the level + 1) << 32 | packets done -- the high half is added when the count is final.
// Ready words (DParams::q_ready), one per 64-ray packet of the levels >= 1: QR_LAST | its ray count once the wave that filled it has
// stored its rays (a packet has one producer and is published whole).
constexpr uint32_t QC_FINISHED = 0u, QC_ERROR = 32u, QC_HEADS = 64u /* level 0: one claim counter per XCD band, 32 words apart */, QC_LEVEL0 = 384u,
                   QC_LEVEL_WORDS = 128u, QC_COUNT = 0u, QC_CLAIMED = 32u, QC_STATE = 64u;
constexpr uint32_t QC_MAX_LEVELS = 8u;
constexpr uint32_t QC_WORDS = QC_LEVEL0 + QC_LEVEL_WORDS * QC_MAX_LEVELS;
constexpr uint32_t QR_LAST = 0x80000000u;
constexpr uint32_t QR_SLACK = 16u; // spare ready words behind every level's (the holder of a ticket beyond the level's capacity looks at the first: never raised)

struct DStats { // per-launch counters (stats kernel variant only)
    unsigned long long primary_rays, shadow_rays, secondary_rays, nodes_tested, spheres_tested, cuboids_tested,
        triangles_tested, accel_entries, hits;
    // audit of the pruned walk (DParams::audit): skipped nodes / runs, primitives below them that yield a t, those the reference
    // would have accepted (must be 0), smallest (t - limit) / margin among the rest (complemented f64 bits, kept with atomicMax; 0 = no sample)
    unsigned long long audit_nodes, audit_runs, audit_prims, audit_violations, audit_slack_nodes, audit_slack_runs;
    unsigned long long audit_used_nodes; // largest share of a node's margin a skipped primitive needed (f64 bits of a value >= 0, kept with atomicMax)
};

// Per-frame record of the explicit Whitted recursion stack (integrate.rs:69-79), in doubles.
constexpr int FRAME_DOUBLES = 18;
constexpr int STASH_DOUBLES = 13; // p(3) ng(3) ns(3) ss(3) material id

struct DRowTab { uint32_t base, rem; }; // a film row's floor(y*w / n) and (y*w) mod n (strided subsets by lattice column: shade.h, modes 4 / 5)

struct DParams {
    // ---- scene tables
    const DNode *nodes;
    const DNode4 *nodes4; // fast trees: valid at the interior nodes wide records were made for
    const uint32_t *primref;
    const DSphere *spheres;
    const int32_t *sphere_mat;
    const DCuboid *cuboids;
    const int32_t *cuboid_mat;
    const uint32_t *tri_v; // 3 per triangle, global vertex ids
    const uint32_t *tri_n; // 3 per triangle, global normal ids (valid when the mesh has normals)
    const uint32_t *tri_t; // 3 per triangle, global uv ids (valid when the mesh has uvs)
    const float *vpos;
    const float *vnorm;
    const float *vtex;
    const DLeafRec *leaf_soup; // slot j <-> primref[j]
    const DChunk *chunks;      // culling records of the mesh leaves (DNode::pad), each with its run of strip entries
    const DStrip *strips;      // the runs' triangles as strips (DChunk::pad = a run's first entry)
    // fast mode's candidate check (host.h): reference leaf of every sphere / cuboid / triangle / accel (parents: DNode::parent)
    const uint32_t *sphere_ref_leaf, *cuboid_ref_leaf, *tri_ref_leaf, *accel_ref_leaf;
    const DAccel *accels;
    const DMaterial *materials;
    const DLight *lights;
    uint32_t nlights;
    uint32_t recursion;
    int32_t default_material; // id of Material::default()
    uint32_t stack_depth;     // per-lane LDS stack entries
    // ---- camera (camera.rs:6-37), background, ambient
    V3 cam_origin, cam_view, cam_up, cam_aux;
    double image_plane_height, pixel_separation, ss_distance;
    uint32_t ss_root;
    uint32_t boxes_finite; // every node box of the scene is finite: the walk may take the sign-specialised slab test (walk.h, slab_intersects_sg)
    V3 bg_inner, bg_outer;
    double bg_scale;
    V3 ambient;
    // ---- film (film.rs:36-45)
    uint32_t w, h;
    double winv, hinv, aspect;
    // ---- work: either a rectangle of 8x8 tiles or a strided pixel subset (lib.rs:152)
    uint32_t mode; // 0 = rectangle [x0,x1) x [y0,y1); 1 = subset {k + i*n}, 64 consecutive i per tile (4 = the same subset, 64 rows of a lattice column per tile); 2 = the pixel offsets listed in pixel_list[0 .. sub_count);
                   // 3 = the sub_m subsets {ks[j] + q*n} of one n, ks = pixel_list[0 .. sub_m) ascending, work item i = q * sub_m + j
    uint32_t x0, y0, x1, y1;
    uint32_t tiles_x;
    // row-block interleave for multi-GPU balance (mode 0): with ilv_n > 1 the rows [y0, y1) are
    // VIRTUAL rows of a compact tile; image row = ((vy / ilv_b) * ilv_n + ilv_r) * ilv_b + vy % ilv_b
    uint32_t ilv_n, ilv_r, ilv_b;
    unsigned long long sub_k, sub_n, sub_count;
    uint32_t sub_m; // mode 3: how many subsets
    uint32_t sub_cols; // modes 4 / 5 (a subset {k + i*n} / several of them tile by lattice column, shade.h): lattice columns per row = ceil(w / n)
    uint32_t sub_rows; // rows per tile: 64 (mode 4), 64 / sub_m (mode 5)
    uint32_t sub_kk, sub_kdiv; // mode 4: k mod n, k / n
    const DRowTab *sub_rowtab; // per film row y: (floor(y*w / n), (y*w) mod n)
    uint32_t ntiles;
    uint32_t tile_rev; // the megakernel and the queue organisation claim the launch's tiles from the LAST to the first (which tile is rendered when never changes a pixel; capi.cpp, tuned_org)
    uint32_t out_row0; // row of the image stored at out_rgba[0] (0 for a full film, y0 for a row tile)
    uint32_t out_x0;   // mode 0: column of the image stored at out_rgba[0] (0 unless the output is a crop)
    uint32_t out_pitch; // mode 0: pixels per row of the output buffer (w unless the output is a crop)
    uint32_t out_compact; // modes 1, 2: work item i writes output element i (a compact buffer) instead of its pixel offset
    const unsigned long long *pixel_list; // mode 2
    uint8_t *out_rgba;
    double *out_radiance; // optional f64 RGB, same addressing as out_rgba (3 doubles per pixel)
    uint32_t *tile_counter;   // [0] next tile; [16 + 16 * x] next tile of XCD band x (TILE_COUNTER_WORDS)
    double *frames;        // [recursion][FRAME_DOUBLES][nthreads]
    unsigned long long frame_threads;
    double *stash;         // [STASH_DOUBLES][nthreads]: shading frame parked across the shadow traversals
    DStats *stats;
    double *dbg_log;       // lg_trace_pixel only (counting instantiations): [0] = entries used, then 4 doubles per event
    // ---- pipelines: per-work-item state in HBM, SoA, indexed by the dense work index widx = tile * 64 + lane
    unsigned long long n_items; // pixels of a chunk (its tiles * 64): SoA stride of level 0's arrays
    double *frame;              // [STASH_DOUBLES][n_items] shading frame of the hit
    uint32_t *vis;              // [n_items] bit l set <=> light l is visible from the hit
    double *accum;              // [3][n_items] running sum over the pixel's samples (integrate.rs:17-18)
    uint32_t sample_index;      // which supersample this pass renders
    // the level-by-level pipeline with the samples of a pixel SIDE BY SIDE (round 5): a level-0 work tile vt is pixel tile vt / ss_par
    // at sample vt % ss_par, every sample's li() is parked in `accum` ([3][n_items], n_items = pixel tiles * ss_par * 64) and a resolve
    // pass sums a pixel's samples in their order (integrate.rs:17-20).  0 / 1: one sample per launch chain, `accum` the running sum.
    uint32_t ss_par;
    uint32_t split_shift; // megakernel and queue organisation, small launches: a tile handed out in 2^split_shift parts of 64 >> split_shift lanes each (0: whole tiles)
    // ---- wavefront pipeline: li() level by level.  Level d holds the rays of recursion depth d
    // (level 0: the chunk's pixels, dense; deeper: a compacted queue fed by the level above).  Per level: closest-hit
    // pass (misses are finished on the spot, hits are COMPACTED into a hit queue and get their shading frame), any-hit
    // shadow pass over the hit queue, shade pass (radiance of the hit, specular children appended to the next level),
    // then bottom-up: value = (output + spec_r * value[child_r]) + spec_t * value[child_t] * a / pdf (integrate.rs:79,103,129).
    uint32_t wf_level;          // level of this launch
    uint32_t wf_levels;         // number of levels (recursion + 1, or 1 for scenes without glass / mirror)
    uint32_t tile0;             // first tile of the chunk (level 0: pixel tile = tile0 + work tile)
    uint32_t prune;             // reference traversal: the pruned walk (lg_accel_set_prune / the accel's default; walk.h, traverse_ref<.., PRUNE>)
    uint32_t *wf_counts;        // device counters: [d] rays of level d (d >= 1), [wf_levels + d] appended hits of level d
    unsigned long long wf_cap;      // capacity (rays) of this level's arrays
    unsigned long long wf_cap_next; // ... of the next level's
    unsigned long long wf_hit_cap;  // dense part of the hit queue / frame / vis arrays (= capacity of the widest level); appended hits sit behind it
    unsigned long long wf_hit_stride; // their allocated length (SoA stride of the frame)
    double *wf_q;               // [6][wf_cap] origin, direction of this level's rays (levels >= 1)
    double *wf_out;             // [3][wf_cap] output of the hit / background of the miss; the combine pass turns it into li()
    double *wf_spec;            // [8][wf_cap] spectrum of the reflected child (3), of the refracted child (3), |wi.n|, pdf
    uint32_t *wf_child;         // [2][wf_cap] index of the reflected / refracted child in the next level (WF_NONE, WF_MISS)
    double *wf_q_next;          // [6][wf_cap_next]
    double *wf_out_next;        // [3][wf_cap_next] (the combine pass reads the children's values)
    uint32_t *wf_hq;            // [wf_hit_cap] ray index of hit h
    // ---- queue organisation (k_queue.hip): ONE persistent launch runs every recursion level of a chunk.  Level d's rays live in
    // q_rays[d] ([6][n_items << d]; level 0 = the chunk's pixels, no array), its results in q_out[d] / q_spec[d] / q_child[d] (the
    // wavefront pipeline's layout: the combine pass is shared); q_ctl = control words (QC_*), q_ready = one word per 64-ray packet
    // of every level >= 1: how many of its rays have been written
    uint32_t *q_ctl;
    uint32_t *q_ready;
    uint32_t *q_err; // sticky error word in pinned HOST memory (devmem.cpp, g_err_words): set by a wave that gave up waiting, cleared by the host alone
    uint32_t q_units, q_unit_tiles; // level 0's work items: units of q_unit_tiles consecutive 8x8 tiles of the tile SEQUENCE
    // the tile sequence: q_order 0 = the tiles in row order (any addressing mode); 1 (rectangles) = blocks of 32 x 32 tiles in row
    // order, Morton order inside a block, the sequence cut into 8 contiguous bands claimed XCD by XCD (k_queue.hip, q_seq_tile)
    uint32_t q_order, q_blocks_x, q_tiles_y, q_seq_len;
    double *q_rays[8];
    double *q_out[8];
    double *q_spec[8];
    uint32_t *q_child[8];
    // ---- LDS-resident scene (scenes whose node / primref / sphere / cuboid tables fit beside the
    // stacks in the CU's 160 KB): `lds_image` holds those tables in their LDS layout; offsets and
    // strides are in 16-byte units from the start of the image
    const void *lds_image;
    uint32_t lds_image_n16;
    uint32_t mega_lanes;        // the LDS-resident megakernel's workgroup: 0 / 1024 lanes (four waves per SIMD, 128 registers), or 768 (three, 168: k_mega.hip)
    uint32_t lds_node_off;                  // DNode without its pad, LDS_NODE_STRIDE units per node
    uint32_t lds_prim_off;                  // primref[] as dwords from here
    uint32_t lds_soup_off;                  // one 3-unit (48-byte) leaf record per primref slot
    uint32_t lds_accel_off;                 // LDS_ACCEL_UNITS units per accel: what the walk needs of a DAccel (see LDS_ACCEL_UNITS)
    // scenes whose tables stay in L2: just the accel records in their LDS layout (LDS_ACCEL_UNITS units each, unit [6] with global
    // node / primref bases), copied behind the stacks by the 256-lane kernels (walk.h, lvl_set); nullptr: too many accels
    const void *accel_image;
    uint32_t accel_image_n16;
    unsigned long long *stamp_counts; // diagnostic build (-DLG_STAMPS) only
    uint32_t stats_filter;      // counting variant: 0 = all traversals, 1 = closest-hit (primary/secondary) only, 2 = shadow only
    uint32_t audit;             // counting variant: also audit what the pruned walk skips (walk.h, audit_prim)
};

} // namespace lg
