/* include/lasgun_hip.h -- C ABI of liblasgun_hip.so, the MI355X-native drop-in for lasgun's
 * per-pixel ray-trace path.
 *
 * Every entry point in the CORE section replaces one item of the reference's public Rust
 * surface (file:line under nfrasser/lasgun); a Rust shim binds them 1:1.
 * Plain pointers and sizes only; no C++ exception ever crosses this boundary.  Functions that
 * can fail return nonzero / NULL and leave a message in lg_last_error() (thread-local).
 * Panics of the reference (empty aggregate, missing mesh handle, BVH stack overflow) become
 * such errors.  There is NO CPU fallback: without a usable HIP device every render call fails.
 *
 * Film layout (src/film.rs:22-45, src/img.rs:46-67): w*h pixels, row-major, top-left origin,
 * 4 bytes RGBA per pixel, A = 255.
 */
#ifndef LASGUN_HIP_H
#define LASGUN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lg_scene lg_scene;         /* src/scene.rs:11-40      Scene      */
typedef struct lg_aggregate lg_aggregate; /* src/scene/node.rs:25-33 Aggregate  */
typedef struct lg_accel lg_accel;         /* src/lib.rs:42           Accel<'s>  */
typedef struct lg_film lg_film;           /* src/film.rs:7-20        Film       */

/* src/material/mod.rs:3-10 -- `Material` is a Copy enum; here a tagged POD.
 * kind 0 Matte   p = kd[3], sigma
 * kind 1 Plastic p = kd[3], ks[3], roughness
 * kind 2 Metal   p = eta[3], k[3], u_roughness, v_roughness
 * kind 3 Glass   p = kr[3], kt[3], eta
 * kind 4 Mirror  p = kr[3] */
typedef struct lg_material {
    int32_t kind;
    double p[10];
} lg_material;

/* ------------------------------- CORE: the reference's surface ------------------------------ */
const char *lg_last_error(void);

lg_material lg_material_default(void);                                                        /* material/mod.rs:15 */
lg_material lg_material_matte(const double kd[3], double sigma);                              /* material/mod.rs:19 */
lg_material lg_material_plastic(const double kd[3], const double ks[3], double roughness);    /* material/mod.rs:24 */
lg_material lg_material_metal(const double eta[3], const double k[3], double u_roughness, double v_roughness); /* :30 */
lg_material lg_material_glass(const double kr[3], const double kt[3], double eta);            /* material/mod.rs:36 */
lg_material lg_material_mirror(const double kr[3]);                                           /* material/mod.rs:43 */

lg_scene *lg_scene_new(void);                                                                 /* scene.rs:50 */
void lg_scene_free(lg_scene *);
void lg_scene_set_perspective_camera(lg_scene *, double fov);                                 /* scene.rs:69 */
void lg_scene_set_orthographic_camera(lg_scene *, double scale);                              /* scene.rs:74 */
/* The Rust setters return `&mut Camera`; the camera lives in the scene, so these take the scene. */
void lg_camera_look_at(lg_scene *, const double origin[3], const double look[3], const double up[3]); /* camera.rs:85 */
void lg_camera_set_supersampling(lg_scene *, uint8_t base);                                   /* camera.rs:96 */
void lg_camera_set_aperture_radius(lg_scene *, double radius);                                /* camera.rs:100 */
void lg_scene_set_solid_background(lg_scene *, const double color[3]);                        /* scene.rs:79 */
void lg_scene_set_radial_background(lg_scene *, const double inner[3], const double outer[3], double scale); /* :83 */
void lg_scene_set_ambient_light(lg_scene *, const double color[3]);                           /* scene.rs:87 */
void lg_scene_set_mesh_smoothing(lg_scene *, int enabled);                                    /* scene.rs:91 */
void lg_scene_set_max_recursion_depth(lg_scene *, uint32_t max_depth);                        /* scene.rs:95 */
void lg_scene_set_threads(lg_scene *, size_t threads);                                        /* scene.rs:99: caps the devices of lg_set_devices */
void lg_scene_add_point_light(lg_scene *, const double position[3], const double intensity[3], const double falloff[3]); /* :103 */
int lg_scene_parse_obj(lg_scene *, const char *text, size_t len, uint32_t *out_ref);          /* scene.rs:120 -> Result */
int lg_scene_load_obj(lg_scene *, const char *path, uint32_t *out_ref);                       /* scene.rs:127 -> Result */
lg_aggregate *lg_scene_root(lg_scene *);                   /* `scene.root` (pub field, scene.rs:14): BORROWED */
void lg_scene_set_root(lg_scene *, lg_aggregate *moved);   /* scene.rs:132: takes ownership */

lg_aggregate *lg_aggregate_new(void);                                                         /* node.rs:36 */
void lg_aggregate_free(lg_aggregate *);                    /* only for aggregates never moved into a scene/group */
void lg_aggregate_add_group(lg_aggregate *, lg_aggregate *moved);                             /* node.rs:49 */
void lg_aggregate_add_sphere(lg_aggregate *, const double center[3], double radius, const lg_material *); /* node.rs:53 */
void lg_aggregate_add_cube(lg_aggregate *, const double origin[3], double dim, const lg_material *);      /* node.rs:58 */
void lg_aggregate_add_box(lg_aggregate *, const double minbound[3], const double maxbound[3], const lg_material *); /* :63 */
void lg_aggregate_add_obj(lg_aggregate *, uint32_t mesh);                                     /* node.rs:70 */
void lg_aggregate_add_obj_of(lg_aggregate *, uint32_t mesh, const lg_material *);             /* node.rs:75 */
void lg_aggregate_swap_backface(lg_aggregate *);                                              /* node.rs:80 */
void lg_aggregate_translate(lg_aggregate *, const double delta[3]);                           /* node.rs:85 */
void lg_aggregate_scale(lg_aggregate *, double x, double y, double z);                        /* node.rs:91 */
void lg_aggregate_rotate_x(lg_aggregate *, double theta_deg);                                 /* node.rs:96 */
void lg_aggregate_rotate_y(lg_aggregate *, double theta_deg);                                 /* node.rs:101 */
void lg_aggregate_rotate_z(lg_aggregate *, double theta_deg);                                 /* node.rs:106 */
void lg_aggregate_rotate(lg_aggregate *, double theta_deg, const double axis[3]);             /* node.rs:111 */

lg_film *lg_film_new(uint32_t width, uint32_t height);                     /* film.rs:24 (zero-filled) */
lg_film *lg_film_wrap(uint32_t width, uint32_t height, uint8_t *rgba);     /* film.rs:36 new_with_output: caller owns rgba */
uint8_t *lg_film_pixels(lg_film *);
uint32_t lg_film_width(lg_film *);
uint32_t lg_film_height(lg_film *);
void lg_film_free(lg_film *);

/* Accel::from(&scene) (lib.rs:42, bvh.rs:135): builds the nested HLBVH on the host exactly as
 * the reference does, flattens it and uploads it to HBM.  Borrows `scene` until lg_accel_free. */
lg_accel *lg_accel_from(const lg_scene *);
void lg_accel_free(lg_accel *);

/* capture(&scene, &mut film) (lib.rs:55-104): builds the accel, renders every pixel, returns when the film is written.  Devices: the
 * ones named with lg_set_devices / lg_set_device; a process that named none gets EVERY visible device for films of 2^18 pixels and
 * more (the reference takes every core, lib.rs:58-62; `scene.threads` caps the count) -- one accel per device, one RCCL gather --
 * and the HIP current device for smaller films (an accel and a communicator per GPU would cost more than the render). */
int lg_capture(const lg_scene *, lg_film *);                                       /* lib.rs:55 */
/* capture_subset(k, n, &accel, &mut img) (lib.rs:110-162): exactly the pixels {k + i*n < w*h} of the row-major film, no other byte touched;
 * callable over and over on one film (www/renderer.ts:103-120).  Which lane renders which pixel of the subset is the build's business: for
 * 8 <= n <= width a 64-lane tile is 64 rows of one LATTICE COLUMN of the subset (x = (k - y*w) mod n + n*c) -- the densest 64 pixels there are --
 * instead of 64 consecutive i, a strip n*64 pixels long (LASGUN_SUBSET_LATTICE=0: the latter). */
int lg_capture_subset(size_t k, size_t n, const lg_accel *, lg_film *);            /* lib.rs:110 */
lg_film *lg_render(const lg_scene *, uint32_t width, uint32_t height);             /* lib.rs:46 */

/* ----------------------- EXTRAS: no counterpart in the reference ---------------------------- */
typedef struct lg_stats { /* deterministic work counters of one render (stats kernel variant) */
    uint64_t primary_rays, shadow_rays, secondary_rays, nodes_tested, spheres_tested, cuboids_tested, triangles_tested,
        accel_entries, hits;
} lg_stats;

int lg_set_device(int device);      /* HIP device new accels are created on (default 0) */
int lg_device_count(void);
/* Device allocations of 1 MiB and more are recycled through a small per-process pool (at most 4 GiB parked per device;
 * flushed automatically when an allocation runs out of memory).  lg_trim_pool gives the parked blocks of `device`
 * (-1: of every device) back to the driver and returns the number of bytes freed. */
uint64_t lg_trim_pool(int device);
/* Devices a host-film lg_capture / lg_render is split over, one host thread per device -- the counterpart
 * of the reference's split over `scene.threads` CPU threads (lib.rs:58-104): count == 0 selects every visible
 * device; a non-zero `scene.threads` caps how many of them are used.  Each device renders the 64-row blocks
 * {r, r+n, ...} of the film (contiguous row tiles when the height is not a multiple of 64*n) from its own copy
 * of the scene and copies them straight into the host film; there is no inter-device traffic.  Until this is
 * called, lg_capture uses the one device of lg_set_device.  An index may repeat (its shares then run
 * concurrently on that device).  lg_accel handles stay bound to the device they were created on. */
int lg_set_devices(const int *device_ids, int count);

/* Accel::from on a given device (lg_accel_from uses the device of lg_set_device). */
lg_accel *lg_accel_from_on(const lg_scene *, int device);

/* ONE film on several GPUs of this process, gathered over xGMI -- the node-level counterpart of the reference's fan-out
 * over threads (lib.rs:55-104).  lg_multi_create builds the scene's accel on every device of the list (device_ids[0] is
 * the ROOT) and, when the list names more than one distinct device, one RCCL communicator per device (ncclCommInitAll;
 * RCCL is dlopen'ed at that moment, never for single-device use).  A capture renders rank r's share on its device --
 * the `block_rows`-row blocks {r, r+n, ...} when the height is a multiple of block_rows * n (64 balances the load to a few
 * per cent), contiguous row tiles otherwise or when block_rows is 0 -- and then moves every share with ONE grouped RCCL
 * exchange (ncclSend on the owners, ncclRecv on the root, inside one ncclGroupStart / ncclGroupEnd) straight to its place
 * in the film in the ROOT's device memory; the root's own contiguous tile is rendered in place.  A device may repeat in the
 * list (its shares are then copied device-locally): that is how a 1-GPU box rehearses the split; with the environment
 * variable LASGUN_MULTI_FORCE_RCCL=1 such shares travel through RCCL as well (send / recv to self).
 * lg_multi_capture_device: dev_rgba is width*height*4 bytes on the root device; synchronous.  lg_multi_capture: the
 * same into a host film (+ one D2H copy).  The scene must outlive the lg_multi. */
typedef struct lg_multi lg_multi;
lg_multi *lg_multi_create(const lg_scene *, const int *device_ids, int count, uint32_t block_rows);
void lg_multi_free(lg_multi *);
int lg_multi_capture_device(lg_multi *, uint32_t width, uint32_t height, void *dev_rgba_on_root);
int lg_multi_capture(lg_multi *, lg_film *);
/* The all-gather form: EVERY rank's device receives the whole film -- dev_rgba[r] is width*height*4 bytes on
 * rank r's device.  One ncclAllGather per device inside one group; contiguous tiles arrive in row order, interleaved blocks are
 * put in row order by n strided device copies.  One device per rank (no repeats); the height must split evenly.  Synchronous. */
int lg_multi_capture_device_all(lg_multi *, uint32_t width, uint32_t height, void *const *dev_rgba);
int lg_multi_rank_count(const lg_multi *);
lg_accel *lg_multi_accel(const lg_multi *, int rank);   /* rank's accel (to select traversal mode / organisation per rank) */
int lg_multi_uses_rccl(const lg_multi *);                /* 1 when a communicator was created */

/* Render image rows [y0, y1) of a width x height film straight into DEVICE memory, no host copy:
 * dev_rgba[0] is pixel (0, row0) of the image.  `hip_stream` is the hipStream_t to enqueue on,
 * used as given (NULL = HIP's default stream, as everywhere in HIP; lg_accel_stream() = the
 * accel's own non-blocking stream); the call only enqueues.  One stream at a time per accel.  This is the multi-GPU row-tile entry point. */
int lg_capture_rows_device(const lg_accel *, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, uint32_t row0,
                           void *dev_rgba, void *hip_stream);
/* Balanced multi-GPU sharding: render the rows y with (y / block_rows) % n == r into a COMPACT
 * device tile of height/n rows (tile row vy <-> image row ((vy/block_rows)*n + r)*block_rows +
 * vy%block_rows).  height must be a multiple of block_rows*n. */
int lg_capture_interleaved_device(const lg_accel *, uint32_t width, uint32_t height, uint32_t block_rows, uint32_t n,
                                  uint32_t r, void *dev_rgba, void *hip_stream);
/* capture_subset into a full width*height device film (pixels outside the subset untouched). */
int lg_capture_subset_device(size_t k, size_t n, const lg_accel *, uint32_t width, uint32_t height, void *dev_rgba,
                             void *hip_stream);
/* SEVERAL subsets {ks[j] + i*n} of one n as ONE render: the pixels written are exactly those of the `count` calls
 * lg_capture_subset(ks[j], n, ...) and no others (repeated k values count once, a k behind the film is an empty subset, as at
 * lib.rs:152).  No counterpart in the reference: its progressive caller (www/renderer.ts:103-120) calls capture_subset a hundred times
 * in a row, and a launch chain per call fills a fraction of a GPU; a batch costs count/n of a frame, and the batch of every k of
 * 0 .. n-1 IS the frame.  Host film: one D2H copy per batch; device film: only enqueues on `hip_stream`, like the calls above. */
int lg_capture_subsets(const size_t *ks, size_t count, size_t n, const lg_accel *, lg_film *);
int lg_capture_subsets_device(const size_t *ks, size_t count, size_t n, const lg_accel *, uint32_t width, uint32_t height,
                              void *dev_rgba, void *hip_stream);
void *lg_accel_stream(const lg_accel *);      /* the accel's own hipStream_t */
int lg_accel_synchronize(const lg_accel *);    /* hipStreamSynchronize(lg_accel_stream()) */

/* f64 radiance before quantisation for subset (k, n); rgb = width*height*3 doubles on the HOST,
 * pixels outside the subset are left untouched. */
int lg_capture_radiance(size_t k, size_t n, const lg_accel *, uint32_t width, uint32_t height, double *rgb);
/* Any list of pixels (offset = y * width + x < width * height) of a width x height film; results are compact and in
 * list order on the HOST: rgba_out[4*i..] and / or rgb_out[3*i..] (f64 radiance before quantisation) for offsets[i];
 * either output may be NULL.  For samples and crops of films too large to move whole (tests, tooling). */
int lg_capture_pixels(const lg_accel *, uint32_t width, uint32_t height, const uint64_t *offsets, size_t count,
                      uint8_t *rgba_out, double *rgb_out);
/* The crop [x0, x1) x [y0, y1) of a width x height film, compact and row-major on the HOST (either output may be NULL). */
int lg_capture_rect(const lg_accel *, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                    uint8_t *rgba_out, double *rgb_out);
/* Work counters for rendering rows [y0, y1) (runs the counting kernel variant once). */
int lg_capture_stats(const lg_accel *, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, lg_stats *out);

/* Audit of the pruned reference walk (lg_accel_set_prune) on real data: rows [y0, y1) are rendered once
 * with the counting variant of the pruned walk, and every node and every run of triangles it SKIPS (although the reference's own
 * box test passes) is also walked the reference's way.  A primitive found there that the reference would have ACCEPTED at that
 * moment -- `t < isect.t` for a closest-hit ray (sphere.rs:86, cuboid.rs:95, triangle.rs:251), t < 1 for a shadow ray
 * (point.rs:49) -- is a violation of the property the walk's exactness rests on: `violations` must be 0.  For the others,
 * (t - limit) / margin is sampled (margin: the skipping rule's own, eps' * |1/d_axis|): min_slack_* = how far beyond the limit the
 * nearest skipped primitive was (+inf: no sample; a box whose own plane parameter IS the slab entry sits at 1 + a rounding).  Reference traversal only (not the fast mode); the pruned walk is forced
 * on for this render whatever the accel's setting. */
typedef struct lg_prune_audit {
    uint64_t skipped_nodes, skipped_runs, primitives, violations;
    double min_slack_nodes, min_slack_runs;
    double max_margin_used_nodes; /* largest (slab entry parameter - t) / margin over the primitives under skipped nodes: the share of the
                                   * shipped margin that property (P) actually needed on this scene (1 would be the edge of a violation) */
} lg_prune_audit;
int lg_audit_prune(const lg_accel *, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, lg_prune_audit *out);
/* The same for the opt-in FAST mode (lg_accel_set_mode(1)), end to end: every ray of rows [y0, y1) -- primary, shadow and secondary, as the
 * render generates them -- is traced by the fast walk as shipped (its trees, its pruning, its candidate check and tie fallback) AND by the
 * reference walk, and the answers are compared on the device: the same primitive in the same accel at the same t, bit for bit, for a
 * closest-hit ray; the same `t < 1` verdict for a shadow ray (point.rs:49).  `violations` must be 0 for the film to be the reference's;
 * `fallbacks` counts the rays the fast walk itself sent to the reference walk (exact ties, winners the reference tree would not have tested).
 * Fast mode's margins are argued, not derived: this is its measurement, frame by frame.  The accel must be in fast mode. */
typedef struct lg_fast_audit { uint64_t rays, fallbacks, violations; } lg_fast_audit;
int lg_audit_fast(const lg_accel *, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, lg_fast_audit *out);

/* Traversal mode of an accel.  0 (default) = the reference's own traversal over the reference's
 * own BVH: the parity path.  1 = opt-in FAST mode: a binned-SAH BVH (one primitive per leaf)
 * over the same primitives, front-to-back with pruning beyond the best hit; same primitive tests
 * and arithmetic.  Its winner is put to the reference tree's own box tests (leaf to root, through every
 * nested accel); a winner that fails them, or an exact tie in t, re-traces the ray with the reference
 * traversal.  Verified byte-identical to mode 0 on every test and benchmark config; its two rounding margins are argued, not
 * PROVEN, and it CAN differ from the reference where a ray lies within rounding of a far triangle's plane (DESIGN.md section 3.3:
 * 5 pixels in 4,100 fuzz scenes whose meshes span ten orders of magnitude).  Refused -- non-zero return, lg_last_error -- for a scene with a
 * transform that does not invert (rotate() about a non-unit axis) or a mesh whose largest |coordinate| exceeds 2^20 times the longest
 * edge of its smallest triangle.  The fast trees cost 5-10x the
 * reference build, so lg_accel_from does not build them: the first lg_accel_set_mode(accel, 1) does (it
 * synchronises the device and uploads the tables again; the Scene must still be alive, as for any use of the accel). */
int lg_accel_set_mode(const lg_accel *, int mode);
/* Pruned form of the reference traversal (mode 0 only): the reference's tree and visit order, but a node is skipped when on
 * some axis the ray reaches its slab only beyond the best accepted hit (closest hit) or beyond the light (shadow rays) by
 * more than a margin derived from the rounding of the reference's own intersection formulas: every
 * primitive below such a node would be rejected by the reference's `t >= isect.t` (sphere.rs:86, cuboid.rs:95,
 * triangle.rs:251), so every pixel is what the unpruned walk gives.  Nodes over a nested BVHAccel are never skipped; inside a
 * mesh only the ray's dominant axis counts.  -1 (default): on for scenes that carry a mesh of >= 4096 triangles (the
 * reference's 254-triangle leaves are where it pays; below that it measured 5-30 % slower and its leaf records are a third to a half of
 * the accel build: profiles/r05_prune_threshold.jsonl -- rounds 3-5 had 256), off otherwise; 0 / 1: off / on (1 on an accel whose
 * tables were built without the leaf records builds them then). */
int lg_accel_set_prune(const lg_accel *, int enabled);
int lg_accel_get_prune(const lg_accel *); /* the effective setting (accel default, LASGUN_PRUNE, lg_accel_set_prune, fast mode): 0 / 1 */

/* Kernel organisation (same arithmetic, same bytes either way).  1 (default): the organisation of a launch is MEASURED -- the SECOND
 * API CALL that launches a kind in the process (the scene's shape, the device, the launch's size class and addressing mode) renders with
 * every organisation that can take it, on the caller's stream WITH THE HOST WAITING (this is the one place where a *_device entry point
 * blocks: once per kind; never on a stream that is being captured into a graph; pin choices with lg_tune_import, or set
 * LASGUN_AUTOTUNE=0, where that cannot be had), and the fastest is kept for the process (capture() rebuilds its accel per frame, so the
 * memory is keyed by the scene's shape); later launches of the kind only enqueue.  An organisation that cannot run (no memory for its
 * buffers) drops out of the measurement instead of failing the render.  The FIRST call that launches a kind -- all of its launches: the
 * four row bands of a big lg_capture, the shares of lg_multi_* -- takes the fitted rule's choice, so that a program that renders one
 * frame and exits pays nothing for a measurement worth 30-50 frames (LASGUN_AUTOTUNE=2: measure at the first call already --
 * benchmarks).  LASGUN_AUTOTUNE=0 keeps
 * the fitted rule of rounds 2-4 throughout: level by level in the WAVEFRONT pipeline (below) for scenes with <= 32 lights and at least
 * 512 spheres / boxes from 2^21 pixels a launch, and for a scene small enough to live in LDS glass / mirror frames of up to 2^20
 * pixels and plain frames of one sample per pixel from 2^18; the queue organisation for glass / mirror over a big mesh; the single
 * persistent megakernel for everything else.  0 = the megakernel only.
 * 2 = use the pipeline wherever it is possible (tests).
 * 3 = the QUEUE organisation wherever it is possible (reference traversal, <= 32 lights, recursion depth <= 7): ONE persistent
 * launch per chunk of the film whose waves pull 64-ray packets from per-level ray queues -- level 0's packets are the 8x8 pixel
 * tiles, level d + 1's are filled by level d's glass / mirror hits -- deepest level first, each packet taken through closest hit,
 * shadow rays and shading by the wave that claimed it; the levels are combined bottom-up as in the wavefront pipeline.  It is
 * the default for scenes with glass / mirror over a big mesh (long, uneven walks; sparse deep levels), where the megakernel
 * runs deep levels with a few lanes per wave and the level-by-level pipeline ends every launch with its slowest wave's tail.
 * Returns non-zero (lg_last_error) for any other value. */
int lg_accel_set_streaming(const lg_accel *, int enabled);
/* The table of measured choices from outside.  An entry is a kind of launch (twelve opaque words) and the choice remembered for it (the
 * encoding of lg_accel_last_organisation).  lg_tune_export writes up to `capacity` entries and returns how many the table holds
 * (out = NULL: just the count); lg_tune_import pins entries -- a kind that has one is never measured, and an entry the launch cannot
 * take (another build, less memory) falls back to the fitted rule's choice instead of failing; lg_tune_clear forgets every choice and
 * every first sight.  A caller that knows its workload exports once and imports at start-up; a test runs on a fixed table.
 * LASGUN_TUNE_FILE=<path> does the same without a line of code: the table is read from the file before the first look-up and rewritten after every
 * choice that is remembered, so a program that renders one frame and exits -- and therefore never measures -- runs on what an earlier run
 * (LASGUN_AUTOTUNE=2, or any program that rendered the kind twice) measured. */
typedef struct lg_tune_entry { uint64_t key[12]; int32_t choice; int32_t reserved; } lg_tune_entry;
size_t lg_tune_export(lg_tune_entry *out, size_t capacity);
int lg_tune_import(const lg_tune_entry *entries, size_t count);
void lg_tune_clear(void);
int lg_accel_last_organisation(const lg_accel *); /* what the accel's last launch ran as: 0 megakernel, 1 level by level, 2 queue, + 16 when its tiles were claimed bottom-up, + 64 when from the middle row outwards, + 128 when the megakernel handed its tiles out in parts, + 32 when the megakernel took a supersampled pixel's samples one after the other (otherwise side by side: every organisation's way since round 5); -1: none yet */
/* The direction in which the megakernel and the queue organisation claim a launch's 8x8 tiles: 0 = from the film's top (row order), 1 = from
 * its bottom, 2 = from its middle row outwards (what a frame shows tends to sit in its middle, and a launch should END on cheap tiles: the
 * 100k-triangle glass torus 36.2 -> 32.8 ms in the megakernel, profiles/r05_ab_tile_middle.jsonl), -1 (default) = middle-out unless the
 * measurement above finds another faster (a launch ends with the recursion trees of its last tiles: 6-9 % either way on scenes with
 * mirrors / glass, profiles/r05_ab_tile_order.jsonl).  Which tile is rendered when never changes a pixel. */
int lg_accel_set_tile_order(const lg_accel *, int order);
/* A supersampled pixel's samples (Camera::sample, camera.rs:113-146; integrate.rs:17-20 sums them in their order): 0 = SIDE BY SIDE -- a
 * launch's level-0 work items are (8x8 tile, sample) pairs, every sample's li() is parked as three doubles and a resolve pass sums each
 * pixel's samples in the reference's order, so the film is the same bytes; a 9-sample 512^2 frame then fills the machine nine times over
 * instead of running nine thin launch chains (level by level, queue) or nine samples in a row on each wave (megakernel): 1.2 - 10 x on the
 * reference's example scenes at their own sizes (profiles/r05_ss_par.jsonl).  1 = one sample after the other (rounds 1-4).  -1 (default) =
 * side by side, except that the megakernel's form is one more thing the measurement above times (frames of 1024^2 and more of a cheap
 * scene run faster with the samples in a row: a ninth of the tile claims). */
int lg_accel_set_sample_order(const lg_accel *, int order);
/* The work item of the megakernel and of the queue organisation's level 0: a whole 8x8 tile per wave (1), or a tile in 2 / 4 / 8 parts of
 * 32 / 16 / 8 lanes each.  A launch of fewer tiles than the grid has waves is as slow as its slowest tile's recursion tree; a part is a shorter
 * tree, four times the waves are at work, and the queue organisation's deeper packets stay as narrow (the 100k-triangle glass torus at 256^2
 * 8.46 -> 5.42 ms in the queue organisation, the metal torus 2.19 -> 1.81 in the megakernel; cheap scenes and big frames lose:
 * profiles/r05_ab_split.jsonl, r05_ab_split_orgs.jsonl).  -1 (default) = whole tiles unless the measurement above finds quarters faster for a small launch
 * (lg_accel_last_organisation: + 128).  Same bytes either way. */
int lg_accel_set_tile_parts(const lg_accel *, int parts);

/* The WAVEFRONT pipeline is li() level by level: per recursion level a closest-hit pass (hits compacted into a queue, misses
 * finished on the spot), an any-hit shadow pass and a shade pass that appends the specular children to the next level's ray
 * queue, then the levels are combined bottom-up in the reference's (output + reflected) + refracted order.  It serves every
 * scene, glass / mirror included.  (lg_accel_set_wavefront and lg_accel_set_packet -- the switches of round 1's three-kernel
 * pipeline and of the packet walk, both slower than this on every measured scene -- left the ABI in round 4.) */

/* Wavefront pipeline, launches of 2 Mpixel and more: cut the launch into `bands` row bands (2..8) rendered on internal streams,
 * each with launch state of its own, so one band's closest pass fills the tails of another band's shadow and shade passes
 * and the sparse deeper levels of a recursive scene run beside other bands' level 0; the caller's stream forks into them
 * and joins them, results are unchanged.  0 = the default (LASGUN_WF_SPLIT, else 1 = off).  Worth it for a caller that
 * renders one frame at a time (headline frame 7.79 -> 7.50 ms with 4 bands); a caller that keeps several frames in
 * flight on streams of its own already has that overlap and loses with it (7.20 -> 7.46 ms).  The internal streams overlap
 * only if the HIP runtime gives them hardware queues of their own (GPU_MAX_HW_QUEUES, default 4, counts every stream of
 * the process).  No counterpart in the reference. */
int lg_accel_set_wf_split(const lg_accel *, int bands);

/* LDS-resident scene (reference traversal; streaming pipeline and megakernel): when the scene's node,
 * primref, sphere and cuboid tables fit beside 1024 per-lane stacks in one CU's 160 KB of LDS, the
 * kernels that traverse run as one 1024-lane workgroup per CU that copies those tables into LDS once
 * and walks them there (same records, same arithmetic, same bytes out).  On by default; returns
 * 1 when the accel's scene qualifies, 0 when it does not (the setting is then without effect). */
int lg_accel_set_lds_scene(const lg_accel *, int enabled);


/* Kernel timing with HIP events on the launch stream: enable, render, then read. */
void lg_profile_enable(const lg_accel *, int enabled);
int lg_profile_read(const lg_accel *, double *total_ms, uint64_t *launches); /* synchronises; resets the tally */
/* Per-kernel HIP-event time.  kind 0 closest-hit trace (+ shading frame), 1 combine, 2 shadow trace, 3 shade; 4 = the megakernel
 * or the queue organisation's persistent kernel. */
int lg_profile_read_kinds(const lg_accel *, double ms[5], uint64_t launches[5]);
/* lg_capture_stats restricted to one kind of traversal: 0 all, 1 closest-hit (primary/secondary), 2 shadow. */
int lg_capture_stats_kind(const lg_accel *, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, int kind, lg_stats *out);

/* Scene-structure introspection (tests) */
int lg_accel_dump(const lg_accel *, const double **f, size_t *nf, const int64_t **i, size_t *ni);
int lg_accel_info(const lg_accel *, uint64_t out[8]); /* nodes, primrefs, spheres, cuboids, triangles, accels, max_stack, bytes */
void lg_aggregate_get_transform(lg_aggregate *, double m[16], double minv[16]);
/* Host-only HLBVH build + flatten of a scene, no device needed: structure dump + counts
 * (nodes, primrefs, spheres, cuboids, triangles, accels, max_stack, has_specular). */
int lg_host_build_dump(const lg_scene *, const double **f, size_t *nf, const int64_t **i, size_t *ni, uint64_t info[8]);
/* Host-only self-check of the fast mode's wide node records against the binary fast trees they are collapsed from (no device):
 * out = { records, children, leaves reached, deepest stack of a walk that pushes every child but one, violations, the stack
 * depth the flattening reserved, 0, 0 }.  Violations: a child box not containing its node's box, a leaf reached twice or
 * never, a dangling link. */
int lg_host_check_wide_records(const lg_scene *, uint64_t out[8]);
/* Host-only self-check of the triangle strips the pruned walk's leaf loop streams (no device): out = { mesh leaves with culling
 * records, runs, triangles, strip entries, violations, FNV-1a of the culling records and the leaves' record words, FNV-1a of the strip
 * entries, 0 } (the hashes: the tables must not depend on how many host threads made them, LASGUN_HOST_THREADS).  Violations: a triangle of such a leaf that is not exactly one
 * strip triangle, a strip triangle whose three vertices are not its slot's three vertices, a run whose counts disagree. */
int lg_host_check_strips(const lg_scene *, uint64_t out[8]);

/* One pixel traced by a single lane (sample 0): out = { t, primref, accel instance, number of lights, then per light
 * the shadow ray's { t, primref }, then the shadow rays' origin (3) } -- primref is 4294967295 for "no hit"; out_len >=
 * 7 + 2 * lights.  `fast` selects the traversal mode.
 * With out_len >= 7 + 2 * lights + 16001 the event log of the primary ray's walk follows: a count, then up to 4000
 * events of 4 doubles (code, a, b, c: node pairs / nodes visited, primitive tests, accel entries and returns; the codes
 * are listed at lasgun_amd.HipApi.trace_pixel_log).  A debugging / test hook: it is how a film difference is traced
 * back to the ray, and the ray to the box, that caused it. */
int lg_trace_pixel(const lg_accel *, uint32_t width, uint32_t height, uint32_t x, uint32_t y, int fast, double *out, size_t out_len);

/* Known-answer and arithmetic probes: run the DEVICE intersectors / math on one thread. */
int lg_kat_intersect(int kind, const double *params, const char *obj_text, size_t obj_len, const double origin[3],
                     const double d[3], double out[8]);
int lg_kat_surface_interaction(const double origin[3], const double d[3], double t, const double dpdu[3],
                               const double dpdv[3], double out_ng[3]);
int lg_math_eval(int op, size_t n, const double *a, const double *b, double *out);
/* Measured rates of the current device in GB/s: what 0 = HBM copy (16 bytes per lane, 1 GiB, read + written bytes),
 * 1 = aggregate LDS read rate (ds_read_b128, every CU streaming): the measured denominators of bench.py's roofline. */
int lg_probe_rate(int what, double *gbps);

#ifdef __cplusplus
}
#endif
#endif /* LASGUN_HIP_H */
