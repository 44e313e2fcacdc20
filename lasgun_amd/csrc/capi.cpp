// lasgun_amd/csrc/capi.cpp -- the C ABI of include/lasgun_hip.h over the host-side units (internal.h): host objects, captures, switches,
// measurement and test hooks.  No torch types, no C++ exceptions across the boundary, no CPU render path.
#include "internal.h"

// ============================================================================================
extern "C" {

const char *lg_last_error(void) { return tl_error.c_str(); }

// ---- the measured choice's table from outside (tune.h): a caller that knows its workload pins the choice and never pays for a race;
// a test runs with a fixed table.  An entry is the twelve words of a kind and the remembered choice, opaque to the caller.
size_t lg_tune_export(lg_tune_entry *out, size_t capacity) {
    const size_t n = lg::tune::snapshot(nullptr, nullptr, 0);
    if (!out || capacity == 0) return n;
    std::vector<lg::tune::Key> keys(capacity);
    std::vector<int> choices(capacity);
    const size_t m = lg::tune::snapshot(keys.data(), choices.data(), capacity);
    for (size_t i = 0; i < std::min(m, capacity); ++i) {
        std::memcpy(out[i].key, keys[i].v, sizeof out[i].key);
        out[i].choice = choices[i];
        out[i].reserved = 0;
    }
    return m;
}
int lg_tune_import(const lg_tune_entry *entries, size_t count) {
    if (count != 0 && !entries) return fail("lg_tune_import: entries is NULL");
    for (size_t i = 0; i < count; ++i) { // (a choice is checked against the launch when it is used: one the launch cannot take falls back to the rule's)
        if (entries[i].choice < 0 || entries[i].choice >= 256 || (entries[i].choice & 15) > 2 /* the queue organisation */) return fail("lg_tune_import: entry " + std::to_string(i) + " holds no choice this library makes");
    }
    for (size_t i = 0; i < count; ++i) {
        lg::tune::Key k;
        std::memcpy(k.v, entries[i].key, sizeof k.v);
        lg::tune::remember(k, entries[i].choice);
    }
    return 0;
}
void lg_tune_clear(void) { lg::tune::clear(); }
void lg_set_last_error(const char *msg) { tl_error = msg ? msg : ""; } // (multi.cpp reports through the same thread-local message)

static lg_material pack(const Material &m) { lg_material r; r.kind = m.kind; std::memcpy(r.p, m.p, sizeof r.p); return r; }
static Material unpack(const lg_material *m) { Material r; r.kind = m->kind; std::memcpy(r.p, m->p, sizeof r.p); return r; }
static lg_material mat2(int kind, const double a[3], const double b[3], double s0, double s1) {
    lg_material m{}; m.kind = kind;
    for (int i = 0; i < 3; ++i) { m.p[i] = a[i]; if (b) m.p[3 + i] = b[i]; }
    m.p[6] = s0; m.p[7] = s1;
    return m;
}
lg_material lg_material_default(void) { return pack(material_default()); }
lg_material lg_material_matte(const double kd[3], double sigma) { return pack(material_matte(kd, sigma)); }
lg_material lg_material_plastic(const double kd[3], const double ks[3], double roughness) { return mat2(MAT_PLASTIC, kd, ks, roughness, 0.0); }
lg_material lg_material_metal(const double eta[3], const double k[3], double u, double v) { return mat2(MAT_METAL, eta, k, u, v); }
lg_material lg_material_glass(const double kr[3], const double kt[3], double eta) { return mat2(MAT_GLASS, kr, kt, eta, 0.0); }
lg_material lg_material_mirror(const double kr[3]) { return mat2(MAT_MIRROR, kr, nullptr, 0.0, 0.0); }

lg_scene *lg_scene_new(void) { return new lg_scene(); }
void lg_scene_free(lg_scene *s) { delete s; }
void lg_scene_set_perspective_camera(lg_scene *s, double fov) { s->s.camera.init(true, fov); }
void lg_scene_set_orthographic_camera(lg_scene *s, double scale) { s->s.camera.init(false, scale); }
void lg_camera_look_at(lg_scene *s, const double o[3], const double l[3], const double u[3]) {
    s->s.camera.look_at(V3{o[0], o[1], o[2]}, V3{l[0], l[1], l[2]}, V3{u[0], u[1], u[2]});
}
void lg_camera_set_supersampling(lg_scene *s, uint8_t base) { s->s.camera.set_supersampling(base); }
void lg_camera_set_aperture_radius(lg_scene *s, double r) { s->s.camera.aperture_radius = r; }
void lg_scene_set_solid_background(lg_scene *s, const double c[3]) {
    s->s.bg_inner = V3{c[0], c[1], c[2]}; s->s.bg_outer = s->s.bg_inner; s->s.bg_scale = 1.0; // background.rs:18-20
}
void lg_scene_set_radial_background(lg_scene *s, const double i[3], const double o[3], double scale) {
    s->s.bg_inner = V3{i[0], i[1], i[2]}; s->s.bg_outer = V3{o[0], o[1], o[2]}; s->s.bg_scale = scale;
}
void lg_scene_set_ambient_light(lg_scene *s, const double c[3]) { s->s.ambient = V3{c[0], c[1], c[2]}; }
void lg_scene_set_mesh_smoothing(lg_scene *s, int e) { s->s.smoothing = e != 0; }
void lg_scene_set_max_recursion_depth(lg_scene *s, uint32_t d) { s->s.recursion = d; }
void lg_scene_set_threads(lg_scene *s, size_t t) { s->s.threads = t; }
void lg_scene_add_point_light(lg_scene *s, const double p[3], const double i[3], const double f[3]) {
    Light l;
    std::memcpy(l.pos, p, sizeof l.pos); std::memcpy(l.intensity, i, sizeof l.intensity); std::memcpy(l.falloff, f, sizeof l.falloff);
    s->s.lights.push_back(l);
}
int lg_scene_parse_obj(lg_scene *s, const char *text, size_t len, uint32_t *out_ref) { // scene.rs:109-123
    return guarded([&] {
        std::unique_ptr<Obj> obj(new Obj());
        parse_obj_text(text, len, *obj);
        if (!s->s.smoothing) obj->normal.clear();
        *out_ref = (uint32_t)s->s.meshes.size();
        s->s.meshes.push_back(std::move(obj));
    });
}
int lg_scene_load_obj(lg_scene *s, const char *path, uint32_t *out_ref) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(std::string("cannot open ") + path);
    std::string buf;
    char tmp[65536];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.append(tmp, n);
    std::fclose(f);
    return lg_scene_parse_obj(s, buf.data(), buf.size(), out_ref);
}
lg_aggregate *lg_scene_root(lg_scene *s) { return reinterpret_cast<lg_aggregate *>(s->s.root.get()); }
void lg_scene_set_root(lg_scene *s, lg_aggregate *moved) {
    s->s.root.reset(new Aggregate(std::move(moved->a)));
    delete moved;
}

// lg_aggregate is layout-compatible with its only member, so a borrowed `Aggregate*` (scene
// root) can be handed out as lg_aggregate*.
static_assert(sizeof(lg_aggregate) == sizeof(Aggregate), "lg_aggregate must wrap Aggregate exactly");
lg_aggregate *lg_aggregate_new(void) { return new lg_aggregate(); }
void lg_aggregate_free(lg_aggregate *a) { delete a; }
static SceneNode node_of(SceneNode::Kind k) { SceneNode n; n.kind = k; n.mat = material_default(); return n; }
void lg_aggregate_add_group(lg_aggregate *a, lg_aggregate *moved) {
    SceneNode n = node_of(SceneNode::GROUP);
    n.group.reset(new Aggregate(std::move(moved->a)));
    delete moved;
    a->a.contents.push_back(std::move(n));
}
void lg_aggregate_add_sphere(lg_aggregate *a, const double c[3], double r, const lg_material *m) {
    SceneNode n = node_of(SceneNode::SPHERE);
    std::memcpy(n.a, c, sizeof n.a); n.b[0] = r; n.mat = unpack(m); n.has_mat = true;
    a->a.contents.push_back(std::move(n));
}
void lg_aggregate_add_cube(lg_aggregate *a, const double o[3], double dim, const lg_material *m) {
    SceneNode n = node_of(SceneNode::CUBE);
    std::memcpy(n.a, o, sizeof n.a); n.b[0] = dim; n.mat = unpack(m); n.has_mat = true;
    a->a.contents.push_back(std::move(n));
}
void lg_aggregate_add_box(lg_aggregate *a, const double mn[3], const double mx[3], const lg_material *m) {
    SceneNode n = node_of(SceneNode::CUBOID);
    std::memcpy(n.a, mn, sizeof n.a); std::memcpy(n.b, mx, sizeof n.b); n.mat = unpack(m); n.has_mat = true;
    a->a.contents.push_back(std::move(n));
}
void lg_aggregate_add_obj(lg_aggregate *a, uint32_t mesh) {
    SceneNode n = node_of(SceneNode::MESH); n.obj = mesh;
    a->a.contents.push_back(std::move(n));
}
void lg_aggregate_add_obj_of(lg_aggregate *a, uint32_t mesh, const lg_material *m) {
    SceneNode n = node_of(SceneNode::MESH); n.obj = mesh; n.mat = unpack(m); n.has_mat = true;
    a->a.contents.push_back(std::move(n));
}
void lg_aggregate_swap_backface(lg_aggregate *a) { a->a.swap_backface = !a->a.swap_backface; }
void lg_aggregate_translate(lg_aggregate *a, const double d[3]) { transform_concat_self(a->a.transform, transform_translate(d)); }
void lg_aggregate_scale(lg_aggregate *a, double x, double y, double z) { transform_concat_self(a->a.transform, transform_scale(x, y, z)); }
void lg_aggregate_rotate_x(lg_aggregate *a, double t) { transform_concat_self(a->a.transform, transform_rotate_x(t)); }
void lg_aggregate_rotate_y(lg_aggregate *a, double t) { transform_concat_self(a->a.transform, transform_rotate_y(t)); }
void lg_aggregate_rotate_z(lg_aggregate *a, double t) { transform_concat_self(a->a.transform, transform_rotate_z(t)); }
void lg_aggregate_rotate(lg_aggregate *a, double t, const double axis[3]) { transform_concat_self(a->a.transform, transform_rotate(t, axis)); }
void lg_aggregate_get_transform(lg_aggregate *a, double m[16], double minv[16]) {
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) { m[4 * c + r] = a->a.transform.m.m[c][r]; minv[4 * c + r] = a->a.transform.minv.m[c][r]; }
}

lg_film *lg_film_new(uint32_t w, uint32_t h) {
    lg_film *f = new lg_film();
    f->w = w; f->h = h;
    f->owned.assign((size_t)w * h * 4, 0);
    f->px = f->owned.data();
    return f;
}
lg_film *lg_film_wrap(uint32_t w, uint32_t h, uint8_t *rgba) {
    lg_film *f = new lg_film();
    f->w = w; f->h = h; f->px = rgba;
    return f;
}
uint8_t *lg_film_pixels(lg_film *f) { return f->px; }
uint32_t lg_film_width(lg_film *f) { return f->w; }
uint32_t lg_film_height(lg_film *f) { return f->h; }
void lg_film_free(lg_film *f) { delete f; }

int lg_set_device(int device) {
    g_device = device;
    g_device_chosen = true;
    return guarded([] { use_device(); });
}
int lg_set_devices(const int *ids, int count) {
    return guarded([&] {
        if (count < 0 || (count > 0 && !ids)) throw Error("bad device list");
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw Error("no HIP device available: liblasgun_hip has no CPU fallback");
        std::vector<int> v;
        if (count == 0) for (int i = 0; i < n; ++i) v.push_back(i); // 0 ids = every visible device
        for (int i = 0; i < count; ++i) {
            if (ids[i] < 0 || ids[i] >= n) throw Error("device index out of range");
            v.push_back(ids[i]); // an index may repeat: its shares then run concurrently on that device
        }
        g_devices = v;
    });
}
uint64_t lg_trim_pool(int device) { return (uint64_t)g_pool.trim(device); }
int lg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Host build (reference trees; with `with_fast` also the fast mode's), upload, and everything derived from the tables.
// Called at creation without the fast trees -- they cost 5-10x the reference build and only mode 1 walks them -- and
// once more, with them, by the first lg_accel_set_mode(accel, 1).

lg_accel *lg_accel_from(const lg_scene *s) { return accel_from_on(s, g_device); }
lg_accel *lg_accel_from_on(const lg_scene *s, int device) { return accel_from_on(s, device); }
void lg_accel_free(lg_accel *a) {
    if (!a) return;
    // its buffers go back to the pool and may be handed out again at once: nothing on any stream may still use them
    // (hipFree used to imply the same wait)
    if (hipSetDevice(a->device) == hipSuccess) (void)hipDeviceSynchronize();
    delete a;
}
void *lg_accel_stream(const lg_accel *a) { return (void *)a->stream; }
int lg_accel_synchronize(const lg_accel *a) {
    return guarded([&] { std::lock_guard<std::mutex> g(a->mtx); use_device(a->device); sync_checked(*a); });
}

int lg_capture_rows_device(const lg_accel *a, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t row0, void *dev_rgba, void *hip_stream) {
    return guarded([&] {
        if (y1 > h || y0 > y1 || row0 > y0) throw Error("bad row range");
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        DParams P = base_params(*a, w, h);
        set_rect(P, 0, y0, w, y1);
        P.out_row0 = row0;
        P.out_rgba = (uint8_t *)dev_rgba;
        enqueue(*a, P, false, (hipStream_t)hip_stream);
    });
}
int lg_capture_interleaved_device(const lg_accel *a, uint32_t w, uint32_t h, uint32_t block_rows, uint32_t n, uint32_t r, void *dev_rgba, void *hip_stream) {
    return guarded([&] {
        if (n == 0 || r >= n || block_rows == 0 || h % (block_rows * n) != 0) throw Error("height must be a multiple of block_rows * n");
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        DParams P = base_params(*a, w, h);
        set_rect(P, 0, 0, w, h / n); // virtual rows of the compact tile
        P.ilv_n = n; P.ilv_r = r; P.ilv_b = block_rows;
        P.out_row0 = 0;
        P.out_rgba = (uint8_t *)dev_rgba;
        enqueue(*a, P, false, (hipStream_t)hip_stream);
    });
}
int lg_capture_subset_device(size_t k, size_t n, const lg_accel *a, uint32_t w, uint32_t h, void *dev_rgba, void *hip_stream) {
    return guarded([&] {
        if (n == 0) throw Error("n must be > 0");
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        DParams P = base_params(*a, w, h);
        if (n == 1 && k == 0) set_rect(P, 0, 0, w, h);
        else set_subset(*a, (hipStream_t)hip_stream, P, k, n, w, h);
        P.out_row0 = 0;
        P.out_rgba = (uint8_t *)dev_rgba;
        enqueue(*a, P, false, (hipStream_t)hip_stream);
    });
}

// capture_subset into a HOST film (lib.rs:110-162).  The reference calls this from n threads at once on one
// Accel and one film (lib.rs:67-103: write-disjoint pixel sets), so concurrent calls must not disturb each other:
// every call renders into a device buffer of its own -- the whole film for (0, 1), otherwise a COMPACT buffer of
// just the subset's pixels, one word per work item -- and only the owned pixels {k + i*n} of the host film are
// written.  Nothing is uploaded and no other pixel of the film is touched (lib.rs:152).
int lg_capture_subset(size_t k, size_t n, const lg_accel *a, lg_film *film) {
    return guarded([&] {
        if (n == 0) throw Error("n must be > 0");
        const uint32_t w = film->w, h = film->h;
        const unsigned long long area = (unsigned long long)w * h;
        const bool whole = n == 1 && k == 0;
        const unsigned long long count = whole ? area : subset_count(area, k, n);
        if (count == 0) return;
        DevBuf<uint32_t> buf; // this call's own output (returned to the pool when the call ends)
        // A whole film of 2^22 pixels and more goes out in ROW BANDS: the bands are rendered one after the other on the accel's stream, and
        // each is copied home on a second stream as soon as it is done -- band j's 16 MiB cross PCIe while band j + 1 renders; round 4
        // issued ONE copy of the whole film after the last kernel (config 3: 1.2 of capture()'s 9.96 ms).  Same pixels, same bytes: a
        // pixel's value does not depend on the launch it is rendered in.  (Bands side by side on four streams were measured first: their
        // persistent grids share the chip, all four finish together and the copies still come last -- 8.57 against 8.29 ms.)
        static const unsigned bands_env = [] { const char *e = std::getenv("LASGUN_CAPTURE_BANDS"); return e ? (unsigned)std::atoi(e) : 4u; }();
        const unsigned nbands = whole && area >= (1ull << 22) && bands_env >= 2 && h >= 64 ? std::min(bands_env, 16u) : 1u;
        const uint32_t rows8 = (h + 7u) / 8u; // bands are whole rows of 8x8 tiles
        auto band_rows = [&](unsigned j, uint32_t &y0, uint32_t &y1) {
            y0 = (uint32_t)((unsigned long long)rows8 * j / nbands) * 8u;
            y1 = std::min(h, (uint32_t)((unsigned long long)rows8 * (j + 1) / nbands) * 8u);
        };
        hipStream_t copy_stream = nullptr;
        struct Events { // (destroyed on every way out: an enqueue that throws must not leak the bands' events)
            std::vector<hipEvent_t> v;
            ~Events() { for (hipEvent_t e : v) (void)hipEventDestroy(e); }
        } rendered_events;
        std::vector<hipEvent_t> &rendered = rendered_events.v;
        {
            std::lock_guard<std::mutex> g(a->mtx);
            use_device(a->device);
            buf.alloc((size_t)count);
            DParams P = base_params(*a, w, h);
            P.out_row0 = 0;
            P.out_rgba = (uint8_t *)buf.p;
            if (nbands > 1) {
                ensure_aux_streams(*a, 1);
                copy_stream = a->aux_streams[0];
                try {
                    for (unsigned j = 0; j < nbands; ++j) {
                        uint32_t y0, y1;
                        band_rows(j, y0, y1);
                        hipEvent_t ev = nullptr;
                        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                        rendered.push_back(ev);
                        if (y1 > y0) {
                            DParams B = P;
                            set_rect(B, 0, y0, w, y1);
                            enqueue(*a, B, false, a->stream);
                        }
                        HIP_TRY(hipEventRecord(ev, a->stream));
                    }
                } catch (...) {
                    (void)hipStreamSynchronize(a->stream); // (the bands already enqueued still write `buf`, which goes back to the pool on the way out)
                    throw;
                }
            } else {
                if (whole) set_rect(P, 0, 0, w, h);
                else { set_subset(*a, a->stream, P, k, n, w, h); P.out_compact = 1; }
                enqueue(*a, P, false, a->stream);
                if (whole) HIP_TRY(hipMemcpyAsync(film->px, buf.p, (size_t)area * 4, hipMemcpyDeviceToHost, a->stream));
            }
        }
        if (nbands > 1) { // the copies, in band order, every render already enqueued (a copy into pageable memory holds this thread until its band is home)
            use_device(a->device);
            hipError_t err = hipSuccess;
            for (unsigned j = 0; j < nbands && err == hipSuccess; ++j) {
                uint32_t y0, y1;
                band_rows(j, y0, y1);
                if (y1 <= y0) continue;
                const size_t off = (size_t)y0 * w * 4, bytes = (size_t)(y1 - y0) * w * 4;
                err = hipStreamWaitEvent(copy_stream, rendered[j], 0);
                if (err == hipSuccess) err = hipMemcpyAsync(film->px + off, (const uint8_t *)buf.p + off, bytes, hipMemcpyDeviceToHost, copy_stream);
            }
            if (err == hipSuccess) err = hipStreamSynchronize(copy_stream);
            const hipError_t err2 = hipStreamSynchronize(a->stream); // (nothing may still write `buf` when it goes back to the pool)
            if (err != hipSuccess || err2 != hipSuccess) throw Error(std::string("banded capture: ") + hipGetErrorString(err != hipSuccess ? err : err2));
            std::lock_guard<std::mutex> g(a->mtx);
            check_queue_error(*a);
            return;
        }
        if (whole) { use_device(a->device); sync_checked(*a); return; }
        std::vector<uint32_t> host((size_t)count);
        use_device(a->device);
        HIP_TRY(hipMemcpyAsync(host.data(), buf.p, (size_t)count * 4, hipMemcpyDeviceToHost, a->stream));
        sync_checked(*a);
        uint8_t *px = film->px;
        for (unsigned long long i = 0; i < count; ++i) std::memcpy(px + 4 * (k + i * n), &host[(size_t)i], 4);
    });
}
// Several subsets of one n in ONE render (no counterpart in the reference, whose progressive caller -- www/renderer.ts:103-120 -- makes a
// hundred capture_subset calls one after the other; each is then a launch chain that fills a fraction of the GPU).  The pixels written
// are exactly those of the `count` calls lg_capture_subset(ks[j], n, ...), every other pixel is left alone; a batch that lists every
// k of 0 .. n-1 is the frame and is rendered as one (8x8 tiles).
int lg_capture_subsets_device(const size_t *ks, size_t count, size_t n, const lg_accel *a, uint32_t w, uint32_t h, void *dev_rgba, void *hip_stream) {
    return guarded([&] {
        const SubsetBatch b = make_batch(ks, count, n, w, h);
        if (b.items == 0) return;
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        DParams P = base_params(*a, w, h);
        try {
            if (b.whole) set_rect(P, 0, 0, w, h);
            else set_subsets(*a, P, b, (hipStream_t)hip_stream);
            P.out_row0 = 0;
            P.out_rgba = (uint8_t *)dev_rgba;
            enqueue(*a, P, false, (hipStream_t)hip_stream);
            if (!b.whole) subsets_enqueued(*a, (hipStream_t)hip_stream);
        } catch (...) { subsets_abandoned(*a, (hipStream_t)hip_stream); throw; }
    });
}
int lg_capture_subsets(const size_t *ks, size_t count, size_t n, const lg_accel *a, lg_film *film) {
    return guarded([&] {
        const uint32_t w = film->w, h = film->h;
        const SubsetBatch b = make_batch(ks, count, n, w, h);
        if (b.items == 0) return;
        if (b.whole) { if (lg_capture_subset(0, 1, a, film)) throw Error(tl_error); return; }
        // as lg_capture_subset: a compact buffer of this call's own (work item i -> word i), one D2H copy for the batch, and only
        // the owned pixels of the host film are written (concurrent callers on one film own disjoint pixels)
        DevBuf<uint32_t> buf;
        std::vector<uint32_t> host((size_t)b.items);
        {
            std::lock_guard<std::mutex> g(a->mtx);
            use_device(a->device);
            buf.alloc((size_t)b.items);
            DParams P = base_params(*a, w, h);
            try {
                set_subsets(*a, P, b, a->stream);
                P.out_compact = 1;
                P.out_row0 = 0;
                P.out_rgba = (uint8_t *)buf.p;
                enqueue(*a, P, false, a->stream);
                subsets_enqueued(*a, a->stream);
            } catch (...) { subsets_abandoned(*a, a->stream); throw; }
            HIP_TRY(hipMemcpyAsync(host.data(), buf.p, (size_t)b.items * 4, hipMemcpyDeviceToHost, a->stream));
        }
        use_device(a->device);
        sync_checked(*a);
        const unsigned long long area = (unsigned long long)w * h, m = b.ks.size();
        uint8_t *px = film->px;
        auto scatter = [&](unsigned long long q0, unsigned long long q1) { // periods [q0, q1): ascending addresses
            for (unsigned long long q = q0; q < q1; ++q)
                for (unsigned long long j = 0; j < m; ++j) {
                    const unsigned long long off = b.ks[(size_t)j] + q * b.n;
                    if (off < area) std::memcpy(px + 4 * off, &host[(size_t)(q * m + j)], 4);
                }
        };
        const unsigned nthreads = b.items >= (1ull << 21) ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
        if (nthreads <= 1) scatter(0, b.periods);
        else {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < nthreads; ++t) pool.emplace_back(scatter, b.periods * t / nthreads, b.periods * (t + 1) / nthreads);
            for (auto &th : pool) th.join();
        }
    });
}
// Any list of pixels of a width x height film (offset = y * width + x), results compact and in list order:
// rgba_out[4*i ..] and/or rgb_out[3*i ..] (f64 radiance before quantisation) for offsets[i].  Test / tooling hook:
// samples and crops of films too large to move whole.
int lg_capture_pixels(const lg_accel *a, uint32_t w, uint32_t h, const uint64_t *offsets, size_t count, uint8_t *rgba_out, double *rgb_out) {
    return guarded([&] {
        if (count == 0) return;
        if (!offsets) throw Error("offsets is NULL");
        const unsigned long long area = (unsigned long long)w * h;
        for (size_t i = 0; i < count; ++i) if (offsets[i] >= area) throw Error("pixel offset outside the film");
        if (count > 0xFFFFFFFFull * 64ull) throw Error("too many pixels");
        DevBuf<unsigned long long> list;
        DevBuf<uint32_t> rgba;
        DevBuf<double> rad;
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        list.alloc(count);
        HIP_TRY(hipMemcpyAsync(list.p, offsets, count * 8, hipMemcpyHostToDevice, a->stream));
        DParams P = base_params(*a, w, h);
        P.mode = 2; P.pixel_list = list.p; P.sub_count = count; P.out_compact = 1;
        P.ntiles = (uint32_t)((count + 63) / 64);
        if (rgba_out) { rgba.alloc(count); P.out_rgba = (uint8_t *)rgba.p; }
        if (rgb_out) { rad.alloc(count * 3); P.out_radiance = rad.p; }
        enqueue(*a, P, false, a->stream);
        if (rgba_out) HIP_TRY(hipMemcpyAsync(rgba_out, rgba.p, count * 4, hipMemcpyDeviceToHost, a->stream));
        if (rgb_out) HIP_TRY(hipMemcpyAsync(rgb_out, rad.p, count * 24, hipMemcpyDeviceToHost, a->stream));
        sync_checked(*a);
    });
}
// The crop [x0, x1) x [y0, y1) of a width x height film, compact and row-major: rgba_out (x1-x0)*(y1-y0)*4 bytes and/or
// rgb_out (x1-x0)*(y1-y0)*3 doubles on the HOST.
int lg_capture_rect(const lg_accel *a, uint32_t w, uint32_t h, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint8_t *rgba_out, double *rgb_out) {
    return guarded([&] {
        if (x1 > w || y1 > h || x0 > x1 || y0 > y1) throw Error("bad rectangle");
        const size_t count = (size_t)(x1 - x0) * (y1 - y0);
        if (count == 0) return;
        DevBuf<uint32_t> rgba;
        DevBuf<double> rad;
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        DParams P = base_params(*a, w, h);
        set_rect(P, x0, y0, x1, y1);
        P.out_row0 = y0; P.out_x0 = x0; P.out_pitch = x1 - x0;
        if (rgba_out) { rgba.alloc(count); P.out_rgba = (uint8_t *)rgba.p; }
        if (rgb_out) { rad.alloc(count * 3); P.out_radiance = rad.p; }
        enqueue(*a, P, false, a->stream);
        if (rgba_out) HIP_TRY(hipMemcpyAsync(rgba_out, rgba.p, count * 4, hipMemcpyDeviceToHost, a->stream));
        if (rgb_out) HIP_TRY(hipMemcpyAsync(rgb_out, rad.p, count * 24, hipMemcpyDeviceToHost, a->stream));
        sync_checked(*a);
    });
}
// One device's share of a multi-device capture: rows of `film` rendered on `device` and copied home.
// With `block_rows` > 0 the share is the 64-row blocks {r, r+n, ...} (even load, one D2H copy per block);
// otherwise the contiguous row tile r of n.
static int capture_share(const lg_scene *s, lg_film *film, int device, uint32_t n, uint32_t r, uint32_t block_rows) {
    lg_accel *a = accel_from_on(s, device);
    if (!a) return 1;
    const uint32_t w = film->w, h = film->h;
    int rc = guarded([&] {
        use_device(device);
        const size_t row_bytes = (size_t)w * 4;
        if (block_rows) {
            const uint32_t rows = h / n;
            a->staging.alloc((size_t)rows * row_bytes);
            if (lg_capture_interleaved_device(a, w, h, block_rows, n, r, a->staging.p, (void *)a->stream)) throw Error(tl_error);
            for (uint32_t g = 0; g < rows / block_rows; ++g) { // block g of the compact tile is image block g*n + r
                const size_t src = (size_t)g * block_rows * row_bytes, dst = ((size_t)g * n + r) * block_rows * row_bytes;
                HIP_TRY(hipMemcpyAsync(film->px + dst, a->staging.p + src, (size_t)block_rows * row_bytes, hipMemcpyDeviceToHost, a->stream));
            }
        } else {
            const uint32_t base = h / n, rem = h % n;
            const uint32_t y0 = r * base + (r < rem ? r : rem), y1 = y0 + base + (r < rem ? 1u : 0u);
            if (y1 > y0) {
                a->staging.alloc((size_t)(y1 - y0) * row_bytes);
                if (lg_capture_rows_device(a, w, h, y0, y1, y0, a->staging.p, (void *)a->stream)) throw Error(tl_error);
                HIP_TRY(hipMemcpyAsync(film->px + (size_t)y0 * row_bytes, a->staging.p, (size_t)(y1 - y0) * row_bytes, hipMemcpyDeviceToHost, a->stream));
            }
        }
        sync_checked(*a);
    });
    lg_accel_free(a);
    return rc;
}

int lg_capture(const lg_scene *s, lg_film *film) { // lib.rs:55-104: the BVH is (re)built inside every capture
    CallScope call; // (one API call, whatever its bands, shares and host threads launch)
    // The reference splits the film over `scene.threads` CPU threads, 0 = all cores (lib.rs:58-62); here the film is
    // split over devices: the ones named with lg_set_devices, the one named with lg_set_device, or -- a process that
    // named none -- EVERY visible device, capped by `scene.threads` when that is non-zero.  Pixels are independent,
    // so the film is the same for any split.
    std::vector<int> devs = g_devices;
    if (devs.empty() && !g_device_chosen) {
        // the default: every visible device -- for films of 2^18 pixels and more.  A smaller film is a millisecond of work on one
        // GPU; building an accel per device and a communicator over them for it would cost seconds (and every rank of a
        // torchrun job that never named a device would do so on every GPU of the node): it stays on the HIP CURRENT device,
        // which is also what a caller that chose its device through the HIP runtime itself expects
        int n_vis = 0, cur = 0;
        if ((unsigned long long)film->w * film->h >= (1ull << 18)) {
            if (hipGetDeviceCount(&n_vis) == hipSuccess)
                for (int d = 0; d < n_vis; ++d) devs.push_back(d);
        } else if (hipGetDevice(&cur) == hipSuccess) devs.push_back(cur);
    }
    if (s->s.threads != 0 && devs.size() > s->s.threads) devs.resize(s->s.threads);
    if (devs.size() <= 1) {
        lg_accel *a = accel_from_on(s, devs.empty() ? g_device : devs[0]);
        if (!a) return 1;
        int rc = lg_capture_subset(0, 1, a, film);
        lg_accel_free(a);
        return rc;
    }
    const uint32_t n = (uint32_t)devs.size();
    const uint32_t block_rows = film->h % (64u * n) == 0 ? 64u : 0u;
    {   // more than one DISTINCT device: the shares are gathered on the first device over xGMI (one grouped RCCL exchange,
        // multi.cpp) and the film leaves the node with one D2H copy -- the north-star's gather, behind the reference's capture()
        bool distinct = false;
        for (uint32_t r = 1; r < n; ++r) distinct = distinct || devs[r] != devs[0];
        if (distinct && !std::getenv("LASGUN_CAPTURE_NO_RCCL")) {
            // RCCL missing or failing (no librccl, ncclCommInitAll refused, an exchange error) must not fail a capture
            // that the RCCL-free path below can serve: say why under LASGUN_DEBUG and go on
            lg_multi *m = lg_multi_create(s, devs.data(), (int)n, block_rows);
            int rc = m ? lg_multi_capture(m, film) : 1;
            if (m) { std::string e = rc ? tl_error : std::string(); lg_multi_free(m); if (rc) tl_error = e; }
            if (rc == 0) return 0;
            static bool told = false;
            if (!told && std::getenv("LASGUN_DEBUG")) {
                told = true;
                std::fprintf(stderr, "[lasgun] lg_capture: the RCCL gather is unavailable (%s); using one host thread and one D2H copy per device\n", tl_error.c_str());
            }
        }
    }
    std::vector<int> rcs(n, 0);
    std::vector<std::string> errs(n);
    std::vector<std::thread> workers;
    for (uint32_t r = 0; r < n; ++r)
        workers.emplace_back([&, r] { // one host thread per device, like the reference's one thread per core
            rcs[r] = capture_share(s, film, devs[r], n, r, block_rows);
            if (rcs[r]) errs[r] = tl_error;
        });
    for (auto &t : workers) t.join();
    for (uint32_t r = 0; r < n; ++r)
        if (rcs[r]) return fail("device " + std::to_string(devs[r]) + ": " + errs[r]);
    return 0;
}
lg_film *lg_render(const lg_scene *s, uint32_t w, uint32_t h) { // lib.rs:46-50
    lg_film *f = lg_film_new(w, h);
    if (lg_capture(s, f)) { lg_film_free(f); return nullptr; }
    return f;
}

int lg_capture_radiance(size_t k, size_t n, const lg_accel *a, uint32_t w, uint32_t h, double *rgb) {
    return guarded([&] {
        if (n == 0) throw Error("n must be > 0");
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        size_t count = (size_t)w * h * 3;
        if (a->staging_rad.n < count) { sync_checked(*a); a->staging_rad.alloc(count); }
        HIP_TRY(hipMemcpyAsync(a->staging_rad.p, rgb, count * 8, hipMemcpyHostToDevice, a->stream));
        DParams P = base_params(*a, w, h);
        if (n == 1 && k == 0) set_rect(P, 0, 0, w, h);
        else set_subset(*a, a->stream, P, k, n, w, h);
        P.out_row0 = 0;
        P.out_radiance = a->staging_rad.p;
        enqueue(*a, P, false, a->stream);
        HIP_TRY(hipMemcpyAsync(rgb, a->staging_rad.p, count * 8, hipMemcpyDeviceToHost, a->stream));
        sync_checked(*a);
    });
}
static int capture_stats_impl(const lg_accel *a, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t filter, lg_stats *out);
int lg_capture_stats(const lg_accel *a, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, lg_stats *out) {
    return capture_stats_impl(a, w, h, y0, y1, 0u, out);
}
int lg_capture_stats_kind(const lg_accel *a, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, int kind, lg_stats *out) {
    if (kind < 0 || kind > 2) return fail("kind: 0 all, 1 closest-hit traversals, 2 shadow traversals");
    return capture_stats_impl(a, w, h, y0, y1, (uint32_t)kind, out);
}
static int capture_stats_impl(const lg_accel *a, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t filter, lg_stats *out) {
    return guarded([&] {
        if (y1 > h || y0 > y1) throw Error("bad row range");
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        DParams P = base_params(*a, w, h);
        set_rect(P, 0, y0, w, y1);
        P.out_row0 = y0;
        P.stats_filter = filter;
        enqueue(*a, P, true, a->stream);
        DStats s;
        HIP_TRY(hipMemcpyAsync(&s, a->stats.p, sizeof s, hipMemcpyDeviceToHost, a->stream));
        sync_checked(*a);
        *out = lg_stats{s.primary_rays, s.shadow_rays, s.secondary_rays, s.nodes_tested, s.spheres_tested, s.cuboids_tested,
                        s.triangles_tested, s.accel_entries, s.hits};
    });
}

int lg_audit_prune(const lg_accel *a, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, lg_prune_audit *out) {
    return guarded([&] {
        if (y1 > h || y0 > y1) throw Error("bad row range");
        if (!out) throw Error("out is NULL");
        std::lock_guard<std::mutex> g(a->mtx);
        if (a->fast) throw Error("the audit is of the pruned REFERENCE walk: not available in fast mode");
        use_device(a->device);
        if (!a->flat.has_records) { // the audit is of the pruned walk as it runs when it is on: with the mesh leaves' records
            const int before = a->prune;
            a->prune = 1;
            try { rebuild_tables(a, false); } catch (...) { a->prune = before; throw; }
            a->prune = before;
        }
        DParams P = base_params(*a, w, h);
        set_rect(P, 0, y0, w, y1);
        P.out_row0 = y0;
        P.prune = 1u; P.audit = std::getenv("LASGUN_AUDIT_SABOTAGE") ? 2u : 1u; // (2: the walk skips by an unsound rule on purpose -- the audit's self-test)
        enqueue(*a, P, true, a->stream);
        DStats s;
        HIP_TRY(hipMemcpyAsync(&s, a->stats.p, sizeof s, hipMemcpyDeviceToHost, a->stream));
        sync_checked(*a);
        auto slack = [](unsigned long long stored) { // (complemented bits, 0 = no sample: k_mega.hip)
            if (stored == 0ull) return (double)INFINITY;
            const unsigned long long bits = ~stored;
            double v; std::memcpy(&v, &bits, sizeof v);
            return v;
        };
        double used = 0.0;
        std::memcpy(&used, &s.audit_used_nodes, sizeof used);
        *out = lg_prune_audit{s.audit_nodes, s.audit_runs, s.audit_prims, s.audit_violations, slack(s.audit_slack_nodes), slack(s.audit_slack_runs), used};
    });
}
int lg_audit_fast(const lg_accel *a, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, lg_fast_audit *out) {
    return guarded([&] {
        if (y1 > h || y0 > y1) throw Error("bad row range");
        if (!out) throw Error("out is NULL");
        std::lock_guard<std::mutex> g(a->mtx);
        if (!a->fast) throw Error("the audit is of the FAST walk: select it with lg_accel_set_mode(accel, 1) first");
        use_device(a->device);
        DParams P = base_params(*a, w, h);
        set_rect(P, 0, y0, w, y1);
        P.out_row0 = y0;
        P.audit = std::getenv("LASGUN_AUDIT_SABOTAGE") ? 2u : 1u; // (2: the counting fast walk prunes by half its limit on purpose -- the audit's self-test)
        enqueue(*a, P, true, a->stream); // the counting instantiation of the megakernel, fast mode: every ray also walked the reference's way
        DStats s;
        HIP_TRY(hipMemcpyAsync(&s, a->stats.p, sizeof s, hipMemcpyDeviceToHost, a->stream));
        sync_checked(*a);
        *out = lg_fast_audit{s.audit_prims, s.audit_runs, s.audit_violations};
    });
}
int lg_accel_set_lds_scene(const lg_accel *a, int enabled) {
    std::lock_guard<std::mutex> lk(a->mtx);
    a->lds_scene = enabled != 0;
    return a->ldss_blocks ? 1 : 0; // 1: the scene's tables fit in LDS (the variant exists for this accel)
}
int lg_accel_set_wf_split(const lg_accel *a, int bands) {
    if (bands < 0 || bands > 8) return fail("bands must be 0 (default) .. 8 (more than 4 are rendered as 4)");
    std::lock_guard<std::mutex> g(a->mtx);
    a->wf_split = (unsigned)bands;
    return 0;
}
int lg_accel_set_streaming(const lg_accel *a, int enabled) {
    std::lock_guard<std::mutex> g(a->mtx);
    if (enabled < 0 || enabled > 3) return fail("streaming must be 0 (megakernel), 1 (default), 2 (wavefront pipeline) or 3 (queue organisation)");
    a->streaming = enabled != 0;
    a->streaming_forced = enabled == 2; // 2 = use it whatever the scene and the launch size (tests)
    a->queue = enabled == 3 ? 1 : enabled == 1 ? -1 : 0; // 3 = the queue organisation whatever the scene; 1 = the accel's defaults
    return 0;
}
int lg_accel_set_tile_order(const lg_accel *a, int order) { // the direction the megakernel and the queue organisation claim a launch's tiles in
    std::lock_guard<std::mutex> g(a->mtx);
    if (order < -1 || order > 2) return fail("tile order must be -1 (default: measured), 0 (top-down), 1 (bottom-up) or 2 (from the middle outwards)");
    a->tile_order = order;
    return 0;
}
int lg_accel_set_tile_parts(const lg_accel *a, int parts) { // the megakernel: a tile handed out whole or in parts of 64 / parts lanes
    std::lock_guard<std::mutex> g(a->mtx);
    if (!(parts == -1 || parts == 1 || parts == 2 || parts == 4 || parts == 8)) return fail("tile parts must be -1 (default), 1, 2, 4 or 8");
    a->tile_parts = parts;
    return 0;
}
int lg_accel_set_sample_order(const lg_accel *a, int order) { // a supersampled pixel's samples: side by side in one launch chain, or one after the other
    std::lock_guard<std::mutex> g(a->mtx);
    if (order < -1 || order > 1) return fail("sample order must be -1 (default), 0 (side by side) or 1 (one after the other)");
    a->sample_order = order;
    return 0;
}
int lg_accel_last_organisation(const lg_accel *a) { // what the accel's last launch ran as: 0 megakernel, 1 level by level, 2 queue, + 16 when its tiles were claimed bottom-up; -1 before the first
    std::lock_guard<std::mutex> g(a->mtx);
    return a->last_org;
}
int lg_accel_get_prune(const lg_accel *a) { // the EFFECTIVE setting of the pruned walk: what a render of this accel uses right now
    std::lock_guard<std::mutex> g(a->mtx);
    return (int)base_params(*a, 8, 8).prune;
}
int lg_accel_set_prune(const lg_accel *a, int enabled) {
    if (enabled < -1 || enabled > 1) return fail("prune must be -1 (default), 0 or 1");
    std::lock_guard<std::mutex> g(a->mtx);
    const int before = a->prune;
    a->prune = enabled;
    if (enabled == 1 && !a->flat.has_records) { // the mesh leaves' culling records were left out of this accel's tables (a small mesh): build them now
        int rc = guarded([&] { rebuild_tables(a, false); });
        if (rc) { a->prune = before; return rc; }
    }
    return 0;
}
int lg_accel_set_mode(const lg_accel *a, int mode) {
    if (mode != 0 && mode != 1) return fail("mode must be 0 (reference traversal) or 1 (fast)");
    std::lock_guard<std::mutex> g(a->mtx);
    if (mode == 1) {
        int rc = guarded([&] { rebuild_tables(a, true); });
        if (rc) return rc;
    }
    if (mode == 1 && !a->fast_available) return fail(a->fast_refusal);
    a->fast = mode == 1;
    return 0;
}
void lg_profile_enable(const lg_accel *a, int enabled) {
    std::lock_guard<std::mutex> g(a->mtx);
    a->profiling = enabled != 0;
}
int lg_profile_read(const lg_accel *a, double *total_ms, uint64_t *launches) {
    return guarded([&] {
        std::lock_guard<std::mutex> g(a->mtx);
        double total = 0.0;
        for (auto &e : a->events) {
            HIP_TRY(hipEventSynchronize(e.second));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
            total += ms;
            (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
        }
        *total_ms = total;
        *launches = a->events.size();
        a->events.clear();
    });
}

int lg_profile_read_kinds(const lg_accel *a, double ms[5], uint64_t launches[5]) {
    return guarded([&] {
        std::lock_guard<std::mutex> g(a->mtx);
        for (int k = 0; k < 5; ++k) {
            double total = 0.0;
            for (auto &e : a->kind_events[k]) {
                HIP_TRY(hipEventSynchronize(e.second));
                float t = 0.f;
                HIP_TRY(hipEventElapsedTime(&t, e.first, e.second));
                if (std::getenv("LASGUN_DEBUG")) std::fprintf(stderr, "[lasgun] kind %d launch: %.3f ms\n", k, t);
                total += t;
                (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
            }
            ms[k] = total; launches[k] = a->kind_events[k].size();
            a->kind_events[k].clear();
        }
    });
}
int lg_accel_dump(const lg_accel *a, const double **f, size_t *nf, const int64_t **i, size_t *ni) {
    *f = a->flat.dump_f.data(); *nf = a->flat.dump_f.size();
    *i = a->flat.dump_i.data(); *ni = a->flat.dump_i.size();
    return 0;
}
// Host-only: run the HLBVH build + flattening of `scene` WITHOUT touching a device and hand back
// the structure dump (same format as lg_accel_dump).  Lets CPU-only tests check the host builder.
int lg_host_build_dump(const lg_scene *s, const double **f, size_t *nf, const int64_t **i, size_t *ni, uint64_t info[8]) {
    static thread_local FlatScene flat;
    return guarded([&] {
        flatten_scene(s->s, flat);
        *f = flat.dump_f.data(); *nf = flat.dump_f.size();
        *i = flat.dump_i.data(); *ni = flat.dump_i.size();
        info[0] = flat.nodes.size(); info[1] = flat.primref.size(); info[2] = flat.spheres.size(); info[3] = flat.cuboids.size();
        info[4] = flat.tri_v.size() / 3; info[5] = flat.accels.size(); info[6] = flat.max_stack; info[7] = flat.has_specular ? 1 : 0;
    });
}
// Host-only self-check of the fast mode's wide records (DNode4) against the binary fast trees they were collapsed from:
// out[0] records, out[1] children, out[2] leaves reached, out[3] the deepest stack a walk that pushes every child but one would need
// (frames of nested accels not counted), out[4] violations (a child box that does not contain its node's f64 box, a leaf reached
// twice or never, a link that is not a node of the tree), out[5] = FlatScene::max_stack_fast1.
// Host-only self-check of the triangle strips (DStrip) against the leaves they are made from: out = { mesh leaves with records, runs,
// triangles, strip entries, violations, hash of the records and the leaves' pad words, hash of the strips, 0 }.  Every triangle slot of a mesh leaf with records must come up in exactly one
// run, exactly once, as a STRIP_TRI entry whose three vertices (the two entries before it and its own) are the slot's three
// vertices in some order; the entry counts must match the run records.
int lg_host_check_strips(const lg_scene *s, uint64_t out[8]) {
    return guarded([&] {
        FlatScene flat;
        flatten_scene(s->s, flat, false);
        for (int k = 0; k < 8; ++k) out[k] = 0;
        std::vector<char> seen_node(flat.nodes.size(), 0);
        for (const DAccel &A : flat.accels) {
            if (!(A.flags & AF_MESH)) continue;
            std::vector<uint32_t> todo{0};
            while (!todo.empty()) {
                const uint32_t nidx = todo.back(); todo.pop_back();
                if (seen_node[A.node_base + nidx]) continue;
                seen_node[A.node_base + nidx] = 1;
                const DNode &nd = flat.nodes[A.node_base + nidx];
                if (!(nd.meta & NODE_LEAF)) { todo.push_back(nidx + 1); todo.push_back(nd.link); continue; }
                const uint32_t nrec = nd.pad >> 24, rec0 = nd.pad & 0x00FFFFFFu;
                if (nrec == 0) continue; // (a leaf without records: walked in the reference's order)
                out[0]++;
                const size_t first = (size_t)A.prim_base + nd.link, count = nd.meta & 0xFFFFu;
                std::vector<int> hits(count, 0);
                for (uint32_t r = rec0; r < rec0 + nrec; ++r) {
                    const DChunk &k = flat.chunks[r];
                    if (k.start == CHUNK_IS_GROUP) continue;
                    out[1]++;
                    const uint32_t ntri = k.count & 0xFFu, nent = k.count >> 8;
                    uint32_t tris = 0;
                    for (uint32_t e = k.pad; e < k.pad + nent; ++e) {
                        out[3]++;
                        const DStrip &E = flat.strips[e];
                        if (!(E.code & STRIP_TRI)) continue;
                        ++tris; out[2]++;
                        const uint32_t slot = E.code & STRIP_SLOT_MASK;
                        if (slot < first || slot >= first + count || e < k.pad + 2) { out[4]++; continue; }
                        hits[slot - first]++;
                        uint32_t want[3][3], got[3][3];
                        std::memcpy(want, flat.leaf_soup[slot].w, 36);
                        std::memcpy(got[0], &flat.strips[e - 2].x, 12); std::memcpy(got[1], &flat.strips[e - 1].x, 12); std::memcpy(got[2], &E.x, 12);
                        bool used[3] = {false, false, false};
                        int matched = 0;
                        for (int i = 0; i < 3; ++i)
                            for (int j = 0; j < 3; ++j)
                                if (!used[j] && std::memcmp(got[i], want[j], 12) == 0) { used[j] = true; ++matched; break; }
                        if (matched != 3) out[4]++;
                    }
                    if (tris != ntri) out[4]++;
                }
                for (size_t i = 0; i < count; ++i) if (hits[i] != 1) out[4]++;
            }
        }
        // FNV-1a over the tables as they would be uploaded (the threaded build must give what one thread gives) and over the leaves' pad words
        auto fnv = [](uint64_t h, const void *p, size_t n) { const unsigned char *b = static_cast<const unsigned char *>(p); for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } return h; };
        uint64_t h = 14695981039346656037ull;
        h = fnv(h, flat.chunks.data(), flat.chunks.size() * sizeof(DChunk));
        for (const DNode &nd : flat.nodes) h = fnv(h, &nd.pad, sizeof nd.pad);
        out[5] = h;
        out[6] = fnv(14695981039346656037ull, flat.strips.data(), flat.strips.size() * sizeof(DStrip));
    });
}
int lg_host_check_wide_records(const lg_scene *s, uint64_t out[8]) {
    return guarded([&] {
        FlatScene flat;
        flatten_scene(s->s, flat, true);
        for (int k = 0; k < 8; ++k) out[k] = 0;
        out[5] = flat.max_stack_fast1;
        std::vector<uint32_t> seen_tree;
        for (const DAccel &A : flat.accels) {
            if (std::find(seen_tree.begin(), seen_tree.end(), A.fnode_base) != seen_tree.end()) continue; // (mesh instances share a tree)
            seen_tree.push_back(A.fnode_base);
            // the binary tree: its nodes (children follow their parent; DNode::link = second child) and its leaves by first slot
            std::map<uint32_t, uint32_t> leaf_by_start; // first slot -> node
            std::set<uint32_t> interior;
            std::vector<uint32_t> bin{0u};
            while (!bin.empty()) {
                const uint32_t i = bin.back(); bin.pop_back();
                const DNode &d = flat.nodes[A.fnode_base + i];
                if (d.meta & NODE_LEAF) leaf_by_start[d.link] = i;
                else { interior.insert(i); bin.push_back(i + 1u); bin.push_back(d.link); }
            }
            if (flat.nodes[A.fnode_base].meta & NODE_LEAF) continue; // a single leaf: no record
            std::map<uint32_t, int> reached;
            struct It { uint32_t node, depth; };
            std::vector<It> st{{0u, 0u}};
            auto contains = [](const float box[6], const DNode &d) {
                for (int a = 0; a < 3; ++a)
                    if (!((double)box[a] <= d.bmin[a]) || !((double)box[3 + a] >= d.bmax[a])) return false;
                return true;
            };
            while (!st.empty()) {
                const It it = st.back(); st.pop_back();
                const DNode4 &w = flat.nodes4[A.fnode_base + it.node];
                out[0]++;
                uint32_t k = 0;
                for (int c = 0; c < WIDE; ++c) k += w.link[c] != NO_HIT ? 1u : 0u;
                if (k < 2u) out[4]++;
                out[3] = std::max<uint64_t>(out[3], it.depth + k - 1u);
                for (int c = 0; c < WIDE; ++c) {
                    if (w.link[c] == NO_HIT) continue;
                    out[1]++;
                    uint32_t node;
                    if (w.link[c] & WIDE_LEAF) {
                        const uint32_t start = w.link[c] & WIDE_START_MASK, cnt = (w.link[c] >> WIDE_COUNT_SHIFT) & 7u;
                        auto f = leaf_by_start.find(start);
                        if (f == leaf_by_start.end() || (flat.nodes[A.fnode_base + f->second].meta & 0xFFFFu) != cnt) { out[4]++; continue; }
                        node = f->second;
                        if (reached[start]++) out[4]++;
                        out[2]++;
                    } else {
                        node = w.link[c];
                        if (!interior.count(node)) { out[4]++; continue; }
                        st.push_back({node, it.depth + k - 1u});
                    }
                    if (!contains(w.box[c], flat.nodes[A.fnode_base + node])) out[4]++;
                }
            }
            if (reached.size() != leaf_by_start.size()) out[4]++;
        }
    });
}
int lg_accel_info(const lg_accel *a, uint64_t out[8]) {
    const FlatScene &f = a->flat;
    out[0] = f.nodes.size(); out[1] = f.primref.size(); out[2] = f.spheres.size(); out[3] = f.cuboids.size();
    out[4] = f.tri_v.size() / 3; out[5] = f.accels.size(); out[6] = a->fast ? a->stack_depth_fast1 : a->stack_depth; out[7] = a->device_bytes;
    return 0;
}

int lg_kat_intersect(int kind, const double *params, const char *obj_text, size_t obj_len, const double o[3], const double d[3], double out[8]) {
    return guarded([&] {
        use_device();
        DevBuf<double> dparams, dout;
        DevBuf<float> dpos;
        DevBuf<uint32_t> dtri;
        std::vector<double> pv(params, params + 8);
        dparams.upload(pv);
        dout.alloc(8);
        uint32_t ntri = 0;
        if (kind == 2) {
            Obj obj;
            parse_obj_text(obj_text, obj_len, obj);
            std::vector<uint32_t> tv;
            for (auto &t : obj.tri) tv.push_back(t.v);
            ntri = (uint32_t)(tv.size() / 3);
            dpos.upload(obj.position);
            dtri.upload(tv);
        } else if (kind != 0 && kind != 1) throw Error("bad kind");
        HIP_TRY(launch_kat(kind, dparams.p, dpos.p, dtri.p, ntri, V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]}, dout.p, nullptr));
        HIP_TRY(hipMemcpy(out, dout.p, 8 * sizeof(double), hipMemcpyDeviceToHost));
    });
}
int lg_kat_surface_interaction(const double o[3], const double d[3], double t, const double dpdu[3], const double dpdv[3], double out_ng[3]) {
    return guarded([&] {
        use_device();
        DevBuf<double> dout;
        dout.alloc(3);
        HIP_TRY(launch_kat_si(V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]}, t, V3{dpdu[0], dpdu[1], dpdu[2]}, V3{dpdv[0], dpdv[1], dpdv[2]}, dout.p, nullptr));
        HIP_TRY(hipMemcpy(out_ng, dout.p, 3 * sizeof(double), hipMemcpyDeviceToHost));
    });
}
int lg_trace_pixel(const lg_accel *a, uint32_t w, uint32_t h, uint32_t x, uint32_t y, int fast, double *out, size_t out_len) {
    return guarded([&] {
        if (x >= w || y >= h) throw Error("pixel outside the film");
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        const size_t need = 7 + 2 * a->flat.lights.size();
        if (out_len < need) throw Error("output too small: 7 + 2 * lights doubles");
        if (fast) rebuild_tables(a, true);
        if (fast && !a->fast_available) throw Error(a->fast_refusal);
        DParams P = base_params(*a, w, h);
        DevBuf<double> dout, dlog;
        dout.alloc(need);
        HIP_TRY(hipMemset(dout.p, 0, need * sizeof(double)));
        const size_t log_n = 1 + 4 * 4000;
        if (out_len >= need + log_n) { dlog.alloc(log_n); HIP_TRY(hipMemset(dlog.p, 0, log_n * sizeof(double))); P.dbg_log = dlog.p; }
        HIP_TRY(launch_trace_pixel(P, fast != 0, fast ? a->stack_depth_fast1 : a->stack_depth, x, y, dout.p, a->stream));
        HIP_TRY(hipMemcpyAsync(out, dout.p, need * sizeof(double), hipMemcpyDeviceToHost, a->stream));
        if (P.dbg_log) HIP_TRY(hipMemcpyAsync(out + need, dlog.p, log_n * sizeof(double), hipMemcpyDeviceToHost, a->stream));
        sync_checked(*a);
    });
}
// Measured rates of the current device, GB/s: what 0 = HBM copy (16 B per lane, 1 GiB each way, read + written bytes),
// 1 = aggregate LDS read rate (ds_read_b128, every CU streaming).  The roofline's measured denominators (bench.py).
int lg_probe_rate(int what, double *gbps) {
    return guarded([&] {
        use_device();
        if (what != 0 && what != 1) throw Error("what: 0 = HBM copy, 1 = LDS read");
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        double best = 0.0;
        if (what == 0) {
            const size_t bytes = 1ull << 30;
            DevBuf<uint8_t> a, b;
            a.alloc(bytes); b.alloc(bytes);
            HIP_TRY(hipMemset(a.p, 1, bytes)); HIP_TRY(hipMemset(b.p, 2, bytes));
            for (int rep = 0; rep < 5; ++rep) {
                HIP_TRY(hipEventRecord(e0, nullptr));
                HIP_TRY(launch_probe_copy(a.p, b.p, bytes, nullptr));
                HIP_TRY(hipEventRecord(e1, nullptr));
                HIP_TRY(hipEventSynchronize(e1));
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms > 0.f) best = std::max(best, 2.0 * (double)bytes / (ms * 1e-3) / 1e9);
            }
        } else {
            int cus = 0, dev = 0;
            HIP_TRY(hipGetDevice(&dev));
            HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            DevBuf<uint32_t> sink;
            sink.alloc((size_t)cus);
            const uint32_t iters = 4096;
            for (int rep = 0; rep < 4; ++rep) {
                HIP_TRY(hipEventRecord(e0, nullptr));
                HIP_TRY(launch_probe_lds((uint32_t)cus, iters, sink.p, nullptr));
                HIP_TRY(hipEventRecord(e1, nullptr));
                HIP_TRY(hipEventSynchronize(e1));
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
                const double bytes = (double)cus * 1024.0 * 16.0 * 16.0 * iters;
                if (rep > 0 && ms > 0.f) best = std::max(best, bytes / (ms * 1e-3) / 1e9);
            }
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        *gbps = best;
    });
}
int lg_math_eval(int op, size_t n, const double *a, const double *b, double *out) {
    return guarded([&] {
        use_device();
        if (op < 0 || op > 8) throw Error("bad op");
        DevBuf<double> da, db, dout;
        std::vector<double> va(a, a + n), vb(b, b + n);
        da.upload(va); db.upload(vb); dout.alloc(n);
        HIP_TRY(launch_math(op, n, da.p, db.p, dout.p, nullptr));
        HIP_TRY(hipMemcpy(out, dout.p, n * sizeof(double), hipMemcpyDeviceToHost));
    });
}

} // extern "C"

// Diagnostic (tools/queue_levels.py): the packets each recursion level of the queue organisation's LAST launch on `hip_stream` held
// (QC_COUNT of its control block) -- with the rays per level this says how full the 64-ray packets of the deeper levels are.
extern "C" int lg_debug_queue_packets(const lg_accel *a, void *hip_stream, unsigned long long out[8]) {
    return guarded([&] {
        std::lock_guard<std::mutex> g(a->mtx);
        use_device(a->device);
        HIP_TRY(hipDeviceSynchronize());
        for (int d = 0; d < 8; ++d) out[d] = 0;
        for (auto &c : a->ctxs) {
            if (c->key != (hipStream_t)hip_stream || c->wf_counters.n < QC_WORDS) continue;
            std::vector<uint32_t> w(QC_WORDS);
            HIP_TRY(hipMemcpy(w.data(), c->wf_counters.p, QC_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost));
            for (uint32_t d = 0; d < QC_MAX_LEVELS; ++d) out[d] = w[QC_LEVEL0 + QC_LEVEL_WORDS * d + QC_COUNT];
#ifdef LG_QIDLE // diagnostic build: [7] = the waves' idle time (100 MHz ticks, summed), k_queue.hip
            out[7] = (unsigned long long)w[QC_ERROR] | ((unsigned long long)w[QC_ERROR + 1] << 32);
#endif
        }
    });
}

#if defined(LG_PKT_STATS) || defined(LG_STAMPS) || defined(LG_QIDLE)
extern "C" int lg_debug_stats(const lg_accel *a, int clear, unsigned long long *out9) { // analysis builds only
    return guarded([&] {
        use_device(a->device);
        HIP_TRY(hipDeviceSynchronize());
        if (clear) HIP_TRY(hipMemset(a->stats.p, 0, 2 * sizeof(DStats)));
        else HIP_TRY(hipMemcpy(out9, a->stats.p, 2 * sizeof(DStats), hipMemcpyDeviceToHost));
    });
}
#endif
