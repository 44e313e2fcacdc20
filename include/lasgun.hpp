// include/lasgun.hpp -- header-only C++ host API over the C ABI (include/lasgun_hip.h).
//
// The reference's own toolchain (Rust) is absent from this pipeline, so the host side above
// the C ABI is C++: the same names, argument meaning and ownership rules as the reference's
// public surface, so a scene written against nfrasser/lasgun ports line by line:
//
//   lasgun::Scene      src/scene.rs:49-143        lasgun::Aggregate  src/scene/node.rs:35-115
//   lasgun::Material   src/material/mod.rs:15-46  lasgun::Camera     src/camera.rs:75-102
//   lasgun::Film       src/film.rs:22-45          lasgun::Accel      src/lib.rs:42
//   lasgun::capture / capture_subset / render     src/lib.rs:46-56,110
//
// Rust panics and `Result`s become exceptions (lasgun::Error, lasgun::ObjError).
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <algorithm>
#include <vector>
#include <utility>

#include "lasgun_hip.h"

namespace lasgun {

using Vec3 = std::array<double, 3>;

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct ObjError : Error { // obj::ObjError of scene.rs:120-130
    using Error::Error;
};

class Material { // Copy POD (material/mod.rs:3-4)
  public:
    lg_material m;
    static Material default_() { return {lg_material_default()}; }
    static Material matte(Vec3 kd, double sigma) { return {lg_material_matte(kd.data(), sigma)}; }
    static Material plastic(Vec3 kd, Vec3 ks, double roughness) { return {lg_material_plastic(kd.data(), ks.data(), roughness)}; }
    static Material metal(Vec3 eta, Vec3 k, double u, double v) { return {lg_material_metal(eta.data(), k.data(), u, v)}; }
    static Material glass(Vec3 kr, Vec3 kt, double eta) { return {lg_material_glass(kr.data(), kt.data(), eta)}; }
    static Material mirror(Vec3 kr) { return {lg_material_mirror(kr.data())}; }
};

using ObjRef = uint32_t; // scene.rs:44

class Aggregate {
  public:
    Aggregate() : h_(lg_aggregate_new()), owned_(true) {}
    Aggregate(const Aggregate &) = delete;
    Aggregate &operator=(const Aggregate &) = delete;
    Aggregate(Aggregate &&o) noexcept : h_(o.h_), owned_(o.owned_) { o.h_ = nullptr; }
    ~Aggregate() { if (h_ && owned_) lg_aggregate_free(h_); }

    void add_group(Aggregate &&child) { lg_aggregate_add_group(h_, child.release()); } // moves, as in Rust
    void add_sphere(Vec3 center, double radius, Material m) { lg_aggregate_add_sphere(h_, center.data(), radius, &m.m); }
    void add_cube(Vec3 origin, double dim, Material m) { lg_aggregate_add_cube(h_, origin.data(), dim, &m.m); }
    void add_box(Vec3 mn, Vec3 mx, Material m) { lg_aggregate_add_box(h_, mn.data(), mx.data(), &m.m); }
    void add_obj(ObjRef mesh) { lg_aggregate_add_obj(h_, mesh); }
    void add_obj_of(ObjRef mesh, Material m) { lg_aggregate_add_obj_of(h_, mesh, &m.m); }
    void swap_backface() { lg_aggregate_swap_backface(h_); }
    Aggregate &translate(Vec3 d) { lg_aggregate_translate(h_, d.data()); return *this; }
    Aggregate &scale(double x, double y, double z) { lg_aggregate_scale(h_, x, y, z); return *this; }
    Aggregate &rotate_x(double deg) { lg_aggregate_rotate_x(h_, deg); return *this; }
    Aggregate &rotate_y(double deg) { lg_aggregate_rotate_y(h_, deg); return *this; }
    Aggregate &rotate_z(double deg) { lg_aggregate_rotate_z(h_, deg); return *this; }
    Aggregate &rotate(double deg, Vec3 axis) { lg_aggregate_rotate(h_, deg, axis.data()); return *this; }

  private:
    friend class Scene;
    Aggregate(lg_aggregate *borrowed, bool owned) : h_(borrowed), owned_(owned) {}
    lg_aggregate *release() {
        if (!owned_) throw Error("cannot move a borrowed Aggregate");
        lg_aggregate *h = h_;
        h_ = nullptr;
        return h;
    }
    lg_aggregate *h_;
    bool owned_;
};

class Scene;
class Camera { // a view of scene.camera (the Rust setters hand out &mut Camera)
  public:
    void look_at(Vec3 origin, Vec3 look, Vec3 up) { lg_camera_look_at(s_, origin.data(), look.data(), up.data()); }
    void set_supersampling(uint8_t base) { lg_camera_set_supersampling(s_, base); }
    void set_aperture_radius(double r) { lg_camera_set_aperture_radius(s_, r); }

  private:
    friend class Scene;
    explicit Camera(lg_scene *s) : s_(s) {}
    lg_scene *s_;
};

class Scene {
  public:
    Scene() : h_(lg_scene_new()) {}
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;
    Scene(Scene &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    ~Scene() { if (h_) lg_scene_free(h_); }

    Aggregate root() { return Aggregate(lg_scene_root(h_), false); } // `scene.root` (pub field)
    Camera set_perspective_camera(double fov) { lg_scene_set_perspective_camera(h_, fov); return Camera(h_); }
    Camera set_orthographic_camera(double scale) { lg_scene_set_orthographic_camera(h_, scale); return Camera(h_); }
    void set_solid_background(Vec3 c) { lg_scene_set_solid_background(h_, c.data()); }
    void set_radial_background(Vec3 inner, Vec3 outer, double scale) { lg_scene_set_radial_background(h_, inner.data(), outer.data(), scale); }
    void set_ambient_light(Vec3 c) { lg_scene_set_ambient_light(h_, c.data()); }
    void set_mesh_smoothing(bool enabled) { lg_scene_set_mesh_smoothing(h_, enabled ? 1 : 0); }
    void set_max_recursion_depth(uint32_t d) { lg_scene_set_max_recursion_depth(h_, d); }
    void set_threads(size_t t) { lg_scene_set_threads(h_, t); }
    void add_point_light(Vec3 position, Vec3 intensity, Vec3 falloff) { lg_scene_add_point_light(h_, position.data(), intensity.data(), falloff.data()); }
    ObjRef parse_obj(const std::string &text) {
        ObjRef r = 0;
        if (lg_scene_parse_obj(h_, text.data(), text.size(), &r)) throw ObjError(lg_last_error());
        return r;
    }
    ObjRef load_obj(const std::string &path) {
        ObjRef r = 0;
        if (lg_scene_load_obj(h_, path.c_str(), &r)) throw ObjError(lg_last_error());
        return r;
    }
    void set_root(Aggregate &&node) { lg_scene_set_root(h_, node.release()); }
    const lg_scene *handle() const { return h_; }

  private:
    lg_scene *h_;
};

class Film {
  public:
    Film(uint32_t w, uint32_t h) : h_(lg_film_new(w, h)) {}
    Film(uint32_t w, uint32_t h, uint8_t *rgba) : h_(lg_film_wrap(w, h, rgba)) {} // Film::new_with_output
    explicit Film(lg_film *adopt) : h_(adopt) {}
    Film(const Film &) = delete;
    Film &operator=(const Film &) = delete;
    Film(Film &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    ~Film() { if (h_) lg_film_free(h_); }
    uint32_t w() const { return lg_film_width(h_); }
    uint32_t h() const { return lg_film_height(h_); }
    const uint8_t *pixels() const { return lg_film_pixels(h_); }
    lg_film *handle() { return h_; }

  private:
    lg_film *h_;
};

class Accel { // `Accel::from(&scene)`: borrows the scene, which must outlive it
  public:
    static Accel from(const Scene &scene) {
        lg_accel *a = lg_accel_from(scene.handle());
        if (!a) throw Error(lg_last_error());
        return Accel(a);
    }
    Accel(const Accel &) = delete;
    Accel &operator=(const Accel &) = delete;
    Accel(Accel &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    ~Accel() { if (h_) lg_accel_free(h_); }
    const lg_accel *handle() const { return h_; }

  private:
    explicit Accel(lg_accel *a) : h_(a) {}
    lg_accel *h_;
};

// GPU-side counterpart of `scene.threads`: the devices capture() / render() split a host film over
// (empty = every visible device); see lg_set_devices.
inline void set_devices(const std::vector<int> &ids = {}) {
    if (lg_set_devices(ids.data(), (int)ids.size())) throw Error(lg_last_error());
}
inline void capture(const Scene &scene, Film &film) { // lib.rs:55
    if (lg_capture(scene.handle(), film.handle())) throw Error(lg_last_error());
}
inline void capture_subset(size_t k, size_t n, const Accel &root, Film &film) { // lib.rs:110
    if (lg_capture_subset(k, n, root.handle(), film.handle())) throw Error(lg_last_error());
}
// several subsets of one n as ONE render: the pixels of the calls capture_subset(k, n, ...) for k in ks (lasgun_hip.h, lg_capture_subsets)
inline void capture_subsets(const std::vector<size_t> &ks, size_t n, const Accel &root, Film &film) {
    if (lg_capture_subsets(ks.data(), ks.size(), n, root.handle(), film.handle())) throw Error(lg_last_error());
}
// the table of measured kernel-organisation choices (lasgun_hip.h, lg_tune_*): export it once, import it at start-up, and no launch of a known kind is measured again
inline std::vector<lg_tune_entry> tune_export() {
    std::vector<lg_tune_entry> v(lg_tune_export(nullptr, 0));
    v.resize(std::min(v.size(), lg_tune_export(v.data(), v.size())));
    return v;
}
inline void tune_import(const std::vector<lg_tune_entry> &entries) {
    if (lg_tune_import(entries.data(), entries.size())) throw Error(lg_last_error());
}
inline void tune_clear() { lg_tune_clear(); }
inline Film render(const Scene &scene, std::pair<uint32_t, uint32_t> resolution) { // lib.rs:46
    lg_film *f = lg_render(scene.handle(), resolution.first, resolution.second);
    if (!f) throw Error(lg_last_error());
    return Film(f);
}

} // namespace lasgun
