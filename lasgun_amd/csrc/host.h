// lasgun_amd/csrc/host.h -- host side of the product: scene description, OBJ reader,
// reference-faithful HLBVH builder and the flattening into device tables.
//
// Mirrors the reference's host-side surface for the render path:
//   Scene      /root/reference/src/scene.rs:11-143
//   Aggregate  /root/reference/src/scene/node.rs:7-115
//   Camera     /root/reference/src/camera.rs:6-194
//   Transform  /root/reference/src/space/transform.rs:49-197
//   BVH build  /root/reference/src/accelerators/bvh.rs:135-453,525-635
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "dscene.h"

namespace lg {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct Material { // POD handed across the C ABI (lg_material)
    int32_t kind;
    double p[10];
};
Material material_default();
Material material_matte(const double kd[3], double sigma);

// Full 4x4, column-major m[col][row]; concat is done on the host exactly like cgmath.
struct Mat4 {
    double m[4][4];
};
struct Transform {
    Mat4 m, minv;
};
Transform transform_identity();
void transform_concat_self(Transform &self, const Transform &other);
Transform transform_translate(const double d[3]);
Transform transform_scale(double x, double y, double z);
Transform transform_rotate_x(double deg);
Transform transform_rotate_y(double deg);
Transform transform_rotate_z(double deg);
Transform transform_rotate(double deg, const double axis[3]);

struct Bounds {
    V3 min, max;
};

struct Obj {
    std::vector<float> position, texture, normal;
    struct Tuple {
        uint32_t v;
        int32_t t, n;
    };
    std::vector<Tuple> tri; // 3 per `f` line, file order
};
void parse_obj_text(const char *text, size_t len, Obj &out); // throws Error

struct Aggregate;
struct SceneNode {
    enum Kind { SPHERE, CUBE, CUBOID, MESH, GROUP } kind = SPHERE;
    double a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
    Material mat{};
    bool has_mat = false;
    uint32_t obj = 0;
    std::unique_ptr<Aggregate> group;
};
struct Aggregate {
    std::vector<SceneNode> contents;
    Transform transform = transform_identity();
    bool swap_backface = false;
};

struct Camera {
    V3 origin{0, 0, 0}, view{0, 0, 1}, up{0, 1, 0}, aux{1, 0, 0};
    bool perspective = true;
    double param = 45.0;
    uint32_t ss_root = 1;
    double ss_distance = 1.0;
    double aperture_radius = 0.0;
    double image_plane_height = 0.0, pixel_separation = 0.0;
    void init(bool persp, double p);
    void look_at(V3 o, V3 look, V3 upv);
    void set_supersampling(uint8_t base);
};

struct Light {
    double pos[3], intensity[3], falloff[3];
};

struct Scene {
    std::unique_ptr<Aggregate> root{new Aggregate()};
    Camera camera;
    V3 bg_inner{0, 0, 0}, bg_outer{0, 0, 0};
    double bg_scale = 1.0;
    V3 ambient{0, 0, 0};
    bool smoothing = true;
    uint32_t recursion = 3;
    size_t threads = 0;
    std::vector<Light> lights;
    std::vector<std::unique_ptr<Obj>> meshes;
    Scene() { camera.init(true, 45.0); }
};

// ---- flattened scene (host copies of the device tables) ----------------------------------
struct FlatScene {
    std::vector<DNode> nodes;
    std::vector<DNode4> nodes4; // wide records of the fast trees' interior nodes (empty without fast trees)
    std::vector<uint32_t> primref;
    std::vector<DSphere> spheres;
    std::vector<int32_t> sphere_mat;
    std::vector<DCuboid> cuboids;
    std::vector<int32_t> cuboid_mat;
    std::vector<uint32_t> tri_v, tri_n, tri_t;
    std::vector<float> vpos, vnorm, vtex;
    std::vector<DLeafRec> leaf_soup; // one per primref slot (+2 spare)
    std::vector<DLeafRec> leaf_soup2; // leaf_soup with every mesh leaf's triangles in k-d order (word 9: original slot)
    std::vector<DChunk> chunks;      // culling records (runs and groups of runs) of leaf_soup2: of the pruned walk's fat mesh leaves
    std::vector<DStrip> strips;      // the runs' triangles as triangle strips: what the pruned walk's leaf loop streams (dscene.h, DStrip)
    // fast mode's candidate check: a hit found through the fast tree counts only if the REFERENCE tree would have tested
    // that primitive for this ray, i.e. if every box on its root-to-leaf path in the reference tree passes the slab test
    std::vector<uint32_t> sphere_ref_leaf, cuboid_ref_leaf, tri_ref_leaf; // per primitive: the leaf of ITS accel's reference tree holding it
    std::vector<uint32_t> accel_ref_leaf; // per accel: the leaf of its parent's reference tree that holds it (NO_HIT for the root)
    std::vector<DAccel> accels;
    std::vector<DMaterial> materials;
    std::vector<DLight> lights;
    int32_t default_material = 0;
    uint32_t max_stack = 0;      // worst-case per-lane traversal stack entries
    uint32_t max_stack_fast1 = 0; // the fast trees under the wide walk (one word per pending child, 3-word level frames)
    bool has_specular = false;   // any glass / mirror material present
    bool has_fast = false;       // the fast mode's trees are part of the tables
    bool has_records = false;    // the mesh leaves' culling records and strips are part of the tables (or there is no mesh)
    bool boxes_finite = false;   // every node box is finite with bmin <= bmax on every axis (DParams::boxes_finite: the sign-specialised slab test is exact then)
    // structure dump in the same format as the oracle's orc_accel_dump (build-parity tests)
    std::vector<double> dump_f;
    std::vector<int64_t> dump_i;
};
// `with_fast`: also build the fast mode's trees (binned SAH; 5-10x the reference build's time): only when that mode is asked for
// (with_records: the culling records and strips of the pruned walk's mesh leaves -- a third to a half of a mesh accel's build; without them
// every mesh leaf is walked in the reference's order, pruned walk or not)
void flatten_scene(const Scene &scene, FlatScene &out, bool with_fast = false, bool with_records = true); // throws Error

} // namespace lg
