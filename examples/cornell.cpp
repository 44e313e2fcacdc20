// examples/cornell.cpp -- the reference's src/examples/cornell.rs:7-69 written against
// include/lasgun.hpp, line for line (the plane mesh is the inline fixture of
// src/shape/triangle.rs:412-421 because the reference's plane.obj is a Git-LFS stub).
// Writes the raw RGBA8 film to argv[1] (default cornell.rgba); PNG encoding is out of scope.
#include <cstdio>
#include <cstdlib>

#include "lasgun.hpp"

using namespace lasgun;

static Scene cornell() {
    Scene scene;
    scene.set_ambient_light({0.2, 0.2, 0.2});

    Camera camera = scene.set_perspective_camera(60.);
    camera.look_at({0., 0., 5.}, {0., 0., 0.}, {0., 1., 0.});
    camera.set_supersampling(2);

    Material white = Material::plastic({0.9, 0.9, 0.9}, {0.5, 0.7, 0.5}, 0.25);
    Material r = Material::plastic({1.0, 0.0, 0.0}, {0.5, 0.7, 0.5}, 0.25);
    Material g = Material::plastic({0.0, 1.0, 0.0}, {0.5, 0.7, 0.5}, 0.25);
    Material glass = Material::glass({1.0, 0.7, 1.0}, {0.7, 1.0, 0.7}, 1.25);

    ObjRef plane = scene.parse_obj("o plane\nv -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\n\nf 1 2 3\nf 1 3 4\n");

    scene.add_point_light({0.0, 1.75, 0.0}, {0.9, 0.9, 0.9}, {1.0, 0.0, 0.0});

    Aggregate floor;
    floor.scale(2.0, 1.0, 2.0);
    floor.translate({0.0, -2.0, 0.0});
    floor.add_obj_of(plane, white);
    scene.root().add_group(std::move(floor));

    Aggregate ceiling;
    ceiling.scale(2.0, 1.0, 2.0);
    ceiling.translate({0.0, 2.0, 0.0});
    ceiling.add_obj_of(plane, white);
    scene.root().add_group(std::move(ceiling));

    Aggregate left;
    left.scale(2.0, 1.0, 2.0);
    left.rotate_z(90.0);
    left.translate({-2.0, 0.0, 0.0});
    left.add_obj_of(plane, r);
    scene.root().add_group(std::move(left));

    Aggregate right;
    right.scale(2.0, 1.0, 2.0);
    right.rotate_z(90.0);
    right.translate({2.0, 0.0, 0.0});
    right.add_obj_of(plane, g);
    scene.root().add_group(std::move(right));

    Aggregate back;
    back.scale(2.0, 1.0, 2.0);
    back.rotate_x(90.0);
    back.translate({0.0, 0.0, -2.0});
    back.add_obj_of(plane, white);
    scene.root().add_group(std::move(back));

    scene.root().add_sphere({1.0, -1.25, 0.0}, 1.0, glass);
    scene.root().add_cube({-1.999, -1.999, 0.0}, 1.0, glass);
    return scene;
}

int main(int argc, char **argv) {
    const char *out = argc > 1 ? argv[1] : "cornell.rgba";
    uint32_t size = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 512;
    try {
        Scene scene = cornell();
        Film film = render(scene, {size, size}); // output::render minus the PNG encode
        FILE *f = std::fopen(out, "wb");
        if (!f) { std::perror(out); return 2; }
        std::fwrite(film.pixels(), 1, (size_t)film.w() * film.h() * 4, f);
        std::fclose(f);
        std::printf("wrote %s (%ux%u RGBA8)\n", out, film.w(), film.h());
    } catch (const lasgun::Error &e) {
        std::fprintf(stderr, "lasgun error: %s\n", e.what());
        return 1;
    }
    return 0;
}
