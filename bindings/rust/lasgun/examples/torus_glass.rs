// BASELINE.json configs[3]: a generated 100,352-triangle torus of glass in a transformed group plus a mirror sphere in
// the room (secondary rays, recursion 3), 4096 x 4096.  `--fast` opts into the pruned-tree traversal mode of the core
// (`Accel::set_fast_mode`), which needs the lower-level `capture_subset` entry point.
use ::lasgun::{ scene::{Aggregate, Scene}, Accel, Material, capture_subset, output };

mod common;

fn main() {
    let scene = torus_room();
    if std::env::args().any(|a| a == "--fast") {
        let mut film = output::film([4096, 4096]);
        let accel = Accel::from(&scene);
        accel.set_fast_mode(true);
        capture_subset(0, 1, &accel, &mut film);
        film.save("torus_glass.png");
    } else {
        output::render(&scene, [4096, 4096], "torus_glass.png");
    }
}

fn torus_room() -> Scene {
    let mut scene = Scene::new();
    scene.set_mesh_smoothing(true);
    common::room(&mut scene, 0);
    let torus = scene.parse_obj(&common::torus_obj(224, 224, 0.9, 0.35)).expect("the generated torus parses");
    let mut group = Aggregate::new();
    group.scale(1.2, 1.2, 1.2).rotate_x(35.0).rotate_y(30.0);
    group.add_obj_of(torus, Material::glass([1.0, 0.7, 1.0], [0.7, 1.0, 0.7], 1.25));
    scene.root.add_group(group);
    scene.root.add_sphere([1.1, -1.4, 0.6], 0.6, Material::mirror([0.5, 0.5, 0.5]));
    scene
}
