// Progressive rendering with `capture_subset`: the film is refined in PASSES interleaved pixel subsets {k, k + n, ...}
// (nfrasser/lasgun src/lib.rs:110-162 is the entry point the reference's wasm front end drives the same way).  Each
// subset is one GPU launch over a pixel list; a preview is written after every quarter of the passes.
use ::lasgun::{ scene::Scene, Accel, Material, capture_subset, output };

mod common;

const PASSES: usize = 16;

fn main() {
    let mut scene = Scene::new();
    let white = common::room(&mut scene, 0);
    scene.root.add_sphere([1.0, -1.25, 0.0], 1.0, white);
    scene.root.add_cube([-1.999, -1.999, 0.0], 1.0, Material::glass([1.0, 0.7, 1.0], [0.7, 1.0, 0.7], 1.25));

    let accel = Accel::from(&scene); // built once, shared by every pass
    let mut film = output::film([1024, 1024]);
    // a fixed permutation of the passes (5 is coprime to 16): early previews cover the image evenly
    for step in 0..PASSES {
        let pass = (step * 5 + 3) % PASSES;
        capture_subset(pass, PASSES, &accel, &mut film);
        if (step + 1) % (PASSES / 4) == 0 { film.save(&format!("progressive_{:02}.png", step + 1)) }
    }
}
