// Progressive rendering: the film is refined in PASSES interleaved pixel subsets {k, k + n, ...} (nfrasser/lasgun
// src/lib.rs:110-162 is the entry point the reference's wasm front end drives the same way, one `capture_subset` per
// pass).  Here the passes between two previews go out as ONE batch (`capture_subsets`: one render, one copy home), which
// writes exactly the pixels the single calls would; a preview is written after every quarter of the passes.
use ::lasgun::{ scene::Scene, Accel, Material, capture_subsets, output };

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
    let order: Vec<usize> = (0..PASSES).map(|step| (step * 5 + 3) % PASSES).collect();
    for (quarter, batch) in order.chunks(PASSES / 4).enumerate() {
        capture_subsets(batch, PASSES, &accel, &mut film);
        film.save(&format!("progressive_{:02}.png", (quarter + 1) * (PASSES / 4)))
    }
}
