// The headline scene of the MI355X build (BASELINE.json configs[2]): the room plus 1024 random plastic spheres,
// 4096 x 4096, one sample per pixel -- rendered through `output::render`, i.e. `capture()` on the GPU.
use ::lasgun::{ scene::Scene, Material, output };

mod common;

const PALETTE: [[f64; 3]; 8] = [
    [0.9, 0.9, 0.9], [1.0, 0.2, 0.2], [0.2, 1.0, 0.2], [0.2, 0.3, 1.0],
    [1.0, 0.8, 0.1], [0.8, 0.2, 0.9], [0.1, 0.8, 0.8], [0.95, 0.5, 0.1],
];

fn main() { output::render(&spheres(1024, 0x1A56_0001), [4096, 4096], "spheres1024.png"); }

fn spheres(count: usize, seed: u64) -> Scene {
    let mut scene = Scene::new();
    common::room(&mut scene, 0);
    let mut rng = common::SplitMix64(seed);
    // draw order per sphere: cx, cy, cz, radius, palette index
    for _ in 0..count {
        let centre = [rng.uniform(-1.8, 1.8), rng.uniform(-1.8, 1.8), rng.uniform(-1.8, 1.8)];
        let radius = rng.uniform(0.02, 0.06);
        let pick = ((rng.next_f64() * PALETTE.len() as f64) as usize).min(PALETTE.len() - 1);
        scene.root.add_sphere(centre, radius, Material::plastic(PALETTE[pick], [0.5, 0.7, 0.5], 0.25));
    }
    scene
}
