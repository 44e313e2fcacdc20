// Shared pieces of this crate's example programs: the deterministic generator and the room every benchmark scene of
// the MI355X build sits in (the same scenes `lasgun_amd/scenes.py` builds through the Python binding).
#![allow(dead_code)]
use ::lasgun::{ scene::{Aggregate, Scene}, Material };

/// SplitMix64: scene generation only, never used at render time.
pub struct SplitMix64(pub u64);

impl SplitMix64 {
    pub fn next_u64(&mut self) -> u64 {
        self.0 = self.0.wrapping_add(0x9E37_79B9_7F4A_7C15);
        let mut z = self.0;
        z = (z ^ (z >> 30)).wrapping_mul(0xBF58_476D_1CE4_E5B9);
        z = (z ^ (z >> 27)).wrapping_mul(0x94D0_49BB_1331_11EB);
        z ^ (z >> 31)
    }
    /// 53 random bits as a double in [0, 1)
    pub fn next_f64(&mut self) -> f64 { (self.next_u64() >> 11) as f64 * (1.0 / 9007199254740992.0) }
    pub fn uniform(&mut self, lo: f64, hi: f64) -> f64 { lo + (hi - lo) * self.next_f64() }
}

/// A unit plane in the xz plane as OBJ text (two triangles).
pub const PLANE: &str = "o plane\nv -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\n\nf 1 2 3\nf 1 3 4\n";

fn wall(scene: &mut Scene, plane: ::lasgun::scene::ObjRef, turn: Option<(char, f64)>, shift: [f64; 3], material: Material) {
    let mut side = Aggregate::new();
    side.scale(2.0, 1.0, 2.0);
    match turn {
        Some(('x', deg)) => { side.rotate_x(deg); }
        Some(('z', deg)) => { side.rotate_z(deg); }
        _ => {}
    }
    side.translate(shift);
    side.add_obj_of(plane, material);
    scene.root.add_group(side);
}

/// Camera at z = 5 looking at the origin, one point light under the ceiling, and a 4 x 4 x 4 room open towards the
/// camera: white floor, ceiling and back wall, a red and a green side wall.  Returns the white wall material.
pub fn room(scene: &mut Scene, supersampling: u8) -> Material {
    scene.set_ambient_light([0.2, 0.2, 0.2]);
    let camera = scene.set_perspective_camera(60.0);
    camera.look_at([0.0, 0.0, 5.0], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0]);
    camera.set_supersampling(supersampling);
    let white = Material::plastic([0.9, 0.9, 0.9], [0.5, 0.7, 0.5], 0.25);
    let red = Material::plastic([1.0, 0.0, 0.0], [0.5, 0.7, 0.5], 0.25);
    let green = Material::plastic([0.0, 1.0, 0.0], [0.5, 0.7, 0.5], 0.25);
    let plane = scene.parse_obj(PLANE).expect("the inline plane parses");
    scene.add_point_light([0.0, 1.75, 0.0], [0.9, 0.9, 0.9], [1.0, 0.0, 0.0]);
    wall(scene, plane, None, [0.0, -2.0, 0.0], white);
    wall(scene, plane, None, [0.0, 2.0, 0.0], white);
    wall(scene, plane, Some(('z', 90.0)), [-2.0, 0.0, 0.0], red);
    wall(scene, plane, Some(('z', 90.0)), [2.0, 0.0, 0.0], green);
    wall(scene, plane, Some(('x', 90.0)), [0.0, 0.0, -2.0], white);
    white
}

/// OBJ text of a torus: nu * nv quads as 2 * nu * nv triangles with per-vertex normals (224 x 224 -> 100,352 triangles).
pub fn torus_obj(nu: usize, nv: usize, big: f64, small: f64) -> String {
    use std::f64::consts::PI;
    use std::fmt::Write;
    let mut text = String::from("o torus\n");
    let mut normals = String::new();
    for i in 0..nu {
        let (su, cu) = (2.0 * PI * i as f64 / nu as f64).sin_cos();
        for j in 0..nv {
            let (sv, cv) = (2.0 * PI * j as f64 / nv as f64).sin_cos();
            writeln!(text, "v {:.6} {:.6} {:.6}", (big + small * cv) * cu, small * sv, (big + small * cv) * su).unwrap();
            writeln!(normals, "vn {:.6} {:.6} {:.6}", cv * cu, sv, cv * su).unwrap();
        }
    }
    text.push_str(&normals);
    let vid = |i: usize, j: usize| (i % nu) * nv + (j % nv) + 1;
    for i in 0..nu {
        for j in 0..nv {
            let (a, b, c, d) = (vid(i, j), vid(i + 1, j), vid(i + 1, j + 1), vid(i, j + 1));
            writeln!(text, "f {0}//{0} {1}//{1} {2}//{2}", a, d, c).unwrap();
            writeln!(text, "f {0}//{0} {1}//{1} {2}//{2}", a, c, b).unwrap();
        }
    }
    text
}
