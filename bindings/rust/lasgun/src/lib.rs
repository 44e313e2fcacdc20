//! `lasgun` over liblasgun_hip.so -- the reference's public surface for the render path, same names, same argument
//! meaning, rendering on an MI355X through the C ABI of include/lasgun_hip.h (crate `lasgun-hip-sys`).
//!
//! What is mirrored (file:line under nfrasser/lasgun):
//!   `Accel`, `render`, `capture`, `capture_subset`          src/lib.rs:42-56,110
//!   `scene::Scene`, `scene::ObjRef`                          src/scene.rs:11-143
//!   `scene::Aggregate`                                       src/scene/node.rs:25-115
//!   `Material`                                               src/material/mod.rs:3-46
//!   `Camera`                                                 src/camera.rs:75-102
//!   `Film`, `Img`, `Pixel`, `PixelBuffer`                    src/film.rs:7-45, src/img.rs:9-52
//!   `output::render`                                         src/output.rs:5-18 (PNG written by a small built-in encoder)
//!
//! Differences a port has to know about:
//!   * there is no CPU path: every render call needs a HIP device and panics with the library's message otherwise
//!     (the reference panics on a failed thread join, a missing mesh, a BVH deeper than 64, an empty aggregate);
//!   * `scene.camera` is reached through the `&mut Camera` the `set_*_camera` calls return, as in every example;
//!   * `scene.threads` caps the number of GPUs a `capture` is split over (`set_devices`), not CPU threads.
use std::ffi::{CStr, CString};
use std::marker::PhantomData;
use std::ops::{Index, IndexMut};
use std::path::Path;

use lasgun_hip_sys as sys;

fn last_error() -> String {
    unsafe { CStr::from_ptr(sys::lg_last_error()).to_string_lossy().into_owned() }
}

// ---------------------------------------------------------------------------------------------------------------
// Material (src/material/mod.rs:3-46): a Copy value
// ---------------------------------------------------------------------------------------------------------------
#[derive(Clone, Copy, Debug)]
pub struct Material(sys::lg_material);

impl Material {
    /// Default material for cases where a specific one may not be required (material/mod.rs:15)
    pub fn default() -> Material { Material(unsafe { sys::lg_material_default() }) }
    pub fn matte(kd: [f64; 3], sigma: f64) -> Material { Material(unsafe { sys::lg_material_matte(&kd, sigma) }) }
    pub fn plastic(kd: [f64; 3], ks: [f64; 3], roughness: f64) -> Material { Material(unsafe { sys::lg_material_plastic(&kd, &ks, roughness) }) }
    pub fn metal(eta: [f64; 3], k: [f64; 3], u_roughness: f64, v_roughness: f64) -> Material {
        Material(unsafe { sys::lg_material_metal(&eta, &k, u_roughness, v_roughness) })
    }
    pub fn glass(kr: [f64; 3], kt: [f64; 3], eta: f64) -> Material { Material(unsafe { sys::lg_material_glass(&kr, &kt, eta) }) }
    pub fn mirror(kr: [f64; 3]) -> Material { Material(unsafe { sys::lg_material_mirror(&kr) }) }
}

// ---------------------------------------------------------------------------------------------------------------
// Camera (src/camera.rs:75-102): lives in the scene; the setters take effect there
// ---------------------------------------------------------------------------------------------------------------
pub struct Camera {
    scene: *mut sys::lg_scene,
}

impl Camera {
    pub fn look_at(&mut self, origin: [f64; 3], look: [f64; 3], up: [f64; 3]) { unsafe { sys::lg_camera_look_at(self.scene, &origin, &look, &up) } }
    pub fn set_supersampling(&mut self, base: u8) { unsafe { sys::lg_camera_set_supersampling(self.scene, base) } }
    pub fn set_aperture_radius(&mut self, radius: f64) { unsafe { sys::lg_camera_set_aperture_radius(self.scene, radius) } }
}

// ---------------------------------------------------------------------------------------------------------------
// scene::{Scene, Aggregate, ObjRef} (src/scene.rs, src/scene/node.rs)
// ---------------------------------------------------------------------------------------------------------------
pub mod scene {
    use super::*;

    /// Opaque reference to a .obj-powered mesh in a scene (scene.rs:42-44)
    #[derive(Clone, Copy, Debug, PartialEq, Eq)]
    pub struct ObjRef(pub(crate) u32);

    /// The one `Result` on the reference's path (obj::ObjError, scene.rs:120-130)
    #[derive(Debug)]
    pub struct ObjError(pub String);
    impl std::fmt::Display for ObjError {
        fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result { write!(f, "{}", self.0) }
    }
    impl std::error::Error for ObjError {}

    /// A collection of scene nodes under one transformation (node.rs:25-33).  Owned until it is moved into a scene
    /// or a parent group; `scene.root` is a borrowed view of the scene's own root.
    pub struct Aggregate {
        pub(crate) ptr: *mut sys::lg_aggregate,
        owned: bool,
    }

    impl Aggregate {
        pub fn new() -> Aggregate { Aggregate { ptr: unsafe { sys::lg_aggregate_new() }, owned: true } }
        pub(crate) fn borrowed(ptr: *mut sys::lg_aggregate) -> Aggregate { Aggregate { ptr, owned: false } }
        fn into_raw(mut self) -> *mut sys::lg_aggregate {
            assert!(self.owned, "the scene's root aggregate cannot be moved");
            self.owned = false; // ownership goes to the C side
            self.ptr
        }

        pub fn add_group(&mut self, aggregate: Aggregate) { unsafe { sys::lg_aggregate_add_group(self.ptr, aggregate.into_raw()) } }
        pub fn add_sphere(&mut self, center: [f64; 3], radius: f64, material: Material) { unsafe { sys::lg_aggregate_add_sphere(self.ptr, &center, radius, &material.0) } }
        pub fn add_cube(&mut self, origin: [f64; 3], dim: f64, material: Material) { unsafe { sys::lg_aggregate_add_cube(self.ptr, &origin, dim, &material.0) } }
        pub fn add_box(&mut self, minbound: [f64; 3], maxbound: [f64; 3], material: Material) { unsafe { sys::lg_aggregate_add_box(self.ptr, &minbound, &maxbound, &material.0) } }
        /// Add a mesh that provides its own material properties (or defaults to Material::default())
        pub fn add_obj(&mut self, mesh: ObjRef) { unsafe { sys::lg_aggregate_add_obj(self.ptr, mesh.0) } }
        /// Add a mesh that is made of a single material
        pub fn add_obj_of(&mut self, mesh: ObjRef, material: Material) { unsafe { sys::lg_aggregate_add_obj_of(self.ptr, mesh.0, &material.0) } }
        pub fn swap_backface(&mut self) { unsafe { sys::lg_aggregate_swap_backface(self.ptr) } }
        pub fn translate(&mut self, delta: [f64; 3]) -> &mut Self { unsafe { sys::lg_aggregate_translate(self.ptr, &delta) }; self }
        pub fn scale(&mut self, x: f64, y: f64, z: f64) -> &mut Self { unsafe { sys::lg_aggregate_scale(self.ptr, x, y, z) }; self }
        pub fn rotate_x(&mut self, theta: f64) -> &mut Self { unsafe { sys::lg_aggregate_rotate_x(self.ptr, theta) }; self }
        pub fn rotate_y(&mut self, theta: f64) -> &mut Self { unsafe { sys::lg_aggregate_rotate_y(self.ptr, theta) }; self }
        pub fn rotate_z(&mut self, theta: f64) -> &mut Self { unsafe { sys::lg_aggregate_rotate_z(self.ptr, theta) }; self }
        pub fn rotate(&mut self, theta: f64, axis: [f64; 3]) -> &mut Self { unsafe { sys::lg_aggregate_rotate(self.ptr, theta, &axis) }; self }
    }

    impl Drop for Aggregate {
        fn drop(&mut self) {
            if self.owned { unsafe { sys::lg_aggregate_free(self.ptr) } }
        }
    }

    /// Description of the world to render and how it should be rendered (scene.rs:11-40)
    pub struct Scene {
        pub(crate) ptr: *mut sys::lg_scene,
        /// The root node: `scene.root.add_sphere(..)` as in the reference's examples (scene.rs:14)
        pub root: Aggregate,
        camera: Camera,
    }

    impl Scene {
        pub fn new() -> Scene {
            let ptr = unsafe { sys::lg_scene_new() };
            Scene { ptr, root: Aggregate::borrowed(unsafe { sys::lg_scene_root(ptr) }), camera: Camera { scene: ptr } }
        }
        pub fn set_perspective_camera(&mut self, fov: f64) -> &mut Camera { unsafe { sys::lg_scene_set_perspective_camera(self.ptr, fov) }; &mut self.camera }
        pub fn set_orthographic_camera(&mut self, scale: f64) -> &mut Camera { unsafe { sys::lg_scene_set_orthographic_camera(self.ptr, scale) }; &mut self.camera }
        pub fn set_solid_background(&mut self, color: [f64; 3]) { unsafe { sys::lg_scene_set_solid_background(self.ptr, &color) } }
        pub fn set_radial_background(&mut self, inner: [f64; 3], outer: [f64; 3], scale: f64) { unsafe { sys::lg_scene_set_radial_background(self.ptr, &inner, &outer, scale) } }
        pub fn set_ambient_light(&mut self, color: [f64; 3]) { unsafe { sys::lg_scene_set_ambient_light(self.ptr, &color) } }
        pub fn set_mesh_smoothing(&mut self, enabled: bool) { unsafe { sys::lg_scene_set_mesh_smoothing(self.ptr, enabled as i32) } }
        pub fn set_max_recursion_depth(&mut self, max_depth: u32) { unsafe { sys::lg_scene_set_max_recursion_depth(self.ptr, max_depth) } }
        /// Zero: every GPU selected with `set_devices`; otherwise a cap on how many of them a capture is split over
        pub fn set_threads(&mut self, threads: usize) { unsafe { sys::lg_scene_set_threads(self.ptr, threads) } }
        pub fn add_point_light(&mut self, position: [f64; 3], intensity: [f64; 3], falloff: [f64; 3]) {
            unsafe { sys::lg_scene_add_point_light(self.ptr, &position, &intensity, &falloff) }
        }
        /// Triangle mesh from the string contents of a .obj file (scene.rs:120-123)
        pub fn parse_obj(&mut self, obj: &str) -> Result<ObjRef, ObjError> {
            let mut reference = 0u32;
            let rc = unsafe { sys::lg_scene_parse_obj(self.ptr, obj.as_ptr() as *const _, obj.len(), &mut reference) };
            if rc != 0 { Err(ObjError(last_error())) } else { Ok(ObjRef(reference)) }
        }
        /// Load the .obj file at the given path (scene.rs:127-130)
        pub fn load_obj(&mut self, obj_path: &Path) -> Result<ObjRef, ObjError> {
            let c = CString::new(obj_path.to_string_lossy().as_bytes()).map_err(|e| ObjError(e.to_string()))?;
            let mut reference = 0u32;
            let rc = unsafe { sys::lg_scene_load_obj(self.ptr, c.as_ptr(), &mut reference) };
            if rc != 0 { Err(ObjError(last_error())) } else { Ok(ObjRef(reference)) }
        }
        pub fn set_root(&mut self, node: Aggregate) {
            unsafe { sys::lg_scene_set_root(self.ptr, node.into_raw()) };
            self.root = Aggregate::borrowed(unsafe { sys::lg_scene_root(self.ptr) });
        }
    }

    impl Drop for Scene {
        fn drop(&mut self) { unsafe { sys::lg_scene_free(self.ptr) } }
    }
}

pub use crate::scene::Scene;

// ---------------------------------------------------------------------------------------------------------------
// Img / Film (src/img.rs:9-67, src/film.rs:7-45)
// ---------------------------------------------------------------------------------------------------------------
pub type Pixel = [u8; 4];

/// Store of pixels in row-major order that can be saved somewhere (img.rs:9-13)
pub trait PixelBuffer: Index<usize, Output = Pixel> + IndexMut<usize> {
    fn save(&self, filename: &str);
}
impl PixelBuffer for Vec<Pixel> {
    fn save(&self, _filename: &str) {}
}

/// What `capture_subset` writes to (img.rs:16-52)
pub trait Img {
    fn w(&self) -> u32;
    fn h(&self) -> u32;
    fn set(&mut self, x: u32, y: u32, color: &Pixel);
}

pub struct Film {
    pub w: u32,
    pub h: u32,
    pub winv: f64,
    pub hinv: f64,
    pub aspect: f64,
    output: Box<dyn PixelBuffer<Output = Pixel>>,
}

impl Film {
    /// A film of the given dimensions, every pixel black and transparent (film.rs:24)
    pub fn new(width: u32, height: u32) -> Film {
        let area = (width as usize) * (height as usize);
        Film::new_with_output(width, height, Box::new(vec![[0u8, 0, 0, 0]; area]))
    }
    /// A film over a pre-allocated pixel store of width * height pixels (film.rs:36)
    pub fn new_with_output(width: u32, height: u32, output: Box<dyn PixelBuffer<Output = Pixel>>) -> Film {
        Film { w: width, h: height, winv: 1. / width as f64, hinv: 1. / height as f64, aspect: width as f64 / height as f64, output }
    }
    pub fn save(&self, filename: &str) { self.output.save(filename) }
}
impl Index<usize> for Film {
    type Output = Pixel;
    fn index(&self, at: usize) -> &Pixel { &self.output[at] }
}
impl IndexMut<usize> for Film {
    fn index_mut(&mut self, at: usize) -> &mut Pixel { &mut self.output[at] }
}
impl Img for Film {
    fn w(&self) -> u32 { self.w }
    fn h(&self) -> u32 { self.h }
    fn set(&mut self, x: u32, y: u32, color: &Pixel) {
        let at = (y as usize) * (self.w as usize) + x as usize;
        self.output[at] = *color
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Accel, render, capture, capture_subset (src/lib.rs:42-162)
// ---------------------------------------------------------------------------------------------------------------
/// The acceleration structure of a scene: the reference's nested HLBVH, built on the host exactly as the reference
/// builds it, flattened and resident in HBM.  Borrows the scene for its whole life (bvh.rs:45-46).
pub struct Accel<'s> {
    ptr: *mut sys::lg_accel,
    scene: PhantomData<&'s Scene>,
}

impl<'s> Accel<'s> {
    pub fn from(scene: &'s Scene) -> Accel<'s> {
        let ptr = unsafe { sys::lg_accel_from(scene.ptr) };
        if ptr.is_null() { panic!("lasgun: {}", last_error()) }
        Accel { ptr, scene: PhantomData }
    }
    /// false (default): the reference's own traversal over the reference's own BVH; true: the opt-in fast mode
    pub fn set_fast_mode(&self, fast: bool) {
        if unsafe { sys::lg_accel_set_mode(self.ptr, fast as i32) } != 0 { panic!("lasgun: {}", last_error()) }
    }
}
impl<'s> Drop for Accel<'s> {
    fn drop(&mut self) { unsafe { sys::lg_accel_free(self.ptr) } }
}
// The reference shares `&Accel` between its capture threads (lib.rs:67-103).  The C side guards every entry point that takes
// an accel with that accel's own mutex (internal.h: lg_accel), so the handle may be used and dropped from any thread.
unsafe impl<'s> Send for Accel<'s> {}
unsafe impl<'s> Sync for Accel<'s> {}

/// GPUs a `capture` / `render` is split over (none given: every visible device); see `Scene::set_threads`
pub fn set_devices(devices: &[i32]) {
    if unsafe { sys::lg_set_devices(devices.as_ptr(), devices.len() as i32) } != 0 { panic!("lasgun: {}", last_error()) }
}

/// Render the given scene (lib.rs:46-50)
pub fn render(scene: &Scene, resolution: (u32, u32)) -> Film {
    let mut film = Film::new(resolution.0, resolution.1);
    capture(scene, &mut film);
    film
}

/// Record an image of the scene on the given film (lib.rs:55-104): the BVH is (re)built inside, the call returns when
/// every pixel is written.
pub fn capture(scene: &Scene, film: &mut Film) {
    let (w, h) = (film.w, film.h);
    let staging = unsafe { sys::lg_film_new(w, h) };
    let rc = unsafe { sys::lg_capture(scene.ptr, staging) };
    if rc != 0 {
        unsafe { sys::lg_film_free(staging) };
        panic!("lasgun: {}", last_error())
    }
    let px = unsafe { std::slice::from_raw_parts(sys::lg_film_pixels(staging), (w as usize) * (h as usize) * 4) };
    for (at, p) in px.chunks_exact(4).enumerate() { film[at] = [p[0], p[1], p[2], p[3]] }
    unsafe { sys::lg_film_free(staging) }
}

/// Capture subset k of n: every pixel {k + i*n} of the row-major pixel buffer and no other (lib.rs:110-162)
pub fn capture_subset(k: usize, n: usize, root: &Accel, img: &mut impl Img) {
    assert!(n > 0);
    let (w, h) = (img.w(), img.h());
    let area = (w as usize) * (h as usize);
    let offsets: Vec<u64> = (k..area).step_by(n).map(|o| o as u64).collect();
    if offsets.is_empty() { return }
    let mut rgba = vec![0u8; offsets.len() * 4];
    let rc = unsafe { sys::lg_capture_pixels(root.ptr, w, h, offsets.as_ptr(), offsets.len(), rgba.as_mut_ptr(), std::ptr::null_mut()) };
    if rc != 0 { panic!("lasgun: {}", last_error()) }
    for (i, offset) in offsets.iter().enumerate() {
        let (x, y) = ((*offset as usize % w as usize) as u32, (*offset as usize / w as usize) as u32);
        img.set(x, y, &[rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2], rgba[4 * i + 3]])
    }
}

/// Several subsets of one `n` in ONE render: writes what the calls `capture_subset(k, n, ..)` for `k` in `ks` write, and nothing else.
/// No counterpart in the reference, whose progressive front end calls `capture_subset` a hundred times in a row
/// (www/renderer.ts:103-120); on a GPU a batch costs `ks.len() / n` of a frame instead of a launch chain per subset.
pub fn capture_subsets(ks: &[usize], n: usize, root: &Accel, img: &mut impl Img) {
    assert!(n > 0);
    let (w, h) = (img.w(), img.h());
    let area = (w as usize) * (h as usize);
    let staging = unsafe { sys::lg_film_new(w, h) };
    let rc = unsafe { sys::lg_capture_subsets(ks.as_ptr(), ks.len(), n, root.ptr, staging) };
    if rc != 0 {
        unsafe { sys::lg_film_free(staging) };
        panic!("lasgun: {}", last_error())
    }
    let px = unsafe { std::slice::from_raw_parts(sys::lg_film_pixels(staging), area * 4) };
    for &k in ks {
        for o in (k..area).step_by(n) {
            img.set((o % w as usize) as u32, (o / w as usize) as u32, &[px[4 * o], px[4 * o + 1], px[4 * o + 2], px[4 * o + 3]])
        }
    }
    unsafe { sys::lg_film_free(staging) }
}

// ---------------------------------------------------------------------------------------------------------------
// The table of measured kernel-organisation choices (no counterpart in the reference; include/lasgun_hip.h, lg_tune_*):
// export it once, import it at start-up, and no launch of a known kind is ever measured again.
// ---------------------------------------------------------------------------------------------------------------
pub mod tune {
    use super::*;
    pub use sys::lg_tune_entry as Entry;
    pub fn export() -> Vec<Entry> {
        let n = unsafe { sys::lg_tune_export(std::ptr::null_mut(), 0) };
        let mut v = vec![Entry::default(); n];
        let m = unsafe { sys::lg_tune_export(v.as_mut_ptr(), n) };
        v.truncate(m.min(n));
        v
    }
    pub fn import(entries: &[Entry]) { if unsafe { sys::lg_tune_import(entries.as_ptr(), entries.len()) } != 0 { panic!("lasgun: {}", last_error()) } }
    pub fn clear() { unsafe { sys::lg_tune_clear() } }
}

// ---------------------------------------------------------------------------------------------------------------
// output::render (src/output.rs:5-18): render to a PNG file
// ---------------------------------------------------------------------------------------------------------------
pub mod output {
    use super::*;
    use std::io::Write;

    struct Image { w: u32, h: u32, px: Vec<Pixel> }
    impl Index<usize> for Image {
        type Output = Pixel;
        fn index(&self, at: usize) -> &Pixel { &self.px[at] }
    }
    impl IndexMut<usize> for Image {
        fn index_mut(&mut self, at: usize) -> &mut Pixel { &mut self.px[at] }
    }
    impl PixelBuffer for Image {
        fn save(&self, filename: &str) { write_png(filename, self.w, self.h, &self.px).unwrap() }
    }

    pub fn render(scene: &Scene, resolution: [u32; 2], filename: &str) {
        let mut film = self::film(resolution);
        capture(scene, &mut film);
        film.save(filename)
    }
    /// A film in the given dimensions whose `save` writes a PNG
    pub fn film(resolution: [u32; 2]) -> Film {
        let (w, h) = (resolution[0], resolution[1]);
        Film::new_with_output(w, h, Box::new(Image { w, h, px: vec![[0, 0, 0, 0]; (w as usize) * (h as usize)] }))
    }

    fn crc32(data: &[u8]) -> u32 {
        let mut c = 0xFFFF_FFFFu32;
        for &b in data {
            c ^= b as u32;
            for _ in 0..8 { c = if c & 1 != 0 { 0xEDB8_8320 ^ (c >> 1) } else { c >> 1 } }
        }
        !c
    }
    fn chunk(out: &mut Vec<u8>, kind: &[u8; 4], data: &[u8]) {
        out.extend_from_slice(&(data.len() as u32).to_be_bytes());
        let mut body = kind.to_vec();
        body.extend_from_slice(data);
        out.extend_from_slice(&body);
        out.extend_from_slice(&crc32(&body).to_be_bytes());
    }
    /// 8-bit RGBA PNG, filter 0 rows in stored (uncompressed) deflate blocks
    fn write_png(filename: &str, w: u32, h: u32, px: &[Pixel]) -> std::io::Result<()> {
        let mut raw = Vec::with_capacity((w as usize * 4 + 1) * h as usize);
        for row in px.chunks((w as usize).max(1)) {
            raw.push(0u8);
            for p in row { raw.extend_from_slice(p) }
        }
        let (mut a, mut b) = (1u32, 0u32); // adler32
        for &v in &raw { a = (a + v as u32) % 65521; b = (b + a) % 65521 }
        let mut z = vec![0x78u8, 0x01];
        if raw.is_empty() { z.extend_from_slice(&[1, 0, 0, 0xFF, 0xFF]) } // a film without pixels: one empty final block
        let mut blocks = raw.chunks(65535).peekable();
        while let Some(block) = blocks.next() {
            z.push(if blocks.peek().is_none() { 1 } else { 0 });
            z.extend_from_slice(&(block.len() as u16).to_le_bytes());
            z.extend_from_slice(&(!(block.len() as u16)).to_le_bytes());
            z.extend_from_slice(block);
        }
        z.extend_from_slice(&((b << 16) | a).to_be_bytes());
        let mut out = vec![0x89, b'P', b'N', b'G', 0x0D, 0x0A, 0x1A, 0x0A];
        let mut ihdr = Vec::new();
        ihdr.extend_from_slice(&w.to_be_bytes());
        ihdr.extend_from_slice(&h.to_be_bytes());
        ihdr.extend_from_slice(&[8, 6, 0, 0, 0]);
        chunk(&mut out, b"IHDR", &ihdr);
        chunk(&mut out, b"IDAT", &z);
        chunk(&mut out, b"IEND", &[]);
        std::fs::File::create(filename)?.write_all(&out)
    }
}
