"""The Rust shim (bindings/rust) cannot be compiled in this image (no Rust toolchain), so its FFI layer is checked
against the C ABI textually: every entry point of include/lasgun_hip.h is declared in lasgun-hip-sys with the same
name and arity, the committed file is exactly what tools/gen_rust_sys.py generates from the header, the safe
crate only calls declared symbols, and it offers the reference's public names (src/lib.rs:42-56,110,
src/scene.rs:49-143, src/scene/node.rs:35-115, src/material/mod.rs:15-46, src/camera.rs:85-102, src/film.rs:22-45)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYS = os.path.join(ROOT, "bindings", "rust", "lasgun-hip-sys", "src", "lib.rs")
SAFE = os.path.join(ROOT, "bindings", "rust", "lasgun", "src", "lib.rs")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def split_args(text):
    """Top-level comma split (Rust types such as `*const [f64; 3]` carry brackets)."""
    out, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([<":
            depth += 1
        elif ch in ")]>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def header_functions():
    import gen_rust_sys
    return {name: (ret, params) for ret, name, params in gen_rust_sys.declarations(open(os.path.join(ROOT, "include", "lasgun_hip.h")).read())}


def rust_functions():
    text = open(SYS).read()
    block = text[text.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (lg_\w+)\((.*?)\)( -> [^;]+)?;", block):
        out[m.group(1)] = (split_args(m.group(2)), (m.group(3) or "").replace(" -> ", "").strip())
    return out


def test_sys_crate_declares_every_entry_point_with_the_same_arity():
    c, r = header_functions(), rust_functions()
    assert len(c) > 80
    assert set(c) == set(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name, (ret, params) in c.items():
        rargs, rret = r[name]
        assert len(rargs) == len(params), name
        assert (ret == "void") == (rret == ""), name
        for ctype, rarg in zip(params, rargs):
            rtype = rarg.split(":", 1)[1].strip()
            assert ("*" in ctype or "[" in ctype) == rtype.startswith("*"), (name, ctype, rtype)  # pointers stay pointers
            if "double" in ctype:
                assert "f64" in rtype, (name, ctype, rtype)
            if re.search(r"\[(\d+)\]", ctype):
                assert "; %s]" % re.search(r"\[(\d+)\]", ctype).group(1) in rtype, (name, ctype, rtype)


def test_committed_sys_crate_is_what_the_generator_produces():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py")], capture_output=True, text=True, check=True).stdout
    assert out == open(SYS).read(), "regenerate: python tools/gen_rust_sys.py > bindings/rust/lasgun-hip-sys/src/lib.rs"


def test_safe_crate_calls_only_declared_symbols_and_offers_the_reference_surface():
    src = open(SAFE).read()
    used = set(re.findall(r"sys::(lg_\w+)\(", src))
    declared = set(rust_functions())
    assert used and used <= declared, sorted(used - declared)
    # the reference's public names for this path
    for needle in ("pub struct Accel<'s>", "pub fn from(scene: &'s Scene) -> Accel<'s>", "pub fn render(scene: &Scene, resolution: (u32, u32)) -> Film",
                   "pub fn capture(scene: &Scene, film: &mut Film)", "pub fn capture_subset(k: usize, n: usize, root: &Accel, img: &mut impl Img)",
                   "pub fn new_with_output(width: u32, height: u32", "pub root: Aggregate", "pub struct ObjRef",
                   "pub fn parse_obj(&mut self, obj: &str) -> Result<ObjRef, ObjError>", "pub fn load_obj(&mut self, obj_path: &Path) -> Result<ObjRef, ObjError>"):
        assert needle in src, needle
    scene_fns = ("set_perspective_camera", "set_orthographic_camera", "set_solid_background", "set_radial_background", "set_ambient_light",
                 "set_mesh_smoothing", "set_max_recursion_depth", "set_threads", "add_point_light", "set_root")
    node_fns = ("add_group", "add_sphere", "add_cube", "add_box", "add_obj", "add_obj_of", "swap_backface", "translate", "scale", "rotate_x",
                "rotate_y", "rotate_z", "rotate")
    mat_fns = ("default", "matte", "plastic", "metal", "glass", "mirror")
    cam_fns = ("look_at", "set_supersampling", "set_aperture_radius")
    for fn in scene_fns + node_fns + mat_fns + cam_fns:
        assert re.search(r"pub fn %s\(" % fn, src), fn


def test_example_programs_use_only_what_the_shim_offers():
    """The crate's examples are this build's own programs (configs[2], configs[3], a progressive capture_subset render);
    every method they call must exist in the safe crate, and none of them may be a copy of a reference example."""
    src = open(SAFE).read()
    offered = set(re.findall(r"pub fn (\w+)\(", src))
    ex_dir = os.path.join(ROOT, "bindings", "rust", "lasgun", "examples")
    names = sorted(n for n in os.listdir(ex_dir) if n.endswith(".rs"))
    assert names == ["progressive.rs", "spheres1024.rs", "torus_glass.rs"]
    manifest = open(os.path.join(ROOT, "bindings", "rust", "lasgun", "Cargo.toml")).read()
    seen = 0
    for name in names + [os.path.join("common", "mod.rs")]:
        text = open(os.path.join(ex_dir, name)).read()
        if not name.startswith("common"):
            assert 'name = "%s"' % name[:-3] in manifest
            assert "use ::lasgun::{" in text and "mod common;" in text and "fn main()" in text
        for call in (re.findall(r"(?:scene|camera|scene\.root|side|group|accel|film)\.(\w+)\(", text) + re.findall(r"Material::(\w+)\(", text)
                     + re.findall(r"output::(\w+)\(", text) + re.findall(r"\b(capture_subset)\(", text)):
            assert call in offered, (name, call)
            seen += 1
    assert seen > 30
    ref_examples = os.path.join("/root/reference", "src", "examples")
    if os.path.isdir(ref_examples):  # (this container only; the GPU box has no reference)
        import difflib
        for name in names:
            mine = [l.strip() for l in open(os.path.join(ex_dir, name)) if l.strip()]
            for ref in os.listdir(ref_examples):
                if ref.endswith(".rs"):
                    theirs = [l.strip() for l in open(os.path.join(ref_examples, ref)) if l.strip()]
                    assert difflib.SequenceMatcher(None, mine, theirs).ratio() < 0.3, (name, ref)
