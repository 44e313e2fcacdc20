// Where the example scenes find their .obj meshes: the reference looks in ./www/public/meshes/<name>.obj
// (src/examples/meshes/mod.rs); its mesh files are git-lfs stubs, so the plane the Cornell walls are made of is also
// kept here as text -- the reference's own inline fixture (src/shape/triangle.rs:412-421).
#![allow(dead_code)]
use std::path::PathBuf;

pub fn path(name: &str) -> PathBuf {
    let mut path = PathBuf::new();
    path.push(".");
    path.push("www");
    path.push("public");
    path.push("meshes");
    path.push(name);
    path.set_extension("obj");
    path
}

pub const PLANE_OBJ: &str = "o plane\nv -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\n\nf 1 2 3\nf 1 3 4\n";
