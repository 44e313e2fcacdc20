"""Scenes x sizes of the committed golden fixtures (tests/golden/*.npz)."""
from lasgun_amd import scenes as S

CASES = {
    "readme": (S.readme_scene, 48, 48),
    "simple_ss2": (lambda api: S.simple_scene(api, supersampling=2), 48, 48),
    "simplereflect": (lambda api: S.simple_scene(api, supersampling=0, reflect=True), 48, 48),
    "cornell_plastic": (lambda api: S.cornell_scene(api, "plastic"), 64, 64),
    "cornell_glass": (lambda api: S.cornell_scene(api, "glass"), 64, 64),
    "spheres1024": (S.spheres_scene, 64, 64),
    "mesh_glass": (lambda api: S.mesh_scene(api, 24, 24, "glass"), 48, 48),
    "mesh_metal_flat": (lambda api: S.mesh_scene(api, 24, 24, "metal", smoothing=False), 48, 40),
    "mesh_default": (lambda api: S.mesh_scene(api, 16, 16, "default"), 40, 48),
}
