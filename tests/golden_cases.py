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
    # the reference's remaining example programs (src/examples/playground.rs, spooky.rs, simplecows.rs) with generated stand-in meshes
    "playground": (lambda api: S.playground_scene(api, 24, 14), 48, 48),
    "spooky": (lambda api: S.spooky_scene(api, 20, 12), 48, 48),
    "simplecows": (S.simplecows_scene, 48, 48),
}

# 64x64 crops of the FULL-SIZE films of BASELINE.json's configs[3] (100k-triangle mesh, glass / metal, mirror sphere:
# secondary rays) and configs[4] (8192x8192 mixed mesh + spheres): name -> (builder, film w, film h, x0, y0, crop w, crop h).
# The crops sit on the torus' silhouette / its refractions and on sphere clusters in front of the mesh.
CROPS = {
    "config4_glass_crop_a": (lambda api: S.mesh_scene(api, 224, 224, "glass"), 4096, 4096, 1080, 1500, 64, 64),
    "config4_glass_crop_b": (lambda api: S.mesh_scene(api, 224, 224, "glass"), 4096, 4096, 1800, 1900, 64, 64),
    "config4_metal_crop": (lambda api: S.mesh_scene(api, 224, 224, "metal"), 4096, 4096, 1080, 1500, 64, 64),
    "config5_mixed_crop_a": (lambda api: S.mixed_scene(api), 8192, 8192, 4000, 3000, 64, 64),
    "config5_mixed_crop_b": (lambda api: S.mixed_scene(api), 8192, 8192, 3000, 4500, 64, 64),
}
