"""The oracle against an independent plain-Python restatement of the reference's shading path (tests/pyref.py):
camera, sphere differentials, SurfaceInteraction, all five materials and their BxDFs, lights and shadow rays, specular
recursion, background, quantisation -- on sphere-only scenes, libm trigonometry on both sides.  CPU only."""
import math

import numpy as np
import pytest

import pyref
from oracle_lib import oracle

import lasgun_amd as la

S = la.scenes


def witness_scene(api, variant):
    """Spheres of every material, well separated (no exact ties in t), two lights with different falloffs, ambient light,
    radial background; variants change the camera, the supersampling and the recursion depth."""
    M = api.Material
    scene = api.Scene.new()
    scene.set_ambient_light([0.15, 0.1, 0.2])
    scene.set_radial_background([0.26, 0.78, 0.67], [0.1, 0.09, 0.33], 0.5)
    if variant == "ortho":
        cam = scene.set_orthographic_camera(9.0)
    else:
        cam = scene.set_perspective_camera(50.0)
    cam.look_at([0.4, 1.1, 9.0], [0.1, 0.0, 0.0], [0.05, 1.0, 0.0])
    if variant == "ss":
        cam.set_supersampling(1)
    scene.set_max_recursion_depth({"shallow": 1, "deep": 5}.get(variant, 3))
    scene.add_point_light([-6.0, 7.0, 8.0], [0.9, 0.85, 0.8], [1.0, 0.0, 0.0])
    scene.add_point_light([5.0, 3.0, 6.0], [0.5, 0.6, 0.9], [0.4, 0.03, 0.002])
    root = scene.root
    root.add_sphere([0.0, -101.5, 0.0], 100.0, M.matte([0.55, 0.5, 0.45], 0.0))
    root.add_sphere([-2.6, -0.3, 0.4], 1.1, M.plastic([0.7, 0.2, 0.2], [0.5, 0.7, 0.5], 0.25))
    root.add_sphere([-0.2, -0.5, 1.6], 0.9, M.glass([1.0, 0.7, 1.0], [0.7, 1.0, 0.7], 1.25))
    root.add_sphere([2.3, -0.1, 0.2], 1.3, M.metal([0.2, 0.9, 1.1], [3.9, 2.4, 2.2], 0.15, 0.3))
    root.add_sphere([0.9, 1.9, -1.5], 1.0, M.mirror([0.8, 0.8, 0.6]))
    root.add_sphere([-1.5, 1.8, -2.0], 0.8, M.matte([0.3, 0.5, 0.8], 35.0))
    root.add_sphere([3.4, 2.2, -2.5], 0.7, M.plastic([0.0, 0.0, 0.0], [0.9, 0.6, 0.2], 0.4))
    return scene


def test_constants():
    assert pyref.FRAC_1_PI == 1.0 / math.pi or abs(pyref.FRAC_1_PI - 1.0 / math.pi) <= 2.0 ** -54
    assert pyref.signum(0.0) == 1.0 and pyref.signum(-0.0) == -1.0
    assert [pyref.to_byte(v) for v in (-1.0, 0.0, 0.5 / 255.0, 1.0, 7.0, float("nan"))] == [0, 0, 1, 255, 255, 0]


@pytest.mark.parametrize("name, w, h", [("readme", 40, 40), ("base", 44, 33), ("ortho", 36, 27), ("ss", 24, 18), ("shallow", 36, 27), ("deep", 36, 27),
                                        ("simplereflect", 40, 30)])
def test_oracle_matches_the_python_witness(name, w, h):
    def build(api):
        if name == "readme":
            return S.readme_scene(api)
        if name == "simplereflect":  # src/examples/simplereflect.rs without its cube and mesh: glass and mirror spheres, recursion 4
            return simple_spheres(api)
        return witness_scene(api, name)

    o = oracle()
    oacc = o.Accel(build(o))
    ofilm = o.Film(w, h)
    o.capture_subset_mt(0, 1, oacc, ofilm, 8)
    orad = o.capture_radiance(oacc, w, h, nthreads=8)  # libm trigonometry (trig mode 0), as the reference
    prad, prgba = pyref.render(build(pyref.Api), w, h)
    prad = np.asarray(prad, dtype=np.float64)
    prgba = np.asarray(prgba, dtype=np.uint8)
    assert np.array_equal(prgba, ofilm.pixels()), "RGBA8 differs at %d pixels" % int((prgba != ofilm.pixels()).any(axis=-1).sum())
    # two restatements, one operation order: equal to the last bit in practice; the assertion leaves room for a reassociation
    assert np.allclose(prad, orad, rtol=1e-12, atol=1e-15)
    same = int((prad.view(np.uint64) == np.asarray(orad).view(np.uint64)).all(axis=-1).sum())
    print("%s: %d of %d pixels bit-identical in f64 radiance" % (name, same, w * h))
    assert same >= (w * h * 99) // 100


def simple_spheres(api):
    scene = api.Scene.new()
    scene.set_ambient_light([0.2, 0.2, 0.2])
    scene.set_radial_background([0.93, 0.87, 0.36], [0.94, 0.6, 0.1], 0.5)
    scene.set_max_recursion_depth(4)
    camera = scene.set_perspective_camera(45.0)
    camera.look_at([25.0, 0.0, 800.0], [25.0, 0.0, 0.0], [0.0, 1.0, 0.0])
    M = api.Material
    mat0 = M.glass([0.7, 1.0, 0.7], [0.5, 0.7, 0.5], 1.333)
    mat1 = M.mirror([0.5, 0.5, 0.5])
    mat2 = M.glass([1.0, 0.6, 0.1], [0.7, 0.7, 1.0], 1.75)
    scene.add_point_light([-100.0, 150.0, 400.0], [0.9, 0.9, 0.9], [1.0, 0.0, 0.0])
    scene.add_point_light([400.0, 100.0, 150.0], [0.7, 0.0, 0.7], [1.0, 0.0, 0.0])
    root = scene.root
    root.add_sphere([0.0, 0.0, -400.0], 100.0, mat0)
    root.add_sphere([200.0, 50.0, -100.0], 150.0, mat0)
    root.add_sphere([0.0, -1200.0, -500.0], 1000.0, mat1)
    root.add_sphere([-100.0, 25.0, -300.0], 50.0, mat2)
    root.add_sphere([0.0, 100.0, -250.0], 25.0, mat0)
    return scene
