"""Closed-form known answers that need no reference run (hand-derived from the cited formulas, evaluated in numpy):
camera rays of corner / centre pixels through the radial background (src/camera.rs:113-146, src/material/background.rs:25-34),
and the Lambertian radiance of a sphere pixel under one point light (src/shape/sphere.rs:30-69, src/integrate/integrate.rs:47-67,
src/light/point.rs:42-54, src/material/matte.rs:18-26).  They do not PIN parity -- only the reference's 17 inline tests do --
but they are independent of the oracle's code and of the device's, so a transcription slip shared by neither would show here.
Run on the oracle (CPU) and, with -m gpu, on the device."""
import math

import numpy as np
import pytest

from oracle_lib import oracle


def camera_ray(eye, look, up, fov, w, h, x, y):
    """Camera::look_at + Camera::sample for supersampling 0 (one sample per pixel), perspective."""
    eye, look, up = (np.array(v, float) for v in (eye, look, up))
    view = look - eye
    aux = np.cross(view, up)
    upn = np.cross(aux, view); upn = upn * (1.0 / math.sqrt(upn @ upn))
    aux = aux * (1.0 / math.sqrt(aux @ aux))
    H = math.sqrt(view @ view) * math.tan(fov * math.pi / 360.0) * 2.0
    winv, hinv, aspect = 1.0 / w, 1.0 / h, w / h
    W = H * aspect
    pixel = H * hinv
    sox = (x * winv - 0.5) * W
    soy = (0.5 - (y + 1) * hinv) * H
    d = view + soy * upn + sox * aux
    half = (upn * pixel) * 0.5 + (aux * pixel) * 0.5  # sample separation = distance (1) * pixel size
    return eye, d + half


def background(d, inner, outer, scale):
    d = d * (1.0 / math.sqrt(d @ d))
    t = min(math.sqrt(1.0 - abs(d[2]) ** 2) / scale, 1.0)
    return np.array(inner) * (1.0 - t) + np.array(outer) * t


def bg_scene(api, eye, look, fov):
    sc = api.Scene.new()
    cam = sc.set_perspective_camera(fov)
    cam.look_at(eye, look, [0.0, 1.0, 0.0])
    sc.set_radial_background([0.26, 0.78, 0.67], [0.1, 0.09, 0.33], 0.5)
    sc.root.add_sphere([0.0, 1.0e6, 0.0], 1.0, api.Material.matte([0.5, 0.5, 0.5], 0.0))  # far outside the view: every pixel is background
    return sc


def check_background(api, radiance):
    w, h, fov = 64, 48, 50.0
    for eye, look in (([0.0, 0.0, 0.0], [0.0, 0.0, 1.0]), ([1.0, 2.0, 3.0], [4.0, 1.0, -2.0])):
        acc = api.Accel(bg_scene(api, eye, look, fov))
        rad = radiance(acc, w, h)
        for x, y in ((0, 0), (w - 1, h - 1), (w // 2, h // 2), (w - 1, 0)):
            _, d = camera_ray(eye, look, [0.0, 1.0, 0.0], fov, w, h, x, y)
            want = background(d, [0.26, 0.78, 0.67], [0.1, 0.09, 0.33], 0.5)
            assert np.allclose(rad[y, x], want, rtol=1e-13, atol=0.0), (eye, x, y, rad[y, x], want)
    # Background::bg at d = (0, 0, 1) is the inner colour, at d = (1, 0, 0) the outer one (scale <= 1)
    assert np.allclose(background(np.array([0.0, 0.0, 1.0]), [0.26, 0.78, 0.67], [0.1, 0.09, 0.33], 0.5), [0.26, 0.78, 0.67])
    assert np.allclose(background(np.array([1.0, 0.0, 0.0]), [0.26, 0.78, 0.67], [0.1, 0.09, 0.33], 0.5), [0.1, 0.09, 0.33])


def check_lambert_sphere(api, radiance):
    w, h, fov = 33, 33, 45.0
    eye, look = [0.0, 0.0, 0.0], [0.0, 0.0, 1.0]
    kd, ambient = np.array([0.7, 1.0, 0.7]), np.array([0.1, 0.2, 0.3])
    lpos, lint, falloff = np.array([30.0, 60.0, -40.0]), np.array([0.8, 0.6, 0.9]), np.array([1.0, 0.001, 0.0])
    c, r = np.array([0.0, 0.0, 100.0]), 50.0
    sc = api.Scene.new()
    sc.set_perspective_camera(fov).look_at(eye, look, [0.0, 1.0, 0.0])
    sc.set_ambient_light(ambient.tolist())
    sc.add_point_light(lpos.tolist(), lint.tolist(), falloff.tolist())
    sc.root.add_sphere(c.tolist(), r, api.Material.matte(kd.tolist(), 0.0))
    rad = radiance(api.Accel(sc), w, h)
    for x, y in ((16, 16), (12, 20), (20, 13)):
        o, d = camera_ray(eye, look, [0.0, 1.0, 0.0], fov, w, h, x, y)
        l = o - c
        a, b, cc = d @ d, 2.0 * (d @ l), l @ l - r * r
        t = (-b - math.sqrt(b * b - 4.0 * a * cc)) / (2.0 * a)  # nearer root: the camera is outside the sphere
        p = o + d * t
        n = (p - c) / r
        wi = lpos - p
        dist = math.sqrt(wi @ wi)
        f_att = falloff[0] + falloff[1] * dist + falloff[2] * dist * dist
        wi = wi / dist
        want = lint * kd * (wi @ n) / f_att + ambient * kd / math.pi  # pi * I * (kd / pi) * cos / f_att + ambient * (kd / pi)
        assert (wi @ n) > 0.0
        assert np.allclose(rad[y, x], want, rtol=1e-9, atol=0.0), (x, y, rad[y, x], want)


def test_closed_forms_on_the_oracle():
    o = oracle()
    check_background(o, lambda acc, w, h: o.capture_radiance(acc, w, h, nthreads=2))
    check_lambert_sphere(o, lambda acc, w, h: o.capture_radiance(acc, w, h, nthreads=2))


@pytest.mark.gpu
def test_closed_forms_on_the_device():
    import lasgun_amd as la
    G = la.api
    for streaming in (0, 2):
        def radiance(acc, w, h):
            G.set_streaming(acc, streaming)
            return G.capture_radiance(acc, w, h)
        check_background(G, radiance)
        check_lambert_sphere(G, radiance)
