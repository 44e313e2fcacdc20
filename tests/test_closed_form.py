"""Closed-form known answers that need no reference run (hand-derived from the cited formulas, evaluated in numpy):
camera rays of corner / centre pixels through the radial background (src/camera.rs:113-146, src/material/background.rs:25-34),
and the Lambertian radiance of a sphere pixel under one point light (src/shape/sphere.rs:30-69, src/integrate/integrate.rs:47-67,
src/light/point.rs:42-54, src/material/matte.rs:18-26), the radiance a mirror sphere reflects from that background (src/core/bxdf/specular.rs:17-24,
src/integrate/integrate.rs:84-108), and a lit / shadowed point on a box face (src/shape/cuboid.rs:55-102).  They do not PIN parity -- only the reference's 17 inline tests do --
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


def camera_rays(eye, look, up, w, h, x, y, base=0, fov=None, ortho=None):
    """Camera::look_at + Camera::sample (camera.rs:85-146) in full: every sample of pixel (x, y) for supersampling base `base` ((base + 1)^2
    samples, camera.rs:189-193), perspective (fov in degrees; pixel_separation 0: one origin) or orthographic (image plane height `ortho`;
    pixel_separation 1: the origin moves with the pixel).  [(origin, direction)] in the reference's order (i outer over aux, j inner over up)."""
    eye, look, up = (np.array(v, float) for v in (eye, look, up))
    view = look - eye
    aux = np.cross(view, up)
    upn = np.cross(aux, view); upn = upn * (1.0 / math.sqrt(upn @ upn))
    aux = aux * (1.0 / math.sqrt(aux @ aux))
    H = ortho if ortho is not None else math.sqrt(view @ view) * math.tan(fov * math.pi / 360.0) * 2.0
    sep = 1.0 if ortho is not None else 0.0
    W = H * (w / h)
    pixel = H * (1.0 / h)
    root = base + 1
    ssep = (1.0 / root) * pixel
    sox = (x * (1.0 / w) - 0.5) * W
    soy = (0.5 - (y + 1) * (1.0 / h)) * H
    origin = eye + (soy * sep) * upn + (sox * sep) * aux
    d = view + soy * upn + sox * aux
    updiff, auxdiff = upn * ssep, aux * ssep
    half = updiff * 0.5 + auxdiff * 0.5
    return [(origin, d + j * updiff + i * auxdiff + half) for i in range(root) for j in range(root)]


def check_supersampled_background(api, radiance):
    """3 x 3 samples per pixel (set_supersampling(2)): the pixel is the mean of the nine samples' background values, summed in the reference's
    order and multiplied by the weight 1 / 9 (lib.rs:147-159, integrate.rs:16-20)."""
    w, h, fov = 24, 18, 55.0
    eye, look = [1.0, 2.0, 3.0], [4.0, 1.0, -2.0]
    inner, outer, scale = [0.26, 0.78, 0.67], [0.1, 0.09, 0.33], 0.5
    sc = api.Scene.new()
    cam = sc.set_perspective_camera(fov)
    cam.look_at(eye, look, [0.0, 1.0, 0.0])
    cam.set_supersampling(2)
    sc.set_radial_background(inner, outer, scale)
    sc.root.add_sphere([0.0, 1.0e6, 0.0], 1.0, api.Material.matte([0.5, 0.5, 0.5], 0.0))
    rad = radiance(api.Accel(sc), w, h)
    for x, y in ((0, 0), (w - 1, h - 1), (w // 2, h // 2), (3, 14)):
        rays = camera_rays(eye, look, [0.0, 1.0, 0.0], w, h, x, y, base=2, fov=fov)
        assert len(rays) == 9
        total = np.zeros(3)
        for _, d in rays:
            total = total + background(d, inner, outer, scale)
        assert np.allclose(rad[y, x], total * (1.0 / 9.0), rtol=1e-13, atol=0.0), (x, y)


def check_orthographic_lambert(api, radiance):
    """The orthographic camera (camera.rs:139-146 with pixel_separation 1: the ray's origin moves across the image plane with its pixel, and the
    direction keeps the pixel's offset as well -- the reference's formula as written) on a Lambertian sphere."""
    w, h, height = 33, 33, 6.0
    eye, look = [0.0, 0.0, 0.0], [0.0, 0.0, 2.0]
    kd, ambient = np.array([0.7, 1.0, 0.7]), np.array([0.1, 0.2, 0.3])
    lpos, lint, falloff = np.array([3.0, 6.0, -4.0]), np.array([0.8, 0.6, 0.9]), np.array([1.0, 0.001, 0.0])
    c, r = np.array([0.2, -0.1, 10.0]), 2.5
    sc = api.Scene.new()
    sc.set_orthographic_camera(height).look_at(eye, look, [0.0, 1.0, 0.0])
    sc.set_ambient_light(ambient.tolist())
    sc.add_point_light(lpos.tolist(), lint.tolist(), falloff.tolist())
    sc.root.add_sphere(c.tolist(), r, api.Material.matte(kd.tolist(), 0.0))
    rad = radiance(api.Accel(sc), w, h)
    checked = 0
    for x, y in ((16, 16), (15, 16), (17, 17), (16, 15), (17, 15), (15, 18)):  # (near the centre: the direction keeps the pixel's offset too, so the rays fan out)
        (o, d), = camera_rays(eye, look, [0.0, 1.0, 0.0], w, h, x, y, base=0, ortho=height)
        t = sphere_hit(o, d, c, r)
        if t is None:
            continue
        p = o + d * t
        n = (p - c) / r
        wi = lpos - p
        dist = math.sqrt(wi @ wi)
        f_att = falloff[0] + falloff[1] * dist + falloff[2] * dist * dist
        wi = wi / dist
        if not (wi @ n) > 0.05:
            continue
        want = lint * kd * (wi @ n) / f_att + ambient * kd / math.pi
        assert np.allclose(rad[y, x], want, rtol=1e-9, atol=0.0), (x, y, rad[y, x], want)
        checked += 1
    assert checked >= 4


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


def sphere_hit(o, d, c, r):
    l = o - c
    a, b, cc = d @ d, 2.0 * (d @ l), l @ l - r * r
    disc = b * b - 4.0 * a * cc
    if disc < 0.0:
        return None
    return (-b - math.sqrt(disc)) / (2.0 * a)


def check_mirror_sphere(api, radiance):
    """A mirror sphere in front of the radial background, no light, no ambient, recursion 1: li = clamp(kr / |cos theta|, 0, 1) * bg(reflect(wo, n)) --
    Reflection::sample_f (core/bxdf/specular.rs:17-24: the NoOp substance returns 1, the spectrum is divided by |cos theta_i|) and
    specular_reflect (integrate.rs:84-108), which traces reflect(wo, ns) and multiplies the spectrum in WITHOUT the |cos| / pdf factor."""
    w, h, fov = 41, 31, 40.0
    eye, look = [0.3, -0.2, 0.0], [0.1, 0.0, 1.0]
    kr = np.array([0.9, 0.6, 0.8])
    inner, outer, scale = [0.26, 0.78, 0.67], [0.1, 0.09, 0.33], 0.5
    c, r = np.array([-1.3, 1.4, 8.0]), 2.5  # on the view axis (eye + 8 (look - eye))
    sc = api.Scene.new()
    sc.set_perspective_camera(fov).look_at(eye, look, [0.0, 1.0, 0.0])
    sc.set_radial_background(inner, outer, scale)
    sc.set_max_recursion_depth(1)
    sc.root.add_sphere(c.tolist(), r, api.Material.mirror(kr.tolist()))
    rad = radiance(api.Accel(sc), w, h)
    seen = 0
    for x, y in ((20, 15), (17, 12), (24, 18), (22, 10), (15, 17)):
        o, d = camera_ray(eye, look, [0.0, 1.0, 0.0], fov, w, h, x, y)
        t = sphere_hit(o, d, c, r)
        if t is None:
            continue
        seen += 1
        n = (o + d * t - c) / r
        wo = -d / math.sqrt(d @ d)
        cos = wo @ n
        wr = -wo + 2.0 * cos * n
        want = np.clip(kr / abs(cos), 0.0, 1.0) * background(wr, inner, outer, scale)  # BSDF::sample_f clamps the spectrum to [0, 1] (interaction/bsdf.rs:133)
        assert np.allclose(rad[y, x], want, rtol=1e-9, atol=0.0), (x, y, rad[y, x], want)
    assert seen >= 3


def fresnel_unpolarised(cos_i, n1, n2):
    """Fresnel reflectance of a dielectric interface for unpolarised light (Snell + the two amplitude ratios), cos_i > 0: n1 -> n2."""
    sin_t = n1 / n2 * math.sqrt(max(0.0, 1.0 - cos_i * cos_i))
    if sin_t >= 1.0:
        return 1.0
    cos_t = math.sqrt(max(0.0, 1.0 - sin_t * sin_t))
    rs = (n1 * cos_i - n2 * cos_t) / (n1 * cos_i + n2 * cos_t)
    rp = (n2 * cos_i - n1 * cos_t) / (n2 * cos_i + n1 * cos_t)
    return 0.5 * (rs * rs + rp * rp)


def check_glass_sphere_reflection(api, radiance):
    """A glass sphere (eta 1.5) in front of the radial background, no light, no ambient, recursion 1: the reflected ray sees the background,
    weighted by the Fresnel reflectance of the interface (core/bxdf/fresnel.rs:37-64) times kr over |cos theta|, clamped to [0, 1]
    (interaction/bsdf.rs:133); the refracted ray ends on the sphere's inside at depth 1, where nothing is lit and nothing is traced on
    (integrate.rs:69-77) -- it contributes exactly 0."""
    w, h, fov = 41, 31, 40.0
    eye, look = [0.3, -0.2, 0.0], [0.1, 0.0, 1.0]
    kr, kt, eta = np.array([1.0, 0.7, 0.9]), [0.8, 1.0, 0.6], 1.5
    inner, outer, scale = [0.26, 0.78, 0.67], [0.1, 0.09, 0.33], 0.5
    c, r = np.array([-1.3, 1.4, 8.0]), 2.5
    sc = api.Scene.new()
    sc.set_perspective_camera(fov).look_at(eye, look, [0.0, 1.0, 0.0])
    sc.set_radial_background(inner, outer, scale)
    sc.set_max_recursion_depth(1)
    sc.root.add_sphere(c.tolist(), r, api.Material.glass(kr.tolist(), kt, eta))
    rad = radiance(api.Accel(sc), w, h)
    seen = 0
    for x, y in ((20, 15), (17, 12), (24, 18), (22, 10), (15, 17), (27, 15), (13, 14)):
        o, d = camera_ray(eye, look, [0.0, 1.0, 0.0], fov, w, h, x, y)
        t = sphere_hit(o, d, c, r)
        if t is None:
            continue
        seen += 1
        n = (o + d * t - c) / r
        wo = -d / math.sqrt(d @ d)
        cos = wo @ n
        wr = -wo + 2.0 * cos * n
        want = np.clip(fresnel_unpolarised(cos, 1.0, eta) * kr / abs(cos), 0.0, 1.0) * background(wr, inner, outer, scale)
        assert np.allclose(rad[y, x], want, rtol=1e-9, atol=1e-15), (x, y, rad[y, x], want)
    assert seen >= 4


def plastic_f(wo, wi, n, kd, ks, alpha):
    """The plastic BSDF of the headline's every surface, written from the published formulas (Walter et al. 2007 / pbrt): a Lambertian lobe
    kd / pi plus the Torrance-Sparrow microfacet lobe ks D G F / (4 cos_i cos_o) with the isotropic Trowbridge-Reitz (GGX) distribution of
    width alpha, Smith's G = 1 / (1 + Lambda_o + Lambda_i) and the Fresnel reflectance of an air -> 1.5 interface at the half vector.
    Reference: material/plastic.rs:20-37 (roughness IS alpha), core/bxdf/microfacet.rs:31-66, 101-115."""
    ci, co = wi @ n, wo @ n
    if not (ci > 0.0 and co > 0.0):
        return None
    wh = wi + wo
    wh = wh / math.sqrt(wh @ wh)
    ch = wh @ n

    def tan2(c):
        return (1.0 - c * c) / (c * c)

    def lam(c):
        return (math.sqrt(1.0 + alpha * alpha * tan2(c)) - 1.0) / 2.0

    D = 1.0 / (math.pi * alpha * alpha * ch ** 4 * (1.0 + tan2(ch) / (alpha * alpha)) ** 2)
    G = 1.0 / (1.0 + lam(co) + lam(ci))
    F = fresnel_unpolarised(wi @ wh, 1.0, 1.5)
    return kd / math.pi + ks * (D * G * F / (4.0 * ci * co))


def check_plastic_sphere(api, radiance):
    """The headline's material on a sphere under one point light with ambient light: li = pi I f(wo, wi) (wi . n) / f_att + ambient f(wo, n)
    (integrate.rs:47-67)."""
    w, h, fov = 33, 33, 45.0
    eye, look = [0.0, 0.0, 0.0], [0.0, 0.0, 1.0]
    kd, ks, alpha = np.array([0.9, 0.2, 0.2]), np.array([0.5, 0.7, 0.5]), 0.25
    ambient = np.array([0.2, 0.2, 0.2])
    lpos, lint, falloff = np.array([4.0, 7.0, -2.0]), np.array([0.9, 0.9, 0.9]), np.array([1.0, 0.0, 0.0])
    c, r = np.array([0.3, -0.2, 8.0]), 2.5
    sc = api.Scene.new()
    sc.set_perspective_camera(fov).look_at(eye, look, [0.0, 1.0, 0.0])
    sc.set_ambient_light(ambient.tolist())
    sc.add_point_light(lpos.tolist(), lint.tolist(), falloff.tolist())
    sc.root.add_sphere(c.tolist(), r, api.Material.plastic(kd.tolist(), ks.tolist(), alpha))
    rad = radiance(api.Accel(sc), w, h)
    checked = 0
    for x, y in ((16, 16), (19, 12), (13, 18), (21, 15), (17, 10), (12, 13)):
        o, d = camera_ray(eye, look, [0.0, 1.0, 0.0], fov, w, h, x, y)
        t = sphere_hit(o, d, c, r)
        if t is None:
            continue
        p = o + d * t
        n = (p - c) / r
        wo = -d / math.sqrt(d @ d)
        wi = lpos - p
        dist = math.sqrt(wi @ wi)
        f_att = falloff[0] + falloff[1] * dist + falloff[2] * dist * dist
        wi = wi / dist
        f_light, f_amb = plastic_f(wo, wi, n, kd, ks, alpha), plastic_f(wo, n, n, kd, ks, alpha)
        if f_light is None or f_amb is None:
            continue
        want = math.pi * lint * f_light * (wi @ n) / f_att + ambient * f_amb
        assert np.allclose(rad[y, x], want, rtol=1e-9, atol=0.0), (x, y, rad[y, x], want)
        checked += 1
    assert checked >= 3


def ggx_d_g(wo, wi, n, alpha):
    wh = wi + wo
    wh = wh / math.sqrt(wh @ wh)
    ch, ci, co = wh @ n, wi @ n, wo @ n
    tan2 = lambda c: (1.0 - c * c) / (c * c)  # noqa: E731
    lam = lambda c: (math.sqrt(1.0 + alpha * alpha * tan2(c)) - 1.0) / 2.0  # noqa: E731
    D = 1.0 / (math.pi * alpha * alpha * ch ** 4 * (1.0 + tan2(ch) / (alpha * alpha)) ** 2)
    return wh, D, 1.0 / (1.0 + lam(co) + lam(ci))


def fresnel_conductor_complex(cos_i, eta, k):
    """Fresnel reflectance of a conductor from its COMPLEX refractive index n = eta + i k (air outside), unpolarised: the textbook amplitude
    ratios evaluated in complex arithmetic -- the reference (core/bxdf/fresnel.rs:69-92, after pbrt) uses an equivalent real-valued form."""
    n2 = complex(eta, k) ** 2
    s2 = 1.0 - cos_i * cos_i
    root = (n2 - s2) ** 0.5
    rs = (cos_i - root) / (cos_i + root)
    rp = (n2 * cos_i - root) / (n2 * cos_i + root)
    return 0.5 * (abs(rs) ** 2 + abs(rp) ** 2)


def sphere_pixels(api, radiance, material, pixels=((16, 16), (19, 12), (13, 18), (21, 15), (17, 10), (12, 13))):
    """(radiance, wo, wi, n, f_att) of sphere pixels under one point light and ambient light"""
    w, h, fov = 33, 33, 45.0
    eye, look = [0.0, 0.0, 0.0], [0.0, 0.0, 1.0]
    ambient = np.array([0.2, 0.1, 0.3])
    lpos, lint, falloff = np.array([4.0, 7.0, -2.0]), np.array([0.9, 0.8, 0.7]), np.array([1.0, 0.02, 0.0])
    c, r = np.array([0.3, -0.2, 8.0]), 2.5
    sc = api.Scene.new()
    sc.set_perspective_camera(fov).look_at(eye, look, [0.0, 1.0, 0.0])
    sc.set_ambient_light(ambient.tolist())
    sc.add_point_light(lpos.tolist(), lint.tolist(), falloff.tolist())
    sc.root.add_sphere(c.tolist(), r, material)
    rad = radiance(api.Accel(sc), w, h)
    out = []
    for x, y in pixels:
        o, d = camera_ray(eye, look, [0.0, 1.0, 0.0], fov, w, h, x, y)
        t = sphere_hit(o, d, c, r)
        if t is None:
            continue
        p = o + d * t
        n = (p - c) / r
        wo = -d / math.sqrt(d @ d)
        wi = lpos - p
        dist = math.sqrt(wi @ wi)
        f_att = falloff[0] + falloff[1] * dist + falloff[2] * dist * dist
        wi = wi / dist
        if (wi @ n) > 0.05 and (wo @ n) > 0.05:
            out.append((rad[y, x], wo, wi, n, f_att, lint, ambient))
    assert len(out) >= 3
    return out


def check_metal_sphere(api, radiance):
    """Metal (material/metal.rs:17-27: a white microfacet lobe, the given roughness as alpha, conductor Fresnel with the given eta and k) with
    the Fresnel term from complex arithmetic."""
    eta, k, alpha = np.array([0.2, 0.92, 1.1]), np.array([3.9, 2.45, 2.14]), 0.3
    for got, wo, wi, n, f_att, lint, ambient in sphere_pixels(api, radiance, api.Material.metal(eta.tolist(), k.tolist(), alpha, alpha)):
        def f(wo, wi):
            wh, D, G = ggx_d_g(wo, wi, n, alpha)
            F = np.array([fresnel_conductor_complex(wi @ wh, e, kk) for e, kk in zip(eta, k)])
            return F * (D * G / (4.0 * (wi @ n) * (wo @ n)))
        want = math.pi * lint * f(wo, wi) * (wi @ n) / f_att + ambient * f(wo, n)
        assert np.allclose(got, want, rtol=1e-9, atol=0.0), (got, want)


def check_oren_nayar_sphere(api, radiance):
    """Matte with sigma = 20 degrees: Oren and Nayar's qualitative model, f = kd / pi (A + B max(0, cos(phi_i - phi_o)) sin(alpha) tan(beta)) --
    with the reference's A AS WRITTEN, 1 - (sigma^2 / 2) (sigma^2 + 0.33) (core/bxdf/diffuse.rs:31: the published A divides by 2 (sigma^2 + 0.33);
    the product is the reference's, and is what a drop-in has to compute), B = 0.45 sigma^2 / (sigma^2 + 0.09).  The azimuth difference is
    taken from the directions' projections onto the tangent plane, not from a local frame."""
    kd, sigma_deg = np.array([0.8, 0.6, 0.4]), 20.0
    sg2 = math.radians(sigma_deg) ** 2
    A, B = 1.0 - (sg2 / 2.0 * (sg2 + 0.33)), 0.45 * sg2 / (sg2 + 0.09)
    for got, wo, wi, n, f_att, lint, ambient in sphere_pixels(api, radiance, api.Material.matte(kd.tolist(), sigma_deg)):
        def f(wo, wi):
            ci, co = wi @ n, wo @ n
            si, so = math.sqrt(max(0.0, 1.0 - ci * ci)), math.sqrt(max(0.0, 1.0 - co * co))
            max_cos = 0.0
            if si > 1e-4 and so > 1e-4:
                pi_, po_ = wi - ci * n, wo - co * n
                max_cos = max(0.0, (pi_ @ po_) / (si * so))
            sin_a, tan_b = (so, si / abs(ci)) if abs(ci) > abs(co) else (si, so / abs(co))
            return kd / math.pi * (A + B * max_cos * sin_a * tan_b)
        want = math.pi * lint * f(wo, wi) * (wi @ n) / f_att + ambient * f(wo, n)
        assert np.allclose(got, want, rtol=1e-9, atol=0.0), (got, want)


def check_glass_pane(api, radiance):
    """An OPEN glass surface -- one quad of a mesh (triangle.rs:161-307), tilted 35 degrees towards the camera -- in front of the radial
    background, recursion 1: the reflected ray and the ray refracted INTO the glass (Snell, 1 -> 1.5; it never meets a second surface) both see
    the background.  li = clamp(F kr / |cos i|) bg(reflect) + clamp((1 - F) kt / |cos t|) |cos t| bg(refract): Transmission::sample_f
    (core/bxdf/specular.rs:43-62), specular_transmit's |wi . n| / pdf (integrate.rs:110-132), both spectra clamped (interaction/bsdf.rs:133)."""
    w, h, fov = 33, 33, 40.0
    eye, look = [0.0, 0.0, 0.0], [0.0, 0.0, 1.0]
    kr, kt, eta = np.array([0.9, 0.8, 1.0]), np.array([0.7, 0.9, 0.6]), 1.5
    inner, outer, scale = [0.26, 0.78, 0.67], [0.1, 0.09, 0.33], 0.5
    th = math.radians(-55.0)  # the plane y = 0 (normal +y) turned about x: its normal ends up 35 degrees off the view axis, facing the camera
    sc = api.Scene.new()
    sc.set_perspective_camera(fov).look_at(eye, look, [0.0, 1.0, 0.0])
    sc.set_radial_background(inner, outer, scale)
    sc.set_max_recursion_depth(1)
    plane = sc.parse_obj("o plane\nv -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\n\nf 1 2 3\nf 1 3 4\n")
    g = api.Aggregate.new()
    g.scale(6.0, 1.0, 6.0)
    g.rotate_x(-55.0)
    g.translate([0.0, 0.0, 7.0])
    g.add_obj_of(plane, api.Material.glass(kr.tolist(), kt.tolist(), eta))
    sc.root.add_group(g)
    rad = radiance(api.Accel(sc), w, h)
    R = np.array([[1.0, 0.0, 0.0], [0.0, math.cos(th), -math.sin(th)], [0.0, math.sin(th), math.cos(th)]])
    n_plane = R @ np.array([0.0, 1.0, 0.0])
    p0 = np.array([0.0, 0.0, 7.0])
    checked = 0
    for x, y in ((16, 16), (12, 14), (20, 18), (15, 20), (19, 12)):
        o, d = camera_ray(eye, look, [0.0, 1.0, 0.0], fov, w, h, x, y)
        t = ((p0 - o) @ n_plane) / (d @ n_plane)
        q = R.T @ (o + d * t - p0)
        if not (t > 0.0 and abs(q[0]) < 5.9 and abs(q[2]) < 5.9 and abs(q[1]) < 1e-9):
            continue
        wo = -d / math.sqrt(d @ d)
        n = n_plane if (wo @ n_plane) > 0.0 else -n_plane  # the flat mesh's normal faces the ray's origin (triangle.rs:301-303)
        ci = wo @ n
        wr = -wo + 2.0 * ci * n
        e = 1.0 / eta
        ct = math.sqrt(1.0 - e * e * (1.0 - ci * ci))
        wt = -e * wo + (e * ci - ct) * n
        F = fresnel_unpolarised(ci, 1.0, eta)
        want = (np.clip(F * kr / abs(ci), 0.0, 1.0) * background(wr, inner, outer, scale)
                + np.clip((1.0 - F) * kt / ct, 0.0, 1.0) * ct * background(wt, inner, outer, scale))
        assert np.allclose(rad[y, x], want, rtol=1e-9, atol=0.0), (x, y, rad[y, x], want)
        checked += 1
    assert checked >= 3


def check_box_face_and_shadow(api, radiance):
    """Lambert on the top face of an axis-aligned box (cuboid.rs:55-102: the face's normal is the axis), lit by one point light -- and the same
    face with a small sphere between it and the light: where the shadow ray is blocked (point.rs:42-54) only the ambient term is left."""
    w, h, fov = 33, 33, 45.0
    eye, look = [0.0, 6.0, -8.0], [0.0, 0.0, 0.0]
    kd, ambient = np.array([0.6, 0.5, 0.9]), np.array([0.05, 0.1, 0.15])
    lpos, lint, falloff = np.array([0.0, 10.0, 0.0]), np.array([0.7, 0.8, 0.9]), np.array([1.0, 0.0, 0.01])
    for blocker in (False, True):
        sc = api.Scene.new()
        sc.set_perspective_camera(fov).look_at(eye, look, [0.0, 1.0, 0.0])
        sc.set_ambient_light(ambient.tolist())
        sc.add_point_light(lpos.tolist(), lint.tolist(), falloff.tolist())
        sc.root.add_box([-3.0, -1.0, -3.0], [3.0, 0.0, 3.0], api.Material.matte(kd.tolist(), 0.0))
        if blocker:
            sc.root.add_sphere([0.0, 5.0, 0.0], 0.5, api.Material.matte([0.1, 0.1, 0.1], 0.0))
        rad = radiance(api.Accel(sc), w, h)
        checked = 0
        for x, y in ((16, 16), (10, 20), (22, 14), (16, 22)):
            o, d = camera_ray(eye, look, [0.0, 1.0, 0.0], fov, w, h, x, y)
            t = (0.0 - o[1]) / d[1]  # the plane y = 0 of the top face
            p = o + d * t
            if not (t > 0.0 and abs(p[0]) < 2.9 and abs(p[2]) < 2.9):
                continue
            if blocker and sphere_hit(o, d, np.array([0.0, 5.0, 0.0]), 0.5) is not None:
                continue  # (the camera ray itself meets the blocker)
            n = np.array([0.0, 1.0, 0.0])
            wi = lpos - p
            dist = math.sqrt(wi @ wi)
            f_att = falloff[0] + falloff[1] * dist + falloff[2] * dist * dist
            wi = wi / dist
            lit = lint * kd * (wi @ n) / f_att
            amb = ambient * kd / math.pi
            shadowed = blocker and sphere_hit(p, lpos - p, np.array([0.0, 5.0, 0.0]), 0.5) is not None
            want = amb if shadowed else lit + amb
            assert np.allclose(rad[y, x], want, rtol=1e-9, atol=1e-15), (blocker, x, y, rad[y, x], want, shadowed)
            checked += 1
        assert checked >= 3


def check_transformed_sphere(api, radiance):
    """Lambert on a unit sphere inside a group that is scaled by (2, 1, 1.5), rotated 30 degrees about y and translated (scene/node.rs:63-115:
    later calls apply after earlier ones; bvh.rs:462 takes the ray into the group's space with the inverse, transform.rs:243-264 brings the
    tangents back with the matrix): an ellipsoid.  Closed form: intersect the unit sphere with M^-1 applied to the ray (same t), normal =
    M^-T n_local normalised."""
    w, h, fov = 33, 33, 45.0
    eye, look = [0.0, 0.0, 0.0], [0.0, 0.0, 1.0]
    kd, ambient = np.array([0.8, 0.7, 0.6]), np.array([0.02, 0.03, 0.04])
    lpos, lint, falloff = np.array([5.0, 8.0, -3.0]), np.array([0.9, 0.8, 0.7]), np.array([1.0, 0.01, 0.0])
    th = math.radians(30.0)
    S_ = np.diag([2.0, 1.0, 1.5])
    R = np.array([[math.cos(th), 0.0, math.sin(th)], [0.0, 1.0, 0.0], [-math.sin(th), 0.0, math.cos(th)]])
    T = np.array([0.5, -0.25, 9.0])
    M = R @ S_
    sc = api.Scene.new()
    sc.set_perspective_camera(fov).look_at(eye, look, [0.0, 1.0, 0.0])
    sc.set_ambient_light(ambient.tolist())
    sc.add_point_light(lpos.tolist(), lint.tolist(), falloff.tolist())
    g = api.Aggregate.new()
    g.scale(2.0, 1.0, 1.5)
    g.rotate_y(30.0)
    g.translate(T.tolist())
    g.add_sphere([0.0, 0.0, 0.0], 1.0, api.Material.matte(kd.tolist(), 0.0))
    sc.root.add_group(g)
    rad = radiance(api.Accel(sc), w, h)
    Minv = np.linalg.inv(M)
    checked = 0
    for x, y in ((16, 16), (14, 17), (19, 15), (17, 18)):
        o, d = camera_ray(eye, look, [0.0, 1.0, 0.0], fov, w, h, x, y)
        ol, dl = Minv @ (o - T), Minv @ d
        t = sphere_hit(ol, dl, np.zeros(3), 1.0)
        if t is None:
            continue
        n = np.linalg.inv(M).T @ (ol + dl * t)
        n = n / math.sqrt(n @ n)
        p = o + d * t
        wi = lpos - p
        dist = math.sqrt(wi @ wi)
        f_att = falloff[0] + falloff[1] * dist + falloff[2] * dist * dist
        wi = wi / dist
        if not (wi @ n) > 0.05:
            continue  # (a point facing away from the light: the unclamped wi . n of the reference is another test's business)
        want = lint * kd * (wi @ n) / f_att + ambient * kd / math.pi
        assert np.allclose(rad[y, x], want, rtol=1e-8, atol=0.0), (x, y, rad[y, x], want)
        checked += 1
    assert checked >= 2


def test_closed_forms_on_the_oracle():
    o = oracle()
    check_background(o, lambda acc, w, h: o.capture_radiance(acc, w, h, nthreads=2))
    check_lambert_sphere(o, lambda acc, w, h: o.capture_radiance(acc, w, h, nthreads=2))
    check_mirror_sphere(o, lambda acc, w, h: o.capture_radiance(acc, w, h, nthreads=2))
    check_box_face_and_shadow(o, lambda acc, w, h: o.capture_radiance(acc, w, h, nthreads=2))
    check_transformed_sphere(o, lambda acc, w, h: o.capture_radiance(acc, w, h, nthreads=2))
    check_glass_sphere_reflection(o, lambda acc, w, h: o.capture_radiance(acc, w, h, nthreads=2))
    check_plastic_sphere(o, lambda acc, w, h: o.capture_radiance(acc, w, h, nthreads=2))
    check_metal_sphere(o, lambda acc, w, h: o.capture_radiance(acc, w, h, nthreads=2))
    check_oren_nayar_sphere(o, lambda acc, w, h: o.capture_radiance(acc, w, h, nthreads=2))
    check_glass_pane(o, lambda acc, w, h: o.capture_radiance(acc, w, h, nthreads=2))
    check_supersampled_background(o, lambda acc, w, h: o.capture_radiance(acc, w, h, nthreads=2))
    check_orthographic_lambert(o, lambda acc, w, h: o.capture_radiance(acc, w, h, nthreads=2))


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
        check_mirror_sphere(G, radiance)
        check_box_face_and_shadow(G, radiance)
        check_transformed_sphere(G, radiance)
        check_glass_sphere_reflection(G, radiance)
        check_plastic_sphere(G, radiance)
        check_metal_sphere(G, radiance)
        check_oren_nayar_sphere(G, radiance)
        check_glass_pane(G, radiance)
        check_supersampled_background(G, radiance)
        check_orthographic_lambert(G, radiance)


def test_to_byte_is_rusts_rounding_and_saturating_cast():
    """`to_byte` (img.rs:65-67): clamp to [0, 1], times 255, `f64::round` (half away from zero), `as u8` (saturating; NaN -> 0) -- written here
    from Rust's documented semantics with exact rational arithmetic, against the oracle's (the device is held to the oracle's in
    tests/test_gpu_parity.py::test_to_byte_quantisation)."""
    from fractions import Fraction
    rng = np.random.default_rng(3)
    k = np.arange(0, 256)
    a = np.concatenate([rng.uniform(-0.5, 1.5, 20000), (k + 0.5) / 255.0, np.nextafter((k + 0.5) / 255.0, 0), np.nextafter((k + 0.5) / 255.0, 2),
                        k / 255.0, [np.nan, np.inf, -np.inf, -0.0, 1.0, 0.0, 5e-324, 1.0 - 2.0 ** -53]])

    def rust(c):
        if c != c:
            c = 0.0  # f64::max / min return the other operand for a NaN: NaN.max(0.0) = 0.0
        c = min(max(c, 0.0), 1.0)
        v = c * 255.0  # one f64 multiplication, rounded
        f = Fraction(v)
        n = int(f)  # truncation; v >= 0
        return min(255, n + 1 if f - n >= Fraction(1, 2) else n)
    got = oracle().math_eval(8, a)
    want = np.array([rust(float(c)) for c in a], dtype=np.float64)
    assert np.array_equal(got, want)
