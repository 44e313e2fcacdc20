"""The inequalities behind the pruned walk, checked numerically: the reference's intersection formulas
are evaluated in f64 exactly as the reference orders them, the geometry they are measured against in exact rational
arithmetic (fractions.Fraction), on random and on deliberately ill-conditioned inputs -- tangent rays, origins on the surface,
far origins, needle and nearly edge-on triangles.  CPU only; no oracle, no device: this pins the DERIVATION, the constants
shipped in lasgun_amd/csrc/dscene.h are then compared with what was observed."""
import math
import random
from fractions import Fraction as Fr

U = 2.0 ** -53


def fsignum(x):
    return math.copysign(1.0, x)


def sphere_t(o, d, c, r):
    """Sphere::intersect_t + quad_roots (sphere.rs:30-69, core/math.rs:7-31): every accepted candidate t, f64, reference order."""
    l = [o[i] - c[i] for i in range(3)]
    a = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]
    b = 2.0 * ((d[0] * l[0] + d[1] * l[1]) + d[2] * l[2])
    cc = ((l[0] * l[0] + l[1] * l[1]) + l[2] * l[2]) - r * r
    disc = b * b - (4.0 * a) * cc
    if disc < 0.0:
        return None
    q = -(b + fsignum(b) * math.sqrt(disc)) / 2.0
    r0 = q / a
    r1 = r0 if q == 0.0 else cc / q
    t0, t1 = min(r0, r1), max(r0, r1)
    t = t1 if t0 < 0.0 else t0
    return None if t < 0.0 else t


def test_sphere_root_residual_is_linear_in_u():
    """||x(t) - c|^2 - r^2| <= 114 u W^2 for the accepted root, W^2 = |o - c|^2 + r^2 -- tangent rays included, where the root
    itself is only good to sqrt(u)."""
    rng = random.Random(7)
    worst = 0.0
    n = 0
    for case in range(6000):
        c = [rng.uniform(-3, 3) for _ in range(3)]
        r = 10.0 ** rng.uniform(-3, 1)
        kind = case % 4
        if kind == 0:      # generic
            o = [rng.uniform(-8, 8) for _ in range(3)]
            d = [rng.uniform(-1, 1) for _ in range(3)]
        elif kind == 1:    # aimed at a point close to the silhouette: nearly tangent
            o = [c[i] + rng.uniform(5, 300) * rng.choice((-1, 1)) for i in range(3)]
            u_ = [rng.gauss(0, 1) for _ in range(3)]
            to_c = [c[i] - o[i] for i in range(3)]
            # a direction perpendicular to (c - o), scaled to radius r (1 -+ tiny): target on the silhouette circle
            dot = sum(u_[i] * to_c[i] for i in range(3)) / sum(v * v for v in to_c)
            perp = [u_[i] - dot * to_c[i] for i in range(3)]
            pl = math.sqrt(sum(v * v for v in perp))
            k = r * (1.0 + rng.choice((-1, 1)) * 10.0 ** rng.uniform(-16, -3)) / pl
            d = [to_c[i] + perp[i] * k for i in range(3)]
        elif kind == 2:    # origin on the surface (a secondary ray leaving the sphere)
            n_ = [rng.gauss(0, 1) for _ in range(3)]
            nl = math.sqrt(sum(v * v for v in n_))
            o = [c[i] + n_[i] / nl * r for i in range(3)]
            d = [rng.uniform(-1, 1) for _ in range(3)]
        else:              # far origin, tiny sphere
            o = [c[i] + rng.uniform(100, 1000) * rng.choice((-1, 1)) for i in range(3)]
            r = 10.0 ** rng.uniform(-3, -1)
            d = [c[i] - o[i] + rng.uniform(-1, 1) * r for i in range(3)]
        t = sphere_t(o, d, c, r)
        if t is None:
            continue
        n += 1
        x = [Fr(o[i]) + Fr(t) * Fr(d[i]) for i in range(3)]
        g = sum((x[i] - Fr(c[i])) ** 2 for i in range(3)) - Fr(r) ** 2
        w2 = sum((Fr(o[i]) - Fr(c[i])) ** 2 for i in range(3)) + Fr(r) ** 2
        ratio = float(abs(g) / (Fr(U) * w2))
        worst = max(worst, ratio)
    assert n > 3000
    assert worst <= 114.0, worst          # the derivation's bound
    assert worst * 8 <= 2048.0            # shipped: e2 = 2^-42 / r_min = 2048 u / r_min (dscene.h), a factor >= 8 above what occurs
    print("sphere residual: worst %.1f u W^2 over %d accepted roots" % (worst, n))


def tri_t(p, o, d):
    """Triangle::intersect up to t (triangle.rs:161-251), f64, reference order; returns (t, kz) or None."""
    ad = [abs(v) for v in d]
    kz = 0 if (ad[0] > ad[1] and ad[0] > ad[2]) else (1 if ad[1] > ad[2] else 2)
    kx = (kz + 1) % 3
    ky = (kx + 1) % 3
    dx, dy, dz = d[kx], d[ky], d[kz]
    pt = [[p[i][kx] - o[kx], p[i][ky] - o[ky], p[i][kz] - o[kz]] for i in range(3)]
    sx, sy, sz = -dx / dz, -dy / dz, 1.0 / dz
    for v in pt:
        v[0] += sx * v[2]
        v[1] += sy * v[2]
    e0 = pt[1][0] * pt[2][1] - pt[1][1] * pt[2][0]
    e1 = pt[2][0] * pt[0][1] - pt[2][1] * pt[0][0]
    e2 = pt[0][0] * pt[1][1] - pt[0][1] * pt[1][0]
    if (e0 < 0.0 or e1 < 0.0 or e2 < 0.0) and (e0 > 0.0 or e1 > 0.0 or e2 > 0.0):
        return None
    det = e0 + e1 + e2
    if det == 0.0:
        return None
    z = [v[2] * sz for v in pt]
    ts = e0 * z[0] + e1 * z[1] + e2 * z[2]
    if (det < 0.0 and ts >= 0.0) or (det > 0.0 and ts <= 0.0):
        return None
    return ts * (1.0 / det), kz, z


def outside_triangle(p, x):
    """How far the point x (taken in the triangle's plane) lies outside the triangle: the largest signed distance beyond an edge line (0 inside)."""
    a = [p[1][i] - p[0][i] for i in range(3)]
    b = [p[2][i] - p[0][i] for i in range(3)]
    n = [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]
    worst = 0.0
    for k in range(3):
        q0, q1, q2 = p[k], p[(k + 1) % 3], p[(k + 2) % 3]
        e = [q1[i] - q0[i] for i in range(3)]
        inward = [n[1] * e[2] - n[2] * e[1], n[2] * e[0] - n[0] * e[2], n[0] * e[1] - n[1] * e[0]]  # n x e: points to the triangle's side of the edge
        il = math.sqrt(sum(v * v for v in inward))
        if il == 0.0:
            continue
        if sum(inward[i] * (q2[i] - q0[i]) for i in range(3)) < 0.0:
            inward = [-v for v in inward]
        worst = max(worst, -sum(inward[i] * (x[i] - q0[i]) for i in range(3)) / il)
    return worst


def f32(x):
    import struct
    return struct.unpack("f", struct.pack("f", x))[0]


def test_triangle_t_lies_between_the_vertices_plane_parameters_on_the_dominant_axis():
    """The dominant-axis rule: an accepted t is within 8 u max|z_i| of [min z_i, max z_i], z_i = fl(fl(p_i,kz - o_kz) * fl(1 / d_kz))
    -- the slab test's own expression -- however degenerate or edge-on the triangle."""
    rng = random.Random(11)
    n = worst = 0
    for case in range(40000):
        base = [rng.uniform(-2, 2) for _ in range(3)]
        kind = case % 3
        if kind == 0:
            p = [[f32(base[i] + rng.uniform(-1, 1)) for i in range(3)] for _ in range(3)]
        elif kind == 1:  # needle
            e = [rng.uniform(-1, 1) for _ in range(3)]
            p = [[f32(base[i]) for i in range(3)], [f32(base[i] + e[i]) for i in range(3)],
                 [f32(base[i] + 0.5 * e[i] + rng.uniform(-1, 1) * 10.0 ** rng.uniform(-7, -2)) for i in range(3)]]
        else:            # axis-aligned, seen exactly edge-on below
            p = [[f32(base[0]), f32(base[1]), f32(base[2])], [f32(base[0] + 1.0), f32(base[1]), f32(base[2])], [f32(base[0]), f32(base[1] + 1.0), f32(base[2])]]
        o = [rng.uniform(-6, 6) for _ in range(3)]
        if kind == 2 and rng.random() < 0.7:   # a ray IN the triangle's plane z = base[2]
            o[2] = p[0][2]
            d = [rng.uniform(-1, 1), rng.uniform(-1, 1), 0.0]
        else:
            tgt = [sum(p[k][i] for k in range(3)) / 3.0 + rng.uniform(-0.6, 0.6) for i in range(3)]
            d = [tgt[i] - o[i] for i in range(3)]
        if max(abs(v) for v in d) == 0.0:
            continue
        r = tri_t(p, o, d)
        if r is None:
            continue
        t, kz, z = r
        n += 1
        zl, zh, zm = min(z), max(z), max(abs(v) for v in z)
        if zm > 0.0:
            worst = max(worst, (zl - t) / (U * zm), (t - zh) / (U * zm))
        assert zl - 8.0 * U * zm <= t <= zh + 8.0 * U * zm, (p, o, d, t, z)
    assert n > 3000
    print("triangle t vs its vertices' plane parameters: worst %.2f u max|z| outside over %d accepted hits" % (worst, n))


def test_triangle_hit_point_stays_near_the_triangle_unless_edge_on():
    """The lateral rule: with sigma = |n . d| / |d| > 0, an accepted hit point o + t d lies within m = 4608 u / sigma^3 * R^2 * l^2 / h^3
    of the triangle's bounds on every axis (R: 1-norm distance from the origin to the farthest vertex, l the longest edge, h the
    smallest altitude).  Edge-on rays (sigma -> 0) are what the bound gives up on -- and the last loop shows why: there the
    reference accepts hit points far outside the triangle's bounds."""
    rng = random.Random(23)
    n = 0
    worst = 0.0
    far_outside = 0
    for case in range(60000):
        base = [rng.uniform(-2, 2) for _ in range(3)]
        e1 = [rng.uniform(-1, 1) for _ in range(3)]
        e2 = [rng.uniform(-1, 1) * (10.0 ** rng.uniform(-3, 0)) for _ in range(3)]
        p = [[f32(base[i]) for i in range(3)], [f32(base[i] + e1[i]) for i in range(3)], [f32(base[i] + e2[i]) for i in range(3)]]
        a = [p[1][i] - p[0][i] for i in range(3)]
        b = [p[2][i] - p[0][i] for i in range(3)]
        nrm = [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]
        area2 = math.sqrt(sum(v * v for v in nrm))
        if area2 == 0.0:
            continue
        edges = [a, b, [p[2][i] - p[1][i] for i in range(3)]]
        lmax = max(math.sqrt(sum(v * v for v in e)) for e in edges)
        h = area2 / lmax
        # aim at a point on or just outside an edge of the triangle, from a random origin: the rays that decide the margin
        w = [rng.uniform(0, 1) for _ in range(3)]
        if rng.random() < 0.7:
            w[rng.randrange(3)] = rng.choice((0.0, 1e-12, -1e-12, 1e-9, -1e-9, -1e-6))
        ws = sum(w)
        if ws == 0.0:
            continue
        tgt = [sum(w[k] * p[k][i] for k in range(3)) / ws for i in range(3)]
        o = [tgt[i] + rng.uniform(-5, 5) for i in range(3)]
        d = [tgt[i] - o[i] for i in range(3)]
        dl = math.sqrt(sum(v * v for v in d))
        if dl == 0.0:
            continue
        sigma = abs(sum(nrm[i] * d[i] for i in range(3))) / (area2 * dl)
        r = tri_t(p, o, d)
        if r is None:
            continue
        t = r[0]
        x = [o[i] + t * d[i] for i in range(3)]
        out = max(outside_triangle(p, x), max(max(min(p[k][i] for k in range(3)) - x[i], x[i] - max(p[k][i] for k in range(3)), 0.0) for i in range(3)))
        R = max(sum(abs(p[k][i] - o[i]) for i in range(3)) for k in range(3))
        if sigma >= 1e-3 and h * h * sigma >= 2.0 ** -40 * R * R:
            n += 1
            m = 4608.0 * U / sigma ** 3 * R * R * lmax * lmax / h ** 3
            worst = max(worst, out / m)
            assert out <= m, (p, o, d, t, out, m, sigma)
    assert n > 20000
    # grazing rays (sigma between 1e-3 and 0.05) aimed just beside the triangle, at distances around the margin
    grazing = 0
    for case in range(60000):
        base = [rng.uniform(-1, 1) for _ in range(3)]
        p = [[f32(base[i]) for i in range(3)], [f32(base[i] + rng.uniform(-1, 1)) for i in range(3)], [f32(base[i] + rng.uniform(-1, 1)) for i in range(3)]]
        a = [p[1][i] - p[0][i] for i in range(3)]
        b = [p[2][i] - p[0][i] for i in range(3)]
        nrm = [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]
        area2 = math.sqrt(sum(v * v for v in nrm))
        if area2 < 1e-3:
            continue
        edges = [a, b, [p[2][i] - p[1][i] for i in range(3)]]
        lmax = max(math.sqrt(sum(v * v for v in e)) for e in edges)
        h = area2 / lmax
        nu = [v / area2 for v in nrm]
        ga, gb = rng.uniform(-1, 1), rng.uniform(-1, 1)
        inplane = [ga * a[i] + gb * b[i] for i in range(3)]
        il = math.sqrt(sum(v * v for v in inplane)) or 1.0
        sig = 10.0 ** rng.uniform(-3, -1.3)
        d = [inplane[i] / il + sig * nu[i] for i in range(3)]
        w0 = rng.uniform(0, 1)
        on_edge = [p[0][i] + w0 * a[i] for i in range(3)]                    # a point of edge p0-p1 ...
        out_dir = [a[1] * nu[2] - a[2] * nu[1], a[2] * nu[0] - a[0] * nu[2], a[0] * nu[1] - a[1] * nu[0]]
        ol = math.sqrt(sum(v * v for v in out_dir)) or 1.0
        sgn = 1.0 if sum(out_dir[i] * b[i] for i in range(3)) < 0.0 else -1.0   # ... pushed away from the third vertex
        delta = 10.0 ** rng.uniform(-15, -6)
        tgt = [on_edge[i] + sgn * out_dir[i] / ol * delta for i in range(3)]
        back = rng.uniform(0.5, 6.0)
        o = [tgt[i] - back * d[i] for i in range(3)]
        dl = math.sqrt(sum(v * v for v in d))
        sigma = abs(sum(nrm[i] * d[i] for i in range(3))) / (area2 * dl)
        r = tri_t(p, o, d)
        if r is None:
            continue
        x = [o[i] + r[0] * d[i] for i in range(3)]
        out = max(outside_triangle(p, x), max(max(min(p[k][i] for k in range(3)) - x[i], x[i] - max(p[k][i] for k in range(3)), 0.0) for i in range(3)))
        R = max(sum(abs(p[k][i] - o[i]) for i in range(3)) for k in range(3))
        if sigma >= 1e-3 and h * h * sigma >= 2.0 ** -40 * R * R:
            grazing += 1
            m = 4608.0 * U / sigma ** 3 * R * R * lmax * lmax / h ** 3
            worst = max(worst, out / m)
            assert out <= m, (p, o, d, r[0], out, m, sigma)
    assert grazing > 2000
    # ... and edge-on: rays lying (to rounding) in the plane of a tilted triangle, passing BESIDE it
    for case in range(20000):
        base = [rng.uniform(-1, 1) for _ in range(3)]
        a = [rng.uniform(-1, 1) for _ in range(3)]
        b = [rng.uniform(-1, 1) for _ in range(3)]
        p = [[f32(base[i]) for i in range(3)], [f32(base[i] + a[i]) for i in range(3)], [f32(base[i] + b[i]) for i in range(3)]]
        a = [p[1][i] - p[0][i] for i in range(3)]
        b = [p[2][i] - p[0][i] for i in range(3)]
        al, be = rng.uniform(2, 4), rng.uniform(-3, 3)          # a point of the plane well outside the triangle ...
        o = [p[0][i] + al * a[i] + be * b[i] for i in range(3)]
        ga, gb = rng.uniform(-1, 1), rng.uniform(-1, 1)         # ... and a direction within the plane that does not point at it
        d = [ga * a[i] + gb * b[i] for i in range(3)]
        if max(abs(v) for v in d) == 0.0:
            continue
        r = tri_t(p, o, d)
        if r is None:
            continue
        x = [o[i] + r[0] * d[i] for i in range(3)]
        lo = [min(p[k][i] for k in range(3)) for i in range(3)]
        hi = [max(p[k][i] for k in range(3)) for i in range(3)]
        ext = max(hi[i] - lo[i] for i in range(3))
        if max(max(lo[i] - x[i], x[i] - hi[i]) for i in range(3)) > 0.25 * ext:
            far_outside += 1
    assert far_outside > 0, "an in-plane ray is accepted although it passes the triangle at a distance: the reason the lateral rule needs sigma > 0"
    print("triangle hit point vs bounds: worst %.3g of the margin over %d + %d grazing hits; %d in-plane rays accepted a quarter of its size or more beside the triangle" % (worst, n, grazing, far_outside))
