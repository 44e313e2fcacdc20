"""The BVH half of the independent witness (tests/pyref.py is the rest) -- TEST INFRASTRUCTURE.

Plain-Python restatement of BVHAccel::new / build / emit_lbvh / build_upper_sah / flatten_bvh_tree and of
BVHAccel::intersect (accelerators/bvh.rs:164-522, 525-635; space/bounds.rs), written from the Rust source: primitive
bounds, Morton codes from (z, y, z), the 5-pass radix sort, LBVH treelets on the top 12 code bits, the upper tree by binned
SAH with the `partition` crate's two-pointer in-place partition (third-party, restated as documented in SURVEY.md 8c), the
depth-first flattening -- and the traversal that walks it (near child by dir_is_neg[axis], a 64-entry stack, leaf primitives in
order[], nested accels in place).  With it the witness renders through the reference's own visit order, so scenes with exact
ties in t and rays that graze a box are inside its reach too.
"""
import sys

import pyref
from pyref import INF, add, cross, dot, neg, sub, transform_normal, transform_point, transform_vector, _div

F64_MAX = sys.float_info.max
NBUCKETS = 12


# ---- Bounds3 (space/bounds.rs) -----------------------------------------------------------------------
def _min(a, b): return a if a < b else b   # bounds.rs:186-194 (NOT f64::min)
def _max(a, b): return b if a < b else a
def b_none(): return ((F64_MAX,) * 3, (-F64_MAX,) * 3)
def b_new(p0, p1): return (tuple(_min(p0[i], p1[i]) for i in range(3)), tuple(_max(p0[i], p1[i]) for i in range(3)))
def b_union(a, b): return (tuple(_min(a[0][i], b[0][i]) for i in range(3)), tuple(_max(a[1][i], b[1][i]) for i in range(3)))
def b_point_union(a, p): return (tuple(_min(a[0][i], p[i]) for i in range(3)), tuple(_max(a[1][i], p[i]) for i in range(3)))


def b_surface_area(b):
    d = sub(b[1], b[0])
    half = d[0] * d[1] + d[0] * d[2] + d[1] * d[2]
    return half + half


def b_maximum_extent(b):  # bounds.rs:125-130: `d.z > d.z` -- never 0
    d = sub(b[1], b[0])
    if d[0] > d[1] and d[2] > d[2]:
        return 0
    return 1 if d[1] > d[2] else 2


def b_offset(b, p):
    o = list(sub(p, b[0]))
    for i in range(3):
        if b[1][i] > b[0][i]:
            o[i] = o[i] / (b[1][i] - b[0][i])
    return tuple(o)


def b_transform(m, b):  # transform_bounds, transform.rs:219-240
    cols = [m[0][:3], m[1][:3], m[2][:3]]
    lo = [tuple(c[r] * b[0][k] for r in range(3)) for k, c in enumerate(cols)]
    hi = [tuple(c[r] * b[1][k] for r in range(3)) for k, c in enumerate(cols)]
    zmin = lambda a, c: tuple(_min(a[i], c[i]) for i in range(3))
    zmax = lambda a, c: tuple(_max(a[i], c[i]) for i in range(3))
    mn = add(add(zmin(lo[0], hi[0]), zmin(lo[1], hi[1])), zmin(lo[2], hi[2]))
    mx = add(add(zmax(lo[0], hi[0]), zmax(lo[1], hi[1])), zmax(lo[2], hi[2]))
    w = m[3]
    return b_new((mn[0] + w[0], mn[1] + w[1], mn[2] + w[2]), (mx[0] + w[0], mx[1] + w[1], mx[2] + w[2]))


def slab_intersects(b, o, dinv):  # Bounds::intersects, cuboid.rs:104-121
    tnear, tfar = -INF, INF
    for i in range(3):
        t1 = (b[0][i] - o[i]) * dinv[i]
        t2 = (b[1][i] - o[i]) * dinv[i]
        tnear = pyref.fmax(tnear, pyref.fmin(t1, t2))
        tfar = pyref.fmin(tfar, pyref.fmax(t1, t2))
    return tnear <= tfar and tfar > 0.0


# ---- Morton codes and the radix sort (bvh.rs:575-635) -------------------------------------------------
def as_u32(x):  # `as u32`: saturating, NaN -> 0
    if x != x or x <= 0.0:
        return 0
    return 0xFFFFFFFF if x >= 4294967296.0 else int(x)


def left_shift_3(x):
    if x == (1 << 10):
        x -= 1
    x = (x | (x << 16)) & 0b00000011000000000000000011111111
    x = (x | (x << 8)) & 0b00000011000000001111000000001111
    x = (x | (x << 4)) & 0b00000011000011000011000011000011
    x = (x | (x << 2)) & 0b00001001001001001001001001001001
    return x & 0xFFFFFFFF


def encode_morton_3(v):  # (z, y, z): x never contributes
    return ((left_shift_3(as_u32(v[2])) << 2) | (left_shift_3(as_u32(v[1])) << 1) | left_shift_3(as_u32(v[2]))) & 0xFFFFFFFF


def radix_sort(v):
    temp = [None] * len(v)
    for p in range(5):
        lowbit = p * 6
        src, dst = (v, temp) if p & 1 == 0 else (temp, v)
        count = [0] * 64
        for mp in src:
            count[(mp[1] >> lowbit) & 63] += 1
        out = [0] * 64
        for i in range(1, 64):
            out[i] = out[i - 1] + count[i - 1]
        for mp in src:
            k = (mp[1] >> lowbit) & 63
            dst[out[k]] = mp
            out[k] += 1
    return temp  # five passes: the result sits in `temp`, which the reference swaps into v


# ---- the accel ------------------------------------------------------------------------------------------
class Node:
    __slots__ = ("bounds", "leaf", "a", "b", "c0", "c1")


class Accel:
    """One BVHAccel: primitives in insertion order (nested accels among them), the flattened nodes, order[]."""

    def __init__(self, prims, transform_m, transform_minv, material, swap):
        self.prims, self.m, self.minv, self.material, self.swap = prims, transform_m, transform_minv, material, swap
        n = len(prims)
        self.max_prims = min(n, 255)
        self.order = [None] * n
        info = []
        for i, p in enumerate(prims):
            b = p["bound"]
            info.append((i, b, tuple(0.5 * b[0][k] + 0.5 * b[1][k] for k in range(3))))
        root = self._build(info)
        self.nodes = []
        self._flatten(root)

    # -- build (bvh.rs:203-270)
    def _build(self, info):
        bounds = b_none()
        for _, b, _c in info:
            bounds = b_union(bounds, b)
        morton = []
        for i, _b, c in info:
            off = b_offset(bounds, c)
            morton.append((i, encode_morton_3((off[0] * 1024.0, off[1] * 1024.0, off[2] * 1024.0))))
        morton = radix_sort(morton)
        treelets = []
        start = 0
        self._ordered = 0
        mask = 0b00111111111111000000000000000000
        for end in range(1, len(morton) + 1):
            if end == len(morton) or (morton[start][1] & mask) != (morton[end][1] & mask):
                treelets.append(self._emit_lbvh(morton, start, end - start, info, 29 - 12))
                start = end
        return self._upper_sah(treelets)

    def _emit_lbvh(self, mp, base, nprims, info, bit):  # bvh.rs:275-345 (mp[base:] is the reference's slice)
        if bit == -1 or nprims < self.max_prims:
            first = self._ordered
            self._ordered += nprims
            bounds = b_none()
            for i in range(nprims):
                idx = mp[base + i][0]
                self.order[first + i] = idx
                bounds = b_union(bounds, info[idx][1])
            n = Node()
            n.bounds, n.leaf, n.a, n.b, n.c0, n.c1 = bounds, True, first, nprims, None, None
            return n
        mask = 1 << bit
        if (mp[base][1] & mask) == (mp[base + nprims - 1][1] & mask):
            return self._emit_lbvh(mp, base, nprims, info, bit - 1)
        lo, hi = 0, nprims - 1
        while lo + 1 != hi:
            mid = (lo + hi) // 2
            if (mp[base + lo][1] & mask) == (mp[base + mid][1] & mask):
                lo = mid
            else:
                hi = mid
        split = hi
        c0 = self._emit_lbvh(mp, base, split, info, bit - 1)
        c1 = self._emit_lbvh(mp, base + split, nprims - split, info, bit - 1)
        n = Node()
        n.bounds, n.leaf, n.a, n.b, n.c0, n.c1 = b_union(c0.bounds, c1.bounds), False, bit % 3, 0, c0, c1
        return n

    def _upper_sah(self, roots):  # bvh.rs:348-432
        if len(roots) == 1:
            return roots[0]
        bounds, cb = b_none(), b_none()
        for r in roots:
            bounds = b_union(bounds, r.bounds)
        for r in roots:
            cb = b_point_union(cb, tuple(0.5 * (r.bounds[0][k] + r.bounds[1][k]) for k in range(3)))
        dim = b_maximum_extent(cb)

        def bucket_a(r):  # bvh.rs:377-381 (`(min + max) * 0.5`)
            centroid = (r.bounds[0][dim] + r.bounds[1][dim]) * 0.5
            return bucket_of(centroid)

        def bucket_b(r):  # bvh.rs:415-420 (`0.5 * (min + max)`: the same product)
            centroid = 0.5 * (r.bounds[0][dim] + r.bounds[1][dim])
            return bucket_of(centroid)

        def bucket_of(centroid):
            b0 = _div(centroid - cb[0][dim], cb[1][dim] - cb[0][dim])
            b = as_u32(float(NBUCKETS) * b0)
            return NBUCKETS - 1 if b == NBUCKETS else b

        counts = [0] * NBUCKETS
        bb = [b_none() for _ in range(NBUCKETS)]
        for r in roots:
            b = bucket_a(r)
            if b >= NBUCKETS:
                raise IndexError("bucket out of range: the reference panics here")
            counts[b] += 1
            bb[b] = b_union(bb[b], r.bounds)
        cost = [0.0] * NBUCKETS
        for i in range(NBUCKETS):
            b0, c0 = b_none(), 0
            for j in range(0, i + 1):
                b0, c0 = b_union(b0, bb[j]), c0 + counts[j]
            b1, c1 = b_none(), 0
            for j in range(i + 1, NBUCKETS):
                b1, c1 = b_union(b1, bb[j]), c1 + counts[j]
            cost[i] = 0.125 + _div(float(c0) * b_surface_area(b0) + float(c1) * b_surface_area(b1), b_surface_area(bounds))
        split = 0
        for i, c in enumerate(cost):
            if c < cost[split]:
                split = i
        # partition ^0.1: in place, two pointers, unstable
        data = list(roots)
        pred = lambda r: bucket_b(r) <= split
        n = len(data)
        l, r = 0, n - 1
        while True:
            while l < n and pred(data[l]):
                l += 1
            while r > 0 and not pred(data[r]):
                r -= 1
            if l >= r:
                break
            data[l], data[r] = data[r], data[l]
        if l == 0 or l == n:
            raise RecursionError("degenerate SAH split: the reference recurses forever here")
        node = Node()
        c0, c1 = self._upper_sah(data[:l]), self._upper_sah(data[l:])
        node.bounds, node.leaf, node.a, node.b, node.c0, node.c1 = b_union(c0.bounds, c1.bounds), False, dim, 0, c0, c1
        return node

    def _flatten(self, node):  # bvh.rs:435-453
        me = len(self.nodes)
        self.nodes.append([node.bounds, node.leaf, node.a, (node.b & 0xFFFF) if node.leaf else node.b])  # Leaf(prim_offset as u32, nprims as u16)
        if not node.leaf:
            self._flatten(node.c0)
            self.nodes[me][3] = self._flatten(node.c1)
        return me

    def bound(self):  # BVHAccel::bound, bvh.rs:457-459
        return b_transform(self.m, self.nodes[0][0])

    # -- intersect (bvh.rs:461-522) -> isect dict in the parent's space, or None
    def intersect(self, o, d, best_t):
        o_l, d_l = transform_point(self.minv, o), transform_vector(self.minv, d)
        dinv = (_div(1.0, d_l[0]), _div(1.0, d_l[1]), _div(1.0, d_l[2]))
        neg_dir = (dinv[0] < 0.0, dinv[1] < 0.0, dinv[2] < 0.0)
        hit = None
        stack = []
        cur = 0
        while True:
            bounds, leaf, a, b = self.nodes[cur]
            if not slab_intersects(bounds, o_l, dinv):
                if not stack:
                    break
                cur = stack.pop()
                continue
            if leaf:
                for i in range(b):
                    r = self.prims[self.order[a + i]]["isect"](o_l, d_l, dinv, best_t)
                    if r is not None:
                        hit = r
                        best_t = r["t"]
                if not stack:
                    break
                cur = stack.pop()
            else:
                if len(stack) >= 64:
                    raise IndexError("more than 64 pending nodes: the reference panics here")
                if neg_dir[a]:
                    stack.append(cur + 1)
                    cur = b
                else:
                    stack.append(b)
                    cur += 1
        if hit is None:
            return None
        g = (transform_vector(self.m, hit["g"][0]), transform_vector(self.m, hit["g"][1]))
        sfc = (transform_vector(self.m, hit["s"][0]), transform_vector(self.m, hit["s"][1])) if hit["g"] != hit["s"] else g
        n = None if hit["n"] is None else transform_normal(self.minv, hit["n"])
        mat = hit["mat"]
        if self.material is not None:
            mat = self.material
        if self.swap:
            g, sfc = (g[1], g[0]), (sfc[1], sfc[0])
            n = None if n is None else neg(n)
        return {"t": hit["t"], "g": g, "s": sfc, "n": n, "mat": mat, "own": hit["own"]}

    def dump(self, f, i):  # the layout of the oracle's orc_accel_dump (oracle/lasgun_oracle.cpp, dump_accel)
        i += [len(self.nodes), len(self.order), 1 if self.material is not None else 0, 1 if self.swap else 0]
        for bounds, leaf, a, b in self.nodes:
            f += list(bounds[0]) + list(bounds[1])
            i += [1 if leaf else 0, a, b]
        i += list(self.order)
        for mat in (self.m, self.minv):
            for c in range(4):
                f += [mat[c][r] for r in range(4)]
        for p in self.prims:
            if p.get("accel") is not None:
                p["accel"].dump(f, i)


def _sphere_prim(cen, rad, mat):
    def isect(o, d, dinv, best_t):
        t, inside = pyref.sphere_t(o, d, cen, rad)
        if t < 0.0 or t >= best_t:
            return None
        dpdu, dpdv = pyref.sphere_isect(o, d, cen, rad, t, inside)
        return {"t": t, "g": (dpdu, dpdv), "s": (dpdu, dpdv), "n": None, "mat": pyref.DEFAULT_MATERIAL, "own": mat}
    r3 = (rad, rad, rad)
    return {"bound": b_new(sub(cen, r3), add(cen, r3)), "isect": isect}


def _cuboid_prim(mn, mx, mat):
    def isect(o, d, dinv, best_t):
        r = pyref.cuboid_isect(o, d, dinv, mn, mx, best_t)
        if r is None:
            return None
        t, dp0, dp1, n = r
        return {"t": t, "g": (dp0, dp1), "s": (dp0, dp1), "n": n, "mat": pyref.DEFAULT_MATERIAL, "own": mat}
    return {"bound": b_new(mn, mx), "isect": isect}


def _triangle_prim(obj, poly):
    def isect(o, d, dinv, best_t):
        r = pyref.triangle_isect(obj, poly, o, d, best_t)
        if r is None:
            return None
        t, g, sfc, n = r
        return {"t": t, "g": g, "s": sfc, "n": n, "mat": pyref.DEFAULT_MATERIAL, "own": None}
    p0, p1, p2 = obj.position[poly[0][0]], obj.position[poly[1][0]], obj.position[poly[2][0]]
    return {"bound": b_point_union(b_new(p0, p1), p2), "isect": isect}


def _accel_prim(acc):
    return {"bound": acc.bound(), "isect": lambda o, d, dinv, best_t: acc.intersect(o, d, best_t), "accel": acc}


def build(agg):
    """BVHAccel::from_aggregate (bvh.rs:149-162) over a pyref.Aggregate."""
    prims = []
    for node in agg.contents:
        if node[0] == "sphere":
            prims.append(_sphere_prim(node[1], node[2], node[3]))
        elif node[0] == "cuboid":
            prims.append(_cuboid_prim(node[1], node[2], node[3]))
        elif node[0] == "mesh":  # from_mesh (bvh.rs:141-147): identity transform, the mesh's material as the accel's default
            ident = pyref.mat_identity()
            tris = [_triangle_prim(node[1], poly) for poly in node[1].polys]
            prims.append(_accel_prim(Accel(tris, ident, pyref.mat_identity(), node[2], False)))
        else:
            prims.append(_accel_prim(build(node[1])))
    return Accel(prims, agg.transform.m, agg.transform.minv, None, agg.swap)


def install(scene):
    """Make pyref.li() find its hits through the BVH of `scene` instead of by brute force."""
    root = build(scene.root)

    def closest(scn, o, d):
        r = root.intersect(o, d, INF)
        if r is None:
            return None
        # li(): shape.material().unwrap_or(isect.material) (integrate.rs:29): spheres and cuboids carry their own
        r = dict(r)
        r["mat"] = r["own"] if r["own"] is not None else r["mat"]
        return r
    scene._closest = closest
    return root
